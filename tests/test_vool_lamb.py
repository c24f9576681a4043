"""SemAbsVOOL forward + Lamb.step: oracle vs the reference's golden vectors on CPU, HIP vs golden on the GPU."""
import numpy as np
import pytest
import torch

import semabs_amd  # noqa: F401
from oracle import vool as ov
from semabs_amd.weights import make_semabsvool_state_dict

SCENE_BOUNDS = [[-1.0, -1.0, -0.1], [1.0, 1.0, 1.9]]
REL = [["behind"], ["on"], ["in front of"]]
SHAPES = [(64, 33), (7,), (128, 128), (5, 3, 3, 3, 3), (1,)]


def _inputs(S, N, M, P, seed):
    rng = np.random.default_rng(seed)
    lo, hi = np.array(SCENE_BOUNDS[0]), np.array(SCENE_BOUNDS[1])
    xyz = (lo + (hi - lo) * rng.random((1, N, 3))).astype(np.float32)
    xyz[0, : N // 8] = xyz[0, N // 8: 2 * (N // 8)] + np.float32(1e-3)
    feat = (rng.standard_normal((1, P, N, 1)) * 0.5).astype(np.float32)
    q = (lo - 0.05 + (hi - lo + 0.1) * rng.random((1, P, M, 3))).astype(np.float32)
    return xyz, feat, q


def _lamb_stream():
    rng = np.random.default_rng(77)
    ws = [(rng.standard_normal(s) * (0.0 if i == 1 else 0.3)).astype(np.float32) for i, s in enumerate(SHAPES)]
    grads = [[(rng.standard_normal(s) * 0.1).astype(np.float32) for s in SHAPES] for _ in range(3)]
    return ws, grads


def test_oracle_vool_forward(golden):
    g = golden("g12_vool_lamb")
    S, N, M, D, seed, wseed = (int(v) for v in g["vool_meta"])
    sd = make_semabsvool_state_dict(seed=wseed)
    xyz, feat, q = _inputs(S, N, M, 2 * D, seed)
    with torch.no_grad():
        out = ov.vool_forward(sd, torch.from_numpy(xyz), torch.from_numpy(feat[:, :D]), torch.from_numpy(feat[:, D:]), torch.from_numpy(q[:, :D]),
                              REL, SCENE_BOUNDS, (S, S, S))
    np.testing.assert_allclose(out.numpy(), g["vool_out"], rtol=2e-3, atol=2e-3)      # logits are cos / 0.07, |.| <= 14.3


def test_oracle_lamb(golden):
    g = golden("g12_vool_lamb")
    ws, grads = _lamb_stream()
    ms = [np.zeros_like(w) for w in ws]; vs = [np.zeros_like(w) for w in ws]
    stats = [None] * len(ws)
    for step in range(3):
        for i in range(len(ws)):
            ws[i], ms[i], vs[i], stats[i] = ov.lamb_step(ws[i], grads[step][i], ms[i], vs[i], lr=1e-3, weight_decay=1e-5)
    for i in range(len(ws)):
        np.testing.assert_allclose(ws[i], g[f"lamb_w{i}"], rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(ms[i], g[f"lamb_m{i}"], rtol=2e-6, atol=2e-8)
        np.testing.assert_allclose(vs[i], g[f"lamb_v{i}"], rtol=2e-6, atol=1e-10)
        np.testing.assert_allclose(np.asarray(stats[i], np.float32), g[f"lamb_stats{i}"], rtol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("precision,tol", [("exact", 3e-3), ("fp16", 6e-2)])
def test_gpu_vool_forward(golden, precision, tol):
    from semabs_amd.net import SemAbsVOOL
    g = golden("g12_vool_lamb")
    S, N, M, D, seed, wseed = (int(v) for v in g["vool_meta"])
    m = SemAbsVOOL(pointing_method="cosine_sim", pointing_dim=64, device="cuda", decoder_concat_xyz_pts=True, voxel_shape=(S, S, S),
                   scene_bounds=SCENE_BOUNDS, unet_num_channels=16, unet_f_maps=16, unet_num_groups=8, unet_num_levels=6,
                   network_inputs=["saliency"], use_pts_feat_extractor=True, pts_feat_extractor_hidden_dim=128, reduce_method="max",
                   batch_size=1, precision=precision)
    m.load_state_dict(make_semabsvool_state_dict(seed=wseed))
    m.eval()                                                # like visualize.py:453; in training mode the forward records the backward tape instead
    xyz, feat, q = _inputs(S, N, M, 2 * D, seed)
    out = m(output_xyz_pts=torch.from_numpy(q[:, :D]), spatial_relation_name=REL, input_xyz_pts=torch.from_numpy(xyz),
            input_target_saliency_pts=torch.from_numpy(feat[:, :D]), input_reference_saliency_pts=torch.from_numpy(feat[:, D:]), tsdf_vol=None)
    assert out.grad_fn is None
    err = np.abs(out.detach().cpu().numpy() - g["vool_out"]).max()
    print(f"{precision}: VOOL logit Linf {err:.3e} (max|ref| {np.abs(g['vool_out']).max():.2f})")
    assert err <= tol * np.abs(g["vool_out"]).max()


@pytest.mark.gpu
def test_gpu_lamb(golden):
    from semabs_amd.optim import Lamb
    g = golden("g12_vool_lamb")
    ws, grads = _lamb_stream()
    params = [torch.nn.Parameter(torch.from_numpy(w).cuda()) for w in ws]
    opt = Lamb(params, lr=1e-3, weight_decay=1e-5)
    for step in range(3):
        for i, p in enumerate(params):
            if p.grad is None:
                p.grad = torch.from_numpy(grads[step][i]).cuda()
            else:
                p.grad.copy_(torch.from_numpy(grads[step][i]))
        opt.step()
    for i, p in enumerate(params):
        st = opt.state[p]
        np.testing.assert_allclose(p.detach().cpu().numpy(), g[f"lamb_w{i}"], rtol=3e-6, atol=1e-7)
        np.testing.assert_allclose(st["exp_avg"].cpu().numpy(), g[f"lamb_m{i}"], rtol=3e-6, atol=2e-8)
        np.testing.assert_allclose(st["exp_avg_sq"].cpu().numpy(), g[f"lamb_v{i}"], rtol=3e-6, atol=1e-10)
        got = np.asarray([float(st["weight_norm"]), float(st["adam_norm"]), float(st["trust_ratio"])], np.float32)
        np.testing.assert_allclose(got, g[f"lamb_stats{i}"], rtol=1e-5)
        assert st["step"] == 3
    with pytest.raises(ValueError):
        Lamb(params, lr=-1.0)


@pytest.mark.gpu
def test_gpu_lamb_resume_from_state_dict(golden):
    """The reference's resume flow (utils.py:289 `optimizer.load_state_dict(checkpoint["optimizer"])`) between steps: after one step the state
    dict is saved (to the host, like torch.save / torch.load would), loaded into a FRESH optimizer over new parameter tensors AND back into
    the running one (whose launch plan caches the old moment buffers' addresses); both must continue exactly like the uninterrupted run
    (the golden's three steps).  Guards the cached-pointer bug: the loaded moments must be the ones the kernels read and write."""
    import copy
    from semabs_amd.optim import Lamb
    g = golden("g12_vool_lamb")
    ws, grads = _lamb_stream()

    def set_grads(params, step):
        for i, p in enumerate(params):
            p.grad = torch.from_numpy(grads[step][i]).cuda()

    params = [torch.nn.Parameter(torch.from_numpy(w).cuda()) for w in ws]
    opt = Lamb(params, lr=1e-3, weight_decay=1e-5)
    set_grads(params, 0)
    opt.step()
    ckpt = copy.deepcopy({"optimizer": opt.state_dict(), "w": [p.detach().cpu().clone() for p in params]})
    ckpt["optimizer"]["state"] = {k: {kk: (vv.cpu() if torch.is_tensor(vv) else vv) for kk, vv in st.items()} for k, st in ckpt["optimizer"]["state"].items()}
    # (a) a fresh optimizer over fresh tensors
    params2 = [torch.nn.Parameter(w.clone().cuda()) for w in ckpt["w"]]
    opt2 = Lamb(params2, lr=1e-3, weight_decay=1e-5)
    opt2.load_state_dict(ckpt["optimizer"])
    # (b) the running optimizer: poison its current moments first - if the stale plan were used, the poison would show up in the result
    for p in params:
        opt.state[p]["exp_avg"].fill_(123.0); opt.state[p]["exp_avg_sq"].fill_(456.0)
    opt.load_state_dict(ckpt["optimizer"])
    for o, ps in ((opt, params), (opt2, params2)):
        for step in (1, 2):
            set_grads(ps, step)
            o.step()
        for i, p in enumerate(ps):
            st = o.state[p]
            np.testing.assert_allclose(p.detach().cpu().numpy(), g[f"lamb_w{i}"], rtol=3e-6, atol=1e-7)
            np.testing.assert_allclose(st["exp_avg"].cpu().numpy(), g[f"lamb_m{i}"], rtol=3e-6, atol=2e-8)
            np.testing.assert_allclose(st["exp_avg_sq"].cpu().numpy(), g[f"lamb_v{i}"], rtol=3e-6, atol=1e-10)
            assert st["step"] == 3
