"""GPU parity for the OVSSC voxel-inference half: HIP kernels (C ABI) vs plain torch fp32 references of the same
ops, then SemAbs3D end to end vs the oracle and the golden vectors captured from the reference.
fp16 mode: fp16 activations / operands, fp32 accumulate (tolerances stated per test); exact mode: fp32 activations with
hi/lo-split operands, which must match the fp32 reference to ~1e-5."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import semabs_amd  # noqa: F401
from oracle import semabs3d as os3
from semabs_amd.weights import make_semabs3d_state_dict

pytestmark = pytest.mark.gpu
SCENE_BOUNDS = [[-1.0, -1.0, -0.1], [1.0, 1.0, 1.9]]


def _cl(x):      # NCDHW -> channels-last
    return x.permute(0, 2, 3, 4, 1).contiguous()


def _unet(precision, sd=None, levels=6):
    from semabs_amd.unet3d import ResidualUNet3D
    u = ResidualUNet3D(16, 16, f_maps=16, num_groups=8, num_levels=levels, precision=precision)
    u.load_state_dict(sd if sd is not None else make_semabs3d_state_dict(seed=3, unet_num_levels=levels), prefix="vol_feature_extractor.")
    return u


@pytest.mark.parametrize("precision,tol", [("exact", 2e-5), ("fp16", 4e-3)])
@pytest.mark.parametrize("cin,cout,S,B", [(16, 16, 12, 2), (16, 32, 8, 1), (32, 32, 8, 3), (64, 128, 4, 2), (512, 512, 2, 2), (32, 64, 6, 1)])
def test_conv3d_gn_relu_resid(precision, tol, cin, cout, S, B, bar):
    """GroupNorm -> Conv3d 3^3 -> (+residual) -> ReLU against torch fp32."""
    from semabs_amd import _lib
    from semabs_amd.unet3d import _Conv
    rng = np.random.default_rng(cin + cout)
    x = torch.from_numpy(rng.standard_normal((B, cin, S, S + 1, S + 2)).astype(np.float32)) * 2 + 0.5
    w = torch.from_numpy((rng.standard_normal((cout, cin, 3, 3, 3)) / np.sqrt(27 * cin)).astype(np.float32))
    gw = torch.from_numpy((1 + 0.2 * rng.standard_normal(cin)).astype(np.float32))
    gb = torch.from_numpy((0.2 * rng.standard_normal(cin)).astype(np.float32))
    res = torch.from_numpy(rng.standard_normal((B, cout, S, S + 1, S + 2)).astype(np.float32))
    G = 8 if cin >= 8 else 1
    ref = F.relu(F.conv3d(F.group_norm(x, G, gw, gb, 1e-5), w, None, padding=1) + res)
    u = _unet(precision)
    conv = _Conv(w, gw, gb, None, 8, u.dev)
    xd, rd = _cl(x).cuda().to(u.act_dtype), _cl(res).cuda().to(u.act_dtype)
    if precision == "fp16":       # compare against the reference evaluated on the same rounded inputs
        ref = F.relu(F.conv3d(F.group_norm(xd.float().cpu().permute(0, 4, 1, 2, 3), G, gw, gb, 1e-5), w, None, padding=1)
                     + rd.float().cpu().permute(0, 4, 1, 2, 3))
    y = u._conv(xd, conv, relu=True, resid=rd)
    got = y.float().cpu().permute(0, 4, 1, 2, 3)
    # (fp16 mode: the error of a single layer is one or two fp16 ulps of the output and moves by an ulp from run to run - the GroupNorm statistics are
    #  accumulated with atomics - so its bar stays at the fixed value; the exact mode's bars are per case, conftest.bar)
    assert bar("vs_torch", (got - ref).abs().max().item(), tol * max(1.0, ref.abs().max().item()), floor=None if precision == "exact" else tol * max(1.0, ref.abs().max().item()))


@pytest.mark.parametrize("precision,tol", [("exact", 2e-5), ("fp16", 4e-3)])
@pytest.mark.parametrize("dims,B,resid", [((8, 8, 16), 1, False), ((16, 24, 32), 2, True), ((8, 16, 48), 3, True)])
def test_conv16_lds_brick_kernel(precision, tol, dims, B, resid, bar):
    """The level-0 LDS-halo kernel (Cin = Cout = 16, bricks of 8 x 8 x 16) vs torch fp32 and vs the generic gather kernel."""
    from semabs_amd import _lib
    from semabs_amd.unet3d import _Conv
    rng = np.random.default_rng(dims[2] + B)
    x = torch.from_numpy(rng.standard_normal((B, 16, *dims)).astype(np.float32)) * 1.5 + 0.3
    x[:, :, ::3] = 0                                                  # sparse like a scattered volume
    w = torch.from_numpy((rng.standard_normal((16, 16, 3, 3, 3)) / np.sqrt(27 * 16)).astype(np.float32))
    gw = torch.from_numpy((1 + 0.2 * rng.standard_normal(16)).astype(np.float32))
    gb = torch.from_numpy((0.2 * rng.standard_normal(16)).astype(np.float32))
    res = torch.from_numpy(rng.standard_normal((B, 16, *dims)).astype(np.float32))
    u = _unet(precision)
    conv = _Conv(w, gw, gb, None, 8, u.dev)
    xd, rd = _cl(x).cuda().to(u.act_dtype), _cl(res).cuda().to(u.act_dtype)
    ref = F.conv3d(F.group_norm(xd.float().cpu().permute(0, 4, 1, 2, 3), 8, gw, gb, 1e-5), w, None, padding=1)
    if resid:
        ref = ref + rd.float().cpu().permute(0, 4, 1, 2, 3)
    ref = F.relu(ref)
    y_lds = u._conv(xd, conv, relu=True, resid=rd if resid else None).float().cpu().permute(0, 4, 1, 2, 3)
    y_gen = u._conv(xd, conv, relu=True, resid=rd if resid else None, generic=True).float().cpu().permute(0, 4, 1, 2, 3)
    assert bar("vs_torch", (y_lds - ref).abs().max().item(), tol * max(1.0, ref.abs().max().item()), floor=None if precision == "exact" else tol * max(1.0, ref.abs().max().item()))
    assert bar("vs_generic", (y_lds - y_gen).abs().max().item(), (1e-5 if precision == "exact" else 2e-3) * max(1.0, ref.abs().max().item()), floor=None if precision == "exact" else (1e-5 if precision == "exact" else 2e-3) * max(1.0, ref.abs().max().item()))


@pytest.mark.parametrize("precision,tol", [("exact", 2e-5), ("fp16", 4e-3)])
@pytest.mark.parametrize("cin,cout,dims,B,resid", [(16, 32, (4, 8, 16), 2, False), (32, 32, (8, 16, 32), 3, True), (64, 64, (4, 16, 16), 2, True),
                                                   (32, 64, (8, 16, 32), 64, True), (64, 64, (8, 16, 32), 64, False), (16, 64, (8, 16, 32), 64, True),
                                                   (128, 128, (4, 8, 16), 1, True), (256, 256, (8, 8, 8), 2, True), (128, 256, (4, 16, 8), 3, False),
                                                   (32, 32, (4, 8, 24), 1, True), (32, 16, (8, 16, 32), 3, True), (64, 16, (4, 8, 16), 2, False)])
def test_conv_brick_kernel(precision, tol, cin, cout, dims, B, resid, bar):
    """k_conv_brick (64^3 .. 16^3 levels: LDS halo bricks of 4 x 8 x 16, weights software-pipelined, lane-transposed epilogue) vs torch fp32 and
    vs the generic gather kernel; the B = 64 cases have enough bricks for the four-output-block (in-place weight re-request) variants, the Cout = 16
    cases take the one-output-block variant (the data gradient of the network's 16 -> 32 convolution)."""
    from semabs_amd.unet3d import _Conv
    rng = np.random.default_rng(cin * 7 + cout + B)
    x = torch.from_numpy(rng.standard_normal((B, cin, *dims)).astype(np.float32)) * 1.5 + 0.3
    w = torch.from_numpy((rng.standard_normal((cout, cin, 3, 3, 3)) / np.sqrt(27 * cin)).astype(np.float32))
    gw = torch.from_numpy((1 + 0.2 * rng.standard_normal(cin)).astype(np.float32))
    gb = torch.from_numpy((0.2 * rng.standard_normal(cin)).astype(np.float32))
    res = torch.from_numpy(rng.standard_normal((B, cout, *dims)).astype(np.float32))
    u = _unet(precision)
    conv = _Conv(w, gw, gb, None, 8, u.dev)
    xd, rd = _cl(x).cuda().to(u.act_dtype), _cl(res).cuda().to(u.act_dtype)
    ref = F.conv3d(F.group_norm(xd.float().cpu().permute(0, 4, 1, 2, 3), 8, gw, gb, 1e-5), w, None, padding=1)
    if resid:
        ref = ref + rd.float().cpu().permute(0, 4, 1, 2, 3)
    ref = F.relu(ref)
    y_brick = u._conv(xd, conv, relu=True, resid=rd if resid else None).float().cpu().permute(0, 4, 1, 2, 3)
    y_gen = u._conv(xd, conv, relu=True, resid=rd if resid else None, generic=True).float().cpu().permute(0, 4, 1, 2, 3)
    scale = max(1.0, ref.abs().max().item())
    assert bar("vs_torch", (y_brick - ref).abs().max().item(), tol * scale, floor=None if precision == "exact" else tol * scale)
    assert bar("vs_generic", (y_brick - y_gen).abs().max().item(), (1e-5 if precision == "exact" else 2e-3) * scale, floor=None if precision == "exact" else (1e-5 if precision == "exact" else 2e-3) * scale)


@pytest.mark.parametrize("precision", ["exact", "fp16"])
@pytest.mark.parametrize("cin,cout,dims", [(32, 32, (8, 16, 32)), (64, 64, (4, 8, 8)), (16, 32, (4, 8, 16)), (128, 64, (3, 5, 6)), (16, 16, (8, 8, 16))])
def test_packed_weights_equal_rowmajor(precision, cin, cout, dims):
    """SEMABS_CONV_PACKED (flag bit 9: the kernels read the fragment-packed copy behind the [Cout, Kp] matrix) only changes WHERE a weight fragment
    comes from: brick, level-0 and gather kernels give bit-identical outputs with and without it, and so does the transposed convolution."""
    from semabs_amd.unet3d import _Conv, _ConvT
    rng = np.random.default_rng(cin + 3 * cout)
    u = _unet(precision)
    x = _cl(torch.from_numpy(rng.standard_normal((2, cin, *dims)).astype(np.float32))).cuda().to(u.act_dtype)
    w = torch.from_numpy((rng.standard_normal((cout, cin, 3, 3, 3)) / np.sqrt(27 * cin)).astype(np.float32))
    conv = _Conv(w, torch.ones(cin), torch.zeros(cin), None, 8, u.dev)
    assert conv.packed == 512
    # (no GroupNorm in front: its statistics are reduced with floating-point atomics, so two calls may differ by an ulp of the affine)
    y1 = u._conv(x, conv, relu=True, gn=False)
    y1g = u._conv(x, conv, relu=True, gn=False, generic=True)
    conv.packed = 0
    y0 = u._conv(x, conv, relu=True, gn=False)
    y0g = u._conv(x, conv, relu=True, gn=False, generic=True)
    scale = float(y0.float().abs().max())
    # (bit-equal where the kernel is deterministic; the gather kernel's split-K partial sums are added with fp32 atomics)
    assert float((y1.float() - y0.float()).abs().max()) <= 2e-6 * scale and float((y1g.float() - y0g.float()).abs().max()) <= 2e-6 * scale
    if cin % 32 == 0 and cout % 16 == 0:
        wt = torch.from_numpy((rng.standard_normal((cin, cout, 3, 3, 3)) / np.sqrt(27 * cin / 8)).astype(np.float32))
        ct = _ConvT(wt, torch.zeros(cout), u.dev)
        skip = torch.zeros(2, 2 * dims[0], 2 * dims[1], 2 * dims[2], cout, device="cuda", dtype=u.act_dtype)
        z1 = u._up(x, skip, ct)
        assert ct.packed == 512
        ct.packed = 0
        z0 = u._up(x, skip, ct)
        assert float((z1.float() - z0.float()).abs().max()) <= 2e-6 * max(1.0, float(z0.float().abs().max()))


@pytest.mark.parametrize("precision,tol", [("exact", 2e-5), ("fp16", 4e-3)])
@pytest.mark.parametrize("cin,cout,S,B", [(32, 16, 6, 2), (64, 32, 4, 1), (512, 256, 2, 2), (128, 64, 3, 1)])
def test_convtranspose3d_skip(precision, tol, cin, cout, S, B, bar):
    from semabs_amd.unet3d import _ConvT
    rng = np.random.default_rng(cin)
    x = torch.from_numpy(rng.standard_normal((B, cin, S, S + 1, S + 2)).astype(np.float32))
    w = torch.from_numpy((rng.standard_normal((cin, cout, 3, 3, 3)) / np.sqrt(27 * cin / 8)).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal(cout).astype(np.float32))
    skip = torch.from_numpy(rng.standard_normal((B, cout, 2 * S, 2 * S + 2, 2 * S + 4)).astype(np.float32))
    u = _unet(precision)
    xd, sd_ = _cl(x).cuda().to(u.act_dtype), _cl(skip).cuda().to(u.act_dtype)
    ref = skip_r = sd_.float().cpu().permute(0, 4, 1, 2, 3) + F.conv_transpose3d(xd.float().cpu().permute(0, 4, 1, 2, 3), w, b, stride=2,
                                                                                 padding=1, output_padding=1)
    y = u._up(xd, sd_, _ConvT(w, b, u.dev))
    got = y.float().cpu().permute(0, 4, 1, 2, 3)
    assert bar("vs_torch", (got - ref).abs().max().item(), tol * max(1.0, ref.abs().max().item()), floor=None if precision == "exact" else tol * max(1.0, ref.abs().max().item()))


@pytest.mark.parametrize("precision", ["exact", "fp16"])
def test_maxpool_and_1x1(precision):
    from semabs_amd.unet3d import _Conv
    rng = np.random.default_rng(1)
    u = _unet(precision)
    x = torch.from_numpy(rng.standard_normal((2, 32, 6, 4, 8)).astype(np.float32))
    xd = _cl(x).cuda().to(u.act_dtype)
    got = u._pool(xd).float().cpu().permute(0, 4, 1, 2, 3)
    assert torch.equal(got, F.max_pool3d(xd.float().cpu().permute(0, 4, 1, 2, 3), 2))
    x = torch.from_numpy(rng.standard_normal((2, 16, 5, 4, 3)).astype(np.float32))
    w = torch.from_numpy(rng.standard_normal((16, 16, 1, 1, 1)).astype(np.float32) * 0.3)
    b = torch.from_numpy(rng.standard_normal(16).astype(np.float32))
    xd = _cl(x).cuda().to(u.act_dtype)
    y = u._conv(xd, _Conv(w, None, None, b, 8, u.dev), relu=False, gn=False).float().cpu().permute(0, 4, 1, 2, 3)
    ref = F.conv3d(xd.float().cpu().permute(0, 4, 1, 2, 3), w, b)
    assert (y - ref).abs().max().item() <= (2e-5 if precision == "exact" else 3e-3) * ref.abs().max().item()


def semabs_inputs(S, N, M, P, seed):
    rng = np.random.default_rng(seed)
    lo, hi = np.array(SCENE_BOUNDS[0]), np.array(SCENE_BOUNDS[1])
    xyz = (lo + (hi - lo) * rng.random((1, N, 3))).astype(np.float32)
    xyz[0, : N // 8] = xyz[0, N // 8: 2 * (N // 8)] + np.float32(1e-3)
    feat = (rng.standard_normal((1, P, N, 1)) * 0.5).astype(np.float32)
    q = (lo - 0.05 + (hi - lo + 0.1) * rng.random((1, P, M, 3))).astype(np.float32)
    return xyz, feat, q


def _model(S, precision, stats="init"):
    from semabs_amd.net import SemAbs3D
    m = SemAbs3D(voxel_shape=(S, S, S), scene_bounds=SCENE_BOUNDS, unet_num_channels=16, unet_f_maps=16, unet_num_groups=8,
                 unet_num_levels=6, network_inputs=["saliency"], use_pts_feat_extractor=True, pts_feat_extractor_hidden_dim=128,
                 reduce_method="max", output_dim=1, device="cuda", decoder_concat_xyz_pts=True, batch_size=1, precision=precision)
    m.load_state_dict(make_semabs3d_state_dict(seed=3, stats=stats))
    return m


@pytest.mark.parametrize("stats", ["init", "trained"])
@pytest.mark.parametrize("precision,tol_feat,tol_out", [("exact", 2e-4, 2e-4), ("fp16", 2e-2, 1e-2)])
def test_semabs3d_forward_vs_golden(golden, precision, tol_feat, tol_out, bar, stats):
    """SemAbs3D.forward at 32^3 (point MLP -> scatter-mean -> 6-level UNet -> decoder) vs the reference's outputs.  stats = "trained" (golden g30): GroupNorm
    gains spread over 0.02 - 6, offsets of +-0.5, dominant convolution channels (weights.make_semabs3d_state_dict) - the same relative bars."""
    g = golden("g9_semabs3d" if stats == "init" else "g30_semabs3d_trained")
    S, N, M, P, seed, wseed = (int(v) for v in g["meta"])
    m = _model(S, precision, stats)
    xyz, feat, q = semabs_inputs(S, N, M, P, seed)
    taps = {}
    f = m.feature_volume(torch.from_numpy(xyz[0]).cuda(), torch.from_numpy(feat[0, :, :, 0]).cuda(), taps=taps)
    sc = taps["scatter"].float().cpu().permute(0, 4, 1, 2, 3)
    assert (sc[:, 0] != 0).sum().item() == g["scatter_nonzero"]              # same occupied voxels
    np.testing.assert_allclose(sc.numpy()[:, :, ::3, ::3, ::3], g["scatter_sub"], rtol=2e-3 if precision == "fp16" else 1e-5,
                               atol=2e-3 if precision == "fp16" else 1e-5)
    feats = f.float().cpu().permute(0, 4, 1, 2, 3).numpy()
    ref = g["unet_sub"]
    err = np.abs(feats[:, :, ::3, ::3, ::3] - ref).max()
    print(f"{precision}: UNet feature Linf {err:.3e} (max|ref| {np.abs(ref).max():.3f})")
    assert bar("unet_feature", err, tol_feat * np.abs(ref).max())
    out = m.forward(torch.from_numpy(xyz), torch.from_numpy(feat), None, torch.from_numpy(q)).cpu().numpy()
    err = np.abs(out - g["out"]).max()
    print(f"{precision}: logit Linf {err:.3e} (max|ref| {np.abs(g['out']).max():.3f})")
    assert bar("logits", err, tol_out * max(1.0, np.abs(g["out"]).max()))
    vvf = m.visual_volumetric_features
    assert tuple(vvf.shape) == (P, 16, S, S, S)


def test_scatter_mean_bit_exact_and_decoder():
    """scatter-mean is deterministic and sums in point order: bit-exact vs the oracle given the same point features;
    decoder (trilinear + MLP) vs the oracle in fp32."""
    from semabs_amd import _lib
    S, N, M, P = 16, 5000, 3000, 3
    m = _model(S, "exact")
    rng = np.random.default_rng(2)
    xyz, _, q = semabs_inputs(S, N, M, P, 9)
    pf = torch.from_numpy(rng.standard_normal((P, N, 16)).astype(np.float32))
    ref = os3.scatter_mean(torch.from_numpy(xyz).repeat(P, 1, 1), pf, SCENE_BOUNDS, (S, S, S))
    xyzd = torch.from_numpy(xyz[0]).cuda()
    flat = m.vg.flat_idxs(xyzd)
    vol = torch.zeros(P, S, S, S, 16, dtype=torch.float32, device="cuda")
    head = torch.full((S ** 3,), -1, dtype=torch.int32, device="cuda")
    nxt = torch.empty(N, dtype=torch.int32, device="cuda")
    pf_d = pf.cuda()
    _lib.call("semabs_scatter_mean", _lib.ptr(flat), _lib.ptr(pf_d), _lib.ptr(head), _lib.ptr(nxt), _lib.ptr(vol), P, N, 16, S ** 3, 1,
              _lib.stream())
    assert torch.equal(vol.cpu().permute(0, 4, 1, 2, 3), ref)
    sd = make_semabs3d_state_dict(seed=3)
    feats = torch.from_numpy(rng.standard_normal((P, 16, S, S, S)).astype(np.float32))
    refd = os3.decoder(sd, feats, torch.from_numpy(q[0]), SCENE_BOUNDS, (S, S, S), True)[..., 0]
    got = m.decode(_cl(feats).cuda(), torch.from_numpy(q[0]).cuda())
    np.testing.assert_allclose(got.cpu().numpy(), refd.numpy(), rtol=1e-4, atol=1e-5)
    got_s = m.decode(_cl(feats).cuda(), torch.from_numpy(q[0, 0]).cuda(), shared=True)
    refs = os3.decoder(sd, feats, torch.from_numpy(q[0, :1]).repeat(P, 1, 1), SCENE_BOUNDS, (S, S, S), True)[..., 0]
    np.testing.assert_allclose(got_s.cpu().numpy(), refs.numpy(), rtol=1e-4, atol=1e-5)


def test_point_mlp():
    from semabs_amd import _lib
    m = _model(16, "fp16")
    rng = np.random.default_rng(4)
    N, P = 1000, 3
    xyz = torch.from_numpy(rng.standard_normal((N, 3)).astype(np.float32))
    feat = torch.from_numpy(rng.standard_normal((P, N)).astype(np.float32))
    sd = make_semabs3d_state_dict(seed=3)
    ref = os3.point_mlp(sd, xyz[None].repeat(P, 1, 1), feat[..., None])
    pf = torch.empty(P, N, 16, dtype=torch.float32, device="cuda")
    m._sync()                                        # derive the kernel operands from the module's parameters
    w = m._w
    xyz_d, feat_d = xyz.cuda(), feat.cuda()          # keep the device buffers alive across the launch
    _lib.call("semabs_point_mlp", _lib.ptr(xyz_d), _lib.ptr(feat_d), _lib.ptr(w["w1"]), _lib.ptr(w["b1"]), _lib.ptr(w["w2"]),
              _lib.ptr(w["b2"]), _lib.ptr(w["w3"]), _lib.ptr(w["b3"]), _lib.ptr(pf), P, N, 128, 16, _lib.stream())
    np.testing.assert_allclose(pf.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-5)
    pf2 = torch.empty_like(pf)                      # the fp32 FMA kernel (its own entry point) against the same oracle
    _lib.call("semabs_point_mlp_fma", _lib.ptr(xyz_d), _lib.ptr(feat_d), _lib.ptr(w["w1"]), _lib.ptr(w["b1"]), _lib.ptr(w["w2"]),
              _lib.ptr(w["b2"]), _lib.ptr(w["w3"]), _lib.ptr(w["b3"]), _lib.ptr(pf2), P, N, 128, 16, _lib.stream())
    np.testing.assert_allclose(pf2.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("precision,tol", [("exact", 3e-4), ("fp16", 3e-2)])
def test_unet128_vs_golden(golden, precision, tol, bar):
    """One 128^3 ResidualUNet3D forward (config 3 shape) against sampled voxels of the reference's output."""
    g = golden("g10_unet128")
    u = _unet(precision, make_semabs3d_state_dict(seed=int(g["meta"][1])))
    rng = np.random.default_rng(int(g["meta"][0]))
    x = np.zeros((1, 16, 128, 128, 128), np.float32)
    occ = rng.random((128, 128, 128)) < 0.03
    x[0][:, occ] = rng.standard_normal((16, int(occ.sum()))).astype(np.float32)
    y = u.forward(torch.from_numpy(x)).cpu().numpy()
    err = np.abs(y.reshape(-1)[g["si"]] - g["y_s"]).max()
    print(f"{precision}: unet128 Linf {err:.3e} (max|ref| {np.abs(g['y_s']).max():.3f})")
    assert bar("unet128", err, tol * np.abs(g["y_s"]).max())


def test_decoder_lattice_walk_is_bit_identical():
    """The lattice hint only changes the order in which the kernel visits the queries: same bits, same output layout."""
    S, P = 32, 3
    m = _model(S, "exact")
    rng = np.random.default_rng(4)
    feats = _cl(torch.from_numpy(rng.standard_normal((P, 16, S, S, S)).astype(np.float32))).cuda()
    lo, hi = np.asarray(SCENE_BOUNDS[0], np.float32), np.asarray(SCENE_BOUNDS[1], np.float32)
    g = np.stack(np.meshgrid(np.arange(S), np.arange(S), np.arange(S), indexing="ij"), axis=-1).astype(np.float32)
    pts = torch.from_numpy((g * ((hi - lo) / np.float32(S - 1)) + lo).reshape(-1, 3).astype(np.float32)).cuda()
    plain = m.decode(feats, pts, shared=True)
    fast = m.decode(feats, pts, shared=True, lattice=(S, S, S))
    assert torch.equal(plain, fast)
    per_label = pts[None].repeat(P, 1, 1).contiguous()
    assert torch.equal(m.decode(feats, per_label, lattice=(S, S, S)), plain)
    # a lattice whose dims do not tile (32, 2, 4) silently takes the plain walk
    assert torch.equal(m.decode(feats, pts[: 30 * 32 * 32], shared=True, lattice=(30, 32, 32)), plain[:, : 30 * 32 * 32])


@pytest.mark.parametrize("precision,tol", [("exact", 2e-5), ("fp16", 4e-3)])
@pytest.mark.parametrize("cin,cout,dims,B", [(32, 16, (4, 8, 16), 2), (64, 32, (8, 8, 16), 1), (128, 64, (4, 16, 32), 1)])
def test_convtranspose3d_brick_kernel(precision, tol, cin, cout, dims, B, bar):
    """Input dims that tile into 4 x 8 x 16 bricks take the two-launch LDS-halo kernel: against torch and against the parity-class gather
    launches (same arithmetic, different summation order across taps)."""
    from semabs_amd import _lib
    from semabs_amd.unet3d import _ConvT
    rng = np.random.default_rng(cin + 1)
    D0, D1, D2 = dims
    x = torch.from_numpy(rng.standard_normal((B, cin, D0, D1, D2)).astype(np.float32))
    w = torch.from_numpy((rng.standard_normal((cin, cout, 3, 3, 3)) / np.sqrt(27 * cin / 8)).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal(cout).astype(np.float32))
    skip = torch.from_numpy(rng.standard_normal((B, cout, 2 * D0, 2 * D1, 2 * D2)).astype(np.float32))
    u = _unet(precision)
    xd, sd_ = _cl(x).cuda().to(u.act_dtype), _cl(skip).cuda().to(u.act_dtype)
    ref = sd_.float().cpu().permute(0, 4, 1, 2, 3) + F.conv_transpose3d(xd.float().cpu().permute(0, 4, 1, 2, 3), w, b, stride=2, padding=1,
                                                                         output_padding=1)
    ct = _ConvT(w, b, u.dev)
    y_brick = u._up(xd, sd_, ct).float().cpu().permute(0, 4, 1, 2, 3)
    y_gather = u._up(xd, sd_, ct, generic=True).float().cpu().permute(0, 4, 1, 2, 3)
    scale = max(1.0, ref.abs().max().item())
    assert bar("vs_torch", (y_brick - ref).abs().max().item(), tol * scale, floor=None if precision == "exact" else tol * scale)
    assert bar("vs_generic", (y_brick - y_gather).abs().max().item(), (1e-5 if precision == "exact" else 2e-3) * scale, floor=None if precision == "exact" else (1e-5 if precision == "exact" else 2e-3) * scale)


def test_decoder_folded_final_conv_matches_materialised():
    """decode(feature_volume(skip_final=True), pre_final=True) == decode(feature_volume()): the final 1x1x1 conv commutes with the trilinear
    interpolation; only fp32 summation order differs."""
    S, N, M, P = 32, 3000, 2048, 2
    m = _model(S, "exact")
    xyz, feat, q = semabs_inputs(S, N, M, P, 11)
    xyzd = torch.from_numpy(xyz[0]).cuda()
    featd = torch.from_numpy(feat[0, :, :, 0]).cuda().contiguous()
    qd = torch.from_numpy(q[0]).cuda()
    full = m.decode(m.feature_volume(xyzd, featd), qd)
    folded = m.decode(m.feature_volume(xyzd, featd, skip_final=True), qd, pre_final=True)
    scale = float(full.abs().max())
    assert float((full - folded).abs().max()) <= 2e-5 * scale and scale > 0


@pytest.mark.parametrize("precision", ["exact", "fp16"])
@pytest.mark.parametrize("cin,cout,dims,B", [(16, 16, (16, 24, 32), 3), (16, 16, (8, 8, 16), 1), (16, 32, (8, 8, 16), 2), (32, 64, (3, 5, 6), 2),
                                            (32, 32, (8, 16, 32), 2), (64, 64, (8, 16, 32), 64), (128, 256, (4, 8, 8), 2), (256, 512, (4, 8, 8), 1)])
def test_conv3d_output_statistics(precision, cin, cout, dims, B):
    """semabs_conv3d_stats: the GroupNorm statistics of the output handed to the next layer - fused into the level-0 kernel's epilogue
    (Cin = Cout = 16 on 8 x 8 x 16 bricks) and the brick kernel's (4 x 8 x 16 / 4 x 8 x 8 tiles), a statistics pass elsewhere - equal the sums of the stored output, and the output itself
    is the plain semabs_conv3d result."""
    from semabs_amd.unet3d import _Conv
    rng = np.random.default_rng(cin * 7 + cout)
    u = _unet(precision)
    x = _cl(torch.from_numpy(rng.standard_normal((B, cin, *dims)).astype(np.float32)) * 1.5 + 0.3).cuda().to(u.act_dtype)
    w = torch.from_numpy((rng.standard_normal((cout, cin, 3, 3, 3)) / np.sqrt(27 * cin)).astype(np.float32))
    gw = torch.from_numpy((1 + 0.2 * rng.standard_normal(cin)).astype(np.float32))
    gb = torch.from_numpy((0.2 * rng.standard_normal(cin)).astype(np.float32))
    res = _cl(torch.from_numpy(rng.standard_normal((B, cout, *dims)).astype(np.float32))).cuda().to(u.act_dtype)
    conv = _Conv(w, gw, gb, None, 8, u.dev)
    y0 = u._conv(x, conv, relu=True, resid=res)
    y, sums = u._conv(x, conv, relu=True, resid=res, out_groups=8)
    # (not bit-equal: the input statistics of each call are reduced with floating-point atomics, so the GroupNorm affine may differ by an ulp)
    assert (y.float() - y0.float()).abs().max().item() <= (1e-5 if precision == "exact" else 4e-3) * float(y0.float().abs().max())
    yd = y.double().reshape(B, -1, 8, cout // 8)                       # [B, voxels, group, channel in group]
    want = torch.stack([yd.sum(dim=(1, 3)), (yd * yd).sum(dim=(1, 3))], dim=-1)
    np.testing.assert_allclose(sums.cpu().numpy(), want.cpu().numpy(), rtol=2e-6, atol=1e-6 * float(want.abs().max()))


@pytest.mark.parametrize("precision", ["exact", "fp16"])
@pytest.mark.parametrize("cin,cout,dims,B", [(32, 16, (4, 8, 16), 2), (64, 32, (4, 8, 16), 1), (256, 128, (4, 8, 16), 1), (32, 16, (3, 4, 5), 2)])
def test_convtranspose3d_output_statistics(precision, cin, cout, dims, B):
    """semabs_convtranspose3d_stats: GroupNorm statistics of (ConvTranspose3d + skip) fused into the brick kernel's epilogue (last case:
    the per-class fallback + a statistics pass) equal the sums of the stored output."""
    from semabs_amd.unet3d import _ConvT
    rng = np.random.default_rng(cin + cout)
    u = _unet(precision)
    x = _cl(torch.from_numpy(rng.standard_normal((B, cin, *dims)).astype(np.float32))).cuda().to(u.act_dtype)
    w = torch.from_numpy((rng.standard_normal((cin, cout, 3, 3, 3)) / np.sqrt(27 * cin / 8)).astype(np.float32))
    bias = torch.from_numpy((0.1 * rng.standard_normal(cout)).astype(np.float32))
    skip = _cl(torch.from_numpy(rng.standard_normal((B, cout, *[2 * d for d in dims])).astype(np.float32))).cuda().to(u.act_dtype)
    ct = _ConvT(w, bias, u.dev)
    y0 = u._up(x, skip, ct)
    y, sums = u._up(x, skip, ct, out_groups=8)
    assert torch.equal(y, y0)
    yd = y.double().reshape(B, -1, 8, cout // 8)
    want = torch.stack([yd.sum(dim=(1, 3)), (yd * yd).sum(dim=(1, 3))], dim=-1)
    np.testing.assert_allclose(sums.cpu().numpy(), want.cpu().numpy(), rtol=3e-6, atol=2e-6 * float(want.abs().max()))


@pytest.mark.parametrize("vol_f32", [1, 0])
def test_scatter_mean_statistics(vol_f32):
    """semabs_scatter_mean_stats: same volume as semabs_scatter_mean, and the GroupNorm statistics of the dense volume summed over the
    occupied voxels only (N not a multiple of 64: waves that straddle two label volumes take the per-lane path)."""
    from semabs_amd import _lib
    rng = np.random.default_rng(9)
    P, N, C, S = 3, 1000, 16, 12
    nvox = S ** 3
    flat = torch.from_numpy(rng.integers(0, nvox, size=N).astype(np.int64)).cuda()
    feat = torch.from_numpy(rng.standard_normal((P, N, C)).astype(np.float32)).cuda()
    dt = torch.float32 if vol_f32 else torch.float16
    vols = []
    for with_stats in (False, True):
        vol = torch.zeros(P, nvox, C, dtype=dt, device="cuda")
        head = torch.full((nvox,), -1, dtype=torch.int32, device="cuda")
        nxt = torch.empty(N, dtype=torch.int32, device="cuda")
        if with_stats:
            sums = torch.zeros(P, 8, 2, dtype=torch.float64, device="cuda")
            _lib.call("semabs_scatter_mean_stats", _lib.ptr(flat), _lib.ptr(feat), _lib.ptr(head), _lib.ptr(nxt), _lib.ptr(vol), P, N, C, nvox, vol_f32,
                      _lib.ptr(sums), _lib.stream())
        else:
            _lib.call("semabs_scatter_mean", _lib.ptr(flat), _lib.ptr(feat), _lib.ptr(head), _lib.ptr(nxt), _lib.ptr(vol), P, N, C, nvox, vol_f32, _lib.stream())
        vols.append(vol)
    assert torch.equal(vols[0], vols[1])
    vd = vols[1].double().reshape(P, nvox, 8, 2)
    want = torch.stack([vd.sum(dim=(1, 3)), (vd * vd).sum(dim=(1, 3))], dim=-1)
    np.testing.assert_allclose(sums.cpu().numpy(), want.cpu().numpy(), rtol=3e-6, atol=2e-6 * float(want.abs().max()))


@pytest.mark.parametrize("precision", ["exact", "fp16"])
def test_sparse_scatter_equals_dense(precision):
    """semabs_scatter_mean_sparse + semabs_conv3d_sparse_stats (the volume is neither zero-filled nor read where the occupancy bitmap is clear) against the dense
    form on the same points: the UNet's first block and everything behind it see the same values.  The scattered volume of the dense run is poisoned with NaN
    bit patterns in a second sparse run's allocation to show that empty voxels are never read."""
    import semabs_amd.net as snet
    S, N, P = 32, 3000, 3
    m = _model(S, precision)
    xyz, feat, _ = semabs_inputs(S, N, 64, P, 5)
    x, f = torch.from_numpy(xyz[0]).cuda(), torch.from_numpy(feat[0, :, :, 0]).cuda()
    old = snet.SPARSE_SCATTER
    try:
        snet.SPARSE_SCATTER = False
        dense = m.feature_volume(x, f).clone()
        snet.SPARSE_SCATTER = True
        # poison what the caching allocator will hand to the sparse run's torch.empty volume
        junk = torch.full((P, S, S, S, 16), float("nan"), dtype=m.vol_feature_extractor.act_dtype, device="cuda")
        del junk
        sparse = m.feature_volume(x, f).clone()
    finally:
        snet.SPARSE_SCATTER = old
    assert bool(torch.isfinite(sparse).all())
    d = float((sparse.float() - dense.float()).abs().max())
    print(f"{precision}: sparse vs dense scatter, UNet feature L-inf {d:.3e} (max {float(dense.float().abs().max()):.3f})")
    # the first GroupNorm's statistics are fp64 atomics in both forms: a few ulps of run-to-run spread, nothing more
    assert d <= (2e-5 if precision == "exact" else 2e-2) * float(dense.float().abs().max())
