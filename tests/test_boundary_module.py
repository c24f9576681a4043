"""The nn.Module boundary the reference's own caller needs (utils.get_net, /root/reference/utils.py:225-296, re-created here -
no reference file is imported or copied):

    net = net_class(**kwargs).to(device); get_n_params(net); Lamb(net.parameters(), ...); DistributedDataParallel(module=net, ...);
    net.load_state_dict(ckpt["net"])  (with the "module." prefix under DDP); net.eval(); net.steps

CPU part: constructor, parameter inventory (counts equal the reference's: 35 403 969 / 35 407 633), state-dict keys and
round trip, strict / non-strict loading, DDP wrapping over gloo.  GPU part (marked): forward through the HIP kernels after `.to(device)`, a
LAMB step over `net.parameters()` changes the next forward (the derived kernel operands follow the parameters), checkpoint round trip."""
import os

import numpy as np
import pytest
import torch

import semabs_amd  # noqa: F401
from semabs_amd.weights import make_semabs3d_state_dict, make_semabsvool_state_dict

BOUNDS = [[-1.0, -1.0, -0.1], [1.0, 1.0, 1.9]]
KW = dict(voxel_shape=(32, 32, 32), scene_bounds=BOUNDS, unet_num_channels=16, unet_f_maps=16, unet_num_groups=8, unet_num_levels=6,
          network_inputs=["saliency"], use_pts_feat_extractor=True, pts_feat_extractor_hidden_dim=128, reduce_method="max", output_dim=1,
          decoder_concat_xyz_pts=True, batch_size=1)


def get_n_params(model):                                     # what utils.get_n_params computes
    return sum(int(np.prod(p.size())) for p in model.parameters())


def test_semabs3d_is_a_module_with_the_reference_inventory():
    from semabs_amd.net import SemAbs3D
    torch.manual_seed(0)
    net = SemAbs3D(device="cpu", **KW)
    assert isinstance(net, torch.nn.Module)
    assert get_n_params(net) == 35403969
    ref = make_semabs3d_state_dict(seed=3)
    sd = net.state_dict()
    assert list(sd.keys()) and set(sd.keys()) == set(ref.keys()) and all(tuple(sd[k].shape) == tuple(ref[k].shape) for k in ref)
    assert "steps" in dict(net.named_buffers()) and "steps" not in dict(net.named_parameters()) and float(net.steps) == 0.0
    assert net.device == "cpu" and net.training and net.eval() is net and not net.training and net.train() is net
    # seeded construction is reproducible, like seed_all(seed) -> net_class(**kwargs)
    torch.manual_seed(0)
    again = SemAbs3D(device="cpu", **KW)
    assert all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), again.state_dict().values()))
    # load / round trip; DDP-saved keys; strictness like nn.Module
    assert not net.load_state_dict(ref).missing_keys
    assert all(torch.equal(net.state_dict()[k], ref[k]) for k in ref)
    net.load_state_dict({"module." + k: v for k, v in ref.items()})
    partial = {k: v for k, v in ref.items() if "final_conv" not in k}
    with pytest.raises(RuntimeError, match="Missing key"):
        net.load_state_dict(partial)
    res = net.load_state_dict(partial, strict=False)
    assert sorted(res.missing_keys) == ["vol_feature_extractor.final_conv.bias", "vol_feature_extractor.final_conv.weight"]
    with pytest.raises(RuntimeError, match="Unexpected key"):
        net.load_state_dict(dict(ref, bogus=torch.zeros(1)))


def test_semabs3d_tsdf_input_shapes_and_unsupported_inputs():
    from semabs_amd.net import SemAbs3D
    net = SemAbs3D(device="cpu", **dict(KW, network_inputs=["saliency", "tsdf"]))
    assert tuple(net.state_dict()["pts_feat_extractor.4.weight"].shape) == (15, 128)        # one UNet input channel is the TSDF (net.py:365-367)
    assert net.vol_feature_extractor.in_channels == 16
    with pytest.raises(NotImplementedError):
        SemAbs3D(device="cpu", **dict(KW, network_inputs=["rgb"]))


def test_semabsvool_is_a_module_with_the_reference_inventory():
    from semabs_amd.net import SemAbsVOOL
    kw = {k: v for k, v in KW.items() if k != "decoder_concat_xyz_pts"}
    net = SemAbsVOOL(pointing_method="cosine_sim", pointing_dim=64, device="cpu", decoder_concat_xyz_pts=True, **kw)
    assert get_n_params(net) == 35407633
    ref = make_semabsvool_state_dict(seed=1)
    assert set(net.state_dict().keys()) == set(ref.keys())
    assert "relation_embeddings.on the left of" in dict(net.named_parameters()) and "completion_net.steps" in net.state_dict()
    net.load_state_dict(ref)
    assert all(torch.equal(net.state_dict()[k], ref[k]) for k in ref)


def test_distributed_data_parallel_wraps_the_module():
    """utils.get_net: DistributedDataParallel(module=net, device_ids=[device], find_unused_parameters=True) - here over gloo on the host."""
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel
    from semabs_amd.net import SemAbs3D
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(29800 + os.getpid() % 150)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        net = DistributedDataParallel(module=SemAbs3D(device="cpu", **dict(KW, unet_num_levels=3)), find_unused_parameters=True)
        assert all(k.startswith("module.") for k in net.state_dict())
        inner = SemAbs3D(device="cpu", **dict(KW, unet_num_levels=3))
        inner.load_state_dict(net.state_dict())                             # a DDP checkpoint loads into the bare module
        assert all(torch.equal(a, b) for a, b in zip(inner.state_dict().values(), net.module.state_dict().values()))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_get_net_sequence_on_the_gpu():
    from semabs_amd.net import SemAbs3D
    from semabs_amd.optim import Lamb
    device = "cuda"
    torch.manual_seed(3)
    net = SemAbs3D(device=device, **KW).to(device)
    optimizer = Lamb(net.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-5, adam=False)
    ref_sd = make_semabs3d_state_dict(seed=3)
    net.load_state_dict({"module." + k: v for k, v in ref_sd.items()})
    net.eval()
    rng = np.random.default_rng(0)
    lo, hi = np.array(BOUNDS[0]), np.array(BOUNDS[1])
    xyz = torch.from_numpy((lo + (hi - lo) * rng.random((1, 3000, 3))).astype(np.float32))
    feat = torch.from_numpy((rng.standard_normal((1, 2, 3000, 1)) * 0.5).astype(np.float32))
    q = torch.from_numpy((lo + (hi - lo) * rng.random((1, 2, 500, 3))).astype(np.float32))
    out0 = net(input_xyz_pts=xyz, input_feature_pts=feat, tsdf_vol=None, output_xyz_pts=q)
    assert tuple(out0.shape) == (1, 2, 500) and out0.is_cuda and torch.isfinite(out0).all()
    assert tuple(net.visual_volumetric_features.shape) == (2, 16, 32, 32, 32)
    # same weights through the oracle
    from oracle import semabs3d as os3
    with torch.no_grad():
        ref = os3.semabs3d_forward(ref_sd, xyz, feat, q, BOUNDS, (32, 32, 32))
    assert float((out0.cpu() - ref).abs().max()) <= 2e-4 * max(1.0, float(ref.abs().max()))
    # an optimizer step over net.parameters() must be seen by the next forward (kernel operands are derived from the parameters)
    for p in net.parameters():
        p.grad = torch.full_like(p, 1e-2)
    optimizer.step()
    out1 = net(input_xyz_pts=xyz, input_feature_pts=feat, tsdf_vol=None, output_xyz_pts=q)
    assert float((out1 - out0).abs().max()) > 1e-4
    # checkpoint round trip restores the first result (to the last bits: GroupNorm statistics are accumulated with floating-point atomics)
    net.load_state_dict(ref_sd)
    out2 = net(input_xyz_pts=xyz, input_feature_pts=feat, tsdf_vol=None, output_xyz_pts=q)
    assert float((out2 - out0).abs().max()) <= 1e-5 * float(out0.abs().max())


@pytest.mark.gpu
def test_semabs3d_tsdf_network_input_vs_reference(golden):
    """network_inputs = ["saliency", "tsdf"] (net.py:346-357, 411-419): the TSDF volume becomes UNet input channel 0, the point MLP fills the
    other 15.  Golden g17 = the unmodified reference module's output with its own parameters (stored in the fixture)."""
    from semabs_amd.net import SemAbs3D
    g = golden("g17_semabs3d_tsdf")
    S, L = int(g["meta"][0]), int(g["meta"][4])
    kw = dict(KW, voxel_shape=(S, S, S), unet_num_levels=L, network_inputs=["saliency", "tsdf"])
    net = SemAbs3D(device="cuda", **kw).to("cuda").eval()
    net.load_state_dict({k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd::")})
    out = net(input_xyz_pts=torch.from_numpy(g["xyz"]), input_feature_pts=torch.from_numpy(g["feat"]), tsdf_vol=torch.from_numpy(g["tsdf"]),
              output_xyz_pts=torch.from_numpy(g["q"]))
    err = float(np.abs(out.cpu().numpy() - g["out"]).max())
    print(f"tsdf-input SemAbs3D: logit L-inf {err:.3e} (max|ref| {np.abs(g['out']).max():.3f})")
    assert err <= 2e-4 * max(1.0, float(np.abs(g["out"]).max()))
    with pytest.raises(ValueError, match="tsdf_vol"):
        net(input_xyz_pts=torch.from_numpy(g["xyz"]), input_feature_pts=torch.from_numpy(g["feat"]), tsdf_vol=None, output_xyz_pts=torch.from_numpy(g["q"]))


@pytest.mark.gpu
def test_semabs3d_tsdf_batch_pairing_quirk_vs_oracle():
    """B = 2 scenes x P = 2 label volumes with "tsdf" input: the reference concatenates `tsdf_vol.unsqueeze(1).repeat(P, 1, 1, 1, 1)` (order
    b0, b1, b0, b1) to the b-major feature stack (b0p0, b0p1, b1p0, b1p1), net.py:411-419 - volume (b, p) gets tsdf_vol[(b P + p) % B].
    The oracle restates that line literally (and is pinned to the reference module at B = 1 by g17); the HIP path must pair the same way."""
    from oracle import semabs3d as os3
    from semabs_amd.net import SemAbs3D
    S, L, N, M = 16, 3, 1500, 300
    kw = dict(KW, voxel_shape=(S, S, S), unet_num_levels=L, network_inputs=["saliency", "tsdf"])
    sd = make_semabs3d_state_dict(seed=6, unet_num_levels=L)
    sd["pts_feat_extractor.4.weight"] = sd["pts_feat_extractor.4.weight"][:15].clone()
    sd["pts_feat_extractor.4.bias"] = sd["pts_feat_extractor.4.bias"][:15].clone()
    net = SemAbs3D(device="cuda", **kw).to("cuda").eval()
    net.load_state_dict(sd)
    rng = np.random.default_rng(3)
    lo, hi = np.array(BOUNDS[0]), np.array(BOUNDS[1])
    xyz = torch.from_numpy((lo + (hi - lo) * rng.random((2, N, 3))).astype(np.float32))
    feat = torch.from_numpy((rng.standard_normal((2, 2, N, 1)) * 0.5).astype(np.float32))
    q = torch.from_numpy((lo + (hi - lo) * rng.random((2, 2, M, 3))).astype(np.float32))
    tsdf = torch.from_numpy(rng.uniform(-1, 1, (2, S, S, S)).astype(np.float32))
    out = net(input_xyz_pts=xyz, input_feature_pts=feat, tsdf_vol=tsdf, output_xyz_pts=q)
    with torch.no_grad():
        ref = os3.semabs3d_forward(sd, xyz, feat, q, BOUNDS, (S, S, S), num_levels=L, tsdf_vol=tsdf)
        own = os3.semabs3d_forward(sd, xyz, feat, q, BOUNDS, (S, S, S), num_levels=L, tsdf_vol=tsdf[[0, 0]])   # what "every volume its own scene" would give for b = 0
    err = float((out.cpu() - ref).abs().max())
    assert err <= 2e-4 * max(1.0, float(ref.abs().max())), err
    assert float((ref[0, 1] - own[0, 1]).abs().max()) > 1e-3          # the quirk is visible in this input: (b0, p1) reads scene 1's TSDF


def test_parameter_order_is_the_references(golden):
    """`net.parameters()` is positional for everything the reference builds on it - `Lamb(net.parameters())`, `optimizer.state_dict()` /
    `load_state_dict` in the checkpoint (utils.py:264-266, 278-296).  g20's `names` = `named_parameters()` of the unmodified reference SemAbsVOOL:
    the drop-in module, the seeded state-dict generator and the fused trainer's optimizer must list the parameters in that order."""
    from semabs_amd.net import SemAbs3D, SemAbsVOOL
    ref = [str(k) for k in golden("g20_vool_train64")["names"]]
    kw = {k: v for k, v in KW.items() if k != "decoder_concat_xyz_pts"}
    net = SemAbsVOOL(pointing_method="cosine_sim", pointing_dim=64, device="cpu", decoder_concat_xyz_pts=True, **kw)
    assert [k for k, _ in net.named_parameters()] == ref
    sd = make_semabsvool_state_dict(seed=1)
    assert [k for k, v in sd.items() if torch.is_floating_point(v) and not k.endswith("steps")] == ref
    inner = SemAbs3D(device="cpu", **dict(KW, decoder_concat_xyz_pts=False))
    assert ["completion_net." + k for k, _ in inner.named_parameters()] == [k for k in ref if k.startswith("completion_net.")]
