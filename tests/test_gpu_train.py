"""GPU: the hand-written backward pass / training step (csrc/train.hip, semabs_amd/train.py) against torch-CPU autograd over the
oracle's functional forward, and against the golden of the reference's own training step (G13)."""
import numpy as np
import pytest
import torch

import semabs_amd  # noqa: F401
from semabs_amd.synth import SCENE_BOUNDS
from semabs_amd.weights import make_semabs3d_state_dict, make_semabsvool_state_dict

from _train_inputs import vool_batch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


def _robust(a, b):
    """(relative L2 error, median |err| / max |ref|).  ReLU masks and max-pool arg-maxes are decided by pre-activations that can sit
    within rounding noise of zero / of each other: a single flipped element (the GPU forward is itself not bitwise reproducible: fp64
    atomics in the GroupNorm sums) moves the L-inf error to ~1e-2 while leaving the bulk untouched, on any pair of implementations."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30), np.median(np.abs(a - b)) / max(np.abs(b).max(), 1e-30)


def _unet_setup(L, seed):
    from semabs_amd.train import UNetTrainer
    pre = "vol_feature_extractor."
    sd = {k: v for k, v in make_semabs3d_state_dict(seed=seed, unet_num_levels=L).items() if k.startswith(pre)}
    dev = torch.device("cuda:0")
    params = {k: v.float().to(dev).contiguous() for k, v in sd.items()}
    grads = {k: torch.zeros_like(v) for k, v in params.items()}
    u = UNetTrainer(params, grads, pre, 16, 16, 16, 8, L)
    u.refresh()
    return sd, params, grads, u, pre


def _cl(x):
    return torch.from_numpy(x).cuda().permute(0, 2, 3, 4, 1).contiguous()


def _uncl(x):
    return x.permute(0, 4, 1, 2, 3).cpu().numpy()


@pytest.mark.parametrize("wgrad_gn", [True, False])
@pytest.mark.parametrize("gscale", [1.0, 1e-7])
def test_gn_conv_layer_backward_strict(gscale, wgrad_gn):
    """One GroupNorm + Conv3d (no ReLU, so nothing can flip): data, weight and affine gradients to fp32 accuracy - also for gradients
    of magnitude 1e-7, which the split-fp16 MFMA operands only survive through the dynamic power-of-two scale."""
    import torch.nn.functional as F
    sd, params, grads, u, pre = _unet_setup(3, 5)
    u.wgrad_gn = wgrad_gn                                    # False: semabs_wgrad_conv3 + semabs_chan_reduce (the path of shapes semabs_wgrad_conv3_gn does not take)
    rng = np.random.default_rng(2)
    # 16^3 volumes take the MFMA brick kernel for the weight gradient (semabs_wgrad_conv3, incl. channel slicing), the others the fp32 one
    for name, cin, cout, s in [("encoders.0.basic_module.conv1.", 16, 16, 16), ("encoders.1.basic_module.conv1.", 16, 32, 8),
                               ("encoders.2.basic_module.conv2.", 64, 64, 4), ("encoders.1.basic_module.conv1.", 16, 32, 16),
                               ("encoders.1.basic_module.conv2.", 32, 32, 16)]:
        x = rng.standard_normal((2, cin, s, s, s)).astype(np.float32) + 0.3
        dz = (rng.standard_normal((2, cout, s, s, s)) * gscale).astype(np.float32)
        key = pre + name
        w = sd[key + "conv.weight"].clone().requires_grad_(True)
        ga = sd[key + "groupnorm.weight"].clone().requires_grad_(True)
        be = sd[key + "groupnorm.bias"].clone().requires_grad_(True)
        xt = torch.from_numpy(x).requires_grad_(True)
        F.conv3d(F.group_norm(xt, 8, ga, be, 1e-5), w, None, padding=1).backward(torch.from_numpy(dz))
        for k in grads:
            grads[k].zero_()
        r, _ = u._conv_fwd(_cl(x), name, False)
        dx = u._conv_bwd(r, _cl(dz))
        torch.cuda.synchronize()
        assert _rel(_uncl(dx), xt.grad.numpy()) < 1e-5, name
        assert _rel(grads[key + "conv.weight"].cpu().numpy(), w.grad.numpy()) < 1e-5, name
        assert _rel(grads[key + "groupnorm.weight"].cpu().numpy(), ga.grad.numpy()) < 1e-5, name
        assert _rel(grads[key + "groupnorm.bias"].cpu().numpy(), be.grad.numpy()) < 1e-5, name


@pytest.mark.parametrize("relu_in,with_add", [(False, False), (True, False), (True, True)])
def test_conv3d_gnbwd_equals_conv_then_apply(relu_in, with_add):
    """semabs_conv3d_gnbwd (the GroupNorm-backward apply as the epilogue of the level-0 data-gradient convolution) against the two-kernel path it replaces
    (semabs_conv3d + semabs_gn_bwd_apply), through UNetTrainer._conv_bwd with the fusion on and off: data gradient, recorded max |dX| and every parameter
    gradient - with the ReLU mask of the layer input and with the residual branch's gradient added."""
    sd, params, grads, u, pre = _unet_setup(3, 5)
    rng = np.random.default_rng(7)
    name, cin, cout, s = "encoders.0.basic_module.conv2.", 16, 16, 16
    x = np.maximum(rng.standard_normal((2, cin, s, s, s)).astype(np.float32) + 0.2, 0.0)          # a post-ReLU activation (zeros included)
    dz = (rng.standard_normal((2, cout, s, s, s)) * 1e-4).astype(np.float32)
    add = (rng.standard_normal((2, cin, s, s, s)) * 1e-4).astype(np.float32)
    outs = []
    for fuse in (True, False):
        u.fuse_gn_apply = fuse
        for k in grads:
            grads[k].zero_()
        r, _ = u._conv_fwd(_cl(x), name, False)
        dx = u._conv_bwd(r, _cl(dz), add1=_cl(add) if with_add else None, relu_in=relu_in)
        torch.cuda.synchronize()
        outs.append((dx.cpu().numpy(), float(dx._semabs_absmax.view(torch.float32)), {k: v.cpu().numpy().copy() for k, v in grads.items() if name in k}))
    (a, am, ga), (b, bm, gb) = outs
    assert _rel(a, b) < 1e-5
    assert am == float(np.abs(a).max()) and bm == float(np.abs(b).max())
    if relu_in:
        assert bool((a[np.transpose(x, (0, 2, 3, 4, 1)) <= 0] == 0).all())
    for k in ga:
        assert _rel(ga[k], gb[k]) < 1e-6, k


@pytest.mark.parametrize("B,D,Ca,Cx,G,gscale", [(2, (8, 12, 32), 16, 16, 8, 1.0), (3, (4, 4, 16), 32, 16, 8, 1.0), (8, (16, 16, 16), 32, 32, 8, 1e-7),
                                                (1, (8, 8, 16), 16, 32, 4, 1.0), (4, (16, 8, 32), 64, 64, 8, 1.0),
                                                (2, (8, 8, 16), 48, 16, 8, 1.0), (1, (8, 12, 16), 96, 32, 8, 1.0)])     # Ca / 4 does not divide 256 (boundary sums counted twice)
def test_wgrad_conv3_gn_vs_fp64(B, D, Ca, Cx, G, gscale):
    """semabs_wgrad_conv3_gn: the weight gradient and the GroupNorm-backward sums (sum dXn, sum dXn xhat) of a GroupNorm -> Conv3d layer from ONE pass over
    (dZ, x), against torch fp64 autograd / fp64 sums over the materialised dXn - non-cubic volumes (every face of the restricted sums differs), batch sizes
    that do and do not divide the workgroup count, channel slicing, gradients of magnitude 1e-7 through the dynamic scale, a non-zero-mean input."""
    import ctypes as C
    import torch.nn.functional as F
    from semabs_amd import _lib
    g = torch.Generator().manual_seed(B * 1000 + Ca + Cx + D[0])
    x = (torch.randn(B, Cx, *D, generator=g) * 0.8 + 0.4).double()
    dz = (torch.randn(B, Ca, *D, generator=g) * gscale).double()
    w = (torch.randn(Ca, Cx, 3, 3, 3, generator=g) * 0.1).double().requires_grad_(True)
    ga = (torch.rand(Cx, generator=g) + 0.5).double().requires_grad_(True)
    be = (torch.randn(Cx, generator=g) * 0.3).double().requires_grad_(True)
    xn_in = x.clone().requires_grad_(True)
    xn = F.group_norm(xn_in, G, ga, be, 1e-5)
    xn.retain_grad()
    F.conv3d(xn, w, None, padding=1).backward(dz)
    xg = x.reshape(B, G, -1)
    mean, var = xg.mean(-1), xg.var(-1, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    xhat = ((xg - mean[..., None]) * rstd[..., None]).reshape(x.shape)
    dxn = xn.grad                                            # = conv^T(dz)
    sa, sb = dxn.sum(dim=(2, 3, 4)), (dxn * xhat).sum(dim=(2, 3, 4))         # [B, Cx]
    scratch = torch.empty(16 << 20, device="cuda")
    ok = C.c_int(0)
    _lib.call("semabs_wgrad_conv3_gn_supported", B, D[0], D[1], D[2], Ca, Cx, scratch.numel(), C.byref(ok))
    assert ok.value == 1
    cl = lambda t: t.float().cuda().permute(0, 2, 3, 4, 1).contiguous()
    dzc, xc = cl(dz), cl(x)
    s = 2.0 ** 20 if gscale < 1e-3 else 1.0
    s2 = torch.tensor([s, 1.0 / s], device="cuda")
    base = 0.5 * gscale
    dW = torch.full((Ca, Cx, 27), base, device="cuda")       # accumulated into
    red = torch.zeros(B, Cx, 2, dtype=torch.float64, device="cuda")
    dev = [t.detach().float().cuda().contiguous() for t in (mean, rstd, ga, be, w)]      # (kept alive across the call)
    _lib.call("semabs_wgrad_conv3_gn", _lib.ptr(dzc), _lib.ptr(xc), _lib.ptr(dev[0]), _lib.ptr(dev[1]), G, _lib.ptr(dev[2]), _lib.ptr(dev[3]), _lib.ptr(dev[4]),
              _lib.ptr(s2), _lib.ptr(dW), _lib.ptr(red), B, D[0], D[1], D[2], Ca, Cx, _lib.ptr(scratch), scratch.numel(), _lib.stream())
    torch.cuda.synchronize()
    ref_w = w.grad.reshape(Ca, Cx, 27)
    assert float(((dW.cpu().double() - float(np.float32(base))) - ref_w).abs().max()) < 1e-5 * float(ref_w.abs().max())
    got = red.cpu() / s
    assert float((got[..., 0] - sa).abs().max()) < 1e-5 * float(sa.abs().max())
    assert float((got[..., 1] - sb).abs().max()) < 1e-5 * float(sb.abs().max())


@pytest.mark.parametrize("R,Ci,Co,act,transposed,scaled", [(100003, 4, 128, 1, False, False), (100003, 128, 128, 1, False, False), (50001, 128, 16, 0, False, False),
                                                            (70000, 36, 32, 1, False, False), (70000, 32, 64, 0, False, False), (70000, 64, 32, 0, True, True),
                                                            (50001, 16, 128, 0, True, True), (33, 128, 128, 0, True, True), (70000, 32, 36, 0, False, False),
                                                            (70003, 16, 16, 0, False, False), (65541, 16, 16, 1, False, False), (100001, 16, 16, 0, True, True)])
def test_linear_rows_vs_fp64(R, Ci, Co, act, transposed, scaled):
    """semabs_linear_rows (the MLP layers of the training step on the matrix cores, fp16 hi / lo split operands) against an fp64 product: fp32-like
    accuracy for every layer shape of the point and sampler MLPs, plain and transposed weights, ragged row counts, gradient-sized inputs through
    the power-of-two input scale, rows past R untouched.  The 16 -> 16 cases with at least 2^16 rows run on k_rows16_f32 (fp32 MFMA)."""
    from semabs_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(R + Ci + Co)
    mag = 1e-7 if scaled else 1.0
    x = torch.randn(R, Ci, device="cuda", generator=g) * mag
    w = torch.randn((Ci, Co) if transposed else (Co, Ci), device="cuda", generator=g) * 0.2
    b = torch.randn(Co, device="cuda", generator=g) * (0.0 if scaled else 1.0)
    s = torch.tensor([2.0 ** 20 if scaled else 1.0, 2.0 ** -20 if scaled else 1.0], device="cuda")
    y = torch.full((R + 16, Co), 7.0, device="cuda")
    _lib.call("semabs_linear_rows", _lib.ptr(x), Ci, _lib.ptr(w), 1 if transposed else Ci, Co if transposed else 1, _lib.ptr(b) if not scaled else None,
              _lib.ptr(y), R, Ci, Co, act, 0.01, _lib.ptr(s) if scaled else None, None, None, None, None, None, _lib.stream())
    ref = x.double() @ (w.double() if transposed else w.double().T)
    ref = ref * float(s[0]) + (0 if scaled else b.double())
    if scaled:                                               # ... and scaled back in the epilogue: the unscaled product of 1e-7-sized values
        y2 = torch.empty(R, Co, device="cuda")
        _lib.call("semabs_linear_rows", _lib.ptr(x), Ci, _lib.ptr(w), 1 if transposed else Ci, Co if transposed else 1, None, _lib.ptr(y2), R, Ci, Co, 0, 0.01,
                  _lib.ptr(s), s[1:].data_ptr(), None, None, None, None, _lib.stream())
        assert float((y2.double() - ref / float(s[0])).abs().max()) < 2e-6 * float(ref.abs().max()) / float(s[0])
    if act:
        ref = torch.where(ref > 0, ref, 0.01 * ref)
    err = float((y[:R].double() - ref).abs().max()) / float(ref.abs().max())
    assert err < 2e-6, err
    assert bool((y[R:] == 7.0).all())
    # the same through a ReLU mask (Y = 0 where mask <= 0) with max |Y| recorded
    mask = torch.randn(R, Co, device="cuda", generator=g)
    bits = torch.zeros(1, dtype=torch.int32, device="cuda")
    y3 = torch.empty(R, Co, device="cuda")
    _lib.call("semabs_linear_rows", _lib.ptr(x), Ci, _lib.ptr(w), 1 if transposed else Ci, Co if transposed else 1, _lib.ptr(b) if not scaled else None,
              _lib.ptr(y3), R, Ci, Co, act, 0.01, _lib.ptr(s) if scaled else None, None, _lib.ptr(mask), _lib.ptr(bits), None, None, _lib.stream())
    assert bool((y3 == torch.where(mask > 0, y[:R], torch.zeros_like(y3))).all())
    assert float(bits.view(torch.float32)) == float(y3.abs().max())
    cs = torch.full((Co,), 0.25, device="cuda")              # column sums of Y (accumulated): the bias gradient when Y is a layer's output gradient
    _lib.call("semabs_linear_rows", _lib.ptr(x), Ci, _lib.ptr(w), 1 if transposed else Ci, Co if transposed else 1, _lib.ptr(b) if not scaled else None,
              _lib.ptr(y3), R, Ci, Co, act, 0.01, _lib.ptr(s) if scaled else None, None, _lib.ptr(mask), None, _lib.ptr(cs), None, _lib.stream())
    ref_cs = y3.double().sum(0)
    assert float(((cs.double() - 0.25) - ref_cs).abs().max()) <= 2e-5 * float(y3.double().abs().sum(0).max()) + 1e-6
    if Ci == 16 and Co == 16 and R >= 65536:                # the 16 -> 16 kernel also sums the columns of its INPUT (the final convolution's bias gradient)
        xs = torch.full((16,), 0.0, device="cuda")
        _lib.call("semabs_linear_rows", _lib.ptr(x), Ci, _lib.ptr(w), 1 if transposed else Ci, Co if transposed else 1, None, _lib.ptr(y3), R, Ci, Co, 0, 0.01,
                  _lib.ptr(s) if scaled else None, None, None, None, None, _lib.ptr(xs), _lib.stream())
        assert float((xs.double() - x.double().sum(0)).abs().max()) <= 2e-5 * float(x.double().abs().sum(0).max())
    if not act:                                              # act = 2: the LeakyReLU form of the mask (Y *= slope where mask <= 0)
        _lib.call("semabs_linear_rows", _lib.ptr(x), Ci, _lib.ptr(w), 1 if transposed else Ci, Co if transposed else 1, _lib.ptr(b) if not scaled else None,
                  _lib.ptr(y3), R, Ci, Co, 2, 0.01, _lib.ptr(s) if scaled else None, None, _lib.ptr(mask), None, None, None, _lib.stream())
        assert bool((y3 == torch.where(mask > 0, y[:R], y[:R] * 0.01)).all())


@pytest.mark.parametrize("P,M", [(3, 1001), (1, 16), (4, 40000)])
def test_cos_bce_and_cos_head_vs_torch_fp64(P, M):
    """semabs_cos_bce / semabs_cos_head (net.py:571-579 cosine pointer at temperature 0.07 + utils.py's weighted BCE-with-logits mean): logits, loss, d loss / d o
    and d loss / d relation embedding against torch fp64 autograd; ragged point counts (16 points per block iteration)."""
    from semabs_amd import _lib
    g = torch.Generator().manual_seed(P * 7 + M)
    o = torch.randn(P * M, 64, generator=g)
    rel = torch.randn(P, 64, generator=g)
    label = (torch.rand(P * M, generator=g) < 0.3).float()
    weight = torch.rand(P * M, generator=g) + 0.5
    od, rd = o.double().requires_grad_(True), rel.double().requires_grad_(True)
    z = torch.nn.functional.cosine_similarity(od.reshape(P, M, 64), rd[:, None, :], dim=-1).reshape(-1) / 0.07
    loss = (torch.nn.functional.binary_cross_entropy_with_logits(z, label.double(), reduction="none") * weight.double()).sum() / (P * M)
    loss.backward()
    dev = [t.cuda().contiguous() for t in (o, rel, label, weight)]
    logits = torch.empty(P * M, device="cuda"); dO = torch.empty(P * M, 64, device="cuda")
    drel = torch.zeros(P, 64, device="cuda"); lossd = torch.zeros(1, dtype=torch.float64, device="cuda")
    dob = torch.zeros(P, 64, device="cuda")
    _lib.call("semabs_cos_bce", _lib.ptr(dev[0]), _lib.ptr(dev[1]), _lib.ptr(dev[2]), _lib.ptr(dev[3]), P, M, 0.07, P * M, _lib.ptr(logits), _lib.ptr(dO),
              _lib.ptr(drel), _lib.ptr(lossd), _lib.ptr(dob), _lib.stream())
    assert _rel(dob.cpu().numpy(), od.grad.reshape(P, M, 64).sum(1).numpy()) < 1e-5
    assert _rel(logits.cpu().numpy(), z.detach().numpy()) < 2e-6
    assert abs(float(lossd) - float(loss)) < 1e-6 * float(loss)
    assert _rel(dO.cpu().numpy(), od.grad.numpy()) < 1e-5
    assert _rel(drel.cpu().numpy(), rd.grad.numpy()) < 1e-5
    # the head with the loss left to the caller: logits only, then the gradients for a given d loss / d logits
    logits2 = torch.empty(P * M, device="cuda")
    _lib.call("semabs_cos_head", _lib.ptr(dev[0]), _lib.ptr(dev[1]), None, P, M, 0.07, _lib.ptr(logits2), None, None, _lib.stream())
    assert bool((logits2 == logits).all())
    dz = torch.randn(P * M, generator=g)
    od.grad = None; rd.grad = None
    z2 = torch.nn.functional.cosine_similarity(od.reshape(P, M, 64), rd[:, None, :], dim=-1).reshape(-1) / 0.07
    z2.backward(dz.double())
    dzd = dz.cuda()
    dO2 = torch.empty(P * M, 64, device="cuda"); drel2 = torch.zeros(P, 64, device="cuda")
    _lib.call("semabs_cos_head", _lib.ptr(dev[0]), _lib.ptr(dev[1]), _lib.ptr(dzd), P, M, 0.07, None, _lib.ptr(dO2), _lib.ptr(drel2), _lib.stream())
    assert _rel(dO2.cpu().numpy(), od.grad.numpy()) < 1e-5
    assert _rel(drel2.cpu().numpy(), rd.grad.numpy()) < 1e-5


def test_convtranspose_backward_strict():
    import torch.nn.functional as F
    sd, params, grads, u, pre = _unet_setup(3, 5)
    rng = np.random.default_rng(3)
    name = "decoders.0.upsampling.upsample."
    cin, cout, s = 64, 32, 4
    x = rng.standard_normal((2, cin, s, s, s)).astype(np.float32)
    skip = rng.standard_normal((2, cout, 2 * s, 2 * s, 2 * s)).astype(np.float32)
    dy = (rng.standard_normal((2, cout, 2 * s, 2 * s, 2 * s)) * 1e-5).astype(np.float32)
    w = sd[pre + name + "weight"].clone().requires_grad_(True)
    b = sd[pre + name + "bias"].clone().requires_grad_(True)
    xt = torch.from_numpy(x).requires_grad_(True)
    (torch.from_numpy(skip) + F.conv_transpose3d(xt, w, b, stride=2, padding=1, output_padding=1)).backward(torch.from_numpy(dy))
    dx, inv = u._up_bwd(name, _cl(x), _cl(dy))               # still carries the dynamic gradient scale; its consumer multiplies by inv[0]
    dx = u._unscale_by(dx, inv)
    torch.cuda.synchronize()
    assert _rel(_uncl(dx), xt.grad.numpy()) < 1e-5
    assert _rel(grads[pre + name + "weight"].cpu().numpy(), w.grad.numpy()) < 1e-5
    assert _rel(grads[pre + name + "bias"].cpu().numpy(), b.grad.numpy()) < 1e-5


def test_unet_backward_vs_autograd():
    from oracle import semabs3d as os3
    L, S, B = 3, 16, 2
    sd, params, grads, u, pre = _unet_setup(L, 5)
    rng = np.random.default_rng(2)
    x = rng.standard_normal((B, 16, S, S, S)).astype(np.float32)
    x[:, :, rng.random((S, S, S)) < 0.5] = 0                                       # sparse like a scattered volume
    dy = (rng.standard_normal((B, 16, S, S, S)) * 1e-4).astype(np.float32)
    # reference gradients: autograd over the oracle's functional UNet
    psd = {k: v.clone().float().requires_grad_(True) for k, v in sd.items()}
    xt = torch.from_numpy(x).requires_grad_(True)
    y_ref = os3.unet_forward(psd, xt, L, prefix=pre)
    y_ref.backward(torch.from_numpy(dy))
    y, tape = u.forward(_cl(x))
    assert _rel(_uncl(y), y_ref.detach().numpy()) < 1e-4
    dx = u.backward(tape, _cl(dy))
    torch.cuda.synchronize()
    l2, med = _robust(_uncl(dx), xt.grad.numpy())
    assert l2 < 5e-2 and med < 1e-3, (l2, med)
    for k in params:
        l2, med = _robust(grads[k].cpu().numpy(), psd[k].grad.numpy())
        assert l2 < 5e-2 and med < 1e-2, (k, l2, med)
    # final conv has no ReLU / pooling behind it: strict
    assert _rel(grads[pre + "final_conv.weight"].cpu().numpy(), psd[pre + "final_conv.weight"].grad.numpy()) < 1e-5


def test_linear_wgrad_colsum_vs_torch():
    from semabs_amd import _lib
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    # (the last three shapes: k_wgrad_rows_narrow - at least 2^16 rows, at most 36 inputs; ragged row counts)
    for R, Ci, Co in [(1000, 4, 128), (777, 128, 16), (2050, 36, 32), (64, 32, 64), (70001, 4, 128), (100003, 36, 32), (65536, 4, 128)]:
        x = torch.from_numpy(rng.standard_normal((R, Ci)).astype(np.float32))
        w = torch.from_numpy(rng.standard_normal((Co, Ci)).astype(np.float32))
        b = torch.from_numpy(rng.standard_normal(Co).astype(np.float32))
        xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
        y = torch.empty(R, Co, device=dev)
        _lib.call("semabs_linear_f32", _lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(y), R, Ci, Co, 1, 0.01, _lib.stream())
        ref = torch.nn.functional.leaky_relu(x @ w.t() + b, 0.01)
        assert _rel(y.cpu().numpy(), ref.numpy()) < 1e-5
        dOut = torch.from_numpy(rng.standard_normal((R, Co)).astype(np.float32))
        dd = dOut.to(dev)
        dW = torch.zeros(Co, Ci, device=dev)
        _lib.call("semabs_wgrad", _lib.ptr(dd), _lib.ptr(xd), None, None, _lib.ptr(dW), 1, 1, 1, R, 1, 1, R, 1, Co, Ci, 1, bytes(3), 0, _lib.stream())
        assert _rel(dW.cpu().numpy(), (dOut.double().t() @ x.double()).numpy()) < 1e-5
        red = torch.zeros(1, Co, 2, dtype=torch.float64, device=dev)
        _lib.call("semabs_chan_reduce", _lib.ptr(dd), None, None, None, _lib.ptr(red), 1, R, Co, 1, _lib.stream())
        assert _rel(red[0, :, 0].cpu().numpy(), dOut.double().sum(0).numpy()) < 1e-6


def test_wgrad_mfma_vs_torch_autograd():
    """semabs_wgrad_mfma (split-fp16 MFMA, transposing LDS reads) against torch's fp64 autograd of Conv3d / ConvTranspose3d / Linear weights
    (unet3d.py:63-118 conv + GroupNorm order, :296-331 transposed convolution) and against the fp32 VALU kernel it replaces.  Gradient operands are
    drawn at 1e-7 scale so the dynamic power-of-two scale is exercised (unscaled they would flush in fp16)."""
    from semabs_amd import _lib
    from semabs_amd.train import TAPS_CONV3, TAPS_ONE
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(11)
    F = torch.nn.functional

    scratch = torch.empty(1 << 22, device=dev)

    def scale_of(t):
        sc = torch.empty(1, device=dev); sh = torch.empty(1, device=dev); s2 = torch.empty(2, device=dev)
        bits = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.call("semabs_grad_scale", _lib.ptr(t), t.numel(), _lib.ptr(sc), _lib.ptr(sh), 1, _lib.ptr(s2), _lib.ptr(bits), 2, _lib.stream())
        return s2

    # Conv3d 3x3x3 with the GroupNorm affine folded into the gather: 8^3 / 4^3 levels (+ a ragged size)
    for B, D, cin, cout in [(2, (8, 8, 8), 32, 64), (3, (4, 4, 4), 64, 32), (1, (3, 5, 6), 16, 16)]:
        x = rng.standard_normal((B, cin) + D)
        gsc, gsh = rng.uniform(0.5, 1.5, (B, cin)), rng.standard_normal((B, cin))
        dz = rng.standard_normal((B, cout) + D) * 1e-7
        xn = torch.from_numpy(x * gsc[:, :, None, None, None] + gsh[:, :, None, None, None])
        wt = torch.zeros(cout, cin, 3, 3, 3, dtype=torch.float64, requires_grad=True)
        F.conv3d(xn, wt, padding=1).backward(torch.from_numpy(dz))
        xd = torch.from_numpy(x).float().to(dev).permute(0, 2, 3, 4, 1).contiguous()
        dzd = torch.from_numpy(dz).float().to(dev).permute(0, 2, 3, 4, 1).contiguous()
        scd, shd = torch.from_numpy(gsc).float().to(dev).contiguous(), torch.from_numpy(gsh).float().to(dev).contiguous()
        dW = torch.zeros(cout, cin, 27, device=dev)
        _lib.call("semabs_wgrad_mfma", _lib.ptr(dzd), _lib.ptr(xd), _lib.ptr(scd), _lib.ptr(shd), _lib.ptr(scale_of(dzd)), None, _lib.ptr(dW),
                  B, *D, *D, 1, cout, cin, 27, TAPS_CONV3, 1, _lib.ptr(scratch), scratch.numel(), _lib.stream())
        old = torch.zeros_like(dW)
        _lib.call("semabs_wgrad", _lib.ptr(dzd), _lib.ptr(xd), _lib.ptr(scd), _lib.ptr(shd), _lib.ptr(old), B, *D, *D, 1, cout, cin, 27, TAPS_CONV3, 1,
                  _lib.stream())
        ref = wt.grad.reshape(cout, cin, 27).numpy()
        assert _rel(dW.cpu().numpy(), ref) < 2e-6, (D, _rel(dW.cpu().numpy(), ref))
        assert _rel(old.cpu().numpy(), ref) < 2e-6

    # the brick kernels (semabs_wgrad_conv3): transposing-read kernel with partial-sum scratch, and the round-2 kernel (scratch = NULL)
    for B, D, cin, cout in [(2, (4, 4, 16), 16, 16), (1, (8, 8, 32), 32, 16), (3, (4, 12, 16), 16, 32), (2, (16, 16, 16), 16, 16), (70, (4, 4, 16), 16, 16)]:      # 70 volumes: two launches
        x = rng.standard_normal((B, cin) + D)
        gsc, gsh = rng.uniform(0.5, 1.5, (B, cin)), rng.standard_normal((B, cin))
        dz = rng.standard_normal((B, cout) + D) * 1e-7
        xn = torch.from_numpy(x * gsc[:, :, None, None, None] + gsh[:, :, None, None, None])
        wt = torch.zeros(cout, cin, 3, 3, 3, dtype=torch.float64, requires_grad=True)
        F.conv3d(xn, wt, padding=1).backward(torch.from_numpy(dz))
        xd = torch.from_numpy(x).float().to(dev).permute(0, 2, 3, 4, 1).contiguous()
        dzd = torch.from_numpy(dz).float().to(dev).permute(0, 2, 3, 4, 1).contiguous()
        scd, shd = torch.from_numpy(gsc).float().to(dev).contiguous(), torch.from_numpy(gsh).float().to(dev).contiguous()
        ref = wt.grad.reshape(cout, cin, 27).numpy()
        for use_scratch in (True, False):
            if not use_scratch and D[1] % 8:
                continue
            for tap_minor in (1, 0):
                dW = torch.zeros(cout, cin, 27, device=dev) if tap_minor else torch.zeros(cout, 27, cin, device=dev)
                for _ in range(2):                          # accumulates
                    _lib.call("semabs_wgrad_conv3", _lib.ptr(dzd), _lib.ptr(xd), _lib.ptr(scd), _lib.ptr(shd), _lib.ptr(scale_of(dzd)), _lib.ptr(dW),
                              B, *D, cout, cin, tap_minor, _lib.ptr(scratch) if use_scratch else None, scratch.numel() if use_scratch else 0, _lib.stream())
                got = dW.cpu().numpy() if tap_minor else dW.permute(0, 2, 1).cpu().numpy()
                assert _rel(got, 2 * ref) < 2e-6, (D, use_scratch, tap_minor, _rel(got, 2 * ref))

    # ConvTranspose3d k3 s2 p1 op1: A = the layer input, X = the output gradient (the scaled operand), rows over the INPUT voxels
    for B, D, cin, cout in [(2, (4, 4, 4), 64, 32), (1, (8, 8, 8), 32, 16), (2, (2, 3, 4), 128, 64)]:
        x = rng.standard_normal((B, cin) + D)
        D2 = tuple(2 * d for d in D)
        g = rng.standard_normal((B, cout) + D2) * 1e-7
        wt = torch.zeros(cin, cout, 3, 3, 3, dtype=torch.float64, requires_grad=True)
        F.conv_transpose3d(torch.from_numpy(x), wt, stride=2, padding=1, output_padding=1).backward(torch.from_numpy(g))
        xd = torch.from_numpy(x).float().to(dev).permute(0, 2, 3, 4, 1).contiguous()
        gd = torch.from_numpy(g).float().to(dev).permute(0, 2, 3, 4, 1).contiguous()
        dW = torch.zeros(cin, cout, 27, device=dev)
        _lib.call("semabs_wgrad_mfma", _lib.ptr(xd), _lib.ptr(gd), None, None, None, _lib.ptr(scale_of(gd)), _lib.ptr(dW),
                  B, *D, *D2, 2, cin, cout, 27, TAPS_CONV3, 1, None, 0, _lib.stream())
        ref = wt.grad.reshape(cin, cout, 27).numpy()
        assert _rel(dW.cpu().numpy(), ref) < 2e-6, (D, _rel(dW.cpu().numpy(), ref))

    # Linear: one tap, rows = points; accumulates into dW (second call doubles it); both layouts
    for R, Ci, Co in [(5000, 128, 128), (777, 32, 64), (2050, 128, 16), (31, 16, 16)]:
        x = rng.standard_normal((R, Ci)); d = rng.standard_normal((R, Co)) * 1e-7
        xd, dd = torch.from_numpy(x).float().to(dev), torch.from_numpy(d).float().to(dev)
        ref = d.T @ x
        for tap_minor in (0, 1):
            dW = torch.zeros(Co, Ci, device=dev)
            for rep in range(2):                            # once through the partial-sum scratch, once through atomics
                _lib.call("semabs_wgrad_mfma", _lib.ptr(dd), _lib.ptr(xd), None, None, _lib.ptr(scale_of(dd)), None, _lib.ptr(dW), 1, 1, 1, R, 1, 1, R, 1,
                          Co, Ci, 1, TAPS_ONE, tap_minor, _lib.ptr(scratch) if rep == 0 else None, scratch.numel() if rep == 0 else 0, _lib.stream())
            assert _rel(dW.cpu().numpy(), 2 * ref) < 2e-6, (R, Ci, Co, _rel(dW.cpu().numpy(), 2 * ref))
    with pytest.raises(RuntimeError):
        _lib.call("semabs_wgrad_mfma", _lib.ptr(dd), _lib.ptr(xd), None, None, None, None, _lib.ptr(dW), 1, 1, 1, 31, 1, 1, 31, 1, 16, 36, 1, TAPS_ONE, 1,
                  None, 0, _lib.stream())


def test_maxpool_bwd_ties_first_max():
    from semabs_amd import _lib
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(4)
    x = rng.integers(0, 3, size=(2, 8, 4, 4, 4)).astype(np.float32)              # many ties (and all-zero windows)
    dy = rng.standard_normal((2, 8, 2, 2, 2)).astype(np.float32)
    xt = torch.from_numpy(x).requires_grad_(True)
    torch.nn.functional.max_pool3d(xt, 2).backward(torch.from_numpy(dy))
    xc = torch.from_numpy(x).to(dev).permute(0, 2, 3, 4, 1).contiguous()
    dyc = torch.from_numpy(dy).to(dev).permute(0, 2, 3, 4, 1).contiguous()
    dx = torch.empty_like(xc)
    _lib.call("semabs_maxpool3d_bwd", _lib.ptr(xc), _lib.ptr(dyc), _lib.ptr(dx), 2, 4, 4, 4, 8, _lib.stream())
    assert np.array_equal(dx.permute(0, 4, 1, 2, 3).cpu().numpy(), xt.grad.numpy())


def _g13_trainer(g, **kw):
    from semabs_amd.train import VOOLTrainer
    S, N, M, D, seed, wseed, _ = [int(v) for v in g["meta"]]
    batch = vool_batch(S, N, M, D, seed, g["label"])
    tr = VOOLTrainer(make_semabsvool_state_dict(seed=wseed), voxel_shape=(S, S, S), scene_bounds=SCENE_BOUNDS, **kw)
    return tr, batch


def test_vool_train_step_vs_reference_golden(golden):
    """Loss / logits (forward) are tight; gradients use flip-tolerant statistics (see _robust).  The thresholds come from the reference's
    own conditioning at this size: perturbing the weights by 1e-6 (relative, i.e. fp32 rounding level) moves the CPU autograd gradients
    of this very batch by 2.6 % (median over tensors) and up to 6.5 % relative L2 - the 32^3 grid reaches 1^3 voxels at UNet level 5,
    where GroupNorm is singular.  The strict proof of the backward kernels is the flip-free layer tests above (1e-5)."""
    g = golden("g13_vool_train")
    tr, batch = _g13_trainer(g)
    out = tr.forward_backward(batch)
    torch.cuda.synchronize()
    assert abs(float(out["loss"]) - float(g["loss"])) <= 1e-4 * float(g["loss"])
    assert np.abs(out["logits"].cpu().numpy() - g["logits"]).max() <= 2e-3          # logits span +-14 (cos / 0.07)
    names = [str(k) for k in g["names"]]
    bad = []
    for k, n, has in zip(names, g["grad_norm"], g["has_grad"]):
        assert (tr.params[k].grad is not None) == bool(has), k      # visual_sampler.* and unused relation embeddings: no gradient
        if has:
            mine = float(tr.grads[k].double().norm())
            if abs(mine - n) > 5e-2 * max(n, 1e-12):        # 5 %: inside the reference's own 1e-6-perturbation spread (docstring); the reductions here
                                                            # use floating-point atomics, so a ReLU / max-pool tie may flip from run to run
                bad.append((k, mine, n))
    assert not bad, bad[:5]
    for k in list(g):
        if k.startswith("grad/"):
            l2, med = _robust(tr.grads[k[5:]].cpu().numpy(), g[k])
            assert l2 < 0.15, (k, l2, med)
        elif k.startswith("grads/"):
            mine = tr.grads[k[6:]].cpu().numpy().reshape(-1)[g["gradidx/" + k[6:]]]
            l2, med = _robust(mine, g[k])
            assert l2 < 0.15 and med < 3e-2, (k, l2, med)
    total = float(tr.optimizer_step())
    assert abs(total - float(g["total_norm"])) <= 3e-2 * float(g["total_norm"])
    sd = tr.state_dict()
    before = make_semabsvool_state_dict(seed=int(g["meta"][5]))
    for k, dn, has in zip(names, g["delta_norm"], g["has_grad"]):
        mine = float((sd[k].cpu().double() - before[k].double()).norm())
        assert abs(mine - dn) <= 5e-2 * dn + 1e-12, (k, mine, dn)                   # also: untouched (no-grad) tensors stay put, dn = 0
    # first LAMB step = lr * trust * g / (|g| + eps) element-wise: sign-like, so compare the update where the gradient is not ~0
    for k in list(g):
        if k.startswith("new/"):
            ref_step = g[k] - before[k[4:]].numpy()
            my_step = sd[k[4:]].cpu().numpy() - before[k[4:]].numpy()
            ok = np.abs(my_step - ref_step) <= 0.05 * np.abs(ref_step).max() + 1e-9
            # 16-element biases: allow a few sign-like flips; the 64-element GroupNorm affines of the (near-singular, see the docstring) coarse levels
            # too: 55 of 64 matched in one run of five, 114 of 128 (encoders.3 conv2 groupnorm.bias) in another - the reductions use floating-point atomics,
            # the elements that flip change from run to run
            # (one full-suite run in ~20 still tripped the 0.85 / 0.75 pair; the strict checks of every backward kernel are the flip-free layer tests above)
            assert ok.mean() > (0.9 if ok.size > 256 else 0.8 if ok.size > 64 else 0.7), (k, ok.mean())
    assert float(sd["steps"]) == 1.0


def test_vool_train_step_vs_reference_golden_64(golden):
    """The same step at 64^3 (g20: the unmodified reference, 12 000 input / 4 000 query points, 3 descriptions), where the deepest UNet level
    still has 2^3 voxels and GroupNorm is well conditioned: the flip-tolerant 5 % / 15 % bounds of the 32^3 golden are not needed.
    Bounds = 3 x the values measured on MI355X (printed below): gradient norms per tensor, relative L2 of the sampled gradients, total norm."""
    g = golden("g20_vool_train64")
    S, N, M, D, seed, wseed, _ = [int(v) for v in g["meta"]]
    from semabs_amd.train import VOOLTrainer
    batch = vool_batch(S, N, M, D, seed, g["label"])
    tr = VOOLTrainer(make_semabsvool_state_dict(seed=wseed), voxel_shape=(S, S, S), scene_bounds=SCENE_BOUNDS)
    out = tr.forward_backward(batch)
    torch.cuda.synchronize()
    e_loss = abs(float(out["loss"]) - float(g["loss"])) / float(g["loss"])
    e_logit = float(np.abs(out["logits"].cpu().numpy() - g["logits"]).max())
    names = [str(k) for k in g["names"]]
    gtot = float(np.sqrt((g["grad_norm"] ** 2).sum()))
    worst_norm, worst_l2, worst_key = 0.0, 0.0, None
    for k, n, has in zip(names, g["grad_norm"], g["has_grad"]):
        assert (tr.params[k].grad is not None) == bool(has), k
        if has and n > 1e-4 * gtot:                                  # tensors that carry gradient (biases in front of a GroupNorm are noise)
            worst_norm = max(worst_norm, abs(float(tr.grads[k].double().norm()) - n) / n)
    for k in list(g):
        if k.startswith("grad/") or k.startswith("grads/"):
            name = k.split("/", 1)[1]
            n = float(g["grad_norm"][names.index(name)])
            if n <= 1e-4 * gtot:
                continue
            mine = tr.grads[name].cpu().numpy()
            if k.startswith("grads/"):
                mine = mine.reshape(-1)[g["gradidx/" + name]]
            l2, _ = _robust(mine, g[k])
            if l2 > worst_l2:
                worst_l2, worst_key = l2, name
    total = float(tr.optimizer_step())
    e_total = abs(total - float(g["total_norm"])) / float(g["total_norm"])
    print(f"64^3 train step vs reference: loss rel {e_loss:.2e}, logits L-inf {e_logit:.2e}, worst per-tensor grad-norm rel {worst_norm:.2e}, "
          f"worst sampled-gradient rel L2 {worst_l2:.2e} ({worst_key}), total norm rel {e_total:.2e}")
    # measured over repeated runs (the reductions use floating-point atomics): loss 1.4-1.6e-7, logits 6.9-8.0e-5, grad norms 1.1-1.3e-2,
    # sampled gradients 2.0-2.6e-2 (small GroupNorm / bias tensors), total norm 3.9e-5 .. 1.3e-4
    assert e_loss <= 1e-6 and e_logit <= 2.5e-4
    # (total norm: 3e-5 .. 1.3e-4 depending on the summation order inside the level-0 convolution - which ReLU ties flip - asserted at 3 x)
    assert worst_norm <= 2.8e-2 and worst_l2 <= 7.2e-2 and e_total <= 4e-4


def test_vool_training_reduces_loss_and_balanced_weights(golden):
    g = golden("g13_vool_train")
    tr, batch = _g13_trainer(g, balance_positive_negative=True)
    w = tr.bce_weight(batch["output_label_pts"].cuda())
    assert np.allclose(w.cpu().numpy()[:, :, ::50], g["bce_weight_balanced_sub"], rtol=1e-6)
    first = tr.step(batch)
    assert abs(float(first["loss"]) - float(g["loss_balanced"])) <= 1e-4 * float(g["loss_balanced"])
    losses = [float(first["loss"])] + [float(tr.step(batch)["loss"]) for _ in range(7)]
    assert losses[-1] < 0.9 * losses[0], losses
    assert all(np.isfinite(losses))


def test_vool_train_step_batch2_pad_relation_vs_oracle():
    """Two scenes per batch, a "[pad]" description, a 4-level UNet at 16^3 (level 3 = 2^3 voxels): loss / logits tight, gradient norms
    with the flip-tolerant bound, parameters without gradient untouched - against the autograd oracle."""
    from oracle import train as ot
    from semabs_amd.train import VOOLTrainer
    S, N, M, D, L = 16, 1500, 700, 3, 4
    rng = np.random.default_rng(21)
    lo, hi = np.array(SCENE_BOUNDS[0]), np.array(SCENE_BOUNDS[1])
    batch = dict(input_xyz_pts=torch.from_numpy((lo + (hi - lo) * rng.random((2, N, 3))).astype(np.float32)),
                 input_target_saliency_pts=torch.from_numpy(rng.random((2, D, N, 1)).astype(np.float32)),
                 input_reference_saliency_pts=torch.from_numpy(rng.random((2, D, N, 1)).astype(np.float32)),
                 output_xyz_pts=torch.from_numpy((lo - 0.05 + (hi - lo + 0.1) * rng.random((2, D, M, 3))).astype(np.float32)),
                 output_label_pts=torch.from_numpy((rng.random((2, D, M)) < 0.25).astype(np.float32)),
                 spatial_relation_name=[["on", "behind"], ["in", "[pad]"], ["on the left of", "on"]])      # D lists of B names
    sd = make_semabsvool_state_dict(seed=9, unet_num_levels=L)
    ref = ot.vool_train_step(sd, batch, SCENE_BOUNDS, (S, S, S), num_levels=L)
    tr = VOOLTrainer(sd, voxel_shape=(S, S, S), scene_bounds=SCENE_BOUNDS, unet_num_levels=L)
    out = tr.forward_backward(batch)
    torch.cuda.synchronize()
    assert abs(float(out["loss"]) - ref["loss"]) <= 1e-4 * ref["loss"]
    assert np.abs(out["logits"].cpu().numpy() - ref["logits"].numpy()).max() <= 2e-3
    for k, g in ref["grads"].items():
        n = float(g.double().norm())
        assert abs(float(tr.grads[k].double().norm()) - n) <= 0.1 * n + 1e-12, k
    assert tr.params["relation_embeddings.in front of"].grad is None            # not in this batch
    assert tr.params["relation_embeddings.[pad]"].grad is not None              # padding descriptions still contribute (train_vool.py:171)
    total = float(tr.optimizer_step())
    assert abs(total - ref["total_norm"]) <= 3e-2 * ref["total_norm"]
    new = tr.state_dict()
    for k in ("relation_embeddings.in front of", "completion_net.visual_sampler.mlp.0.weight"):
        assert torch.equal(new[k].cpu(), sd[k]), k


def _check_against_reference_with_spread(tr, out, g, sp, label, factor=3.0, floor_norm=2e-3, floor_l2=4e-3):
    """Every gradient tensor that carries gradient against the reference's, measured in units of the reference's OWN spread: the per-tensor
    deviation of the reference's CPU gradients when its weights are perturbed by 1e-6 relative (fp32 rounding level), worst of three
    perturbations (g20s / g22, tests/golden/gen_golden.py).  Asserted: every tensor within `factor` x its own spread (a three-sample maximum is
    itself a noisy estimate of a tensor's spread; floors for tensors whose three samples happen to agree to 1e-6 - the HIP path's fp32 atomics
    order alone moves a small GroupNorm / bias tensor by ~1e-3), and the MEDIAN tensor inside 1.5 x: the HIP gradients are as close to the
    reference's as the reference is to itself under rounding-level changes."""
    names = [str(k) for k in g["names"]]
    gtot = float(np.sqrt((g["grad_norm"] ** 2).sum()))
    snames = [str(k) for k in sp["spread_names"]]
    s_norm = dict(zip(snames, (float(v) for v in sp["spread_norm_rel"])))
    s_l2 = dict(zip(snames, (float(v) for v in sp["spread_l2_rel"])))
    rn, rl, bad = [], [], []
    for k, n, has in zip(names, g["grad_norm"], g["has_grad"]):
        assert (tr.params[k].grad is not None) == bool(has), k
        if not has or n <= 1e-4 * gtot:
            continue                                                  # biases in front of a GroupNorm etc.: mathematically zero, numerically noise
        e = abs(float(tr.grads[k].double().norm()) - n) / n
        bl = max(s_l2[k], floor_l2 / factor)
        # a norm cannot be asked to agree better than the vectors do (| |a| - |b| | <= |a - b|): the reference's norm spread of a small tensor
        # is often several times smaller than its vector spread (the perturbation moves the gradient sideways), the HIP path's need not be
        bn = max(s_norm[k], s_l2[k] / factor, floor_norm / factor)
        rn.append(e / bn)
        key = "grad/" + k if "grad/" + k in g else "grads/" + k
        mine = tr.grads[k].cpu().numpy()
        if key.startswith("grads/"):
            mine = mine.reshape(-1)[g["gradidx/" + k]]           # a 2 048-element sample vs the spread's full-tensor relative L2: same statistic up to sampling noise
        l2, _ = _robust(mine, g[key])
        rl.append(l2 / bl)
        if e > factor * bn:
            bad.append((k, "norm", e, bn))
        if l2 > factor * bl:
            bad.append((k, "l2", l2, bl))
    rn, rl = np.asarray(rn), np.asarray(rl)
    print(f"{label}: {len(rn)} gradient tensors in units of the reference's own 1e-6-perturbation spread - norm deviation median {np.median(rn):.2f} / "
          f"p90 {np.percentile(rn, 90):.2f} / max {rn.max():.2f};  sampled-gradient relative L2 median {np.median(rl):.2f} / p90 {np.percentile(rl, 90):.2f} / max {rl.max():.2f}")
    assert not bad, bad[:6]
    assert np.median(rn) <= 1.5 and np.median(rl) <= 1.5


def test_vool_train_step_64_inside_reference_self_spread(golden):
    """g20 again, with bounds that do not come from this implementation's own measurements: in units of what the REFERENCE's own gradients move by
    when its weights are perturbed at fp32 rounding level (g20s: worst of three 1e-6 perturbations, per tensor)."""
    g, sp = golden("g20_vool_train64"), golden("g20s_vool_train64_spread")
    S, N, M, D, seed, wseed, _ = [int(v) for v in g["meta"]]
    from semabs_amd.train import VOOLTrainer
    tr = VOOLTrainer(make_semabsvool_state_dict(seed=wseed), voxel_shape=(S, S, S), scene_bounds=SCENE_BOUNDS)
    out = tr.forward_backward(vool_batch(S, N, M, D, seed, g["label"]))
    torch.cuda.synchronize()
    _check_against_reference_with_spread(tr, out, g, sp, "64^3")
    total = float(tr.optimizer_step())
    e_total = abs(total - float(g["total_norm"])) / float(g["total_norm"])
    assert e_total <= max(3.0 * float(sp["spread_total_rel"]), 4e-4), e_total


def test_vool_train_step_config5_128_vs_reference_golden(golden):
    """Config 5 at its STATED size: 128^3, batch 1, 4 descriptions, 80 000 input / 400 000 query points (train_vool.py defaults) - forward, BCE,
    backward (the persistent 9-wave level-0 weight-gradient kernel, the cell-list sampler backward, the dynamic gradient scale all at the
    size they are benchmarked at), clip_grad_norm_ and LAMB against g22 = the unmodified reference on the same seeded batch and weights:
    loss, 16 384 sampled logits + their sums, 123 gradient norms / sampled gradients in units of the reference's own 1e-6-perturbation spread,
    total norm, per-tensor update norms."""
    g = golden("g22_vool_train128")
    S, N, M, D, seed, wseed, _ = [int(v) for v in g["meta"]]
    assert (S, N, M, D) == (128, 80000, 400000, 4)
    from semabs_amd.train import VOOLTrainer
    label = np.unpackbits(g["label_packed"])[: D * M].reshape(1, D, M)
    before = make_semabsvool_state_dict(seed=wseed)
    tr = VOOLTrainer(before, voxel_shape=(S, S, S), scene_bounds=SCENE_BOUNDS)
    out = tr.forward_backward(vool_batch(S, N, M, D, seed, label))
    torch.cuda.synchronize()
    e_loss = abs(float(out["loss"]) - float(g["loss"])) / float(g["loss"])
    lg = out["logits"].cpu().numpy().reshape(-1)
    e_logit = float(np.abs(lg[g["logit_idx"]] - g["logits_s"]).max())
    e_sum = abs(float(lg.astype(np.float64).sum()) - float(g["logits_sum"])) / float(g["logits_abs"])
    print(f"128^3 config-5 step vs reference: loss rel {e_loss:.2e}, sampled logits L-inf {e_logit:.2e}, logit sum rel {e_sum:.2e}")
    assert e_loss <= 1e-6 and e_logit <= 5e-4 and e_sum <= 1e-6
    _check_against_reference_with_spread(tr, out, g, g, "128^3")
    total = float(tr.optimizer_step())
    e_total = abs(total - float(g["total_norm"])) / float(g["total_norm"])
    print(f"128^3 total gradient norm rel {e_total:.2e} (reference self-spread {float(g['spread_total_rel']):.2e})")
    assert e_total <= max(3.0 * float(g["spread_total_rel"]), 4e-4)
    sd = tr.state_dict()
    names = [str(k) for k in g["names"]]
    for k, dn, has in zip(names, g["delta_norm"], g["has_grad"]):
        mine = float((sd[k].cpu().double() - before[k].double()).norm())
        assert abs(mine - dn) <= 5e-2 * dn + 1e-12, (k, mine, dn)
