"""Debug helper: layer-by-layer comparison of the backward kernels against torch-CPU autograd (run on the GPU box)."""
import sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
import semabs_amd  # noqa
from semabs_amd import _lib
from semabs_amd.train import UNetTrainer, TAPS_CONV3
from semabs_amd.weights import make_semabs3d_state_dict
from oracle import semabs3d as os3

dev = torch.device("cuda:0")


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


def cl(x):
    return torch.from_numpy(x).to(dev).permute(0, 2, 3, 4, 1).contiguous()


def uncl(x):
    return x.permute(0, 4, 1, 2, 3).cpu().numpy()


L, S, B = 3, 16, 2
pre = "vol_feature_extractor."
sd = {k: v for k, v in make_semabs3d_state_dict(seed=5, unet_num_levels=L).items() if k.startswith(pre)}
params = {k: v.float().to(dev).contiguous() for k, v in sd.items()}
grads = {k: torch.zeros_like(v) for k, v in params.items()}
u = UNetTrainer(params, grads, pre, 16, 16, 16, 8, L)
u.refresh()
rng = np.random.default_rng(2)

# 1. single GN + conv (no relu) layers
for name, cin, cout, s in [("encoders.0.basic_module.conv1.", 16, 16, 16), ("encoders.1.basic_module.conv1.", 16, 32, 8),
                           ("encoders.2.basic_module.conv2.", 64, 64, 4)]:
    x = rng.standard_normal((B, cin, s, s, s)).astype(np.float32) + 0.3
    dz = rng.standard_normal((B, cout, s, s, s)).astype(np.float32)
    key = pre + name
    w = sd[key + "conv.weight"].clone().requires_grad_(True)
    ga = sd[key + "groupnorm.weight"].clone().requires_grad_(True)
    be = sd[key + "groupnorm.bias"].clone().requires_grad_(True)
    xt = torch.from_numpy(x).requires_grad_(True)
    y = F.conv3d(F.group_norm(xt, 8, ga, be, 1e-5), w, None, padding=1)
    y.backward(torch.from_numpy(dz))
    for k in grads:
        grads[k].zero_()
    r = u._conv_fwd(cl(x), name, False)
    dx = u._conv_bwd(r, cl(dz))
    torch.cuda.synchronize()
    print(name, "fwd", rel(uncl(r.y), y.detach().numpy()), "dx", rel(uncl(dx), xt.grad.numpy()), "dW", rel(grads[key + "conv.weight"].cpu().numpy(), w.grad.numpy()),
          "dgamma", rel(grads[key + "groupnorm.weight"].cpu().numpy(), ga.grad.numpy()), "dbeta", rel(grads[key + "groupnorm.bias"].cpu().numpy(), be.grad.numpy()))
    e = np.abs(uncl(dx) - xt.grad.numpy())
    print("    per-batch dx err", e[0].max(), e[1].max())

# 2. full UNet, per-batch error and error histogram
x = rng.standard_normal((B, 16, S, S, S)).astype(np.float32)
x[:, :, rng.random((S, S, S)) < 0.5] = 0
dy = rng.standard_normal((B, 16, S, S, S)).astype(np.float32)
psd = {k: v.clone().float().requires_grad_(True) for k, v in sd.items()}
xt = torch.from_numpy(x).requires_grad_(True)
taps_ref = {}
y_ref = os3.unet_forward(psd, xt, L, prefix=pre, taps=taps_ref)
for v in taps_ref.values():
    v.retain_grad()
y_ref.backward(torch.from_numpy(dy))
for k in grads:
    grads[k].zero_()
u.debug = []
y, tape = u.forward(cl(x))
dx = u.backward(tape, cl(dy))
torch.cuda.synchronize()
e = np.abs(uncl(dx) - xt.grad.numpy())
mx = np.abs(xt.grad.numpy()).max()
print("unet dx rel", e.max() / mx, "per-batch", e[0].max() / mx, e[1].max() / mx, "frac > 1e-4", (e > 1e-4 * mx).mean(), "frac > 1e-3", (e > 1e-3 * mx).mean())
for k in params:
    print(f"  {k[len(pre):]:55s} {rel(grads[k].cpu().numpy(), psd[k].grad.numpy()):.2e}")

# incoming gradient of every block (= gradient w.r.t. that block's output) against autograd's retained grads
order = [f"dec{i}" for i in reversed(range(L - 1))] + [f"enc{i}" for i in reversed(range(L))]
blocks = [gg for kind, gg in u.debug if kind == "block"]
for name, gg in zip(order, blocks):
    ref = taps_ref[name].grad.numpy()
    print("  d", name, rel(uncl(gg), ref), "fwd", rel(uncl([r for r in tape if r[0] == "block"][0][3].y) if False else 0, 1))
fw = [it for it in tape if it[0] == "block"]
names_f = [f"enc{i}" for i in range(L)] + [f"dec{i}" for i in range(L - 1)]
for name, it in zip(names_f, fw):
    print("  fwd", name, rel(uncl(it[3].y), taps_ref[name].detach().numpy()))

print("---- block-level: encoders.0 on sparse / dense inputs, twice each")
bp = pre + "encoders.0.basic_module."
for tag in ["sparse", "dense", "sparse", "dense"]:
    rs = np.random.default_rng(7)
    xb = rs.standard_normal((B, 16, S, S, S)).astype(np.float32)
    if tag == "sparse":
        xb[:, :, rs.random((S, S, S)) < 0.5] = 0
    dyb = rs.standard_normal((B, 16, S, S, S)).astype(np.float32)
    psd = {k: v.clone().float().requires_grad_(True) for k, v in sd.items()}
    xt = torch.from_numpy(xb).requires_grad_(True)
    o1 = os3._single_conv(psd, bp + "conv1.", xt, 8, True)
    o2 = os3._single_conv(psd, bp + "conv2.", o1, 8, True)
    z3 = os3._single_conv(psd, bp + "conv3.", o2, 8, False)
    yb = F.relu(z3 + o1)
    for t_ in (o1, o2, z3):
        t_.retain_grad()
    yb.backward(torch.from_numpy(dyb))
    for k in grads:
        grads[k].zero_()
    tp = []
    yg = u._block_fwd(cl(xb), "encoders.0.basic_module.", tp)
    r1, r2, r3 = tp[0][1:]
    dS = u._ew(cl(dyb), r3.y, 0)
    d2 = u._conv_bwd(r3, dS)
    dz2 = u._ew(d2, r2.y, 0)
    d1 = u._conv_bwd(r2, dz2, add1=dS)
    dz1 = u._ew(d1, r1.y, 0)
    dxb = u._conv_bwd(r1, dz1)
    torch.cuda.synchronize()
    m1 = ((uncl(r1.y) > 0) != (o1.detach().numpy() > 0)).sum()
    m2 = ((uncl(r2.y) > 0) != (o2.detach().numpy() > 0)).sum()
    m3 = ((uncl(r3.y) > 0) != (yb.detach().numpy() > 0)).sum()
    print(tag, "mask flips", m1, m2, m3, "| d out2", rel(uncl(d2), o2.grad.numpy()), "d out1", rel(uncl(d1), o1.grad.numpy()), "dx", rel(uncl(dxb), xt.grad.numpy()),
          "| near-zero o1", (np.abs(o1.detach().numpy()) < 1e-6).mean() - (o1.detach().numpy() == 0).mean())
