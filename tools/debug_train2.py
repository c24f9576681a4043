import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import semabs_amd  # noqa
from semabs_amd import _lib
from semabs_amd.train import UNetTrainer
from semabs_amd.weights import make_semabs3d_state_dict
from oracle import semabs3d as os3
dev = torch.device("cuda:0")
rel = lambda a, b: np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(np.abs(b).max(), 1e-12)
cl = lambda x: torch.from_numpy(x).to(dev).permute(0, 2, 3, 4, 1).contiguous()
uncl = lambda x: x.permute(0, 4, 1, 2, 3).cpu().numpy()
L, S, B = 3, 16, 2
pre = "vol_feature_extractor."
sd = {k: v for k, v in make_semabs3d_state_dict(seed=5, unet_num_levels=L).items() if k.startswith(pre)}
params = {k: v.float().to(dev).contiguous() for k, v in sd.items()}
grads = {k: torch.zeros_like(v) for k, v in params.items()}
u = UNetTrainer(params, grads, pre, 16, 16, 16, 8, L)
u.refresh()
rng = np.random.default_rng(2)
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
order = [f"dec{i}" for i in reversed(range(L - 1))] + [f"enc{i}" for i in reversed(range(L))]
for trial in range(8):
    _lib.call("semabs_conv_set_config", 1 if trial < 4 else 0)
    for k in grads:
        grads[k].zero_()
    u.debug = []
    y, tape = u.forward(cl(x))
    dx = u.backward(tape, cl(dy))
    torch.cuda.synchronize()
    blocks = [gg for kind, gg in u.debug if kind == "block"]
    errs = " ".join(f"{n}:{rel(uncl(gg), taps_ref[n].grad.numpy()):.1e}" for n, gg in zip(order, blocks))
    wmax = max((rel(grads[k].cpu().numpy(), psd[k].grad.numpy()), k[len(pre):]) for k in params)
    print(f"trial {trial} lds={int(trial < 4)} fwd {rel(uncl(y), y_ref.detach().numpy()):.1e} dx {rel(uncl(dx), xt.grad.numpy()):.1e} | {errs} | worst w {wmax[0]:.1e} {wmax[1]}")

print("---- run-to-run mask differences")
base = None
for trial in range(6):
    y, tape = u.forward(cl(x))
    torch.cuda.synchronize()
    masks = {}
    for it in tape:
        if it[0] == "block":
            for r in it[1:]:
                masks[r.name] = (r.y > 0).cpu().numpy()
                masks[r.name + "val"] = r.y.cpu().numpy()
    if base is None:
        base = masks
        continue
    out = []
    for k in masks:
        if k.endswith("val"):
            continue
        d = (masks[k] != base[k]).sum()
        if d:
            vals = np.abs(base[k + "val"][masks[k] != base[k]]).max()
            out.append(f"{k}:{d} (|y|<={vals:.1e})")
    print("trial", trial, out)
# how close to zero do pre-activations get?  z is not kept, but y = relu(z): smallest positive outputs
for it in tape:
    if it[0] == "block":
        for r in it[1:]:
            yv = r.y.cpu().numpy()
            pos = yv[yv > 0]
            print(f"  {r.name:40s} zeros {np.mean(yv == 0):.3f} min+ {pos.min():.2e}  count(<1e-5) {(pos < 1e-5).sum()}")
