"""Stage-by-stage comparison of the HIP closed-form ViT-B rollout with the oracle's fp32 intermediates on the golden tiles (GPU): x1c / fc / x2c / feat, the kept
softmax row, V, every tensor of the VJP chain, the per-tile relevance, and the rollout kernel alone on the oracle's operands.  How round 6 found that the softmax
row of block 11 carries the error on trained-checkpoint statistics (tests/test_gpu_trained_stats.py).

    python tools/stage_errors.py [ViT-B/32|ViT-B/16] [init|trained] [repeats of the 3 tiles: 1 = small-batch launch sequence, 16 / 4 = the benchmarked one]
"""
import sys, torch, numpy as np
sys.path.insert(0, '.')
import semabs_amd
from oracle import relevancy as orl, preprocess as op
from semabs_amd.synth import synth_rgb
from semabs_amd.weights import make_clip_state_dict
from semabs_amd.clip import ClipWrapper
arch = sys.argv[1] if len(sys.argv) > 1 else "ViT-B/32"
stats = sys.argv[2] if len(sys.argv) > 2 else "trained"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
tag = "b32" if arch.endswith("32") else "b16"
g = dict(np.load(f"tests/golden/g29_vit_{tag}.npz")) if stats == "trained" else dict(np.load(f"tests/golden/g3g4_vit_{tag}.npz"))
sd = make_clip_state_dict(arch, 0, text_tower=False, stats=stats)
sizes = [120, 80, 60, 30, 97]
tiles = torch.from_numpy(np.stack([op.preprocess_tile(synth_rgb(sizes[i % 5], sizes[i % 5], seed=7 + i)) for i in range(3)]))
w_text = torch.from_numpy(g["w_text"])
heads, layers = 12, 12
with torch.no_grad():
    feat, last = orl.vit_forward(sd, tiles)
    pre = f"visual.transformer.resblocks.{layers - 1}."
    n, E = feat.shape
    nrm = feat.norm(dim=-1, keepdim=True); fh = feat / nrm
    wl = w_text.T[:, None, :]
    dfeat = 100.0 * (wl - fh[None] * (fh[None] * wl).sum(-1, keepdim=True)) / nrm[None]
    sc = dfeat.abs().amax(-1, keepdim=True)
    dy = (dfeat / sc) @ sd["visual.proj"].T
    x2c = last["x_final_cls"]
    dx2 = orl._ln_vjp(x2c[None], sd["visual.ln_post.weight"], dy)
    x1c = last["x1"][:, 0, :]; fc = last["fc"][:, 0, :]
    dact = dx2 @ sd[pre + "mlp.c_proj.weight"]
    sg = torch.sigmoid(1.702 * fc)
    dfc = dact * (sg * (1 + 1.702 * fc * (1 - sg)))[None]
    dh2 = dfc @ sd[pre + "mlp.c_fc.weight"]
    g1 = dx2 + orl._ln_vjp(x1c[None], sd[pre + "ln_2.weight"], dh2)
    u = g1 @ sd[pre + "attn.out_proj.weight"]
    L = u.shape[0]
    v = last["v"]; probs = last["probs"][:, :, 0, :]
    grad = torch.einsum("nhjd,lnhd->lnhj", v, u.view(L, n, heads, -1)) * sc[..., None]
    cam = grad * probs[None]
ClipWrapper.engine = None
ClipWrapper(arch, state_dict=sd, chunk_tiles=3 * reps, max_labels=4)
eng = ClipWrapper.engine
wt = w_text.T.contiguous().cuda()
t = tiles.repeat(reps, 1, 1, 1).cuda()
def cmp(name, ours, ref):
    ours = ours.float().cpu(); ref = ref.float()
    e = float((ours - ref).abs().max()); m = float(ref.abs().max())
    print(f"  {name:8s} rel L-inf {e / m:.3e}   max|ref| {m:.3e}  median|ref| {float(ref.abs().median()):.3e}")
for pos in (True, False):
    rel, logits, f = eng.gradcam_tiles(t, wt, pos)
    ws = eng._workspace()
    N = 3 * reps
    print(f"{arch} {stats} reps {reps} pos {pos}")
    T = eng.T; D = eng.D
    cmp("x1c", ws["x1c"][:3], x1c); cmp("fc", ws["fc"][:3], fc); cmp("x2c", ws["x2c"][:3], x2c); cmp("feat", ws["feat"][:3], feat)
    cmp("probs", ws["probs"][:3], probs)
    cmp("v", ws["v16"][: 3 * T].view(3, T, heads, 64).transpose(1, 2), v)
    R = L * N
    pick = lambda a: a[:R].view(L, N, -1)[:, :3]
    cmp("dy", pick(ws["dy"]), dy); cmp("dx2", pick(ws["dx2"]), dx2); cmp("dact", pick(ws["dact"]), dact); cmp("dfc", pick(ws["dfc"]), dfc)
    cmp("dh2", pick(ws["dh2"]), dh2); cmp("g1", pick(ws["g1h"]), g1); cmp("u", pick(ws["u"]), u)
    c = cam.clamp(min=0) if pos else cam
    ref = c.mean(2)[:, :, 1:]
    cmp("rel", rel[:, :3].reshape(L, 3, -1), ref)
    # the rollout kernel alone on the ORACLE's operands
    import ctypes
    from semabs_amd import _lib
    ws["probs"][:3].copy_(probs.cuda()); ws["v16"][: 3 * T].copy_(v.transpose(1, 2).reshape(3 * T, D).half().cuda())
    uu = torch.zeros_like(ws["u"]); uu[:R].view(L, N, D)[:, :3] = u.cuda(); ws["u"].copy_(uu)
    s2 = torch.zeros_like(ws["scale"]); s2[:R].view(L, N)[:, :3] = sc[..., 0].cuda(); ws["scale"].copy_(s2)
    out = torch.zeros_like(rel)
    _lib.call("semabs_rollout", _lib.ptr(ws["probs"]), _lib.ptr(ws["v16"]), _lib.ptr(ws["u"]), _lib.ptr(ws["scale"]), _lib.ptr(out), N, T, heads, L, int(pos), int(out.shape[1]), 0, _lib.stream())
    cmp("rollout kernel on oracle operands", out[:, :3].reshape(L, 3, -1), ref)
