"""Kernel-level A/B of the LayerNorm CONSUMER epilogue (k_gemm8p<EPI, LNC>) against the plain persistent kernel at the benchmark batch (M = 2448 x 197), same operands,
same process, launches interleaved (the bench line showed the consumer QKV at +11 % over block 0's plain QKV; DESIGN said +60 us).

    python tools/ln_consumer_ab.py            -> per shape: plain us, consumer us, difference; in three launch contexts (back to back / behind an attention-sized
                                                 pause / alternating with the other shape)
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import semabs_amd  # noqa
from semabs_amd.clip.vit import gemm, gemm_ln

M = int(os.environ.get("PROBE_M", 2448 * 197))
K = 768
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn(M, K, device="cuda", generator=g).half()
rowac = torch.stack([1.0 + 0.1 * torch.rand(M, device="cuda", generator=g), 0.1 * torch.randn(M, device="cuda", generator=g)], 1).contiguous()
ops = {}
for name, N, epi in (("QKV", 2304, 0), ("c_fc", 3072, 1)):
    B = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half()
    ops[name] = dict(N=N, epi=epi, B=B, bias=torch.randn(N, device="cuda", generator=g), colsum=torch.randn(N, device="cuda", generator=g),
                     C=torch.empty(M, N, dtype=torch.float16, device="cuda"))


def launch(name, consumer, reverse=0):
    o = ops[name]
    if consumer:
        gemm_ln(A, o["B"], o["C"], o["bias"], M, o["N"], K, K, K, o["N"], o["epi"], rowac=rowac, colsum=o["colsum"], reverse=reverse)
    else:
        gemm(A, o["B"], o["C"], o["bias"], M, o["N"], K, K, K, o["N"], o["epi"], kernel=2 | (reverse << 8))


def timed(fn, reps):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return t[len(t) // 2], t[0]


for _ in range(10):
    launch("QKV", False); launch("c_fc", True)
print(f"M = {M}; median (min) us over 40 launches, alternating plain / consumer so both see the same clock state")
for name in ("QKV", "c_fc"):
    res = {False: [], True: []}
    for rep in range(40):
        for consumer in (False, True):
            res[consumer].append(timed(lambda: launch(name, consumer, rep & 1), 1)[0])
    med = {c: sorted(v)[len(v) // 2] for c, v in res.items()}
    mn = {c: min(v) for c, v in res.items()}
    fl = 2.0 * M * ops[name]["N"] * K
    print(f"  {name:5s} plain {med[False]:7.1f} ({mn[False]:7.1f}) us = {fl / med[False] / 1e6:6.0f} TF/s | consumer {med[True]:7.1f} ({mn[True]:7.1f}) us = {fl / med[True] / 1e6:6.0f} TF/s | "
          f"difference {med[True] - med[False]:+6.1f} us ({100 * (med[True] / med[False] - 1):+.1f} %)")
# sustained: 30 back-to-back launches of one variant (the clock settles to the variant's own power)
for name in ("QKV", "c_fc"):
    for consumer in (False, True):
        for _ in range(5):
            launch(name, consumer)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for r in range(30):
            launch(name, consumer, r & 1)
        b.record(); torch.cuda.synchronize()
        print(f"  sustained x30  {name:5s} {'consumer' if consumer else 'plain   '} {a.elapsed_time(b) / 30 * 1e3:7.1f} us")
