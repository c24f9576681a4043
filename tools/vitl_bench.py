"""SURVEY.md 8 f4 measurement: ViT-L/14 forward + 13-layer rollout on one MI355X.   python tools/vitl_bench.py [tiles_per_chunk] [labels] [chunks]
Prints one JSON line: ms per chunk for the forward (24 blocks, 13 of them keeping their intermediates) and for the backward + rollout
(L x n sequences through blocks 23..12), tiles/s, and the time a 480 x 480 "ours" scene (2 448 tile forwards) would take."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import semabs_amd  # noqa
from semabs_amd.clip import ClipWrapper
from semabs_amd.weights import make_clip_state_dict

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
L = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ClipWrapper.engine = None
ClipWrapper("ViT-L/14", state_dict=make_clip_state_dict("ViT-L/14", 0, text_tower=False), chunk_tiles=n, max_labels=L)
eng = ClipWrapper.engine
G, Kp = eng.g * eng.g, 3 * eng.p * eng.p
patches = (torch.randn(n * G, Kp, device="cuda") * 0.5).half()
w = torch.randn(L, eng.E, device="cuda"); w = w / w.norm(dim=-1, keepdim=True)
rel = torch.zeros(L, n, eng.g, eng.g, device="cuda")
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
fw, bw = [], []
for r in range(reps + 1):
    ev[0].record()
    eng.embed(patches, n); eng.trunk(n); eng.head(n)
    ev[1].record()
    eng.rollout(n, w, True, rel, 0)
    ev[2].record()
    torch.cuda.synchronize()
    if r:
        fw.append(ev[0].elapsed_time(ev[1])); bw.append(ev[1].elapsed_time(ev[2]))
f, b = float(np.median(fw)), float(np.median(bw))
T, D = eng.T, eng.D
fwd_flops = n * (24 * (24 * T * D * D + 4 * T * T * D) + 2 * (T - 1) * Kp * D)
bwd_gemm_flops = L * n * T * 13 * 2 * D * D * (4 + 4 + 1) + L * n * T * 12 * 2 * D * D * 3
bwd_attn_flops = L * n * eng.H * (13 * 2 * 2 * T * T * 64 + 12 * 3 * 2 * T * T * 64)          # S, dP (+ dQ, dK, dV) as dense products
print(json.dumps({"arch": "ViT-L/14", "tiles_per_chunk": n, "labels": L, "forward_ms": f, "backward_rollout_ms": b,
                  "tiles_per_s": n / ((f + b) * 1e-3), "scene_seconds_2448_forwards": 2448 / n * (f + b) * 1e-3,
                  "forward_tflops": fwd_flops / (f * 1e-3) / 1e12, "backward_gemm_tflops_if_alone": bwd_gemm_flops / (b * 1e-3) / 1e12,
                  "backward_algorithmic_tflops": (bwd_gemm_flops + bwd_attn_flops) / (b * 1e-3) / 1e12,
                  "peak_hbm_gb": torch.cuda.max_memory_allocated() / 1e9, "rel_absmax": float(rel.abs().max())}))
