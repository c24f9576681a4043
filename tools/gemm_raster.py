"""Raster experiment for k_gemm8 (tuning build): which (group_m, super-column width) keeps an XCD's operand panels in its 4 MiB L2.
   python tools/gemm_raster.py time            per-shape us / TFLOP/s of every raster in CONFIGS (and checks the outputs are identical)
   python tools/gemm_raster.py seq             launches every (shape, raster) exactly once, in the order of `sequence()` (for tools/pmc_seq.py)
"""
import json, os, sys
os.environ.setdefault("SEMABS_TUNE_LIB", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import semabs_amd  # noqa
from semabs_amd import _lib
from semabs_amd.clip.vit import gemm

M = int(os.environ.get("PROBE_M", 2448 * 197))
SHAPES = {"qkv": (2304, 768, 0), "fc": (3072, 768, 1), "proj": (768, 3072, 2), "out": (768, 768, 2), "kv32": (768, 768, 3)}
CONFIGS = {
    "qkv": [(8, 0), (4, 0), (2, 0), (1, 0), (1, 3), (2, 3), (4, 3), (8, 3), (1, 5), (2, 5), (16, 3), (64, 3)],
    "fc": [(8, 0), (4, 0), (1, 0), (1, 6), (2, 6), (1, 4), (2, 4), (4, 4), (1, 3), (8, 4), (16, 4), (64, 4)],
    "proj": [(8, 0), (4, 0), (2, 0), (1, 0), (16, 0)],
    "out": [(8, 0), (4, 0), (2, 0), (1, 0), (16, 0)],
    "kv32": [(8, 0), (1, 0)],
}
tune = lambda k, v: _lib.call("semabs_gemm_tune", k, v)


def sequence():
    return [(name, gm, sc) for name in SHAPES for (gm, sc) in CONFIGS[name]]


def operands(name):
    n, k, epi = SHAPES[name]
    g = torch.Generator(device="cuda").manual_seed(n * 7 + k)
    A = torch.randn(M, k, device="cuda", generator=g).half()
    B = (torch.randn(n, k, device="cuda", generator=g) * 0.05).half()
    bias = torch.randn(n, device="cuda", generator=g)
    C = torch.zeros(M, n, device="cuda", dtype=torch.float16 if epi in (0, 1) else torch.float32)
    return A, B, bias, C


def run(name, ops, gm, sc):
    n, k, epi = SHAPES[name]
    A, B, bias, C = ops
    tune(1, gm); tune(6, sc)
    gemm(A, B, C, bias, M, n, k, k, k, n, epi, kernel=2)


mode = sys.argv[1] if len(sys.argv) > 1 else "time"
out = {}
for name in SHAPES:
    ops = operands(name)
    n, k, epi = SHAPES[name]
    if mode == "seq":
        for gm, sc in CONFIGS[name]:
            if epi == 2:
                ops[3].zero_()
            run(name, ops, gm, sc)
        torch.cuda.synchronize()
        del ops
        continue
    ref = None
    for rep in range(2):
        for gm, sc in CONFIGS[name]:
            if epi == 2:
                ops[3].zero_()
            run(name, ops, gm, sc)
            if rep == 0:
                if ref is None:
                    ref = ops[3].clone()
                else:
                    assert torch.equal(ref, ops[3]), f"raster {gm},{sc} changed the result of {name}"
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6):
                run(name, ops, gm, sc)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 6
            out[f"{name}_g{gm}_s{sc}_r{rep}"] = dict(us=round(ms * 1e3, 1), tflops=round(2.0 * M * n * k / ms / 1e9))
            print(f"{name:5s} group_m {gm:3d} sc_w {sc}  rep {rep}: {ms * 1e3:8.1f} us  {2.0 * M * n * k / ms / 1e9:6.0f} TFLOP/s", flush=True)
    del ops, ref
tune(1, 0); tune(6, 0)
if mode == "time":
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "gemm_raster")
    os.makedirs(d, exist_ok=True)
    json.dump(out, open(os.path.join(d, "time.json"), "w"), indent=1)
