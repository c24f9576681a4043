"""Tuning aid (SEMABS_TUNE_LIB=1 -> libsemabs_hip_tune.so): per-shape TFLOP/s of the ViT trunk GEMMs at the benchmark batch (M = 2448 * 197),
ablations, first-wave stagger sweep, and per-workgroup {start, main-loop end, end} traces.   python tools/gemm_probe.py [what ...]"""
import json, os, sys
os.environ.setdefault("SEMABS_TUNE_LIB", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import semabs_amd  # noqa
from semabs_amd import _lib
from semabs_amd.clip.vit import gemm

M = int(os.environ.get("PROBE_M", 2448 * 197))
SHAPES = [("qkv", 2304, 768, 0), ("out", 768, 768, 2), ("fc", 3072, 768, 1), ("proj", 768, 3072, 2), ("kv32", 768, 768, 3)]
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "gemm_probe")
os.makedirs(OUT, exist_ok=True)
tune = lambda k, v: _lib.call("semabs_gemm_tune", k, v)
bufs = {}


def operands(n, k, epi):
    key = (n, k, epi)
    if key not in bufs:
        g = torch.Generator(device="cuda").manual_seed(n * 7 + k)
        A = torch.randn(M, k, device="cuda", generator=g).half()
        B = (torch.randn(n, k, device="cuda", generator=g) * 0.05).half()
        bias = torch.randn(n, device="cuda", generator=g)
        C = torch.zeros(M, n, device="cuda", dtype=torch.float16 if epi in (0, 1) else torch.float32)
        bufs[key] = (A, B, bias, C)
    return bufs[key]


KERNEL = 0


def time_shape(n, k, epi, reps=6):
    A, B, bias, C = operands(n, k, epi)
    for _ in range(2):
        gemm(A, B, C, bias, M, n, k, k, k, n, epi, kernel=KERNEL)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        gemm(A, B, C, bias, M, n, k, k, k, n, epi, kernel=KERNEL)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return ms, 2.0 * M * n * k / ms / 1e9


def line(label):
    res = {}
    for name, n, k, epi in SHAPES:
        ms, tf = time_shape(n, k, epi)
        res[name] = (round(ms * 1e3, 1), round(tf))
    print(f"{label:34s} " + "  ".join(f"{nm} {v[0]:7.1f}us {v[1]:5d}TF" for nm, v in res.items()), flush=True)
    return res


what = sys.argv[1:] or ["base", "ablate", "trace"]
log = {}
if "datadep" in what:
    # The PRODUCTION kernels on operands of decreasing switching activity: the same instruction stream, the same bytes moved - what changes is the power the
    # matrix pipe draws, i.e. the clock the part holds.  (SEMABS_TUNE_LIB=0 python tools/gemm_probe.py datadep)
    for kind in ("random (the trunk's statistics)", "constant 0.5", "zeros"):
        for key in list(bufs):
            del bufs[key]
        for name, n, k, epi in SHAPES:
            A, B, bias, C = operands(n, k, epi)
            if kind != "random (the trunk's statistics)":
                v = 0.5 if kind.startswith("constant") else 0.0
                A.fill_(v); B.fill_(v * 0.1); bias.fill_(v); C.zero_()
        log["datadep " + kind] = line("operands: " + kind)

if "table" in what:
    # per-shape table of the production configuration + the vendor library's PLAIN fp16 GEMM (torch.matmul -> hipBLASLt, no bias / GELU /
    # residual) as calibration of what a large fp16 GEMM reaches on this box at these shapes
    tab = {}
    for _ in range(2):                                   # clocks up before the first measured shape
        for name, n, k, epi in SHAPES:
            time_shape(n, k, epi, reps=3)
    for name, n, k, epi in SHAPES:
        ms, tf = time_shape(n, k, epi)
        A, B, _, _ = operands(n, k, epi)
        Bt = B.t().contiguous()
        for _ in range(2):
            torch.matmul(A, Bt)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(6):
            torch.matmul(A, Bt)
        e1.record(); torch.cuda.synchronize()
        vms = e0.elapsed_time(e1) / 6
        bytes_alg = M * k * 2 + n * k * 2 + M * n * (2 if epi in (0, 1) else (8 if epi == 2 else 4))
        tab[name] = dict(M=M, N=n, K=k, epilogue=epi, us=round(ms * 1e3, 1), tflops=round(tf, 1), vendor_plain_us=round(vms * 1e3, 1),
                         vendor_plain_tflops=round(2.0 * M * n * k / vms / 1e9, 1), algorithmic_bytes=bytes_alg,
                         hbm_floor_us=round(bytes_alg / 8e6, 1), mfma_floor_us=round(2.0 * M * n * k / 2.5e9, 1))
        print(name, tab[name], flush=True)
        del Bt
    json.dump(tab, open(os.path.join(OUT, "shapes.json"), "w"), indent=1)
    what = [w for w in what if w != "table"]
if "base" in what:
    for rep in range(2):
        for pf in (1, 0):
            tune(3, pf)
            log[f"base_pf{pf}_{rep}"] = line(f"fragment prefetch {pf} (rep {rep})")
    tune(3, 1)
if "ring" in what:
    for rep in range(3):
        for kern, label in ((2, "phased, 4 phases per K = 64"), (2 | 512, "ring, 1 phase per K = 32")):
            KERNEL = kern
            log[f"ring{kern}_{rep}"] = line(f"{label} (rep {rep})")
    KERNEL = 0
if "deep" in what:
    for rep in range(3):
        for kern, label in ((2 | 2048, "phased PF (4 half-tiles in flight)"), (2 | 4096, "deep, one workgroup per tile"), (2, "deep + persistent, prologue in the drain (fp16 out)")):
            KERNEL = kern
            log[f"deep{kern}_{rep}"] = line(f"{label} (rep {rep})")
    KERNEL = 0
if "cfgs" in what:
    for rep in range(2):
        for c in (0, 9, 3, 7, 4, 5):
            tune(0, c)
            log[f"cfg{c}_{rep}"] = line(f"ring-kernel configuration {c} (0 = production) rep {rep}")
    tune(0, 0)
if "persist" in what:
    for rep in range(2):
        for ps in (0, 1):
            tune(5, ps)
            log[f"persist{ps}_{rep}"] = line(f"persistent {ps} (rep {rep})")
    tune(5, 0)
if "phases" in what:
    # per-phase timeline of ONE workgroup in the middle of the grid (tuning build, GEMM8_STAMP): for both wave rows, averaged over the inner K tiles,
    # shader-clock cycles from phase start to: past the SYNC barrier (DMA issue + vmcnt wait + barrier), MFMA block issued, past the END barrier
    for name, n, k, epi in SHAPES[:4]:
        A, B, bias, C = operands(n, k, epi)
        nb = ((M + 255) // 256) * (n // 256)
        tr = torch.zeros(2, 1024, dtype=torch.int64, device="cuda")
        gemm(A, B, C, bias, M, n, k, k, k, n, epi); torch.cuda.synchronize()
        tune(8, nb // 2 + 3); tune(7, tr.data_ptr())
        gemm(A, B, C, bias, M, n, k, k, k, n, epi); torch.cuda.synchronize()
        tune(7, 0)
        t = tr.cpu().numpy().reshape(2, 64, 4, 4)[:, :k // 64]          # [row, K tile, phase, point]
        for row in range(2):
            x = t[row, 1:-2].astype(np.float64)                          # inner K tiles
            sync = (x[:, :, 1] - x[:, :, 0]).mean(0); mma = (x[:, :, 2] - x[:, :, 1]).mean(0); end = (x[:, :, 3] - x[:, :, 2]).mean(0)
            per_kt = (t[row, 2:-1, 0, 0] - t[row, 1:-2, 0, 0]).mean()
            print(f"phases {name} row {row}: cycles per K tile {per_kt:7.0f} | per phase (0-3): to SYNC {np.round(sync)}  MFMA issue {np.round(mma)}  to END {np.round(end)}", flush=True)
        log[f"phases_{name}"] = t.tolist()
if "ablate2" in what:
    # components of the steady-state loop, epilogue always off (bit 3): bit 0 no in-loop DMA, bit 1 no in-loop LDS fragment reads, bit 2 no DMA waits
    SH = SHAPES
    for kern in (2 | 2048, 2):
        KERNEL = kern
        SHAPES = [x for x in SH if x[3] in (0, 2)]            # the instantiated epilogues
        for ab, label in [(0, "full kernel"), (8, "no epilogue"), (9, "no epilogue, no DMA"), (10, "no epilogue, no LDS reads"), (11, "no epilogue, no DMA, no LDS reads"),
                          (12, "no epilogue, no DMA waits"), (15, "MFMA + barriers only")]:
            tune(2, ab)
            log[f"ablate2_{kern}_{ab}"] = line(("PF   " if kern & 2048 else "deep ") + label)
        tune(2, 0)
    SHAPES = SH; KERNEL = 0
if "ablate" in what:
    for ab, label in [(8, "no epilogue"), (16, "no B staging / B reads"), (24, "no B staging / reads, no epilogue"), (9, "no DMA, no epilogue")]:
        tune(2, ab)
        log[f"ablate{ab}"] = line(label)
    tune(2, 0)
if "abltrace" in what:
    # per ablation (compile-time instantiations): wall time AND the in-kernel main-loop clock / cycles per K tile of the QKV and c_proj shapes
    for kern in (2 | 2048, 2):
        for ab, label in [(0, "full kernel"), (8, "no epilogue"), (9, "no epilogue, no DMA"), (10, "no epilogue, no LDS reads"), (12, "no epilogue, no DMA waits"), (15, "MFMA + barriers only")]:
            tune(2, ab)
            for name, n, k, epi in (SHAPES[0], SHAPES[3]):
                A, B, bias, C = operands(n, k, epi)
                nb = ((M + 255) // 256) * (n // 256)
                tr = torch.zeros(nb, 8, dtype=torch.int64, device="cuda")
                for _ in range(4):
                    gemm(A, B, C, bias, M, n, k, k, k, n, epi, kernel=kern)
                torch.cuda.synchronize()
                tune(4, tr.data_ptr())
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    gemm(A, B, C, bias, M, n, k, k, k, n, epi, kernel=kern)
                e1.record(); torch.cuda.synchronize()
                tune(4, 0)
                t = tr.cpu().numpy()
                us = e0.elapsed_time(e1) / 4 * 1e3
                ghz = ((t[:, 5] - t[:, 4]) / np.maximum(1, t[:, 1] - t[:, 0])).mean() * 0.1
                print(f"{'PF  ' if kern & 2048 else 'deep'} {label:28s} {name:5s} {us:7.1f}us {2.0 * M * n * k / us / 1e6:6.0f}TF  main loop {((t[:, 1] - t[:, 0]) / 100.0).mean():6.2f}us = "
                      f"{(t[:, 5] - t[:, 4]).mean() / (k // 64):6.0f} cycles per K tile at {ghz:.3f} GHz; tile end - start {((t[:, 2] - t[:, 0]) / 100.0).mean():6.2f}us", flush=True)
    tune(2, 0)
if "power" in what:
    # ENERGY attribution (tuning build): per compile-time ablation of the deep schedule and per operand kind, the QKV GEMM launched back to back for ~4 s with rocm-smi
    # sampled from a thread: us per launch, socket power, sclk, joules per launch.  (python tools/gemm_probe.py power)
    import re, subprocess, threading, time
    def smi():
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
        pw = re.search(r"Power \(W\): ([0-9.]+)", out); ck = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
        return (float(pw.group(1)) if pw else float("nan"), float(ck.group(1)) if ck else float("nan"))
    name, n, k, epi = SHAPES[0]
    for kind in ("random", "zeros"):
        for key in list(bufs):
            del bufs[key]
        A, B, bias, C = operands(n, k, epi)
        if kind == "zeros":
            A.zero_(); B.zero_(); bias.zero_()
        for ab, label in [(0, "full kernel"), (8, "no epilogue"), (9, "no epilogue, no DMA"), (10, "no epilogue, no LDS reads"), (15, "MFMA + barriers only")]:
            tune(2, ab)
            for _ in range(3):
                gemm(A, B, C, bias, M, n, k, k, k, n, epi, kernel=2 | 4096)
            torch.cuda.synchronize()
            samples, stop = [], threading.Event()
            th = threading.Thread(target=lambda: [samples.append(smi()) or time.sleep(0.3) for _ in iter(lambda: stop.is_set(), True)])
            t0 = time.time(); launches = 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            th.start(); e0.record()
            while time.time() - t0 < 4.0:
                for _ in range(40):
                    gemm(A, B, C, bias, M, n, k, k, k, n, epi, kernel=2 | 4096)
                launches += 40
                torch.cuda.synchronize()
            e1.record(); torch.cuda.synchronize(); stop.set(); th.join()
            us = e0.elapsed_time(e1) / launches * 1e3
            sm = np.array(samples[2:] or samples)
            pw, ck = float(np.nanmean(sm[:, 0])), float(np.nanmean(sm[:, 1]))
            print(f"{kind:7s} {label:28s} {us:7.1f} us  {2.0 * M * n * k / us / 1e6:6.0f} TF  {pw:6.0f} W  sclk {ck:5.0f} MHz  {pw * us * 1e-6:5.2f} J per launch", flush=True)
    tune(2, 0)
if "policy" in what:
    # cache-policy experiment on the persistent fp16-output kernel (tuning build): nt on the C stores / the A stream / the W stream
    SH = SHAPES; SHAPES = [SH[0], SH[2]]; KERNEL = 2
    for rep in range(2):
        for ab, label in [(0, "default policy"), (32, "nt C stores"), (64, "nt A loads"), (128, "nt W loads"), (96, "nt C stores + nt A loads"),
                          (256, "sc0 sc1 C stores"), (512, "nt sc1 C stores"), (1024, "sc0 C stores"), (2048, "sc1 C stores"), (4096, "sc0 nt sc1 C stores")]:
            tune(2, ab)
            log[f"policy{ab}_{rep}"] = line(f"k_gemm8p {label} (rep {rep})")
    tune(2, 0); SHAPES = SH; KERNEL = 0
if "v3trace" in what:
    # persistent cross-tile prefetch (V3) vs one workgroup per tile on the fp16-output shapes: per-tile timeline from the in-kernel stamps
    for kern, label in ((2 | 4096, "deep, one workgroup per tile"), (2, "deep + persistent (k_gemm8p)")):
        for name, n, k, epi in (SHAPES[0], SHAPES[2]):
            A, B, bias, C = operands(n, k, epi)
            nb = ((M + 255) // 256) * (n // 256)
            tr = torch.zeros(nb, 8, dtype=torch.int64, device="cuda")
            for _ in range(4):
                gemm(A, B, C, bias, M, n, k, k, k, n, epi, kernel=kern)
            torch.cuda.synchronize()
            tune(4, tr.data_ptr())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                gemm(A, B, C, bias, M, n, k, k, k, n, epi, kernel=kern)
            e1.record(); torch.cuda.synchronize()
            tune(4, 0)
            t = tr.cpu().numpy()
            us = e0.elapsed_time(e1) / 4 * 1e3
            ghz = ((t[:, 5] - t[:, 4]) / np.maximum(1, t[:, 1] - t[:, 0])).mean() * 0.1
            print(f"{label:32s} {name:5s} {us:7.1f}us {2.0 * M * n * k / us / 1e6:6.0f}TF  tile start -> K loop end {((t[:, 1] - t[:, 0]) / 100.0).mean():6.2f}us = "
                  f"{(t[:, 5] - t[:, 4]).mean() / (k // 64):6.0f} cycles per K tile at {ghz:.3f} GHz; -> epilogue done {((t[:, 2] - t[:, 1]) / 100.0).mean():6.2f}us; first wait + barrier {((t[:, 6] - t[:, 0]) / 100.0).mean() if t[:, 6].any() else float('nan'):6.2f}us; launch / tiles per CU {us / (nb / 256):6.2f}us", flush=True)
if "tracepf" in what:
    KERNEL = 2 | 2048; what = what + ["trace"]
if "trace" in what:
    for name, n, k, epi in SHAPES[:4]:
        A, B, bias, C = operands(n, k, epi)
        nb = ((M + 255) // 256) * (n // 256)
        tr = torch.zeros(nb, 8, dtype=torch.int64, device="cuda")
        gemm(A, B, C, bias, M, n, k, k, k, n, epi, kernel=KERNEL)
        torch.cuda.synchronize()
        tune(4, tr.data_ptr())
        gemm(A, B, C, bias, M, n, k, k, k, n, epi, kernel=KERNEL)
        torch.cuda.synchronize()
        tune(4, 0)
        t = tr.cpu().numpy()
        np.save(os.path.join(OUT, f"trace_{name}.npy"), t)
        t0 = t[:, 0].min()
        start, main, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, (t[:, 2] - t0) / 100.0     # us (100 MHz)
        total = end.max()
        hw = t[:, 3]; hwid = hw & 0xffffffff; xcc = hw >> 32
        cuid = (xcc << 16) | ((hwid >> 8) & 0xf) | (((hwid >> 12) & 1) << 4) | (((hwid >> 13) & 7) << 5)
        gaps = []
        for c in np.unique(cuid):
            idx = np.where(cuid == c)[0]
            o = idx[np.argsort(start[idx])]
            gaps += list(start[o][1:] - end[o][:-1])
        gaps = np.asarray(gaps)
        ghz = ((t[:, 5] - t[:, 4]) / np.maximum(1, t[:, 1] - t[:, 0])).mean() * 0.1
        print(f"trace {name}: clock in the main loop {ghz:.3f} GHz, {(t[:, 5] - t[:, 4]).mean() / (k // 64):.0f} shader cycles per K tile (2048 = matrix pipe back to back)")
        print(f"trace {name}: total {total:.0f}us  tiles {nb}  CUs {len(np.unique(cuid))}  main loop mean {(main - start).mean():.1f}  epilogue (to stores complete) mean {(end - main).mean():.1f}  "
              f"gap to next tile mean {gaps.mean():.2f} p90 {np.percentile(gaps, 90):.2f}  per-tile period {total / (nb / 256):.1f}", flush=True)
json.dump(log, open(os.path.join(OUT, "probe.json"), "w"))
