#!/usr/bin/env python
"""bench.py — scenes/s of the relevancy -> fusion -> OVSSC hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is ONE synthetic scene end to end on one GPU: 480x480 RGB-D, 16 labels, ViT-B/16, the "ours" saliency config
(204 tiles x 6 images (5 colour-jitter copies) x 2 flips = 2 448 ViT forwards), analytic attention x gradient rollout,
multi-scale aggregation, depth -> points -> voxel indices, point MLP, scatter-mean, 6-level ResidualUNet3D on 16 label
volumes of 128^3, implicit decoder at the 128^3 voxel centres, TSDF integration and the OVSSC post-mask.  Inputs
(uint8 images, fp32 depth, weights) are resident in HBM before the timed region.  Ranks process disjoint scenes
(scene sharding, no data-path collective) -> weak scaling; value = all scenes / max-over-ranks time.

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (dominant kernel = the fp16 MFMA GEMM, timed
live inside the timed region with HIP events carried by the launches' own dispatch packets) and `cpu_baseline` (the oracle on the host cores, bounded
sample, N = 1).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import semabs_amd  # noqa: E402,F401

ARCH = "ViT-B/16"
N_LABELS = 16
IMG = 480
VOXEL = 128
FLOPS_PER_TILE = {"ViT-B/16": 35.127e9, "ViT-B/32": 8.818e9}      # SURVEY.md §8d per-tile forward count
PEAK_F16_TFLOPS = 2500.0                                          # dense MFMA peak (MI355X_MICROARCH.md)


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


def _median3(fn, reps=3):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def cpu_baseline(arch, n_labels, tiles_sample=64):
    """Oracle ("port" of the reference's CPU path, oracle/) timed on this host's cores: EVERY stage of one scene on a bounded sample of the
    same workload (same synthetic scene family, "ours" config, same arch / label count / grid), median of 3 per stage, scaled to one scene.
    The oracle restates the reference's arithmetic (torch-CPU fp32, analytic rollout instead of L autograd passes - i.e. it is FASTER than the
    reference's own CPU path, which took 720 s for the relevancy stage of this scene on the 8 cores of the build container)."""
    from oracle import geometry as og
    from oracle import relevancy as orl
    from oracle import scene as osc
    from oracle import semabs3d as os3
    from semabs_amd.synth import SCENE_BOUNDS, synth_scene
    from semabs_amd.weights import make_clip_state_dict, make_semabs3d_state_dict
    sd = make_clip_state_dict(arch, 0, text_tower=False)
    # thread count: all logical CPUs is NOT the fastest setting for torch-CPU at these sizes (256 threads on the 2 x 64-core host ran the ViT
    # 15 x slower than 64) - calibrate on one batch of 16 tiles (the oracle runs batches of 32) and keep the fastest of {1/2 (= the physical
    # cores), 1/4, 1/8, 1/16 of the logical CPUs, >= 8}
    ncpu = os.cpu_count() or 1
    cal_tiles = torch.zeros(16, 3, 224, 224)
    cal_w = torch.zeros(512, n_labels)
    best = (None, float("inf"))
    for t in sorted({max(8, ncpu // d) for d in (2, 4, 8, 16)} | {min(ncpu, 8)}):
        if t > ncpu:
            continue
        torch.set_num_threads(t)
        from oracle import relevancy as _orl
        with torch.no_grad():
            _orl.gradcam_tiles(sd, cal_tiles[:1], cal_w, True)
            t0 = time.perf_counter()
            _orl.gradcam_tiles(sd, cal_tiles, cal_w, True)
            dt = time.perf_counter() - t0
        if dt < best[1]:
            best = (t, dt)
    threads = best[0]
    torch.set_num_threads(threads)
    nsd = make_semabs3d_state_dict(seed=3)
    rng = np.random.default_rng(0)
    w = torch.from_numpy(rng.standard_normal((512, n_labels)).astype(np.float32))
    sc = synth_scene(IMG, IMG, seed=1)
    cfg = orl.saliency_configs["ours"](IMG)
    table = orl.tile_table(IMG, IMG, 1, cfg["cropping_augmentations"])                    # 204 tiles of one image
    pick = table[np.linspace(0, len(table) - 1, tiles_sample).astype(int)]
    st = {}
    st["tiles_preprocess"] = _median3(lambda: orl.make_tile_images([sc["rgb"]], pick)) / tiles_sample * 1224
    tiles = orl.make_tile_images([sc["rgb"]], pick)
    with torch.no_grad():
        orl.gradcam_tiles(sd, tiles[:2], w, True)                                         # warm-up
        st["vit_rollout"] = _median3(lambda: [orl.gradcam_tiles(sd, tiles[i:i + 32], w, True) for i in range(0, tiles_sample, 32)]) / tiles_sample * 2448
        # aggregation of one image's tiles (both flip passes already averaged), x 6 images
        rel = torch.from_numpy(rng.standard_normal((n_labels, len(table), 14 if "16" in arch else 7, 14 if "16" in arch else 7)).astype(np.float32) * 1e-3)
        st["aggregate"] = _median3(lambda: orl.aggregate(rel, table, IMG, IMG, tile_sizes=[a["tile_size"] for a in cfg["cropping_augmentations"]]), reps=1) * 6
        # geometry: unprojection, bounds, voxel indices, frustum of the lattice, TSDF
        def geometry():
            pts = og.get_pointcloud(sc["depth"], sc["cam_intr"], sc["cam_pose"]).astype(np.float32)
            m = og.filter_pts_bounds(pts, np.asarray(SCENE_BOUNDS, np.float64))
            og.flatten_idxs(og.points_grid_idxs(pts[m], SCENE_BOUNDS, (VOXEL,) * 3), (VOXEL,) * 3)
        st["geometry"] = _median3(geometry)
        q = osc.grid_points(SCENE_BOUNDS, VOXEL)
        def tsdf_frustum():
            lo, hi = np.asarray(SCENE_BOUNDS[0], np.float64), np.asarray(SCENE_BOUNDS[1], np.float64)
            tv = og.TSDFVolume(np.stack([lo, hi], axis=1), (hi[0] - lo[0]) / VOXEL)
            tv.integrate(sc["rgb"], sc["depth"], sc["cam_intr"], sc["cam_pose"])
            og.check_pts_in_frustum(q.astype(np.float64), (IMG, IMG), sc["cam_pose"], sc["cam_intr"])
        st["tsdf_frustum"] = _median3(tsdf_frustum, reps=1)
        # voxel stage for ONE label volume, x n_labels: point MLP + scatter-mean, UNet, decoder at the 128^3 lattice
        pts = og.get_pointcloud(sc["depth"], sc["cam_intr"], sc["cam_pose"]).astype(np.float32)
        pts = pts[og.filter_pts_bounds(pts, np.asarray(SCENE_BOUNDS, np.float64))]
        xyz = torch.from_numpy(pts[rng.integers(0, len(pts), size=80000)])[None]
        feat = torch.from_numpy(rng.standard_normal((1, 80000, 1)).astype(np.float32))
        st["point_mlp_scatter"] = _median3(lambda: os3.scatter_mean(xyz, os3.point_mlp(nsd, xyz, feat), SCENE_BOUNDS, (VOXEL,) * 3)) * n_labels
        vol = os3.scatter_mean(xyz, os3.point_mlp(nsd, xyz, feat), SCENE_BOUNDS, (VOXEL,) * 3)
        feats = [None]
        def unet():
            feats[0] = os3.unet_forward(nsd, vol, 6)
        st["unet"] = _median3(unet) * n_labels
        st["decoder"] = _median3(lambda: os3.decoder(nsd, feats[0], torch.from_numpy(q)[None], SCENE_BOUNDS, (VOXEL,) * 3, True), reps=1) * n_labels
    scene_s = float(sum(st.values()))
    return {"value": 1.0 / scene_s, "unit": "scenes/s", "cores": threads, "logical_cpus": ncpu, "cpu_model": _cpu_model(), "kind": "port",
            "sample": f"per stage, median of 3, scaled to one scene: {tiles_sample} of 2448 tile forwards + {tiles_sample} of 1224 tile "
                      f"preprocessings ({arch}, {n_labels} labels, analytic rollout), 1 of 6 images' aggregation, full geometry / TSDF / frustum, "
                      f"1 of {n_labels} label volumes through point MLP + scatter, the 128^3 UNet and the decoder; torch-CPU fp32 oracle",
            "scene_seconds_estimate": scene_s, "stage_seconds_per_scene": {k: round(v, 3) for k, v in st.items()},
            "reference_cpu_measured": "the unmodified reference's get_clip_saliency for this scene shape (6 images, 2448 forwards): 720 s on the 8 "
                                      "cores of the build container (tests/golden/g16_headline_aug5.npz: seconds, cores)"}


def parity_report(pipe, arch, precision):
    """Outside the timed region, rank 0 / N = 1: the benchmarked kernels against the committed REFERENCE goldens (tests/golden/: outputs of the
    unmodified reference run in the build container) - the headline relevancy maps on the benchmarked workload itself (6 images, 2 448 tile
    forwards, one ViT batch), the 128^3 UNet of the voxel stage, and the bit-exact voxel indices."""
    import hashlib
    from semabs_amd.clip import ClipWrapper, saliency_configs
    from semabs_amd.synth import SCENE_BOUNDS, synth_jitter, synth_rgb, synth_scene
    gdir = os.path.join(ROOT, "tests", "golden")
    out = {"source": "tests/golden g16 (get_clip_saliency 480x480 / 16 labels / ViT-B/16 'ours', 5 injected augmentations), g10 (ResidualUNet3D 128^3), "
                     "g8 (get_pointcloud + VirtualGrid indices 480x480 / 128^3) - produced by the unmodified reference"}
    if arch == "ViT-B/16":
        g = np.load(os.path.join(gdir, "g16_headline_aug5.npz"))
        img = synth_rgb(IMG, IMG, seed=0)
        cfg = saliency_configs["ours"](IMG)
        images = ClipWrapper.make_images(img, cfg["augmentations"], jittered_images=[synth_jitter(img, k) for k in range(cfg["augmentations"])])
        maps = ClipWrapper.relevancy_device(images, torch.from_numpy(g["text"]).cuda().contiguous(), cfg["cropping_augmentations"],
                                            cfg["horizontal_flipping"], cfg["positive_attn_only"]).cpu().numpy()
        err = max(float(np.abs(maps[:, ::4, ::4] - g["sub"]).max()), float(np.abs(maps[:, g["rows_idx"], :] - g["rows"]).max()))
        out["relevancy_map_abs_linf"] = err
        out["relevancy_map_rel_linf"] = err / float(g["absmax"].max())
    g = np.load(os.path.join(gdir, "g10_unet128.npz"))
    rng = np.random.default_rng(int(g["meta"][0]))
    x = np.zeros((1, 16, 128, 128, 128), np.float32)
    occ = rng.random((128, 128, 128)) < 0.03
    x[0][:, occ] = rng.standard_normal((16, int(occ.sum()))).astype(np.float32)
    if VOXEL == 128 and int(g["meta"][1]) == 3:                     # the pipeline's UNet carries exactly the golden's weights (net seed 3)
        y = pipe.net.vol_feature_extractor.forward(torch.from_numpy(x)).cpu().numpy()
        out["unet128_feature_abs_linf"] = float(np.abs(y.reshape(-1)[g["si"]] - g["y_s"]).max())
        out["unet128_feature_rel_linf"] = out["unet128_feature_abs_linf"] / float(np.abs(g["y_s"]).max())
        out["unet_precision"] = precision
    # the text tower, reported separately: it runs once per label SET (ClipWrapper.set_classes), not per scene; token ids from the committed
    # fixture (the BPE merge table is third-party data the GPU box does not have)
    tk = os.path.join(gdir, "tokens_default.npz")
    if os.path.exists(tk):
        from semabs_amd.clip.vit import TextEncoder
        from semabs_amd.weights import make_clip_state_dict
        t = np.load(tk)
        enc = TextEncoder(make_clip_state_dict(arch, 0, text_tower=True))
        tokens = torch.from_numpy(t["tokens"][:N_LABELS])
        enc.zeroshot_weights(tokens, N_LABELS, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            enc.zeroshot_weights(tokens, N_LABELS, 1)
        torch.cuda.synchronize()
        out["text_tower_ms_per_label_set"] = (time.perf_counter() - t0) / 5 * 1e3
        del enc
    g = np.load(os.path.join(gdir, "g8_geometry.npz"))
    from semabs_amd.point_cloud import pointcloud_device
    sc = synth_scene(480, 480, seed=5)
    xyz, _ = pointcloud_device(torch.from_numpy(sc["depth"]).cuda(), sc["cam_intr"], sc["cam_pose"], np.array(SCENE_BOUNDS))
    flat = pipe.net.vg.flat_idxs(xyz).cpu().numpy().astype(np.int64)
    sha = np.frombuffer(hashlib.sha256(np.ascontiguousarray(flat).tobytes()).digest(), dtype=np.uint8)
    out["voxel_indices_bit_exact"] = bool(VOXEL == 128 and np.array_equal(sha, g["480_flat_sha"]))
    return out



class SmiSampler:
    """Shader clock and socket power of THIS rank's GPU sampled on a side thread while the timed region runs (the power-cap
    explanation of the GEMM fraction must be visible in the driver-run line, not only in builder-run logs).  Sources, first that works: the amdsmi
    python binding (amdsmi_get_power_info / amdsmi_get_clock_info GFX / amdsmi_get_power_cap_info), the amdgpu hwmon files (power1_average |
    power1_input, power1_cap, freq1_input), `rocm-smi --showpower --showclocks` (slow: ~1 sample / s).  Reads cost ~0.1-1 ms of host time each and
    take no GPU time; the period is 50 ms."""

    def __init__(self, device_index=0, period_s=0.05):
        import threading
        self.period = period_s
        self.samples = []                                   # (t, sclk_mhz | None, power_w | None)
        self.cap_w = None
        self.source = None
        self._stop = threading.Event()
        self._thread = None
        self._read = None
        bdf = None
        try:
            pr = torch.cuda.get_device_properties(device_index)
            bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
        except Exception:
            bdf = None
        for mk in (self._mk_amdsmi, self._mk_hwmon, self._mk_cli):
            try:
                rd = mk(device_index, bdf)
                if rd is not None and rd() != (None, None):
                    self._read = rd
                    break
            except Exception:
                continue

    # -- sources --
    def _mk_amdsmi(self, idx, bdf):
        import amdsmi
        amdsmi.amdsmi_init()
        hs = amdsmi.amdsmi_get_processor_handles()
        h = None
        if bdf:
            for c in hs:
                try:
                    if amdsmi.amdsmi_get_gpu_device_bdf(c).lower().startswith(bdf):
                        h = c
                        break
                except Exception:
                    pass
        if h is None:
            h = hs[idx if idx < len(hs) else 0]
        try:
            ci = amdsmi.amdsmi_get_power_cap_info(h)
            cap = float(ci.get("power_cap", 0))
            self.cap_w = cap / 1e6 if cap > 1e5 else (cap if cap > 0 else None)      # microwatts in this binding; watts in older ones
        except Exception:
            pass
        num = lambda v: float(v) if isinstance(v, (int, float)) else None

        def rd():
            w = f = None
            try:
                pi = amdsmi.amdsmi_get_power_info(h)
                for k in ("current_socket_power", "average_socket_power", "socket_power"):
                    v = num(pi.get(k))
                    if v is not None and v > 0:
                        w = v
                        break
                if self.cap_w is None and num(pi.get("power_limit")):
                    pl = num(pi["power_limit"])
                    self.cap_w = pl / 1e6 if pl > 1e5 else pl
            except Exception:
                pass
            try:
                ck = amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)
                for k in ("clk", "cur_clk", "current_clk"):
                    v = num(ck.get(k))
                    if v is not None and v > 0:
                        f = v
                        break
            except Exception:
                pass
            return f, w
        self.source = "amdsmi (amdsmi_get_clock_info GFX, amdsmi_get_power_info socket power, amdsmi_get_power_cap_info)"
        return rd

    def _mk_hwmon(self, idx, bdf):
        import glob
        cands = []
        for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            dev = os.path.realpath(os.path.join(hw, "..", ".."))
            if any(os.path.exists(os.path.join(hw, f)) for f in ("power1_average", "power1_input")):
                cands.append((hw, dev))
        if not cands:
            return None
        pick = next((c for c in cands if bdf and bdf in c[1].lower()), cands[idx if idx < len(cands) else 0])
        hw = pick[0]
        pf = next(f for f in ("power1_average", "power1_input") if os.path.exists(os.path.join(hw, f)))
        rdnum = lambda f: float(open(os.path.join(hw, f)).read().strip())
        try:
            self.cap_w = rdnum("power1_cap") / 1e6
        except Exception:
            pass

        def rd():
            w = f = None
            try:
                w = rdnum(pf) / 1e6
            except Exception:
                pass
            try:
                f = rdnum("freq1_input") / 1e6
            except Exception:
                pass
            return f, w
        self.source = f"amdgpu hwmon ({pf}, freq1_input, power1_cap)"
        return rd

    def _mk_cli(self, idx, bdf):
        import re
        import shutil
        import subprocess
        exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
        if not os.path.exists(exe):
            return None

        def rd():
            w = f = None
            try:
                o = subprocess.run([exe, "-d", str(idx), "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True, timeout=10).stdout
                m = re.search(r"Socket Graphics Package Power \(W\):\s*([0-9.]+)", o) or re.search(r"Power \(W\):\s*([0-9.]+)", o)
                w = float(m.group(1)) if m else None
                m = re.search(r"sclk clock level:.*?\((\d+)Mhz\)", o)
                f = float(m.group(1)) if m else None
                m = re.search(r"Max Graphics Package Power \(W\):\s*([0-9.]+)", o)
                if m and self.cap_w is None:
                    self.cap_w = float(m.group(1))
            except Exception:
                pass
            return f, w
        self.period = max(self.period, 0.5)
        self.source = "rocm-smi --showpower --showclocks (subprocess, ~1 sample / s)"
        return rd

    # -- control --
    def start(self):
        import threading
        if self._read is None:
            return
        self.samples = []
        self._stop.clear()

        def loop():
            while not self._stop.is_set():
                f, w = self._read()
                self.samples.append((time.perf_counter(), f, w))
                self._stop.wait(self.period)
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()

    def stop(self):
        if self._thread is not None:
            self._stop.set()
            self._thread.join(timeout=15)
            self._thread = None

    def summary(self):
        """Fields merged into `roofline`: always present (None when no source answered on this box)."""
        fs = [s[1] for s in self.samples if s[1]]
        ws = [s[2] for s in self.samples if s[2]]
        med = lambda v: float(np.median(v)) if v else None
        return {"sclk_mhz_under_load": med(fs), "sclk_mhz_min_max": [float(min(fs)), float(max(fs))] if fs else None,
                "power_w": med(ws), "power_w_max": float(max(ws)) if ws else None, "power_cap_w": self.cap_w,
                "smi_samples": len(self.samples), "smi_source": self.source,
                "smi_note": "medians over the timed region (side thread, 50 ms period, this rank's GPU); the whole scene runs in it, not only the GEMMs: "
                            "the UNet / attention / LayerNorm phases draw less than the GEMM phases, so the GEMM-only figures are at or above these"}


HBM_PEAK_TBS = 8.0                                                # HBM3E spec (MI355X_MICROARCH.md); ~6.3 TB/s is what a copy reaches
HBM_COPY_TBS = 6.3                                                # what a float4 copy kernel reaches on this part (same guide)
EPI_NAMES = {0: "bias -> fp16", 1: "bias + QuickGELU -> fp16", 2: "bias + fp32 residual read-modify-write", 3: "bias -> fp32", 4: "row-remapped fp32", 5: "row-table multiply -> fp16"}




# GEMMs of the last block and the VJP chain whose A operand is an [hi | lo] pair (K = 2 x the layer's width; clip/vit.py head_split): algorithmic flops = half the issued ones
SPLIT_GEMMS = {(768, 1536, 3): "last-block K / CLS q / W_o^T g1", (768, 1536, 2): "CLS out-proj", (3072, 1536, 3): "CLS c_fc / W_pr^T dx2", (768, 6144, 3): "W_fc^T dfc",
               (768, 6144, 2): "CLS c_proj", (512, 1536, 3): "CLS feature projection", (768, 1024, 3): "proj^T dfeat"}


def gemm_shape_name(N, K, epi, M=None):
    """The GEMM shapes of the ViT-B trunk by role (width D = 768; reference: CLIP/clip/auxiliary.py:129,340, model_explainability.py:210-217)."""
    role = {(2304, 768, 0): "QKV", (768, 768, 2): "out-proj", (3072, 768, 1): "c_fc", (768, 3072, 2): "c_proj", (1536, 768, 3): "K|V (last block)",
            (768, 768, 4): "patch embedding", **{k: v + ", [hi | lo] A operand" for k, v in SPLIT_GEMMS.items()}}.get((int(N), int(K), int(epi)))
    small = M is not None and M < 2048
    return f"{role or ('small' if small else 'other')} N={int(N)} K={int(K)} ({EPI_NAMES.get(int(epi), epi)})"


def gemm_roof_ms(fl, by):
    """Per-launch roof: max(flops / dense fp16 MFMA peak, algorithmic bytes / HBM peak) in ms, and which one binds."""
    t_mfma, t_hbm = fl / (PEAK_F16_TFLOPS * 1e12) * 1e3, by / (HBM_PEAK_TBS * 1e12) * 1e3
    return max(t_mfma, t_hbm), ("mfma" if t_mfma >= t_hbm else "hbm"), t_mfma, t_hbm


def _classify(name, a):
    """C-ABI entry point + its arguments -> (kernel class, algorithmic flops, algorithmic bytes) of that launch; None = small / host-side."""
    if name == "semabs_gemm_f16_ln":
        M, N, K, epi = a[4], a[5], a[6], a[10]
        return ("fp16 GEMM: " + gemm_shape_name(N, K, epi, M) + (" + LayerNorm producer (fp16 x*gamma copy, row partials)" if a[11] else " + LayerNorm consumer"),
                2.0 * M * N * K, M * K * 2 + N * K * 2 + M * N * (2 if epi in (0, 1) else 8) + (M * N * 2 if a[11] else 0))
    if name.startswith("semabs_gemm_f16"):
        M, N, K, epi = a[5], a[6], a[7], a[11]
        return ("fp16 GEMM: " + gemm_shape_name(N, K, epi, M), (1.0 if (int(N), int(K), int(epi)) in SPLIT_GEMMS else 2.0) * M * N * K,
                M * K * 2 + N * K * 2 + M * N * (2 if epi in (0, 1, 5) else (8 if epi == 2 else 4)))
    if name == "semabs_attention":
        n, T, H = a[3], a[4], a[5]
        return "attention (k_attention)", 4.0 * n * H * T * T * 64, n * T * H * 64 * 2 * 4
    if name in ("semabs_layernorm", "semabs_add_layernorm"):
        M, D = a[4 if name == "semabs_layernorm" else 5], a[5 if name == "semabs_layernorm" else 6]
        of32 = (a[7] & 1) if name == "semabs_layernorm" else 0      # bit 0 = fp32 output; bits 1-2 carry the zigzag row order (vit.layernorm)
        return "LayerNorm (k_layernorm)", 0.0, M * D * (4 + (4 if of32 else 2))
    if name in ("semabs_conv3d", "semabs_conv3d_stats"):
        B, D0, D1, D2, cin, cout, k, flags, resid = a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[16], a[7]
        vox, eb = B * D0 * D1 * D2, 4 if (flags & 1) else 2
        fl, by = 2.0 * k ** 3 * cin * cout * vox, vox * (cin + cout * (2 if resid else 1)) * eb
        iss = 3.0 if (flags & 1) else 1.0                   # exact mode = fp16 hi / lo split: three MFMA products per algorithmic product
        if k == 3 and cin == 16 and cout == 16 and D0 % 8 == 0 and D1 % 8 == 0 and D2 % 16 == 0:
            return "Conv3d 128^3 16->16 (k_conv16_lds)", fl, by, iss
        if k == 3 and cout % 32 == 0 and D0 % 4 == 0 and D1 % 8 == 0 and (D2 % 16 == 0 or (D2 % 8 == 0 and cin % 32 == 0)):
            return "Conv3d 64^3..8^3 (k_conv_brick)", fl, by, iss
        return "Conv3d 4^3 / 1x1x1 (k_conv gather)", fl, by, iss
    if name in ("semabs_convtranspose3d", "semabs_convtranspose3d_stats"):
        B, D0, D1, D2, cin, cout, flags = a[7], a[8], a[9], a[10], a[11], a[12], a[13]
        vox, eb = B * D0 * D1 * D2, 4 if (flags & 1) else 2
        return "ConvTranspose3d (k_convT_brick / gather)", 2.0 * 27 * cin * cout * vox, vox * cin * eb + 8 * vox * cout * eb * 2, (3.0 if (flags & 1) else 1.0)
    if name == "semabs_decoder":
        P, M, f32 = a[10], a[11], a[13]
        S = a[4]
        vox = int(S[0]) * int(S[1]) * int(S[2])
        return "implicit decoder (k_decoder)", 2.0 * P * M * (19 * 16 + 16), P * vox * 16 * (4 if f32 else 2) + P * M * 4
    if name in ("semabs_gn_stats", "semabs_maxpool3d", "semabs_scatter_mean", "semabs_scatter_mean_stats", "semabs_point_mlp"):
        return {"semabs_gn_stats": "GroupNorm statistics", "semabs_maxpool3d": "max-pool", "semabs_point_mlp": "point MLP"}.get(name, "scatter-mean"), 0.0, 0
    if name in ("semabs_tile_patches", "semabs_aggregate", "semabs_color_jitter", "semabs_rollout", "semabs_attention_cls"):
        return "tiling / aggregation / rollout / CLS attention", 0.0, 0
    return "other (geometry, TSDF, small element-wise)", 0.0, 0


class _CallTimer:
    """_lib.CALL_HOOK: one torch event pair around every C-ABI launch of a profiled scene (outside the timed region)."""

    def __init__(self):
        self.rec = []

    def before(self, name, args):
        cls = _classify(name, args)
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        return cls, e0

    def after(self, tok):
        cls, e0 = tok
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.rec.append((cls, e0, e1))

    def table(self, scenes):
        torch.cuda.synchronize()
        agg = {}
        for c, e0, e1 in self.rec:
            cls, fl, by = c[0], c[1], c[2]
            d = agg.setdefault(cls, dict(ms=0.0, launches=0, flops=0.0, bytes=0.0, issued=0.0, roof=0.0))
            d["ms"] += e0.elapsed_time(e1); d["launches"] += 1; d["flops"] += fl; d["bytes"] += by; d["issued"] += fl * (c[3] if len(c) > 3 else 1.0)
            d["roof"] += gemm_roof_ms(fl, by)[0]            # roofs are per LAUNCH (a class can mix MFMA- and HBM-bound launches) and summed
        out = {}
        for cls, d in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
            ms = d["ms"] / scenes
            row = {"ms_per_scene": round(ms, 3), "launches_per_scene": d["launches"] // scenes}
            if d["flops"] or d["bytes"]:
                fl, by = d["flops"] / scenes, d["bytes"] / scenes
                t_mfma, t_hbm = fl / (PEAK_F16_TFLOPS * 1e12) * 1e3, by / (HBM_PEAK_TBS * 1e12) * 1e3
                row.update(algorithmic_tflop=round(fl / 1e12, 3), algorithmic_gb=round(by / 1e9, 2), tflops=round(fl / ms / 1e9, 1) if fl else None,
                           tb_per_s=round(by / ms / 1e9, 2) if by else None, bound="mfma" if t_mfma >= t_hbm else "hbm",
                           roof_ms=round(d["roof"] / scenes, 3), frac_of_roof=round(d["roof"] / scenes / ms, 3))
                if d["issued"] > d["flops"]:                 # exact-mode convolutions: the roof of the arithmetic actually chosen (3 MFMA products per product)
                    t_iss = d["issued"] / scenes / (PEAK_F16_TFLOPS * 1e12) * 1e3
                    row.update(issued_mfma_tflop=round(d["issued"] / scenes / 1e12, 3), issued_roof_ms=round(max(t_iss, t_hbm), 3),
                               frac_of_issued_roof=round(max(t_iss, t_hbm) / ms, 3))
            out[cls] = row
        # every fp16 GEMM launch together (the class the headline `roofline` object reports), roof = the per-launch roofs summed
        g = [d for cls, d in agg.items() if cls.startswith("fp16 GEMM")]
        if g:
            ms = sum(d["ms"] for d in g) / scenes
            fl, by, rf = sum(d["flops"] for d in g) / scenes, sum(d["bytes"] for d in g) / scenes, sum(d["roof"] for d in g) / scenes
            out["fp16 GEMM: all shapes"] = {"ms_per_scene": round(ms, 3), "launches_per_scene": sum(d["launches"] for d in g) // scenes, "algorithmic_tflop": round(fl / 1e12, 3),
                                            "algorithmic_gb": round(by / 1e9, 2), "tflops": round(fl / ms / 1e9, 1), "tb_per_s": round(by / ms / 1e9, 2),
                                            "roof_ms": round(rf, 3), "frac_of_roof": round(rf / ms, 3), "mfma_only_roof_ms": round(fl / (PEAK_F16_TFLOPS * 1e12) * 1e3, 3),
                                            "note": "sum of the rows above; roof_ms = sum over launches of max(flops / 2.5 PF, bytes / 8 TB/s): the out-proj and K|V launches are HBM-bound"}
        return out


def stage_report(pipe, scenes, w_text, arch, precision):
    """Driver-witnessed numbers for the other BASELINE configs and a per-kernel-class table, measured by this process OUTSIDE the headline timed
    region (N = 1): config 2 (relevancy stage alone, one 480^2 / 16-label image), config 3 (ResidualUNet3D forward on [16, 16, 128^3], exact and
    fp16), config 5 (one VOOL optimisation step at 128^3 / 4 descriptions / 80 000 + 400 000 points), and event-timed ms per scene of every
    kernel class with its algorithmic flops / bytes (conv flops are the reference's 2 * 27 * Cin * Cout per voxel - the exact mode issues 3 x
    that on the matrix pipe) against max(flops / 2.5 PF, bytes / 8 TB/s)."""
    from semabs_amd import _lib
    from semabs_amd.clip import ClipWrapper, saliency_configs
    cfg = saliency_configs["ours"](IMG)
    ev = lambda: torch.cuda.Event(enable_timing=True)

    def timed(fn, reps=3):
        fn()
        ts = []
        for _ in range(reps):
            a, b = ev(), ev()
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        return float(np.median(ts))

    out = {}
    sc = scenes[0]
    def relevancy():
        images = ClipWrapper.make_images(sc["rgb"], cfg["augmentations"], img_dev=sc.get("rgb_dev"), seed=0)
        ClipWrapper.relevancy_device(images, w_text, cfg["cropping_augmentations"], cfg["horizontal_flipping"], cfg["positive_attn_only"])
    out["relevancy_ms"] = round(timed(relevancy), 2)
    out["relevancy_config"] = f"config 2: {IMG}x{IMG}, {N_LABELS} labels, {arch}, 'ours' (2448 tile forwards): colour jitter + tiling + ViT + rollout + aggregation"
    # ---- kernel classes over one full scene ----
    ct = _CallTimer()
    pipe.run(sc, w_text, seed=0)
    torch.cuda.synchronize()
    _lib.CALL_HOOK = ct
    try:
        for i in range(2):
            pipe.run(scenes[i % len(scenes)], w_text, seed=i)
    finally:
        _lib.CALL_HOOK = None
    out["kernel_classes"] = ct.table(2)
    kc = out["kernel_classes"]
    trunk = [v for k, v in kc.items() if (k.startswith("fp16 GEMM: ") and k != "fp16 GEMM: all shapes") or k.startswith("LayerNorm") or k.startswith("attention")]
    if trunk:
        fl = sum(v.get("algorithmic_tflop") or 0.0 for v in trunk); gb = sum(v.get("algorithmic_gb") or 0.0 for v in trunk); ms = sum(v["ms_per_scene"] for v in trunk)
        counted = None
        for name in ("r06_gemm_pmc.json", "r05_gemm_pmc.json"):
            pth = os.path.join(ROOT, "profiles", name)
            if os.path.exists(pth):
                try:
                    counted = {"gb_per_scene": json.load(open(pth)).get("trunk_counted_gb_per_scene"), "source": f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE, imported; not measured in this run)"}
                except Exception:
                    counted = None
                if counted and counted["gb_per_scene"]:
                    break
        out["trunk_roof"] = {"what": "ViT trunk = every fp16 GEMM + LayerNorm + attention launch of a scene (the relevancy stage minus tiling / rollout / aggregation)",
                             "ms_per_scene": round(ms, 2), "algorithmic_tflop": round(fl, 2), "algorithmic_gb": round(gb, 1),
                             "mfma_floor_ms": round(fl / PEAK_F16_TFLOPS * 1e3, 2), "hbm_floor_ms_at_8_tbs": round(gb / HBM_PEAK_TBS, 2), "hbm_floor_ms_at_6p3_tbs": round(gb / HBM_COPY_TBS, 2),
                             "bound": "hbm" if gb / HBM_PEAK_TBS > fl / PEAK_F16_TFLOPS * 1e3 else "mfma",
                             "frac_of_roof": round(max(fl / PEAK_F16_TFLOPS * 1e3, gb / HBM_PEAK_TBS) / ms, 3), "counted_traffic": counted,
                             "note": "the trunk as designed moves about as many bytes as it has flops to hide them behind: both floors are stated; the fp32 residual stream "
                                     "(read-modify-write in out-proj / c_proj, read by both LayerNorms) is ~40 % of a block's bytes"}
    out["kernel_classes_note"] = ("torch event pairs around every C-ABI launch of 2 profiled scenes (events between launches add ~1-3 % to short kernels); "
                                  "roof = max(algorithmic flops / 2.5 PF dense fp16 MFMA, algorithmic bytes / 8 TB/s)")
    # ---- config 3: the UNet alone ----
    net = pipe.net
    u = net.vol_feature_extractor
    x = torch.zeros(N_LABELS, VOXEL, VOXEL, VOXEL, 16, dtype=torch.float32, device="cuda")
    occ = torch.rand(VOXEL, VOXEL, VOXEL, device="cuda") < 0.03
    x[:, occ] = torch.randn(N_LABELS, int(occ.sum()), 16, device="cuda")
    out["unet128x16_ms_" + precision] = round(timed(lambda: u.forward_cl(x.to(u.act_dtype))), 2)
    from semabs_amd.unet3d import ResidualUNet3D
    other = "fp16" if precision == "exact" else "exact"
    u2 = ResidualUNet3D(in_channels=16, out_channels=16, f_maps=16, num_groups=8, num_levels=6, precision=other)
    u2.load_state_dict(u.state_dict())
    u2.to("cuda")
    out["unet128x16_ms_" + other] = round(timed(lambda: u2.forward_cl(x.to(u2.act_dtype))), 2)
    out["unet_config"] = f"config 3: ResidualUNet3D forward, {N_LABELS} volumes of 16 x {VOXEL}^3, channels-last, incl. the final 1x1x1 convolution"
    del u2, x
    torch.cuda.empty_cache()
    # ---- f1 / f5 at the size the reference runs them (visualize.py:163-164, 360-361): 240^3 sampling lattice, 2^20 query points per pass ----
    from semabs_amd.inference import process_batch_ovssc, process_batch_vool
    from semabs_amd.net import SemAbsVOOL
    from semabs_amd.synth import SCENE_BOUNDS as _SB
    from semabs_amd.weights import make_semabsvool_state_dict as _mk_vool
    rngv = np.random.default_rng(7)
    pts = torch.from_numpy(rngv.uniform([-0.95, -0.95, -0.05], [0.95, 0.95, 1.85], size=(150000, 3)).astype(np.float32))
    classes = [f"class {i}" for i in range(N_LABELS)]
    ob = {"ovssc_obj_classes": classes, "input_xyz_pts": pts, "input_feature_pts": torch.from_numpy(rngv.standard_normal((N_LABELS, len(pts))).astype(np.float32)),
          "rgb": sc["rgb"], "depth": sc["depth"], "cam_intr": sc["cam_intr"], "cam_extr": sc["cam_pose"]}
    SQ, PASS = 240, 2 ** 20
    chunks = -(-SQ ** 3 // PASS)
    out["process_batch_ovssc_240_ms"] = round(timed(lambda: process_batch_ovssc(net, ob, _SB, "cuda", 80000, sampling_shape=(SQ,) * 3, num_pts_per_pass=PASS, seed=0), reps=2), 1)
    if VOXEL == 128:
        vool = SemAbsVOOL(pointing_method="cosine_sim", pointing_dim=64, device="cuda", decoder_concat_xyz_pts=True, voxel_shape=(VOXEL,) * 3, scene_bounds=_SB,
                          unet_num_channels=16, unet_f_maps=16, unet_num_groups=8, unet_num_levels=6, network_inputs=["saliency"], use_pts_feat_extractor=True,
                          pts_feat_extractor_hidden_dim=128, reduce_method="max", output_dim=1, batch_size=1)
        vool.load_state_dict(_mk_vool(seed=3))
        vool.eval()
        ND = 4
        vb = {"descriptions": [f"the thing {i} on the other thing" for i in range(ND)], "spatial_relation_name": ["on", "behind", "in", "on the left of"], "input_xyz_pts": pts,
              "input_target_saliency_pts": torch.from_numpy(rngv.standard_normal((ND, len(pts))).astype(np.float32)),
              "input_reference_saliency_pts": torch.from_numpy(rngv.standard_normal((ND, len(pts))).astype(np.float32))}
        out["process_batch_vool_240_ms"] = round(timed(lambda: process_batch_vool(vool, vb, _SB, "cuda", 80000, sampling_shape=(SQ,) * 3, num_pts_per_pass=PASS, seed=0), reps=2), 1)
        del vool
    out["inference_240_config"] = (f"f1 / f5: process_batch_ovssc ({N_LABELS} classes) and process_batch_vool (4 descriptions) as visualize.py calls them: sampling_shape {SQ}^3 "
                                   f"= {SQ ** 3} lattice points in {chunks} passes of 2^20, 80000 input points, {VOXEL}^3 feature volumes; incl. TSDF integration at {SQ}^3, frustum test, "
                                   "post-mask and the host copy of the result.  UNet passes: the reference re-runs point MLP + scatter + UNet for every pass - "
                                   f"{N_LABELS * chunks} (OVSSC) / {2 * 4 * chunks} (VOOL) - here the feature volumes are computed once: {N_LABELS} / {2 * 4}")
    torch.cuda.empty_cache()
    # ---- config 5: one VOOL optimisation step ----
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from train_bench import synth_batch
    from semabs_amd.synth import SCENE_BOUNDS
    from semabs_amd.train import VOOLTrainer
    from semabs_amd.weights import make_semabsvool_state_dict
    tr = VOOLTrainer(make_semabsvool_state_dict(seed=3), voxel_shape=(VOXEL,) * 3, scene_bounds=SCENE_BOUNDS)
    batch = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in synth_batch(VOXEL, 80000, 400000, 4, seed=0).items()}
    out["vool_train_step_ms"] = round(timed(lambda: tr.step(batch), reps=3), 2)
    out["vool_train_config"] = f"config 5: SemAbsVOOL {VOXEL}^3, batch 1, 4 descriptions, 80000 input / 400000 query points: forward + BCE + backward + clip + LAMB, 1 GPU"
    if VOXEL == 128:
        # roof of the step: the UNet on 8 volumes (2 per description) dominates.  Forward 0.34081 TFLOP / 0.838 G activation elements per volume
        # (tools/unet_bench.py); the backward pass is a data-gradient and a weight-gradient convolution per layer (2 x the forward flops) and
        # ~3 x its tensor passes (dZ read twice, X read for the weight gradient and twice by the GroupNorm backward, dXn written and read
        # twice, dX written).  Exact mode issues 3 MFMAs per product.
        vols, f_tf, f_gb = 8, 0.34081, 0.838 * 4
        alg_tf, issued_tf, gb = 3 * vols * f_tf, 9 * vols * f_tf, 4 * vols * f_gb
        mfma_ms, hbm_ms = issued_tf / PEAK_F16_TFLOPS * 1e3, gb / (HBM_PEAK_TBS * 1e3) * 1e3
        out["vool_train_roof"] = {"algorithmic_tflop": round(alg_tf, 2), "issued_mfma_tflop": round(issued_tf, 2), "compulsory_gb_estimate": round(gb, 1),
                                  "mfma_floor_ms": round(mfma_ms, 2), "hbm_floor_ms": round(hbm_ms, 2), "bound": "hbm" if hbm_ms > mfma_ms else "mfma",
                                  "frac_of_roof": round(max(mfma_ms, hbm_ms) / out["vool_train_step_ms"], 3),
                                  "note": "UNet forward + backward on 8 volumes only (point MLPs, sampler, head, optimiser are < 10 % of the step); bytes = 4 x the forward's compulsory activation traffic"}
    del tr
    torch.cuda.empty_cache()
    return out


def train_workload(args, rank, world, dist):
    """--workload train: config 5 data-parallel.  Every rank takes one VOOL optimisation step per step on ITS OWN scene (weak scaling): forward + BCE +
    backward, ONE flat sum all-reduce of the gradients (+ the per-relation "used" flags) over RCCL, clip, LAMB - `VOOLTrainer.step`.  value = samples / s
    over all ranks; the all-reduce alone is timed separately on the same buffer."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from train_bench import synth_batch
    from semabs_amd import dist as sdist
    from semabs_amd.synth import SCENE_BOUNDS
    from semabs_amd.train import VOOLTrainer
    from semabs_amd.weights import make_semabsvool_state_dict
    tr = VOOLTrainer(make_semabsvool_state_dict(seed=3), voxel_shape=(VOXEL,) * 3, scene_bounds=SCENE_BOUNDS)
    batches = [{k: (v.cuda() if torch.is_tensor(v) else v) for k, v in synth_batch(VOXEL, 80000, 400000, 4, seed=1000 * rank + i).items()} for i in range(2)]
    for i in range(args.warmup):
        tr.step(batches[i % 2])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    sdist.reset_stats()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = tr.step(batches[i % 2])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    ar_ms = None
    per_rank = None
    if dist is not None:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {"rank": rank, "steps": args.steps, "collectives": sdist.stats_snapshot()})
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        sdist_t = t.cpu() if args.backend == "gloo" else t
        dist.all_reduce(sdist_t, op=dist.ReduceOp.MAX)
        dt = float(sdist_t.item())
        buf = tr.flat_grad.clone()
        sdist.allreduce_flat_gradients(buf, 0); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            sdist.allreduce_flat_gradients(buf, 0)
        torch.cuda.synchronize()
        ar_ms = (time.perf_counter() - t1) / 3 * 1e3
    # every rank must hold the same parameters after the step
    import hashlib
    dig = int.from_bytes(hashlib.sha256(torch.cat([p.detach().reshape(-1) for p in tr.params.values()]).cpu().numpy().tobytes()).digest()[:7], "big")
    same = True
    if dist is not None:
        allsum = sdist.gather_results(torch.tensor([dig], dtype=torch.int64, device="cuda")).cpu().numpy()
        same = bool((allsum == allsum[0]).all())
    if rank == 0:
        print(json.dumps({
            "metric": "VOOL optimisation steps (samples)/sec, 128^3 x 4 descriptions, 80000 / 400000 points, data-parallel (config 5)",
            "value": args.steps * world / dt, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16 hi/lo-split MFMA operands, fp32 accumulate (fp32-equivalent)", "data": "synthetic",
            "config": {"workload": f"SemAbsVOOL {VOXEL}^3 training step per rank: forward + BCE + backward + flat gradient all-reduce + clip + LAMB", "parallelism": f"data-parallel x{world}",
                       "backend": args.backend if world > 1 else None},
            "collectives": {"allreduce_bytes_per_rank_per_step": int(tr.flat_grad.numel() * 4), "allreduce_ms_alone": ar_ms, "parameters_identical_across_ranks": same,
                            "overlapped_with_backward": bool(tr.overlap_allreduce), "buckets_bytes": [int((b - a) * 4) for a, b in tr.buckets.ranges],
                            "exposed_ms_per_step": None if per_rank is None else max(
                                (r["collectives"].get("all_reduce_wait", {}).get("device_ms", 0.0) + r["collectives"].get("all_reduce", {}).get("device_ms", 0.0)) / args.steps for r in per_rank),
                            "exposed_ms_note": "device time the COMPUTE stream spent waiting for the gradient all-reduce (HIP events around the waits of BucketedAllReduce.finish / "
                                               "around the single blocking all-reduce), per step, maximum over the ranks: the communication the backward pass did not hide",
                            "per_rank": per_rank,
                            "per_rank_note": "payload bytes, HOST-side and DEVICE-side milliseconds of every collective each rank issued inside the timed region, by kind: "
                                             "all_reduce_bucket = the asynchronous bucket launches (host time = launch cost), all_reduce_wait = making the compute "
                                             "stream wait for them (device_ms = exposed communication)"},
            "loss": float(out["loss"])}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--arch", default=ARCH)
    ap.add_argument("--precision", default="exact", choices=["exact", "fp16"], help="UNet arithmetic (see DESIGN.md)")
    ap.add_argument("--chunk", type=int, default=2448, help="tile forwards per ViT batch.  2448 = the whole scene in one batch (11 GB of activations of "
                    "the 288 GB): the GEMMs run 22 / 66 / 88 waves of tiles deep, so launch, prologue and last-wave effects are amortised - "
                    "901 vs 853 TFLOP/s and 136.7 vs 142.7 ms per scene against 220 (which was tuned to have no ragged last wave: 1.99 / "
                    "5.98 / 7.97 waves); 663 / 1224: 139 ms.  The maps are bit-identical for every chunk size (tools/chunk_equiv.py)")
    ap.add_argument("--streams", type=int, default=1, help="HIP streams the tile chunks are pipelined over when --chunk cuts the scene into several "
                    "batches (2 = +4%% scenes/s at --chunk 220, but overlapping kernels blur the per-launch timing of the roofline leg)")
    ap.add_argument("--time-every", type=int, default=1, help="roofline leg: attach timing events to one GEMM launch in n of the timed region (chosen by a "
                    "hash of the launch index).  1 = every launch: free with one ViT batch per scene (~60 launches); with small batches "
                    "(--chunk 220: 660 launches per scene) it costs 1.9 %% of scenes/s - every dispatch packet then carries a completion signal - "
                    "and 5 gives the same TFLOP/s figure")
    ap.add_argument("--mode", default="throughput", choices=["throughput", "latency"], help="throughput (default, the driver's contract): ranks own disjoint scenes, "
                    "weak scaling.  latency: every step is ONE scene split over all ranks (ScenePipeline.run_sharded: tile-sharded relevancy + one all-gather of "
                    "the per-tile relevances, label-sharded voxel inference + one all-gather of the logits) - strong scaling; the JSON carries the bytes each "
                    "rank put on the wire and a cross-rank checksum of the maps and the labels")
    ap.add_argument("--workload", default="scene", choices=["scene", "train"], help="scene (default): the headline relevancy -> fusion -> OVSSC step.  train: config 5, "
                    "one data-parallel VOOL optimisation step per step (every rank its own 128^3 scene, 4 descriptions, 80 000 / 400 000 points; ONE flat "
                    "all-reduce of the gradients, /root/reference/utils.py:255-258) - samples/s over all ranks, all-reduce bytes and time in the JSON")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend for --gpus > 1: nccl = RCCL over xGMI, one rank per GPU.  gloo: the "
                    "ranks may share one device (contract tests on a 1-GPU box; device collectives are staged through the host)")
    ap.add_argument("--cu-split", default="0", help="throughput mode: N[:layout] = run scene i's voxel stage on a stream masked to N CUs concurrently with scene "
                    "i + 1's relevancy stage on the remaining CUs (semabs_amd.scene.CuPartition; layout balanced | low).  0 (default) = one stream, stages in sequence")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity leg (reference goldens, outside the timed region)")
    ap.add_argument("--no-stages", action="store_true", help="skip the stage / kernel-class leg (configs 2, 3, 5 and the per-class table, outside the timed region)")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (MI355X); there is no CPU fallback")
    if args.gpus > 1 and "RANK" not in os.environ:
        # Plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, RCCL), exactly as the driver would
        # (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...), and hand back its exit code.
        if torch.cuda.device_count() < args.gpus and args.backend != "gloo":
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {torch.cuda.device_count()} HIP device(s) visible")
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} (or run plainly and let bench.py spawn the ranks)")
    if local >= torch.cuda.device_count() and args.backend != "gloo":
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but only {torch.cuda.device_count()} HIP device(s) visible")
    local = local % torch.cuda.device_count()                                # (gloo: ranks may share a device)
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    if args.workload == "train":
        return train_workload(args, rank, world, dist)

    from semabs_amd.clip import vit as vitmod
    from semabs_amd.scene import build_default
    from semabs_amd.synth import synth_scene
    pipe = build_default(args.arch, precision=args.precision, chunk_tiles=args.chunk, max_labels=N_LABELS, voxel=VOXEL, text_tower=False)
    rng = np.random.default_rng(0)
    w = rng.standard_normal((N_LABELS, 512)).astype(np.float32)          # synthetic unit-norm zero-shot text weights
    w /= np.linalg.norm(w, axis=1, keepdims=True)
    w_text = torch.from_numpy(w).cuda()
    n_scenes = args.steps + args.warmup
    from semabs_amd.clip import ClipWrapper, saliency_configs
    ClipWrapper.n_streams = max(1, args.streams)
    cfg = saliency_configs["ours"](IMG)
    scenes = [pipe.upload(synth_scene(IMG, IMG, seed=1000 * rank + i)) for i in range(n_scenes)]      # RGB-D frames resident in HBM
    # The reference encodes the label set on EVERY get_clip_saliency call (CLIP/clip/__init__.py:116-117 -> zeroshot_classifier), so the text tower is
    # part of a step: token ids of 16 labels from the committed fixture (the BPE merge table is third-party data the GPU box does not have), random-init
    # text weights of the benchmarked architecture, resident in HBM.  The per-label weights it produces drive the rollout of that scene.
    text_enc, text_tokens = None, None
    tk = os.path.join(ROOT, "tests", "golden", "tokens_default.npz")
    if os.path.exists(tk):
        from semabs_amd.clip.vit import TextEncoder
        from semabs_amd.weights import make_clip_state_dict
        text_enc = TextEncoder(make_clip_state_dict(args.arch, 0, text_tower=True))
        text_tokens = torch.from_numpy(np.load(tk)["tokens"][:N_LABELS]).to(torch.int64).cuda()     # the label set's token ids, resident in HBM like the frame

    latency = args.mode == "latency"
    if latency:                                          # every rank holds the SAME scenes: one scene per step, split over the ranks
        scenes = [pipe.upload(synth_scene(IMG, IMG, seed=i)) for i in range(n_scenes)]

    def step(i):                                        # everything from the raw frame on is inside the timed region (incl. the colour jitter and the text tower)
        w = text_enc.zeroshot_weights(text_tokens, N_LABELS, 1) if text_enc is not None else w_text
        return pipe.run_sharded(scenes[i], w, seed=i) if latency else pipe.run(scenes[i], w, seed=i)

    part = None
    if args.cu_split != "0" and not latency:
        from semabs_amd.scene import CuPartition
        ncu, _, lay = args.cu_split.partition(":")
        part = CuPartition(int(ncu), lay or "balanced")

    def run_range(lo, hi):
        res = None
        if part is None:
            for i in range(lo, hi):
                res = step(i)
            return res
        # CU-partitioned pipeline: relevancy(i) is queued on the large partition, then voxels(i - 1) on the small one - they run concurrently
        prev = None
        cur = torch.cuda.current_stream()
        for s_ in part.streams:
            s_.wait_stream(cur)
        for i in range(lo, hi):
            with torch.cuda.stream(part.vit):
                w = text_enc.zeroshot_weights(text_tokens, N_LABELS, 1) if text_enc is not None else w_text
                st = pipe.run_relevancy(scenes[i], w, seed=i)
            if prev is not None:
                with torch.cuda.stream(part.voxel):
                    res = pipe.run_voxels(prev)
            prev = st
        with torch.cuda.stream(part.voxel):
            res = pipe.run_voxels(prev)
        for s_ in part.streams:
            cur.wait_stream(s_)
        return res

    run_range(0, args.warmup)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    timer = vitmod.GemmTimer(every=args.time_every)
    vitmod.GEMM_TIMER = timer
    if dist is not None:
        from semabs_amd import dist as _sdist
        _sdist.reset_stats()
    smi = SmiSampler(local) if rank == 0 else None
    torch.cuda.synchronize()
    if smi is not None:
        smi.start()
    t0 = time.perf_counter()
    res = run_range(args.warmup, n_scenes)
    # the job's results leave the ranks through RCCL: every rank's last label volume is all-gathered (scene-shard mode has no other payload collective;
    # in latency mode the per-step all-gathers already carried the relevances and the logits)
    gathered = None
    if dist is not None and res is not None and res.labels is not None:
        from semabs_amd import dist as sdist
        gathered = sdist.gather_results(res.labels)                           # [world, S^3] int32 on every rank
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if smi is not None:
        smi.stop()
    vitmod.GEMM_TIMER = None
    if dist is not None:
        t = torch.tensor([dt], device="cpu" if args.backend == "gloo" else "cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    gs = timer.summary()
    total_scenes = args.steps * (1 if latency else world)
    value = total_scenes / dt
    wire = None
    if world > 1:
        from semabs_amd import dist as sdist
        per_rank = [None] * world                            # (snapshot first: the checksum exchange below is not part of the timed region)
        dist.all_gather_object(per_rank, {"rank": rank, "scenes": args.steps, "collectives": sdist.stats_snapshot()})
        # what this rank sent per collective (payload only) and cross-rank checksums: in latency mode all ranks must end with identical maps / labels
        import hashlib
        from semabs_amd.clip import saliency_configs as _sc
        from semabs_amd.clip import plan_tiles as _pt
        ncfg = _sc["ours"](IMG)
        n_tiles = len(_pt(IMG, IMG, ncfg["augmentations"] + 1, ncfg["cropping_augmentations"])[0])
        g = 14 if "16" in args.arch else 7
        per = -(-N_LABELS // world)
        wire = {"gather_results_bytes_per_rank": int(gathered[0].numel() * gathered[0].element_size()) if gathered is not None else 0,
                "gather_results_ranks_seen": int(gathered.shape[0]) if gathered is not None else 0}
        if latency:
            wire.update(tile_relevance_allgather_bytes_per_rank_per_scene=int(2 * N_LABELS * (-(-n_tiles // world)) * g * g * 4),
                        logits_allgather_bytes_per_rank_per_scene=int(per * VOXEL ** 3 * 4))
        digest = lambda t: int.from_bytes(hashlib.sha256(t.detach().cpu().numpy().tobytes()).digest()[:7], "big")
        mine = torch.tensor([digest(res.relevancies), digest(res.labels)], dtype=torch.int64, device="cuda")
        from semabs_amd import dist as sdist
        allsum = sdist.gather_results(mine).cpu().numpy()
        wire["per_rank"] = per_rank
        wire["per_rank_note"] = "payload bytes this rank contributed and host-side milliseconds of every collective it issued inside the timed region, by kind"
        wire["maps_checksums_by_rank"] = [int(x) for x in allsum[:, 0]]
        wire["labels_checksums_by_rank"] = [int(x) for x in allsum[:, 1]]
        wire["identical_across_ranks"] = bool((allsum == allsum[0]).all())
        if gathered is not None:
            wire["gathered_label_volumes_checksum"] = digest(gathered)
    if rank == 0:
        ach = gs["flops"] / (gs["total_ms"] * 1e-3) / 1e12 if gs["total_ms"] > 0 else 0.0
        # HBM bytes per GEMM launch: PMC counters cannot be read from inside the process that is being timed (rocprofv3 owns them and
        # serialises the kernels), so this figure is IMPORTED from the committed counter run of this same command and labelled as such
        traffic, traffic_src = None, None
        for name in ("r06_gemm_pmc.json", "r05_gemm_pmc.json"):
            pmc = os.path.join(ROOT, "profiles", name)
            if os.path.exists(pmc):
                try:
                    traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
                    traffic_src = f"imported from profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; FETCH_SIZE x 2 per the gfx950 correction), not measured in this run"
                    break
                except Exception:
                    traffic = None
        out = {
            "metric": "scenes/sec (relevancy+3D-UNet infer), 480x480x16-label x128^3" + (" - single-scene LATENCY mode: every scene tile- and label-sharded over all ranks" if latency else ""),
            "value": value, "unit": "scenes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if latency else "weak", "vs_baseline": None,
            "dtype": "f16" if args.precision == "fp16" else "f16 (MFMA operands, fp32 accumulate; UNet hi/lo-split = fp32-equivalent)",
            "data": "synthetic",
            "config": {"workload": f"end-to-end relevancy->fusion->OVSSC per scene: {IMG}x{IMG} RGB-D, {N_LABELS} labels, {args.arch}, "
                                   f"'ours' saliency config (2448 tile forwards), {VOXEL}^3 voxels, 80000 input points; scene-sharded",
                       "arch": args.arch, "unet_precision": args.precision, "tile_chunk_streams": args.streams, "scenes_per_gpu": args.steps, "parallelism": (f"tile-shard + label-shard x{world} (one scene per step)" if latency else f"scene-shard x{world}"),
                       "mode": args.mode, "backend": args.backend if world > 1 else None,
                       "cu_split": None if part is None else {"voxel_cus": part.cus[0], "vit_cus": part.cus[1], "layout": part.layout}},
            "relevancy_tflops_algorithmic": 2448 * FLOPS_PER_TILE[args.arch] * total_scenes / dt / 1e12,
            "roofline": {"kernel": "fp16 GEMM: k_gemm8 (large shapes) + k_gemm_f16 (small), all epilogues", "bound": "mfma", "achieved": ach, "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / PEAK_F16_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": gs["bytes"] / max(1, gs["launches"]),
                         "traffic_over_algorithmic": (traffic / (gs["bytes"] / max(1, gs["launches"]))) if (traffic and gs["bytes"]) else None,
                         "algorithmic_bytes_note": "A + W + C per launch (fp16 operands; C fp16 2 B, fp32 4 B, fp32 residual read-modify-write 8 B per element), averaged over the timed launches",
                         "launches": gs["launches"], "launches_in_timed_region": gs["seen"],
                         "sampling": ("every GEMM launch of the timed region carries start / stop events" if args.time_every == 1 else
                                      f"1 in {args.time_every} GEMM launches of the timed region (hashed launch index) carries start / stop events"),
                         "avg_launch_us": gs["total_ms"] * 1e3 / max(1, gs["launches"]),
                         "per_shape": {gemm_shape_name(k[0], k[1], k[2]): {"launches": v[0], "avg_us": round(v[1] * 1e3 / v[0], 1), "tflops": round(v[2] / v[1] / 1e9, 1),
                                                                             "roof_us": round(gemm_roof_ms(v[2] / v[0], v[3] / v[0])[0] * 1e3, 1), "bound": gemm_roof_ms(v[2] / v[0], v[3] / v[0])[1],
                                                                             "frac_of_roof": round(gemm_roof_ms(v[2] / v[0], v[3] / v[0])[0] * v[0] / v[1], 3)}
                                       for k, v in sorted(gs["shapes"].items(), key=lambda kv: -kv[1][1]) if v[1] > 0},
                         "frac_of_per_launch_roofs": (sum(gemm_roof_ms(v[2] / v[0], v[3] / v[0])[0] * v[0] for v in gs["shapes"].values() if v[0]) / gs["total_ms"]) if gs["total_ms"] > 0 else None,
                         "gemm_share_of_step": gs["total_ms"] * 1e-3 * gs["seen"] / max(1, gs["launches"]) / dt if world == 1 else None,
                         },
        }
        out["roofline"].update(smi.summary())
        # context beside `frac` (which stays against the 2.4 GHz dense peak): the same achieved rate against the peak at the shader clock the part actually held
        # during the timed region - at its power cap the part clocks down under matrix load, so this is the fraction of the matrix pipe's cycles that did work
        sclk = out["roofline"].get("sclk_mhz_under_load")
        if sclk:
            out["roofline"]["peak_at_measured_sclk"] = PEAK_F16_TFLOPS * float(sclk) / 2400.0
            out["roofline"]["frac_at_measured_sclk"] = ach / (PEAK_F16_TFLOPS * float(sclk) / 2400.0)
            out["roofline"]["frac_at_measured_sclk_note"] = ("achieved / (peak x sclk_mhz_under_load / 2400): the scene-wide median clock; the GEMM phases alone hold a LOWER "
                                                             "clock than the median (profiles/r04_mfma_probes.txt), so this is a lower bound of the matrix pipe's busy fraction")
        out["timed_region"] = ("per scene, everything from the uint8 frame + fp32 depth resident in HBM to the label volume: " +
                               ("the text tower on the 16 labels' token ids (the reference encodes the label set on every get_clip_saliency call), " if text_enc is not None else "") +
                               "colour jitter, tiling, ViT + rollout, aggregation, unprojection + compaction + sub-sample, point MLP, scatter, UNet, decoder, TSDF, "
                               "frustum mask of the lattice (computed on the device per scene), post-mask.  NOT timed: " +
                               ("" if text_enc is not None else "the text tower (token fixture tests/golden/tokens_default.npz missing: synthetic unit-norm weights), ") +
                               "tokenisation of the label strings (host, the BPE table is not on the GPU box: token ids come from the committed fixture) and the "
                               "host->HBM upload of the frame (5 MB, 0.08 ms over PCIe Gen5)")
        out["text_tower_in_timed_region"] = text_enc is not None
        if wire is not None:
            wire["exposed_ms_per_step"] = max(sum(v.get("device_ms", 0.0) for v in r["collectives"].values()) / args.steps for r in per_rank)
            wire["exposed_ms_note"] = ("device time the compute stream spent inside the (blocking) all-gathers of the timed region, per step, maximum over the ranks - HIP events "
                                       "around each call (semabs_amd.dist); scene-shard mode has ONE gather of the final label volumes in the whole run")
            out["collectives"] = wire
        if world == 1 and not args.no_parity:
            out["parity"] = parity_report(pipe, args.arch, args.precision)
        if world == 1 and not args.no_stages:
            out["stages"] = stage_report(pipe, scenes, w_text, args.arch, args.precision)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.arch, N_LABELS)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
