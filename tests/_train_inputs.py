"""Seeded inputs of the G13 VOOL training golden (same construction as tests/golden/gen_golden.py:_semabs_inputs / g13_vool_train)."""
import numpy as np
import torch

from semabs_amd.synth import SCENE_BOUNDS

REL_NAMES = [["behind"], ["on"], ["behind"], ["in"]]          # gen_golden.py: g13 / g20 use the first three, g22 (4 descriptions) all four


def vool_batch(S, N, M, D, seed, label):
    rng = np.random.default_rng(seed)
    lo, hi = np.array(SCENE_BOUNDS[0]), np.array(SCENE_BOUNDS[1])
    P = 2 * D
    xyz = (lo + (hi - lo) * rng.random((1, N, 3))).astype(np.float32)
    xyz[0, : N // 8] = xyz[0, N // 8: 2 * (N // 8)] + np.float32(1e-3)
    feat = (rng.standard_normal((1, P, N, 1)) * 0.5).astype(np.float32)
    q = (lo - 0.05 + (hi - lo + 0.1) * rng.random((1, P, M, 3))).astype(np.float32)
    return dict(input_xyz_pts=torch.from_numpy(xyz), input_target_saliency_pts=torch.from_numpy(feat[:, :D]),
                input_reference_saliency_pts=torch.from_numpy(feat[:, D:]), output_xyz_pts=torch.from_numpy(q[:, :D]),
                output_label_pts=torch.from_numpy(np.asarray(label, np.float32)), spatial_relation_name=[list(r) for r in REL_NAMES[:D]])


def worst_param_deviation(sd, ref_sd, before, ref_g, noise_floor=1e-3, sel_frac=5e-2, quantile=1.0):
    """Two optimisation steps from the same start compared parameter by parameter.  LAMB's first step is sign-like (m / (sqrt(v) + eps) = g / |g|):
    where a gradient element is within noise of zero its update flips sign on any pair of runs, so compare only where |g| is well above the
    tensor's noise floor, and skip tensors whose whole gradient is noise.  All arguments: dicts of numpy arrays (ref_g: the reference run's
    gradients, only for tensors that have one).  -> worst |sd - ref_sd| over the selected elements, as a fraction of the tensor's own max step;
    tensors without gradient must be bit-identical.  quantile < 1: per tensor that quantile of the selected elements instead of their maximum -
    one ReLU / max-pool tie flipping between two runs moves a handful of gradient elements by O(1), a wrong mapping moves whole tensors."""
    gnorm = np.sqrt(sum(float((v.astype(np.float64) ** 2).sum()) for v in ref_g.values()))
    worst = 0.0
    for k, v in ref_sd.items():
        if k not in ref_g:
            assert np.array_equal(sd[k], v), k
            continue
        g = ref_g[k]
        if float(np.linalg.norm(g.astype(np.float64))) < noise_floor * gnorm or np.abs(g).max() == 0:
            continue
        sel = np.abs(g) > sel_frac * np.abs(g).max()
        step = np.abs(v - before[k]).max()
        if sel.any() and step > 0:
            dev = np.abs(sd[k] - v)[sel] / step
            worst = max(worst, float(np.quantile(dev, quantile)) if quantile < 1.0 else float(dev.max()))
    return worst
