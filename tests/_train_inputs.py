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
