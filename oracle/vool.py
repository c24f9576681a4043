"""ORACLE (test infrastructure, not product code) — CPU restatement of `SemAbsVOOL.forward` (net.py:506-579, cosine_sim
pointer :300-309) and of `Lamb.step` (arm/optim/lamb.py:59-127).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import semabs3d as os3
from .geometry import grid_constants


def vool_forward(sd, input_xyz_pts, target_sal, reference_sal, output_xyz_pts, relation_names, scene_bounds, grid_shape,
                 num_levels: int = 6, temperature: float = 0.07):
    """input_xyz_pts [B, N, 3]; *_sal [B, D, N, 1]; output_xyz_pts [B, D, M, 3]; relation_names [D][B] -> [B, D, M]."""
    csd = {k[len("completion_net."):]: v for k, v in sd.items() if k.startswith("completion_net.")}
    B, D, N = target_sal.shape[:3]
    M = output_xyz_pts.shape[2]
    xyz = input_xyz_pts.unsqueeze(1).repeat(1, D, 1, 1).view(B * D, N, 3)

    def feature_vol(sal):
        feat = os3.point_mlp(csd, xyz, sal.reshape(B * D, N, 1))
        return os3.unet_forward(csd, os3.scatter_mean(xyz, feat, scene_bounds, grid_shape), num_levels)

    vol = torch.cat((feature_vol(target_sal), feature_vol(reference_sal)), dim=1)            # [B*D, 32, S, S, S]
    off, sc = grid_constants(scene_bounds, grid_shape)
    q = (output_xyz_pts.reshape(B * D, M, 3).float() + torch.from_numpy(off)) * torch.from_numpy(sc)
    S = torch.tensor(grid_shape, dtype=torch.float32)
    q = torch.minimum(torch.maximum(q, torch.zeros(3)), S - 1) / S
    qn = 2.0 * q - 1.0
    samp = F.grid_sample(vol, qn.view(B * D, M, 1, 1, 3), mode="bilinear", padding_mode="border", align_corners=True)
    samp = samp.view(B * D, vol.shape[1], M).permute(0, 2, 1).reshape(B * D * M, -1)
    samp = torch.cat((samp, qn.reshape(B * D * M, 3)), dim=-1)
    h = F.leaky_relu(F.linear(samp, sd["spatial_sampler.mlp.0.weight"], sd["spatial_sampler.mlp.0.bias"]), 0.01)
    o = F.linear(h, sd["spatial_sampler.mlp.2.weight"], sd["spatial_sampler.mlp.2.bias"]).view(B * D, M, -1)
    rel = torch.stack([torch.stack([sd["relation_embeddings." + relation_names[d][b]] for b in range(B)]) for d in range(D)])
    rel = rel.permute(1, 0, 2).reshape(B * D, 1, -1)
    return (F.cosine_similarity(o, rel, dim=-1) / temperature).view(B, D, M)


def lamb_step(w, g, m, v, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0):
    """One step on fp32 numpy arrays (updated copies returned): -> (w, m, v, (weight_norm, adam_norm, trust_ratio))."""
    b1, b2 = np.float32(betas[0]), np.float32(betas[1])
    m = m * b1 + g * np.float32(1 - betas[0])
    v = v * b2 + np.float32(1 - betas[1]) * g * g
    wn = np.float32(min(max(np.sqrt(np.float32((w.astype(np.float32) ** 2).sum())), 0.0), 10.0))
    s = m / (np.sqrt(v) + np.float32(eps))
    if weight_decay != 0:
        s = s + np.float32(weight_decay) * w
    an = np.float32(np.sqrt(np.float32((s ** 2).sum())))
    trust = np.float32(1.0) if (wn == 0 or an == 0) else np.float32(wn / an)
    w = w + np.float32(-lr * float(trust)) * s
    return w.astype(np.float32), m.astype(np.float32), v.astype(np.float32), (wn, an, trust)
