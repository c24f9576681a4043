"""ORACLE (test infrastructure, not product code) — CPU restatement of the OVSSC voxel-inference half.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this.

  SemAbs3D.forward                     net.py:383-439
  pts_feat_extractor (point MLP)       net.py:358-367, 395-404
  VirtualGrid.scatter_points (MEAN!)   net.py:185-201  (reduce_method argument is ignored; default "mean")
  ResidualUNet3D                       unet3d.py:596-621, 190-259 (ExtResNetBlock, order "gcr"), 262-444
  ImplicitVolumetricDecoder            net.py:215-256  (divide by S not S-1; x -> W axis of grid_sample)

Functional torch-CPU fp32 over a plain state dict (key names = `SemAbs3D.state_dict()`).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from .geometry import flatten_idxs, grid_constants, points_grid_idxs


def point_mlp(sd, xyz: torch.Tensor, feat: torch.Tensor) -> torch.Tensor:
    x = torch.cat((xyz, feat), dim=-1)
    x = F.leaky_relu(F.linear(x, sd["pts_feat_extractor.0.weight"], sd["pts_feat_extractor.0.bias"]), 0.01)
    x = F.leaky_relu(F.linear(x, sd["pts_feat_extractor.2.weight"], sd["pts_feat_extractor.2.bias"]), 0.01)
    return F.linear(x, sd["pts_feat_extractor.4.weight"], sd["pts_feat_extractor.4.bias"])


def scatter_mean(xyz: torch.Tensor, feat: torch.Tensor, scene_bounds, grid_shape) -> torch.Tensor:
    """xyz [B, N, 3], feat [B, N, C] -> [B, C, S, S, S]; empty voxels 0."""
    B, N, C = feat.shape
    flat = torch.from_numpy(flatten_idxs(points_grid_idxs(xyz.numpy(), scene_bounds, grid_shape), grid_shape))
    nvox = int(np.prod(grid_shape))
    out = torch.zeros(B, nvox, C)
    out.scatter_add_(1, flat.unsqueeze(-1).expand(B, N, C), feat)
    cnt = torch.zeros(B, nvox)
    cnt.scatter_add_(1, flat, torch.ones(B, N))
    out = out / cnt.clamp(min=1).unsqueeze(-1)
    return out.view(B, *grid_shape, C).permute(0, 4, 1, 2, 3).contiguous()


def _single_conv(sd, pre, x, groups, relu):
    c = x.shape[1]
    g = groups if c >= groups else 1
    x = F.group_norm(x, g, sd[pre + "groupnorm.weight"], sd[pre + "groupnorm.bias"], 1e-5)
    x = F.conv3d(x, sd[pre + "conv.weight"], None, padding=1)
    return F.relu(x) if relu else x


def _res_block(sd, pre, x, groups):
    out = _single_conv(sd, pre + "conv1.", x, groups, True)
    res = out
    out = _single_conv(sd, pre + "conv2.", out, groups, True)
    out = _single_conv(sd, pre + "conv3.", out, groups, False)
    return F.relu(out + res)


def unet_forward(sd, x: torch.Tensor, num_levels: int, groups: int = 8,
                 prefix: str = "vol_feature_extractor.", taps: dict | None = None) -> torch.Tensor:
    feats = []
    for i in range(num_levels):
        if i > 0:
            x = F.max_pool3d(x, 2)
        x = _res_block(sd, f"{prefix}encoders.{i}.basic_module.", x, groups)
        if taps is not None:
            taps[f"enc{i}"] = x
        feats.insert(0, x)
    for i, skip in enumerate(feats[1:]):
        up = f"{prefix}decoders.{i}.upsampling.upsample."
        x = F.conv_transpose3d(x, sd[up + "weight"], sd[up + "bias"], stride=2, padding=1, output_padding=1)
        x = skip + x
        x = _res_block(sd, f"{prefix}decoders.{i}.basic_module.", x, groups)
        if taps is not None:
            taps[f"dec{i}"] = x
    return F.conv3d(x, sd[prefix + "final_conv.weight"], sd[prefix + "final_conv.bias"])


def decoder(sd, vol: torch.Tensor, query: torch.Tensor, scene_bounds, grid_shape, concat_xyz: bool) -> torch.Tensor:
    """vol [B, C, S, S, S], query [B, M, 3] -> [B, M, out]  (net.py:215-256)."""
    off, sc = grid_constants(scene_bounds, grid_shape)
    q = (query.float() + torch.from_numpy(off)) * torch.from_numpy(sc)
    S = torch.tensor(grid_shape, dtype=torch.float32)
    q = torch.minimum(torch.maximum(q, torch.zeros(3)), S - 1)
    q = q / S
    qn = 2.0 * q - 1.0
    B, M = qn.shape[:2]
    samp = F.grid_sample(vol, qn.view(B, M, 1, 1, 3), mode="bilinear", padding_mode="border", align_corners=True)
    samp = samp.view(B, vol.shape[1], M).permute(0, 2, 1).reshape(B * M, -1)
    if concat_xyz:
        samp = torch.cat((samp, qn.reshape(B * M, 3)), dim=-1)
    h = F.leaky_relu(F.linear(samp, sd["visual_sampler.mlp.0.weight"], sd["visual_sampler.mlp.0.bias"]), 0.01)
    return F.linear(h, sd["visual_sampler.mlp.2.weight"], sd["visual_sampler.mlp.2.bias"]).view(B, M, -1)


def semabs3d_forward(sd, input_xyz_pts, input_feature_pts, output_xyz_pts, scene_bounds, grid_shape,
                     num_levels: int = 6, groups: int = 8, concat_xyz: bool = True, taps: dict | None = None, tsdf_vol=None):
    """input_xyz_pts [B, N, 3], input_feature_pts [B, P, N, F], output_xyz_pts [B, P, M, 3] -> [B, P, M].
    tsdf_vol [B, S, S, S] (network_inputs contains "tsdf", net.py:411-419): concatenated in front of the scattered point features - the
    point MLP then has C - 1 outputs (net.py:365-367); repeated over the label volumes exactly like the reference's
    `tsdf_vol.unsqueeze(1).repeat(num_patches, 1, 1, 1, 1)`."""
    B, P, N = input_feature_pts.shape[:3]
    xyz = input_xyz_pts.unsqueeze(1).repeat(1, P, 1, 1).view(B * P, N, 3)
    feat = point_mlp(sd, xyz, input_feature_pts.reshape(B * P, N, -1))
    vol = scatter_mean(xyz, feat, scene_bounds, grid_shape)
    if tsdf_vol is not None:
        vol = torch.cat((tsdf_vol.unsqueeze(1).repeat(P, 1, 1, 1, 1), vol), dim=1)
    if taps is not None:
        taps["scatter"] = vol
    vol = unet_forward(sd, vol, num_levels, groups, taps=taps)
    if taps is not None:
        taps["unet"] = vol
    M = output_xyz_pts.shape[2]
    out = decoder(sd, vol, output_xyz_pts.reshape(B * P, M, 3), scene_bounds, grid_shape, concat_xyz)
    return out.view(B, P, M, -1).squeeze(-1)
