"""Drop-in for the on-path classes of the reference's `net.py`: `VirtualGrid` (net.py:24-201) and `SemAbs3D`
(net.py:319-439), computed by HIP kernels.  Baselines (SemanticAware*, ClipSpatialVOOL) are out of scope.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np
import torch

from . import _lib


class VirtualGrid:
    def __init__(self, scene_bounds, grid_shape: Tuple[int, int, int] = (32, 32, 32), batch_size: int = 8,
                 device=None, int_dtype=torch.int64, float_dtype=torch.float32, reduce_method: str = "mean"):
        self.lower_corner = tuple(scene_bounds[0])
        self.upper_corner = tuple(scene_bounds[1])
        self.grid_shape = tuple(int(s) for s in grid_shape)
        self.batch_size = int(batch_size)
        self.device = device
        self.int_dtype = int_dtype
        self.float_dtype = float_dtype
        self.reduce_method = reduce_method
        # fp32 constants formed with the reference's op order (net.py:91-95): offsets = -lc, scales = (S-1)/(uc-lc)
        lc = np.asarray(self.lower_corner, np.float64).astype(np.float32)
        uc = np.asarray(self.upper_corner, np.float64).astype(np.float32)
        idx_scale = np.asarray(self.grid_shape, np.float32) - np.float32(1)
        self.offsets = (-lc).astype(np.float32)
        self.scales = (idx_scale / (uc - lc)).astype(np.float32)

    @property
    def num_grids(self):
        return int(np.prod((self.batch_size,) + self.grid_shape))

    def flat_idxs(self, points: torch.Tensor) -> torch.Tensor:
        """points fp32 [..., 3] on the GPU -> int64 [...] flat voxel index (get_points_grid_idxs + flatten_idxs)."""
        _lib.require_gpu()
        pts = points.contiguous().view(-1, 3)
        flat = torch.empty(pts.shape[0], dtype=torch.int64, device=pts.device)
        _lib.call("semabs_voxel_index", _lib.ptr(pts), pts.shape[0], _lib.farr(self.offsets), _lib.farr(self.scales),
                  _lib.iarr(self.grid_shape), _lib.ptr(flat), None, _lib.stream())
        return flat.view(points.shape[:-1])

    def get_points_grid_idxs(self, points: torch.Tensor, cast_to_int=True, batch_idx=None):
        assert cast_to_int and batch_idx is None, "only the integer, un-batched form is on the path"
        _lib.require_gpu()
        pts = points.contiguous().view(-1, 3)
        idx3 = torch.empty(pts.shape[0], 3, dtype=torch.int32, device=pts.device)
        _lib.call("semabs_voxel_index", _lib.ptr(pts), pts.shape[0], _lib.farr(self.offsets), _lib.farr(self.scales),
                  _lib.iarr(self.grid_shape), None, _lib.ptr(idx3), _lib.stream())
        return idx3.to(self.int_dtype).view(*points.shape[:-1], 3)

    def flatten_idxs(self, idxs: torch.Tensor, keepdim=False):
        S0, S1, S2 = self.grid_shape
        assert idxs.shape[-1] == 3
        flat = idxs[..., 0] * (S1 * S2) + idxs[..., 1] * S2 + idxs[..., 2]
        return flat.unsqueeze(-1) if keepdim else flat


class SemAbs3D:
    """Drop-in (inference) for `net.SemAbs3D` (net.py:319-439): point MLP -> scatter-MEAN into the voxel grid ->
    ResidualUNet3D -> trilinear implicit decoder, all on HIP kernels, channels-last in between.

    Quirks kept on purpose: the grid is built without a reduce method so points are reduced with MEAN although the
    constructor asserts "max" (net.py:339-344, 369, 185-199); the decoder divides by S (not S - 1) and feeds point-x
    to grid_sample's innermost axis (net.py:221-239).
    """

    def __init__(self, voxel_shape, scene_bounds, unet_num_channels, unet_f_maps, unet_num_groups, unet_num_levels,
                 network_inputs: List[str], use_pts_feat_extractor: bool, pts_feat_extractor_hidden_dim: int,
                 reduce_method: str, output_dim=1, device: str = "cuda", decoder_concat_xyz_pts: bool = False,
                 precision: str = "fp16", **kwargs):
        from .unet3d import ResidualUNet3D
        self.device = device
        self.vg = VirtualGrid(scene_bounds=np.array(scene_bounds), batch_size=kwargs.get("batch_size", 1),
                              grid_shape=tuple(voxel_shape), device=torch.device(device) if isinstance(device, str) else device)
        self.steps = torch.zeros(1)
        self.network_inputs = list(network_inputs)
        self.use_pts_feat_extractor = use_pts_feat_extractor
        self.reduce_method = reduce_method
        self.pts_feature_dim = (("saliency" in self.network_inputs) + ("rgb" in self.network_inputs) * 3
                                + ("patch_masks" in self.network_inputs))
        if not (use_pts_feat_extractor and self.pts_feature_dim == 1 and "tsdf" not in self.network_inputs and output_dim == 1):
            raise NotImplementedError("the HIP path covers the released OVSSC configuration: network_inputs=['saliency'], "
                                      "use_pts_feat_extractor=True, output_dim=1")
        assert self.reduce_method == "max"          # asserted by the reference too (and then ignored)
        self.hidden = pts_feat_extractor_hidden_dim
        self.C = unet_num_channels
        self.concat_xyz = bool(decoder_concat_xyz_pts)
        self.precision = precision
        self.vol_feature_extractor = ResidualUNet3D(in_channels=unet_num_channels, out_channels=unet_num_channels,
                                                    f_maps=unet_f_maps, num_groups=unet_num_groups,
                                                    num_levels=unet_num_levels, precision=precision)
        self._sd = {}
        self._w = None
        self.features_cl = None

    # ---- weights -----------------------------------------------------------------------------------
    def load_state_dict(self, sd, strict: bool = True):
        sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}     # DDP checkpoints
        dev = _lib.require_gpu()
        f = lambda k: sd[k].float().to(dev).contiguous()
        self._w = {k: f(f"pts_feat_extractor.{i}.{n}") for k, (i, n) in
                   dict(w1=(0, "weight"), b1=(0, "bias"), w2=(2, "weight"), b2=(2, "bias"), w3=(4, "weight"), b3=(4, "bias")).items()}
        self._dec = {k: np.ascontiguousarray(sd[f"visual_sampler.mlp.{i}.{n}"].float().cpu().numpy().reshape(-1)) for k, (i, n) in
                     dict(w1=(0, "weight"), b1=(0, "bias"), w2=(2, "weight"), b2=(2, "bias")).items()}
        self.vol_feature_extractor.load_state_dict(sd, strict=strict, prefix="vol_feature_extractor.")
        self._final = (np.ascontiguousarray(sd["vol_feature_extractor.final_conv.weight"].float().cpu().numpy().reshape(-1)),
                       np.ascontiguousarray(sd["vol_feature_extractor.final_conv.bias"].float().cpu().numpy().reshape(-1)))
        if "steps" in sd:
            self.steps = sd["steps"].clone()
        self._sd = {k: v.detach().clone() for k, v in sd.items()}
        return self

    def state_dict(self):
        return dict(self._sd)

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    # ---- stages ------------------------------------------------------------------------------------
    def feature_volume(self, xyz: torch.Tensor, feat: torch.Tensor, taps: dict | None = None, skip_final: bool = False) -> torch.Tensor:
        """xyz fp32 [N, 3], feat fp32 [P, N] (one scene, P label volumes) -> UNet features channels-last [P, S, S, S, C].
        skip_final: stop in front of the UNet's final 1x1x1 convolution (see `decode(pre_final=True)`)."""
        dev = _lib.require_gpu()
        P, N = int(feat.shape[0]), int(feat.shape[1])
        S0, S1, S2 = self.vg.grid_shape
        nvox = S0 * S1 * S2
        st = _lib.stream()
        w = self._w
        pf = torch.empty(P, N, self.C, dtype=torch.float32, device=dev)
        xyz, feat = xyz.contiguous(), feat.contiguous()      # held in locals: the kernels read these buffers
        _lib.call("semabs_point_mlp", _lib.ptr(xyz), _lib.ptr(feat), _lib.ptr(w["w1"]), _lib.ptr(w["b1"]),
                  _lib.ptr(w["w2"]), _lib.ptr(w["b2"]), _lib.ptr(w["w3"]), _lib.ptr(w["b3"]), _lib.ptr(pf), P, N, self.hidden, self.C, st)
        flat = self.vg.flat_idxs(xyz)
        unet = self.vol_feature_extractor
        vol = torch.zeros(P, S0, S1, S2, self.C, dtype=unet.act_dtype, device=dev)
        head = torch.full((nvox,), -1, dtype=torch.int32, device=dev)
        nxt = torch.empty(N, dtype=torch.int32, device=dev)
        sums = None
        if self.C == 16 and unet.in_channels == 16 and unet.enc[0][0].groups == 8:     # statistics for the first GroupNorm come out of the scatter
            sums = torch.zeros(P, 8, 2, dtype=torch.float64, device=dev)
            _lib.call("semabs_scatter_mean_stats", _lib.ptr(flat), _lib.ptr(pf), _lib.ptr(head), _lib.ptr(nxt), _lib.ptr(vol), P, N, self.C, nvox,
                      unet.f32, _lib.ptr(sums), st)
        else:
            _lib.call("semabs_scatter_mean", _lib.ptr(flat), _lib.ptr(pf), _lib.ptr(head), _lib.ptr(nxt), _lib.ptr(vol), P, N, self.C, nvox,
                      unet.f32, st)
        if taps is not None:
            taps["scatter"] = vol
            taps["point_feat"] = pf
        return unet.forward_cl(vol, taps=taps, skip_final=skip_final, in_sums=sums)

    def decode(self, features_cl: torch.Tensor, query: torch.Tensor, shared: bool = False, lattice=None, pre_final: bool = False) -> torch.Tensor:
        """features [P, S, S, S, C]; query fp32 [P, M, 3] (or [M, 3] with shared=True) -> logits fp32 [P, M].
        lattice=(G0, G1, G2): the M queries are a dense C-order lattice (e.g. `VirtualGrid.get_grid_points`): same results, faster walk.
        pre_final: `features_cl` is `feature_volume(..., skip_final=True)`, i.e. the activation in front of the UNet's final 1x1x1
        convolution; that (linear) layer is applied to the sampled features inside the decoder kernel instead."""
        dev = _lib.require_gpu()
        P = int(features_cl.shape[0])
        M = int(query.shape[-2])
        out = torch.empty(P, M, dtype=torch.float32, device=dev)
        d = self._dec
        fp = lambda a: a.ctypes.data
        query = query.contiguous()
        _lib.call("semabs_decoder", _lib.ptr(features_cl), _lib.ptr(query), _lib.farr(self.vg.offsets), _lib.farr(self.vg.scales),
                  _lib.iarr(self.vg.grid_shape), fp(d["w1"]), fp(d["b1"]), fp(d["w2"]), fp(d["b2"]), int(self.concat_xyz), P, M,
                  0 if shared else M * 3, self.vol_feature_extractor.f32, _lib.ptr(out), None if lattice is None else _lib.iarr(lattice),
                  fp(self._final[0]) if pre_final else None, fp(self._final[1]) if pre_final else None, _lib.stream())
        return out

    # ---- reference surface ---------------------------------------------------------------------------
    def forward(self, input_xyz_pts, input_feature_pts, tsdf_vol, output_xyz_pts, **kwargs):
        dev = _lib.require_gpu()
        B, P, N = input_feature_pts.shape[:3]
        M = output_xyz_pts.shape[2]
        outs, feats = [], []
        for b in range(B):
            f = self.feature_volume(input_xyz_pts[b].to(dev, torch.float32), input_feature_pts[b].to(dev, torch.float32).reshape(P, N))
            feats.append(f)
            outs.append(self.decode(f, output_xyz_pts[b].to(dev, torch.float32)))
        self.features_cl = torch.cat(feats, dim=0)
        return torch.stack(outs, dim=0).view(B, P, M)

    __call__ = forward

    @property
    def visual_volumetric_features(self):
        """[B*P, C, S, S, S] fp32 like the reference's cached attribute (net.py:425-427)."""
        return None if self.features_cl is None else self.features_cl.permute(0, 4, 1, 2, 3).float()


class SemAbsVOOL:
    """Drop-in (inference) for `net.SemAbsVOOL` (net.py:469-579) with `pointing_method="cosine_sim"` (the default,
    utils.py:87-91): two SemAbs3D feature volumes (target / reference saliency), the 35 -> 32 -> 64 spatial sampler and the
    cosine-similarity pointer against the relation embedding, fused in one HIP kernel (`semabs_vool_head`)."""

    RELATIONS = ["in", "behind", "in front of", "on the left of", "on the right of", "on", "[pad]"]

    def __init__(self, pointing_method: str, pointing_dim: int, device: str, decoder_concat_xyz_pts: bool, **kwargs):
        if pointing_method != "cosine_sim" or pointing_dim != 64 or not decoder_concat_xyz_pts:
            raise NotImplementedError("the HIP VOOL head covers pointing_method='cosine_sim', pointing_dim=64, decoder_concat_xyz_pts=True")
        self.device = device
        self.steps = torch.zeros(1)
        self.completion_net = SemAbs3D(device=device, **kwargs)       # like the reference: built without the xyz concat
        self.pointing_temperature = 0.07
        self._prm = None
        self._rel = {}
        self._sd = {}

    def load_state_dict(self, sd, strict: bool = True):
        sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
        dev = _lib.require_gpu()
        self.completion_net.load_state_dict({k[len("completion_net."):]: v for k, v in sd.items() if k.startswith("completion_net.")}, strict)
        flat = [sd[f"spatial_sampler.mlp.{i}.{n}"].float().reshape(-1) for i, n in ((0, "weight"), (0, "bias"), (2, "weight"), (2, "bias"))]
        assert [t.numel() for t in flat] == [32 * 35, 32, 64 * 32, 64], "spatial sampler must be 35 -> 32 -> 64"
        self._prm = torch.cat(flat).to(dev).contiguous()
        self._rel = {k: sd["relation_embeddings." + k].float().to(dev) for k in self.RELATIONS}
        self._sd = {k: v.detach().clone() for k, v in sd.items()}
        return self

    def state_dict(self):
        return dict(self._sd)

    def eval(self):
        return self

    def forward(self, output_xyz_pts, spatial_relation_name, input_xyz_pts, input_target_saliency_pts, input_reference_saliency_pts,
                tsdf_vol=None, **kwargs):
        dev = _lib.require_gpu()
        net = self.completion_net
        batch_size, num_descs = np.array(spatial_relation_name).T.shape
        M = int(output_xyz_pts.shape[-2])
        outs = []
        for b in range(batch_size):
            xyz = input_xyz_pts[b].to(dev, torch.float32)
            N = xyz.shape[0]
            ft = net.feature_volume(xyz, input_target_saliency_pts[b].to(dev, torch.float32).reshape(num_descs, N))
            fr = net.feature_volume(xyz, input_reference_saliency_pts[b].to(dev, torch.float32).reshape(num_descs, N))
            rel = torch.stack([self._rel[spatial_relation_name[d][b]] for d in range(num_descs)], dim=0).contiguous()
            q = output_xyz_pts[b].to(dev, torch.float32).reshape(num_descs, M, 3).contiguous()
            out = torch.empty(num_descs, M, dtype=torch.float32, device=dev)
            _lib.call("semabs_vool_head", _lib.ptr(ft), _lib.ptr(fr), _lib.ptr(q), _lib.ptr(self._prm), _lib.ptr(rel), _lib.farr(net.vg.offsets),
                      _lib.farr(net.vg.scales), _lib.iarr(net.vg.grid_shape), float(self.pointing_temperature), num_descs, M,
                      net.vol_feature_extractor.f32, _lib.ptr(out), _lib.stream())
            outs.append(out)
        return torch.stack(outs, dim=0).view(batch_size, num_descs, M)

    __call__ = forward
