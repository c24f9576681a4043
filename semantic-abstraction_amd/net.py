"""Drop-in for the on-path classes of the reference's `net.py`: `VirtualGrid` (net.py:24-201) and `SemAbs3D`
(net.py:319-439), computed by HIP kernels.  Baselines (SemanticAware*, ClipSpatialVOOL) are out of scope.
"""
from __future__ import annotations

from typing import List, Tuple

import os
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


SPARSE_SCATTER = os.environ.get("SEMABS_SPARSE_SCATTER", "1") == "1"


class SemAbs3D(torch.nn.Module):
    """Drop-in (inference) for `net.SemAbs3D` (net.py:319-439): point MLP -> scatter-MEAN into the voxel grid ->
    ResidualUNet3D -> trilinear implicit decoder, all on HIP kernels, channels-last in between.

    A real `torch.nn.Module`: parameters / the `steps` buffer live under the reference's state-dict keys, so the reference's caller
    works unchanged - `net_class(**kwargs).to(device)`, `get_n_params`, `Lamb(net.parameters())`, `DistributedDataParallel(module=net)`,
    `net.load_state_dict(ckpt["net"])`, `net.eval()` (utils.py:225-296).  `forward` is inference-only (no autograd graph; training runs
    through `semabs_amd.train`).

    Quirks kept on purpose: the grid is built without a reduce method so points are reduced with MEAN although the
    constructor asserts "max" (net.py:339-344, 369, 185-199); the decoder divides by S (not S - 1) and feeds point-x
    to grid_sample's innermost axis (net.py:221-239).
    """

    def __init__(self, voxel_shape, scene_bounds, unet_num_channels, unet_f_maps, unet_num_groups, unet_num_levels,
                 network_inputs: List[str], use_pts_feat_extractor: bool, pts_feat_extractor_hidden_dim: int,
                 reduce_method: str, output_dim=1, device: str = "cuda", decoder_concat_xyz_pts: bool = False,
                 precision: str = "exact", **kwargs):
        super().__init__()
        from .module import register_tree
        from .unet3d import ResidualUNet3D
        from .weights import make_semabs3d_state_dict
        self.device = device
        self.vg = VirtualGrid(scene_bounds=np.array(scene_bounds), batch_size=kwargs.get("batch_size", 1),
                              grid_shape=tuple(voxel_shape), device=torch.device(device) if isinstance(device, str) else device)
        self.network_inputs = list(network_inputs)
        self.use_pts_feat_extractor = use_pts_feat_extractor
        self.reduce_method = reduce_method
        self.pts_feature_dim = (("saliency" in self.network_inputs) + ("rgb" in self.network_inputs) * 3
                                + ("patch_masks" in self.network_inputs))
        self.with_tsdf = "tsdf" in self.network_inputs
        if not (use_pts_feat_extractor and self.pts_feature_dim == 1 and output_dim == 1):
            raise NotImplementedError("the HIP path covers network_inputs = ['saliency'] or ['saliency', 'tsdf'] with "
                                      "use_pts_feat_extractor=True, output_dim=1 (the released OVSSC / VOOL configurations)")
        assert self.reduce_method == "max"          # asserted by the reference too (and then ignored)
        self.hidden = pts_feat_extractor_hidden_dim
        self.C = unet_num_channels
        self.concat_xyz = bool(decoder_concat_xyz_pts)
        self.precision = precision                   # "exact" by default: the reference's fp32 results; "fp16" = opt-in fast mode
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        init = make_semabs3d_state_dict(seed=seed, unet_num_channels=unet_num_channels, unet_f_maps=unet_f_maps, unet_num_groups=unet_num_groups,
                                        unet_num_levels=unet_num_levels, pts_feat_extractor_hidden_dim=pts_feat_extractor_hidden_dim,
                                        decoder_concat_xyz_pts=decoder_concat_xyz_pts)
        if self.with_tsdf:                          # the point MLP leaves one UNet input channel to the TSDF volume (net.py:365-367)
            init["pts_feat_extractor.4.weight"] = init["pts_feat_extractor.4.weight"][: unet_num_channels - 1].clone()
            init["pts_feat_extractor.4.bias"] = init["pts_feat_extractor.4.bias"][: unet_num_channels - 1].clone()
        # Registration order = the reference's (net.py:346-381: pts_feat_extractor, vol_feature_extractor, visual_sampler): `net.parameters()` is
        # positional for everything built on it - Lamb(net.parameters()), optimizer.state_dict() / load_state_dict (utils.py:264-266, 289)
        own = {k: v for k, v in init.items() if not k.startswith("vol_feature_extractor.")}
        register_tree(self, {k: v for k, v in own.items() if k.startswith("pts_feat_extractor.")})
        self.vol_feature_extractor = ResidualUNet3D(in_channels=unet_num_channels, out_channels=unet_num_channels,
                                                    f_maps=unet_f_maps, num_groups=unet_num_groups,
                                                    num_levels=unet_num_levels, precision=precision)
        register_tree(self, {k: v for k, v in own.items() if not k.startswith("pts_feat_extractor.")}, buffers=("steps",))
        self._w = self._dec = self._final = None
        self._sig = None
        self.features_cl = None

    # ---- weights -----------------------------------------------------------------------------------
    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        from .module import strip_module_prefix
        return super().load_state_dict(strip_module_prefix(state_dict), strict=strict, **kw)

    def _sync(self):
        """(Re)derive the kernels' operand layouts when a parameter changed (see module.py)."""
        dev = _lib.require_gpu()
        if any(p.device.type != "cuda" for p in self.parameters()):
            self.to(dev)                                # there is no CPU path: a module left on the host is moved to the HIP device on first use
        own = [t for n, t in list(self.named_parameters()) + list(self.named_buffers()) if not n.startswith("vol_feature_extractor.")]
        sig = tuple((t.data_ptr(), t._version) for t in own)
        if sig == self._sig:
            return
        sd = {k: v.detach() for k, v in self.state_dict().items() if not k.startswith("vol_feature_extractor.")}
        f = lambda k: sd[k].float().to(dev).contiguous()
        self._w = {k: f(f"pts_feat_extractor.{i}.{n}") for k, (i, n) in
                   dict(w1=(0, "weight"), b1=(0, "bias"), w2=(2, "weight"), b2=(2, "bias"), w3=(4, "weight"), b3=(4, "bias")).items()}
        if self.with_tsdf:
            # 15 output channels: run the 16-channel kernel with a zero row in front - channel 0 of the scattered volume then stays 0 and
            # receives the TSDF (torch.cat((tsdf, features), dim=1), net.py:411-419)
            self._w["w3"] = torch.cat([torch.zeros(1, self.hidden, device=dev), self._w["w3"]], dim=0).contiguous()
            self._w["b3"] = torch.cat([torch.zeros(1, device=dev), self._w["b3"]], dim=0).contiguous()
        self._dec = {k: np.ascontiguousarray(sd[f"visual_sampler.mlp.{i}.{n}"].float().cpu().numpy().reshape(-1)) for k, (i, n) in
                     dict(w1=(0, "weight"), b1=(0, "bias"), w2=(2, "weight"), b2=(2, "bias")).items()}
        self._sig = sig

    def _final_conv(self):
        """Host copies of the UNet's final 1x1x1 convolution (kernel arguments of the decoder when it applies that layer itself); re-read from the
        device only when the parameters changed - a device -> host copy per scene is a host synchronisation on the hot path."""
        u = self.vol_feature_extractor
        w, b = u.final_conv.weight, u.final_conv.bias
        sig = (w.data_ptr(), w._version, b.data_ptr(), b._version)
        if getattr(self, "_final_sig", None) != sig:
            self._final_host = (np.ascontiguousarray(w.detach().float().cpu().numpy().reshape(-1)),
                                np.ascontiguousarray(b.detach().float().cpu().numpy().reshape(-1)))
            self._final_sig = sig
        return self._final_host

    # ---- stages ------------------------------------------------------------------------------------
    @torch.no_grad()
    def feature_volume(self, xyz: torch.Tensor, feat: torch.Tensor, taps: dict | None = None, skip_final: bool = False,
                       tsdf_vol: torch.Tensor | None = None) -> torch.Tensor:
        """xyz fp32 [N, 3], feat fp32 [P, N] (one scene, P label volumes) -> UNet features channels-last [P, S, S, S, C].
        skip_final: stop in front of the UNet's final 1x1x1 convolution (see `decode(pre_final=True)`).
        tsdf_vol fp32 [S, S, S] or [P, S, S, S] (required iff "tsdf" is a network input): becomes input channel 0 of every (of its) label
        volume (net.py:411-419)."""
        dev = _lib.require_gpu()
        self._sync()
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
        unet._sync()
        head = _lib.filled((nvox,), torch.int32, -1, dev)
        nxt = torch.empty(N, dtype=torch.int32, device=dev)
        sums = None
        # Sparse scatter (round 6): the labels' volumes are scatters of the SAME points, 1 - 4 % of the voxels.  When the first convolution can read a
        # bitmap-described input (the 128^3 level-0 kernel) the volume is neither zero-filled (2.1 GB per 16-label scene) nor read where it is empty; tests
        # that tap the dense scatter volume, the TSDF input and other shapes take the dense form.  SEMABS_SPARSE_SCATTER=0: A/B.
        sparse = (taps is None and not self.with_tsdf and self.C == 16 and unet.in_channels == 16 and unet.enc[0][0].groups == 8 and SPARSE_SCATTER
                  and unet.sparse_input_supported((P, S0, S1, S2, self.C), unet.enc[0][0]))
        if sparse:
            vol = torch.empty(P, S0, S1, S2, self.C, dtype=unet.act_dtype, device=dev)
            occ = _lib.filled(((nvox + 31) // 32,), torch.int32, 0, dev)
            sums = _lib.filled((P, 8, 2), torch.float64, 0, dev)
            _lib.call("semabs_scatter_mean_sparse", _lib.ptr(flat), _lib.ptr(pf), _lib.ptr(head), _lib.ptr(nxt), _lib.ptr(vol), _lib.ptr(occ), P, N, self.C, nvox,
                      unet.f32, _lib.ptr(sums), st)
            return unet.forward_cl(vol, skip_final=skip_final, in_sums=sums, occ=occ)
        vol = _lib.filled((P, S0, S1, S2, self.C), unet.act_dtype, 0, dev)
        if self.with_tsdf:
            if tsdf_vol is None:
                raise ValueError("network_inputs contains 'tsdf': pass tsdf_vol")
            _lib.call("semabs_scatter_mean", _lib.ptr(flat), _lib.ptr(pf), _lib.ptr(head), _lib.ptr(nxt), _lib.ptr(vol), P, N, self.C, nvox,
                      unet.f32, st)
            # channel 0 (kept zero by the padded point MLP) <- the TSDF: a strided copy, no arithmetic; GroupNorm statistics then come from the
            # generic pass over the (now dense) volume, not from the occupied-voxel shortcut of the scatter kernel
            vol[..., 0] = tsdf_vol.to(dev, unet.act_dtype).reshape(-1, S0, S1, S2)       # [S, S, S] for every volume, or one per volume [P, S, S, S]
        elif self.C == 16 and unet.in_channels == 16 and unet.enc[0][0].groups == 8:     # statistics for the first GroupNorm come out of the scatter
            sums = _lib.filled((P, 8, 2), torch.float64, 0, dev)
            _lib.call("semabs_scatter_mean_stats", _lib.ptr(flat), _lib.ptr(pf), _lib.ptr(head), _lib.ptr(nxt), _lib.ptr(vol), P, N, self.C, nvox,
                      unet.f32, _lib.ptr(sums), st)
        else:
            _lib.call("semabs_scatter_mean", _lib.ptr(flat), _lib.ptr(pf), _lib.ptr(head), _lib.ptr(nxt), _lib.ptr(vol), P, N, self.C, nvox,
                      unet.f32, st)
        if taps is not None:
            taps["scatter"] = vol
            taps["point_feat"] = pf
        return unet.forward_cl(vol, taps=taps, skip_final=skip_final, in_sums=sums)

    @torch.no_grad()
    def decode(self, features_cl: torch.Tensor, query: torch.Tensor, shared: bool = False, lattice=None, pre_final: bool = False) -> torch.Tensor:
        """features [P, S, S, S, C]; query fp32 [P, M, 3] (or [M, 3] with shared=True) -> logits fp32 [P, M].
        lattice=(G0, G1, G2): the M queries are a dense C-order lattice (e.g. `VirtualGrid.get_grid_points`): same results, faster walk.
        pre_final: `features_cl` is `feature_volume(..., skip_final=True)`, i.e. the activation in front of the UNet's final 1x1x1
        convolution; that (linear) layer is applied to the sampled features inside the decoder kernel instead."""
        dev = _lib.require_gpu()
        self._sync()
        P = int(features_cl.shape[0])
        M = int(query.shape[-2])
        out = torch.empty(P, M, dtype=torch.float32, device=dev)
        d = self._dec
        final = self._final_conv() if pre_final else None
        fp = lambda a: a.ctypes.data
        query = query.contiguous()
        _lib.call("semabs_decoder", _lib.ptr(features_cl), _lib.ptr(query), _lib.farr(self.vg.offsets), _lib.farr(self.vg.scales),
                  _lib.iarr(self.vg.grid_shape), fp(d["w1"]), fp(d["b1"]), fp(d["w2"]), fp(d["b2"]), int(self.concat_xyz), P, M,
                  0 if shared else M * 3, self.vol_feature_extractor.f32, _lib.ptr(out), None if lattice is None else _lib.iarr(lattice),
                  fp(final[0]) if pre_final else None, fp(final[1]) if pre_final else None, _lib.stream())
        return out

    # ---- reference surface ---------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, input_xyz_pts, input_feature_pts, tsdf_vol, output_xyz_pts, **kwargs):
        """input_xyz_pts [B, N, 3], input_feature_pts [B, P, N, 1], tsdf_vol [B, S, S, S] (used iff "tsdf" is a network input),
        output_xyz_pts [B, P, M, 3] -> logits [B, P, M]   (net.py:383-439)."""
        dev = _lib.require_gpu()
        B, P, N = input_feature_pts.shape[:3]
        M = output_xyz_pts.shape[2]
        outs, feats = [], []
        for b in range(B):
            tv = None
            if self.with_tsdf and tsdf_vol is not None:
                # the reference pairs volume i = b * P + p of the b-major feature stack with `tsdf_vol.unsqueeze(1).repeat(P, 1, 1, 1, 1)[i]`,
                # i.e. tsdf_vol[i % B] (net.py:411-419) - for B > 1 and P > 1 not the volume's own scene; kept as it is
                tv = torch.stack([tsdf_vol[(b * P + p_) % B] for p_ in range(P)], dim=0)
            f = self.feature_volume(input_xyz_pts[b].to(dev, torch.float32), input_feature_pts[b].to(dev, torch.float32).reshape(P, N), tsdf_vol=tv)
            feats.append(f)
            outs.append(self.decode(f, output_xyz_pts[b].to(dev, torch.float32)))
        self.features_cl = torch.cat(feats, dim=0)
        return torch.stack(outs, dim=0).view(B, P, M)

    @property
    def visual_volumetric_features(self):
        """[B*P, C, S, S, S] fp32 like the reference's cached attribute (net.py:425-427)."""
        return None if self.features_cl is None else self.features_cl.permute(0, 4, 1, 2, 3).float()


class SemAbsVOOL(torch.nn.Module):
    """Drop-in (inference) for `net.SemAbsVOOL` (net.py:469-579) with `pointing_method="cosine_sim"` (the default,
    utils.py:87-91): two SemAbs3D feature volumes (target / reference saliency), the 35 -> 32 -> 64 spatial sampler and the
    cosine-similarity pointer against the relation embedding, fused in one HIP kernel (`semabs_vool_head`).
    A real nn.Module with the reference's keys: `completion_net.*`, `spatial_sampler.mlp.{0,2}.*`, `relation_embeddings.<name>`, `steps`."""

    RELATIONS = ["in", "behind", "in front of", "on the left of", "on the right of", "on", "[pad]"]

    def __init__(self, pointing_method: str, pointing_dim: int, device: str, decoder_concat_xyz_pts: bool, **kwargs):
        super().__init__()
        if pointing_method != "cosine_sim" or pointing_dim != 64 or not decoder_concat_xyz_pts:
            raise NotImplementedError("the HIP VOOL head covers pointing_method='cosine_sim', pointing_dim=64, decoder_concat_xyz_pts=True")
        from .module import register_tree
        from .weights import make_semabsvool_state_dict
        self.device = device
        self.completion_net = SemAbs3D(device=device, **kwargs)       # like the reference: built without the xyz concat
        self.pointing_temperature = 0.07
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        init = make_semabsvool_state_dict(seed=seed, pointing_dim=pointing_dim, unet_num_channels=kwargs["unet_num_channels"],
                                          unet_f_maps=kwargs["unet_f_maps"], unet_num_groups=kwargs["unet_num_groups"],
                                          unet_num_levels=kwargs["unet_num_levels"],
                                          pts_feat_extractor_hidden_dim=kwargs["pts_feat_extractor_hidden_dim"])
        register_tree(self, {k: v for k, v in init.items() if not k.startswith("completion_net.")}, buffers=("steps",))
        self._prm = None
        self._rel = {}
        self._sig = None

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        from .module import strip_module_prefix
        return super().load_state_dict(strip_module_prefix(state_dict), strict=strict, **kw)

    def _sync(self):
        dev = _lib.require_gpu()
        if any(p.device.type != "cuda" for p in self.parameters()):
            self.to(dev)
        own = [t for n, t in self.named_parameters() if not n.startswith("completion_net.")]
        sig = tuple((t.data_ptr(), t._version) for t in own)
        if sig == self._sig:
            return
        sd = {k: v.detach() for k, v in self.state_dict().items() if not k.startswith("completion_net.")}
        flat = [sd[f"spatial_sampler.mlp.{i}.{n}"].float().reshape(-1) for i, n in ((0, "weight"), (0, "bias"), (2, "weight"), (2, "bias"))]
        assert [t.numel() for t in flat] == [32 * 35, 32, 64 * 32, 64], "spatial sampler must be 35 -> 32 -> 64"
        self._prm = torch.cat(flat).to(dev).contiguous()
        self._rel = {k: sd["relation_embeddings." + k].float().to(dev) for k in self.RELATIONS}
        self._sig = sig

    def forward(self, output_xyz_pts, spatial_relation_name, input_xyz_pts, input_target_saliency_pts, input_reference_saliency_pts,
                tsdf_vol=None, **kwargs):
        """-> logits [B, D, M] (net.py:506-579).  Under `torch.no_grad()` or after `net.eval()`: the fused inference kernels (no graph).  In training
        mode with grad mode on and trainable parameters the result carries a `grad_fn`: `loss.backward()` on anything computed from it runs the hand-written backward
        pass of `semabs_amd.train` and leaves the gradients in `p.grad` of this module's parameters - what `utils.loop` needs
        (utils.py:404-417: `loss.backward(); clip_grad_norm_(net.parameters(), ...); optimizer.step()`), and what DistributedDataParallel's
        gradient hooks listen to."""
        # training mode + grad mode (train_vool.py / utils.loop); `net.eval()` (visualize.py:453, which calls the net WITHOUT torch.no_grad() and
        # detaches the result) and torch.no_grad() (utils.loop's validation branch, utils.py:424) both take the fused inference kernels
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return self._forward_train(dict(output_xyz_pts=output_xyz_pts, spatial_relation_name=spatial_relation_name, input_xyz_pts=input_xyz_pts,
                                            input_target_saliency_pts=input_target_saliency_pts,
                                            input_reference_saliency_pts=input_reference_saliency_pts))
        return self._forward_infer(output_xyz_pts, spatial_relation_name, input_xyz_pts, input_target_saliency_pts, input_reference_saliency_pts)

    def _engine(self):
        """The training engine bound to THIS module's parameters (built on first use; follows `.to()` / `load_state_dict` through the
        parameters' storage)."""
        _lib.require_gpu()
        if any(p.device.type != "cuda" for p in self.parameters()):
            self.to(_lib.require_gpu())
        key = tuple(p.data_ptr() for p in self.parameters())
        if getattr(self, "_train_engine", None) is None or self._train_key != key:
            from .train import VOOLTrainer
            net = self.completion_net
            u = net.vol_feature_extractor
            self.__dict__["_train_engine"] = VOOLTrainer(None, voxel_shape=net.vg.grid_shape, scene_bounds=[net.vg.lower_corner, net.vg.upper_corner],
                                                         unet_num_channels=net.C, unet_f_maps=u.f_maps, unet_num_groups=u.num_groups,
                                                         unet_num_levels=len(u.f_maps), pts_feat_extractor_hidden_dim=net.hidden,
                                                         pointing_temperature=self.pointing_temperature, module=self)
            self.__dict__["_train_key"] = key
        return self._train_engine

    def _forward_train(self, batch: dict):
        eng = self._engine()
        names = np.array(batch["spatial_relation_name"]).T
        used = sorted(set(names.reshape(-1).tolist()), key=self.RELATIONS.index)
        pnames = [k for k in eng.graph_params(used) if eng.params[k].requires_grad]
        return _VOOLFunction.apply(eng, batch, pnames, *[eng.params[k] for k in pnames])

    @torch.no_grad()
    def feature_volumes(self, xyz: torch.Tensor, target_saliency: torch.Tensor, reference_saliency: torch.Tensor):
        """The part of the forward pass that does not depend on the query points (net.py:521-549): point MLP + scatter + UNet on the target and on
        the reference saliency of every description.  xyz fp32 [N, 3], saliencies fp32 [D, N] -> (ft, fr), each [D, S, S, S, C] channels-last.
        `process_batch_vool` computes these ONCE per description and re-uses them for every chunk of query points."""
        dev = _lib.require_gpu()
        self._sync()
        net = self.completion_net
        xyz = xyz.to(dev, torch.float32).contiguous()
        D, N = int(target_saliency.shape[0]), int(xyz.shape[0])
        ft = net.feature_volume(xyz, target_saliency.to(dev, torch.float32).reshape(D, N).contiguous())
        fr = net.feature_volume(xyz, reference_saliency.to(dev, torch.float32).reshape(D, N).contiguous())
        return ft, fr

    @torch.no_grad()
    def point(self, ft: torch.Tensor, fr: torch.Tensor, relation_names, query_xyz: torch.Tensor) -> torch.Tensor:
        """Sampler + spatial MLP + cosine pointer (net.py:550-579) against cached feature volumes.  relation_names: one per description;
        query_xyz fp32 [D, M, 3] (or [M, 3], shared by all descriptions) -> logits fp32 [D, M]."""
        dev = _lib.require_gpu()
        self._sync()
        net = self.completion_net
        D = int(ft.shape[0])
        q = query_xyz.to(dev, torch.float32)
        if q.dim() == 2:
            q = q[None].expand(D, -1, -1)
        q = q.contiguous()
        M = int(q.shape[1])
        rel = torch.stack([self._rel[str(n)] for n in relation_names], dim=0).contiguous()
        out = torch.empty(D, M, dtype=torch.float32, device=dev)
        _lib.call("semabs_vool_head", _lib.ptr(ft), _lib.ptr(fr), _lib.ptr(q), _lib.ptr(self._prm), _lib.ptr(rel), _lib.farr(net.vg.offsets),
                  _lib.farr(net.vg.scales), _lib.iarr(net.vg.grid_shape), float(self.pointing_temperature), D, M,
                  net.vol_feature_extractor.f32, _lib.ptr(out), _lib.stream())
        return out

    @torch.no_grad()
    def _forward_infer(self, output_xyz_pts, spatial_relation_name, input_xyz_pts, input_target_saliency_pts, input_reference_saliency_pts):
        batch_size, num_descs = np.array(spatial_relation_name).T.shape
        M = int(output_xyz_pts.shape[-2])
        if input_xyz_pts.dim() == 2:                            # visualize.py:387-412 hands over the cloud without a batch dimension
            input_xyz_pts = input_xyz_pts[None]
        outs = []
        for b in range(batch_size):
            N = int(input_xyz_pts[b].shape[0])
            ft, fr = self.feature_volumes(input_xyz_pts[b], input_target_saliency_pts[b].reshape(num_descs, N),
                                          input_reference_saliency_pts[b].reshape(num_descs, N))
            outs.append(self.point(ft, fr, [spatial_relation_name[d][b] for d in range(num_descs)], output_xyz_pts[b].reshape(num_descs, M, 3)))
        return torch.stack(outs, dim=0).view(batch_size, num_descs, M)


class _VOOLFunction(torch.autograd.Function):
    """Autograd node of `SemAbsVOOL.forward`: forward = the HIP forward with a tape, backward = the hand-written backward pass against it
    (`VOOLTrainer.forward_tape` / `backward_tape`).  Its differentiable inputs are exactly the parameters the reference's autograd graph of
    this batch would contain (not `visual_sampler.*`, not relation embeddings no description names), so unused ones keep `grad = None` and
    DistributedDataParallel(find_unused_parameters=True) sees the same used / unused split as with the reference (utils.py:255-258)."""

    @staticmethod
    def forward(ctx, eng, batch, pnames, *params):
        logits, tape = eng.forward_tape(batch)
        ctx.eng, ctx.tape, ctx.pnames = eng, tape, pnames
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        if ctx.tape is None:
            raise RuntimeError("SemAbsVOOL: the backward tape is single-use (backward through the same forward twice is not supported)")
        grads = ctx.eng.backward_tape(ctx.tape, dlogits)
        ctx.tape = None
        return (None, None, None) + tuple(grads[k] for k in ctx.pnames)
