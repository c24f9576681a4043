"""End-to-end OVSSC inference for one RGB-D frame: relevancy -> geometry -> voxel inference, all on one GPU, nothing
copied to the host in between.  This is the recipe of `visualize.prep_data` + `process_batch_ovssc`
(visualize.py:61-154, 157-248) with one change that leaves the outputs identical for a fixed sub-sample: the
reference re-runs scatter + UNet for every 2^20-point chunk of query points (visualize.py:180-211); here the feature
volume is computed once per (scene, label) and only the decoder is evaluated per query point.

Stage map (reference file:line -> kernel file):
  ClipWrapper.get_clip_saliency * 50                visualize.py:93-101     tiles.hip, gemm.hip, vit.hip
  get_pointcloud -> float32, filter_pts_bounds      visualize.py:103-108    geometry.hip
  relevancies -= mean over labels; per-class points visualize.py:109-122    geometry.hip (gather_point_features)
  in-bounds compaction + np.random.choice(80 000)   visualize.py:103-108,193  geometry.hip (compact_subsample: counter-based seeded draw)
  SemAbs3D.forward                                  net.py:383-439          unet.hip
  TSDFVolume.integrate                              visualize.py:217-227    geometry.hip
  argmax / cutoff / frustum / tsdf mask             visualize.py:228-247    geometry.hip (ovssc_labels)
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib
from . import dist as sdist
from .clip import ClipWrapper, plan_tiles, saliency_configs
from .fusion import TSDFVolume
from .net import SemAbs3D
from .point_cloud import frustum_mask_device, frustum_params, pointcloud_device, pointcloud_params
from .synth import SCENE_BOUNDS

DEFAULT_NET_KWARGS = dict(voxel_shape=(128, 128, 128), scene_bounds=SCENE_BOUNDS, unet_num_channels=16, unet_f_maps=16,
                          unet_num_groups=8, unet_num_levels=6, network_inputs=["saliency"], use_pts_feat_extractor=True,
                          pts_feat_extractor_hidden_dim=128, reduce_method="max", output_dim=1, decoder_concat_xyz_pts=True,
                          batch_size=1)


@dataclass
class SceneResult:
    relevancies: torch.Tensor          # fp32 [L, H, W]   RAW get_clip_saliency maps (NOT x 50, NOT mean-subtracted: that is applied inside the point-feature gather)
    logits: torch.Tensor               # fp32 [L, S^3]    decoder outputs at the voxel centres
    labels: Optional[torch.Tensor]     # int32 [S^3]      argmax with the cutoff / frustum / tsdf mask, -1 = empty
    tsdf: Optional[torch.Tensor]       # fp32 [S, S, S]
    n_in: torch.Tensor                 # int64 [1] on the device: points of the depth image inside scene_bounds

    @property
    def n_in_bounds(self) -> int:
        """Host value (synchronises).  Zero in-bounds points is an error in the reference (np.random.choice on an empty population,
        visualize.py:193); the device path cannot raise at launch time, so it is raised here, on first read - and `ScenePipeline.run` has
        already poisoned that scene's `logits` (NaN) and `labels` (-1) on the device, so a caller that never reads this still cannot mistake the
        result for a valid one."""
        n = int(self.n_in.item())
        if n == 0:
            raise RuntimeError("no point of the depth image falls inside scene_bounds")
        return n


class ScenePipeline:
    def __init__(self, net: SemAbs3D, num_input_pts: int = 80000, config: str = "ours", subtract_mean: bool = True,
                 with_tsdf: bool = True, cutoff: float = -3.0, fold_final_conv: bool = True):
        self.net = net
        self.num_input_pts = int(num_input_pts)
        self.config = config
        self.subtract_mean = subtract_mean
        self.with_tsdf = with_tsdf
        self.cutoff = cutoff
        self.fold_final_conv = bool(fold_final_conv)
        self.dev = _lib.require_gpu()
        S0, S1, S2 = net.vg.grid_shape
        lc = np.asarray(net.vg.lower_corner, np.float32)
        uc = np.asarray(net.vg.upper_corner, np.float32)
        # VirtualGrid.get_grid_points (net.py:63-82): idx * (uc - lc) / (S - 1) + lc, fp32
        scales = (uc - lc) / (np.asarray([S0, S1, S2], np.float32) - np.float32(1))
        g = np.stack(np.meshgrid(np.arange(S0), np.arange(S1), np.arange(S2), indexing="ij"), axis=-1).astype(np.float32)
        self.grid_points_np = (g * scales + lc).reshape(-1, 3).astype(np.float32)
        self.grid_points = torch.from_numpy(self.grid_points_np).to(self.dev)
        self._grid_points64 = None

    def upload(self, scene: dict) -> dict:
        """Host -> HBM once, outside the timed region."""
        d = dict(scene)
        d["rgb_dev"] = torch.from_numpy(np.ascontiguousarray(scene["rgb"])).to(self.dev)
        d["depth_dev"] = torch.from_numpy(np.ascontiguousarray(scene["depth"], dtype=np.float32)).to(self.dev)
        # the camera of the frame travels with it: the argument blocks of the point-cloud, TSDF and frustum kernels (22 / 15 / 16 doubles) are
        # part of the upload, so a scene issues no host -> device copy of its own
        net = self.net
        bounds = np.array([net.vg.lower_corner, net.vg.upper_corner], np.float64)
        d["_pc_prm"] = pointcloud_params(scene["cam_intr"], scene["cam_pose"], bounds)
        d["_fr_prm"] = frustum_params(scene["cam_pose"], scene["cam_intr"])
        if self.with_tsdf:
            d["_tsdf_prm"] = self._tsdf_volume_spec()[2](scene["cam_pose"])
        d["_cam_key"] = self._cam_key(scene)                  # the cached blocks belong to THIS camera (checked on every use: _fresh_camera)
        return d

    @staticmethod
    def _cam_key(scene: dict) -> bytes:
        return np.asarray(scene["cam_pose"], np.float64).tobytes() + np.asarray(scene["cam_intr"], np.float64).tobytes()

    def _fresh_camera(self, scene: dict) -> dict:
        """An uploaded dict whose cam_pose / cam_intr were changed afterwards (a frame stream re-using one dict) must not run with the stale device argument
        blocks of `upload`: they are dropped, the kernels then build them from the current camera (one small host -> device copy each)."""
        if "_cam_key" in scene and scene["_cam_key"] != self._cam_key(scene):
            for k in ("_pc_prm", "_fr_prm", "_tsdf_prm", "_cam_key"):
                scene.pop(k, None)
        return scene

    def _tsdf_volume_spec(self):
        """(bounds [3, 2], voxel size, cam_pose -> device argument block) of the scene's TSDF volume (visualize.py:217-227)."""
        net = self.net
        S = net.vg.grid_shape[0]
        lo, hi = np.asarray(net.vg.lower_corner, np.float64), np.asarray(net.vg.upper_corner, np.float64)
        vs = (hi[0] - lo[0]) / S

        return np.stack([lo, hi], axis=1), vs, (lambda cam_pose, obs_weight=1.0: TSDFVolume.params_for(vs, cam_pose, obs_weight, self.dev))

    def run(self, scene: dict, w_text: torch.Tensor, seed: int = 0, jittered_images=None, images_dev: torch.Tensor | None = None) -> SceneResult:
        """scene: dict(rgb uint8 [H, W, 3], depth fp32 [H, W], cam_intr, cam_pose [+ rgb_dev / depth_dev]);
        w_text fp32 [L, E] on the GPU (zero-shot weights of the labels)."""
        return self.run_voxels(self.run_relevancy(scene, w_text, seed, jittered_images, images_dev))

    def run_sharded(self, scene: dict, w_text: torch.Tensor, seed: int = 0, jittered_images=None) -> SceneResult:
        """ONE scene split over the ranks of the default process group - the single-scene latency mode of SURVEY.md 8(e), next to the
        scene-sharded throughput mode `bench.py` measures.  Relevancy is tile-sharded: every rank runs a contiguous slice of the tile
        forwards for all labels and one all-gather of the per-rank slices of the per-tile relevances (RCCL) completes them everywhere (the aggregation is
        then replicated - it is cheap and deterministic).  Voxel inference is label-sharded: each rank runs the UNet / decoder on its
        slice of the label volumes and the logits are all-gathered.  Geometry is replicated.  World size 1 degenerates to `run`."""
        return self.run_voxels(self.run_relevancy(scene, w_text, seed, jittered_images, shard_tiles=True), shard_labels=True)

    def run_relevancy(self, scene: dict, w_text: torch.Tensor, seed: int = 0, jittered_images=None, images_dev: torch.Tensor | None = None,
                      geo_stream: torch.cuda.Stream | None = None, vit_stream: torch.cuda.Stream | None = None, shard_tiles: bool = False) -> dict:
        """First half of a scene: geometry (point cloud, in-bounds compaction, seeded sub-sample) and the relevancy maps.  With streams
        given, geometry runs on `geo_stream` (its host sync then waits for nothing else) and the ViT on `vit_stream`; the returned state
        carries the events `run_voxels` waits for - this is what lets a caller overlap scene i's voxel stage with scene i + 1's ViT."""
        dev, net = self.dev, self.net
        scene = self._fresh_camera(scene)
        H, W = scene["depth"].shape
        cfg = saliency_configs[self.config](H)
        cur = torch.cuda.current_stream()
        gs, vs = geo_stream or cur, vit_stream or cur
        # ---- geometry first (cheap; its results are needed by the voxel stage only) -------------------------------------------------------
        with torch.cuda.stream(gs):
            depth_dev = scene.get("depth_dev")
            if depth_dev is None:
                depth_dev = torch.from_numpy(np.ascontiguousarray(scene["depth"], dtype=np.float32)).to(dev)
            bounds = np.array([net.vg.lower_corner, net.vg.upper_corner], np.float64)
            xyz, mask = pointcloud_device(depth_dev, scene["cam_intr"], scene["cam_pose"], bounds, prm=scene.get("_pc_prm"))
            # in-bounds compaction + the seeded draw of num_input_pts points with replacement, on the device: the host never learns the point
            # count, so a scene has NO host synchronisation (several ranks sharing one host each used to block on torch.nonzero per scene)
            pix = torch.empty(H * W, dtype=torch.int64, device=dev)
            n_in = torch.empty(1, dtype=torch.int64, device=dev)
            sel = torch.empty(self.num_input_pts, dtype=torch.int64, device=dev)
            _lib.call("semabs_compact_subsample", _lib.ptr(mask), H * W, int(seed) & 0xFFFFFFFFFFFFFFFF, self.num_input_pts, _lib.ptr(pix),
                      _lib.ptr(n_in), _lib.ptr(sel), _lib.stream())
            geo_done = torch.cuda.Event()
            geo_done.record(gs)
        # ---- relevancy --------------------------------------------------------------------------------
        with torch.cuda.stream(vs):
            if images_dev is None:
                images_dev = ClipWrapper.make_images(scene["rgb"], cfg["augmentations"], jittered_images, img_dev=scene.get("rgb_dev"), seed=seed)
            rank, world = sdist.rank_world()
            if shard_tiles and world > 1:
                n_img = int(images_dev.shape[0])
                table, _ = plan_tiles(H, W, n_img, cfg["cropping_augmentations"])
                rel, _, scales = ClipWrapper.relevancy_device(images_dev, w_text, cfg["cropping_augmentations"], cfg["horizontal_flipping"],
                                                              cfg["positive_attn_only"], tile_range=sdist.shard_range(len(table), rank, world),
                                                              return_tiles=True)
                rel = sdist.allgather_tile_relevance(rel, len(table))
                maps = ClipWrapper.aggregate_device(rel, scales, n_img, H, W)
            else:
                maps = ClipWrapper.relevancy_device(images_dev, w_text, cfg["cropping_augmentations"], cfg["horizontal_flipping"],
                                                    cfg["positive_attn_only"])               # [L, H, W]
            maps_c = maps.contiguous()
            vit_done = torch.cuda.Event()
            vit_done.record(vs)
        return dict(scene=scene, L=int(w_text.shape[0]), H=H, W=W, depth_dev=depth_dev, xyz=xyz, sel=sel, n_in=n_in, maps=maps, maps_c=maps_c,
                    geo_done=geo_done, vit_done=vit_done, streams=(gs, vs))

    def run_voxels(self, state: dict, shard_labels: bool = False) -> SceneResult:
        """Second half: per-point features, voxel inference, TSDF and the label volume, on the current stream.
        shard_labels: each rank infers a contiguous slice of the label volumes, the logits are all-gathered."""
        dev, net = self.dev, self.net
        scene, L, H, W = state["scene"], state["L"], state["H"], state["W"]
        depth_dev, xyz, sel, maps, maps_c, n_in = state["depth_dev"], state["xyz"], state["sel"], state["maps"], state["maps_c"], state["n_in"]
        cur = torch.cuda.current_stream()
        cur.wait_event(state["geo_done"]); cur.wait_event(state["vit_done"])
        for t in (xyz, sel, maps, maps_c, depth_dev, n_in):
            t.record_stream(cur)                                                              # produced on other streams, read here
        st = _lib.stream()
        feat = torch.empty(L, self.num_input_pts, dtype=torch.float32, device=dev)
        xyz_sub = torch.empty(self.num_input_pts, 3, dtype=torch.float32, device=dev)
        _lib.call("semabs_gather_point_features", _lib.ptr(maps_c), _lib.ptr(sel), _lib.ptr(xyz), L, H * W, self.num_input_pts, 50.0,
                  int(self.subtract_mean), _lib.ptr(feat), _lib.ptr(xyz_sub), st)
        # ---- voxel inference ----------------------------------------------------------------------------
        # the UNet's final 1x1x1 convolution is folded into the decoder (applied to the sampled features): its output volume - 2 GB written
        # and read back at 128^3 x 16 labels - is never materialised on this path; SemAbs3D.forward keeps producing it
        rank, world = sdist.rank_world()
        if shard_labels and world > 1:
            per = (L + world - 1) // world                                                    # equal slices (the last one padded) for all_gather
            l0, l1 = min(L, rank * per), min(L, (rank + 1) * per)
            part = _lib.filled((per, self.grid_points.shape[0]), torch.float32, 0, dev)
            if l1 > l0:
                f_r = net.feature_volume(xyz_sub, feat[l0:l1].contiguous(), skip_final=self.fold_final_conv)
                part[:l1 - l0] = net.decode(f_r, self.grid_points, shared=True, lattice=net.vg.grid_shape, pre_final=self.fold_final_conv)
            logits = sdist.gather_results(part).reshape(world * per, -1)[:L].contiguous()
            net.features_cl = None
        else:
            features = net.feature_volume(xyz_sub, feat, skip_final=self.fold_final_conv)
            logits = net.decode(features, self.grid_points, shared=True, lattice=net.vg.grid_shape, pre_final=self.fold_final_conv)   # [L, S^3]
            net.features_cl = None if self.fold_final_conv else features
        tsdf = labels = None
        if self.with_tsdf:
            bnds, vs, _ = self._tsdf_volume_spec()
            tv = TSDFVolume(bnds.copy(), vs)
            rgb_dev = scene.get("rgb_dev")
            tv.integrate(rgb_dev if rgb_dev is not None else scene["rgb"], depth_dev, scene["cam_intr"], scene["cam_pose"], prm=scene.get("_tsdf_prm"))
            tsdf = tv._tsdf_vol
            if tuple(int(d) for d in tv._vol_dim) != tuple(net.vg.grid_shape):
                # ceil((hi - lo) / voxel_size) can come out as S + 1 for extents that are not exactly representable / not cubic; the label kernel
                # indexes the TSDF with the logits' strides, so a mismatch must not pass silently (the reference fails on the shape mismatch)
                raise RuntimeError(f"TSDF volume {tuple(tv._vol_dim)} does not match the voxel grid {tuple(net.vg.grid_shape)}")
            fr = self._frustum(scene, H, W)
            labels = torch.empty(logits.shape[1], dtype=torch.int32, device=dev)
            tsdf_flat = tsdf.reshape(-1)
            _lib.call("semabs_ovssc_labels", _lib.ptr(logits), _lib.ptr(fr), _lib.ptr(tsdf_flat), L, int(logits.shape[1]), float(self.cutoff),
                      _lib.ptr(labels), st)
        # An empty in-bounds cloud is an error in the reference (np.random.choice on an empty population, visualize.py:193).  The device path learns the
        # count without a host synchronisation, so it cannot raise here; it must not hand back plausible-looking output either (the sub-sample would be
        # 80 000 copies of pixel 0): the outputs are poisoned on the device - NaN logits, label -1 everywhere - and `n_in_bounds` raises on first read.
        _lib.call("semabs_poison_empty", _lib.ptr(n_in), _lib.ptr(logits), logits.numel(), _lib.ptr(labels), 0 if labels is None else labels.numel(), st)
        # `relevancies` = the raw relevancy maps; prep_data's x 50 / mean subtraction (visualize.py:100-112) lives in semabs_gather_point_features
        return SceneResult(relevancies=maps, logits=logits, labels=labels, tsdf=tsdf, n_in=n_in)

    def _frustum(self, scene, H, W):
        """In-frustum mask of the voxel-centre lattice for THIS scene's pose, computed on the device inside the step (no host round trip, no
        per-pose cache: every scene pays for it, like in the reference's process_batch_ovssc)."""
        if self._grid_points64 is None:
            self._grid_points64 = self.grid_points.double()          # the reference hands fp32 lattice points to an f64 routine (visualize.py:231-236)
        return frustum_mask_device(self._grid_points64, H, W, scene["cam_pose"], scene["cam_intr"], prm=scene.get("_fr_prm"))


def cu_mask_words(n_cus: int, select, total: int = 256):
    """32-bit mask words with bit i set for the first n_cus CU indices i (in increasing order) for which select(i) holds."""
    words = [0] * ((total + 31) // 32)
    left = n_cus
    for i in range(total):
        if left and select(i):
            words[i // 32] |= 1 << (i % 32)
            left -= 1
    return words


class CuPartition:
    """Two HIP streams with complementary CU masks (round 6): `voxel` confined to `voxel_cus` CUs, `vit` to the rest.  In throughput mode
    scene i's voxel stage (UNet / decoder / TSDF: HBM-bound, ~1.1 kW) runs on the small partition WHILE scene i + 1's relevancy stage (persistent GEMMs at the
    socket's power cap) runs on the large one - the library's persistent kernels size their grids to their stream's CUs (semabs_stream_cu_count).  Results are
    bit-identical to the sequential schedule (tests/test_gpu_scene.py).
    layout: "balanced" = the same number of CUs of every XCD (CU indices i with i % 32 < voxel_cus / 8: evenly spread whether the runtime enumerates CUs
    XCD-major or round-robin over the XCDs), "low" = CU indices 0 .. voxel_cus - 1."""

    def __init__(self, voxel_cus: int = 64, layout: str = "balanced"):
        import ctypes as C
        n = C.c_int(0)
        _lib.call("semabs_stream_cu_count", None, C.byref(n))
        total = int(n.value)
        assert 0 < voxel_cus < total and voxel_cus % 8 == 0 and layout in ("balanced", "low")
        per = voxel_cus // 8
        sel = (lambda i: i % (total // 8) < per) if layout == "balanced" else (lambda i: i < voxel_cus)
        vox = cu_mask_words(voxel_cus, sel, total)
        vit = [(~w) & 0xFFFFFFFF for w in vox]
        if total % 32:
            vit[-1] &= (1 << (total % 32)) - 1
        self.total, self.voxel_cus, self.layout = total, voxel_cus, layout
        self._handles = []
        self.streams = []
        for words in (vox, vit):
            arr = (C.c_uint * len(words))(*words)
            h = C.c_void_p()
            _lib.call("semabs_stream_create_cumask", arr, len(words), C.byref(h))
            self._handles.append(h)
            self.streams.append(torch.cuda.ExternalStream(h.value))
        self.voxel, self.vit = self.streams
        got = []
        for h in self._handles:
            _lib.call("semabs_stream_cu_count", h, C.byref(n))
            got.append(int(n.value))
        self.cus = tuple(got)                                 # what the runtime reports for the two streams

    def close(self):
        """Drain the two streams.  They are NOT destroyed: torch's caching allocator keeps per-stream block pools keyed by the stream handle and touches it again
        when those blocks are re-used (destroying the handles under it segfaulted in the GPU suite); two idle streams per partition live until process exit.
        (`semabs_stream_destroy` exists for callers that own their allocations.)"""
        torch.cuda.synchronize()


def build_default(arch: str = "ViT-B/16", precision: str = "exact", clip_seed: int = 0, net_seed: int = 3, chunk_tiles: int = 2448,
                  max_labels: int = 16, voxel: int = 128, text_tower: bool = True, **pipe_kwargs) -> ScenePipeline:
    """Seeded random-init weights of the released architectures (no checkpoints / network here)."""
    from .weights import make_clip_state_dict, make_semabs3d_state_dict
    ClipWrapper.engine = None
    ClipWrapper(arch, state_dict=make_clip_state_dict(arch, clip_seed, text_tower=text_tower), chunk_tiles=chunk_tiles, max_labels=max_labels)
    kw = dict(DEFAULT_NET_KWARGS, voxel_shape=(voxel, voxel, voxel), precision=precision)
    net = SemAbs3D(**kw)
    net.load_state_dict(make_semabs3d_state_dict(seed=net_seed))
    return ScenePipeline(net, **pipe_kwargs)
