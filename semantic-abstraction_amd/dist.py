"""Sharding helpers for multi-GPU runs (one process per GPU, `torch.distributed`; backend "nccl" = RCCL over xGMI on
the GPU node, "gloo" in the CPU tests).

The path shards three ways, none of which needs a collective on the data path itself:
  * scene sharding   (config 4: 64 scenes over 8 GPUs)        - ranks own disjoint scenes; results stay on the rank or
                                                                 are gathered once at the end (`gather_results`).
  * tile sharding    (single-scene latency)                    - each rank runs a contiguous slice of the (tile, flip)
                                                                 forwards for all labels; ONE all-gather of the per-rank
                                                                 slices of the per-tile relevances [L, N, g, g] completes
                                                                 them on every rank before aggregation.
  * label sharding   (voxel inference)                         - 16 label volumes / 8 GPUs = 2 each; logits all-gathered.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


# Per-process collective accounting: payload bytes this rank contributed, host-side wall time AND device-side time of every
# collective issued through this module, by kind - bench.py prints it per rank in every N > 1 line, so the first real multi-GPU run explains itself.
# device_ms: HIP events on the COMPUTE stream around the call.  For a blocking collective (the all-gathers) that is the time the compute stream spent in it
# (issue -> data complete), i.e. exposed communication; for the asynchronous gradient buckets the pair around `finish()`'s waits ("all_reduce_wait") is the
# part of the all-reduce the backward pass did NOT hide - `exposed_ms` in bench.py's line - while the buckets' own entries show issue cost only.
STATS: dict = {}
_EVENTS: list = []          # (kind, start event, stop event) not yet resolved (elapsed_time needs the events to have completed)
_EVENT_POOL: list = []


def _account(kind: str, nbytes: int, seconds: float, calls: int = 1) -> None:
    d = STATS.setdefault(kind, {"calls": 0, "bytes": 0, "host_ms": 0.0, "device_ms": 0.0})
    d["calls"] += calls
    d["bytes"] += int(nbytes)
    d["host_ms"] += seconds * 1e3


class _DeviceTimed:
    """with _DeviceTimed(kind, on_cuda): ... - a pair of timing events on the current stream around the body; resolved lazily by stats_snapshot()."""

    def __init__(self, kind: str, on: bool):
        self.kind, self.on = kind, bool(on) and torch.cuda.is_available()

    def __enter__(self):
        if self.on:
            self.a = _EVENT_POOL.pop() if _EVENT_POOL else torch.cuda.Event(enable_timing=True)
            self.b = _EVENT_POOL.pop() if _EVENT_POOL else torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.b.record()
            _EVENTS.append((self.kind, self.a, self.b))
        return False


def _resolve_events() -> None:
    if not _EVENTS:
        return
    torch.cuda.synchronize()
    for kind, a, b in _EVENTS:
        STATS.setdefault(kind, {"calls": 0, "bytes": 0, "host_ms": 0.0, "device_ms": 0.0})["device_ms"] += a.elapsed_time(b)
        _EVENT_POOL.extend((a, b))
    _EVENTS.clear()


def reset_stats() -> None:
    _resolve_events()
    STATS.clear()


def stats_snapshot() -> dict:
    _resolve_events()
    return {k: dict(v, host_ms=round(v["host_ms"], 3), device_ms=round(v["device_ms"], 3)) for k, v in STATS.items()}


def rank_world() -> Tuple[int, int]:
    """(rank, world size) of the default process group; (0, 1) when torch.distributed is not initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous balanced slice [lo, hi) of n items for `rank` (first n % world ranks get one extra)."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_list(items: Sequence, rank: int, world: int) -> List:
    lo, hi = shard_range(len(items), rank, world)
    return list(items[lo:hi])


def _all_gather_list(local: torch.Tensor) -> List[torch.Tensor]:
    """dist.all_gather of equally shaped tensors; gloo has no device collectives, so CUDA tensors are staged through the host there
    (the CPU / single-GPU tests); with "nccl" (= RCCL) the device buffers go over xGMI directly."""
    import time
    world = dist.get_world_size()
    t0 = time.perf_counter()
    with _DeviceTimed("all_gather", local.is_cuda):
        if local.is_cuda and dist.get_backend() == "gloo":
            host = local.detach().cpu().contiguous()
            out = [torch.empty_like(host) for _ in range(world)]
            dist.all_gather(out, host)
            res = [o.to(local.device) for o in out]
        else:
            res = [torch.empty_like(local) for _ in range(world)]
            dist.all_gather(res, local.contiguous())
    _account("all_gather", local.numel() * local.element_size(), time.perf_counter() - t0)
    return res


def _all_gather_into(recv: torch.Tensor, send: torch.Tensor) -> None:
    """recv [world, *send.shape] <- every rank's `send`, in place (pre-allocated arenas: no allocation, no list of tensors).  RCCL: all_gather_into_tensor on the
    device buffers; gloo (CPU / single-GPU tests) has no such collective for device memory: staged through the host."""
    import time
    t0 = time.perf_counter()
    with _DeviceTimed("all_gather", send.is_cuda):
        if send.is_cuda and dist.get_backend() == "gloo":
            host = send.detach().cpu()
            out = [torch.empty_like(host) for _ in range(dist.get_world_size())]
            dist.all_gather(out, host)
            recv.copy_(torch.stack(out))
        else:
            dist.all_gather_into_tensor(recv, send)
    _account("all_gather", send.numel() * send.element_size(), time.perf_counter() - t0)


_ARENAS: dict = {}


def _arena(key, shape, dtype, device, zero=False) -> torch.Tensor:
    """A buffer that lives as long as the process, keyed by use and shape (the latency path re-uses the same shapes every scene)."""
    k = (key, tuple(shape), dtype, str(device))
    if k not in _ARENAS:
        _ARENAS[k] = (torch.zeros if zero else torch.empty)(*shape, dtype=dtype, device=device)
    return _ARENAS[k]


def allgather_tile_relevance(rel_local: List[torch.Tensor], n_tiles: int) -> List[torch.Tensor]:
    """Tile sharding: rank r computed rel_local[pass] = [L, hi_r - lo_r, g, g] for its contiguous slice `shard_range(n_tiles, r, world)`
    of the tile table.  ONE all-gather of the (padded-to-equal) slices of all passes completes [L, n_tiles, g, g] per pass on every rank:
    each rank sends only what it computed (1 / world of the 2 x 15 MB at the BASELINE shape) - the zero-padded all-reduce this replaces sent
    the full tensors and added zeros."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return rel_local
    rank, world = dist.get_rank(), dist.get_world_size()
    passes = len(rel_local)
    L, g = int(rel_local[0].shape[0]), int(rel_local[0].shape[2])
    ranges = [shard_range(n_tiles, r, world) for r in range(world)]
    pad = max(hi - lo for lo, hi in ranges)
    lo, hi = ranges[rank]
    assert all(int(r.shape[1]) == hi - lo for r in rel_local), "rel_local must hold exactly this rank's tile slice"
    # No ATen kernel on this path (round 6): send / receive arenas allocated once per shape, pack and unpack as pitched copies of the library
    # (semabs_copy2d: L rows per pass), one all_gather_into_tensor.  The returned tensors are views of an arena: valid until the next call with the same shapes
    # (the caller aggregates them right away on the same stream).  Host tensors (the CPU tests) take the slice / cat form of the same layout.
    t = rel_local[0]
    gg = g * g
    if not t.is_cuda:
        send = torch.zeros(passes, L, pad, g, g, dtype=t.dtype)
        for p in range(passes):
            send[p, :, : hi - lo] = rel_local[p]
        parts = _all_gather_list(send)
        return [torch.cat([parts[r][p, :, : ranges[r][1] - ranges[r][0]] for r in range(world)], dim=1).contiguous() for p in range(passes)]
    from . import _lib
    assert t.dtype == torch.float32
    send = _arena("tile_send", (passes, L, pad * gg), t.dtype, t.device, zero=True)
    recv = _arena("tile_recv", (world, passes, L, pad * gg), t.dtype, t.device)
    out = _arena("tile_out", (passes, L, n_tiles * gg), t.dtype, t.device)
    st = _lib.stream()
    for p in range(passes):
        src = rel_local[p]
        assert src.is_contiguous()
        _lib.call("semabs_copy2d", _lib.ptr(src), (hi - lo) * gg * 4, _lib.ptr(send[p]), pad * gg * 4, (hi - lo) * gg * 4, L, st)
    _all_gather_into(recv, send)
    for r in range(world):
        rlo, rhi = ranges[r]
        if rhi == rlo:
            continue
        for p in range(passes):
            dst = out[p].view(-1)[rlo * gg:]
            _lib.call("semabs_copy2d", _lib.ptr(recv[r, p]), pad * gg * 4, _lib.ptr(dst), n_tiles * gg * 4, (rhi - rlo) * gg * 4, L, st)
    return [out[p].view(L, n_tiles, g, g) for p in range(passes)]


def gather_results(local: torch.Tensor) -> torch.Tensor:
    """All-gather equally shaped per-rank results along a new leading dim (label / scene shards -> rank 0 and all)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local[None]
    return torch.stack(_all_gather_list(local), dim=0)


def allreduce_flat_gradients(flat: torch.Tensor, n_flags: int = 0) -> Tuple[float, torch.Tensor | None]:
    """Data-parallel gradient exchange of the VOOL training step (what DistributedDataParallel's bucketed all-reduce does in the
    reference, utils.py:255-258 with find_unused_parameters=True): ONE sum all-reduce of the flat gradient buffer - a single large
    collective suits xGMI's per-link-bound rings better than 100+ per-tensor ones.  The last `n_flags` entries of `flat` are per-
    parameter "used on this rank" flags travelling in the same message; -> (scale = 1 / world to apply to the gradients, host bool
    tensor "used on any rank" or None when n_flags == 0).  World size 1 / no process group: nothing is sent."""
    import time
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    if world > 1:
        t0 = time.perf_counter()
        with _DeviceTimed("all_reduce", flat.is_cuda):
            _allreduce_sum(flat)
        _account("all_reduce", flat.numel() * flat.element_size(), time.perf_counter() - t0)
    flags = None
    if n_flags:
        flags = flat[flat.numel() - n_flags:].detach().cpu() > 0
    return 1.0 / world, flags


def _allreduce_sum(t: torch.Tensor, async_op: bool = False):
    """Sum all-reduce of a contiguous tensor in place.  RCCL ("nccl"): device buffers over xGMI, optionally asynchronous (the work object is returned;
    the collective runs on the process group's own stream behind everything queued on the current stream).  gloo (CPU / single-GPU tests): staged
    through the host, always synchronous."""
    if t.is_cuda and dist.get_backend() == "gloo":
        host = t.detach().cpu()
        dist.all_reduce(host)
        t.copy_(host)
        return None
    return dist.all_reduce(t, async_op=True) if async_op else dist.all_reduce(t)


class BucketedAllReduce:
    """Data-parallel gradient exchange overlapped with the backward pass (what the reference gets from DistributedDataParallel's bucketed all-reduce,
    utils.py:255-258): the flat gradient buffer is cut into a few CONTIGUOUS buckets in the order the backward pass completes them; `ready(i)` starts
    bucket i's sum all-reduce asynchronously (RCCL: on the process group's stream, behind the kernels already queued on the compute stream), `finish()`
    reduces whatever was not started and makes the compute stream wait for all of them.  Summing disjoint slices separately or the whole buffer at once
    gives the same numbers (element-wise sums; bit-identical for two ranks, where the order of the two addends cannot matter).
    World size 1 / no process group: every call is a no-op."""

    def __init__(self, flat: torch.Tensor, ranges: Sequence[Tuple[int, int]]):
        self.flat = flat
        self.ranges = [(int(a), int(b)) for a, b in ranges]
        assert all(a < b for a, b in self.ranges)
        cover = sorted(self.ranges)
        assert cover[0][0] == 0 and cover[-1][1] == flat.numel() and all(cover[i][1] == cover[i + 1][0] for i in range(len(cover) - 1)), \
            "buckets must tile the flat buffer"
        self.started = [False] * len(self.ranges)
        self.works = []
        self.exposed_ms = 0.0

    @property
    def active(self) -> bool:
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def begin_step(self) -> None:
        # Collectives of a previous forward_backward that never reached finish() (a loss-only call, an exception in between) are still running on the process
        # group's stream: the new step's writes to the flat buffer must not race with them, and the ranks' collective sequences must stay aligned
        for w in self.works:
            w.wait()
        self.started = [False] * len(self.ranges)
        self.works = []

    def ready(self, i: int) -> None:
        import time
        if not self.active or self.started[i]:
            return
        a, b = self.ranges[i]
        t0 = time.perf_counter()
        w = _allreduce_sum(self.flat[a:b], async_op=True)
        _account("all_reduce_bucket", (b - a) * self.flat.element_size(), time.perf_counter() - t0)
        self.started[i] = True
        if w is not None:
            self.works.append(w)

    def finish(self) -> float:
        """Start the buckets nobody announced, wait for all; -> 1 / world (the scale the caller applies to the summed gradients)."""
        import time
        if not self.active:
            return 1.0
        for i in range(len(self.ranges)):
            self.ready(i)
        t0 = time.perf_counter()
        with _DeviceTimed("all_reduce_wait", self.flat.is_cuda):                 # device_ms of this entry = the all-reduce time the backward pass did not hide
            for w in self.works:
                w.wait()                                     # (RCCL: the compute stream waits for the collective's stream; the host does not block)
        self.works = []
        _account("all_reduce_wait", 0, time.perf_counter() - t0)
        return 1.0 / dist.get_world_size()
