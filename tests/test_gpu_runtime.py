"""GPU tests of the runtime helpers behind bench.py and the tile-chunk pipeline: per-launch GEMM timing through the dispatch packet
(semabs_gemm_time_next) and CU-partitioned streams (semabs_stream_create_cumask)."""
import ctypes as C

import numpy as np
import pytest
import torch

import semabs_amd  # noqa: F401
from semabs_amd import _lib

pytestmark = pytest.mark.gpu


def test_gemm_dispatch_packet_timing():
    from semabs_amd.clip import vit
    M, N, K = 4096, 768, 768
    A = torch.randn(M, K, device="cuda").half()
    B = (torch.randn(N, K, device="cuda") * 0.05).half()
    Cm = torch.zeros(M, N, device="cuda", dtype=torch.float16)
    bias = torch.zeros(N, device="cuda")
    ref = (A.float() @ B.float().t()).cpu()
    timer = vit.GemmTimer()
    vit.GEMM_TIMER = timer
    try:
        for _ in range(5):
            vit.gemm(A, B, Cm, bias, M, N, K, K, K, N, 0)
    finally:
        vit.GEMM_TIMER = None
    s = timer.summary()
    assert s["launches"] == 5 and s["flops"] == 5 * 2.0 * M * N * K
    assert 0.0 < s["total_ms"] < 50.0                                   # five ~10 us kernels; the events were filled by the launches themselves
    # un-timed launches afterwards are unaffected, and the timed ones computed the product
    vit.gemm(A, B, Cm, bias, M, N, K, K, K, N, 0)
    torch.cuda.synchronize()
    np.testing.assert_allclose(Cm.float().cpu().numpy(), ref.numpy(), rtol=2e-3, atol=2e-3 * float(ref.abs().max()))


def _probe(stream_handle, n=1024):
    out = torch.zeros(n, 2, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    _lib.call("semabs_probe_placement", out.data_ptr(), n, stream_handle)
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    hw = o[:, 1]
    return o[:, 0], set(zip(o[:, 0].tolist(), ((hw >> 13) & 7).tolist(), ((hw >> 12) & 1).tolist(), ((hw >> 8) & 15).tolist()))


def test_cu_partitioned_streams_are_disjoint_and_keep_the_xcd_round_robin():
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    words = (n_cu + 31) // 32
    halves = []
    for k in range(2):
        bits = [0] * words
        for b in range(k * n_cu // 2, (k + 1) * n_cu // 2):
            bits[b // 32] |= 1 << (b % 32)
        h = C.c_void_p()
        _lib.call("semabs_stream_create_cumask", C.byref(h), (C.c_uint32 * words)(*bits), words)
        xcc, places = _probe(h)
        _lib.call("semabs_stream_destroy", h)
        assert len(places) == n_cu // 2, len(places)                   # the mask is honoured
        assert len(set(xcc.tolist())) == 8                             # every XCD keeps a share ...
        assert all(int(xcc[i]) == int(xcc[i + 8]) for i in range(64))  # ... and workgroup b still runs on XCD (b + const) % 8
        halves.append(places)
    assert not (halves[0] & halves[1])


def test_clip_wrapper_cu_partition_gives_identical_maps():
    from semabs_amd.clip import ClipWrapper, saliency_configs
    from semabs_amd.synth import synth_rgb
    from semabs_amd.weights import make_clip_state_dict
    ClipWrapper.engine = None
    ClipWrapper("ViT-B/32", state_dict=make_clip_state_dict("ViT-B/32", 0), chunk_tiles=16, max_labels=4)
    img = synth_rgb(120, 120, 3)
    cfg = saliency_configs["chefer_et_al"](120)
    w = torch.from_numpy(np.random.default_rng(0).standard_normal((2, 512)).astype(np.float32)).cuda()
    images = ClipWrapper.make_images(img, cfg["augmentations"])
    old = (ClipWrapper.n_streams, ClipWrapper.cu_partition, ClipWrapper._streams)
    try:
        outs = []
        for part in (False, True):
            ClipWrapper.n_streams, ClipWrapper.cu_partition, ClipWrapper._streams = 2, part, None
            outs.append(ClipWrapper.relevancy_device(images, w, cfg["cropping_augmentations"], cfg["horizontal_flipping"], cfg["positive_attn_only"]).clone())
        assert torch.equal(outs[0], outs[1])
    finally:
        ClipWrapper.n_streams, ClipWrapper.cu_partition, ClipWrapper._streams = old
