"""GPU tests of the runtime helper behind bench.py: per-launch GEMM timing through the dispatch packet
(semabs_gemm_f16_ex with start / stop events)."""
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
