"""Host-side buffer utilities of the C ABI (round 5): fills, replication, empty-cloud poisoning, strided un-flip average, EOT row gather - the
launches that replaced torch.zeros / torch.full / Tensor.repeat / masked_fill_ / slice copies / index_select on the hot path."""
import numpy as np
import pytest
import torch

import semabs_amd  # noqa: F401
from semabs_amd import _lib

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,off", [(1, 0), (3, 1), (4, 0), (1000, 3), (1 << 20, 0), ((1 << 22) + 7, 2)])
def test_fill_u32_patterns_and_ragged_ends(n, off):
    buf = torch.full((n + 8,), 7, dtype=torch.int32, device="cuda")
    view = buf[off:off + n]
    _lib.call("semabs_fill_u32", view.data_ptr(), n * 4, 0xDEADBEEF, _lib.stream())
    h = buf.cpu().numpy().view(np.uint32)
    assert (h[off:off + n] == 0xDEADBEEF).all() and (h[:off] == 7).all() and (h[off + n:] == 7).all()
    assert torch.equal(_lib.filled((5, 3), torch.float32, -1.0), -torch.ones(5, 3, device="cuda"))
    assert torch.equal(_lib.filled((7,), torch.int32, -1), torch.full((7,), -1, dtype=torch.int32, device="cuda"))
    z = _lib.filled((3, 2, 2), torch.float64, 0)
    assert z.dtype == torch.float64 and float(z.abs().sum()) == 0.0
    with pytest.raises(RuntimeError):
        _lib.call("semabs_fill_u32", view.data_ptr(), 6, 0, _lib.stream())


def test_replicate_equals_repeat():
    src = torch.randint(0, 255, (480, 480, 3), dtype=torch.uint8, device="cuda")
    dst = torch.empty(6, 480, 480, 3, dtype=torch.uint8, device="cuda")
    _lib.call("semabs_replicate", _lib.ptr(src), _lib.ptr(dst), src.numel(), 6, _lib.stream())
    assert torch.equal(dst, src[None].repeat(6, 1, 1, 1))


def test_poison_empty_only_when_the_cloud_is_empty():
    logits = torch.randn(3, 1000, device="cuda")
    labels = torch.randint(0, 3, (1000,), dtype=torch.int32, device="cuda")
    keep_l, keep_b = logits.clone(), labels.clone()
    n_in = torch.tensor([5], dtype=torch.int64, device="cuda")
    _lib.call("semabs_poison_empty", _lib.ptr(n_in), _lib.ptr(logits), logits.numel(), _lib.ptr(labels), labels.numel(), _lib.stream())
    assert torch.equal(logits, keep_l) and torch.equal(labels, keep_b)
    n_in.zero_()
    _lib.call("semabs_poison_empty", _lib.ptr(n_in), _lib.ptr(logits), logits.numel(), _lib.ptr(labels), labels.numel(), _lib.stream())
    assert bool(torch.isnan(logits).all()) and bool((labels == -1).all())
    logits2 = torch.randn(10, device="cuda")
    _lib.call("semabs_poison_empty", _lib.ptr(n_in), _lib.ptr(logits2), 10, None, 0, _lib.stream())
    assert bool(torch.isnan(logits2).all())


def test_unflip_average_rows_equals_the_two_buffer_form():
    L, N, g = 5, 37, 14
    both = torch.randn(L, 2 * N, g, g, device="cuda")
    a, b = both[:, :N].contiguous(), both[:, N:].contiguous()
    want = torch.empty_like(a)
    _lib.call("semabs_unflip_average", _lib.ptr(a), _lib.ptr(b), _lib.ptr(want), L * N, g, _lib.stream())
    got = torch.empty_like(a)
    _lib.call("semabs_unflip_average_rows", _lib.ptr(both), both.data_ptr() + N * g * g * 4, _lib.ptr(got), L, N, 2 * N, g, _lib.stream())
    assert torch.equal(got, want)
    assert torch.equal(want, (a + b.flip(-1)) / 2)


def test_eot_rows_gather_equals_argmax_index_select():
    B, T, D = 11, 77, 512
    tokens = torch.randint(1, 40000, (B, T), dtype=torch.int64)
    eot = torch.randint(2, T, (B,))
    for i in range(B):
        tokens[i, eot[i]] = 49407
        tokens[i, eot[i] + 1:] = 0
    x = torch.randn(B * T, D, device="cuda")
    dst = torch.empty(B, D, device="cuda")
    td = tokens.cuda()
    _lib.call("semabs_eot_rows_gather", _lib.ptr(td), _lib.ptr(x), _lib.ptr(dst), B, T, D, _lib.stream())
    rows = torch.arange(B) * T + tokens.argmax(dim=-1)
    assert torch.equal(dst.cpu(), x.cpu()[rows])
