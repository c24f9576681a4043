"""GPU: the two fp16 MFMA GEMM kernels pinned per call (`semabs_gemm_f16_ex(kernel=...)`) against an fp64 product of the same fp16 operands.

* kernel 2 = `k_gemm8`, the phased 256 x 256 x 64 kernel that runs every large contraction of the ViT trunk (60 % of a scene): all five
  epilogues, M exactly one row panel / with a ragged last panel / the 256-tile batch size, N in {768, 2304, 3072}, K in {128, 768, 3072},
  strided A and B (lda > K: the K|V weight slices and the CLS-row gather use it), the row-remapped epilogue of the patch embedding;
* kernel 1 = `k_gemm_f16`, the ring kernel, on the same problems; the two must agree with each other to fp32 summation-order noise.

Tolerances (stated per output type): fp32 outputs 1e-5 of max|ref| (fp32 accumulation of exact fp16 products: measured ~1e-6);
fp16 outputs 5e-4 of max|ref| + 1e-3 relative (one fp16 rounding of the result = 2^-11)."""
import numpy as np
import pytest
import torch

import semabs_amd  # noqa: F401

pytestmark = pytest.mark.gpu

EPI_F16, EPI_GELU_F16, EPI_RESID_F32, EPI_F32, EPI_ROWMAP = 0, 1, 2, 3, 4


def _operands(M, N, K, lda, ldb, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    A = torch.randn(M, lda, device="cuda", generator=g).half()
    B = (torch.randn(N, ldb, device="cuda", generator=g) * 0.05).half()
    bias = torch.randn(N, device="cuda", generator=g)
    ref = A[:, :K].double() @ B[:, :K].double().T + bias.double()            # asymmetric operands: catches transposes
    return A, B, bias, ref


def _check32(got, ref, what):
    scale = float(ref.abs().max())
    err = float((got.double() - ref).abs().max())
    assert err <= 1e-5 * scale, f"{what}: max err {err:.3e} vs max|ref| {scale:.3e}"


def _check16(got, ref, what):
    scale = float(ref.abs().max())
    d = (got.double() - ref).abs()
    bad = d > (5e-4 * scale + 1e-3 * ref.abs())
    assert not bool(bad.any()), f"{what}: {int(bad.sum())} elements off, max err {float(d.max()):.3e} vs max|ref| {scale:.3e}"


SHAPES = [(2048, 768, 768), (2381, 768, 768), (2381, 2304, 768), (2381, 3072, 768), (2381, 768, 3072), (2049, 768, 128),
          (50432, 2304, 768), (50432, 768, 3072), (256, 768, 768), (300, 3072, 768),
          # K tiles per output tile nk = 2 / 3 / 5: the persistent kernel's cross-tile prologue with the shortest loop it admits (nk = 2) and the odd
          # counts that must fall back to one workgroup per tile
          (4100, 768, 128), (4100, 512, 192), (4100, 768, 320)]


@pytest.mark.parametrize("kernel", [2, 1, 2 | 512, 2 | 2048, 2 | 4096])          # 2: the phased kernel (deep schedule; persistent workgroups k_gemm8p for the fp16 outputs), | 512: its K = 32 ring schedule, | 2048: its round-3 PF schedule, | 4096: deep schedule with one workgroup per tile also for the fp16 outputs (NO persistent workgroups)
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_gemm_kernels_all_epilogues(kernel, M, N, K):
    from semabs_amd.clip.vit import gemm
    A, B, bias, ref = _operands(M, N, K, K, K, seed=M + N + K)
    tag = f"kernel {kernel} {M}x{N}x{K}"
    c32 = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
    gemm(A, B, c32, bias, M, N, K, K, K, N, EPI_F32, kernel=kernel)
    _check32(c32, ref, tag + " f32")
    gemm(A, B, c32, None, M, N, K, K, K, N, EPI_F32, kernel=kernel)                    # no bias
    _check32(c32, ref - bias.double(), tag + " f32 no bias")
    c16 = torch.full((M, N), float("nan"), dtype=torch.float16, device="cuda")
    gemm(A, B, c16, bias, M, N, K, K, K, N, EPI_F16, kernel=kernel)
    _check16(c16, ref, tag + " f16")
    gemm(A, B, c16, bias, M, N, K, K, K, N, EPI_GELU_F16, kernel=kernel)
    _check16(c16, ref * torch.sigmoid(1.702 * ref), tag + " gelu")
    c16.fill_(float("nan"))
    gemm(A, B, c16, None, M, N, K, K, K, N, EPI_F16, kernel=kernel)                    # fp16 output without a bias (the persistent kernel's bias row is optional)
    _check16(c16, ref - bias.double(), tag + " f16 no bias")
    g = torch.Generator(device="cuda").manual_seed(1)
    res = torch.randn(M, N, device="cuda", generator=g)
    c = res.clone()
    gemm(A, B, c, bias, M, N, K, K, K, N, EPI_RESID_F32, kernel=kernel)
    _check32(c, ref + res.double(), tag + " residual")
    # rows past M must not be touched (ragged last row panel): the output buffer carries a guard band
    guard = torch.full((M + 256, N), 7.0, dtype=torch.float32, device="cuda")
    gemm(A, B, guard, bias, M, N, K, K, K, N, EPI_F32, kernel=kernel)
    assert bool((guard[M:] == 7.0).all()), tag + ": wrote past row M"


@pytest.mark.parametrize("kernel", [2, 1])
def test_gemm_strided_operands_and_slices(kernel):
    """lda / ldb > K and a row slice of B (the last block's K|V weights are rows D..3D of in_proj_weight; its CLS rows are a strided A)."""
    from semabs_amd.clip.vit import gemm
    M, N, K, lda, ldb = 2300, 768, 768, 3 * 768, 768 + 64
    A, Bfull, _, _ = _operands(M, 3 * N, K, lda, ldb, seed=11)
    bias_full = torch.randn(3 * N, device="cuda")
    B = Bfull[N:2 * N]                                                          # contiguous row slice, row stride ldb
    bias = bias_full[N:2 * N].contiguous()
    ref = A[:, :K].double() @ B[:, :K].double().T + bias.double()
    c = torch.empty(M, N, dtype=torch.float32, device="cuda")
    gemm(A, B, c, bias, M, N, K, lda, ldb, N, EPI_F32, kernel=kernel)
    _check32(c, ref, f"kernel {kernel} strided")
    # strided C (ldc > N): a column block of a wider fp16 buffer
    wide = torch.full((M, 3 * N), 3.0, dtype=torch.float16, device="cuda")
    cview = wide[:, N:2 * N]
    from semabs_amd import _lib
    _lib.call("semabs_gemm_f16_ex", A.data_ptr(), B.data_ptr(), cview.data_ptr(), bias.data_ptr(), None, M, N, K, lda, ldb, 3 * N, EPI_F16, None,
              kernel, None, None, _lib.stream())
    _check16(wide[:, N:2 * N], ref, f"kernel {kernel} strided C")
    assert bool((wide[:, :N] == 3.0).all()) and bool((wide[:, 2 * N:] == 3.0).all())


@pytest.mark.parametrize("kernel", [2, 1])
def test_gemm_rowmap_epilogue(kernel):
    """epi 4: out row = (m / g_in) * g_out + g_off + m % g_in, plus addend[g_off + m % g_in] - the patch embedding writes token rows 1..T-1 of
    every tile and adds the positional embedding (model_explainability.py:325-343)."""
    from semabs_amd.clip.vit import gemm
    n, G, T, N, K = 11, 196, 197, 768, 768
    A, B, _, _ = _operands(n * G, N, K, K, K, seed=3)
    pos = torch.randn(T, N, device="cuda")
    out = torch.full((n * T, N), -7.0, dtype=torch.float32, device="cuda")
    gemm(A, B, out, None, n * G, N, K, K, K, N, EPI_ROWMAP, addend=pos, rowmap=(G, T, 1), kernel=kernel)
    ref = (A.double() @ B.double().T).view(n, G, N) + pos[1:].double()
    got = out.view(n, T, N)
    _check32(got[:, 1:], ref, f"kernel {kernel} rowmap")
    assert bool((got[:, 0] == -7.0).all())                                       # class-token rows are left to semabs_embed_finish


def test_gemm_quickgelu_vjp_epilogue():
    """epi 5 (phased kernel): C fp16 = (A B^T + bias) * table[m % n_x, :] with table = semabs_quickgelu_grad(pre) - the QuickGELU VJP of the ViT-L rollout
    (model_explainability.py:199-201 QuickGELU, differentiated by torch.autograd in clip_gradcam.py:90-97) fused into the W_pr^T GEMM; against
    fp64 and against the unfused pair (fp32 GEMM + semabs_gelu_bwd)."""
    from semabs_amd import _lib
    from semabs_amd.clip.vit import gemm
    for M, N, K, n_x in [(2381, 1024, 256, 700), (4100, 768, 768, 4100), (2048, 256, 128, 257)]:
        A, B, bias, ref = _operands(M, N, K, K, K, seed=M)
        g = torch.Generator(device="cuda").manual_seed(5)
        pre = torch.randn(n_x, N, device="cuda", generator=g) * 2.0
        x = pre.double()[torch.arange(M, device="cuda") % n_x]
        sg = torch.sigmoid(1.702 * x)
        ref5 = ref * (sg * (1.0 + 1.702 * x * (1.0 - sg)))
        out = torch.full((M, N), 7.0, dtype=torch.float16, device="cuda")
        table = torch.empty_like(pre)
        _lib.call("semabs_quickgelu_grad", _lib.ptr(pre), _lib.ptr(table), pre.numel(), _lib.stream())
        gemm(A, B, out, bias, M, N, K, K, K, N, 5, addend=table, rowmap=(n_x, 1, 0))
        _check16(out, ref5, f"epi 5 {M}x{N}x{K}")
        d32 = torch.empty(M, N, device="cuda")
        gemm(A, B, d32, bias, M, N, K, K, K, N, EPI_F32, kernel=2)
        two = torch.empty(M, N, dtype=torch.float16, device="cuda")
        _lib.call("semabs_gelu_bwd", _lib.ptr(d32), _lib.ptr(pre), _lib.ptr(two), M, N, n_x, 0, _lib.stream())
        assert float((out.float() - two.float()).abs().max()) <= 2e-3 * float(ref5.abs().max())      # two fp16 roundings of nearly equal fp32 values
    with pytest.raises(RuntimeError):                                            # small M: the ring kernel has no such epilogue
        gemm(A[:300], B, out[:300], bias, 300, N, K, K, K, N, 5, addend=table, rowmap=(n_x, 1, 0))


@pytest.mark.parametrize("epi", [EPI_F16, EPI_GELU_F16, EPI_RESID_F32])
@pytest.mark.parametrize("M,N,K", [(2381, 2304, 768), (50432, 3072, 768), (4100, 768, 128), (20000, 768, 3072)])
def test_gemm_schedules_bit_identical(epi, M, N, K):
    """Every schedule of the phased kernel - round-3 PF (| 2048), deep with one workgroup per tile (| 4096), deep + persistent workgroups (the default for
    the fp16 outputs) - runs the same fragments through the same MFMA order and the same fp32 epilogue order: the results must be EQUAL, not close
    (tools/gemm_biteq.py as a test).  The K = 32 ring schedule (| 512) sums in a different order and is only close."""
    from semabs_amd.clip.vit import gemm
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn(M, K, device="cuda", generator=g).half()
    B = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half()
    bias = torch.randn(N, device="cuda", generator=g)
    outs = []
    for kern in (2 | 2048, 2 | 4096, 2):
        C = (torch.full((M, N), float("nan"), dtype=torch.float16, device="cuda") if epi != EPI_RESID_F32 else
             torch.sin(torch.arange(M * N, device="cuda", dtype=torch.float32)).view(M, N).contiguous())
        gemm(A, B, C, bias, M, N, K, K, K, N, epi, kernel=kern)
        outs.append(C)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_gemm_kernels_agree_and_heuristic_picks_the_phased_kernel():
    """Both kernels compute the same sums (fp32 accumulation order differs); the heuristic (kernel 0) must give exactly the phased kernel's
    result for a trunk-sized problem and exactly the ring kernel's for a small one."""
    from semabs_amd.clip.vit import gemm
    for (M, N, K), same_as in (((4925, 2304, 768), 2), ((591, 2304, 768), 1)):
        A, B, bias, ref = _operands(M, N, K, K, K, seed=5)
        outs = {}
        for kernel in (0, 1, 2):
            c = torch.empty(M, N, dtype=torch.float32, device="cuda")
            gemm(A, B, c, bias, M, N, K, K, K, N, EPI_F32, kernel=kernel)
            outs[kernel] = c
        assert torch.equal(outs[0], outs[same_as])
        assert float((outs[1] - outs[2]).abs().max()) <= 1e-5 * float(ref.abs().max())


def test_gemm_ex_rejects_bad_options():
    from semabs_amd import _lib
    h = _lib.lib()
    a = torch.zeros(256, 128, dtype=torch.float16, device="cuda")
    c = torch.zeros(256, 128, dtype=torch.float32, device="cuda")
    rc = h.semabs_gemm_f16_ex(a.data_ptr(), a.data_ptr(), c.data_ptr(), None, None, 256, 128, 128, 128, 128, 128, 3, None, 5, None, None, None)
    assert rc == -1 and b"kernel must be" in h.semabs_last_error()
    ev = torch.cuda.Event(enable_timing=True)
    rc = h.semabs_gemm_f16_ex(a.data_ptr(), a.data_ptr(), c.data_ptr(), None, None, 256, 128, 128, 128, 128, 128, 3, None, 0, 1, None, None)
    assert rc == -1 and b"go together" in h.semabs_last_error()


# ---- LayerNorm folded into the GEMMs on either side of it (gemm.hip LNP / LNC, semabs_gemm_f16_ln, semabs_ln_rowstats) -------------------------------
@pytest.mark.parametrize("M,K", [(2048, 768), (2381, 768), (2381, 3072), (50432, 768)])
def test_gemm_layernorm_producer_epilogue(M, K):
    """LNP: the residual GEMM's epilogue also emits xg = fp16(x_new * gamma) (16-byte stores assembled from lane pairs) and per-row partial sums per
    256-column tile.  x_new must be BIT-identical to the plain residual epilogue's, xg bit-identical to fp16(x_new * gamma), the partials exact to fp32
    summation order, rows past M untouched, a second launch bit-identical (fixed reduction order, no atomics)."""
    from semabs_amd.clip.vit import gemm, gemm_ln
    N = 768
    A, B, bias, ref = _operands(M, N, K, K, K, seed=M + K)
    g = torch.Generator(device="cuda").manual_seed(2)
    res = torch.randn(M, N, device="cuda", generator=g) * 3 + 0.5
    gamma = torch.randn(N, device="cuda", generator=g)
    x0 = res.clone()
    gemm(A, B, x0, bias, M, N, K, K, K, N, EPI_RESID_F32, kernel=2)
    for reverse in (0, 1):
        x1 = torch.full((M + 256, N), 7.0, dtype=torch.float32, device="cuda")
        x1[:M] = res
        xg = torch.full((M + 256, N), 7.0, dtype=torch.float16, device="cuda")
        part = torch.full((M + 256, N // 256, 2), 7.0, dtype=torch.float32, device="cuda")
        gemm_ln(A, B, x1, bias, M, N, K, K, K, N, EPI_RESID_F32, xg=xg, gamma=gamma, part=part, reverse=reverse)
        assert torch.equal(x1[:M], x0)
        assert torch.equal(xg[:M], (x0 * gamma).half())
        assert bool((x1[M:] == 7.0).all()) and bool((xg[M:] == 7.0).all()) and bool((part[M:] == 7.0).all())
        t = x0.double().view(M, N // 256, 256)
        ref_p = torch.stack([t.sum(-1), (t * t).sum(-1)], dim=-1)
        assert float((part[:M].double() - ref_p).abs().max()) <= 2e-6 * float(ref_p.abs().max())
        x2 = res.clone()
        part2 = torch.empty(M, N // 256, 2, dtype=torch.float32, device="cuda")
        gemm_ln(A, B, x2, bias, M, N, K, K, K, N, EPI_RESID_F32, xg=xg[:M], gamma=gamma, part=part2, reverse=reverse)
        assert torch.equal(part2, part[:M])


@pytest.mark.parametrize("M,N,epi", [(2048, 2304, EPI_F16), (2381, 2304, EPI_F16), (2381, 3072, EPI_GELU_F16), (50432, 3072, EPI_GELU_F16), (50432, 2304, EPI_F16)])
def test_gemm_layernorm_consumer_epilogue_equals_layernorm_then_gemm(M, N, epi):
    """LNC + semabs_ln_rowstats: rstd * (xg W^T) - mean rstd colsum + (b + W beta) against LayerNorm (fp64) followed by the product, and against the
    unfused kernels (LayerNorm kernel -> fp16 -> GEMM): same result to the fp16 rounding of the two different A operands."""
    from semabs_amd.clip.vit import gemm, gemm_ln, layernorm, ln_rowstats
    K = D = 768
    g = torch.Generator(device="cuda").manual_seed(M + N)
    x = torch.randn(M, D, device="cuda", generator=g) * torch.linspace(0.3, 3.0, D, device="cuda") + 0.4       # channel-dependent spread, non-zero mean
    gamma = 1.0 + 0.2 * torch.randn(D, device="cuda", generator=g)
    beta = 0.1 * torch.randn(D, device="cuda", generator=g)
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half()
    b = torch.randn(N, device="cuda", generator=g)
    xg = (x * gamma).half()
    t = x.view(M, D // 256, 256)
    part = torch.stack([t.sum(-1), (t * t).sum(-1)], dim=-1).contiguous()
    rowac = torch.empty(M, 2, dtype=torch.float32, device="cuda")
    ln_rowstats(part, M, D // 256, D, rowac)
    mean, var = x.double().mean(-1), x.double().var(-1, unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    assert float((rowac[:, 0].double() - rstd).abs().max()) <= 1e-5 * float(rstd.max())
    assert float((rowac[:, 1].double() + mean * rstd).abs().max()) <= 1e-5 * float((mean * rstd).abs().max())
    colsum = (W.double() @ gamma.double()).float()
    bias_f = (b.double() + W.double() @ beta.double()).float()
    ln = torch.nn.functional.layer_norm(x.double(), (D,), gamma.double(), beta.double(), 1e-5)
    ref = ln @ W.double().T + b.double()
    if epi == EPI_GELU_F16:
        ref = ref * torch.sigmoid(1.702 * ref)
    scale = float(ref.abs().max())
    h = torch.empty(M, D, dtype=torch.float16, device="cuda")
    layernorm(x, gamma, beta, h, M, D)
    out_u = torch.empty(M, N, dtype=torch.float16, device="cuda")
    gemm(h, W, out_u, b, M, N, K, K, K, N, epi, kernel=2)
    err_u = float((out_u.double() - ref).abs().max())
    outs = []
    for reverse in (0, 1):
        out = torch.full((M + 256, N), 7.0, dtype=torch.float16, device="cuda")
        gemm_ln(xg, W, out, bias_f, M, N, K, K, K, N, epi, rowac=rowac, colsum=colsum, reverse=reverse)
        err = float((out[:M].double() - ref).abs().max())
        print(f"LN-folded GEMM {M}x{N} epi {epi} reverse {reverse}: L-inf {err:.3e} vs LayerNorm-then-GEMM kernels {err_u:.3e} (max|ref| {scale:.2f})")
        assert err <= 2.5e-3 * scale and err <= 2.0 * err_u + 5e-4 * scale          # as accurate as the unfused path (both round one A operand to fp16)
        assert bool((out[M:] == 7.0).all())
        outs.append(out)
    assert torch.equal(outs[0], outs[1])                                            # the tile order does not enter the result


@pytest.mark.parametrize("dc", [4.0, 20.0])
@pytest.mark.parametrize("N,epi", [(2304, EPI_F16), (3072, EPI_GELU_F16)])
def test_gemm_layernorm_fold_with_row_dc_offset_and_massive_channels(N, epi, dc):
    """The whole fold (producer epilogue -> semabs_ln_rowstats -> consumer) on rows the released checkpoints produce and random-init weights do not:
    every row shifted by its own DC offset of ~dc sigma of the bulk (|mean| >> spread) and three channels at 80 x the spread.  An un-centred
    fp16(x * gamma) copy spends its 11 bits on the offset; the producer therefore centres each row on the previous LayerNorm's mean of that row
    (`ln_center`), which this test provides with the error a real chain has (the mean before the residual update).  Bar: as accurate as the
    LayerNorm kernel followed by the plain GEMM (which centres exactly before rounding)."""
    from semabs_amd.clip.vit import gemm, gemm_ln, layernorm, ln_rowstats
    M, K = 4099, 768
    D = K
    g = torch.Generator(device="cuda").manual_seed(N + int(dc))
    x_prev = torch.randn(M, D, device="cuda", generator=g)
    x_prev += dc * (1.0 + 0.5 * torch.randn(M, 1, device="cuda", generator=g))              # per-row offset
    x_prev[:, [5, 300, 701]] += torch.tensor([80.0, -60.0, 70.0], device="cuda")
    gamma = (1.0 + 0.2 * torch.randn(D, device="cuda", generator=g)) * torch.exp(0.35 * torch.randn(D, device="cuda", generator=g))
    gamma[[5, 300, 701]] = torch.tensor([0.1, 0.3, 0.05], device="cuda")
    beta = 0.3 * torch.randn(D, device="cuda", generator=g)
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half()
    b = torch.randn(N, device="cuda", generator=g)
    # the residual update the producer GEMM applies: x = x_prev + A Wp^T + bp (a delta of the size one sub-layer adds, incl. its own DC component)
    Kp = 768
    A = (torch.randn(M, Kp, device="cuda", generator=g) * 0.5).half()
    Wp = (torch.randn(D, Kp, device="cuda", generator=g) * 0.03).half()
    bp = 0.3 + 0.1 * torch.randn(D, device="cuda", generator=g)
    colsum = (W.double() @ gamma.double()).float()
    bias_f = (b.double() + W.double() @ beta.double()).float()
    errs = {}
    for centred in (False, True):
        x = x_prev.clone()
        xg = torch.empty(M, D, dtype=torch.float16, device="cuda")
        part = torch.empty(M, D // 256, 2, dtype=torch.float32, device="cuda")
        center = x_prev.mean(-1).contiguous() if centred else None                            # what the previous LayerNorm knew about the row
        gemm_ln(A, Wp, x, bp, M, D, Kp, Kp, Kp, D, EPI_RESID_F32, xg=xg, gamma=gamma, part=part, center=center)
        rowac = torch.empty(M, 2, dtype=torch.float32, device="cuda")
        center2 = torch.empty(M, dtype=torch.float32, device="cuda")
        ln_rowstats(part, M, D // 256, D, rowac, center=center, center_out=center2)
        mean = x.double().mean(-1)
        assert float((center2.double() - mean).abs().max()) <= 2e-6 * float(mean.abs().max())
        rstd = (x.double().var(-1, unbiased=False) + 1e-5).rsqrt()
        assert float((rowac[:, 0].double() - rstd).abs().max()) <= 2e-5 * float(rstd.max())
        if centred:
            assert torch.equal(xg, ((x - center[:, None]) * gamma).half())
        ref = torch.nn.functional.layer_norm(x.double(), (D,), gamma.double(), beta.double(), 1e-5) @ W.double().T + b.double()
        if epi == EPI_GELU_F16:
            ref = ref * torch.sigmoid(1.702 * ref)
        scale = float(ref.abs().max())
        out = torch.empty(M, N, dtype=torch.float16, device="cuda")
        gemm_ln(xg, W, out, bias_f, M, N, K, K, K, N, epi, rowac=rowac, colsum=colsum)
        errs[centred] = float((out.double() - ref).abs().max())
    h = torch.empty(M, D, dtype=torch.float16, device="cuda")
    layernorm(x, gamma, beta, h, M, D)
    out_u = torch.empty(M, N, dtype=torch.float16, device="cuda")
    gemm(h, W, out_u, b, M, N, K, K, K, N, epi, kernel=2)
    err_u = float((out_u.double() - ref).abs().max())
    print(f"LayerNorm fold, rows with a {dc:.0f}-sigma DC offset + 3 massive channels, {M}x{N} epi {epi}: centred fold {errs[True] / scale:.3e}, un-centred fold "
          f"{errs[False] / scale:.3e}, LayerNorm-then-GEMM kernels {err_u / scale:.3e} of max|ref| {scale:.2f}")
    assert errs[True] <= 1.5 * err_u + 3e-4 * scale
    assert errs[True] < errs[False]


def test_trunk_with_and_without_layernorm_fold():
    """VisionRollout.trunk with the fold on (default) against the unfused launch sequence on the same weights and input: the residual stream after
    11 blocks agrees to the accumulated fp16-operand noise of either path."""
    from semabs_amd.clip.vit import VisionRollout
    from semabs_amd.weights import make_clip_state_dict
    eng = VisionRollout(make_clip_state_dict("ViT-B/32", 0, text_tower=False), chunk_tiles=48, max_labels=4)
    n = 48                                                   # 48 x 50 tokens = 2 400 rows >= 2 048: the phased kernel and the fold are active
    g = torch.Generator(device="cuda").manual_seed(0)
    patches = torch.randn(n * 49, 3 * 32 * 32, device="cuda", generator=g).half()
    outs = []
    for fold in (True, False):
        eng.ln_fold = fold
        eng.embed(patches, n)
        eng.trunk(n)
        outs.append(eng._workspace()["x"][: n * 50].clone())
    d = float((outs[0] - outs[1]).abs().max()) / float(outs[1].abs().max())
    print(f"trunk, LayerNorm fold on vs off: relative L-inf {d:.2e} of the residual stream")
    assert d <= 3e-3 and bool(torch.isfinite(outs[0]).all())


# ---- precision = "parity": low fp16 halves of q | k out of the QKV GEMM, attention scores from hi + lo pairs -------------------------------------------
@pytest.mark.parametrize("M", [2048, 2381, 50432])
@pytest.mark.parametrize("consumer", [False, True])
def test_gemm_low_halves_of_the_first_columns(M, consumer):
    """k_gemm8p<.., QKLO>: C is bit-identical to the kernel without the option; lo[m, n] = fp16(v - fp16(v)) for the first lo_cols columns, i.e. C + lo
    reproduces the fp32 value to 2^-22 relative; columns past lo_cols and rows past M are not written (tiles with and without low halves alternate on
    a workgroup: both positional wait counts are exercised)."""
    from semabs_amd.clip.vit import gemm, gemm_ln, ln_rowstats
    N, K, LO = 2304, 768, 1536
    g = torch.Generator(device="cuda").manual_seed(M)
    A = torch.randn(M, K, device="cuda", generator=g).half()
    B = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half()
    bias = torch.randn(N, device="cuda", generator=g)
    kw = {}
    if consumer:
        part = torch.rand(M, 3, 2, device="cuda", generator=g) * 100 + 50
        part[..., 1] = part[..., 0] ** 2 / 256 + 300.0
        rowac = torch.empty(M, 2, device="cuda")
        ln_rowstats(part, M, 3, K, rowac)
        kw = dict(rowac=rowac, colsum=torch.randn(N, device="cuda", generator=g))
    for reverse in (0, 1):
        c0 = torch.full((M + 256, N), 7.0, dtype=torch.float16, device="cuda")
        if consumer:
            gemm_ln(A, B, c0, bias, M, N, K, K, K, N, EPI_F16, reverse=reverse, **kw)
        else:
            gemm(A, B, c0, bias, M, N, K, K, K, N, EPI_F16, kernel=2 | (reverse << 8))
        c1 = torch.full((M + 256, N), 7.0, dtype=torch.float16, device="cuda")
        lo = torch.full((M + 256, LO + 256), 7.0, dtype=torch.float16, device="cuda")
        gemm_ln(A, B, c1, bias, M, N, K, K, K, N, EPI_F16, reverse=reverse, lo=lo, lo_cols=LO, **kw)
        assert torch.equal(c1, c0)
        assert bool((lo[M:] == 7.0).all()) and bool((lo[:, LO:] == 7.0).all())
        ref = A.double() @ B.double().T
        if consumer:
            ref = ref * rowac[:, :1].double() + rowac[:, 1:].double() * kw["colsum"].double() + bias.double()
        else:
            ref = ref + bias.double()
        both = c1[:M, :LO].double() + lo[:M, :LO].double()
        e_hi = float((c1[:M, :LO].double() - ref[:, :LO]).abs().max()) / float(ref.abs().max())
        e_both = float((both - ref[:, :LO]).abs().max()) / float(ref.abs().max())
        print(f"M {M} consumer {consumer}: fp16 output {e_hi:.2e}, hi + lo {e_both:.2e} of max|ref|")
        assert e_both < 4e-6 and e_both < e_hi / 50


def test_attention_split_scores():
    """semabs_attention_split against an fp64 attention on q = q_hi + q_lo, k = k_hi + k_lo, and against semabs_attention on the hi halves alone: the
    split result must be closer to the fp64 result of the UNROUNDED q, k by an order of magnitude."""
    from semabs_amd import _lib
    n, T, H, D = 5, 197, 12, 768
    g = torch.Generator(device="cuda").manual_seed(3)
    qkv32 = torch.randn(n, T, 3 * D, device="cuda", generator=g) * 1.5
    qkv32[..., :D] *= 0.125 * 4.0                              # sharpened scores: the softmax is far from uniform
    hi = qkv32.half()
    lo = (qkv32[..., :2 * D] - hi[..., :2 * D].float()).half().contiguous()
    out_s = torch.empty(n, T, D, dtype=torch.float16, device="cuda")
    out_h = torch.empty_like(out_s)
    _lib.call("semabs_attention_split", _lib.ptr(hi), _lib.ptr(lo), _lib.ptr(out_s), None, n, T, H, 64, 3 * D, 2 * D, 0, _lib.stream())
    _lib.call("semabs_attention", _lib.ptr(hi), _lib.ptr(out_h), None, n, T, H, 64, 3 * D, 0, _lib.stream())
    q, k = (qkv32[..., i * D:(i + 1) * D].double().view(n, T, H, 64).transpose(1, 2) for i in (0, 1))
    v = hi[..., 2 * D:].double().view(n, T, H, 64).transpose(1, 2)
    ref = (torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ v).transpose(1, 2).reshape(n, T, D)
    e_s = float((out_s.double() - ref).abs().max())
    e_h = float((out_h.double() - ref).abs().max())
    print(f"attention vs fp64 on unrounded q, k: hi only {e_h:.3e}, hi + lo {e_s:.3e}")
    assert e_s < e_h / 3 and e_s < 5e-3                       # measured 2.0e-2 / 3.7e-3 (what is left: fp16 P, V and the fp16 output, |o| up to ~5)
