// Transformer kernels around the GEMMs: LayerNorm, fused multi-head attention (MFMA, softmax in registers),
// CLS-row attention of the last block (keeps the softmax row), embedding glue, and the element-wise pieces of
// the closed-form attention x gradient rollout.
//
// Replaces (reference file:line):
//   LayerNorm (fp32)                       CLIP/clip/model_explainability.py:188-194
//   multi_head_attention_forward           CLIP/clip/auxiliary.py:207, 260-340   (q scaling is folded into W_q)
//   class/positional embedding             CLIP/clip/model_explainability.py:329-344
//   ClipGradcam.forward / interpret        CLIP/clip/clip_gradcam.py:58-132  (autograd replaced by the analytic VJP)
//   token embedding / EOT gather           CLIP/clip/model_explainability.py:469-482
//   zeroshot_classifier normalise + mean   CLIP/clip/clip_gradcam.py:24-27
#include "semabs_common.h"

// "Split" fp16 outputs (round 6): a row of W values is stored as [hi | lo] with a pitch of 2 W fp16 - hi = fp16(v), lo = fp16(v - hi) - so that the
// GEMM that consumes it (K = 2 W against the weight matrix repeated twice along K) sees the value to ~2^-22 instead of 2^-11.  Used for the operands of
// the LAST block and of the VJP chain behind it (n or L n rows - a percent of the trunk's flops): with trained-checkpoint statistics (peaked softmax:
// CLS scores up to ~60) the fp16 rounding of the last LayerNorm output alone moves the kept softmax row by 2e-3 (tests/test_gpu_trained_stats.py).
__device__ __forceinline__ void store_split4(f16* row, int W, int c0, float a, float b, float c, float d) {
    f16x4 h; h[0] = (f16)a; h[1] = (f16)b; h[2] = (f16)c; h[3] = (f16)d;
    f16x4 l; l[0] = (f16)(a - (float)h[0]); l[1] = (f16)(b - (float)h[1]); l[2] = (f16)(c - (float)h[2]); l[3] = (f16)(d - (float)h[3]);
    *reinterpret_cast<f16x4*>(row + c0) = h;
    *reinterpret_cast<f16x4*>(row + W + c0) = l;
}

// =================================================================================================
// LayerNorm: one wave per row, row held in registers (D <= 1024, D % 256 == 0), two-pass mean / variance.
// =================================================================================================
template <int VPL, int OUT_MODE>   // VPL = float4 vectors per lane (D = 256 * VPL); OUT_MODE 0 = fp16, 1 = fp32, 2 = fp16 [hi | lo] (pitch 2 D)
__global__ void k_layernorm(const float* __restrict__ x, long ld_in, const float* __restrict__ gamma,
                            const float* __restrict__ beta, void* __restrict__ out, long M, float eps, int order, float* __restrict__ mean_out) {
    const int lane = threadIdx.x & 63;
    // order: 0 = rows in dispatch order; 1 / 2 = XCD-contiguous runs walked forwards / backwards (zigzag with the neighbouring GEMMs)
    const long blk = order ? semabs_xcd_item((int)blockIdx.x, (int)gridDim.x, order == 2) : (long)blockIdx.x;
    const long row = blk * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= M) return;
    const int D = 256 * VPL;
    const float* xr = x + row * ld_in;
    float4 v[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr + (i * 64 + lane) * 4));     // streaming: 346 -> 318 us per pass
        v[i] = make_float4(t[0], t[1], t[2], t[3]);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) / D;
    if (mean_out && lane == 0) mean_out[row] = mean;       // the row centre the LayerNorm-fold producer behind this pass starts from (gemm.hip LNP)
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        q += (a * a + b * b) + (c * c + d * d);
    }
    const float rstd = rsqrtf(wave_sum(q) / D + eps);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c0 = (i * 64 + lane) * 4;
        float4 gm = *reinterpret_cast<const float4*>(gamma + c0);
        float4 bt = *reinterpret_cast<const float4*>(beta + c0);
        float4 o;
        o.x = (v[i].x - mean) * rstd * gm.x + bt.x; o.y = (v[i].y - mean) * rstd * gm.y + bt.y;
        o.z = (v[i].z - mean) * rstd * gm.z + bt.z; o.w = (v[i].w - mean) * rstd * gm.w + bt.w;
        if (OUT_MODE == 1) {
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + row * D + c0) = o;
        } else if (OUT_MODE == 2) {
            store_split4(reinterpret_cast<f16*>(out) + row * 2 * D, D, c0, o.x, o.y, o.z, o.w);
        } else {
            f16x4 h; h[0] = (f16)o.x; h[1] = (f16)o.y; h[2] = (f16)o.z; h[3] = (f16)o.w;
            *reinterpret_cast<f16x4*>(reinterpret_cast<f16*>(out) + row * D + c0) = h;
        }
    }
}

// Two LayerNorms in one pass (round 6): out1 = LN(x; g1, b1) in fp32 (may alias x) and out2 = fp16 LN(out1; g2, b2) - ln_pre followed by block 0's ln_1
// (model_explainability.py:345, 232-255) - with the same per-lane partial sums and wave reductions as two k_layernorm launches: bit-identical to them, one
// read of the 1.48 GB row set less.  mean_out (optional) = the row means of out1 (the first centre of the LayerNorm-fold chain).
template <int VPL>
__global__ void k_layernorm2(const float* __restrict__ x, const float* __restrict__ g1, const float* __restrict__ b1, float* __restrict__ out1,
                             const float* __restrict__ g2, const float* __restrict__ b2, f16* __restrict__ out2, long M, float eps, int order, float* __restrict__ mean_out) {
    const int lane = threadIdx.x & 63;
    const long blk = order ? semabs_xcd_item((int)blockIdx.x, (int)gridDim.x, order == 2) : (long)blockIdx.x;
    const long row = blk * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= M) return;
    const int D = 256 * VPL;
    const float* xr = x + row * D;
    float4 v[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr + (i * 64 + lane) * 4));
        v[i] = make_float4(t[0], t[1], t[2], t[3]);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    float mean = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        q += (a * a + b * b) + (c * c + d * d);
    }
    float rstd = rsqrtf(wave_sum(q) / D + eps);
    s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c0 = (i * 64 + lane) * 4;
        const float4 gm = *reinterpret_cast<const float4*>(g1 + c0), bt = *reinterpret_cast<const float4*>(b1 + c0);
        v[i].x = (v[i].x - mean) * rstd * gm.x + bt.x; v[i].y = (v[i].y - mean) * rstd * gm.y + bt.y;
        v[i].z = (v[i].z - mean) * rstd * gm.z + bt.z; v[i].w = (v[i].w - mean) * rstd * gm.w + bt.w;
        *reinterpret_cast<float4*>(out1 + row * D + c0) = v[i];
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    mean = wave_sum(s) / D;
    if (mean_out && lane == 0) mean_out[row] = mean;
    q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        q += (a * a + b * b) + (c * c + d * d);
    }
    rstd = rsqrtf(wave_sum(q) / D + eps);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c0 = (i * 64 + lane) * 4;
        const float4 gm = *reinterpret_cast<const float4*>(g2 + c0), bt = *reinterpret_cast<const float4*>(b2 + c0);
        f16x4 h;
        h[0] = (f16)((v[i].x - mean) * rstd * gm.x + bt.x); h[1] = (f16)((v[i].y - mean) * rstd * gm.y + bt.y);
        h[2] = (f16)((v[i].z - mean) * rstd * gm.z + bt.z); h[3] = (f16)((v[i].w - mean) * rstd * gm.w + bt.w);
        *reinterpret_cast<f16x4*>(out2 + row * D + c0) = h;
    }
}
extern "C" int semabs_layernorm2(const float* x, const float* g1, const float* b1, float* out1, const float* g2, const float* b2, void* out2, long M, int D, float eps,
                                 int order, float* mean_out, void* stream) {
    if (M == 0) return SEMABS_OK;
    SEMABS_REQUIRE(x && g1 && b1 && out1 && g2 && b2 && out2 && M > 0, "semabs_layernorm2: bad args");
    SEMABS_REQUIRE(D % 256 == 0 && D >= 256 && D <= 1024 && order >= 0 && order <= 2, "semabs_layernorm2: D must be 256..1024 step 256");
    dim3 grid(semabs_cdiv(M, 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    switch (D / 256) {
        case 1: hipLaunchKernelGGL(k_layernorm2<1>, grid, block, 0, s, x, g1, b1, out1, g2, b2, (f16*)out2, M, eps, order, mean_out); break;
        case 2: hipLaunchKernelGGL(k_layernorm2<2>, grid, block, 0, s, x, g1, b1, out1, g2, b2, (f16*)out2, M, eps, order, mean_out); break;
        case 3: hipLaunchKernelGGL(k_layernorm2<3>, grid, block, 0, s, x, g1, b1, out1, g2, b2, (f16*)out2, M, eps, order, mean_out); break;
        case 4: hipLaunchKernelGGL(k_layernorm2<4>, grid, block, 0, s, x, g1, b1, out1, g2, b2, (f16*)out2, M, eps, order, mean_out); break;
    }
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// Residual add fused into the LayerNorm pass: x[row, :] += delta[row, :] (fp16 GEMM output), x written back, out = fp16 LayerNorm(x).
// The fp32 read-modify-write of the residual stream leaves the GEMM epilogue (where it serialises with the MFMA work of the tile: the
// out-proj GEMM spent 37 of its 80 us there) for this streaming kernel, which already reads the row.
template <int VPL>
__global__ void k_add_layernorm(float* __restrict__ x, const f16* __restrict__ delta, const float* __restrict__ gamma,
                                const float* __restrict__ beta, f16* __restrict__ out, long M, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= M) return;
    const int D = 256 * VPL;
    float* xr = x + row * D;
    const f16* dr = delta + row * D;
    float4 v[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c0 = (i * 64 + lane) * 4;
        v[i] = *reinterpret_cast<const float4*>(xr + c0);
        const f16x4 d = *reinterpret_cast<const f16x4*>(dr + c0);
        v[i].x += (float)d[0]; v[i].y += (float)d[1]; v[i].z += (float)d[2]; v[i].w += (float)d[3];
        *reinterpret_cast<float4*>(xr + c0) = v[i];
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    if (!out) return;                                               // add only (uniform)
    const float mean = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        q += (a * a + b * b) + (c * c + d * d);
    }
    const float rstd = rsqrtf(wave_sum(q) / D + eps);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c0 = (i * 64 + lane) * 4;
        float4 gm = *reinterpret_cast<const float4*>(gamma + c0);
        float4 bt = *reinterpret_cast<const float4*>(beta + c0);
        f16x4 h;
        h[0] = (f16)((v[i].x - mean) * rstd * gm.x + bt.x); h[1] = (f16)((v[i].y - mean) * rstd * gm.y + bt.y);
        h[2] = (f16)((v[i].z - mean) * rstd * gm.z + bt.z); h[3] = (f16)((v[i].w - mean) * rstd * gm.w + bt.w);
        *reinterpret_cast<f16x4*>(out + row * D + c0) = h;
    }
}
// x fp32 [M, D] += delta fp16 [M, D] in place; out fp16 [M, D] = LayerNorm(x) (out = NULL: the addition only)
extern "C" int semabs_add_layernorm(float* x, const void* delta, const float* gamma, const float* beta, void* out, long M, int D,
                                    float eps, void* stream) {
    if (M == 0) return SEMABS_OK;
    SEMABS_REQUIRE(x && delta && M > 0 && (!out || (gamma && beta)), "semabs_add_layernorm: bad args");
    SEMABS_REQUIRE(D % 256 == 0 && D >= 256 && D <= 1024, "semabs_add_layernorm: D must be 256..1024 step 256");
    dim3 grid(semabs_cdiv(M, 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    switch (D / 256) {
        case 1: hipLaunchKernelGGL(k_add_layernorm<1>, grid, block, 0, s, x, (const f16*)delta, gamma, beta, (f16*)out, M, eps); break;
        case 2: hipLaunchKernelGGL(k_add_layernorm<2>, grid, block, 0, s, x, (const f16*)delta, gamma, beta, (f16*)out, M, eps); break;
        case 3: hipLaunchKernelGGL(k_add_layernorm<3>, grid, block, 0, s, x, (const f16*)delta, gamma, beta, (f16*)out, M, eps); break;
        case 4: hipLaunchKernelGGL(k_add_layernorm<4>, grid, block, 0, s, x, (const f16*)delta, gamma, beta, (f16*)out, M, eps); break;
    }
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// x fp32 rows (stride ld_in elements) -> out [M, D] dense, fp16 (out_f32 = 0) or fp32 (1; may alias x when ld_in == D); mean_out (optional) [M] = the row means
// out_f32 bit 3 (value 8): fp16 [hi | lo] rows of pitch 2 D (see store_split4)
extern "C" int semabs_layernorm(const float* x, const float* gamma, const float* beta, void* out, long M, int D,
                                float eps, int out_f32, long ld_in, float* mean_out, void* stream) {
    if (M == 0) return SEMABS_OK;
    SEMABS_REQUIRE(x && gamma && beta && out && M > 0, "semabs_layernorm: bad args");
    SEMABS_REQUIRE(D % 256 == 0 && D >= 256 && D <= 1024 && ld_in % 4 == 0, "semabs_layernorm: D must be 256..1024 step 256");
    dim3 grid(semabs_cdiv(M, 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    const int order = (out_f32 >> 1) & 3;                   // bits 1-2 of out_f32: 0 dispatch order, 1 / 2 XCD-contiguous runs forwards / backwards
#define LN_CASE(V)                                                                                              \
    case V:                                                                                                     \
        if (out_f32 & 1) hipLaunchKernelGGL((k_layernorm<V, 1>), grid, block, 0, s, x, ld_in, gamma, beta, out, M, eps, order, mean_out); \
        else if (out_f32 & 8) hipLaunchKernelGGL((k_layernorm<V, 2>), grid, block, 0, s, x, ld_in, gamma, beta, out, M, eps, order, mean_out); \
        else hipLaunchKernelGGL((k_layernorm<V, 0>), grid, block, 0, s, x, ld_in, gamma, beta, out, M, eps, order, mean_out);        \
        break;
    switch (D / 256) { LN_CASE(1) LN_CASE(2) LN_CASE(3) LN_CASE(4) }
#undef LN_CASE
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// =================================================================================================
// LayerNorm folded into the GEMMs (gemm.hip, LNP / LNC): the producer GEMM leaves per-row partial sums (sum x, sum x^2) per 256-column tile;
// this turns them into the two per-row numbers the consumer GEMM's epilogue applies: (rstd, -mean * rstd).   model_explainability.py:188-194
// The tiles' fp32 sums are combined in fp64 (the variance is a difference of two large numbers when |mean| >> sigma).
// Centred form (round 6): the producer subtracted a per-row centre c (the row mean the PREVIOUS LayerNorm of this row saw) before forming xg and the
// partials, so the sums are those of y = x - c: rstd is unchanged, the consumer's second coefficient becomes -mean(y) * rstd (the part of the mean the
// fp16 copy still carries), and center_out = c + mean(y) = the row's true mean, the centre of the next producer on this row.
// =================================================================================================
__global__ void k_ln_rowstats(const float* __restrict__ part, long M, int ntile, int D, float eps, float* __restrict__ rowac,
                              const float* __restrict__ center_in, float* __restrict__ center_out) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= M) return;
    double s1 = 0.0, s2 = 0.0;
    for (int t = 0; t < ntile; ++t) { s1 += (double)part[(r * ntile + t) * 2]; s2 += (double)part[(r * ntile + t) * 2 + 1]; }
    const double mean = s1 / D;
    double var = s2 / D - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    rowac[r * 2] = rstd;
    rowac[r * 2 + 1] = (float)(-mean) * rstd;
    if (center_out) center_out[r] = (float)((center_in ? (double)center_in[r] : 0.0) + mean);
}
extern "C" int semabs_ln_rowstats(const float* part, long M, int ntile, int D, float eps, float* rowac, const float* center_in, float* center_out, void* stream) {
    if (M == 0) return SEMABS_OK;
    SEMABS_REQUIRE(part && rowac && ntile > 0 && D > 0, "semabs_ln_rowstats: bad args");
    hipLaunchKernelGGL(k_ln_rowstats, dim3(semabs_cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream, part, M, ntile, D, eps, rowac, center_in, center_out);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// =================================================================================================
// CLS rows of the token matrix: x[n, 0, :] = class_embedding + pos[0, :]   (patch rows come from the GEMM epilogue)
// =================================================================================================
__global__ void k_embed_cls(float* __restrict__ x, const float* __restrict__ cls, const float* __restrict__ pos0,
                            int n, int T, int D) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)n * D) return;
    int d = (int)(i % D); long t = i / D;
    x[t * T * D + d] = cls[d] + pos0[d];
}

extern "C" int semabs_embed_finish(float* x, const float* cls, const float* pos, int n, int T, int D, void* stream) {
    if (n == 0) return SEMABS_OK;
    SEMABS_REQUIRE(x && cls && pos && n > 0 && T > 0 && D > 0, "semabs_embed_finish: bad args");
    hipLaunchKernelGGL(k_embed_cls, dim3(semabs_cdiv((long)n * D, 256)), dim3(256), 0, (hipStream_t)stream, x, cls, pos, n, T, D);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// =================================================================================================
// Fused attention, head dim 64.  One workgroup (4 waves) per (sequence, head): K [TP, 64] (swizzled) and
// V^T [64, TP + 4] in LDS; each wave owns 32-query blocks.  S^T = K Q^T with v_mfma_f32_32x32x16_f16 so that a
// lane holds one query's scores (row max/sum = in-lane + one lane^32 exchange), softmax in fp32, P stays in
// registers as the B operand of O^T = V^T P^T (the MFMA k-slot permutation is matched on the V^T side).
// =================================================================================================
__device__ __forceinline__ int kswz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// SPLIT (precision = "parity"): q and k arrive as fp16 hi + lo pairs (qk_lo [n_seq, T, ld_lo] holds the low halves, q | k at column
// offsets 0 / D; written by the QKV GEMM's epilogue) and the scores are S = q_hi k_hi + q_hi k_lo + q_lo k_hi in fp32 - the rounding of q and k to fp16
// is the largest single term of the relevancy maps' deviation from the fp32 reference (tests/test_vit_precision_budget.py: 1.5e-3 of 2.05e-3 per tile).
// Costs a third LDS image (K_lo) - one workgroup per CU instead of two - and two more score MFMAs per k-step.
template <int NKB, bool CAUSAL, bool SPLIT = false>   // number of 32-key blocks: TP = 32 * NKB
__global__ __launch_bounds__(64 * ((NKB > 4) ? 8 : 4), SPLIT ? 1 : 2) void k_attention(const f16* __restrict__ qkv, f16* __restrict__ out,
                                                   float* __restrict__ stats, int T, int H, int ld, int D, int order,
                                                   const f16* __restrict__ qk_lo = nullptr, int ld_lo = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TP = 32 * NKB;
    constexpr int VS = TP + 4;                    // V^T row stride (elements): keeps ds_read_b64 8-byte aligned
    char* sK = smem;
    f16* sV = reinterpret_cast<f16*>(smem + TP * 128);
    [[maybe_unused]] char* sKl = smem + TP * 128 + 64 * VS * 2;      // SPLIT: K_lo, same swizzled layout as K
    constexpr int NWAVE = (NKB > 4) ? 8 : 4;        // one 32-query block per wave when there are more than 4 of them
    constexpr int NTHR = 64 * NWAVE;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wg = order ? semabs_xcd_item((int)blockIdx.x, (int)gridDim.x, order == 2) : (int)blockIdx.x;    // (see k_layernorm: zigzag order)
    const int seq = wg / H, h = wg % H;
    const f16* base = qkv + (long)seq * T * ld + h * 64;
    [[maybe_unused]] const f16* base_lo = SPLIT ? qk_lo + (long)seq * T * ld_lo + h * 64 : nullptr;

    // Staging: ALL of a thread's K / V rows are requested before the first LDS store (the loop used to alternate load - store, i.e. one
    // ~1-2 us HBM round trip per iteration, 4-7 of them in a row: SQ_WAIT_ANY was 48 % of the wave cycles), and the query fragments of the
    // wave's first block are requested before the barrier as well.
    constexpr int NIT_ST = (TP * 8 + NTHR - 1) / NTHR;
    f16x8 kvs[NIT_ST], vvs[NIT_ST];
    [[maybe_unused]] f16x8 kls[SPLIT ? NIT_ST : 1];
#pragma unroll
    for (int it = 0; it < NIT_ST; ++it) {
        const int c = tid + it * NTHR;
        const int row = c >> 3, kc = c & 7;
        if (c < TP * 8 && row < T) {
            kvs[it] = *reinterpret_cast<const f16x8*>(base + (long)row * ld + D + kc * 8);
            vvs[it] = *reinterpret_cast<const f16x8*>(base + (long)row * ld + 2 * D + kc * 8);
            if constexpr (SPLIT) kls[it] = *reinterpret_cast<const f16x8*>(base_lo + (long)row * ld_lo + D + kc * 8);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) { kvs[it][e] = (f16)0.f; vvs[it][e] = (f16)0.f; if constexpr (SPLIT) kls[it][e] = (f16)0.f; }
        }
    }
    const int ql = lane & 31, hi = lane >> 5;
    f16x8 fq0[4];
    [[maybe_unused]] f16x8 fq0l[SPLIT ? 4 : 1];
    {
        const int q = wid * 32 + ql;
        const int qc = q < T ? q : T - 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            fq0[ks] = *reinterpret_cast<const f16x8*>(base + (long)qc * ld + (ks * 2 + hi) * 8);
            if constexpr (SPLIT) fq0l[ks] = *reinterpret_cast<const f16x8*>(base_lo + (long)qc * ld_lo + (ks * 2 + hi) * 8);
        }
    }
#pragma unroll
    for (int it = 0; it < NIT_ST; ++it) {
        const int c = tid + it * NTHR;
        const int row = c >> 3, kc = c & 7;
        if (c < TP * 8) {
            *reinterpret_cast<f16x8*>(sK + kswz(row, kc)) = kvs[it];
            if constexpr (SPLIT) *reinterpret_cast<f16x8*>(sKl + kswz(row, kc)) = kls[it];
#pragma unroll
            for (int e = 0; e < 8; ++e) sV[(kc * 8 + e) * VS + row] = vvs[it][e];
        }
    }
    __syncthreads();

    int koff[4];                                   // swizzled K-row offsets: row-block independent
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = kswz(ql, ks * 2 + hi);
#pragma unroll 1
    for (int qb = wid; qb < NKB; qb += NWAVE) {
        const int q = qb * 32 + ql;
        const int qc = q < T ? q : T - 1;
        f16x8 fq[4];
        [[maybe_unused]] f16x8 fql[SPLIT ? 4 : 1];
        if (qb == wid) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { fq[ks] = fq0[ks]; if constexpr (SPLIT) fql[ks] = fq0l[ks]; }
        } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                fq[ks] = *reinterpret_cast<const f16x8*>(base + (long)qc * ld + (ks * 2 + hi) * 8);
                if constexpr (SPLIT) fql[ks] = *reinterpret_cast<const f16x8*>(base_lo + (long)qc * ld_lo + (ks * 2 + hi) * 8);
            }
        }
        // One score block (4 MFMAs) at a time instead of keeping all NKB blocks in registers (112 VGPRs at T = 197): exponentiate, feed the
        // un-normalised probabilities straight into P.V; O is scaled by 1 / sum at the end.  ~100 VGPRs -> two workgroups per CU.
        auto scores = [&](int kb, f32x16& sc, float init) {      // init = -(reference maximum): the MFMA accumulator does the subtraction
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[r] = init;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                f16x8 fk = *reinterpret_cast<const f16x8*>(sK + kb * 4096 + koff[ks]);
                if constexpr (SPLIT) {                       // the two small products first, the large one last (fp32 accumulation)
                    const f16x8 fkl = *reinterpret_cast<const f16x8*>(sKl + kb * 4096 + koff[ks]);
                    sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fkl, fq[ks], sc, 0, 0, 0);
                    sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fk, fql[ks], sc, 0, 0, 0);
                }
                sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fk, fq[ks], sc, 0, 0, 0);
            }
            // sc[r] = score(key = kb*32 + (r&3) + 8*(r>>2) + 4*hi, query = q); only a block that reaches past T (or, causal, past the query) needs masking
            if (CAUSAL || kb * 32 + 31 >= T) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const bool dead = key >= T || (CAUSAL && key > q);
                    sc[r] = dead ? -INFINITY : sc[r];
                }
            }
        };
        // Single pass with a LAZY running maximum: block kb's scores come out of the MFMA already shifted by the current reference m (the
        // accumulator starts at -m); m is only raised - and O and the running sum rescaled - when some query of the wave sees a score more
        // than 8 above it, i.e. almost never after the first block (un-normalised probabilities up to e^8 = 2981 are exact enough in fp16
        // and the accumulations are fp32; the final 1 / sum removes the reference).  Saves the separate max pass: 28 of the 84 MFMAs and
        // 28 of the K fragment reads per query block.
        float mref = 0.f;                                   // per query (the two lanes of a query keep the same value)
        float sum = 0.f;
        f32x16 o[2];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
#pragma unroll 1
        for (int kb = 0; kb < NKB; ++kb) {
            f32x16 sc;
            scores(kb, sc, -mref);
            float bm = sc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) bm = fmaxf(bm, sc[r]);
            bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
            const bool raise = kb == 0 || bm > 8.f;          // first block: adopt its maximum as the reference
            if (__ballot(raise) != 0) {                      // wave-uniform branch
                const float d = raise ? bm : 0.f;            // (a fully masked block cannot occur: key 0 is always live)
                const float alpha = __builtin_amdgcn_exp2f(-d * 1.44269504088896340736f);
                mref += d;
                sum *= alpha;
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[r] -= d;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) { sc[r] = __builtin_amdgcn_exp2f(sc[r] * 1.44269504088896340736f); sum += sc[r]; }
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                f16x8 p;
#pragma unroll
                for (int j = 0; j < 8; ++j) p[j] = (f16)sc[hf * 8 + j];
                const int kbase = kb * 32 + hf * 16 + 4 * hi;
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const f16* vr = sV + (db * 32 + ql) * VS + kbase;
                    f16x4 v0 = *reinterpret_cast<const f16x4*>(vr);
                    f16x4 v1 = *reinterpret_cast<const f16x4*>(vr + 8);
                    f16x8 fv;
                    fv[0] = v0[0]; fv[1] = v0[1]; fv[2] = v0[2]; fv[3] = v0[3];
                    fv[4] = v1[0]; fv[5] = v1[1]; fv[6] = v1[2]; fv[7] = v1[3];
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fv, p, o[db], 0, 0, 0);
                }
            }
        }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;
        // optional: the softmax normalisation of every query, P[q, k] = exp(S[q, k] - stats[.., 0]) * stats[.., 1] - what the attention backward
        // of the multi-layer rollout (vitl.hip) needs to rebuild P without a statistics pass of its own
        if (stats && q < T && hi == 0) {
            float* st = stats + (((long)seq * H + h) * T + q) * 2;
            st[0] = mref; st[1] = inv;
        }
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] *= inv;
        // o[db][r] = O[query q][d = db*32 + (r&3) + 8*(r>>2) + 4*hi]
        if (q < T) {
            f16* orow = out + ((long)seq * T + q) * D + h * 64;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    f16x4 hv;
#pragma unroll
                    for (int j = 0; j < 4; ++j) hv[j] = (f16)o[db][rq * 4 + j];
                    *reinterpret_cast<f16x4*>(orow + db * 32 + rq * 8 + 4 * hi) = hv;
                }
        }
    }
}

// =================================================================================================
// k_attention2: the same computation with K AND V staged row-major by direct-to-LDS DMA and V read through the gfx950 transposing LDS read.
//   * staging: `buffer_load ... lds` (16 B per lane, 1 KB = 8 token rows per wave instruction): no staging VGPRs, no ds_write at all.  The
//     round-2 kernel bounced every row through registers and scattered V^T with 2-byte ds_write_b16 (4-way bank conflicts, 29 % of the
//     LDS-active cycles).  Rows >= T fall off the buffer descriptor (extent = this sequence's T rows) and land as zeros.
//   * LDS images: K [TP][64] with the 16-byte chunk XOR-swizzle of kswz (conflict-free ds_read_b128 fragments, as before); V [TP][64] with
//     chunk ^= ((row >> 1) & 1) << 2, applied on the DMA SOURCE address (the DMA's LDS image is lane-linear).
//   * P.V:  O^T += V^T P^T needs, per lane (d = lane & 31, k half = lane >> 5), V[key0 .. key0 + 3][d] - a COLUMN of the row-major image.
//     `ds_read_b64_tr_b16`: the 16 lanes of a group each address 8 bytes of a [4 keys][16 d] block (lane i: key i / 4, d chunk i % 4) and
//     get back its column i.  With the swizzle above the four key rows of a group pair occupy four disjoint 16-bank windows: conflict-free.
// Selected per call by bits 1 / 2 of semabs_attention's `causal` word (A/B in tests and tools); measured slower than k_attention, see the launcher.
// =================================================================================================
__device__ __forceinline__ int vswz_chunk(int row, int chunk) { return chunk ^ (((row >> 1) & 1) << 2); }
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int NKB, bool CAUSAL, bool DMA>          // DMA: stage K / V by direct-to-LDS loads; else through registers with 16-byte LDS stores (same images)
__global__ __launch_bounds__(64 * ((NKB > 4) ? 8 : 4), 2) void k_attention2(const f16* __restrict__ qkv, f16* __restrict__ out,
                                                    float* __restrict__ stats, int T, int H, int ld, int D) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TP = 32 * NKB;
    char* sK = smem;
    char* sV = smem + TP * 128;
    constexpr int NWAVE = (NKB > 4) ? 8 : 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int seq = blockIdx.x / H, h = blockIdx.x % H;
    const f16* base = qkv + (long)seq * T * ld + h * 64;
    const int ql = lane & 31, hi = lane >> 5;

    // query fragments of the wave's first block: ordinary loads, issued BEFORE the LDS-DMA loads (an ordinary load queued behind outstanding
    // DMA loads can be reported complete early on gfx950 - DESIGN.md 6a - so the two kinds never share the queue in that order)
    f16x8 fq0[4];
    {
        const int q = wid * 32 + ql;
        const int qc = q < T ? q : T - 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fq0[ks] = *reinterpret_cast<const f16x8*>(base + (long)qc * ld + (ks * 2 + hi) * 8);
    }
    if constexpr (!DMA) {
        // register staging: every K / V row chunk of the thread is requested before the first LDS store (as in the round-2 kernel), then
        // written with ONE ds_write_b128 each - V row-major in the swizzled image instead of eight 2-byte V^T scatters
        constexpr int NTHR = 64 * NWAVE;
        constexpr int NIT_ST = (TP * 8 + NTHR - 1) / NTHR;
        f16x8 kvs[NIT_ST], vvs[NIT_ST];
#pragma unroll
        for (int it = 0; it < NIT_ST; ++it) {
            const int c = tid + it * NTHR;
            const int row = c >> 3, kc = c & 7;
            if (c < TP * 8 && row < T) {
                kvs[it] = *reinterpret_cast<const f16x8*>(base + (long)row * ld + D + kc * 8);
                vvs[it] = *reinterpret_cast<const f16x8*>(base + (long)row * ld + 2 * D + kc * 8);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { kvs[it][e] = (f16)0.f; vvs[it][e] = (f16)0.f; }
            }
        }
#pragma unroll
        for (int it = 0; it < NIT_ST; ++it) {
            const int c = tid + it * NTHR;
            const int row = c >> 3, kc = c & 7;
            if (c < TP * 8) {
                *reinterpret_cast<f16x8*>(sK + kswz(row, kc)) = kvs[it];
                *reinterpret_cast<f16x8*>(sV + row * 128 + (vswz_chunk(row, kc) << 4)) = vvs[it];
            }
        }
    } else {
        // one resource descriptor per workgroup: this sequence's T token rows; rows >= T read as zeros
        const long bytes = ((long)(T - 1) * ld + 3L * D) * 2 - (long)h * 128;
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(base), 0, (int)(bytes > 0x7fffffffL ? 0x7fffffffL : bytes), 0x00020000);
        constexpr int PIECES = TP / 8;                       // 1 KB pieces (8 rows) per operand
        const int prow = lane >> 3, pc = lane & 7;
#pragma unroll
        for (int it = 0; it < (2 * PIECES + NWAVE - 1) / NWAVE; ++it) {
            const int p = it * NWAVE + wid;                  // wave-uniform piece id: K pieces first, then V
            if (p < 2 * PIECES) {
                const bool isv = p >= PIECES;
                const int row = (isv ? p - PIECES : p) * 8 + prow;
                const int lc = isv ? vswz_chunk(row, pc) : (pc ^ ((row >> 1) & 7));          // logical chunk that lives at physical chunk pc
                const unsigned voff = (unsigned)row * (unsigned)ld * 2u + (unsigned)((isv ? 2 * D : D) + lc * 8) * 2u;
                char* dst = (isv ? sV : sK) + (isv ? p - PIECES : p) * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)dst, 16, voff, 0u, 0, 0);
            }
        }
    }
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int koff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = kswz(ql, ks * 2 + hi);
    // transposing V reads: lane -> (key row i / 4 + 4 hi of the group's 4-key block, 8-byte piece i % 4 of the 16-d half `dsel`); the swizzle
    // bit of the row is (i >> 3) & 1 for every block (key0 is a multiple of 4 with bit 1 clear), so the lane's offset inside a row is constant
    const int gi = lane & 15, dsel = (lane >> 4) & 1;
    const int vrow_off = ((gi >> 2) + 4 * hi) * 128;
    const int vcol0 = (((dsel * 2 + ((gi & 3) >> 1)) ^ (((gi >> 3) & 1) << 2)) << 4) + (gi & 1) * 8;         // d block 0; d block 1 = ^ 64
    auto vfrag = [&](int key0, int db) -> f16x8 {
        const char* pa = sV + key0 * 128 + vrow_off + (vcol0 ^ (db << 6));
        const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)pa);
        const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pa + 8 * 128));
        f16x8 v;
        const f16x4 fa = __builtin_bit_cast(f16x4, a), fb = __builtin_bit_cast(f16x4, b);
        v[0] = fa[0]; v[1] = fa[1]; v[2] = fa[2]; v[3] = fa[3]; v[4] = fb[0]; v[5] = fb[1]; v[6] = fb[2]; v[7] = fb[3];
        return v;
    };
#pragma unroll 1
    for (int qb = wid; qb < NKB; qb += NWAVE) {
        const int q = qb * 32 + ql;
        const int qc = q < T ? q : T - 1;
        f16x8 fq[4];
        if (qb == wid) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fq[ks] = fq0[ks];
        } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fq[ks] = *reinterpret_cast<const f16x8*>(base + (long)qc * ld + (ks * 2 + hi) * 8);
        }
        float mref = 0.f, sum = 0.f;
        f32x16 o[2];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
#pragma unroll 1
        for (int kb = 0; kb < NKB; ++kb) {
            f32x16 sc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[r] = -mref;       // the MFMA accumulator does the subtraction of the reference maximum
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const f16x8 fk = *reinterpret_cast<const f16x8*>(sK + kb * 4096 + koff[ks]);
                sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fk, fq[ks], sc, 0, 0, 0);
            }
            // the V fragments of this key block do not depend on the scores: request them now, they arrive under the softmax arithmetic
            f16x8 fv[2][2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int db = 0; db < 2; ++db) fv[hf][db] = vfrag(kb * 32 + hf * 16, db);
            if (CAUSAL || kb * 32 + 31 >= T) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const bool dead = key >= T || (CAUSAL && key > q);
                    sc[r] = dead ? -INFINITY : sc[r];
                }
            }
            float bm = sc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) bm = fmaxf(bm, sc[r]);
            bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
            const bool raise = kb == 0 || bm > 8.f;
            if (__ballot(raise) != 0) {
                const float d = raise ? bm : 0.f;
                const float alpha = __builtin_amdgcn_exp2f(-d * 1.44269504088896340736f);
                mref += d;
                sum *= alpha;
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[r] -= d;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) { sc[r] = __builtin_amdgcn_exp2f(sc[r] * 1.44269504088896340736f); sum += sc[r]; }
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                f16x8 p;
#pragma unroll
                for (int j = 0; j < 8; ++j) p[j] = (f16)sc[hf * 8 + j];
#pragma unroll
                for (int db = 0; db < 2; ++db) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fv[hf][db], p, o[db], 0, 0, 0);
            }
        }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;
        if (stats && q < T && hi == 0) {
            float* st = stats + (((long)seq * H + h) * T + q) * 2;
            st[0] = mref; st[1] = inv;
        }
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] *= inv;
        if (q < T) {
            f16* orow = out + ((long)seq * T + q) * D + h * 64;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    f16x4 hv;
#pragma unroll
                    for (int j = 0; j < 4; ++j) hv[j] = (f16)o[db][rq * 4 + j];
                    *reinterpret_cast<f16x4*>(orow + db * 32 + rq * 8 + 4 * hi) = hv;
                }
        }
    }
}

// qkv fp16 [n_seq, T, ld] with q | k | v at column offsets 0, D, 2D (q already scaled); out fp16 [n_seq, T, D]
// row_stats (optional, may be NULL) fp32 [n_seq, H, T, 2] = (reference maximum, 1 / sum) of every query's softmax
// causal: bit 0 = causal mask (text tower); bit 1 = k_attention2 with register staging, bit 2 = k_attention2 with LDS-DMA staging (A/B; all three give identical results)
static int attention_impl(const void* qkv, void* out, void* row_stats, int n_seq, int T, int H, int head_dim, int ld, int causal,
                          const void* qk_lo, int ld_lo, void* stream);
extern "C" int semabs_attention(const void* qkv, void* out, void* row_stats, int n_seq, int T, int H, int head_dim,
                                int ld, int causal, void* stream) {
    return attention_impl(qkv, out, row_stats, n_seq, T, H, head_dim, ld, causal, nullptr, 0, stream);
}
// The same with q and k as fp16 hi + lo pairs: qk_lo fp16 [n_seq, T, ld_lo], q_lo | k_lo at column offsets 0 / D (precision = "parity": three score
// products in fp32; k_attention<.., SPLIT>).  causal bits 1-2 (the k_attention2 A/B variants) are not available here.
extern "C" int semabs_attention_split(const void* qkv, const void* qk_lo, void* out, void* row_stats, int n_seq, int T, int H, int head_dim,
                                      int ld, int ld_lo, int causal, void* stream) {
    SEMABS_REQUIRE(qk_lo && ld_lo % 8 == 0 && ld_lo >= 2 * H * 64 && (causal & 6) == 0, "semabs_attention_split: needs qk_lo with a 16-byte aligned row pitch >= 2 D and the default kernel");
    return attention_impl(qkv, out, row_stats, n_seq, T, H, head_dim, ld, causal, qk_lo, ld_lo, stream);
}
static int attention_impl(const void* qkv, void* out, void* row_stats, int n_seq, int T, int H, int head_dim, int ld, int causal,
                          const void* qk_lo, int ld_lo, void* stream) {
    if (n_seq == 0) return SEMABS_OK;
    SEMABS_REQUIRE(qkv && out && n_seq > 0 && T > 0 && H > 0, "semabs_attention: bad args");
    SEMABS_REQUIRE(head_dim == 64, "semabs_attention: head_dim must be 64");
    SEMABS_REQUIRE(T <= 288 && ld % 8 == 0, "semabs_attention: T must be <= 288");
    const int D = H * 64;
    const int nkb = (T + 31) / 32;
    const bool is_causal = (causal & 1) != 0;
    // Default = k_attention (register staging, V^T in LDS).  k_attention2 (row-major V + transposing LDS reads; bit 1: register staging, bit 2:
    // LDS-DMA staging) removes every LDS bank conflict and 61 % of the LDS-active cycles (profiles/r03_pmc_attention_ab.json) but runs 12 %
    // SLOWER at the bench shape (797 / 803 vs 707 us per 2 448-tile launch): kept selectable per call for A/B, identical results.
    const bool legacy = (causal & 6) == 0;
    const int order = (causal >> 3) & 3;                    // bits 3-4: 0 dispatch order, 1 / 2 XCD-contiguous runs of (sequence, head) forwards / backwards (k_attention only)
    const bool dma = (causal & 4) != 0 && (long)T * ld * 2 < (1L << 31);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(n_seq * H);
#define ATT_LAUNCH(N, C)                                                                                             \
    {                                                                                                                \
        size_t lds = (size_t)(32 * N) * 128 + 64 * (32 * N + 4) * 2;                                                 \
        static SemabsLdsAttr attr;                                                                                   \
        semabs_ensure_lds(&k_attention<N, C>, (int)lds, attr);                                                       \
        hipLaunchKernelGGL((k_attention<N, C>), grid, dim3(64 * ((N > 4) ? 8 : 4)), lds, s, (const f16*)qkv, (f16*)out, (float*)row_stats, T, H, ld, D, order); \
    }
#define ATT2_LAUNCH(N, C, DM)                                                                                        \
    {                                                                                                                \
        size_t lds = (size_t)(32 * N) * 256;                                                                         \
        static SemabsLdsAttr attr2;                                                                                  \
        semabs_ensure_lds(&k_attention2<N, C, DM>, (int)lds, attr2);                                                 \
        hipLaunchKernelGGL((k_attention2<N, C, DM>), grid, dim3(64 * ((N > 4) ? 8 : 4)), lds, s, (const f16*)qkv, (f16*)out, (float*)row_stats, T, H, ld, D); \
    }
#define ATT_SPLIT_LAUNCH(N, C)                                                                                       \
    {                                                                                                                \
        size_t lds = (size_t)(32 * N) * 128 * 2 + 64 * (32 * N + 4) * 2;                                             \
        static SemabsLdsAttr attr_s;                                                                                 \
        semabs_ensure_lds(&k_attention<N, C, true>, (int)lds, attr_s);                                               \
        hipLaunchKernelGGL((k_attention<N, C, true>), grid, dim3(64 * ((N > 4) ? 8 : 4)), lds, s, (const f16*)qkv, (f16*)out, (float*)row_stats, T, H, ld, D, order, (const f16*)qk_lo, ld_lo); \
    }
#define ATT_CASE(N) { if (qk_lo) { if (is_causal) ATT_SPLIT_LAUNCH(N, true) else ATT_SPLIT_LAUNCH(N, false) }                      \
                      else if (legacy) { if (is_causal) ATT_LAUNCH(N, true) else ATT_LAUNCH(N, false) }                            \
                      else if (dma) { if (is_causal) ATT2_LAUNCH(N, true, true) else ATT2_LAUNCH(N, false, true) }                 \
                      else { if (is_causal) ATT2_LAUNCH(N, true, false) else ATT2_LAUNCH(N, false, false) } }
    switch (nkb) {
        case 1: case 2: ATT_CASE(2) break;
        case 3: ATT_CASE(3) break;
        case 4: ATT_CASE(4) break;
        case 5: ATT_CASE(5) break;
        case 6: ATT_CASE(6) break;
        case 7: ATT_CASE(7) break;
        case 8: ATT_CASE(8) break;
        default: ATT_CASE(9) break;                          // T = 257: ViT-L/14
    }
#undef ATT_CASE
#undef ATT_LAUNCH
#undef ATT_SPLIT_LAUNCH
#undef ATT2_LAUNCH
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// =================================================================================================
// Last block, CLS query only (only x[CLS] reaches the image feature): one wave per (tile, head).
//   q fp32 [n, D] (scaled), kv fp32 [n, T, 2D] (k | v)  ->  probs fp32 [n, H, T] (kept for the rollout),
//   o fp16 [n, D] (input of out_proj)
// =================================================================================================
__global__ __launch_bounds__(256) void k_attention_cls(const float* __restrict__ q, const float* __restrict__ kmat, const f16* __restrict__ vmat,
                                                       float* __restrict__ probs, f16* __restrict__ o, int n, int T, int H, int split) {
    __shared__ float sp[4][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long wh = (long)blockIdx.x * 4 + w;
    if (wh >= (long)n * H) return;
    const int D = H * 64;
    const int i = (int)(wh / H), h = (int)(wh % H);
    const float* qr = q + (long)i * D + h * 64;
    const float* kb = kmat + (long)i * T * D + h * 64;
    if (!kmat) {                                              // the softmax rows are given (semabs_cls_scores): P . V only
#pragma unroll
        for (int t = 0; t < 4; ++t) { const int j = t * 64 + lane; sp[w][j] = j < T ? probs[wh * T + j] : 0.f; }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        const f16* vb0 = vmat + (long)i * T * D + h * 64 + lane;
        float a0 = 0.f;
        for (int j = 0; j < T; ++j) a0 += sp[w][j] * (float)vb0[(long)j * D];
        if (split) {
            const f16 hi = (f16)a0;
            o[(long)i * 2 * D + h * 64 + lane] = hi;
            o[(long)i * 2 * D + D + h * 64 + lane] = (f16)(a0 - (float)hi);
        } else {
            o[(long)i * D + h * 64 + lane] = (f16)a0;
        }
        return;
    }
    float sc[4];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        int j = t * 64 + lane;
        float acc = -INFINITY;
        if (j < T) {
            const float4* kr = reinterpret_cast<const float4*>(kb + (long)j * D);
            const float4* q4 = reinterpret_cast<const float4*>(qr);
            acc = 0.f;
#pragma unroll
            for (int d = 0; d < 16; ++d) {
                float4 a = q4[d], b = kr[d];
                acc += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
            }
        }
        sc[t] = acc;
        mx = fmaxf(mx, acc);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) { sc[t] = (t * 64 + lane < T) ? __expf(sc[t] - mx) : 0.f; sum += sc[t]; }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        int j = t * 64 + lane;
        float p = sc[t] * inv;
        sp[w][j] = p;
        if (j < T) probs[wh * T + j] = p;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    const f16* vb = vmat + (long)i * T * D + h * 64 + lane;       // lane = d
    float acc = 0.f;
    for (int j = 0; j < T; ++j) acc += sp[w][j] * (float)vb[(long)j * D];
    if (split) {                                                 // [hi | lo] rows of pitch 2 D (store_split4's layout)
        const f16 hi = (f16)acc;
        o[(long)i * 2 * D + h * 64 + lane] = hi;
        o[(long)i * 2 * D + D + h * 64 + lane] = (f16)(acc - (float)hi);
    } else {
        o[(long)i * D + h * 64 + lane] = (f16)acc;
    }
}

extern "C" int semabs_attention_cls(const float* q, const float* k, const void* v, float* probs, void* o, int n, int T, int H,
                                    int head_dim, int split, void* stream) {
    if (n == 0) return SEMABS_OK;
    SEMABS_REQUIRE((q || !k) && v && probs && o && n > 0, "semabs_attention_cls: bad args (k = NULL: probs are an input, semabs_cls_scores)");
    SEMABS_REQUIRE(head_dim == 64 && T <= 256 && T > 0, "semabs_attention_cls: head_dim 64, T <= 256");
    hipLaunchKernelGGL(k_attention_cls, dim3(semabs_cdiv((long)n * H, 4)), dim3(256), 0, (hipStream_t)stream, q, k, (const f16*)v, probs,
                       (f16*)o, n, T, H, split);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// =================================================================================================
// Last block, CLS query: the kept softmax row WITHOUT the K projection (round 6).  Only the CLS query's scores are ever needed, and
//   s[h, j] = q_h . (W_k,h x_j + b_k,h) = (W_k,h^T q_h) . x_j + q_h . b_k,h,
// so instead of K = X W_k^T for every token (an [M, D] x [D, D] GEMM with fp32 output: 1.1 ms per scene and 1.48 GB written and read back) a workgroup per tile
// forms R[h, :] = W_k,h^T q_h (12 x D values, fp32 FMAs over the fp16-exact weights), splits it into fp16 hi / lo planes in LDS - the A operand, rows = heads
// padded to 16 - and multiplies it with the tile's LayerNorm-1 rows x_j ([hi | lo] rows of pitch 2 D, or plain fp16 rows) on v_mfma_f32_16x16x32_f16: three
// products, so both factors enter to ~2^-22.  Softmax per head in fp32 -> probs [n, H, T], the input of semabs_attention_cls(k = NULL).
//   q fp32 [n, D] (scaled CLS query incl. bias), wk fp16 [D, D] ([out, in] rows of in_proj_weight[D:2D]), bk fp32 [D], x fp16 rows of pitch ldx.
// CLIP/clip/auxiliary.py:307-337 for the one query row that clip_gradcam.py:124-131 reads.
// =================================================================================================
template <int D>
__global__ __launch_bounds__(384) void k_cls_scores(const float* __restrict__ q, const f16* __restrict__ wk, const float* __restrict__ bk, const f16* __restrict__ x,
                                                    long ldx, int split, float* __restrict__ probs, int T) {
    constexpr int H = D / 64, PITCH = D + 8, CG = D / 8;    // CG threads cover one weight row with 16-byte loads
    static_assert(H <= 16 && CG * 4 == 384, "built for D = 768 (12 heads, 96 column groups x 4 head slots)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f16* r_hi = reinterpret_cast<f16*>(smem);               // [16][PITCH]
    f16* r_lo = r_hi + 16 * PITCH;
    float* s_q = reinterpret_cast<float*>(r_lo + 16 * PITCH);   // [D]
    float* s_sb = s_q + D;                                  // [16]: q_h . b_k,h
    float* s_s = s_sb + 16;                                 // [16][224] scores
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const long tile = blockIdx.x;
    for (int c = tid; c < D; c += 384) s_q[c] = q[tile * D + c];
    for (int c = tid; c < 4 * PITCH; c += 384) { r_hi[12 * PITCH + c] = (f16)0.f; r_lo[12 * PITCH + c] = (f16)0.f; }     // rows 12 .. 15 of the A operand
    __syncthreads();
    if (tid < H) {
        float a = 0.f;
        for (int d = 0; d < 64; ++d) a += s_q[tid * 64 + d] * bk[tid * 64 + d];
        s_sb[tid] = a;
    }
    // R: thread (cg, hs) owns 8 columns of heads hs, hs + 4, hs + 8; a weight row is read by 96 consecutive threads (1 536 contiguous bytes)
    const int cg = tid % CG, hs = tid / CG;
#pragma unroll 1
    for (int hh = 0; hh < H / 4; ++hh) {
        const int h = hh * 4 + hs;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const f16* wrow = wk + (long)(h * 64) * D + cg * 8;
#pragma unroll 8
        for (int d = 0; d < 64; ++d) {
            const f16x8 w = *reinterpret_cast<const f16x8*>(wrow + (long)d * D);
            const float qv = s_q[h * 64 + d];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += qv * (float)w[e];
        }
        f16x8 vh, vl;
#pragma unroll
        for (int e = 0; e < 8; ++e) { vh[e] = (f16)acc[e]; vl[e] = (f16)(acc[e] - (float)vh[e]); }
        *reinterpret_cast<f16x8*>(r_hi + h * PITCH + cg * 8) = vh;
        *reinterpret_cast<f16x8*>(r_lo + h * PITCH + cg * 8) = vl;
    }
    __syncthreads();
    // scores: a wave takes every sixth block of 16 tokens; lane = (token of the block, 8-wide k chunk)
    const int l15 = lane & 15, kg = lane >> 4;
    const int nblk = (T + 15) / 16;
    for (int tb = wid; tb < nblk; tb += 6) {
        int j = tb * 16 + l15;
        if (j >= T) j = T - 1;                              // clamped rows are computed and dropped
        const f16* xr = x + (tile * T + j) * ldx + kg * 8;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 6
        for (int ks = 0; ks < D / 32; ++ks) {
            const f16x8 bh = *reinterpret_cast<const f16x8*>(xr + ks * 32);
            const f16x8 ah = *reinterpret_cast<const f16x8*>(r_hi + l15 * PITCH + ks * 32 + kg * 8);
            const f16x8 al = *reinterpret_cast<const f16x8*>(r_lo + l15 * PITCH + ks * 32 + kg * 8);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
            if (split) {
                const f16x8 bl = *reinterpret_cast<const f16x8*>(xr + D + ks * 32);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
            }
        }
        // acc[r] = s[head 4 kg + r][token tb * 16 + l15]
        if (tb * 16 + l15 < T) {
#pragma unroll
            for (int r = 0; r < 4; ++r) s_s[(kg * 4 + r) * 224 + tb * 16 + l15] = acc[r];
        }
    }
    __syncthreads();
    for (int h = wid; h < H; h += 6) {
        float v[4];
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int j = t * 64 + lane;
            v[t] = j < T ? s_s[h * 224 + j] + s_sb[h] : -INFINITY;
            mx = fmaxf(mx, v[t]);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) { v[t] = (t * 64 + lane < T) ? __expf(v[t] - mx) : 0.f; sum += v[t]; }
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int j = t * 64 + lane;
            if (j < T) probs[(tile * H + h) * T + j] = v[t] * inv;
        }
    }
}
extern "C" int semabs_cls_scores(const float* q, const void* wk, const float* bk, const void* x, long ldx, int split, float* probs, int n, int T, int D, void* stream) {
    if (n == 0) return SEMABS_OK;
    SEMABS_REQUIRE(q && wk && bk && x && probs && n > 0, "semabs_cls_scores: bad args");
    SEMABS_REQUIRE(D == 768 && T > 0 && T <= 224 && ldx >= (split ? 2 : 1) * (long)D && ldx % 8 == 0, "semabs_cls_scores: D = 768, T <= 224, 16-byte aligned rows of pitch >= D (2 D when split)");
    constexpr int PITCH = 768 + 8;
    const size_t lds = (size_t)2 * 16 * PITCH * 2 + (768 + 16 + 16 * 224) * 4;
    static SemabsLdsAttr attr;
    semabs_ensure_lds(&k_cls_scores<768>, (int)lds, attr);
    hipLaunchKernelGGL(k_cls_scores<768>, dim3((unsigned)n), dim3(384), lds, (hipStream_t)stream, q, (const f16*)wk, bk, (const f16*)x, ldx, split, probs, T);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// =================================================================================================
// Small element-wise pieces
// =================================================================================================
// strided row gather: dst[r, :] = src[r * src_stride + offset ...][:cols]  (fp32), e.g. the CLS rows of x
__global__ void k_rows_gather(const float* __restrict__ src, float* __restrict__ dst, long rows, int cols, long stride, long off) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    long r = i / cols; int c = (int)(i % cols);
    dst[i] = src[r * stride + off + c];
}
extern "C" int semabs_rows_gather(const float* src, float* dst, long rows, int cols, long src_stride, long offset, void* stream) {
    if (rows == 0) return SEMABS_OK;
    SEMABS_REQUIRE(src && dst && rows > 0 && cols > 0, "semabs_rows_gather: bad args");
    hipLaunchKernelGGL(k_rows_gather, dim3(semabs_cdiv(rows * cols, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, rows, cols, src_stride, offset);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// EOT rows of the text tower: dst[b, :] = x[b * T + argmax_t tokens[b, t], :] (`x[torch.arange(B), text.argmax(dim=-1)]`, model_explainability.py:480;
// the end-of-text token is the largest id of a sequence; first occurrence on ties, like torch.argmax).  One wave per sequence.
__global__ __launch_bounds__(64) void k_eot_rows_gather(const long long* __restrict__ tokens, const float* __restrict__ x, float* __restrict__ dst, int T, int D) {
    const int b = blockIdx.x, lane = threadIdx.x;
    long long best = -0x7fffffffffffffffLL - 1; int at = 0;
    for (int t = lane; t < T; t += 64) { const long long v = tokens[(long)b * T + t]; if (v > best) { best = v; at = t; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const long long ov = __shfl_xor(best, o, 64); const int oa = __shfl_xor(at, o, 64);
        if (ov > best || (ov == best && oa < at)) { best = ov; at = oa; }
    }
    const float* src = x + ((long)b * T + at) * D;
    for (int c = lane; c < D; c += 64) dst[(long)b * D + c] = src[c];
}
extern "C" int semabs_eot_rows_gather(const long long* tokens, const float* x, float* dst, int B, int T, int D, void* stream) {
    if (B == 0) return SEMABS_OK;
    SEMABS_REQUIRE(tokens && x && dst && B > 0 && T > 0 && D > 0, "semabs_eot_rows_gather: bad args");
    hipLaunchKernelGGL(k_eot_rows_gather, dim3(B), dim3(64), 0, (hipStream_t)stream, tokens, x, dst, T, D);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// quick-GELU forward on fp32 pre-activations -> fp16 (CLS-row MLP of the last block keeps fc for the VJP)
// split_w > 0: rows of split_w values stored as [hi | lo] with pitch 2 split_w (store_split4's layout)
__global__ void k_quickgelu(const float* __restrict__ fc, f16* __restrict__ act, long n, int split_w) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = fc[i];
    const float a = v / (1.f + __expf(-1.702f * v));
    if (split_w > 0) {
        const long r = i / split_w; const int c = (int)(i - r * split_w);
        const f16 hi = (f16)a;
        act[r * 2 * split_w + c] = hi;
        act[r * 2 * split_w + split_w + c] = (f16)(a - (float)hi);
    } else {
        act[i] = (f16)a;
    }
}
extern "C" int semabs_quickgelu(const float* fc, void* act, long n, int split_w, void* stream) {
    if (n == 0) return SEMABS_OK;
    SEMABS_REQUIRE(fc && act && n > 0 && split_w >= 0 && (split_w == 0 || n % split_w == 0), "semabs_quickgelu: bad args");
    hipLaunchKernelGGL(k_quickgelu, dim3(semabs_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, fc, (f16*)act, n, split_w);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// quick-GELU derivative table: gd = s (1 + 1.702 x (1 - s)), s = sigmoid(1.702 x), fp32 -> fp32.  The multi-layer rollout differentiates the same
// pre-activations for every label: the table is built once per (block, tile chunk) and the W_pr^T GEMM's epilogue (semabs_gemm_f16 epi 5)
// multiplies by its rows - no transcendental work per (label, element).
__global__ __launch_bounds__(256) void k_quickgelu_grad(const float* __restrict__ fc, float* __restrict__ gd, long n4) {
    const long i0 = (long)blockIdx.x * 1024 + threadIdx.x;
    float4 x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const long i = i0 + u * 256 < n4 ? i0 + u * 256 : 0; x[u] = *reinterpret_cast<const float4*>(fc + i * 4); }
    auto d = [](float v) { const float sg = 1.f / (1.f + __expf(-1.702f * v)); return sg * (1.f + 1.702f * v * (1.f - sg)); };
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long i = i0 + u * 256;
        if (i < n4) *reinterpret_cast<float4*>(gd + i * 4) = make_float4(d(x[u].x), d(x[u].y), d(x[u].z), d(x[u].w));
    }
}
extern "C" int semabs_quickgelu_grad(const float* fc, float* gd, long n, void* stream) {
    if (n == 0) return SEMABS_OK;
    SEMABS_REQUIRE(fc && gd && n > 0 && n % 4 == 0, "semabs_quickgelu_grad: bad args (n % 4 == 0)");
    hipLaunchKernelGGL(k_quickgelu_grad, dim3(semabs_cdiv(n / 4, 1024)), dim3(256), 0, (hipStream_t)stream, fc, gd, n / 4);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// logits + d logit / d feat, one wave per tile (clip_gradcam.py:63-67 and the head of the analytic VJP):
//   fh = f / |f|;  logit[i, l] = 100 fh . w_l;  dfeat[l, i, :] = 100 (w_l - fh (fh . w_l)) / |f|
// Each gradient row is normalised to max-abs 1 before the fp16 GEMM chain (the VJP is linear, the rollout
// multiplies the factor back in: scale[l, i]).  w_text fp32 [L, E] (row per label).
template <int PER>   // E = 64 * PER
__global__ __launch_bounds__(256) void k_logit_grad(const float* __restrict__ feat, const float* __restrict__ wt, int n, int L,
                                                    float* __restrict__ logits, f16* __restrict__ dfeat,
                                                    float* __restrict__ scale, int split) {
    constexpr int E = 64 * PER;
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    float f[PER];
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) { f[k] = feat[(long)i * E + k * 64 + lane]; ss += f[k] * f[k]; }
    const float nrm = sqrtf(wave_sum(ss));
#pragma unroll
    for (int k = 0; k < PER; ++k) f[k] = f[k] / nrm;
    for (int l = 0; l < L; ++l) {
        float w[PER], g[PER];
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < PER; ++k) { w[k] = wt[(long)l * E + k * 64 + lane]; dot += f[k] * w[k]; }
        dot = wave_sum(dot);
        if (logits && lane == 0) logits[(long)i * L + l] = 100.f * dot;
        float mx = 0.f;
#pragma unroll
        for (int k = 0; k < PER; ++k) { g[k] = 100.f * (w[k] - f[k] * dot) / nrm; mx = fmaxf(mx, fabsf(g[k])); }
        mx = wave_max(mx);
        const float sc = mx > 0.f ? mx : 1.f;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const float gv = g[k] / sc;
            const f16 hi = (f16)gv;
            if (split) {                                         // [hi | lo] rows of pitch 2 E
                dfeat[((long)l * n + i) * 2 * E + k * 64 + lane] = hi;
                dfeat[((long)l * n + i) * 2 * E + E + k * 64 + lane] = (f16)(gv - (float)hi);
            } else {
                dfeat[((long)l * n + i) * E + k * 64 + lane] = hi;
            }
        }
        if (lane == 0) scale[(long)l * n + i] = sc;
    }
}
extern "C" int semabs_logit_grad(const float* feat, const float* w_text, int n, int L, int E, float* logits, void* dfeat,
                                 float* scale, int split, void* stream) {
    if (n == 0 || L == 0) return SEMABS_OK;
    SEMABS_REQUIRE(feat && w_text && dfeat && scale, "semabs_logit_grad: bad args");
    SEMABS_REQUIRE(E == 512 || E == 768, "semabs_logit_grad: embed dim must be 512 or 768");
    dim3 grid(semabs_cdiv(n, 4)), block(256);
    if (E == 512) hipLaunchKernelGGL(k_logit_grad<8>, grid, block, 0, (hipStream_t)stream, feat, w_text, n, L, logits, (f16*)dfeat, scale, split);
    else hipLaunchKernelGGL(k_logit_grad<12>, grid, block, 0, (hipStream_t)stream, feat, w_text, n, L, logits, (f16*)dfeat, scale, split);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// LayerNorm VJP wrt the input, rows (l, i): x row = i = m % n_x.  out = [resid +] rstd (g - mean(g) - xh mean(g xh)),
// g = gy * gamma.  Writes fp32 (out32, optional) and fp16 (out16, optional).
template <int VPL>
__global__ void k_ln_bwd(const float* __restrict__ x, long ld_x, const float* __restrict__ gamma, const float* __restrict__ gy,
                         const float* __restrict__ resid, float* __restrict__ out32, f16* __restrict__ out16, long M, int n_x,
                         float eps, int split) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int D = 256 * VPL;
    const float* xr = x + (row % n_x) * ld_x;
    float4 v[VPL], g[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        v[i] = *reinterpret_cast<const float4*>(xr + (i * 64 + lane) * 4);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
        q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    const float rstd = rsqrtf(wave_sum(q) / D + eps);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c0 = (i * 64 + lane) * 4;
        float4 gm = *reinterpret_cast<const float4*>(gamma + c0);
        float4 gg = *reinterpret_cast<const float4*>(gy + row * D + c0);
        g[i].x = gg.x * gm.x; g[i].y = gg.y * gm.y; g[i].z = gg.z * gm.z; g[i].w = gg.w * gm.w;
        v[i].x *= rstd; v[i].y *= rstd; v[i].z *= rstd; v[i].w *= rstd;   // xh
        sg += (g[i].x + g[i].y) + (g[i].z + g[i].w);
        sgx += (g[i].x * v[i].x + g[i].y * v[i].y) + (g[i].z * v[i].z + g[i].w * v[i].w);
    }
    const float mg = wave_sum(sg) / D, mgx = wave_sum(sgx) / D;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c0 = (i * 64 + lane) * 4;
        float4 o;
        o.x = rstd * (g[i].x - mg - v[i].x * mgx); o.y = rstd * (g[i].y - mg - v[i].y * mgx);
        o.z = rstd * (g[i].z - mg - v[i].z * mgx); o.w = rstd * (g[i].w - mg - v[i].w * mgx);
        if (resid) {
            float4 r = *reinterpret_cast<const float4*>(resid + row * D + c0);
            o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        if (out32) *reinterpret_cast<float4*>(out32 + row * D + c0) = o;
        if (out16) {
            if (split) {
                store_split4(out16 + row * 2 * D, D, c0, o.x, o.y, o.z, o.w);
            } else {
                f16x4 h; h[0] = (f16)o.x; h[1] = (f16)o.y; h[2] = (f16)o.z; h[3] = (f16)o.w;
                *reinterpret_cast<f16x4*>(out16 + row * D + c0) = h;
            }
        }
    }
}
extern "C" int semabs_ln_bwd(const float* x, const float* gamma, const float* gy, const float* resid, float* out32,
                             void* out16, long M, int D, int n_x, long ld_x, float eps, int split, void* stream) {
    if (M == 0) return SEMABS_OK;
    SEMABS_REQUIRE(x && gamma && gy && (out32 || out16) && M > 0 && n_x > 0, "semabs_ln_bwd: bad args");
    SEMABS_REQUIRE(D % 256 == 0 && D >= 256 && D <= 1024, "semabs_ln_bwd: D must be 256..1024 step 256");
    dim3 grid(semabs_cdiv(M, 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    switch (D / 256) {
        case 1: hipLaunchKernelGGL(k_ln_bwd<1>, grid, block, 0, s, x, ld_x, gamma, gy, resid, out32, (f16*)out16, M, n_x, eps, split); break;
        case 2: hipLaunchKernelGGL(k_ln_bwd<2>, grid, block, 0, s, x, ld_x, gamma, gy, resid, out32, (f16*)out16, M, n_x, eps, split); break;
        case 3: hipLaunchKernelGGL(k_ln_bwd<3>, grid, block, 0, s, x, ld_x, gamma, gy, resid, out32, (f16*)out16, M, n_x, eps, split); break;
        case 4: hipLaunchKernelGGL(k_ln_bwd<4>, grid, block, 0, s, x, ld_x, gamma, gy, resid, out32, (f16*)out16, M, n_x, eps, split); break;
    }
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// quick-GELU VJP: dfc[m, :] = dact[m, :] * (s (1 + 1.702 fc (1 - s))),  s = sigmoid(1.702 fc[m % n_x, :])   -> fp16
// One workgroup per row m: the pre-activation row m % n_x is a scalar decision, 4 float4 per operand in flight per thread (one float4 per
// thread with two 64-bit divisions per element ran at 3.7 TB/s: 1.79 ms per ViT-L block at 16 labels x 63 tiles).
__global__ __launch_bounds__(256) void k_gelu_bwd(const float* __restrict__ dact, const float* __restrict__ fc, f16* __restrict__ dfc, long M, int W, int n_x, int split) {
    const long m = blockIdx.x;
    const float* a_row = dact + m * W;
    const float* f_row = fc + (m % n_x) * W;
    f16* o_row = dfc + m * W * (split ? 2 : 1);
    auto d = [](float x) { float s = 1.f / (1.f + __expf(-1.702f * x)); return s * (1.f + 1.702f * x * (1.f - s)); };
    for (int c0 = threadIdx.x * 4; c0 < W; c0 += 4096) {
        float4 a[4], f[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * 1024 < W ? c0 + u * 1024 : c0;
            const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a_row + c));
            a[u] = make_float4(t[0], t[1], t[2], t[3]);
            f[u] = *reinterpret_cast<const float4*>(f_row + c);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * 1024;
            if (c < W) {
                const float r0 = a[u].x * d(f[u].x), r1 = a[u].y * d(f[u].y), r2 = a[u].z * d(f[u].z), r3 = a[u].w * d(f[u].w);
                if (split) {
                    store_split4(o_row, W, c, r0, r1, r2, r3);
                } else {
                    f16x4 h;
                    h[0] = (f16)r0; h[1] = (f16)r1; h[2] = (f16)r2; h[3] = (f16)r3;
                    *reinterpret_cast<f16x4*>(o_row + c) = h;
                }
            }
        }
    }
}
extern "C" int semabs_gelu_bwd(const float* dact, const float* fc, void* dfc, long M, int W, int n_x, int split, void* stream) {
    if (M == 0) return SEMABS_OK;
    SEMABS_REQUIRE(dact && fc && dfc && M > 0 && M < (1L << 31) && W % 4 == 0 && n_x > 0, "semabs_gelu_bwd: bad args");
    hipLaunchKernelGGL(k_gelu_bwd, dim3((unsigned)M), dim3(256), 0, (hipStream_t)stream, dact, fc, (f16*)dfc, M, W, n_x, split);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// =================================================================================================
// Rollout (closed form of clip_gradcam.py:90-131 for the only contributing block):
//   rel[l, i, j-1] = scale[l, i] / H * sum_h clampmin0?(A[i, h, j] * (V[i, j, h, :] . u[l, i, h, :])),  j = 1..T-1
//   probs fp32 [n, H, T]; v fp16 [n, T, D] (the last block's V); u fp32 [L, n, D]; out fp32 [L, n_total, T-1]
//   written at tile offset `tile0` (so chunks of tiles fill one [L, N, g, g] array).
// Round 2: the V . u dots are a [labels x 64] x [64 x tokens] product per (tile, head) and run on v_mfma_f32_16x16x32_f16: A = u rows of a
// group of 16 labels (fp32 split into fp16 hi + lo: two MFMAs, products exact to 2^-22), B = V rows of 16 tokens, so a lane owns one token
// and four labels, V is read ONCE per group of 16 labels (the thread-per-token VALU kernel it replaces re-read it per group of 4: 3.2 GB
// for a 0.74 GB tensor, 0.81 ms per scene) and the stores of a label are 64 contiguous bytes per 16 lanes.  One workgroup (4 waves) per
// (tile, label group); a wave takes every fourth block of 16 tokens and keeps its (at most five) accumulators across the heads, so that u
// is fetched once per head.
// =================================================================================================
__global__ __launch_bounds__(256) void k_rollout(const float* __restrict__ probs, const f16* __restrict__ vmat,
                                                 const float* __restrict__ u, const float* __restrict__ scale,
                                                 float* __restrict__ rel, int n, int T, int H, int L, int positive_only,
                                                 long n_total, long tile0) {
    constexpr int MB = 5;                                   // token blocks per wave: 4 waves x 5 x 16 = 320 >= T
    const int D = H * 64;
    const int i = blockIdx.x;
    const int l0 = blockIdx.y * 16;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int vl = lane & 15, kg = lane >> 4;
    const int la = min(l0 + vl, L - 1);                     // this lane's label row of the A operand
    const float* urow = u + ((long)la * n + i) * D + kg * 8;
    int jc[MB];
    f32x4 acc[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        jc[m] = min((wid + 4 * m) * 16 + vl, T - 1);
        acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int h = 0; h < H; ++h) {
        f16x8 ah[2], al[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const float4 x0 = *reinterpret_cast<const float4*>(urow + h * 64 + ks * 32);
            const float4 x1 = *reinterpret_cast<const float4*>(urow + h * 64 + ks * 32 + 4);
            const float xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) { ah[ks][e] = (f16)xs[e]; al[ks][e] = (f16)(xs[e] - (float)ah[ks][e]); }
        }
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            if ((wid + 4 * m) * 16 >= T) continue;          // wave-uniform
            const f16* vrow = vmat + ((long)i * T + jc[m]) * D + h * 64 + kg * 8;
            f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const f16x8 b = *reinterpret_cast<const f16x8*>(vrow + ks * 32);
                d = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks], b, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[ks], b, d, 0, 0, 0);
            }
            const float a = probs[((long)i * H + h) * T + jc[m]];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float c = a * d[e];
                acc[m][e] += positive_only ? fmaxf(c, 0.f) : c;
            }
        }
    }
    // acc[m][e] = sum over heads for (label l0 + 4 kg + e, token (wid + 4 m) * 16 + vl)
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        const int j = (wid + 4 * m) * 16 + vl;
        if (j < 1 || j >= T) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int l = l0 + 4 * kg + e;
            if (l < L) rel[((long)l * n_total + tile0 + i) * (T - 1) + (j - 1)] = acc[m][e] / H * scale[(long)l * n + i];
        }
    }
}
extern "C" int semabs_rollout(const float* probs, const void* v, const float* u, const float* scale, float* rel, int n,
                              int T, int H, int L, int positive_only, long n_total, long tile0, void* stream) {
    if (n == 0 || L == 0) return SEMABS_OK;
    SEMABS_REQUIRE(probs && v && u && scale && rel && n > 0 && T > 1 && H > 0, "semabs_rollout: bad args");
    SEMABS_REQUIRE(T <= 320, "semabs_rollout: T must be <= 320");
    hipLaunchKernelGGL(k_rollout, dim3(n, (L + 15) / 16), dim3(256), 0, (hipStream_t)stream, probs, (const f16*)v, u, scale, rel, n, T, H, L, positive_only, n_total, tile0);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// =================================================================================================
// Text tower glue
// =================================================================================================
// x[b, t, :] = token_embedding[tokens[b, t], :] + positional_embedding[t, :]
__global__ void k_gather_text(const long long* __restrict__ tokens, const float* __restrict__ emb, const float* __restrict__ pos,
                              float* __restrict__ x, int B, int T, int D) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * T * D) return;
    int d = (int)(i % D); long bt = i / D; int t = (int)(bt % T);
    x[i] = emb[tokens[bt] * D + d] + pos[(long)t * D + d];
}
extern "C" int semabs_gather_text(const long long* tokens, const float* emb, const float* pos, float* x, int B, int T, int D, void* stream) {
    if (B == 0) return SEMABS_OK;
    SEMABS_REQUIRE(tokens && emb && pos && x && B > 0 && T > 0 && D > 0, "semabs_gather_text: bad args");
    hipLaunchKernelGGL(k_gather_text, dim3(semabs_cdiv((long)B * T * D, 256)), dim3(256), 0, (hipStream_t)stream, tokens, emb, pos, x, B, T, D);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// e fp32 [C * P, E] (class-major) -> w fp32 [C, E] = mean_p e / |e|   (not re-normalised)
__global__ void k_text_finish(const float* __restrict__ e, float* __restrict__ w, int C, int P, int E) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    for (int k = lane; k < E; k += 64) w[(long)c * E + k] = 0.f;
    for (int p = 0; p < P; ++p) {
        const float* r = e + ((long)c * P + p) * E;
        float ss = 0.f;
        for (int k = lane; k < E; k += 64) ss += r[k] * r[k];
        const float nrm = sqrtf(wave_sum(ss));
        for (int k = lane; k < E; k += 64) w[(long)c * E + k] += r[k] / nrm;
    }
    for (int k = lane; k < E; k += 64) w[(long)c * E + k] /= (float)P;
}
extern "C" int semabs_text_finish(const float* e, float* w, int C, int P, int E, void* stream) {
    if (C == 0) return SEMABS_OK;
    SEMABS_REQUIRE(e && w && C > 0 && P > 0 && E > 0, "semabs_text_finish: bad args");
    hipLaunchKernelGGL(k_text_finish, dim3(semabs_cdiv(C, 4)), dim3(256), 0, (hipStream_t)stream, e, w, C, P, E);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}
