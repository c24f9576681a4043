// fp16 MFMA GEMM with fused epilogues:  C[M, N] = epi(A[M, K] . B[N, K]^T + bias[N])
//
// Used for every dense contraction of the CLIP ViT / text tower on the path (reference: F.linear calls at
// CLIP/clip/auxiliary.py:129,340 and model_explainability.py:210-217,253-254; patch conv :325; proj :353) and for
// the batched CLS-token VJP GEMMs of the attention x gradient rollout (clip_gradcam.py:90-97 autograd.grad).
// Both operands are K-contiguous (torch Linear weight layout [N, K]); weights are fp16-exact because the
// reference rounds them to fp16 (`convert_weights`, model_explainability.py:501-527).  fp32 accumulate.
//
// Structure (gfx950): 256 threads = 4 waves (2 x 2), block tile 128 x 128 x 64, wave tile 64 x 64 as 2 x 2
// v_mfma_f32_32x32x16_f16 accumulators; global -> VGPR -> LDS double buffer, one barrier per K tile; LDS rows are
// 128 B with the 16-B chunk index XOR-swizzled by (row >> 1) & 7 so ds_read_b128 fragment reads and
// ds_write_b128 staging writes are bank-conflict-free; epilogue stages each wave's 64 x 64 fp32 tile through
// its own 16 KB LDS slice to emit 16-byte row-contiguous stores; block ids are remapped so the blocks that share
// an A row-panel run on one XCD (private L2).
#include "semabs_common.h"

#define BM 128
#define BN 128
#define BK 64
#define GEMM_THREADS 256
#define TILE_BYTES (BM * BK * 2)          // 16 KB per operand tile
#define GEMM_LDS (4 * TILE_BYTES)         // A0 B0 A1 B1

enum { EPI_BIAS_F16 = 0, EPI_BIAS_GELU_F16 = 1, EPI_BIAS_RESID_F32 = 2, EPI_BIAS_F32 = 3, EPI_ROWMAP_ADD_F32 = 4 };

struct GemmArgs {
    const f16* A; const f16* B; void* C; const float* bias; const float* addend;
    long M; int N, K; long lda; int ldb; long ldc;
    int g_in, g_out, g_off;     // EPI_ROWMAP_ADD_F32: out row = (m / g_in) * g_out + g_off + m % g_in
    int n_tiles_n; int n_blocks;
};

__device__ __forceinline__ int swz_off(int row, int chunk) { return row * (BK * 2) + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int EPI>
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_gemm_f16(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;

    // XCD-aware bijective remap: hardware places block b on XCD b % 8; give each XCD a contiguous run of tiles.
    int b = blockIdx.x;
    {
        const int nb = g.n_blocks, q = nb >> 3, r = nb & 7, xcd = b & 7, k = b >> 3;
        b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const long m0 = (long)(b / g.n_tiles_n) * BM;
    const int n0 = (b % g.n_tiles_n) * BN;

    // staging assignment: 1024 16-byte chunks per operand tile, 4 per thread
    const f16* a_src[4]; const f16* b_src[4]; int lds_dst[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int c = tid + i * GEMM_THREADS, row = c >> 3, kc = c & 7;
        long am = m0 + row; if (am > g.M - 1) am = g.M - 1;
        a_src[i] = g.A + am * g.lda + kc * 8;
        b_src[i] = g.B + (long)(n0 + row) * g.ldb + kc * 8;
        lds_dst[i] = swz_off(row, kc);
    }
    f16x8 ra[4], rb[4];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = *reinterpret_cast<const f16x8*>(a_src[i] + (long)kt * BK);
            rb[i] = *reinterpret_cast<const f16x8*>(b_src[i] + (long)kt * BK);
        }
    };
    auto lstore = [&](int buf) {
        char* sa = smem + buf * 2 * TILE_BYTES; char* sb = sa + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f16x8*>(sa + lds_dst[i]) = ra[i];
            *reinterpret_cast<f16x8*>(sb + lds_dst[i]) = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = g.K / BK;
    gload(0);
    lstore(0);
    __syncthreads();
    const int frow = lane & 31, fk = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        const char* sa = smem + buf * 2 * TILE_BYTES; const char* sb = sa + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            f16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = *reinterpret_cast<const f16x8*>(sa + swz_off(wm * 64 + i * 32 + frow, ks * 2 + fk));
                fb[i] = *reinterpret_cast<const f16x8*>(sb + swz_off(wn * 64 + i * 32 + frow, ks * 2 + fk));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: wave-private 64 x 64 fp32 staging (16 KB), then row-contiguous 16-byte accesses ----
    float* st = reinterpret_cast<float*>(smem + wid * 16384);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                st[row * 64 + j * 32 + (lane & 31)] = acc[i][j][r];
            }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): own writes visible to own wave
    __builtin_amdgcn_wave_barrier();
    const int cchunk = lane & 15;          // 4 floats
    const int ncol = n0 + wn * 64 + cchunk * 4;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.bias) bv = *reinterpret_cast<const float4*>(g.bias + ncol);
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
        int row = it * 4 + (lane >> 4);
        long m = m0 + wm * 64 + row;
        if (m >= g.M) continue;
        float4 v = *reinterpret_cast<const float4*>(st + row * 64 + cchunk * 4);
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        if (EPI == EPI_BIAS_F16 || EPI == EPI_BIAS_GELU_F16) {
            if (EPI == EPI_BIAS_GELU_F16) {
                v.x = v.x / (1.f + __expf(-1.702f * v.x)); v.y = v.y / (1.f + __expf(-1.702f * v.y));
                v.z = v.z / (1.f + __expf(-1.702f * v.z)); v.w = v.w / (1.f + __expf(-1.702f * v.w));
            }
            f16x4 h; h[0] = (f16)v.x; h[1] = (f16)v.y; h[2] = (f16)v.z; h[3] = (f16)v.w;
            *reinterpret_cast<f16x4*>(reinterpret_cast<f16*>(g.C) + m * g.ldc + ncol) = h;
        } else if (EPI == EPI_BIAS_RESID_F32) {
            float4* p = reinterpret_cast<float4*>(reinterpret_cast<float*>(g.C) + m * g.ldc + ncol);
            float4 o = *p;
            o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
            *p = o;
        } else if (EPI == EPI_BIAS_F32) {
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(g.C) + m * g.ldc + ncol) = v;
        } else {
            long grp = m / g.g_in; int within = (int)(m - grp * g.g_in);
            long orow = grp * g.g_out + g.g_off + within;
            if (g.addend) {
                float4 a = *reinterpret_cast<const float4*>(g.addend + (long)(g.g_off + within) * g.N + ncol);
                v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
            }
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(g.C) + orow * g.ldc + ncol) = v;
        }
    }
}

template <int EPI>
static int launch(const GemmArgs& g, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_f16<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
        attr_set = true;
    }
    hipLaunchKernelGGL(k_gemm_f16<EPI>, dim3(g.n_blocks), dim3(GEMM_THREADS), GEMM_LDS, s, g);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// C ABI.  A fp16 [M, K] (row stride lda elements), B fp16 [N, K] (row stride ldb), C per `epi`:
//   0 fp16 = acc + bias        1 fp16 = quickgelu(acc + bias)      2 fp32 += acc + bias (in place)
//   3 fp32 = acc + bias        4 fp32 row-remapped store + addend  (rowmap = {g_in, g_out, g_off})
// bias fp32 [N] or NULL.  Requires N % 128 == 0, K % 64 == 0, 16-byte aligned rows.
extern "C" int semabs_gemm_f16(const void* A, const void* B, void* C, const float* bias, const float* addend,
                               long M, int N, int K, long lda, int ldb, long ldc, int epi, const int* rowmap3,
                               void* stream) {
    SEMABS_REQUIRE(A && B && C, "semabs_gemm_f16: null operand");
    SEMABS_REQUIRE(M > 0 && N > 0 && K > 0, "semabs_gemm_f16: empty problem");
    SEMABS_REQUIRE(N % BN == 0 && K % BK == 0, "semabs_gemm_f16: N must be a multiple of 128 and K of 64");
    SEMABS_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0, "semabs_gemm_f16: leading dimensions must keep 16-byte alignment");
    GemmArgs g;
    g.A = (const f16*)A; g.B = (const f16*)B; g.C = C; g.bias = bias; g.addend = addend;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.g_in = 1; g.g_out = 1; g.g_off = 0;
    if (epi == EPI_ROWMAP_ADD_F32) {
        SEMABS_REQUIRE(rowmap3 && rowmap3[0] > 0, "semabs_gemm_f16: epi 4 needs rowmap");
        g.g_in = rowmap3[0]; g.g_out = rowmap3[1]; g.g_off = rowmap3[2];
    }
    g.n_tiles_n = N / BN;
    long mt = (M + BM - 1) / BM;
    SEMABS_REQUIRE(mt * g.n_tiles_n < (1L << 30), "semabs_gemm_f16: grid too large");
    g.n_blocks = (int)(mt * g.n_tiles_n);
    hipStream_t s = (hipStream_t)stream;
    switch (epi) {
        case EPI_BIAS_F16: return launch<EPI_BIAS_F16>(g, s);
        case EPI_BIAS_GELU_F16: return launch<EPI_BIAS_GELU_F16>(g, s);
        case EPI_BIAS_RESID_F32: return launch<EPI_BIAS_RESID_F32>(g, s);
        case EPI_BIAS_F32: return launch<EPI_BIAS_F32>(g, s);
        case EPI_ROWMAP_ADD_F32: return launch<EPI_ROWMAP_ADD_F32>(g, s);
    }
    semabs_set_error("semabs_gemm_f16: unknown epilogue");
    return SEMABS_EINVAL;
}
