// fp16 MFMA GEMM with fused epilogues:  C[M, N] = epi(A[M, K] . B[N, K]^T + bias[N])
//
// Used for every dense contraction of the CLIP ViT / text tower on the path (reference: F.linear calls at
// CLIP/clip/auxiliary.py:129,340 and model_explainability.py:210-217,253-254; patch conv :325; proj :353) and for
// the batched CLS-token VJP GEMMs of the attention x gradient rollout (clip_gradcam.py:90-97 autograd.grad).
// Both operands are K-contiguous (torch Linear weight layout [N, K]); weights are fp16-exact because the
// reference rounds them to fp16 (`convert_weights`, model_explainability.py:501-527).  fp32 accumulate.
//
// Two kernels (gfx950):
//  * k_gemm8 (M >= 2048, N % 256 == 0, K >= 128 - every GEMM of the ViT trunk): 256 x 256 x 64 tile, 8 waves, four phases per K tile with one
//    half-tile of direct-to-LDS DMA issued per phase and four half-tiles in flight across barriers, skewed wave rows, fragment reads
//    prefetched inside the MFMA blocks, buffer-descriptor addressing, and an epilogue that transposes each wave's accumulators through its
//    own slice of LDS so that stores cover whole row segments.  Described in full at its definition below.
//  * k_gemm_f16 (everything else - small M, N % 256 != 0, K = 64): 128 x 128 x 64 (or 128 x 256 x 32) tile of 32x32x16 MFMAs, operands
//    global -> LDS by direct DMA (global_load_lds_dwordx4) in a 2-3 stage ring with counted vmcnt + raw barriers, LDS rows XOR-swizzled
//    (applied on the DMA source address and on the read) so ds_read_b128 fragment reads are bank-conflict-free, fragment ping-pong, and an
//    epilogue that stages 32-row slabs of each wave's fp32 tile through a private 8 KB LDS slice to emit 16-byte row-contiguous stores.
// Both: fused epilogues (bias / QuickGELU / fp32 residual read-modify-write / row-remap + addend), block ids remapped so the blocks that
// share operand panels run on one XCD (private L2) with grouped rasterisation.
#include "semabs_common.h"
#include <hip/hip_ext.h>
#include <type_traits>
#include <mutex>

#define BK 64   // K granularity required by the ABI (both K-tile sizes divide it)
enum { EPI_BIAS_F16 = 0, EPI_BIAS_GELU_F16 = 1, EPI_BIAS_RESID_F32 = 2, EPI_BIAS_F32 = 3, EPI_ROWMAP_ADD_F32 = 4, EPI_MULROW_F16 = 5 };

struct GemmArgs {
    const f16* A; const f16* B; void* C; const float* bias; const float* addend;
    long M; int N, K; long lda; int ldb; long ldc;
    int g_in, g_out, g_off;     // EPI_ROWMAP_ADD_F32: out row = (m / g_in) * g_out + g_off + m % g_in
    int n_tiles_n; int n_tiles_m; int group_m; int n_blocks;
    int sc_w;                   // column panels per super-column of the raster (see tile_of in k_gemm8); 0 = all of them
    int reverse;                // k_gemm8: every XCD walks its run of tiles backwards (zigzag with the producer of A, semabs_common.h)
    // LayerNorm folded into the GEMMs on either side of it (semabs_gemm_f16_ln; k_gemm8's LNP / k_gemm8p's LNC template parameters):
    f16* ln_xg; const float* ln_gamma; float* ln_part;      // producer (fp32 residual epilogue): fp16 ((x_new - centre) * gamma) [M, N], gamma [N], row partials [M, N / 256, 2]
    const float* ln_center;                                 // producer: per-row centre [M] subtracted before the fp16 copy and the partial sums (NULL = 0)
    const float* ln_rowac; const float* ln_colsum;          // consumer (fp16 outputs): per row (rstd, -mean * rstd) [M, 2], per column sum_k gamma_k W[n, k] [N]
    f16* lo_out; int lo_cols; long ld_lo;                   // k_gemm8p<.., QKLO>: the LOW fp16 half of the first lo_cols output columns, lo_out[m, n] = fp16(v - fp16(v))
#ifdef SEMABS_TUNING
    int ablate;     // tuning build only (k_gemm_f16; k_gemm8 takes its ablations as the template parameter ABL): bit0 skip in-loop DMA, bit1 skip in-loop LDS reads, bit2 skip in-loop waits+barrier, bit3 skip epilogue
    unsigned long long* trace;  // tuning build only: per-workgroup {start, main loop end, end} s_memrealtime stamps + hw id
    unsigned long long* ptrace; int ptrace_wg;   // tuning build only: s_memtime stamps of every phase of ONE workgroup (waves 0 and 4 = one per wave row), see GEMM8_STAMP
#endif
};
// Ablation switches exist only in the tuning build (build.py --tuning -> libsemabs_hip_tune.so, used by tools/); in the production library
// GEMM_ABL() is the constant false and every ablation branch is compiled out of the hot loops.
#ifdef SEMABS_TUNING
#define GEMM_ABL(bit) ((g.ablate & (bit)) != 0)
#else
#define GEMM_ABL(bit) (false)
#endif

// LDS tile rows hold BKT fp16 of K (128 B for BKT = 64, 64 B for BKT = 32); the 16-byte chunk index is XOR-ed with a
// function of the row chosen so that every 16-lane group of a ds_read_b128 fragment read hits 16 distinct 16-byte
// slots of the 256-byte bank row.
template <int BKT>
__device__ __forceinline__ int swz_off(int row, int chunk) {
    if (BKT == 64) return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
    return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// BM x BN block tile, WGM x WGN waves, each wave (BM/WGM) x (BN/WGN) as 32x32 MFMA tiles, K tile BKT, STAGES LDS buffers.
// Staging: direct-to-LDS DMA (global_load_lds_dwordx4), STAGES - 1 K tiles in flight; counted s_waitcnt vmcnt + raw
// s_barrier so the prefetches stay in flight across barriers.  One wave instruction fills 1 KB of the lane-linear LDS
// image, so the swizzle is applied to the per-lane SOURCE address (and again on the read).
template <int EPI, int BM, int BN, int BKT, int WGM, int WGN, int STAGES>
__global__ __launch_bounds__(64 * WGM * WGN, ((BM / WGM) * (BN / WGN) <= 64 * 64 && WGM * WGN >= 8) ? 4 : 2) void k_gemm_f16(GemmArgs g) {
    constexpr int NW = WGM * WGN;
    constexpr int WTM = BM / WGM, WTN = BN / WGN, TM = WTM / 32, TN = WTN / 32;
    constexpr int ROWB = BKT * 2;                           // bytes per tile row
    constexpr int RPP = 1024 / ROWB;                        // tile rows per 1 KB DMA piece
    constexpr int CPR = ROWB / 16;                          // 16-byte chunks per row
    constexpr int A_BYTES = BM * ROWB, BUF = (BM + BN) * ROWB;
    constexpr int PPW = (BM + BN) / RPP / NW;               // DMA pieces per wave per K tile
    static_assert((BM + BN) / RPP % NW == 0 && WTN == 64 && WTM % 32 == 0 && STAGES >= 2, "tile / wave layout");
    static_assert(STAGES * BUF >= NW * 8192, "epilogue staging needs 8 KB per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WGN, wn = wid % WGN;

    // XCD-aware bijective remap: hardware places block b on XCD b % 8; give each XCD a contiguous run of tiles.
    int b = blockIdx.x;
    {
        const int nb = g.n_blocks, q = nb >> 3, r = nb & 7, xcd = b & 7, k = b >> 3;
        b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    // grouped rasterisation: consecutive ids walk GROUP_M row-panels (fastest) x the column panels, so the ~32 blocks
    // resident on one XCD form a compact patch of the output and share operand panels through that XCD's L2
    int tm, tn;
    {
        const int per_group = g.group_m * g.n_tiles_n, gid = b / per_group, first_m = gid * g.group_m;
        const int gsz = (g.n_tiles_m - first_m < g.group_m) ? g.n_tiles_m - first_m : g.group_m;
        const int r = b - gid * per_group;
        tm = first_m + r % gsz; tn = r / gsz;
    }
    const long m0 = (long)tm * BM;
    const int n0 = tn * BN;

    const f16* src[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int p = wid * PPW + j;                        // wave-uniform piece id: A pieces first, then B
        const int lr = lane / CPR, cp = lane % CPR;
        if (p < BM / RPP) {
            const int row = p * RPP + lr;
            long am = m0 + row; if (am > g.M - 1) am = g.M - 1;
            src[j] = g.A + am * g.lda + ((swz_off<BKT>(row, cp) - row * ROWB) >> 1);
        } else {
            const int row = (p - BM / RPP) * RPP + lr;
            src[j] = g.B + (long)(n0 + row) * g.ldb + ((swz_off<BKT>(row, cp) - row * ROWB) >> 1);
        }
    }
    auto stage = [&](int kt) {
        char* dst = smem + (kt % STAGES) * BUF + wid * PPW * 1024;
#pragma unroll
        for (int j = 0; j < PPW; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + (long)kt * BKT),
                                             (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, 0, 0);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = g.K / BKT;
    constexpr int KS = BKT / 16;                            // MFMA k-steps per K tile (even)
    // prologue: STAGES - 1 tiles in flight, wait for the first
#pragma unroll
    for (int t = 0; t < STAGES - 1; ++t)
        if (t < nk) stage(t);
    if (nk >= STAGES - 1) wait_vmcnt<(STAGES - 2) * PPW>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    const int frow = lane & 31, fk = lane >> 5;
    // fragment ping-pong: k-step s+1 (or the first k-step of the next K tile) is read from LDS while the MFMAs of
    // k-step s run, so the matrix pipe is not parked behind ds_read latency.
    f16x8 fa[2][TM], fb[2][TN];
    auto load_frags = [&](int which, int kt, int ks) {
        const char* sa = smem + (kt % STAGES) * BUF; const char* sb = sa + A_BYTES;
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[which][i] = *reinterpret_cast<const f16x8*>(sa + swz_off<BKT>(wm * WTM + i * 32 + frow, ks * 2 + fk));
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[which][j] = *reinterpret_cast<const f16x8*>(sb + swz_off<BKT>(wn * WTN + j * 32 + frow, ks * 2 + fk));
    };
    load_frags(0, 0, 0);
    if GEMM_ABL(2) load_frags(1, 0, 1);
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + STAGES - 1 < nk && !GEMM_ABL(1)) stage(kt + STAGES - 1);  // buffer last read in iteration kt - 1 (reads retired before its barrier)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks + 1 < KS) {
                if (!GEMM_ABL(2)) load_frags(nxt, kt, ks + 1);
            } else if GEMM_ABL(4) {
                if (kt + 1 < nk && !GEMM_ABL(2)) load_frags(nxt, kt + 1, 0);
            } else {
                // tile kt + 1 must have landed for every wave before anyone reads it; later tiles stay in flight
                if (kt + STAGES - 1 < nk) wait_vmcnt<(STAGES - 2) * PPW>();
                else wait_vmcnt<0>();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (kt + 1 < nk && !GEMM_ABL(2)) load_frags(nxt, kt + 1, 0);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
        }
    }
    __builtin_amdgcn_s_barrier();                          // every wave is done with the operand tiles before they become epilogue scratch

    // ---- epilogue: per wave, 32-row slabs of its tile through a private 8 KB LDS slice -> 16-byte row accesses ----
    if (GEMM_ABL(8) && acc[0][0][0] != 12345.678f) return;
    float* st = reinterpret_cast<float*>(smem + wid * 8192);
    const int cchunk = lane & 15;          // 4 floats
    const int ncol = n0 + wn * WTN + cchunk * 4;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.bias) bv = *reinterpret_cast<const float4*>(g.bias + ncol);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                st[row * 64 + j * 32 + (lane & 31)] = acc[i][j][r];
            }
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): own writes visible to own wave
        __builtin_amdgcn_wave_barrier();
        if (EPI == EPI_BIAS_F16 || EPI == EPI_BIAS_GELU_F16) {
            // fp16 output: 8 columns (16 bytes) per lane, 8 rows per store instruction (dwordx4 stores: the store
            // tail is issue-bound, not bandwidth-bound, so half the instructions = half the tail)
            const int c8 = lane & 7;
            const int ncol8 = n0 + wn * WTN + c8 * 8;
            float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
            if (g.bias) { b0 = *reinterpret_cast<const float4*>(g.bias + ncol8); b1 = *reinterpret_cast<const float4*>(g.bias + ncol8 + 4); }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 8 + (lane >> 3);
                const long m = m0 + wm * WTM + i * 32 + row;
                if (m >= g.M) continue;
                const float4 v0 = *reinterpret_cast<const float4*>(st + row * 64 + c8 * 8);
                const float4 v1 = *reinterpret_cast<const float4*>(st + row * 64 + c8 * 8 + 4);
                float v[8] = {v0.x + b0.x, v0.y + b0.y, v0.z + b0.z, v0.w + b0.w, v1.x + b1.x, v1.y + b1.y, v1.z + b1.z, v1.w + b1.w};
                f16x8 h;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (EPI == EPI_BIAS_GELU_F16) v[e] = v[e] * __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v[e]));   // 1-ulp rcp << fp16 rounding of the result
                    h[e] = (f16)v[e];
                }
                if (GEMM_ABL(16) && v[0] != 12345.678f) continue;
                *reinterpret_cast<f16x8*>(reinterpret_cast<f16*>(g.C) + m * g.ldc + ncol8) = h;
            }
        } else {
#pragma unroll 4
        for (int it = 0; it < 8; ++it) {
            int row = it * 4 + (lane >> 4);
            long m = m0 + wm * WTM + i * 32 + row;
            if (m >= g.M) continue;
            float4 v = *reinterpret_cast<const float4*>(st + row * 64 + cchunk * 4);
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            if (GEMM_ABL(16) && v.x != 12345.678f) continue;
            if (EPI == EPI_BIAS_RESID_F32) {
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
        __builtin_amdgcn_wave_barrier();      // slab reads done before the next slab overwrites the slice
    }
}

// Per-call launch options (no process-global state: the library is re-entrant per stream).
//   kernel      : 0 = heuristic, 1 = the 128 x 128 ring kernel, 2 = the 256 x 256 phased kernel (when the shape allows it)
//   ev_start/stop: optional HIP events filled by the launch's own dispatch packet (hipExtLaunchKernelGGL) - no extra barrier packets
//                  around the kernel, unlike hipEventRecord before and after it
struct GemmOpts { int kernel; hipEvent_t ev_start, ev_stop; int ring; int v2; int nopers; };

#ifdef SEMABS_TUNING
// tuning build only (libsemabs_hip_tune.so, tools/): knobs for ablations / alternative tile configurations
static int g_group_m = 0, g_ablate = 0, g_force_cfg = 0, g_prefetch = 1, g_persist = 0, g_sc_w = 0;
static unsigned long long* g_trace = nullptr;
static unsigned long long* g_ptrace = nullptr; static int g_ptrace_wg = 0;
extern "C" int semabs_gemm_tune(int key, long long value) {
    switch (key) {
        case 0: g_force_cfg = (int)value; break;     // alternative ring-kernel tile configurations (see launch())
        case 1: g_group_m = value < 0 ? 0 : (int)value; break;     // 0 = the per-shape heuristic of gemm8_group_m()
        case 2: g_ablate = (int)value; break;
        case 3: g_prefetch = (int)value; break;
        case 5: g_persist = (int)value; break;        // 1 = persistent workgroups (one per CU)       // 1 = fragment reads in the MFMA shadow (production), 0 = read block before the barrier
        case 4: g_trace = (unsigned long long*)value; break;
        case 7: g_ptrace = (unsigned long long*)value; break;      // phase trace buffer: 2 rows x 1024 stamps (8 B each), zero = off
        case 8: g_ptrace_wg = (int)value; break;                   // which workgroup (virtual block id) is traced
        case 6: g_sc_w = (int)value; break;           // column panels per super-column (0 = one super-column)
        default: return SEMABS_EINVAL;
    }
    return SEMABS_OK;
}
#define GEMM_GROUP_M g_group_m
#define GEMM_SC_W g_sc_w
#define GEMM_PREFETCH g_prefetch
#define GEMM_TUNE_ARGS(g) do { (g).ablate = g_ablate; (g).trace = g_trace; (g).ptrace = g_ptrace; (g).ptrace_wg = g_ptrace_wg; } while (0)
#else
#define GEMM_GROUP_M 0
#define GEMM_SC_W 0
#define GEMM_PREFETCH 1
#define GEMM_TUNE_ARGS(g) do { } while (0)
#endif

template <typename Kern>
static inline void gemm_dispatch(Kern kernel, dim3 grid, dim3 block, size_t lds, hipStream_t s, const GemmArgs& g, const GemmOpts& o) {
    if (o.ev_start && o.ev_stop) hipExtLaunchKernelGGL(kernel, grid, block, lds, s, o.ev_start, o.ev_stop, 0, g);
    else hipLaunchKernelGGL(kernel, grid, block, lds, s, g);
}

template <int EPI, int BM, int BN, int BKT, int WGM, int WGN, int STAGES>
static int launch_cfg(GemmArgs g, hipStream_t s, const GemmOpts& o) {
    constexpr int LDS = STAGES * (BM + BN) * BKT * 2;
    static SemabsLdsAttr attr;
    semabs_ensure_lds(&k_gemm_f16<EPI, BM, BN, BKT, WGM, WGN, STAGES>, LDS, attr);
    g.n_tiles_n = g.N / BN;
    long mt = (g.M + BM - 1) / BM;
    g.n_tiles_m = (int)mt;
    g.group_m = GEMM_GROUP_M > 0 ? GEMM_GROUP_M : 8;
    GEMM_TUNE_ARGS(g);
    if (mt * g.n_tiles_n >= (1L << 30)) { semabs_set_error("semabs_gemm_f16: grid too large"); return SEMABS_EINVAL; }
    g.n_blocks = (int)(mt * g.n_tiles_n);
    gemm_dispatch(k_gemm_f16<EPI, BM, BN, BKT, WGM, WGN, STAGES>, dim3(g.n_blocks), dim3(64 * WGM * WGN), LDS, s, g, o);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// =================================================================================================
// 256 x 256 x 64 "phased" kernel for the large GEMMs (M >= 2048, N % 256 == 0, K >= 128).
//
// 8 waves as 2 (M) x 4 (N); a wave owns 128 x 64 of the tile as four quadrants: rows = 64 from each A half-tile, columns = 32 from
// each B half-tile, so that every 16 KB half-tile (A0, B0, B1, A1) is read by the whole workgroup in exactly ONE of the four
// phases of a K tile.  Phase = { ds_read this quadrant's new fragments | issue ONE half-tile of a later K tile as direct-to-LDS
// DMA | s_waitcnt vmcnt(8) | barrier | 16 x v_mfma_f32_16x16x32_f16 | barrier }.  The two wave rows run one barrier apart, so on
// every SIMD one wave is in its MFMA block while the other reads LDS / issues DMA (s_setprio favours the MFMA wave).
// Hazard schedule (phase numbers are global, 4 per K tile t; two LDS buffers by K-tile parity):
//   reads :  A0(t), B0(t) at 4t      B1(t) at 4t+1      A1(t) at 4t+2      (4t+3 reuses registers only)
//   stages:  B1(t+1) at 4t           A1(t+1) at 4t+1    A0(t+2) at 4t+2    B0(t+2) at 4t+3
//   -> every half-tile is re-staged >= 2 phases after its last read (needed because the wave rows are skewed by one barrier) and
//      is staged >= 4 phases before the phase that precedes its first read, so "vmcnt(8)" (four half-tiles = 8 DMA instructions
//      per lane still in flight) in every phase retires it in time: 64 KB of operands stay in flight per CU across barriers.
// Operands are swapped in the MFMA (D = B_frag x A_frag) and the B fragment rows are interleaved (n = (i >> 2) * 8 + jt * 4 + (i & 3))
// so that a lane ends up with 8 CONSECUTIVE columns of one output row (16 B of fp16 / 2 x 16 B of fp32): the unit the epilogue moves.  The
// epilogue itself does NOT store those registers directly (round 1 did: row-per-lane 16-byte stores are served at ~1 lane / clk by the
// texture-address unit); it transposes them through the wave's own 16 KB of the idle operand buffers into row-contiguous order first - see
// the comment at `epilogue` below.
// =================================================================================================
__device__ __forceinline__ int swzA8(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int swzB8(int row, int chunk) { return row * 128 + ((chunk ^ (((row >> 1) & 1) | (((row >> 3) & 3) << 1))) << 4); }

// Buffer addressing (raw buffers, stride 0): address = 48-bit base in a scalar resource descriptor + 32-bit per-lane byte offset + scalar
// byte offset.  One VGPR per lane serves every access of a tile - no 64-bit vector address arithmetic in the hot loop or the epilogue - and
// accesses past `bytes` are dropped by the hardware (loads return 0, stores are discarded), which is how the ragged last row panel is
// handled: no per-row guards anywhere.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t gemm_rsrc(const void* p, long bytes) {
    const int n = bytes > 0x7fffffffL ? 0x7fffffff : (bytes < 0 ? 0 : (int)bytes);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, n, 0x00020000);
}
// AUX = cache policy bits of the access (bit 1 = nt, "non-temporal": streaming).  The fp32 epilogues (residual read-modify-write, fp32 outputs)
// use nt for their loads and stores - the tile is touched once and should not push operand tiles out of L2: out-proj 666 -> 696, c_proj
// 1 028 -> 1 044, K|V 829 -> 846 TFLOP/s - while the fp16 outputs, which the next kernel reads, are slower with it (QKV -4 %, c_fc -5 %).
template <int AUX>
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX));
}
template <int AUX>
__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, AUX);
}

// LNP (LayerNorm fold, PRODUCER side; EPI_BIAS_RESID_F32 only): besides x += acc + bias the epilogue writes xg = fp16(x_new * gamma) - the A operand of
// the GEMM behind the LayerNorm that follows - with 16-byte stores (two rows' 8-byte halves exchanged between neighbouring lanes by DPP), and per row the
// partial sums (sum x_new, sum x_new^2) over this tile's 256 columns into ln_part[row][tile column]: reduced over the 8 lanes of a row by DPP, then over
// the 4 wave columns x 2 B halves through 16 KB of LDS behind the operand buffers in a FIXED order (no atomics: bit-reproducible, batch-size independent).
// semabs_ln_rowstats turns the partials into (rstd, -mean * rstd) per row; k_gemm8p<EPI, true> (the consumer) applies them:
//   sum_k ((x_k - mu) rstd gamma_k + beta_k) W[n,k] = rstd * sum_k xg_k W[n,k] - mu rstd * sum_k gamma_k W[n,k] + sum_k beta_k W[n,k].
template <int EPI, bool PF, bool PERS = false, bool RING = false, bool V2 = false, int ABL = 0, bool LNP = false>      // ABL: compile-time ablation bits (tuning build only; 0 in the product)
//       // PERS: persistent workgroups (see the end of the kernel); RING: the K = 32 ring schedule (see `ring`); V2: the deep schedule (see `v2`)
__global__ __launch_bounds__(512) void k_gemm8(GemmArgs g) {
    static_assert(!LNP || EPI == EPI_BIAS_RESID_F32, "the LayerNorm producer lives in the fp32 residual epilogue");
#define GABL(bit) ((ABL & (bit)) != 0)      /* 1 no in-loop DMA, 2 no in-loop LDS reads, 4 no DMA waits, 8 no epilogue, 16 no B staging */
    constexpr int HT = 16384;                               // one half-tile: 128 rows x 64 fp16
    constexpr int OFF_A0 = 0, OFF_B0 = HT, OFF_B1 = 2 * HT, OFF_A1 = 3 * HT, BUFSZ = 4 * HT;
    constexpr bool OUT16 = EPI == EPI_BIAS_F16 || EPI == EPI_BIAS_GELU_F16;
    constexpr int ES = OUT16 ? 2 : 4;                       // bytes per output element
    constexpr int EAUX = OUT16 ? 0 : 2;                     // epilogue cache policy: nt for the fp32 tiles (see buf_store4)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int l15 = lane & 15, kg = lane >> 4;

    // block id -> output tile.  XCD-aware bijective remap (hardware places block b on XCD b % 8) + grouped rasterisation, as in k_gemm_f16.
    auto tile_of = [&](int vb, long& m0, int& n0) {
        const int b = semabs_xcd_item(vb, g.n_blocks, g.reverse != 0);
        // super-columns: the column panels are walked sc_w at a time over ALL row panels, so that the sc_w weight panels of a super-column
        // (sc_w x 256 x K fp16) stay in the XCD's 4 MiB L2 for its whole pass while the activation panels stream through once per pass
        const int scw = g.sc_w > 0 && g.sc_w < g.n_tiles_n ? g.sc_w : g.n_tiles_n;
        const int per_super = g.n_tiles_m * scw;
        int sc = b / per_super;
        const int nsc = (g.n_tiles_n + scw - 1) / scw;
        if (sc > nsc - 1) sc = nsc - 1;                     // only the last super-column can be narrower
        const int rb = b - sc * per_super;
        const int width = g.n_tiles_n - sc * scw < scw ? g.n_tiles_n - sc * scw : scw;
        const int per_group = g.group_m * width, gid = rb / per_group, first_m = gid * g.group_m;
        const int gsz = (g.n_tiles_m - first_m < g.group_m) ? g.n_tiles_m - first_m : g.group_m;
        const int r = rb - gid * per_group;
        m0 = (long)(first_m + r % gsz) * 256;
        n0 = (sc * scw + r / gsz) * 256;
    };

    // ---- DMA sources: thread -> (row = j * 64 + tid / 8, 16-byte chunk position tid % 8) of every half-tile, j = 0, 1 ----
    const int srow = tid >> 3, scp = tid & 7;
    unsigned aoff[2], boff[2];                              // [j] byte offsets from the tile's first row; the half-tile / K tile add scalar offsets
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = j * 64 + srow;
        aoff[j] = (unsigned)(row * (int)g.lda * 2 + (swzA8(row, scp) - row * 128));
        boff[j] = (unsigned)(row * g.ldb * 2 + (swzB8(row, scp) - row * 128));
    }
    // (the swizzle terms of rows r and r + 64 are equal, so piece j = 1 is piece 0 + 64 rows: the deep schedule passes that through the scalar offset)
    const unsigned a_half = (unsigned)(128 * (int)g.lda * 2), b_half = (unsigned)(128 * g.ldb * 2);
    __amdgpu_buffer_rsrc_t rA, rB;                          // per tile: rows past M read as zeros (they only feed output rows that are never stored)
    auto set_tile = [&](long m0, int n0) {
        const long rows = g.M - m0 < 256 ? g.M - m0 : 256;
        rA = gemm_rsrc(g.A + m0 * g.lda, ((rows - 1) * g.lda + g.K) * 2);
        rB = gemm_rsrc(g.B + (long)n0 * g.ldb, (255L * g.ldb + g.K) * 2);
    };
    auto stage_a = [&](int h, int kt) {
        char* dst = smem + (kt & 1) * BUFSZ + (h ? OFF_A1 : OFF_A0) + wid * 1024;
        const unsigned soff = (unsigned)kt * 128 + (h ? a_half : 0u);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(dst + j * 8192), 16, aoff[j], soff, 0, 0);
    };
    auto stage_b = [&](int h, int kt) {
        char* dst = smem + (kt & 1) * BUFSZ + (h ? OFF_B1 : OFF_B0) + wid * 1024;
        const unsigned soff = (unsigned)kt * 128 + (h ? b_half : 0u);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (__attribute__((address_space(3))) void*)(dst + j * 8192), 16, boff[j], soff, 0, 0);
    };
    auto prologue = [&]() { stage_a(0, 0); stage_b(0, 0); stage_b(1, 0); stage_a(1, 0); stage_a(0, 1); stage_b(0, 1); };

    // ---- fragment read offsets (bytes within a half-tile) ----
    int offA[2], offB[2][2];
    {
        const int rowa = wr * 64 + l15;                     // + it * 16: swizzle term does not depend on it
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) offA[kk] = swzA8(rowa, kk * 4 + kg);
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            const int rowb = wc * 32 + (l15 >> 2) * 8 + jt * 4 + (l15 & 3);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) offB[jt][kk] = swzB8(rowb, kk * 4 + kg);
        }
    }
    f32x4 acc[2][4][2][2];                                  // [A half][row tile][B half][col tile]
    f16x8 fa[4][2], fb[2][2][2];                            // fa[row tile][kk] (A0 then A1 reuse it); fb[B half][col tile][kk]
    // PF (fragment prefetch in the shadow of the MFMAs): the LDS reads of the NEXT phase are issued inside the current phase's MFMA block -
    // A1 then needs its own registers (it is read while A0 is multiplied) and B0 of the next K tile a second set (read during the last
    // phase, which still multiplies this tile's B0)
    f16x8 fa1[PF ? 4 : 1][2], fb0n[PF ? 2 : 1][2];

    auto read_a = [&](int h, int par) {
        const char* base = smem + par * BUFSZ + (h ? OFF_A1 : OFF_A0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) fa[i][kk] = *reinterpret_cast<const f16x8*>(base + i * 2048 + offA[kk]);
    };
    auto read_b = [&](int h, int par) {
        const char* base = smem + par * BUFSZ + (h ? OFF_B1 : OFF_B0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) fb[h][j][kk] = *reinterpret_cast<const f16x8*>(base + offB[j][kk]);
    };
    auto mma = [&](int ha, int hb) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[ha][i][hb][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[hb][j][kk], fa[i][kk], acc[ha][i][hb][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    // between the read / DMA-issue block and the MFMA block of a phase
#define GEMM8_SYNC(staged)                                                       \
    do {                                                                         \
        if (!GABL(4)) { if (staged) wait_vmcnt<8>(); else wait_vmcnt<0>(); } /* tuning: bit 2 = no DMA waits (wrong results) */ \
        __builtin_amdgcn_sched_barrier(0);                                       \
        __builtin_amdgcn_s_barrier();                                            \
        __builtin_amdgcn_s_waitcnt(0xc07f);   /* lgkmcnt(0); the builtin (not inline asm) so that the compiler's own counter model sees it */ \
        __builtin_amdgcn_sched_barrier(0);                                       \
    } while (0)
#define GEMM8_END()                                                              \
    do {                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                       \
        __builtin_amdgcn_s_barrier();                                            \
        __builtin_amdgcn_sched_barrier(0);                                       \
    } while (0)

    // ---- epilogue: registers -> wave-private LDS transpose -> row-contiguous 16-byte accesses ----
    // After the MFMAs a lane holds, per 16 x 16 tile pair, 8 consecutive columns of ONE row (lane l15 = row, kg = column group).  Stored
    // straight from registers, the 16 lanes of a quarter wave hit 16 different rows: the texture-address unit then handles about one lane per
    // clock (measured: 16 B / clk / CU - 4-5 us per fp16 tile, 16-20 us per fp32 read-modify-write tile = as long as the K = 768 main loop,
    // with the later waves queueing behind the earlier ones).  So every (B half, A half) pass of a wave - 64 rows x 32 columns - goes
    // through 4 / 8 KB of the wave's own 16 KB slice of the (now idle) operand buffers and comes back with consecutive lanes on consecutive
    // 16-byte chunks of a row (4 lanes = a 64-byte fp16 row segment, 8 lanes = a 128-byte fp32 one): quad-coalesced dwordx4 accesses at the
    // full 64 B / clk.  No workgroup barrier: a wave only reads what it wrote (LDS is in-order per wave); the rows' 16-byte chunks are
    // XOR-swizzled so that neither the row-per-lane writes nor the row-contiguous reads conflict.  The residual tile of the fp32
    // read-modify-write is loaded in the same coalesced layout, one pass ahead.
    // Output addressing: one resource descriptor per (tile, wave) whose extent ends with the last valid row - accesses of the ragged last row
    // panel fall off it - plus a per-lane byte offset and a scalar offset per pass / instruction.
    auto epilogue = [&](const long m0, const int n0) {
        // Store-data hazard (found in round 4, gfx950 / ROCm 7.2): a VALU write to the data registers of a 16-byte buffer store issued a few instructions
        // earlier can still reach the store - the last four lanes of each 16-lane group of the LAST store of a pass carried the next pass's first
        // v_pk_add_f32 result in their second dword (non-deterministic, 0.3 % of the QuickGELU outputs, only in the instantiation whose register
        // allocation put that VALU write right behind the store).  The compiler pads this hazard only for stores WITHOUT a scalar offset register
        // (its hazard table assumes an SGPR offset makes the store issue late enough); these stores have one.  So every pass ends with 16 wait states
        // before the next pass's arithmetic may touch a register.
        auto store_pad = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        constexpr int ROWB = OUT16 ? 64 : 128;              // bytes of a pass row (32 columns)
        constexpr int LPR = ROWB / 16;                      // lanes (16-byte chunks) per row in the coalesced layout: 4 / 8
        constexpr int RPI = 64 / LPR;                       // rows per wave instruction: 16 / 8
        constexpr int NIT = 64 / RPI;                       // instructions per pass: 4 / 8
        constexpr int PASSB = 64 * ROWB;                    // 4 / 8 KB; two passes alternate inside the wave's 16 KB
        char* const ep = smem + wid * 16384;
        float bia[2][8];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            const int ncol = n0 + hb * 128 + wc * 32 + kg * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) bia[hb][e] = 0.f;
            if (g.bias) {
                const float4 b0 = *reinterpret_cast<const float4*>(g.bias + ncol), b1 = *reinterpret_cast<const float4*>(g.bias + ncol + 4);
                bia[hb][0] = b0.x; bia[hb][1] = b0.y; bia[hb][2] = b0.z; bia[hb][3] = b0.w; bia[hb][4] = b1.x; bia[hb][5] = b1.y; bia[hb][6] = b1.z; bia[hb][7] = b1.w;
            }
        }
        // register layout -> LDS (row = i * 16 + l15; this lane's 8 columns = chunk kg (fp16) / chunks 2 kg, 2 kg + 1 (fp32))
        const int wswz = OUT16 ? ((l15 >> 1) & 3) : (l15 & 7);
        // Scheduling fences around the LDS stores are load-bearing.  Left to itself the compiler interleaved the next values' VALU work with
        // the 16-byte LDS stores and re-used a store's first data VGPR five VALU instructions after the ds_write_b128 that sources it; with
        // the other wave row hammering the LDS at the same time, the last lane of each quad then stored the NEW register value
        // (non-deterministic, ~1e-4 of the fp16 outputs wrong; fp32 untouched only because its registers happened not to be re-used).  The
        // documented hazard (VALU write of the data of a > 64-bit store: 1-2 wait states) is what the compiler pads for; back-to-back wide
        // stores evidently keep reading their operands for longer.  So: all values of a pass first, fence, then the stores, fence - the
        // stores' operands are only overwritten by the NEXT pass, after this pass's LDS reads have returned.
        auto lds_write = [&](int pass) {
            const int hb = pass >> 1, ha = pass & 1;
            char* buf = ep + (pass & 1) * PASSB;
            f32x4 w[4][OUT16 ? 1 : 2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = acc[ha][i][hb][0][e] + bia[hb][e]; v[4 + e] = acc[ha][i][hb][1][e] + bia[hb][4 + e]; }
                if constexpr (OUT16) {
                    f16x8 h;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if (EPI == EPI_BIAS_GELU_F16) v[e] = v[e] * __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v[e]));
                        h[e] = (f16)v[e];
                    }
                    w[i][0] = __builtin_bit_cast(f32x4, h);                       // same access type as lds_read (no type-based reordering)
                } else {
                    w[i][0] = f32x4{v[0], v[1], v[2], v[3]};
                    w[i][OUT16 ? 0 : 1] = f32x4{v[4], v[5], v[6], v[7]};
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                char* row = buf + (i * 16 + l15) * ROWB;
                if constexpr (OUT16) {
                    *reinterpret_cast<f32x4*>(row + ((kg ^ wswz) << 4)) = w[i][0];
                } else {
                    *reinterpret_cast<f32x4*>(row + (((2 * kg) ^ wswz) << 4)) = w[i][0];
                    *reinterpret_cast<f32x4*>(row + (((2 * kg + 1) ^ wswz) << 4)) = w[i][OUT16 ? 0 : 1];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // coalesced layout: instruction `it` of a pass covers rows it * RPI + lane / LPR, this lane's chunk = lane % LPR
        const int crow = lane / LPR, cchunk = lane % LPR;
        auto lds_read = [&](int pass, int it) {
            const int row = it * RPI + crow;
            const int rswz = OUT16 ? ((row >> 1) & 3) : (row & 7);
            return *reinterpret_cast<const f32x4*>(ep + (pass & 1) * PASSB + row * ROWB + ((cchunk ^ rswz) << 4));
        };
        const long mw = m0 + wr * 64;                       // first row of this wave's 64-row strip of A half 0 (half 1: + 128)
        const unsigned ldcb = (unsigned)g.ldc * ES;         // output row pitch in bytes
        if constexpr (EPI == EPI_ROWMAP_ADD_F32) {
            // out row = (m / g_in) * g_out + g_off + m % g_in: a per-lane row map, still monotonic in m - the descriptor starts at the strip's
            // first output row and ends after the output row of the last valid m
            const long grp0 = mw / g.g_in; const long orow0 = grp0 * g.g_out + g.g_off + (mw - grp0 * g.g_in);
            const long mlast = g.M - 1; const long grpl = mlast / g.g_in; const long orowl = grpl * g.g_out + g.g_off + (mlast - grpl * g.g_in);
            const __amdgpu_buffer_rsrc_t rC = gemm_rsrc(reinterpret_cast<const float*>(g.C) + orow0 * g.ldc + (n0 + wc * 32),
                                                        (orowl - orow0) * (long)ldcb + (long)(g.N - n0 - wc * 32) * ES);
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int hb = pass >> 1, ha = pass & 1;
                lds_write(pass);
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const long m = mw + ha * 128 + it * RPI + crow;
                    const long grp = m / g.g_in; const int within = (int)(m - grp * g.g_in);
                    const long orow = grp * g.g_out + g.g_off + within;
                    f32x4 v = lds_read(pass, it);
                    if (g.addend) {
                        const float4 a = *reinterpret_cast<const float4*>(g.addend + (long)(g.g_off + within) * g.N + (n0 + hb * 128 + wc * 32 + cchunk * 4));
                        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
                    }
                    buf_store4<EAUX>(rC, (unsigned)(orow - orow0) * ldcb + (unsigned)(cchunk * 16), (unsigned)(hb * 128 * ES), v);
                }
                store_pad();
            }
        } else {
            const long rows = g.M - mw;                     // valid rows from the strip's first row on (may be <= 0: nothing is stored)
            const __amdgpu_buffer_rsrc_t rC = gemm_rsrc(reinterpret_cast<const char*>(g.C) + (mw * g.ldc + n0 + wc * 32) * ES,
                                                        rows > 0 ? (rows - 1) * (long)ldcb + (long)(g.N - n0 - wc * 32) * ES : 0);
            const unsigned voff = (unsigned)crow * ldcb + (unsigned)(cchunk * 16);
            auto soff_of = [&](int pass, int it) { return (unsigned)((pass & 1) * 128 + it * RPI) * ldcb + (unsigned)((pass >> 1) * 128 * ES); };
            if constexpr (EPI == EPI_MULROW_F16) {
                // C fp16 = (acc + bias) * addend[m % g_in, :] (addend fp32 [g_in, N]): the QuickGELU VJP of the ViT-L rollout in the epilogue of the
                // W_pr^T GEMM, with the derivative as a per-tile-row table (vit.hip k_quickgelu_grad: the 16 labels of a tile share it) - the fp32
                // product (4.2 GB per block at 16 labels x 63 tiles) is neither written nor read back.  The tile goes through LDS in fp32 like the
                // other fp32 epilogues; the table rows are loaded in the same coalesced layout one pass ahead; a lane ends up with 4 consecutive
                // columns = one 8-byte store.  (With the sigmoid evaluated here - v_exp + v_rcp per element and label - this GEMM took 2.48 ms
                // against 1.33 ms for the plain fp32-output GEMM of the same shape.)
                const unsigned ldob = (unsigned)g.ldc * 2u;
                const __amdgpu_buffer_rsrc_t rO = gemm_rsrc(reinterpret_cast<const char*>(g.C) + (mw * g.ldc + n0 + wc * 32) * 2,
                                                            rows > 0 ? (rows - 1) * (long)ldob + (long)(g.N - n0 - wc * 32) * 2 : 0);
                const unsigned ovoff = (unsigned)crow * ldob + (unsigned)(cchunk * 8);
                const float* fcol = g.addend + (n0 + wc * 32 + cchunk * 4);
                f32x4 fc[2][NIT];
                auto issue = [&](int pass) {
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        const unsigned m = (unsigned)(mw + (pass & 1) * 128 + it * RPI + crow);
                        const unsigned fr = m % (unsigned)g.g_in;
                        const float4 t = *reinterpret_cast<const float4*>(fcol + (long)fr * g.N + (pass >> 1) * 128);
                        fc[pass & 1][it] = f32x4{t.x, t.y, t.z, t.w};
                    }
                };
                issue(0);
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
                    lds_write(pass);
                    if (pass + 1 < 4) issue(pass + 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        const f32x4 v = lds_read(pass, it);
                        const f32x4 x = fc[pass & 1][it];
                        f16x4 h;
#pragma unroll
                        for (int e = 0; e < 4; ++e) h[e] = (f16)(v[e] * x[e]);
                        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, h), rO, ovoff,
                                                              (unsigned)((pass & 1) * 128 + it * RPI) * ldob + (unsigned)((pass >> 1) * 128 * 2), 0);
                    }
                    store_pad();
                }
            } else if constexpr (EPI == EPI_BIAS_RESID_F32) {
                f32x4 res[2][NIT];
                auto issue = [&](int pass) {
#pragma unroll
                    for (int it = 0; it < NIT; ++it) res[pass & 1][it] = buf_load4<EAUX>(rC, voff, soff_of(pass, it));
                };
                // LNP: the fp16 (x_new * gamma) copy and the row partials of the LayerNorm that follows (see the template comment).  Coalesced layout
                // here: 8 lanes per row (cchunk = lane % 8 holds 4 consecutive columns), 8 rows per instruction.  xg: instructions it and it + 1
                // of a pass are paired - the even lane of a pair collects row `it`'s 8 columns, the odd lane row `it + 1`'s - one 16-byte store per pair.
                [[maybe_unused]] __amdgpu_buffer_rsrc_t rG = rC;
                [[maybe_unused]] f32x4 gam[2];
                [[maybe_unused]] float* const sred = reinterpret_cast<float*>(smem + 2 * BUFSZ);        // [256 rows][4 wave columns][2 B halves][2]
                [[maybe_unused]] const unsigned ldgb = (unsigned)g.N * 2u;
                [[maybe_unused]] const bool odd = (cchunk & 1) != 0;
                [[maybe_unused]] unsigned gvoff = 0;
                if constexpr (LNP) {
                    rG = gemm_rsrc(reinterpret_cast<const char*>(g.ln_xg) + (mw * (long)g.N + n0 + wc * 32) * 2,
                                   rows > 0 ? (rows - 1) * (long)ldgb + (long)(g.N - n0 - wc * 32) * 2 : 0);
#pragma unroll
                    for (int hb = 0; hb < 2; ++hb) {
                        const float4 t = *reinterpret_cast<const float4*>(g.ln_gamma + n0 + hb * 128 + wc * 32 + cchunk * 4);
                        gam[hb] = f32x4{t.x, t.y, t.z, t.w};
                    }
                    gvoff = (unsigned)(crow + (odd ? RPI : 0)) * ldgb + (unsigned)((cchunk & ~1) * 8);
                }
                // the rows' centres (round 6): the mean the previous LayerNorm of each row saw, subtracted before the fp16 rounding and the partial sums - an
                // un-centred copy of a row with |mean| >> spread (trained checkpoints: a per-token DC offset of several sigma) spends its 11 bits on the offset
                [[maybe_unused]] float cen[2][NIT];
                if constexpr (LNP) {
                    const __amdgpu_buffer_rsrc_t rCen = gemm_rsrc(g.ln_center ? g.ln_center + mw : nullptr, (g.ln_center && rows > 0) ? rows * 4 : 0);
#pragma unroll
                    for (int ha = 0; ha < 2; ++ha)
#pragma unroll
                        for (int it = 0; it < NIT; ++it)
                            cen[ha][it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rCen, (unsigned)(ha * 128 + it * RPI + crow) * 4u, 0, 0));
                }
                auto dpp = [](float x, auto ctrl) {            // x of the lane selected by the DPP control word (all rows / banks, no bound control)
                    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, false));
                };
                using XOR1 = std::integral_constant<int, 0xB1>;  // quad_perm [1, 0, 3, 2]
                using XOR2 = std::integral_constant<int, 0x4E>;  // quad_perm [2, 3, 0, 1]
                using HMIR = std::integral_constant<int, 0x141>; // row_half_mirror: lane i <-> 7 - i of each 8
                issue(0);
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
                    lds_write(pass);
                    if (pass + 1 < 4) issue(pass + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    [[maybe_unused]] unsigned keep0 = 0, keep1 = 0;        // the even instruction's fp16 pairs, until its partner instruction arrives
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        const f32x4 v = lds_read(pass, it);
                        const f32x4 o = res[pass & 1][it];
                        const f32x4 nw = f32x4{o[0] + v[0], o[1] + v[1], o[2] + v[2], o[3] + v[3]};
                        buf_store4<EAUX>(rC, voff, soff_of(pass, it), nw);
                        if constexpr (LNP) {
                            const int hb = pass >> 1;
                            typedef f16 f16x2v __attribute__((ext_vector_type(2)));
                            const float cr = cen[pass & 1][it];
                            const f32x4 y = f32x4{nw[0] - cr, nw[1] - cr, nw[2] - cr, nw[3] - cr};
                            const f16x2v h01 = f16x2v{(f16)(y[0] * gam[hb][0]), (f16)(y[1] * gam[hb][1])};
                            const f16x2v h23 = f16x2v{(f16)(y[2] * gam[hb][2]), (f16)(y[3] * gam[hb][3])};
                            const unsigned u01 = __builtin_bit_cast(unsigned, h01), u23 = __builtin_bit_cast(unsigned, h23);
                            if ((it & 1) == 0) { keep0 = u01; keep1 = u23; }
                            else {
                                // even lanes send this (odd) instruction's values and receive the partner's even-instruction values; odd lanes the reverse
                                const unsigned snd0 = odd ? keep0 : u01, snd1 = odd ? keep1 : u23;
                                const unsigned rcv0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)snd0, 0xB1, 0xf, 0xf, false);
                                const unsigned rcv1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)snd1, 0xB1, 0xf, 0xf, false);
                                const u32x4 w = odd ? u32x4{rcv0, rcv1, u01, u23} : u32x4{keep0, keep1, rcv0, rcv1};
                                __builtin_amdgcn_raw_buffer_store_b128(w, rG, gvoff, (unsigned)((pass & 1) * 128 + (it - 1) * RPI) * ldgb + (unsigned)(hb * 128 * 2), 0);
                                __builtin_amdgcn_sched_barrier(0);      // (store-data hazard, see store_pad: no VALU write to w right behind the store)
                                asm volatile("s_nop 7" ::: "memory");
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            float s1 = (y[0] + y[1]) + (y[2] + y[3]);
                            float s2 = (y[0] * y[0] + y[1] * y[1]) + (y[2] * y[2] + y[3] * y[3]);
                            s1 += dpp(s1, XOR1{}); s2 += dpp(s2, XOR1{});
                            s1 += dpp(s1, XOR2{}); s2 += dpp(s2, XOR2{});
                            s1 += dpp(s1, HMIR{}); s2 += dpp(s2, HMIR{});
                            if (cchunk == 0) {
                                const int row = (pass & 1) * 128 + wr * 64 + it * RPI + crow;
                                *reinterpret_cast<float2*>(sred + ((row * 4 + wc) * 2 + hb) * 2) = make_float2(s1, s2);
                            }
                            // nw is the data of the 16-byte store above: keep its registers LIVE (not re-used) down to here.  With the centring the compiler formed
                            // y in place (v_sub nw, nw, c right behind the store) and the store-data hazard described at store_pad put y[1] into x for the last four
                            // lanes of every 16 (found with tools/gemm_probe.py-style row-index centres, round 6)
                            asm volatile("" :: "v"(nw[0]), "v"(nw[1]), "v"(nw[2]), "v"(nw[3]));
                        }
                    }
                    store_pad();
                }
                if constexpr (LNP) {
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_s_barrier();
                    if (tid < 256 && m0 + tid < g.M) {
                        const float* pr = sred + tid * 16;
                        float s1 = 0.f, s2 = 0.f;
#pragma unroll
                        for (int q = 0; q < 8; ++q) { s1 += pr[q * 2]; s2 += pr[q * 2 + 1]; }
                        *reinterpret_cast<float2*>(g.ln_part + ((m0 + tid) * g.n_tiles_n + n0 / 256) * 2) = make_float2(s1, s2);
                    }
                }
            } else {
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
                    lds_write(pass);
#pragma unroll
                    for (int it = 0; it < NIT; ++it) buf_store4<EAUX>(rC, voff, soff_of(pass, it), lds_read(pass, it));
                    store_pad();
                }
            }
        }
    };

    auto read_a1 = [&](int par) {                            // PF: A1 -> its own registers
        const char* base = smem + par * BUFSZ + OFF_A1;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) fa1[PF ? i : 0][kk] = *reinterpret_cast<const f16x8*>(base + i * 2048 + offA[kk]);
    };
    auto read_b0n = [&](int par) {                           // PF: B0 of the next K tile -> the spare set
        const char* base = smem + par * BUFSZ + OFF_B0;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) fb0n[PF ? j : 0][kk] = *reinterpret_cast<const f16x8*>(base + offB[j][kk]);
    };
    auto mma_a1 = [&](int hb) {                              // PF: quadrants (A1, *) from fa1
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[1][i][hb][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[hb][j][kk], fa1[PF ? i : 0][kk], acc[1][i][hb][j], 0, 0, 0);
    };
    auto mma_a0 = [&](int hb) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[0][i][hb][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[hb][j][kk], fa[i][kk], acc[0][i][hb][j], 0, 0, 0);
    };
    // one DS read slotted after every MFMA for the first `nread` MFMAs of a 16-MFMA block, the rest of the MFMAs back to back
#define GEMM8_INTERLEAVE(nread)                                                  \
    do {                                                                         \
        _Pragma("unroll") for (int q_ = 0; q_ < (nread); ++q_) {                 \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                   \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                   \
        }                                                                        \
        __builtin_amdgcn_sched_group_barrier(0x008, 16 - (nread), 0);            \
    } while (0)

#ifdef SEMABS_TUNING
    // phase trace (tools/gemm_probe.py phases): lane 0 of waves 0 and 4 writes the shader clock to the spare LDS behind the operand buffers at four
    // points of every phase: 0 phase start (before the DMA issue), 1 past the SYNC barrier, 2 MFMA block issued, 3 past the END barrier
#define GEMM8_STAMP(t_, ph_, pt_)                                                                                             \
    do {                                                                                                                      \
        if (ptr_on) *reinterpret_cast<unsigned long long*>(smem + 2 * BUFSZ + (wid >> 2) * 8192 + (((t_) * 4 + (ph_)) * 4 + (pt_)) * 8) = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define GEMM8_STAMP(t_, ph_, pt_) do { } while (0)
#endif
    const int nk = g.K / 64;                                // >= 2 (checked by the launcher)
    int vb = blockIdx.x;                                    // (a do-while whose condition is the constant false unless PERS: written as a for loop the compiler could not prove the single trip and
    do {                                                    //  kept every invariant of the epilogue live across the K loop - in a kernel with no register to spare)
    long m0; int n0;
    tile_of(vb, m0, n0);
    set_tile(m0, n0);
#ifdef SEMABS_TUNING
    const bool ptr_on = g.ptrace && vb == g.ptrace_wg && lane == 0 && (wid == 0 || wid == 4) && g.K / 64 <= 64;
#endif
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][i][c][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef SEMABS_TUNING
    unsigned long long t_start = 0, t_main = 0, c_start = 0, c_main = 0;
    if (g.trace) { t_start = __builtin_amdgcn_s_memrealtime(); c_start = __builtin_amdgcn_s_memtime(); }
#endif
    if constexpr (RING) {
        // ---- K tiles of 32 in a ring of four 32 KB buffers [A: 256 rows x 64 B | B: 256 rows x 64 B] ----------------------------------------------
        // The phased schedule above hands the matrix pipe from one wave row to the other eight times per 64 of K, and every hand-over is a
        // workgroup barrier's ~90 - 100 ticks (tools/gemm_probe.py phases): 18 % of a K tile.  Here a phase is a whole K tile of 32 = 32 MFMAs per
        // wave (all 32 accumulators once, no dependencies inside the block), i.e. half the hand-overs per flop.  Phase t of a wave row:
        //   { ds_read the 12 fragments of tile t | issue tile t + 3 (4 x 1 KB of LDS-DMA per wave) | vmcnt: tile t + 1 retired | lgkmcnt(0) |
        //     barrier | 32 MFMAs | barrier },   the two rows one barrier apart as before.
        // Hazards: tile t + 3 overwrites the buffer of tile t - 1, whose fragment reads every wave COMPLETED (lgkmcnt(0)) before its phase t - 1
        // barrier, i.e. before the other row's phase-(t - 1) END barrier that precedes this stage; tile t is read at the start of phase t, after
        // the END barrier of phase t - 1, which the other row reaches only after the vmcnt wait that retired ITS share of tile t (three tiles =
        // 96 KB per CU stay in flight across barriers).  Rows are 64 B: chunk ^= {0, 2, 3, 1}[(row >> 2) & 3] puts the 16 lanes a ds_read_b128 is
        // served with ({rows 0-3, 12-15} of one k group + {rows 4-11} of the next) on 16 different 16-byte slots, for the A rows and for the
        // interleaved B rows alike; the swizzle is applied on the DMA source address.
        constexpr int RB = 32768;
        auto g4 = [](int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; };
        unsigned raoff[2], rboff[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (wid + 8 * j) * 16 + (lane >> 2);
            const int sc = (lane & 3) ^ g4(row);
            raoff[j] = (unsigned)(row * (int)g.lda * 2 + sc * 16);
            rboff[j] = (unsigned)(row * g.ldb * 2 + sc * 16);
        }
        auto stage_r = [&](int kt) {
            char* dst = smem + (kt & 3) * RB + wid * 1024;
            const unsigned soff = (unsigned)kt * 64;
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(dst + j * 8192), 16, raoff[j], soff, 0, 0);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (__attribute__((address_space(3))) void*)(dst + 16384 + j * 8192), 16, rboff[j], soff, 0, 0);
        };
        const int roffA = (wr * 64 + l15) * 64 + ((kg ^ g4(l15)) << 4);
        int roffB[2];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            const int rowb = wc * 32 + (l15 >> 2) * 8 + jt * 4 + (l15 & 3);
            roffB[jt] = 16384 + rowb * 64 + ((kg ^ g4(rowb)) << 4);
        }
        f16x8 fra[2][4], frb[2][2];
        const int nk32 = g.K / 32;                           // >= 4 (K >= 128)
        stage_r(0); stage_r(1); stage_r(2);
        wait_vmcnt<8>();                                     // this wave's share of tile 0
        __builtin_amdgcn_s_barrier();
        if (wr == 1) __builtin_amdgcn_s_barrier();           // skew the second wave row by one barrier
        for (int t = 0; t < nk32; ++t) {
            const char* buf = smem + (t & 3) * RB;
            if (t + 3 < nk32) stage_r(t + 3);               // (DMA first: its issue slots are the long part of the segment, the reads' latency hides behind them)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ha = 0; ha < 2; ++ha)
#pragma unroll
                for (int i = 0; i < 4; ++i) fra[ha][i] = *reinterpret_cast<const f16x8*>(buf + ha * 8192 + i * 1024 + roffA);
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) frb[hb][jt] = *reinterpret_cast<const f16x8*>(buf + hb * 8192 + roffB[jt]);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 3 < nk32) wait_vmcnt<8>();
            else if (t + 2 < nk32) wait_vmcnt<4>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0): the fragment reads are COMPLETE before the barrier (see the hazards above)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ha = 0; ha < 2; ++ha)
#pragma unroll
                for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[ha][i][hb][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(frb[hb][j], fra[ha][i], acc[ha][i][hb][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
    prologue();
    if constexpr (PF) stage_b(1, 1);                        // the prefetching loop stages one phase earlier (below)
    if constexpr (V2) { stage_a(1, 1); wait_vmcnt<10>(); }  // the deep schedule another one: K tiles 0 and 1 whole, the first three half-tiles retired
    else wait_vmcnt<8>();                                   // A0(0), B0(0) (PF: and B1(0)) have landed (this wave's share)
    __builtin_amdgcn_s_barrier();
    if constexpr (PF && V2) {
        // The deep schedule re-stages the slots of A0(0) / B0(0) in phases 0 / 1 of the first K tile, so their reads - which precede the loop - must
        // be complete in EVERY wave before the first wave enters phase 0: read here, before the skew, and close with a barrier of all eight waves
        // (found the hard way: with the reads behind the skew barrier the second wave row read A0(2) where it expected A0(0)).
        read_b(0, 0); read_a(0, 0);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
    }
    if (wr == 1) __builtin_amdgcn_s_barrier();              // skew the second wave row by one barrier
    if constexpr (PF && V2) {
        // ---- "deep" schedule: the PF schedule below with FIVE half-tiles (80 KB) of DMA in flight per CU instead of four, in the same eight slots, and
        // exact counted waits while the queue drains.  Stage index s = 4 t + {A0: 0, B0: 1, B1: 2, A1: 3}; stage s is ISSUED in global phase s - 8 (the
        // prologue issues stages 0 - 7), READ in phase s - 2 (inside that phase's MFMA block, for the next phase) and its slot is re-staged in phase s:
        // exactly two phases after the last read of the slot's previous occupant - the PF schedule re-stages after three, that slack is the fifth
        // half-tile.  Per K tile t: phase 0 stages A0(t + 2), phase 1 B0(t + 2), phase 2 B1(t + 2), phase 3 A1(t + 2); reads as in the PF schedule.
        //   landed: a wave's wait in phase P leaves its five newest stage calls (P + 4 .. P + 8) in flight, so its share of stages <= P + 3 has landed;
        //           the reader of stage P + 2 in phase P has passed a barrier that the other wave row reached after ITS wait of phase P - 1 (<= P + 2).
        //   free  : the slot of stage s - 8 was read in phase s - 10; every wave completes those reads (lgkmcnt(0) IN FRONT of the barrier here) before
        //           its SYNC barrier of phase s - 9, which precedes - for either wave row - the END barrier that opens phase s - 8.
        //   drain : K tile nk - 2 stages nothing and waits vmcnt 8 / 6 / 4 / 2 (the last K tile's half-tiles, one read per phase from the next phase
        //           on), K tile nk - 1 vmcnt(0) once - the PF schedule waits vmcnt(0) as soon as nothing is staged and exposes a full memory latency
        //           per output tile there.
        // The priority raise and the LDS-read completion wait sit in front of the SYNC barrier: what follows the barrier is matrix-pipe idle time.
        // Lean load segments: what a wave executes between its END barrier and its next SYNC barrier runs beside the other row's MFMA block, but what
        // sits between the SYNC barrier and the first MFMA is matrix-pipe idle time on all four SIMDs, eight times per K tile - so the fragment
        // addresses of the coming block (base + parity offset: two VALU adds) are computed in front of the barrier and pinned there with an empty asm,
        // the LDS-read completion wait and the priority raise sit in front of it too, the steady-state K tiles (t + 2 < nk) carry no scalar branch, and
        // the second column tile of a B fragment pair is the first one + 512 bytes (four rows further, same swizzle term): an immediate offset.
        typedef const __attribute__((address_space(3))) char* lds_cptr;
        typedef const __attribute__((address_space(3))) f16x8* lds_frag;
        auto pin = [&](int x) { lds_cptr p = (lds_cptr)smem + x; asm volatile("" : "+v"(p)); return p; };
        auto ktile = [&](auto steady_c, const int t) {
            constexpr bool STEADY = decltype(steady_c)::value;
            const int pb = (t & 1) * BUFSZ, pn = pb ^ BUFSZ;
            const bool pen = t + 2 == nk;
            lds_cptr r0, r1;
#define GEMM8_SYNCD(k_)                                                          \
    do {                                                                         \
        if (!GABL(4)) {                                                          \
            if (STEADY) wait_vmcnt<10>();                                        \
            else if (pen) wait_vmcnt<8 - 2 * (k_)>();                            \
            else if ((k_) == 0) wait_vmcnt<0>();                                 \
        }                                                                        \
        __builtin_amdgcn_s_waitcnt(0xc07f);                                      \
        __builtin_amdgcn_sched_barrier(0);                                       \
        __builtin_amdgcn_s_setprio(1);                                           \
        __builtin_amdgcn_s_barrier();                                            \
        __builtin_amdgcn_sched_barrier(0);                                       \
    } while (0)
#define GEMM8_ENDD()                                                             \
    do {                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                       \
        __builtin_amdgcn_s_setprio(0);                                           \
        __builtin_amdgcn_s_barrier();                                            \
        __builtin_amdgcn_sched_barrier(0);                                       \
    } while (0)
            // phase 0: quadrant (A0, B0); reads B1(t)
            if (STEADY && !GABL(1)) stage_a(0, t + 2);
            r0 = pin(offB[0][0] + pb); r1 = pin(offB[0][1] + pb);
            GEMM8_SYNCD(0);
            if (!GABL(2)) {
#pragma unroll
                for (int j = 0; j < 2; ++j) { fb[1][j][0] = *(lds_frag)(r0 + (OFF_B1 + j * 512)); fb[1][j][1] = *(lds_frag)(r1 + (OFF_B1 + j * 512)); }
            }
            mma_a0(0);
            if (!GABL(2)) GEMM8_INTERLEAVE(4);
            GEMM8_ENDD();
            // phase 1: quadrant (A0, B1); reads A1(t)
            if (STEADY && !GABL(1)) stage_b(0, t + 2);
            r0 = pin(offA[0] + pb); r1 = pin(offA[1] + pb);
            GEMM8_SYNCD(1);
            if (!GABL(2)) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { fa1[PF ? i : 0][0] = *(lds_frag)(r0 + (OFF_A1 + i * 2048)); fa1[PF ? i : 0][1] = *(lds_frag)(r1 + (OFF_A1 + i * 2048)); }
            }
            mma_a0(1);
            if (!GABL(2)) GEMM8_INTERLEAVE(8);
            GEMM8_ENDD();
            // phase 2: quadrant (A1, B1); reads A0(t + 1) (last K tile: a stale slot, never used - keeps the block branch-free)
            if (STEADY && !GABL(1)) stage_b(1, t + 2);
            r0 = pin(offA[0] + pn); r1 = pin(offA[1] + pn);
            GEMM8_SYNCD(2);
            if (!GABL(2)) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { fa[i][0] = *(lds_frag)(r0 + (OFF_A0 + i * 2048)); fa[i][1] = *(lds_frag)(r1 + (OFF_A0 + i * 2048)); }
            }
            mma_a1(1);
            if (!GABL(2)) GEMM8_INTERLEAVE(8);
            GEMM8_ENDD();
            // phase 3: quadrant (A1, B0); reads B0(t + 1) into the spare set
            if (STEADY && !GABL(1)) stage_a(1, t + 2);
            r0 = pin(offB[0][0] + pn); r1 = pin(offB[0][1] + pn);
            GEMM8_SYNCD(3);
            if (!GABL(2)) {
#pragma unroll
                for (int j = 0; j < 2; ++j) { fb0n[PF ? j : 0][0] = *(lds_frag)(r0 + (OFF_B0 + j * 512)); fb0n[PF ? j : 0][1] = *(lds_frag)(r1 + (OFF_B0 + j * 512)); }
            }
            mma_a1(0);
            if (!GABL(2)) GEMM8_INTERLEAVE(4);
            GEMM8_ENDD();
            __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) if (!GABL(2)) fb[0][j][kk] = fb0n[PF ? j : 0][kk];
        };
        if (GABL(2)) { read_b(1, 0); read_a1(0); }      // ablation: every fragment register holds real data once
        int t = 0;
        for (; t + 2 < nk; ++t) ktile(std::integral_constant<bool, true>{}, t);
        for (; t < nk; ++t) ktile(std::integral_constant<bool, false>{}, t);
#undef GEMM8_SYNCD
#undef GEMM8_ENDD
    } else
    if constexpr (PF) {
        // Fragment reads one phase ahead, inside the MFMA blocks (see the register comment above).  What is read where, K tile t:
        //   phase 0 (A0, B0): B1(t)      phase 1 (A0, B1): A1(t)      phase 2 (A1, B1): A0(t + 1)      phase 3 (A1, B0): B0(t + 1) -> spare set
        // Reading a half-tile one phase earlier means the OTHER wave row (one barrier behind) must have retired its DMA share of it one
        // phase earlier too, so the staging schedule moves one phase earlier as well (the slots are free by then - their last reads also
        // moved up).  Global phase numbers P = 4 t + phase; reads at P need stages at <= P - 5 (vmcnt(8) leaves the 4 newest stage calls in
        // flight, and the reader's barrier pairs with the other row's previous phase):
        //   stages:  A1(t + 1) at 4t      A0(t + 2) at 4t + 1      B0(t + 2) at 4t + 2      B1(t + 2) at 4t + 3
        //   reads :  B1(t) at 4t [staged 4t - 5]   A1(t) at 4t + 1 [4t - 4]   A0(t + 1) at 4t + 2 [4t - 3]   B0(t + 1) at 4t + 3 [4t - 2]
        // and every slot is re-staged >= 2 phases after its last read (A1: read 4t - 3, staged 4t; A0: 4t - 2 / 4t + 1; B0: 4t - 1 / 4t + 2;
        // B1: 4t / 4t + 3).  The prologue therefore also stages B1(1) (seven half-tiles, the first three retired by vmcnt(8)).
        read_b(0, 0); read_a(0, 0);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        for (int t = 0; t < nk; ++t) {
            const int par = t & 1;
            const bool s1 = t + 1 < nk && !GABL(1), s2 = t + 2 < nk && !GABL(1);
            GEMM8_STAMP(t, 0, 0);
            if (s1) stage_a(1, t + 1);
            GEMM8_SYNC(s1);
            GEMM8_STAMP(t, 0, 1);
            __builtin_amdgcn_s_setprio(1);
            if (!GABL(16)) read_b(1, par);          // tuning bit 4: no B staging, no B fragment reads (wrong results: what would B from registers buy?)
            mma_a0(0);
            GEMM8_INTERLEAVE(4);
            __builtin_amdgcn_s_setprio(0);
            GEMM8_STAMP(t, 0, 2);
            GEMM8_END();
            GEMM8_STAMP(t, 0, 3);
            GEMM8_STAMP(t, 1, 0);
            if (s2) stage_a(0, t + 2);
            GEMM8_SYNC(s2);
            GEMM8_STAMP(t, 1, 1);
            __builtin_amdgcn_s_setprio(1);
            read_a1(par); mma_a0(1);
            GEMM8_INTERLEAVE(8);
            __builtin_amdgcn_s_setprio(0);
            GEMM8_STAMP(t, 1, 2);
            GEMM8_END();
            GEMM8_STAMP(t, 1, 3);
            GEMM8_STAMP(t, 2, 0);
            if (s2 && !GABL(16)) stage_b(0, t + 2);
            GEMM8_SYNC(s2);
            GEMM8_STAMP(t, 2, 1);
            __builtin_amdgcn_s_setprio(1);
            read_a(0, par ^ 1);                      // (last K tile: reads a stale slot, never used - keeps the block branch-free)
            mma_a1(1);
            GEMM8_INTERLEAVE(8);
            __builtin_amdgcn_s_setprio(0);
            GEMM8_STAMP(t, 2, 2);
            GEMM8_END();
            GEMM8_STAMP(t, 2, 3);
            GEMM8_STAMP(t, 3, 0);
            if (s2 && !GABL(16)) stage_b(1, t + 2);
            GEMM8_SYNC(s2);
            GEMM8_STAMP(t, 3, 1);
            __builtin_amdgcn_s_setprio(1);
            if (!GABL(16)) read_b0n(par ^ 1);
            mma_a1(0);
            GEMM8_INTERLEAVE(4);
            __builtin_amdgcn_s_setprio(0);
            GEMM8_STAMP(t, 3, 2);
            GEMM8_END();
            GEMM8_STAMP(t, 3, 3);
            __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) fb[0][j][kk] = fb0n[PF ? j : 0][kk];
        }
    } else
    for (int t = 0; t < nk; ++t) {
        const int par = t & 1;
        const bool s1 = t + 1 < nk && !GABL(1), s2 = t + 2 < nk && !GABL(1);
        // phase 0: quadrant (A0, B0)
        if (!GABL(2) || t == 0) { read_b(0, par); __builtin_amdgcn_sched_barrier(0); read_a(0, par); }
        if (s1) stage_b(1, t + 1);
        GEMM8_SYNC(s1);
        mma(0, 0);
        GEMM8_END();
        // phase 1: quadrant (A0, B1)
        if (!GABL(2) || t == 0) read_b(1, par);
        if (s1) stage_a(1, t + 1);
        GEMM8_SYNC(s1);
        mma(0, 1);
        GEMM8_END();
        // phase 2: quadrant (A1, B1)
        if (!GABL(2) || t == 0) read_a(1, par);
        if (s2) stage_a(0, t + 2);
        GEMM8_SYNC(s2);
        mma(1, 1);
        GEMM8_END();
        // phase 3: quadrant (A1, B0) - fragments already in registers
        if (s2) stage_b(0, t + 2);
        GEMM8_SYNC(s2);
        mma(1, 0);
        GEMM8_END();
    }
    }                                                       // (RING / phased)
    if (wr == 0) __builtin_amdgcn_s_barrier();              // balance the skew barrier
    // Every DMA has been waited for (vmcnt(0) in the last K tile's phases): the epilogue's ordinary loads (bias, residual) never share the
    // vmcnt queue with LDS-DMA loads.  That matters: counted vmcnt waits assume in-order return, and on gfx950 an ordinary load issued
    // BEHIND outstanding LDS-DMA loads was observed to be reported complete too early (a persistent variant that issued the next tile's
    // first half-tiles before the epilogue returned garbage in lanes 12-15 of the first bias fragment, non-deterministically).
#ifdef SEMABS_TUNING
    if (g.trace) { t_main = __builtin_amdgcn_s_memrealtime(); c_main = __builtin_amdgcn_s_memtime(); }
#endif
    if (!GABL(8)) epilogue(m0, n0);
    else {                                                  // ablation: keep the accumulators (and with them the whole K loop) alive
        float keep = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int j = 0; j < 2; ++j) keep += acc[a][i][c][j][0] + acc[a][i][c][j][1] + acc[a][i][c][j][2] + acc[a][i][c][j][3];
        if (keep == 12345.678f) reinterpret_cast<float*>(g.C)[tid] = keep;
    }
#ifdef SEMABS_TUNING
    if (g.ptrace && vb == g.ptrace_wg) {                    // every wave of the traced workgroup takes part in the barrier; waves 0 and 4 copy their rows out
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        if (wid == 0 || wid == 4) {
            const int nst = g.K / 64 * 16;
            for (int i = lane; i < nst && i < 1024; i += 64)
                g.ptrace[(wid >> 2) * 1024 + i] = *reinterpret_cast<unsigned long long*>(smem + 2 * BUFSZ + (wid >> 2) * 8192 + i * 8);
        }
    }
#endif
#ifdef SEMABS_TUNING
    if (g.trace) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned long long* tr = g.trace + (size_t)vb * 8;       // 100 MHz stamps, hardware id, shader-clock stamps (effective clock = cycles / time)
            tr[0] = t_start; tr[1] = t_main; tr[2] = t_end; tr[3] = ((unsigned long long)xcc << 32) | hw; tr[4] = c_start; tr[5] = c_main;
        }
    }
#endif
    if (PERS) {
        // Persistent workgroups (one per CU): the next tile's first half-tiles are requested while this tile's stores drain - a workgroup is
        // not retired before its stores complete, and the next one cannot start before it is (the operand buffers take 128 of the 160 KB):
        // 0.9 - 5 us per tile (tools/gemm_probe.py trace).  Every wave is done with its epilogue slice of LDS before the DMA overwrites it;
        // no ordinary load is outstanding at this point (the residual rows were consumed before their stores).
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
    }
    vb += (int)gridDim.x;
    } while (PERS && vb < g.n_blocks);
#undef GABL
#undef GEMM8_SYNC
#undef GEMM8_END
#undef GEMM8_INTERLEAVE
#undef GEMM8_STAMP
}

// =================================================================================================
// k_gemm8p: the deep schedule as PERSISTENT workgroups with the next tile's prologue inside the current tile's drain (fp16-output epilogues).
//
// A 256 x 256 x 768 tile is ~16 us of steady-state K loop between ~8 us of fixed cost (tools/gemm_probe.py abltrace: 2 795 cycles per K tile over
// a K = 768 tile against 2 214 in the steady state of K = 3 072; trace: 1.4 us from a tile's last store to the next workgroup's first instruction):
// the wait for the first half-tiles of the prologue, the drain of the DMA queue, the epilogue, the acknowledgement of its stores (a workgroup is not
// retired before that) and the dispatch of the next workgroup.  Round 4's first attempt ("V3", a variant of k_gemm8 since removed: all eight half-tiles of the next tile
// requested between the K loop and the epilogue, vmcnt(0) at the next tile's start) lost: 128 DMA instructions in front of the epilogue's stores
// and a wait for the acknowledgement of those stores.  Here the workgroup stays and the stage sequence simply NEVER ENDS: stage index
// s = 4 t + j keeps counting across tiles, so the last two K tiles of a tile - which stage nothing in k_gemm8 - stage the next tile's K tiles 0 and 1
// through the next tile's descriptors, one half-tile per phase as always, into the slots the steady-state rule frees anyway (slot of stage s - 8,
// last read two phases ago).  Consequences:
//   * no prologue latency, no drain, no dispatch gap after the first tile; no extra DMA issue time (the drain phases' issue slots were idle);
//   * every wait is the steady-state vmcnt(10).  Behind the epilogue the queue also holds its 16 stores per wave, OLDER than the DMA a wait is
//     meant to leave in flight: "at most 10 operations outstanding" then retires more than needed (some stores too), never less - the only property
//     relied on is that LOADS retire in order among loads;
//   * the epilogue must not touch the operand slots (they are being refilled): it transposes through 2 KB per wave BEHIND the slots (8 half
//     passes of 32 rows x 32 columns), and loads nothing from memory: the tile's bias row (256 fp32 = 1 KB) travels by LDS-DMA as well - one more
//     "half-tile" issued by wave 0 in front of the tile's first stage, into one of two 1 KB buffers (the next tile's row lands while this tile's
//     epilogue reads its own) - so the kernel issues no ordinary load at all, only LDS-DMA loads and stores.  The bias is added in the epilogue,
//     after the accumulation, as in k_gemm8 (starting the accumulators at it changed the fp32 summation order: headline L-inf 9.0e-4 -> 9.8e-4).
// Requires an even number of K tiles (buffer parity continues across tiles).  Slot map, fragment layout and MFMA order are k_gemm8's: results are
// bit-identical to it.
// =================================================================================================
// LNC (LayerNorm fold, CONSUMER side; fp16 outputs): C = a_row * acc + (c_row * colsum[n] + bias'[n]) with (a, c) = (rstd, -mean * rstd) of the row
// (semabs_ln_rowstats) - LN(x) W^T + b without materialising LN(x); A is the producer's xg = fp16(x * gamma), bias' = b + W beta.  The tile's colsum row
// (1 KB) and its 256 (a, c) pairs (2 KB) travel by LDS-DMA like the bias row - waves 1, 2, 3 issue one more 16-byte-per-lane load each in front of the
// tile's first stage, exactly where wave 0 issues the bias row - so the kernel still issues no ordinary load and every counted wait keeps its meaning.
// QKLO (precision = "parity"; EPI_BIAS_F16 only): tiles whose columns lie below g.lo_cols (the q | k columns of the QKV projection; lo_cols % 256 == 0)
// also store the LOW half of every output, lo_out[m, n] = fp16(v - fp16(v)), through the same transposition - the attention kernel then forms the
// scores from hi + lo pairs (vit.hip k_attention<.., SPLIT>).  Such a tile issues 32 stores per wave instead of 16: the positional waits behind its
// epilogue allow 42 outstanding operations where they allow 26 behind a plain tile (see VM_AFTER_EPI).
template <int EPI, bool LNC = false, bool QKLO = false>
__global__ __launch_bounds__(512) void k_gemm8p(GemmArgs g) {
    static_assert(!QKLO || EPI == EPI_BIAS_F16, "the low halves exist for the plain fp16 output only");
    static_assert(EPI == EPI_BIAS_F16 || EPI == EPI_BIAS_GELU_F16 || EPI == EPI_BIAS_RESID_F32, "fp16-output epilogues and the fp32 residual read-modify-write");
    static_assert(!LNC || EPI != EPI_BIAS_RESID_F32, "the LayerNorm consumer has fp16 outputs");
    constexpr bool OUT16 = EPI != EPI_BIAS_RESID_F32;
    constexpr int HT = 16384, OFF_A0 = 0, OFF_B0 = HT, OFF_B1 = 2 * HT, OFF_A1 = 3 * HT, BUFSZ = 4 * HT;
    constexpr int OFF_SCR = 2 * BUFSZ, OFF_BIAS = OFF_SCR + 8 * 2048;      // epilogue scratch (2 KB per wave), bias rows of this and the next tile (2 x 1 KB)
    constexpr int OFF_CS = OFF_BIAS + 2 * 1024, OFF_RA = OFF_CS + 2 * 1024; // LNC: colsum rows (2 x 1 KB) and (rstd, -mean rstd) pairs of the tile's 256 rows (2 x 2 KB), by tile parity
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int l15 = lane & 15, kg = lane >> 4;
    const bool has_bias = g.bias != nullptr;

    auto tile_of = [&](int vb, long& m0, int& n0) {          // (as in k_gemm8)
        const int b = semabs_xcd_item(vb, g.n_blocks, g.reverse != 0);
        const int per_group = g.group_m * g.n_tiles_n, gid = b / per_group, first_m = gid * g.group_m;
        const int gsz = (g.n_tiles_m - first_m < g.group_m) ? g.n_tiles_m - first_m : g.group_m;
        const int r = b - gid * per_group;
        m0 = (long)(first_m + r % gsz) * 256;
        n0 = (r / gsz) * 256;
    };
    const int srow = tid >> 3, scp = tid & 7;
    const unsigned aoff = (unsigned)(srow * (int)g.lda * 2 + (swzA8(srow, scp) - srow * 128));
    const unsigned boff = (unsigned)(srow * g.ldb * 2 + (swzB8(srow, scp) - srow * 128));
    const unsigned a_half = (unsigned)(128 * (int)g.lda * 2), b_half = (unsigned)(128 * g.ldb * 2);
    const unsigned a_q = (unsigned)(64 * (int)g.lda * 2), b_q = (unsigned)(64 * g.ldb * 2);      // second 64-row piece of a half-tile: same swizzle term
    __amdgpu_buffer_rsrc_t rA, rB, rAn, rBn;                // this tile's / the next tile's operand panels
    const __amdgpu_buffer_rsrc_t rBias = gemm_rsrc(g.bias, has_bias ? (long)g.N * 4 : 0);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rCs = gemm_rsrc(g.ln_colsum, LNC ? (long)g.N * 4 : 0);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rRa = gemm_rsrc(g.ln_rowac, LNC ? (g.M * 8 > 0x7fffffffL ? 0x7fffffffL : g.M * 8) : 0);
    auto rsrc_a = [&](long m0) { const long rows = g.M - m0 < 256 ? g.M - m0 : 256; return gemm_rsrc(g.A + m0 * g.lda, ((rows - 1) * g.lda + g.K) * 2); };
    auto rsrc_b = [&](int n0) { return gemm_rsrc(g.B + (long)n0 * g.ldb, (255L * g.ldb + g.K) * 2); };
    // stage j (0 A0, 1 B0, 2 B1, 3 A1) of K tile kt of this (nx = false) or the next tile: two 8 KB pieces per half-tile, 1 KB per wave each
    auto stage = [&](const int j, const int kt, const bool nx) {
        const bool isa = j == 0 || j == 3, hi = j >= 2;
        char* dst = smem + (kt & 1) * BUFSZ + (j == 0 ? OFF_A0 : (j == 1 ? OFF_B0 : (j == 2 ? OFF_B1 : OFF_A1))) + wid * 1024;
        const unsigned soff = (unsigned)kt * 128 + (hi ? (isa ? a_half : b_half) : 0u);
        const __amdgpu_buffer_rsrc_t r = isa ? (nx ? rAn : rA) : (nx ? rBn : rB);
#ifdef SEMABS_TUNING
        if ((isa && (g.ablate & 64)) || (!isa && (g.ablate & 128))) {      // cache-policy experiment: streaming (nt) operand loads
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(dst), 16, isa ? aoff : boff, soff, 0, 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(dst + 8192), 16, isa ? aoff : boff, soff + (isa ? a_q : b_q), 0, 2);
            return;
        }
#endif
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(dst), 16, isa ? aoff : boff, soff, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(dst + 8192), 16, isa ? aoff : boff, soff + (isa ? a_q : b_q), 0, 0);
    };
    auto stage_bias = [&](const long m0, const int n0, const int bpar) {    // wave 0: bias[n0 .. n0 + 255] -> LDS (two 1 KB buffers, by tile parity), 16 bytes per lane
        if (wid == 0 && has_bias)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rBias, (__attribute__((address_space(3))) void*)(smem + OFF_BIAS + bpar * 1024), 16, (unsigned)lane * 16u, (unsigned)n0 * 4u, 0, 0);
        if constexpr (LNC) {                                 // wave 1: colsum[n0 .. n0 + 255]; waves 2, 3: (a, c) of rows m0 .. m0 + 127 / m0 + 128 .. m0 + 255 (8 bytes per row)
            if (wid == 1)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rCs, (__attribute__((address_space(3))) void*)(smem + OFF_CS + bpar * 1024), 16, (unsigned)lane * 16u, (unsigned)n0 * 4u, 0, 0);
            if (wid == 2 || wid == 3)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rRa, (__attribute__((address_space(3))) void*)(smem + OFF_RA + bpar * 2048 + (wid - 2) * 1024), 16,
                                                         (unsigned)lane * 16u, (unsigned)(m0 * 8 + (wid - 2) * 1024), 0, 0);
        }
    };
    int offA[2], offB[2];                                   // fragment read offsets inside a half-tile (column tile jt = 1 of B: + 512 bytes)
    {
        const int rowa = wr * 64 + l15, rowb = wc * 32 + (l15 >> 2) * 8 + (l15 & 3);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) { offA[kk] = swzA8(rowa, kk * 4 + kg); offB[kk] = swzB8(rowb, kk * 4 + kg); }
    }
    f32x4 acc[2][4][2][2];
    f16x8 fa[4][2], fa1[4][2], fb[2][2][2], fb0n[2][2];
    auto mma = [&](const int ha, const int hb, const f16x8 (&a)[4][2], const f16x8 (&b)[2][2]) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[ha][i][hb][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j][kk], a[i][kk], acc[ha][i][hb][j], 0, 0, 0);
    };
    typedef const __attribute__((address_space(3))) char* lds_cptr;
    typedef const __attribute__((address_space(3))) f16x8* lds_frag;
    auto pin = [&](int x) { lds_cptr p = (lds_cptr)smem + x; asm volatile("" : "+v"(p)); return p; };
    auto rd_a = [&](f16x8 (&dst)[4][2], lds_cptr r0, lds_cptr r1, const int off) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { dst[i][0] = *(lds_frag)(r0 + (off + i * 2048)); dst[i][1] = *(lds_frag)(r1 + (off + i * 2048)); }
    };
    auto rd_b = [&](f16x8 (&dst)[2][2], lds_cptr r0, lds_cptr r1, const int off) {
#pragma unroll
        for (int j = 0; j < 2; ++j) { dst[j][0] = *(lds_frag)(r0 + (off + j * 512)); dst[j][1] = *(lds_frag)(r1 + (off + j * 512)); }
    };
#define P_INTERLEAVE(nread)                                                      \
    do {                                                                         \
        _Pragma("unroll") for (int q_ = 0; q_ < (nread); ++q_) {                 \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                   \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                   \
        }                                                                        \
        __builtin_amdgcn_sched_group_barrier(0x008, 16 - (nread), 0);            \
    } while (0)
    const int nk = g.K / 64;                                // even, >= 2 (checked by the launcher)
    // operations that may be outstanding in the first five phases behind an epilogue: 10 + what the epilogue issued per wave (16 stores; the fp32
    // read-modify-write issues 32 loads + 32 stores: 74 does not fit the 6-bit counter, 63 is merely a little stricter - it also retires the first
    // residual loads, which completed long ago)
    constexpr int VM_AFTER_EPI = OUT16 ? 26 : 63;
    // One K tile = four phases; STEADY: a half-tile is staged in every phase (kts / nx say which) and the wait is vmcnt(10); else (the last two K
    // tiles of the workgroup's LAST tile) nothing is staged and the queue drains with exact counts, as in k_gemm8.
    // Waits behind an epilogue.  From a workgroup's second tile on, the 16 stores of the previous tile's epilogue sit in the queue between the prologue
    // stages (0 - 7 of this tile, issued during the previous tile's last two K tiles) and the stages issued since.  vmcnt(n) is positional - everything
    // older than the n most recently ISSUED operations has completed (in-order retirement: the property every counted wait of this kernel, and the
    // compiler's own for mixed loads and stores on this target, rests on) - so "stages <= P + 3 landed" in global phase P of such a tile reads:
    // 2 (4 - P) older-stage instructions + 16 stores + 2 (P + 1) newer ones = 26 may be outstanding, for P = 0 .. 4; from P = 5 on the stage wanted is
    // younger than the stores and the count is the steady-state 10 again.  Waiting vmcnt(10) there instead forced the acknowledgement of the stores:
    // 0.5 us per QKV tile, 1.95 us per c_fc tile (whose QuickGELU epilogue issues its stores late) - tools/gemm_probe.py v3trace.
    bool prev_lo = false;                                   // QKLO: the previous tile of this workgroup stored low halves too (32 stores in the queue)
    auto ktile = [&](auto steady_c, const int t, const int kts, const bool nx, const int st_ph) {
        constexpr bool STEADY = decltype(steady_c)::value;
        const int pb = (t & 1) * BUFSZ, pn = pb ^ BUFSZ;
        const bool pen = t + 2 == nk;
        lds_cptr r0, r1;
#define P_SYNC(k_)                                                               \
    do {                                                                         \
        if (STEADY) { if ((k_) < st_ph) { if (QKLO && prev_lo) wait_vmcnt<VM_AFTER_EPI + 16>(); else wait_vmcnt<VM_AFTER_EPI>(); } else wait_vmcnt<10>(); } \
        else if (pen) wait_vmcnt<8 - 2 * (k_)>();                                \
        else if ((k_) == 0) wait_vmcnt<0>();                                     \
        __builtin_amdgcn_s_waitcnt(0xc07f);                                      \
        __builtin_amdgcn_sched_barrier(0);                                       \
        __builtin_amdgcn_s_setprio(1);                                           \
        __builtin_amdgcn_s_barrier();                                            \
        __builtin_amdgcn_sched_barrier(0);                                       \
    } while (0)
#define P_END()                                                                  \
    do {                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                       \
        __builtin_amdgcn_s_setprio(0);                                           \
        __builtin_amdgcn_s_barrier();                                            \
        __builtin_amdgcn_sched_barrier(0);                                       \
    } while (0)
        if (STEADY) stage(0, kts, nx);                      // A0
        r0 = pin(offB[0] + pb); r1 = pin(offB[1] + pb);
        P_SYNC(0);
        rd_b(fb[1], r0, r1, OFF_B1); mma(0, 0, fa, fb[0]);  // reads B1(t)
        P_INTERLEAVE(4);
        P_END();
        if (STEADY) stage(1, kts, nx);                      // B0
        r0 = pin(offA[0] + pb); r1 = pin(offA[1] + pb);
        P_SYNC(1);
        rd_a(fa1, r0, r1, OFF_A1); mma(0, 1, fa, fb[1]);    // reads A1(t)
        P_INTERLEAVE(8);
        P_END();
        if (STEADY) stage(2, kts, nx);                      // B1
        r0 = pin(offA[0] + pn); r1 = pin(offA[1] + pn);
        P_SYNC(2);
        rd_a(fa, r0, r1, OFF_A0); mma(1, 1, fa1, fb[1]);    // reads A0(t + 1) (= the next tile's A0(0) in the last K tile)
        P_INTERLEAVE(8);
        P_END();
        if (STEADY) stage(3, kts, nx);                      // A1
        r0 = pin(offB[0] + pn); r1 = pin(offB[1] + pn);
        P_SYNC(3);
        rd_b(fb0n, r0, r1, OFF_B0); mma(1, 0, fa1, fb[0]);  // reads B0(t + 1) into the spare set
        P_INTERLEAVE(4);
        P_END();
        __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) fb[0][j][kk] = fb0n[j][kk];
    };
    // ---- epilogue: the lane's 8 consecutive columns of a row -> 2 KB of wave-private LDS (32 rows x 64 B, 16-byte chunks XOR-swizzled) -> row-contiguous
    // 16-byte stores (4 lanes = one 64-byte row segment), as in k_gemm8 but in half passes and with every lane-derived value recomputed from an
    // opaque copy of the thread id, so that nothing of it is hoisted across the K loops of the persistent tile loop (the kernel has no register to spare)
    auto epilogue = [&](const long m0, const int n0, const int bpar) {
        int t_ = tid; asm volatile("" : "+v"(t_));
        const int ln = t_ & 63, r15 = ln & 15, q4 = ln >> 4;
        char* const ep = smem + OFF_SCR + wid * 2048;
        // bias of the lane's 8 columns per B half, from this tile's bias buffer (the other buffer already holds the next tile's row): added here, after
        // the accumulation, in k_gemm8's order - the results of the two kernels are bit-identical
        f32x4 bia[2][2];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                bia[hb][j] = has_bias ? *reinterpret_cast<const f32x4*>(smem + OFF_BIAS + bpar * 1024 + (hb * 128 + wc * 32 + q4 * 8 + j * 4) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        [[maybe_unused]] f32x4 csm[LNC ? 2 : 1][2];
        if constexpr (LNC) {
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    csm[hb][j] = *reinterpret_cast<const f32x4*>(smem + OFF_CS + bpar * 1024 + (hb * 128 + wc * 32 + q4 * 8 + j * 4) * 4);
        }
        const long mw = m0 + wr * 64;
        const long rows = g.M - mw;
        if constexpr (!OUT16) {
            // fp32 residual read-modify-write (out-proj, c_proj): quarter passes of 16 rows x 32 columns (2 KB: a lane's two 16-byte halves of its row,
            // chunks XOR-swizzled by the row) -> 8 lanes per 128-byte row segment; the residual rows are fetched in that coalesced layout one quarter
            // pass ahead (ordinary loads BEHIND the next tile's LDS-DMA in the queue: the compiler's waits for them are positional like ours)
            const int wswz8 = r15 & 7;
            const int crow8 = ln >> 3, cchunk8 = ln & 7;
            const unsigned ldcb = (unsigned)g.ldc * 4u;
            const __amdgpu_buffer_rsrc_t rC = gemm_rsrc(reinterpret_cast<const char*>(g.C) + (mw * g.ldc + n0 + wc * 32) * 4,
                                                        rows > 0 ? (rows - 1) * (long)ldcb + (long)(g.N - n0 - wc * 32) * 4 : 0);
            const unsigned voff = (unsigned)crow8 * ldcb + (unsigned)(cchunk8 * 16);
            auto soff_of = [&](const int q, const int it) {      // quarter pass q = pass * 4 + i: rows ha * 128 + i * 16 + it * 8 .., columns hb * 128 ..
                const int pass = q >> 2, i = q & 3;
                return (unsigned)((pass & 1) * 128 + i * 16 + it * 8) * ldcb + (unsigned)((pass >> 1) * 128 * 4);
            };
            f32x4 res[2][2];
            auto issue = [&](const int q) {
#pragma unroll
                for (int it = 0; it < 2; ++it) res[q & 1][it] = buf_load4<2>(rC, voff, soff_of(q, it));
            };
            issue(0);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int pass = q >> 2, i = q & 3, hb = pass >> 1, ha = pass & 1;
                f32x4 w0, w1;
#pragma unroll
                for (int e = 0; e < 4; ++e) { w0[e] = acc[ha][i][hb][0][e] + bia[hb][0][e]; w1[e] = acc[ha][i][hb][1][e] + bia[hb][1][e]; }
                __builtin_amdgcn_sched_barrier(0);
                *reinterpret_cast<f32x4*>(ep + r15 * 128 + (((2 * q4) ^ wswz8) << 4)) = w0;
                *reinterpret_cast<f32x4*>(ep + r15 * 128 + (((2 * q4 + 1) ^ wswz8) << 4)) = w1;
                __builtin_amdgcn_sched_barrier(0);
                if (q + 1 < 16) issue(q + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int row = it * 8 + crow8;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(ep + row * 128 + ((cchunk8 ^ (row & 7)) << 4));
                    const f32x4 o = res[q & 1][it];
                    buf_store4<2>(rC, voff, soff_of(q, it), f32x4{o[0] + v[0], o[1] + v[1], o[2] + v[2], o[3] + v[3]});
                }
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
        const int wswz = (r15 >> 1) & 3;
        const int crow = ln >> 2, cchunk = ln & 3;
        const unsigned ldcb = (unsigned)g.ldc * 2u;
        const __amdgpu_buffer_rsrc_t rC = gemm_rsrc(reinterpret_cast<const char*>(g.C) + (mw * g.ldc + n0 + wc * 32) * 2,
                                                    rows > 0 ? (rows - 1) * (long)ldcb + (long)(g.N - n0 - wc * 32) * 2 : 0);
        const unsigned voff = (unsigned)crow * ldcb + (unsigned)(cchunk * 16);
        [[maybe_unused]] const bool tile_lo = QKLO && n0 < g.lo_cols;
        [[maybe_unused]] const unsigned ldlob = (unsigned)g.ld_lo * 2u;
        [[maybe_unused]] const __amdgpu_buffer_rsrc_t rLo = gemm_rsrc(reinterpret_cast<const char*>(g.lo_out) + (mw * g.ld_lo + n0 + wc * 32) * 2,
                                                                       (QKLO && tile_lo && rows > 0) ? (rows - 1) * (long)ldlob + (long)(g.lo_cols - n0 - wc * 32) * 2 : 0);
        [[maybe_unused]] const unsigned volo = (unsigned)crow * ldlob + (unsigned)(cchunk * 16);
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int hb = pass >> 1, ha = pass & 1;
            // LNC: the (rstd, -mean rstd) pairs of this lane's four rows of the pass, all four LDS reads in flight together (one per use cost
            // ~0.9 us per tile in dependent LDS round trips)
            [[maybe_unused]] float2 rap[LNC ? 4 : 1];
            if constexpr (LNC) {
#pragma unroll
                for (int i = 0; i < 4; ++i) rap[i] = *reinterpret_cast<const float2*>(smem + OFF_RA + bpar * 2048 + (ha * 128 + wr * 64 + i * 16 + r15) * 8);
            }
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                f32x4 w[2];
                [[maybe_unused]] f32x4 wl[QKLO ? 2 : 1];
#pragma unroll
                for (int il = 0; il < 2; ++il) {
                    const int i = hf * 2 + il;
                    float v[8];
                    if constexpr (LNC) {
                        const float2 ra = rap[i];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = acc[ha][i][hb][0][e] * ra.x + (ra.y * csm[hb][0][e] + bia[hb][0][e]);
                            v[4 + e] = acc[ha][i][hb][1][e] * ra.x + (ra.y * csm[hb][1][e] + bia[hb][1][e]);
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[e] = acc[ha][i][hb][0][e] + bia[hb][0][e]; v[4 + e] = acc[ha][i][hb][1][e] + bia[hb][1][e]; }
                    }
                    f16x8 h;
                    [[maybe_unused]] f16x8 hl;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if (EPI == EPI_BIAS_GELU_F16) v[e] = v[e] * __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v[e]));
                        h[e] = (f16)v[e];
                        if constexpr (QKLO) hl[e] = (f16)(v[e] - (float)h[e]);
                    }
                    w[il] = __builtin_bit_cast(f32x4, h);
                    if constexpr (QKLO) wl[il] = __builtin_bit_cast(f32x4, hl);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int il = 0; il < 2; ++il) *reinterpret_cast<f32x4*>(ep + (il * 16 + r15) * 64 + ((q4 ^ wswz) << 4)) = w[il];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int row = it * 16 + crow;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(ep + row * 64 + ((cchunk ^ ((row >> 1) & 3)) << 4));
#ifdef SEMABS_TUNING
                    if (g.ablate & 32) { buf_store4<2>(rC, voff, (unsigned)(ha * 128 + hf * 32 + it * 16) * ldcb + (unsigned)(hb * 128 * 2), v); continue; }
                    if (g.ablate & 256) { buf_store4<17>(rC, voff, (unsigned)(ha * 128 + hf * 32 + it * 16) * ldcb + (unsigned)(hb * 128 * 2), v); continue; }     // sc0 sc1
                    if (g.ablate & 512) { buf_store4<18>(rC, voff, (unsigned)(ha * 128 + hf * 32 + it * 16) * ldcb + (unsigned)(hb * 128 * 2), v); continue; }     // nt sc1
                    if (g.ablate & 1024) { buf_store4<1>(rC, voff, (unsigned)(ha * 128 + hf * 32 + it * 16) * ldcb + (unsigned)(hb * 128 * 2), v); continue; }     // sc0
                    if (g.ablate & 2048) { buf_store4<16>(rC, voff, (unsigned)(ha * 128 + hf * 32 + it * 16) * ldcb + (unsigned)(hb * 128 * 2), v); continue; }    // sc1
                    if (g.ablate & 4096) { buf_store4<19>(rC, voff, (unsigned)(ha * 128 + hf * 32 + it * 16) * ldcb + (unsigned)(hb * 128 * 2), v); continue; }    // sc0 nt sc1
#endif
                    buf_store4<0>(rC, voff, (unsigned)(ha * 128 + hf * 32 + it * 16) * ldcb + (unsigned)(hb * 128 * 2), v);
                }
                // (store-data hazard: see store_pad in k_gemm8's epilogue)
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (QKLO) {
                    if (tile_lo) {                           // the same half pass once more for the low halves (the wave's LDS accesses are in order)
#pragma unroll
                        for (int il = 0; il < 2; ++il) *reinterpret_cast<f32x4*>(ep + (il * 16 + r15) * 64 + ((q4 ^ wswz) << 4)) = wl[il];
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int it = 0; it < 2; ++it) {
                            const int row = it * 16 + crow;
                            const f32x4 v = *reinterpret_cast<const f32x4*>(ep + row * 64 + ((cchunk ^ ((row >> 1) & 3)) << 4));
                            buf_store4<0>(rLo, volo, (unsigned)(ha * 128 + hf * 32 + it * 16) * ldlob + (unsigned)(hb * 128 * 2), v);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
    };

    // ---- persistent tile loop ----
    int vb = blockIdx.x;
    long m0; int n0;
    tile_of(vb, m0, n0);
    rA = rsrc_a(m0); rB = rsrc_b(n0);
    bool first = true;
    int bpar = 0;                                           // bias buffer of the current tile
    stage_bias(m0, n0, 0);
    stage(0, 0, false); stage(1, 0, false); stage(2, 0, false); stage(3, 0, false); stage(0, 1, false); stage(1, 1, false); stage(2, 1, false); stage(3, 1, false);
    for (;;) {
        const int nvb = vb + (int)gridDim.x;
        const bool has_next = nvb < g.n_blocks;
        long m0n = 0; int n0n = 0;
        if (has_next) { tile_of(nvb, m0n, n0n); rAn = rsrc_a(m0n); rBn = rsrc_b(n0n); } else { rAn = gemm_rsrc(g.A, 0); rBn = gemm_rsrc(g.B, 0); }
        // stages 0 - 2 of this tile (and its bias row) have landed: 16 (wave 0: 17) prologue DMA instructions, the 10 newest may be in flight
#ifdef SEMABS_TUNING
        unsigned long long t_start = 0, t_main = 0, c_start = 0, c_main = 0, t_first = 0;
        if (g.trace) { t_start = __builtin_amdgcn_s_memrealtime(); c_start = __builtin_amdgcn_s_memtime(); }
#endif
        if (first) wait_vmcnt<10>(); else if (QKLO && prev_lo) wait_vmcnt<VM_AFTER_EPI + 16>(); else wait_vmcnt<VM_AFTER_EPI>();     // (see ktile: behind an epilogue its loads / stores are inside the window)
        __builtin_amdgcn_s_barrier();
#ifdef SEMABS_TUNING
        if (g.trace) t_first = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[a][i][c][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        {   // B0(0), A0(0): complete in EVERY wave before the first wave enters phase 0 (their slots are re-staged in phases 0 / 1)
            lds_cptr r0 = pin(offB[0]), r1 = pin(offB[1]);
            rd_b(fb[0], r0, r1, OFF_B0);
            r0 = pin(offA[0]); r1 = pin(offA[1]);
            rd_a(fa, r0, r1, OFF_A0);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        if (wr == 1) __builtin_amdgcn_s_barrier();          // skew the second wave row by one barrier
        // (the workgroup's LAST tile stages through zero-extent descriptors: the same instruction stream, the loads fetch nothing and write zeros into
        //  slots nobody reads again - one body for every K tile of the kernel instead of a steady-state and a drain variant, which the register
        //  allocator could not hold side by side: 588 spills)
        for (int t = 0; t < nk; ++t) {
            const bool nx = t + 2 >= nk;
            if (has_next && t + 2 == nk) stage_bias(m0n, n0n, bpar ^ 1);   // the next tile's bias row (LNC: + colsum row and row pairs), in front of its first stage
            ktile(std::integral_constant<bool, true>{}, t, nx ? t + 2 - nk : t + 2, nx, first ? 0 : (t == 0 ? 4 : (t == 1 ? 1 : 0)));
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();          // balance the skew barrier
#ifdef SEMABS_TUNING
        if (g.trace) { t_main = __builtin_amdgcn_s_memrealtime(); c_main = __builtin_amdgcn_s_memtime(); }
#endif
        epilogue(m0, n0, bpar);
#ifdef SEMABS_TUNING
        if (g.trace && tid == 0) {                          // {tile start, K loop end, epilogue issued, first wait passed} in 100 MHz ticks; shader-clock stamps
            unsigned long long* tr = g.trace + (size_t)vb * 8;
            tr[0] = t_start; tr[1] = t_main; tr[2] = __builtin_amdgcn_s_memrealtime(); tr[3] = 0; tr[4] = c_start; tr[5] = c_main; tr[6] = t_first;
        }
#endif
        if (!has_next) { wait_vmcnt<0>(); break; }          // nothing may be in flight towards this workgroup's LDS when it ends
        if constexpr (QKLO) prev_lo = n0 < g.lo_cols;
        vb = nvb; m0 = m0n; n0 = n0n; rA = rAn; rB = rBn; first = false; bpar ^= 1;
    }
#undef P_INTERLEAVE
#undef P_SYNC
#undef P_END
}

// Raster group size (row panels walked fastest) by the number of 256-wide column panels.  An XCD runs 32 tiles at a time = group_m row
// panels x 32 / group_m column panels, and an operand panel is only re-used out of its 4 MiB L2 while the panels of the resident tiles
// fit next to the C tiles streaming through (tools/gemm_raster.py + tools/pmc_seq.py, M = 482 256; FETCH x 2 + WRITE over the algorithmic
// bytes / TFLOP/s):  N = 768 (out-proj, K = 768): group 8 1.19 / 677, 2 1.02 / 714;  N = 768 (c_proj, K = 3072): 8 1.55 / 1030, 2 1.40 / 1048;
// N = 2304 (QKV): 8 2.31 / 981, 4 1.76 / 998, 1 2.52 / 976;  N = 3072 (c_fc): 8 1.77 / 918, 4 1.86 / 909, 1 2.68 / 884.  With three column
// panels all of W stays resident and A is read once; with 9 - 12 the 3.5 - 4.7 MB of W cannot, and A is re-read ~2.5 x whatever the raster
// (super-columns that keep a W slice resident re-read A once per slice instead: QKV 1.61 - 1.68 at -3 .. -5 % TFLOP/s, not taken).
static inline int gemm8_group_m(int n_tiles_n) { return n_tiles_n <= 3 ? 2 : (n_tiles_n <= 9 ? 4 : 8); }

// CUs the launch stream can occupy (the device's multiprocessor count, or the popcount of the stream's CU mask): semabs_stream_cus, cached per stream

template <int EPI>
static int launch_gemm8(GemmArgs g, hipStream_t s, const GemmOpts& o) {
#ifdef SEMABS_TUNING
    constexpr int LDS = 2 * 4 * 16384 + 16384;             // + the phase-trace rows (GEMM8_STAMP)
#else
    constexpr int LDS = 2 * 4 * 16384;
#endif
    static SemabsLdsAttr attr_pf, attr_nopf;
    semabs_ensure_lds(&k_gemm8<EPI, true>, LDS, attr_pf);
    semabs_ensure_lds(&k_gemm8<EPI, false>, LDS, attr_nopf);
    g.n_tiles_n = g.N / 256;
    const long mt = (g.M + 255) / 256;
    g.n_tiles_m = (int)mt;
    g.group_m = GEMM_GROUP_M > 0 ? GEMM_GROUP_M : gemm8_group_m(g.n_tiles_n);
    g.sc_w = GEMM_SC_W;
    GEMM_TUNE_ARGS(g);
    if (mt * g.n_tiles_n >= (1L << 30)) { semabs_set_error("semabs_gemm_f16: grid too large"); return SEMABS_EINVAL; }
    g.n_blocks = (int)(mt * g.n_tiles_n);
#ifdef SEMABS_TUNING
    if (g_persist) {
        static SemabsLdsAttr attr_p;
        semabs_ensure_lds(&k_gemm8<EPI, true, true>, LDS, attr_p);
        const int ncu = semabs_stream_cus(s);
        gemm_dispatch(k_gemm8<EPI, true, true>, dim3(g.n_blocks < ncu ? g.n_blocks : ncu), dim3(512), LDS, s, g, o);
        SEMABS_CHECK_LAUNCH();
        return SEMABS_OK;
    }
#endif
#ifdef SEMABS_TUNING
    // compile-time ablations (no run-time branches in the loops they measure): g_ablate picks an instantiation, for the QKV / out-proj epilogues only
    if constexpr (EPI == EPI_BIAS_F16 || EPI == EPI_BIAS_RESID_F32) {
        if (g_ablate & 31) {
            auto go = [&](auto kern) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
                gemm_dispatch(kern, dim3(g.n_blocks), dim3(512), LDS, s, g, o);
            };
#define GEMM8_ABL_CASE(n) case n: if (o.v2) go(k_gemm8<EPI, true, false, false, true, n>); else go(k_gemm8<EPI, true, false, false, false, n>); break;
            switch (g_ablate & 31) {
                GEMM8_ABL_CASE(8) GEMM8_ABL_CASE(9) GEMM8_ABL_CASE(10) GEMM8_ABL_CASE(11) GEMM8_ABL_CASE(12) GEMM8_ABL_CASE(13) GEMM8_ABL_CASE(15)
                default: semabs_set_error("semabs_gemm_tune: this ablation mask is not instantiated"); return SEMABS_EINVAL;
            }
#undef GEMM8_ABL_CASE
            SEMABS_CHECK_LAUNCH();
            return SEMABS_OK;
        }
    }
#endif
    if constexpr (EPI == EPI_BIAS_RESID_F32) {
        if (g.ln_xg) {                                       // LayerNorm producer (deep schedule): + 16 KB of LDS behind the operand buffers for the row partials
            static SemabsLdsAttr attr_lp;
            semabs_ensure_lds(&k_gemm8<EPI, true, false, false, true, 0, true>, LDS + 16384, attr_lp);
            gemm_dispatch(k_gemm8<EPI, true, false, false, true, 0, true>, dim3(g.n_blocks), dim3(512), LDS + 16384, s, g, o);
            SEMABS_CHECK_LAUNCH();
            return SEMABS_OK;
        }
    }
    if constexpr (EPI == EPI_BIAS_F16 || EPI == EPI_BIAS_GELU_F16) {
        if (g.ln_rowac || g.lo_out) {                        // LayerNorm consumer and / or low halves (persistent workgroups): + 2 x 1 KB colsum rows + 2 x 2 KB row pairs
            if ((g.K / 64) % 2 != 0) { semabs_set_error("semabs_gemm_f16_ln: the persistent kernel needs an even number of 64-wide K tiles"); return SEMABS_EINVAL; }
            constexpr int LDSC = 2 * 4 * 16384 + 8 * 2048 + 2 * 1024 + 2 * 1024 + 2 * 2048;
            const int ncu = semabs_stream_cus(s);
            const dim3 grid(g.n_blocks < ncu ? g.n_blocks : ncu);
            if constexpr (EPI == EPI_BIAS_F16) {
                if (g.lo_out && g.ln_rowac) {
                    static SemabsLdsAttr a1; semabs_ensure_lds(&k_gemm8p<EPI, true, true>, LDSC, a1);
                    gemm_dispatch(k_gemm8p<EPI, true, true>, grid, dim3(512), LDSC, s, g, o);
                    SEMABS_CHECK_LAUNCH();
                    return SEMABS_OK;
                }
                if (g.lo_out) {
                    static SemabsLdsAttr a2; semabs_ensure_lds(&k_gemm8p<EPI, false, true>, LDSC, a2);
                    gemm_dispatch(k_gemm8p<EPI, false, true>, grid, dim3(512), LDSC, s, g, o);
                    SEMABS_CHECK_LAUNCH();
                    return SEMABS_OK;
                }
            }
            if (g.lo_out) { semabs_set_error("semabs_gemm_f16_ln: low halves exist for epi 0 only"); return SEMABS_EINVAL; }
            static SemabsLdsAttr attr_lc;
            semabs_ensure_lds(&k_gemm8p<EPI, true>, LDSC, attr_lc);
            gemm_dispatch(k_gemm8p<EPI, true>, grid, dim3(512), LDSC, s, g, o);
            SEMABS_CHECK_LAUNCH();
            return SEMABS_OK;
        }
    }
    if (o.ring) {                                           // kernel | 512: the K = 32 ring schedule (A/B relic, one workgroup per tile); overrides the schedule bits
        static SemabsLdsAttr attr_r;
        semabs_ensure_lds(&k_gemm8<EPI, false, false, true>, LDS, attr_r);
        gemm_dispatch(k_gemm8<EPI, false, false, true>, dim3(g.n_blocks), dim3(512), LDS, s, g, o);
        SEMABS_CHECK_LAUNCH();
        return SEMABS_OK;
    }
    if constexpr (EPI == EPI_BIAS_F16 || EPI == EPI_BIAS_GELU_F16) {
        // fp16-output epilogues: persistent workgroups (one per CU), the next tile's prologue inside the current tile's drain - k_gemm8p.  Needs an even
        // number of K tiles and no super-columns; kernel | 4096 selects one workgroup per tile (A/B).  The kernel also carries the fp32 residual
        // read-modify-write epilogue (quarter passes of 2 KB): bit-identical, but SLOWER than one workgroup per tile on the same box - out-proj 796 vs 767 us,
        // c_proj 1 980 vs 1 925 us (16 dependent load - transpose - store rounds per tile instead of 4) - so it is not dispatched (not instantiated).
        if (o.v2 && !o.nopers && (g.K / 64) % 2 == 0 && g.sc_w == 0) {
            constexpr int LDSP = 2 * 4 * 16384 + 8 * 2048 + 2 * 1024;
            static SemabsLdsAttr attr_p4;
            semabs_ensure_lds(&k_gemm8p<EPI>, LDSP, attr_p4);
            const int ncu = semabs_stream_cus(s);
            gemm_dispatch(k_gemm8p<EPI>, dim3(g.n_blocks < ncu ? g.n_blocks : ncu), dim3(512), LDSP, s, g, o);
            SEMABS_CHECK_LAUNCH();
            return SEMABS_OK;
        }
    }
    if (o.v2) {
        static SemabsLdsAttr attr_v;
        semabs_ensure_lds(&k_gemm8<EPI, true, false, false, true>, LDS, attr_v);
        gemm_dispatch(k_gemm8<EPI, true, false, false, true>, dim3(g.n_blocks), dim3(512), LDS, s, g, o);
        SEMABS_CHECK_LAUNCH();
        return SEMABS_OK;
    }
    if (GEMM_PREFETCH) gemm_dispatch(k_gemm8<EPI, true>, dim3(g.n_blocks), dim3(512), LDS, s, g, o);
    else gemm_dispatch(k_gemm8<EPI, false>, dim3(g.n_blocks), dim3(512), LDS, s, g, o);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

template <int EPI>
static int launch(const GemmArgs& g, hipStream_t s, const GemmOpts& o) {
    // what the phased kernel can run (its buffer addressing keeps per-tile byte offsets in 32 bits)
    const bool big_ok = g.N % 256 == 0 && g.K >= 128 && g.lda < (1L << 20) && g.ldb < (1 << 20) && g.ldc < (1L << 20);
#ifdef SEMABS_TUNING
    if (g_force_cfg == 3 && g.N % 256 == 0) return launch_cfg<EPI, 256, 256, 32, 4, 4, 4>(g, s, o);      // 16 waves, 64 x 64 wave tiles
    if (g_force_cfg == 4 && g.N % 256 == 0) return launch_cfg<EPI, 256, 128, 32, 4, 2, 3>(g, s, o);      // 8 waves, 64 x 64 wave tiles, 2 blocks / CU
    if (g_force_cfg == 5 && g.N % 256 == 0) return launch_cfg<EPI, 128, 256, 32, 2, 4, 3>(g, s, o);
    if (g_force_cfg == 6) return launch_cfg<EPI, 128, 128, 32, 2, 2, 3>(g, s, o);                         // 4 waves, 48 KB: 3 blocks / CU
    if (g_force_cfg == 7 && g.N % 256 == 0) return launch_cfg<EPI, 256, 128, 32, 2, 2, 3>(g, s, o);      // 4 waves, 128 x 64 wave tiles, 72 KB: 2 blocks / CU
    if (g_force_cfg == 9 && g.N % 256 == 0) return launch_cfg<EPI, 256, 256, 32, 2, 4, 4>(g, s, o);
#endif
    const bool big = o.kernel == 2 ? big_ok : (o.kernel == 1 ? false : (big_ok && g.M >= 2048));
    if (big) return launch_gemm8<EPI>(g, s, o);                                              // 256 x 256 x 64 phased kernel
    if (o.kernel == 0 && g.N % 256 == 0 && g.M >= 2048) return launch_cfg<EPI, 128, 256, 32, 2, 4, 3>(g, s, o);      // K = 64: one K tile, nothing to pipeline
    return launch_cfg<EPI, 128, 128, 64, 2, 2, 2>(g, s, o);
}

// C ABI.  A fp16 [M, K] (row stride lda elements), B fp16 [N, K] (row stride ldb), C per `epi`:
//   0 fp16 = acc + bias        1 fp16 = quickgelu(acc + bias)      2 fp32 += acc + bias (in place)
//   3 fp32 = acc + bias        4 fp32 row-remapped store + addend  (rowmap = {g_in, g_out, g_off})
//   5 fp16 = (acc + bias) * addend[m % rowmap[0], :]   (addend = fp32 table [rowmap[0], N], e.g. semabs_quickgelu_grad's; phased kernel only)
// bias fp32 [N] or NULL.  Requires N % 128 == 0, K % 64 == 0, 16-byte aligned rows.
extern "C" int semabs_gemm_f16_ex(const void* A, const void* B, void* C, const float* bias, const float* addend,
                                  long M, int N, int K, long lda, int ldb, long ldc, int epi, const int* rowmap3,
                                  int kernel, void* start_event, void* stop_event, void* stream) {
    SEMABS_REQUIRE(A && B && C, "semabs_gemm_f16: null operand");
    SEMABS_REQUIRE(M > 0 && N > 0 && K > 0, "semabs_gemm_f16: empty problem");
    SEMABS_REQUIRE(N % 128 == 0 && K % BK == 0, "semabs_gemm_f16: N must be a multiple of 128 and K of 64");
    SEMABS_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0 && ((epi > 1 && epi != 5) || ldc % 8 == 0), "semabs_gemm_f16: leading dimensions must keep 16-byte alignment");
    SEMABS_REQUIRE((kernel & 255) >= 0 && (kernel & 255) <= 2 && (kernel >> 13) == 0, "semabs_gemm_f16_ex: kernel must be 0 (heuristic), 1 (ring kernel k_gemm_f16) or 2 (phased kernel k_gemm8 / k_gemm8p: deep schedule, persistent workgroups for the fp16 outputs), optionally | 256 (reversed tile order) | 512 (phased kernel: K = 32 ring schedule, one workgroup per tile) | 2048 (phased kernel: round-3 PF schedule, one workgroup per tile) | 4096 (phased kernel: one workgroup per tile also for the fp16 outputs, i.e. NO persistent workgroups)");
    SEMABS_REQUIRE((start_event == nullptr) == (stop_event == nullptr), "semabs_gemm_f16_ex: start and stop events go together");
    GemmArgs g;
    g.A = (const f16*)A; g.B = (const f16*)B; g.C = C; g.bias = bias; g.addend = addend;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.g_in = 1; g.g_out = 1; g.g_off = 0;
    if (epi == EPI_ROWMAP_ADD_F32 || epi == EPI_MULROW_F16) {
        SEMABS_REQUIRE(rowmap3 && rowmap3[0] > 0, "semabs_gemm_f16: epi 4 / 5 need rowmap");
        g.g_in = rowmap3[0]; g.g_out = rowmap3[1]; g.g_off = rowmap3[2];
    }
    g.n_tiles_n = 0; g.n_blocks = 0; g.sc_w = 0; g.reverse = (kernel >> 8) & 1;
    g.ln_xg = nullptr; g.ln_gamma = nullptr; g.ln_part = nullptr; g.ln_rowac = nullptr; g.ln_colsum = nullptr; g.ln_center = nullptr;
    g.lo_out = nullptr; g.lo_cols = 0; g.ld_lo = 0;
    const int ring = (kernel >> 9) & 1;                     // bit 9: the K = 32 ring schedule of the phased kernel (A/B)
    const int v2 = ((kernel >> 11) & 1) == 0;               // the deep schedule of the phased kernel is the default since round 4; bit 11 selects the round-3 PF schedule (A/B), bit 10 is accepted and ignored
    const int nopers = (kernel >> 12) & 1;                  // bit 12: one workgroup per tile also for the fp16-output epilogues (A/B against k_gemm8p)
    kernel &= 255;
    GemmOpts o{kernel, (hipEvent_t)start_event, (hipEvent_t)stop_event, ring, v2, nopers};
    hipStream_t s = (hipStream_t)stream;
    switch (epi) {
        case EPI_BIAS_F16: return launch<EPI_BIAS_F16>(g, s, o);
        case EPI_BIAS_GELU_F16: return launch<EPI_BIAS_GELU_F16>(g, s, o);
        case EPI_BIAS_RESID_F32: return launch<EPI_BIAS_RESID_F32>(g, s, o);
        case EPI_BIAS_F32: return launch<EPI_BIAS_F32>(g, s, o);
        case EPI_ROWMAP_ADD_F32: return launch<EPI_ROWMAP_ADD_F32>(g, s, o);
        case EPI_MULROW_F16: {
            const bool big_ok = g.N % 256 == 0 && g.K >= 128 && g.lda < (1L << 20) && g.ldb < (1 << 20) && g.ldc < (1L << 20);
            SEMABS_REQUIRE(big_ok && g.M >= 2048 && g.M < (1L << 31) && addend && g.g_in > 0, "semabs_gemm_f16: epi 5 (row-table multiply) needs the phased kernel's shapes (M >= 2048, N % 256 == 0, K >= 128), the table in addend and its row count in rowmap[0]");
            return launch_gemm8<EPI_MULROW_F16>(g, s, o);
        }
    }
    semabs_set_error("semabs_gemm_f16: unknown epilogue");
    return SEMABS_EINVAL;
}

// LayerNorm folded into the GEMMs on either side of it (large shapes only: M >= 2048, N % 256 == 0, K >= 128; k_gemm8's LNP / k_gemm8p's LNC).
//   epi 2 (x += A W^T + b, fp32) with ln_xg / ln_gamma / ln_part: also xg = fp16(x_new * gamma) [M, N] and the row partials [M, N / 256, 2];
//   epi 0 / 1 (fp16 outputs) with ln_rowac / ln_colsum: C = rstd_row * (A W^T) - mean_row rstd_row * colsum + bias, A being such an xg (K % 128 == 0).
//   epi 0 with lo_out (with or without the consumer operands): also lo_out[m, n] = fp16(v - fp16(v)) for the columns n < lo_cols (lo_cols % 256 == 0; row
//   pitch ld_lo elements) - the low halves of q | k for precision = "parity" (semabs_attention_split).
// reverse: walk the tiles backwards per XCD (the zigzag schedule of the trunk, like kernel | 256 of semabs_gemm_f16_ex).
extern "C" int semabs_gemm_f16_ln(const void* A, const void* B, void* C, const float* bias, long M, int N, int K, long lda, int ldb, long ldc, int epi,
                                  void* ln_xg, const float* ln_gamma, float* ln_part, const float* ln_center, const float* ln_rowac, const float* ln_colsum,
                                  void* lo_out, int lo_cols, long ld_lo, int reverse,
                                  void* start_event, void* stop_event, void* stream) {
    SEMABS_REQUIRE(A && B && C, "semabs_gemm_f16_ln: null operand");
    SEMABS_REQUIRE(M >= 2048 && N % 256 == 0 && K >= 128 && K % BK == 0, "semabs_gemm_f16_ln: needs M >= 2048, N % 256 == 0, K >= 128 (the phased kernel)");
    SEMABS_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0 && (epi > 1 || ldc % 8 == 0) && lda < (1L << 20) && ldb < (1 << 20) && ldc < (1L << 20),
                   "semabs_gemm_f16_ln: leading dimensions");
    const bool producer = epi == EPI_BIAS_RESID_F32 && ln_xg && ln_gamma && ln_part && !ln_rowac && !ln_colsum;
    const bool consumer = (epi == EPI_BIAS_F16 || epi == EPI_BIAS_GELU_F16) && ln_rowac && ln_colsum && !ln_xg && !ln_gamma && !ln_part;
    const bool plain_lo = epi == EPI_BIAS_F16 && lo_out && !ln_rowac && !ln_colsum && !ln_xg && !ln_gamma && !ln_part;
    SEMABS_REQUIRE(producer || consumer || plain_lo, "semabs_gemm_f16_ln: epi 2 with (xg, gamma, partials), epi 0 / 1 with (rowac, colsum), or epi 0 with lo_out");
    SEMABS_REQUIRE(!(consumer || plain_lo) || K % 128 == 0, "semabs_gemm_f16_ln: the persistent kernel needs K % 128 == 0");
    SEMABS_REQUIRE(!lo_out || (epi == EPI_BIAS_F16 && lo_cols > 0 && lo_cols % 256 == 0 && lo_cols <= N && ld_lo % 8 == 0 && ld_lo >= lo_cols && ld_lo < (1L << 20)),
                   "semabs_gemm_f16_ln: lo_out needs epi 0, lo_cols a multiple of 256 within N and a 16-byte aligned row pitch >= lo_cols");
    SEMABS_REQUIRE((start_event == nullptr) == (stop_event == nullptr), "semabs_gemm_f16_ln: start and stop events go together");
    GemmArgs g;
    g.A = (const f16*)A; g.B = (const f16*)B; g.C = C; g.bias = bias; g.addend = nullptr;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.g_in = 1; g.g_out = 1; g.g_off = 0;
    g.n_tiles_n = 0; g.n_blocks = 0; g.sc_w = 0; g.reverse = reverse ? 1 : 0;
    g.ln_xg = (f16*)ln_xg; g.ln_gamma = ln_gamma; g.ln_part = ln_part; g.ln_rowac = ln_rowac; g.ln_colsum = ln_colsum;
    g.ln_center = producer ? ln_center : nullptr;
    g.lo_out = (f16*)lo_out; g.lo_cols = lo_out ? lo_cols : 0; g.ld_lo = ld_lo;
    GemmOpts o{2, (hipEvent_t)start_event, (hipEvent_t)stop_event, 0, 1, 0};
    hipStream_t s = (hipStream_t)stream;
    switch (epi) {
        case EPI_BIAS_F16: return launch_gemm8<EPI_BIAS_F16>(g, s, o);
        case EPI_BIAS_GELU_F16: return launch_gemm8<EPI_BIAS_GELU_F16>(g, s, o);
        default: return launch_gemm8<EPI_BIAS_RESID_F32>(g, s, o);
    }
}

extern "C" int semabs_gemm_f16(const void* A, const void* B, void* C, const float* bias, const float* addend,
                               long M, int N, int K, long lda, int ldb, long ldc, int epi, const int* rowmap3,
                               void* stream) {
    return semabs_gemm_f16_ex(A, B, C, bias, addend, M, N, K, lda, ldb, ldc, epi, rowmap3, 0, nullptr, nullptr, stream);
}
