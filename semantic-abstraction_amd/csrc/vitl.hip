// Multi-layer attention x gradient rollout for deep CLIP towers (ViT-L/14: 24 blocks, 13 of which enter the rollout) - SURVEY.md 8 f4.
//
// Reference: ClipGradcam.interpret (CLIP/clip/clip_gradcam.py:70-132).  For every block i > num_layers (10) it takes
//     grad_i = d logit_l / d attn_probs_i   (torch.autograd.grad, one backward pass per label),
//     cam_i  = mean_h clamp(grad_i * attn_probs_i, min = 0)      (clamp iff positive_attn_only),        R <- R + cam_i R,
// and returns R[:, :, 0, 1:].  Only row 0 of the final R is used and R_final = (I + cam_23) ... (I + cam_11), so a ROW VECTOR is enough:
//     r <- e_0;   for i = 23 ... 11:   r <- r + r cam_i.
// grad_i[h, q, k] = dO_i[q, h, :] . V_i[k, h, :] with dO_i the gradient wrt the attention output of block i (before out_proj), so no T x T
// matrix is ever materialised: P is rebuilt from Q, K and the forward's row statistics, dP = dO V^T is formed on the fly, and both the
// rollout update  c[k] += r[q] act(P dP) / H  and the attention backward (dQ, dK, dV) that carries the gradient down to the previous
// block are accumulated block by block.  ViT-B never gets here: only its last block enters the rollout (closed form in vit.hip).
//
// Round 2: both kernels run on the matrix pipe (v_mfma_f32_32x32x16_f16, one workgroup per (sequence, head), one wave per 32-row block,
// T <= 288, head_dim 64).  An MFMA result D = A B has its ROWS in registers and its COLUMNS in lanes, and can be fed back only as a B
// operand, i.e. contracted over its rows.  Hence two orientations:
//   k_attn_bwd_dq   wave = query block.  S^T = K Q^T, dP^T = V dO^T (rows = keys, lanes = queries: the per-query softmax statistics and
//                   delta = dO . O are one value per lane), dS^T = P^T (dP^T - delta), dQ^T += K^T dS^T (contracts over keys).
//   k_attn_bwd_dkv  wave = key block.   S = Q K^T, dP = dO V^T (rows = queries, lanes = keys), the rollout update is an in-lane sum over
//                   the rows, dV^T += dO^T P and dK^T += Q^T dS (contract over queries).
// delta_q = sum_k P dP = dO_q . O_q needs the attention OUTPUT of the forward pass, and P = exp(S - m) * inv its row statistics
// (semabs_attention's row_stats); dS goes through fp16 as an MFMA operand scaled by 2^4 (fp32 accumulators, un-scaled at the store).
// The first, vector-ALU versions of these kernels (v_dot2, thread = row) took 78 % of the ViT-L/14 step.
#include "semabs_common.h"

__device__ __forceinline__ int kswz_l(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
#define DS_SCALE 16.f

// qkv fp16 [n, T, 3D] (q pre-scaled | k | v) of the block; att fp16 [n, T, D] its attention output; fstats fp32 [n, H, T, 2] the forward's
// (reference maximum, 1 / sum); dO fp16 [R, T, D] with R = L * n sequences ordered (label, tile); rvec fp32 [R, T] (the rollout row
// BEFORE this block).  Writes stats fp32 [R, H, T, 4] = (maximum, 1 / sum, delta, rvec) for k_attn_bwd_dkv and, when GRAD, the dQ third of
// dqkv fp16 [R, T, 3D].
template <int NKB, bool GRAD>
__global__ __launch_bounds__(64 * NKB) void k_attn_bwd_dq(const f16* __restrict__ qkv, const f16* __restrict__ att, const float* __restrict__ fstats,
                                                          const f16* __restrict__ dO, const float* __restrict__ rvec, float* __restrict__ stats,
                                                          f16* __restrict__ dqkv, int n, int T, int H) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TP = 32 * NKB, VS = TP + 4, NTHR = 64 * NKB;
    char* sK = smem;                                        // [TP][64] fp16, 16-byte chunks swizzled
    char* sV = smem + TP * 128;
    f16* sKT = reinterpret_cast<f16*>(smem + TP * 256);     // [64][VS]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int D = H * 64, ld = 3 * D;
    const int r = blockIdx.x / H, h = blockIdx.x % H;
    const int tile = r % n;
    const f16* base = qkv + (size_t)tile * T * ld + h * 64;
    const int ql = lane & 31, hi = lane >> 5;
    const int q = wid * 32 + ql;
    const bool live = q < T;
    const int qc = live ? q : T - 1;
    f16x8 fq[4], fo[4];
    float delta = 0.f;
    {
        const f16* ob = dO + ((size_t)r * T + qc) * D + h * 64;
        const f16* ab = att + ((size_t)tile * T + qc) * D + h * 64;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            fq[ks] = *reinterpret_cast<const f16x8*>(base + (size_t)qc * ld + (ks * 2 + hi) * 8);
            fo[ks] = *reinterpret_cast<const f16x8*>(ob + (ks * 2 + hi) * 8);
            const f16x8 fa = *reinterpret_cast<const f16x8*>(ab + (ks * 2 + hi) * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) delta = fmaf((float)fo[ks][e], (float)fa[e], delta);
        }
    }
    delta += __shfl_xor(delta, 32, 64);
    const float* fs = fstats + (((size_t)tile * H + h) * T + qc) * 2;
    const float mref = fs[0];
    const float inv = live ? fs[1] : 0.f;
    if (live && hi == 0) *reinterpret_cast<float4*>(stats + (((size_t)r * H + h) * T + q) * 4) = make_float4(mref, inv, delta, rvec[(size_t)r * T + q]);
    if (!GRAD) return;
    constexpr int NIT = (TP * 8 + NTHR - 1) / NTHR;
    {
        f16x8 kvs[NIT], vvs[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = tid + it * NTHR, row = c >> 3, kc = c & 7;
            if (c < TP * 8 && row < T) {
                kvs[it] = *reinterpret_cast<const f16x8*>(base + (size_t)row * ld + D + kc * 8);
                vvs[it] = *reinterpret_cast<const f16x8*>(base + (size_t)row * ld + 2 * D + kc * 8);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { kvs[it][e] = (f16)0.f; vvs[it][e] = (f16)0.f; }
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = tid + it * NTHR, row = c >> 3, kc = c & 7;
            if (c < TP * 8) {
                *reinterpret_cast<f16x8*>(sK + kswz_l(row, kc)) = kvs[it];
                *reinterpret_cast<f16x8*>(sV + kswz_l(row, kc)) = vvs[it];
#pragma unroll
                for (int e = 0; e < 8; ++e) sKT[(kc * 8 + e) * VS + row] = kvs[it][e];
            }
        }
    }
    __syncthreads();
    int koff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = kswz_l(ql, ks * 2 + hi);
    f32x16 o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[db][e] = 0.f;
#pragma unroll 1
    for (int kb = 0; kb < NKB; ++kb) {
        f32x16 sc, dp;
#pragma unroll
        for (int e = 0; e < 16; ++e) { sc[e] = -mref; dp[e] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const f16x8 fk = *reinterpret_cast<const f16x8*>(sK + kb * 4096 + koff[ks]);
            sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fk, fq[ks], sc, 0, 0, 0);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const f16x8 fv = *reinterpret_cast<const f16x8*>(sV + kb * 4096 + koff[ks]);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_f16(fv, fo[ks], dp, 0, 0, 0);
        }
        // sc[e] = S[key, q] - m_q, dp[e] = dP[key, q] with key = kb*32 + (e&3) + 8*(e>>2) + 4*hi
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int key = kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
            const float p = key < T ? __builtin_amdgcn_exp2f(sc[e] * 1.44269504088896340736f) * inv : 0.f;
            sc[e] = p * (dp[e] - delta) * DS_SCALE;                              // dS^T (scaled)
        }
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            f16x8 pk;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) pk[jj] = (f16)sc[hf * 8 + jj];
            const int kbase = kb * 32 + hf * 16 + 4 * hi;
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const f16* kr = sKT + (db * 32 + ql) * VS + kbase;
                const f16x4 v0 = *reinterpret_cast<const f16x4*>(kr);
                const f16x4 v1 = *reinterpret_cast<const f16x4*>(kr + 8);
                f16x8 fa;
                fa[0] = v0[0]; fa[1] = v0[1]; fa[2] = v0[2]; fa[3] = v0[3];
                fa[4] = v1[0]; fa[5] = v1[1]; fa[6] = v1[2]; fa[7] = v1[3];
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, pk, o[db], 0, 0, 0);
            }
        }
    }
    // o[db][e] = dQ^T[d = db*32 + (e&3) + 8*(e>>2) + 4*hi][q] * DS_SCALE
    if (live) {
        f16* orow = dqkv + ((size_t)r * T + q) * ld + h * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                f16x4 hv;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) hv[jj] = (f16)(o[db][rq * 4 + jj] * (1.f / DS_SCALE));
                *reinterpret_cast<f16x4*>(orow + db * 32 + rq * 8 + 4 * hi) = hv;
            }
    }
}

// stats fp32 [R, H, T, 4] from k_attn_bwd_dq; gscale fp32 [R] (true gradient = stored gradient * gscale);
// c fp32 [R, T] += (1 / H) sum_q rvec[q] act(P[q, k] dP[q, k] gscale)   (atomic over heads); GRAD: the dK and dV thirds of dqkv.
template <int NKB, bool GRAD>
__global__ __launch_bounds__(64 * NKB) void k_attn_bwd_dkv(const f16* __restrict__ qkv, const f16* __restrict__ dO, const float* __restrict__ stats,
                                                           const float* __restrict__ gscale, float* __restrict__ c, f16* __restrict__ dqkv,
                                                           int n, int T, int H, int positive_only) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TP = 32 * NKB, VS = TP + 4, NTHR = 64 * NKB;
    char* sQ = smem;                                        // [TP][64] fp16 swizzled
    char* sO = smem + TP * 128;                             // dO, same layout
    float* sS = reinterpret_cast<float*>(smem + TP * 256);  // [TP][4]
    f16* sQT = reinterpret_cast<f16*>(smem + TP * 256 + TP * 16);   // [64][VS]   (GRAD only)
    f16* sOT = sQT + 64 * VS;                               // [64][VS]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int D = H * 64, ld = 3 * D;
    const int r = blockIdx.x / H, h = blockIdx.x % H;
    const int tile = r % n;
    const f16* base = qkv + (size_t)tile * T * ld + h * 64;
    const f16* obase = dO + (size_t)r * T * D + h * 64;
    constexpr int NIT = (TP * 8 + NTHR - 1) / NTHR;
    {
        f16x8 qs[NIT], os[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int cidx = tid + it * NTHR, row = cidx >> 3, kc = cidx & 7;
            if (cidx < TP * 8 && row < T) {
                qs[it] = *reinterpret_cast<const f16x8*>(base + (size_t)row * ld + kc * 8);
                os[it] = *reinterpret_cast<const f16x8*>(obase + (size_t)row * D + kc * 8);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { qs[it][e] = (f16)0.f; os[it][e] = (f16)0.f; }
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int cidx = tid + it * NTHR, row = cidx >> 3, kc = cidx & 7;
            if (cidx < TP * 8) {
                *reinterpret_cast<f16x8*>(sQ + kswz_l(row, kc)) = qs[it];
                *reinterpret_cast<f16x8*>(sO + kswz_l(row, kc)) = os[it];
                if (GRAD) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { sQT[(kc * 8 + e) * VS + row] = qs[it][e]; sOT[(kc * 8 + e) * VS + row] = os[it][e]; }
                }
            }
        }
        for (int i = tid; i < TP; i += NTHR)                 // dead query rows: 1 / sum = 0 -> P = 0
            *reinterpret_cast<float4*>(sS + i * 4) = i < T ? *reinterpret_cast<const float4*>(stats + (((size_t)r * H + h) * T + i) * 4)
                                                           : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const int ql = lane & 31, hi = lane >> 5;
    const int k = wid * 32 + ql;
    const bool livek = k < T;
    const int kc2 = livek ? k : T - 1;
    f16x8 fk[4], fv[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        fk[ks] = *reinterpret_cast<const f16x8*>(base + (size_t)kc2 * ld + D + (ks * 2 + hi) * 8);
        fv[ks] = *reinterpret_cast<const f16x8*>(base + (size_t)kc2 * ld + 2 * D + (ks * 2 + hi) * 8);
    }
    int koff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = kswz_l(ql, ks * 2 + hi);
    f32x16 ok[2], ov[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) { ok[db][e] = 0.f; ov[db][e] = 0.f; }
    float csum = 0.f;
#pragma unroll 1
    for (int qb = 0; qb < NKB; ++qb) {
        f32x16 sc, dp;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int qq = qb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
            sc[e] = -sS[qq * 4];                                              // accumulator starts at - m_q: the subtraction is free
            dp[e] = 0.f;
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const f16x8 fa = *reinterpret_cast<const f16x8*>(sQ + qb * 4096 + koff[ks]);
            sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fk[ks], sc, 0, 0, 0);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const f16x8 fa = *reinterpret_cast<const f16x8*>(sO + qb * 4096 + koff[ks]);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fv[ks], dp, 0, 0, 0);
        }
        // sc[e] = S[qq, k] - m_qq, dp[e] = dP[qq, k]; afterwards sc = P, dp = dS * DS_SCALE
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int qq = qb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
            const float4 st = *reinterpret_cast<const float4*>(sS + qq * 4);  // (m, 1 / sum, delta, r_q)
            const float p = livek ? __builtin_amdgcn_exp2f(sc[e] * 1.44269504088896340736f) * st.y : 0.f;
            const float t = p * dp[e];
            csum = fmaf(st.w, positive_only ? fmaxf(t, 0.f) : t, csum);
            sc[e] = p;
            dp[e] = p * (dp[e] - st.z) * DS_SCALE;
        }
        if (GRAD) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                f16x8 pp, dd;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) { pp[jj] = (f16)sc[hf * 8 + jj]; dd[jj] = (f16)dp[hf * 8 + jj]; }
                const int qbase = qb * 32 + hf * 16 + 4 * hi;
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const f16* orow = sOT + (db * 32 + ql) * VS + qbase;
                    const f16* qrow = sQT + (db * 32 + ql) * VS + qbase;
                    const f16x4 a0 = *reinterpret_cast<const f16x4*>(orow), a1 = *reinterpret_cast<const f16x4*>(orow + 8);
                    const f16x4 b0 = *reinterpret_cast<const f16x4*>(qrow), b1 = *reinterpret_cast<const f16x4*>(qrow + 8);
                    f16x8 fa, fb;
                    fa[0] = a0[0]; fa[1] = a0[1]; fa[2] = a0[2]; fa[3] = a0[3]; fa[4] = a1[0]; fa[5] = a1[1]; fa[6] = a1[2]; fa[7] = a1[3];
                    fb[0] = b0[0]; fb[1] = b0[1]; fb[2] = b0[2]; fb[3] = b0[3]; fb[4] = b1[0]; fb[5] = b1[1]; fb[6] = b1[2]; fb[7] = b1[3];
                    ov[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, pp, ov[db], 0, 0, 0);        // dV^T += dO^T P
                    ok[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb, dd, ok[db], 0, 0, 0);        // dK^T += Q^T dS
                }
            }
        }
    }
    csum += __shfl_xor(csum, 32, 64);
    if (livek && hi == 0) atomicAdd(&c[(size_t)r * T + k], csum * (gscale[r] / (float)H));
    if (GRAD && livek) {
        f16* orow = dqkv + ((size_t)r * T + k) * ld + h * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                f16x4 hk, hv;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) { hk[jj] = (f16)(ok[db][rq * 4 + jj] * (1.f / DS_SCALE)); hv[jj] = (f16)ov[db][rq * 4 + jj]; }
                *reinterpret_cast<f16x4*>(orow + D + db * 32 + rq * 8 + 4 * hi) = hk;
                *reinterpret_cast<f16x4*>(orow + 2 * D + db * 32 + rq * 8 + 4 * hi) = hv;
            }
    }
}

// Attention backward of one block for R = L * n sequences (+ the rollout update of that block).  dqkv NULL: rollout update only (the
// last block that enters the rollout needs no gradient below it).  stats: scratch fp32 [R, H, T, 4].
extern "C" int semabs_attention_bwd(const void* qkv, const void* att, const float* fwd_stats, const void* dO, const float* rvec, const float* gscale,
                                    float* c, float* stats, void* dqkv, int n, int L, int T, int H, int head_dim, int positive_only, void* stream) {
    if (n == 0 || L == 0) return SEMABS_OK;
    SEMABS_REQUIRE(qkv && att && fwd_stats && dO && rvec && gscale && c && stats && n > 0 && L > 0 && H > 0, "semabs_attention_bwd: bad args");
    SEMABS_REQUIRE(head_dim == 64 && T > 0 && T <= 288, "semabs_attention_bwd: head_dim must be 64 and T <= 288");
    const long R = (long)L * n;
    SEMABS_REQUIRE(R * H < (1L << 31), "semabs_attention_bwd: too many sequences");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)(R * H));
    const int nkb = (T + 31) / 32;
#define BWD_LAUNCH(N)                                                                                                               \
    {                                                                                                                                \
        constexpr int TPc = 32 * N, VSc = TPc + 4;                                                                                   \
        const size_t lq = (size_t)TPc * 256 + 64 * VSc * 2, lkv = (size_t)TPc * 256 + TPc * 16 + 2 * 64 * VSc * 2, lr = (size_t)TPc * 256 + TPc * 16; \
        static SemabsLdsAttr attr_q, attr_kv, attr_r;                                                                                \
        semabs_ensure_lds(&k_attn_bwd_dq<N, true>, (int)lq, attr_q);                                                                 \
        semabs_ensure_lds(&k_attn_bwd_dkv<N, true>, (int)lkv, attr_kv);                                                              \
        semabs_ensure_lds(&k_attn_bwd_dkv<N, false>, (int)lr, attr_r);                                                               \
        if (dqkv) {                                                                                                                  \
            hipLaunchKernelGGL((k_attn_bwd_dq<N, true>), grid, dim3(64 * N), lq, s, (const f16*)qkv, (const f16*)att, fwd_stats, (const f16*)dO, rvec, stats, (f16*)dqkv, n, T, H); \
            hipLaunchKernelGGL((k_attn_bwd_dkv<N, true>), grid, dim3(64 * N), lkv, s, (const f16*)qkv, (const f16*)dO, stats, gscale, c, (f16*)dqkv, n, T, H, positive_only);     \
        } else {                                                                                                                     \
            hipLaunchKernelGGL((k_attn_bwd_dq<N, false>), grid, dim3(64 * N), 0, s, (const f16*)qkv, (const f16*)att, fwd_stats, (const f16*)dO, rvec, stats, (f16*)nullptr, n, T, H); \
            hipLaunchKernelGGL((k_attn_bwd_dkv<N, false>), grid, dim3(64 * N), lr, s, (const f16*)qkv, (const f16*)dO, stats, gscale, c, (f16*)nullptr, n, T, H, positive_only);   \
        }                                                                                                                            \
    }
    switch (nkb) {
        case 1: case 2: BWD_LAUNCH(2) break;
        case 3: BWD_LAUNCH(3) break;
        case 4: BWD_LAUNCH(4) break;
        case 5: BWD_LAUNCH(5) break;
        case 6: BWD_LAUNCH(6) break;
        case 7: BWD_LAUNCH(7) break;
        case 8: BWD_LAUNCH(8) break;
        default: BWD_LAUNCH(9) break;                        // T = 257: ViT-L/14
    }
#undef BWD_LAUNCH
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// Per-sequence renormalisation: g32 [R, len] *= 2^k with max |g| * 2^k in [0.5, 1); g16 = fp16(g32); gscale[r] /= 2^k.
__global__ __launch_bounds__(1024) void k_seq_rescale(float* __restrict__ g32, f16* __restrict__ g16, float* __restrict__ gscale, long len) {
    __shared__ float red[16];
    __shared__ float s_mul;
    float* g = g32 + (size_t)blockIdx.x * len;
    float mx = 0.f;
    for (long i = threadIdx.x * 4L; i < len; i += blockDim.x * 4L) {
        const float4 v = *reinterpret_cast<const float4*>(g + i);
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = 0.f;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) m = fmaxf(m, red[i]);
        float mul = 1.f;
        if (m > 0.f && isfinite(m)) {
            int e;
            (void)frexpf(m, &e);                            // m = f * 2^e, f in [0.5, 1)
            mul = ldexpf(1.f, -e);
        }
        s_mul = mul;
        gscale[blockIdx.x] = gscale[blockIdx.x] / mul;
    }
    __syncthreads();
    const float mul = s_mul;
    f16* h = g16 ? g16 + (size_t)blockIdx.x * len : nullptr;
    for (long i = threadIdx.x * 4L; i < len; i += blockDim.x * 4L) {
        float4 v = *reinterpret_cast<const float4*>(g + i);
        v.x *= mul; v.y *= mul; v.z *= mul; v.w *= mul;
        *reinterpret_cast<float4*>(g + i) = v;
        if (h) { f16x4 hv; hv[0] = (f16)v.x; hv[1] = (f16)v.y; hv[2] = (f16)v.z; hv[3] = (f16)v.w; *reinterpret_cast<f16x4*>(h + i) = hv; }
    }
}
extern "C" int semabs_seq_rescale(float* g32, void* g16, float* gscale, long R, long len, void* stream) {
    if (R == 0) return SEMABS_OK;
    SEMABS_REQUIRE(g32 && gscale && R > 0 && len > 0 && len % 4 == 0, "semabs_seq_rescale: bad args");
    hipLaunchKernelGGL(k_seq_rescale, dim3((unsigned)R), dim3(1024), 0, (hipStream_t)stream, g32, (f16*)g16, gscale, len);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

__global__ void k_rollout_step(float* __restrict__ r, float* __restrict__ c, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { r[i] += c[i]; c[i] = 0.f; }
}
// r += c; c = 0   (one rollout layer:  r <- r + r cam_i, with c = r cam_i accumulated by semabs_attention_bwd)
extern "C" int semabs_rollout_step(float* r, float* c, long n, void* stream) {
    if (n == 0) return SEMABS_OK;
    SEMABS_REQUIRE(r && c && n > 0, "semabs_rollout_step: bad args");
    hipLaunchKernelGGL(k_rollout_step, dim3(semabs_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, r, c, n);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}
