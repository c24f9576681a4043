// Multi-layer attention x gradient rollout for deep CLIP towers (ViT-L/14: 24 blocks, 13 of which enter the rollout) - SURVEY.md 8 f4.
//
// Reference: ClipGradcam.interpret (CLIP/clip/clip_gradcam.py:70-132).  For every block i > num_layers (10) it takes
//     grad_i = d logit_l / d attn_probs_i   (torch.autograd.grad, one backward pass per label),
//     cam_i  = mean_h clamp(grad_i * attn_probs_i, min = 0)      (clamp iff positive_attn_only),        R <- R + cam_i R,
// and returns R[:, :, 0, 1:].  Only row 0 of the final R is used and R_final = (I + cam_23) ... (I + cam_11), so a ROW VECTOR is enough:
//     r <- e_0;   for i = 23 ... 11:   r <- r + r cam_i.
// grad_i[h, q, k] = dO_i[q, h, :] . V_i[k, h, :] with dO_i the gradient wrt the attention output of block i (before out_proj), so no T x T
// matrix is ever materialised: one pass over (query, key) pairs per (label, tile, head) recomputes P from Q, K, forms dP = dO V^T on the
// fly and accumulates both the rollout update  c[k] += r[q] act(P dP) / H  and the attention backward (dQ, dK, dV) that carries the
// gradient down to the previous block.  ViT-B never gets here: only its last block enters the rollout (closed form in vit.hip).
//
// Kernels (fp32 accumulate, fp16 operands = what the forward stored; one workgroup per (sequence, head), T <= 320, head_dim 64):
//   k_attn_bwd_q   thread = query: row max / sum (online softmax), delta = sum_k P dP, rollout update, dQ
//   k_attn_bwd_kv  thread = key:   dK = dS^T Q, dV = P^T dO   (reads the row statistics k_attn_bwd_q left behind)
//   k_seq_rescale  per sequence power-of-two renormalisation of the residual gradient (the chain is linear and cam is positively
//                  homogeneous in it, so the factor is divided out of the rollout update) - keeps the fp16 GEMM operands in range
//   k_rollout_step r += c, c = 0
// These first versions run on the vector ALU (v_dot2c_f32_f16 for the 64-long dots); an MFMA formulation is the obvious next step.
#include "semabs_common.h"

typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float dot64(const f16x8 (&a)[8], const f16x8* __restrict__ b) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const f16x8 y = b[c];
#pragma unroll
        for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_fdot2(f16x2v{a[c][2 * e], a[c][2 * e + 1]}, f16x2v{y[2 * e], y[2 * e + 1]}, s, false);
    }
    return s;
}

// qkv fp16 [n, T, 3D] (q pre-scaled | k | v) of the block; dO fp16 [R, T, D] with R = L * n sequences ordered (label, tile);
// rvec fp32 [R, T] (the rollout row BEFORE this block), gscale fp32 [R] (true gradient = stored gradient * gscale);
// c fp32 [R, T] += (1 / H) sum_q rvec[q] act(P[q, k] dP[q, k] gscale)          (atomic over heads);
// stats fp32 [R, H, T, 4] = (row max, 1 / row sum, delta, -) for k_attn_bwd_kv;  dqkv fp16 [R, T, 3D]: the dQ third (null: rollout only).
__global__ __launch_bounds__(320) void k_attn_bwd_q(const f16* __restrict__ qkv, const f16* __restrict__ dO, const float* __restrict__ rvec,
                                                    const float* __restrict__ gscale, float* __restrict__ c, float* __restrict__ stats,
                                                    f16* __restrict__ dqkv, int n, int T, int H, int positive_only) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f16* sK = reinterpret_cast<f16*>(smem);                 // [T][64]
    f16* sV = sK + (size_t)T * 64;                          // [T][64]
    float* sC = reinterpret_cast<float*>(sV + (size_t)T * 64);   // [T]
    const int D = H * 64;
    const int r = blockIdx.x / H, h = blockIdx.x % H;
    const int tile = r % n;
    const f16* base = qkv + (size_t)tile * T * 3 * D + h * 64;
    for (int i = threadIdx.x; i < T * 8; i += blockDim.x) {
        const int row = i >> 3, ch = i & 7;
        *reinterpret_cast<f16x8*>(sK + row * 64 + ch * 8) = *reinterpret_cast<const f16x8*>(base + (size_t)row * 3 * D + D + ch * 8);
        *reinterpret_cast<f16x8*>(sV + row * 64 + ch * 8) = *reinterpret_cast<const f16x8*>(base + (size_t)row * 3 * D + 2 * D + ch * 8);
    }
    for (int i = threadIdx.x; i < T; i += blockDim.x) sC[i] = 0.f;
    __syncthreads();
    const int q = threadIdx.x;
    const bool live = q < T;
    const int qc = live ? q : T - 1;
    f16x8 fq[8], fo[8];
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
        fq[ch] = *reinterpret_cast<const f16x8*>(base + (size_t)qc * 3 * D + ch * 8);
        fo[ch] = *reinterpret_cast<const f16x8*>(dO + ((size_t)r * T + qc) * D + h * 64 + ch * 8);
    }
    // pass A: row max and sum (online)
    float m = -INFINITY, sum = 0.f;
    for (int k = 0; k < T; ++k) {
        const float s = dot64(fq, reinterpret_cast<const f16x8*>(sK + k * 64));
        const float mn = fmaxf(m, s);
        sum = sum * __expf(m - mn) + __expf(s - mn);
        m = mn;
    }
    const float inv = 1.f / sum;
    // pass B: delta and the rollout update
    const float rq = live ? rvec[(size_t)r * T + q] : 0.f;
    const float gs = gscale[r] / (float)H;
    float delta = 0.f;
    for (int k = 0; k < T; ++k) {
        const float s = dot64(fq, reinterpret_cast<const f16x8*>(sK + k * 64));
        const float p = __expf(s - m) * inv;
        const float dp = dot64(fo, reinterpret_cast<const f16x8*>(sV + k * 64));
        const float pd = p * dp;
        delta += pd;
        float t = pd * gs;
        if (positive_only) t = fmaxf(t, 0.f);
        t = wave_sum(rq * t);                               // dead lanes carry rq = 0
        if ((threadIdx.x & 63) == 0) atomicAdd(&sC[k], t);
    }
    if (live) {
        float* st = stats + (((size_t)r * H + h) * T + q) * 4;
        st[0] = m; st[1] = inv; st[2] = delta;
    }
    // pass C: dQ = dS K with dS = P (dP - delta)
    if (dqkv) {
        float dq[64];
#pragma unroll
        for (int d = 0; d < 64; ++d) dq[d] = 0.f;
        for (int k = 0; k < T; ++k) {
            const f16x8* kr = reinterpret_cast<const f16x8*>(sK + k * 64);
            const float s = dot64(fq, kr);
            const float p = __expf(s - m) * inv;
            const float dp = dot64(fo, reinterpret_cast<const f16x8*>(sV + k * 64));
            const float ds = p * (dp - delta);
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) {
                const f16x8 kv = kr[ch];
#pragma unroll
                for (int e = 0; e < 8; ++e) dq[ch * 8 + e] = fmaf(ds, (float)kv[e], dq[ch * 8 + e]);
            }
        }
        if (live) {
            f16* o = dqkv + ((size_t)r * T + q) * 3 * D + h * 64;
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) {
                f16x8 hv;
#pragma unroll
                for (int e = 0; e < 8; ++e) hv[e] = (f16)dq[ch * 8 + e];
                *reinterpret_cast<f16x8*>(o + ch * 8) = hv;
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < T; i += blockDim.x) atomicAdd(&c[(size_t)r * T + i], sC[i]);
}

// dK[k] = sum_q dS[q, k] Q[q],  dV[k] = sum_q P[q, k] dO[q]   -> the K and V thirds of dqkv
__global__ __launch_bounds__(320) void k_attn_bwd_kv(const f16* __restrict__ qkv, const f16* __restrict__ dO, const float* __restrict__ stats,
                                                     f16* __restrict__ dqkv, int n, int T, int H) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f16* sQ = reinterpret_cast<f16*>(smem);                 // [T][64]
    f16* sO = sQ + (size_t)T * 64;                          // [T][64]  (dO)
    float* sS = reinterpret_cast<float*>(sO + (size_t)T * 64);   // [T][4]
    const int D = H * 64;
    const int r = blockIdx.x / H, h = blockIdx.x % H;
    const int tile = r % n;
    const f16* base = qkv + (size_t)tile * T * 3 * D + h * 64;
    for (int i = threadIdx.x; i < T * 8; i += blockDim.x) {
        const int row = i >> 3, ch = i & 7;
        *reinterpret_cast<f16x8*>(sQ + row * 64 + ch * 8) = *reinterpret_cast<const f16x8*>(base + (size_t)row * 3 * D + ch * 8);
        *reinterpret_cast<f16x8*>(sO + row * 64 + ch * 8) = *reinterpret_cast<const f16x8*>(dO + ((size_t)r * T + row) * D + h * 64 + ch * 8);
    }
    for (int i = threadIdx.x; i < T; i += blockDim.x)
        *reinterpret_cast<float4*>(sS + i * 4) = *reinterpret_cast<const float4*>(stats + (((size_t)r * H + h) * T + i) * 4);
    __syncthreads();
    const int k = threadIdx.x;
    const bool live = k < T;
    const int kc = live ? k : T - 1;
    f16x8 fk[8], fv[8];
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
        fk[ch] = *reinterpret_cast<const f16x8*>(base + (size_t)kc * 3 * D + D + ch * 8);
        fv[ch] = *reinterpret_cast<const f16x8*>(base + (size_t)kc * 3 * D + 2 * D + ch * 8);
    }
    float dk[64], dv[64];
#pragma unroll
    for (int d = 0; d < 64; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
    for (int q = 0; q < T; ++q) {
        const f16x8* qr = reinterpret_cast<const f16x8*>(sQ + q * 64);
        const f16x8* orow = reinterpret_cast<const f16x8*>(sO + q * 64);
        const float4 st = *reinterpret_cast<const float4*>(sS + q * 4);
        const float s = dot64(fk, qr);
        const float p = __expf(s - st.x) * st.y;
        const float dp = dot64(fv, orow);
        const float ds = p * (dp - st.z);
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) {
            const f16x8 qv = qr[ch], ov = orow[ch];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                dk[ch * 8 + e] = fmaf(ds, (float)qv[e], dk[ch * 8 + e]);
                dv[ch * 8 + e] = fmaf(p, (float)ov[e], dv[ch * 8 + e]);
            }
        }
    }
    if (live) {
        f16* o = dqkv + ((size_t)r * T + k) * 3 * D + h * 64;
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) {
            f16x8 hk, hv;
#pragma unroll
            for (int e = 0; e < 8; ++e) { hk[e] = (f16)dk[ch * 8 + e]; hv[e] = (f16)dv[ch * 8 + e]; }
            *reinterpret_cast<f16x8*>(o + D + ch * 8) = hk;
            *reinterpret_cast<f16x8*>(o + 2 * D + ch * 8) = hv;
        }
    }
}

// Attention backward of one block for R = L * n sequences (+ the rollout update of that block).  dqkv NULL: rollout update only (the
// last block that enters the rollout needs no gradient below it).  stats: scratch fp32 [R, H, T, 4].
extern "C" int semabs_attention_bwd(const void* qkv, const void* dO, const float* rvec, const float* gscale, float* c, float* stats, void* dqkv,
                                    int n, int L, int T, int H, int head_dim, int positive_only, void* stream) {
    if (n == 0 || L == 0) return SEMABS_OK;
    SEMABS_REQUIRE(qkv && dO && rvec && gscale && c && stats && n > 0 && L > 0 && H > 0, "semabs_attention_bwd: bad args");
    SEMABS_REQUIRE(head_dim == 64 && T > 0 && T <= 320, "semabs_attention_bwd: head_dim must be 64 and T <= 320");
    const long R = (long)L * n;
    SEMABS_REQUIRE(R * H < (1L << 31), "semabs_attention_bwd: too many sequences");
    const size_t lds = (size_t)T * 64 * 2 * 2 + (size_t)T * 4 * 4;
    static bool set = false;
    if (!set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_bwd_q), hipFuncAttributeMaxDynamicSharedMemorySize, 320 * 64 * 4 + 320 * 16);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_bwd_kv), hipFuncAttributeMaxDynamicSharedMemorySize, 320 * 64 * 4 + 320 * 16);
        set = true;
    }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_attn_bwd_q, dim3((unsigned)(R * H)), dim3(320), lds, s, (const f16*)qkv, (const f16*)dO, rvec, gscale, c, stats, (f16*)dqkv,
                       n, T, H, positive_only);
    if (dqkv)
        hipLaunchKernelGGL(k_attn_bwd_kv, dim3((unsigned)(R * H)), dim3(320), lds, s, (const f16*)qkv, (const f16*)dO, stats, (f16*)dqkv, n, T, H);
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
