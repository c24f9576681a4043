// LAMB optimizer step as two multi-tensor HIP kernels (config 5: the VOOL / OVSSC training recipe uses
// arm/optim/lamb.py through utils.py:260-266, 416).
//
// Replaces `Lamb.step` (arm/optim/lamb.py:59-127), no bias correction:
//   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g g;  s = m / (sqrt(v) + eps) + wd w
//   r = clamp(|w|, 0, 10) / |s|   (1 when either norm is 0);   w -= lr r s
// Pass 1 updates the moments and accumulates |w|^2, |s|^2 per tensor (fp64 atomics); pass 2 recomputes s from the new
// moments and applies the update, so the step direction is never written to HBM.  HBM-bound: 5 reads + 3 writes of
// the parameter count per step.
#include "semabs_common.h"

#define LAMB_CHUNK 16384     // elements per workgroup

struct LambHyper { float lr, beta1, beta2, one_m_beta1, one_m_beta2, eps, wd; };

// chunk table: int64 [n_chunks, 3] = (tensor id, element offset, element count); ptrs: int64 [4, n_tensors] = w, g, m, v
__global__ __launch_bounds__(256) void k_lamb_moments(const long long* __restrict__ chunks, const long long* __restrict__ ptrs, int n_tensors,
                                                      LambHyper h, double* __restrict__ norms) {
    const long long tid_ = chunks[blockIdx.x * 3], off = chunks[blockIdx.x * 3 + 1], cnt = chunks[blockIdx.x * 3 + 2];
    const float* w = reinterpret_cast<const float*>(ptrs[tid_]) + off;
    const float* g = reinterpret_cast<const float*>(ptrs[n_tensors + tid_]) + off;
    float* m = reinterpret_cast<float*>(ptrs[2 * n_tensors + tid_]) + off;
    float* v = reinterpret_cast<float*>(ptrs[3 * n_tensors + tid_]) + off;
    float sw = 0.f, ss = 0.f;
    for (long long i = threadIdx.x; i < cnt; i += 256) {
        const float gi = g[i], wi = w[i];
        const float mi = m[i] * h.beta1 + gi * h.one_m_beta1;
        const float vi = v[i] * h.beta2 + h.one_m_beta2 * gi * gi;
        m[i] = mi; v[i] = vi;
        float s = mi / (sqrtf(vi) + h.eps);
        if (h.wd != 0.f) s += h.wd * wi;
        sw += wi * wi; ss += s * s;
    }
    __shared__ float red[2][4];
    sw = wave_sum(sw); ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sw; red[1][threadIdx.x >> 6] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&norms[tid_ * 2], (double)((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])));
        atomicAdd(&norms[tid_ * 2 + 1], (double)((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])));
    }
}

__global__ __launch_bounds__(256) void k_lamb_apply(const long long* __restrict__ chunks, const long long* __restrict__ ptrs, int n_tensors,
                                                    LambHyper h, const double* __restrict__ norms, float* __restrict__ stats, int adam) {
    const long long tid_ = chunks[blockIdx.x * 3], off = chunks[blockIdx.x * 3 + 1], cnt = chunks[blockIdx.x * 3 + 2];
    float* w = reinterpret_cast<float*>(ptrs[tid_]) + off;
    const float* m = reinterpret_cast<const float*>(ptrs[2 * n_tensors + tid_]) + off;
    const float* v = reinterpret_cast<const float*>(ptrs[3 * n_tensors + tid_]) + off;
    float wn = (float)sqrt(norms[tid_ * 2]);
    wn = fminf(fmaxf(wn, 0.f), 10.f);
    const float an = (float)sqrt(norms[tid_ * 2 + 1]);
    const float trust = (wn == 0.f || an == 0.f) ? 1.f : wn / an;
    if (off == 0 && threadIdx.x == 0 && stats) { stats[tid_ * 3] = wn; stats[tid_ * 3 + 1] = an; stats[tid_ * 3 + 2] = trust; }
    const float alpha = -h.lr * (adam ? 1.f : trust);
    for (long long i = threadIdx.x; i < cnt; i += 256) {
        const float wi = w[i];
        float s = m[i] / (sqrtf(v[i]) + h.eps);
        if (h.wd != 0.f) s += h.wd * wi;
        w[i] = wi + alpha * s;
    }
}

// chunks int64 [n_chunks, 3], ptrs int64 [4, n_tensors] (device arrays of device pointers to fp32 w, g, m, v),
// norms fp64 [n_tensors, 2] scratch (zeroed here), stats fp32 [n_tensors, 3] = weight_norm, adam_norm, trust_ratio (optional)
extern "C" int semabs_lamb_step(const long long* chunks, int n_chunks, const long long* ptrs, int n_tensors, double lr, double beta1,
                                double beta2, double eps, double weight_decay, int adam, double* norms, float* stats, void* stream) {
    if (n_chunks == 0 || n_tensors == 0) return SEMABS_OK;
    SEMABS_REQUIRE(chunks && ptrs && norms, "semabs_lamb_step: null pointer");
    SEMABS_REQUIRE(lr >= 0. && eps >= 0. && beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1., "semabs_lamb_step: invalid hyper-parameter");
    hipStream_t s = (hipStream_t)stream;
    // the reference forms 1 - beta in python doubles and lets torch cast the scalar to fp32
    LambHyper h{(float)lr, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)weight_decay};
    semabs_fill32(norms, sizeof(double) * 2 * n_tensors, 0u, s);
    hipLaunchKernelGGL(k_lamb_moments, dim3(n_chunks), dim3(256), 0, s, chunks, ptrs, n_tensors, h, norms);
    hipLaunchKernelGGL(k_lamb_apply, dim3(n_chunks), dim3(256), 0, s, chunks, ptrs, n_tensors, h, norms, stats, adam);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}
