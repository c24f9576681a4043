// Evaluation metrics after the path: the reference voxelises point predictions with three scatter-max passes and then walks
// every (scene, description) pair in Python.  Here: one pass over the points with integer atomics, one counting pass; the ratios are formed
// on the host from exact integer counts, in the reference's own operand order.
//
//   utils.voxelize_points      utils.py:617-665   (VirtualGrid(reduce_method="max").scatter_points x 3, net.py:185-201)
//   utils.prediction_analysis  utils.py:340-380   (+ utils.iou :329-337)
#include "semabs_common.h"

// flat int64 [BP, N] voxel index of every point; pred / label / ignore uint8 [BP, N] (non-zero = true)
// -> vol int32 [BP, nvox, 3] (zero-filled): [0] any positive prediction, [1] label state (0 empty, 1 only negatives, 2 any positive), [2] any ignore
__global__ void k_voxelize_eval(const long long* __restrict__ flat, const unsigned char* __restrict__ pred, const unsigned char* __restrict__ label,
                                const unsigned char* __restrict__ ignore, int* __restrict__ vol, long BP, long N, long nvox) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BP * N) return;
    const long bp = i / N;
    int* v = vol + (bp * nvox + flat[i]) * 3;
    if (pred[i]) atomicMax(v, 1);
    atomicMax(v + 1, label[i] ? 2 : 1);
    if (ignore[i]) atomicMax(v + 2, 1);
}
// vol -> prediction / label / ignore uint8 [BP, nvox]   (utils.py:648-664: missing label = empty voxel -> ignored)
__global__ void k_voxelize_finish(const int* __restrict__ vol, unsigned char* __restrict__ pred, unsigned char* __restrict__ label,
                                  unsigned char* __restrict__ ignore, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int p = vol[i * 3], s = vol[i * 3 + 1], g = vol[i * 3 + 2];
    pred[i] = p > 0; label[i] = s == 2; ignore[i] = (g > 0) || (s == 0);
}
extern "C" int semabs_voxelize_eval(const long long* flat, const unsigned char* pred, const unsigned char* label, const unsigned char* ignore,
                                    int* vol_scratch, unsigned char* out_pred, unsigned char* out_label, unsigned char* out_ignore, long BP, long N,
                                    long nvox, void* stream) {
    if (BP == 0) return SEMABS_OK;
    SEMABS_REQUIRE(flat && pred && label && ignore && vol_scratch && out_pred && out_label && out_ignore && N >= 0 && nvox > 0, "semabs_voxelize_eval: bad args");
    hipStream_t s = (hipStream_t)stream;
    semabs_fill32(vol_scratch, sizeof(int) * 3 * BP * nvox, 0u, s);
    if (N > 0) hipLaunchKernelGGL(k_voxelize_eval, dim3(semabs_cdiv(BP * N, 256)), dim3(256), 0, s, flat, pred, label, ignore, vol_scratch, BP, N, nvox);
    hipLaunchKernelGGL(k_voxelize_finish, dim3(semabs_cdiv(BP * nvox, 256)), dim3(256), 0, s, vol_scratch, out_pred, out_label, out_ignore, BP * nvox);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// counts int64 [BP, 6] += (valid, label, pred, label & pred, label | pred, ...) over the M elements of each row, ignoring ignore != 0:
//   [0] n valid   [1] positive labels   [2] positive predictions   [3] true positives   [4] union   [5] reserved
__global__ __launch_bounds__(256) void k_pred_counts(const unsigned char* __restrict__ pred, const unsigned char* __restrict__ label,
                                                     const unsigned char* __restrict__ ignore, unsigned long long* __restrict__ counts, long M) {
    const long bp = blockIdx.y;
    const unsigned char* p = pred + bp * M; const unsigned char* l = label + bp * M; const unsigned char* g = ignore + bp * M;
    int c[5] = {0, 0, 0, 0, 0};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < M; i += (long)gridDim.x * 256) {
        if (g[i]) continue;
        const int pv = p[i] != 0, lv = l[i] != 0;
        c[0] += 1; c[1] += lv; c[2] += pv; c[3] += lv & pv; c[4] += lv | pv;
    }
    __shared__ int sh[5];
    if (threadIdx.x < 5) sh[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        float v = wave_sum((float)c[k]);                    // < 2^24 per wave: exact
        if ((threadIdx.x & 63) == 0) atomicAdd(&sh[k], (int)v);
    }
    __syncthreads();
    if (threadIdx.x < 5) atomicAdd(&counts[bp * 6 + threadIdx.x], (unsigned long long)sh[threadIdx.x]);
}
extern "C" int semabs_prediction_counts(const unsigned char* pred, const unsigned char* label, const unsigned char* ignore, unsigned long long* counts,
                                        long BP, long M, void* stream) {
    if (BP == 0) return SEMABS_OK;
    SEMABS_REQUIRE(pred && label && ignore && counts && M >= 0 && BP < 65536, "semabs_prediction_counts: bad args");
    hipStream_t s = (hipStream_t)stream;
    semabs_fill32(counts, sizeof(unsigned long long) * 6 * BP, 0u, s);
    if (M > 0) {
        int bx = semabs_cdiv(M, 256 * 8); if (bx < 1) bx = 1; if (bx > 64) bx = 64;
        hipLaunchKernelGGL(k_pred_counts, dim3(bx, (unsigned)BP), dim3(256), 0, s, pred, label, ignore, counts, M);
    }
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}
