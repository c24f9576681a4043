// Skeleton probe for k_gemm8: 8 waves = 2 wave rows that alternate MFMA blocks separated by workgroup barriers (row 1 runs one barrier
// behind row 0, as in the kernel), nothing else.  How much of the matrix pipe survives the alternation, by block length and MFMA shape?
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_pingpong.cpp -o /tmp/pp && /tmp/pp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 0: both barriers of a phase (SYNC + END), rows skewed; 1: one barrier per block, rows skewed; 2: no barriers at all
template <int NBLK, int MODE, bool BIG>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    const int wr = threadIdx.x >> 8;                 // wave row 0 / 1
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
    f32x4 acc[16]; f32x16 accb[4];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) accb[i][e] = 0.f;
    if (MODE != 2 && wr == 1) __builtin_amdgcn_s_barrier();
    for (int it = 0; it < iters; ++it) {
        if (MODE != 2) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
        __builtin_amdgcn_s_setprio(1);
        if (BIG) {
#pragma unroll
            for (int i = 0; i < NBLK / 2; ++i) accb[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, accb[i & 3], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < NBLK; ++i) acc[i & 15] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i & 15], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        if (MODE == 0) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
    }
    if (MODE != 2 && wr == 0) __builtin_amdgcn_s_barrier();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0];
    for (int i = 0; i < 4; ++i) s += accb[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K>
static double run(K kern, int iters, int nblk, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
    return 5.0 * 256 * 8 * (double)iters * nblk * (2.0 * 16 * 16 * 32) / (ms * 1e-3) / 1e12;
}
#define ROW(N) printf("%3d MFMAs (16x16x32 equiv.) per block:  two barriers %7.1f   one barrier %7.1f   none %7.1f   |  32x32x16: two %7.1f   one %7.1f   none %7.1f  TFLOP/s\n", N, \
    run(k<N, 0, false>, 200000 / N, N, out), run(k<N, 1, false>, 200000 / N, N, out), run(k<N, 2, false>, 200000 / N, N, out), \
    run(k<N, 0, true>, 200000 / N, N, out), run(k<N, 1, true>, 200000 / N, N, out), run(k<N, 2, true>, 200000 / N, N, out));
int main() {
    float* out; (void)hipMalloc(&out, 256L * 512 * 4);
    ROW(8) ROW(16) ROW(32) ROW(64) ROW(128)
    return 0;
}
