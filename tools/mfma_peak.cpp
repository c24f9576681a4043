// Pure MFMA issue-rate probe (no memory traffic): TFLOP/s of back-to-back independent v_mfma_f32_16x16x32_f16 / v_mfma_f32_32x32x16_f16 with
// W waves per SIMD.   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.cpp -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// RANDOM: operands = pseudo-random fp16 in (-2, 2) per lane (the data a GEMM sees) instead of near-constants: the matrix pipe's power, and with it the clock
// the part holds, depends on how many operand bits toggle
__device__ __forceinline__ _Float16 rnd16(unsigned& s) { s = s * 1664525u + 1013904223u; return (_Float16)(((int)(s >> 8) & 0xffff) * (4.f / 65536.f) - 2.f); }
template <int NACC, bool RANDOM = false>
__global__ __launch_bounds__(512) void k16(float* out, int iters) {
    f16x8 a, b;
    unsigned seed = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int e = 0; e < 8; ++e) { a[e] = RANDOM ? rnd16(seed) : (_Float16)(threadIdx.x * 0.001f + e); b[e] = RANDOM ? rnd16(seed) : (_Float16)(e * 0.5f); }
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(512) void k32(float* out, int iters) {
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K>
static double run(K kern, int threads, int blocks, int iters, double flop_per_mfma, int nacc, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)blocks * threads / 64.0;
    return 5.0 * waves * iters * nacc * flop_per_mfma / (ms * 1e-3) / 1e12;
}
int main() {
    float* out; hipMalloc(&out, 4096L * 512 * 4);
    const int iters = 200000;      // ~0.5 s per launch: long enough for the power management to settle
    for (int wpc : {4, 8, 16}) {                 // waves per workgroup = waves per CU when one workgroup per CU fits
        const int threads = wpc * 64 > 512 ? 512 : wpc * 64, blocks = 256 * (wpc * 64 / threads);
        printf("waves/CU %2d: 16x16x32 f16, 16 acc: %7.1f TFLOP/s (random operands: %7.1f)   4 acc: %7.1f   32x32x16 f16, 4 acc: %7.1f   2 acc: %7.1f\n", wpc,
               run(k16<16>, threads, blocks, iters, 2.0 * 16 * 16 * 32, 16, out), run(k16<16, true>, threads, blocks, iters, 2.0 * 16 * 16 * 32, 16, out),
               run(k16<4>, threads, blocks, iters, 2.0 * 16 * 16 * 32, 4, out),
               run(k32<4>, threads, blocks, iters, 2.0 * 32 * 32 * 16, 4, out), run(k32<2>, threads, blocks, iters, 2.0 * 32 * 32 * 16, 2, out));
    }
    return 0;
}
