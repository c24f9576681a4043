// Do vector-ALU instructions of one wave overlap with MFMAs of ANOTHER wave on the same SIMD?  (round 4, attention kernel analysis)
// One workgroup of 8 waves = 2 per SIMD: waves 0-3 run `a`, waves 4-7 run `b`, a / b in {MFMA 32x32x16 chain pairs, v_exp_f32, v_mul_f32, idle}.
//   hipcc --offload-arch=gfx950 -O3 tools/coissue_probe.cpp -o /tmp/ci && /tmp/ci
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float work(int kind, int iters, float seed) {
    if (kind == 0) {                                  // 8 MFMAs per iteration on two independent accumulators
        f16x8 a, b;
#pragma unroll
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(seed - i); }
        f32x16 c0 = {0}, c1 = {0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            }
        }
        return c0[0] + c1[3];
    }
    float r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = seed * 0.001f + i;
    if (kind == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#define S(i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
                S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#undef S
            }
        }
    } else if (kind == 2) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#define S(i) asm volatile("v_mul_f32 %0, 0x3f7fff00, %0" : "+v"(r[i]));
                S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#undef S
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += r[i];
    return s;
}

__global__ __launch_bounds__(512) void k(float* out, unsigned long long* clk, int ka, int kb, int ia, int ib) {
    const int wid = threadIdx.x >> 6;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float v = 0.f;
    if (wid < 4) { if (ka >= 0) v = work(ka, ia, (float)threadIdx.x); }
    else { if (kb >= 0) v = work(kb, ib, (float)threadIdx.x); }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = v;
    if ((threadIdx.x & 63) == 0) clk[wid] = t1 - t0;
}

int main() {
    float* out; unsigned long long* clk;
    (void)hipMalloc(&out, 512 * sizeof(float)); (void)hipMalloc(&clk, 8 * sizeof(unsigned long long));
    const char* nm[] = {"idle", "MFMA 32x32x16 x8", "v_exp_f32 x32", "v_mul_f32 x32"};
    struct Case { int ka, kb, ia, ib; };
    // iteration counts chosen so that each side alone takes about the same time: 8 MFMAs = 256 cycles; 32 v_exp = 263; 32 v_mul = 80..160
    const Case cases[] = {{0, -1, 2000, 0}, {-1, 1, 0, 2000}, {-1, 2, 0, 4000}, {0, 1, 2000, 2000}, {0, 2, 2000, 4000}, {1, 1, 2000, 2000}, {0, 0, 2000, 2000}, {2, 2, 4000, 4000}};
    for (const Case& c : cases) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, out, clk, c.ka, c.kb, c.ia, c.ib);
        (void)hipDeviceSynchronize();
        unsigned long long h[8];
        (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
        printf("waves 0-3: %-18s waves 4-7: %-18s -> cycles  a: %8llu   b: %8llu\n", nm[c.ka + 1], nm[c.kb + 1], h[0], h[4]);
    }
    return 0;
}
