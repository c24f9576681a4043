// Issue-rate probe for the vector-ALU instructions of the attention softmax (round 4): cycles per wave64 instruction on one SIMD, measured with s_memtime
// around a loop of independent instructions, at 1 and 4 waves per SIMD.    hipcc --offload-arch=gfx950 -O3 tools/valu_rate.cpp -o /tmp/vr && /tmp/vr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
template <int OP>
__global__ void k(float* out, unsigned long long* clk, int iters) {
    float r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = (float)(threadIdx.x + i) * 0.001f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (OP == 0) {
#define S(i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
                REP8(S)
#undef S
            } else if (OP == 1) {
#define S(i) asm volatile("v_mul_f32 %0, 0x3f7fff00, %0" : "+v"(r[i]));
                REP8(S)
#undef S
            } else if (OP == 2) {
#define S(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*(double*)&r[2 * i]) : "v"(*(double*)&r[14]));
                S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(0)
#undef S
            } else if (OP == 3) {
#define S(i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(r[i]) : "v"(r[i + 8]));
                REP8(S)
#undef S
            } else if (OP == 4) {
#define S(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(r[i + 8]), "v"(r[(i + 1) & 7]));
                REP8(S)
#undef S
            } else if (OP == 5) {
#define S(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r[i]), "+v"(r[i + 8]));
                REP8(S)
#undef S
            } else if (OP == 6) {
#define S(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
                REP8(S)
#undef S
            } else if (OP == 7) {
#define S(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(r[i + 8]));
                REP8(S)
#undef S
            } else if (OP == 8) {
#define S(i) asm volatile("v_exp_f16 %0, %0" : "+v"(r[i]));
                REP8(S)
#undef S
            } else if (OP == 9) {
#define S(i) asm volatile("v_exp_f32 %0, %0\n\tv_mul_f32 %1, 0x3f7fff00, %1" : "+v"(r[i]), "+v"(r[i + 8]));
                REP8(S)
#undef S
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
static void run(const char* name, int per_iter) {
    float* out; unsigned long long* clk;
    hipMalloc(&out, 1024 * 4 * sizeof(float)); hipMalloc(&clk, 64 * sizeof(unsigned long long));
    for (int threads : {256, 1024}) {
        const int iters = 2000;
        hipLaunchKernelGGL(k<OP>, dim3(1), dim3(threads), 0, 0, out, clk, iters);
        hipLaunchKernelGGL(k<OP>, dim3(1), dim3(threads), 0, 0, out, clk, iters);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(threads / 64);
        hipMemcpy(h.data(), clk, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        unsigned long long mx = 0;
        for (auto v : h) mx = v > mx ? v : mx;
        const double n = (double)iters * per_iter * (threads / 256);      // instructions issued on one SIMD
        printf("%-28s %d wave(s)/SIMD: %.2f cycles per wave64 instruction on the SIMD\n", name, threads / 256, (double)mx / n);
    }
    hipFree(out); hipFree(clk);
}

int main() {
    run<1>("v_mul_f32", 32);
    run<7>("v_add_f32", 32);
    run<2>("v_pk_fma_f32", 32);
    run<3>("v_cvt_pk_f16_f32", 32);
    run<4>("v_max3_f32", 32);
    run<5>("v_permlane32_swap_b32", 32);
    run<0>("v_exp_f32", 32);
    run<8>("v_exp_f16", 32);
    run<6>("v_rcp_f32", 32);
    run<9>("v_exp_f32 + v_mul_f32 (pairs)", 64);
    return 0;
}
