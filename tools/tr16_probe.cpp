#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
// LDS image: 64 rows x 64 halfs (row stride 128 B), value = row * 64 + col.  Each lane supplies its own 8-byte address.
__global__ void k(const int* lane_off, short* out) {
    __shared__ __attribute__((aligned(16))) short lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (short)i;
    __syncthreads();
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + lane_off[threadIdx.x]));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
    int h_off[64]; short h_out[256];
    int* d_off; short* d_out;
    hipMalloc(&d_off, sizeof(h_off)); hipMalloc(&d_out, sizeof(h_out));
    // experiment A: lane i of each 16-lane group g points at (row = 4 g + i / 4, col chunk = (i % 4) * 4)
    for (int l = 0; l < 64; ++l) { int g = l >> 4, i = l & 15; h_off[l] = (4 * g + i / 4) * 64 + (i % 4) * 4; }
    hipMemcpy(d_off, h_off, sizeof(h_off), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_off, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("A: lane -> (row, col) of its 4 elements\n");
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" (%d,%d)", h_out[l * 4 + j] / 64, h_out[l * 4 + j] % 64); printf("\n"); }
    // experiment B: lane i points at (row = i % 4 + 4 g, chunk = i / 4): the other plausible ordering
    for (int l = 0; l < 64; ++l) { int g = l >> 4, i = l & 15; h_off[l] = (4 * g + i % 4) * 64 + (i / 4) * 4; }
    hipMemcpy(d_off, h_off, sizeof(h_off), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_off, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("B:\n");
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" (%d,%d)", h_out[l * 4 + j] / 64, h_out[l * 4 + j] % 64); printf("\n"); }
    return 0;
}
