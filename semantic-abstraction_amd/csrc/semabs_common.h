// Shared helpers for libsemabs_hip.so (gfx950 only; no CUDA paths, no hipify output).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <mutex>

#define SEMABS_OK 0
#define SEMABS_EINVAL (-1)   // bad argument (null pointer, unsupported shape)
#define SEMABS_EHIP (-2)     // HIP launch / runtime error; see semabs_last_error()

extern "C" const char* semabs_last_error(void);
void semabs_set_error(const char* msg);

#define SEMABS_REQUIRE(cond, msg)            \
    do {                                     \
        if (!(cond)) {                       \
            semabs_set_error(msg);           \
            return SEMABS_EINVAL;            \
        }                                    \
    } while (0)

#define SEMABS_CHECK_LAUNCH()                          \
    do {                                               \
        hipError_t e__ = hipGetLastError();            \
        if (e__ != hipSuccess) {                       \
            semabs_set_error(hipGetErrorString(e__));  \
            return SEMABS_EHIP;                        \
        }                                              \
    } while (0)

static inline int semabs_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Per-kernel "maximum dynamic LDS" attribute, raised on demand.  One of these per launch site (a function-local static): thread-safe, and - unlike a
// set-once flag - correct when a later call needs MORE than the first one did (sizes that depend on the problem shape).
struct SemabsLdsAttr { std::mutex m; int have = 0; };
template <typename Kern>
static inline void semabs_ensure_lds(Kern kern, int bytes, SemabsLdsAttr& st) {
    std::lock_guard<std::mutex> lock(st.m);
    if (bytes > st.have) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        st.have = bytes;
    }
}

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// wave64 reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Workgroup id -> work item with XCD locality and a direction.  The hardware places workgroup vb on XCD vb % 8; item ids are handed out so that
// XCD x owns the contiguous run [lo_x, lo_x + cnt_x) of the nb items and walks it in dispatch order - forwards, or (reverse) backwards.
// Consecutive kernels of a producer -> consumer chain alternate the direction ("zigzag"): a consumer then starts with the rows its producer
// wrote LAST, which are still in the 256 MiB Infinity Cache (and partly in that XCD's L2), instead of with the rows written first.
__device__ __forceinline__ int semabs_xcd_item(int vb, int nb, bool reverse) {
    const int q = nb >> 3, r = nb & 7, x = vb & 7, k = vb >> 3;
    const int lo = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q, cnt = q + (x < r ? 1 : 0);
    return lo + (reverse ? cnt - 1 - k : k);
}

// CUs a launch on stream `s` can occupy: the popcount of the stream's CU mask (hipExtStreamCreateWithCUMask; the whole device for ordinary streams), cached per
// stream handle (runtime.hip).  Persistent kernels size their grids with it, so that two streams with complementary masks each fill exactly their share.
int semabs_stream_cus(hipStream_t s);

// Fill nbytes (a multiple of 4, pointer 4-byte aligned) with a 32-bit pattern.  A kernel launch, not hipMemsetAsync: on this runtime a memset
// queued between kernels costs ~1.5 ms of stream time (measured on the 4-byte fill of semabs_grad_scale: 78.8 -> 77.1 ms per training step).
static __global__ __launch_bounds__(256) void semabs_k_fill32(unsigned int* __restrict__ p, long n, unsigned int v) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] = v;
}
static inline void semabs_fill32(void* p, size_t nbytes, unsigned int v, hipStream_t s) {
    const long n = (long)(nbytes / 4);
    if (n == 0) return;
    long nb = (n + 1023) / 1024; if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(semabs_k_fill32, dim3((unsigned)nb), dim3(256), 0, s, (unsigned int*)p, n, v);
}
