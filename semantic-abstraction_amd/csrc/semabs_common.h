// Shared helpers for libsemabs_hip.so (gfx950 only; no CUDA paths, no hipify output).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

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

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
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
