// Library-level entry points: error string, version, device sanity.
#include "semabs_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void semabs_set_error(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}

extern "C" const char* semabs_last_error(void) { return g_err; }

extern "C" int semabs_abi_version(void) { return 1; }

// Returns 0 and fills name/cu_count for the current device; the library only carries gfx950 code objects.
extern "C" int semabs_device_info(char* name, int name_len, int* cu_count, long long* hbm_bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { semabs_set_error("no HIP device"); return SEMABS_EHIP; }
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) { semabs_set_error("hipGetDeviceProperties failed"); return SEMABS_EHIP; }
    if (name && name_len > 0) { strncpy(name, p.gcnArchName, name_len - 1); name[name_len - 1] = 0; }
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (long long)p.totalGlobalMem;
    return SEMABS_OK;
}

// HIP event helpers for the per-launch GEMM timing of bench.py (events with timing enabled; see semabs_gemm_time_next)
extern "C" int semabs_event_create(void** ev) {
    SEMABS_REQUIRE(ev, "semabs_event_create: null pointer");
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) { semabs_set_error("hipEventCreate failed"); return SEMABS_EHIP; }
    *ev = (void*)e;
    return SEMABS_OK;
}
extern "C" int semabs_event_destroy(void* ev) {
    if (ev) (void)hipEventDestroy((hipEvent_t)ev);
    return SEMABS_OK;
}
extern "C" int semabs_event_elapsed_ms(void* start, void* stop, float* ms) {
    SEMABS_REQUIRE(start && stop && ms, "semabs_event_elapsed_ms: null pointer");
    if (hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop) != hipSuccess) { semabs_set_error("hipEventElapsedTime failed"); return SEMABS_EHIP; }
    return SEMABS_OK;
}

// ---- CU-partitioned streams -----------------------------------------------------------------------------------------------------------
// A HIP stream whose kernels only run on the compute units named in `mask` (bit i = CU i in the driver's numbering; `probe` below shows
// where the workgroups of such a stream actually land).  Used to run two independent tile-chunk pipelines side by side, each on its own half
// of the chip, so that the HBM-bound store phase of one overlaps the MFMA phase of the other.
extern "C" int semabs_stream_create_cumask(void** stream, const uint32_t* mask, int n_words) {
    SEMABS_REQUIRE(stream && mask && n_words > 0, "semabs_stream_create_cumask: bad arguments");
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, mask) != hipSuccess) { semabs_set_error("hipExtStreamCreateWithCUMask failed"); return SEMABS_EHIP; }
    *stream = (void*)s;
    return SEMABS_OK;
}
extern "C" int semabs_stream_destroy(void* stream) {
    if (stream) (void)hipStreamDestroy((hipStream_t)stream);
    return SEMABS_OK;
}
__global__ void k_probe_placement(int* out) {
    if (threadIdx.x == 0) {
        unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));      // HW_REG_XCC_ID[3:0]
        unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));       // HW_REG_HW_ID
        out[2 * blockIdx.x] = (int)xcc;
        out[2 * blockIdx.x + 1] = (int)hw;
    }
    __builtin_amdgcn_s_sleep(100);
}
// out: int32[n_blocks, 2] = (XCC id, HW_ID register) of the CU each workgroup of a 1-wave-per-block grid ran on
extern "C" int semabs_probe_placement(int* out, int n_blocks, void* stream) {
    SEMABS_REQUIRE(out && n_blocks > 0, "semabs_probe_placement: bad arguments");
    hipLaunchKernelGGL(k_probe_placement, dim3(n_blocks), dim3(64), 0, (hipStream_t)stream, out);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}
