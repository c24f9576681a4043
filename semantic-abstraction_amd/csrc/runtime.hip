// Library-level entry points: error string, version, device sanity.
#include "semabs_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void semabs_set_error(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}

extern "C" const char* semabs_last_error(void) { return g_err; }

extern "C" int semabs_abi_version(void) { return 6; }   // bumped whenever an exported signature changes or disappears (round 3: + semabs_cos_head, semabs_compact_subsample; 4: semabs_wgrad_conv3 takes a scratch buffer, + semabs_wgrad_mfma, semabs_quickgelu_grad, GEMM epilogue 5; 5: + semabs_unflip_average; 6 (round 5): + semabs_color_jitter_op, hue factor range checked)

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

// HIP event helpers for the per-launch GEMM timing of bench.py (events with timing enabled; passed to semabs_gemm_f16_ex)
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
