// Library-level entry points: error string, version, device sanity.
#include "semabs_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void semabs_set_error(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}

extern "C" const char* semabs_last_error(void) { return g_err; }

extern "C" int semabs_abi_version(void) { return 8; }   // bumped whenever an exported signature changes or disappears (round 3: + semabs_cos_head, semabs_compact_subsample; 4: semabs_wgrad_conv3 takes a scratch buffer, + semabs_wgrad_mfma, semabs_quickgelu_grad, GEMM epilogue 5; 5: + semabs_unflip_average; 6 (round 5): + semabs_color_jitter_op, hue factor range checked; 7: + semabs_wgrad_conv3_gn, semabs_vool_sample_bwd takes absmax_bits; 8 (round 6): row centres in semabs_gemm_f16_ln / semabs_ln_rowstats / semabs_layernorm)

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

// ------------------------------------------------------------------------------------------------
// CU-masked streams (round 6): a stream created with a CU mask confines its kernels to those CUs; the persistent kernels ask how many that is.
// ------------------------------------------------------------------------------------------------
#include <mutex>
#include <unordered_map>
static int device_cus() {
    static int n = 0;
    if (!n) { int dev = 0; (void)hipGetDevice(&dev); if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256; }
    return n;
}
static std::mutex g_cu_mu;
static std::unordered_map<hipStream_t, int> g_cu_cache;     // a stream's mask is fixed at creation; semabs_stream_destroy / _create_cumask drop a handle's entry
int semabs_stream_cus(hipStream_t s) {
    const int all = device_cus();
    if (!s) return all;
    {
        std::lock_guard<std::mutex> lk(g_cu_mu);
        auto it = g_cu_cache.find(s);
        if (it != g_cu_cache.end()) return it->second;
    }
    uint32_t mask[16] = {0};
    int n = all;
    if (hipExtStreamGetCUMask(s, 16, mask) == hipSuccess) {
        int c = 0;
        for (int i = 0; i < 16; ++i) c += __builtin_popcount(mask[i]);
        if (c > 0 && c < all) n = c;
    } else {
        (void)hipGetLastError();
    }
    std::lock_guard<std::mutex> lk(g_cu_mu);
    g_cu_cache[s] = n;
    return n;
}
// A stream whose kernels run on the CUs of `mask` only (n_words x 32 bits, bit i = CU i in the runtime's enumeration).  The caller owns it (semabs_stream_destroy).
extern "C" int semabs_stream_create_cumask(const unsigned int* mask, int n_words, void** stream) {
    SEMABS_REQUIRE(mask && n_words > 0 && n_words <= 16 && stream, "semabs_stream_create_cumask: bad args");
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, mask) != hipSuccess) { (void)hipGetLastError(); semabs_set_error("hipExtStreamCreateWithCUMask failed"); return SEMABS_EHIP; }
    { std::lock_guard<std::mutex> lk(g_cu_mu); g_cu_cache.erase(s); }
    *stream = (void*)s;
    return SEMABS_OK;
}
extern "C" int semabs_stream_destroy(void* stream) {
    if (stream) {
        { std::lock_guard<std::mutex> lk(g_cu_mu); g_cu_cache.erase((hipStream_t)stream); }
        (void)hipStreamDestroy((hipStream_t)stream);
    }
    return SEMABS_OK;
}
extern "C" int semabs_stream_cu_count(void* stream, int* cus) {
    SEMABS_REQUIRE(cus, "semabs_stream_cu_count: null pointer");
    *cus = semabs_stream_cus((hipStream_t)stream);
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

// ------------------------------------------------------------------------------------------------
// Buffer utilities of the host side (round 5): the hot path issues no ATen kernel - fills, replications and the
// empty-cloud poisoning of a scene are launches of this library on the caller's stream.
// ------------------------------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
static __global__ __launch_bounds__(256) void k_fill128(u32x4* __restrict__ p, long n16, unsigned int v) {
    const u32x4 w = u32x4{v, v, v, v};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) __builtin_nontemporal_store(w, p + i);
}
// nbytes % 4 == 0, p 4-byte aligned: 16-byte streaming stores over the aligned body, 4-byte stores for the ragged ends
extern "C" int semabs_fill_u32(void* p, long long nbytes, unsigned int value, void* stream) {
    if (nbytes == 0) return SEMABS_OK;
    SEMABS_REQUIRE(p && nbytes > 0 && nbytes % 4 == 0 && ((uintptr_t)p & 3) == 0, "semabs_fill_u32: pointer and byte count must be multiples of 4");
    hipStream_t s = (hipStream_t)stream;
    char* c = (char*)p;
    const long long head = ((16 - ((uintptr_t)c & 15)) & 15) < nbytes ? ((16 - ((uintptr_t)c & 15)) & 15) : nbytes;
    if (head) semabs_fill32(c, (size_t)head, value, s);
    const long long body = (nbytes - head) / 16 * 16;
    if (body) {
        const long n16 = body / 16;
        long nb = (n16 + 255) / 256; if (nb > 8192) nb = 8192;
        hipLaunchKernelGGL(k_fill128, dim3((unsigned)nb), dim3(256), 0, s, (u32x4*)(c + head), n16, value);
    }
    if (nbytes - head - body) semabs_fill32(c + head + body, (size_t)(nbytes - head - body), value, s);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// dst = reps copies of src back to back (nbytes each; nbytes % 4 == 0, both 4-byte aligned): the image stack of make_images, the broadcast of one TSDF to the label volumes
static __global__ __launch_bounds__(256) void k_replicate(const unsigned int* __restrict__ src, unsigned int* __restrict__ dst, long n, int reps) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const unsigned int v = src[i];
        for (int r = 0; r < reps; ++r) dst[(long)r * n + i] = v;
    }
}
extern "C" int semabs_replicate(const void* src, void* dst, long long nbytes, int reps, void* stream) {
    if (nbytes == 0 || reps == 0) return SEMABS_OK;
    SEMABS_REQUIRE(src && dst && nbytes > 0 && reps > 0 && nbytes % 4 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 3) == 0, "semabs_replicate: bad args");
    const long n = (long)(nbytes / 4);
    long nb = (n + 255) / 256; if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(k_replicate, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const unsigned int*)src, (unsigned int*)dst, n, reps);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// rows x width 32-bit words from a pitched source to a pitched destination (pitches in words): the pack / unpack of the tile-sharded relevancy all-gather
// (dist.allgather_tile_relevance) without torch.cat / slice-assign kernels.
static __global__ __launch_bounds__(256) void k_copy2d(const unsigned int* __restrict__ src, long src_pitch, unsigned int* __restrict__ dst, long dst_pitch, long width, long rows) {
    const long n = width * rows;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long r = i / width, c = i - r * width;
        dst[r * dst_pitch + c] = src[r * src_pitch + c];
    }
}
extern "C" int semabs_copy2d(const void* src, long long src_pitch_bytes, void* dst, long long dst_pitch_bytes, long long width_bytes, long long rows, void* stream) {
    if (width_bytes == 0 || rows == 0) return SEMABS_OK;
    SEMABS_REQUIRE(src && dst && width_bytes > 0 && rows > 0 && (width_bytes | src_pitch_bytes | dst_pitch_bytes) % 4 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 3) == 0 &&
                   src_pitch_bytes >= width_bytes && dst_pitch_bytes >= width_bytes, "semabs_copy2d: 4-byte aligned pointers / pitches / width, pitches >= width");
    const long n = (long)(width_bytes / 4 * rows);
    long nb = (n + 255) / 256; if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(k_copy2d, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const unsigned int*)src, (long)(src_pitch_bytes / 4), (unsigned int*)dst,
                       (long)(dst_pitch_bytes / 4), (long)(width_bytes / 4), (long)rows);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// A scene whose depth image has no point inside scene_bounds is an error in the reference (np.random.choice on an empty population, visualize.py:193).
// The device path learns the count without a host synchronisation: when *n_in == 0 the logits become NaN and the labels -1 (one load and an early
// exit otherwise), so the result cannot be mistaken for a valid one; SceneResult.n_in_bounds raises on first read.
static __global__ __launch_bounds__(256) void k_poison_empty(const long long* __restrict__ n_in, float* __restrict__ logits, long n_logits,
                                                             int* __restrict__ labels, long n_labels) {
    if (*n_in != 0) return;
    const float nan = __builtin_nanf("");
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_logits; i += (long)gridDim.x * 256) logits[i] = nan;
    if (labels)
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_labels; i += (long)gridDim.x * 256) labels[i] = -1;
}
extern "C" int semabs_poison_empty(const long long* n_in, float* logits, long long n_logits, int* labels, long long n_labels, void* stream) {
    SEMABS_REQUIRE(n_in && logits && n_logits >= 0 && n_labels >= 0, "semabs_poison_empty: bad args");
    hipLaunchKernelGGL(k_poison_empty, dim3(1024), dim3(256), 0, (hipStream_t)stream, n_in, logits, (long)n_logits, labels, (long)n_labels);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}
