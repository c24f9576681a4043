// Relevancy storage format either side of the hot path: what the reference does to the maps between
// ClipWrapper.get_clip_saliency and the HDF5 file, and between the file and the network input.
//
//   pack    generate_relevancy.py:95-118   nearest-exact resize [L, H, W] -> [L, h, w] (torch interpolate), append the mean-over-labels
//                                          map ("mean" row); text features [L, E] -> append their mean, L2-normalise every row
//   unpack  dataset.py:821-871             stored maps [P, h, w] (- the "mean" map when subtract_mean_relevancy) -> bilinear
//                                          (align_corners = False) to the image size [P, H, W]  (x 50 happens at dataset.py:1053)
//
// The HDF5 container itself (gzip chunks, region references, file locks) is storage and stays out of scope.
// Index / weight arithmetic follows ATen (UpSample.h: nearest_exact_idx, area_pixel_compute_source_index) in fp32.
#include "semabs_common.h"

// nearest-exact: src = min(floor((dst + 0.5) * (in / out)), in - 1), scale and product in fp32
__device__ __forceinline__ int nearest_exact_src(int dst, int in_size, float scale) {
    const int s = (int)floorf(((float)dst + 0.5f) * scale);
    return s < in_size - 1 ? s : in_size - 1;
}

// out [L + 1, h, w]: rows 0..L-1 the resized maps, row L their mean over labels.  One thread per output pixel.
__global__ void k_rel_pack(const float* __restrict__ maps, float* __restrict__ out, int L, int H, int W, int h, int w, float sy, float sx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= h * w) return;
    const int oy = i / w, ox = i - oy * w;
    const int iy = nearest_exact_src(oy, H, sy), ix = nearest_exact_src(ox, W, sx);
    float acc = 0.f;
    for (int l = 0; l < L; ++l) {
        const float v = maps[((long)l * H + iy) * W + ix];
        out[(long)l * h * w + i] = v;
        acc += v;                                            // torch.mean over dim 0: sequential fp32 sum, then one division
    }
    out[(long)L * h * w + i] = acc / (float)L;
}
extern "C" int semabs_relevancy_pack(const float* maps, float* out, int L, int H, int W, int h, int w, void* stream) {
    if (L == 0) return SEMABS_OK;
    SEMABS_REQUIRE(maps && out && L > 0 && H > 0 && W > 0 && h > 0 && w > 0, "semabs_relevancy_pack: bad args");
    hipLaunchKernelGGL(k_rel_pack, dim3(semabs_cdiv((long)h * w, 256)), dim3(256), 0, (hipStream_t)stream, maps, out, L, H, W, h, w,
                       (float)H / (float)h, (float)W / (float)w);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// feats [L, E] -> out [L + 1, E]: rows 0..L-1 the inputs, row L their mean; then every row divided by its L2 norm.  One wave per output row.
__global__ __launch_bounds__(64) void k_text_pack(const float* __restrict__ feats, float* __restrict__ out, int L, int E) {
    const int r = blockIdx.x, lane = threadIdx.x;
    float ss = 0.f;
    for (int k = lane; k < E; k += 64) {
        float v;
        if (r < L) v = feats[(long)r * E + k];
        else { float a = 0.f; for (int l = 0; l < L; ++l) a += feats[(long)l * E + k]; v = a / (float)L; }
        out[(long)r * E + k] = v;
        ss += v * v;
    }
    const float nrm = sqrtf(wave_sum(ss));
    for (int k = lane; k < E; k += 64) out[(long)r * E + k] /= nrm;
}
extern "C" int semabs_text_pack(const float* feats, float* out, int L, int E, void* stream) {
    if (L == 0) return SEMABS_OK;
    SEMABS_REQUIRE(feats && out && L > 0 && E > 0, "semabs_text_pack: bad args");
    hipLaunchKernelGGL(k_text_pack, dim3(L + 1), dim3(64), 0, (hipStream_t)stream, feats, out, L, E);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// bilinear, align_corners = False: src = max(0, (dst + 0.5) * scale - 0.5); i0 = floor(src), i1 = min(i0 + 1, in - 1), l1 = src - i0, l0 = 1 - l1
__device__ __forceinline__ void lin_src(int dst, int in_size, float scale, int& i0, int& i1, float& l0, float& l1) {
    float s = ((float)dst + 0.5f) * scale - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s; if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = s - (float)i0; l0 = 1.f - l1;
}
// stored [P, h, w] (row index rows[p] of a larger [R, h, w] array, or p itself when rows == nullptr), optional mean map [h, w] subtracted
// first, -> out [P, H, W] = out_scale * bilinear(...)
__global__ void k_rel_unpack(const float* __restrict__ stored, const long long* __restrict__ rows, const float* __restrict__ mean_map,
                             float* __restrict__ out, int P, int h, int w, int H, int W, float sy, float sx, float out_scale) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)P * H * W) return;
    const int ox = (int)(i % W); const long t = i / W; const int oy = (int)(t % H); const int p = (int)(t / H);
    int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
    lin_src(oy, h, sy, y0, y1, ly0, ly1);
    lin_src(ox, w, sx, x0, x1, lx0, lx1);
    const float* src = stored + (rows ? rows[p] : (long long)p) * h * w;
    auto at = [&](int y, int x) { float v = src[y * w + x]; if (mean_map) v -= mean_map[y * w + x]; return v; };
    const float top = lx0 * at(y0, x0) + lx1 * at(y0, x1), bot = lx0 * at(y1, x0) + lx1 * at(y1, x1);
    out[i] = out_scale * (ly0 * top + ly1 * bot);
}
extern "C" int semabs_relevancy_unpack(const float* stored, const long long* rows, const float* mean_map, float* out, int P, int h, int w, int H,
                                       int W, float out_scale, void* stream) {
    if (P == 0) return SEMABS_OK;
    SEMABS_REQUIRE(stored && out && P > 0 && h > 0 && w > 0 && H > 0 && W > 0, "semabs_relevancy_unpack: bad args");
    hipLaunchKernelGGL(k_rel_unpack, dim3(semabs_cdiv((long)P * H * W, 256)), dim3(256), 0, (hipStream_t)stream, stored, rows, mean_map, out, P, h,
                       w, H, W, (float)h / (float)H, (float)w / (float)W, out_scale);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}
