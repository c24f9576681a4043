// Multi-scale sliding-window tiling (front) and aggregation (back) of the relevancy extractor.
// Compiled with -ffp-contract=off: the uint8 resampler is an integer, bit-exact target.
//
// Replaces (reference file:line):
//   ClipWrapper.create_tiles crop + preprocess        CLIP/clip/__init__.py:254-281
//   _transform: Resize(224, BICUBIC) / ToTensor / Normalize   CLIP/clip/clip_explainability.py:98-108
//       (the resize itself is Pillow's ImagingResample: 22-bit fixed-point separable bicubic, uint8 between passes)
//   horizontal flip of the tile batch                  CLIP/clip/__init__.py:171-173
//   VisionTransformer.conv1 im2col (the GEMM A operand) CLIP/clip/model_explainability.py:325-328
//   un-flip average, bilinear upsample, fp16 canvases, count normalise, mean over scales   __init__.py:196-233
#include "semabs_common.h"
#include <math.h>
#include <vector>

#define OUT_RES 224
#define PRECISION_BITS 22
#define KMAX 24            // max taps supported (ksize = 2 * ceil(2 * max(scale, 1)) + 1  ->  scale <= 5.5)

// ------------------------------------------------------------------------------------------------
// Host: Pillow's precompute_coeffs + normalize_coeffs_8bpc for the bicubic filter (a = -0.5).
// Writes xmin[out_size], kk[out_size * KMAX] (zero padded) and *ksize.  Pure host arithmetic in double.
// ------------------------------------------------------------------------------------------------
static double bicubic(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

extern "C" int semabs_resize_coeffs(int in_size, int out_size, int* xmin_out, int* kk_out, int kmax, int* ksize_out) {
    SEMABS_REQUIRE(in_size > 0 && out_size > 0 && xmin_out && kk_out && ksize_out, "semabs_resize_coeffs: bad args");
    double scale = (double)((float)in_size - 0.0f) / out_size;
    double filterscale = scale < 1.0 ? 1.0 : scale;
    double support = 2.0 * filterscale;
    int ksize = (int)ceil(support) * 2 + 1;
    SEMABS_REQUIRE(ksize <= kmax, "semabs_resize_coeffs: down-scale factor too large for KMAX taps");
    std::vector<double> k(ksize);
    for (int xx = 0; xx < out_size; ++xx) {
        double center = 0.0 + (xx + 0.5) * scale;
        double ww = 0.0, ss = 1.0 / filterscale;
        int lo = (int)(center - support + 0.5);
        if (lo < 0) lo = 0;
        int hi = (int)(center + support + 0.5);
        if (hi > in_size) hi = in_size;
        int n = hi - lo;
        for (int x = 0; x < n; ++x) {
            double w = bicubic((x + lo - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < n; ++x)
            if (ww != 0.0) k[x] /= ww;
        for (int x = 0; x < kmax; ++x) {
            int v = 0;
            if (x < n) v = k[x] < 0 ? (int)(-0.5 + k[x] * (1 << PRECISION_BITS)) : (int)(0.5 + k[x] * (1 << PRECISION_BITS));
            kk_out[xx * kmax + x] = v;
        }
        xmin_out[xx] = lo;
    }
    *ksize_out = ksize;
    return SEMABS_OK;
}

// ------------------------------------------------------------------------------------------------
// Tile -> patch matrix.  grid = (n_tiles, 224 / BAND); one workgroup resamples a band of output rows:
//   phase A: horizontal pass of the input rows the band needs -> uint8 in LDS
//   phase B: vertical pass -> uint8 -> (u/255 - mean)/std via a 768-entry fp16 LUT -> im2col store
// tiles: int32 [n, 5] = (image, row0, col0, tile_size, coef_id)
// coef_xmin int32 [n_sizes, 224], coef_kk int32 [n_sizes, 224, KMAX], coef_ksize int32 [n_sizes]
// lut fp16 [3, 256];  patches fp16 [n * g * g, 3 * p * p], column = c * p*p + iy * p + ix
// ------------------------------------------------------------------------------------------------
#define BAND 16
#define MAX_IN_ROWS 64      // input rows a 16-row output band can touch: 16 * scale + 2 * support + 2  (scale <= 2.2)

__device__ __forceinline__ int clamp_u8(int a) { return a < 0 ? 0 : (a > 255 ? 255 : a); }

// Horizontal pass of one band: thread x owns output column x - its taps (weights, byte offsets) sit in registers and it walks the band's input rows two
// at a time, every byte load of a row pair in flight before the first multiply (with a run-time tap loop each tap waited for its own three loads:
// ~3 000 cycles of exposed latency per output).  KS = taps known at compile time (0: generic loop).
template <int KS>
__device__ __forceinline__ void tile_hpass(const unsigned char* __restrict__ src, int W, int ts, int ksize, const int* __restrict__ kx, int x0,
                                           int r_lo, int n_rows, unsigned char (*sh)[OUT_RES][3], int x) {
    if (KS == 0) {
        for (int rr = 0; rr < n_rows; ++rr) {
            const unsigned char* srow = src + (long)(r_lo + rr) * W * 3;
            int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
            for (int k = 0; k < ksize; ++k) {
                int xi = x0 + k; if (xi > ts - 1) xi = ts - 1;        // taps past the edge carry weight 0
                const int w = kx[k];
                a0 += (int)srow[xi * 3] * w; a1 += (int)srow[xi * 3 + 1] * w; a2 += (int)srow[xi * 3 + 2] * w;
            }
            sh[rr][x][0] = (unsigned char)clamp_u8(a0 >> PRECISION_BITS);
            sh[rr][x][1] = (unsigned char)clamp_u8(a1 >> PRECISION_BITS);
            sh[rr][x][2] = (unsigned char)clamp_u8(a2 >> PRECISION_BITS);
        }
        return;
    }
    constexpr int K = KS ? KS : 1;
    int w[K], off[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        int xi = x0 + k; if (xi > ts - 1) xi = ts - 1;
        w[k] = kx[k]; off[k] = xi * 3;
    }
    const long pitch = (long)W * 3;
    const unsigned char* srow = src + (long)r_lo * pitch;
    int rr = 0;
    for (; rr + 2 <= n_rows; rr += 2, srow += 2 * pitch) {
        unsigned char u[2][K][3];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int k = 0; k < K; ++k) {
                u[q][k][0] = srow[q * pitch + off[k]]; u[q][k][1] = srow[q * pitch + off[k] + 1]; u[q][k][2] = srow[q * pitch + off[k] + 2];
            }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
#pragma unroll
            for (int k = 0; k < K; ++k) { a0 += (int)u[q][k][0] * w[k]; a1 += (int)u[q][k][1] * w[k]; a2 += (int)u[q][k][2] * w[k]; }
            sh[rr + q][x][0] = (unsigned char)clamp_u8(a0 >> PRECISION_BITS);
            sh[rr + q][x][1] = (unsigned char)clamp_u8(a1 >> PRECISION_BITS);
            sh[rr + q][x][2] = (unsigned char)clamp_u8(a2 >> PRECISION_BITS);
        }
    }
    if (rr < n_rows) {
        int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
#pragma unroll
        for (int k = 0; k < K; ++k) { a0 += (int)srow[off[k]] * w[k]; a1 += (int)srow[off[k] + 1] * w[k]; a2 += (int)srow[off[k] + 2] * w[k]; }
        sh[rr][x][0] = (unsigned char)clamp_u8(a0 >> PRECISION_BITS);
        sh[rr][x][1] = (unsigned char)clamp_u8(a1 >> PRECISION_BITS);
        sh[rr][x][2] = (unsigned char)clamp_u8(a2 >> PRECISION_BITS);
    }
}

// Vertical pass of one output row for the pixel pair (x, x + 1), x even: the pair's 6 bytes are 3 aligned 16-bit LDS reads per tap.  -> v[2][3]
template <int KS>
__device__ __forceinline__ void tile_vpass(const unsigned short* __restrict__ shs, const int* __restrict__ ky, int ksize, int yb, int n_rows, int x,
                                           int (&v)[2][3]) {
    int a[2][3];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int c = 0; c < 3; ++c) a[q][c] = 1 << (PRECISION_BITS - 1);
    auto tap = [&](int k) {
        int rr = yb + k; if (rr > n_rows - 1) rr = n_rows - 1;
        const int w = ky[k];
        const unsigned short* s = shs + ((rr * OUT_RES + x) >> 1) * 3;
        const int s0 = s[0], s1 = s[1], s2 = s[2];                   // (R0 G0) (B0 R1) (G1 B1)
        a[0][0] += (s0 & 255) * w; a[0][1] += (s0 >> 8) * w; a[0][2] += (s1 & 255) * w;
        a[1][0] += (s1 >> 8) * w; a[1][1] += (s2 & 255) * w; a[1][2] += (s2 >> 8) * w;
    };
    if constexpr (KS == 0) {
        for (int k = 0; k < ksize; ++k) tap(k);
    } else {
#pragma unroll
        for (int k = 0; k < KS; ++k) tap(k);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int c = 0; c < 3; ++c) v[q][c] = clamp_u8(a[q][c] >> PRECISION_BITS);
}

__global__ __launch_bounds__(256) void k_tile_patches(const unsigned char* __restrict__ images, int H, int W,
                                                      const int* __restrict__ tiles, const int* __restrict__ coef_xmin,
                                                      const int* __restrict__ coef_kk, const int* __restrict__ coef_ksize,
                                                      const f16* __restrict__ lut, f16* __restrict__ patches, int p,
                                                      int flip, int n_tiles) {
    __shared__ __attribute__((aligned(16))) unsigned char sh[MAX_IN_ROWS][OUT_RES][3];
    __shared__ int s_xmin[OUT_RES];
    __shared__ int s_ky[BAND][KMAX];
    __shared__ f16 s_lut[3 * 256];
    const int t = blockIdx.x, band = blockIdx.y, tid = threadIdx.x;
    const int img = tiles[t * 5 + 0], row0 = tiles[t * 5 + 1], col0 = tiles[t * 5 + 2], ts = tiles[t * 5 + 3], cid = tiles[t * 5 + 4];
    const int* xmin = coef_xmin + cid * OUT_RES;
    const int* kk = coef_kk + (long)cid * OUT_RES * KMAX;
    const int ksize = coef_ksize[cid];
    const int y_first = band * BAND, y_last = y_first + BAND - 1;
    for (int i = tid; i < OUT_RES; i += 256) s_xmin[i] = xmin[i];
    for (int i = tid; i < BAND * KMAX; i += 256) s_ky[i / KMAX][i % KMAX] = kk[(y_first + i / KMAX) * KMAX + i % KMAX];
    for (int i = tid; i < 3 * 256; i += 256) s_lut[i] = lut[i];
    __syncthreads();
    const int r_lo = s_xmin[y_first];
    int r_hi = s_xmin[y_last] + ksize;          // exclusive
    if (r_hi > ts) r_hi = ts;
    const int n_rows = r_hi - r_lo;
    const unsigned char* src = images + ((long)img * H + row0) * W * 3 + (long)col0 * 3;
    const bool identity = (ts == OUT_RES);
    // ---- phase A: horizontal pass ----
    if (identity) {
        for (int e = tid; e < n_rows * OUT_RES; e += 256) {
            const int rr = e / OUT_RES, x = e - rr * OUT_RES;
            const unsigned char* srow = src + (long)(r_lo + rr) * W * 3;
            sh[rr][x][0] = srow[x * 3]; sh[rr][x][1] = srow[x * 3 + 1]; sh[rr][x][2] = srow[x * 3 + 2];
        }
    } else if (tid < OUT_RES) {
        const int* kx = kk + tid * KMAX;
        const int x0 = s_xmin[tid];
        switch (ksize) {                          // workgroup-uniform
            case 5: tile_hpass<5>(src, W, ts, ksize, kx, x0, r_lo, n_rows, sh, tid); break;
            case 7: tile_hpass<7>(src, W, ts, ksize, kx, x0, r_lo, n_rows, sh, tid); break;
            case 9: tile_hpass<9>(src, W, ts, ksize, kx, x0, r_lo, n_rows, sh, tid); break;
            case 11: tile_hpass<11>(src, W, ts, ksize, kx, x0, r_lo, n_rows, sh, tid); break;
            default: tile_hpass<0>(src, W, ts, ksize, kx, x0, r_lo, n_rows, sh, tid); break;
        }
    }
    __syncthreads();
    // ---- phase B: vertical pass + normalise + im2col; thread = (row group of 8, pixel pair) ----
    if (tid >= OUT_RES) return;
    const int g = OUT_RES / p, pp = p * p, Kp = 3 * pp;
    const int rg = tid / (OUT_RES / 2), x = (tid - rg * (OUT_RES / 2)) * 2;
    const unsigned short* shs = reinterpret_cast<const unsigned short*>(&sh[0][0][0]);
    // flip = 2: the resampled tile is written twice, as is into patches[0 .. n_tiles) and mirrored into patches[n_tiles .. 2 n_tiles)
    const int xf = OUT_RES - 2 - x;               // mirrored pair starts here (even), pixel order swapped
    const int px0 = x / p, ix0 = x - px0 * p, px1 = xf / p, ix1 = xf - px1 * p;
    for (int yy = rg * (BAND / 2); yy < (rg + 1) * (BAND / 2); ++yy) {
        const int y = y_first + yy;
        int v[2][3];
        if (identity) {
            const unsigned short* s = shs + (((y - r_lo) * OUT_RES + x) >> 1) * 3;
            const int s0 = s[0], s1 = s[1], s2 = s[2];
            v[0][0] = s0 & 255; v[0][1] = s0 >> 8; v[0][2] = s1 & 255; v[1][0] = s1 >> 8; v[1][1] = s2 & 255; v[1][2] = s2 >> 8;
        } else {
            const int yb = s_xmin[y] - r_lo;
            switch (ksize) {
                case 5: tile_vpass<5>(shs, s_ky[yy], ksize, yb, n_rows, x, v); break;
                case 7: tile_vpass<7>(shs, s_ky[yy], ksize, yb, n_rows, x, v); break;
                case 9: tile_vpass<9>(shs, s_ky[yy], ksize, yb, n_rows, x, v); break;
                case 11: tile_vpass<11>(shs, s_ky[yy], ksize, yb, n_rows, x, v); break;
                default: tile_vpass<0>(shs, s_ky[yy], ksize, yb, n_rows, x, v); break;
            }
        }
        f16x2 o[3], of[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const f16 l = s_lut[c * 256 + v[0][c]], r = s_lut[c * 256 + v[1][c]];
            o[c][0] = l; o[c][1] = r; of[c][0] = r; of[c][1] = l;
        }
        const int py = y / p, iy = y - py * p;
        if (flip != 1) {
            f16* dst = patches + ((long)t * g * g + py * g + px0) * Kp + iy * p + ix0;
#pragma unroll
            for (int c = 0; c < 3; ++c) *reinterpret_cast<f16x2*>(dst + c * pp) = o[c];
        }
        if (flip != 0) {
            f16* dst = patches + ((long)(flip == 2 ? t + n_tiles : t) * g * g + py * g + px1) * Kp + iy * p + ix1;
#pragma unroll
            for (int c = 0; c < 3; ++c) *reinterpret_cast<f16x2*>(dst + c * pp) = of[c];
        }
    }
}

extern "C" int semabs_tile_patches(const unsigned char* images, int n_img, int H, int W, const int* tiles_dev, int n_tiles,
                                   const int* coef_xmin, const int* coef_kk, const int* coef_ksize, const void* lut,
                                   void* patches, int patch, int flip, int max_ksize, void* stream) {
    if (n_tiles == 0) return SEMABS_OK;
    SEMABS_REQUIRE(images && tiles_dev && coef_xmin && coef_kk && coef_ksize && lut && patches, "semabs_tile_patches: null pointer");
    SEMABS_REQUIRE(n_img > 0 && H > 0 && W > 0 && (patch == 16 || patch == 32 || patch == 14), "semabs_tile_patches: bad shape / patch size");
    SEMABS_REQUIRE(flip >= 0 && flip <= 2, "semabs_tile_patches: flip must be 0 (as is), 1 (mirrored) or 2 (both: patches holds 2 n_tiles tiles)");
    SEMABS_REQUIRE(OUT_RES % patch == 0, "semabs_tile_patches: patch must divide 224");
    // a 16-row band touches at most 16 * scale + ksize input rows; ksize = 2 ceil(2 scale) + 1
    SEMABS_REQUIRE(max_ksize <= KMAX && (max_ksize - 1) / 4.0 * BAND + max_ksize + 2 <= MAX_IN_ROWS,
                   "semabs_tile_patches: tile_size / 224 too large for the LDS band (tile_size must be <= ~490)");
    hipLaunchKernelGGL(k_tile_patches, dim3(n_tiles, OUT_RES / BAND), dim3(256), 0, (hipStream_t)stream, images, H, W, tiles_dev,
                       coef_xmin, coef_kk, coef_ksize, (const f16*)lut, (f16*)patches, patch, flip, n_tiles);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// fp32 tile images [n, 3, 224, 224] (already preprocessed, e.g. by a caller of ClipGradcam.forward) -> patch matrix
__global__ void k_patchify(const float* __restrict__ x, f16* __restrict__ patches, int n, int p, int flip) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long tot = (long)n * 3 * OUT_RES * OUT_RES;
    if (i >= tot) return;
    int xq = (int)(i % OUT_RES); long r = i / OUT_RES;
    int y = (int)(r % OUT_RES); r /= OUT_RES;
    int c = (int)(r % 3); long t = r / 3;
    const int g = OUT_RES / p, pp = p * p;
    const int xo = flip ? OUT_RES - 1 - xq : xq;
    const int py = y / p, iy = y % p, px = xo / p, ix = xo % p;
    patches[((long)t * g * g + py * g + px) * 3 * pp + c * pp + iy * p + ix] = (f16)x[i];
}
extern "C" int semabs_patchify(const float* x, void* patches, int n, int patch, int flip, void* stream) {
    if (n == 0) return SEMABS_OK;
    SEMABS_REQUIRE(x && patches && n > 0 && OUT_RES % patch == 0, "semabs_patchify: bad args");
    hipLaunchKernelGGL(k_patchify, dim3(semabs_cdiv((long)n * 3 * OUT_RES * OUT_RES, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       (f16*)patches, n, patch, flip);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// ------------------------------------------------------------------------------------------------
// Aggregation, gather form: one thread per (label, pixel) walks the tiles that cover the pixel in the
// reference's accumulation order (image -> column start -> row start), bilinearly samples each tile's g x g map
// (align_corners = False), adds into an fp16 accumulator per scale (rounded after every add, as the reference's
// half canvases are), then sum_s (acc_s / count_s) / n_scales.  No atomics; deterministic.
//   rel / rel_flip fp32 [L, N, g, g] (tile order = the tile table's); rel_flip may be null (no flip pass)
//   scales int32 [n_scales, 5] = (tile_size, stride, n_row_starts, n_col_starts, first tile index within an image)
// ------------------------------------------------------------------------------------------------
struct AggArgs {
    int L, N, g, H, W, n_img, tiles_per_img, n_scales;
};

__device__ __forceinline__ void bil_setup(int d, float scale, int g, int& i0, int& i1, float& l0, float& l1) {      // scale = (float)g / (float)tile size
    float s = scale * ((float)d + 0.5f) - 0.5f;
    if (s < 0.f) s = 0.f;
    i0 = (int)s;
    if (i0 > g - 1) i0 = g - 1;
    i1 = i0 + (i0 < g - 1 ? 1 : 0);
    l1 = s - (float)i0;
    l0 = 1.f - l1;
}

__global__ __launch_bounds__(256) void k_aggregate(const float* __restrict__ rel, const float* __restrict__ rel_flip,
                                                   const int* __restrict__ scales, AggArgs a, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long tot = (long)a.L * a.H * a.W;
    if (i >= tot) return;
    const int c = (int)(i % a.W); const long t = i / a.W;
    const int r = (int)(t % a.H); const int l = (int)(t / a.H);
    const int g = a.g, gg = g * g;
    float total = 0.f;
    for (int s = 0; s < a.n_scales; ++s) {
        const int ts = scales[s * 5], stride = scales[s * 5 + 1], nx = scales[s * 5 + 2], ny = scales[s * 5 + 3], base = scales[s * 5 + 4];
        // covering row starts x0 = ix * stride with x0 <= r < x0 + ts
        int ix_hi = r / stride; if (ix_hi > nx - 1) ix_hi = nx - 1;
        int ix_lo = r - ts + 1 <= 0 ? 0 : (r - ts + 1 + stride - 1) / stride;
        int iy_hi = c / stride; if (iy_hi > ny - 1) iy_hi = ny - 1;
        int iy_lo = c - ts + 1 <= 0 ? 0 : (c - ts + 1 + stride - 1) / stride;
        __half acc = __float2half(0.f);
        float cnt = 1e-5f;
        const float bscale = (float)g / (float)ts;       // one division per scale, not per covering tile (~300 per pixel)
        for (int im = 0; im < a.n_img; ++im) {
            for (int iy = iy_lo; iy <= iy_hi; ++iy) {
                int w0, w1; float lw0, lw1;
                bil_setup(c - iy * stride, bscale, g, w0, w1, lw0, lw1);
                for (int ix = ix_lo; ix <= ix_hi; ++ix) {
                    int h0, h1; float lh0, lh1;
                    bil_setup(r - ix * stride, bscale, g, h0, h1, lh0, lh1);
                    const long tile = (long)im * a.tiles_per_img + base + iy * nx + ix;
                    const float* p = rel + ((long)l * a.N + tile) * gg;
                    float v00 = p[h0 * g + w0], v01 = p[h0 * g + w1], v10 = p[h1 * g + w0], v11 = p[h1 * g + w1];
                    if (rel_flip) {
                        const float* q = rel_flip + ((long)l * a.N + tile) * gg;
                        v00 = (v00 + q[h0 * g + (g - 1 - w0)]) / 2; v01 = (v01 + q[h0 * g + (g - 1 - w1)]) / 2;
                        v10 = (v10 + q[h1 * g + (g - 1 - w0)]) / 2; v11 = (v11 + q[h1 * g + (g - 1 - w1)]) / 2;
                    }
                    const float val = lh0 * (lw0 * v00 + lw1 * v01) + lh1 * (lw0 * v10 + lw1 * v11);
                    acc = __float2half(__half2float(acc) + val);
                    cnt += 1.0f;
                }
            }
        }
        total = total + __half2float(acc) / cnt;
    }
    out[i] = total / (float)a.n_scales;
}

extern "C" int semabs_aggregate(const float* rel, const float* rel_flip, int L, int N, int g, int H, int W,
                                const int* scales_dev, int n_scales, int n_img, int tiles_per_img, float* out, void* stream) {
    if (L == 0) return SEMABS_OK;
    SEMABS_REQUIRE(rel && scales_dev && out && L > 0 && N > 0 && g > 0 && H > 0 && W > 0 && n_scales > 0 && n_img > 0,
                   "semabs_aggregate: bad args");
    SEMABS_REQUIRE((long)n_img * tiles_per_img == N, "semabs_aggregate: tile count does not match n_img * tiles_per_img");
    AggArgs a{L, N, g, H, W, n_img, tiles_per_img, n_scales};
    hipLaunchKernelGGL(k_aggregate, dim3(semabs_cdiv((long)L * H * W, 256)), dim3(256), 0, (hipStream_t)stream, rel, rel_flip,
                       scales_dev, a, out);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// un-flip average of the two passes (CLIP/clip/__init__.py:196-204): out[m, h, w] = (rel[m, h, w] + rel_flip[m, h, g - 1 - w]) / 2, once per map cell.
// `semabs_aggregate` with rel_flip does the same average per covered PIXEL per covering tile (8 loads per tile instead of 4, ~500 covering tiles per
// pixel at the headline shape): callers that have both passes average first and aggregate the result - same operation on the same operands, bit-identical.
__global__ void k_unflip_average(const float* __restrict__ rel, const float* __restrict__ rel_flip, float* __restrict__ out, long n, int g,
                                 long per, long in_stride) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int w = (int)(i % g);
    const long l = i / per, src = l * in_stride + (i - l * per);      // (per == in_stride: src == i)
    out[i] = (rel[src] + rel_flip[src - w + (g - 1 - w)]) / 2;
}
extern "C" int semabs_unflip_average(const float* rel, const float* rel_flip, float* out, long n_maps, int g, void* stream) {
    if (n_maps == 0) return SEMABS_OK;
    SEMABS_REQUIRE(rel && rel_flip && out && n_maps > 0 && g > 0, "semabs_unflip_average: bad args");
    const long n = n_maps * g * g;
    hipLaunchKernelGGL(k_unflip_average, dim3(semabs_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, rel, rel_flip, out, n, g, n, n);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}
// The same average for L label rows of maps_per_label maps each whose INPUT rows are in_label_stride maps apart (both passes of a call live in one
// [L, 2 N, g, g] buffer: rel = its first N maps of every label, rel_flip = rel + N maps): out [L, maps_per_label, g, g] contiguous - no slice copies.
extern "C" int semabs_unflip_average_rows(const float* rel, const float* rel_flip, float* out, int L, long maps_per_label, long in_label_stride,
                                          int g, void* stream) {
    if (L == 0 || maps_per_label == 0) return SEMABS_OK;
    SEMABS_REQUIRE(rel && rel_flip && out && L > 0 && maps_per_label > 0 && in_label_stride >= maps_per_label && g > 0, "semabs_unflip_average_rows: bad args");
    const long per = maps_per_label * g * g, n = (long)L * per;
    hipLaunchKernelGGL(k_unflip_average, dim3(semabs_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, rel, rel_flip, out, n, g, per, in_label_stride * g * g);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// ------------------------------------------------------------------------------------------------
// Colour jitter for the augmentation copies (ClipWrapper.jittering_transforms = ColorJitter(0.6, 0.6, 0.6, 0.1) on the PIL image,
// CLIP/clip/__init__.py:55-57, 246-247).  The reference's DRAW is random (torch's global generator), so (order, factors) are caller
// arguments; the ARITHMETIC is pinned: torchvision 0.13.1's PIL path = Pillow's ImageEnhance.Brightness / Contrast / Color
// (Image.blend(degenerate, image, factor): fp32 d + alpha (x - d), truncated, clipped when alpha is outside [0, 1]; degenerate = black /
// the rounded mean of convert("L") / convert("L") itself, L = ITU-R 601-2 luma in 16.16 fixed point) and adjust_hue (convert("HSV") with
// uint8-quantised H, S, V - Convert.c rgb2hsv_row / hsv2rgb, float / double steps as in the C source - and H += uint8(hue_factor * 255)
// modulo 256).  Byte-exact against Pillow: tests/golden/g28_color_jitter.npz (all 24 op orders at 480 x 480), oracle/preprocess.py
// color_jitter (checked on all 2^24 colours), tests/test_gpu_relevancy.py::test_color_jitter_*.  This file is built with -ffp-contract=off.
//   op: 0 brightness, 1 contrast (needs `mean` = rounded mean of the grey image), 2 saturation, 3 hue   (torchvision's fn_idx)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int grey_u8(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }
__device__ __forceinline__ unsigned char blend_u8(float a, float b, float f) {
    float t = a + f * (b - a);
    return (unsigned char)(t <= 0.f ? 0.f : (t >= 255.f ? 255.f : t));
}
__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
// Pillow rgb2hsv_row: s, rc, gc, bc in fp32 (IEEE division); h = 2 + rc - bc / 4 + gc - rc in double rounded to fp32 (bc - gc: fp32);
// h = (float)fmod(h / 6.0 + 1.0, 1.0); H = (int)(h * 255.0), S = (int)(s * 255.0) (double products, truncation), V = max
__device__ __forceinline__ void rgb2hsv_u8(int r, int g, int b, int& uh, int& us, int& uv) {
    const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
    uv = maxc;
    if (minc == maxc) { uh = 0; us = 0; return; }
    const float cr = (float)(maxc - minc);
    const float s = __fdiv_rn(cr, (float)maxc);
    const float rc = __fdiv_rn((float)(maxc - r), cr), gc = __fdiv_rn((float)(maxc - g), cr), bc = __fdiv_rn((float)(maxc - b), cr);
    float h;
    if (r == maxc) h = __fsub_rn(bc, gc);
    else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
    else h = (float)(4.0 + (double)gc - (double)rc);
    h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
    uh = clip8((int)((double)h * 255.0));
    us = clip8((int)((double)s * 255.0));
}
// Pillow hsv2rgb: i = floor(h * 6 / 255) (double), f = fp32 remainder, fs = (float)(s / 255.0); p / q / t = round-half-away(v * (1 - ...)) in double
__device__ __forceinline__ void hsv2rgb_u8(int h, int s, int v, int& r, int& g, int& b) {
    if (s == 0) { r = g = b = v; return; }
    const double hd = (double)(float)h * 6.0 / 255.0;
    const int i = (int)floor(hd);
    const double f = (double)(float)(hd - (double)(float)i);
    const double fs = (double)(float)((double)(float)s / 255.0);
    const double vf = (double)(float)v;
    const int p = clip8((int)round(vf * (1.0 - fs)));
    const int q = clip8((int)round(vf * (1.0 - fs * f)));
    const int t = clip8((int)round(vf * (1.0 - fs * (1.0 - f))));
    switch (i % 6) {
        case 0: r = v; g = t; b = p; break;
        case 1: r = q; g = v; b = p; break;
        case 2: r = p; g = v; b = t; break;
        case 3: r = p; g = q; b = v; break;
        case 4: r = t; g = p; b = v; break;
        default: r = v; g = p; b = q; break;
    }
}

__global__ void k_grey_sum(const unsigned char* __restrict__ img, long n, unsigned long long* __restrict__ sum) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    float v = 0.f;
    if (i < n) v = (float)grey_u8(img[i * 3], img[i * 3 + 1], img[i * 3 + 2]);
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) atomicAdd(sum, (unsigned long long)(v + 0.5f));
}

__global__ void k_jitter_op(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, long n, int op, float f,
                            const unsigned long long* __restrict__ grey_sum) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int r = src[i * 3], g = src[i * 3 + 1], b = src[i * 3 + 2];
    if (op == 0) {
        r = blend_u8(0.f, r, f); g = blend_u8(0.f, g, f); b = blend_u8(0.f, b, f);
    } else if (op == 1) {
        float m = (float)(int)((double)(*grey_sum) / (double)n + 0.5);
        r = blend_u8(m, r, f); g = blend_u8(m, g, f); b = blend_u8(m, b, f);
    } else if (op == 2) {
        float m = (float)grey_u8(r, g, b);
        r = blend_u8(m, r, f); g = blend_u8(m, g, f); b = blend_u8(m, b, f);
    } else {
        // f carries the hue shift already reduced to uint8 by the host: (int)(hue_factor * 255) mod 256 (torchvision: np.uint8(hue_factor * 255))
        int uh, us, uv;
        rgb2hsv_u8(r, g, b, uh, us, uv);
        uh = (uh + (int)f) & 255;
        hsv2rgb_u8(uh, us, uv, r, g, b);
    }
    dst[i * 3] = (unsigned char)r; dst[i * 3 + 1] = (unsigned char)g; dst[i * 3 + 2] = (unsigned char)b;
}

// ONE adjustment, in place.  img uint8 [H, W, 3]; op as above; factor = the torchvision factor (hue: in [-0.5, 0.5]); scratch = 8 bytes of device memory
extern "C" int semabs_color_jitter_op(unsigned char* img, int H, int W, int op, float factor, void* scratch8, void* stream) {
    SEMABS_REQUIRE(img && scratch8 && H > 0 && W > 0, "semabs_color_jitter_op: bad args");
    SEMABS_REQUIRE(op >= 0 && op < 4, "semabs_color_jitter_op: op must be 0..3");
    hipStream_t s = (hipStream_t)stream;
    const long n = (long)H * W;
    float f = factor;
    if (op == 1) {
        semabs_fill32(scratch8, 8, 0u, s);
        hipLaunchKernelGGL(k_grey_sum, dim3(semabs_cdiv(n, 256)), dim3(256), 0, s, img, n, (unsigned long long*)scratch8);
    } else if (op == 3) {
        SEMABS_REQUIRE(f >= -0.5f && f <= 0.5f, "semabs_color_jitter_op: hue factor must lie in [-0.5, 0.5] (torchvision adjust_hue)");
        f = (float)(((int)((double)f * 255.0) % 256 + 256) % 256);      // np.uint8(hue_factor * 255): truncation toward zero, uint8 wrap-around
    }
    hipLaunchKernelGGL(k_jitter_op, dim3(semabs_cdiv(n, 256)), dim3(256), 0, s, img, img, n, op, f, (const unsigned long long*)scratch8);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// img uint8 [H, W, 3] jittered IN PLACE by the four adjustments in the order order4 (a permutation of 0..3), factors4[op] each (ColorJitter.forward);
// order4 / factors4 are host arrays; scratch = 8 bytes of device memory
extern "C" int semabs_color_jitter(unsigned char* img, int H, int W, const int* order4, const float* factors4,
                                   void* scratch8, void* stream) {
    SEMABS_REQUIRE(img && order4 && factors4 && scratch8 && H > 0 && W > 0, "semabs_color_jitter: bad args");
    for (int k = 0; k < 4; ++k) {
        SEMABS_REQUIRE(order4[k] >= 0 && order4[k] < 4, "semabs_color_jitter: op must be 0..3");
        const int rc = semabs_color_jitter_op(img, H, W, order4[k], factors4[order4[k]], scratch8, stream);
        if (rc != SEMABS_OK) return rc;
    }
    return SEMABS_OK;
}
