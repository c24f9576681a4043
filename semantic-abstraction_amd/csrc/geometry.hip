// Geometry kernels: depth -> world points, bounds mask, voxel indices, TSDF integration, frustum mask.
// All HBM-bound elementwise work; integer outputs are bit-exact targets, so this file is compiled with
// -ffp-contract=off and every multiply-add that the CPU path fuses (BLAS dgemm k-loops) is an explicit fma().
//
// Replaces (reference file:line):
//   get_pointcloud / transform_pointcloud      point_cloud.py:34-66, 8-21
//   filter_pts_bounds                          point_cloud.py:24-31
//   VirtualGrid.get_points_grid_idxs/flatten   net.py:84-133
//   TSDFVolume.vox2world/cam2pix/integrate     fusion.py:85-195
//   check_pts_in_frustum                       point_cloud.py:88-110
#include "semabs_common.h"

// ------------------------------------------------------------------------------------------------
// depth [H, W] fp32 -> xyz fp32 [H*W, 3] (f64 math, cast at the end as visualize.py:103-105 does) and the
// inclusive AABB mask of the fp32 points against f64 bounds.
// intr = {fx, fy, cx, cy}; pose = row-major 3x4 [R | t]; bounds = {lo[3], hi[3]} (may be null with mask).
__global__ void k_pointcloud(const float* __restrict__ depth, int H, int W, const double* __restrict__ prm,
                             int has_pose, float* __restrict__ xyz, unsigned char* __restrict__ mask, double* __restrict__ xyz64) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)H * W) return;
    const double fx = prm[0], fy = prm[1], cx = prm[2], cy = prm[3];
    int v = (int)(i / W), u = (int)(i - (long)v * W);
    double d = (double)depth[i];
    double x = ((double)u - cx) * (d / fx);
    double y = ((double)v - cy) * (d / fy);
    double z = d;
    if (has_pose) {
        const double* P = prm + 4;
        // np.dot(R, pts.T): dgemm k-loop = fma chain in k order; translation added afterwards
        double wx = fma(P[2], z, fma(P[1], y, P[0] * x)) + P[3];
        double wy = fma(P[6], z, fma(P[5], y, P[4] * x)) + P[7];
        double wz = fma(P[10], z, fma(P[9], y, P[8] * x)) + P[11];
        x = wx; y = wy; z = wz;
    }
    if (xyz64) { xyz64[i * 3 + 0] = x; xyz64[i * 3 + 1] = y; xyz64[i * 3 + 2] = z; }     // the reference's own return type (point_cloud.py:51-66)
    float fx32 = (float)x, fy32 = (float)y, fz32 = (float)z;
    if (xyz) { xyz[i * 3 + 0] = fx32; xyz[i * 3 + 1] = fy32; xyz[i * 3 + 2] = fz32; }
    if (mask) {
        const double* B = prm + 16;
        bool m = (double)fx32 >= B[0] && (double)fx32 <= B[3] && (double)fy32 >= B[1] && (double)fy32 <= B[4] &&
                 (double)fz32 >= B[2] && (double)fz32 <= B[5];
        mask[i] = m ? 1 : 0;
    }
}

extern "C" int semabs_pointcloud(const float* depth, int H, int W, const double* params_dev, int has_pose,
                                 float* xyz, unsigned char* mask, void* stream) {
    SEMABS_REQUIRE(depth && params_dev && xyz && H > 0 && W > 0, "semabs_pointcloud: bad args");
    long n = (long)H * W;
    hipLaunchKernelGGL(k_pointcloud, dim3(semabs_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, depth, H, W,
                       params_dev, has_pose, xyz, mask, (double*)nullptr);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// The same unprojection returning the f64 points themselves - `get_pointcloud`'s return type in the reference (its on-path caller casts to
// fp32 at once, visualize.py:103-105, which is what semabs_pointcloud hands out; other callers get the full-precision values here).
extern "C" int semabs_pointcloud_f64(const float* depth, int H, int W, const double* params_dev, int has_pose, double* xyz64, void* stream) {
    SEMABS_REQUIRE(depth && params_dev && xyz64 && H > 0 && W > 0, "semabs_pointcloud_f64: bad args");
    long n = (long)H * W;
    hipLaunchKernelGGL(k_pointcloud, dim3(semabs_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, depth, H, W,
                       params_dev, has_pose, (float*)nullptr, (unsigned char*)nullptr, xyz64);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// ------------------------------------------------------------------------------------------------
// pts fp32 [N, 3] -> flat int64 [N] = ix*S1*S2 + iy*S2 + iz with i = clamp(trunc((p + off) * sc), 0, S-1).
// off/sc are the fp32 constants the host forms exactly as net.py:91-95 does.
__global__ void k_voxel_index(const float* __restrict__ pts, long N, float ox, float oy, float oz, float sx,
                              float sy, float sz, int S0, int S1, int S2, long long* __restrict__ flat,
                              int* __restrict__ idx3) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float px = pts[i * 3 + 0], py = pts[i * 3 + 1], pz = pts[i * 3 + 2];
    float fx = (px + ox) * sx, fy = (py + oy) * sy, fz = (pz + oz) * sz;   // two roundings each (no contraction)
    long long ix = (long long)fx, iy = (long long)fy, iz = (long long)fz;  // trunc toward zero
    ix = ix < 0 ? 0 : (ix > S0 - 1 ? S0 - 1 : ix);
    iy = iy < 0 ? 0 : (iy > S1 - 1 ? S1 - 1 : iy);
    iz = iz < 0 ? 0 : (iz > S2 - 1 ? S2 - 1 : iz);
    if (flat) flat[i] = ix * (long long)S1 * S2 + iy * S2 + iz;
    if (idx3) { idx3[i * 3] = (int)ix; idx3[i * 3 + 1] = (int)iy; idx3[i * 3 + 2] = (int)iz; }
}

extern "C" int semabs_voxel_index(const float* pts, long N, const float* off3, const float* scale3,
                                  const int* shape3, long long* flat, int* idx3, void* stream) {
    if (N == 0) return SEMABS_OK;   // empty point set: nothing to do (pointers may be null)
    SEMABS_REQUIRE(pts && off3 && scale3 && shape3 && (flat || idx3) && N > 0, "semabs_voxel_index: bad args");
    hipLaunchKernelGGL(k_voxel_index, dim3(semabs_cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, pts, N, off3[0],
                       off3[1], off3[2], scale3[0], scale3[1], scale3[2], shape3[0], shape3[1], shape3[2], flat, idx3);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// ------------------------------------------------------------------------------------------------
// One fused TSDF integration step over a [D0, D1, D2] volume (fusion.py:121-195).
// prm (device, doubles): [0..11] inv(pose) rows 0-2 (3x4), [12] voxel_size, [13] trunc_margin, [14] obs_weight
// origin/intr passed by value as fp32 (the reference casts both to fp32).
struct TsdfArgs {
    float ox, oy, oz;
    float fx, fy, cx, cy;
    int D0, D1, D2, H, W;
};

__global__ void k_tsdf_integrate(const unsigned char* __restrict__ color, const float* __restrict__ depth,
                                 const double* __restrict__ prm, TsdfArgs a, float* __restrict__ tsdf,
                                 float* __restrict__ weight, float* __restrict__ colvol,
                                 long long* __restrict__ pix_out) {
    long n = (long)a.D0 * a.D1 * a.D2;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int k = (int)(i % a.D2);
    int j = (int)((i / a.D2) % a.D1);
    int ii = (int)(i / ((long)a.D2 * a.D1));
    const double vs = prm[12], trunc = prm[13], ow = prm[14];
    // vox2world: f32(f64(origin) + f64(vs) * f64(f32(idx)))
    float wx = (float)((double)a.ox + vs * (double)(float)ii);
    float wy = (float)((double)a.oy + vs * (double)(float)j);
    float wz = (float)((double)a.oz + vs * (double)(float)k);
    // rigid_transform: np.dot(T, [x y z 1]^T) in f64 (dgemm k-loop = fma chain)
    double X = wx, Y = wy, Z = wz;
    double cxm = fma(prm[3], 1.0, fma(prm[2], Z, fma(prm[1], Y, prm[0] * X)));
    double cym = fma(prm[7], 1.0, fma(prm[6], Z, fma(prm[5], Y, prm[4] * X)));
    double czm = fma(prm[11], 1.0, fma(prm[10], Z, fma(prm[9], Y, prm[8] * X)));
    // cam2pix: round-half-even of x * fx / z + cx with fp32 intrinsics promoted to f64
    double rx = rint(cxm * (double)a.fx / czm + (double)a.cx);
    double ry = rint(cym * (double)a.fy / czm + (double)a.cy);
    bool finite = isfinite(rx) && isfinite(ry);
    long long px = finite ? (long long)rx : (long long)0x8000000000000000LL;
    long long py = finite ? (long long)ry : (long long)0x8000000000000000LL;
    if (pix_out) { pix_out[i * 2] = px; pix_out[i * 2 + 1] = py; }
    bool valid_pix = finite && px >= 0 && px < a.W && py >= 0 && py < a.H && czm > 0.0;
    double depth_val = valid_pix ? (double)depth[py * a.W + px] : 0.0;
    double diff = depth_val - czm;
    bool valid = depth_val > 0.0 && diff >= -trunc;
    if (!valid) return;
    double dist = fmax(-1.0, fmin(1.0, diff / trunc));
    float w_old = weight[i], t_old = tsdf[i];
    float w_new = (float)((double)w_old + ow);
    float t_new = (float)(((double)(w_old * t_old) + ow * dist) / (double)w_new);
    weight[i] = w_new;
    tsdf[i] = t_new;
    if (colvol && color) {
        const float cc = 65536.0f;
        const unsigned char* c = color + ((long)py * a.W + px) * 3;
        float nw = floorf((float)c[2] * cc + (float)c[1] * 256.0f + (float)c[0]);
        float old = colvol[i];
        float ob = floorf(old / cc), og = floorf((old - ob * cc) / 256.0f), orr = old - ob * cc - og * 256.0f;
        float nb = floorf(nw / cc), ng = floorf((nw - nb * cc) / 256.0f), nr = nw - nb * cc - ng * 256.0f;
        float owf = (float)ow;
        nb = fminf(255.0f, rintf((w_old * ob + owf * nb) / w_new));
        ng = fminf(255.0f, rintf((w_old * og + owf * ng) / w_new));
        nr = fminf(255.0f, rintf((w_old * orr + owf * nr) / w_new));
        colvol[i] = nb * cc + ng * 256.0f + nr;
    }
}

extern "C" int semabs_tsdf_integrate(const unsigned char* color, const float* depth, int H, int W,
                                     const double* params_dev, const float* origin3, const float* intr4,
                                     const int* dims3, float* tsdf, float* weight, float* colvol,
                                     long long* pix_out, void* stream) {
    SEMABS_REQUIRE(depth && params_dev && origin3 && intr4 && dims3 && tsdf && weight, "semabs_tsdf_integrate: bad args");
    TsdfArgs a;
    a.ox = origin3[0]; a.oy = origin3[1]; a.oz = origin3[2];
    a.fx = intr4[0]; a.fy = intr4[1]; a.cx = intr4[2]; a.cy = intr4[3];
    a.D0 = dims3[0]; a.D1 = dims3[1]; a.D2 = dims3[2]; a.H = H; a.W = W;
    long n = (long)a.D0 * a.D1 * a.D2;
    SEMABS_REQUIRE(n > 0 && H > 0 && W > 0, "semabs_tsdf_integrate: empty volume or image");
    hipLaunchKernelGGL(k_tsdf_integrate, dim3(semabs_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, color, depth,
                       params_dev, a, tsdf, weight, colvol, pix_out);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// ------------------------------------------------------------------------------------------------
// pts f64 [M, 3] -> in-frustum mask (float pixel coordinates, no rounding) (point_cloud.py:88-110)
// prm: [0..11] inv(pose) 3x4, [12] fx, [13] fy, [14] cx, [15] cy (all f64, from the f64 cam_intr)
__global__ void k_frustum(const double* __restrict__ pts, long M, const double* __restrict__ prm, int H, int W,
                          unsigned char* __restrict__ mask) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    double x = pts[i * 3], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
    double cx_ = fma(prm[2], z, fma(prm[1], y, prm[0] * x)) + prm[3];
    double cy_ = fma(prm[6], z, fma(prm[5], y, prm[4] * x)) + prm[7];
    double cz_ = fma(prm[10], z, fma(prm[9], y, prm[8] * x)) + prm[11];
    double px = (prm[12] / cz_) * cx_ + prm[14];
    double py = (prm[13] / cz_) * cy_ + prm[15];
    mask[i] = (px >= 0 && px < (double)W && py >= 0 && py < (double)H && cz_ > 0) ? 1 : 0;
}

extern "C" int semabs_frustum_mask(const double* pts, long M, const double* params_dev, int H, int W,
                                   unsigned char* mask, void* stream) {
    if (M == 0) return SEMABS_OK;
    SEMABS_REQUIRE(pts && params_dev && mask && M > 0, "semabs_frustum_mask: bad args");
    hipLaunchKernelGGL(k_frustum, dim3(semabs_cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream, pts, M, params_dev, H,
                       W, mask);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// ------------------------------------------------------------------------------------------------
// In-bounds compaction + seeded sub-sample, entirely on the device (visualize.py:103-108 `input_xyz_pts[in_bounds_mask]` and :193
// `np.random.choice(len(pts), size=num_input_pts)`): no host round trip for the point count.
//   pix[0 .. n_in) = ascending indices i with mask[i] != 0   (what torch.nonzero / boolean indexing produce)
//   sel[j] = pix[ mulhi64( splitmix64(seed * 0x9E3779B97F4A7C15 + j), n_in ) ]        a uniform draw with replacement, counter-based:
//            reproducible from (seed, j, n_in) alone - oracle/scene.py:subsample_indices restates it
// Two small launches compact the mask (H * W = 230 400 at the BASELINE shape), a third draws.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// Two launches of 256-thread workgroups over 4 096-byte blocks of the mask (one 16-byte load per thread): per-block counts, then every block
// sums the counts in front of it (<= a few hundred), scans its own threads and writes its indices - ascending, like torch.nonzero.
// (Round 3's first version was ONE 1 024-thread workgroup with byte loads: 294 us for 230 400 pixels; this one: a few microseconds.)
__device__ __forceinline__ int mask16_count(const uint4 v, unsigned& bits) {
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    bits = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int b = 0; b < 4; ++b) bits |= (((w[q] >> (8 * b)) & 0xffu) ? 1u : 0u) << (q * 4 + b);
    return __popc(bits);
}
__device__ __forceinline__ uint4 mask16_load(const unsigned char* mask, long n, long base) {
    if (base + 16 <= n) return *reinterpret_cast<const uint4*>(mask + base);        // (mask is 16-byte aligned: a torch allocation)
    unsigned w[4] = {0, 0, 0, 0};
    for (int e = 0; e < 16; ++e) if (base + e < n && mask[base + e]) w[e >> 2] |= 1u << (8 * (e & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}
__global__ __launch_bounds__(256) void k_compact_count(const unsigned char* __restrict__ mask, long n, int* __restrict__ blk_cnt) {
    const long base = ((long)blockIdx.x * 256 + threadIdx.x) * 16;
    unsigned bits;
    int c = base < n ? mask16_count(mask16_load(mask, n, base), bits) : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    __shared__ int ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blk_cnt[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
__global__ __launch_bounds__(256) void k_compact_write(const unsigned char* __restrict__ mask, long n, const int* __restrict__ blk_cnt, int nblk,
                                                       long long* __restrict__ pix, long long* __restrict__ n_out) {
    __shared__ int ws[4];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int pre = 0;                                             // counts of the blocks in front of this one
    for (int b = tid; b < (int)blockIdx.x; b += 256) pre += blk_cnt[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pre += __shfl_xor(pre, o, 64);
    if (lane == 0) ws[w] = pre;
    __syncthreads();
    if (tid == 0) s_base = ws[0] + ws[1] + ws[2] + ws[3];
    __syncthreads();
    const long base = ((long)blockIdx.x * 256 + tid) * 16;
    unsigned bits = 0;
    const int cnt = base < n ? mask16_count(mask16_load(mask, n, base), bits) : 0;
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
    __syncthreads();
    if (lane == 63) ws[w] = incl;
    __syncthreads();
    int woff = 0;
    for (int q = 0; q < w; ++q) woff += ws[q];
    long pos = (long)s_base + woff + incl - cnt;
    for (int e = 0; e < 16; ++e) if (bits & (1u << e)) pix[pos++] = base + e;
    if ((int)blockIdx.x == nblk - 1 && tid == 255) *n_out = (long long)s_base + woff + incl;
}
__global__ void k_subsample(const long long* __restrict__ pix, const long long* __restrict__ n_in, unsigned long long seed, long num, long long* __restrict__ sel) {
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= num) return;
    const unsigned long long n = (unsigned long long)*n_in;
    const unsigned long long r = splitmix64(seed * 0x9E3779B97F4A7C15ull + (unsigned long long)j);
    sel[j] = n ? pix[__umul64hi(r, n)] : 0;
}
// mask uint8 [n]; pix int64 [n] scratch (first *n_in entries valid afterwards); n_in int64 [1] on the device; sel int64 [num]
// mask uint8 [n] (16-byte aligned); pix int64 [n] (the first *n_in entries are valid afterwards); n_in int64 [1] on the device; sel int64 [num],
// num >= 1.  The per-block counts (ceil(n / 4096) ints) are parked in `sel` between the first two launches - k_subsample overwrites it last.
extern "C" int semabs_compact_subsample(const unsigned char* mask, long n, unsigned long long seed, long num, long long* pix, long long* n_in,
                                        long long* sel, void* stream) {
    SEMABS_REQUIRE(mask && pix && n_in && sel && n > 0 && num > 0, "semabs_compact_subsample: bad args");
    SEMABS_REQUIRE((reinterpret_cast<uintptr_t>(mask) & 15) == 0, "semabs_compact_subsample: mask must be 16-byte aligned");
    const int nblk = semabs_cdiv(n, 4096);
    SEMABS_REQUIRE(num * 2 >= nblk, "semabs_compact_subsample: num must be >= ceil(n / 4096) / 2 (sel doubles as the block-count scratch)");
    int* blk_cnt = reinterpret_cast<int*>(sel);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_compact_count, dim3(nblk), dim3(256), 0, s, mask, n, blk_cnt);
    hipLaunchKernelGGL(k_compact_write, dim3(nblk), dim3(256), 0, s, mask, n, blk_cnt, nblk, pix, n_in);
    hipLaunchKernelGGL(k_subsample, dim3(semabs_cdiv(num, 256)), dim3(256), 0, s, pix, n_in, seed, num, sel);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// ------------------------------------------------------------------------------------------------
// Point features for SemAbs3D from the relevancy maps (visualize.py:93-122 + dataset.py:1049-1056 recipe):
//   r = rel * 50;  r -= mean over labels (if subtract_mean);  feat[l, j] = r[l, sel[j]];  xyz_out[j] = xyz[sel[j]]
// rel fp32 [L, HW], sel int64 [n] (pixel indices of the sub-sampled in-bounds points), xyz fp32 [HW, 3]
// ------------------------------------------------------------------------------------------------
__global__ void k_gather_point_features(const float* __restrict__ rel, const long long* __restrict__ sel,
                                        const float* __restrict__ xyz, int L, long HW, long n, float mult, int subtract_mean,
                                        float* __restrict__ feat, float* __restrict__ xyz_out) {
    long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const long p = sel[j];
    float mean = 0.f;
    if (subtract_mean) {
        float s = 0.f;
        for (int l = 0; l < L; ++l) s += rel[(long)l * HW + p] * mult;
        mean = s / (float)L;
    }
    for (int l = 0; l < L; ++l) feat[(long)l * n + j] = rel[(long)l * HW + p] * mult - mean;
    if (xyz_out) { xyz_out[j * 3] = xyz[p * 3]; xyz_out[j * 3 + 1] = xyz[p * 3 + 1]; xyz_out[j * 3 + 2] = xyz[p * 3 + 2]; }
}

extern "C" int semabs_gather_point_features(const float* rel, const long long* sel, const float* xyz, int L, long HW, long n,
                                            float mult, int subtract_mean, float* feat, float* xyz_out, void* stream) {
    if (n == 0 || L == 0) return SEMABS_OK;
    SEMABS_REQUIRE(rel && sel && feat && HW > 0 && (xyz_out == nullptr || xyz != nullptr), "semabs_gather_point_features: bad args");
    hipLaunchKernelGGL(k_gather_point_features, dim3(semabs_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, rel, sel, xyz, L, HW, n,
                       mult, subtract_mean, feat, xyz_out);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// ------------------------------------------------------------------------------------------------
// OVSSC post-mask (visualize.py:228-247): prediction = argmax_l logit; a voxel keeps its class unless every logit is
// below `cutoff`, it is outside the frustum, or tsdf > 0.   logits fp32 [L, M] -> label int32 [M] (-1 = empty)
// ------------------------------------------------------------------------------------------------
__global__ void k_ovssc_labels(const float* __restrict__ logits, const unsigned char* __restrict__ in_frustum,
                               const float* __restrict__ tsdf, int L, long M, float cutoff, int* __restrict__ label) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    float best = logits[i]; int arg = 0;
    for (int l = 1; l < L; ++l) { float v = logits[(long)l * M + i]; if (v > best) { best = v; arg = l; } }
    bool empty = !(best >= cutoff);                           // (logprobs < cutoff).all()
    if (in_frustum && !in_frustum[i]) empty = true;
    if (tsdf && tsdf[i] > 0.f) empty = true;
    label[i] = empty ? -1 : arg;
}
extern "C" int semabs_ovssc_labels(const float* logits, const unsigned char* in_frustum, const float* tsdf, int L, long M,
                                   float cutoff, int* label, void* stream) {
    if (M == 0) return SEMABS_OK;
    SEMABS_REQUIRE(logits && label && L > 0, "semabs_ovssc_labels: bad args");
    hipLaunchKernelGGL(k_ovssc_labels, dim3(semabs_cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream, logits, in_frustum, tsdf, L, M, cutoff, label);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}
