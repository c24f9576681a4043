// Backward / training kernels for the VOOL training step (config 5: train_vool.py -> utils.loop, utils.py:404-417):
// 3D-UNet backward (weight gradients, GroupNorm / ReLU / MaxPool / ConvTranspose backward), point-MLP and sampler-MLP
// linear layers, scatter-mean and trilinear-sampling backward, cosine-similarity pointer backward, BCE-with-logits.
// Data gradients of Conv3d / ConvTranspose3d reuse the forward implicit-GEMM kernels of unet.hip with re-laid-out weights.
// Everything here is fp32 (training uses the "exact" activation mode).
//
// Replaces what torch.autograd does for (reference file:line):
//   SemAbsVOOL.forward graph                         net.py:506-579
//   SemAbs3D.forward graph                           net.py:383-439, 185-201 (scatter), 215-256 (decoder)
//   ResidualUNet3D                                   unet3d.py:190-259, 298-317, 385-444, 596-621
//   binary_cross_entropy_with_logits                 train_vool.py:171-178
//   clip_grad_norm_                                  utils.py:415
#include "semabs_common.h"

// max |.| reduction into one global word: a wave-level max, then an atomic only when it would raise the current value (the plain read
// is a filter, not a synchronisation: millions of waves hitting one address with atomicMax serialise otherwise)
__device__ __forceinline__ void absmax_commit(unsigned int* bits, float m) {
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0 && m > 0.f) {
        const unsigned int u = __float_as_uint(m);
        if (u > __hip_atomic_load(bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(bits, u);
    }
}

// 16-byte load with the streaming ("non-temporal") cache policy: the element-wise passes of the backward read their 1-2 GB operands once
__device__ __forceinline__ float4 ld_nt4(const float* p) {
    const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return make_float4(t[0], t[1], t[2], t[3]);
}

// =================================================================================================
// Weight gradient as a split-K "A^T B" reduction over rows (voxels / points):
//   dW[ca][tap * Cx + cx] += sum_rows A[row][ca] * Xn(neighbour(row, tap))[cx]
// A fp32 [R, Ca] (rows over the index space [B, M0, M1, M2]); X fp32 [B, I0, I1, I2, Cx] gathered at m * is + td with
// zero padding, optionally through a GroupNorm affine (scale / shift per (b, cx)).  A workgroup owns a 16 x 64 tile of dW and
// a chunk of rows; 32-row batches of A and the gathered X go through LDS; fp32 atomics combine the row chunks.
// =================================================================================================
struct WgradArgs {
    const float* A; const float* X; const float* gn_scale; const float* gn_shift; float* dW;
    int B, M0, M1, M2, I0, I1, I2, is;
    int Ca, Cx, ntaps, tap_minor; long rows_per_block;
    signed char td0[28], td1[28], td2[28];
};

template <int TA>                     // a thread owns TA x 4 outputs: workgroup tile = (16 TA) x 64 of dW; TA = 4 -> two 16-byte LDS reads feed 16 FMAs
__global__ __launch_bounds__(256) void k_wgrad(WgradArgs a) {
    __shared__ __attribute__((aligned(16))) float sA[32][16 * TA];
    __shared__ __attribute__((aligned(16))) float sB[32][64];
    const int t = threadIdx.x;
    const int ca0 = blockIdx.y * 16 * TA, n0 = blockIdx.z * 64;
    const int N = a.ntaps * a.Cx;
    const long R = (long)a.B * a.M0 * a.M1 * a.M2;
    const long r_begin = (long)blockIdx.x * a.rows_per_block;
    long r_end = r_begin + a.rows_per_block; if (r_end > R) r_end = R;
    const int cg = t & 15, ng = t >> 4;
    float acc[TA][4];
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[i][k] = 0.f;
    // register double buffer: the global loads of row batch i + 1 are in flight while batch i is multiplied out of LDS (with a load ->
    // LDS -> barrier -> FMA -> barrier loop every batch paid a full memory round trip: the 8^3 / 4^3 levels were latency-bound)
    float ra[2 * TA];
    float4 rx[2];
    auto fetch = [&](long rb) {
#pragma unroll
        for (int k = 0; k < 2 * TA; ++k) {
            const int e = t + k * 256, rr = e / (16 * TA), c = e % (16 * TA);
            const long row = rb + rr;
            ra[k] = (row < r_end) ? a.A[row * a.Ca + ca0 + c] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int e = t + k * 256, rr = e >> 4, q = e & 15;
            const long row = rb + rr;
            const int n = n0 + q * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < r_end && n < N) {
                const int tap = n / a.Cx, cx = n - tap * a.Cx;
                long m = row;
                const int m2 = (int)(m % a.M2); m /= a.M2;
                const int m1 = (int)(m % a.M1); m /= a.M1;
                const int m0 = (int)(m % a.M0); const int b = (int)(m / a.M0);
                const int i0 = m0 * a.is + a.td0[tap], i1 = m1 * a.is + a.td1[tap], i2 = m2 * a.is + a.td2[tap];
                if (i0 >= 0 && i0 < a.I0 && i1 >= 0 && i1 < a.I1 && i2 >= 0 && i2 < a.I2) {
                    v = *reinterpret_cast<const float4*>(a.X + ((((long)b * a.I0 + i0) * a.I1 + i1) * a.I2 + i2) * a.Cx + cx);
                    if (a.gn_scale) {
                        const float4 s = *reinterpret_cast<const float4*>(a.gn_scale + (long)b * a.Cx + cx);
                        const float4 h = *reinterpret_cast<const float4*>(a.gn_shift + (long)b * a.Cx + cx);
                        v.x = v.x * s.x + h.x; v.y = v.y * s.y + h.y; v.z = v.z * s.z + h.z; v.w = v.w * s.w + h.w;
                    }
                }
            }
            rx[k] = v;
        }
    };
    if (r_begin < r_end) fetch(r_begin);
    for (long rb = r_begin; rb < r_end; rb += 32) {
#pragma unroll
        for (int k = 0; k < 2 * TA; ++k) {
            const int e = t + k * 256;
            sA[e / (16 * TA)][e % (16 * TA)] = ra[k];
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int e = t + k * 256;
            *reinterpret_cast<float4*>(&sB[e >> 4][(e & 15) * 4]) = rx[k];
        }
        __syncthreads();
        if (rb + 32 < r_end) fetch(rb + 32);
#pragma unroll 8
        for (int rr = 0; rr < 32; ++rr) {
            float av[TA];
            if (TA == 4) { const float4 q = *reinterpret_cast<const float4*>(&sA[rr][cg * 4]); av[0] = q.x; av[1] = q.y; av[2] = q.z; av[3] = q.w; }
            else if (TA == 2) { const float2 q = *reinterpret_cast<const float2*>(&sA[rr][cg * 2]); av[0] = q.x; av[1] = q.y; }
            else av[0] = sA[rr][cg];
            const float4 bv = *reinterpret_cast<const float4*>(&sB[rr][ng * 4]);
#pragma unroll
            for (int i = 0; i < TA; ++i) { acc[i][0] += av[i] * bv.x; acc[i][1] += av[i] * bv.y; acc[i][2] += av[i] * bv.z; acc[i][3] += av[i] * bv.w; }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int n = n0 + ng * 4 + k, ca = ca0 + cg * TA + i;
            if (n < N) {
                const int tap = n / a.Cx, cx = n - tap * a.Cx;
                const long idx = a.tap_minor ? ((long)ca * a.Cx + cx) * a.ntaps + tap : (long)ca * N + n;
                atomicAdd(&a.dW[idx], acc[i][k]);
            }
        }
}

// dW fp32 is ACCUMULATED into: [Ca, ntaps, Cx] (tap_minor = 0, the forward kernels' K order) or [Ca, Cx, ntaps] (tap_minor = 1, the
// layout of torch's Conv3d / ConvTranspose3d / Linear weights, so gradients land in the parameter-shaped buffer directly).
// taps int8 [ntaps, 3] host array.  Ca % 16 == 0, Cx % 4 == 0.
// One tap, 16 x 16 outputs, millions of rows (the final 1x1x1 convolution: dW[16][16] += sum_r g[r][:]^T x[r][:] over 16.7 M voxels): a pure
// streaming reduction.  The tiled kernel above took 1.82 ms for it (every row chunk goes through LDS for 256 outputs); here a lane pair owns
// a row - lane parity h takes output rows 8 h .. 8 h + 7, i.e. 8 x 16 accumulators in registers - reads its 96 bytes with 16-byte loads (a wave
// covers 32 consecutive rows of both operands), and the partial sums meet in a wave reduction, an LDS reduction over the four waves and one fp32 atomic per output per workgroup.
__global__ __launch_bounds__(256) void k_wgrad_rows16(const float* __restrict__ A, const float* __restrict__ X, float* __restrict__ dW, long R) {
    const int h = threadIdx.x & 1;
    float acc[8][16];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    const long stride = (long)gridDim.x * 128;
    for (long r = (long)blockIdx.x * 128 + (threadIdx.x >> 1); r < R; r += stride) {
        const float4 a0 = ld_nt4(A + r * 16 + h * 8), a1 = ld_nt4(A + r * 16 + h * 8 + 4);
        const float4 x0 = ld_nt4(X + r * 16), x1 = ld_nt4(X + r * 16 + 4), x2 = ld_nt4(X + r * 16 + 8), x3 = ld_nt4(X + r * 16 + 12);
        const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        const float xv[16] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w, x2.x, x2.y, x2.z, x2.w, x3.x, x3.y, x3.z, x3.w};
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[i][j] += av[i] * xv[j];
    }
    __shared__ float sred[4][256];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float v = acc[i][j];
#pragma unroll
            for (int o = 2; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);      // over the 32 lanes of the same parity
            if ((threadIdx.x & 63) < 2) sred[threadIdx.x >> 6][(h * 8 + i) * 16 + j] = v;
        }
    __syncthreads();
    // one atomic per output per WORKGROUP (per wave it was 2 M atomics on 256 addresses: 3.6 ms of serialised read-modify-writes)
    atomicAdd(&dW[threadIdx.x], (sred[0][threadIdx.x] + sred[1][threadIdx.x]) + (sred[2][threadIdx.x] + sred[3][threadIdx.x]));
}

// Linear layers with a NARROW input (the point MLP's first layer, 4 inputs; the sampler MLP's, 36): dW[co][ci] += sum_r A[r][co] X[r][ci].  k_wgrad walks
// 32-row batches through LDS with two barriers each - one memory round trip per 32 rows: 0.7 - 1.0 TB/s on these shapes.  Here a thread owns a 4 x 4 block
// of dW (one quad of outputs, one quad of inputs: two 16-byte loads feed 16 FMAs), a row group = (Ca / 4) (Cx / 4) threads, the block's row groups take
// every n-th row, four rows in flight per thread.  Block sums meet in LDS, then one atomic per (co, ci) and block.
__global__ __launch_bounds__(256) void k_wgrad_rows_narrow(const float* __restrict__ A, const float* __restrict__ X, float* __restrict__ dW, long R, int Ca, int Cx) {
    extern __shared__ float sred[];                         // [row groups][Ca][Cx]
    const int nqi = Cx / 4, tpg = (Ca / 4) * nqi, nrg = 256 / tpg;
    const int rg = threadIdx.x / tpg, tg = threadIdx.x % tpg, qo = tg / nqi, qi = tg % nqi;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[i][k] = 0.f;
    if (rg < nrg) {
        const long step = (long)gridDim.x * nrg;
        long r = (long)blockIdx.x * nrg + rg;
        auto fma16 = [&](const float4& d, const float4& x) {
            const float dv[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) { acc[i][0] += dv[i] * x.x; acc[i][1] += dv[i] * x.y; acc[i][2] += dv[i] * x.z; acc[i][3] += dv[i] * x.w; }
        };
        for (; r + 3 * step < R; r += 4 * step) {
            float4 d[4], x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                d[u] = *reinterpret_cast<const float4*>(A + (r + u * step) * Ca + qo * 4);
                x[u] = *reinterpret_cast<const float4*>(X + (r + u * step) * Cx + qi * 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) fma16(d[u], x[u]);
        }
        for (; r < R; r += step) fma16(*reinterpret_cast<const float4*>(A + r * Ca + qo * 4), *reinterpret_cast<const float4*>(X + r * Cx + qi * 4));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) sred[((long)rg * Ca + qo * 4 + i) * Cx + qi * 4 + k] = acc[i][k];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < Ca * Cx; e += 256) {
        float t = 0.f;
        for (int g = 0; g < nrg; ++g) t += sred[(long)g * Ca * Cx + e];
        atomicAdd(&dW[e], t);
    }
}
extern "C" int semabs_wgrad(const float* A, const float* X, const float* gn_scale, const float* gn_shift, float* dW, int B, int M0,
                            int M1, int M2, int I0, int I1, int I2, int in_stride, int Ca, int Cx, int ntaps, const signed char* taps,
                            int tap_minor, void* stream) {
    if (B == 0) return SEMABS_OK;
    SEMABS_REQUIRE(A && X && dW && taps, "semabs_wgrad: null pointer");
    SEMABS_REQUIRE(Ca % 16 == 0 && Cx % 4 == 0 && ntaps >= 1 && ntaps <= 28, "semabs_wgrad: Ca % 16, Cx % 4, 1 <= ntaps <= 28");
    WgradArgs a;
    a.A = A; a.X = X; a.gn_scale = gn_scale; a.gn_shift = gn_shift; a.dW = dW;
    a.B = B; a.M0 = M0; a.M1 = M1; a.M2 = M2; a.I0 = I0; a.I1 = I1; a.I2 = I2; a.is = in_stride; a.Ca = Ca; a.Cx = Cx; a.ntaps = ntaps;
    a.tap_minor = tap_minor;
    for (int i = 0; i < ntaps; ++i) { a.td0[i] = taps[i * 3]; a.td1[i] = taps[i * 3 + 1]; a.td2[i] = taps[i * 3 + 2]; }
    const long R = (long)B * M0 * M1 * M2;
    if (ntaps == 1 && Ca == 16 && Cx == 16 && !gn_scale && in_stride == 1 && taps[0] == 0 && taps[1] == 0 && taps[2] == 0 && I0 == M0 && I1 == M1 &&
        I2 == M2 && R >= (1L << 18)) {
        int nb = semabs_cdiv(R, 128 * 64); if (nb > 1024) nb = 1024;
        hipLaunchKernelGGL(k_wgrad_rows16, dim3(nb), dim3(256), 0, (hipStream_t)stream, A, X, dW, R);
        SEMABS_CHECK_LAUNCH();
        return SEMABS_OK;
    }
    if (ntaps == 1 && !gn_scale && in_stride == 1 && taps[0] == 0 && taps[1] == 0 && taps[2] == 0 && I0 == M0 && I1 == M1 && I2 == M2 && Cx <= 36 &&
        (Ca / 4) * (Cx / 4) <= 256 && R >= (1L << 16)) {
        const int nrg = 256 / ((Ca / 4) * (Cx / 4));
        int nb = (int)semabs_cdiv(R, (long)nrg * 64); if (nb > 2048) nb = 2048;
        hipLaunchKernelGGL(k_wgrad_rows_narrow, dim3(nb), dim3(256), (size_t)nrg * Ca * Cx * 4, (hipStream_t)stream, A, X, dW, R, Ca, Cx);
        SEMABS_CHECK_LAUNCH();
        return SEMABS_OK;
    }
    const int TA = (Ca % 64 == 0) ? 4 : (Ca % 32 == 0 ? 2 : 1);
    const int ytiles = Ca / (16 * TA), ztiles = semabs_cdiv((long)ntaps * Cx, 64);
    // enough row chunks to fill the chip a few times over, but at least 1024 rows each (keeps the atomic traffic small)
    long target_blocks = 4096 / ((long)ytiles * ztiles); if (target_blocks < 1) target_blocks = 1;
    if (target_blocks > 1024) target_blocks = 1024;      // a single-tile output (linear layers, 1x1x1 convolution): every row chunk adds to the same addresses
    long rpb = (R + target_blocks - 1) / target_blocks; if (rpb < 1024) rpb = 1024;
    rpb = (rpb + 31) / 32 * 32;
    a.rows_per_block = rpb;
    dim3 grid(semabs_cdiv(R, rpb), ytiles, ztiles);
    if (TA == 4) hipLaunchKernelGGL(k_wgrad<4>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else if (TA == 2) hipLaunchKernelGGL(k_wgrad<2>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_wgrad<1>, grid, dim3(256), 0, (hipStream_t)stream, a);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// =================================================================================================
// The same reduction on the matrix cores (split fp16: Ahi Xhi + Ahi Xlo + Alo Xhi, fp32 accumulate) for the shapes the brick kernel
// (k_wgrad16_lds) does not take: the 8^3 / 4^3 levels, the transposed convolutions (stride 2) and the linear layers.  The contraction runs
// over ROWS, which is the slow index of both operands in memory, so a 32-row batch of A [rows][16 CAB channels] and of the gathered
// X [rows][16 n-blocks of 16 (tap, cx) columns] goes through LDS in its natural row-major form - planes of [32 rows][16 channels] fp16,
// 32 B per row - and the fragments come back through the transposing LDS read (ds_read_b64_tr_b16: the 16 lanes of a group address a
// [4 rows][16 channels] block and each receives one channel's 4 rows).  Which rows form a lane's 8 k values is free as long as both
// operands agree: k group g takes rows 4 g .. 4 g + 3 and 16 + 4 g .. 16 + 4 g + 3, so a half wave reads 8 consecutive rows = 256
// contiguous bytes of a plane (conflict-free), and the plane pitch of 1088 B spreads the 8-byte staging stores over all banks.
// Workgroup = 4 waves: tile CAB x 16 output channels x 16 n-blocks, wave w owns n-blocks 4 w .. 4 w + 3; the global loads of batch i + 1 are
// in flight while batch i is multiplied; row chunks meet in fp32 atomics.  Gradient operands arrive with their dynamic power-of-two scale
// (sA2 / sX2 = (s, 1 / s) device scalars or null): multiplied by s on the way into fp16, the sums by 1 / s on the way out.
// =================================================================================================
struct WgradMArgs {
    const float* A; const float* X; const float* gn_scale; const float* gn_shift; const float* sA2; const float* sX2; float* dW;
    float* part;                // row-chunk partial sums [chunk][Ca][npad] (reduced by k_wgrad_mfma_reduce), or null: fp32 atomics on dW
    int B, M0, M1, M2, I0, I1, I2, is, Ca, Cx, ntaps, tap_minor, npad;
    unsigned R, rows_per_block;
    signed char td0[28], td1[28], td2[28];
};
typedef short wg_s16x4 __attribute__((ext_vector_type(4)));

template <int CAB>
__global__ __launch_bounds__(256, 2) void k_wgrad_mfma(WgradMArgs a) {
    constexpr int PP = 1088;                                // plane pitch (bytes)
    constexpr int NA = (CAB * 128 + 255) / 256;             // float4 of A per thread per batch
    __shared__ __attribute__((aligned(16))) char sA[2][CAB * PP];      // [hi / lo][channel block][32 rows][16 channels] fp16
    __shared__ __attribute__((aligned(16))) char sX[2][16 * PP];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int ca0 = blockIdx.y * CAB * 16, nb0 = blockIdx.z * 16;
    const int cpb = a.Cx >> 4, NB = a.ntaps * cpb;
    const unsigned r_begin = blockIdx.x * a.rows_per_block;
    unsigned r_end = r_begin + a.rows_per_block; if (r_end > a.R) r_end = a.R;
    const float sa = a.sA2 ? a.sA2[0] : 1.f, sx = a.sX2 ? a.sX2[0] : 1.f;

    // X slots of this thread: row xr of the batch, quarter xq of n-blocks 2 i + (xs >> 2), i = 0 .. 7
    const int xr = t >> 3, xs = t & 7, xq = xs & 3;
    int xtd[8], xcx[8];                                     // packed tap offsets, first channel (-1: past the last n-block)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int nb = nb0 + 2 * i + (xs >> 2);
        xtd[i] = 0; xcx[i] = -1;
        if (nb < NB) {
            const int tap = nb / cpb;
            xcx[i] = (nb - tap * cpb) * 16 + xq * 4;
            xtd[i] = (a.td0[tap] & 0xff) | ((a.td1[tap] & 0xff) << 8) | ((a.td2[tap] & 0xff) << 16);
        }
    }
    float4 pa[NA], px[8];
    auto fetch = [&](unsigned rb) {
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            const int e = t + k * 256, rr = e / (CAB * 4), q = e % (CAB * 4);
            const unsigned row = rb + rr;
            pa[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < CAB * 128 && row < r_end) pa[k] = *reinterpret_cast<const float4*>(a.A + (long)row * a.Ca + ca0 + q * 4);
        }
        const unsigned row = rb + xr;
        unsigned m = row;
        const int m2 = (int)(m % (unsigned)a.M2); m /= (unsigned)a.M2;
        const int m1 = (int)(m % (unsigned)a.M1); m /= (unsigned)a.M1;
        const int m0 = (int)(m % (unsigned)a.M0); const int b = (int)(m / (unsigned)a.M0);
        const int j0 = m0 * a.is, j1 = m1 * a.is, j2 = m2 * a.is;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < r_end && xcx[i] >= 0) {
                const int i0 = j0 + (int)(signed char)(xtd[i] & 0xff), i1 = j1 + (int)(signed char)((xtd[i] >> 8) & 0xff),
                          i2 = j2 + (int)(signed char)((xtd[i] >> 16) & 0xff);
                if (i0 >= 0 && i0 < a.I0 && i1 >= 0 && i1 < a.I1 && i2 >= 0 && i2 < a.I2) {
                    v = *reinterpret_cast<const float4*>(a.X + ((((long)b * a.I0 + i0) * a.I1 + i1) * a.I2 + i2) * a.Cx + xcx[i]);
                    if (a.gn_scale) {
                        const float4 sc = *reinterpret_cast<const float4*>(a.gn_scale + (long)b * a.Cx + xcx[i]);
                        const float4 sh = *reinterpret_cast<const float4*>(a.gn_shift + (long)b * a.Cx + xcx[i]);
                        v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
                    }
                }
            }
            px[i] = v;
        }
    };
    auto split_store = [&](float4 v, float s, char* hi, char* lo) {
        const float f[4] = {v.x * s, v.y * s, v.z * s, v.w * s};
        f16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) { h[e] = (f16)f[e]; l[e] = (f16)(f[e] - (float)h[e]); }
        *reinterpret_cast<f16x4*>(hi) = h; *reinterpret_cast<f16x4*>(lo) = l;
    };
    auto store = [&]() {
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            const int e = t + k * 256, rr = e / (CAB * 4), q = e % (CAB * 4);
            if (e < CAB * 128) { const int o = (q >> 2) * PP + rr * 32 + (q & 3) * 8; split_store(pa[k], sa, sA[0] + o, sA[1] + o); }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int o = (2 * i + (xs >> 2)) * PP + xr * 32 + xq * 8; split_store(px[i], sx, sX[0] + o, sX[1] + o); }
    };
    const int g16 = lane >> 4, li = lane & 15;
    const int fo = (g16 * 4 + (li >> 2)) * 32 + (li & 3) * 8;
    auto frag = [&](const char* plane) -> f16x8 {
        const wg_s16x4 u = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_s16x4 __attribute__((address_space(3)))*)(plane + fo));
        const wg_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_s16x4 __attribute__((address_space(3)))*)(plane + fo + 512));
        const f16x4 fu = __builtin_bit_cast(f16x4, u), fv = __builtin_bit_cast(f16x4, v);
        f16x8 r;
        r[0] = fu[0]; r[1] = fu[1]; r[2] = fu[2]; r[3] = fu[3]; r[4] = fv[0]; r[5] = fv[1]; r[6] = fv[2]; r[7] = fv[3];
        return r;
    };
    f32x4 acc[CAB][4];
#pragma unroll
    for (int c = 0; c < CAB; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[c][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool active = nb0 + w * 4 < NB;                   // wave-uniform: this wave has columns to compute

    if (r_begin < r_end) fetch(r_begin);
    for (unsigned rb = r_begin; rb < r_end; rb += 32) {
        store();
        __syncthreads();
        if (rb + 32 < r_end) fetch(rb + 32);
        if (active) {
            f16x8 ah[CAB], al[CAB];
#pragma unroll
            for (int c = 0; c < CAB; ++c) { ah[c] = frag(sA[0] + c * PP); al[c] = frag(sA[1] + c * PP); }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f16x8 xh = frag(sX[0] + (w * 4 + j) * PP), xl = frag(sX[1] + (w * 4 + j) * PP);
#pragma unroll
                for (int c = 0; c < CAB; ++c) {
                    acc[c][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[c], xh, acc[c][j], 0, 0, 0);
                    acc[c][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[c], xl, acc[c][j], 0, 0, 0);
                    acc[c][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[c], xh, acc[c][j], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    if (!active) return;
    const float os = (a.sA2 ? a.sA2[1] : 1.f) * (a.sX2 ? a.sX2[1] : 1.f);
    const int N = a.ntaps * a.Cx;
    if (a.part) {                                           // plain 64-byte-run stores of this row chunk's tile; summed (and laid out) by the reduce pass
        float* dst = a.part + (long)blockIdx.x * a.Ca * a.npad;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < CAB; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    dst[(long)(ca0 + c * 16 + g16 * 4 + r) * a.npad + (nb0 + w * 4 + j) * 16 + li] = acc[c][j][r] * os;
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int nb = nb0 + w * 4 + j;
        if (nb >= NB) break;
        const int tap = nb / cpb, cx = (nb - tap * cpb) * 16 + li;
#pragma unroll
        for (int c = 0; c < CAB; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ca = ca0 + c * 16 + g16 * 4 + r;
                const long idx = a.tap_minor ? ((long)ca * a.Cx + cx) * a.ntaps + tap : (long)ca * N + tap * a.Cx + cx;
                if (gridDim.x == 1) a.dW[idx] += acc[c][j][r] * os;       // the only workgroup with this tile
                else atomicAdd(&a.dW[idx], acc[c][j][r] * os);
            }
    }
}
// dW[layout(ca, n)] += sum_chunks part[chunk][ca][n]   (thread = one (ca, n): reads are coalesced over n, 4 partial sums in flight)
__global__ __launch_bounds__(256) void k_wgrad_mfma_reduce(const float* __restrict__ part, float* __restrict__ dW, int chunks, int Ca, int N, int npad,
                                                           int Cx, int ntaps, int tap_minor) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)Ca * N) return;
    const int ca = (int)(e / N), n = (int)(e - (long)ca * N);
    const float* p = part + (long)ca * npad + n;
    const long cs = (long)Ca * npad;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int c = 0;
    for (; c + 4 <= chunks; c += 4) { s0 += p[c * cs]; s1 += p[(c + 1) * cs]; s2 += p[(c + 2) * cs]; s3 += p[(c + 3) * cs]; }
    for (; c < chunks; ++c) s0 += p[c * cs];
    const int tap = n / Cx, cx = n - tap * Cx;
    const long idx = tap_minor ? ((long)ca * Cx + cx) * ntaps + tap : e;
    dW[idx] += (s0 + s1) + (s2 + s3);
}

// semabs_wgrad on the matrix cores; additionally Cx % 16 == 0, fewer than 2^31 rows.  sA2 / sX2: (s, 1 / s) of semabs_grad_scale for the operand that
// is a gradient (A for Conv3d / Linear, X for ConvTranspose3d), or NULL.  scratch (scratch_floats fp32, or NULL): room for the row chunks'
// partial tiles - with it the chunks are combined by a second, deterministic pass instead of fp32 atomics (measured at 16 G atomics / s on
// scattered addresses: the 8^3 level's 12 M atomics took 0.6 of its 0.75 ms).
extern "C" int semabs_wgrad_mfma(const float* A, const float* X, const float* gn_scale, const float* gn_shift, const float* sA2, const float* sX2,
                                 float* dW, int B, int M0, int M1, int M2, int I0, int I1, int I2, int in_stride, int Ca, int Cx, int ntaps,
                                 const signed char* taps, int tap_minor, float* scratch, long scratch_floats, void* stream) {
    if (B == 0) return SEMABS_OK;
    SEMABS_REQUIRE(A && X && dW && taps, "semabs_wgrad_mfma: null pointer");
    SEMABS_REQUIRE(Ca % 16 == 0 && Cx % 16 == 0 && ntaps >= 1 && ntaps <= 28, "semabs_wgrad_mfma: Ca % 16, Cx % 16, 1 <= ntaps <= 28");
    SEMABS_REQUIRE((gn_scale == nullptr) == (gn_shift == nullptr), "semabs_wgrad_mfma: gn_scale and gn_shift go together");
    const long R = (long)B * M0 * M1 * M2;
    SEMABS_REQUIRE(R > 0 && R < (1L << 31) - 64, "semabs_wgrad_mfma: row count out of range");
    WgradMArgs a;
    a.A = A; a.X = X; a.gn_scale = gn_scale; a.gn_shift = gn_shift; a.sA2 = sA2; a.sX2 = sX2; a.dW = dW;
    a.B = B; a.M0 = M0; a.M1 = M1; a.M2 = M2; a.I0 = I0; a.I1 = I1; a.I2 = I2; a.is = in_stride; a.Ca = Ca; a.Cx = Cx; a.ntaps = ntaps;
    a.tap_minor = tap_minor; a.R = (unsigned)R;
    for (int i = 0; i < ntaps; ++i) { a.td0[i] = taps[i * 3]; a.td1[i] = taps[i * 3 + 1]; a.td2[i] = taps[i * 3 + 2]; }
    const int CAB = (Ca % 64 == 0) ? 4 : (Ca % 32 == 0 ? 2 : 1);
    const int ytiles = Ca / (16 * CAB), ztiles = semabs_cdiv((long)ntaps * (Cx / 16), 16);
    a.npad = ztiles * 256;
    // row chunks: ~1024 workgroups in all, at least 128 rows each
    long chunks = 1024 / ((long)ytiles * ztiles); if (chunks < 1) chunks = 1;
    const long tile_floats = (long)Ca * a.npad;
    if (scratch && chunks > scratch_floats / tile_floats) chunks = scratch_floats / tile_floats;
    if (chunks < 1) chunks = 1;
    long rpb = (R + chunks - 1) / chunks; if (rpb < 128) rpb = 128;
    rpb = (rpb + 31) / 32 * 32;
    a.rows_per_block = (unsigned)rpb;
    chunks = semabs_cdiv(R, rpb);
    a.part = (scratch && chunks > 1 && chunks * tile_floats <= scratch_floats) ? scratch : nullptr;
    dim3 grid((unsigned)chunks, ytiles, ztiles);
    hipStream_t s = (hipStream_t)stream;
    if (CAB == 4) hipLaunchKernelGGL(k_wgrad_mfma<4>, grid, dim3(256), 0, s, a);
    else if (CAB == 2) hipLaunchKernelGGL(k_wgrad_mfma<2>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_wgrad_mfma<1>, grid, dim3(256), 0, s, a);
    if (a.part)
        hipLaunchKernelGGL(k_wgrad_mfma_reduce, dim3(semabs_cdiv((long)Ca * ntaps * Cx, 256)), dim3(256), 0, s, a.part, dW, (int)chunks, Ca, ntaps * Cx,
                           a.npad, Cx, ntaps, tap_minor);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// =================================================================================================
// Per-step weight layouts in ONE launch per matrix: out_hi / out_lo[i] = fp16 hi / lo split of src[idx[i]] (idx < 0: zero padding).
// The MFMA kernels read every convolution / linear weight as split-fp16 matrices in their own layouts (tap-major rows, flipped + transposed
// for the data gradient, parity-class matrices of the transposed convolution, each followed by its fragment-packed copy); the weights change
// every optimisation step, and building those layouts with torch ops was ~830 tiny launches (3.8 ms of GPU time) per step.  The index map of
// a layout depends on the shapes only: the host builds it once by running the layout code on an index tensor (train.py: _WeightLayouts).
// =================================================================================================
__global__ __launch_bounds__(256) void k_gather_split16(const float* __restrict__ src, const int* __restrict__ idx, long n, f16* __restrict__ hi,
                                                        f16* __restrict__ lo) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 4 <= n) {
        const int4 k = *reinterpret_cast<const int4*>(idx + i);
        const int kk[4] = {k.x, k.y, k.z, k.w};
        f16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = kk[e] >= 0 ? src[kk[e]] : 0.f;
            h[e] = (f16)v; l[e] = (f16)(v - (float)h[e]);
        }
        *reinterpret_cast<f16x4*>(hi + i) = h;
        *reinterpret_cast<f16x4*>(lo + i) = l;
    } else {
        for (long j = i; j < n; ++j) {
            const float v = idx[j] >= 0 ? src[idx[j]] : 0.f;
            const f16 h = (f16)v;
            hi[j] = h; lo[j] = (f16)(v - (float)h);
        }
    }
}
// src fp32 (any shape, contiguous); idx int32 [n] (16-byte aligned), hi / lo fp16 [n] (8-byte aligned)
extern "C" int semabs_gather_split16(const float* src, const int* idx, long n, void* hi, void* lo, void* stream) {
    if (n == 0) return SEMABS_OK;
    SEMABS_REQUIRE(src && idx && hi && lo, "semabs_gather_split16: null pointer");
    hipLaunchKernelGGL(k_gather_split16, dim3(semabs_cdiv(n, 1024)), dim3(256), 0, (hipStream_t)stream, src, idx, n, (f16*)hi, (f16*)lo);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// All layouts of a step in ONE launch (round 6): a device table of {src, idx, hi, lo, n, first block} per matrix; a block finds its matrix by binary
// search over the first-block column.  78 launches of 5 - 60 us (0.98 ms per step, most of them shorter than their launch gap) become one.
struct GatherJob { const float* src; const int* idx; f16* hi; f16* lo; long n; long block0; };
__global__ __launch_bounds__(256) void k_gather_split16_batched(const GatherJob* __restrict__ jobs, int njobs) {
    int lo_j = 0, hi_j = njobs - 1;
    const long blk = blockIdx.x;
    while (lo_j < hi_j) {                                   // last job whose first block is <= blk (uniform per workgroup)
        const int mid = (lo_j + hi_j + 1) >> 1;
        if (jobs[mid].block0 <= blk) lo_j = mid; else hi_j = mid - 1;
    }
    const GatherJob j = jobs[lo_j];
    const long i = ((blk - j.block0) * 256 + threadIdx.x) * 4;
    if (i >= j.n) return;
    if (i + 4 <= j.n) {
        const int4 k = *reinterpret_cast<const int4*>(j.idx + i);
        const int kk[4] = {k.x, k.y, k.z, k.w};
        f16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = kk[e] >= 0 ? j.src[kk[e]] : 0.f;
            h[e] = (f16)v; l[e] = (f16)(v - (float)h[e]);
        }
        *reinterpret_cast<f16x4*>(j.hi + i) = h;
        *reinterpret_cast<f16x4*>(j.lo + i) = l;
    } else {
        for (long q = i; q < j.n; ++q) {
            const float v = j.idx[q] >= 0 ? j.src[j.idx[q]] : 0.f;
            const f16 h = (f16)v;
            j.hi[q] = h; j.lo[q] = (f16)(v - (float)h);
        }
    }
}
// jobs: device array of njobs x 6 64-bit words {src, idx, hi, lo, n, first block} with first block = running sum of ceil(n / 1024); total_blocks = its end
extern "C" int semabs_gather_split16_batched(const void* jobs, int njobs, long total_blocks, void* stream) {
    if (njobs == 0 || total_blocks == 0) return SEMABS_OK;
    SEMABS_REQUIRE(jobs && njobs > 0 && total_blocks > 0 && total_blocks < (1L << 31), "semabs_gather_split16_batched: bad args");
    static_assert(sizeof(GatherJob) == 48, "the host packs six 64-bit words per job");
    hipLaunchKernelGGL(k_gather_split16_batched, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, (const GatherJob*)jobs, njobs);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// =================================================================================================
// Per-(batch, channel) reductions over voxels: Sa = sum dY, Sb = sum dY * xhat (xhat = (X - mean_g) * rstd_g), fp64 atomics.
// Used for bias gradients (X = null) and GroupNorm backward.
// =================================================================================================
__global__ __launch_bounds__(256) void k_chan_reduce(const float* __restrict__ dY, const float* __restrict__ X, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, double* __restrict__ out, long nvox, int C, int G) {
    const int b = blockIdx.y;
    const int cpv = C / 4;
    const long chunks = nvox * cpv;
    long stride = (long)gridDim.x * 256; stride -= stride % cpv;
    const long start = (long)blockIdx.x * 256 + threadIdx.x;
    __shared__ float sh[2][512];
    for (int c = threadIdx.x; c < C; c += 256) { sh[0][c] = 0.f; sh[1][c] = 0.f; }
    __syncthreads();
    if (start < stride) {
        const int c0 = (int)(start % cpv) * 4;
        float mu[4] = {0, 0, 0, 0}, rs[4] = {1, 1, 1, 1};
        if (X) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int g = (c0 + j) / (C / G); mu[j] = mean[b * G + g]; rs[j] = rstd[b * G + g]; }
        }
        float sa[4] = {0, 0, 0, 0}, sb[4] = {0, 0, 0, 0};
        const float* dyb = dY + (long)b * nvox * C; const float* xb = X ? X + (long)b * nvox * C : nullptr;
        long i = start;
        for (; i + 3 * stride < chunks; i += 4 * stride) {      // four independent loads per operand in flight
            float4 d[4], x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { d[u] = ld_nt4(dyb + (i + u * stride) * 4); if (xb) x[u] = ld_nt4(xb + (i + u * stride) * 4); }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                sa[0] += d[u].x; sa[1] += d[u].y; sa[2] += d[u].z; sa[3] += d[u].w;
                if (xb) {
                    sb[0] += d[u].x * ((x[u].x - mu[0]) * rs[0]); sb[1] += d[u].y * ((x[u].y - mu[1]) * rs[1]);
                    sb[2] += d[u].z * ((x[u].z - mu[2]) * rs[2]); sb[3] += d[u].w * ((x[u].w - mu[3]) * rs[3]);
                }
            }
        }
        for (; i < chunks; i += stride) {
            const float4 d = ld_nt4(dyb + i * 4);
            sa[0] += d.x; sa[1] += d.y; sa[2] += d.z; sa[3] += d.w;
            if (xb) {
                const float4 x = ld_nt4(xb + i * 4);
                sb[0] += d.x * ((x.x - mu[0]) * rs[0]); sb[1] += d.y * ((x.y - mu[1]) * rs[1]);
                sb[2] += d.z * ((x.z - mu[2]) * rs[2]); sb[3] += d.w * ((x.w - mu[3]) * rs[3]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { atomicAdd(&sh[0][c0 + j], sa[j]); atomicAdd(&sh[1][c0 + j], sb[j]); }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        atomicAdd(&out[((long)b * C + c) * 2], (double)sh[0][c]);
        atomicAdd(&out[((long)b * C + c) * 2 + 1], (double)sh[1][c]);
    }
}
// dY (and X) fp32 [B, nvox, C]; mean / rstd fp32 [B, G] (with X); out fp64 [B, C, 2] accumulated (zero it first)
extern "C" int semabs_chan_reduce(const float* dY, const float* X, const float* mean, const float* rstd, double* out, int B, long nvox, int C,
                                  int G, void* stream) {
    if (B == 0 || nvox == 0) return SEMABS_OK;
    SEMABS_REQUIRE(dY && out && C % 4 == 0 && C <= 512 && (!X || (mean && rstd && G > 0 && C % G == 0)), "semabs_chan_reduce: bad args");
    long chunks = nvox * (C / 4);
    // ~2048 workgroups in all (8 per CU keep enough loads in flight; 128 x B left half the chip idle for the B = 1 bias-gradient sums:
    // 0.84 ms for 1 GB); every workgroup ends with 2 C fp64 atomics
    int cap = 2048 / B; if (cap < 16) cap = 16;
    int bx = semabs_cdiv(chunks, 256 * 16); if (bx < 1) bx = 1; if (bx > cap) bx = cap;
    hipLaunchKernelGGL(k_chan_reduce, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, dY, X, mean, rstd, out, nvox, C, G > 0 ? G : 1);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// GroupNorm forward statistics in the form the backward needs: mean / rstd per (b, g) from the fp64 sums of semabs_gn_stats
__global__ void k_gn_meanrstd(const double* __restrict__ sums, float* __restrict__ mean, float* __restrict__ rstd, int n, double count, float eps) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double m = sums[i * 2] / count;
    double var = sums[i * 2 + 1] / count - m * m; if (var < 0) var = 0;
    mean[i] = (float)m; rstd[i] = (float)(1.0 / sqrt(var + (double)eps));
}
extern "C" int semabs_gn_meanrstd(const double* sums, float* mean, float* rstd, int B, int G, long count, float eps, void* stream) {
    if (B == 0) return SEMABS_OK;
    SEMABS_REQUIRE(sums && mean && rstd && count > 0, "semabs_gn_meanrstd: bad args");
    hipLaunchKernelGGL(k_gn_meanrstd, dim3(semabs_cdiv((long)B * G, 256)), dim3(256), 0, (hipStream_t)stream, sums, mean, rstd, B * G, (double)count, eps);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// GroupNorm backward, coefficient stage: red fp64 [B, C, 2] (Sa, Sb) ->
//   coef fp32 [B, C, 3] = (rstd * gamma_c, rstd * s1 / n, rstd * s2 / n) with s1 = sum_{c in g} gamma_c Sa, s2 = sum gamma_c Sb
//   dgamma[c] += sum_b Sb[b, c];  dbeta[c] += sum_b Sa[b, c]
__global__ void k_gn_bwd_coef(const double* __restrict__ red, const float* __restrict__ gamma, const float* __restrict__ rstd,
                              const float* __restrict__ inv_scale, float* __restrict__ coef, float* __restrict__ dgamma,
                              float* __restrict__ dbeta, int B, int C, int G, double n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i % C, cg = C / G, g = c / cg;
    const double inv = inv_scale ? (double)inv_scale[0] : 1.0;        // dXn (and so red) carry the dynamic gradient scale
    double s1 = 0, s2 = 0;
    for (int k = g * cg; k < (g + 1) * cg; ++k) { s1 += (double)gamma[k] * red[((long)b * C + k) * 2]; s2 += (double)gamma[k] * red[((long)b * C + k) * 2 + 1]; }
    s1 *= inv; s2 *= inv;
    const float rs = rstd[b * G + g];
    coef[i * 3] = (float)(rs * gamma[c] * inv); coef[i * 3 + 1] = (float)(rs * s1 / n); coef[i * 3 + 2] = (float)(rs * s2 / n);
    atomicAdd(&dgamma[c], (float)(red[((long)b * C + c) * 2 + 1] * inv));
    atomicAdd(&dbeta[c], (float)(red[((long)b * C + c) * 2] * inv));
}
extern "C" int semabs_gn_bwd_coef(const double* red, const float* gamma, const float* rstd, const float* inv_scale, float* coef, float* dgamma,
                                  float* dbeta, int B, int C, int G, long nvox, void* stream) {
    if (B == 0) return SEMABS_OK;
    SEMABS_REQUIRE(red && gamma && rstd && coef && dgamma && dbeta && C % G == 0, "semabs_gn_bwd_coef: bad args");
    hipLaunchKernelGGL(k_gn_bwd_coef, dim3(semabs_cdiv((long)B * C, 256)), dim3(256), 0, (hipStream_t)stream, red, gamma, rstd, inv_scale, coef,
                       dgamma, dbeta, B, C, G, (double)nvox * (C / G));
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// dX = (k1 * dXn - k2 - xhat * k3 [+ add1] [+ add2]) [* (mask_y > 0)]; optionally max |dX| -> bits (for the next dynamic gradient scale)
// grid = (chunks of 256 * 4 float4 per volume, B): a thread's four channels - hence its mean / rstd / coefficients - are the same for every
// element it touches when C / 4 divides 256 (every channel count of the network), so they are loaded once; 4 independent float4 per operand
// in flight per thread.  (One float4 per thread with the (b, channel) decode and nine scalar coefficient loads per element ran at 4.7 TB/s.)
template <bool FIXED>                                       // FIXED: C / 4 divides 256
__global__ __launch_bounds__(256) void k_gn_bwd_apply(const float* __restrict__ dXn, const float* __restrict__ X, const float* __restrict__ mean,
                               const float* __restrict__ rstd, const float* __restrict__ coef, const float* __restrict__ add1,
                               const float* __restrict__ add2, const float* __restrict__ mask_y, unsigned int* __restrict__ bits,
                               float* __restrict__ dX, long per_vol, int C, int G) {
    const int b = blockIdx.y;
    const int cpv = C / 4;
    const long vbase = (long)b * per_vol;                   // in float4 units
    float mu[4], rs[4], k0[4], k1[4], k2[4];
    auto setup = [&](int c0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + j, g = c / (C / G);
            mu[j] = mean[b * G + g]; rs[j] = rstd[b * G + g];
            const float* k = coef + ((long)b * C + c) * 3;
            k0[j] = k[0]; k1[j] = k[1]; k2[j] = k[2];
        }
    };
    if (FIXED) setup((threadIdx.x % cpv) * 4);
    float m = 0.f;
    const long i0 = (long)blockIdx.x * 1024 + threadIdx.x;
    float4 d[4], x[4], a1[4], a2[4], y[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long i = i0 + u * 256;
        ok[u] = i < per_vol;
        const long e = (vbase + (ok[u] ? i : 0)) * 4;
        d[u] = ld_nt4(dXn + e); x[u] = ld_nt4(X + e);
        if (add1) a1[u] = ld_nt4(add1 + e);
        if (add2) a2[u] = ld_nt4(add2 + e);
        if (mask_y && mask_y != X) y[u] = ld_nt4(mask_y + e);           // (mask_y == X - the layer's input is the post-ReLU activation - is the usual case: no second load)
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long i = i0 + u * 256;
        if (!FIXED) setup((int)(i % cpv) * 4);
        const float dv[4] = {d[u].x, d[u].y, d[u].z, d[u].w}, xv[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = k0[j] * dv[j] - k1[j] - ((xv[j] - mu[j]) * rs[j]) * k2[j];
        if (add1) { o[0] += a1[u].x; o[1] += a1[u].y; o[2] += a1[u].z; o[3] += a1[u].w; }
        if (add2) { o[0] += a2[u].x; o[1] += a2[u].y; o[2] += a2[u].z; o[3] += a2[u].w; }
        if (mask_y) {                                      // X is the post-ReLU output of the previous layer only when the caller says so
            const float4 mk = mask_y == X ? x[u] : y[u];
            o[0] = mk.x > 0.f ? o[0] : 0.f; o[1] = mk.y > 0.f ? o[1] : 0.f; o[2] = mk.z > 0.f ? o[2] : 0.f; o[3] = mk.w > 0.f ? o[3] : 0.f;
        }
        if (ok[u]) {
            *reinterpret_cast<float4*>(dX + (vbase + i) * 4) = make_float4(o[0], o[1], o[2], o[3]);
            m = fmaxf(m, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
        }
    }
    if (bits) absmax_commit(bits, m);
}
extern "C" int semabs_gn_bwd_apply(const float* dXn, const float* X, const float* mean, const float* rstd, const float* coef, const float* add1,
                                   const float* add2, const float* mask_y, unsigned int* absmax_bits, float* dX, int B, long nvox, int C, int G,
                                   void* stream) {
    if (B == 0) return SEMABS_OK;
    SEMABS_REQUIRE(dXn && X && mean && rstd && coef && dX && C % 4 == 0 && C % G == 0, "semabs_gn_bwd_apply: bad args");
    const long per_vol = nvox * (C / 4);
    SEMABS_REQUIRE(B < 65536 && semabs_cdiv(per_vol, 1024) < (1L << 31), "semabs_gn_bwd_apply: grid too large");
    const dim3 grid((unsigned)semabs_cdiv(per_vol, 1024), B);
    if (256 % (C / 4) == 0)
        hipLaunchKernelGGL(k_gn_bwd_apply<true>, grid, dim3(256), 0, (hipStream_t)stream, dXn, X, mean, rstd, coef, add1, add2, mask_y, absmax_bits, dX, per_vol, C, G);
    else
        hipLaunchKernelGGL(k_gn_bwd_apply<false>, grid, dim3(256), 0, (hipStream_t)stream, dXn, X, mean, rstd, coef, add1, add2, mask_y, absmax_bits, dX, per_vol, C, G);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// =================================================================================================
// Element-wise: masked gradients and sums
//   mode 0: out = dY * (Y > 0)                 (ReLU)
//   mode 1: out = dY * (Y > 0 ? 1 : slope)     (LeakyReLU, sign of the output = sign of the pre-activation)
//   mode 2: out = a + b                        (gradient fan-in;  Y = second addend)
//   mode 3: out = a * b[0]                     (undo the dynamic gradient scale, b = device scalar)
// =================================================================================================
__global__ __launch_bounds__(256) void k_ew(const float* __restrict__ dY, const float* __restrict__ Y, float* __restrict__ out, long n4, int mode, float slope,
                     unsigned int* __restrict__ bits, const float* __restrict__ in_scale) {
    // four float4 per operand in flight per thread (one per thread ran at 5.2 TB/s)
    const long i0 = (long)blockIdx.x * 1024 + threadIdx.x;
    const float kin = in_scale ? in_scale[0] : 1.f;        // the producer's dynamic gradient scale undone on the way in (a power of two: exact)
    const float k3 = mode == 3 ? Y[0] : 0.f;
    const float s = mode == 0 ? 0.f : slope;
    float4 d[4], y[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long i = i0 + u * 256 < n4 ? i0 + u * 256 : 0;
        d[u] = ld_nt4(dY + i * 4);
        if (mode != 3) y[u] = ld_nt4(Y + i * 4);
    }
    float m = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long i = i0 + u * 256;
        float4 o;
        const float4 dd = make_float4(d[u].x * kin, d[u].y * kin, d[u].z * kin, d[u].w * kin);
        if (mode == 3) o = make_float4(dd.x * k3, dd.y * k3, dd.z * k3, dd.w * k3);
        else if (mode == 2) o = make_float4(dd.x + y[u].x, dd.y + y[u].y, dd.z + y[u].z, dd.w + y[u].w);
        else o = make_float4(dd.x * (y[u].x > 0.f ? 1.f : s), dd.y * (y[u].y > 0.f ? 1.f : s), dd.z * (y[u].z > 0.f ? 1.f : s), dd.w * (y[u].w > 0.f ? 1.f : s));
        if (i < n4) {
            *reinterpret_cast<float4*>(out + i * 4) = o;
            m = fmaxf(m, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
        }
    }
    if (bits) absmax_commit(bits, m);                       // max |out| for the dynamic gradient scale of the next convolution
}
// absmax_bits: optional uint32 (zero it first) receiving the bit pattern of max |out| - feeds semabs_grad_scale(have_bits = 1)
extern "C" int semabs_ew(const float* a, const float* b, float* out, long n, int mode, float slope, unsigned int* absmax_bits, void* stream) {
    if (n == 0) return SEMABS_OK;
    SEMABS_REQUIRE(a && b && out && n % 4 == 0 && mode >= 0 && mode <= 3, "semabs_ew: bad args (n % 4 == 0)");
    hipLaunchKernelGGL(k_ew, dim3(semabs_cdiv(n / 4, 1024)), dim3(256), 0, (hipStream_t)stream, a, b, out, n / 4, mode, slope, absmax_bits, (const float*)nullptr);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}
// the same with the first operand multiplied by the device scalar in_scale[0] first (modes 0 - 2): folds the separate "undo the dynamic gradient
// scale" pass (mode 3) of a data-gradient convolution's output into the element-wise pass that consumes it
extern "C" int semabs_ew_scaled(const float* a, const float* b, const float* in_scale, float* out, long n, int mode, float slope, unsigned int* absmax_bits,
                                void* stream) {
    if (n == 0) return SEMABS_OK;
    SEMABS_REQUIRE(a && b && in_scale && out && n % 4 == 0 && mode >= 0 && mode <= 2, "semabs_ew_scaled: bad args (n % 4 == 0, mode 0..2)");
    hipLaunchKernelGGL(k_ew, dim3(semabs_cdiv(n / 4, 1024)), dim3(256), 0, (hipStream_t)stream, a, b, out, n / 4, mode, slope, absmax_bits, in_scale);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// MaxPool3d(2) backward, channels-last fp32: the FIRST maximal element of each 2x2x2 window (scan order d, h, w) gets dY
template <bool RELU_MASK>                                   // RELU_MASK: X is a post-ReLU activation and the caller wants the gradient in FRONT of that ReLU
__global__ void k_maxpool_bwd(const float* __restrict__ X, const float* __restrict__ dY, float* __restrict__ dX, int B, int O0, int O1, int O2, int C,
                              const float* __restrict__ add, unsigned int* __restrict__ bits) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long tot = (long)B * O0 * O1 * O2 * C;
    if (i >= tot) { if (bits) absmax_commit(bits, 0.f); return; }      // (every lane takes part in the wave-level maximum)
    const int c = (int)(i % C); long v = i / C;
    const int o2 = (int)(v % O2); v /= O2;
    const int o1 = (int)(v % O1); v /= O1;
    const int o0 = (int)(v % O0); const int b = (int)(v / O0);
    const int I1 = 2 * O1, I2 = 2 * O2;
    float best = -INFINITY; int arg = 0; long idx[8];
    unsigned pos = 0;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        idx[d] = ((((long)b * 2 * O0 + 2 * o0 + (d >> 2)) * I1 + 2 * o1 + ((d >> 1) & 1)) * I2 + 2 * o2 + (d & 1)) * C + c;
        const float x = X[idx[d]];
        if (x > best) { best = x; arg = d; }
        pos |= (x > 0.f ? 1u : 0u) << d;
    }
    const float g = dY[i];
    float m = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        float o = (d == arg) ? g : 0.f;
        if (add) o += add[idx[d]];                           // the skip connection's gradient (the pooled tensor is also a decoder input): one pass less
        if (RELU_MASK && !((pos >> d) & 1u)) o = 0.f;        // ... and the ReLU that produced X: the block's own mask pass (read dX, read X, write) is gone
        dX[idx[d]] = o;
        m = fmaxf(m, fabsf(o));
    }
    if (bits) absmax_commit(bits, m);
}
extern "C" int semabs_maxpool3d_bwd(const float* X, const float* dY, float* dX, int B, int D0, int D1, int D2, int C, void* stream) {
    if (B == 0) return SEMABS_OK;
    SEMABS_REQUIRE(X && dY && dX && D0 % 2 == 0 && D1 % 2 == 0 && D2 % 2 == 0, "semabs_maxpool3d_bwd: bad args");
    const long tot = (long)B * (D0 / 2) * (D1 / 2) * (D2 / 2) * C;
    hipLaunchKernelGGL(k_maxpool_bwd<false>, dim3(semabs_cdiv(tot, 256)), dim3(256), 0, (hipStream_t)stream, X, dY, dX, B, D0 / 2, D1 / 2, D2 / 2, C,
                       (const float*)nullptr, (unsigned int*)nullptr);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}
// the same + `add` (like dX, e.g. the skip connection's gradient) added on the way out, and (optional) max |dX| -> absmax_bits (zero it first)
extern "C" int semabs_maxpool3d_bwd_add(const float* X, const float* dY, const float* add, float* dX, unsigned int* absmax_bits, int relu_mask, int B, int D0,
                                        int D1, int D2, int C, void* stream) {
    if (B == 0) return SEMABS_OK;
    SEMABS_REQUIRE(X && dY && add && dX && D0 % 2 == 0 && D1 % 2 == 0 && D2 % 2 == 0, "semabs_maxpool3d_bwd_add: bad args");
    const long tot = (long)B * (D0 / 2) * (D1 / 2) * (D2 / 2) * C;
    if (relu_mask) hipLaunchKernelGGL(k_maxpool_bwd<true>, dim3(semabs_cdiv(tot, 256)), dim3(256), 0, (hipStream_t)stream, X, dY, dX, B, D0 / 2, D1 / 2, D2 / 2, C, add, absmax_bits);
    else hipLaunchKernelGGL(k_maxpool_bwd<false>, dim3(semabs_cdiv(tot, 256)), dim3(256), 0, (hipStream_t)stream, X, dY, dX, B, D0 / 2, D1 / 2, D2 / 2, C, add, absmax_bits);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// =================================================================================================
// Dense fp32 linear layer for the point / sampler MLPs: Y[R, Co] = act(X[R, Ci] . W[Co, Ci]^T + b)
// act: 0 none, 1 LeakyReLU(slope).  One thread = one row x 16 outputs; the 16 x Ci weight slice sits in LDS.
// =================================================================================================
__global__ __launch_bounds__(256) void k_linear(const float* __restrict__ X, const float* __restrict__ W, const float* __restrict__ bias,
                                                float* __restrict__ Y, long R, int Ci, int Co, int act, float slope) {
    extern __shared__ float sw[];                     // [16][Ci]
    const int co0 = blockIdx.y * 16;
    for (int i = threadIdx.x; i < 16 * Ci; i += 256) { const int o = i / Ci, k = i - o * Ci; sw[i] = (co0 + o < Co) ? W[(long)(co0 + o) * Ci + k] : 0.f; }
    __syncthreads();
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    float acc[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) acc[o] = (bias && co0 + o < Co) ? bias[co0 + o] : 0.f;
    const float* xr = X + r * Ci;
    for (int k = 0; k < Ci; ++k) {
        const float xv = xr[k];
#pragma unroll
        for (int o = 0; o < 16; ++o) acc[o] += sw[o * Ci + k] * xv;
    }
#pragma unroll
    for (int o = 0; o < 16; ++o)
        if (co0 + o < Co) { float v = acc[o]; if (act == 1) v = v > 0.f ? v : slope * v; Y[r * Co + co0 + o] = v; }
}
extern "C" int semabs_linear_f32(const float* X, const float* W, const float* bias, float* Y, long R, int Ci, int Co, int act, float slope, void* stream) {
    if (R == 0) return SEMABS_OK;
    SEMABS_REQUIRE(X && W && Y && Ci > 0 && Co > 0 && Ci <= 1024, "semabs_linear_f32: bad args");
    hipLaunchKernelGGL(k_linear, dim3(semabs_cdiv(R, 256), semabs_cdiv(Co, 16)), dim3(256), (size_t)16 * Ci * 4, (hipStream_t)stream, X, W, bias, Y, R, Ci, Co, act, slope);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// =================================================================================================
// Tall-and-skinny linear layers of the training step (point MLP 4 -> 128 -> 128 -> 16 on P * N rows, sampler MLP 36 -> 32 -> 64 on D * M rows, and
// their data gradients) on the matrix cores with fp32-like accuracy: Y[R, Co] = act((s X)[R, Ci] . W^T + b), X / Y fp32 row-major, W fp32 addressed as
// W[n * w_sn + k * w_sk] (so the transpose needs no copy), operands split into fp16 hi + lo on the fly (three MFMA products, fp32 accumulation - the
// scheme of the convolutions).  Round 5: these layers ran as 1 x 1 x 1 "convolutions" on the generic gather kernel (0.7 - 0.9 ms each at 640 k - 1.6 M rows,
// 0.02 of their roof) or on the fp32 FMA kernel k_linear (0.73 ms): ~5.5 ms of the 62.8 ms step.
// A workgroup (4 waves) stages W once as fp16 hi / lo planes in LDS ([Co16][Kp + 8]: the 16-byte pad puts the 16 rows of a fragment read on distinct bank
// groups), then each wave walks 16-row tiles: per k-step of 32 a lane loads its row's 8 fp32 (2 x 16 B), scales and splits them, and multiplies them
// against the Co16 / 16 column tiles.  s (optional, a device scalar) is the dynamic power-of-two scale of a gradient input - REQUIRED for one: values of
// 1e-7 are fp16 subnormals before the split; the accumulator is multiplied by out_scale (optional device scalar, e.g. 1 / s) in front of the bias.
// =================================================================================================
__device__ __forceinline__ float row16_sum(float v) {      // sum over the 16 lanes of a DPP row, in every lane
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));     // quad_perm [1, 0, 3, 2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));     // quad_perm [2, 3, 0, 1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));    // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));    // row_mirror
    return v;
}
template <int NT>                                            // column tiles of 16 kept in registers per pass (Co16 / 16 <= NT)
__global__ __launch_bounds__(256) void k_linear_rows(const float* __restrict__ X, long ldx, const float* __restrict__ W, long w_sn, long w_sk,
                                                     const float* __restrict__ bias, float* __restrict__ Y, long R, int Ci, int Co, int act, float slope,
                                                     const float* __restrict__ in_scale, const float* __restrict__ out_scale,
                                                     const float* __restrict__ relu_mask, unsigned int* __restrict__ bits, float* __restrict__ colsum) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int Kp = (Ci + 31) / 32 * 32, KS = Kp + 8;         // padded K and LDS row stride (fp16 elements)
    const int Co16 = (Co + 15) / 16 * 16;
    f16* s_hi = reinterpret_cast<f16*>(smem);
    f16* s_lo = s_hi + (long)Co16 * KS;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < Co16 * Kp; i += 256) {
        const int n = i / Kp, k = i - n * Kp;
        const float w = (n < Co && k < Ci) ? W[n * w_sn + k * w_sk] : 0.f;
        const f16 h = (f16)w;
        s_hi[n * KS + k] = h; s_lo[n * KS + k] = (f16)(w - (float)h);
    }
    __syncthreads();
    const float sx = in_scale ? in_scale[0] : 1.f, so = out_scale ? out_scale[0] : 1.f;
    const int vl = lane & 15, kg = lane >> 4;
    const int nt = Co16 / 16;
    // (the weights are the A operand, the rows the B operand: a lane's four results are four CONSECUTIVE output channels of one row - 16-byte stores and mask
    //  loads; with the operands the other way round a lane held one channel of four rows: 32 scalar stores and 32 scalar mask loads per tile for 128 outputs,
    //  and the masked 128 x 128 data gradient ran at 0.6 ms for 1 GB)
    float bv[NT][4];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) bv[j][i] = (bias && j < nt && j * 16 + 4 * kg + i < Co) ? bias[j * 16 + 4 * kg + i] : 0.f;
    const bool vec4 = Co % 4 == 0;
    float cs[NT][4];                                         // colsum: this lane's share of the column sums of Y (the bias gradient when Y is a layer's dOut)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) cs[j][i] = 0.f;
    const long ntiles = (R + 15) / 16;
    float amax = 0.f;
    for (long t = (long)blockIdx.x * 4 + wid; t < ntiles; t += (long)gridDim.x * 4) {
        const long row = t * 16 + vl;
        const float* xr = X + (row < R ? row : R - 1) * ldx;
        f32x4 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < Kp; ks += 32) {
            const int k0 = ks + kg * 8;
            float v[8];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int k = k0 + hf * 4;
                if (k + 4 <= Ci) { const float4 q = *reinterpret_cast<const float4*>(xr + k); v[hf * 4] = q.x; v[hf * 4 + 1] = q.y; v[hf * 4 + 2] = q.z; v[hf * 4 + 3] = q.w; }
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[hf * 4 + e] = (k + e < Ci) ? xr[k + e] : 0.f;
                }
            }
            f16x8 xh, xl;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float q = v[e] * sx; xh[e] = (f16)q; xl[e] = (f16)(q - (float)xh[e]); }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (j < nt) {
                    const f16x8 wh = *reinterpret_cast<const f16x8*>(s_hi + (j * 16 + vl) * KS + k0);
                    const f16x8 wl = *reinterpret_cast<const f16x8*>(s_lo + (j * 16 + vl) * KS + k0);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh, acc[j], 0, 0, 0);
                }
            }
        }
        // acc[j][i] = Y[t * 16 + vl][j * 16 + 4 * kg + i]
        const long r = t * 16 + vl;
        if (r < R) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int c0 = j * 16 + 4 * kg;
                if (j < nt && c0 < Co) {
                    float o[4], mk[4] = {1.f, 1.f, 1.f, 1.f};
                    if (relu_mask) {
                        if (vec4) { const float4 m4 = *reinterpret_cast<const float4*>(relu_mask + r * Co + c0); mk[0] = m4.x; mk[1] = m4.y; mk[2] = m4.z; mk[3] = m4.w; }
                        else {
#pragma unroll
                            for (int i = 0; i < 4; ++i) if (c0 + i < Co) mk[i] = relu_mask[r * Co + c0 + i];
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        o[i] = acc[j][i] * so + bv[j][i];
                        if (act == 1) o[i] = o[i] > 0.f ? o[i] : slope * o[i];
                        // Y is a gradient in front of the ReLU (act 0) / LeakyReLU (act 2: factor `slope`) that produced relu_mask
                        if (relu_mask && !(mk[i] > 0.f)) o[i] = act == 2 ? o[i] * slope : 0.f;
                    }
                    if (colsum) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) cs[j][i] += (c0 + i < Co) ? o[i] : 0.f;
                    }
                    if (vec4) {
                        *reinterpret_cast<float4*>(Y + r * Co + c0) = make_float4(o[0], o[1], o[2], o[3]);
                        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) if (c0 + i < Co) { Y[r * Co + c0 + i] = o[i]; amax = fmaxf(amax, fabsf(o[i])); }
                    }
                }
            }
        }
    }
    if (bits) absmax_commit(bits, amax);
    if (colsum) {                                            // rows of a wave by DPP, waves of the block through LDS (the weight planes are dead), one atomic per column and block
        __syncthreads();
        float* s_cs = reinterpret_cast<float*>(smem);
        for (int c = tid; c < Co16; c += 256) s_cs[c] = 0.f;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (j < nt) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = row16_sum(cs[j][i]);
                    if (vl == 0) atomicAdd(&s_cs[j * 16 + 4 * kg + i], v);
                }
            }
        }
        __syncthreads();
        for (int c = tid; c < Co; c += 256) atomicAdd(&colsum[c], s_cs[c]);
    }
}
// The 16 -> 16 case (the UNet's final 1 x 1 x 1 convolution and its data gradient over 16.7 M voxels) on the FP32 matrix instruction: Y^T = W X^T with
// v_mfma_f32_16x16x4_f32, four of them per 16-row tile.  A lane's operands are ONE float4 of W (loaded once) and ONE float4 of its row (k = 4 kg + s for MFMA s),
// its result four consecutive output channels of one row: 16-byte loads and stores, a wave reads and writes 1 KB contiguous per tile, no operand split, no
// LDS - the products are exact fp32.  (k_linear_rows<2> on these shapes: 0.66 / 0.52 ms for 2.1 GB; this kernel runs at the memory rate.)
__global__ __launch_bounds__(256) void k_rows16_f32(const float* __restrict__ X, const float* __restrict__ W, long w_sn, long w_sk, const float* __restrict__ bias,
                                                    float* __restrict__ Y, long R, int act, float slope, const float* __restrict__ in_scale,
                                                    const float* __restrict__ out_scale, const float* __restrict__ relu_mask, unsigned int* __restrict__ bits,
                                                    float* __restrict__ colsum, float* __restrict__ x_colsum) {
    const int lane = threadIdx.x & 63, vl = lane & 15, kg = lane >> 4;
    float cs[4] = {0.f, 0.f, 0.f, 0.f}, cx[4] = {0.f, 0.f, 0.f, 0.f};
    float wv[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) wv[s4] = W[vl * w_sn + (4 * kg + s4) * w_sk];        // A[m = vl][k = 4 kg + s]
    const float sc = (in_scale ? in_scale[0] : 1.f) * (out_scale ? out_scale[0] : 1.f);     // (powers of two: exact)
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) bv = *reinterpret_cast<const float4*>(bias + 4 * kg);
    const long ntiles = (R + 15) / 16, stride = (long)gridDim.x * 4;
    float amax = 0.f;
    for (long t = (long)blockIdx.x * 4 + (threadIdx.x >> 6); t < ntiles; t += stride) {
        const long row = t * 16 + vl;
        const bool ok = row < R;
        const float4 xv = ld_nt4(X + (ok ? row : R - 1) * 16 + 4 * kg);                    // B[k = 4 kg + s][n = row vl]
        float4 mk = make_float4(1.f, 1.f, 1.f, 1.f);
        if (relu_mask && ok) mk = ld_nt4(relu_mask + row * 16 + 4 * kg);
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[0], xv.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[1], xv.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[2], xv.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[3], xv.w, acc, 0, 0, 0);
        // acc[i] = Y[row vl][cout = 4 kg + i]
        float o[4] = {acc[0] * sc + bv.x, acc[1] * sc + bv.y, acc[2] * sc + bv.z, acc[3] * sc + bv.w};
        const float mv[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (act == 1) o[i] = o[i] > 0.f ? o[i] : slope * o[i];
            if (relu_mask && !(mv[i] > 0.f)) o[i] = act == 2 ? o[i] * slope : 0.f;
        }
        if (ok) {
            *reinterpret_cast<float4*>(Y + row * 16 + 4 * kg) = make_float4(o[0], o[1], o[2], o[3]);
            amax = fmaxf(amax, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
            cs[0] += o[0]; cs[1] += o[1]; cs[2] += o[2]; cs[3] += o[3];
            cx[0] += xv.x; cx[1] += xv.y; cx[2] += xv.z; cx[3] += xv.w;
        }
    }
    if (bits) absmax_commit(bits, amax);
    if (colsum || x_colsum) {
        __shared__ float s_cs[32];
        if (threadIdx.x < 32) s_cs[threadIdx.x] = 0.f;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = row16_sum(cs[i]), u = row16_sum(cx[i]);
            if (vl == 0) { atomicAdd(&s_cs[4 * kg + i], v); atomicAdd(&s_cs[16 + 4 * kg + i], u); }
        }
        __syncthreads();
        if (threadIdx.x < 16) { if (colsum) atomicAdd(&colsum[threadIdx.x], s_cs[threadIdx.x]); if (x_colsum) atomicAdd(&x_colsum[threadIdx.x], s_cs[16 + threadIdx.x]); }
    }
}
extern "C" int semabs_linear_rows(const float* X, long ldx, const float* W, long w_sn, long w_sk, const float* bias, float* Y, long R, int Ci, int Co,
                                  int act, float slope, const float* in_scale, const float* out_scale, const float* relu_mask, unsigned int* absmax_bits,
                                  float* colsum, float* x_colsum, void* stream) {
    if (R == 0) return SEMABS_OK;
    SEMABS_REQUIRE(X && W && Y && R > 0 && Ci > 0 && Co > 0, "semabs_linear_rows: bad args");
    SEMABS_REQUIRE(Ci % 4 == 0 && ldx % 4 == 0 && Ci <= 512 && Co <= 128 && (act == 0 || act == 1 || (act == 2 && relu_mask)), "semabs_linear_rows: Ci % 4 == 0, Ci <= 512, Co <= 128, act 0 / 1 (2 with relu_mask)");
    if (Ci == 16 && Co == 16 && ldx == 16 && R >= (1L << 16)) {
        long nb16 = ((R + 15) / 16 + 3) / 4; if (nb16 > 4096) nb16 = 4096;
        hipLaunchKernelGGL(k_rows16_f32, dim3((unsigned)nb16), dim3(256), 0, (hipStream_t)stream, X, W, w_sn, w_sk, bias, Y, R, act, slope, in_scale, out_scale,
                           relu_mask, absmax_bits, colsum, x_colsum);
        SEMABS_CHECK_LAUNCH();
        return SEMABS_OK;
    }
    SEMABS_REQUIRE(!x_colsum, "semabs_linear_rows: x_colsum only with the 16 -> 16 kernel (Ci = Co = ldx = 16, at least 2^16 rows)");
    const int Kp = (Ci + 31) / 32 * 32, Co16 = (Co + 15) / 16 * 16;
    const size_t lds = (size_t)Co16 * (Kp + 8) * 2 * 2;
    SEMABS_REQUIRE(lds <= 160 * 1024, "semabs_linear_rows: the weight matrix does not fit LDS");
    const long ntiles = (R + 15) / 16;
    // every block stages the whole weight matrix (through strides when it is the transpose: 4-byte reads 4 Co bytes apart): with the large matrices, whose LDS
    // footprint allows two blocks per CU anyway, one wave of blocks - 2 per CU - instead of 2 048 (the 128 x 128 transposed layer: 640 -> us below)
    long nb = (ntiles + 3) / 4; const long cap = lds > 48 * 1024 ? 512 : 2048; if (nb > cap) nb = cap;
    hipStream_t s = (hipStream_t)stream;
#define LR_LAUNCH(NT)                                                                                                           \
    {                                                                                                                           \
        static SemabsLdsAttr attr;                                                                                              \
        semabs_ensure_lds(&k_linear_rows<NT>, 160 * 1024, attr);                                                                \
        hipLaunchKernelGGL(k_linear_rows<NT>, dim3((unsigned)nb), dim3(256), lds, s, X, ldx, W, w_sn, w_sk, bias, Y, R, Ci, Co, act, slope, in_scale, out_scale, relu_mask, absmax_bits, colsum); \
    }
    if (Co16 <= 32) LR_LAUNCH(2) else if (Co16 <= 64) LR_LAUNCH(4) else LR_LAUNCH(8)
#undef LR_LAUNCH
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// =================================================================================================
// Scatter-mean backward: dpf[b, p, :] = dvol[b, flat[p], :] / count[flat[p]]   (count via an integer histogram)
// =================================================================================================
__global__ void k_voxel_count(const long long* __restrict__ flat, long N, int* __restrict__ count) {
    long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < N) atomicAdd(&count[flat[p]], 1);
}
__global__ void k_scatter_mean_bwd(const long long* __restrict__ flat, const int* __restrict__ count, const float* __restrict__ dvol,
                                   float* __restrict__ dpf, int P, long N, int C, long nvox) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)P * N * (C / 4)) return;
    const int c4 = (int)(i % (C / 4)); const long bp = i / (C / 4);
    const long p = bp % N; const int b = (int)(bp / N);
    const long v = flat[p];
    const float inv = 1.f / (float)count[v];
    float4 d = *reinterpret_cast<const float4*>(dvol + ((long)b * nvox + v) * C + c4 * 4);
    d.x *= inv; d.y *= inv; d.z *= inv; d.w *= inv;
    *reinterpret_cast<float4*>(dpf + bp * C + c4 * 4) = d;
}
// count int32 [nvox] must be zero-filled by the caller
extern "C" int semabs_scatter_mean_bwd(const long long* flat, int* count, const float* dvol, float* dpf, int P, long N, int C, long nvox, void* stream) {
    if (P == 0 || N == 0) return SEMABS_OK;
    SEMABS_REQUIRE(flat && count && dvol && dpf && C % 4 == 0, "semabs_scatter_mean_bwd: bad args");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_voxel_count, dim3(semabs_cdiv(N, 256)), dim3(256), 0, s, flat, N, count);
    hipLaunchKernelGGL(k_scatter_mean_bwd, dim3(semabs_cdiv((long)P * N * (C / 4), 256)), dim3(256), 0, s, flat, count, dvol, dpf, P, N, C, nvox);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// =================================================================================================
// VOOL head, decomposed for training: trilinear sampling of the two volumes (+ qn) -> f [R, 36] (35 used, 1 pad), its
// backward (scatter of df into the two gradient volumes with the same corner weights), cosine pointer forward/backward.
// =================================================================================================
struct SampArgs { float off[3], sc[3]; int S0, S1, S2; };
__device__ __forceinline__ void samp_setup(const float* qp, const SampArgs& a, float* qn, int& x0, int& y0, int& z0, float* w) {
    const int S[3] = {a.S0, a.S1, a.S2};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float t = (qp[k] + a.off[k]) * a.sc[k];
        t = fminf(fmaxf(t, 0.f), (float)(S[k] - 1));
        t = t / (float)S[k];
        qn[k] = 2.0f * t - 1.0f;
    }
    float ix = ((qn[0] + 1.f) / 2.f) * (float)(a.S2 - 1), iy = ((qn[1] + 1.f) / 2.f) * (float)(a.S1 - 1), iz = ((qn[2] + 1.f) / 2.f) * (float)(a.S0 - 1);
    ix = fminf(fmaxf(ix, 0.f), (float)(a.S2 - 1)); iy = fminf(fmaxf(iy, 0.f), (float)(a.S1 - 1)); iz = fminf(fmaxf(iz, 0.f), (float)(a.S0 - 1));
    const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    x0 = (int)fx; y0 = (int)fy; z0 = (int)fz;
    w[0] = (fx + 1.f) - ix; w[1] = ix - fx; w[2] = (fy + 1.f) - iy; w[3] = iy - fy; w[4] = (fz + 1.f) - iz; w[5] = iz - fz;
}
// thread = (point, volume selector v in {0: target, 1: reference}): 16 channels
__global__ void k_vool_sample(const float* __restrict__ vol_t, const float* __restrict__ vol_r, const float* __restrict__ query, SampArgs a, int P,
                              long M, float* __restrict__ f) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)P * M * 2) return;
    const int sel = (int)(i & 1); const long pt = i >> 1; const int d = (int)(pt / M);
    float qn[3], w[6]; int x0, y0, z0;
    samp_setup(query + pt * 3, a, qn, x0, y0, z0, w);
    const float* vb = (sel ? vol_r : vol_t) + (long)d * a.S0 * a.S1 * a.S2 * 16;
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int zz = z0 + (k >> 2), yy = y0 + ((k >> 1) & 1), xx = x0 + (k & 1);
        if (zz > a.S0 - 1 || yy > a.S1 - 1 || xx > a.S2 - 1) continue;
        const float wt = w[k & 1] * w[2 + ((k >> 1) & 1)] * w[4 + (k >> 2)];
        const float4* p = reinterpret_cast<const float4*>(vb + (((long)zz * a.S1 + yy) * a.S2 + xx) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float4 v = p[q]; acc[4 * q] += v.x * wt; acc[4 * q + 1] += v.y * wt; acc[4 * q + 2] += v.z * wt; acc[4 * q + 3] += v.w * wt; }
    }
    float* fo = f + pt * 36 + sel * 16;
#pragma unroll
    for (int c = 0; c < 16; ++c) fo[c] = acc[c];
    if (sel == 0) { f[pt * 36 + 32] = qn[0]; f[pt * 36 + 33] = qn[1]; f[pt * 36 + 34] = qn[2]; f[pt * 36 + 35] = 0.f; }
}
// Backward of the sampling as a GATHER: points are first threaded into per-cell lists (cell = floor corner of the point), then every
// voxel of the gradient volumes sums w * df over the points of the 8 cells that touch it - no floating-point atomics (the scatter form
// issued 8 x 16 of them per point and volume and was atomic-throughput bound), and the volumes need no zero-fill.
__global__ void k_vool_cells(const float* __restrict__ query, SampArgs a, int P, long M, int* __restrict__ head, int* __restrict__ next) {
    const long pt = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pt >= (long)P * M) return;
    const int d = (int)(pt / M);
    float qn[3], w[6]; int x0, y0, z0;
    samp_setup(query + pt * 3, a, qn, x0, y0, z0, w);
    const long cell = (((long)d * a.S0 + z0) * a.S1 + y0) * a.S2 + x0;
    next[pt] = atomicExch(&head[cell], (int)(pt - (long)d * M));            // list entries are point indices within the description
}
// thread = (description, voxel): the 16 + 16 channels of BOTH gradient volumes in registers.  (One thread per (voxel, volume) walked every cell list twice and
// set up every visited point twice; with 0.2 points per cell almost every wave has some lane with a non-empty list at every corner, so the list walk - not
// the 64 B stores - is what the kernel costs: VALU issue 0.68 at 1.1 ms.)
__global__ __launch_bounds__(256) void k_vool_sample_bwd(const float* __restrict__ df, const float* __restrict__ query, SampArgs a, int P, long M,
                                                         const int* __restrict__ head, const int* __restrict__ next,
                                                         float* __restrict__ dvol_t, float* __restrict__ dvol_r, unsigned int* __restrict__ bits) {
    const long nvox = (long)a.S0 * a.S1 * a.S2;
    const long dv = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (dv >= (long)P * nvox) return;
    const int d = (int)(dv / nvox); long v = dv - (long)d * nvox;
    const int vx = (int)(v % a.S2); v /= a.S2;
    const int vy = (int)(v % a.S1); const int vz = (int)(v / a.S1);
    float acc[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) acc[c] = 0.f;
    int hp[8];                                              // the eight list heads first: independent loads, one memory latency instead of eight
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int cz = vz - (k >> 2), cy = vy - ((k >> 1) & 1), cx = vx - (k & 1);          // this voxel is corner k of cell (cz, cy, cx)
        hp[k] = (cz < 0 || cy < 0 || cx < 0) ? -1 : head[(((long)d * a.S0 + cz) * a.S1 + cy) * a.S2 + cx];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        int p = hp[k];
        while (p >= 0) {
            const long pt = (long)d * M + p;
            float qn[3], w[6]; int x0, y0, z0;
            samp_setup(query + pt * 3, a, qn, x0, y0, z0, w);
            const float wt = w[k & 1] * w[2 + ((k >> 1) & 1)] * w[4 + (k >> 2)];
            const float4* g = reinterpret_cast<const float4*>(df + pt * 36);
#pragma unroll
            for (int q = 0; q < 8; ++q) { const float4 gv = g[q]; acc[4 * q] += gv.x * wt; acc[4 * q + 1] += gv.y * wt; acc[4 * q + 2] += gv.z * wt; acc[4 * q + 3] += gv.w * wt; }
            p = next[pt];
        }
    }
    float4* ot = reinterpret_cast<float4*>(dvol_t + dv * 16);
    float4* orf = reinterpret_cast<float4*>(dvol_r + dv * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) { ot[q] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]); orf[q] = make_float4(acc[16 + 4 * q], acc[17 + 4 * q], acc[18 + 4 * q], acc[19 + 4 * q]); }
    if (bits) {                                             // max |dvol| for the dynamic gradient scale of the first backward layer (saves its pass over dvol)
        float m = 0.f;
#pragma unroll
        for (int c = 0; c < 32; ++c) m = fmaxf(m, fabsf(acc[c]));
        absmax_commit(bits, m);
    }
}
static void fill_samp(SampArgs& a, const float* off3, const float* sc3, const int* shape3) {
    for (int k = 0; k < 3; ++k) { a.off[k] = off3[k]; a.sc[k] = sc3[k]; }
    a.S0 = shape3[0]; a.S1 = shape3[1]; a.S2 = shape3[2];
}
// vol_t / vol_r fp32 [P, S, S, S, 16]; query fp32 [P, M, 3]; f fp32 [P*M, 36]
extern "C" int semabs_vool_sample(const float* vol_t, const float* vol_r, const float* query, const float* off3, const float* sc3, const int* shape3,
                                  int P, long M, float* f, void* stream) {
    if (P == 0 || M == 0) return SEMABS_OK;
    SEMABS_REQUIRE(vol_t && vol_r && query && off3 && sc3 && shape3 && f, "semabs_vool_sample: null pointer");
    SampArgs a; fill_samp(a, off3, sc3, shape3);
    hipLaunchKernelGGL(k_vool_sample, dim3(semabs_cdiv((long)P * M * 2, 256)), dim3(256), 0, (hipStream_t)stream, vol_t, vol_r, query, a, P, M, f);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}
// df fp32 [P*M, 36] -> dvol_t / dvol_r fp32 [P, S, S, S, 16] (fully written).  head int32 [P * S^3] and next int32 [P * M] are scratch.
extern "C" int semabs_vool_sample_bwd(const float* df, const float* query, const float* off3, const float* sc3, const int* shape3, int P, long M,
                                      int* head, int* next, float* dvol_t, float* dvol_r, unsigned int* absmax_bits, void* stream) {
    if (P == 0) return SEMABS_OK;
    SEMABS_REQUIRE(df && query && off3 && sc3 && shape3 && head && next && dvol_t && dvol_r, "semabs_vool_sample_bwd: null pointer");
    SEMABS_REQUIRE(M < (1L << 31), "semabs_vool_sample_bwd: at most 2^31 - 1 query points per description");
    SampArgs a; fill_samp(a, off3, sc3, shape3);
    hipStream_t s = (hipStream_t)stream;
    const long nvox = (long)a.S0 * a.S1 * a.S2;
    semabs_fill32(head, sizeof(int) * P * nvox, 0xffffffffu, s);
    if (M > 0) hipLaunchKernelGGL(k_vool_cells, dim3(semabs_cdiv((long)P * M, 256)), dim3(256), 0, s, query, a, P, M, head, next);
    hipLaunchKernelGGL(k_vool_sample_bwd, dim3(semabs_cdiv((long)P * nvox, 256)), dim3(256), 0, s, df, query, a, P, M, head, next, dvol_t, dvol_r, absmax_bits);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// cosine pointer (+ BCE): logit = cos(o, rel_d) / T;  BCE mode (label != null): loss += w * BCEwithlogits(logit, y) / n_total and dz = d loss / d logit is
// formed here; head mode: dz = dz_in (null = logits only).  16 lanes per point (E = 64 = one float4 per lane), four points per wave, row reductions by DPP:
// ~25 instructions per point (one wave per point with two 64-lane reductions and the transcendentals in every lane: ~120, 760 us for 1.6 M points - VALU-bound
// at 0.6 GB of traffic).  Writes logits, dO [R, 64]; accumulates drel [P, 64] and the loss.
__global__ __launch_bounds__(256) void k_cos_rows(const float* __restrict__ o, const float* __restrict__ rel, const float* __restrict__ label,
                                                  const float* __restrict__ weight, const float* __restrict__ dz_in, int P, long M, float inv_temp, float inv_n,
                                                  float* __restrict__ logits, float* __restrict__ dO, float* __restrict__ drel, double* __restrict__ loss,
                                                  float* __restrict__ dbias) {
    __shared__ float s_drel[64];
    __shared__ float s_db[64];
    if (threadIdx.x < 64) s_db[threadIdx.x] = 0.f;
    float acc_db[4] = {0.f, 0.f, 0.f, 0.f};                 // column sums of dO (the bias gradient of the layer that produced o), per description
    __shared__ float s_loss;
    if (threadIdx.x < 64) s_drel[threadIdx.x] = 0.f;
    if (threadIdx.x == 0) s_loss = 0.f;
    __syncthreads();
    const int q = threadIdx.x & 15;
    // blocks are aligned to descriptions: blockIdx.y = description, blockIdx.x strides over its M points (16 per iteration)
    const int d = blockIdx.y;
    const float4 rv = *reinterpret_cast<const float4*>(rel + d * 64 + q * 4);
    const float nr = fmaxf(sqrtf(row16_sum((rv.x * rv.x + rv.y * rv.y) + (rv.z * rv.z + rv.w * rv.w))), 1e-8f);
    const float inr = 1.f / nr;
    const float rh[4] = {rv.x / nr, rv.y / nr, rv.z / nr, rv.w / nr};
    const bool grad = label || dz_in;
    float acc_drel[4] = {0.f, 0.f, 0.f, 0.f}, acc_loss = 0.f;
    for (long m = (long)blockIdx.x * 16 + (threadIdx.x >> 4); m < M; m += (long)gridDim.x * 16) {
        const long pt = (long)d * M + m;
        const float4 ov = *reinterpret_cast<const float4*>(o + pt * 64 + q * 4);
        const float no = fmaxf(sqrtf(row16_sum((ov.x * ov.x + ov.y * ov.y) + (ov.z * ov.z + ov.w * ov.w))), 1e-8f);
        const float oh[4] = {ov.x / no, ov.y / no, ov.z / no, ov.w / no};
        const float cs = row16_sum((oh[0] * rh[0] + oh[1] * rh[1]) + (oh[2] * rh[2] + oh[3] * rh[3]));
        const float z = cs * inv_temp;
        if (logits && q == 0) logits[pt] = z;
        if (!grad) continue;
        float dz;
        if (label) {
            const float y = label[pt], wgt = weight ? weight[pt] : 1.f;
            // numerically stable BCE with logits: max(z, 0) - z y + log(1 + exp(-|z|))
            const float l = fmaxf(z, 0.f) - z * y + log1pf(__expf(-fabsf(z)));
            dz = (1.f / (1.f + __expf(-z)) - y) * wgt * inv_n;
            if (q == 0) acc_loss += l * wgt * inv_n;
        } else dz = dz_in[pt];
        const float dc = dz * inv_temp, ino = dc / no, dcr = dc * inr;
        const float dv[4] = {(rh[0] - oh[0] * cs) * ino, (rh[1] - oh[1] * cs) * ino, (rh[2] - oh[2] * cs) * ino, (rh[3] - oh[3] * cs) * ino};
        *reinterpret_cast<float4*>(dO + pt * 64 + q * 4) = make_float4(dv[0], dv[1], dv[2], dv[3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc_drel[j] += (oh[j] - rh[j] * cs) * dcr; acc_db[j] += dv[j]; }
    }
    if (!grad) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) { atomicAdd(&s_drel[q * 4 + j], acc_drel[j]); if (dbias) atomicAdd(&s_db[q * 4 + j], acc_db[j]); }
    if (label && q == 0) atomicAdd(&s_loss, acc_loss);
    __syncthreads();
    if (threadIdx.x < 64) { atomicAdd(&drel[d * 64 + threadIdx.x], s_drel[threadIdx.x]); if (dbias) atomicAdd(&dbias[d * 64 + threadIdx.x], s_db[threadIdx.x]); }
    if (label && threadIdx.x == 0) atomicAdd(loss, (double)s_loss);
}
// o fp32 [P*M, 64]; rel fp32 [P, 64]; label / weight fp32 [P*M] (weight optional); outputs: logits [P*M] (optional), dO [P*M, 64],
// drel fp32 [P, 64] and loss fp64 [1] ACCUMULATED (zero them first).  n_total = element count of the mean reduction.  dbias (optional) fp32 [P, 64], accumulated:
// the column sums of dO per description (summed over P they are the bias gradient of the layer that produced o).
extern "C" int semabs_cos_bce(const float* o, const float* rel, const float* label, const float* weight, int P, long M, float temperature,
                              long n_total, float* logits, float* dO, float* drel, double* loss, float* dbias, void* stream) {
    if (P == 0 || M == 0) return SEMABS_OK;
    SEMABS_REQUIRE(o && rel && label && dO && drel && loss && temperature > 0.f && n_total > 0, "semabs_cos_bce: bad args");
    int bx = semabs_cdiv(M, 16 * 8); if (bx < 1) bx = 1; if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(k_cos_rows, dim3(bx, P), dim3(256), 0, (hipStream_t)stream, o, rel, label, weight, (const float*)nullptr, P, M, 1.0f / temperature,
                       1.0f / (float)n_total, logits, dO, drel, loss, dbias);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// The same pointer head with the loss left to the CALLER (the reference's own loop: `loss = BCE(net(**batch), label); loss.backward()`,
// train_vool.py:171-178, utils.py:404-417 - semabs_amd.net.SemAbsVOOL under autograd): dz == NULL -> logits only (forward);
// dz = d loss / d logits [P*M] -> dO = d loss / d o and drel (accumulated) = d loss / d rel.
extern "C" int semabs_cos_head(const float* o, const float* rel, const float* dlogits, int P, long M, float temperature, float* logits, float* dO,
                               float* drel, void* stream) {
    if (P == 0 || M == 0) return SEMABS_OK;
    SEMABS_REQUIRE(o && rel && temperature > 0.f && (dlogits ? (dO && drel) : (logits != nullptr)), "semabs_cos_head: bad args");
    int bx = semabs_cdiv(M, 16 * 8); if (bx < 1) bx = 1; if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(k_cos_rows, dim3(bx, P), dim3(256), 0, (hipStream_t)stream, o, rel, (const float*)nullptr, (const float*)nullptr, dlogits, P, M,
                       1.0f / temperature, 0.f, logits, dO, drel, (double*)nullptr, (float*)nullptr);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// =================================================================================================
// Dynamic per-tensor gradient scale for the split-fp16 data-gradient convolutions: the MFMA operands are fp16 hi + lo pairs, which
// keep ~22 bits only for magnitudes inside fp16's normal range, while gradients are routinely 1e-6 and smaller.  s = 2^k puts the
// tensor's max |x| into [256, 512); the convolution applies it through its (GroupNorm-style) per-(b, c) input affine, and the
// consumer multiplies by 1 / s.  Powers of two: the scaling itself is exact.
// =================================================================================================
__global__ __launch_bounds__(256) void k_absmax(const float* __restrict__ x, long n4, unsigned int* __restrict__ bits) {
    float m = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = *reinterpret_cast<const float4*>(x + i * 4);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    absmax_commit(bits, m);                                 // non-negative floats order like their bit patterns
}
__global__ void k_scale_fill(const unsigned int* __restrict__ bits, float* __restrict__ scale_arr, float* __restrict__ shift_arr, int n_arr,
                             float* __restrict__ s2) {
    const float m = __uint_as_float(*bits);
    float s = 1.f;
    if (m > 0.f && m < INFINITY) { int e = 8 - ilogbf(m); e = e < -100 ? -100 : (e > 100 ? 100 : e); s = ldexpf(1.f, e); }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_arr) { scale_arr[i] = s; shift_arr[i] = 0.f; }
    if (i == 0) { s2[0] = s; s2[1] = 1.f / s; }
}
// x fp32 [n] (n % 4 == 0) -> scale_arr[n_arr] = s, shift_arr[n_arr] = 0 (the conv's input affine), s2 = (s, 1 / s); bits: uint32 scratch, or
// (have_bits = 1) the max |x| bit pattern already produced by semabs_ew, in which case x is not read again; have_bits = 2: bits is already zero
// (sub-allocated from the step's zeroed arena), so no memset is queued
extern "C" int semabs_grad_scale(const float* x, long n, float* scale_arr, float* shift_arr, int n_arr, float* s2, unsigned int* bits, int have_bits,
                                 void* stream) {
    SEMABS_REQUIRE(scale_arr && shift_arr && s2 && bits && n_arr > 0 && (have_bits == 1 || (x && n > 0 && n % 4 == 0)), "semabs_grad_scale: bad args");
    hipStream_t s = (hipStream_t)stream;
    if (have_bits != 1) {
        if (!have_bits) semabs_fill32(bits, sizeof(unsigned int), 0u, s);
        int bx = semabs_cdiv(n / 4, 256 * 8); if (bx > 2048) bx = 2048;
        hipLaunchKernelGGL(k_absmax, dim3(bx), dim3(256), 0, s, x, n / 4, bits);
    }
    hipLaunchKernelGGL(k_scale_fill, dim3(semabs_cdiv(n_arr, 256)), dim3(256), 0, s, bits, scale_arr, shift_arr, n_arr, s2);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// =================================================================================================
// Global gradient norm over a chunk table (same format as semabs_lamb_step) and in-place scaling of all gradients:
// clip_grad_norm_: g *= min(1, max_norm / (norm + 1e-6))
// =================================================================================================
__global__ __launch_bounds__(256) void k_grad_sqsum(const long long* __restrict__ chunks, const long long* __restrict__ ptrs, int n_tensors, double* __restrict__ sq) {
    const long long tid_ = chunks[blockIdx.x * 3], off = chunks[blockIdx.x * 3 + 1], cnt = chunks[blockIdx.x * 3 + 2];
    const float* g = reinterpret_cast<const float*>(ptrs[n_tensors + tid_]) + off;
    float s = 0.f;
    for (long long i = threadIdx.x; i < cnt; i += 256) s += g[i] * g[i];
    s = wave_sum(s);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sq, (double)((red[0] + red[1]) + (red[2] + red[3])));
}
__global__ __launch_bounds__(256) void k_grad_scale(const long long* __restrict__ chunks, const long long* __restrict__ ptrs, int n_tensors,
                                                    const double* __restrict__ sq, float max_norm, float extra_scale) {
    const long long tid_ = chunks[blockIdx.x * 3], off = chunks[blockIdx.x * 3 + 1], cnt = chunks[blockIdx.x * 3 + 2];
    float* g = reinterpret_cast<float*>(ptrs[n_tensors + tid_]) + off;
    const float norm = (float)sqrt(*sq) * extra_scale;
    float coef = max_norm / (norm + 1e-6f); if (coef > 1.f) coef = 1.f;
    coef *= extra_scale;
    if (coef == 1.f) return;
    for (long long i = threadIdx.x; i < cnt; i += 256) g[i] *= coef;
}
// sq fp64 [1] scratch (zeroed here) holds sum g^2 afterwards; gradients are first multiplied by extra_scale (e.g. 1 / world size)
extern "C" int semabs_clip_grad_norm(const long long* chunks, int n_chunks, const long long* ptrs, int n_tensors, float max_norm, float extra_scale,
                                     double* sq, void* stream) {
    if (n_chunks == 0) return SEMABS_OK;
    SEMABS_REQUIRE(chunks && ptrs && sq && max_norm > 0.f, "semabs_clip_grad_norm: bad args");
    hipStream_t s = (hipStream_t)stream;
    semabs_fill32(sq, sizeof(double), 0u, s);
    hipLaunchKernelGGL(k_grad_sqsum, dim3(n_chunks), dim3(256), 0, s, chunks, ptrs, n_tensors, sq);
    hipLaunchKernelGGL(k_grad_scale, dim3(n_chunks), dim3(256), 0, s, chunks, ptrs, n_tensors, sq, max_norm, extra_scale);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// =================================================================================================
// Conv3d 3x3x3 weight gradient on MFMA, for volumes that tile into 4 x 8 x 16 bricks (levels 0-3 of the 128^3 UNet).
//   dW[ca][tap][cx] += inv_s * sum_vox (s * dZ[vox][ca]) * GN(X)[vox + tap][cx]
// The contraction runs over VOXELS, so both MFMA operands must be voxel-major while HBM holds them channel-major: a persistent
// workgroup (9 waves) transposes one brick at a time into LDS - the 6 x 10 x 18 halo of X (GroupNorm affine applied, zero padded) and
// the 4 x 8 x 16 brick of dZ (times the dynamic gradient scale s), each as split fp16 hi + lo, laid out [z][y][channel][x] so that 8
// consecutive x of one channel are one aligned 16-byte ds_read.  Wave w owns the tap pairs (dz, dy) = (w / 3 - 1, w % 3 - 1); its
// three dx taps share one aligned row read, the +-1 shifts are done in registers with v_alignbit on one extra dword.  Every staged
// element feeds 27 x 16 MACs; the 27 x 16 x 16 partial result stays in registers across all bricks and is added to dW once.
// Channel counts above 16 are sliced: blockIdx.y = ca slice, blockIdx.z = cx slice.
// =================================================================================================
#define WG_T0 4
#define WG_T1 8
#define WG_T2 16
#define WG_H0 (WG_T0 + 2)
#define WG_H1 (WG_T1 + 2)
#define WG_XROW 24                           // element stride of the (z, y, channel) rows of the X halo: x = -1 .. 16 sit at 7 .. 24 of a row, i.e. a
                                             // row's last two slots are the (never used) first two of the next row.  48 B = 3 x 16 B: the 16 channel
                                             // rows of a fragment read land in 16 different 16-byte slots of the 64 banks (at 64 B they shared 4)
#define WG_AYROW (16 * WG_T2 + 8)            // element stride of a (z, y) block of dZ rows: the + 16 B puts the two y rows of a lane group in
                                             // different slots
#define WG_NTHR 576
struct Wgrad16Args {
    const float* A; const float* X; const float* gn_scale; const float* gn_shift; const float* s2; float* dW;
    int B, D0, D1, D2, Ca, Cx, tap_minor;
};

__device__ __forceinline__ unsigned int pack_hi_lo(float v0, float v1, unsigned int& lo) {
    const f16 h0 = (f16)v0, h1 = (f16)v1;
    const f16 l0 = (f16)(v0 - (float)h0), l1 = (f16)(v1 - (float)h1);
    lo = (unsigned int)__builtin_bit_cast(unsigned short, l0) | ((unsigned int)__builtin_bit_cast(unsigned short, l1) << 16);
    return (unsigned int)__builtin_bit_cast(unsigned short, h0) | ((unsigned int)__builtin_bit_cast(unsigned short, h1) << 16);
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(WG_NTHR) void k_wgrad16_lds(Wgrad16Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int XH_EL = WG_H0 * WG_H1 * 16 * WG_XROW + 8;        // elements in one X plane (hi or lo); + 8: the last row's tail
    constexpr int A_EL = WG_T0 * WG_T1 * WG_AYROW;
    unsigned short* sXh = reinterpret_cast<unsigned short*>(smem);
    unsigned short* sXl = sXh + XH_EL;
    unsigned short* sAh = sXl + XH_EL;
    unsigned short* sAl = sAh + A_EL;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ca0 = blockIdx.y * 16, cx0 = blockIdx.z * 16;
    const int n0 = a.D0 / WG_T0, n1 = a.D1 / WG_T1, n2 = a.D2 / WG_T2;
    const int nbricks = a.B * n0 * n1 * n2;
    const float s_in = a.s2 ? a.s2[0] : 1.f;
    const int dz = wid / 3 - 1, dy = wid % 3 - 1;
    const int i16 = lane & 15, kg = lane >> 4;
    f32x4 acc[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};

    for (int brick = blockIdx.x; brick < nbricks; brick += gridDim.x) {
        int t = brick;
        const int t2 = t % n2; t /= n2;
        const int t1 = t % n1; t /= n1;
        const int t0 = t % n0; const int b = t / n0;
        const int z0 = t0 * WG_T0, y0 = t1 * WG_T1, x0 = t2 * WG_T2;
        __syncthreads();                                            // previous brick's fragments are no longer being read
        // ---- stage the X halo: task = (hz, hy, pair of x); elements 6 + 2 px, 7 + 2 px  <->  x = 2 px - 2, 2 px - 1 ----
        for (int task = tid; task < WG_H0 * WG_H1 * 10; task += WG_NTHR) {
            const int px = task % 10, hy = (task / 10) % WG_H1, hz = task / (10 * WG_H1);
            const int gz = z0 + hz - 1, gy = y0 + hy - 1;
            float v[2][16];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int xl = 2 * px - 2 + e;                      // local x, -2 .. 17 (only -1 .. 16 are ever read)
                const int gx = x0 + xl;
                const bool ok = xl >= -1 && xl <= WG_T2 && gz >= 0 && gz < a.D0 && gy >= 0 && gy < a.D1 && gx >= 0 && gx < a.D2;
                if (ok) {
                    const float4* p = reinterpret_cast<const float4*>(a.X + ((((long)b * a.D0 + gz) * a.D1 + gy) * a.D2 + gx) * a.Cx + cx0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const float4 w = p[q]; v[e][4 * q] = w.x; v[e][4 * q + 1] = w.y; v[e][4 * q + 2] = w.z; v[e][4 * q + 3] = w.w; }
                    if (a.gn_scale) {
#pragma unroll
                        for (int c = 0; c < 16; ++c) v[e][c] = v[e][c] * a.gn_scale[(long)b * a.Cx + cx0 + c] + a.gn_shift[(long)b * a.Cx + cx0 + c];
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 16; ++c) v[e][c] = 0.f;
                }
            }
            const int rowbase = ((hz * WG_H1 + hy) * 16) * WG_XROW + 6 + 2 * px;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                unsigned int lo;
                const unsigned int hi = pack_hi_lo(v[0][c], v[1][c], lo);
                *reinterpret_cast<unsigned int*>(sXh + rowbase + c * WG_XROW) = hi;
                *reinterpret_cast<unsigned int*>(sXl + rowbase + c * WG_XROW) = lo;
            }
        }
        // ---- stage the dZ brick: task = (z, y, pair of x) ----
        for (int task = tid; task < WG_T0 * WG_T1 * 8; task += WG_NTHR) {
            const int px = task % 8, yy = (task / 8) % WG_T1, zz = task / (8 * WG_T1);
            float v[2][16];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float4* p = reinterpret_cast<const float4*>(a.A + ((((long)b * a.D0 + z0 + zz) * a.D1 + y0 + yy) * a.D2 + x0 + 2 * px + e) * a.Ca + ca0);
#pragma unroll
                for (int q = 0; q < 4; ++q) { const float4 w = p[q]; v[e][4 * q] = w.x * s_in; v[e][4 * q + 1] = w.y * s_in; v[e][4 * q + 2] = w.z * s_in; v[e][4 * q + 3] = w.w * s_in; }
            }
            const int rowbase = (zz * WG_T1 + yy) * WG_AYROW + 2 * px;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                unsigned int lo;
                const unsigned int hi = pack_hi_lo(v[0][c], v[1][c], lo);
                *reinterpret_cast<unsigned int*>(sAh + rowbase + c * WG_T2) = hi;
                *reinterpret_cast<unsigned int*>(sAl + rowbase + c * WG_T2) = lo;
            }
        }
        __syncthreads();
        // ---- 16 k-steps of 32 voxels (2 rows x 16 x): A = dZ[ca][vox], B = X[vox + tap][cx] ----
#pragma unroll 2
        for (int ks = 0; ks < 16; ++ks) {
            // lane group kg -> (y row, x half) = (kg & 1, kg >> 1): the ds_read_b128 lane groups pair kg 0 with kg 1 (and 2 with 3), i.e. two
            // y rows of the same x half - with the strides above every group of 16 lanes covers 16 distinct 16-byte bank slots (before:
            // SQ_LDS_BANK_CONFLICT = 73 % of the LDS-active cycles, and LDS was the busiest unit of the kernel)
            const int zz = ks >> 2, yy = (ks & 3) * 2 + (kg & 1), xg = kg >> 1;
            const int aoff = (zz * WG_T1 + yy) * WG_AYROW + i16 * WG_T2 + xg * 8;
            const f16x8 ah = *reinterpret_cast<const f16x8*>(sAh + aoff);
            const f16x8 al = *reinterpret_cast<const f16x8*>(sAl + aoff);
            const int boff = (((zz + 1 + dz) * WG_H1 + (yy + 1 + dy)) * 16 + i16) * WG_XROW + 8 + xg * 8;
            const u32x4 gh = *reinterpret_cast<const u32x4*>(sXh + boff);
            const u32x4 gl = *reinterpret_cast<const u32x4*>(sXl + boff);
            const unsigned int ph = *reinterpret_cast<const unsigned int*>(sXh + boff - 2), pl = *reinterpret_cast<const unsigned int*>(sXl + boff - 2);
            const unsigned int nh = *reinterpret_cast<const unsigned int*>(sXh + boff + 8), nl = *reinterpret_cast<const unsigned int*>(sXl + boff + 8);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                u32x4 bh, bl;
                if (d == 1) { bh = gh; bl = gl; }
                else if (d == 2) {
                    bh = u32x4{__builtin_amdgcn_alignbit(gh[1], gh[0], 16), __builtin_amdgcn_alignbit(gh[2], gh[1], 16), __builtin_amdgcn_alignbit(gh[3], gh[2], 16), __builtin_amdgcn_alignbit(nh, gh[3], 16)};
                    bl = u32x4{__builtin_amdgcn_alignbit(gl[1], gl[0], 16), __builtin_amdgcn_alignbit(gl[2], gl[1], 16), __builtin_amdgcn_alignbit(gl[3], gl[2], 16), __builtin_amdgcn_alignbit(nl, gl[3], 16)};
                } else {
                    bh = u32x4{__builtin_amdgcn_alignbit(gh[0], ph, 16), __builtin_amdgcn_alignbit(gh[1], gh[0], 16), __builtin_amdgcn_alignbit(gh[2], gh[1], 16), __builtin_amdgcn_alignbit(gh[3], gh[2], 16)};
                    bl = u32x4{__builtin_amdgcn_alignbit(gl[0], pl, 16), __builtin_amdgcn_alignbit(gl[1], gl[0], 16), __builtin_amdgcn_alignbit(gl[2], gl[1], 16), __builtin_amdgcn_alignbit(gl[3], gl[2], 16)};
                }
                const f16x8 xh = __builtin_bit_cast(f16x8, bh), xl = __builtin_bit_cast(f16x8, bl);
                acc[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xh, acc[d], 0, 0, 0);
                acc[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, xh, acc[d], 0, 0, 0);
                acc[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xl, acc[d], 0, 0, 0);
            }
        }
    }
    // acc[d][r] = dW[ca = 4 * kg + r][tap (dz, dy, d - 1)][cx = i16].  Park the 16 x 16 x 27 block in LDS in the order of the output
    // layout, then flush it with lane-consecutive atomics (for a fixed ca the block is one contiguous run of the output row).
    const float inv = a.s2 ? a.s2[1] : 1.f;
    float* sOut = reinterpret_cast<float*>(smem);
    __syncthreads();
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int tap = ((dz + 1) * 3 + (dy + 1)) * 3 + d;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ca = 4 * kg + r;
            sOut[a.tap_minor ? (ca * 16 + i16) * 27 + tap : (ca * 27 + tap) * 16 + i16] = acc[d][r] * inv;
        }
    }
    __syncthreads();
    const long N = 27L * a.Cx;
    for (int e = tid; e < 16 * 432; e += WG_NTHR) {
        const int ca = e / 432, rem = e - ca * 432;
        long idx;
        if (a.tap_minor) idx = ((long)(ca0 + ca) * a.Cx + cx0) * 27 + rem;                       // rem = cx * 27 + tap
        else idx = (long)(ca0 + ca) * N + (long)(rem >> 4) * a.Cx + cx0 + (rem & 15);            // rem = tap * 16 + cx
        atomicAdd(&a.dW[idx], sOut[e]);
    }
}

// =================================================================================================
// The same weight gradient with the operands in their NATURAL order in LDS ([voxel][16 channels] fp16, 32 B per voxel, hi and lo planes)
// and the voxel-major fragments produced by the transposing LDS read (ds_read_b64_tr_b16, as in k_wgrad_mfma): staging is an 8-byte store per
// (voxel, 4 channels) instead of sixteen 4-byte scatter stores per voxel pair, a fragment read of 16 consecutive x is one contiguous 512 B
// (conflict-free at any dx shift, so the three dx taps are plain address offsets - no v_alignbit), and 4 x 4 x 16 bricks (58 KB) leave room for
// the accumulators.
//
// Round 5: a workgroup = 16 waves, one per CU, in two roles (the round-4 kernel did both jobs in every wave, one after the other, and was bound by VALU
// issue - ~700 VALU instructions per thread and brick for addresses, the fp16 hi / lo split and operand assembly, beside 84 MFMAs - at 0.28 matrix-pipe busy):
//  * waves 8 .. 15 LOAD: brick i + 2's global loads are in flight in registers while brick i + 1 is split into fp16 hi / lo planes in the LDS buffer the
//    multiplying waves are not reading (two 58 KB buffers); one barrier per brick.  A slot's address is a per-thread constant plus a per-brick scalar
//    (see `relx` below), whether a slot is outside the volume comes from six per-thread bit sets: ~25 VALU instructions per 16-byte slot, all of them the
//    GroupNorm affine and the split.
//  * waves 0 .. 7 MULTIPLY: wave w takes z slab w & 3 of the brick and half of the 27 taps.  A k-step is the pair of x rows (kk, kk + 2) of the slab
//    (kk = 0, 1), so that tap dy of k-step kk needs the halo rows (j, j + 2) with j = dy + 1 + kk: the SAME fragment serves (kk = 0, dy) and (kk = 1, dy - 1),
//    and for one tap group g = (dz, dx) the four fragments j = 0 .. 3 feed all six (kk, dy) products (with rows (kk, kk + 1) a product needed its own).
//    A slab's nine groups are split 5 + 4 between the two waves that share its SIMD (the slabs alternate which of the two gets five: 81 MFMAs per brick and
//    SIMD pair, no tap computed twice); a group's six products run kk-major - hi.hi for all six, then lo.hi, then hi.lo - so that MFMAs on one accumulator
//    are three apart.
// The workgroup's 27 x 16 x 16 sums meet in LDS and go to a partial-sum buffer, one row per workgroup; k_wgrad3_reduce adds the rows into dW (no atomics:
// deterministic up to the order of four LDS float adds per element, and 16 G atomics / s was a third of the round-2 kernel).
// Measured (tools/wgrad_time.py, [8, 128^3, 16 x 16], random operands, 1 500 launches back to back): 740 us at 1 869 MHz and 1 391 W of the 1 400 W cap,
// against 955 us at 2 365 MHz and 1 374 W before - BOTH versions run at the socket's power cap, so the time is the energy a launch takes (1.03 J vs 1.31 J);
// the multiplying waves alone (no loads, no split) take 494 us at 1 199 W.  In the training step (post-ReLU gradients: half the operands are zeros) the six
// level-0 launches take 700 - 800 us each (1 000 - 1 230 before), the step's weight gradients 8.7 ms (12.4).
// =================================================================================================
__device__ __forceinline__ constexpr int w3_frag_off(int g, int j) { return (((g / 3) * 6 + j) * 18 + g % 3) * 32; }   // bytes from the slab's first halo voxel
struct Wgrad3Args {
    const float* A; const float* X; const float* gn_scale; const float* gn_shift; const float* s2; float* part;
    int B, D0, D1, D2, Ca, Cx;
    // per-volume mode (semabs_wgrad_conv3_gn): X is staged as xhat = (x - mean) rstd of its GroupNorm (no affine), every workgroup stays inside ONE volume
    // (blockIdx.x % B) and its partial row also carries Q = the 27 boundary-restricted sums of dZ (W3_ROW_PV floats per row); rows keep the gradient scale.
    const float* gn_mean; const float* gn_rstd; int G, per_vol;
};
#define W3_ROW 6912
#define W3_ROW_PV (6912 + 16)
#define W3_HV (6 * 6 * 18)
#define W3_LDS (2 * W3_HV * 32 + 2 * 256 * 32)
#ifndef W3_NLD
#define W3_NLD 512                                          // loader threads (8 waves) beside the 512 multiplying ones
#endif
#define W3_NTHR (512 + W3_NLD)
#define W3_NX ((W3_HV * 4 + W3_NLD - 1) / W3_NLD)
#define W3_NA (1024 / W3_NLD)

__global__ __launch_bounds__(W3_NTHR) void k_wgrad3_tr(Wgrad3Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const sG = reinterpret_cast<float*>(smem + 2 * W3_LDS);  // [B][scale 16 | shift 16] of this channel slice (no global loads per brick)
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool loader = w >= 8;                             // waves 8 .. 15 stage bricks, waves 0 .. 7 multiply them (see the header comment)
    const int tid = loader ? (int)threadIdx.x - 512 : (int)threadIdx.x;   // index within the role
    const int kp = w & 3, half = (w >> 2) & 1;              // k-steps 2 kp, 2 kp + 1; taps 13 half .. 13 half + 13
    const int ca0 = blockIdx.y * 16, cx0 = blockIdx.z * 16;
    const int n0 = a.D0 / 4, n1 = a.D1 / 4, n2 = a.D2 / 16;
    const float s_in = a.s2 ? a.s2[0] : 1.f;
    const int q = tid & 3;                                  // this thread's 4-channel quarter in every staging slot
    float* const sQ = sG + a.B * 32;                        // [16 ca]: per-volume mode's sum of dZ over the workgroup's bricks
    for (int e = threadIdx.x; e < a.B * 32; e += W3_NTHR) {
        const int b = e >> 5, c = e & 15;
        if (a.gn_mean) {
            const int g = (cx0 + c) / (a.Cx / a.G);
            const float rs = a.gn_rstd[b * a.G + g];
            sG[e] = (e & 16) ? -a.gn_mean[b * a.G + g] * rs : rs;
        } else
            sG[e] = a.gn_scale ? ((e & 16) ? a.gn_shift[(long)b * a.Cx + cx0 + c] : a.gn_scale[(long)b * a.Cx + cx0 + c]) : ((e & 16) ? 0.f : 1.f);
    }
    if (threadIdx.x < 16) sQ[threadIdx.x] = 0.f;
    float4 px[W3_NX], pa[W3_NA];
    unsigned okmask = 0;
    // Staging addresses.  A slot's voxel differs from the brick's first halo voxel by a fixed (hz, hy, hx), so its byte offset is a per-thread constant
    // `rel` plus a per-brick UNIFORM base (scalar arithmetic): one v_add per load.  Nothing is clamped: the arithmetic wraps modulo 2^32, the buffer
    // descriptors carry the tensors' exact sizes (< 2^31 bytes, the launcher's guarantee), so a slot outside the volume either lands outside the
    // descriptor (the load returns 0) or on some other voxel of the tensor - and store() zeroes every such slot: whether slot i is outside depends
    // only on which faces of the volume the brick touches (uniform, 6 bits) and on which faces of the halo the slot lies on (six per-thread bit sets
    // `face[k]` over i).  Before, three clamps, three compares and three 24-bit multiplies per slot made the fetch ~250 VALU instructions per brick
    // and thread - and VALU issue, not the matrix pipe or the LDS, was what bounded this kernel.
    unsigned relx[W3_NX], rela[W3_NA], face[6] = {0u, 0u, 0u, 0u, 0u, 0u};
    const unsigned xq = (unsigned)(cx0 + q * 4) * 4u, aq = (unsigned)(ca0 + q * 4) * 4u, xvb = (unsigned)a.Cx * 4u, avb = (unsigned)a.Ca * 4u;
#pragma unroll
    for (int i = 0; i < W3_NX; ++i) {
        int hv = (i * W3_NLD + tid) >> 2; hv = hv < W3_HV ? hv : W3_HV - 1;
        const int hz = hv / 108, hy = (hv / 18) % 6, hx = hv % 18;
        relx[i] = (unsigned)((hz * a.D1 + hy) * a.D2 + hx) * xvb + xq;
        face[0] |= (hz == 0 ? 1u : 0u) << i; face[1] |= (hz == 5 ? 1u : 0u) << i;
        face[2] |= (hy == 0 ? 1u : 0u) << i; face[3] |= (hy == 5 ? 1u : 0u) << i;
        face[4] |= (hx == 0 ? 1u : 0u) << i; face[5] |= (hx == 17 ? 1u : 0u) << i;
    }
#pragma unroll
    for (int i = 0; i < W3_NA; ++i) {
        const int av = (i * W3_NLD + tid) >> 2;
        rela[i] = (unsigned)(((av >> 6) * a.D1 + ((av >> 4) & 3)) * a.D2 + (av & 15)) * avb + aq;
    }
    float tsum[4] = {0.f, 0.f, 0.f, 0.f};                   // per-volume mode: this thread's running sum of its four dZ channels over all its slots
    const unsigned vox_all = (unsigned)a.B * (unsigned)(a.D0 * a.D1) * (unsigned)a.D2;
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X), 0, (int)(vox_all * xvb), 0x00020000);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.A), 0, (int)(vox_all * avb), 0x00020000);
    auto fetch = [&](int brick) {
        int t = brick;                                     // (uniform: scalar arithmetic)
        const int t2 = t % n2; t /= n2;
        const int t1 = t % n1; t /= n1;
        const int t0 = t % n0; const int b = t / n0;
        const int z0 = t0 * 4, y0 = t1 * 4, x0 = t2 * 16;
        const unsigned bx = (unsigned)(((b * a.D0 + z0 - 1) * a.D1 + (y0 - 1)) * a.D2 + (x0 - 1)) * xvb;
        const unsigned ba = (unsigned)(((b * a.D0 + z0) * a.D1 + y0) * a.D2 + x0) * avb;
        okmask = ~((z0 == 0 ? face[0] : 0u) | (z0 + 4 == a.D0 ? face[1] : 0u) | (y0 == 0 ? face[2] : 0u) | (y0 + 4 == a.D1 ? face[3] : 0u) |
                   (x0 == 0 ? face[4] : 0u) | (x0 + 16 == a.D2 ? face[5] : 0u));
#pragma unroll
        for (int i = 0; i < W3_NX; ++i) { const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rX, relx[i] + bx, 0, 0)); px[i] = make_float4(v[0], v[1], v[2], v[3]); }
#pragma unroll
        for (int i = 0; i < W3_NA; ++i) { const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, rela[i] + ba, 0, 0)); pa[i] = make_float4(v[0], v[1], v[2], v[3]); }
    };
    auto split_store = [&](const float f[4], char* hi, char* lo) {
        f16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) { h[e] = (f16)f[e]; l[e] = (f16)(f[e] - (float)h[e]); }
        *reinterpret_cast<f16x4*>(hi) = h; *reinterpret_cast<f16x4*>(lo) = l;
    };
    auto store = [&](int brick, char* buf) {
        char* const sXh = buf; char* const sXl = sXh + W3_HV * 32; char* const sAh = sXl + W3_HV * 32; char* const sAl = sAh + 256 * 32;
        const int b = brick / (n0 * n1 * n2);
        const float4 gsc = *reinterpret_cast<const float4*>(sG + b * 32 + q * 4), gsh = *reinterpret_cast<const float4*>(sG + b * 32 + 16 + q * 4);
#pragma unroll
        for (int i = 0; i < W3_NX; ++i) {
            const int hv = (i * W3_NLD + tid) >> 2;
            if (hv < W3_HV) {
                const bool ok = (okmask >> i) & 1u;          // the affine applies to voxels inside the volume only: the padding is zero AFTER GroupNorm
                const float f[4] = {ok ? px[i].x * gsc.x + gsh.x : 0.f, ok ? px[i].y * gsc.y + gsh.y : 0.f, ok ? px[i].z * gsc.z + gsh.z : 0.f,
                                    ok ? px[i].w * gsc.w + gsh.w : 0.f};
                split_store(f, sXh + hv * 32 + q * 8, sXl + hv * 32 + q * 8);
            }
        }
#pragma unroll
        for (int i = 0; i < W3_NA; ++i) {
            const int av = (i * W3_NLD + tid) >> 2;
            const float f[4] = {pa[i].x * s_in, pa[i].y * s_in, pa[i].z * s_in, pa[i].w * s_in};
            split_store(f, sAh + av * 32 + q * 8, sAl + av * 32 + q * 8);
            if (a.per_vol) {
#pragma unroll
                for (int j = 0; j < 4; ++j) tsum[j] += f[j];
            }
        }
    };
    const int g16 = lane >> 4, li = lane & 15;
    const int fo = (g16 * 4 + (li >> 2)) * 32 + (li & 3) * 8;      // rows = x 0 .. 15 of one x row; the second read takes the next y row
    auto frag = [&](const char* p0, int second) -> f16x8 {
        const wg_s16x4 u = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_s16x4 __attribute__((address_space(3)))*)(p0));
        const wg_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_s16x4 __attribute__((address_space(3)))*)(p0 + second));
        const f16x4 fu = __builtin_bit_cast(f16x4, u), fv = __builtin_bit_cast(f16x4, v);
        f16x8 r;
        r[0] = fu[0]; r[1] = fu[1]; r[2] = fu[2]; r[3] = fu[3]; r[4] = fv[0]; r[5] = fv[1]; r[6] = fv[2]; r[7] = fv[3];
        return r;
    };

    // this workgroup's bricks: brick0, brick0 + G, .. < nbricks (per-volume mode: the bricks of volume blockIdx.x % B only)
    const int nbv = n0 * n1 * n2;
    const int G = a.per_vol ? (int)gridDim.x / a.B : (int)gridDim.x;
    const int brick0 = a.per_vol ? ((int)blockIdx.x % a.B) * nbv + (int)blockIdx.x / a.B : (int)blockIdx.x;
    const int nbricks = a.per_vol ? ((int)blockIdx.x % a.B + 1) * nbv : a.B * nbv;
    float* const sOut = reinterpret_cast<float*>(smem);
    __syncthreads();                                        // sG
    // The two roles run SEPARATE loops with the same number of barriers (a wave's role never changes, but in one shared loop the register allocator
    // keeps the loaders' staging registers alive through the multiplying code and spills the accumulators).
    if (loader) {
        if (brick0 < nbricks) {
            fetch(brick0);
            store(brick0, smem);
            // unconditional (past the end it re-reads the last brick): a conditional fetch makes the staging registers a merge of old and new
            // values, and the copies the compiler inserts for it wait for the loads at the merge
            fetch(brick0 + G < nbricks ? brick0 + G : brick0);
        }
        __syncthreads();
        int it = 0;
        for (int brick = brick0; brick < nbricks; brick += G, ++it) {
            // brick + G arrived in registers while the multiplying waves worked on the brick before this one: split it into the OTHER buffer (its
            // last readers passed the barrier that ended the previous iteration), then start brick + 2 G on its way
            if (brick + G < nbricks) {
                store(brick + G, smem + ((it + 1) & 1) * W3_LDS);
                fetch(brick + 2 * G < nbricks ? brick + 2 * G : brick + G);
            }
            __syncthreads();
        }
        if (a.per_vol) {
#pragma unroll
            for (int j = 0; j < 4; ++j) atomicAdd(sQ + q * 4 + j, tsum[j]);
        }
        __syncthreads();                                    // (the multiplying waves zero sOut between these two)
        __syncthreads();
    } else {
        f32x4 acc[15];
#pragma unroll
        for (int i = 0; i < 15; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();                                    // buffer 0 is complete
        // The loop exists once per tap half, with `half` a compile-time constant: every tap's fragment address is then the wave's base register plus
        // an immediate (with a run-time `half` the 28 offsets are loop-invariant VGPRs, and at 128 registers they spill).
        auto multiply = [&](auto lo_c, auto hi_c) {
        constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
        int it = 0;
        for (int brick = brick0; brick < nbricks; brick += G, ++it) {
            const char* const sXh = smem + (it & 1) * W3_LDS + (kp * 6 * 18) * 32 + fo;          // slab kp's first halo voxel, this lane's part of a fragment
            const char* const sAh = smem + (it & 1) * W3_LDS + 2 * W3_HV * 32 + (kp * 4 * 16) * 32 + fo;
            auto rows2 = [&](const char* p, const char* p2) -> f16x8 {                           // the x rows at p and p2 -> one 32-voxel fragment
                const wg_s16x4 u = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_s16x4 __attribute__((address_space(3)))*)(p));
                const wg_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_s16x4 __attribute__((address_space(3)))*)(p2));
                const f16x4 fu = __builtin_bit_cast(f16x4, u), fv = __builtin_bit_cast(f16x4, v);
                f16x8 r;
                r[0] = fu[0]; r[1] = fu[1]; r[2] = fu[2]; r[3] = fu[3]; r[4] = fv[0]; r[5] = fv[1]; r[6] = fv[2]; r[7] = fv[3];
                return r;
            };
            f16x8 ah[2], al[2], xh[2][4], xl[4];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) { ah[kk] = rows2(sAh + kk * 512, sAh + kk * 512 + 1024); al[kk] = rows2(sAh + 256 * 32 + kk * 512, sAh + 256 * 32 + kk * 512 + 1024); }
            // Halo rows 2 and 3 are the second row of one fragment and the first of another.  Left to itself the compiler reads them once and assembles the
            // MFMA operands with v_mov - VALU issue this kernel does not have.  The second rows are read through a base the optimiser cannot relate to the first.
            int second = 2 * 18 * 32;
            asm volatile("" : "+v"(second));
            const char* const sXh2 = sXh + second;
#pragma unroll
            for (int j = 0; j < 4; ++j) xh[0][j] = rows2(sXh + w3_frag_off(LO, j), sXh2 + w3_frag_off(LO, j));
#pragma unroll
            for (int g = LO; g < HI; ++g) {
                const int cur = (g - LO) & 1, a0 = 3 * (g - LO);
                // product e = 3 kk + (dy + 1): fragment j = kk + dy + 1, accumulator a0 + dy + 1
#pragma unroll
                for (int j = 0; j < 4; ++j) xl[j] = rows2(sXh + W3_HV * 32 + w3_frag_off(g, j), sXh2 + W3_HV * 32 + w3_frag_off(g, j));
#pragma unroll
                for (int e = 0; e < 6; ++e) acc[a0 + e % 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[e / 3], xh[cur][e / 3 + e % 3], acc[a0 + e % 3], 0, 0, 0);
                if (g + 1 < HI) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) xh[cur ^ 1][j] = rows2(sXh + w3_frag_off(g + 1, j), sXh2 + w3_frag_off(g + 1, j));
                }
#pragma unroll
                for (int e = 0; e < 6; ++e) acc[a0 + e % 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[e / 3], xh[cur][e / 3 + e % 3], acc[a0 + e % 3], 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 6; ++e) acc[a0 + e % 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[e / 3], xl[e / 3 + e % 3], acc[a0 + e % 3], 0, 0, 0);
            }
            __syncthreads();                                // this buffer is free again, the other one is complete
        }
        };
        using I0 = std::integral_constant<int, 0>; using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>; using I9 = std::integral_constant<int, 9>;
        const int glo = half ? 5 - (kp & 1) : 0, ghi = half ? 9 : 5 - (kp & 1);
        if (half == 0) { if (kp & 1) multiply(I0{}, I4{}); else multiply(I0{}, I5{}); }
        else           { if (kp & 1) multiply(I4{}, I9{}); else multiply(I5{}, I9{}); }
        // acc[tt][r] = this wave's share of dW[ca = 4 g16 + r][tap][cx = li]: the four k-step waves of a tap half meet in LDS, ordered [ca][cx][tap]
        const float inv = a.s2 && !a.per_vol ? a.s2[1] : 1.f;
        __syncthreads();
        for (int e = tid; e < 6912; e += 512) sOut[e] = 0.f;
        __syncthreads();
#pragma unroll
        for (int tt = 0; tt < 15; ++tt) {
            const int g = glo + tt / 3, tap = ((g / 3) * 3 + tt % 3) * 3 + g % 3;       // group (dz, dx) = (g / 3, g % 3), dy + 1 = tt % 3
            if (g < ghi) {
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicAdd(&sOut[((4 * g16 + r) * 16 + li) * 27 + tap], acc[tt][r] * inv);
            }
        }
    }
    __syncthreads();
    float* dst = a.part + (((long)blockIdx.x * gridDim.y + blockIdx.y) * gridDim.z + blockIdx.z) * (a.per_vol ? W3_ROW_PV : W3_ROW);
    for (int e = threadIdx.x; e < 6912; e += W3_NTHR) dst[e] = sOut[e];
    if (a.per_vol && threadIdx.x < 16) dst[6912 + threadIdx.x] = sQ[threadIdx.x];
}
// part [nx][Ca / 16][Cx / 16][16 ca][16 cx][27] -> dW += sum over nx
__global__ __launch_bounds__(256) void k_wgrad3_reduce(const float* __restrict__ part, float* __restrict__ dW, int nx, int ny, int nz, int Cx, int tap_minor) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long per = (long)ny * nz * 6912;
    if (i >= per) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int c = 0;
    for (; c + 4 <= nx; c += 4) { s0 += part[c * per + i]; s1 += part[(c + 1) * per + i]; s2 += part[(c + 2) * per + i]; s3 += part[(c + 3) * per + i]; }
    for (; c < nx; ++c) s0 += part[c * per + i];
    const int e = (int)(i % 6912); const int yz = (int)(i / 6912);
    const int ca = (yz / nz) * 16 + e / 432, rem = e % 432, cx = (yz % nz) * 16 + rem / 27, tap = rem % 27;
    const long idx = tap_minor ? ((long)ca * Cx + cx) * 27 + tap : ((long)ca * 27 + tap) * Cx + cx;
    dW[idx] += (s0 + s1) + (s2 + s3);
}

// dZ fp32 [B, D0, D1, D2, Ca], X fp32 [B, D0, D1, D2, Cx] (+ GroupNorm affine [B, Cx]); dW fp32 [Ca, 27, Cx] or (tap_minor) [Ca, Cx, 27], accumulated.
// s2 = (s, 1 / s) device scalars of semabs_grad_scale for dZ, or null.  Needs D0 % 4 == 0, D1 % 4 == 0, D2 % 16 == 0, Ca % 16 == Cx % 16 == 0.
// scratch (scratch_floats fp32): partial sums of the transposing-read kernel k_wgrad3_tr; NULL selects the round-2 kernel (D1 % 8 == 0, atomics).
// Which kernel semabs_wgrad_conv3 would run for a shape: 2 = the transposing-read kernel (needs scratch), 1 = the 4 x 8 x 16 brick kernel, 0 = neither
// (the caller must use semabs_wgrad_mfma / semabs_wgrad).  ONE predicate for the entry point and for host-side routing (semabs_wgrad_conv3_supported).
static int wgrad_conv3_route(int D0, int D1, int D2, int Ca, int Cx, bool have_scratch, long scratch_floats, long* bmax_out) {
    if (D0 <= 0 || D1 <= 0 || D2 <= 0 || Ca <= 0 || Cx <= 0) return 0;
    if (!(D0 % 4 == 0 && D1 % 4 == 0 && D2 % WG_T2 == 0 && Ca % 16 == 0 && Cx % 16 == 0)) return 0;
    const int combos = (Ca / 16) * (Cx / 16);
    const long vpv = (long)D0 * D1 * D2;                    // voxels per volume
    // volumes per launch of the transposing-read kernel: its GroupNorm table holds 64 volumes and its staging offsets are 32-bit
    long bmax = 64;
    if ((1L << 24) / vpv < bmax) bmax = (1L << 24) / vpv;
    if (((1L << 31) - 1) / (vpv * (Ca > Cx ? Ca : Cx) * 4) < bmax) bmax = ((1L << 31) - 1) / (vpv * (Ca > Cx ? Ca : Cx) * 4);
    if (bmax_out) *bmax_out = bmax;
    if (have_scratch && bmax >= 1 && (long)D0 * D1 < (1L << 24) / bmax && scratch_floats >= (long)combos * 6912) return 2;
    return D1 % WG_T1 == 0 ? 1 : 0;
}
extern "C" int semabs_wgrad_conv3_supported(int D0, int D1, int D2, int Ca, int Cx, long scratch_floats, int* kernel) {
    SEMABS_REQUIRE(kernel, "semabs_wgrad_conv3_supported: null pointer");
    *kernel = wgrad_conv3_route(D0, D1, D2, Ca, Cx, scratch_floats > 0, scratch_floats, nullptr);
    return SEMABS_OK;
}
extern "C" int semabs_wgrad_conv3(const float* dZ, const float* X, const float* gn_scale, const float* gn_shift, const float* s2, float* dW, int B,
                                  int D0, int D1, int D2, int Ca, int Cx, int tap_minor, float* scratch, long scratch_floats, void* stream) {
    if (B == 0) return SEMABS_OK;
    SEMABS_REQUIRE(dZ && X && dW, "semabs_wgrad_conv3: null pointer");
    SEMABS_REQUIRE(D0 % 4 == 0 && D1 % 4 == 0 && D2 % WG_T2 == 0 && Ca % 16 == 0 && Cx % 16 == 0,
                   "semabs_wgrad_conv3: needs D0 % 4 == 0, D1 % 4 == 0, D2 % 16 == 0 and channel counts that are multiples of 16");
    SEMABS_REQUIRE((gn_scale == nullptr) == (gn_shift == nullptr), "semabs_wgrad_conv3: gn_scale and gn_shift go together");
    const int combos = (Ca / 16) * (Cx / 16);
    hipStream_t s = (hipStream_t)stream;
    const long vpv = (long)D0 * D1 * D2;                    // voxels per volume
    long bmax = 0;
    const int route = wgrad_conv3_route(D0, D1, D2, Ca, Cx, scratch != nullptr, scratch_floats, &bmax);
    if (route == 2) {
        static SemabsLdsAttr attr3;
        semabs_ensure_lds(&k_wgrad3_tr, 2 * W3_LDS + 64 * 128 + 64, attr3);
        for (int b0 = 0; b0 < B; b0 += (int)bmax) {          // (one launch for every call of the 128^3 training step)
            const int Bc = B - b0 < bmax ? B - b0 : (int)bmax;
            // one persistent workgroup per CU in all, every workgroup leaves one 27 x 16 x 16 partial sum in scratch
            const int nbricks = Bc * (D0 / 4) * (D1 / 4) * (D2 / 16);
            int bx = 256 / combos; if (bx < 1) bx = 1; if (bx > nbricks) bx = nbricks;
            if ((long)bx * combos * 6912 > scratch_floats) bx = (int)(scratch_floats / ((long)combos * 6912));
            Wgrad3Args a;
            a.A = dZ + (long)b0 * vpv * Ca; a.X = X + (long)b0 * vpv * Cx;
            a.gn_scale = gn_scale ? gn_scale + (long)b0 * Cx : nullptr; a.gn_shift = gn_shift ? gn_shift + (long)b0 * Cx : nullptr;
            a.s2 = s2; a.part = scratch; a.B = Bc; a.D0 = D0; a.D1 = D1; a.D2 = D2; a.Ca = Ca; a.Cx = Cx;
            a.gn_mean = nullptr; a.gn_rstd = nullptr; a.G = 1; a.per_vol = 0;
            hipLaunchKernelGGL(k_wgrad3_tr, dim3(bx, Ca / 16, Cx / 16), dim3(W3_NTHR), 2 * W3_LDS + (size_t)Bc * 128 + 64, s, a);
            hipLaunchKernelGGL(k_wgrad3_reduce, dim3(semabs_cdiv((long)combos * 6912, 256)), dim3(256), 0, s, scratch, dW, bx, Ca / 16, Cx / 16, Cx, tap_minor);
        }
        SEMABS_CHECK_LAUNCH();
        return SEMABS_OK;
    }
    SEMABS_REQUIRE(D1 % WG_T1 == 0, "semabs_wgrad_conv3: without (enough) scratch, or for volumes above 2^24 voxels, the 4 x 8 x 16 brick kernel runs and needs D1 % 8 == 0");
    Wgrad16Args a;
    a.A = dZ; a.X = X; a.gn_scale = gn_scale; a.gn_shift = gn_shift; a.s2 = s2; a.dW = dW; a.B = B; a.D0 = D0; a.D1 = D1; a.D2 = D2; a.Ca = Ca; a.Cx = Cx; a.tap_minor = tap_minor;
    const size_t lds = (size_t)(WG_H0 * WG_H1 * 16 * WG_XROW + 8 + WG_T0 * WG_T1 * WG_AYROW) * 2 * 2;
    static SemabsLdsAttr attr16;
    semabs_ensure_lds(&k_wgrad16_lds, (int)lds, attr16);
    const int nbricks = B * (D0 / WG_T0) * (D1 / WG_T1) * (D2 / WG_T2);
    int bx = 768 / combos; if (bx < 8) bx = 8; if (bx > nbricks) bx = nbricks;
    hipLaunchKernelGGL(k_wgrad16_lds, dim3(bx, Ca / 16, Cx / 16), dim3(WG_NTHR), lds, s, a);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// =================================================================================================
// Weight gradient AND the GroupNorm-backward reductions of a 3^3 convolution layer from ONE pass over (dZ, x)  (round 5).
// The layer is y = conv(xn), xn = gamma xhat + beta inside the volume and 0 in the padding, xhat = (x - mean) rstd.  With the per-volume correlations
//     Chat[b, ca, tap, cx] = sum_u dZ[b, u, ca] xhat[b, u + tap, cx]            (u + tap inside the volume; k_wgrad3_tr with xhat staged instead of xn)
//     Z[b, ca, tap]        = sum_u dZ[b, u, ca] [u + tap inside]
// everything the backward pass needs besides the data gradient follows WITHOUT reading the data gradient dXn = conv^T(dZ) or x again:
//     dW[ca, cx, tap]                 = sum_b gamma[cx] Chat[b, ca, tap, cx] + beta[cx] Z[b, ca, tap]
//     Sa[b, c] = sum_v dXn[b, v, c]          = sum_{ca, tap} W[ca, c, tap] Z[b, ca, tap]
//     Sb[b, c] = sum_v dXn[b, v, c] xhat[..] = sum_{ca, tap} W[ca, c, tap] Chat[b, ca, tap, c]
// (Sa, Sb) are what semabs_chan_reduce computed from dXn and x - a pass over two activation-sized tensors per layer (2.1 GB at 8 x 128^3 x 16, 4.0 ms of the
// training step).  Z comes from 27 restricted sums of dZ, Q[sz, sy, sx] with s = 0 (all), 1 (first slab only), 2 (last slab only) per axis: the total
// Q[0, 0, 0] is accumulated by the loader waves of k_wgrad3_tr from the dZ values they stage anyway, the other 26 (faces, edges, corners: 5 % of the
// voxels at 128^3) by k_dz_boundary_sums.  Tap offset d = -1 excludes the first slab of its axis, d = +1 the last, so
//     Z[b, ca, (dz, dy, dx)] = sum over the subsets S of {axes with d != 0} of (-1)^|S| Q[the excluded slab on the axes in S, all on the others].
// Rows of `part`: [bx][Ca / 16][Cx / 16][W3_ROW_PV = 6912 + 16 totals], row x belongs to volume x % B and keeps dZ's dynamic gradient scale.
// (Accumulating the 26 boundary sums in the loader waves too - LDS float atomics from the boundary bricks - cost 2 ms per step: 16-way address conflicts.)
// =================================================================================================
__device__ __forceinline__ float w3_z_from_q(const float* __restrict__ q, int ca_stride, int tap) {     // q -> Q[0][ca], entries ca_stride apart
    const int ex[3] = {tap / 9 == 0 ? 1 : tap / 9 == 2 ? 2 : 0, (tap / 3) % 3 == 0 ? 1 : (tap / 3) % 3 == 2 ? 2 : 0, tap % 3 == 0 ? 1 : tap % 3 == 2 ? 2 : 0};
    float z = 0.f;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        if (((m & 1) && !ex[0]) || ((m & 2) && !ex[1]) || ((m & 4) && !ex[2])) continue;
        const int idx = (((m & 1) ? ex[0] : 0) * 3 + ((m & 2) ? ex[1] : 0)) * 3 + ((m & 4) ? ex[2] : 0);
        const float v = q[(long)idx * ca_stride];
        z += (__popc(m) & 1) ? -v : v;
    }
    return z;
}
// qb fp32 [B][27][Ca] (zeroed by the caller; entry 0 stays 0): sums of s dZ over the voxels whose coordinate on every axis with s != 0 is the first (1) /
// last (2) index.  grid = (26 regions, B, chunks); thread = (voxel of the chunk, 4 channels).
__global__ __launch_bounds__(256) void k_dz_boundary_sums(const float* __restrict__ dZ, const float* __restrict__ s2, float* __restrict__ qb, int D0, int D1, int D2, int Ca) {
    const int idx = (int)blockIdx.x + 1, b = blockIdx.y;
    const int sel[3] = {idx / 9, (idx / 3) % 3, idx % 3}, D[3] = {D0, D1, D2};
    int n[3], lo[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { n[k] = sel[k] ? 1 : D[k]; lo[k] = sel[k] == 2 ? D[k] - 1 : 0; }
    const long nv = (long)n[0] * n[1] * n[2];
    const int cq = Ca / 4, tq = threadIdx.x % cq;
    const int vpb = 256 / cq;                               // voxel slots per pass; Ca / 4 need not divide 256 (Ca = 48, 80, 96, 112: the routes admit any Ca % 16 == 0)
    const bool live = (int)threadIdx.x / cq < vpb;          // ... then the surplus threads would form slot `vpb` = the next pass's slot 0 and count its voxels twice
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long v = (long)blockIdx.z * vpb + threadIdx.x / cq; live && v < nv; v += (long)gridDim.z * vpb) {
        const int x = lo[2] + (int)(v % n[2]), y = lo[1] + (int)((v / n[2]) % n[1]), z = lo[0] + (int)(v / n[2] / n[1]);
        const float4 t = *reinterpret_cast<const float4*>(dZ + ((((long)b * D0 + z) * D1 + y) * D2 + x) * Ca + tq * 4);
        acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
    }
    __shared__ float4 sm[256];
    sm[threadIdx.x] = acc;
    __syncthreads();
    if ((int)threadIdx.x < cq) {
        float4 t = sm[threadIdx.x];
        for (int k = threadIdx.x + cq; k < 256; k += cq) { const float4 u = sm[k]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        const float sc = s2 ? s2[0] : 1.f;
        float* dst = qb + ((long)b * 27 + idx) * Ca + tq * 4;
        atomicAdd(dst, t.x * sc); atomicAdd(dst + 1, t.y * sc); atomicAdd(dst + 2, t.z * sc); atomicAdd(dst + 3, t.w * sc);
    }
}
// rows -> chat [B][ny][nz][6912] (sum over the volume's rows) and qv [B][ny][432] (totals from the cx block 0 rows, the rest from qb)
__global__ __launch_bounds__(256) void k_wgrad3_pv_reduce(const float* __restrict__ part, const float* __restrict__ qb, float* __restrict__ chat,
                                                           float* __restrict__ qv, int bx, int B, int ny, int nz) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long nchat = (long)B * ny * nz * 6912, nq = (long)B * ny * 432;
    if (i >= nchat + nq) return;
    int b, y, z, e;
    if (i < nchat) { e = (int)(i % 6912); long t = i / 6912; z = (int)(t % nz); t /= nz; y = (int)(t % ny); b = (int)(t / ny); }
    else {
        const long k = i - nchat;
        const int idx = (int)(k % 432) / 16, c16 = (int)(k % 16);
        y = (int)((k / 432) % ny); b = (int)(k / 432 / ny);
        if (idx) { qv[k] = qb[((long)b * 27 + idx) * (ny * 16) + y * 16 + c16]; return; }
        e = 6912 + c16; z = 0;
    }
    float s0 = 0.f, s1 = 0.f;
    const long rstride = (long)B * ny * nz * W3_ROW_PV;     // from row x to row x + B (the same volume's next workgroup)
    const float* p0 = part + (((long)b * ny + y) * nz + z) * W3_ROW_PV + e;
    int x = 0;
    for (; x + 2 * B <= bx; x += 2 * B) { s0 += p0[0]; s1 += p0[rstride]; p0 += 2 * rstride; }
    if (x < bx) s0 += p0[0];
    if (i < nchat) chat[i] = s0 + s1; else qv[i - nchat] = s0 + s1;
}
// blocks [0, nw): dW[ca][cx][tap] += inv sum_b (gamma[cx] Chat + beta[cx] Z);  blocks [nw, ..): red[b][c] = (Sa, Sb) in dZ's scale (fp64, like semabs_chan_reduce)
__global__ __launch_bounds__(256) void k_wgrad3_pv_finish(const float* __restrict__ chat, const float* __restrict__ qv, const float* __restrict__ W,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ s2,
                                                           float* __restrict__ dW, double* __restrict__ red, int B, int Ca, int Cx, int nw) {
    const int ny = Ca / 16, nz = Cx / 16;
    if ((int)blockIdx.x < nw) {
        const long i = (long)blockIdx.x * 256 + threadIdx.x;            // = (ca Cx + cx) 27 + tap, the layout of dW
        if (i >= (long)Ca * Cx * 27) return;
        const int tap = (int)(i % 27), cx = (int)((i / 27) % Cx), ca = (int)(i / 27 / Cx);
        const float g = gamma[cx], bt = beta[cx];
        float acc = 0.f;
        for (int b = 0; b < B; ++b) {
            const float ch = chat[((((long)b * ny + ca / 16) * nz + cx / 16) * 6912) + ((ca % 16) * 16 + cx % 16) * 27 + tap];
            const float z = w3_z_from_q(qv + ((long)b * ny + ca / 16) * 432 + ca % 16, 16, tap);
            acc += g * ch + bt * z;
        }
        dW[i] += acc * (s2 ? s2[1] : 1.f);
        return;
    }
    // one block per (volume b, 16-channel slice of c): thread = (c % 16, one of 16 ca classes); Z of the block's volume goes through LDS once
    __shared__ float sz[128 * 27];
    __shared__ double sred[16][16][2];
    const int blk = (int)blockIdx.x - nw, b = blk / nz, cz = blk % nz;
    for (int e = threadIdx.x; e < Ca * 27; e += 256) { const int ca = e / 27; sz[e] = w3_z_from_q(qv + ((long)b * ny + ca / 16) * 432 + ca % 16, 16, e % 27); }
    __syncthreads();
    const int cl = threadIdx.x & 15, part = threadIdx.x >> 4, c = cz * 16 + cl;
    double sa = 0.0, sb = 0.0;
    for (int ca = part; ca < Ca; ca += 16) {
        const float* wrow = W + ((long)ca * Cx + c) * 27;
        const float* crow = chat + ((((long)b * ny + ca / 16) * nz + cz) * 6912) + ((ca % 16) * 16 + cl) * 27;
        float pa = 0.f, pb = 0.f;
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) { pa += wrow[tap] * sz[ca * 27 + tap]; pb += wrow[tap] * crow[tap]; }
        sa += (double)pa; sb += (double)pb;
    }
    sred[part][cl][0] = sa; sred[part][cl][1] = sb;
    __syncthreads();
    if (threadIdx.x < 32) {
        const int k = threadIdx.x & 1, cc = threadIdx.x >> 1;
        double t = 0.0;
        for (int p2 = 0; p2 < 16; ++p2) t += sred[p2][cc][k];
        red[((long)b * Cx + cz * 16 + cc) * 2 + k] = t;
    }
}
// bx (workgroups along x, a multiple of B) for the per-volume launch, or 0 if the shape / scratch does not allow it
static int wgrad_conv3_gn_bx(int B, int D0, int D1, int D2, int Ca, int Cx, long scratch_floats) {
    long bmax = 0;
    if (B <= 0 || Ca > 128 || wgrad_conv3_route(D0, D1, D2, Ca, Cx, true, scratch_floats, &bmax) != 2 || B > bmax || B > 64) return 0;
    const int combos = (Ca / 16) * (Cx / 16);
    const int nbv = (D0 / 4) * (D1 / 4) * (D2 / 16);
    int per = 256 / combos / B;                             // workgroups per volume
    if (per > nbv) per = nbv;
    const long fixed = (long)B * combos * 6912 + (long)B * (Ca / 16) * 432 + (long)B * 27 * Ca;      // chat + qv + qb behind the rows
    while (per >= 1 && (long)per * B * combos * W3_ROW_PV + fixed > scratch_floats) --per;
    return per >= 1 ? per * B : 0;
}
extern "C" int semabs_wgrad_conv3_gn_supported(int B, int D0, int D1, int D2, int Ca, int Cx, long scratch_floats, int* ok) {
    SEMABS_REQUIRE(ok, "semabs_wgrad_conv3_gn_supported: null pointer");
    *ok = wgrad_conv3_gn_bx(B, D0, D1, D2, Ca, Cx, scratch_floats) > 0 ? 1 : 0;
    return SEMABS_OK;
}
// dZ fp32 [B, D0, D1, D2, Ca] (scaled by s2[0] on the way in, s2 = semabs_grad_scale's (s, 1 / s) or null), X fp32 [B, D0, D1, D2, Cx] = the layer's
// GroupNorm INPUT with its statistics mean / rstd [B, G] and affine gamma / beta [Cx]; W = the layer's weights fp32 [Ca, Cx, 27].
// dW fp32 [Ca, Cx, 27] += the weight gradient; red fp64 [B, Cx, 2] = (sum dXn, sum dXn xhat) in dZ's scale - the output of semabs_chan_reduce(dXn, X, ..).
extern "C" int semabs_wgrad_conv3_gn(const float* dZ, const float* X, const float* mean, const float* rstd, int G, const float* gamma, const float* beta,
                                     const float* W, const float* s2, float* dW, double* red, int B, int D0, int D1, int D2, int Ca, int Cx,
                                     float* scratch, long scratch_floats, void* stream) {
    if (B == 0) return SEMABS_OK;
    SEMABS_REQUIRE(dZ && X && mean && rstd && gamma && beta && W && dW && red && scratch, "semabs_wgrad_conv3_gn: null pointer");
    SEMABS_REQUIRE(G > 0 && Cx % G == 0, "semabs_wgrad_conv3_gn: the group count must divide the input channels");
    SEMABS_REQUIRE(Ca <= 128, "semabs_wgrad_conv3_gn: at most 128 output channels");
    const int bx = wgrad_conv3_gn_bx(B, D0, D1, D2, Ca, Cx, scratch_floats);
    SEMABS_REQUIRE(bx > 0, "semabs_wgrad_conv3_gn: shape / scratch not supported (ask semabs_wgrad_conv3_gn_supported first)");
    hipStream_t s = (hipStream_t)stream;
    const int ny = Ca / 16, nz = Cx / 16, combos = ny * nz;
    static SemabsLdsAttr attr3;
    semabs_ensure_lds(&k_wgrad3_tr, 2 * W3_LDS + 64 * 128 + 64, attr3);
    Wgrad3Args a;
    a.A = dZ; a.X = X; a.gn_scale = nullptr; a.gn_shift = nullptr; a.s2 = s2; a.part = scratch; a.B = B; a.D0 = D0; a.D1 = D1; a.D2 = D2; a.Ca = Ca; a.Cx = Cx;
    a.gn_mean = mean; a.gn_rstd = rstd; a.G = G; a.per_vol = 1;
    hipLaunchKernelGGL(k_wgrad3_tr, dim3(bx, ny, nz), dim3(W3_NTHR), 2 * W3_LDS + (size_t)B * 128 + 64, s, a);
    float* chat = scratch + (long)bx * combos * W3_ROW_PV;
    float* qv = chat + (long)B * combos * 6912;
    float* qb = qv + (long)B * ny * 432;                    // [B][27][Ca]
    (void)hipMemsetAsync(qb, 0, (size_t)B * 27 * Ca * 4, s);
    {
        const long face = (long)(D0 > D1 ? D0 : D1) * (D1 > D2 ? D1 : D2);       // voxels of the largest face
        const int vpb = 256 / (Ca / 4);
        int chunks = (int)semabs_cdiv(face, (long)vpb * 8); if (chunks > 64) chunks = 64; if (chunks < 1) chunks = 1;
        hipLaunchKernelGGL(k_dz_boundary_sums, dim3(26, B, chunks), dim3(256), 0, s, dZ, s2, qb, D0, D1, D2, Ca);
    }
    hipLaunchKernelGGL(k_wgrad3_pv_reduce, dim3(semabs_cdiv((long)B * combos * 6912 + (long)B * ny * 432, 256)), dim3(256), 0, s, scratch, qb, chat, qv, bx, B, ny, nz);
    const int nw = (int)semabs_cdiv((long)Ca * Cx * 27, 256);
    hipLaunchKernelGGL(k_wgrad3_pv_finish, dim3(nw + B * nz), dim3(256), 0, s, chat, qv, W, gamma, beta, s2, dW, red, B, Ca, Cx, nw);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}
