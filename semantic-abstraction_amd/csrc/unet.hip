// OVSSC voxel-inference kernels: point-feature MLP, deterministic scatter-mean, GroupNorm statistics,
// implicit-GEMM Conv3d / ConvTranspose3d / 1x1x1 conv on MFMA with GroupNorm-apply fused into the operand load and
// ReLU / residual / skip-add / bias fused into the store, MaxPool3d, trilinear decoder + MLP.
//
// Replaces (reference file:line):
//   SemAbs3D.pts_feat_extractor                   net.py:358-367, 395-404
//   VirtualGrid.scatter_points (reduce = MEAN)    net.py:185-201  (torch_scatter.scatter)
//   create_conv "gcr" / ExtResNetBlock            unet3d.py:20-95, 190-259   (GroupNorm -> Conv3d 3^3 no bias -> ReLU)
//   Encoder MaxPool3d / Decoder ConvTranspose3d   unet3d.py:298-317, 428-444, 385-396 (sum joining)
//   final_conv                                    unet3d.py:579
//   ImplicitVolumetricDecoder                     net.py:215-256   (divide by S, x -> innermost axis quirks kept)
//
// Layout: activations are channels-last [B, D0, D1, D2, C] (fp16 in HBM, or fp32 in "exact" mode) so that the 8
// input channels one MFMA lane needs are one 16-byte load and the 16 channels of a voxel are one 32-byte line.
// MFMA: v_mfma_f32_16x16x32_f16 with the WEIGHTS as the A operand (rows = output channels) and 16 voxels as the
// B operand columns, so each lane ends up with 4 consecutive output channels of one voxel (8-byte coalesced store).
// K order is (tap, cin); for Cin = 16 one MFMA k-step covers two taps, otherwise one tap x 32 input channels.
// "exact" mode splits both operands into fp16 hi + lo and issues 3 MFMAs (hi*hi + lo*hi + hi*lo): ~fp32 accuracy.
//
// Convolution kernels by level of the 128^3 UNet (all share the ConvArgs contract and the [Cout, (tap, cin)] split weight matrices):
//   k_conv16_lds    128^3 x 16 -> 16      persistent, wave-specialised (producer / consumer), LDS-double-buffered halo bricks
//   k_conv_brick    64^3 .. 16^3           LDS halo bricks, input channels staged 32 at a time, Cout sliced over grid z
//   k_convT_brick   ConvTranspose3d of the three upper levels, two launches (output parity along axis 0)
//   k_conv          everything else: gather implicit GEMM (1x1x1, strided data gradients, deepest levels; split-K + k_conv_finish
//                   for the 8^3 / 4^3 levels in exact mode)
#include "semabs_common.h"

// The hardware places workgroup b of a launch on XCD b % 8, and each XCD has its own 4 MiB L2.  Tiles that share halo voxels (neighbouring
// bricks, the output-channel slices of one brick) should therefore NOT have consecutive workgroup ids.  Bijective remap: XCD x runs the
// contiguous run of tile ids [x * n / 8, (x + 1) * n / 8) in dispatch order, so a tile's neighbours are resident on - or were just run by -
// the same XCD and its halo re-reads hit that L2 instead of going out to the fabric (round 3; the GEMMs do the same, gemm.hip tile_of).
__device__ __forceinline__ int xcd_remap(int vb, int nb) {
    const int q = nb >> 3, r = nb & 7, x = vb & 7, k = vb >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}
// the same for persistent workgroups (grid G, G % 8 == 0): iteration `it` of workgroup vb -> tile id, or -1 when this XCD's run is exhausted
__device__ __forceinline__ int xcd_persistent_tile(int vb, int G, int it, int total) {
    if (G & 7) { const long t = (long)it * G + vb; return t < total ? (int)t : -1; }
    const int x = vb & 7, j = vb >> 3, g8 = G >> 3;
    const int q = total >> 3, r = total & 7;
    const int lo = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q, cnt = q + (x < r ? 1 : 0);
    const long o = (long)it * g8 + j;
    return o < cnt ? lo + (int)o : -1;
}


// =================================================================================================
// Point MLP: (xyz | feat) 4 -> H -> H -> C, LeakyReLU(0.01); fp32 FMAs, weights in LDS, one thread per (label, point)
// =================================================================================================
template <int HID, int COUT>
__global__ __launch_bounds__(256, 2) void k_point_mlp(const float* __restrict__ xyz, const float* __restrict__ feat,
                                                   const float* __restrict__ w1, const float* __restrict__ b1,
                                                   const float* __restrict__ w2, const float* __restrict__ b2,
                                                   const float* __restrict__ w3, const float* __restrict__ b3,
                                                   float* __restrict__ out, int P, long N) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sw2 = reinterpret_cast<float*>(smem);          // [HID][HID]
    float* sw3 = sw2 + HID * HID;                         // [COUT][HID]
    float* sw1 = sw3 + COUT * HID;                        // [HID][4]
    float* sb = sw1 + HID * 4;                            // b1[HID] b2[HID] b3[COUT]
    for (int i = threadIdx.x; i < HID * HID; i += 256) sw2[i] = w2[i];
    for (int i = threadIdx.x; i < COUT * HID; i += 256) sw3[i] = w3[i];
    for (int i = threadIdx.x; i < HID * 4; i += 256) sw1[i] = w1[i];
    for (int i = threadIdx.x; i < HID; i += 256) { sb[i] = b1[i]; sb[HID + i] = b2[i]; }
    for (int i = threadIdx.x; i < COUT; i += 256) sb[2 * HID + i] = b3[i];
    __syncthreads();
    const long total = (long)P * N;
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < total; r += (long)gridDim.x * 256) {
        const long n = r % N;
        const float x = xyz[n * 3], y = xyz[n * 3 + 1], z = xyz[n * 3 + 2], f = feat[r];
        float h1[HID];
#pragma unroll
        for (int jb = 0; jb < HID; jb += 8) {
#pragma unroll
            for (int j = jb; j < jb + 8; ++j) {
                const float4 w = *reinterpret_cast<const float4*>(sw1 + j * 4);
                float v = sb[j] + w.x * x + w.y * y + w.z * z + w.w * f;
                h1[j] = v > 0.f ? v : 0.01f * v;
            }
            __builtin_amdgcn_sched_barrier(0);     // stop the scheduler hoisting all 128 LDS reads (register blow-up)
        }
        float o[COUT];
#pragma unroll
        for (int c = 0; c < COUT; ++c) o[c] = sb[2 * HID + c];
#pragma unroll 1
        for (int j = 0; j < HID; ++j) {
            float v = sb[HID + j];
            const float4* wr = reinterpret_cast<const float4*>(sw2 + j * HID);
#pragma unroll
            for (int kb = 0; kb < HID / 4; kb += 8) {
#pragma unroll
                for (int k = kb; k < kb + 8; ++k) {
                    const float4 w = wr[k];
                    v += w.x * h1[4 * k] + w.y * h1[4 * k + 1] + w.z * h1[4 * k + 2] + w.w * h1[4 * k + 3];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            v = v > 0.f ? v : 0.01f * v;
#pragma unroll
            for (int c = 0; c < COUT; ++c) o[c] += sw3[c * HID + j] * v;
        }
        float4* dst = reinterpret_cast<float4*>(out + r * COUT);
#pragma unroll
        for (int c = 0; c < COUT / 4; ++c) dst[c] = make_float4(o[4 * c], o[4 * c + 1], o[4 * c + 2], o[4 * c + 3]);
    }
}

// -------------------------------------------------------------------------------------------------
// The same MLP on the matrix pipe (exact mode: operands split into fp16 hi + lo, 3 MFMAs per product, fp32 accumulate).  A wave owns 64
// points (4 tiles of 16 = MFMA columns) and never leaves its registers between the layers:
//   layer 1 (4 -> 128) is computed on the VALU directly in the B-operand layout of layer 2 (lane (point, kg) produces the 8 channels
//   ks * 32 + kg * 8 .. + 7 of k-step ks);
//   layer 2 (128 -> 128): W2 hi / lo are the A operand from LDS (rows 288 B apart: conflict-free ds_read_b128 for the real lane groups),
//   32 accumulators [8 output tiles][4 point tiles];
//   layer 3 (128 -> 16) takes its B operand straight from those accumulators: a lane holds channels 16 ot + 4 kg + r, so k-step ks' is
//   DEFINED as the channels {16 (2 ks') + 4 kg + e, 16 (2 ks' + 1) + 4 kg + e : e < 4} and W3 is staged in LDS in that k order.
// 48.5 GFLOP per scene: 1.45 ms on fp32 FMAs -> matrix pipe.
// -------------------------------------------------------------------------------------------------
#define PM_ROW 144                              // fp16 elements per W2 row in LDS (128 + 16 pad = 288 B)
__global__ __launch_bounds__(512) void k_point_mlp_mfma(const float* __restrict__ xyz, const float* __restrict__ feat,
                                                        const float* __restrict__ w1, const float* __restrict__ b1,
                                                        const float* __restrict__ w2, const float* __restrict__ b2,
                                                        const float* __restrict__ w3, const float* __restrict__ b3,
                                                        float* __restrict__ out, int P, long N) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f16* sw2h = reinterpret_cast<f16*>(smem);              // [128][PM_ROW]
    f16* sw2l = sw2h + 128 * PM_ROW;
    f16* sw3h = sw2l + 128 * PM_ROW;                        // [4 ks'][4 kg][16 m][8]: the A fragment of layer 3, ready to load
    f16* sw3l = sw3h + 4 * 4 * 16 * 8;
    float* sw1 = reinterpret_cast<float*>(sw3l + 4 * 4 * 16 * 8);   // [128][4]
    float* sb1 = sw1 + 128 * 4;                             // [128]
    float* sb2 = sb1 + 128;                                 // [128]
    const int tid = threadIdx.x;
    for (int i = tid; i < 128 * 128; i += 512) {
        const int r = i >> 7, c = i & 127;
        const float v = w2[i];
        const f16 h = (f16)v;
        sw2h[r * PM_ROW + c] = h; sw2l[r * PM_ROW + c] = (f16)(v - (float)h);
    }
    for (int i = tid; i < 4 * 4 * 16 * 8; i += 512) {
        const int e = i & 7, m = (i >> 3) & 15, kg = (i >> 7) & 3, ks = i >> 9;
        const int ch = (e < 4) ? (2 * ks) * 16 + 4 * kg + e : (2 * ks + 1) * 16 + 4 * kg + (e - 4);
        const float v = w3[m * 128 + ch];
        const f16 h = (f16)v;
        sw3h[i] = h; sw3l[i] = (f16)(v - (float)h);
    }
    for (int i = tid; i < 128 * 4; i += 512) sw1[i] = w1[i];
    for (int i = tid; i < 128; i += 512) { sb1[i] = b1[i]; sb2[i] = b2[i]; }
    __syncthreads();
    const int lane = tid & 63, l15 = lane & 15, kg = lane >> 4;
    const int wave = (int)blockIdx.x * 8 + (tid >> 6), n_waves = (int)gridDim.x * 8;
    const long total = (long)P * N;
    const f32x4 b3v = *reinterpret_cast<const f32x4*>(b3 + 4 * kg);
    for (long g0 = (long)wave * 64; g0 < total; g0 += (long)n_waves * 64) {
        float px[4], py[4], pz[4], pf[4];
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            long r = g0 + pt * 16 + l15; if (r > total - 1) r = total - 1;
            const long n = r % N;
            px[pt] = xyz[n * 3]; py[pt] = xyz[n * 3 + 1]; pz[pt] = xyz[n * 3 + 2]; pf[pt] = feat[r];
        }
        f32x4 acc[8][4];
#pragma unroll
        for (int ot = 0; ot < 8; ++ot)
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) acc[ot][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int ks = 0; ks < 4; ++ks) {
            // layer 1 for the 8 channels this lane feeds into k-step ks, for its 4 points
            f16x8 bh[4], bl[4];
            const int c0 = ks * 32 + kg * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float4 w = *reinterpret_cast<const float4*>(sw1 + (c0 + e) * 4);
                const float bb = sb1[c0 + e];
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) {
                    float v = bb + w.x * px[pt] + w.y * py[pt] + w.z * pz[pt] + w.w * pf[pt];
                    v = v > 0.f ? v : 0.01f * v;
                    const f16 h = (f16)v;
                    bh[pt][e] = h; bl[pt][e] = (f16)(v - (float)h);
                }
            }
#pragma unroll
            for (int ot = 0; ot < 8; ++ot) {
                const int aoff = (ot * 16 + l15) * PM_ROW + ks * 32 + kg * 8;
                const f16x8 ah = *reinterpret_cast<const f16x8*>(sw2h + aoff);
                const f16x8 al = *reinterpret_cast<const f16x8*>(sw2l + aoff);
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) acc[ot][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[pt], acc[ot][pt], 0, 0, 0);
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) acc[ot][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[pt], acc[ot][pt], 0, 0, 0);
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) acc[ot][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[pt], acc[ot][pt], 0, 0, 0);
            }
        }
        // layer 2 epilogue (bias, LeakyReLU) and layer 3, k-step ks' = output tiles (2 ks', 2 ks' + 1)
        f32x4 o[4];
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) o[pt] = b3v;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const f32x4 ba = *reinterpret_cast<const f32x4*>(sb2 + (2 * ks) * 16 + 4 * kg);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(sb2 + (2 * ks + 1) * 16 + 4 * kg);
            const f16x8 wh = *reinterpret_cast<const f16x8*>(sw3h + ((ks * 4 + kg) * 16 + l15) * 8);
            const f16x8 wl = *reinterpret_cast<const f16x8*>(sw3l + ((ks * 4 + kg) * 16 + l15) * 8);
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) {
                f16x8 hh, hl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[2 * ks][pt][e] + ba[e];
                    v = v > 0.f ? v : 0.01f * v;
                    f16 h = (f16)v; hh[e] = h; hl[e] = (f16)(v - (float)h);
                    v = acc[2 * ks + 1][pt][e] + bb[e];
                    v = v > 0.f ? v : 0.01f * v;
                    h = (f16)v; hh[4 + e] = h; hl[4 + e] = (f16)(v - (float)h);
                }
                o[pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, hh, o[pt], 0, 0, 0);
                o[pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, hh, o[pt], 0, 0, 0);
                o[pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, hl, o[pt], 0, 0, 0);
            }
        }
        // o[pt][r] = out[point g0 + 16 pt + l15][channel 4 kg + r]
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            const long r = g0 + pt * 16 + l15;
            if (r < total) *reinterpret_cast<f32x4*>(out + r * 16 + 4 * kg) = o[pt];
        }
    }
}

// xyz fp32 [N, 3] (shared by the P label volumes), feat fp32 [P, N] -> out fp32 [P, N, 16]
extern "C" int semabs_point_mlp(const float* xyz, const float* feat, const float* w1, const float* b1, const float* w2,
                                const float* b2, const float* w3, const float* b3, float* out, int P, long N, int hidden,
                                int cout, void* stream) {
    if (P == 0 || N == 0) return SEMABS_OK;
    SEMABS_REQUIRE(xyz && feat && w1 && b1 && w2 && b2 && w3 && b3 && out, "semabs_point_mlp: null pointer");
    SEMABS_REQUIRE(hidden == 128 && cout == 16, "semabs_point_mlp: built for hidden 128 -> 16 channels (net.py:358-367 defaults)");
    const size_t lds2 = (size_t)(2 * 128 * PM_ROW + 2 * 4 * 4 * 16 * 8) * 2 + (size_t)(128 * 4 + 2 * 128) * 4;
    static SemabsLdsAttr attr2;
    semabs_ensure_lds(&k_point_mlp_mfma, (int)lds2, attr2);
    const long groups = ((long)P * N + 63) / 64;
    long grid2 = (groups + 7) / 8; if (grid2 > semabs_stream_cus((hipStream_t)stream)) grid2 = semabs_stream_cus((hipStream_t)stream);
    hipLaunchKernelGGL(k_point_mlp_mfma, dim3((unsigned)grid2), dim3(512), lds2, (hipStream_t)stream, xyz, feat, w1, b1, w2, b2, w3, b3, out, P, N);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// The same MLP on fp32 FMAs (no matrix pipe): the cross-check of the MFMA kernel in tests/ (its own entry point - no process-global switch).
extern "C" int semabs_point_mlp_fma(const float* xyz, const float* feat, const float* w1, const float* b1, const float* w2,
                                    const float* b2, const float* w3, const float* b3, float* out, int P, long N, int hidden,
                                    int cout, void* stream) {
    if (P == 0 || N == 0) return SEMABS_OK;
    SEMABS_REQUIRE(xyz && feat && w1 && b1 && w2 && b2 && w3 && b3 && out, "semabs_point_mlp_fma: null pointer");
    SEMABS_REQUIRE(hidden == 128 && cout == 16, "semabs_point_mlp_fma: built for hidden 128 -> 16 channels (net.py:358-367 defaults)");
    size_t lds = (size_t)(128 * 128 + 16 * 128 + 128 * 4 + 2 * 128 + 16) * 4;
    static SemabsLdsAttr attr;
    semabs_ensure_lds(&k_point_mlp<128, 16>, (int)lds, attr);
    long total = (long)P * N;
    int grid = semabs_cdiv(total, 256); if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL((k_point_mlp<128, 16>), dim3(grid), dim3(256), lds, (hipStream_t)stream, xyz, feat, w1, b1, w2, b2, w3, b3, out, P, N);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// =================================================================================================
// Scatter-mean, deterministic: per-voxel linked lists (atomicExch on a head table), then the list head sorts its
// (short) list by point index and sums in that order = the order torch_scatter's CPU scatter_add uses.
// The voxel index is shared by all P label volumes of a scene.
// =================================================================================================
__global__ void k_scatter_link(const long long* __restrict__ flat, long N, int* __restrict__ head, int* __restrict__ next) {
    long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    next[p] = atomicExch(&head[flat[p]], (int)p);
}

#define SCATTER_MAXLIST 32
template <typename TOut>
__global__ void k_scatter_mean(const long long* __restrict__ flat, const float* __restrict__ feat, const int* __restrict__ head,
                               const int* __restrict__ next, TOut* __restrict__ vol, int P, long N, int C, long nvox, double* __restrict__ stats,
                               unsigned int* __restrict__ occ) {
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = t < (long)P * N;
    if (!in_range) t = (long)P * N - 1;
    const long p = t % N; const int b = (int)(t / N);
    const long v = flat[p];
    const bool worker = in_range && head[v] == (int)p;     // one worker per occupied voxel (and label)
    if (occ && worker && b == 0) atomicOr(occ + (v >> 5), 1u << (v & 31));     // occupancy bitmap (zeroed by the caller): the volume's other voxels stay unwritten
    // `stats` (optional, C == 16): GroupNorm statistics (8 groups of 2 channels) of the volume being written - empty voxels add nothing, so
    // the sums over the workers' voxels are the sums over the dense volume.  A wave reduces its workers' sums and issues 16 fp64 atomics.
    float gs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (worker) {
    int ids[SCATTER_MAXLIST]; int cnt = 0; int total = 0;
    for (int q = (int)p; q >= 0; q = next[q]) {
        if (cnt < SCATTER_MAXLIST) ids[cnt++] = q;
        ++total;
    }
    if (total <= SCATTER_MAXLIST) {                 // insertion sort ascending: reference accumulation order
        for (int i = 1; i < cnt; ++i) {
            int k = ids[i], j = i - 1;
            while (j >= 0 && ids[j] > k) { ids[j + 1] = ids[j]; --j; }
            ids[j + 1] = k;
        }
    }
    const float denom = (float)total;
    if (C == 16 && total <= SCATTER_MAXLIST) {      // the released configuration: whole 64-byte point rows, same per-channel summation order
        float sv[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) sv[c] = 0.f;
        for (int i = 0; i < cnt; ++i) {
            const float4* fp = reinterpret_cast<const float4*>(feat + ((long)b * N + ids[i]) * 16);
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float4 t = fp[k]; sv[4 * k] += t.x; sv[4 * k + 1] += t.y; sv[4 * k + 2] += t.z; sv[4 * k + 3] += t.w; }
        }
        TOut ov[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            ov[c] = (TOut)(sv[c] / denom);
            if (stats) { const float of = (float)ov[c]; gs[c >> 1] += of; gq[c >> 1] += of * of; }
        }
        TOut* vp = vol + ((long)b * nvox + v) * 16;
#pragma unroll
        for (int c = 0; c < 16; ++c) vp[c] = ov[c];
    } else
    for (int c = 0; c < C; ++c) {
        float s = 0.f;
        if (total <= SCATTER_MAXLIST) {
            for (int i = 0; i < cnt; ++i) s += feat[((long)b * N + ids[i]) * C + c];
        } else {                                    // very crowded voxel: list order (last-bit differences only)
            for (int q = (int)p; q >= 0; q = next[q]) s += feat[((long)b * N + q) * C + c];
        }
        const TOut o = (TOut)(s / denom);
        vol[((long)b * nvox + v) * C + c] = o;
        if (stats && c < 16) { const float of = (float)o; gs[c >> 1] += of; gq[c >> 1] += of * of; }
    }
    }
    if (stats) {                                            // uniform
        if (__ballot(worker) != 0) {
            const int bw = __builtin_amdgcn_readfirstlane(b);       // a wave may straddle two labels only if N % 64 != 0: then per-lane atomics
            const bool uniform_b = __ballot(b != bw) == 0;
            if (uniform_b) {
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    float a = gs[g], q = gq[g];
#pragma unroll
                    for (int m = 1; m < 64; m <<= 1) { a += __shfl_xor(a, m, 64); q += __shfl_xor(q, m, 64); }
                    if ((threadIdx.x & 63) == 0) { atomicAdd(stats + ((long)bw * 8 + g) * 2, (double)a); atomicAdd(stats + ((long)bw * 8 + g) * 2 + 1, (double)q); }
                }
            } else if (worker) {
#pragma unroll
                for (int g = 0; g < 8; ++g) { atomicAdd(stats + ((long)b * 8 + g) * 2, (double)gs[g]); atomicAdd(stats + ((long)b * 8 + g) * 2 + 1, (double)gq[g]); }
            }
        }
    }
}

// flat int64 [N]; feat fp32 [P, N, C]; vol [P, nvox, C] (fp16 or fp32), must be zero-filled by the caller;
// head int32 [nvox] filled with -1 by the caller; next int32 [N] scratch.
static int scatter_mean_impl(const long long* flat, const float* feat, int* head, int* next, void* vol, int P, long N, int C,
                             long nvox, int vol_f32, double* stats, void* stream, unsigned int* occ = nullptr) {
    if (P == 0 || N == 0) return SEMABS_OK;
    SEMABS_REQUIRE(flat && feat && head && next && vol && C > 0 && nvox > 0, "semabs_scatter_mean: bad args");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_scatter_link, dim3(semabs_cdiv(N, 256)), dim3(256), 0, s, flat, N, head, next);
    if (vol_f32) hipLaunchKernelGGL(k_scatter_mean<float>, dim3(semabs_cdiv((long)P * N, 256)), dim3(256), 0, s, flat, feat, head, next, (float*)vol, P, N, C, nvox, stats, occ);
    else hipLaunchKernelGGL(k_scatter_mean<f16>, dim3(semabs_cdiv((long)P * N, 256)), dim3(256), 0, s, flat, feat, head, next, (f16*)vol, P, N, C, nvox, stats, occ);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}
extern "C" int semabs_scatter_mean(const long long* flat, const float* feat, int* head, int* next, void* vol, int P, long N, int C,
                                   long nvox, int vol_f32, void* stream) {
    return scatter_mean_impl(flat, feat, head, next, vol, P, N, C, nvox, vol_f32, nullptr, stream);
}
// + the GroupNorm statistics (8 groups; sum / sum of squares, fp64 [P, 8, 2], zero-filled by the caller) of the scattered volume for the first
// UNet block: only occupied voxels contribute, so the 2 GB dense volume is not read back for them.  C must be 16.
extern "C" int semabs_scatter_mean_stats(const long long* flat, const float* feat, int* head, int* next, void* vol, int P, long N, int C,
                                         long nvox, int vol_f32, double* out_sums, void* stream) {
    SEMABS_REQUIRE(out_sums && C == 16, "semabs_scatter_mean_stats: needs out_sums and C == 16");
    return scatter_mean_impl(flat, feat, head, next, vol, P, N, C, nvox, vol_f32, out_sums, stream);
}
// ... and SPARSE: vol is NOT zero-filled by the caller; occ (uint32 [ceil(nvox / 32)], zero-filled by the caller) receives the occupancy bitmap, one for all P
// volumes (they scatter the same points); only occupied voxels of vol are written.  The consumer is semabs_conv3d_sparse_stats.
extern "C" int semabs_scatter_mean_sparse(const long long* flat, const float* feat, int* head, int* next, void* vol, unsigned int* occ, int P, long N, int C,
                                          long nvox, int vol_f32, double* out_sums, void* stream) {
    SEMABS_REQUIRE(out_sums && occ && C == 16, "semabs_scatter_mean_sparse: needs out_sums, occ and C == 16");
    return scatter_mean_impl(flat, feat, head, next, vol, P, N, C, nvox, vol_f32, out_sums, stream, occ);
}

// =================================================================================================
// GroupNorm statistics: per (volume, group) sum and sum of squares -> fp64 atomics; then per (volume, channel)
// scale = rstd * gamma, shift = beta - mean * scale.
// =================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void k_gn_stats(const T* __restrict__ x, double* __restrict__ sums, long nvox, int C, int G) {
    // grid = (blocks_per_volume, B); thread handles 8 consecutive channels of a voxel
    const int b = blockIdx.y;
    const int cpv = C / 8;                                  // 8-channel chunks per voxel
    const long chunks = nvox * cpv;
    const int cg = C / G;                                   // channels per group (>= 2)
    const T* xb = x + (long)b * nvox * C;
    // a thread always sees the same channel chunk when the stride is a multiple of cpv
    long stride = (long)gridDim.x * 256;
    stride -= stride % cpv;
    const long start = (long)blockIdx.x * 256 + threadIdx.x;
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
    if (start < stride)
        for (long i = start; i < chunks; i += stride) {
            float v[8];
            if (sizeof(T) == 2) {
                f16x8 h = *reinterpret_cast<const f16x8*>(reinterpret_cast<const f16*>(xb) + i * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (float)h[j];
            } else {
                const float4* p4 = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(xb) + i * 8);
                float4 a = p4[0], c = p4[1];
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) { s[j] += v[j]; q[j] += v[j] * v[j]; }
        }
    // block reduction through LDS float atomics, then one fp64 global atomic per (group, moment) per block
    __shared__ float sh[2][512];
    for (int c = threadIdx.x; c < C; c += 256) { sh[0][c] = 0.f; sh[1][c] = 0.f; }
    __syncthreads();
    if (start < stride) {
        const int c0 = (int)(start % cpv) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) { atomicAdd(&sh[0][c0 + j], s[j]); atomicAdd(&sh[1][c0 + j], q[j]); }
    }
    __syncthreads();
    for (int g = threadIdx.x; g < G; g += 256) {
        float ss = 0.f, qq = 0.f;
        for (int k = 0; k < cg; ++k) { ss += sh[0][g * cg + k]; qq += sh[1][g * cg + k]; }
        atomicAdd(&sums[((long)b * G + g) * 2], (double)ss);
        atomicAdd(&sums[((long)b * G + g) * 2 + 1], (double)qq);
    }
}

// x [B, nvox, C] (fp16 or fp32) -> sums fp64 [B, G, 2] (zero-filled by the caller)
extern "C" int semabs_gn_stats(const void* x, double* sums, int B, long nvox, int C, int G, int x_f32, void* stream) {
    if (B == 0) return SEMABS_OK;
    SEMABS_REQUIRE(x && sums && nvox > 0 && C % 8 == 0 && C <= 512 && G > 0 && C % G == 0, "semabs_gn_stats: bad args (C % 8 == 0, C <= 512)");
    long chunks = nvox * (C / 8);
    int bx = semabs_cdiv(chunks, 256 * 16); if (bx < 1) bx = 1; if (bx > 256) bx = 256;
    // the per-thread channel chunk must be loop-invariant: block count * 256 rounded down to a multiple of C/8 inside
    if (x_f32) hipLaunchKernelGGL(k_gn_stats<float>, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, (const float*)x, sums, nvox, C, G);
    else hipLaunchKernelGGL(k_gn_stats<f16>, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, (const f16*)x, sums, nvox, C, G);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

__global__ void k_gn_finalize(const double* __restrict__ sums, const float* __restrict__ gamma, const float* __restrict__ beta,
                              float* __restrict__ scale, float* __restrict__ shift, int B, int C, int G, double count, float eps) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i % C, g = c / (C / G);
    const double mean = sums[((long)b * G + g) * 2] / count;
    double var = sums[((long)b * G + g) * 2 + 1] / count - mean * mean;
    if (var < 0) var = 0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = rstd * gamma[c];
    scale[i] = sc;
    shift[i] = beta[c] - (float)mean * sc;
}
// sums fp64 [B, G, 2] -> scale/shift fp32 [B, C]; count = voxels * channels_per_group
extern "C" int semabs_gn_finalize(const double* sums, const float* gamma, const float* beta, float* scale, float* shift, int B, int C,
                                  int G, long nvox, float eps, void* stream) {
    if (B == 0) return SEMABS_OK;
    SEMABS_REQUIRE(sums && gamma && beta && scale && shift && C % G == 0 && nvox > 0, "semabs_gn_finalize: bad args");
    hipLaunchKernelGGL(k_gn_finalize, dim3(semabs_cdiv((long)B * C, 256)), dim3(256), 0, (hipStream_t)stream, sums, gamma, beta, scale,
                       shift, B, C, G, (double)nvox * (C / G), eps);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// =================================================================================================
// Implicit-GEMM convolution family
// =================================================================================================
#ifdef SEMABS_TUNING
#define CONV16_ABL(c) (c)
#else
#define CONV16_ABL(c) false
#endif
struct ConvArgs {
    const void* x; void* y; const f16* w_hi; const f16* w_lo;
    const float* gn_scale; const float* gn_shift; const float* bias; const void* resid;
    int B, I0, I1, I2;          // input spatial dims
    int O0, O1, O2;             // output spatial dims of the full output tensor
    int M0, M1, M2;             // index space this launch covers (conv: = output dims; convT: = input dims per parity class)
    int os, op0, op1, op2;      // output coordinate = m * os + op
    int Cin, Cout, Kp;          // Kp = padded K (multiple of 32) = weight row stride
    int ntaps; int relu;
    int is;                     // input coordinate = m * is + td (1 except for the ConvTranspose data gradient)
    int ksplit;                 // > 1: blockIdx.z takes a slice of the k-steps and adds its partial sum to y (fp32, pre-zeroed) with atomics;
                                // bias / residual / ReLU are then applied by k_conv_finish
    signed char td0[28], td1[28], td2[28];   // tap offsets: input coordinate = m + td  (conv pad-1: -1..1; convT: 0/+1)
    double* stats;              // optional (k_conv16_lds): fp64 [B, 8, 2] sum / sum of squares of the OUTPUT per GroupNorm group, accumulated
    int ablate;                 // tuning only (k_conv16_lds): 1 producers only, 2 consumers only, 4 consumers without epilogue
    long long* trace;           // tuning only (k_conv_brick, tools/conv_probe.py): [workgroup][8] s_memtime stamps of wave 0
    const f16* wp_hi; const f16* wp_lo;   // optional (k_conv_brick): fragment-packed copy of the weights, see SEMABS_CONV_PACKED
    int ncls; long cls_off[8];            // k_conv only: > 0 = ConvTranspose3d, all 8 output parity classes in ONE launch (blockIdx.z = class:
                                          // taps, weight block and output parity are derived from it in the kernel)
    // k_conv16_lds<.., GNB> only (semabs_conv3d_gnbwd): the convolution is a DATA GRADIENT and its epilogue is the GroupNorm backward of the layer input
    // `resid` = X: y = k0 acc - k1 - ((X - mean) rstd) k2 [masked by X > 0], coef = (k0, k1, k2) per (b, channel), mean / rstd per (b, group of gnb_G)
    const float* gnb_coef = nullptr; const float* gnb_mean = nullptr; const float* gnb_rstd = nullptr; int gnb_G = 0, gnb_relu = 0;
    unsigned int* gnb_bits = nullptr;     // optional: receives the bit pattern of max |y|
    // k_conv16_lds only (semabs_conv3d_sparse_stats, round 6): occupancy bitmap of the INPUT (bit v & 31 of word v >> 5 for voxel v of a volume; shared by the
    // B volumes: they are scatters of the same points).  Voxels whose bit is clear ARE zero and their memory is neither initialised nor read.
    const unsigned int* occ = nullptr;
    const float* gnb_add = nullptr;       // optional: a tensor like y added before the mask (the residual branch's gradient)
};

template <bool F32>
__device__ __forceinline__ void load8(const void* base, long idx, float (&v)[8]) {
    if (F32) {
        const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + idx);
        float4 a = p[0], b = p[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
        f16x8 h = *reinterpret_cast<const f16x8*>(reinterpret_cast<const f16*>(base) + idx);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)h[j];
    }
}

// NW = 16-channel output blocks per wave (1, 2, 4); MW = 2 voxel blocks of 16 per wave; 4 waves per workgroup.
// LDSW (the deep levels, NW = 4): the four waves of a workgroup share their output-channel tiles, i.e. their weight fragments - 8 KB per
// k-step that each of them used to pull through the vector L1 on its own (the kernel is bound by L1 bandwidth there: 12 KB of operands per
// wave per 24 MFMAs).  Wave w loads tile w, parks it in a double-buffered LDS slab, and all four read the k-step's four tiles from there.
template <int NW, bool F32, bool CIN16, bool LDSW = false, bool CLS = false>      // CLS: ConvTranspose3d, blockIdx.z = output parity class
__global__ __launch_bounds__(256) void k_conv(ConvArgs a) {
    // the class-dependent launch parameters as locals (the kernel argument itself is never written: a modified by-value struct moves to
    // scratch and slowed every instantiation by 25 %)
    const int ccls = CLS ? (int)blockIdx.z : 0;
    const int cp0 = CLS ? ccls >> 2 : 0, cp1 = CLS ? (ccls >> 1) & 1 : 0, cp2 = CLS ? ccls & 1 : 0;
    const int k_ntaps = CLS ? (cp0 + 1) * (cp1 + 1) * (cp2 + 1) : a.ntaps;
    const int k_Kp = CLS ? k_ntaps * a.Cin : a.Kp;
    const int k_op0 = CLS ? cp0 : a.op0, k_op1 = CLS ? cp1 : a.op1, k_op2 = CLS ? cp2 : a.op2;
    const long k_woff = CLS ? a.cls_off[ccls] : 0;
    const f16* const k_w_hi = a.w_hi + k_woff; const f16* const k_w_lo = a.w_lo ? a.w_lo + k_woff : nullptr;
    const f16* const k_wp_hi = a.wp_hi ? a.wp_hi + k_woff : nullptr; const f16* const k_wp_lo = a.wp_lo ? a.wp_lo + k_woff : nullptr;
    constexpr int MW = 2;
    static_assert(!LDSW || (NW == 4 && !CIN16), "LDSW: four waves x four output-channel tiles");
    __shared__ __attribute__((aligned(16))) f16x8 s_w[LDSW ? 2 : 1][LDSW ? 4 : 1][2][LDSW ? 64 : 1];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int vl = lane & 15, kg = lane >> 4;
    const long Mtot = (long)a.B * a.M0 * a.M1 * a.M2;
    const long mbase = ((long)blockIdx.x * 4 + wid) * (MW * 16);
    if (!LDSW && mbase >= Mtot) return;                    // (LDSW: every wave takes part in the barriers; its voxels are masked by vok)
    const int n0 = blockIdx.y * (NW * 16);
    const bool has_gn = a.gn_scale != nullptr;

    // this lane's voxel in each m-block
    int vb[MW], v0[MW], v1[MW], v2[MW]; bool vok[MW];
#pragma unroll
    for (int mi = 0; mi < MW; ++mi) {
        long m = mbase + mi * 16 + vl;
        vok[mi] = m < Mtot;
        if (!vok[mi]) m = Mtot - 1;
        v2[mi] = (int)(m % a.M2); m /= a.M2;
        v1[mi] = (int)(m % a.M1); m /= a.M1;
        v0[mi] = (int)(m % a.M0); vb[mi] = (int)(m / a.M0);
    }
    f32x4 acc[MW][NW];
#pragma unroll
    for (int mi = 0; mi < MW; ++mi)
#pragma unroll
        for (int ni = 0; ni < NW; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    // GroupNorm affine for Cin == 16: this lane always touches channels 8*(kg&1) .. +8 of its voxel's volume
    float gsc[MW][8], gsh[MW][8];
    if (CIN16 && has_gn) {
#pragma unroll
        for (int mi = 0; mi < MW; ++mi)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                gsc[mi][j] = a.gn_scale[vb[mi] * 16 + 8 * (kg & 1) + j];
                gsh[mi][j] = a.gn_shift[vb[mi] * 16 + 8 * (kg & 1) + j];
            }
    }
    const int cin_steps = CIN16 ? 1 : a.Cin / 32;
    const int nsteps = CIN16 ? (k_ntaps + 1) / 2 : k_ntaps * cin_steps;
    // The k-loop is software-pipelined by hand: the gathered activations and the weight rows of k-step ks + 1 are requested before
    // the MFMAs of k-step ks.  On the deep levels a wave walks 200-400 k-steps of L2 / HBM weight rows with nothing else to hide their
    // latency (few workgroups exist there), and the compiler does not pipeline a loop with a run-time trip count by itself.
    struct Raw { float v[MW][8]; bool ok[MW]; f16x8 wh[NW], wl[NW]; };
    auto fetch = [&](int ks, Raw& R) {
        int tap, c0;
        if (CIN16) { tap = 2 * ks + (kg >> 1); c0 = 8 * (kg & 1); }
        else { tap = ks / cin_steps; c0 = (ks - tap * cin_steps) * 32 + 8 * kg; }
        const bool tap_ok = tap < k_ntaps;
        int d0 = 0, d1 = 0, d2 = 0;
        if (tap_ok) {
            if (CLS) {                                      // taps of a parity class in (t0, t1, t2) order: offset 1 for the first tap of an odd dimension
                const int t2 = tap % (cp2 + 1), r = tap / (cp2 + 1), t1 = r % (cp1 + 1), t0 = r / (cp1 + 1);
                d0 = (cp0 && t0 == 0) ? 1 : 0; d1 = (cp1 && t1 == 0) ? 1 : 0; d2 = (cp2 && t2 == 0) ? 1 : 0;
            } else { d0 = a.td0[tap]; d1 = a.td1[tap]; d2 = a.td2[tap]; }
        }
#pragma unroll
        for (int mi = 0; mi < MW; ++mi) {
            const int i0 = v0[mi] * a.is + d0, i1 = v1[mi] * a.is + d1, i2 = v2[mi] * a.is + d2;
            const bool ok = tap_ok && vok[mi] && i0 >= 0 && i0 < a.I0 && i1 >= 0 && i1 < a.I1 && i2 >= 0 && i2 < a.I2;
            R.ok[mi] = ok;                                  // outside the volume: zero padding, applied AFTER GroupNorm
            if (ok) load8<F32>(a.x, ((((long)vb[mi] * a.I0 + i0) * a.I1 + i1) * a.I2 + i2) * a.Cin + c0, R.v[mi]);
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) R.v[mi][j] = 0.f;
            }
        }
#pragma unroll
        for (int ni = 0; ni < NW; ++ni) {
            if (LDSW && ni != 0) continue;                  // this wave's own tile only (tile wid), kept in slot 0
            const int cb = (n0 >> 4) + (LDSW ? wid : ni);
            const long widx = k_wp_hi ? ((long)(ks * (a.Cout >> 4) + cb) * 64 + lane) * 8 : (long)(cb * 16 + vl) * k_Kp + ks * 32 + kg * 8;
            R.wh[ni] = *reinterpret_cast<const f16x8*>((k_wp_hi ? k_wp_hi : k_w_hi) + widx);
            if (F32) R.wl[ni] = *reinterpret_cast<const f16x8*>((k_wp_hi ? k_wp_lo : k_w_lo) + widx);
        }
    };
    auto park = [&](int ks, Raw& R) {                       // LDSW: this wave's tile of k-step ks -> LDS, then the workgroup barrier
        s_w[ks & 1][wid][0][lane] = R.wh[0];
        if (F32) s_w[ks & 1][wid][1][lane] = R.wl[0];
        __syncthreads();
    };
    auto consume = [&](int ks, Raw& R) {
        int c0 = 0;
        if (!CIN16) { const int tap = ks / cin_steps; c0 = (ks - tap * cin_steps) * 32 + 8 * kg; }
        f16x8 xh[MW], xl[MW];
#pragma unroll
        for (int mi = 0; mi < MW; ++mi) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = R.v[mi][j];
            const bool ok = R.ok[mi];
            if (ok && has_gn) {
                if (CIN16) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = v[j] * gsc[mi][j] + gsh[mi][j];
                } else {
                    const float4* ps = reinterpret_cast<const float4*>(a.gn_scale + (long)vb[mi] * a.Cin + c0);
                    const float4* pt = reinterpret_cast<const float4*>(a.gn_shift + (long)vb[mi] * a.Cin + c0);
                    float4 s0 = ps[0], s1 = ps[1], t0 = pt[0], t1 = pt[1];
                    v[0] = v[0] * s0.x + t0.x; v[1] = v[1] * s0.y + t0.y; v[2] = v[2] * s0.z + t0.z; v[3] = v[3] * s0.w + t0.w;
                    v[4] = v[4] * s1.x + t1.x; v[5] = v[5] * s1.y + t1.y; v[6] = v[6] * s1.z + t1.z; v[7] = v[7] * s1.w + t1.w;
                }
            }
            if (!ok) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                xh[mi][j] = (f16)v[j];
                if (F32) xl[mi][j] = (f16)(v[j] - (float)xh[mi][j]);
            }
        }
#pragma unroll
        for (int ni = 0; ni < NW; ++ni) {
            const f16x8 wh = LDSW ? s_w[ks & 1][ni][0][lane] : R.wh[ni];
            f16x8 wl;
            if (F32) wl = LDSW ? s_w[ks & 1][ni][1][lane] : R.wl[ni];
#pragma unroll
            for (int mi = 0; mi < MW; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[mi], acc[mi][ni], 0, 0, 0);
            if (F32) {
#pragma unroll
                for (int mi = 0; mi < MW; ++mi) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[mi], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[mi], acc[mi][ni], 0, 0, 0);
                }
            }
        }
    };
    int ks_lo = 0, ks_hi = nsteps;
    if (a.ksplit > 1) {
        const int per = (nsteps + a.ksplit - 1) / a.ksplit;
        ks_lo = blockIdx.z * per; ks_hi = ks_lo + per < nsteps ? ks_lo + per : nsteps;
    }
    Raw R0, R1;
    if (ks_lo < ks_hi) fetch(ks_lo, R0);
    if (LDSW) {
        // k-step ks: its four weight tiles are in slab ks & 1 (parked one step earlier); the loads of k-step ks + 1 are in flight during the
        // MFMAs and are parked into the other slab afterwards - that slab was last read in step ks - 1, before the previous barrier.
        if (ks_lo < ks_hi) park(ks_lo, R0);
        for (int ks = ks_lo; ks < ks_hi; ks += 2) {
            if (ks + 1 < ks_hi) fetch(ks + 1, R1);
            consume(ks, R0);
            if (ks + 1 < ks_hi) {
                park(ks + 1, R1);
                if (ks + 2 < ks_hi) fetch(ks + 2, R0);
                consume(ks + 1, R1);
                if (ks + 2 < ks_hi) park(ks + 2, R0);
            }
        }
    } else
    for (int ks = ks_lo; ks < ks_hi; ks += 2) {
        if (ks + 1 < ks_hi) fetch(ks + 1, R1);
        consume(ks, R0);
        if (ks + 1 < ks_hi) {
            if (ks + 2 < ks_hi) fetch(ks + 2, R0);
            consume(ks + 1, R1);
        }
    }
    // epilogue: acc[mi][ni][r] = out[voxel = lane & 15 of block mi][cout = n0 + ni*16 + 4*(lane >> 4) + r]
#pragma unroll
    for (int mi = 0; mi < MW; ++mi) {
        if (!vok[mi]) continue;
        const long ovox = (((long)vb[mi] * a.O0 + (v0[mi] * a.os + k_op0)) * a.O1 + (v1[mi] * a.os + k_op1)) * a.O2 + (v2[mi] * a.os + k_op2);
#pragma unroll
        for (int ni = 0; ni < NW; ++ni) {
            const int co = n0 + ni * 16 + 4 * kg;
            float o[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
            if (F32 && a.ksplit > 1) {                             // split-K partial: plain sums, finished by k_conv_finish
                float* yp = reinterpret_cast<float*>(a.y) + ovox * a.Cout + co;
#pragma unroll
                for (int e = 0; e < 4; ++e) atomicAdd(yp + e, o[e]);
                continue;
            }
            if (a.bias) {
                const float4 bv = *reinterpret_cast<const float4*>(a.bias + co);
                o[0] += bv.x; o[1] += bv.y; o[2] += bv.z; o[3] += bv.w;
            }
            const long oidx = ovox * a.Cout + co;
            if (a.resid) {                                           // (requesting every residual row up front costs this kernel an occupancy step: 286 -> 311 us)
                if (F32) {
                    const float4 r = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.resid) + oidx);
                    o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w;
                } else {
                    const f16x4 r = *reinterpret_cast<const f16x4*>(reinterpret_cast<const f16*>(a.resid) + oidx);
                    o[0] += (float)r[0]; o[1] += (float)r[1]; o[2] += (float)r[2]; o[3] += (float)r[3];
                }
            }
            if (a.relu == 1) { o[0] = fmaxf(o[0], 0.f); o[1] = fmaxf(o[1], 0.f); o[2] = fmaxf(o[2], 0.f); o[3] = fmaxf(o[3], 0.f); }
            else if (a.relu == 2) {                                  // LeakyReLU(0.01): the point / sampler MLP layers run as 1x1x1 convolutions
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = o[j] > 0.f ? o[j] : 0.01f * o[j];
            }
            if (F32) {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.y) + oidx) = make_float4(o[0], o[1], o[2], o[3]);
            } else {
                f16x4 h; h[0] = (f16)o[0]; h[1] = (f16)o[1]; h[2] = (f16)o[2]; h[3] = (f16)o[3];
                *reinterpret_cast<f16x4*>(reinterpret_cast<f16*>(a.y) + oidx) = h;
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Level-0 specialisation: Cin = Cout = 16, 3x3x3, the six full-resolution convolutions that carry most of the
// UNet's HBM traffic.  One workgroup (8 waves) owns an 8 x 8 x 16 output brick: the 10 x 10 x 18 input halo is
// read from HBM ONCE, GroupNorm-applied, split into fp16 hi (+ lo in exact mode) and parked in LDS (57.6 / 115 KB);
// the 27 taps then come from LDS with conflict-free ds_read_b128 (16 consecutive voxels x 16-byte halves), the weights
// (14 k-steps) live in registers, and every 16-voxel row is stored as one contiguous 512 B / 1 KB line.
// -------------------------------------------------------------------------------------------------
#define C16_T1 8
#define C16_T2 16
#define C16_H1 (C16_T1 + 2)
#define C16_H2 (C16_T2 + 2)

// max |x| over the wave -> one atomicMax on the bit pattern (non-negative floats order like unsigned integers)
__device__ __forceinline__ void conv_absmax_commit(unsigned int* bits, float m) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(bits, __float_as_uint(m));
}
// GNB (round 5, exact mode only): the launch is the DATA GRADIENT of a GroupNorm -> Conv3d layer and its epilogue applies the GroupNorm backward to the
// accumulators (see ConvArgs::gnb_*): the layer input's rows come through the residual registers, the per-(volume, channel) coefficients through a small LDS
// table - the separate apply pass (read dXn, read X, write dX: 0.55 ms per layer at 8 x 128^3 x 16) and the dXn tensor itself are gone.
// SPARSE (round 6): the input is described by an occupancy bitmap (ConvArgs::occ) - a compile-time variant, so that the other instantiations keep their registers.
template <bool F32, int C16_T0, bool GNB = false, bool SPARSE = false>      // brick depth 8 (fp16: 2 x 57.6 KB LDS) or 4 (exact: 2 x 69 KB): one 8-wave workgroup per CU, two halo buffers
__global__ __launch_bounds__(512) void k_conv16_lds(ConvArgs a) {
    constexpr int C16_H0 = C16_T0 + 2, C16_HALO = C16_H0 * C16_H1 * C16_H2;
    // One plane = the 16-byte half-voxels (channels 0-7 or 8-15) of the whole halo.  Its size is rounded up to a multiple of 256 B: a
    // ds_read_b128 is served in lane groups such as {0-3, 12-15, 20-27}, i.e. voxels 0-3 / 12-15 of plane 0 together with voxels 4-11 of
    // plane 1 - the group covers all 64 banks exactly once only if the plane offset is = 0 mod 256 B (unpadded it is 128 mod 256: 2-way
    // conflicts on every fragment read, SQ_LDS_BANK_CONFLICT = 46 % of the LDS-active cycles).
    constexpr int PLANE = (C16_HALO * 8 + 127) / 128 * 128;         // fp16 elements
    constexpr int BUF_EL = PLANE * 2 * (F32 ? 2 : 1);               // fp16 elements per halo buffer: hi planes, then lo planes (exact mode)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int vl = lane & 15, kg = lane >> 4;
    const int n2 = a.I2 / C16_T2, n1 = a.I1 / C16_T1, n0 = a.I0 / C16_T0;
    const int per_vol = n0 * n1 * n2, total = a.B * per_vol;
    const bool has_gn = a.gn_scale != nullptr;
    // Wave specialisation.  The PMC profile of the plain form (every wave stages its brick, then computes it) showed the matrix pipe, the
    // VALU (GroupNorm + fp16 hi/lo conversion of the halo) and LDS each ~28 % busy and never at the same time: two co-resident workgroups
    // doing identical work fall into lockstep.  Here one persistent workgroup per CU splits the roles: waves 4-7 (one per SIMD) PRODUCE
    // the halo of brick i + 1 into the second LDS buffer (HBM -> registers -> affine -> split -> LDS) while waves 0-3 (one per SIMD)
    // CONSUME brick i on the matrix pipe; one barrier per brick.  VALU and MFMA now overlap on every SIMD by construction.
    const bool producer = wid >= 4;
    const int ptid = tid - 256;                                     // producer thread index (valid when producer)
    float* const s_gnb = reinterpret_cast<float*>(smem + (size_t)BUF_EL * 2 * 2);     // GNB: [B][3 = (A, Bx, C)][16 channels] behind the two halo buffers
    if (GNB) {
        // y = k0 acc - k1 - (x - mean) rstd k2 = A acc + Bx x + C
        for (int e = tid; e < a.B * 16; e += 512) {
            const int b = e >> 4, c = e & 15, g = c / (16 / a.gnb_G);
            const float mu = a.gnb_mean[b * a.gnb_G + g], rs = a.gnb_rstd[b * a.gnb_G + g];
            const float* k = a.gnb_coef + (long)e * 3;
            s_gnb[b * 48 + c] = k[0]; s_gnb[b * 48 + 16 + c] = -rs * k[2]; s_gnb[b * 48 + 32 + c] = mu * rs * k[2] - k[1];
        }
    }
    float amax = 0.f;

    auto produce = [&](int brick, int buf) {
        const int b = brick / per_vol; int t = brick - b * per_vol;
        const int t2 = t % n2; t /= n2;
        const int t1 = t % n1; const int t0 = t / n1;
        const int z0 = t0 * C16_T0, y0 = t1 * C16_T1, x0 = t2 * C16_T2;
        f16* s_hi = reinterpret_cast<f16*>(smem) + buf * BUF_EL;
        f16* s_lo = s_hi + PLANE * 2;
        // task = (halo voxel, 8-channel half).  Eight consecutive lanes take eight consecutive voxels of ONE half, the next eight lanes the
        // other half of the same voxels: a wave's loads still cover whole 64-byte voxels, its ds_write_b128 lane groups (8 x 8 lanes) write
        // 128 contiguous bytes of one plane, and a thread keeps ONE channel half for all of its tasks, so the GroupNorm scale / shift of its
        // 8 channels are fetched once per brick (indexing them per element made the compiler emit 32 dependent broadcast loads per voxel).
        const int hsel = (ptid >> 3) & 1;
        const int vsub = (ptid & 7) + ((ptid >> 4) << 3);          // 0 .. 127
        float gs[8], gh[8];
        if (has_gn) {
            const float4* ps = reinterpret_cast<const float4*>(a.gn_scale + b * 16 + hsel * 8);
            const float4* pt = reinterpret_cast<const float4*>(a.gn_shift + b * 16 + hsel * 8);
            const float4 s0 = ps[0], s1 = ps[1], h0 = pt[0], h1 = pt[1];
            gs[0] = s0.x; gs[1] = s0.y; gs[2] = s0.z; gs[3] = s0.w; gs[4] = s1.x; gs[5] = s1.y; gs[6] = s1.z; gs[7] = s1.w;
            gh[0] = h0.x; gh[1] = h0.y; gh[2] = h0.z; gh[3] = h0.w; gh[4] = h1.x; gh[5] = h1.y; gh[6] = h1.z; gh[7] = h1.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) { gs[j] = 1.f; gh[j] = 0.f; }
        }
        // Every halo load of the brick is issued before the first one is used (NT x 8 VGPRs; the producer loop below is a separate branch
        // of the kernel, so these registers overlap the consumers' 112 weight registers).  Out-of-volume taps read a clamped (valid)
        // address and are zeroed afterwards: no divergent branches around the loads.  Addresses = brick base (scalar, 64-bit) + a 32-bit
        // per-lane element offset.
        constexpr int NT = (C16_HALO + 127) / 128;                  // tasks per thread
        const float* xb = reinterpret_cast<const float*>(a.x) + (size_t)b * a.I0 * a.I1 * a.I2 * 16;
        const f16* xbh = reinterpret_cast<const f16*>(a.x) + (size_t)b * a.I0 * a.I1 * a.I2 * 16;
        float raw[NT][8];
        bool inb[NT];
        // Sparse input (a.occ, the scattered point features: ~1 - 4 % of the voxels are occupied): the bitmap words of all tasks are requested first; a task whose
        // voxel is empty then reads the volume's first line (one cache line for all of them) and is zeroed BEFORE the GroupNorm affine - the volume's empty
        // voxels are never written by the scatter and never read here (2.1 GB of zero-fill and 2.1 GB of loads per scene).
        [[maybe_unused]] unsigned oword[NT];
        [[maybe_unused]] int vflat[NT];
        if constexpr (SPARSE) {
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                const int v = min(vsub + k * 128, C16_HALO - 1);
                const int hx = v % C16_H2, hy = (v / C16_H2) % C16_H1, hz = v / (C16_H2 * C16_H1);
                const int cz = min(max(z0 + hz - 1, 0), a.I0 - 1), cy = min(max(y0 + hy - 1, 0), a.I1 - 1), cx = min(max(x0 + hx - 1, 0), a.I2 - 1);
                vflat[k] = (cz * a.I1 + cy) * a.I2 + cx;
                oword[k] = a.occ[vflat[k] >> 5];
            }
        }
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const int v = min(vsub + k * 128, C16_HALO - 1);
            const int hx = v % C16_H2, hy = (v / C16_H2) % C16_H1, hz = v / (C16_H2 * C16_H1);
            const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
            inb[k] = gz >= 0 && gz < a.I0 && gy >= 0 && gy < a.I1 && gx >= 0 && gx < a.I2;
            const int cz = min(max(gz, 0), a.I0 - 1), cy = min(max(gy, 0), a.I1 - 1), cx = min(max(gx, 0), a.I2 - 1);
            unsigned eo = (unsigned)(((cz * a.I1 + cy) * a.I2 + cx) * 16 + hsel * 8);      // < 2^31 elements per volume (checked on the host)
            if constexpr (SPARSE) {
                const bool full = (oword[k] >> (vflat[k] & 31)) & 1u;
                oword[k] = full ? 1u : 0u;
                if (!full) eo = (unsigned)(hsel * 8);
            }
            {
                if (F32) {
                    const float4 q0 = *reinterpret_cast<const float4*>(xb + eo), q1 = *reinterpret_cast<const float4*>(xb + eo + 4);
                    raw[k][0] = q0.x; raw[k][1] = q0.y; raw[k][2] = q0.z; raw[k][3] = q0.w; raw[k][4] = q1.x; raw[k][5] = q1.y; raw[k][6] = q1.z; raw[k][7] = q1.w;
                } else {
                    const f16x8 q = *reinterpret_cast<const f16x8*>(xbh + eo);
#pragma unroll
                    for (int c = 0; c < 8; ++c) raw[k][c] = (float)q[c];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const int v = vsub + k * 128;
            f16x8 h, l;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float rv = (SPARSE && !oword[k]) ? 0.f : raw[k][c];            // an empty voxel of a sparse input is zero (its memory holds anything)
                const float val = inb[k] ? rv * gs[c] + gh[c] : 0.f;                 // zero padding AFTER the normalisation
                h[c] = (f16)val;
                if (F32) l[c] = (f16)(val - (float)h[c]);
            }
            if (v < C16_HALO) {                                     // two planes of 16-byte half-voxels: [channels 0-7][v], [channels 8-15][v]
                *reinterpret_cast<f16x8*>(s_hi + hsel * PLANE + v * 8) = h;
                if (F32) *reinterpret_cast<f16x8*>(s_lo + hsel * PLANE + v * 8) = l;
            }
        }
    };

    // consumers: weights (A operand, rows = cout) for all k-steps live in registers.  The 27 taps are grouped by dy (round 2): class dy holds its
    // nine (dz, dx) taps q = dz * 3 + dx as five k-steps {q = 2 s, 2 s + 1} (the tenth slot reads the zero-padded tail of the weight row),
    // because an activation fragment of halo row rho = r + dy with both of its taps in ONE dy class serves every output row r = rho - dy of
    // the wave: 10 halo rows feed the 24 (row, dy) pairs of 8 output rows - 100 fragment reads per 8 rows instead of 224, for 15 instead of
    // 14 k-steps per row (the LDS reads, not the matrix pipe, bounded the consumers: 1.25 ms alone vs 0.81 ms of MFMAs).
    f16x8 wh[3][5], wl[3][5];
    auto load_weights = [&]() {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int st = 0; st < 5; ++st) {
                const int qa = 2 * st, qb = 2 * st + 1;
                const int tapa = (qa / 3) * 9 + dy * 3 + (qa % 3);
                const int tapb = qb < 9 ? (qb / 3) * 9 + dy * 3 + (qb % 3) : 27;      // 27: zeros (Kp = 448 > 27 * 16)
                const long widx = (long)vl * a.Kp + ((kg >> 1) ? tapb : tapa) * 16 + (kg & 1) * 8;
                wh[dy][st] = *reinterpret_cast<const f16x8*>(a.w_hi + widx);
                if (F32) wl[dy][st] = *reinterpret_cast<const f16x8*>(a.w_lo + widx);
            }
    };
    const int half = kg & 1, tsel = kg >> 1;
    f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.bias) bias4 = *reinterpret_cast<const f32x4*>(a.bias + 4 * kg);
    const float relu_floor = a.relu ? 0.f : -__builtin_inff();
    // Fused GroupNorm statistics of the output (8 groups of 2 channels): a lane's four output channels 4 kg .. 4 kg + 3 are the groups 2 kg
    // and 2 kg + 1.  fp32 partial sums per 4-row group, fp64 running sums per wave while the volume index stays the same, then a 16-lane
    // reduction and one fp64 atomic per (group, moment) per wave per volume.
    double st[4] = {0.0, 0.0, 0.0, 0.0};                            // sum / sum of squares of group 2 kg, then of group 2 kg + 1
    int st_b = -1;
    auto flush_stats = [&]() {
        if (st_b >= 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                double v = st[j];
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) v += __shfl_xor(v, m, 64);
                if (vl == 0) atomicAdd(a.stats + ((long)st_b * 8 + 2 * kg + (j >> 1)) * 2 + (j & 1), v);
                st[j] = 0.0;
            }
        }
    };
    constexpr int ROWS_PER_WAVE = C16_T0 * 2;                        // 4 consumer waves share the brick's T0 * 8 rows

    auto consume = [&](int brick, int buf) {
        const int b = brick / per_vol; int t = brick - b * per_vol;
        const int t2 = t % n2; t /= n2;
        const int t1 = t % n1; const int t0 = t / n1;
        const int z0 = t0 * C16_T0, y0 = t1 * C16_T1, x0 = t2 * C16_T2;
        const f16* s_hi = reinterpret_cast<const f16*>(smem) + buf * BUF_EL;
        const f16* s_lo = s_hi + PLANE * 2;
        const size_t vol_off = (size_t)b * a.I0 * a.I1 * a.I2 * 16;
        if (a.stats && b != st_b) { flush_stats(); st_b = b; }
        constexpr int MR = 8;                                       // one group = the 8 output rows (y = 0..7) of one z slice of the brick
        int tapoff[5];                                              // per lane: element offset of its tap (A for kg < 2, B otherwise) of k-step st
#pragma unroll
        for (int st = 0; st < 5; ++st) {
            const int qa = 2 * st, qb = 2 * st + 1 < 9 ? 2 * st + 1 : 2 * st;          // the missing tenth tap reads tap A's voxels (its weights are 0)
            const int q = tsel ? qb : qa;
            tapoff[st] = ((q / 3) * C16_H1 * C16_H2 + (q % 3)) * 8;
        }
#pragma unroll 1
        for (int pr = 0; pr < ROWS_PER_WAVE / MR; ++pr) {
            const int zl = (wid * ROWS_PER_WAVE + pr * MR) >> 3;    // z slice of this group inside the brick
            const int fbase = half * PLANE + (zl * C16_H1 * C16_H2 + vl) * 8;          // halo voxel (zl + dz, rho, vl + dx) = fbase + tapoff + rho * H2 * 8
            f32x4 acc[MR];
#pragma unroll
            for (int mi = 0; mi < MR; ++mi) acc[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
            // Output offsets and the residual rows are fetched now: the loads have the whole k-loop to arrive, and the epilogue after the last
            // MFMA is a dozen branch-free instructions per row (the matrix pipe idles while it runs).
            unsigned ooff[MR];                                      // element offset inside volume b; < 2^31 (checked on the host)
            f32x4 res[MR];
#pragma unroll
            for (int mi = 0; mi < MR; ++mi) {
                ooff[mi] = (unsigned)((((z0 + zl) * a.I1 + (y0 + mi)) * a.I2 + (x0 + vl)) * 16 + 4 * kg);
                res[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (a.resid) {
#pragma unroll
                for (int mi = 0; mi < MR; ++mi) {
                    if (F32) res[mi] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.resid) + vol_off + ooff[mi]);
                    else {
                        const f16x4 r = *reinterpret_cast<const f16x4*>(reinterpret_cast<const f16*>(a.resid) + vol_off + ooff[mi]);
                        res[mi] = f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
                    }
                }
            }
            // Fragment ring over the 50 (k-step, halo row) pairs, THREE slots, requests two pairs ahead (round 4).  A pair carries 3 x (rows it
            // serves) MFMAs: 3 for the halo rows 0 and 9, 6 for rows 1 and 8, 9 for the rest.  With two slots the fragment of pair i + 1 was
            // requested after the first MFMA of pair i, so behind a 3-MFMA pair it had 32 cycles to come out of LDS (~130 needed) and the matrix pipe
            // waited - four such pairs per k-step, back to back across the k-step boundary (rows 8, 9, 0, 1): the consumers alone ran at 0.70 of their
            // MFMA time.  Now (a) the halo rows of a k-step are walked in the order 2 0 3 1 4 9 5 8 6 7, so that a short pair always sits between
            // two 9-MFMA pairs, and (b) pair i + 2 is requested behind the first MFMA of pair i: every fragment has at least 11 MFMAs = 176 cycles.
            // Inside a pair the MFMAs go product-major (wh.xh for every row served, then wl.xh, then wh.xl): dependent MFMAs are up to 3 issues
            // apart.  The sched_barriers pin this order (left alone the compiler merges the buffers and issues the requests next to their first use).
            f16x8 xh[3], xl[3];
            constexpr int RHO[10] = {2, 0, 3, 1, 4, 9, 5, 8, 6, 7};
            auto request = [&](int i, int slot) {
                const int st = i / 10, rho = RHO[i % 10];
                const int off = fbase + tapoff[st] + rho * (C16_H2 * 8);
                xh[slot] = *reinterpret_cast<const f16x8*>(s_hi + off);
                if (F32) xl[slot] = *reinterpret_cast<const f16x8*>(s_lo + off);
            };
            request(0, 0);
            request(1, 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int st = 0; st < 5; ++st) {
#pragma unroll
                for (int ri = 0; ri < 10; ++ri) {
                    const int i = st * 10 + ri, slot = i % 3, rho = RHO[ri];
                    bool first = true;
#pragma unroll
                    for (int kind = 0; kind < (F32 ? 3 : 1); ++kind) {
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy) {
                            const int r = rho - dy;
                            if (r < 0 || r >= MR) continue;
                            acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kind == 1 ? wl[dy][st] : wh[dy][st], kind == 2 ? xl[slot] : xh[slot], acc[r], 0, 0, 0);
                            if (first) {
                                first = false;
                                if (i + 2 < 50) request(i + 2, (i + 2) % 3);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#ifdef SEMABS_TUNING
            if (a.ablate == 4) { if (acc[0][0] != 1234.5f) continue; }
#endif
            // acc[mi][r] = out[voxel x0 + vl of row mi][cout = 4 * kg + r]
            float ps[4] = {0.f, 0.f, 0.f, 0.f};
            f32x4 cA, cB, cC;
            // the optional added tensor's rows are requested two rows ahead of their use, behind the last MFMA (all eight at once, like the residual rows, do not
            // fit: the kernel is at 253 of 256 registers)
            f32x4 ad[MR];
            if (GNB && a.gnb_add) {
                ad[0] = *reinterpret_cast<const f32x4*>(a.gnb_add + vol_off + ooff[0]);
                ad[1] = *reinterpret_cast<const f32x4*>(a.gnb_add + vol_off + ooff[1]);
            }
            if (GNB) {
                cA = *reinterpret_cast<const f32x4*>(s_gnb + b * 48 + 4 * kg); cB = *reinterpret_cast<const f32x4*>(s_gnb + b * 48 + 16 + 4 * kg);
                cC = *reinterpret_cast<const f32x4*>(s_gnb + b * 48 + 32 + 4 * kg);
            }
#pragma unroll
            for (int mi = 0; mi < MR; ++mi) {
                float o[4];
                if (GNB) {
                    if (a.gnb_add && mi + 2 < MR) { ad[mi + 2] = *reinterpret_cast<const f32x4*>(a.gnb_add + vol_off + ooff[mi + 2]); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[e] = acc[mi][e] * cA[e] + (res[mi][e] * cB[e] + cC[e]);
                        if (a.gnb_add) o[e] += ad[mi][e];
                        if (a.gnb_relu) o[e] = res[mi][e] > 0.f ? o[e] : 0.f;
                        amax = fmaxf(amax, fabsf(o[e]));
                    }
                } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = fmaxf(acc[mi][e] + bias4[e] + res[mi][e], relu_floor);
                }
                if (!F32) {                                          // statistics of the values as stored
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (float)(f16)o[e];
                }
                ps[0] += o[0] + o[1]; ps[1] += o[0] * o[0] + o[1] * o[1];
                ps[2] += o[2] + o[3]; ps[3] += o[2] * o[2] + o[3] * o[3];
                if (F32) {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.y) + vol_off + ooff[mi]) = make_float4(o[0], o[1], o[2], o[3]);
                } else {
                    f16x4 h; h[0] = (f16)o[0]; h[1] = (f16)o[1]; h[2] = (f16)o[2]; h[3] = (f16)o[3];
                    *reinterpret_cast<f16x4*>(reinterpret_cast<f16*>(a.y) + vol_off + ooff[mi]) = h;
                }
            }
            if (a.stats) {
#pragma unroll
                for (int j = 0; j < 4; ++j) st[j] += (double)ps[j];
            }
        }
    };

    // Two role loops with the same barrier count (one per brick + one for the first halo).  They are separate branches so that the register
    // allocation of each role only carries its own live values (the weights are not live in the producer loop).
    // The barriers only order LDS traffic (halo buffers); __syncthreads() would also wait for the consumers' global stores to complete
    // (vmcnt(0)) once per brick.
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    // brick of iteration `it`: this XCD's contiguous run of the brick list (xcd_persistent_tile), so that the halo planes neighbouring bricks
    // share are read through ONE L2
    auto brick_of = [&](int it) { return xcd_persistent_tile((int)blockIdx.x, (int)gridDim.x, it, total); };
    if (producer) {
        int it = 0, brick = brick_of(0);
        if (brick >= 0) produce(brick, 0);
        lds_barrier();
        int buf = 0;
        for (; brick >= 0; buf ^= 1) {
            const int next = brick_of(++it);
            if (next >= 0 && !CONV16_ABL(a.ablate == 2 || a.ablate >= 4)) produce(next, buf ^ 1);   // the other buffer: consumed one iteration ago, before the barrier
            lds_barrier();
            brick = next;
        }
    } else {
        load_weights();
        lds_barrier();
        int buf = 0, it = 0;
        for (int brick = brick_of(0); brick >= 0; brick = brick_of(++it), buf ^= 1) {
            if (!CONV16_ABL(a.ablate == 1)) consume(brick, buf);
            lds_barrier();
        }
        if (a.stats) flush_stats();
        if (GNB && a.gnb_bits) conv_absmax_commit(a.gnb_bits, amax);
    }
}

// `act_f32` of the convolution entry points is a flag word: bit 0 = fp32 activations ("exact" mode), bit 8 (SEMABS_CONV_GENERIC) = run
// the generic gather kernel even where a brick kernel exists - the per-call cross-check used by tests/ (no process-global switch).
#define SEMABS_CONV_GENERIC 256
// bit 9 (SEMABS_CONV_PACKED): the w_hi / w_lo buffers carry, right after the [Cout, Kp] matrix (after all eight class matrices for the transposed
// convolution, each packed on its own), a FRAGMENT-PACKED copy: [k-step = Kp / 32][Cout / 16][lane = kg * 16 + row][8] fp16 with
// k = 32 * k-step + 8 * kg + e, so that the A operand of one MFMA (16 output channels x 32 k) is 1 KB of contiguous memory read with
// lane-linear 16-byte pieces.  From the row-major matrix the same fragment is 16 rows x 64 bytes, Kp * 2 bytes apart, and a 128-byte line is
// half used whenever consecutive k-steps of a kernel are not adjacent in k (k_conv_brick walks (channel chunk, tap)): the rolled k-loops
// of the 32^3 / 16^3 levels ran at 0.43 of the MFMA rate on it, at the MFMA rate on the packed copy.
#define SEMABS_CONV_PACKED 512
#ifdef SEMABS_TUNING
static int g_conv16_ablate = 0;      // tuning build only (libsemabs_hip_tune.so)
static long long* g_conv_trace = nullptr;
static int g_brick_persist = 0;      // k_conv_brick: persistent-workgroup variant (measured slower, kept in the tuning build for A/B)
static int g_convt_no_line = 0;      // k_convT_brick: 1 = the per-class half-line stores of round 5 (A/B of the LINE variant)
extern "C" int semabs_conv_tune(int key, long long value) {
    if (key == 0) g_conv16_ablate = (int)value & 7;
    else if (key == 1) g_conv_trace = reinterpret_cast<long long*>(value);
    else if (key == 2) g_brick_persist = (int)value;
    else if (key == 3) g_convt_no_line = (int)value;
    return SEMABS_OK;
}
#endif

static int conv16_lds_launch(const ConvArgs& a_in, int f32, hipStream_t s) {
    ConvArgs a = a_in;
#ifdef SEMABS_TUNING
    a.ablate = g_conv16_ablate;
#endif
    if (f32) {
        constexpr int T0 = 4;
        const size_t lds = (size_t)(((T0 + 2) * C16_H1 * C16_H2 * 8 + 127) / 128 * 128) * 2 * 2 * 2 * 2;   // two half-voxel planes (256-B padded), hi + lo, two buffers; fp16
        const long total = (long)a.B * (a.I0 / T0) * (a.I1 / C16_T1) * (a.I2 / C16_T2);
        long nb = semabs_stream_cus(s); if (nb > total) nb = total;
        if (a.gnb_coef) {
            static SemabsLdsAttr attrg;
            semabs_ensure_lds(&k_conv16_lds<true, T0, true>, (int)(lds + 32 * 192), attrg);
            hipLaunchKernelGGL((k_conv16_lds<true, T0, true>), dim3((unsigned)nb), dim3(512), lds + (size_t)a.B * 192, s, a);
            SEMABS_CHECK_LAUNCH();
            return SEMABS_OK;
        }
        if (a.occ) {
            static SemabsLdsAttr attrs;
            semabs_ensure_lds(&k_conv16_lds<true, T0, false, true>, (int)lds, attrs);
            hipLaunchKernelGGL((k_conv16_lds<true, T0, false, true>), dim3((unsigned)nb), dim3(512), lds, s, a);
            SEMABS_CHECK_LAUNCH();
            return SEMABS_OK;
        }
        static SemabsLdsAttr attr;
        semabs_ensure_lds(&k_conv16_lds<true, T0>, (int)lds, attr);
        hipLaunchKernelGGL((k_conv16_lds<true, T0>), dim3((unsigned)nb), dim3(512), lds, s, a);
    } else {
        constexpr int T0 = 8;
        const size_t lds = (size_t)(((T0 + 2) * C16_H1 * C16_H2 * 8 + 127) / 128 * 128) * 2 * 2 * 2;       // two half-voxel planes (256-B padded), two buffers; fp16
        const long total = (long)a.B * (a.I0 / T0) * (a.I1 / C16_T1) * (a.I2 / C16_T2);
        long nb = semabs_stream_cus(s); if (nb > total) nb = total;
        if (a.occ) {
            static SemabsLdsAttr attrs;
            semabs_ensure_lds(&k_conv16_lds<false, T0, false, true>, (int)lds, attrs);
            hipLaunchKernelGGL((k_conv16_lds<false, T0, false, true>), dim3((unsigned)nb), dim3(512), lds, s, a);
            SEMABS_CHECK_LAUNCH();
            return SEMABS_OK;
        }
        static SemabsLdsAttr attr;
        semabs_ensure_lds(&k_conv16_lds<false, T0>, (int)lds, attr);
        hipLaunchKernelGGL((k_conv16_lds<false, T0>), dim3((unsigned)nb), dim3(512), lds, s, a);
    }
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// -------------------------------------------------------------------------------------------------
// LDS-halo brick kernel for the levels below full resolution (3x3x3, Cin = 16 or a multiple of 32, Cout a multiple of 32) - the
// generic gather kernel re-reads every voxel 27 times through L1 / the texture path and runs at < 10 % of either roof there.  Same
// idea as k_conv16_lds: a workgroup (8 waves) owns a 4 x 8 x 16 output brick and NB x 16 output channels (blockIdx.z slices Cout),
// stages the 6 x 10 x 18 input halo of 32 input channels at a time (GroupNorm applied, split fp16 hi/lo, [voxel][channel] with the
// 16-byte chunk index XOR-ed by (voxel >> 2) & 3, resp. (voxel >> 3) & 1 for Cin = 16, so that the 16 consecutive-x voxels of a
// fragment read cover all 64 banks) and runs the 27 taps of that channel chunk out of LDS.  A wave keeps 4 rows x NB x 16 output
// channels in registers; the weights (too many for registers) stream from L2 once per k-step and wave.
// -------------------------------------------------------------------------------------------------
// TX = tile extent along the innermost axis: 16 (one fragment = 16 consecutive voxels of a row, a wave owns 4 rows) or 8 (volumes 8 wide, the
// 8^3 level: one fragment = two y-rows x 8 voxels, a wave owns 2 fragments; round 2 - that level used to run on the gather kernel at 0.11 of
// its bound).
// GNB (round 5): the data-gradient launch whose epilogue is the GroupNorm backward of the layer input (ConvArgs::gnb_*, as in k_conv16_lds<.., GNB>).
template <bool F32, int CW, int NB, bool ONE, bool PERSIST, int TX = 16, bool GNB = false>   // CW = channels staged at a time (16 or 32); ONE: Cin == CW (compile-time offsets)
__global__ __launch_bounds__(512) void k_conv_brick(ConvArgs a, int total) {
    constexpr int T0 = 4, H0 = T0 + 2, HX = TX + 2, HALO = H0 * C16_H1 * HX, NTHR = 512;
    constexpr int R = TX / 4;                               // fragments (16 voxels each) per wave: the tile has 2 * TX of them
    constexpr int CPV = CW / 8;                             // 16-byte chunks per staged voxel
    constexpr int NSTEPS = CW == 32 ? 27 : 14;
    // Round 2: PERSISTENT workgroups (one or two per CU, tiles t = blockIdx.x, + gridDim.x, ...) and a software pipeline over the staging
    // units (tile, channel chunk): the halo of unit u + 1 is requested into registers before (NB <= 2) or right after (NB = 4: registers)
    // the k-loop of unit u, so its HBM / L2 round trip runs under the MFMAs and the epilogue.  With one workgroup per tile, 7.7 us per
    // tile (of 25) passed between the last store of one workgroup and the first load of the next on the same CU, and staging + epilogue
    // added another 40 % with the matrix pipe idle (tools/conv_probe.py).
    constexpr bool EARLY = false;                           // (true: requested before the k-loop - 72 more live registers: every variant spilled)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS layout: one plane per 8-channel chunk, [CPV][PLANE] 16-byte half-voxels, plane size a multiple of 256 B.  A ds_read_b128 is served
    // in lane groups such as {0-3, 12-15, 20-27} = voxels 0-3 / 12-15 of chunk kg and voxels 4-11 of chunk kg ^ 1: with the planes a
    // multiple of 256 B apart those are 16 distinct 16-byte slots of the 64 banks for any start voxel (the former [voxel][chunk ^ f(voxel)]
    // swizzle assumed contiguous 16-lane groups: SQ_LDS_BANK_CONFLICT was 42 % of the LDS-active cycles).
    constexpr int PLANE = (HALO * 8 + 127) / 128 * 128;    // fp16 elements
    f16* s_hi = reinterpret_cast<f16*>(smem);
    f16* s_lo = s_hi + PLANE * CPV;                         // exact mode only
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int vl = lane & 15, kg = lane >> 4;
    const int n2 = a.I2 / TX, n1 = a.I1 / C16_T1, n0 = a.I0 / T0;
    const int per_vol = n0 * n1 * n2;
    const bool has_gn = a.gn_scale != nullptr;
    const int Cin = ONE ? CW : a.Cin;                       // compile-time when there is a single chunk: immediate weight offsets
    const int nchunks = ONE ? 1 : Cin / CW;
#ifdef SEMABS_TUNING
    long long* trc = a.trace ? a.trace + (long)blockIdx.x * 8 : nullptr;      // stamps of the LAST tile this workgroup processed
#define BRICK_STAMP(i) do { if (trc && tid == 0) trc[i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define BRICK_STAMP(i) do { } while (0)
#endif
    // tile index -> (output-channel slice, volume, brick); the slice is slowest so that concurrently running workgroups share their weights in L2
    auto decode = [&](int t, int& b, int& cout0, int& z0, int& y0, int& x0) {
        int br = t % per_vol; t /= per_vol;
        b = t % a.B; cout0 = (t / a.B) * (NB * 16);
        const int t2 = br % n2; br /= n2;
        const int t1 = br % n1; const int t0 = br / n1;
        z0 = t0 * T0; y0 = t1 * C16_T1; x0 = t2 * TX;
    };
    // fragment f = wid * R + r of the tile -> voxel of lane-slot v (0..15): TX = 16: (z, y, x) = (f >> 3, f & 7, v); TX = 8: (f >> 2, 2 (f & 3) + (v >> 3), v & 7)
    auto frag_z = [&](int f) { return TX == 16 ? f >> 3 : f >> 2; };
    auto frag_y = [&](int f, int v) { return TX == 16 ? (f & 7) : 2 * (f & 3) + (v >> 3); };
    auto frag_x = [&](int v) { return TX == 16 ? v : (v & 7); };
    int rbase[R];                                           // halo voxel index of the centre tap of this lane's voxel in each fragment
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int f = wid * R + r;
        rbase[r] = ((frag_z(f) + 1) * C16_H1 + (frag_y(f, vl) + 1)) * HX + (frag_x(vl) + 1);
    }
    // ---- staging: halo of channels [cc * CW, +CW) -> GroupNorm affine -> fp16 hi/lo -> LDS; task = (halo voxel, 8-channel chunk) ----
    // Chunk fastest over the lanes: a quad of lanes reads ONE 128-byte line (4 x 32 B of a voxel, or 2 voxels x 2 x 32 B for Cin = 16).  A
    // thread keeps its chunk for all of its tasks (one GroupNorm scale / shift fetch per unit); all of its halo loads are in flight
    // together; out-of-volume taps read a clamped address and are zeroed afterwards (no divergent branches around the loads).
    constexpr int VPI = NTHR / CPV;                         // halo voxels staged per iteration
    constexpr int NIT = (HALO + VPI - 1) / VPI;
    const int c = tid % CPV;
    const int vsub = tid / CPV;
    float raw[NIT][8];
    float gs[8], gh[8];
    unsigned inb = 0;                                       // bit it: task it lies inside the volume
    auto issue = [&](int tile, int cc) {
        int b, cout0, z0, y0, x0;
        decode(tile, b, cout0, z0, y0, x0);
        const int ch = cc * CW + c * 8;
        if (has_gn) {
            const float4* ps = reinterpret_cast<const float4*>(a.gn_scale + (long)b * Cin + ch);
            const float4* pt = reinterpret_cast<const float4*>(a.gn_shift + (long)b * Cin + ch);
            const float4 s0 = ps[0], s1 = ps[1], h0 = pt[0], h1 = pt[1];
            gs[0] = s0.x; gs[1] = s0.y; gs[2] = s0.z; gs[3] = s0.w; gs[4] = s1.x; gs[5] = s1.y; gs[6] = s1.z; gs[7] = s1.w;
            gh[0] = h0.x; gh[1] = h0.y; gh[2] = h0.z; gh[3] = h0.w; gh[4] = h1.x; gh[5] = h1.y; gh[6] = h1.z; gh[7] = h1.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) { gs[j] = 1.f; gh[j] = 0.f; }
        }
        inb = 0;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int v = min(vsub + it * VPI, HALO - 1);
            const int hx = v % HX, hy = (v / HX) % C16_H1, hz = v / (HX * C16_H1);
            const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
            if (gz >= 0 && gz < a.I0 && gy >= 0 && gy < a.I1 && gx >= 0 && gx < a.I2) inb |= 1u << it;
            const int cz = min(max(gz, 0), a.I0 - 1), cy = min(max(gy, 0), a.I1 - 1), cx = min(max(gx, 0), a.I2 - 1);
            load8<F32>(a.x, ((((long)b * a.I0 + cz) * a.I1 + cy) * a.I2 + cx) * Cin + ch, raw[it]);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int v = vsub + it * VPI;
            const bool in = (inb >> it) & 1u;
            f16x8 h, l;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float val = in ? raw[it][j] * gs[j] + gh[j] : 0.f;          // zero padding AFTER the normalisation
                h[j] = (f16)val; if (F32) l[j] = (f16)(val - (float)h[j]);
            }
            if (v < HALO) {
                const int off = c * PLANE + v * 8;
                *reinterpret_cast<f16x8*>(s_hi + off) = h;
                if (F32) *reinterpret_cast<f16x8*>(s_lo + off) = l;
            }
        }
    };

    int tile = PERSIST ? (int)blockIdx.x : xcd_remap((int)blockIdx.x, (int)gridDim.x);      // (not persistent: the grid is the tile list)
    if (tile >= total) return;
    BRICK_STAMP(6);
    issue(tile, 0);
    for (;;) {
    int b, cout0, z0, y0, x0;
    decode(tile, b, cout0, z0, y0, x0);
    BRICK_STAMP(0);
    f32x4 acc[R][NB];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[r][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int cc = 0; cc < nchunks; ++cc) {
        __syncthreads();                                    // every wave is done reading the previous LDS image
        commit();
        BRICK_STAMP(2);
        __syncthreads();
        BRICK_STAMP(3);
        int ntile = tile, ncc = cc + 1;
        if (ncc == nchunks) { ntile = PERSIST ? tile + (int)gridDim.x : total; ncc = 0; }
        if (EARLY && ntile < total) issue(ntile, ncc);
        // ---- 27 taps of this channel chunk; wave = 4 rows (16 voxels along x each) x NB x 16 output channels ----
        // The weights of k-step ks + 1 are requested BEFORE the MFMAs of k-step ks, and all eight activation fragments of a step are
        // requested up front so that the LDS latency is paid once per step.  Fully unrolled with immediate offsets when Cin is a compile-time
        // constant; a rolled loop for the multi-chunk variant.
        auto step_geom = [&](int ks, int& voff, int& chunk, long& kofs) {     // kofs: k index of this lane's 8 weights = tap * Cin + channel
            if (CW == 32) {
                chunk = kg; voff = ((ks / 9) - 1) * C16_H1 * HX + (((ks / 3) % 3) - 1) * HX + ((ks % 3) - 1);
                kofs = (long)ks * Cin + cc * 32 + kg * 8;
            } else {
                const int ta = 2 * ks, tb = (2 * ks + 1 < 27) ? 2 * ks + 1 : 26;  // tap 27 has zero weights: reuse tap 26's address
                const int offa = ((ta / 9) - 1) * C16_H1 * HX + (((ta / 3) % 3) - 1) * HX + ((ta % 3) - 1);
                const int offb = ((tb / 9) - 1) * C16_H1 * HX + (((tb / 3) % 3) - 1) * HX + ((tb % 3) - 1);
                voff = (kg >> 1) ? offb : offa; chunk = kg & 1;
                // k index = tap * Cin + channel; tap 27 (second tap of the last step) does not exist: with Cin == 16 the padded weight row holds
                // zeros there, otherwise its lanes read tap 26's weights again and load_w zeroes them
                const int tap = ONE ? 2 * ks + (kg >> 1) : min(2 * ks + (kg >> 1), 26);
                kofs = (long)tap * Cin + cc * 16 + (kg & 1) * 8;
            }
        };
        const bool packed = a.wp_hi != nullptr;
        const int ncb = a.Cout >> 4;                        // output-channel blocks of 16 in the packed layout
        auto w_index = [&](int ks, int nb, long kofs) -> long {          // element index of this lane's 8 weights of (k-step ks, block nb)
            return packed ? ((long)((CW == 32 ? ks * (Cin >> 5) + cc : ks) * ncb + (cout0 >> 4) + nb) * 64 + lane) * 8
                          : (long)(cout0 + nb * 16 + vl) * a.Kp + kofs;
        };
        const f16* const wbase_h = packed ? a.wp_hi : a.w_hi;
        const f16* const wbase_l = packed ? a.wp_lo : a.w_lo;
        auto load_w = [&](int ks, f16x8* wh, f16x8* wl) {
            int voff, chunk; long kofs;
            step_geom(ks, voff, chunk, kofs);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const long widx = w_index(ks, nb, kofs);
                wh[nb] = *reinterpret_cast<const f16x8*>(wbase_h + widx);
                if (F32) wl[nb] = *reinterpret_cast<const f16x8*>(wbase_l + widx);
                if (CW == 16 && !ONE) {
                    const bool dead = ks == NSTEPS - 1 && (kg >> 1);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { wh[nb][e] = dead ? (f16)0.f : wh[nb][e]; if (F32) wl[nb][e] = dead ? (f16)0.f : wl[nb][e]; }
                }
            }
        };
        // Activation fragments are loop-carried: xh / xl hold the fragments of the CURRENT k-step on entry, and each of them is re-requested
        // in place for the next k-step right behind its last MFMA (in the last output-channel block), so that neither the LDS round trip nor the
        // address arithmetic of the rolled loops stands between two MFMA blocks (they did: the rolled k-loop ran at 0.7 / 0.43 of the MFMA
        // rate with two / four output blocks).
        f16x8 xh[R], xl[R];
        auto read_x = [&](int ks) {
            int voff, chunk; long kofs;
            step_geom(ks, voff, chunk, kofs);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int off = chunk * PLANE + (rbase[r] + voff) * 8;
                xh[r] = *reinterpret_cast<const f16x8*>(s_hi + off);
                if (F32) xl[r] = *reinterpret_cast<const f16x8*>(s_lo + off);
            }
        };
        // one k-step.  whn / wln != nullptr: the next step's weights go to a second register set, requested before the MFMAs (NB <= 2);
        // nullptr: every block's weights are re-requested in place behind its own MFMAs (NB = 4)
        auto step = [&](int ks, f16x8* wh, f16x8* wl, f16x8* whn, f16x8* wln) {
            const int ksn = ks + 1 < NSTEPS ? ks + 1 : ks;      // (the last step re-requests its own operands: no branch in the loop)
            int voffn, chunkn; long kofsn;
            step_geom(ksn, voffn, chunkn, kofsn);
            int offn[R];
#pragma unroll
            for (int r = 0; r < R; ++r) offn[r] = chunkn * PLANE + (rbase[r] + voffn) * 8;
            if (whn) load_w(ksn, whn, wln);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const bool lastb = nb == NB - 1;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    acc[r][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[nb], xh[r], acc[r][nb], 0, 0, 0);
                    if (!F32 && lastb) { xh[r] = *reinterpret_cast<const f16x8*>(s_hi + offn[r]); __builtin_amdgcn_sched_barrier(0); }
                }
                if (F32) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        acc[r][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[nb], xh[r], acc[r][nb], 0, 0, 0);
                        if (lastb) { xh[r] = *reinterpret_cast<const f16x8*>(s_hi + offn[r]); __builtin_amdgcn_sched_barrier(0); }
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        acc[r][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[nb], xl[r], acc[r][nb], 0, 0, 0);
                        if (lastb) { xl[r] = *reinterpret_cast<const f16x8*>(s_lo + offn[r]); __builtin_amdgcn_sched_barrier(0); }
                    }
                }
                if (!whn) {
                    __builtin_amdgcn_sched_barrier(0);
                    const long widx = w_index(ksn, nb, kofsn);
                    wh[nb] = *reinterpret_cast<const f16x8*>(wbase_h + widx);
                    if (F32) wl[nb] = *reinterpret_cast<const f16x8*>(wbase_l + widx);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        read_x(0);
        if constexpr (NB <= 2) {
            f16x8 wa_h[NB], wa_l[NB], wb_h[NB], wb_l[NB];      // two weight fragment sets, alternating by k-step parity
            load_w(0, wa_h, wa_l);
            if (ONE && !PERSIST) {                          // (full unroll: in the persistent kernel its hoisted invariants spill)
#pragma unroll
                for (int ks = 0; ks < NSTEPS; ks += 2) {
                    step(ks, wa_h, wa_l, wb_h, wb_l);
                    if (ks + 1 < NSTEPS) step(ks + 1, wb_h, wb_l, wa_h, wa_l);
                }
            } else {
                // rolled two steps at a time (the unrolled form of the multi-chunk variant makes the scheduler hoist dozens of run-time-addressed
                // weight loads and spill); NSTEPS is odd for CW == 32 -> one tail step
#pragma unroll 1
                for (int ks = 0; ks + 1 < NSTEPS; ks += 2) {
                    step(ks, wa_h, wa_l, wb_h, wb_l);
                    step(ks + 1, wb_h, wb_l, wa_h, wa_l);
                }
                if (NSTEPS & 1) step(NSTEPS - 1, wa_h, wa_l, wb_h, wb_l);
            }
        } else {
            // four output-channel blocks: a second weight set would not fit the registers.  The blocks are walked one after the other and a
            // block's fragments are re-requested IN PLACE for the next k-step as soon as its twelve MFMAs are issued: every request has the
            // other three blocks' MFMAs (3/4 of a step) to arrive.
            f16x8 wh[NB], wl[NB];
            load_w(0, wh, wl);
            constexpr int KU = (ONE && !PERSIST) ? NSTEPS : 1;
#pragma unroll KU
            for (int ks = 0; ks < NSTEPS; ++ks) step(ks, wh, wl, nullptr, nullptr);
        }
        if (!EARLY && ntile < total) issue(ntile, ncc);
    }
    BRICK_STAMP(4);
    // acc[r][nb][e] = out[voxel x0 + vl of row r][cout = cout0 + nb * 16 + 4 * kg + e]: lane = kg * 16 + voxel, i.e. consecutive lanes are
    // consecutive VOXELS, Cout * 4 bytes apart - a store in that layout touches four different lines per lane quad.  The accumulators are
    // therefore transposed across lanes first (ds_bpermute: register exchange through the LDS crossbar, no memory): lane' = voxel * 4 + kg,
    // so that a quad of lanes owns 64 contiguous bytes of one voxel; bias, residual and ReLU are applied in that layout.
    {
        const int v2 = lane >> 2, k2 = lane & 3;
        const int src = (k2 * 16 + v2) << 2;                 // byte address of the source lane for ds_bpermute
        // (the lanes' values are passed as scalars through __float_as_int: `__builtin_bit_cast(int, acc[r][nb][e])` on the vector element made
        //  clang 19 / ROCm 7.2 emit ONE ds_bpermute per accumulator and copy its result to all four elements)
        long obase[R];                                       // element index of (voxel, cout0 + 4 * k2); output block nb is 16 channels further
        float q[R][NB][4];
        float st_s[NB], st_q[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) { st_s[nb] = 0.f; st_q[nb] = 0.f; }
        // GNB: (k0, k1, k2) of this lane's four channels per output block - 12 contiguous floats - and their group's mean / rstd, requested together
        // with the rows of the layer input (q): one round trip
        float gk[NB][12], gmu[NB], grs[NB];
        float amax = 0.f;
        if (GNB) {
            const int cg = a.Cout / a.gnb_G;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int co = cout0 + nb * 16 + 4 * k2;
                const float4* kp = reinterpret_cast<const float4*>(a.gnb_coef + ((long)b * a.Cout + co) * 3);
                const float4 k0 = kp[0], k1 = kp[1], k2v = kp[2];
                gk[nb][0] = k0.x; gk[nb][1] = k0.y; gk[nb][2] = k0.z; gk[nb][3] = k0.w; gk[nb][4] = k1.x; gk[nb][5] = k1.y; gk[nb][6] = k1.z; gk[nb][7] = k1.w;
                gk[nb][8] = k2v.x; gk[nb][9] = k2v.y; gk[nb][10] = k2v.z; gk[nb][11] = k2v.w;
                gmu[nb] = a.gnb_mean[b * a.gnb_G + co / cg]; grs[nb] = a.gnb_rstd[b * a.gnb_G + co / cg];
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int f = wid * R + r;
            const long ovox = (((long)b * a.I0 + (z0 + frag_z(f))) * a.I1 + (y0 + frag_y(f, v2))) * a.I2 + (x0 + frag_x(v2));
            obase[r] = ovox * a.Cout + (cout0 + 4 * k2);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const long oi = obase[r] + nb * 16;
                if (a.resid) {                               // every residual row is requested before the first store (one round trip, not 4 x NB)
                    if (F32) { const float4 t = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.resid) + oi); q[r][nb][0] = t.x; q[r][nb][1] = t.y; q[r][nb][2] = t.z; q[r][nb][3] = t.w; }
                    else { const f16x4 t = *reinterpret_cast<const f16x4*>(reinterpret_cast<const f16*>(a.resid) + oi); q[r][nb][0] = (float)t[0]; q[r][nb][1] = (float)t[1]; q[r][nb][2] = (float)t[2]; q[r][nb][3] = (float)t[3]; }
                } else {
                    q[r][nb][0] = q[r][nb][1] = q[r][nb][2] = q[r][nb][3] = 0.f;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int co = cout0 + nb * 16 + 4 * k2;
                const long oi = obase[r] + nb * 16;
                const float a0 = acc[r][nb][0], a1 = acc[r][nb][1], a2 = acc[r][nb][2], a3 = acc[r][nb][3];
                float o[4];
                o[0] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(a0)));
                o[1] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(a1)));
                o[2] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(a2)));
                o[3] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(a3)));
                if (GNB) {                                   // o = the data gradient dXn of channels co .. co + 3, q = the layer input x: dX = k0 dXn - k1 - (x - mean) rstd k2
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float xv = q[r][nb][j];
                        float t = gk[nb][3 * j] * o[j] - gk[nb][3 * j + 1] - ((xv - gmu[nb]) * grs[nb]) * gk[nb][3 * j + 2];
                        if (a.gnb_relu) t = xv > 0.f ? t : 0.f;
                        o[j] = t;
                        amax = fmaxf(amax, fabsf(t));
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] += q[r][nb][j];
                }
                if (a.bias) { const float4 bv = *reinterpret_cast<const float4*>(a.bias + co); o[0] += bv.x; o[1] += bv.y; o[2] += bv.z; o[3] += bv.w; }
                if (a.relu) { o[0] = fmaxf(o[0], 0.f); o[1] = fmaxf(o[1], 0.f); o[2] = fmaxf(o[2], 0.f); o[3] = fmaxf(o[3], 0.f); }
                if (F32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.y) + oi) = make_float4(o[0], o[1], o[2], o[3]);
                else {
                    f16x4 h; h[0] = (f16)o[0]; h[1] = (f16)o[1]; h[2] = (f16)o[2]; h[3] = (f16)o[3]; *reinterpret_cast<f16x4*>(reinterpret_cast<f16*>(a.y) + oi) = h;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (float)h[e];          // statistics of the values as stored
                }
                st_s[nb] += (o[0] + o[1]) + (o[2] + o[3]);
                st_q[nb] += (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
            }
        }
        if (GNB && a.gnb_bits) conv_absmax_commit(a.gnb_bits, amax);
        // Fused GroupNorm statistics of the output (8 groups; a lane's four channels always lie in one group): fp32 per lane over its
        // R fragments -> reduction over the 16 voxel lanes -> LDS float atomics per group of this NB x 16-channel slice -> one fp64
        // atomic per (group, moment) per workgroup.  Saves the statistics pass over y that the next layer's GroupNorm needs.
        if (a.stats) {
            const int cg = a.Cout >> 3;                     // channels per group (>= 4)
            float* s_stat = reinterpret_cast<float*>(smem); // [local group][2]; the operand planes are dead after the barrier
            __syncthreads();                                // every wave is done with the k-loop's LDS reads
            if (tid < 16) s_stat[tid] = 0.f;
            __syncthreads();
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                float vs = st_s[nb], vq = st_q[nb];
#pragma unroll
                for (int m = 4; m < 64; m <<= 1) { vs += __shfl_xor(vs, m, 64); vq += __shfl_xor(vq, m, 64); }
                if (v2 == 0) {
                    const int gl = (nb * 16 + 4 * k2) / cg;                  // group index local to this slice
                    atomicAdd(&s_stat[gl * 2], vs); atomicAdd(&s_stat[gl * 2 + 1], vq);
                }
            }
            __syncthreads();
            const int ngl = (NB * 16 >= cg) ? NB * 16 / cg : 1;
            if (tid < ngl * 2) atomicAdd(a.stats + ((long)b * 8 + cout0 / cg + (tid >> 1)) * 2 + (tid & 1), (double)s_stat[tid]);
            if (PERSIST) __syncthreads();
        }
    }
    BRICK_STAMP(5);
    if (!PERSIST) break;
    tile += gridDim.x;
    if (tile >= total) break;
    }
    BRICK_STAMP(7);
}

template <bool F32, int CW, int NB, bool ONE, bool PERSIST, int TX = 16, bool GNB = false>
static int conv_brick_launch_p(const ConvArgs& a, hipStream_t s) {
    const size_t lds = (size_t)((6 * C16_H1 * (TX + 2) * 8 + 127) / 128 * 128) * (CW / 8) * 2 * (F32 ? 2 : 1);      // CW / 8 planes (256-B padded), hi (+ lo), fp16
    static int per_cu = 0;                                  // co-resident workgroups per CU (LDS and registers), queried once
    if (!per_cu) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_brick<F32, CW, NB, ONE, PERSIST, TX, GNB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_conv_brick<F32, CW, NB, ONE, PERSIST, TX, GNB>, 512, lds) != hipSuccess || n < 1) n = 1;
        per_cu = n;
    }
    const long total = (long)(a.I0 / 4) * (a.I1 / C16_T1) * (a.I2 / TX) * a.B * (a.Cout / (NB * 16));
    SEMABS_REQUIRE(total < (1L << 30), "conv brick: too many tiles");
    long nwg = total;
    if (PERSIST) { nwg = (long)semabs_stream_cus(s) * per_cu; if (nwg > total) nwg = total; }   // as many workgroups as fit the chip at once
    ConvArgs at = a;
#ifdef SEMABS_TUNING
    at.trace = g_conv_trace;
#endif
    hipLaunchKernelGGL((k_conv_brick<F32, CW, NB, ONE, PERSIST, TX, GNB>), dim3((unsigned)nwg), dim3(512), lds, s, at, (int)total);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}
template <bool F32, int CW, int NB, bool ONE>
static int conv_brick_launch_t(const ConvArgs& a, hipStream_t s) {
#ifdef SEMABS_TUNING
    if (g_brick_persist) return conv_brick_launch_p<F32, CW, NB, ONE, true>(a, s);
#endif
    return conv_brick_launch_p<F32, CW, NB, ONE, false>(a, s);
}
static int conv_brick_launch(const ConvArgs& a, int f32, hipStream_t s) {
    if (a.gnb_coef) {                                       // semabs_conv3d_gnbwd below full resolution: exact mode, Cin % 32 == 0, Cout % 32 == 0, two output blocks per wave
        if (a.Cin == 32) return conv_brick_launch_p<true, 32, 2, true, false, 16, true>(a, s);
        return conv_brick_launch_p<true, 32, 2, false, false, 16, true>(a, s);
    }
    if (a.I2 % C16_T2 != 0) {                               // 8 wide (the 8^3 level; Cin >= 32 there): 4 x 8 x 8 tiles, two output blocks
        return f32 ? conv_brick_launch_p<true, 32, 2, false, false, 8>(a, s) : conv_brick_launch_p<false, 32, 2, false, false, 8>(a, s);
    }
    const long bricks = (long)a.B * (a.I0 / 4) * (a.I1 / C16_T1) * (a.I2 / C16_T2);
    if (a.Cout == 16) {                                     // one 16-channel output block per wave (round 5: the data gradient of the 16 -> 32 convolution at
        // 64^3 - a 32 -> 16 convolution - ran on the generic gather kernel: 1.29 ms against 0.2 ms for the 16 -> 32 forward)
        if (a.Cin == 32) return f32 ? conv_brick_launch_t<true, 32, 1, true>(a, s) : conv_brick_launch_t<false, 32, 1, true>(a, s);
        return f32 ? conv_brick_launch_t<true, 32, 1, false>(a, s) : conv_brick_launch_t<false, 32, 1, false>(a, s);
    }
    const bool nb4 = a.Cout % 64 == 0 && bricks * (a.Cout / 64) >= 512;      // wider Cout slices only while the grid still fills the chip
    if (a.Cin == 16) {
        if (nb4) return f32 ? conv_brick_launch_t<true, 16, 4, true>(a, s) : conv_brick_launch_t<false, 16, 4, true>(a, s);
        return f32 ? conv_brick_launch_t<true, 16, 2, true>(a, s) : conv_brick_launch_t<false, 16, 2, true>(a, s);
    }
    // (Tried in round 2: 16-channel chunks for Cin >= 32 - half the LDS, two workgroups per CU so that one stages / stores while the other is in
    //  its k-loop.  Slower, 1 198 vs 1 027 us at 64^3 x 32 ch: two co-resident workgroups doing identical work fall into lockstep, both stage
    //  and both multiply at the same time - the same behaviour that led to the producer / consumer split of k_conv16_lds.)
    if (a.Cin == 32) {
        if (nb4) return f32 ? conv_brick_launch_t<true, 32, 4, true>(a, s) : conv_brick_launch_t<false, 32, 4, true>(a, s);
        return f32 ? conv_brick_launch_t<true, 32, 2, true>(a, s) : conv_brick_launch_t<false, 32, 2, true>(a, s);
    }
    if (nb4) return f32 ? conv_brick_launch_t<true, 32, 4, false>(a, s) : conv_brick_launch_t<false, 32, 4, false>(a, s);
    return f32 ? conv_brick_launch_t<true, 32, 2, false>(a, s) : conv_brick_launch_t<false, 32, 2, false>(a, s);
}

// split-K epilogue: y = act(y + bias + resid) in place, fp32 channels-last
__global__ void k_conv_finish(float* __restrict__ y, const float* __restrict__ bias, const float* __restrict__ resid, long n4, int C, int relu) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 v = *reinterpret_cast<float4*>(y + i * 4);
    if (bias) { const float4 b = *reinterpret_cast<const float4*>(bias + (i * 4) % C); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
    if (resid) { const float4 r = *reinterpret_cast<const float4*>(resid + i * 4); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
    if (relu == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    else if (relu == 2) { v.x = v.x > 0.f ? v.x : 0.01f * v.x; v.y = v.y > 0.f ? v.y : 0.01f * v.y; v.z = v.z > 0.f ? v.z : 0.01f * v.z; v.w = v.w > 0.f ? v.w : 0.01f * v.w; }
    *reinterpret_cast<float4*>(y + i * 4) = v;
}

static int conv_launch(const ConvArgs& a_in, int f32, hipStream_t s) {
    ConvArgs a = a_in;
    const long Mtot = (long)a.B * a.M0 * a.M1 * a.M2;
    // The deepest levels (8^3, 4^3 voxels x 256 / 512 channels) have so few voxels that even 16-channel output slices leave the k-loop
    // (216 / 432 steps) as the only parallelism left - and every slice re-gathers the input.  In exact mode those launches split the
    // k-steps over grid z instead (wide 64-channel slices, 4x less input traffic), accumulate with fp32 atomics into a zeroed output and
    // apply bias / residual / ReLU in a second, tiny pass.
    a.ksplit = 1;
    if (f32 && a.os == 1 && a.Cout % 64 == 0 && a.Cin % 32 == 0 && a.ntaps == 27 && Mtot <= 16384) {
        const long slices = semabs_cdiv(Mtot, 4 * 32) * (a.Cout / 64);
        int ks = (int)(2048 / (slices > 0 ? slices : 1));
        if (ks > 27) ks = 27;
        if (ks >= 3) {
            a.ksplit = ks;
            semabs_fill32(a.y, (size_t)Mtot * a.Cout * sizeof(float), 0u, s);
            dim3 grid(semabs_cdiv(Mtot, 4 * 32), a.Cout / 64, ks), block(256);
            hipLaunchKernelGGL((k_conv<4, true, false, true>), grid, block, 0, s, a);
            const long n4 = Mtot * a.Cout / 4;
            hipLaunchKernelGGL(k_conv_finish, dim3(semabs_cdiv(n4, 256)), dim3(256), 0, s, reinterpret_cast<float*>(a.y), a.bias,
                               reinterpret_cast<const float*>(a.resid), n4, a.Cout, a.relu);
            SEMABS_CHECK_LAUNCH();
            return SEMABS_OK;
        }
    }
    // output channels per wave: 64 / 32 / 16.  Wide slices reuse each gathered input fragment more, but the deep UNet levels have only
    // a few thousand voxels: there the k-loop (up to 432 steps of dependent L2 weight loads) is latency-bound and the chip is filled
    // by slicing Cout finer instead (4^3 x 512 channels: 32 -> 128 workgroups).
    const int ncls = a.ncls > 0 ? a.ncls : 1;
    int nw = a.Cout >= 64 ? 4 : (a.Cout >= 32 ? 2 : 1);
    while (nw > 1 && (long)semabs_cdiv(Mtot, 4 * 32) * (a.Cout / (nw * 16)) * ncls < 1024) nw >>= 1;
    dim3 grid(semabs_cdiv(Mtot, 4 * 32), a.Cout / (nw * 16), ncls), block(256);
    const bool c16 = a.Cin == 16;
    if (a.ncls > 0) {                                       // ConvTranspose3d, eight parity classes in grid z (Cin % 32 == 0)
#define CONVC_GO(NW_, F_) hipLaunchKernelGGL((k_conv<NW_, F_, false, false, true>), grid, block, 0, s, a)
        if (f32) { if (nw == 4) CONVC_GO(4, true); else if (nw == 2) CONVC_GO(2, true); else CONVC_GO(1, true); }
        else { if (nw == 4) CONVC_GO(4, false); else if (nw == 2) CONVC_GO(2, false); else CONVC_GO(1, false); }
#undef CONVC_GO
        SEMABS_CHECK_LAUNCH();
        return SEMABS_OK;
    }
#define CONV_GO(NW_, F_, C_) hipLaunchKernelGGL((k_conv<NW_, F_, C_>), grid, block, 0, s, a)
#define CONV_NW(F_, C_) { if (nw == 4) CONV_GO(4, F_, C_); else if (nw == 2) CONV_GO(2, F_, C_); else CONV_GO(1, F_, C_); }
    if (nw == 4 && !c16) {                                  // four waves x four shared weight tiles: weights through LDS
        if (f32) hipLaunchKernelGGL((k_conv<4, true, false, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((k_conv<4, false, false, true>), grid, block, 0, s, a);
    }
    else if (f32) { if (c16) CONV_NW(true, true) else CONV_NW(true, false) }
    else { if (c16) CONV_NW(false, true) else CONV_NW(false, false) }
#undef CONV_NW
#undef CONV_GO
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

static int conv_common_checks(const void* x, const void* w_hi, const void* w_lo, void* y, int Cin, int Cout, int f32) {
    SEMABS_REQUIRE(x && w_hi && y, "semabs conv: null pointer");
    SEMABS_REQUIRE(!f32 || w_lo, "semabs conv: exact (fp32) mode needs the lo half of the split weights");
    SEMABS_REQUIRE(Cin == 16 || Cin % 32 == 0, "semabs conv: Cin must be 16 or a multiple of 32");
    SEMABS_REQUIRE(Cout % 16 == 0 && (Cout < 64 || Cout % 64 == 0), "semabs conv: Cout must be 16, 32 or a multiple of 64");
    return SEMABS_OK;
}

// Conv3d k (1 or 3), stride 1, padding k/2.  x [B, D0, D1, D2, Cin] -> y [B, D0, D1, D2, Cout] (fp16, or fp32 when
// act_f32).  w_hi / w_lo fp16 [Cout, Kp], k index = ((kd*3 + kh)*3 + kw) * Cin + cin, Kp = K rounded up to 32.
// gn_scale / gn_shift fp32 [B, Cin] (null = no GroupNorm in front); bias fp32 [Cout] or null; resid like y or null.
// relu: 0 none, 1 ReLU, 2 LeakyReLU(0.01) (generic kernel only: ksize 1, i.e. the MLP layers of the training step).
// out_sums (optional): fp64 [B, out_groups, 2], ZERO-FILLED by the caller; receives the GroupNorm statistics (sum, sum of squares per group)
// of the output y - fused into the convolution's epilogue where the kernel supports it (the 128^3 x 16-channel level), otherwise by a
// statistics pass over y on the same stream.
static int conv3d_impl(const void* x, const void* w_hi, const void* w_lo, void* y, const float* gn_scale, const float* gn_shift,
                       const float* bias, const void* resid, int B, int D0, int D1, int D2, int Cin, int Cout, int ksize,
                       int relu, int act_flags, double* out_sums, int out_groups, void* stream, const unsigned int* occ = nullptr) {
    if (B == 0) return SEMABS_OK;
    const int act_f32 = act_flags & 1;
    const bool bricks = !(act_flags & SEMABS_CONV_GENERIC);
    int rc = conv_common_checks(x, w_hi, w_lo, y, Cin, Cout, act_f32);
    if (rc) return rc;
    SEMABS_REQUIRE(ksize == 1 || ksize == 3, "semabs_conv3d: kernel size must be 1 or 3");
    SEMABS_REQUIRE((gn_scale == nullptr) == (gn_shift == nullptr), "semabs_conv3d: gn_scale and gn_shift go together");
    ConvArgs a;
    a.x = x; a.y = y; a.w_hi = (const f16*)w_hi; a.w_lo = (const f16*)w_lo; a.gn_scale = gn_scale; a.gn_shift = gn_shift;
    a.bias = bias; a.resid = resid; a.B = B; a.I0 = D0; a.I1 = D1; a.I2 = D2; a.O0 = D0; a.O1 = D1; a.O2 = D2;
    a.M0 = D0; a.M1 = D1; a.M2 = D2; a.os = 1; a.is = 1; a.op0 = a.op1 = a.op2 = 0; a.Cin = Cin; a.Cout = Cout; a.relu = relu;
    a.ntaps = ksize * ksize * ksize;
    a.Kp = ((a.ntaps * Cin + 31) / 32) * 32;
    a.stats = nullptr; a.ablate = 0; a.trace = nullptr; a.wp_hi = nullptr; a.wp_lo = nullptr; a.ncls = 0;
    if (act_flags & SEMABS_CONV_PACKED) {
        a.wp_hi = a.w_hi + (long)Cout * a.Kp;
        if (w_lo) a.wp_lo = a.w_lo + (long)Cout * a.Kp;
    }
    int t = 0;
    for (int kd = 0; kd < ksize; ++kd)
        for (int kh = 0; kh < ksize; ++kh)
            for (int kw = 0; kw < ksize; ++kw, ++t) { a.td0[t] = kd - ksize / 2; a.td1[t] = kh - ksize / 2; a.td2[t] = kw - ksize / 2; }
    SEMABS_REQUIRE(relu == 0 || relu == 1 || (relu == 2 && ksize == 1), "semabs_conv3d: relu must be 0, 1, or 2 (LeakyReLU, ksize 1 only)");
    SEMABS_REQUIRE(!out_sums || (out_groups > 0 && Cout % out_groups == 0), "semabs_conv3d_stats: Cout must be a multiple of out_groups");
    bool fused = false;
    if (bricks && ksize == 3 && Cin == 16 && Cout == 16 && D0 % 8 == 0 && D1 % C16_T1 == 0 && D2 % C16_T2 == 0 && (long)D0 * D1 * D2 * 16 < (1L << 31)) {
        if (out_sums && out_groups == 8) { a.stats = out_sums; fused = true; }
        a.occ = occ;
        rc = conv16_lds_launch(a, act_f32, (hipStream_t)stream);
    } else if (occ) {
        semabs_set_error("semabs_conv3d_sparse_stats: the occupancy bitmap is read by the 16 -> 16 level-0 kernel only (D0 % 8, D1 % 8, D2 % 16 == 0)");
        return SEMABS_EINVAL;
    } else if (bricks && ksize == 3 && (Cout % 32 == 0 || (Cout == 16 && Cin % 32 == 0 && D2 % C16_T2 == 0)) && (Cin == 16 || Cin % 32 == 0) && relu != 2 &&
               D0 % 4 == 0 && D1 % C16_T1 == 0 && (D2 % C16_T2 == 0 || (D2 % 8 == 0 && Cin % 32 == 0))) {
        if (out_sums && out_groups == 8 && Cout >= 32) { a.stats = out_sums; fused = true; }     // (a lane's four output channels lie in one of the 8 groups: Cout >= 32)
        rc = conv_brick_launch(a, act_f32, (hipStream_t)stream);
    } else {
        rc = conv_launch(a, act_f32, (hipStream_t)stream);
    }
    if (rc == SEMABS_OK && out_sums && !fused) rc = semabs_gn_stats(y, out_sums, B, (long)D0 * D1 * D2, Cout, out_groups, act_f32, stream);
    return rc;
}
extern "C" int semabs_conv3d(const void* x, const void* w_hi, const void* w_lo, void* y, const float* gn_scale, const float* gn_shift,
                             const float* bias, const void* resid, int B, int D0, int D1, int D2, int Cin, int Cout, int ksize,
                             int relu, int act_f32, void* stream) {
    return conv3d_impl(x, w_hi, w_lo, y, gn_scale, gn_shift, bias, resid, B, D0, D1, D2, Cin, Cout, ksize, relu, act_f32, nullptr, 0, stream);
}
extern "C" int semabs_conv3d_stats(const void* x, const void* w_hi, const void* w_lo, void* y, const float* gn_scale, const float* gn_shift,
                                   const float* bias, const void* resid, int B, int D0, int D1, int D2, int Cin, int Cout, int ksize,
                                   int relu, int act_f32, double* out_sums, int out_groups, void* stream) {
    SEMABS_REQUIRE(out_sums, "semabs_conv3d_stats: out_sums is null");
    return conv3d_impl(x, w_hi, w_lo, y, gn_scale, gn_shift, bias, resid, B, D0, D1, D2, Cin, Cout, ksize, relu, act_f32, out_sums, out_groups, stream);
}

// semabs_conv3d_stats on a SPARSE input: occ = occupancy bitmap of x (semabs_scatter_mean_stats(.., occ)), bit v & 31 of word v >> 5 for voxel v, one bitmap for
// all B volumes.  Empty voxels are zero by definition: x need not be initialised there and is not read.  16 -> 16 channels, 3 x 3 x 3, D0 % 8 == D1 % 8 == D2 % 16 == 0.
extern "C" int semabs_conv3d_sparse_stats(const void* x, const unsigned int* occ, const void* w_hi, const void* w_lo, void* y, const float* gn_scale, const float* gn_shift,
                                          const float* bias, const void* resid, int B, int D0, int D1, int D2, int Cin, int Cout, int ksize,
                                          int relu, int act_f32, double* out_sums, int out_groups, void* stream) {
    SEMABS_REQUIRE(out_sums && occ, "semabs_conv3d_sparse_stats: out_sums / occ is null");
    SEMABS_REQUIRE(!(act_f32 & SEMABS_CONV_GENERIC), "semabs_conv3d_sparse_stats: the generic kernel reads the whole volume");
    return conv3d_impl(x, w_hi, w_lo, y, gn_scale, gn_shift, bias, resid, B, D0, D1, D2, Cin, Cout, ksize, relu, act_f32, out_sums, out_groups, stream, occ);
}

// Data gradient of a GroupNorm -> Conv3d 3x3x3 layer WITH the GroupNorm backward applied in the epilogue (round 5):
//   dX = k0 conv(dZ, Wbwd) - k1 - ((X - mean) rstd) k2 [+ add1]  [0 where X <= 0 if relu_mask],   coef fp32 [B, Cin_of_the_layer, 3] = (k0, k1, k2) of semabs_gn_bwd_coef,
// i.e. semabs_conv3d(dZ ..) followed by semabs_gn_bwd_apply(.., add1) without the intermediate tensor.  dZ fp32 [B, D0, D1, D2, 16] with its dynamic
// scale as the input affine (in_scale / in_shift), X fp32 = the layer's GroupNorm input, mean / rstd fp32 [B, G].  Shapes: 16 -> 16 channels, D0 % 8 == 0,
// D1 % 8 == 0, D2 % 16 == 0, B <= 32 (the level-0 kernel), or channel counts that are multiples of 32 with D0 % 4 == 0 and without add1 (the brick kernel);
// ask semabs_conv3d_gnbwd_supported.
// 1 = the level-0 kernel (16 -> 16), 2 = the brick kernel (Cin % 32 == 0, Cout % 32 == 0: a lane's four output channels lie in one GroupNorm group; no add1), 0 = neither
static int conv3d_gnbwd_route(int B, int D0, int D1, int D2, int Cin, int Cout, int G, bool have_add) {
    if (B <= 0 || G <= 0) return 0;
    if (B <= 32 && Cin == 16 && Cout == 16 && D0 % 8 == 0 && D1 % C16_T1 == 0 && D2 % C16_T2 == 0 && (long)D0 * D1 * D2 * 16 < (1L << 31) && 16 % G == 0) return 1;
    if (!have_add && Cin % 32 == 0 && Cout % 32 == 0 && Cout % G == 0 && (Cout / G) % 4 == 0 && D0 % 4 == 0 && D1 % C16_T1 == 0 && D2 % C16_T2 == 0) return 2;
    return 0;
}
extern "C" int semabs_conv3d_gnbwd_supported(int B, int D0, int D1, int D2, int Cin, int Cout, int G, int have_add1, int* ok) {
    SEMABS_REQUIRE(ok, "semabs_conv3d_gnbwd_supported: null pointer");
    *ok = conv3d_gnbwd_route(B, D0, D1, D2, Cin, Cout, G, have_add1 != 0) ? 1 : 0;
    return SEMABS_OK;
}
extern "C" int semabs_conv3d_gnbwd(const void* dZ, const void* w_hi, const void* w_lo, void* dX, const float* in_scale, const float* in_shift, const void* X,
                                   const float* mean, const float* rstd, const float* coef, int G, const float* add1, int relu_mask, unsigned int* absmax_bits,
                                   int B, int D0, int D1, int D2, int Cin, int Cout, int act_flags, void* stream) {
    if (B == 0) return SEMABS_OK;
    SEMABS_REQUIRE(dZ && w_hi && w_lo && dX && X && mean && rstd && coef, "semabs_conv3d_gnbwd: null pointer");
    const int route = conv3d_gnbwd_route(B, D0, D1, D2, Cin, Cout, G, add1 != nullptr);
    SEMABS_REQUIRE((act_flags & 1) && route, "semabs_conv3d_gnbwd: exact mode and a supported shape (see semabs_conv3d_gnbwd_supported)");
    SEMABS_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "semabs_conv3d_gnbwd: in_scale and in_shift go together");
    ConvArgs a;
    a.x = dZ; a.y = dX; a.w_hi = (const f16*)w_hi; a.w_lo = (const f16*)w_lo; a.gn_scale = in_scale; a.gn_shift = in_shift;
    a.bias = nullptr; a.resid = X; a.B = B; a.I0 = D0; a.I1 = D1; a.I2 = D2; a.O0 = D0; a.O1 = D1; a.O2 = D2;
    a.M0 = D0; a.M1 = D1; a.M2 = D2; a.os = 1; a.is = 1; a.op0 = a.op1 = a.op2 = 0; a.Cin = Cin; a.Cout = Cout; a.relu = 0;
    a.ntaps = 27; a.Kp = ((27 * Cin + 31) / 32) * 32;
    a.stats = nullptr; a.ablate = 0; a.trace = nullptr; a.wp_hi = nullptr; a.wp_lo = nullptr; a.ncls = 0; a.ksplit = 1;
    int t = 0;
    for (int kd = 0; kd < 3; ++kd)
        for (int kh = 0; kh < 3; ++kh)
            for (int kw = 0; kw < 3; ++kw, ++t) { a.td0[t] = kd - 1; a.td1[t] = kh - 1; a.td2[t] = kw - 1; }
    a.gnb_coef = coef; a.gnb_mean = mean; a.gnb_rstd = rstd; a.gnb_G = G; a.gnb_relu = relu_mask; a.gnb_bits = absmax_bits; a.gnb_add = add1;
    if (act_flags & SEMABS_CONV_PACKED) { a.wp_hi = a.w_hi + (long)Cout * a.Kp; a.wp_lo = a.w_lo + (long)Cout * a.Kp; }
    return route == 1 ? conv16_lds_launch(a, 1, (hipStream_t)stream) : conv_brick_launch(a, 1, (hipStream_t)stream);
}

// General gather convolution (used for the data gradients of training): y[b, m, :] = sum_taps W_tap . x[b, m * in_stride + td_tap, :]
// with zero padding outside x.  x [B, I0, I1, I2, Cin] -> y [B, M0, M1, M2, Cout]; w_hi / w_lo fp16 [Cout, Kp], k = tap * Cin + cin,
// Kp = ntaps * Cin rounded up to 32; taps int8 [ntaps, 3] (host); in_scale / in_shift fp32 [B, Cin] = optional input affine (null = none).
extern "C" int semabs_conv3d_gather(const void* x, const void* w_hi, const void* w_lo, void* y, const float* in_scale, const float* in_shift,
                                    int B, int I0, int I1, int I2, int M0, int M1, int M2, int in_stride, int Cin, int Cout, int ntaps,
                                    const signed char* taps, int act_f32, void* stream) {
    if (B == 0) return SEMABS_OK;
    const int act_flags = act_f32;
    act_f32 = act_flags & 1;
    int rc = conv_common_checks(x, w_hi, w_lo, y, Cin, Cout, act_f32);
    if (rc) return rc;
    SEMABS_REQUIRE(taps && ntaps >= 1 && ntaps <= 28 && in_stride >= 1, "semabs_conv3d_gather: bad taps / stride");
    ConvArgs a;
    SEMABS_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "semabs_conv3d_gather: in_scale and in_shift go together");
    a.x = x; a.y = y; a.w_hi = (const f16*)w_hi; a.w_lo = (const f16*)w_lo; a.gn_scale = in_scale; a.gn_shift = in_shift;
    a.bias = nullptr; a.resid = nullptr; a.stats = nullptr; a.ablate = 0; a.trace = nullptr; a.wp_hi = nullptr; a.wp_lo = nullptr; a.ncls = 0; a.B = B; a.I0 = I0; a.I1 = I1; a.I2 = I2; a.O0 = M0; a.O1 = M1; a.O2 = M2;
    a.M0 = M0; a.M1 = M1; a.M2 = M2; a.os = 1; a.is = in_stride; a.op0 = a.op1 = a.op2 = 0; a.Cin = Cin; a.Cout = Cout; a.relu = 0;
    a.ntaps = ntaps; a.Kp = ((ntaps * Cin + 31) / 32) * 32;
    if (act_flags & SEMABS_CONV_PACKED) { a.wp_hi = a.w_hi + (long)Cout * a.Kp; if (a.w_lo) a.wp_lo = a.w_lo + (long)Cout * a.Kp; }
    for (int t = 0; t < ntaps; ++t) { a.td0[t] = taps[t * 3]; a.td1[t] = taps[t * 3 + 1]; a.td2[t] = taps[t * 3 + 2]; }
    return conv_launch(a, act_f32, (hipStream_t)stream);
}

// -------------------------------------------------------------------------------------------------
// ConvTranspose3d in TWO launches (output parity along axis 0) for volumes that tile into 4 x 8 x 16 input bricks (the three upper levels).  The eight parity-class
// launches of the gather kernel read the input once per tap (27 x the tensor through L2) and write every output line in 64-byte halves;
// here a workgroup (8 waves) stages the 5 x 9 x 17 input halo of 32 channels at a time in LDS (split fp16, bank-swizzled like
// k_conv_brick) and produces four classes of its 8 x 16 x 32 output brick: wave = 4 input rows, 4 classes x 16 output channels in
// registers (16 accumulators), weights streamed from L2.  y = skip + convT(x) + bias as before; Cout sliced by 16 over grid z.
// -------------------------------------------------------------------------------------------------
struct ConvTArgs {
    const void* x; void* y; const f16* w_hi; const f16* w_lo; const float* bias; const void* skip;
    long class_off[8];
    int B, D0, D1, D2, Cin, Cout;
    double* stats;              // optional: fp64 [B, 8, 2] GroupNorm statistics (8 groups) of the output (after bias and skip), accumulated
    long packed_off;            // > 0: element offset of the fragment-packed copies of the class matrices (SEMABS_CONV_PACKED), same class offsets
};
// LINE (round 6; fp32, Cout == 16: the 128^3 level, 3/4 of the kernel's bytes): the outputs of input voxel x for the classes p2 = 0 / 1 are the adjacent 64-byte
// records of output voxels 2 x, 2 x + 1 = ONE 128-byte line per lane quad, but a store instruction of one class touched only its half of 16 different lines (and the
// skip loads likewise).  Neighbouring lanes (x, x + 1) swap one class (DPP) so that each instruction covers WHOLE lines: the even-x lines in the first, the odd-x
// lines in the second.  Same values, same arithmetic, same addresses in total: bit-identical output.
template <bool F32, int P0, bool LINE = false>           // P0 = output parity along axis 0: two launches of four classes each (acc registers)
__global__ __launch_bounds__(512) void k_convT_brick(ConvTArgs a) {
    constexpr int T0 = 4, T1 = 8, T2 = 16, H0 = T0 + 1, H1 = T1 + 1, H2 = T2 + 1, HALO = H0 * H1 * H2, NTHR = 512, CW = 32, CPV = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f16* s_hi = reinterpret_cast<f16*>(smem);               // [HALO][32], chunk-swizzled
    f16* s_lo = s_hi + HALO * CW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int vl = lane & 15, kg = lane >> 4;
    const int b = blockIdx.y, cout0 = blockIdx.z * 16;
    const int n2 = a.D2 / T2, n1 = a.D1 / T1;
    int t = (gridDim.y * gridDim.z == 1 || gridDim.x % 8 == 0) ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;   // x is the fastest grid axis: with gridDim.x % 8 == 0 the XCD of a workgroup is blockIdx.x % 8
    const int t2 = t % n2; t /= n2;
    const int t1 = t % n1; const int t0 = t / n1;
    const int z0 = t0 * T0, y0 = t1 * T1, x0 = t2 * T2;
    auto swz = [](int v, int chunk) { return chunk ^ ((v >> 2) & 3); };

    f32x4 acc[4][4];                                        // [input row of this wave][parity class (p1, p2)]
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    int rbase[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = wid * 4 + r;                        // z * 8 + y of the input brick
        rbase[r] = ((row >> 3) * H1 + (row & 7)) * H2 + vl;
    }
    const int nchunks = a.Cin / CW;
    for (int cc = 0; cc < nchunks; ++cc) {
        if (cc) __syncthreads();
        constexpr int NIT = (HALO * CPV + NTHR - 1) / NTHR;
        const int c = tid % CPV;
        const int ch = cc * CW + c * 8;
        float raw[NIT][8];
        bool inb[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int v = (tid + it * NTHR) / CPV;
            const int hx = v % H2, hy = (v / H2) % H1, hz = v / (H2 * H1);
            const int gz = z0 + hz, gy = y0 + hy, gx = x0 + hx;                 // halo on the high side only (taps reach m and m + 1)
            inb[it] = v < HALO && gz < a.D0 && gy < a.D1 && gx < a.D2;
            if (inb[it]) load8<F32>(a.x, ((((long)b * a.D0 + gz) * a.D1 + gy) * a.D2 + gx) * a.Cin + ch, raw[it]);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int v = (tid + it * NTHR) / CPV;
            f16x8 h, l;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float val = inb[it] ? raw[it][j] : 0.f;
                h[j] = (f16)val; if (F32) l[j] = (f16)(val - (float)h[j]);
            }
            if (v < HALO) {
                const int off = v * CW + swz(v, c) * 8;
                *reinterpret_cast<f16x8*>(s_hi + off) = h;
                if (F32) *reinterpret_cast<f16x8*>(s_lo + off) = l;
            }
        }
        __syncthreads();
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            constexpr int p0 = P0;
            const int cls = P0 * 4 + c4, p1 = (c4 >> 1) & 1, p2 = c4 & 1;
            const int ntaps = (p0 + 1) * (p1 + 1) * (p2 + 1);
            const long kp = (long)ntaps * a.Cin;                                // row stride of this class's weight matrix
            const f16* wh_c = a.w_hi + a.packed_off + a.class_off[cls];
            const f16* wl_c = F32 ? a.w_lo + a.packed_off + a.class_off[cls] : nullptr;
            int tt = 0;
#pragma unroll
            for (int q0 = 0; q0 <= p0; ++q0)
#pragma unroll
                for (int q1 = 0; q1 <= p1; ++q1)
#pragma unroll
                    for (int q2 = 0; q2 <= p2; ++q2, ++tt) {
                        const int d0 = p0 ? (q0 == 0 ? 1 : 0) : 0, d1 = p1 ? (q1 == 0 ? 1 : 0) : 0, d2 = p2 ? (q2 == 0 ? 1 : 0) : 0;
                        const int voff = (d0 * H1 + d1) * H2 + d2;
                        const long widx = a.packed_off ? ((long)((tt * (a.Cin >> 5) + cc) * (a.Cout >> 4) + (cout0 >> 4)) * 64 + lane) * 8
                                                       : (long)(cout0 + vl) * kp + (long)tt * a.Cin + cc * 32 + kg * 8;
                        const f16x8 wh = *reinterpret_cast<const f16x8*>(wh_c + widx);
                        f16x8 wl;
                        if (F32) wl = *reinterpret_cast<const f16x8*>(wl_c + widx);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int v = rbase[r] + voff;
                            const int off = v * CW + swz(v, kg) * 8;
                            const f16x8 xh = *reinterpret_cast<const f16x8*>(s_hi + off);
                            acc[r][c4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh, acc[r][c4], 0, 0, 0);
                            if (F32) {
                                const f16x8 xl = *reinterpret_cast<const f16x8*>(s_lo + off);
                                acc[r][c4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh, acc[r][c4], 0, 0, 0);
                                acc[r][c4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl, acc[r][c4], 0, 0, 0);
                            }
                        }
                        if (tt & 1) __builtin_amdgcn_sched_barrier(0);          // keep the scheduler from hoisting all 9 / 18 taps' weight rows (spills)
                    }
        }
    }
    // acc[r][c4][e] = convT output at (2 z + P0, 2 y + p1, 2 (x0 + vl) + p2), channel cout0 + 4 kg + e
    const int O1 = 2 * a.D1, O2 = 2 * a.D2;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias) bv = *reinterpret_cast<const float4*>(a.bias + cout0 + 4 * kg);
    // fused GroupNorm statistics of the output: a lane's channels 4 kg, 4 kg + 1 | 4 kg + 2, 4 kg + 3 are two halves that each lie inside one
    // of the 8 groups (Cout / 8 >= 2 channels per group); fp32 per lane -> 16-lane reduction -> LDS float atomics per group of this
    // 16-channel slice -> one fp64 atomic per (group, moment) per workgroup
    float* s_stat = reinterpret_cast<float*>(smem);         // [8 local groups][2]
    float st_s[2] = {0.f, 0.f}, st_q[2] = {0.f, 0.f};
    if (a.stats) {
        __syncthreads();                                    // every wave is done with the operand tiles
        if (tid < 16) s_stat[tid] = 0.f;
        __syncthreads();
    }
    if constexpr (LINE) {
        static_assert(!LINE || F32, "the full-line store variant is for fp32 activations with 16 output channels");
        const bool odd = (vl & 1) != 0;
        auto xchg = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false)); };   // lane ^ 1
        // slot [r][p1][i]: i = 0 -> the line of the even x of this lane pair, i = 1 -> the odd x's line; this lane's 16 bytes sit at (odd ? 64 : 0) + 16 kg of it
        float sk[4][2][2][4];
        long oidx[4][2][2];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = wid * 4 + r;
            const int z = z0 + (row >> 3), y = y0 + (row & 7);
#pragma unroll
            for (int p1 = 0; p1 < 2; ++p1)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int x = x0 + ((vl & ~1) | i);
                    const long ovox = (((long)b * 2 * a.D0 + 2 * z + P0) * O1 + 2 * y + p1) * O2 + 2 * x + (odd ? 1 : 0);
                    oidx[r][p1][i] = ovox * 16 + 4 * kg;
                    if (a.skip) {
                        const f32x4 q = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.skip) + oidx[r][p1][i]));
                        sk[r][p1][i][0] = q[0]; sk[r][p1][i][1] = q[1]; sk[r][p1][i][2] = q[2]; sk[r][p1][i][3] = q[3];
                    } else {
                        sk[r][p1][i][0] = sk[r][p1][i][1] = sk[r][p1][i][2] = sk[r][p1][i][3] = 0.f;
                    }
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int p1 = 0; p1 < 2; ++p1) {
                float o0[4], o1[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float va = acc[r][2 * p1][e], vb = acc[r][2 * p1 + 1][e];          // this lane's class p2 = 0 / p2 = 1 values
                    const float got = xchg(odd ? va : vb);                                  // even lanes receive the odd neighbour's p2 = 0, odd lanes the even neighbour's p2 = 1
                    o0[e] = ((odd ? got : va) + bb[e]) + sk[r][p1][0][e];                    // goes to the even x's line
                    o1[e] = ((odd ? vb : got) + bb[e]) + sk[r][p1][1][e];                    // goes to the odd x's line
                }
                __builtin_nontemporal_store(f32x4{o0[0], o0[1], o0[2], o0[3]}, reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.y) + oidx[r][p1][0]));
                __builtin_nontemporal_store(f32x4{o1[0], o1[1], o1[2], o1[3]}, reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.y) + oidx[r][p1][1]));
                st_s[0] += (o0[0] + o0[1]) + (o1[0] + o1[1]); st_q[0] += (o0[0] * o0[0] + o0[1] * o0[1]) + (o1[0] * o1[0] + o1[1] * o1[1]);
                st_s[1] += (o0[2] + o0[3]) + (o1[2] + o1[3]); st_q[1] += (o0[2] * o0[2] + o0[3] * o0[3]) + (o1[2] * o1[2] + o1[3] * o1[3]);
            }
    } else {
    // The skip rows of ALL 16 (row, class) outputs are requested before the first store: `skip` and `y` may alias as far as the compiler
        // knows, so load - add - store per output was 16 dependent round trips.
        float sk[4][4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = wid * 4 + r;
            const int z = z0 + (row >> 3), y = y0 + (row & 7), x = x0 + vl;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const int p0 = P0, p1 = (c4 >> 1) & 1, p2 = c4 & 1;
                const long ovox = (((long)b * 2 * a.D0 + 2 * z + p0) * O1 + 2 * y + p1) * O2 + 2 * x + p2;
                const long oidx = ovox * a.Cout + cout0 + 4 * kg;
                if (a.skip) {
                    if (F32) { const f32x4 q = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.skip) + oidx)); sk[r][c4][0] = q[0]; sk[r][c4][1] = q[1]; sk[r][c4][2] = q[2]; sk[r][c4][3] = q[3]; }
                    else { const f16x4 q = *reinterpret_cast<const f16x4*>(reinterpret_cast<const f16*>(a.skip) + oidx); sk[r][c4][0] = (float)q[0]; sk[r][c4][1] = (float)q[1]; sk[r][c4][2] = (float)q[2]; sk[r][c4][3] = (float)q[3]; }
                } else {
                    sk[r][c4][0] = sk[r][c4][1] = sk[r][c4][2] = sk[r][c4][3] = 0.f;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = wid * 4 + r;
            const int z = z0 + (row >> 3), y = y0 + (row & 7), x = x0 + vl;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const int p0 = P0, p1 = (c4 >> 1) & 1, p2 = c4 & 1;
                const long ovox = (((long)b * 2 * a.D0 + 2 * z + p0) * O1 + 2 * y + p1) * O2 + 2 * x + p2;
                const long oidx = ovox * a.Cout + cout0 + 4 * kg;
                float o[4] = {acc[r][c4][0] + bv.x + sk[r][c4][0], acc[r][c4][1] + bv.y + sk[r][c4][1], acc[r][c4][2] + bv.z + sk[r][c4][2], acc[r][c4][3] + bv.w + sk[r][c4][3]};
                if (F32) __builtin_nontemporal_store(f32x4{o[0], o[1], o[2], o[3]}, reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.y) + oidx));
                else {
                    f16x4 h; h[0] = (f16)o[0]; h[1] = (f16)o[1]; h[2] = (f16)o[2]; h[3] = (f16)o[3]; *reinterpret_cast<f16x4*>(reinterpret_cast<f16*>(a.y) + oidx) = h;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (float)h[e];                  // statistics of the values as stored
                }
                st_s[0] += o[0] + o[1]; st_q[0] += o[0] * o[0] + o[1] * o[1];
                st_s[1] += o[2] + o[3]; st_q[1] += o[2] * o[2] + o[3] * o[3];
            }
        }
    }
    if (a.stats) {
        const int cg = a.Cout / 8;                          // channels per group
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            float vs = st_s[hf], vq = st_q[hf];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) { vs += __shfl_xor(vs, m, 64); vq += __shfl_xor(vq, m, 64); }
            if (vl == 0) {
                const int gl = (cg >= 16) ? 0 : (4 * kg + 2 * hf) / cg;          // group index local to this 16-channel slice
                atomicAdd(&s_stat[gl * 2], vs); atomicAdd(&s_stat[gl * 2 + 1], vq);
            }
        }
        __syncthreads();
        const int ngl = (cg >= 16) ? 1 : 16 / cg;
        if (tid < ngl * 2) atomicAdd(a.stats + ((long)b * 8 + cout0 / cg + (tid >> 1)) * 2 + (tid & 1), (double)s_stat[tid]);
    }
}
#ifdef SEMABS_TUNING
#define CONVT_NO_LINE (g_convt_no_line != 0)                 // tuning build: semabs_conv_tune(3, 1)
#else
#define CONVT_NO_LINE (false)
#endif
template <bool F32>
static int convT_brick_launch(const ConvTArgs& a, hipStream_t s) {
    const size_t lds = (size_t)5 * 9 * 17 * 32 * 2 * (F32 ? 2 : 1);
    static SemabsLdsAttr attr0, attr1;
    semabs_ensure_lds(&k_convT_brick<F32, 0>, (int)lds, attr0);
    semabs_ensure_lds(&k_convT_brick<F32, 1>, (int)lds, attr1);
    dim3 grid((a.D0 / 4) * (a.D1 / 8) * (a.D2 / 16), a.B, a.Cout / 16);
    if constexpr (F32) {
        if (a.Cout == 16 && !CONVT_NO_LINE) {                // whole-line stores / skip loads (see the kernel's LINE comment)
            static SemabsLdsAttr attr0l, attr1l;
            semabs_ensure_lds(&k_convT_brick<true, 0, true>, (int)lds, attr0l);
            semabs_ensure_lds(&k_convT_brick<true, 1, true>, (int)lds, attr1l);
            hipLaunchKernelGGL((k_convT_brick<true, 0, true>), grid, dim3(512), lds, s, a);
            hipLaunchKernelGGL((k_convT_brick<true, 1, true>), grid, dim3(512), lds, s, a);
            SEMABS_CHECK_LAUNCH();
            return SEMABS_OK;
        }
    }
    hipLaunchKernelGGL((k_convT_brick<F32, 0>), grid, dim3(512), lds, s, a);
    hipLaunchKernelGGL((k_convT_brick<F32, 1>), grid, dim3(512), lds, s, a);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// ConvTranspose3d k3 s2 p1 output_padding 1 (+bias) fused with the sum-joining skip: y = skip + convT(x) + bias.
// x [B, D0, D1, D2, Cin] -> y [B, 2 D0, 2 D1, 2 D2, Cout].  One launch per output parity class (8 of them): output
// o = 2 m + p takes input m (kernel index 1) when p = 0, inputs m + 1 (k = 0) and m (k = 2) when p = 1.
// w_hi / w_lo: 8 matrices, class c = p0*4 + p1*2 + p2 at offset class_off[c] (elements), each [Cout, ntaps_c * Cin]
// with tap order (t0, t1, t2) over each dimension's candidate list above.
static int convT_impl(const void* x, const void* w_hi, const void* w_lo, const long* class_off, void* y,
                      const float* bias, const void* skip, int B, int D0, int D1, int D2, int Cin, int Cout,
                      int act_flags, double* out_sums, int out_groups, void* stream) {
    if (B == 0) return SEMABS_OK;
    const int act_f32 = act_flags & 1;
    const bool bricks = !(act_flags & SEMABS_CONV_GENERIC);
    int rc = conv_common_checks(x, w_hi, w_lo, y, Cin, Cout, act_f32);
    if (rc) return rc;
    SEMABS_REQUIRE(class_off && Cin % 32 == 0, "semabs_convtranspose3d: Cin must be a multiple of 32");
    if (bricks && D0 % 4 == 0 && D1 % 8 == 0 && D2 % 16 == 0) {           // one launch, input read once, all eight classes per brick
        ConvTArgs ta;
        ta.x = x; ta.y = y; ta.w_hi = (const f16*)w_hi; ta.w_lo = (const f16*)w_lo; ta.bias = bias; ta.skip = skip;
        for (int c = 0; c < 8; ++c) ta.class_off[c] = class_off[c];
        ta.B = B; ta.D0 = D0; ta.D1 = D1; ta.D2 = D2; ta.Cin = Cin; ta.Cout = Cout;
        ta.packed_off = (act_flags & SEMABS_CONV_PACKED) ? 27L * Cin * Cout : 0;
        const bool fused = out_sums && out_groups == 8 && Cout % 16 == 0;
        ta.stats = fused ? out_sums : nullptr;
        rc = act_f32 ? convT_brick_launch<true>(ta, (hipStream_t)stream) : convT_brick_launch<false>(ta, (hipStream_t)stream);
        if (rc == SEMABS_OK && out_sums && !fused) rc = semabs_gn_stats(y, out_sums, B, 8L * D0 * D1 * D2, Cout, out_groups, act_f32, stream);
        return rc;
    }
    {   // the small (deep) levels: the eight output parity classes as ONE launch of the gather kernel, blockIdx.z = class (eight launches of
        // 16 - 64 workgroups each were 50 us apiece: 0.48 ms for the 4^3 -> 8^3 layer, whose arithmetic is 4 us)
        ConvArgs a;
        a.x = x; a.y = y; a.w_hi = (const f16*)w_hi; a.w_lo = w_lo ? (const f16*)w_lo : nullptr;
        a.gn_scale = nullptr; a.gn_shift = nullptr; a.bias = bias; a.resid = skip; a.stats = nullptr; a.ablate = 0; a.trace = nullptr; a.wp_hi = nullptr; a.wp_lo = nullptr;
        if (act_flags & SEMABS_CONV_PACKED) { a.wp_hi = a.w_hi + 27L * Cin * Cout; if (a.w_lo) a.wp_lo = a.w_lo + 27L * Cin * Cout; }
        a.B = B; a.I0 = D0; a.I1 = D1; a.I2 = D2; a.O0 = 2 * D0; a.O1 = 2 * D1; a.O2 = 2 * D2;
        a.M0 = D0; a.M1 = D1; a.M2 = D2; a.os = 2; a.is = 1; a.op0 = a.op1 = a.op2 = 0; a.Cin = Cin; a.Cout = Cout; a.relu = 0;
        a.ntaps = 8; a.Kp = 8 * Cin;                        // (per class in the kernel)
        for (int t = 0; t < 28; ++t) { a.td0[t] = 0; a.td1[t] = 0; a.td2[t] = 0; }
        a.ncls = 8;
        for (int c = 0; c < 8; ++c) a.cls_off[c] = class_off[c];
        rc = conv_launch(a, act_f32, (hipStream_t)stream);
        if (rc) return rc;
    }
    if (out_sums) return semabs_gn_stats(y, out_sums, B, 8L * D0 * D1 * D2, Cout, out_groups, act_f32, stream);
    return SEMABS_OK;
}
extern "C" int semabs_convtranspose3d(const void* x, const void* w_hi, const void* w_lo, const long* class_off, void* y,
                                      const float* bias, const void* skip, int B, int D0, int D1, int D2, int Cin, int Cout,
                                      int act_f32, void* stream) {
    return convT_impl(x, w_hi, w_lo, class_off, y, bias, skip, B, D0, D1, D2, Cin, Cout, act_f32, nullptr, 0, stream);
}
// + the GroupNorm statistics of the output (fp64 [B, out_groups, 2], zero-filled by the caller) for the block that follows: fused into the
// brick kernel's epilogue for 8 groups, a statistics pass over y otherwise
extern "C" int semabs_convtranspose3d_stats(const void* x, const void* w_hi, const void* w_lo, const long* class_off, void* y,
                                            const float* bias, const void* skip, int B, int D0, int D1, int D2, int Cin, int Cout,
                                            int act_f32, double* out_sums, int out_groups, void* stream) {
    SEMABS_REQUIRE(out_sums && out_groups > 0 && Cout % out_groups == 0, "semabs_convtranspose3d_stats: bad out_sums / out_groups");
    return convT_impl(x, w_hi, w_lo, class_off, y, bias, skip, B, D0, D1, D2, Cin, Cout, act_f32, out_sums, out_groups, stream);
}

// =================================================================================================
// MaxPool3d(2), channels-last
// =================================================================================================
template <typename T>
__global__ void k_maxpool(const T* __restrict__ x, T* __restrict__ y, int B, int O0, int O1, int O2, int C) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c8 = C / 8;
    const long tot = (long)B * O0 * O1 * O2 * c8;
    if (i >= tot) return;
    const int cc = (int)(i % c8); long v = i / c8;
    const int o2 = (int)(v % O2); v /= O2;
    const int o1 = (int)(v % O1); v /= O1;
    const int o0 = (int)(v % O0); const int b = (int)(v / O0);
    const int I1 = 2 * O1, I2 = 2 * O2;
    float m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        const long idx = ((((long)b * 2 * O0 + 2 * o0 + (d >> 2)) * I1 + 2 * o1 + ((d >> 1) & 1)) * I2 + 2 * o2 + (d & 1)) * C + cc * 8;
        float vv[8];
        load8<sizeof(T) == 4>(x, idx, vv);
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], vv[j]);
    }
    const long oidx = ((((long)b * O0 + o0) * O1 + o1) * O2 + o2) * C + cc * 8;
    if (sizeof(T) == 4) {
        float4* p = reinterpret_cast<float4*>(reinterpret_cast<float*>(y) + oidx);
        p[0] = make_float4(m[0], m[1], m[2], m[3]); p[1] = make_float4(m[4], m[5], m[6], m[7]);
    } else {
        f16x8 h;
#pragma unroll
        for (int j = 0; j < 8; ++j) h[j] = (f16)m[j];
        *reinterpret_cast<f16x8*>(reinterpret_cast<f16*>(y) + oidx) = h;
    }
}
extern "C" int semabs_maxpool3d(const void* x, void* y, int B, int D0, int D1, int D2, int C, int act_f32, void* stream) {
    if (B == 0) return SEMABS_OK;
    SEMABS_REQUIRE(x && y && D0 % 2 == 0 && D1 % 2 == 0 && D2 % 2 == 0 && C % 8 == 0, "semabs_maxpool3d: dims must be even, C % 8 == 0");
    const long tot = (long)B * (D0 / 2) * (D1 / 2) * (D2 / 2) * (C / 8);
    if (act_f32) hipLaunchKernelGGL(k_maxpool<float>, dim3(semabs_cdiv(tot, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, B, D0 / 2, D1 / 2, D2 / 2, C);
    else hipLaunchKernelGGL(k_maxpool<f16>, dim3(semabs_cdiv(tot, 256)), dim3(256), 0, (hipStream_t)stream, (const f16*)x, (f16*)y, B, D0 / 2, D1 / 2, D2 / 2, C);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// =================================================================================================
// Implicit decoder: q = clamp((p + off) * sc, 0, S-1) / S; qn = 2q - 1; trilinear grid_sample(border, align_corners)
// with qn.x -> innermost axis (D2), qn.y -> D1, qn.z -> D0; cat(qn); Linear(19,16) LeakyReLU Linear(16, 1).
// =================================================================================================
struct DecArgs {
    float off[3], sc[3];
    int S0, S1, S2; int concat_xyz;
    float w1[16 * 19], b1[16], w2[16], b2;
    int has_final; float fw[16 * 16], fb[16];      // optional: the UNet's final 1x1x1 convolution applied to the SAMPLED features (it commutes
};                                                 // with the trilinear interpolation), so its output volume never has to be written
// LATTICE: the M queries of a label are a dense C-order lattice G0 x G1 x G2 (e.g. all voxel centres).  Query axis 0 is the point's x, which
// the decoder maps to the volume's INNERMOST axis (net.py:221-239), so in query order consecutive threads walk the volume's slowest axis
// (1 MB apart at 128^3, every lane its own cache lines: 2.4 ms per scene).  Here a workgroup takes a 32 x 2 x 4 tile of the lattice with the
// lane index along query axis 0 - reads are contiguous along the volume's innermost axis - and transposes its 256 results through LDS so
// the output still goes out in query order (16-byte pieces).  Same arithmetic per query, same output layout.
// sampled features f[16] (+ normalised query qn) -> [optional folded final 1x1x1 convolution] -> Linear(19 | 16, 16) LeakyReLU Linear(16, 1)
// (CXYZ = concat_xyz as a template parameter: with a run-time row length the compiler copied the 2.4 KB argument struct to scratch and indexed
//  it there - every weight of every query came out of scratch memory instead of the scalar cache)
template <bool CXYZ>
__device__ __forceinline__ float decoder_mlp(const DecArgs& a, float (&f)[16], const float (&qn)[3]) {
    if (a.has_final) {                                      // f <- W_final . f + b_final  (the corner weights of the in-range corners sum to 1)
        float g[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float t = a.fb[j];
#pragma unroll
            for (int c = 0; c < 16; ++c) t += a.fw[j * 16 + c] * f[c];
            g[j] = t;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) f[j] = g[j];
    }
    float o = a.b2;
    constexpr int din = CXYZ ? 19 : 16;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        float h = a.b1[j];
#pragma unroll
        for (int c = 0; c < 16; ++c) h += a.w1[j * din + c] * f[c];
        if (CXYZ) h += a.w1[j * 19 + 16] * qn[0] + a.w1[j * 19 + 17] * qn[1] + a.w1[j * 19 + 18] * qn[2];
        h = h > 0.f ? h : 0.01f * h;
        o += a.w2[j] * h;
    }
    return o;
}
// Round 3 tried FOUR LANES PER VOXEL READ here (a quad interpolating one query together, lane q loading quarter q of each corner voxel so that a
// quad reads 64 contiguous bytes, DPP all-gather, owner lane runs the MLP; results bit-identical): 1.61 -> 2.44 ms per scene - SLOWER.  The
// thread-per-query gathers below are not what bounds this kernel; what did help is the XCD-aware tile order (FETCH_SIZE 4.41 -> 2.65 GB for the
// 2.15 GB volume, L2 hit 82 %), for 1.65 -> 1.61 ms.
template <typename T, bool LATTICE, bool CXYZ>
__global__ __launch_bounds__(256) void k_decoder(const T* __restrict__ vol, const float* __restrict__ query, DecArgs a, int P, long M,
                                                 long q_stride_p, float* __restrict__ out, int G0, int G1, int G2) {
    __shared__ float s_out[LATTICE ? 256 : 1];
    long i; int b; long m;
    if (LATTICE) {
        const int n2 = G2 / 4, n1 = G1 / 2, n0 = G0 / 32;
        const int per = n0 * n1 * n2;
        const int vb = xcd_remap((int)blockIdx.x, (int)gridDim.x);      // neighbouring query tiles read the same voxels: keep them on one XCD's L2
        b = vb / per;
        int t = vb - b * per;
        const int t2 = t % n2; t /= n2;
        const int t1 = t % n1; const int t0 = t / n1;
        const int i0 = t0 * 32 + (threadIdx.x & 31), i1 = t1 * 2 + ((threadIdx.x >> 5) & 1), i2 = t2 * 4 + (threadIdx.x >> 6);
        m = ((long)i0 * G1 + i1) * G2 + i2;
        i = (long)b * M + m;
    } else {
        i = (long)blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= (long)P * M) return;
        b = (int)(i / M); m = i % M;
    }
    const float* qp = query + (long)b * q_stride_p + m * 3;
    float qn[3];
    const int S[3] = {a.S0, a.S1, a.S2};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float t = (qp[k] + a.off[k]) * a.sc[k];
        t = fminf(fmaxf(t, 0.f), (float)(S[k] - 1));
        t = t / (float)S[k];
        qn[k] = 2.0f * t - 1.0f;
    }
    // unnormalise (align_corners=True) and clip (border): x -> D2, y -> D1, z -> D0
    float ix = ((qn[0] + 1.f) / 2.f) * (float)(a.S2 - 1), iy = ((qn[1] + 1.f) / 2.f) * (float)(a.S1 - 1), iz = ((qn[2] + 1.f) / 2.f) * (float)(a.S0 - 1);
    ix = fminf(fmaxf(ix, 0.f), (float)(a.S2 - 1)); iy = fminf(fmaxf(iy, 0.f), (float)(a.S1 - 1)); iz = fminf(fmaxf(iz, 0.f), (float)(a.S0 - 1));
    const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    const float wx1 = ix - fx, wy1 = iy - fy, wz1 = iz - fz, wx0 = (fx + 1.f) - ix, wy0 = (fy + 1.f) - iy, wz0 = (fz + 1.f) - iz;
    float f[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) f[c] = 0.f;
    const T* vb = vol + (long)b * a.S0 * a.S1 * a.S2 * 16;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        const int zz = z0 + (d >> 2), yy = y0 + ((d >> 1) & 1), xx = x0 + (d & 1);
        if (zz > a.S0 - 1 || yy > a.S1 - 1 || xx > a.S2 - 1) continue;       // out-of-bounds corner contributes 0 (its weight is 0 too)
        const float w = ((d & 1) ? wx1 : wx0) * (((d >> 1) & 1) ? wy1 : wy0) * ((d >> 2) ? wz1 : wz0);
        const long idx = (((long)zz * a.S1 + yy) * a.S2 + xx) * 16;
        float v[8];
        load8<sizeof(T) == 4>(vb, idx, v);
#pragma unroll
        for (int c = 0; c < 8; ++c) f[c] += v[c] * w;
        load8<sizeof(T) == 4>(vb, idx + 8, v);
#pragma unroll
        for (int c = 0; c < 8; ++c) f[8 + c] += v[c] * w;
    }
    const float o = decoder_mlp<CXYZ>(a, f, qn);
    if (LATTICE) {
        s_out[(threadIdx.x & 63) * 4 + (threadIdx.x >> 6)] = o;                 // [(i0, i1) pair][i2]
        __syncthreads();
        if (threadIdx.x < 64) {                                                 // one 16-byte store per (i0, i1): 4 consecutive queries along axis 2
            const float4 v = *reinterpret_cast<const float4*>(s_out + threadIdx.x * 4);
            *reinterpret_cast<float4*>(out + i) = v;                            // for these threads i2 = t2 * 4: i is the first of the four
        }
    } else {
        out[i] = o;
    }
}

// vol [P, S0, S1, S2, 16] (fp16 / fp32); query fp32: label b reads query + b * q_stride_p (0 = shared by all labels),
// M points x 3; out fp32 [P, M].  w1 [16, 19 or 16], b1 [16], w2 [1, 16], b2 [1] host pointers (fp32).
// qgrid3 (host int[3] or NULL): promise that every label's M queries form a dense C-order lattice of these dims (M = product) - only
// changes the order in which the kernel walks them, never the results or their layout.
// final_w [16, 16] / final_b [16] (host, or NULL): `vol` is the UNet's activation BEFORE its final 1x1x1 convolution; the decoder applies that
// convolution to the sampled features instead (a linear map commutes with the interpolation) - saves writing and re-reading one volume.
extern "C" int semabs_decoder(const void* vol, const float* query, const float* off3, const float* sc3, const int* shape3,
                              const float* w1, const float* b1, const float* w2, const float* b2, int concat_xyz, int P, long M,
                              long q_stride_p, int vol_f32, float* out, const int* qgrid3, const float* final_w, const float* final_b,
                              void* stream) {
    if (P == 0 || M == 0) return SEMABS_OK;
    SEMABS_REQUIRE(vol && query && off3 && sc3 && shape3 && w1 && b1 && w2 && b2 && out, "semabs_decoder: null pointer");
    DecArgs a;
    for (int k = 0; k < 3; ++k) { a.off[k] = off3[k]; a.sc[k] = sc3[k]; }
    a.S0 = shape3[0]; a.S1 = shape3[1]; a.S2 = shape3[2]; a.concat_xyz = concat_xyz;
    const int din = concat_xyz ? 19 : 16;
    for (int i = 0; i < 16 * din; ++i) a.w1[i] = w1[i];
    for (int i = 0; i < 16; ++i) { a.b1[i] = b1[i]; a.w2[i] = w2[i]; }
    a.b2 = b2[0];
    a.has_final = final_w != nullptr;
    if (final_w) {
        SEMABS_REQUIRE(final_b, "semabs_decoder: final_w needs final_b");
        for (int i = 0; i < 256; ++i) a.fw[i] = final_w[i];
        for (int i = 0; i < 16; ++i) a.fb[i] = final_b[i];
    }
    const bool lattice = qgrid3 && (long)qgrid3[0] * qgrid3[1] * qgrid3[2] == M && qgrid3[0] % 32 == 0 && qgrid3[1] % 2 == 0 && qgrid3[2] % 4 == 0 &&
                         (long)P * (M / 256) < (1L << 31);
#define DEC_LAUNCH(KERN, GRID)                                                                                                        \
    do {                                                                                                                              \
        if (vol_f32) { if (concat_xyz) hipLaunchKernelGGL((KERN(float, true)), GRID, block, 0, (hipStream_t)stream, (const float*)vol, query, a, P, M, q_stride_p, out, g0, g1, g2); \
                       else hipLaunchKernelGGL((KERN(float, false)), GRID, block, 0, (hipStream_t)stream, (const float*)vol, query, a, P, M, q_stride_p, out, g0, g1, g2); }         \
        else { if (concat_xyz) hipLaunchKernelGGL((KERN(f16, true)), GRID, block, 0, (hipStream_t)stream, (const f16*)vol, query, a, P, M, q_stride_p, out, g0, g1, g2);             \
               else hipLaunchKernelGGL((KERN(f16, false)), GRID, block, 0, (hipStream_t)stream, (const f16*)vol, query, a, P, M, q_stride_p, out, g0, g1, g2); }                     \
    } while (0)
#define DEC_K_LAT(T, C) k_decoder<T, true, C>
#define DEC_K_ANY(T, C) k_decoder<T, false, C>
    const dim3 block(256);
    const int g0 = qgrid3 ? qgrid3[0] : 0, g1 = qgrid3 ? qgrid3[1] : 0, g2 = qgrid3 ? qgrid3[2] : 0;
    if (lattice) {
        const dim3 grid((unsigned)(P * (M / 256)));
        DEC_LAUNCH(DEC_K_LAT, grid);
    } else {
        const dim3 grid(semabs_cdiv((long)P * M, 256));
        DEC_LAUNCH(DEC_K_ANY, grid);
    }
#undef DEC_LAUNCH
#undef DEC_K_LAT
#undef DEC_K_ANY
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}

// =================================================================================================
// VOOL head (SemAbsVOOL.forward, net.py:559-579): per (description d, query point m)
//   f = cat(trilinear(target_vol[d]), trilinear(reference_vol[d]))  (32 ch)   -- the channel concat of net.py:556 is never materialised
//   h = LeakyReLU(W1 [32, 35] . cat(f, qn) + b1);  o = W2 [64, 32] . h + b2   -- spatial_sampler (ImplicitVolumetricDecoder, net.py:215-256)
//   out = cosine_similarity(o, relation_embedding[d]) / temperature               -- PointingAttention.cosine_sim, net.py:300-309
// params (device fp32): w1 [32*35] | b1 [32] | w2 [64*32] | b2 [64];  rel fp32 [P, 64]
// =================================================================================================
template <typename T>
__device__ __forceinline__ void trilerp16(const T* __restrict__ vb, int S0, int S1, int S2, int z0, int y0, int x0, float wz1, float wy1,
                                          float wx1, float wz0, float wy0, float wx0, float* f) {
#pragma unroll
    for (int c = 0; c < 16; ++c) f[c] = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        const int zz = z0 + (d >> 2), yy = y0 + ((d >> 1) & 1), xx = x0 + (d & 1);
        if (zz > S0 - 1 || yy > S1 - 1 || xx > S2 - 1) continue;
        const float w = ((d & 1) ? wx1 : wx0) * (((d >> 1) & 1) ? wy1 : wy0) * ((d >> 2) ? wz1 : wz0);
        const long idx = (((long)zz * S1 + yy) * S2 + xx) * 16;
        float v[8];
        load8<sizeof(T) == 4>(vb, idx, v);
#pragma unroll
        for (int c = 0; c < 8; ++c) f[c] += v[c] * w;
        load8<sizeof(T) == 4>(vb, idx + 8, v);
#pragma unroll
        for (int c = 0; c < 8; ++c) f[8 + c] += v[c] * w;
    }
}

struct VoolArgs { float off[3], sc[3]; int S0, S1, S2; float inv_temp; };

template <typename T>
__global__ __launch_bounds__(256) void k_vool_head(const T* __restrict__ vol_t, const T* __restrict__ vol_r, const float* __restrict__ query,
                                                   const float* __restrict__ prm, const float* __restrict__ rel, VoolArgs a, int P, long M,
                                                   float* __restrict__ out) {
    __shared__ float sw[32 * 35 + 32 + 64 * 32 + 64];
    for (int i = threadIdx.x; i < 32 * 35 + 32 + 64 * 32 + 64; i += 256) sw[i] = prm[i];
    __syncthreads();
    const float* w1 = sw; const float* b1 = sw + 32 * 35; const float* w2 = b1 + 32; const float* b2 = w2 + 64 * 32;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)P * M) return;
    const int d = (int)(i / M);
    const float* qp = query + i * 3;
    float qn[3];
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
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    const float wx1 = ix - fx, wy1 = iy - fy, wz1 = iz - fz, wx0 = (fx + 1.f) - ix, wy0 = (fy + 1.f) - iy, wz0 = (fz + 1.f) - iz;
    float f[35];
    const long vstride = (long)a.S0 * a.S1 * a.S2 * 16;
    trilerp16<T>(vol_t + d * vstride, a.S0, a.S1, a.S2, z0, y0, x0, wz1, wy1, wx1, wz0, wy0, wx0, f);
    trilerp16<T>(vol_r + d * vstride, a.S0, a.S1, a.S2, z0, y0, x0, wz1, wy1, wx1, wz0, wy0, wx0, f + 16);
    f[32] = qn[0]; f[33] = qn[1]; f[34] = qn[2];
    float hcur[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        float hv = b1[j];
#pragma unroll
        for (int c = 0; c < 35; ++c) hv += w1[j * 35 + c] * f[c];
        hcur[j] = hv > 0.f ? hv : 0.01f * hv;
    }
    const float* r = rel + (long)d * 64;
    float dot = 0.f, no = 0.f, nr = 0.f;
#pragma unroll 4
    for (int k = 0; k < 64; ++k) {
        float o = b2[k];
#pragma unroll
        for (int j = 0; j < 32; ++j) o += w2[k * 32 + j] * hcur[j];
        const float rk = r[k];
        dot += o * rk; no += o * o; nr += rk * rk;
    }
    const float eps = 1e-8f;
    out[i] = dot / (fmaxf(sqrtf(no), eps) * fmaxf(sqrtf(nr), eps)) * a.inv_temp;
}

// vol_t / vol_r [P, S0, S1, S2, 16] (fp16 / fp32), query fp32 [P, M, 3], params fp32 [3264] (device), rel fp32 [P, 64] (device),
// off3 / sc3 / shape3 host arrays; out fp32 [P, M].  Built for hidden 2*16 -> 32 -> pointing_dim 64 with xyz concat (utils.py defaults).
extern "C" int semabs_vool_head(const void* vol_t, const void* vol_r, const float* query, const float* params, const float* rel,
                                const float* off3, const float* sc3, const int* shape3, float temperature, int P, long M, int vol_f32,
                                float* out, void* stream) {
    if (P == 0 || M == 0) return SEMABS_OK;
    SEMABS_REQUIRE(vol_t && vol_r && query && params && rel && off3 && sc3 && shape3 && out && temperature > 0.f, "semabs_vool_head: bad args");
    VoolArgs a;
    for (int k = 0; k < 3; ++k) { a.off[k] = off3[k]; a.sc[k] = sc3[k]; }
    a.S0 = shape3[0]; a.S1 = shape3[1]; a.S2 = shape3[2]; a.inv_temp = 1.0f / temperature;
    dim3 grid(semabs_cdiv((long)P * M, 256)), block(256);
    if (vol_f32) hipLaunchKernelGGL(k_vool_head<float>, grid, block, 0, (hipStream_t)stream, (const float*)vol_t, (const float*)vol_r, query, params, rel, a, P, M, out);
    else hipLaunchKernelGGL(k_vool_head<f16>, grid, block, 0, (hipStream_t)stream, (const f16*)vol_t, (const f16*)vol_r, query, params, rel, a, P, M, out);
    SEMABS_CHECK_LAUNCH();
    return SEMABS_OK;
}
