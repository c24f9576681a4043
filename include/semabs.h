/* semabs.h — C ABI of libsemabs_hip.so: the MI355X (gfx950) hot path of real-stanford/semantic-abstraction
 * (multi-scale CLIP relevancy extraction -> depth/TSDF geometry -> OVSSC 3D-UNet voxel inference).
 *
 * The reference is pure Python and has no FFI for this path; these are the entry points a maintainer would bind with
 * ctypes from the reference's own call sites.  Every entry point names the reference code it
 * replaces as file:line relative to the reference repository root.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch types.  Unless a parameter says "host", pointers are DEVICE pointers.
 *   - every function returns 0 on success, SEMABS_EINVAL (-1) for a rejected argument, SEMABS_EHIP (-2) when a HIP
 *     call / launch failed; semabs_last_error() returns the message (thread local).  No exceptions cross the ABI.
 *   - `stream` is a hipStream_t (NULL = default stream).  Calls are asynchronous and stream ordered; the library
 *     never allocates device memory and never synchronises.  All buffers are caller owned, dense and row major.
 *   - empty inputs (zero rows / points / tiles) are accepted and do nothing.
 *   - fp16 = IEEE binary16 (the CLIP weights are fp16-exact: model_explainability.py:501-527).
 */
#ifndef SEMABS_H
#define SEMABS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SEMABS_OK 0
#define SEMABS_EINVAL (-1)
#define SEMABS_EHIP (-2)

const char* semabs_last_error(void);
int semabs_abi_version(void);
/* CU-masked streams (round 6; no reference counterpart - the reference runs one CUDA stream per process, utils.py:131-138): a stream whose kernels are
 * confined to the CUs of `mask` (bit i = CU i in the runtime's enumeration, n_words x 32 bits).  The library's persistent kernels size their grids to the CUs
 * of the stream they are launched on (semabs_stream_cu_count), so two streams with complementary masks share the chip without over-subscribing it. */
int semabs_stream_create_cumask(const unsigned int* mask, int n_words, void** stream);
int semabs_stream_destroy(void* stream);
int semabs_stream_cu_count(void* stream, int* cus);

/* Host-side buffer utilities (no reference counterpart: what torch.zeros / torch.full / Tensor.repeat / masked_fill_ did on the hot path; the
 * product issues no ATen kernel inside a scene).  semabs_fill_u32: nbytes (a multiple of 4) of a 32-bit pattern.  semabs_replicate: dst = reps
 * copies of src (nbytes each, a multiple of 4).  semabs_poison_empty: *n_in == 0 (no point of the depth image inside scene_bounds - an error in the
 * reference, visualize.py:193) -> logits = NaN, labels = -1; nothing otherwise. */
int semabs_fill_u32(void* p, long long nbytes, unsigned int value, void* stream);
int semabs_replicate(const void* src, void* dst, long long nbytes, int reps, void* stream);
/* rows x width_bytes from a pitched source to a pitched destination (all multiples of 4 bytes): pack / unpack of the tile-sharded all-gather (dist.py) */
int semabs_copy2d(const void* src, long long src_pitch_bytes, void* dst, long long dst_pitch_bytes, long long width_bytes, long long rows, void* stream);
int semabs_poison_empty(const long long* n_in, float* logits, long long n_logits, int* labels, long long n_labels, void* stream);
int semabs_device_info(char* name /*host*/, int name_len, int* cu_count /*host*/, long long* hbm_bytes /*host*/);

/* ============================ geometry (csrc/geometry.hip) =============================================== */

/* get_pointcloud + transform_pointcloud + filter_pts_bounds          point_cloud.py:34-66, 8-21, 24-31
 * depth fp32 [H, W]; params f64 [22] = {fx, fy, cx, cy, pose 3x4 row major (12), bounds lo[3], hi[3]};
 * xyz fp32 [H*W, 3] = float32(f64 math); mask uint8 [H*W] (NULL = skip) = inclusive AABB test of the fp32 point. */
int semabs_pointcloud(const float* depth, int H, int W, const double* params, int has_pose, float* xyz,
                      unsigned char* mask, void* stream);
/* same, returning the f64 points themselves (the reference's return dtype; point_cloud.py:51-66) */
int semabs_pointcloud_f64(const float* depth, int H, int W, const double* params_dev, int has_pose, double* xyz64, void* stream);

/* VirtualGrid.get_points_grid_idxs + flatten_idxs                    net.py:84-133
 * idx = clamp(trunc((p + off) * scale), 0, S-1) with two separately rounded fp32 ops (bit-exact integer output);
 * off3 / scale3 / shape3 are HOST arrays; flat int64 [N] and / or idx3 int32 [N, 3] (either may be NULL). */
int semabs_voxel_index(const float* pts, long N, const float* off3, const float* scale3, const int* shape3,
                       long long* flat, int* idx3, void* stream);

/* TSDFVolume.vox2world + rigid_transform + cam2pix + integrate_tsdf + integrate      fusion.py:85-195
 * color uint8 [H, W, 3] (NULL = TSDF only), depth fp32 [H, W]; params f64 [15] = {inv(pose) 3x4, voxel_size,
 * trunc_margin, obs_weight}; origin3 / intr4 {fx, fy, cx, cy} / dims3 are HOST arrays (fp32 as the reference casts
 * them); tsdf / weight / colvol fp32 [D0, D1, D2] updated in place; pix_out int64 [D0*D1*D2, 2] optional. */
int semabs_tsdf_integrate(const unsigned char* color, const float* depth, int H, int W, const double* params,
                          const float* origin3, const float* intr4, const int* dims3, float* tsdf, float* weight,
                          float* colvol, long long* pix_out, void* stream);

/* check_pts_in_frustum                                               point_cloud.py:88-110
 * pts f64 [M, 3]; params f64 [16] = {inv(pose) 3x4, fx, fy, cx, cy}; mask uint8 [M]. */
int semabs_frustum_mask(const double* pts, long M, const double* params, int H, int W, unsigned char* mask, void* stream);

/* in-bounds compaction + seeded sub-sample on the device (visualize.py:103-108 boolean indexing, :193 np.random.choice with replacement):
 * pix[0..n_in) = ascending i with mask[i] != 0; sel[j] = pix[mulhi64(splitmix64(seed * 0x9E3779B97F4A7C15 + j), n_in)]; n_in int64 [1] device.
 * mask 16-byte aligned; num >= max(1, ceil(n / 4096) / 2): sel doubles as the scratch of the per-block counts between the launches */
int semabs_compact_subsample(const unsigned char* mask, long n, unsigned long long seed, long num, long long* pix, long long* n_in,
                             long long* sel, void* stream);

/* relevancy -> per-point features                                    visualize.py:93-122 (x50, -mean over labels, gather)
 * rel fp32 [L, HW]; sel int64 [n] pixel ids; xyz fp32 [HW, 3]; feat fp32 [L, n]; xyz_out fp32 [n, 3] (optional). */
int semabs_gather_point_features(const float* rel, const long long* sel, const float* xyz, int L, long HW, long n,
                                 float mult, int subtract_mean, float* feat, float* xyz_out, void* stream);

/* OVSSC post-mask: argmax, cutoff, frustum, tsdf > 0                 visualize.py:228-247
 * logits fp32 [L, M]; in_frustum uint8 [M] / tsdf fp32 [M] optional; label int32 [M] (-1 = empty). */
int semabs_ovssc_labels(const float* logits, const unsigned char* in_frustum, const float* tsdf, int L, long M,
                        float cutoff, int* label, void* stream);

/* ============================ tiling front / back (csrc/tiles.hip) ====================================== */

/* Pillow ImagingResample coefficient tables for BICUBIC (third-party algorithm behind torchvision Resize,
 * clip_explainability.py:98-108).  HOST function, HOST pointers: xmin int32 [out], kk int32 [out, kmax]. */
int semabs_resize_coeffs(int in_size, int out_size, int* xmin, int* kk, int kmax, int* ksize);

/* ClipWrapper.create_tiles crop + _transform + flip + VisionTransformer.conv1 im2col
 *                                          CLIP/clip/__init__.py:254-281, 171-173; model_explainability.py:325-328
 * images uint8 [n_img, H, W, 3]; tiles int32 [n_tiles, 5] = (image, row0, col0, tile_size, coef_id);
 * coef_xmin int32 [n_sizes, 224], coef_kk int32 [n_sizes, 224, 24], coef_ksize int32 [n_sizes]; lut fp16 [3, 256]
 * = fp16((u/255 - mean_c)/std_c); patches fp16 [n_tiles * g*g, 3*p*p] (g = 224/p), column = c*p*p + iy*p + ix.
 * flip 0 = as is, 1 = mirrored (the flip pass, __init__.py:171-173), 2 = both from one resampling: patches fp16 [2, n_tiles * g*g, 3*p*p]. */
int semabs_tile_patches(const unsigned char* images, int n_img, int H, int W, const int* tiles, int n_tiles,
                        const int* coef_xmin, const int* coef_kk, const int* coef_ksize, const void* lut, void* patches,
                        int patch, int flip, int max_ksize, void* stream);

/* same im2col for already preprocessed fp32 tiles [n, 3, 224, 224] (ClipGradcam.forward(x=...), clip_gradcam.py:58-62) */
int semabs_patchify(const float* x, void* patches, int n, int patch, int flip, void* stream);

/* un-flip average, bilinear upsample, fp16 canvases, count normalise, mean over scales   CLIP/clip/__init__.py:196-236
 * rel / rel_flip fp32 [L, N, g, g] (rel_flip NULL = no flip pass); scales int32 [n_scales, 5] =
 * (tile_size, stride, n_row_starts, n_col_starts, first tile index within an image); out fp32 [L, H, W]. */
int semabs_aggregate(const float* rel, const float* rel_flip, int L, int N, int g, int H, int W, const int* scales,
                     int n_scales, int n_img, int tiles_per_img, float* out, void* stream);

/* the un-flip average alone (CLIP/clip/__init__.py:196-204): out[m, h, w] = (rel[m, h, w] + rel_flip[m, h, g - 1 - w]) / 2 for n_maps = L * N maps
 * [g, g]; aggregating `out` with rel_flip = NULL equals semabs_aggregate(rel, rel_flip) bit for bit at half the loads per covering tile. */
int semabs_unflip_average(const float* rel, const float* rel_flip, float* out, long n_maps, int g, void* stream);
/* The same for L label rows whose input maps are in_label_stride maps apart (both passes in one [L, 2 N, g, g] buffer); out [L, maps_per_label, g, g]. */
int semabs_unflip_average_rows(const float* rel, const float* rel_flip, float* out, int L, long maps_per_label, long in_label_stride, int g, void* stream);

/* ColorJitter(0.6, 0.6, 0.6, 0.1) of the augmentation copies             CLIP/clip/__init__.py:55-57, 246-247
 * (torchvision 0.13.1 ColorJitter.forward on the PIL image: functional_pil.adjust_brightness / _contrast / _saturation / _hue = Pillow
 *  ImageEnhance blends and the uint8 HSV hue rotation; byte-exact against Pillow, tests/golden/g28_color_jitter.npz)
 * img uint8 [H, W, 3] in place; order4 (a permutation of the op ids) / factors4 (indexed by op id) HOST arrays;
 * op: 0 brightness 1 contrast 2 saturation 3 hue (hue factor in [-0.5, 0.5]); scratch8: 8 bytes of device memory. */
int semabs_color_jitter(unsigned char* img, int H, int W, const int* order4, const float* factors4, void* scratch8, void* stream);
/* One adjustment of the above, in place (torchvision functional_pil.adjust_*). */
int semabs_color_jitter_op(unsigned char* img, int H, int W, int op, float factor, void* scratch8, void* stream);

/* ============================ dense contractions (csrc/gemm.hip) ======================================== */

/* C = epi(A[M,K] . B[N,K]^T + bias[N]) on fp16 MFMA, fp32 accumulate.   every F.linear / conv1 / proj on the path:
 * auxiliary.py:129,340; model_explainability.py:210-217,253-254,325,353; clip_gradcam.py:90-97 (VJP chain)
 * epi: 0 fp16 = acc+bias | 1 fp16 = quickgelu(acc+bias) | 2 fp32 += acc+bias | 3 fp32 = acc+bias |
 *      4 fp32 row-remapped: out row = (m / g_in) * g_out + g_off + m % g_in, plus addend[(g_off + m % g_in), :]
 *      5 fp16 = (acc+bias) * addend[m % g_in, :]: addend = fp32 table [g_in, N] - with semabs_quickgelu_grad's table the QuickGELU VJP of the multi-layer
 *        rollout (clip_gradcam.py:90-97 differentiating model_explainability.py:199-201); M >= 2048, N % 256 == 0, K >= 128 (phased kernel only)
 * rowmap3 HOST {g_in, g_out, g_off}; N % 128 == 0, K % 64 == 0, lda/ldb % 8 == 0, ldc % 4 == 0. */
int semabs_gemm_f16(const void* A, const void* B, void* C, const float* bias, const float* addend, long M, int N, int K,
                    long lda, int ldb, long ldc, int epi, const int* rowmap3, void* stream);

/* Same contraction with per-call launch options (the library keeps no process-global launch state):
 * kernel: 0 = heuristic, 1 = the 128x128 ring kernel, 2 = the 256x256 phased kernel (needs N % 256 == 0, K >= 128); | 256 = every XCD walks its run of
 *   tiles backwards (zigzag with the producer of A); | 512 = the phased kernel's K = 32 ring schedule (A/B, slower: DESIGN section 11);
 * start_event / stop_event: optional hipEvent_t pair filled by the launch's own dispatch packet (both or neither) - how bench.py times
 * the GEMM launches of the timed region without inserting barrier packets around them. */
int semabs_gemm_f16_ex(const void* A, const void* B, void* C, const float* bias, const float* addend, long M, int N, int K,
                       long lda, int ldb, long ldc, int epi, const int* rowmap3, int kernel, void* start_event, void* stop_event,
                       void* stream);

/* LayerNorm folded into the GEMMs on either side of it (auxiliary.py:129,340 / model_explainability.py:232-255: ln_1 -> in_proj, ln_2 -> c_fc): large
 * shapes (M >= 2048, N % 256 == 0, K >= 128).  epi 2 with (ln_xg, ln_gamma, ln_part): x += A W^T + b AND xg = fp16(x_new * gamma) [M, N], row partials
 * (sum, sum of squares) [M, N / 256, 2]; epi 0 / 1 with (ln_rowac, ln_colsum): C = rstd_row * (A W^T) - mean_row * rstd_row * colsum[n] + bias[n]
 * (A = such an xg, K % 128 == 0; colsum[n] = sum_k gamma_k W[n, k]; bias must already contain sum_k beta_k W[n, k]).  reverse: tile order (zigzag).
 * epi 0 with lo_out (precision = "parity"): additionally lo_out[m, n] = fp16(v - fp16(v)) for n < lo_cols (lo_cols % 256 == 0, row pitch ld_lo): the low
 * halves of q | k, consumed by semabs_attention_split.  semabs_ln_rowstats: partials -> (rstd, -mean * rstd) [M, 2].
 * ln_center (producer, optional, [M]): a per-row centre c - the mean the previous LayerNorm of the row saw - subtracted before the fp16 copy and the
 * partial sums: xg = fp16((x_new - c) * gamma), partials of x_new - c.  semabs_ln_rowstats(center_in, center_out) then yields (rstd, -mean(x - c) * rstd),
 * which makes the consumer's formula exact again, and center_out = c + mean(x - c), the next producer's centre (center_out may alias center_in). */
int semabs_gemm_f16_ln(const void* A, const void* B, void* C, const float* bias, long M, int N, int K, long lda, int ldb, long ldc, int epi,
                       void* ln_xg, const float* ln_gamma, float* ln_part, const float* ln_center, const float* ln_rowac, const float* ln_colsum,
                       void* lo_out, int lo_cols, long ld_lo, int reverse,
                       void* start_event, void* stop_event, void* stream);
int semabs_ln_rowstats(const float* part, long M, int ntile, int D, float eps, float* rowac, const float* center_in, float* center_out, void* stream);

/* ============================ transformer pieces (csrc/vit.hip) ========================================= */

/* LayerNorm (fp32 statistics)                                        model_explainability.py:188-194
 * mean_out (optional, [M]): the row means, the first centre of a folded-LayerNorm chain (semabs_gemm_f16_ln). */
int semabs_layernorm(const float* x, const float* gamma, const float* beta, void* out, long M, int D, float eps,
                     int out_f32, long ld_in, float* mean_out, void* stream);
/* Two LayerNorms in one pass: out1 fp32 = LN(x; g1, b1) (may alias x), out2 fp16 = LN(out1; g2, b2); bit-identical to two semabs_layernorm calls.  ln_pre
 * followed by block 0's ln_1 (model_explainability.py:345, 232-255).  order as in semabs_layernorm (0 / 1 / 2); mean_out (optional): row means of out1. */
int semabs_layernorm2(const float* x, const float* g1, const float* b1, float* out1, const float* g2, const float* b2, void* out2, long M, int D, float eps,
                      int order, float* mean_out, void* stream);
/* Residual add fused into the LayerNorm pass: x fp32 [M, D] += delta fp16 [M, D] (in place), out fp16 [M, D] = LayerNorm(x); out NULL = the
 * addition only.  ResidualAttentionBlock.forward's `x = x + ...; ln_2(x)` (model_explainability.py:232-255) with the read-modify-write of
 * the residual stream taken out of the GEMM epilogue. */
int semabs_add_layernorm(float* x, const void* delta, const float* gamma, const float* beta, void* out, long M, int D, float eps,
                         void* stream);
/* class token rows: x[n, 0, :] = class_embedding + pos[0, :]          model_explainability.py:329-343 */
int semabs_embed_finish(float* x, const float* cls, const float* pos, int n, int T, int D, void* stream);
/* fused multi-head attention, head_dim 64, T <= 288                  auxiliary.py:260-340 (q pre-scaled)
 * row_stats (optional, NULL = none) fp32 [n_seq, H, T, 2]: every query's softmax normalisation (reference maximum, 1 / sum), i.e.
 * attn_probs[q, k] = exp(S[q, k] - row_stats[.., 0]) * row_stats[.., 1] - kept for semabs_attention_bwd */
int semabs_attention(const void* qkv, void* out, void* row_stats, int n_seq, int T, int H, int head_dim, int ld,
                     int causal, void* stream);
/* last block, CLS query only; keeps the softmax row (the hooked attn_probs, auxiliary.py:330-335).  q fp32 [n, D], k fp32 [n, T, D] (the
 * scores stay fp32), v fp16 [n, T, D] (it only enters averaged: o, and the rollout's V . u dots) */
/* semabs_attention with q and k as fp16 hi + lo pairs (precision = "parity"): qk_lo fp16 [n_seq, T, ld_lo], q_lo | k_lo at column offsets 0 / D;
 * scores = q_hi k_hi + q_hi k_lo + q_lo k_hi in fp32                                                      CLIP/clip/auxiliary.py:307-337 */
int semabs_attention_split(const void* qkv, const void* qk_lo, void* out, void* row_stats, int n_seq, int T, int H, int head_dim, int ld, int ld_lo,
                           int causal, void* stream);
/* "split" fp16 outputs (semabs_attention_cls / _quickgelu / _logit_grad / _ln_bwd / _gelu_bwd: `split` != 0; semabs_layernorm: out_f32 bit 3): a row
 * of W values is stored as [hi | lo] with a pitch of 2 W - hi = fp16(v), lo = fp16(v - hi) - and consumed by a GEMM with K = 2 W against the weight
 * matrix repeated twice along K: the operand enters to ~2^-22.  The last block and the VJP chain behind it run this way (clip/vit.py). */
int semabs_attention_cls(const float* q, const float* k, const void* v, float* probs, void* o, int n, int T, int H, int head_dim, int split, void* stream);
/* The kept softmax row of the last block without the K projection: s[h, j] = (W_k,h^T q_h) . x_j + q_h . b_k,h for the CLS query only (the one row
 * clip_gradcam.py:124-131 reads; auxiliary.py:307-337).  q fp32 [n, D] (scaled, bias included), wk fp16 [D, D] = in_proj_weight[D:2D], bk fp32 [D], x = the block's
 * LayerNorm-1 rows, fp16 of pitch ldx ([hi | lo] when split) -> probs fp32 [n, H, T]; then semabs_attention_cls(q = k = NULL) forms o = P . V from them. */
int semabs_cls_scores(const float* q, const void* wk, const float* bk, const void* x, long ldx, int split, float* probs, int n, int T, int D, void* stream);
int semabs_rows_gather(const float* src, float* dst, long rows, int cols, long src_stride, long offset, void* stream);
/* x[arange(B), tokens.argmax(-1)] of the text tower (the EOT rows)        CLIP/clip/model_explainability.py:480
 * tokens int64 [B, T] (device), x fp32 [B * T, D] -> dst fp32 [B, D]. */
int semabs_eot_rows_gather(const long long* tokens, const float* x, float* dst, int B, int T, int D, void* stream);
int semabs_quickgelu(const float* fc, void* act, long n, int split_w, void* stream);    /* model_explainability.py:197-199; split_w > 0: rows of split_w values, split output */
/* gd = d quickgelu(x) / dx on fp32 pre-activations (fp32, n % 4 == 0): the derivative table of semabs_gemm_f16 epi 5 */
int semabs_quickgelu_grad(const float* fc, float* gd, long n, void* stream);
/* logits = 100 f/|f| . w_l and d logit / d f, rows normalised to max-abs 1            clip_gradcam.py:62-67 */
int semabs_logit_grad(const float* feat, const float* w_text, int n, int L, int E, float* logits, void* dfeat, float* scale, int split, void* stream);
int semabs_ln_bwd(const float* x, const float* gamma, const float* gy, const float* resid, float* out32, void* out16,
                  long M, int D, int n_x, long ld_x, float eps, int split, void* stream);
int semabs_gelu_bwd(const float* dact, const float* fc, void* dfc, long M, int W, int n_x, int split, void* stream);
/* closed form of ClipGradcam.interpret for the only contributing block             clip_gradcam.py:70-132 */
int semabs_rollout(const float* probs, const void* v /* fp16 [n, T, D] */, const float* u, const float* scale, float* rel, int n, int T, int H,
                   int L, int positive_only, long n_total, long tile0, void* stream);
/* ---- multi-layer rollout for deep towers (csrc/vitl.hip; ViT-L/14: clip_gradcam.py:51-56, 85-126) -------------------------------------
 * Attention backward of one block for R = L * n sequences ordered (label, tile) + that block's rollout update (two MFMA kernels).
 * qkv fp16 [n, T, 3 D] of the block (q pre-scaled | k | v), att fp16 [n, T, D] = its attention output and fwd_stats fp32 [n, H, T, 2] =
 * semabs_attention's row_stats of the forward pass, dO fp16 [R, T, D] = gradient wrt the attention output (before out_proj),
 * rvec fp32 [R, T] = rollout row before this block, gscale fp32 [R] (true gradient = stored * gscale),
 * c fp32 [R, T] += (1 / H) sum_q rvec[q] act(P dP gscale)  (act = clamp(min 0) iff positive_only), stats fp32 [R, H, T, 4] scratch,
 * dqkv fp16 [R, T, 3 D] <- (dQ | dK | dV), or NULL for the rollout update only.  head_dim 64, T <= 288. */
int semabs_attention_bwd(const void* qkv, const void* att, const float* fwd_stats, const void* dO, const float* rvec, const float* gscale,
                         float* c, float* stats, void* dqkv, int n, int L, int T, int H, int head_dim, int positive_only, void* stream);
/* per sequence r: g32[r, :len] *= 2^k (max |g| -> [0.5, 1)), g16 = fp16(g32) (optional), gscale[r] /= 2^k */
int semabs_seq_rescale(float* g32, void* g16, float* gscale, long R, long len, void* stream);
/* one rollout layer: r += c, c = 0                                                    clip_gradcam.py:121-124 (row 0 of R only) */
int semabs_rollout_step(float* r, float* c, long n, void* stream);
/* text tower glue                                                    model_explainability.py:469-482; clip_gradcam.py:24-27 */
int semabs_gather_text(const long long* tokens, const float* emb, const float* pos, float* x, int B, int T, int D, void* stream);
int semabs_text_finish(const float* e, float* w, int C, int P, int E, void* stream);

/* ============================ OVSSC voxel inference (csrc/unet.hip) ===================================== */

/* SemAbs3D.pts_feat_extractor: (xyz | feat) 4 -> 128 -> 128 -> 16, LeakyReLU(0.01)   net.py:358-367, 395-404 */
int semabs_point_mlp(const float* xyz, const float* feat, const float* w1, const float* b1, const float* w2, const float* b2,
                     const float* w3, const float* b3, float* out, int P, long N, int hidden, int cout, void* stream);
/* the same MLP on fp32 FMAs instead of the matrix pipe: the cross-check of the MFMA kernel (tests) */
int semabs_point_mlp_fma(const float* xyz, const float* feat, const float* w1, const float* b1, const float* w2, const float* b2,
                         const float* w3, const float* b3, float* out, int P, long N, int hidden, int cout, void* stream);
/* VirtualGrid.scatter_points, reduce = MEAN (the reference ignores reduce_method)     net.py:185-201
 * deterministic, sums in point order; vol zero-filled and head filled with -1 by the caller. */
int semabs_scatter_mean(const long long* flat, const float* feat, int* head, int* next, void* vol, int P, long N, int C,
                        long nvox, int vol_f32, void* stream);
/* Same, plus the GroupNorm statistics (8 groups, fp64 [P, 8, 2], zero-filled by the caller) of the scattered volume for the first UNet block
 * (unet3d.py:66-79): summed over the occupied voxels while they are written.  C must be 16. */
int semabs_scatter_mean_stats(const long long* flat, const float* feat, int* head, int* next, void* vol, int P, long N, int C,
                              long nvox, int vol_f32, double* out_sums, void* stream);
/* Sparse form (round 6): vol is NOT zero-filled; occ uint32 [ceil(nvox / 32)] (zero-filled by the caller) receives the occupancy bitmap shared by the P volumes
 * (net.py:185-201 scatters the same points for every label); only occupied voxels are written.  Consumed by semabs_conv3d_sparse_stats, which treats every other
 * voxel as zero without reading it (2.1 GB of fill and 2.1 GB of loads per 16 x 128^3 scene). */
int semabs_scatter_mean_sparse(const long long* flat, const float* feat, int* head, int* next, void* vol, unsigned int* occ, int P, long N, int C, long nvox,
                               int vol_f32, double* out_sums, void* stream);
/* GroupNorm statistics / affine                                       unet3d.py:66-79 (nn.GroupNorm) */
int semabs_gn_stats(const void* x, double* sums, int B, long nvox, int C, int G, int x_f32, void* stream);
int semabs_gn_finalize(const double* sums, const float* gamma, const float* beta, float* scale, float* shift, int B, int C,
                       int G, long nvox, float eps, void* stream);
/* [GroupNorm ->] Conv3d k in {1, 3} pad k/2 [+ bias] [+ residual] [-> ReLU], channels-last   unet3d.py:16-17, 98-128, 247-259, 579
 * act_f32 (here and in the ConvTranspose3d entry points) is a flag word: bit 0 = fp32 activations ("exact" mode), bit 8 = run the generic
 * gather kernel even where an LDS-brick kernel exists (per-call cross-check for tests; results agree to rounding), bit 9 = the w_hi / w_lo
 * buffers carry, behind the [Cout, Kp] matrix (behind all eight class matrices for the transposed convolution, each packed on its own), a
 * fragment-packed copy [Kp / 32][Cout / 16][64 lanes = kg * 16 + row][8] fp16 with k = 32 * k-step + 8 * kg + e, which the kernels then read
 * instead (one MFMA A operand = 1 KB of contiguous memory; same values, same results). */
int semabs_conv3d(const void* x, const void* w_hi, const void* w_lo, void* y, const float* gn_scale, const float* gn_shift,
                  const float* bias, const void* resid, int B, int D0, int D1, int D2, int Cin, int Cout, int ksize, int relu,
                  int act_f32, void* stream);
/* Same convolution, and the GroupNorm statistics of its OUTPUT for the next layer: out_sums fp64 [B, out_groups, 2] (sum, sum of squares
 * per group; zero-filled by the caller) -> semabs_gn_finalize.  Fused into the epilogue at the 128^3 x 16-channel level (saves one full
 * read of the tensor per GroupNorm, unet3d.py:66-79), a statistics pass over y elsewhere. */
int semabs_conv3d_stats(const void* x, const void* w_hi, const void* w_lo, void* y, const float* gn_scale, const float* gn_shift,
                        const float* bias, const void* resid, int B, int D0, int D1, int D2, int Cin, int Cout, int ksize, int relu,
                        int act_f32, double* out_sums, int out_groups, void* stream);
/* semabs_conv3d_stats on a sparse input volume (see semabs_scatter_mean_sparse): 16 -> 16 channels, 3 x 3 x 3, D0 % 8 == D1 % 8 == D2 % 16 == 0 */
int semabs_conv3d_sparse_stats(const void* x, const unsigned int* occ, const void* w_hi, const void* w_lo, void* y, const float* gn_scale, const float* gn_shift,
                               const float* bias, const void* resid, int B, int D0, int D1, int D2, int Cin, int Cout, int ksize, int relu, int act_f32,
                               double* out_sums, int out_groups, void* stream);
/* ConvTranspose3d k3 s2 p1 (output_size = skip size) + bias + sum joining          unet3d.py:428-440, 385-396 */
int semabs_convtranspose3d(const void* x, const void* w_hi, const void* w_lo, const long* class_off /*host*/, void* y,
                           const float* bias, const void* skip, int B, int D0, int D1, int D2, int Cin, int Cout, int act_f32,
                           void* stream);
/* Same, plus the GroupNorm statistics of the output (sum / sum of squares per group, fp64 [B, out_groups, 2], zero-filled by the caller) for
 * the decoder block that follows (unet3d.py:66-79): fused into the brick kernel's epilogue for 8 groups, a statistics pass otherwise. */
int semabs_convtranspose3d_stats(const void* x, const void* w_hi, const void* w_lo, const long* class_off, void* y, const float* bias,
                                 const void* skip, int B, int D0, int D1, int D2, int Cin, int Cout, int act_f32, double* out_sums,
                                 int out_groups, void* stream);
int semabs_maxpool3d(const void* x, void* y, int B, int D0, int D1, int D2, int C, int act_f32, void* stream);   /* unet3d.py:298 */
/* ImplicitVolumetricDecoder: trilinear grid_sample (border, align_corners) + MLP    net.py:215-256
 * off3 / sc3 / shape3 / w1 / b1 / w2 / b2 are HOST arrays.  qgrid3 (host int[3] or NULL): the M queries of every label are a dense C-order
 * lattice of these dims (e.g. all voxel centres) - lets the kernel walk them in the volume's memory order; results and layout unchanged.
 * final_w [16,16] / final_b [16] (host or NULL): vol is the activation in front of the UNet's final 1x1x1 conv (unet3d.py:584-586), which is
 * then applied to the sampled features (it commutes with the interpolation). */
int semabs_decoder(const void* vol, const float* query, const float* off3, const float* sc3, const int* shape3, const float* w1,
                   const float* b1, const float* w2, const float* b2, int concat_xyz, int P, long M, long q_stride_p, int vol_f32,
                   float* out, const int* qgrid3, const float* final_w, const float* final_b, void* stream);

/* SemAbsVOOL head: sample two 16-ch volumes (no concat), spatial sampler 35 -> 32 -> 64, cosine similarity with the
 * relation embedding / temperature                                    net.py:559-579, 215-256, 300-309
 * params fp32 [32*35 | 32 | 64*32 | 64] and rel fp32 [P, 64] on the device; off3 / sc3 / shape3 host. */
int semabs_vool_head(const void* vol_t, const void* vol_r, const float* query, const float* params, const float* rel,
                     const float* off3, const float* sc3, const int* shape3, float temperature, int P, long M, int vol_f32,
                     float* out, void* stream);

/* ============================ optimizer (csrc/optim.hip) ================================================= */

/* Lamb.step over all parameter tensors at once                        arm/optim/lamb.py:59-127
 * chunks int64 [n_chunks, 3] = (tensor, offset, count); ptrs int64 [4, n_tensors] = device pointers to fp32 w, g, m, v;
 * norms fp64 [n_tensors, 2] scratch; stats fp32 [n_tensors, 3] = weight_norm, adam_norm, trust_ratio (optional). */
int semabs_lamb_step(const long long* chunks, int n_chunks, const long long* ptrs, int n_tensors, double lr, double beta1,
                     double beta2, double eps, double weight_decay, int adam, double* norms, float* stats, void* stream);

/* ============================ training step (csrc/train.hip, csrc/unet.hip) ============================== */
/* What torch.autograd + train_vool.py / utils.loop do for one optimisation step of SemAbsVOOL (config 5):
 *   forward graph            net.py:506-579 (SemAbsVOOL), 383-439 (SemAbs3D), unet3d.py:190-259, 596-621
 *   loss                     train_vool.py:171-178 (binary_cross_entropy_with_logits, weight = utils.get_bce_weight utils.py:727-749)
 *   backward + clip + step   utils.py:404-417 (loss.backward, clip_grad_norm_, Lamb.step)
 * All buffers fp32 on the device unless noted; channels-last volumes [B, D0, D1, D2, C]. */

/* General gather convolution y[b, m] = sum_taps W_tap . x[b, m * in_stride + td_tap] (zero padding): the data gradient of Conv3d
 * (flipped taps, stride 1) and of ConvTranspose3d (stride 2).  w_hi / w_lo fp16 [Cout, Kp], k = tap * Cin + cin; taps int8 [ntaps, 3] host;
 * in_scale / in_shift fp32 [B, Cin]: optional per-(b, c) input affine (the dynamic gradient scale of semabs_grad_scale). */
int semabs_conv3d_gather(const void* x, const void* w_hi, const void* w_lo, void* y, const float* in_scale, const float* in_shift, int B,
                         int I0, int I1, int I2, int M0, int M1, int M2, int in_stride, int Cin, int Cout, int ntaps, const signed char* taps,
                         int act_f32, void* stream);

/* Weight gradient dW[ca][tap * Cx + cx] += sum_rows A[row][ca] * GN(X)[neighbour(row, tap)][cx]; rows run over [B, M0, M1, M2], the
 * neighbour is m * in_stride + td with zero padding.  Conv3d: A = dY, X = layer input.  ConvTranspose3d: A = input, X = dY, stride 2.
 * Linear: M0 = M1 = 1, M2 = rows, one tap.  dW is accumulated into; tap_minor = 1 writes torch's parameter layout [Ca, Cx, ntaps] instead of
 * [Ca, ntaps, Cx].  Ca % 16 == 0, Cx % 4 == 0. */
int semabs_wgrad(const float* A, const float* X, const float* gn_scale, const float* gn_shift, float* dW, int B, int M0, int M1, int M2,
                 int I0, int I1, int I2, int in_stride, int Ca, int Cx, int ntaps, const signed char* taps, int tap_minor, void* stream);

/* semabs_wgrad on the matrix cores (split fp16 operands = fp32-like accuracy, transposing LDS reads) for the shapes semabs_wgrad_conv3 does not
 * take: the 8^3 / 4^3 levels, ConvTranspose3d (stride 2), Linear.  Additionally Cx % 16 == 0 and fewer than 2^31 - 64 rows.  sA2 / sX2 =
 * (s, 1 / s) device scalars of semabs_grad_scale for the operand that is a gradient, or NULL.  scratch (scratch_floats fp32) or NULL: room for
 * the row chunks' partial sums - with it they are combined by a second deterministic pass, without it by fp32 atomics.  Replaces the weight-gradient half of
 * loss.backward() for these layers (reference: train_vool.py / utils.py loop -> torch autograd of unet3d.py:63-118, 296-331). */
int semabs_wgrad_mfma(const float* A, const float* X, const float* gn_scale, const float* gn_shift, const float* sA2, const float* sX2, float* dW,
                      int B, int M0, int M1, int M2, int I0, int I1, int I2, int in_stride, int Ca, int Cx, int ntaps, const signed char* taps,
                      int tap_minor, float* scratch, long scratch_floats, void* stream);

/* The same for Conv3d 3x3x3 on MFMA (split fp16, bricks of the volume staged in LDS): dW[ca][tap * Cx + cx] += sum_vox dZ[vox][ca] * GN(X)[vox + tap][cx].
 * s2 = (s, 1 / s) of semabs_grad_scale(dZ) or NULL.  Needs D0 % 4 == 0, D1 % 4 == 0, D2 % 16 == 0, Ca % 16 == 0, Cx % 16 == 0.
 * scratch (scratch_floats fp32, >= (Ca / 16) (Cx / 16) 6912): partial sums of the transposing-read kernel, reduced deterministically;
 * scratch = NULL: the round-2 kernel (4 x 8 x 16 bricks transposed while staging, D1 % 8 == 0, fp32 atomics). */
int semabs_wgrad_conv3(const float* dZ, const float* X, const float* gn_scale, const float* gn_shift, const float* s2, float* dW, int B, int D0,
                       int D1, int D2, int Ca, int Cx, int tap_minor, float* scratch, long scratch_floats, void* stream);
/* Which kernel semabs_wgrad_conv3 runs for a shape with scratch_floats of scratch (0 = no scratch): *kernel = 2 transposing-read kernel, 1 brick
 * kernel (D1 % 8 == 0), 0 unsupported (use semabs_wgrad_mfma / semabs_wgrad) - the entry point's own predicate, for host-side routing. */
int semabs_wgrad_conv3_supported(int D0, int D1, int D2, int Ca, int Cx, long scratch_floats, int* kernel);
/* Weight gradient AND the GroupNorm-backward reductions of a GroupNorm -> Conv3d 3x3x3 layer from ONE pass over (dZ, X): X [B, D0, D1, D2, Cx] is the
 * layer's GroupNorm INPUT with its statistics mean / rstd [B, G] and affine gamma / beta [Cx], W [Ca, Cx, 27] the layer's weights.
 *   dW [Ca, Cx, 27] += d loss / d W;   red fp64 [B, Cx, 2] = (sum_v dXn, sum_v dXn * xhat) with dXn = conv^T(dZ), in dZ's dynamic scale -
 * exactly what semabs_chan_reduce(dXn, X, mean, rstd) returns, without reading dXn or X again (csrc/train.hip has the algebra).  Shapes as
 * semabs_wgrad_conv3 plus B <= 64 and scratch for 256 / ((Ca / 16) (Cx / 16)) rows of 7344 floats; ask semabs_wgrad_conv3_gn_supported first.
 * Replaces, for these layers, the weight-gradient and GroupNorm-statistics halves of loss.backward() (reference: train_vool.py / utils.py loop -> torch
 * autograd of unet3d.py:63-118). */
int semabs_wgrad_conv3_gn_supported(int B, int D0, int D1, int D2, int Ca, int Cx, long scratch_floats, int* ok);
int semabs_wgrad_conv3_gn(const float* dZ, const float* X, const float* mean, const float* rstd, int G, const float* gamma, const float* beta,
                          const float* W, const float* s2, float* dW, double* red, int B, int D0, int D1, int D2, int Ca, int Cx, float* scratch,
                          long scratch_floats, void* stream);

/* Data gradient of a GroupNorm -> Conv3d 3x3x3 layer with the GroupNorm backward applied in the convolution's epilogue:
 *   dX = k0 * conv(dZ, Wbwd) - k1 - ((X - mean) * rstd) * k2 [+ add1]  [0 where X <= 0 if relu_mask],   coef fp32 [B, 16, 3] = (k0, k1, k2) of semabs_gn_bwd_coef
 * = semabs_conv3d(dZ, ..) + semabs_gn_bwd_apply(.., add1) without the intermediate tensor and its two passes (add1: optional, like dX).  w_hi / w_lo = the layer's data-gradient
 * weights (as for semabs_conv3d), in_scale / in_shift = dZ's dynamic scale as an input affine, X = the layer's GroupNorm input, mean / rstd fp32 [B, G];
 * act_flags as semabs_conv3d (exact mode required); absmax_bits (optional, zeroed): max |dX|.  16 -> 16 channels with D0 % 8 == 0, D1 % 8 == 0, D2 % 16 == 0,
 * B <= 32, or - without add1 - channel counts that are multiples of 32 with D0 % 4 == 0: ask semabs_conv3d_gnbwd_supported.  Replaces, for these layers, the data-gradient + GroupNorm halves of loss.backward() (unet3d.py:63-118). */
int semabs_conv3d_gnbwd_supported(int B, int D0, int D1, int D2, int Cin, int Cout, int G, int have_add1, int* ok);
int semabs_conv3d_gnbwd(const void* dZ, const void* w_hi, const void* w_lo, void* dX, const float* in_scale, const float* in_shift, const void* X,
                        const float* mean, const float* rstd, const float* coef, int G, const float* add1, int relu_mask, unsigned int* absmax_bits, int B,
                        int D0, int D1, int D2, int Cin, int Cout, int act_flags, void* stream);

/* out fp64 [B, C, 2] += (sum_v dY, sum_v dY * xhat) per (batch, channel); X = NULL gives plain column sums (bias gradients) */
int semabs_chan_reduce(const float* dY, const float* X, const float* mean, const float* rstd, double* out, int B, long nvox, int C, int G,
                       void* stream);
/* GroupNorm mean / rstd fp32 [B, G] from the fp64 sums of semabs_gn_stats (what the backward needs)      unet3d.py:59-77 */
int semabs_gn_meanrstd(const double* sums, float* mean, float* rstd, int B, int G, long count, float eps, void* stream);
/* GroupNorm backward: per-(b, c) coefficients + dgamma / dbeta (accumulated), then
 * dX = (k1 dXn - k2 - xhat k3 (+ add1 + add2)) (* (mask_y > 0): the ReLU in front of the GroupNorm), max |dX| -> absmax_bits (optional) */
int semabs_gn_bwd_coef(const double* red, const float* gamma, const float* rstd, const float* inv_scale, float* coef, float* dgamma,
                       float* dbeta, int B, int C, int G, long nvox, void* stream);
int semabs_gn_bwd_apply(const float* dXn, const float* X, const float* mean, const float* rstd, const float* coef, const float* add1,
                        const float* add2, const float* mask_y, unsigned int* absmax_bits, float* dX, int B, long nvox, int C, int G, void* stream);

/* element-wise over n (% 4 == 0) floats: mode 0 out = a * (b > 0) (ReLU backward from the output), 1 LeakyReLU backward, 2 out = a + b,
 * 3 out = a * b[0] (b = device scalar); absmax_bits (optional, zeroed uint32) receives the bit pattern of max |out| */
int semabs_ew(const float* a, const float* b, float* out, long n, int mode, float slope, unsigned int* absmax_bits, void* stream);
/* modes 0 - 2 with a * in_scale[0] (device scalar) as the first operand: the "undo the dynamic gradient scale" pass folded into its consumer */
int semabs_ew_scaled(const float* a, const float* b, const float* in_scale, float* out, long n, int mode, float slope, unsigned int* absmax_bits,
                     void* stream);
/* Dynamic power-of-two scale s for a gradient tensor (max |x| * s in [256, 512)) so the split-fp16 data-gradient convolutions keep fp32-like
 * accuracy for tiny gradients: scale_arr[n_arr] = s, shift_arr[n_arr] = 0 (the conv's input affine), s2 = (s, 1 / s); bits = uint32 scratch, or with
 * have_bits = 1 the max |x| pattern semabs_ew already produced (x is then not read) */
int semabs_grad_scale(const float* x, long n, float* scale_arr, float* shift_arr, int n_arr, float* s2, unsigned int* bits, int have_bits,
                      void* stream);
/* MaxPool3d(2) backward: first maximal element of each window takes dY                                   unet3d.py:298-317 */
int semabs_maxpool3d_bwd(const float* X, const float* dY, float* dX, int B, int D0, int D1, int D2, int C, void* stream);
/* + `add` (like dX: the skip connection's gradient) added on the way out, max |dX| -> absmax_bits (optional, zero it first): one pass less per level.
 * relu_mask != 0: X is a post-ReLU activation and dX is wanted in front of that ReLU (dX = 0 where X <= 0): the residual block's mask pass folded in. */
int semabs_maxpool3d_bwd_add(const float* X, const float* dY, const float* add, float* dX, unsigned int* absmax_bits, int relu_mask, int B, int D0, int D1,
                             int D2, int C, void* stream);

/* Y[R, Co] = act(X[R, Ci] . W[Co, Ci]^T + bias), act 0 none / 1 LeakyReLU(slope): the point MLP and sampler MLP layers and, with W
 * transposed by the caller, their data gradients                                                         net.py:358-367, 300-309 */
int semabs_linear_f32(const float* X, const float* W, const float* bias, float* Y, long R, int Ci, int Co, int act, float slope, void* stream);
/* The same layer on the matrix cores with fp32-like accuracy (operands split into fp16 hi + lo, three products): Y = act((s X) W^T + b), W addressed as
 * W[n * w_sn + k * w_sk] (plain: Ci, 1; transposed: 1, Co), in_scale = optional device scalar s (the power-of-two scale of a gradient input; the output
 * stays scaled unless out_scale - a device scalar the accumulator is multiplied by, e.g. 1 / s - is given).  Ci % 4 == 0, Co <= 128, act 0 none / 1 LeakyReLU(slope).                                  net.py:358-367, 215-256 (and their backward)
 * relu_mask (optional, like Y): Y = 0 where relu_mask <= 0 (Y is a gradient in front of the ReLU that produced relu_mask), or with act = 2 Y *= slope
 * there (LeakyReLU; no activation is applied to Y itself); absmax_bits (optional, uint32 [1], zeroed by the caller): bit pattern of max |Y|;
 * colsum (optional, fp32 [Co]): += the column sums of Y (the bias gradient of the layer whose output gradient Y is); x_colsum (optional, fp32 [16], only with
 * Ci = Co = ldx = 16 and at least 2^16 rows): += the column sums of X (unscaled). */
int semabs_linear_rows(const float* X, long ldx, const float* W, long w_sn, long w_sk, const float* bias, float* Y, long R, int Ci, int Co,
                       int act, float slope, const float* in_scale, const float* out_scale, const float* relu_mask, unsigned int* absmax_bits, float* colsum,
                       float* x_colsum, void* stream);

/* scatter-mean backward: dpf[b, p] = dvol[b, flat[p]] / count[flat[p]]; count int32 [nvox] zero-filled by the caller   net.py:185-201 */
int semabs_scatter_mean_bwd(const long long* flat, int* count, const float* dvol, float* dpf, int P, long N, int C, long nvox, void* stream);

/* VOOL head pieces for training: f [P*M, 36] = (trilinear(vol_t), trilinear(vol_r), qn, 0) and the scatter of df back into the two
 * gradient volumes (gather over per-cell point lists; head int32 [P * S^3], next int32 [P * M] scratch)                                                            net.py:215-256, 556-565 */
int semabs_vool_sample(const float* vol_t, const float* vol_r, const float* query, const float* off3, const float* sc3, const int* shape3,
                       int P, long M, float* f, void* stream);
/* absmax_bits (optional, uint32 [1], zeroed by the caller): receives the bit pattern of max |dvol| (what semabs_grad_scale(.., have_max = 1) takes) */
int semabs_vool_sample_bwd(const float* df, const float* query, const float* off3, const float* sc3, const int* shape3, int P, long M,
                           int* head, int* next, float* dvol_t, float* dvol_r, unsigned int* absmax_bits, void* stream);

/* logits = cos(o, rel) / T; loss += sum w BCEwithlogits(logit, label) / n_total; dO and drel (accumulated) = d loss / d o, d rel; dbias (optional,
 * fp32 [P, 64], accumulated) = the column sums of dO per description                                     net.py:566-579, train_vool.py:171-178 */
int semabs_cos_bce(const float* o, const float* rel, const float* label, const float* weight, int P, long M, float temperature, long n_total,
                   float* logits, float* dO, float* drel, double* loss, float* dbias, void* stream);

/* per-step weight layouts: hi / lo[i] = fp16 hi / lo split of src[idx[i]] (idx < 0: zero) - one launch per matrix instead of torch permute / flip /
 * cat / cast chains; the index map of a layout (tap-major, flipped + transposed, parity classes, fragment-packed copy) depends on shapes only */
int semabs_gather_split16(const float* src, const int* idx, long n, void* hi, void* lo, void* stream);
/* ... and all layouts of a step in ONE launch: jobs = device array of njobs x {src, idx, hi, lo, n, first block} (six 64-bit words each; first block =
 * running sum of ceil(n / 1024)), total_blocks = the sum */
int semabs_gather_split16_batched(const void* jobs, int njobs, long total_blocks, void* stream);

/* the pointer head with the loss left to the caller (autograd boundary of SemAbsVOOL: `loss.backward()` of utils.py:404-417):
 * dlogits == NULL: logits = cos(o, rel) / T only; dlogits = d loss / d logits [P*M]: dO, drel (accumulated) = d loss / d o, d rel   net.py:566-579 */
int semabs_cos_head(const float* o, const float* rel, const float* dlogits, int P, long M, float temperature, float* logits, float* dO,
                    float* drel, void* stream);

/* clip_grad_norm_ over the LAMB chunk table: g *= extra_scale, then g *= min(1, max_norm / (norm + 1e-6)); sq fp64 [1] = sum g^2   utils.py:415 */
int semabs_clip_grad_norm(const long long* chunks, int n_chunks, const long long* ptrs, int n_tensors, float max_norm, float extra_scale,
                          double* sq, void* stream);

/* ============================ relevancy storage format (csrc/relio.hip) =================== */
/* What sits between get_clip_saliency and the HDF5 file, and between the file and the network input (the container itself - gzip chunks,
 * region references, file locks - is storage and out of scope). */

/* maps fp32 [L, H, W] -> out fp32 [L + 1, h, w]: nearest-exact resize to the storage dims + the mean-over-labels map as the last row
 *                                                                                              generate_relevancy.py:95-108 */
int semabs_relevancy_pack(const float* maps, float* out, int L, int H, int W, int h, int w, void* stream);
/* feats fp32 [L, E] -> out fp32 [L + 1, E]: append the mean row, L2-normalise every row               generate_relevancy.py:109-118 */
int semabs_text_pack(const float* feats, float* out, int L, int E, void* stream);
/* stored fp32 [R, h, w], rows int64 [P] (or NULL = rows 0..P-1), mean_map fp32 [h, w] (or NULL) -> out fp32 [P, H, W] =
 * out_scale * bilinear(align_corners = False)(stored[rows] - mean_map)                                dataset.py:821-871 (x 50: :1053) */
int semabs_relevancy_unpack(const float* stored, const long long* rows, const float* mean_map, float* out, int P, int h, int w, int H, int W,
                            float out_scale, void* stream);

/* ============================ evaluation metrics (csrc/evalm.hip) ========================= */
/* utils.voxelize_points (utils.py:617-665): flat int64 [BP, N] voxel indices (semabs_voxel_index), pred / label / ignore uint8 [BP, N] ->
 * voxelised prediction / label / ignore uint8 [BP, nvox] (scatter-max semantics, empty voxels ignored); vol_scratch int32 [BP, nvox, 3] */
int semabs_voxelize_eval(const long long* flat, const unsigned char* pred, const unsigned char* label, const unsigned char* ignore,
                         int* vol_scratch, unsigned char* out_pred, unsigned char* out_label, unsigned char* out_ignore, long BP, long N,
                         long nvox, void* stream);
/* utils.prediction_analysis (utils.py:340-380): per row, over the non-ignored elements: counts uint64 [BP, 6] = (valid, positive labels,
 * positive predictions, true positives, union, 0); the ratios are formed on the host from these exact counts */
int semabs_prediction_counts(const unsigned char* pred, const unsigned char* label, const unsigned char* ignore, unsigned long long* counts,
                             long BP, long M, void* stream);

/* ============================ timing helpers ============================================================== */
/* HIP events (timing enabled) for semabs_gemm_f16_ex's start_event / stop_event: the launch records its start / stop timestamps through the
 * dispatch packet itself (hipExtLaunchKernelGGL) - per-launch durations without barrier packets around the kernel (bench.py's roofline leg). */
int semabs_event_create(void** ev);
int semabs_event_destroy(void* ev);
int semabs_event_elapsed_ms(void* start, void* stop, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* SEMABS_H */
