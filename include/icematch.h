/* libicematch — C ABI of the MI355X (gfx950) learned extraction + matching hot path.
 *
 * The reference (franioli/icepy4d) is pure Python: its "plugin" seam for this path is
 * `ImageMatcherBase._match_images(image0, image1, **config)` (`src/icepy4d/matching/matchers.py:276-302`),
 * implemented by `LightGlueMatcher._match_images` (`:1226-1304`) and `SuperGlueMatcher._match_images`
 * (`:892-940`), which call torch modules.  This library is what those two methods bind instead of torch:
 * the host side (`icepy4d_amd/matching/matchers.py`, ctypes) keeps the reference's class / method names.
 *
 * Conventions
 *   - every `d_*` pointer is a DEVICE pointer owned by the caller (e.g. a torch tensor); `h_*` is host memory
 *   - `stream` is a `hipStream_t` passed as void*; calls only enqueue work (no host sync) unless stated
 *   - return value: 0 = ok, negative = error (see `im_last_error`); no exceptions cross the boundary
 *   - a context is not re-entrant: use one context per (process, device, stream)
 *   - dynamic sizes (keypoint counts) stay in device memory: `d_n*` are `int32` device scalars
 */
#ifndef ICEMATCH_H
#define ICEMATCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct im_ctx im_ctx;

/* ---- context ------------------------------------------------------------------------------------ */
int im_version(void);
/* Binds to HIP device `device` (replaces `.to(device)` at `matchers.py:852, 1256-1258`). */
int im_ctx_create(int device, im_ctx** out);
void im_ctx_destroy(im_ctx* ctx);
const char* im_last_error(im_ctx* ctx);
/* Allocates every workspace for images up to max_h x max_w, `max_images` images per call and `max_kpts`
 * keypoints per image. Must be called before any forward; may be called again to grow. Synchronises. */
int im_ctx_reserve(im_ctx* ctx, int max_h, int max_w, int max_images, int max_kpts);

/* Per-launch timing with HIP events recorded on the launch stream (what bench.py's roofline leg reads).
 * begin: synchronises and arms; end: synchronises and writes a JSON object {"kernel": {"count", "total_ms"}}. */
int im_profile_begin(im_ctx* ctx);
int im_profile_end(im_ctx* ctx, char* buf, size_t cap);

/* ---- weights: tensors are passed under their OFFICIAL state-dict key names --------------------------
 * model: "superpoint" (`lightglue/superpoint.py:118-137` == `SuperGlue/models/superpoint.py:122-140`),
 *        "lightglue"  (`lightglue/lightglue.py:350-373`), "superglue" (`SuperGlue/models/superglue.py:221-242`).
 * `h_data` is host fp32, `numel` elements, torch-contiguous layout. Replaces `load_state_dict`
 * (`lightglue/superpoint.py:139-140`, `lightglue/lightglue.py:376-392`, `superglue.py:244-247`). */
int im_set_tensor(im_ctx* ctx, const char* model, const char* key, const float* h_data, size_t numel);
/* Checks that every tensor of `model` was provided, re-packs (conv slabs, head-major q/k/v, folded BatchNorm)
 * and uploads. Synchronises. */
int im_finalize_weights(im_ctx* ctx, const char* model);

/* ---- SuperPoint: `SuperPoint.extract` / `.forward` -----------------------------------------------
 * (`lightglue/superpoint.py:146-231`, `SuperGlue/models/superpoint.py:151-220`).
 * d_img: uint8 [n_images][h][w][channels], channels = 1 (gray) or 3 (RGB, the layout `core/images.py:75` hands to
 * `match()`); the u8 -> float conversion of `_frame2tensor` (`matchers.py:263-274, 1212-1220`) and, for 3 channels, the gray
 * conversion happen per pixel inside the first convolution's producer, exactly as the reference orders them:
 *   flavour 0 = LightGlue: scale to float, then kornia's weights on the FLOAT image (`lightglue/utils.py:35-36`);
 *               border := -1 before the threshold (`lightglue/superpoint.py:177-184`)
 *   flavour 1 = SuperGlue: cv2.cvtColor(RGB2GRAY) on the UINT8 image (fixed point), then scale (`matchers.py:911-917`);
 *               threshold, then the coordinate border mask (`SuperGlue/models/superpoint.py:176-189`)
 * channels = 4 is a float32 gray image [n_images][h][w] (4 bytes per pixel) the caller has already scaled to [0, 1]: the
 * `resize` option of `SuperPoint.extract` (`lightglue/utils.py:30-33`), whose kornia resize runs on the host.
 * max_kpts <= 0 means unlimited (bounded by the reserved max_kpts: see im_superpoint_candidates).
 * Outputs (row stride = reserved max_kpts): d_kpts [n_images][max_kpts][2] (x, y), d_scores [n_images][max_kpts],
 * d_desc [n_images][max_kpts][256] (L2-normalised), d_n [n_images]. */
int im_superpoint_forward(im_ctx* ctx, const uint8_t* d_img, int n_images, int h, int w, int channels,
                          int nms_radius, float threshold, int border, int max_kpts, int flavour,
                          float* d_kpts, float* d_scores, float* d_desc, int32_t* d_n, void* stream);
/* Number of candidates (NMS survivors above the threshold, inside the border) each image of the LAST im_superpoint_forward
 * had BEFORE the top-k / capacity cut; h_counts: host int32 [n_images]. Synchronises the stream. `max_keypoints = -1`
 * (`SuperGlue/models/superpoint.py:176-203`, icepy4d's SuperGlue default `matchers.py:859`) means "all of them": the caller
 * compares with the reserved max_kpts, grows the workspace (im_ctx_reserve) and repeats the forward if any were cut. */
int im_superpoint_candidates(im_ctx* ctx, int n_images, int32_t* h_counts, void* stream);

/* ---- LightGlue: `LightGlue._forward` on the reference's CPU path (`lightglue/lightglue.py:436-556`) ----
 * Inputs for image 0/1 are the two slices of the SuperPoint outputs above (same strides, n_images = 2).
 * h_size: host [2][2] = (W, H) of image 0 and 1 (`image_size`, `lightglue/superpoint.py:229`).
 * Outputs: d_matches [2][max_kpts] int32 (-1 = none; row 0 = matches0, row 1 = matches1),
 * d_mscores [2][max_kpts], d_prune [2][max_kpts] int32, d_info int32[4] = {stop, n0_final, n1_final, 0}. */
typedef struct {
    double depth_confidence; /* <= 0 disables early stop   (`lightglue.py:317`); doubles: the reference holds */
    double width_confidence; /* <= 0 disables point pruning (`lightglue.py:318`); Python floats and derives   */
    double filter_threshold; /* `lightglue.py:319`                                 1 - width_confidence in double */
    int n_layers;            /* 9 */
    int pruning_min_kpts;    /* an image is pruned after a layer only while it holds MORE live points than this
                                (`lightglue.py:326-331, 495, 503, 581-585`): -1 = the reference's CPU path (always), 1024 = its CUDA
                                path, 1536 = CUDA with FlashAttention */
} im_lightglue_conf;
int im_lightglue_forward(im_ctx* ctx, const float* d_kpts, const float* d_desc, const int32_t* d_n,
                         const float* h_size, const im_lightglue_conf* conf,
                         int32_t* d_matches, float* d_mscores, int32_t* d_prune, int32_t* d_info, void* stream);

/* The same for n_pairs independent pairs in ONE sequence of launches (a batch dimension over pairs inside every kernel):
 * image 2p / 2p + 1 of the inputs are pair p; all pairs share h_size (equal-shape tile pairs of an epoch, `matchers.py:367-394`;
 * the epochs of a sequence, `main_dev.py:60`). Needs im_ctx_reserve(max_images >= 2 n_pairs). Outputs: d_matches / d_mscores /
 * d_prune [2 n_pairs][max_kpts] (rows 2p, 2p + 1 = matches0, matches1 of pair p), d_info [n_pairs][4]. Results are
 * bit-identical to n_pairs single calls. */
int im_lightglue_forward_pairs(im_ctx* ctx, int n_pairs, const float* d_kpts, const float* d_desc, const int32_t* d_n,
                               const float* h_size, const im_lightglue_conf* conf, int32_t* d_matches, float* d_mscores,
                               int32_t* d_prune, int32_t* d_info, void* stream);

/* ---- SuperGlue: `SuperGlue.forward` (`SuperGlue/models/superglue.py:250-305`) ---------------------------
 * d_desc rows are [max_kpts][256] (the transposed view of the reference's [256, K]); h_shape: host [2][2] = (H, W)
 * of the image tensors (`data['image0'].shape`). Outputs as for LightGlue (d_info = {0, n0, n1, 0}). */
typedef struct {
    int sinkhorn_iterations; /* icepy4d default 20 (`matchers.py:857`) */
    double match_threshold;  /* icepy4d default 0.3 (`matchers.py:864`) */
    int n_layers;            /* 18 */
} im_superglue_conf;
int im_superglue_forward(im_ctx* ctx, const float* d_kpts, const float* d_scores, const float* d_desc,
                         const int32_t* d_n, const float* h_shape, const im_superglue_conf* conf,
                         int32_t* d_matches, float* d_mscores, int32_t* d_info, void* stream);

/* Match-table record of one pair for the sharded sequence driver (replaces the per-epoch bookkeeping of
 * `main_dev.py:160-173`): int32 [8 + 2 * max_kpts] = {epoch, n0, n1, n_matches, stop, 0, 0, 0}, matches0, scores0 bits. */
int im_pack_record(im_ctx* ctx, const int32_t* d_n, const int32_t* d_matches0, const float* d_mscores0,
                   const int32_t* d_info, int epoch, int32_t* d_record, void* stream);
/* n_pairs records at once from the outputs of im_lightglue_forward_pairs (epochs first_epoch .. first_epoch + n_pairs - 1);
 * d_records [n_pairs][8 + 2 * max_kpts]. With d_kpts != NULL (the [2 n_pairs][max_kpts][2] keypoints of im_superpoint_forward)
 * every record also carries the keypoints of both images as float32 bit patterns: d_records [n_pairs][8 + 6 * max_kpts]
 * (the 98 KB record of a sharded run whose gathered table must be self-contained; the reference keeps them in
 * `Epoch.features`, `main_dev.py:160-173`). */
int im_pack_records(im_ctx* ctx, int n_pairs, const int32_t* d_n, const int32_t* d_matches, const float* d_mscores,
                    const int32_t* d_info, int first_epoch, int32_t* d_records, const float* d_kpts, void* stream);
/* Copies an internal buffer of the last forward ("lg_x", "lg_cos", "lg_sin", "sim", "md", "sp_smap", "sp_nms") for
 * stage-level parity tests. */
int im_debug_read(im_ctx* ctx, const char* name, float* d_dst, size_t nfloats, void* stream);
/* Measurement aid (no reference counterpart): the shader clock the two matrix-core kernel classes hold INSIDE their main loops. arm = 1: from
 * now on every attention launch (the matchers' self / cross blocks) and every Winograd convolution launch of this context has the first wave of
 * each block store the shader cycles and the 100 MHz reference ticks of its main loop into probe words nothing else reads; arm = 2: read,
 * stay armed; arm = 0: read and disarm. h_out[4] = {attention MHz (median over blocks), blocks, convolution MHz, blocks}. bench.py emits them
 * as `roofline.sustained_clock_mhz` (MI355X_MICROARCH.md, DVFS give-back item 6: cycles / reference ticks x 100 MHz). */
int im_debug_clock_probe(im_ctx* ctx, int arm, double* h_out, void* stream);
/* Debugging aid (no reference counterpart; GPU AddressSanitizer is unavailable on the MI355X pool): with IM_DEBUG_GUARDS=1 in the
 * environment when a context is created, every device buffer the library allocates (workspace of im_ctx_reserve, packed weights,
 * scratch) carries 256 bytes of guard words on both sides; they are compared by a small kernel at the end of every forward /
 * stage entry point, before a buffer is freed and in im_ctx_destroy. A changed word fails that call with -90 (im_last_error names
 * the buffer and the side) and is counted here: number of guard failures seen by this process so far (0 when the mode is off). */
int im_debug_guard_failures(void);
/* Self-test of that mode: issues one stray 4-byte store right behind the newest library buffer, expects the check to fail with
 * -90, restores the word. Returns 0 when the stray store was caught, -93 when the mode is off, -94 when it went unnoticed. */
int im_debug_guard_selftest(im_ctx* ctx, void* stream);
/* Compares the guard words now, from the host (0 when the mode is off or every word is intact, -90 otherwise). Forwards recorded into
 * a HIP graph carry no check of their own (a captured check would keep the buffer table of the capture): callers that replay graphs
 * call this after a replay. */
int im_debug_guards_check(im_ctx* ctx, void* stream);

/* ---- stage entry points (what the stage-isolated parity tests call; also usable on their own) ----------- */
/* C[m][n] = alpha * (sum_k A[m][k] W[n][k] + bias[n]); fp32 GEMM. bias may be NULL. big_tile bit 0: 128x128 tiles; bit 1: the product on
 * the bf16 matrix cores with fp32 accuracy (operands cut into three bf16 values, six products per fp32 product) instead of the f32-input MFMA. */
int im_gemm_nt(im_ctx* ctx, const float* d_a, const float* d_w, const float* d_bias, float* d_c,
               int m, int n, int k, float alpha, int big_tile, void* stream);
/* The feed-forward tail of a transformer block / GNN layer in one kernel:
 *   x[z][m][0..256) += W3 . act(W0 . [x[z][m] | att[z][m]] + b0) + b3   for the first d_n[z] (NULL: n_rows) rows of each image z
 * act 0: gelu(layernorm(.)) = LightGlue's ffn (`lightglue/lightglue.py:144-149, 160-162, 212-216`); act 1: relu(.) = SuperGlue's
 * mlp with its BatchNorm folded into W0 / b0 (`SuperGlue/models/superglue.py:51-61, 104-116`; h_ln_g / h_ln_b unused, may be NULL).
 * d_x / d_att: [n_images][n_rows][256]; weights on the HOST in torch layout (W0 [512][512] acting on cat([x, att]), b0 [512],
 * LayerNorm g / b [512], W3 [256][512], b3 [256]): packed + uploaded inside; synchronises. */
int im_ffn_fused(im_ctx* ctx, int act, float* d_x, const float* d_att, const float* h_w0, const float* h_b0, const float* h_ln_g,
                   const float* h_ln_b, const float* h_w3, const float* h_b3, int n_images, int n_rows, const int32_t* d_n, void* stream);
/* 3x3 conv, NHWC fp32, weights in torch layout [cout][cin][3][3] on the HOST (packed + uploaded inside; synchronises) */
int im_conv3x3(im_ctx* ctx, const float* d_in, const float* h_weight, const float* h_bias, float* d_out,
               int b, int h, int w, int cin, int cout, int relu, int pool, void* stream);
/* the same convolution in Winograd F(2x2, 3x3) form (2.25x fewer matrix-core FLOPs; what the forward pass uses) */
int im_conv3x3_winograd(im_ctx* ctx, const float* d_in, const float* h_weight, const float* h_bias, float* d_out,
                        int b, int h, int w, int cin, int cout, int relu, int pool, void* stream);
/* fp32 flash attention: q, k, v [batch][heads][n_max][64]; out [batch][n_max][heads*64]; cross bit 0: kv of image z^1; bit 1: the kernel on the
 * f32-input MFMA (rounds 1-5) instead of the bf16-plane kernel, for A/B; bit 2: the bf16-plane kernel cuts K / V itself while it stages them
 * (the form for callers without the plane workspace) instead of reading the planes of its first launch */
int im_flash_attn(im_ctx* ctx, const float* d_q, const float* d_k, const float* d_v, float* d_out,
                  const int32_t* d_n, int n_max, int batch, int heads, int cross, float scale, void* stream);
/* `simple_nms` (`lightglue/superpoint.py:50-65`): d_scores, d_out [n_images][h][w] */
int im_nms(im_ctx* ctx, const float* d_scores, float* d_out, int n_images, int h, int w, int radius, void* stream);
/* border / threshold / row-major compaction / top-k (`lightglue/superpoint.py:177-200`) on an NMS map */
int im_select_topk(im_ctx* ctx, const float* d_nms, int n_images, int h, int w, int border, float threshold,
                   int max_kpts, float* d_kpts, float* d_scores, int32_t* d_n, void* stream);
/* `sample_descriptors` (`lightglue/superpoint.py:75-87`) incl. the dense per-cell L2 normalisation (`:205`):
 * d_dense_raw = convDb output, NHWC [n_images][hc][wc][256]; d_desc [n_images][max_kpts][256] */
int im_sample_descriptors(im_ctx* ctx, const float* d_dense_raw, int n_images, int hc, int wc,
                          const float* d_kpts, const int32_t* d_n, float* d_desc, void* stream);
/* `sigmoid_log_double_softmax` + `filter_matches` (`lightglue/lightglue.py:253-306`) on a given similarity matrix:
 * d_sim [m][ld], d_z0 [m], d_z1 [n] -> d_matches [2][max(m,n)] int32 etc. (compact index space) */
int im_assign_from_sim(im_ctx* ctx, const float* d_sim, int m, int n, int ld, const float* d_z0, const float* d_z1,
                       float threshold, int32_t* d_m0, int32_t* d_m1, float* d_ms0, float* d_ms1, void* stream);
/* `log_optimal_transport` (`SuperGlue/models/superglue.py:152-186`): d_scores [m][ld] -> d_out [(m+1)][(n+1)] */
int im_log_optimal_transport(im_ctx* ctx, const float* d_scores, int m, int n, int ld, float bin_score, int iters,
                             float* d_out, void* stream);

/* ---- Gaussian pyramid steps on uint8 images [n_images][h][w][channels] (channels interleaved, 1..4), what `Quality` resizing and
 * tile preselection call through OpenCV in the reference (`cv2.pyrDown` / `cv2.pyrUp`, `matchers.py:529-530, 599-609`):
 * im_pyr_down -> [n_images][(h + 1) / 2][(w + 1) / 2][channels], im_pyr_up -> [n_images][2 h][2 w][channels]. Enqueue only. */
int im_pyr_down(im_ctx* ctx, const uint8_t* d_in, uint8_t* d_out, int n_images, int h, int w, int channels, void* stream);
int im_pyr_up(im_ctx* ctx, const uint8_t* d_in, uint8_t* d_out, int n_images, int h, int w, int channels, void* stream);

/* ---- tile mode (`ImageMatcherBase._match_by_tile`, `matchers.py:304-469`) -----------------------------------------------
 * The tail of the tile loop for ALL tile pairs of an image pair in one call (`matchers.py:402-448`): valid matches of every
 * pair are shifted to image coordinates ((kpt + tile origin) + image origin, fp32), concatenated in tile-pair order, and
 * `np.unique(mkpts0, axis=0, return_index=True)` is applied: rows in lexicographic (x, y) order, first occurrence kept.
 *   d_matches [n_pairs][max_kpts] int32 : matches0 of every tile pair (-1 = none), rows = keypoints of its first tile
 *   d_slots   [n_pairs][2] int32        : which entry of the feature bank holds tile 0 / tile 1 of the pair
 *   d_off     [n_pairs][4] float        : (x, y) origin of tile 0 and of tile 1 in their images (`lim0[0:2]`, `lim1[0:2]`)
 *   h_origin  [4] float (host)          : `t0_origin`, `t1_origin`
 *   d_kp_bank [n_tiles][max_kpts][2], d_n_bank [n_tiles] : keypoints and keypoint counts of every extracted tile
 * Outputs (capacity n_pairs * max_kpts rows): d_count = number of unique rows S; d_idx0 / d_idx1 [S] = flat bank row
 * (tile * max_kpts + keypoint) of each surviving match in image 0 / 1 (for im_gather_rows on descriptors and scores);
 * d_kp0 / d_kp1 [S][2] = the matched points in image coordinates. Enqueues only. */
int im_merge_tile_matches(im_ctx* ctx, int n_pairs, int max_kpts, const int32_t* d_matches, const int32_t* d_slots, const float* d_off,
                          const float* h_origin, const float* d_kp_bank, const int32_t* d_n_bank, int32_t* d_count, int32_t* d_idx0,
                          int32_t* d_idx1, float* d_kp0, float* d_kp1, void* stream);
/* d_dst[r][:] = d_src[d_idx[r]][:] for r < n, rows of row_floats floats (descriptor / score banks -> matched features). */
int im_gather_rows(im_ctx* ctx, const float* d_src, int row_floats, const int32_t* d_idx, int n, float* d_dst, void* stream);

/* Fundamental-matrix RANSAC over matched keypoints (`src/icepy4d/matching/geometric_verification.py:11-102`, which
 * calls pydegensac / cv2 USAC_MAGSAC on the CPU): n_hyp seeded 8-point hypotheses scored by Sampson error in parallel.
 * d_p0, d_p1 [n][2] float (x, y); d_F [9] double = matrix of the best hypothesis (unit Frobenius norm); d_mask [n] uint8
 * its inliers; d_info {inlier count, hypothesis index}. The least-squares refit on the inliers is left to the caller. */
int im_ransac_fundamental(im_ctx* ctx, const float* d_p0, const float* d_p1, int n, int n_hyp, double threshold,
                          unsigned int seed, double* d_F, uint8_t* d_mask, int32_t* d_info, void* stream);

/* Relative orientation, device stage (replaces the RANSAC inside `cv2.findEssentialMat`, `src/icepy4d/sfm/geometry.py:64-66`):
 * like im_ransac_fundamental on NORMALISED image coordinates (d_x0, d_x1 [n][2] float), every 8-point hypothesis projected onto
 * the essential manifold (two equal singular values, one zero) before it is scored; d_E [9] double = the best hypothesis
 * (x1^T E x0 = 0, unit Frobenius norm), d_mask / d_info as above. The cheirality test (`cv2.recoverPose`, `geometry.py:70-75`)
 * and the 5-7 correspondence case (five-point solver) stay on the host: one 3 x 3 matrix. */
int im_ransac_essential(im_ctx* ctx, const float* d_x0, const float* d_x1, int n, int n_hyp, double threshold,
                        unsigned int seed, double* d_E, uint8_t* d_mask, int32_t* d_info, void* stream);

/* Linear two-view triangulation of n points on the device (replaces the per-point Python loop of
 * `src/icepy4d/sfm/triangulation.py:153-186`): h_P0, h_P1 = the two 3 x 4 projection matrices (row-major doubles in HOST
 * memory), d_x0, d_x1 [n][3] double homogeneous image points, d_X [n][4] double = homogeneous points normalised to X[3] = 1.
 * The reference's formulation (unknowns X and one depth per view, 6 x 6 system per point, right singular vector of the smallest singular
 * value): equal to the reference's outputs within 1e-9 relative (tests/golden/g10_triangulation.npz). */
int im_triangulate_linear(im_ctx* ctx, const double* h_P0, const double* h_P1, const double* d_x0, const double* d_x1, int n,
                          double* d_X, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ICEMATCH_H */
