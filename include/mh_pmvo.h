/*
 * mh_pmvo.h -- C ABI of libmhpmvo.so: MonoHair's PMVO hot path as hand-written HIP kernels for
 * MI355X (gfx950).  This is the drop-in boundary: the reference has no FFI of its own (the path is
 * plain PyTorch, /root/reference/PMVO.py), so each entry point below names the reference function(s)
 * whose tensor-op sequence it replaces; the Python mirror of the reference classes
 * (monohair_amd/pmvo.py, gabor.py, pmvo_utils.py) binds them with ctypes -- see INTEGRATION.md.
 *
 * Conventions
 *   - every pointer that is not marked "host" is a DEVICE pointer owned by the caller (e.g. a torch
 *     tensor's data_ptr()); nothing is retained after the call returns except inside mh_ctx;
 *   - all calls are asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - return value: 0 on success, negative mh_status on failure; mh_last_error() returns a
 *     thread-local description.  No exceptions cross the ABI.  No global state.
 *   - layouts are the reference's (row-major, fp32): vis[V,N], ori[V,N,2], conf[V,N], mask[V,N],
 *     ori_patch[V,N,P,2], conf_patch[V,N,P], with P = patch*patch taps in row-offset-major order.
 */
#ifndef MH_PMVO_H
#define MH_PMVO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mh_ctx mh_ctx;

enum mh_status {
    MH_OK = 0,
    MH_ERR_ARG = -1,     /* bad argument */
    MH_ERR_HIP = -2,     /* a HIP runtime call failed */
    MH_ERR_STATE = -3,   /* call order violated (e.g. views not set) */
    MH_ERR_NOMEM = -4
};

#define MH_CAM_STRIDE 48 /* floats per camera record: pose[16] | proj[16] | inv(pose[:3,:3])[9] | pad */
#define MH_TOPK 20       /* PMVO.py:341 */

const char *mh_last_error(void);
int mh_version(void);

/* ---- context: owns the packed per-view maps, the camera table and the depth-offset table -------- */
int mh_ctx_create(int device_id, mh_ctx **out);
void mh_ctx_destroy(mh_ctx *ctx);

/* Allocate packed storage for V views of H x W pixels (20 B / pixel: {ori_row, ori_col, conf, depth}
 * + mask).  Replaces the per-view H2D copies of PMVO.__init__ (PMVO.py:21-28). */
int mh_ctx_alloc_views(mh_ctx *ctx, int V, int H, int W);

/* Pack one view.  depth/mask are the reference's [H,W,C] arrays of which channel 0 is used
 * (PMVO.py:485,523): pass the channel count as the pixel stride.  cam_host: MH_CAM_STRIDE floats (host). */
int mh_ctx_set_view(mh_ctx *ctx, int view, const float *cam_host, const float *depth, int depth_stride,
                    const float *ori /*[H,W,2]*/, const float *conf /*[H,W]*/, const float *mask,
                    int mask_stride, void *stream);

/* Same, from the 8-bit image files the maps are stored in (Utils/PMVO_utils.py:255-313: best_ori/ gray u8,
 * conf/ gray u8, hair_mask/ channel 0 u8) without the float64 host decode: lut_host is 256 x {ori_row, ori_col,
 * conf, mask} floats -- the loader's value for each pixel code -- and the decode is a table lookup in the pack
 * kernel (bit-identical to uploading the decoded float maps).  ori_u8/conf_u8/mask_u8: device [H,W] planes. */
int mh_ctx_set_view_u8(mh_ctx *ctx, int view, const float *cam_host, const float *depth, int depth_stride,
                       const unsigned char *ori_u8, const unsigned char *conf_u8, const unsigned char *mask_u8,
                       const float *lut_host /*[256][4]*/, void *stream);

/* The S depth offsets of PMVO.sample_next_3d_pos (PMVO.py:274-278), host pointer, S <= 256. */
int mh_ctx_set_depth_offsets(mh_ctx *ctx, const float *offsets_host, int S);

/* Host -> device copy of a points chunk on `stream` (PMVO.py:40, `torch.from_numpy(points).type(torch.float).to(device)`):
 * a plain hipMemcpyAsync.  `host` should be page-locked for the copy to be asynchronous; the caller keeps it unchanged until the
 * copy has completed (an event of its own).  Exists because the tensor library's own non-blocking copy from a pinned source
 * also records an allocator-tracking event per call: 70 -> 55 us of host time per forward() on the 8-bit loop, where the host
 * has 0.24 ms per iteration to enqueue the next one. */
int mh_upload_async(mh_ctx *ctx, const void *host, void *device, size_t bytes, void *stream);
/* The same copy when `pinned_host` IS page-locked, device-visible memory (hipHostMalloc / a pin_memory tensor): up to 1 MiB
 * (4-byte aligned) it is issued as a KERNEL on `stream` that reads the host buffer over the link, larger copies as
 * hipMemcpyAsync.  Round 6: the 60 KB chunk of an iteration through hipMemcpyAsync stalls ONE call for 6-7 ms every so often
 * (the runtime reclaiming the copy commands that piled up while the host ran ahead: 7 ms of idle GPU when its queue is
 * shallow -- the "one timed round in ten is 20 % slower" of rounds 3-5); the kernel form has no such call, and the iteration
 * is 0.8 % (fp32 maps) / 4 % (8-bit maps) faster with it because the copy no longer waits for a copy engine.  MH_UPLOAD_KERNEL=0
 * in the environment selects hipMemcpyAsync for every size (A/B). */
int mh_upload_pinned(mh_ctx *ctx, const void *pinned_host, void *device, size_t bytes, void *stream);

/* ---- K3+K4+K5: PMVO.Compute_Visible_and_Ori (PMVO.py:346-376) with project_points (:378-397),
 * the gathers (:482-523) and compute_visible (:525-529).  Any output may be NULL.
 * pixf[V,N,2] = unrounded (row, col) of each point in each view (what Camera.uv2pixel returns,
 * Camera_utils.py:60-71); the search kernel reuses it. */
int mh_project_gather(mh_ctx *ctx, const float *points /*[N,3]*/, int N, int patch, float *vis, float *ori,
                      float *conf, float *mask, float *ori_patch, float *conf_patch, float *pixf, void *stream);

/* ---- K6: PMVO.Find_max_conf_from_visible_view (PMVO.py:339-343).  out_idx/out_val are [MH_TOPK,N].
 * Equal values come back in the order torch.topk gives them on the CPU (std::nth_element + std::sort of libstdc++ on
 * (value, view) pairs, restated step for step for one wave per point in csrc/mh_topk_wave.h) -- confidences from 8-bit
 * maps saturate, ties are the rule, and the base views decide which candidates are tried.
 * mh_ctx_set_option("topk_order", 1) selects the simpler rule of round 1 instead (value descending, then view index
 * ascending), 2 the literal one-lane-per-point form (csrc/mh_topk_order.h; slow, cross-check).  Needs V >= MH_TOPK
 * (the reference raises below 20 views). */
int mh_topk_views(mh_ctx *ctx, const float *vis, const float *conf, int N, int32_t *out_idx, float *out_val,
                  void *stream);

/* bytes of scratch mh_search_forward / mh_refine_loss need for N points and this patch size */
size_t mh_search_scratch_bytes(mh_ctx *ctx, int N, int patch);
/* byte offset, inside that scratch, of the [V,N] uint8 array of tap-list lengths the preparation kernels leave there
 * (0 for views that do not see the point): what the search orders its workgroups by, and what a caller can read to
 * count the (candidate, view, tap) evaluations a launch actually executes */
size_t mh_search_counts_offset(mh_ctx *ctx, int N, int patch);

/* ---- K7-K10 fused: the body of PMVO.forward (PMVO.py:50-78): for base-view ranks 0,rank_step,...
 * sample_next_3d_pos (:263-335), compute_reproject_ori (:219-241), compute_prj_loss (:151-209) and the
 * best-so-far update (:57-70); line_ori = normalize(best_sample - point).
 * Inputs are the tensors mh_project_gather produced for the same points.  Optional outputs may be NULL.
 *
 * THE N POINTS OF A CALL ARE ONE BATCH OF THE REFERENCE (PMVO.py:572-574 hands forward() 5000 points at a time), and a
 * point's answer depends on its batch as it does there: the points that share a (rank, base view) select how the sgemms
 * of sample_next_3d_pos round (Utils/Camera_utils.py:50-53,103; option "reproject_rule"), and the last N*S mod 32 samples of
 * the batch are the trailing columns of ATen's sums over the views (option "sum_block") -- see mh_ctx_set_option.  Base view
 * indices of ranks whose base_val is <= 0 may be anything (they are clamped; PMVO.py:64 never takes those ranks).
 * The scratch is opaque and must come from this library's own preparation calls: the search relies on the tap records being
 * unit vectors (|cos| <= 1 + 2^-14 in its integer-key body). */
int mh_search_forward(mh_ctx *ctx, const float *points, int N, int patch, float conf_threshold, int nrank,
                      int rank_step, const float *vis, const float *ori, const float *pixf,
                      const float *ori_patch, const float *conf_patch, const int32_t *base_idx,
                      const float *base_val, void *scratch, size_t scratch_bytes, float *line_ori,
                      float *min_loss, uint8_t *high_conf, float *best_sample, int32_t *best_rank,
                      int32_t *best_s, void *stream);

/* ---- The same path without materialising the patch tensors (what PMVO.forward uses): mh_forward_prepare is
 * the projection / visibility / centre-sample part of Compute_Visible_and_Ori (PMVO.py:346-376) fused with the
 * tap-list preparation, straight from the packed maps (patches of views that fail the depth test are not even
 * gathered); mh_topk_views then ranks the base views; mh_search_prepared runs the fused loss search on the
 * prepared scratch (its tail is work space: the search takes the points in descending order of work, which it
 * derives there).  Results are identical to mh_project_gather + mh_search_forward.  mask may be NULL. */
int mh_forward_prepare(mh_ctx *ctx, const float *points, int N, int patch, float conf_threshold, float *vis,
                       float *ori, float *conf, float *mask, void *scratch, size_t scratch_bytes, void *stream);
int mh_search_prepared(mh_ctx *ctx, const float *points, int N, int patch, float conf_threshold, int nrank,
                       int rank_step, const float *ori, const int32_t *base_idx, const float *base_val,
                       void *scratch, float *line_ori, float *min_loss, uint8_t *high_conf, float *best_sample,
                       int32_t *best_rank, int32_t *best_s, void *stream);
/* PMVO.forward (PMVO.py:39-78) in ONE call: mh_forward_prepare + mh_topk_views + mh_search_prepared on the same stream
 * (base_idx [20,N] int32 and base_val [20,N] receive the ranking).  What monohair_amd.pmvo.PMVO.forward calls: on 8-bit
 * maps an iteration is 0.24 ms of GPU work, so every host-side call per iteration counts.  With the default kernels the
 * ranking kernel also writes the work classes of the search's launch order and counts the batch's points per (rank, base
 * view) (fewer launches than the three calls; same results -- the order only decides WHEN a point is processed). */
int mh_forward(mh_ctx *ctx, const float *points, int N, int patch, float conf_threshold, int nrank, int rank_step,
               float *vis, float *ori, float *conf, float *mask, void *scratch, size_t scratch_bytes, int32_t *base_idx,
               float *base_val, float *line_ori, float *min_loss, uint8_t *high_conf, float *best_sample,
               int32_t *best_rank, int32_t *best_s, void *stream);

/* ---- compute_reproject_ori + compute_prj_loss with ONE given candidate per point: the core of
 * PMVO.refine (PMVO.py:86-90), next = point + ori*step_mul/step_div
 * (0.005 and 4 in the reference, applied in that order).  loss[N] (raw num/den, PMVO.py:199-204). */
int mh_refine_loss(mh_ctx *ctx, const float *points, const float *dir /*[N,3]*/, float step_mul, float step_div,
                   int N, int patch, float conf_threshold, const float *vis, const float *ori_patch,
                   const float *conf_patch, float *loss, uint8_t *high_conf, void *stream);

/* The same loss evaluated straight from the resident maps (no patch tensors): what refine's smoothing loop needs per
 * chunk (PMVO.py:619-623 calls PMVO.refine, i.e. Compute_Visible_and_Ori + this loss).  Bit-identical to
 * mh_project_gather + mh_refine_loss.
 *
 * batch / row0 / total: where these N points sit in the reference's batches.  The loss is a sum over views of a [V, n]
 * tensor per batch of n points (PMVO.py:198-204) and ATen adds the trailing n mod 32 columns of such a sum in another
 * order than the rest (csrc/mh_device.h: mh_row_sum_views), so the last bits of a point's loss depend on where it sits
 * in its batch.  The points are rows row0 .. row0+N-1 of `total` points that the reference processes `batch` at a time
 * (PMVO.py:602-606: 5000); batch = 0: these N points are one batch.  mh_ctx_set_option("sum_block", 0) turns the rule off. */
int mh_refine_loss_maps(mh_ctx *ctx, const float *points, const float *dir, float step_mul, float step_div, int N,
                        int patch, float conf_threshold, float *loss, uint8_t *high_conf, int batch, long long row0,
                        long long total, void *stream);

/* The tail of one chunk of refine's smoothing loop (PMVO.py:91-92, 631-642), in place on slices of the global arrays:
 * update = head_filter && !head_top ? -1 : loss_u;  ori <- center where |cos(center, ori)| < replace_threshold;
 * loss_out = update == -1 ? 0.5 : update.  ori may be NULL (loss only: the single-rank loop applies the replacement with
 * mh_replace_dissimilar inside its dependent chain and evaluates the losses of many chunks in one launch beside it). */
int mh_refine_combine(mh_ctx *ctx, const float *center, const float *loss_u, const unsigned char *head_filter,
                      const unsigned char *head_top, float replace_threshold, float *ori /*[N,3] in/out*/,
                      float *loss_out, int N, void *stream);

/* ---- the intermediate methods of the reference's class PMVO as stand-alone calls (the fused path above does not need
 * them; they keep the class's method surface): project_points (PMVO.py:378-397) for resident view `view`: row_col
 * int32 [N,2] rounded+clamped, z_half = -z/2, out_of_image flags, unrounded (row, col); any output may be NULL. */
int mh_project_points(mh_ctx *ctx, int view, const float *points, int N, int32_t *row_col, float *z_half,
                      unsigned char *out_of_image, float *pixel_unrounded, void *stream);
/* get_depth / get_ori / get_conf / get_mask / get_ori_patch / get_c_patch (PMVO.py:482-523): the resident records
 * {ori_row, ori_col, conf, depth} [N, size*size, 4] and mask [N, size*size] of view `view` at row_col (int64 [N,2]),
 * taps clamped to the image one by one, row offset outer. */
int mh_gather_pixels(mh_ctx *ctx, int view, const long long *row_col, int N, int size, float *records, float *mask,
                     void *stream);
/* compute_visible (PMVO.py:525-529), elementwise. */
int mh_compute_visible(mh_ctx *ctx, const float *depth, const float *z, size_t n, float *out, void *stream);
/* sample_next_3d_pos (PMVO.py:263-335): samples [N,S,3] for base_view [N]; ori = the [V,N,2] centre orientations of
 * Compute_Visible_and_Ori; offsets [S] on the device. */
int mh_sample_next(mh_ctx *ctx, const float *points, const int32_t *base_view, const float *ori, const float *offsets,
                   int N, int S, float *samples, void *stream);
/* compute_reproject_ori / compute_points_prj_ori (PMVO.py:219-260): D [V,N,S,2] = pixel(sample) - pixel(point). */
int mh_reproject_ori(mh_ctx *ctx, const float *points, const float *samples, int N, int S, float *D, void *stream);
/* compute_prj_loss (PMVO.py:151-209) on materialised tensors: loss [N], index int64 [N], high_conf [N];
 * all_loss [N,S] (the per-sample losses after the positive / low-confidence rules) may be NULL. */
int mh_prj_loss(mh_ctx *ctx, const float *D, const float *ori_patch, const float *conf_patch, const float *vis, int V,
                int N, int S, int P, float conf_threshold, float *loss, long long *index, unsigned char *high_conf,
                float *all_loss, void *stream);

/* ---- K13: per-view visibility / mask / confidence votes of PMVO.filter_points (PMVO.py:402-459),
 * PMVO.compute_unvisible_points (:461-480) and PMVO.filter_head_points (:110-137).
 * surface_index/filter_index/unvisible_index/head_filter: uint8 [N]; any may be NULL.
 * batch / row0 / total: as for mh_refine_loss_maps (the votes are sums over views of [V, n] tensors; mask values in
 * (0, 0.2] stay fractional, PMVO.py:427, so the order of the sum can matter). */
int mh_filter_points(mh_ctx *ctx, const float *points, int N, int patch, float conf_threshold,
                     float visible_threshold, uint8_t *surface_index, uint8_t *filter_index,
                     uint8_t *unvisible_index, uint8_t *head_filter, int batch, long long row0, long long total,
                     void *stream);

/* The same votes with the rows taken in the order `order` (int32 [N], a permutation of 0..N-1 on the device, or NULL): the
 * results are those of mh_filter_points, row for row -- a row's votes do not depend on when it is processed -- but a wave of
 * the large-launch kernel then holds 64 spatial neighbours instead of 64 consecutive candidates.  The candidates' own order
 * (a raster over the volume) sweeps every image once per slab: 10 GB of line fetches per pass for isolated 20-byte gathers;
 * in cell order (mh_grid_build's `order` on any grid of a few millimetres: the drivers use cells of 5 mm, or the grid the
 * neighbour search built anyway) the launch over 465 k candidates takes 0.97 instead of 2.15 ms. */
int mh_filter_points_ordered(mh_ctx *ctx, const float *points, int N, int patch, float conf_threshold,
                             float visible_threshold, uint8_t *surface_index, uint8_t *filter_index,
                             uint8_t *unvisible_index, uint8_t *head_filter, int batch, long long row0, long long total,
                             const int32_t *order, void *stream);

/* ---- K11: compute_points_similarity (Utils/PMVO_utils.py:366-382): medoid orientation of each group.
 * Dense form: ori[G,K,3] -> out[G,3], out_index[G].  Segmented form (voxel fit, PMVO.py:717-726):
 * ori[M,3] sorted by group, seg_start[G+1] (device), max_group = largest group size (groups beyond 4096 members take a staged path). */
/* Indexed form: group g = rows index[g*K .. g*K+K) of ori_rows[M,3] (the neighbour lists of refine, PMVO.py:612-618,
 * without materialising ori[index]). */
int mh_medoid_indexed(mh_ctx *ctx, const float *ori_rows, const int32_t *index, int G, int K, float *out,
                      int32_t *out_index, void *stream);
int mh_medoid_dense(mh_ctx *ctx, const float *ori, int G, int K, float *out, int32_t *out_index, void *stream);
int mh_medoid_segmented(mh_ctx *ctx, const float *ori, const int32_t *seg_start, int G, int max_group,
                        float *out, int32_t *out_index, void *stream);

/* refine's in-place replacement rule (PMVO.py:631-636): ori[n] <- center[n] where |cos(center[n], ori[n])| < threshold
 * (0.95 in the reference), cos evaluated as torch.cosine_similarity does on [N,3] fp32 tensors. */
int mh_replace_dissimilar(mh_ctx *ctx, const float *center, float *ori /*in/out [N,3]*/, float threshold, int N,
                          void *stream);

/* Exact k nearest neighbours on a uniform grid: the `points_tree.query(sub_points, 100)` of refine (PMVO.py:612,671).
 * pts_sorted[M,3]: the data points sorted by cell (x fastest), order[M]: their original indices, cell_start[ncell+1];
 * grid_origin_h (host): {ox, oy, oz, cell size}, grid_dims (host): {dx, dy, dz}.  out_idx[Q,k] sorted by (fp64 distance,
 * index); status[Q] != 0 marks queries the kernel could not finish (candidate buffer / ring limit): redo on the host. */
int mh_knn_grid(mh_ctx *ctx, const float *grid_origin_h, const int32_t *grid_dims, const float *pts_sorted,
                const int32_t *order, const int32_t *cell_start, const void *queries /*[Q,3] float32, or float64 if
                query_f64*/, int query_f64, int Q, int k, int first_ring /*cells around the query's cell looked at first
                (>= 1)*/, const int32_t *query_order /*optional [Q]: order in which the queries are taken (locality)*/,
                const unsigned char *valid /*optional [M], by original index: only these points count as neighbours*/,
                int32_t *out_idx, int32_t *status, void *stream);

/* Distance (float64) from each of N float32 points to the nearest of M float64 reference points, exhaustive: the
 * `scalp_tree.query(points, k=1)` of PMVO.filter_head_points (PMVO.py:100-101) -- d = sqrt(min_j ((dx*dx + dy*dy) + dz*dz))
 * evaluated in float64 as scipy's KDTree does.  out_mask (optional, uint8 [N]) = dist < max_dist && z < z_limit in
 * float64: the `head_top_index` of PMVO.py:102-106 (4 cm from the scalp, 1 cm below its top).  out_dist may be NULL. */
int mh_nearest_distance(mh_ctx *ctx, const float *points, int N, const double *ref_points, int M, double *out_dist,
                        double max_dist, double z_limit, unsigned char *out_mask, void *stream);

/* The grid mh_knn_grid searches: points sorted by cell (x fastest) with their original indices and the first sorted
 * position of every cell.  grid_origin_h (host): {ox, oy, oz, cell size}, grid_dims (host): {dx, dy, dz}.  Any of
 * pts_sorted / cell_start [dx*dy*dz + 1] / n_occupied (device int32: number of non-empty cells) may be NULL; order[M]
 * is always written.  (Replaces the torch sort / searchsorted / unique pipeline of the first implementation.) */
/* Bounding box of M float32 points: out6 (device) = {min x, min y, min z, max x, max y, max z} (M == 0: +inf / -inf).  What
 * the grid is laid over (scipy's KDTree needs no such thing: PMVO.py:605). */
int mh_points_bbox(mh_ctx *ctx, const float *points, int M, float *out6, void *stream);
size_t mh_grid_scratch_bytes(int M);
int mh_grid_build(mh_ctx *ctx, const float *grid_origin_h, const int32_t *grid_dims, const float *points, int M,
                  void *scratch, size_t scratch_bytes, float *pts_sorted, int32_t *order, int32_t *cell_start,
                  int32_t *n_occupied, void *stream);

/* Stable ascending sort of n 64-bit keys on their low end_bit bits -> keys_out[n], order[n] (original positions):
 * the grouping of points by voxel of the volume fit, PMVO.py:705-715 (a dict of lists in point order). */
size_t mh_sort_scratch_bytes(int n);
int mh_sort_keys(mh_ctx *ctx, const unsigned long long *keys, int n, int end_bit, void *scratch, size_t scratch_bytes,
                 unsigned long long *keys_out, int32_t *order, void *stream);

/* The grouping step of the volume fit in one call (PMVO.py:697-715 + Utils/PMVO_utils.py:386-404 p2v): voxel keys
 * (x*gy + y)*gz + z of the points evaluated in float64 exactly as numpy does (y, z negated, (p - min) / size, round half
 * to even, x86 int32 cast, clip), stable sort -> keys_sorted[n], order[n]; ori_sorted[n,3] (optional, with ori) = the
 * orientation rows in that order with the sign canonicalisation of :697-698 applied (y > 0 -> negated).  points are
 * float32 or float64 (points_f64); voxel_min (3 doubles) and dims (3 int32) are host arrays.  Nothing is modified in
 * place (the reference's p2v flips its input array). */
size_t mh_voxel_group_scratch_bytes(int n);
int mh_voxel_group(mh_ctx *ctx, const void *points, int points_f64, const float *ori, int n, const double *voxel_min,
                   double voxel_size, const int32_t *dims, void *scratch, size_t scratch_bytes,
                   unsigned long long *keys_sorted, int32_t *order, float *ori_sorted, void *stream);

/* ---- Selection on the device between the stages of refine: the `points[index]` / `ori[index]` of the loss threshold
 * (PMVO.py:651-653), the `centres[keep]` of the shell points (:680-686), their np.concatenate (:690-693) and the run
 * boundaries of the sorted voxel keys (:705-715) -- stable stream compaction, no host round trip.
 * mh_select_rows: element i is selected when (flags[i] && !(veto && veto[i])) != invert.  Selected rows of a / b ([n,3]
 * float32, optional) are written to a_out / b_out starting at row *base (device int32, optional: 0), in their original
 * order; index_out (optional) receives the selected i; *count (device int32) = *base + number selected -- hand it as
 * `base` to a second call to append.  scratch: mh_select_scratch_bytes(n).
 * mh_segment_heads: keys_sorted[n] ascending -> seg_start[G+1] (first position of every run of equal keys, seg_start[G]
 * = n; capacity n+1), head_keys[G] (optional, capacity n), meta (device int32[2]) = {G, largest run}.
 * mh_flag_less: out[i] = x[i] < threshold (float32 comparison; NaN -> 0). */
size_t mh_select_scratch_bytes(int n);
int mh_select_rows(mh_ctx *ctx, const unsigned char *flags, const unsigned char *veto, int invert, int n, const float *a,
                   const float *b, float *a_out, float *b_out, int32_t *index_out, const int32_t *base, int32_t *count,
                   void *scratch, size_t scratch_bytes, void *stream);
int mh_segment_heads(mh_ctx *ctx, const unsigned long long *keys_sorted, int n, int32_t *seg_start,
                     unsigned long long *head_keys, int32_t *meta, void *scratch, size_t scratch_bytes, void *stream);
int mh_flag_less(mh_ctx *ctx, const float *x, float threshold, int n, unsigned char *out, void *stream);
/* flag[0] (device int32) = 1 when two device buffers of `bytes` (a multiple of 4) differ in any bit, else 0: how refine
 * recognises the points a neighbour table was prepared for while optimize ran (monohair_amd/pmvo.py: RefinePrefetch). */
int mh_buffers_differ(mh_ctx *ctx, const void *a, const void *b, size_t bytes, int32_t *flag, void *stream);

/* ---- depth-map producer (the step before the path): Utils/Render_utils.py:310-347 render_bust_hair_depth with
 * the BustObj shader (:146-188) and Renderer.draw/ReadBuffer (:239-262) -- triangles drawn with a LESS depth test,
 * value (-z_camera / 2) * 255, background 255, top-left image origin.  verts[Nv,3] (world, bust offset applied),
 * faces[Nf,3] int32 (several meshes: concatenate them, earlier primitives win depth ties as in GL draw order).
 * pixel_center: 0.5 = OpenGL's sample position, 0.0 = the integer positions PMVO.project_points rounds to.
 * out[H,W,channels] float32 (channels = 3 gives the reference's .npy layout).  scratch: mh_render_scratch_bytes. */
size_t mh_render_scratch_bytes(int Nv, int Nf, int H, int W);
int mh_render_depth(mh_ctx *ctx, const float *cam_host, const float *verts, int Nv, const int32_t *faces, int Nf,
                    int H, int W, float pixel_center, void *scratch, size_t scratch_bytes, float *out, int channels,
                    void *stream);

/* ---- the caller on the other side of the path: Utils/Render_utils.py:269-307 render_data -- the
 * strand segments traced from the exterior volume drawn as GL_LINES of width 3 (StrandsObj, :8-127) over the bust mesh
 * (BustObj, :130-203) with a shared depth test, one of four fragment colourings; the images DeepMVSHair reads
 * (infer_inner.py:60-73).  line_pts / line_tan [2*Nseg,3]: the reference's `Lines` / `tangent` vertex buffers (two
 * vertices per segment).  color_option 0 depth/2, 1 direction, 2 undirected direction (2 theta), 3 white, < 0: strands
 * not drawn; depth_option (mesh fragments) 0 depth/2, 1 black, 2 white; clear: background value of all three channels.
 * out [H,W,3] float32 in the shader's range, top-left origin.  A specified rasteriser (header of csrc/raster.hip) that
 * follows the GL specification's rules; pinned against a real OpenGL implementation (Google SwiftShader,
 * tests/golden/gl_raster.npz) up to what GL leaves to the implementation (sub-pixel snapping, interpolation rounding). */
size_t mh_render_strands_scratch_bytes(int Nv, int Nf, int Nseg, int H, int W);
int mh_render_strands(mh_ctx *ctx, const float *cam_host, const float *verts, int Nv, const int32_t *faces, int Nf,
                      const float *line_pts, const float *line_tan, int Nseg, int H, int W, float pixel_center,
                      int line_width, int color_option, int depth_option, float clear, void *scratch,
                      size_t scratch_bytes, float *out, void *stream);

/* ---- K1+K2: calOrientationGabor.forward with iter=1 (preprocess_capture_data/GaborFilter.py:29-145):
 * 180 real Gabor kernels 17x17 (sigma 1.8/2.4, lambda 4), |response| argmax -> orientation index,
 * response-curve variance -> confidence normalised by the image maximum.
 * image[H,W] (DoG-filtered gray) -> orient_index[H,W] (int32, degrees), conf[H,W], variance[H,W] (un-normalised). */
int mh_gabor_bank(mh_ctx *ctx, const float *image, int H, int W, int32_t *orient_index, float *conf,
                  float *variance, void *stream);
/* Optional: install the 180 x 17 x 17 kernels (host pointer, kernel-major like gabor_fn's output) instead of
 * the bank the library builds on the device; lets the host reproduce the reference's CPU transcendental
 * functions bit for bit. */
int mh_gabor_set_bank(mh_ctx *ctx, const float *bank_host);
/* The difference-of-Gaussians prefilter of the stage (GaborFilter.py:190-192: skimage.filters.difference_of_gaussians(img,
 * 0.4, 10) = img_as_float, two scipy.ndimage.gaussian_filter passes with mode='nearest', their difference) in float64 with
 * scipy's operation order -- two launches.  image: uint8 [H,W] (in_kind 0; codes are multiplied by the double 1/255) or
 * float64 [H,W] (in_kind 1), device.  w_lo / w_hi: HOST pointers to the symmetric halves w[0..r] (w[r] = centre) of the two
 * normalised 1-D Gaussians as scipy builds them (radius = int(4 sigma + 0.5) <= 48).  scratch: mh_dog_scratch_bytes(H, W)
 * device bytes.  out64 and/or out32 (device, [H,W]; either may be NULL): lo - hi and its float32 cast. */
size_t mh_dog_scratch_bytes(int H, int W);
int mh_dog(mh_ctx *ctx, const void *image, int in_kind, int H, int W, const double *w_lo, int r_lo, const double *w_hi,
           int r_hi, void *scratch, double *out64, float *out32, void *stream);
/* One view of the Gabor stage, device to device (calculate_orientation, GaborFilter.py:164-224, without the file IO):
 * gray uint8 [H,W] -> DoG -> bank -> orient_index / conf / variance as mh_gabor_bank, plus (k8, c8 non-NULL) the two 8-bit
 * codes the reference writes to best_ori/<view> and conf/<view> (orientation in degrees; floor(conf*255+0.5)) -- what the
 * PMVO loaders read back (mh_ctx_set_view_u8).  conf may be NULL.  scratch: mh_gabor_view_scratch_bytes(H, W). */
size_t mh_gabor_view_scratch_bytes(int H, int W);
int mh_gabor_view(mh_ctx *ctx, const uint8_t *gray, int H, int W, const double *w_lo, int r_lo, const double *w_hi, int r_hi,
                  void *scratch, int32_t *orient_index, float *conf, float *variance, uint8_t *k8, uint8_t *c8, void *stream);

/* ---- strand tracing on the fitted volume (HairGrow.py).
 * mh_volume_pack: occ[Z,H,W] + ori[Z,H,W,3] (the .mat readers' layout, Utils/PMVO_utils.py:86-113) -> 16-byte voxels
 *   {ori_x, -ori_y, -ori_z, occ} (HairGrowing.__init__, HairGrow.py:41-55); vox: W*H*Z*16 bytes.
 * mh_trace_seeds: HairGrowing.trace (:59-149) for n already-jittered seeds, WITHOUT the flag gate: strand i occupies
 *   out[i][first[i] .. first[i]+len[i]) of a 513-point row.
 * mh_trace_scalp: HairGrowing.traceFromScalp (:154-223): rows of 257 points, len 0 where the reference returns None.
 * mh_strands_accept: the sequential flag gate of GenerateGuideStrandFromScalp (:226-265) / randomlyGenerateSegments
 *   (:269-299) replayed over the finished traces -- HOST pointers, runs on the calling thread. */
int mh_volume_pack(mh_ctx *ctx, const float *occ, const float *ori, int W, int H, int Z, void *vox, void *stream);
int mh_trace_seeds(mh_ctx *ctx, const void *vox, int W, int H, int Z, const float *seeds, int n, float thr_dot,
                   float *out, int32_t *first, int32_t *len, void *stream);
int mh_trace_scalp(mh_ctx *ctx, const void *vox, int W, int H, int Z, const float *seeds, const float *normals, int n,
                   float thr_dot, float *out, int32_t *len, void *stream);
int mh_strands_accept(int W, int H, int Z, float *flag, const float *pts, const int32_t *first, const int32_t *len,
                      int stride, const float *seeds, int n, int mode, uint8_t *accepted);
/* Packs the fixed-stride rows of mh_trace_seeds / mh_trace_scalp (device) into one point list, strand after strand:
 * packed[offsets[i] .. offsets[i]+len[i]) = rows[i][first[i] .. ) (first may be NULL = 0); offsets = exclusive prefix sum
 * of len (device, int64).  mh_strands_accept reads the packed list with stride 0 and first = offsets. */
int mh_strands_compact(mh_ctx *ctx, const float *rows, const int32_t *first, const int32_t *len, const long long *offsets,
                       int n, int stride, float *packed, void *stream);

/* ---- the one exchange of the data path, RCCL over xGMI.  The reference has no multi-GPU path
 * (options.py:112 asserts a single GPU); the voxel fit of refine (PMVO.py:695-726) is sharded here by x-slabs of
 * the volume, every rank fitting the voxels of its slab into a zero-initialised dense [X,Y,Z,C] fp32 volume (C = 4:
 * occupancy + orientation), and rank `root` assembles the shared volume:
 *   mode 0: slab gather -- ownership is disjoint, so every peer ncclSend's its slab straight to the root, which
 *           ncclRecv's it in place ((nranks-1)/nranks of the volume crosses xGMI, one slab per link);
 *   mode 1: dense ncclReduce(sum) of the whole volume (x + 0 is exact: the same result; kept for comparison).
 * slab_host (host, nranks+1 ints): rank r owns x in [slab_host[r], slab_host[r+1]).  `comm` is an ncclComm_t:
 * mh_comm_init makes one from a 128-byte ncclUniqueId that rank 0 obtains with mh_comm_unique_id and hands to the
 * other ranks by any means (torch.distributed broadcast, a file, MPI); any ncclComm_t of the caller works as well.
 * librccl.so.1 is bound at run time; a process that never calls these does not load it. */
int mh_comm_unique_id(void *id_out_host /*128 bytes*/);
int mh_comm_init(mh_ctx *ctx, const void *id_host /*128 bytes*/, int nranks, int rank, void **comm_out);
int mh_comm_destroy(void *comm);
int mh_volume_reduce(mh_ctx *ctx, void *comm, int rank, int nranks, int root, float *volume /*[X,Y,Z,C] in place*/,
                     int X, int Y, int Z, int C, const int32_t *slab_host, int mode, void *stream);
/* The slab gather with slab-sized buffers on the peers (what monohair_amd.dist.voxel_fit_reduced uses): `slab` is this
 * rank's own x-slab [slab_host[rank+1]-slab_host[rank], Y, Z, C] (device, contiguous; may be NULL when the slab is empty),
 * `volume` the dense [X,Y,Z,C] volume on the root only (NULL elsewhere).  The root's own slab is copied into place on the
 * stream unless slab == volume + slab_host[root]*Y*Z*C.  Wire traffic = mode 0 above; a peer holds 1/nranks of the volume.
 * The environment variable MH_RCCL_LIB=<path> binds another library for these entry points (e.g. tests/fake_rccl.cpp, a
 * stand-in built against rccl.h that lets several ranks share one GPU: how the nranks > 1 branches are tested on a
 * one-GPU box); an unloadable path is an error. */
int mh_volume_gather(mh_ctx *ctx, void *comm, int rank, int nranks, int root, const float *slab, float *volume, int X,
                     int Y, int Z, int C, const int32_t *slab_host, void *stream);

/* ---- host-side IO of the volume files: scipy.io.savemat of PMVO.py:753-764 writes dense float64 arrays that are zero
 * except at the occupied voxels.  Creates `path` = prefix (the MAT-v5 header + array tags, built by the caller) + a
 * zero-filled payload of payload_bytes, then stores values[i] at payload element elem_index[i] (float64 elements; later
 * entries win on duplicates).  The file is sparse on disk; the scatter runs on `threads` host threads split by
 * destination range.  Host pointers only; no GPU involved. */
int mh_mat_write_sparse(const char *path, const void *prefix, size_t prefix_bytes, size_t payload_bytes,
                        const long long *elem_index, const double *values, size_t n, int threads);
/* The same file in steps (open -> touch -> store -> close): `touch` makes the pages of the given elements resident without
 * changing them, so a background thread can take the page faults of the zero-filled mapping (16-19 ms for the two volume files
 * of a pass) early -- with every candidate point's voxel, a superset of what can become occupied -- while the GPU still works;
 * `store` writes the occupied elements (later entries win).  The finished file is byte for byte what mh_mat_write_sparse and
 * scipy's dense savemat write (PMVO.py:753-764). */
int mh_mat_sparse_open(const char *path, const void *prefix, size_t prefix_bytes, size_t payload_bytes, void **handle);
int mh_mat_sparse_touch(void *handle, const long long *elem_index, size_t n);
int mh_mat_sparse_store(void *handle, const long long *elem_index, const double *values, size_t n);
/* store from the voxel list of the fit: vox [G,3] (x,y,z) -> element y + Y*(x + X*z) (+ c*X*Y*Z for channel c of Ori);
 * ori == NULL writes 1.0 (Occ), else ori [G,3] float32 (ori_is_f64 = 0) or float64; later rows win */
int mh_mat_sparse_store_voxels(void *handle, const long long *vox, const void *ori, int ori_is_f64, size_t G, int X, int Y,
                               int Z);
int mh_mat_sparse_close(void *handle);

/* Supported options of a context (mh_ctx_set_option): how results are rounded and which shipped form computes them.
 * Anything else is an argument error; the A/B forms and cross-check kernels the tests and bench.py switch between are NOT
 * here but in include/mh_pmvo_lab.h (mh_ctx_set_lab_option).
 *   "reproject_rule": how the sgemms of PMVO.sample_next_3d_pos (Camera.projection / Camera.reprojection of the points that
 *       share a base view, PMVO.py:289,318 -> Utils/Camera_utils.py:50-53,103) round.  0 (default): by the number M of points
 *       of the batch that share the (rank, base view), as MKL does in the reference -- M == 1: single-column projection;
 *       S*M >= "reproject_fma_min_cols" or S*M <= 3: k-ordered fma chain; otherwise separately rounded products.  1: the
 *       mid-size forms for every point (rounds 1-4; independent of the batch).  2: the chain forms for every point.
 *   "reproject_fma_min_cols": the column count from which MKL's sgemm switches to its threaded (fma chain) kernel ON THE
 *       HOST THE REFERENCE RUNS ON -- it moves with that host's thread count: 8 threads 28445 (default: MKL 2024.2, AVX-512,
 *       where the goldens were generated), 4 threads 14223, 2 threads 21334, 1 thread never (2147483647).
 *       `python tools/probe_mkl_forms.py --emit-options` prints the value for a host; PMVO.py takes it as
 *       --PMVO.reference_host=<json>; pinned end to end at 1 / 2 / 4 / 8 threads (tests/golden/pmvo_threads.npz).
 *   "sum_block": 32 (default) ATen's sum(dim=0) adds the trailing (columns mod 32) of a [V, N*S] / [V, N] sum in its
 *       row_sum order (forward: the last samples of the last point of a batch; refine / filter votes: the last points of a
 *       batch), and the [V, 1] sums of a batch of ONE point in its inner-sum order; 0: cascade order everywhere (rounds 1-4).
 *   "topk_order": see mh_topk_views (0 = torch.topk's order among equal confidences, default; 1 = view order).
 *   "gabor_variant": 3 (default) mh_gabor_mfma2_kernel (FP32 MFMA); 0 the direct v_pk_fma kernel.  Same results.
 *   "tap_plane_max_mb" (default 4096; environment MH_TAP_PLANE_MAX_MB at context creation): contexts with fp32 views keep
 *       every pixel once more as a ready-made patch tap (unit orientation, clamped confidence: 16 B per pixel, +1.2 %
 *       iterations/s) only if that plane is at most this large and leaves a quarter of the free device memory: 60 x 1080p
 *       (1 991 MB) gets it, 120 x 4K (15 925 MB) does not unless the budget is raised.  Takes effect before the first view.
 *   "line_rule" (mh_render_strands): 0 = OpenGL's diamond-exit rule (default), 1 = the pixel that holds a segment's end
 *       point is drawn too (what Google SwiftShader does; changes the image: used to compare with that GL).
 *   "raster_subpixel_bits" (mh_render_depth, mh_render_strands): window positions are snapped to 2^-bits pixel, 4..8,
 *       default 8; OpenGL requires at least 4, which is what SwiftShader uses (changes the image at silhouettes). */
int mh_ctx_set_option(mh_ctx *ctx, const char *key, int value);

#ifdef __cplusplus
}
#endif
#endif /* MH_PMVO_H */
