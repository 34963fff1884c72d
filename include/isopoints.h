/*
 * isopoints.h -- C ABI of the MI355X (gfx950) iso-point hot path.
 *
 * One shared library (libisopoints_hip.so), plain pointers + sizes, no torch
 * types.  Every pointer is a DEVICE pointer unless its name ends in `_host`.
 * Every entry point enqueues work on `stream` (a hipStream_t passed as void*,
 * NULL = default stream) and returns without synchronising, except where the
 * comment says "syncs".  Return value: ISO_OK (0) or a negative ISO_ERR_*; the
 * message for the last error of the calling thread is iso_last_error().
 * Nothing is allocated inside the library: outputs and workspaces are caller
 * owned (the `*_bytes` helpers size them).  Thread-safe per stream.
 *
 * Each entry point cites the reference interface it stands in for
 * (paths relative to the yifita/iso-points checkout).
 */
#ifndef ISOPOINTS_H_
#define ISOPOINTS_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ISO_OK 0
#define ISO_ERR_INVALID (-1)     /* bad argument (shape, null pointer, range) */
#define ISO_ERR_UNSUPPORTED (-2) /* valid but outside what the kernels cover   */
#define ISO_ERR_LAUNCH (-3)      /* HIP launch / runtime failure               */
#define ISO_ERR_WORKSPACE (-4)   /* caller workspace too small                 */

const char* iso_version(void);
const char* iso_last_error(void);

/* ------------------------------------------------------------------------
 * A. Newton level-set projection
 *    replaces UniformProjection._project_points + _compute_sdf_and_grad
 *    (DSS/models/levelset_sampling.py:290-351, :142-170).
 *
 *    Per point (points are independent, so the reference's boolean-mask
 *    compaction is an in-register `active` flag here):
 *      repeat: f,g = SDF(p), grad SDF(p); normal = g;
 *              if |f| <= tol -> converged, stop; if it == max_iters -> stop;
 *              m = f*g/sdeno(|g|^2,1e-17); m = m/max(|m|,1e-15)*min(|m|,0.1);
 *              p -= m
 *    points (n,3) f32 packed; normals_out (n,3) = last evaluated gradient (NOT
 *    normalised); mask_out (n) u8 = converged.  In-place (pts_out == pts_in)
 *    is allowed.
 * ---------------------------------------------------------------------- */

/* analytic SDF |x-c| - radius (BASELINE.json configs[0], SURVEY 8(d) cfg 1/3a) */
int iso_project_sphere(const float* pts_in, float* pts_out, float* normals_out,
                       uint8_t* mask_out, int64_t n, float cx, float cy,
                       float cz, float radius, int max_iters, float tol,
                       void* stream);

/* What a projection launch does ON THE SIDE for the stages that consume its result in the iso-point cycle, so that they
 * need no pass of their own over the points it has just written (all pointers device memory):
 *   - the bounding box of the projected points is added to the PENDING BOX of the brick workspace grid_ws (section E;
 *     required, initialised).  iso_bricks_build_pending then makes the grid's header from it -- no box pass, no
 *     launch in between; iso_bricks_box_take returns it as 8 floats (what N ranks all-gather) and clears it.
 *   - views != NULL: the renderable mask of every point (iso_splat_view_mask's rule; normals = the projection's own
 *     normals_out) -> mask_out, and the renderable points per 256-point tile and view -> front_ws
 *     (iso_splat_front_workspace_bytes).  The scan of iso_splat_view_mask_scan (chunk table in front_ws, first_idx_out,
 *     num_pts_out, view_total_out) is done by iso_bricks_build_pending when it is handed the same struct -- the grid
 *     it builds carries the mask as payload, so the two always go together in the cycle.
 * Reference: the consumers are UniformProjection._create_tree (levelset_sampling.py:110-140: bbox of the cloud),
 * SurfaceSplatting._filter_points_with_invalid_depth / backface culling (rasterizer.py:184-254).          */
typedef struct iso_follow {
  void* grid_ws; int64_t grid_n_max;
  const float* views; int n_views; float znear; float zfar; int backface_culling;
  int32_t* mask_out; void* front_ws; int64_t front_ws_bytes;
  int64_t* first_idx_out; int64_t* num_pts_out; int32_t* view_total_out;
} iso_follow;

/* iso_project_sphere + the side work described by *follow (a HOST struct, read during the call). */
int iso_project_sphere_follow(const float* pts_in, float* pts_out, float* normals_out,
                              uint8_t* mask_out, int64_t n, float cx, float cy,
                              float cz, float radius, int max_iters, float tol,
                              const iso_follow* follow_host, void* stream);

/* SIREN SDF (DSS/models/common.py:90-165): dims 3 -> H -> (H)*n_hidden -> 1,
 * h0 = sin(w0*(W0 x+b0)), hi = sin(w*(Wi h+bi)), sdf = WL h + bL.
 * Weights are handed over exactly as torch stores them (row-major
 * [out][in]) in ONE packed f32 buffer:
 *   W0[H*3] b0[H]  { Wi[H*H] bi[H] } * n_hidden   WL[H] bL[1]
 * iso_siren_pack_weights() re-orders the hidden matrices into the MFMA
 * lane-linear images the projection kernel streams (forward + transposed).
 * H must be a multiple of 16, 16 <= H <= 512; 0 <= n_hidden <= 8.            */
int64_t iso_siren_raw_floats(int hidden, int n_hidden);
int64_t iso_siren_packed_floats(int hidden, int n_hidden);
int iso_siren_pack_weights(const float* raw, float* packed, int hidden,
                           int n_hidden, void* stream);
/* How the H x H products of the hidden layers are formed (process-wide switch):
 *   1 (default)  fp16 matrix cores at f32 accuracy: every f32 operand is cut into two fp16
 *                numbers (11 + 11 significant bits) under an exact power-of-two scale -- per layer
 *                for the weights, 2^12 for the activations, per point for the adjoint of the reverse
 *                sweep -- and three partial products are accumulated in f32: same error level as an
 *                f32 GEMM (the accumulation dominates), 5.3x the MFMA rate.  H in {128,256},
 *                n_hidden >= 1.
 *   0            f32 matrix cores (v_mfma_f32_16x16x4_f32; bitwise an fmaf chain).
 * Shapes mode 1 does not cover run in mode 0.  ISO_SIREN_GEMM=f32 in the environment selects 0. */
int iso_siren_set_gemm_mode(int mode);
int iso_siren_get_gemm_mode(void);
/* Newton tail of iso_project_siren (H = 256, split-fp16 mode): the iterations from `first_tail_iteration` on run as ONE
 * launch whose workgroups iterate the survivors of their own tiles until none is left (the reference leaves its loop
 * when nothing is active, levelset_sampling.py:329) -- instead of one launch per iteration, most of them for a few
 * hundred points or none.  -1: the default (4 for max_iters >= 6, else 2), 0: never (a launch per iteration), k >= 1.
 * Results do not depend on it (a point's evaluation does not depend on the tile it sits in).              */
int iso_siren_set_tail_from(int first_tail_iteration);
/* Which tile a workgroup of the step kernels takes next: drawn from a per-launch counter (1, the default: the XCDs of one
 * GPU do not run alike under the power cap, equal static shares end with the slowest) or every gridDim-th (0).  -1: back to
 * the environment (ISO_SIREN_DYN_TILES / ISO_IDR_DYN_TILES = 0) or the default.  Results do not depend on it. */
int iso_siren_set_drawn_tiles(int on);
int iso_idr_set_drawn_tiles(int on);
/* step-kernel launches one iso_project_siren call with these arguments issues (max_iters + 1 without the tail) */
int iso_siren_step_launches(int hidden, int n_hidden, int max_iters);
/* scratch for the per-wave activation-derivative stash */
int64_t iso_project_siren_workspace_bytes(int64_t n, int hidden, int n_hidden);
/* diagnostics: byte offset, inside that workspace, of the 64 int32 counters a projection leaves behind -- [it] = points
 * evaluated by iteration it (it >= 1; iteration 0 evaluates all n) */
int64_t iso_project_siren_counts_offset(int64_t n, int hidden, int n_hidden);
int iso_project_siren(const float* pts_in, float* pts_out, float* normals_out,
                      uint8_t* mask_out, int64_t n, const float* packed,
                      int hidden, int n_hidden, float omega_first,
                      float omega_hidden, int max_iters, float tol,
                      void* workspace, int64_t workspace_bytes, void* stream);
/* one SDF + gradient evaluation, no Newton move (used by tests and by callers
 * that only need _compute_sdf_and_grad, levelset_sampling.py:142-170).
 * grad_out may be NULL: value only, the reverse sweep is skipped (the `sdf` callable of
 * RayTracing, levelset_sampling.py:831-1167, and the candidate evaluations of
 * combined_modeling.py:376-380).                                              */
int iso_siren_sdf_grad(const float* pts, float* sdf_out, float* grad_out,
                       int64_t n, const float* packed, int hidden,
                       int n_hidden, float omega_first, float omega_hidden,
                       void* workspace, int64_t workspace_bytes, void* stream);

/* IDR-style SDF (DSS/models/common.py:220-310): positional encoding with n_freq
 * frequencies (D0 = 3 + 6*n_freq <= 63), n_layers softplus(beta) layers of width `hidden`
 * (128/256/512), optional skip connection [h, e(x)]/sqrt(2) into layer skip_layer (< 0: none;
 * layer skip_layer-1 is then hidden-D0 wide), linear head, tanh.  Weight-norm must already be
 * folded into the weights.  raw = for l = 0..n_layers: W_l (row-major [out][in]) then b_l.
 * hidden 256 / 512 with n_freq <= 6 run on the fp16 matrix cores at f32 accuracy (two fp16 parts
 * per operand under exact power-of-two scales, as iso_siren_set_gemm_mode(1)); ISO_IDR_GEMM=f32 in
 * the environment, hidden 128 and wider encodings use the f32 matrix cores.                     */
int64_t iso_idr_raw_floats(int hidden, int n_layers, int skip_layer, int n_freq);
int64_t iso_idr_packed_floats(int hidden, int n_layers);
int iso_idr_pack_weights(const float* raw, float* packed, int hidden, int n_layers,
                         int skip_layer, int n_freq, void* stream);
int64_t iso_project_idr_workspace_bytes(int64_t n, int hidden, int n_layers);
int iso_project_idr(const float* pts_in, float* pts_out, float* normals_out, uint8_t* mask_out,
                    int64_t n, const float* packed, int hidden, int n_layers, int skip_layer,
                    int n_freq, float beta, int max_iters, float tol, void* workspace,
                    int64_t workspace_bytes, void* stream);
/* grad_out may be NULL (value only, forward sweep only). */
int iso_idr_sdf_grad(const float* pts, float* sdf_out, float* grad_out, int64_t n,
                     const float* packed, int hidden, int n_layers, int skip_layer, int n_freq,
                     float beta, void* workspace, int64_t workspace_bytes, void* stream);

/* Sphere tracing along given rays: SphereTracing.project_points,
 * DSS/models/levelset_sampling.py:679-808 (callers implicit_modeling.py:305,313).
 *   p_0 = ray0; per iteration (max_iters advances, max_iters + 1 evaluations):
 *     f = sdf(p)                                       value only -- the gradient the reference
 *                                                      computes at :745-757 is never used (:806)
 *     still active: |f| > 0.1 * tol  and  inside       (:764)
 *     m = (alpha f) d;  m = m / max(|m|,1e-15) * min(|m|,0.1);  q = p + m        (:771-775)
 *     inside = |q| < bound (= radius + padding);  p = q only while inside        (:776-779)
 *   outputs: pts_out (n,3) the last position inside the bounding sphere, sdf_out (n) the value at
 *   it (`network_eval_on_levelset_points`), mask_out (n) = |sdf| <= tol (:790).
 *   dirs (n,3) are the (normalised) ray directions.  One launch per iteration over a device-side
 *   active list, as iso_project_*; workspaces are those of iso_project_siren / iso_project_idr. */
int iso_trace_sphere(const float* ray0, const float* dirs, float* pts_out, float* sdf_out,
                     uint8_t* mask_out, int64_t n, float cx, float cy, float cz, float radius,
                     float alpha, float bound, int max_iters, float tol, void* stream);
int iso_trace_siren(const float* ray0, const float* dirs, float* pts_out, float* sdf_out,
                    uint8_t* mask_out, int64_t n, const float* packed, int hidden, int n_hidden,
                    float omega_first, float omega_hidden, float alpha, float bound, int max_iters,
                    float tol, void* workspace, int64_t workspace_bytes, void* stream);
int iso_trace_idr(const float* ray0, const float* dirs, float* pts_out, float* sdf_out,
                  uint8_t* mask_out, int64_t n, const float* packed, int hidden, int n_layers,
                  int skip_layer, int n_freq, float beta, float alpha, float bound, int max_iters,
                  float tol, void* workspace, int64_t workspace_bytes, void* stream);

/* Nearest point to each ray (brute force, fused): the (R,M) point-to-ray search of
 * CombinedModel.sample_offsurface_using_isopoints, DSS/models/combined_modeling.py:336-352.
 *   pC = p - origin;  ray_sq = (pC . ray)^2;  dist = |pC|^2 - ray_sq;  nn = argmin_m dist
 *   idx_out (R) i32 = nn (lowest index among equal distances; -1 when there are no points),
 *   raysq_out (R) = ray_sq[r, nn] (the reference's `ray_len` before eps_sqrt().sqrt(), :345,:353),
 *   dist_out (R, may be NULL) = dist[r, nn].  rays (R,3) unit directions, points (M,3).       */
int64_t iso_ray_nearest_point_workspace_bytes(int64_t n_rays);
int iso_ray_nearest_point(const float* rays, int64_t n_rays, float ox, float oy, float oz,
                          const float* points, int64_t n_points, int32_t* idx_out,
                          float* raysq_out, float* dist_out, void* workspace,
                          int64_t workspace_bytes, void* stream);

/* State updates of IDR's two-ended ray marcher between two network evaluations
 * (RayTracing.sphere_tracing / .secant, DSS/models/levelset_sampling.py:920-1032, :1114-1133).
 * Per ray two marching ends: arrays of shape (2,R) hold end 0 (forward from the bounding-sphere
 * entry) in [0,R) and end 1 (backward from the exit) in [R,2R).  cam, dirs (R,3) are the ray
 * origin and unit direction.  Each call writes `list` (<= 2R,3): the points that need a network
 * value next, `slot` (2,R) i32: their position in the list (-1: none), and `*count` (device i32):
 * how many there are -- the caller reads it, evaluates list[0:count] and passes the values on.
 *
 * iso_raymarch_settle (:962-990 and, with check_order, the z0 < z1 test of :1027-1030 that ends
 *   the previous iteration): cur = live ? nxt : 0;  cur <= thr -> 0;  live &= cur > thr;
 *   *count = unfinished ends; when `step`: z0 += cur0, z1 -= cur1, list <- cam + z d of those ends.
 * iso_raymarch_overshoot (:993-1025): nxt <- values (first: ends without a slot get 0; later:
 *   they keep theirs); ends with nxt < 0 stepped through the surface: when `may_backstep`
 *   z0 -= back cur0 / z1 += back cur1, list <- their new points, *count = how many (else 0).
 * iso_raymarch_secant (:1114-1133) on n compacted rays: with f_mid = values at z_pred, move the
 *   bracket end of matching sign (f_mid > 0 -> lo, < 0 -> hi); then (also when f_mid is NULL: the
 *   first call) z_pred = -f_lo (z_hi - z_lo) / (f_hi - f_lo) + z_lo, pts_out = cam + z_pred d.  */
int iso_raymarch_settle(const float* cam, const float* dirs, int64_t n_rays, float* z, float* cur,
                        const float* nxt, uint8_t* live, float sdf_threshold, int check_order,
                        int step, int32_t* slot, float* list, int32_t* count, void* stream);
int iso_raymarch_overshoot(const float* cam, const float* dirs, int64_t n_rays, float* z,
                           const float* cur, float* nxt, const float* values, int first,
                           int may_backstep, float back, int32_t* slot, float* list,
                           int32_t* count, void* stream);
int iso_raymarch_secant(const float* cam, const float* dirs, int64_t n, float* f_lo, float* f_hi,
                        float* z_lo, float* z_hi, float* z_pred, const float* f_mid,
                        float* pts_out, void* stream);

/* Image values at projected points: get_tensor_values, DSS/utils/__init__.py:325-375
 * (= grid_sample(image (B,C,H,W), p, mode, padding_mode='reflection', align_corners=False),
 * squeezed and permuted).  p (B,n,2) in [-1,1] (x, y); out (B,n,C).
 * mode 0 bilinear, 1 nearest, 2 integer indexing trunc((p + 1)(size - 1)/2) (`grid_sample=False`,
 * :357-363; an index outside the image -- an IndexError in the reference -- gives NaN).          */
int iso_image_sample(const float* image, int batch, int channels, int height, int width,
                     const float* p, int64_t n, int mode, float* out, void* stream);

/* ------------------------------------------------------------------------
 * B. Fixed-radius nearest neighbours on a uniform grid
 *    replaces the third-party `frnn` / `prefix_sum` extensions the reference
 *    calls (lxxue/FRNN@eab337f, not vendored):
 *      frnn.frnn_grid_points  levelset_sampling.py:132,182,200
 *                             point_processing.py:74,84,145,179,253
 *                             rasterizer.py:371
 *      frnn.frnn_gather       levelset_sampling.py:213,268-271
 *      frnn._C.insert_points_cuda / counting_sort_cuda   rasterizer.py:909,921
 *      prefix_sum.prefix_sum_cuda                        rasterizer.py:915
 *
 *    Grid parameter block (f32), same layout the reference's kernels read
 *    (DSS/csrc/rasterize_points_backward.cu:10-28):
 *      3-D: [min_x,min_y,min_z, 1/cell, res_x,res_y,res_z, total]  (8 floats)
 *      2-D: [min_x,min_y,       1/cell, res_x,res_y,       total]  (6 floats)
 *    cell index = (x*res_y + y)*res_z + z   (2-D: x*res_y + y).
 * ---------------------------------------------------------------------- */
#define ISO_GRID3_PARAMS 8
#define ISO_GRID2_PARAMS 6
#define ISO_GRID_MAX_RES 256
/* upper bound of `total` for a grid built by iso_frnn_make_grid with max_res = ISO_GRID_MAX_RES */
#define ISO_GRID3_MAX_CELLS ((ISO_GRID_MAX_RES + 1) * (ISO_GRID_MAX_RES + 1) * (ISO_GRID_MAX_RES + 1))

/* Axis-aligned bounding box of each cloud: minmax (N,8) f32 =
 * [min_x,min_y,min_z,0, max_x,max_y,max_z,0] (zeros for an empty cloud).  Feeds the
 * search radius / kernel width of levelset_sampling.py:129-131 and :254-256 without
 * torch's dim-reductions or a host sync.                                         */
int iso_points_bbox(const float* points, const int64_t* lengths, int n_clouds,
                    int64_t p_stride, float* minmax, void* stream);

/* Device-side grid sizing for clouds `points` (N, P, 3) padded, lengths (N)
 * i64 (NULL = all P), radius (N) f32.  Writes params (N, 8).  Cell size is
 * chosen from point density (about 8 points per occupied cell on a surface)
 * but never finer than r/2 or (max extent)/max_res; the query walks Chebyshev
 * rings of cells so the result does not depend on the cell size.  No host
 * sync: the caller picks max_res (1..ISO_GRID_MAX_RES; 128 is plenty below ~250 k
 * points, 1 M points on a surface want 256) and allocates the grid arrays for
 * (max_res+1)^3 cells.                                                          */
int iso_frnn_make_grid(const float* points, const int64_t* lengths,
                       const float* radius, int n_clouds, int64_t p_stride,
                       int max_res, float* grid_params, void* stream);
/* The same with the density target spelled out: about `points_per_cell` points per occupied cell
 * on a surface (iso_frnn_make_grid = 8).  An exact K-nearest search (r = inf) finishes inside the
 * 3x3x3 cell neighbourhood -- the fast path of the query -- when the K-th neighbour is closer
 * than one cell; measured optimum points_per_cell ~ 0.75 K (K = 32, 150 k points on a surface:
 * 0.9 ms instead of 3.7 ms); results do not depend on the choice.                           */
int iso_frnn_make_grid_density(const float* points, const int64_t* lengths, const float* radius,
                               int n_clouds, int64_t p_stride, int max_res, float points_per_cell,
                               float* grid_params, void* stream);

/* frnn._C.insert_points_cuda: cell id and arrival slot of each point;
 * cnt (N,G) must be zero on entry.  dim = 2 or 3 (points have `dim` floats).  */
int iso_frnn_insert_points(const float* points, const int64_t* lengths,
                           const float* grid_params, int32_t* cnt,
                           int32_t* cell, int32_t* idx_in_cell, int n_clouds,
                           int64_t p_stride, int64_t g_stride, int dim,
                           void* stream);
/* prefix_sum.prefix_sum_cuda: exclusive scan of `n` i32 counts (in != out or
 * in == out).  `n` is read on the host; batch = independent rows.
 * workspace: iso_prefix_sum_workspace_bytes(n or g_stride, batch).           */
int64_t iso_prefix_sum_workspace_bytes(int64_t n, int batch);
int iso_prefix_sum(const int32_t* in, int32_t* out, int64_t n, int batch,
                   int64_t row_stride, void* workspace, int64_t workspace_bytes,
                   void* stream);
/* as iso_prefix_sum, but the row length is read ON DEVICE from
 * grid_params[total] and rows are scanned up to g_stride at most             */
int iso_frnn_scan_cells(const int32_t* cnt, int32_t* off,
                        const float* grid_params, int n_clouds,
                        int64_t g_stride, int dim, void* workspace,
                        int64_t workspace_bytes, void* stream);
/* frnn._C.counting_sort_cuda: scatter points into cell order.
 * sorted_points (N,P,dim), sorted_idx (N,P) i32 = original index.            */
int iso_frnn_counting_sort(const float* points, const int64_t* lengths,
                           const int32_t* cell, const int32_t* idx_in_cell,
                           const int32_t* off, float* sorted_points,
                           int32_t* sorted_idx, int n_clouds, int64_t p_stride,
                           int64_t g_stride, int dim, void* stream);

/* Radius-K query (frnn.frnn_grid_points proper).  For every query point
 * q in points1 (N,P1,3): the K nearest points of cloud 2 with
 * d2 = (dx*dx + dy*dy) + dz*dz < r*r (f32, no FMA contraction), ascending by
 * (d2, original index); unused slots are -1 (idx) / -1.0 (dists).
 * idxs_out is i64 (pytorch3d convention, levelset_sampling.py:136).
 * nn_out (N,P1,K,3) may be NULL (return_nn=False); when given, points2
 * (N,P2,3, original order) must be given too.
 * points1 == NULL means "cloud 2 queried against itself": queries are then
 * processed in cell order (neighbouring lanes walk the same cells) and rows
 * are written at their original index.
 * sorted2 / sorted_idx2 / off come from the build calls above.
 * workspace (iso_frnn_query_workspace_bytes, 16-B aligned): list of the queries a single lane
 * could not finish within two rings of cells (a second kernel serves each of them with a whole
 * wave), and the per-call candidate records (x,y,z,index: one 16-B load per candidate).        */
int64_t iso_frnn_query_workspace_bytes(int n_clouds, int64_t p1_stride, int64_t p2_stride);
int iso_frnn_query(const float* points1, const int64_t* lengths1,
                   const float* points2, const float* sorted2,
                   const int32_t* sorted_idx2,
                   const int64_t* lengths2, const int32_t* off,
                   const float* grid_params, const float* radius, int K,
                   float* dists_out, int64_t* idxs_out, float* nn_out,
                   int n_clouds, int64_t p1_stride, int64_t p2_stride,
                   int64_t g_stride, void* workspace, int64_t workspace_bytes,
                   void* stream);

/* frnn.frnn_gather: out[n,i,k,:] = x[n, idx[n,i,k], :], zeros where idx < 0.
 * x (N,P2,U) f32, idx (N,P1,K) i64, out (N,P1,K,U).                           */
int iso_frnn_gather(const float* x, const int64_t* idx, float* out,
                    int n_clouds, int64_t p1, int64_t p2, int K, int U,
                    void* stream);

/* ------------------------------------------------------------------------
 * C. Tangent-plane repulsion step
 *    replaces the body of UniformProjection.resample
 *    (DSS/models/levelset_sampling.py:268-284), single cloud:
 *      n_j = normalize(normals[j]) ; D = p_i - p_j ; w = exp(-|D|^2 * inv_sigma)
 *      Dt = D - (D.n_j) n_j ; move = (sum w + 1) * sum(w Dt) / sdeno(sum w)
 *      out_i = p_i + move_i
 *    Moves the n points [first_point, first_point+n) of `points` (rows 0..n-1 of idx and
 *    points_out belong to them; neighbour indices address the whole array) -- a rank of
 *    a sharded run passes its own slice, a single-GPU caller first_point = 0, n = P.
 *    idx: row i starts at idx + i*idx_row_stride (i64, K entries, -1 padding:
 *    weight 0 there) -- lets the caller pass the [:,1:] view of a (P,K+1)
 *    query result without copying.  inv_sigma (= P/diag, :256) is a DEVICE
 *    scalar so the bounding-box reduction needs no host sync.                 */
int iso_repulse(const float* points, const float* normals, const int64_t* idx,
                int64_t idx_row_stride, float* points_out, int64_t n,
                int64_t first_point, int K, const float* inv_sigma, void* stream);

/* UniformProjection.insert (DSS/models/levelset_sampling.py:172-233) on the device.
 * iso_insert_fathers: father[b][i] = 1 when the nearest of the n_refs (device int32, <= 64) selected reference
 *   points lies within the K = 1 query radius and 0 < d^2 < 4 spacing^2 (:200-206); params (device, 2 floats) =
 *   {(4 r)^2, 4 spacing^2}.
 * iso_insert_children: child = 2 father / 3 + neighbour / 3 over the LAST `patch` columns of knn_idx (:207-209),
 *   cloud b written from row out_first[b], fathers in ascending order (rank_inclusive = inclusive prefix sum of
 *   father over each cloud); neighbour index < 0 = the origin (what frnn_gather returns there).             */
int iso_insert_fathers(const float* points, const int64_t* lengths, int n_clouds, int64_t max_points,
                       const float* refs, const int32_t* n_refs, const float* params, uint8_t* father_out,
                       void* stream);
int iso_insert_children(const float* points, const int64_t* knn_idx, int n_clouds, int64_t max_points, int K,
                        int patch, const uint8_t* father, const int64_t* rank_inclusive, const int64_t* out_first,
                        float* children_out, void* stream);
/* Sparsest-edge candidates of point_processing.upsample
 * (DSS/utils/point_processing.py:326-339): per point p with neighbours nn_k (knn (n,K,3)):
 * mid_k = (nn_k + 2p)/3, spars_k = min_j |mid_k - nn_j|, sparsity = max_k spars_k,
 * candidate = mid_argmax (first maximum).  K <= 64.                               */
int iso_upsample_candidates(const float* points, const float* knn, int64_t n, int K,
                            float* sparsity_out, float* candidates_out, void* stream);

/* Edge-aware candidates of EdgeAwareProjection.upsample (DSS/models/levelset_sampling.py:609-628):
 * per point p (unit normal n) with neighbours nn_k (knn (n,K,3)) of unit normals u_k (knn_normals):
 * mid_k = (nn_k + 2p)/3, d_kj = mid_k - nn_j,
 * m_k = sqrt(max(|min_j (|d_kj| - sum_c (d_kj,c u_k,c)^2)|, 1e-17)), edge_k = (2 - n.u_k)^edge_sensitivity,
 * sparsity = max_k edge_k m_k, candidate = mid_argmax (first maximum).  K <= 64.             */
int iso_ear_candidates(const float* points, const float* normals, const float* knn,
                       const float* knn_normals, int64_t n, int K, float edge_sensitivity,
                       float* sparsity_out, float* candidates_out, void* stream);

/* Farthest-point sampling = torch_cluster.fps as wlop uses it
 * (DSS/utils/point_processing.py:473-499): per cloud n, out_idx[n, 0..n_samples[n]) =
 * start[n], then repeatedly the point farthest from the chosen set (squared f32 distances,
 * ties -> lowest index).  work: iso_farthest_point_sampling_work_floats(N, p_stride) f32 of
 * scratch.  out_idx rows are left untouched beyond n_samples[n].  Clouds of >= 8 k points
 * (p_stride) run grid-wide (cooperative launch, points held in registers), smaller ones in one
 * workgroup per cloud; the sample sequence is the same.                              */
int64_t iso_farthest_point_sampling_work_floats(int n_clouds, int64_t p_stride);
int iso_farthest_point_sampling(const float* points, const int64_t* lengths,
                                const int64_t* n_samples, const int64_t* start, int n_clouds,
                                int64_t p_stride, int64_t out_stride, float* work,
                                int64_t* out_idx, void* stream);

/* ------------------------------------------------------------------------
 * D. EWA surface splatting
 *    replaces SurfaceSplatting (DSS/core/rasterizer.py:103-661), the pybind module
 *    DSS._C (DSS/csrc/ext.cpp:5-18: splat_points, _splat_points_naive,
 *    _splat_points_occ_backward, _splat_points_occ_fast_cuda_backward,
 *    _backward_zbuf) and SurfaceSplattingRenderer.forward (DSS/core/renderer.py:36-82).
 *    Matrices are 4x4 row-major in pytorch3d's row-vector convention
 *    (p' = [p,1] @ M).  Packed clouds: cloud n owns rows
 *    [first_idx[n], first_idx[n]+num_pts[n]).
 * ---------------------------------------------------------------------- */

/* filter_renderable (rasterizer.py:184-254): flags (n_views, P) i32 = 1 when
 * znear <= z_view <= zfar and (backface_culling ? normal_view.z < 0 : 1).        */
int iso_splat_view_flags(const float* points, const float* normals, const float* views,
                         int32_t* flags, int64_t P, int n_views, float znear, float zfar,
                         int backface_culling, void* stream);
/* stable stream compaction of U-float rows: out[offsets[e]] = in[e % P] where
 * flags[e] != 0, e in [0,total) (offsets = exclusive scan of flags).           */
int iso_compact_rows(const float* in, const int32_t* flags, const int32_t* offsets,
                     float* out, int64_t P, int64_t total, int U, void* stream);
/* _compute_isotropic_Vrk (rasterizer.py:367-386): dists (N,p_stride,7) from the
 * K=7 self query -> h (packed) = clamp(0.5*max_{6 nn} d2, 5e-5, 0.01); clouds
 * with fewer than 7 points use d2 = 1e-3.  Row i of cloud n (i < num_pts[n]) is written to
 * h[first_idx[n] + i]; a rank that queried only a sub-range of each cloud passes the
 * sub-range in first_idx/num_pts and the true cloud sizes in cloud_num_pts (NULL = num_pts). */
int iso_splat_vrk_h(const float* dists, const int64_t* first_idx, const int64_t* num_pts,
                    const int64_t* cloud_num_pts, float* h, int n_clouds, int64_t p_stride,
                    void* stream);
/* _get_per_point_info + PointsRasterizer.transform (rasterizer.py:441-563,:618):
 * per packed point: NDC position (xy projected, z = view depth), ellipse (a,b,c),
 * cutoff, bbox radii, EWA normaliser.  views = world->view, projs = full
 * world->NDC, both (n_views,4,4).                                               */
int iso_splat_setup(const float* points, const float* normals, const float* h,
                    const int64_t* first_idx, const int64_t* num_pts, const float* views,
                    const float* projs, int n_views, int64_t max_pts, int image_size,
                    float sigma, float cutoff, float* ndc_out, float* ellipse_out,
                    float* cutoff_out, float* radii_out, float* scaler_out, void* stream);

/* Forward rasterisation = _C.splat_points (rasterize_points.h:461-525).
 * Two calls around one caller-side read of the pair count:
 *   iso_splat_bin_count : tile_cnt (n_clouds*T*T, zero on entry) += splats per
 *                         16x16 tile (T = iso_splat_tiles_per_side(S));
 *   caller: tile_off = exclusive scan (iso_prefix_sum), total = off[last]+cnt[last],
 *           allocate `pairs` (i32, >= total), zero tile_cursor (n_clouds*T*T i32) and
 *           overflow_flag; tile_off needs n_clouds*T*T + 1 entries (the last = total);
 *   iso_splat_forward   : fill + raster (tile_cursor holds the fill cursors first and is then
 *                         overwritten with the heaviest-first tile order the raster walks).  Per pixel the points_per_pixel (<= 150; above 32 a slow kernel that keeps the lists in the outputs)
 *                         smallest (z, idx) hits, entries with z - z0 >
 *                         depth_merging_thres reset to -1, occupancy = any hit.
 * Outputs as the reference: idx i32, zbuf/qvalue f32 (N,S,S,K), occ f32 (N,S,S),
 * image flipped in both axes (+X left, +Y up).  *overflow_flag != 0 afterwards
 * means pair_capacity was too small (result incomplete).
 * workspace (optional, NULL = none): iso_splat_forward_workspace_bytes(tiles of the band, K); with it
 * the tiles that hold many times the mean number of candidates (silhouettes) are rasterised in slices
 * by several workgroups and merged -- same result, the longest work item is a slice.
 * Non-square images (beyond the reference, whose rasteriser is square-only, rasterizer.py:52): image_size
 * = rows H, image_width = columns W (<= 0: W = H); outputs (N,H,W,K).  NDC follows pytorch3d's non-square
 * convention: the shorter side spans [-1,1], the longer [-e,e], e = longer / shorter -- square pixels of
 * 2 / min(H,W); the per-point set-up (iso_splat_setup / iso_splat_front) takes image_size = min(H,W).
 * T = iso_splat_tiles_per_side(H) tile rows, iso_splat_tiles_per_side(W) tile columns.
 * [tile_row_begin, tile_row_end) restricts both calls to a band of tile rows (in NDC pixel
 * order, i.e. before the flip): a rank of a sharded run rasterises only its band and leaves
 * the other pixels of the output tensors untouched; pass 0, T for the whole image.          */
int iso_splat_tiles_per_side(int image_size);
int iso_splat_bin_count(const float* points, const float* radii, const int64_t* first_idx,
                        const int64_t* num_pts, int n_clouds, int64_t max_pts, int image_size,
                        int image_width, int tile_row_begin, int tile_row_end, int32_t* tile_cnt,
                        void* stream);
/* Offsets of the tile lists from the counts of iso_splat_bin_count in ONE launch: tile_off = exclusive scan of
 * tile_cnt[0..n) (n = views x tiles + 1; same result as iso_prefix_sum), tile_cursor[0..n) = 0 (the fill cursors and,
 * in the last word, the overflow flag iso_splat_forward expects cleared), and tile_cnt is cleared as it is read, so
 * a caller that keeps the counter array never clears it again.                                        */
int iso_splat_tile_offsets(int32_t* tile_cnt, int32_t* tile_off, int32_t* tile_cursor, int64_t n, void* stream);
int iso_splat_forward(const float* points, const float* ellipse, const float* cutoff,
                      const float* radii, const int64_t* first_idx, const int64_t* num_pts,
                      int n_clouds, int64_t max_pts, float depth_merging_thres, int image_size,
                      int image_width, int points_per_pixel, int tile_row_begin, int tile_row_end,
                      int32_t* tile_cursor, const int32_t* tile_off,
                      int32_t* pairs, int64_t pair_capacity, int32_t* overflow_flag,
                      int32_t* idx_out, float* zbuf_out, float* qvalue_out, float* occ_out,
                      void* workspace, int64_t workspace_bytes, void* stream);
int64_t iso_splat_forward_workspace_bytes(int64_t n_tiles, int points_per_pixel);
/* iso_splat_forward and iso_splat_composite (below) in one pass: image_out (N,S,S,channels+1) is
 * composited from the pixels' K-best lists while they are in registers -- same arithmetic and order,
 * the (N,S,S,K) lists are written as before (the backward needs them) but not read again.       */
int iso_splat_render(const float* points, const float* ellipse, const float* cutoff,
                     const float* radii, const int64_t* first_idx, const int64_t* num_pts,
                     int n_clouds, int64_t max_pts, float depth_merging_thres, int image_size,
                     int image_width, int points_per_pixel, int tile_row_begin, int tile_row_end,
                     int32_t* tile_cursor, const int32_t* tile_off,
                     int32_t* pairs, int64_t pair_capacity, int32_t* overflow_flag,
                     int32_t* idx_out, float* zbuf_out, float* qvalue_out, float* occ_out,
                     void* workspace, int64_t workspace_bytes, const float* scaler,
                     const float* features, int channels, int norm_weighted, float eps,
                     float* image_out, void* stream);

/* iso_splat_render that also leaves the visible flags of the backward pass (iso_splat_mark_visible: rasterizer.py:850-853
 * of the reference marks the points listed in the pixels' lists) while it writes the lists: visible_out (P,) uint8, ZERO
 * on entry over the rows of the clouds (iso_splat_front_rows clears them), NULL = iso_splat_render.               */
int iso_splat_render_visible(const float* points, const float* ellipse, const float* cutoff,
                             const float* radii, const int64_t* first_idx, const int64_t* num_pts,
                             int n_clouds, int64_t max_pts, float depth_merging_thres, int image_size,
                             int image_width, int points_per_pixel, int tile_row_begin, int tile_row_end,
                             int32_t* tile_cursor, const int32_t* tile_off,
                             int32_t* pairs, int64_t pair_capacity, int32_t* overflow_flag,
                             int32_t* idx_out, float* zbuf_out, float* qvalue_out, float* occ_out,
                             void* workspace, int64_t workspace_bytes, const float* scaler,
                             const float* features, int channels, int norm_weighted, float eps,
                             float* image_out, uint8_t* visible_out, void* stream);

/* The reference's two-stage raster interface, DSS._C._rasterize_coarse / _rasterize_fine (DSS/csrc/ext.cpp:11-12;
 * dispatchers rasterize_points.h:167,257; kernels rasterize_points.cu:293-441, :503-596).  No Python caller in the
 * reference (splat_points runs both internally); here they are a compatibility surface beside iso_splat_forward, whose
 * own binning is the tile lists of iso_splat_bin_count.
 *   coarse: bin_points_out (N, B, B, M) int32, B = 1 + (image_size - 1) / bin_size: bin (by, bx) lists the packed
 *           indices of its cloud's points with z >= 0 whose box [p - r, p + r] overlaps the bin (extent = PixToNdc of
 *           its first / last pixel -+ half a pixel, non-strict), ascending (the reference: arrival order of its atomics
 *           on the GPU, ascending on the CPU), -1 behind them; *overflow_out is set to 1 when a bin had more than M
 *           (cut at M; never cleared here).
 *   fine:   per pixel the K front-most hits among the entries of its bin (negative entries skipped), tests and outputs
 *           of iso_splat_forward: fine(coarse(x)) == iso_splat_forward(x) bit for bit when no bin overflowed.  K <= 150 (above 32: the
 *           lists are kept in the output arrays, as in iso_splat_forward). */
int iso_rasterize_coarse(const float* points, const float* radii, const int64_t* first_idx, const int64_t* num_pts,
                         int n_clouds, int image_size, int bin_size, int max_points_per_bin, int32_t* bin_points_out,
                         int32_t* overflow_out, void* stream);
int iso_rasterize_fine(const float* points, const float* ellipse, const float* cutoff, const float* radii,
                       int64_t n_points, const int32_t* bin_points, int n_clouds, int max_points_per_bin,
                       float depth_merging_thres, int image_size, int bin_size, int points_per_pixel,
                       int32_t* idx_out, float* zbuf_out, float* qvalue_out, float* occ_out, void* stream);

/* DSS/utils/__init__.py:172-185 (gather_with_neg_idx) for one float per point, as SurfaceSplatting.forward uses it for
 * the fragments' scaler (rasterizer.py:635-637): out[i] = idx[i] >= 0 ? values[idx[i]] : 0, i < n.  idx and out 16-byte
 * aligned.                                                                                                            */
int iso_gather_neg_idx(const float* values, const int32_t* idx, int64_t n, float* out, void* stream);

/* renderer.py:53-78: w = exp(-0.5 q) * scaler[idx] (0 where idx < 0);
 * image[..., c] = sum_k w f / max(sum_k w, eps) (norm_weighted) or sum_k w f;
 * image[..., channels] = occupancy.  frag_scaler_out (n_pixels,K) may be NULL.  */
int iso_splat_composite(const int32_t* idx, const float* qvalue, const float* occ,
                        const float* scaler, const float* features, int64_t n_pixels,
                        int points_per_pixel, int channels, int norm_weighted, float eps,
                        float* frag_scaler_out, float* image_out, void* stream);

/* Backward = EllipticalRasterizer.backward, default fast path
 * (rasterizer.py:841-968 + rasterize_points_backward.cu:85-178 +
 * rasterize_points.cu:823-846), point-major and atomic-free:
 *   iso_splat_mark_visible: visible[p] = 1 for points listed in pixels whose first
 *                           slot is filled (visible zeroed by the caller);
 *   iso_splat_backward    : grad_points (P,3): xy = sum over pixels within
 *                           search_radius[n] of grad_occ * d/sdeno(|d|^2,1e-10)
 *                           (visible points; grad>0 pixels outside the splat rect
 *                           skipped), z = sum of grad_zbuf over the slots that list
 *                           the point.  No float atomics: a point's xy terms are added in image
 *                           order by eight lanes whose partial sums meet in a fixed tree, the z sum
 *                           is a pixel-major scatter in 64-bit fixed point (integer atomics, exactly
 *                           rounded) -> bit-stable.
 *                           total_points = rows of `points` (sizes the heavy-point list).
 *                           workspace: iso_splat_backward_workspace_bytes, 16-byte aligned (it starts
 *                           with 64-bit pixel masks of the gradient image).
 * grad_zbuf/idx may be NULL (no z gradient); visible may be NULL (= all).
 * rect_mode = 1 switches the xy support to the slow reference kernel's rectangle
 * |d| <= radii * radii_s (_C._splat_points_occ_backward, rasterize_points.cu:673-760);
 * search_radius is then unused.                                                  */
/* Backward of iso_splat_composite (renderer.py:53-78 composites through pytorch3d's differentiable
 * compositors): grad_image (n_pixels, channels + 1) -> grad_features (P, channels) and grad_scaler (P)
 * ACCUMULATED into the caller's zero-initialised buffers (float atomics), grad_qvalue (n_pixels, K)
 * and grad_occ (n_pixels) written; any output pointer may be NULL.                               */
int iso_splat_composite_backward(const int32_t* idx, const float* qvalue, const float* scaler,
                                 const float* features, const float* grad_image, int64_t n_pixels,
                                 int points_per_pixel, int channels, int norm_weighted, float eps,
                                 float* grad_features, float* grad_qvalue, float* grad_scaler,
                                 float* grad_occ, void* stream);
int iso_splat_mark_visible(const int32_t* idx, int64_t n_pixels, int points_per_pixel,
                           uint8_t* visible, void* stream);
/* search_radius_out[n] = lower-median(radii of the visible points of cloud n, both columns
 * flattened) * radii_s  (rasterizer.py:884, torch.median semantics), 0 for a cloud without
 * visible points.  Exact: radix select on the f32 bit patterns, three histogram passes (11 + 11 + 10 bits) and
 * a final pass.  The workspace holds the histograms: it must be ZERO on entry and is left zero on exit, so a
 * caller that keeps one workspace clears it once (hipMemset) and never again.                        */
int64_t iso_splat_median_radius_workspace_bytes(int n_clouds);
int iso_splat_median_radius(const float* radii, const uint8_t* visible, const int64_t* first_idx,
                            const int64_t* num_pts, int n_clouds, int64_t max_pts, float radii_s,
                            void* workspace, int64_t workspace_bytes, float* search_radius_out,
                            void* stream);
int64_t iso_splat_backward_workspace_bytes(int n_clouds, int image_size, int image_width,
                                           int64_t total_points);
int iso_splat_backward(const float* points, const float* radii, const uint8_t* visible,
                       const float* search_radius, const int64_t* first_idx,
                       const int64_t* num_pts, int n_clouds, int64_t max_pts,
                       const float* grad_occ, const int32_t* idx, const float* grad_zbuf,
                       int image_size, int image_width, int points_per_pixel, int rect_mode, float radii_s,
                       int64_t total_points, void* workspace, int64_t workspace_bytes,
                       float* grad_points, void* stream);
/* _C._backward_zbuf alone (rasterize_points.cu:823-846): z_grad[idx] += grad_zbuf
 * by atomic scatter, accumulating into the caller's (P) buffer.                 */
int iso_splat_zbuf_backward(const int32_t* idx, const float* grad_zbuf, int64_t n_pixels,
                            int points_per_pixel, float* z_grad, void* stream);

/* ------------------------------------------------------------------------
 * E. Brick grid + fused neighbour kernels (the hot path of the iso-point cycle)
 *
 *    The stand-alone FRNN entry points of section B stay the general API.  The cycle itself
 *    (UniformProjection.resample, levelset_sampling.py:239-288, and the K = 7 bandwidth query of
 *    SurfaceSplatting._get_per_point_info, rasterizer.py:367-386) runs on a two-level grid:
 *    coarse BRICKS of 4x4x4 fine cells in global memory (x-major brick id, so a range of brick
 *    ids is an x-slab of space -- the unit the multi-GPU path shards by), fine cells only in LDS:
 *    a workgroup stages its brick plus a one-cell halo, counting-sorted by fine cell, and every
 *    query of the brick walks its 3x3x3 fine cells there.  Exact results (K nearest within r,
 *    ties to the lower id): a query whose K-th distance is not covered by the staged block is
 *    finished by a tail kernel that walks rings of bricks.
 *
 *    One workspace (iso_bricks_workspace_bytes(n_max), 256-B aligned) holds the grid; n_max =
 *    n_own + import_max must be the same in every call on that workspace.
 *
 *    iso_bricks_build: bbox = [min xyz, 0, max xyz, 0] on the device (iso_points_bbox layout; N
 *    ranks reduce it first so that every rank derives the same grid), n_total = points of the
 *    WHOLE cloud.  radius > 0: fixed search radius; radius <= 0: r = sqrt(|bbox diag| / n_total)
 *    * knn_k (levelset_sampling.py:129-131).  Fine cell = cell_scale * sqrt(diag / n_total),
 *    clamped to [extent / (4 (cap-1)), 1.002 r].  Own points get ids id_base + i; the optional
 *    imported records (halo cells of other ranks: rec0 = x,y,z,id-bits, rec1 = nx,ny,nz,payload,
 *    count on the device) are searched but never queried.  payload: one int per own point
 *    carried to the kernels (the view mask for iso_splat_h_fused), or NULL.
 * ---------------------------------------------------------------------- */
int64_t iso_bricks_workspace_bytes(int64_t n_max);
/* Once per workspace, before its first build: zeroes the brick counters and the counter block.  Every build leaves
 * the brick counters zeroed again (the offsets pass clears what it has read), so no build starts with a clearing
 * pass.  The counter block (576 ints at byte 256 of the workspace; behind it, also at a fixed offset, the 2048 zero-on-entry
 * chunk totals of the brick scan): [0..15] counters of the current grid (slots as
 * documented below), [16..31] "sticky" sums of the counters of all earlier grids on this workspace -- a header write
 * adds what it resets -- so that a caller can check the overflow / certification counters of a whole cycle of several
 * grids with one read afterwards.                                                                    */
int iso_bricks_workspace_init(void* workspace, int64_t n_max, void* stream);
/* Diagnostic, synchronises `stream`: ISO_ERR_INVALID when iso_bricks_workspace_init never ran on this workspace (it
 * leaves a mark in the counter block).  A build on such a workspace counts into whatever the memory held and gives a
 * wrong grid with no other sign; the builds themselves cannot check (the mark lives on the device).  Also refuses a
 * workspace whose arrival words or scan totals are not zero (an aborted build left them set: every later build on it
 * would compute wrong offsets).                                                                                  */
int iso_bricks_workspace_check(const void* workspace, int64_t n_max, void* stream);
/* iso_bricks_build for ONE rank that holds the whole cloud (n_total = n, id_base = 0, no imports): the bounding box
 * is taken by the build itself -- no iso_points_bbox pass, no 8-float round trip (six launches per grid instead of
 * twelve).  Same grid, same results as iso_points_bbox + iso_bricks_build.                          */
int iso_bricks_build_whole(const float* points, const float* normals, const int32_t* payload, int64_t n,
                           float radius, int knn_k, float cell_scale, void* workspace,
                           int64_t workspace_bytes, void* stream);
/* iso_bricks_build_whole without its box pass: the bounding box of `points` is the workspace's PENDING BOX, left there by
 * the launch that wrote the points (iso_project_*_follow).  follow_host != NULL with views: the per-tile counts that launch
 * left in follow_host->front_ws are scanned on the side (one workgroup more in the count launch): front_ws, first_idx_out,
 * num_pts_out, view_total_out as iso_splat_view_mask_scan leaves them.                                  */
int iso_bricks_build_pending(const float* points, const float* normals, const int32_t* payload, int64_t n,
                             float radius, int knn_k, float cell_scale, void* workspace,
                             int64_t workspace_bytes, const iso_follow* follow_host, void* stream);
/* the pending box as 8 floats (iso_points_bbox layout), cleared afterwards: N ranks all-gather it for iso_bricks_params */
int iso_bricks_box_take(void* workspace, int64_t n_max, float* box_out, void* stream);
int iso_bricks_build(const float* points, const float* normals, const int32_t* payload,
                     int64_t n_own, int64_t id_base, const float* import_rec0,
                     const float* import_rec1, const int32_t* import_count, int64_t import_max,
                     const float* bbox, int64_t n_total, float radius, int knn_k,
                     float cell_scale, void* workspace, int64_t workspace_bytes, void* stream);
/* N ranks, one x-slab of the cloud each: every rank derives the SAME grid from the
 * reduced bounding box (iso_bricks_params), exports the points other ranks' queries can reach
 * (iso_halo_export: fine x-cell within one cell of another rank's local x-range; rank_boxes =
 * (world, 8) floats, every rank's local iso_points_bbox), the export buffers are all-gathered
 * (one RCCL all-gather of `export_buf`: float4[2][capacity + 1], word 0 = record count), and each
 * rank keeps what its own queries can reach (iso_halo_import) as the imported records of
 * iso_bricks_build (bbox = NULL: header already written).  halo_cells: width of the exchanged band in
 * fine cells (>= 1; the fused kernels certify their results within one cell, the tail kernels -- a query
 * whose K-th neighbour lies farther -- check their search ball against the imported band and count the
 * queries that left it).  The grid's counters: slot 4 an exporter ran out of capacity, slot 5 the import
 * buffer did, slot 6 tail queries that needed more than the imported band.                       */
/* boxes: n_boxes rows of 8 floats (iso_points_bbox layout: every rank's local box as all-gathered); the header is
 * written for their union.                                                                          */
int iso_bricks_params(const float* boxes, int n_boxes, int64_t n_total, int64_t n_own, int64_t id_base, float radius,
                      int knn_k, float cell_scale, void* workspace, int64_t n_max, void* stream);
int iso_halo_export(void* workspace, const float* points, const float* normals, const int32_t* payload,
                    int64_t n_own, const float* rank_boxes, int world, int rank, int halo_cells,
                    float* export_buf, int64_t capacity, void* stream);
int iso_halo_import(void* workspace, int64_t n_max, const float* gathered, const float* rank_boxes, int world,
                    int rank, int halo_cells, int64_t capacity, float* import_rec0, float* import_rec1,
                    int32_t* import_count, int64_t import_capacity, void* stream);
/* One resample move of every own point: K+1 = k_plus_one self-inclusive FRNN query (radius of
 * the build), column 0 dropped, tangent-plane repulsion with inv_sigma = n_total / diag
 * (levelset_sampling.py:254-284; same arithmetic as iso_frnn_query + iso_repulse).  `points`
 * = the own points the grid was built from.  idx_out (n_own, K) int64 / d2_out (n_own, K)
 * optional: the neighbour lists (UniformProjection._knn_idx / _knn_dists).            */
int iso_resample_fused(void* workspace, int64_t n_max, const float* points, int64_t n_own,
                       int k_plus_one, float* points_out, int64_t* idx_out, float* d2_out,
                       void* stream);
/* mask_out[i] bit v = point i is renderable in view v (z range + backface culling,
 * rasterizer.py:184-254; = iso_splat_view_flags for up to 8 views at once);
 * view_count_out[0..7] = points per view.                                            */
int iso_splat_view_mask(const float* points, const float* normals, const float* views, int n_views,
                        int64_t n, float znear, float zfar, int backface_culling,
                        int32_t* mask_out, int32_t* view_count_out, void* stream);
/* h_out[v * n_own + i] = clamp(max d2 of the 6 nearest OTHER points renderable in view v within
 * the build radius / 2, 5e-5, 0.01) for every view v in which own point i is renderable
 * (rasterizer.py:367-386; = iso_frnn_query K = 7 on each filtered view cloud + iso_splat_vrk_h),
 * all views in one pass over the grid (payload = view mask).  view_total[v] = points of the
 * whole cloud renderable in view v (< 7: the reference's 1e-3 branch).                */
int iso_splat_h_fused(void* workspace, int64_t n_max, const float* points, const int32_t* mask,
                      int64_t n_own, const int32_t* view_total, int n_views, float* h_out,
                      void* stream);
/* Front end of the splat for ONE cloud seen by up to 8 cameras, on the unfiltered cloud and without a
 * host read: mask (iso_splat_view_mask) + h (iso_splat_h_fused) -> the packed per-view arrays that
 * _C.splat_points takes (SurfaceSplatting.forward, rasterizer.py:597-661: filter, extend to the
 * cameras, _get_per_point_info, transform).  Packed order = view-major, points ascending (the
 * reference's); first_idx_out / num_pts_out (n_views) int64 and view_total_out (8) int32 are written
 * on the device; the per-point outputs must hold n_views * n_points rows (upper bound), rows beyond
 * first[n_views-1] + num[n_views-1] stay untouched.  features (n_points, channels) -> features_out
 * packed, or features_from_normals = 1: 0.5 (normalize(n) + 1) (3 channels); src_out: original point of
 * every packed row (scatter of the row gradients back to the cloud).  Same arithmetic as
 * iso_splat_view_flags + iso_compact_rows + iso_splat_setup.                                     */
int64_t iso_splat_front_workspace_bytes(int64_t n_points);
int iso_splat_front(const float* points, const float* normals, const float* features, int channels,
                    int features_from_normals, const int32_t* mask, const float* h, int64_t n_points,
                    const float* views, const float* projs, int n_views, int image_size, float sigma,
                    float cutoff, void* workspace, int64_t workspace_bytes, int64_t* first_idx_out,
                    int64_t* num_pts_out, int32_t* view_total_out, float* ndc_out, float* ellipse_out,
                    float* cutoff_out, float* radii_out, float* scaler_out, float* features_out,
                    int32_t* src_out, void* stream);
/* The same front end with two launches fewer and the view totals known BEFORE the bandwidth pass needs them:
 * iso_splat_view_mask_scan = renderable mask (iso_splat_view_mask's rule) + per-chunk counts + their scan: mask_out,
 * first_idx_out / num_pts_out (n_views int64) and view_total_out (8 int32) are final, the workspace
 * (iso_splat_front_workspace_bytes) holds the scanned chunk table; iso_splat_front_rows = the compaction + set-up pass
 * of iso_splat_front alone, on that workspace / mask / first_idx.                                     */
int iso_splat_view_mask_scan(const float* points, const float* normals, const float* views, int n_views,
                             int64_t n_points, float znear, float zfar, int backface_culling, int32_t* mask_out,
                             void* workspace, int64_t workspace_bytes, int64_t* first_idx_out, int64_t* num_pts_out,
                             int32_t* view_total_out, void* stream);
int iso_splat_front_rows(const float* points, const float* normals, const float* features, int channels,
                         int features_from_normals, const int32_t* mask, const float* h, int64_t n_points,
                         const float* views, const float* projs, int n_views, int image_size, float sigma,
                         float cutoff, const void* workspace, int64_t workspace_bytes, const int64_t* first_idx,
                         float* ndc_out, float* ellipse_out, float* cutoff_out, float* radii_out,
                         float* scaler_out, float* features_out, int32_t* src_out,
                         uint8_t* visible_zero_out /* (rows) or NULL: the rows' "visible" flags of the backward pass
                                                      (iso_splat_mark_visible), cleared here instead of by a fill */,
                         int64_t row_capacity /* rows the output arrays hold; <= 0: num_points.sum() rows, whatever that is */,
                         int32_t* overflow_out /* set to 1 when rows were dropped at row_capacity (never cleared here), or NULL */,
                         void* stream);
/* Gradient of the packed NDC rows of iso_splat_front w.r.t. the WORLD points (the part of the reference's
 * backward that autograd runs through cameras.transform_points in SurfaceSplatting.transform,
 * rasterizer.py:565-582,618: per-point set-up is under no_grad :608-610, so d(ndc)/d(point) is all there is).
 * grad_ndc (rows,3) -> grad_points (n_points,3), summed over the views a point is rendered in (ascending view
 * order, deterministic); mask / src / first_idx / num_points as written by iso_splat_view_mask / iso_splat_front. */
int iso_splat_points_backward(const float* points, int64_t n_points, const float* views, const float* projs,
                              int n_views, const int32_t* mask, const int32_t* src, const int64_t* first_idx,
                              const int64_t* num_points, const float* grad_ndc, float* grad_points, void* stream);
/* The z gradient of iso_splat_backward in pieces, for N ranks that each own a band of tile rows:
 * zscale (2 ints: bits of max |grad_zbuf|, exponent) <- iso_splat_z_absmax over the rank's pixels, MAX-
 * reduced over the ranks (word 0); iso_splat_z_scatter derives the exponent and adds the rank's pixels
 * into the zero-initialised 64-bit accumulators acc (total rows); acc is SUM-reduced over the ranks;
 * iso_splat_z_finish converts rows [row0, row0 + n_rows) into grad_points[:, 2].  Exactly rounded and
 * order independent like the single-GPU path (ZbufBackwardKernel, rasterize_points.cu:823-846).     */
int iso_splat_z_absmax(const float* grad_zbuf, int64_t n, int32_t* zscale, void* stream);
int iso_splat_z_scatter(const int32_t* idx, const float* grad_zbuf, int64_t n_pixels, int points_per_pixel,
                        int64_t pixels_per_view, int32_t* zscale, int64_t* acc, void* stream);
int iso_splat_z_finish(const int64_t* acc, const int32_t* zscale, int64_t row0, int64_t n_rows,
                       float* grad_points, void* stream);
/* The same for the band of EVERY view in one call each (a rank's band = rows [y0, y1) of all n_views images of the
 * (N,H,W,K) arrays; idx / grad_zbuf point at the band's first pixel of view 0, view_pixels = H * W, band_pixels =
 * (y1 - y0) * W): iso_splat_band_marks = iso_splat_mark_visible + iso_splat_z_absmax (zscale zeroed first) over
 * the n_views slices, iso_splat_band_z_scatter = iso_splat_z_scatter over them (pixels_per_view = view_pixels).
 * Two launches per call instead of two per view -- a rank's share of a cycle is a chain of small kernels.   */
int iso_splat_band_marks(const int32_t* idx, const float* grad_zbuf, int n_views, int64_t view_pixels,
                         int64_t band_pixels, int points_per_pixel, uint8_t* visible, int32_t* zscale, void* stream);
int iso_splat_band_z_scatter(const int32_t* idx, const float* grad_zbuf, int n_views, int64_t view_pixels,
                             int64_t band_pixels, int points_per_pixel, int32_t* zscale, int64_t* acc, void* stream);
/* N ranks: the packed per-view arrays of the WHOLE cloud (view-major, then rank, then the rank's own
 * order = the single-GPU order when the ranks hold consecutive ranges of the cloud) from the
 * all-gathered per-rank outputs of iso_splat_front.  gathered: world blocks of 12 * capacity floats
 * (ndc 3, ellipse 3, radii 2, scaler 1, features 3 -- the front end's outputs laid out in one
 * buffer); counts (world, 8) int32: the ranks' view_total_out.  own_first / own_num: this rank's
 * rows inside the global layout.                                                              */
int iso_splat_repack(const float* gathered, int64_t capacity, int world, int rank, int n_views,
                     const int32_t* counts, float cutoff, int64_t max_rows, float* ndc_out,
                     float* ellipse_out, float* cutoff_out, float* radii_out, float* scaler_out,
                     float* features_out, int64_t* first_idx_out, int64_t* num_pts_out,
                     int64_t* own_first_out, int64_t* own_num_out, void* stream);

/* ----------------------------------------------------------------------
 * F. N ranks, splat stage: the fragment-side exchange (csrc/band.hip; SURVEY 8(e) -- the reference has no
 *    distributed layer; the cell order it shards by is that of DSS/csrc/rasterize_points_backward.cu:119).
 *
 *    Every rank runs the front end on its own points and rasterises a BAND of 16-pixel tile rows of every view
 *    (the balanced split of the tile rows over the ranks).  A packed row is needed by exactly the bands its
 *    bounding box touches (the binning pass's test), so it is sent to those ranks only, in ONE all-to-all with equal
 *    splits: `send` = world segments of iso_splat_band_segment_floats(cap_pair) floats, each a 16-int header
 *    ([0..7] records per view, [8] records wanted) and cap_pair records of 13 floats (ndc 3, ellipse 3, radii 2,
 *    scaler 1, features 3, the row's GLOBAL id: its position in the single-GPU packed layout).  Rows keep their
 *    order inside a segment.  iso_splat_band_import lays the received segments out view-major and, inside a view, by
 *    source rank: ascending global id, so the (z, id) tie rule of the rasteriser picks what the single-GPU run picks;
 *    iso_splat_band_remap turns the band's per-pixel local row ids into the global ones (= the single-GPU lists).
 *    Backward: the band's per-record results (64-bit fixed-point z sum, visible flag) go back into the slots the
 *    records came in (iso_splat_band_return: world segments of cap_pair x 2 int64), the reverse all-to-all returns
 *    them, iso_splat_band_merge adds them to the owner's rows (integer adds: order-independent).
 *    flags (int32[1], caller-cleared): bit 0 a send segment overflowed cap_pair, bit 1 the local arrays did.
 *    counts: (world, 8) int32 rows per view of every rank (all-gathered view totals of iso_splat_front).        */
int64_t iso_splat_band_segment_floats(int64_t cap_pair);
int64_t iso_splat_band_export_workspace_bytes(int64_t max_rows_per_view, int n_views, int world);
int iso_splat_band_export(const float* ndc, const float* ellipse, const float* radii, const float* scaler,
                          const float* features, const int64_t* first_idx, const int64_t* num_points,
                          int n_views, int64_t max_rows_per_view, const int32_t* counts, int world, int rank,
                          int image_size, int image_width, int64_t cap_pair, float* send, int32_t* sent_row,
                          int64_t* acc_own, uint8_t* vis_own, int64_t* gid_first, int64_t* first_global,
                          int64_t* num_global, int32_t* flags, void* workspace, int64_t workspace_bytes,
                          void* stream);
int iso_splat_band_import(const float* recv, int world, int n_views, int64_t cap_pair, int64_t cap_local,
                          float cutoff, float* ndc, float* ellipse, float* cutoff_out, float* radii,
                          float* scaler, float* features, int32_t* gid, int32_t* origin, int64_t* acc_local,
                          uint8_t* vis_local, int64_t* first_local, int64_t* num_local, int32_t* place,
                          int32_t* flags, void* stream);
int iso_splat_band_remap(const int32_t* idx_local, const int32_t* gid, int n_views, int64_t view_pixels,
                         int64_t band_pixel0, int64_t band_pixels, int points_per_pixel, int32_t* idx_global,
                         void* stream);
int iso_splat_band_return(const int64_t* acc_local, const uint8_t* vis_local, const int32_t* origin,
                          const int64_t* first_local, const int64_t* num_local, int n_views, int64_t cap_local,
                          int64_t* ret, void* stream);
int iso_splat_band_merge(const int64_t* back, const int32_t* sent_row, int world, int n_views,
                         int64_t max_rows_per_view, int64_t cap_pair, int64_t* acc_own, uint8_t* vis_own,
                         const void* workspace, void* stream);
/* iso_splat_median_radius in pieces: pass 0, 1, 2 count the caller's rows into the pass's histograms (the slice
 * workspace + pass * iso_splat_median_pass_words(n_clouds) words of that many words), N ranks sum that slice over
 * the ranks before the next piece; iso_splat_median_final resolves and leaves the workspace zeroed.            */
int64_t iso_splat_median_pass_words(int n_clouds);
int iso_splat_median_pass(int pass, const float* radii, const uint8_t* visible, const int64_t* first_idx,
                          const int64_t* num_pts, int n_clouds, int64_t max_pts, void* workspace,
                          int64_t workspace_bytes, void* stream);
int iso_splat_median_final(void* workspace, int n_clouds, float radii_s, float* search_radius_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ISOPOINTS_H_ */
