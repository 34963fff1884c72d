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
/* scratch for the per-wave activation-derivative stash */
int64_t iso_project_siren_workspace_bytes(int64_t n, int hidden, int n_hidden);
int iso_project_siren(const float* pts_in, float* pts_out, float* normals_out,
                      uint8_t* mask_out, int64_t n, const float* packed,
                      int hidden, int n_hidden, float omega_first,
                      float omega_hidden, int max_iters, float tol,
                      void* workspace, int64_t workspace_bytes, void* stream);
/* one SDF + gradient evaluation, no Newton move (used by tests and by callers
 * that only need _compute_sdf_and_grad, levelset_sampling.py:142-170)        */
int iso_siren_sdf_grad(const float* pts, float* sdf_out, float* grad_out,
                       int64_t n, const float* packed, int hidden,
                       int n_hidden, float omega_first, float omega_hidden,
                       void* workspace, int64_t workspace_bytes, void* stream);

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
#define ISO_GRID_MAX_RES 128
/* upper bound of `total` for a grid built by iso_frnn_make_grid */
#define ISO_GRID3_MAX_CELLS ((ISO_GRID_MAX_RES + 1) * (ISO_GRID_MAX_RES + 1) * (ISO_GRID_MAX_RES + 1))

/* Device-side grid sizing for clouds `points` (N, P, 3) padded, lengths (N)
 * i64 (NULL = all P), radius (N) f32.  Writes params (N, 8).  Cell size is
 * chosen from point density (about 8 points per occupied cell on a surface)
 * but never finer than r/2 or (max extent)/128; the query walks Chebyshev
 * rings of cells so the result does not depend on the cell size.  No host
 * sync: allocate grid arrays for ISO_GRID3_MAX_CELLS.                         */
int iso_frnn_make_grid(const float* points, const int64_t* lengths,
                       const float* radius, int n_clouds, int64_t p_stride,
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
 * sorted2 / sorted_idx2 / off come from the build calls above.               */
int iso_frnn_query(const float* points1, const int64_t* lengths1,
                   const float* points2, const float* sorted2,
                   const int32_t* sorted_idx2,
                   const int64_t* lengths2, const int32_t* off,
                   const float* grid_params, const float* radius, int K,
                   float* dists_out, int64_t* idxs_out, float* nn_out,
                   int n_clouds, int64_t p1_stride, int64_t p2_stride,
                   int64_t g_stride, void* stream);

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
 *    idx: row i starts at idx + i*idx_row_stride (i64, K entries, -1 padding:
 *    weight 0 there) -- lets the caller pass the [:,1:] view of a (P,K+1)
 *    query result without copying.  inv_sigma (= P/diag, :256) is a DEVICE
 *    scalar so the bounding-box reduction needs no host sync.                 */
int iso_repulse(const float* points, const float* normals, const int64_t* idx,
                int64_t idx_row_stride, float* points_out, int64_t n, int K,
                const float* inv_sigma, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ISOPOINTS_H_ */
