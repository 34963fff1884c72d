// Tangent-plane repulsion step (HBM-bound: 100 B/point algorithmic).
// Reference semantics: UniformProjection.resample body,
// DSS/models/levelset_sampling.py:268-284 (single cloud, see isopoints.h C).
//
// One lane per point.  The point's K neighbour indices are read as one
// contiguous 8*K-byte row; neighbour positions / normals are 12-B gathers that
// are served by L2 / Infinity Cache (the arrays are read-only and, when the
// caller keeps the cloud in grid-cell order, spatially coherent).  The normal
// of each neighbour is normalised on the fly exactly as F.normalize does
// (v / max(|v|, 1e-12)), so no normalised copy of the normals is ever written.
#include "iso_common.h"
#include "iso_tile.h"

#pragma clang fp contract(off)

namespace {

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_repulse(
    const float* __restrict__ pts, const float* __restrict__ nrm,
    const int64_t* __restrict__ idx, int64_t idx_stride, float* __restrict__ out,
    int64_t n, int64_t first, int K, const float* __restrict__ inv_sigma_ptr) {
  const float inv_sigma = *inv_sigma_ptr;
  __shared__ __attribute__((aligned(16))) float tile[BLOCK * 3];
  const int t = threadIdx.x;
  const int64_t n_tiles = (n + BLOCK - 1) / BLOCK;
  for (int64_t tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
    const int64_t base = tl * BLOCK;
    const int cnt = (int)((n - base) < BLOCK ? (n - base) : BLOCK);
    iso_tile_load3<BLOCK>(pts, first + base, cnt, tile);
    __syncthreads();
    float px = 0.f, py = 0.f, pz = 0.f;
    if (t < cnt) {
      px = tile[3 * t]; py = tile[3 * t + 1]; pz = tile[3 * t + 2];
      const int64_t* row = idx + (base + t) * idx_stride;
      float sw = 0.f, mx = 0.f, my = 0.f, mz = 0.f;
      for (int k = 0; k < K; ++k) {
        int64_t j = row[k];
        // frnn_gather returns zeros for idx < 0 (levelset_sampling.py:268-271);
        // the weight is then forced to 0 (:275) so the term drops out.
        if (j < 0) continue;
        float qx = pts[j * 3], qy = pts[j * 3 + 1], qz = pts[j * 3 + 2];
        float ux = nrm[j * 3], uy = nrm[j * 3 + 1], uz = nrm[j * 3 + 2];
        float un = sqrtf((ux * ux + uy * uy) + uz * uz);
        un = un > 1e-12f ? un : 1e-12f;
        ux = ux / un; uy = uy / un; uz = uz / un;
        float dx = px - qx, dy = py - qy, dz = pz - qz;
        float d2 = (dx * dx + dy * dy) + dz * dz;
        float w = expf(-d2 * inv_sigma);
        float dn = (dx * ux + dy * uy) + dz * uz;
        float tx = dx - dn * ux, ty = dy - dn * uy, tz = dz - dn * uz;
        sw += w;
        mx += w * tx; my += w * ty; mz += w * tz;
      }
      float dens = sw + 1.0f;
      float den = iso_eps_denom(sw, 1.0e-17f);
      px = px + dens * mx / den;
      py = py + dens * my / den;
      pz = pz + dens * mz / den;
    }
    __syncthreads();
    if (t < cnt) { tile[3 * t] = px; tile[3 * t + 1] = py; tile[3 * t + 2] = pz; }
    __syncthreads();
    iso_tile_store3<BLOCK>(out, base, cnt, tile);
    __syncthreads();
  }
}

}  // namespace

extern "C" int iso_repulse(const float* points, const float* normals,
                           const int64_t* idx, int64_t idx_row_stride,
                           float* points_out, int64_t n, int64_t first_point, int K,
                           const float* inv_sigma, void* stream) {
  ISO_REQUIRE(n >= 0 && K >= 0 && first_point >= 0, ISO_ERR_INVALID, "iso_repulse: bad sizes");
  if (n == 0) return ISO_OK;
  ISO_REQUIRE(points && normals && points_out && inv_sigma && (idx || K == 0),
              ISO_ERR_INVALID, "iso_repulse: null pointer");
  ISO_REQUIRE(idx_row_stride >= K, ISO_ERR_INVALID, "iso_repulse: idx_row_stride < K");
  ISO_REQUIRE(points != points_out, ISO_ERR_INVALID,
              "iso_repulse: in-place not allowed (neighbours are re-read)");
  constexpr int BLOCK = 256;
  hipLaunchKernelGGL(k_repulse<BLOCK>, dim3(iso_stream_grid(n, BLOCK)), dim3(BLOCK),
                     0, (hipStream_t)stream, points, normals, idx, idx_row_stride,
                     points_out, n, first_point, K, inv_sigma);
  ISO_CHECK_LAUNCH("iso_repulse");
  return ISO_OK;
}

// ---------------------------------------------------------------------------------------------
// Sparsest-edge candidates for point_processing.upsample (DSS/utils/point_processing.py:326-339):
// for every point p with neighbours nn_0..nn_{K-1}:
//   mid_k      = (nn_k + 2 p) / 3
//   spars_k    = min_j | mid_k - nn_j |            (norm, not squared: :336-337)
//   sparsity   = max_k spars_k ,  father_nb = argmax_k (first maximum)
//   candidate  = mid_{father_nb}
// The reference materialises the (N,P,K,K,3) difference tensor (11.5 kB/point at K=31); here one
// lane keeps its K neighbours in LDS (K*12 B per lane) and never writes the K^2 distances.
// ---------------------------------------------------------------------------------------------
namespace {

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_upsample_candidates(
    const float* __restrict__ pts, const float* __restrict__ knn /*(n,K,3)*/, int64_t n, int K,
    float* __restrict__ sparsity, float* __restrict__ cand /*(n,3)*/) {
  extern __shared__ float s_nn[];   // [K*3][BLOCK]  (component-major: conflict-free per lane)
  const int t = threadIdx.x;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + t; i < n; i += (int64_t)gridDim.x * BLOCK) {
    const float px = pts[i * 3], py = pts[i * 3 + 1], pz = pts[i * 3 + 2];
    for (int k = 0; k < K * 3; ++k) s_nn[k * BLOCK + t] = knn[i * K * 3 + k];
    float best = -1.0f;
    float bx = px, by = py, bz = pz;
    for (int k = 0; k < K; ++k) {
      const float mx = (s_nn[(k * 3) * BLOCK + t] + 2.0f * px) / 3.0f;
      const float my = (s_nn[(k * 3 + 1) * BLOCK + t] + 2.0f * py) / 3.0f;
      const float mz = (s_nn[(k * 3 + 2) * BLOCK + t] + 2.0f * pz) / 3.0f;
      float mn = 3.0e38f;
      for (int j = 0; j < K; ++j) {
        const float dx = mx - s_nn[(j * 3) * BLOCK + t], dy = my - s_nn[(j * 3 + 1) * BLOCK + t],
                    dz = mz - s_nn[(j * 3 + 2) * BLOCK + t];
        const float d = sqrtf((dx * dx + dy * dy) + dz * dz);
        mn = d < mn ? d : mn;
      }
      if (mn > best) { best = mn; bx = mx; by = my; bz = mz; }
    }
    sparsity[i] = best;
    cand[i * 3] = bx; cand[i * 3 + 1] = by; cand[i * 3 + 2] = bz;
  }
}

}  // namespace

extern "C" int iso_upsample_candidates(const float* points, const float* knn, int64_t n, int K,
                                       float* sparsity_out, float* candidates_out, void* stream) {
  ISO_REQUIRE(n >= 0 && K >= 1 && K <= 64, ISO_ERR_INVALID, "iso_upsample_candidates: bad sizes (1 <= K <= 64)");
  if (n == 0) return ISO_OK;
  ISO_REQUIRE(points && knn && sparsity_out && candidates_out, ISO_ERR_INVALID,
              "iso_upsample_candidates: null pointer");
  constexpr int BLOCK = 64;
  const size_t lds = (size_t)K * 3 * BLOCK * sizeof(float);
  hipLaunchKernelGGL(k_upsample_candidates<BLOCK>, dim3(iso_stream_grid(n, BLOCK)), dim3(BLOCK), lds,
                     (hipStream_t)stream, points, knn, n, K, sparsity_out, candidates_out);
  ISO_CHECK_LAUNCH("iso_upsample_candidates");
  return ISO_OK;
}

// ---------------------------------------------------------------------------------------------
// Edge-aware candidates of EdgeAwareProjection.upsample (DSS/models/levelset_sampling.py:609-628):
// for every point p (normal n) with neighbours nn_k (normals u_k):
//   mid_k    = (nn_k + 2 p) / 3 ,  d_kj = mid_k - nn_j
//   m_k      = sqrt(max(| min_j ( |d_kj| - sum_c (d_kj,c u_k,c)^2 ) |, 1e-17))    (as written at :620-625:
//              the norm minus the sum of SQUARED COMPONENT PRODUCTS with the normal of neighbour k --
//              `knn_normals.unsqueeze(-2)` broadcasts over j -- not a squared dot product)
//   edge_k   = (2 - n . u_k) ^ edge_sensitivity
//   sparsity = max_k edge_k m_k ,  candidate = mid_argmax (first maximum)
// Same lane-owns-a-point layout as k_upsample_candidates with the neighbour normals next to the
// neighbour positions in LDS (K*24 B per lane).
// ---------------------------------------------------------------------------------------------
namespace {

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_ear_candidates(
    const float* __restrict__ pts, const float* __restrict__ nrm, const float* __restrict__ knn,
    const float* __restrict__ knn_nrm, int64_t n, int K, float edge_sensitivity,
    float* __restrict__ sparsity, float* __restrict__ cand) {
  extern __shared__ float s_ear[];   // [K*3][BLOCK] positions, then [K*3][BLOCK] normals
  float* s_p = s_ear;
  float* s_u = s_ear + (size_t)K * 3 * BLOCK;
  const int t = threadIdx.x;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + t; i < n; i += (int64_t)gridDim.x * BLOCK) {
    const float px = pts[i * 3], py = pts[i * 3 + 1], pz = pts[i * 3 + 2];
    const float nx = nrm[i * 3], ny = nrm[i * 3 + 1], nz = nrm[i * 3 + 2];
    for (int k = 0; k < K * 3; ++k) {
      s_p[k * BLOCK + t] = knn[i * K * 3 + k];
      s_u[k * BLOCK + t] = knn_nrm[i * K * 3 + k];
    }
    float best = -__builtin_inff();
    float bx = px, by = py, bz = pz;
    for (int k = 0; k < K; ++k) {
      const float mx = (s_p[(k * 3) * BLOCK + t] + 2.0f * px) / 3.0f;
      const float my = (s_p[(k * 3 + 1) * BLOCK + t] + 2.0f * py) / 3.0f;
      const float mz = (s_p[(k * 3 + 2) * BLOCK + t] + 2.0f * pz) / 3.0f;
      const float ux = s_u[(k * 3) * BLOCK + t], uy = s_u[(k * 3 + 1) * BLOCK + t], uz = s_u[(k * 3 + 2) * BLOCK + t];
      float mn = __builtin_inff();
      for (int j = 0; j < K; ++j) {
        const float dx = mx - s_p[(j * 3) * BLOCK + t], dy = my - s_p[(j * 3 + 1) * BLOCK + t],
                    dz = mz - s_p[(j * 3 + 2) * BLOCK + t];
        const float ax = dx * ux, ay = dy * uy, az = dz * uz;
        const float v = sqrtf((dx * dx + dy * dy) + dz * dz) - ((ax * ax + ay * ay) + az * az);
        mn = v < mn ? v : mn;
      }
      mn = fabsf(mn);
      mn = sqrtf(mn > 1.0e-17f ? mn : 1.0e-17f);
      const float dotn = (nx * ux + ny * uy) + nz * uz;
      float edge = 2.0f - dotn;
      if (edge_sensitivity == 2.0f) edge = edge * edge;
      else if (edge_sensitivity != 1.0f) edge = powf(edge, edge_sensitivity);
      const float val = edge * mn;
      if (val > best) { best = val; bx = mx; by = my; bz = mz; }
    }
    sparsity[i] = best;
    cand[i * 3] = bx; cand[i * 3 + 1] = by; cand[i * 3 + 2] = bz;
  }
}

}  // namespace

extern "C" int iso_ear_candidates(const float* points, const float* normals, const float* knn,
                                  const float* knn_normals, int64_t n, int K, float edge_sensitivity,
                                  float* sparsity_out, float* candidates_out, void* stream) {
  ISO_REQUIRE(n >= 0 && K >= 1 && K <= 64, ISO_ERR_INVALID, "iso_ear_candidates: bad sizes (1 <= K <= 64)");
  if (n == 0) return ISO_OK;
  ISO_REQUIRE(points && normals && knn && knn_normals && sparsity_out && candidates_out, ISO_ERR_INVALID,
              "iso_ear_candidates: null pointer");
  constexpr int BLOCK = 64;
  const size_t lds = (size_t)K * 6 * BLOCK * sizeof(float);   // <= 96 KiB at K = 64
  static bool raised = false;
  if (!raised) {
    if (hipFuncSetAttribute((const void*)k_ear_candidates<BLOCK>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            64 * 6 * BLOCK * (int)sizeof(float)) != hipSuccess) {
      iso_set_error("iso_ear_candidates: cannot raise the dynamic LDS limit");
      return ISO_ERR_LAUNCH;
    }
    raised = true;
  }
  hipLaunchKernelGGL(k_ear_candidates<BLOCK>, dim3(iso_stream_grid(n, BLOCK)), dim3(BLOCK), lds,
                     (hipStream_t)stream, points, normals, knn, knn_normals, n, K, edge_sensitivity,
                     sparsity_out, candidates_out);
  ISO_CHECK_LAUNCH("iso_ear_candidates");
  return ISO_OK;
}
