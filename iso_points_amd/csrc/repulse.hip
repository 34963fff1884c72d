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
