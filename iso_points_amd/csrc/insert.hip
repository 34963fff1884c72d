// Loss-weighted insertion of iso-points (UniformProjection.insert, DSS/models/levelset_sampling.py:172-233) as two
// kernels around one prefix sum: which points get children, and the children themselves.  Everything the
// reference decides on the host (threshold of the metric, size of the selected set, the radius) arrives as
// device-side scalars, so the call has no host read before its result sizes are needed.
#include "iso_common.h"

namespace {

constexpr int kMaxRefs = 64;          // the reference keeps at most min(50, P_ref / 20) selected points (:189-194)

// A point fathers children when the NEAREST selected reference point lies within the query radius and
// 0 < d^2 < 4 * spacing^2 (K = 1 query of radius 4 r, then :204-206).  params: [0] (4 r)^2, [1] 4 * spacing^2.
__global__ void k_insert_fathers(const float* __restrict__ pts, const int64_t* __restrict__ lengths, int64_t P,
                                 const float* __restrict__ refs, const int32_t* __restrict__ n_refs,
                                 const float* __restrict__ params, uint8_t* __restrict__ father) {
  __shared__ float s_ref[kMaxRefs * 3];
  const int b = blockIdx.y;
  const int nr = min(*n_refs, kMaxRefs);
  for (int i = threadIdx.x; i < nr * 3; i += blockDim.x) s_ref[i] = refs[i];
  __syncthreads();
  const float r2 = params[0], lim = params[1];
  const int64_t len = lengths ? lengths[b] : P;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
    uint8_t f = 0;
    if (i < len) {
      const float* p = pts + ((int64_t)b * P + i) * 3;
      const float x = p[0], y = p[1], z = p[2];
      float best = INFINITY;
      for (int r = 0; r < nr; ++r) {
        const float dx = x - s_ref[r * 3], dy = y - s_ref[r * 3 + 1], dz = z - s_ref[r * 3 + 2];
        const float d = (dx * dx + dy * dy) + dz * dz;
        best = d < best ? d : best;
      }
      f = (best < r2 && best < lim && best > 0.f) ? 1 : 0;
    }
    father[(int64_t)b * P + i] = f;
  }
}

// child (rank of the father among its cloud's fathers, k) = 2 father / 3 + neighbour_k / 3 (:209); a missing
// neighbour (index < 0) counts as the origin, as frnn_gather returns it.  out: cloud b starts at row out_first[b].
__global__ void k_insert_children(const float* __restrict__ pts, const int64_t* __restrict__ knn, int64_t P, int K,
                                  int k0, int patch, const uint8_t* __restrict__ father,
                                  const int64_t* __restrict__ rank_incl, const int64_t* __restrict__ out_first,
                                  float* __restrict__ out) {
  const int b = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t g = (int64_t)b * P + i;
    if (!father[g]) continue;
    const int64_t row0 = out_first[b] + (rank_incl[g] - 1) * patch;
    const float fx = pts[g * 3], fy = pts[g * 3 + 1], fz = pts[g * 3 + 2];
    for (int k = 0; k < patch; ++k) {
      const int64_t j = knn[g * K + k0 + k];
      float mx = 0.f, my = 0.f, mz = 0.f;
      if (j >= 0) { const float* m = pts + ((int64_t)b * P + j) * 3; mx = m[0]; my = m[1]; mz = m[2]; }
      float* o = out + (row0 + k) * 3;
      o[0] = 2.0f * fx / 3.0f + mx / 3.0f;
      o[1] = 2.0f * fy / 3.0f + my / 3.0f;
      o[2] = 2.0f * fz / 3.0f + mz / 3.0f;
    }
  }
}

}  // namespace

extern "C" int iso_insert_fathers(const float* points, const int64_t* lengths, int n_clouds, int64_t max_points,
                                  const float* refs, const int32_t* n_refs, const float* params, uint8_t* father_out,
                                  void* stream) {
  ISO_REQUIRE(n_clouds >= 0 && max_points >= 0, ISO_ERR_INVALID, "iso_insert_fathers: bad sizes");
  if (n_clouds == 0 || max_points == 0) return ISO_OK;
  ISO_REQUIRE(points && refs && n_refs && params && father_out, ISO_ERR_INVALID, "iso_insert_fathers: null pointer");
  hipLaunchKernelGGL(k_insert_fathers, dim3(iso_stream_grid(max_points, 256), n_clouds), dim3(256), 0, (hipStream_t)stream,
                     points, lengths, max_points, refs, n_refs, params, father_out);
  ISO_CHECK_LAUNCH("iso_insert_fathers");
  return ISO_OK;
}

extern "C" int iso_insert_children(const float* points, const int64_t* knn_idx, int n_clouds, int64_t max_points, int K,
                                   int patch, const uint8_t* father, const int64_t* rank_inclusive,
                                   const int64_t* out_first, float* children_out, void* stream) {
  ISO_REQUIRE(n_clouds >= 0 && max_points >= 0 && patch >= 1 && K >= patch, ISO_ERR_INVALID, "iso_insert_children: bad sizes");
  if (n_clouds == 0 || max_points == 0) return ISO_OK;
  ISO_REQUIRE(points && knn_idx && father && rank_inclusive && out_first && children_out, ISO_ERR_INVALID,
              "iso_insert_children: null pointer");
  hipLaunchKernelGGL(k_insert_children, dim3(iso_stream_grid(max_points, 256), n_clouds), dim3(256), 0, (hipStream_t)stream,
                     points, knn_idx, max_points, K, K - patch, patch, father, rank_inclusive, out_first, children_out);
  ISO_CHECK_LAUNCH("iso_insert_children");
  return ISO_OK;
}
