// Farthest-point sampling (exact, sequential by nature): one 1024-lane workgroup per cloud.
// Stands in for torch_cluster.fps as the reference's wlop calls it
// (DSS/utils/point_processing.py:473-499, :51).  Every iteration updates the running
// min-distance of all points to the sample set and takes the arg-max (ties -> lowest index)
// by wave shuffles + one LDS exchange.  The min-distance array lives in a caller workspace
// (L2 resident for the reference's cloud sizes, 5k-50k points).
#include <float.h>
#include "iso_common.h"

#pragma clang fp contract(off)

namespace {

constexpr int FPS_BLOCK = 1024;

__global__ __launch_bounds__(FPS_BLOCK) void k_fps(const float* __restrict__ pts,
                                                   const int64_t* __restrict__ lengths,
                                                   const int64_t* __restrict__ n_samples,
                                                   const int64_t* __restrict__ start, int64_t p_stride,
                                                   int64_t out_stride, float* __restrict__ work,
                                                   int64_t* __restrict__ out_idx) {
  __shared__ float s_d[FPS_BLOCK / 64];
  __shared__ int s_i[FPS_BLOCK / 64];
  __shared__ int s_cur;
  const int n = blockIdx.x;
  const int64_t len = lengths ? lengths[n] : p_stride;
  const int64_t ns = n_samples[n] < len ? n_samples[n] : len;
  const float* p = pts + (int64_t)n * p_stride * 3;
  float* mind = work + (int64_t)n * p_stride;
  int64_t* out = out_idx + (int64_t)n * out_stride;
  const int t = threadIdx.x;
  if (len <= 0 || ns <= 0) return;
  for (int64_t i = t; i < len; i += FPS_BLOCK) mind[i] = FLT_MAX;
  int cur = (int)(start[n] < len ? (start[n] < 0 ? 0 : start[n]) : len - 1);
  __syncthreads();
  for (int64_t s = 0; s < ns; ++s) {
    if (t == 0) out[s] = cur;
    const float cx = p[(int64_t)cur * 3], cy = p[(int64_t)cur * 3 + 1], cz = p[(int64_t)cur * 3 + 2];
    float bd = -1.0f;
    int bi = 0x7fffffff;
    for (int64_t i = t; i < len; i += FPS_BLOCK) {
      const float dx = p[i * 3] - cx, dy = p[i * 3 + 1] - cy, dz = p[i * 3 + 2] - cz;
      const float d = (dx * dx + dy * dy) + dz * dz;
      const float m = fminf(mind[i], d);
      mind[i] = m;
      if (m > bd) { bd = m; bi = (int)i; }   // i ascending per lane: first maximum kept
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float od = __shfl_xor(bd, o);
      const int oi = __shfl_xor(bi, o);
      if (od > bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
    }
    if ((t & 63) == 0) { s_d[t >> 6] = bd; s_i[t >> 6] = bi; }
    __syncthreads();
    if (t == 0) {
      float d = s_d[0];
      int i = s_i[0];
      for (int w = 1; w < FPS_BLOCK / 64; ++w)
        if (s_d[w] > d || (s_d[w] == d && s_i[w] < i)) { d = s_d[w]; i = s_i[w]; }
      s_cur = i;
    }
    __syncthreads();
    cur = s_cur;
    __syncthreads();
  }
}

}  // namespace

extern "C" int iso_farthest_point_sampling(const float* points, const int64_t* lengths,
                                           const int64_t* n_samples, const int64_t* start,
                                           int n_clouds, int64_t p_stride, int64_t out_stride,
                                           float* work, int64_t* out_idx, void* stream) {
  ISO_REQUIRE(n_clouds >= 0 && p_stride >= 0 && out_stride >= 0, ISO_ERR_INVALID,
              "iso_farthest_point_sampling: bad sizes");
  if (n_clouds == 0 || p_stride == 0 || out_stride == 0) return ISO_OK;
  ISO_REQUIRE(points && n_samples && start && work && out_idx, ISO_ERR_INVALID,
              "iso_farthest_point_sampling: null pointer");
  ISO_REQUIRE(p_stride < (1ll << 31), ISO_ERR_UNSUPPORTED, "iso_farthest_point_sampling: cloud too large");
  hipLaunchKernelGGL(k_fps, dim3(n_clouds), dim3(FPS_BLOCK), 0, (hipStream_t)stream, points, lengths,
                     n_samples, start, p_stride, out_stride, work, out_idx);
  ISO_CHECK_LAUNCH("iso_farthest_point_sampling");
  return ISO_OK;
}
