// Nearest point to a ray, brute force, no (R,M) matrix in memory.
//
// Reference: CombinedModel.sample_offsurface_using_isopoints, DSS/models/combined_modeling.py:336-352
// ("TODO: faster search"): for R camera rays and M points
//     pC      = p - C                                   (M,3)
//     ray_sq  = (sum_c pC_c * ray_c)^2                  (R,M)
//     dist    = sum_c pC_c^2 - ray_sq                   (R,M)   squared point-to-line distance
//     nn      = argmin_m dist;   ray_len = ray_sq[r, nn]
// It materialises two (R,M) f32 matrices per cloud (3.3 GB for 4096 rays x 100 k points, twice)
// and runs torch.topk over them.  Here one thread owns a ray and walks the points, which every
// workgroup stages through LDS as (pC, |pC|^2) records (one broadcast ds_read_b128 per point and
// wave); the point range is split over blockIdx.y and the partial minima meet in a 64-bit
// atomicMin on (order-preserving distance bits, index) -- ties go to the lowest index.  ~10 VALU
// ops per (ray, point): 4096 x 100 k pairs take ~0.1 ms.  Arithmetic as the reference's
// (f32, no contraction, ((x+y)+z) sums).
#include "iso_common.h"

namespace {

constexpr int kRayBlock = 256;
constexpr int kPtTile = 1024;

__device__ __forceinline__ uint32_t order_bits(float v) {
  const uint32_t u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ void k_ray_keys_init(unsigned long long* keys, int64_t R) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < R) keys[i] = ~0ull;
}

__global__ __launch_bounds__(kRayBlock) void k_ray_nearest(
    const float* __restrict__ rays, int64_t R, float ox, float oy, float oz,
    const float* __restrict__ pts, int64_t M, int64_t per_slice, unsigned long long* keys) {
  __shared__ float4 tile[kPtTile];
  const int64_t r = (int64_t)blockIdx.x * kRayBlock + threadIdx.x;
  float rx = 0.f, ry = 0.f, rz = 0.f;
  if (r < R) { rx = rays[r * 3]; ry = rays[r * 3 + 1]; rz = rays[r * 3 + 2]; }
  const int64_t m0 = (int64_t)blockIdx.y * per_slice;
  const int64_t m1 = m0 + per_slice < M ? m0 + per_slice : M;
  float best = __builtin_inff();
  int bi = -1;
  for (int64_t base = m0; base < m1; base += kPtTile) {
    const int cnt = (int)(m1 - base < kPtTile ? m1 - base : kPtTile);
    __syncthreads();
    for (int i = threadIdx.x; i < cnt; i += kRayBlock) {
      const int64_t m = base + i;
      const float x = pts[m * 3] - ox, y = pts[m * 3 + 1] - oy, z = pts[m * 3 + 2] - oz;
      tile[i] = make_float4(x, y, z, (x * x + y * y) + z * z);
    }
    __syncthreads();
#pragma unroll 4
    for (int i = 0; i < cnt; ++i) {
      const float4 p = tile[i];
      const float proj = (p.x * rx + p.y * ry) + p.z * rz;
      const float d = p.w - proj * proj;
      if (d < best) { best = d; bi = (int)(base + i); }
    }
  }
  if (r < R && bi >= 0)
    atomicMin(keys + r, ((unsigned long long)order_bits(best) << 32) | (unsigned)bi);
}

__global__ void k_ray_nearest_finish(const float* __restrict__ rays, int64_t R, float ox, float oy,
                                     float oz, const float* __restrict__ pts,
                                     const unsigned long long* __restrict__ keys,
                                     int32_t* __restrict__ idx_out, float* __restrict__ raysq_out,
                                     float* __restrict__ dist_out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const unsigned long long k = keys[r];
  if (k == ~0ull) {
    idx_out[r] = -1; raysq_out[r] = 0.f;
    if (dist_out) dist_out[r] = __builtin_inff();
    return;
  }
  const int64_t m = (int64_t)(k & 0xffffffffull);
  const float x = pts[m * 3] - ox, y = pts[m * 3 + 1] - oy, z = pts[m * 3 + 2] - oz;
  const float proj = (x * rays[r * 3] + y * rays[r * 3 + 1]) + z * rays[r * 3 + 2];
  idx_out[r] = (int32_t)m;
  raysq_out[r] = proj * proj;
  if (dist_out) dist_out[r] = ((x * x + y * y) + z * z) - proj * proj;
}

}  // namespace

extern "C" int64_t iso_ray_nearest_point_workspace_bytes(int64_t n_rays) {
  return (n_rays < 0 ? 0 : n_rays) * 8 + 64;
}

extern "C" int iso_ray_nearest_point(const float* rays, int64_t n_rays, float ox, float oy, float oz,
                                     const float* points, int64_t n_points, int32_t* idx_out,
                                     float* raysq_out, float* dist_out, void* workspace,
                                     int64_t workspace_bytes, void* stream) {
  ISO_REQUIRE(n_rays >= 0 && n_points >= 0, ISO_ERR_INVALID, "iso_ray_nearest_point: negative size");
  if (n_rays == 0) return ISO_OK;
  ISO_REQUIRE(n_points < (1ll << 31), ISO_ERR_UNSUPPORTED, "iso_ray_nearest_point: n_points must fit int32");
  ISO_REQUIRE(rays && idx_out && raysq_out && workspace && (points || n_points == 0), ISO_ERR_INVALID,
              "iso_ray_nearest_point: null pointer");
  ISO_REQUIRE(workspace_bytes >= iso_ray_nearest_point_workspace_bytes(n_rays), ISO_ERR_WORKSPACE,
              "iso_ray_nearest_point: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  unsigned long long* keys = (unsigned long long*)workspace;
  const int rb = iso_div_up(n_rays, kRayBlock);
  hipLaunchKernelGGL(k_ray_keys_init, dim3(iso_div_up(n_rays, 256)), dim3(256), 0, s, keys, n_rays);
  if (n_points > 0) {
    // enough (ray block, point slice) workgroups to fill 256 CUs a few times; slices of whole tiles
    int64_t slices = (256 * 8 + rb - 1) / rb;
    const int64_t tiles = (n_points + kPtTile - 1) / kPtTile;
    if (slices > tiles) slices = tiles;
    if (slices < 1) slices = 1;
    if (slices > 65535) slices = 65535;
    const int64_t per_slice = ((tiles + slices - 1) / slices) * kPtTile;
    slices = (n_points + per_slice - 1) / per_slice;
    hipLaunchKernelGGL(k_ray_nearest, dim3(rb, (unsigned)slices), dim3(kRayBlock), 0, s, rays, n_rays, ox, oy,
                       oz, points, n_points, per_slice, keys);
  }
  hipLaunchKernelGGL(k_ray_nearest_finish, dim3(iso_div_up(n_rays, 256)), dim3(256), 0, s, rays, n_rays, ox,
                     oy, oz, points, keys, idx_out, raysq_out, dist_out);
  ISO_CHECK_LAUNCH("iso_ray_nearest_point");
  return ISO_OK;
}
