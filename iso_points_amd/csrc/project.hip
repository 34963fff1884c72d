// Newton level-set projection, analytic-SDF variant (HBM-bound: 37 B/point).
// Reference semantics: UniformProjection._project_points,
// DSS/models/levelset_sampling.py:313-342 (see include/isopoints.h section A).
//
// Layout: points are (n,3) f32 packed.  A 256-point tile (3 KiB) is moved
// HBM<->LDS with full-width coalesced accesses (16 B/lane when aligned) and
// each lane then picks its own xyz out of LDS (stride 3 dwords: odd, so
// conflict-free).  The whole T-iteration Newton loop runs in registers with a
// per-lane `active` flag; nothing goes back to HBM between iterations.
#include <stdlib.h>
#include "iso_common.h"
#include "iso_tile.h"
#include "iso_newton.h"
#include "bricks.h"
#include "follow.h"

namespace {

struct SphereSdf {
  float cx, cy, cz, radius;
  __device__ __forceinline__ void eval(float px, float py, float pz, float& f,
                                       float& gx, float& gy, float& gz) const {
    float dx = px - cx, dy = py - cy, dz = pz - cz;
    float r = sqrtf((dx * dx + dy * dy) + dz * dz);
    f = r - radius;
    float rr = r > 1e-30f ? r : 1e-30f;
    gx = dx / rr;
    gy = dy / rr;
    gz = dz / rr;
  }
};

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_project_sphere(
    const float* __restrict__ pts_in, float* __restrict__ pts_out,
    float* __restrict__ nrm_out, uint8_t* __restrict__ mask_out, int64_t n,
    SphereSdf sdf, int max_iters, float tol) {
  __shared__ __attribute__((aligned(16))) float tile[BLOCK * 3];
  const int t = threadIdx.x;
  const int64_t n_tiles = (n + BLOCK - 1) / BLOCK;
  for (int64_t tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
    const int64_t base = tl * BLOCK;
    const int cnt = (int)((n - base) < BLOCK ? (n - base) : BLOCK);
    iso_tile_load3<BLOCK>(pts_in, base, cnt, tile);
    __syncthreads();
    float px = 0.f, py = 0.f, pz = 0.f, nx = 0.f, ny = 0.f, nz = 0.f;
    bool conv = false;
    if (t < cnt) {
      px = tile[3 * t + 0];
      py = tile[3 * t + 1];
      pz = tile[3 * t + 2];
      for (int it = 0;; ++it) {
        float f;
        sdf.eval(px, py, pz, f, nx, ny, nz);
        if (!(fabsf(f) > tol)) { conv = true; break; }
        if (it == max_iters) break;
        iso_newton_move(f, nx, ny, nz, px, py, pz);
      }
    }
    __syncthreads();
    if (t < cnt) { tile[3 * t] = px; tile[3 * t + 1] = py; tile[3 * t + 2] = pz; }
    __syncthreads();
    iso_tile_store3<BLOCK>(pts_out, base, cnt, tile);
    __syncthreads();
    if (t < cnt) { tile[3 * t] = nx; tile[3 * t + 1] = ny; tile[3 * t + 2] = nz; }
    __syncthreads();
    iso_tile_store3<BLOCK>(nrm_out, base, cnt, tile);
    if (t < cnt) mask_out[base + t] = conv ? 1 : 0;
    __syncthreads();
  }
}

// The same projection with the side work of follow.h (bounding box -> pending box of the grid workspace; renderable mask
// and per-tile counts), taken from the registers that hold the results.
template <int BLOCK, int NV>
__global__ __launch_bounds__(BLOCK) void k_project_sphere_follow(
    const float* __restrict__ pts_in, float* __restrict__ pts_out, float* __restrict__ nrm_out,
    uint8_t* __restrict__ mask_out, int64_t n, SphereSdf sdf, int max_iters, float tol, FollowArgs fa) {
  static_assert(BLOCK == kFollowTile, "one tile of the count table per workgroup round");
  __shared__ __attribute__((aligned(16))) float tile[BLOCK * 3];
  __shared__ float s_box[BLOCK / 64][6];
  __shared__ int s_cnt[2 * (BLOCK / 64)][8];
  const int t = threadIdx.x;
  const int64_t n_tiles = (n + BLOCK - 1) / BLOCK;
  FollowState<NV> fs;
  fs.init(fa);
  for (int64_t tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
    const int64_t base = tl * BLOCK;
    const int cnt = (int)((n - base) < BLOCK ? (n - base) : BLOCK);
    iso_tile_load3<BLOCK>(pts_in, base, cnt, tile);
    __syncthreads();
    float px = 0.f, py = 0.f, pz = 0.f, nx = 0.f, ny = 0.f, nz = 0.f;
    bool conv = false;
    if (t < cnt) {
      px = tile[3 * t + 0];
      py = tile[3 * t + 1];
      pz = tile[3 * t + 2];
      for (int it = 0;; ++it) {
        float f;
        sdf.eval(px, py, pz, f, nx, ny, nz);
        if (!(fabsf(f) > tol)) { conv = true; break; }
        if (it == max_iters) break;
        iso_newton_move(f, nx, ny, nz, px, py, pz);
      }
      fs.point(fa, base + t, px, py, pz, nx, ny, nz);
    }
    __syncthreads();
    if (t < cnt) { tile[3 * t] = px; tile[3 * t + 1] = py; tile[3 * t + 2] = pz; }
    __syncthreads();
    iso_tile_store3<BLOCK>(pts_out, base, cnt, tile);
    __syncthreads();
    if (t < cnt) { tile[3 * t] = nx; tile[3 * t + 1] = ny; tile[3 * t + 2] = nz; }
    __syncthreads();
    iso_tile_store3<BLOCK>(nrm_out, base, cnt, tile);
    if (t < cnt) mask_out[base + t] = conv ? 1 : 0;
    __syncthreads();
    fs.tile_done(fa, tl, s_cnt);
  }
  fs.finish(fa, s_box);
}

// Sphere tracing against the analytic sphere (SphereTracing.project_points,
// DSS/models/levelset_sampling.py:735-779): the whole loop in registers, 12+12 B in, 12+4+1 B out.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_trace_sphere(
    const float* __restrict__ ray0, const float* __restrict__ dirs, float* __restrict__ pts_out,
    float* __restrict__ sdf_out, uint8_t* __restrict__ mask_out, int64_t n, SphereSdf sdf,
    float alpha, float bound, int max_iters, float tol) {
  __shared__ __attribute__((aligned(16))) float tile[BLOCK * 3];
  const int t = threadIdx.x;
  const int64_t n_tiles = (n + BLOCK - 1) / BLOCK;
  const float tol_active = 0.1f * tol;
  for (int64_t tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
    const int64_t base = tl * BLOCK;
    const int cnt = (int)((n - base) < BLOCK ? (n - base) : BLOCK);
    float dx = 0.f, dy = 0.f, dz = 0.f;
    iso_tile_load3<BLOCK>(dirs, base, cnt, tile);
    __syncthreads();
    if (t < cnt) { dx = tile[3 * t]; dy = tile[3 * t + 1]; dz = tile[3 * t + 2]; }
    __syncthreads();
    iso_tile_load3<BLOCK>(ray0, base, cnt, tile);
    __syncthreads();
    float px = 0.f, py = 0.f, pz = 0.f, f = 0.f;
    if (t < cnt) {
      px = tile[3 * t + 0];
      py = tile[3 * t + 1];
      pz = tile[3 * t + 2];
      for (int it = 0;; ++it) {
        float gx, gy, gz;
        sdf.eval(px, py, pz, f, gx, gy, gz);
        if (!(fabsf(f) > tol_active) || it == max_iters) break;
        if (!iso_trace_move(f, dx, dy, dz, alpha, bound, px, py, pz)) break;   // left the sphere: frozen
      }
    }
    __syncthreads();
    if (t < cnt) { tile[3 * t] = px; tile[3 * t + 1] = py; tile[3 * t + 2] = pz; }
    __syncthreads();
    iso_tile_store3<BLOCK>(pts_out, base, cnt, tile);
    if (t < cnt) {
      sdf_out[base + t] = f;
      mask_out[base + t] = fabsf(f) <= tol ? 1 : 0;
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int iso_project_sphere(const float* pts_in, float* pts_out,
                                  float* normals_out, uint8_t* mask_out,
                                  int64_t n, float cx, float cy, float cz,
                                  float radius, int max_iters, float tol,
                                  void* stream) {
  ISO_REQUIRE(n >= 0, ISO_ERR_INVALID, "iso_project_sphere: n < 0");
  ISO_REQUIRE(max_iters >= 0, ISO_ERR_INVALID, "iso_project_sphere: max_iters < 0");
  if (n == 0) return ISO_OK;
  ISO_REQUIRE(pts_in && pts_out && normals_out && mask_out, ISO_ERR_INVALID,
              "iso_project_sphere: null pointer");
  constexpr int BLOCK = 256;
  SphereSdf sdf{cx, cy, cz, radius};
  hipLaunchKernelGGL(k_project_sphere<BLOCK>, dim3(iso_stream_grid(n, BLOCK)),
                     dim3(BLOCK), 0, (hipStream_t)stream, pts_in, pts_out,
                     normals_out, mask_out, n, sdf, max_iters, tol);
  ISO_CHECK_LAUNCH("iso_project_sphere");
  return ISO_OK;
}

extern "C" int iso_project_sphere_follow(const float* pts_in, float* pts_out, float* normals_out, uint8_t* mask_out,
                                         int64_t n, float cx, float cy, float cz, float radius, int max_iters,
                                         float tol, const iso_follow* f, void* stream) {
  ISO_REQUIRE(n >= 0 && max_iters >= 0, ISO_ERR_INVALID, "iso_project_sphere_follow: bad n / max_iters");
  ISO_REQUIRE(f, ISO_ERR_INVALID, "iso_project_sphere_follow: follow is NULL (use iso_project_sphere)");
  ISO_REQUIRE(n == 0 || (pts_in && pts_out && normals_out && mask_out), ISO_ERR_INVALID,
              "iso_project_sphere_follow: null pointer");
  FollowArgs fa;
  const int rc = follow_args(*f, n, normals_out != nullptr, "iso_project_sphere_follow", fa);
  if (rc != ISO_OK) return rc;
  if (n == 0) return ISO_OK;                             // (nothing to add to the pending box, no tile to count)
  constexpr int BLOCK = 256;
  SphereSdf sdf{cx, cy, cz, radius};
  // (one round of resident workgroups, each looping over its tiles: the box is committed once per workgroup)
  static const int cap = []() { const char* e = getenv("ISO_FOLLOW_GRID"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 2048; }();
  int follow_grid = iso_stream_grid(n, BLOCK);
  if (follow_grid > cap) follow_grid = cap;
#define ISO_PSF(NV)                                                                                                    \
  hipLaunchKernelGGL((k_project_sphere_follow<BLOCK, NV>), dim3(follow_grid), dim3(BLOCK), 0, (hipStream_t)stream, \
                     pts_in, pts_out, normals_out, mask_out, n, sdf, max_iters, tol, fa)
  if (fa.n_views == 0) ISO_PSF(0);
  else if (fa.n_views <= 4) ISO_PSF(4);
  else ISO_PSF(8);
#undef ISO_PSF
  ISO_CHECK_LAUNCH("iso_project_sphere_follow");
  return ISO_OK;
}

extern "C" int iso_trace_sphere(const float* ray0, const float* dirs, float* pts_out, float* sdf_out,
                                uint8_t* mask_out, int64_t n, float cx, float cy, float cz,
                                float radius, float alpha, float bound, int max_iters, float tol,
                                void* stream) {
  ISO_REQUIRE(n >= 0, ISO_ERR_INVALID, "iso_trace_sphere: n < 0");
  ISO_REQUIRE(max_iters >= 0, ISO_ERR_INVALID, "iso_trace_sphere: max_iters < 0");
  if (n == 0) return ISO_OK;
  ISO_REQUIRE(ray0 && dirs && pts_out && sdf_out && mask_out, ISO_ERR_INVALID,
              "iso_trace_sphere: null pointer");
  constexpr int BLOCK = 256;
  SphereSdf sdf{cx, cy, cz, radius};
  hipLaunchKernelGGL(k_trace_sphere<BLOCK>, dim3(iso_stream_grid(n, BLOCK)), dim3(BLOCK), 0,
                     (hipStream_t)stream, ray0, dirs, pts_out, sdf_out, mask_out, n, sdf, alpha, bound,
                     max_iters, tol);
  ISO_CHECK_LAUNCH("iso_trace_sphere");
  return ISO_OK;
}
