// Newton level-set projection, analytic-SDF variant (HBM-bound: 37 B/point).
// Reference semantics: UniformProjection._project_points,
// DSS/models/levelset_sampling.py:313-342 (see include/isopoints.h section A).
//
// Layout: points are (n,3) f32 packed.  A 256-point tile (3 KiB) is moved
// HBM<->LDS with full-width coalesced accesses (16 B/lane when aligned) and
// each lane then picks its own xyz out of LDS (stride 3 dwords: odd, so
// conflict-free).  The whole T-iteration Newton loop runs in registers with a
// per-lane `active` flag; nothing goes back to HBM between iterations.
#include "iso_common.h"
#include "iso_tile.h"
#include "iso_newton.h"

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
