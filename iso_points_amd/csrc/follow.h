// Side work of a kernel that writes FINAL point positions (include/isopoints.h: iso_follow): what the next stages of the
// iso-point cycle would otherwise do in passes of their own over the points just written --
//   * bounding box of the result -> the PENDING BOX of a brick workspace (bricks.h): replaces k_brick_bbox /
//     iso_points_bbox; the header is made from it by the count pass of iso_bricks_build_pending;
//   * renderable mask per point + the number of renderable points per 256-point tile and view: replaces
//     k_view_mask_chunks (splat.hip; same expressions); the scan rides in that build's count launch (chunk_scan_job).
// No grid-wide hand-over inside the launch: a workgroup only adds to accumulators / writes its own tile's words.
// Reference: UniformProjection._create_tree (levelset_sampling.py:110-140, bbox of the cloud);
// SurfaceSplatting._filter_points_with_invalid_depth / backface culling (DSS/core/rasterizer.py:184-254).
#pragma once
#include "bricks.h"

constexpr int kFollowTile = 256;        // points per tile of the count table (four tiles = one chunk of the front end)

struct FollowArgs {                     // device-side form of iso_follow
  int32_t* counters;                    // the grid workspace's counter block (pending box)
  const float* views; int n_views; float znear, zfar; int backface;
  int32_t* mask_out; int32_t* tile_cnt; int n_tiles;
};

BrickWs bricks_carve(void* ws, int64_t n_max);

// host: validate *f for a launch over n points and translate it
static inline int follow_args(const iso_follow& f, int64_t n, bool have_normals, const char* who, FollowArgs& a) {
  ISO_REQUIRE(f.grid_ws && f.grid_n_max >= n, ISO_ERR_INVALID, "%s: follow needs an initialised brick workspace with n_max >= n", who);
  ISO_REQUIRE(((uintptr_t)f.grid_ws & 255) == 0, ISO_ERR_INVALID, "%s: brick workspace must be 256-B aligned", who);
  a.counters = bricks_carve(f.grid_ws, f.grid_n_max).counters;
  a.views = f.views; a.n_views = 0; a.znear = f.znear; a.zfar = f.zfar; a.backface = f.backface_culling;
  a.mask_out = nullptr; a.tile_cnt = nullptr; a.n_tiles = 0;
  if (f.views) {
    ISO_REQUIRE(f.n_views >= 1 && f.n_views <= 8, ISO_ERR_UNSUPPORTED, "%s: follow: 1..8 views per call", who);
    ISO_REQUIRE(f.front_ws && (n == 0 || f.mask_out) && (have_normals || !f.backface_culling),
                ISO_ERR_INVALID, "%s: follow: null pointer in the mask part", who);
    ISO_REQUIRE(f.front_ws_bytes >= iso_splat_front_workspace_bytes(n), ISO_ERR_WORKSPACE, "%s: follow.front_ws too small", who);
    a.n_views = f.n_views;
    a.mask_out = f.mask_out;
    a.n_tiles = (int)(((n + kFollowTile - 1) / kFollowTile + 3) / 4 * 4);   // row stride of the table: a multiple of four (= 4 n_chunks)
    a.tile_cnt = (int32_t*)f.front_ws + 8 * ((n + 1023) / 1024 + 1);       // behind the chunk table (front_tile_table)
  }
  return ISO_OK;
}

#ifdef __HIPCC__
// One per thread.  A thread reports at most ONE point of tile `tile` with point(), then tile_done(tile) is called by ALL
// threads of the workgroup, and finish() once at the end by all threads.
// NV: views the state is built for (0: no mask part; the view constants live in scalar registers -- eight views' worth
// spilled 70 of them in a kernel that only wanted the box).
template <int NV>
struct FollowState {
  BkBox box;
  unsigned seen;                        // the pending box as it was when the workgroup started (BkBox::peek)
  float vz[NV > 0 ? NV : 1][4];         // column 2 of every view matrix: z_view = [p, 1] . vz

  __device__ __forceinline__ void init(const FollowArgs& a) {
    box.init();
    seen = BkBox::peek(a.counters);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
#pragma unroll
      for (int r = 0; r < 4; ++r) vz[v][r] = (a.views && v < a.n_views) ? a.views[v * 16 + 4 * r + 2] : 0.f;
    }
  }
  // final position p and normal n of point i
  __device__ __forceinline__ void point(const FollowArgs& a, int64_t i, float x, float y, float z, float nx, float ny, float nz) {
    box.add(x, y, z);
    if (NV == 0 || !a.views) return;
    int m = 0;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      if (v < a.n_views) {                                  // the expressions of k_view_mask_chunks (splat.hip)
        const float zv = ((x * vz[v][0] + y * vz[v][1]) + z * vz[v][2]) + vz[v][3];
        bool ok = (zv >= a.znear) && (zv <= a.zfar);
        if (a.backface) ok = ok && (((nx * vz[v][0] + ny * vz[v][1]) + nz * vz[v][2]) < 0.f);
        if (ok) m |= 1 << v;
      }
    }
    last_m = m;
    a.mask_out[i] = m;
  }
  // all threads, after the tile's points were reported (at most ONE per thread): the per-view counts of the tile.
  // s_cnt: [2][waves][8] ints of LDS (double buffered by the parity of the call: no barrier at the end)
  int parity = 0;
  int last_m = 0;
  __device__ __forceinline__ void tile_done(const FollowArgs& a, int64_t tile, int (*s_cnt)[8]) {
    if (NV == 0 || !a.views) return;
    const int nw = blockDim.x >> 6, w = threadIdx.x >> 6;
    int (*buf)[8] = s_cnt + parity * nw;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const unsigned long long bal = __ballot((last_m >> v) & 1);
      if ((threadIdx.x & 63) == 0) buf[w][v] = __popcll(bal);
    }
    last_m = 0;
    __syncthreads();
    if ((int)threadIdx.x < a.n_views) {
      int c = 0;
      for (int k = 0; k < nw; ++k) c += buf[k][threadIdx.x];
      a.tile_cnt[(int64_t)threadIdx.x * a.n_tiles + tile] = c;
    }
    parity ^= 1;
  }
  // s_box: one row of 6 floats per wave
  __device__ __forceinline__ void finish(const FollowArgs& a, float (*s_box)[6]) {
#ifndef FOLLOW_DBG_NOBOX       // (timing experiment, tools/build_variant.sh: results are then wrong by construction)
    box.commit(a.counters, s_box, seen);
#endif
  }
};
#endif  // __HIPCC__
