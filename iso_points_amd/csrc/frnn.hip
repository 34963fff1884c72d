// Fixed-radius nearest neighbours on a uniform grid, written for gfx950.
// Stands in for the third-party frnn / prefix_sum CUDA extensions the
// reference calls (see include/isopoints.h section B for the call sites).
//
// Pipeline (all sizes decided on the device, no host sync):
//   make_grid   : bbox (atomic min/max on order-preserving uint keys) -> cell
//                 size / resolution
//   insert      : cell id + arrival slot per point (atomicAdd on cell counter)
//   scan_cells  : exclusive prefix sum of the cell counters (wave scan via
//                 DPP shuffles, block scan via LDS, 3-phase across blocks)
//   counting_sort: scatter points into cell order
//   query       : one lane per query point walks Chebyshev rings of cells
//                 around its own cell, keeps the K best (d2, idx) pairs in
//                 registers, stops as soon as the ring guarantee covers the
//                 K-th distance or the radius.  Result = exact K nearest
//                 within r, so it is independent of the cell size.
//
// Distances are d2 = (dx*dx + dy*dy) + dz*dz in f32 with contraction off, the
// same expression the oracle evaluates, so neighbour lists are bit-exact.
#include <float.h>
#include "iso_common.h"

#pragma clang fp contract(off)

namespace {

// ---- order-preserving float <-> uint key ----------------------------------
__device__ __forceinline__ unsigned f2key(float f) {
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
  unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

__global__ void k_bbox_init(unsigned* __restrict__ keys, int n_clouds) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_clouds * 8) {
    int j = i & 7;
    keys[i] = (j < 3) ? 0xffffffffu : 0u;  // min slots: +max key, max slots: 0
  }
}

// keys[n][0..2] = min xyz, keys[n][4..6] = max xyz (uint keys).  Each workgroup reduces its
// grid-stride slice in registers -> wave shuffles -> LDS, and issues ONE set of six atomics
// (a few hundred atomics per cloud in total: same-address atomics serialise in L2).
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_bbox(const float* __restrict__ pts,
                                                const int64_t* __restrict__ lengths,
                                                int64_t p_stride,
                                                unsigned* __restrict__ keys) {
  __shared__ float s_mn[BLOCK / 64][3], s_mx[BLOCK / 64][3];
  const int n = blockIdx.y;
  const int64_t len = lengths ? lengths[n] : p_stride;
  if (len <= 0) return;
  const float* p = pts + (int64_t)n * p_stride * 3;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  // flat float index so consecutive lanes read consecutive dwords; the launcher makes the grid stride a
  // multiple of 3 floats, so a thread stays on one coordinate axis and keeps four loads in flight
  const int64_t nfl = len * 3;
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  const int64_t i_first = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (stride % 3 == 0) {
    float lo = FLT_MAX, hi = -FLT_MAX;
    int64_t i = i_first;
    for (; i + 3 * stride < nfl; i += 4 * stride) {
      const float v0 = p[i], v1 = p[i + stride], v2 = p[i + 2 * stride], v3 = p[i + 3 * stride];
      lo = fminf(fminf(lo, v0), fminf(v1, fminf(v2, v3)));
      hi = fmaxf(fmaxf(hi, v0), fmaxf(v1, fmaxf(v2, v3)));
    }
    for (; i < nfl; i += stride) { const float v = p[i]; lo = fminf(lo, v); hi = fmaxf(hi, v); }
    const int a = (int)(i_first % 3);
#pragma unroll
    for (int c = 0; c < 3; ++c) if (a == c) { mn[c] = lo; mx[c] = hi; }
  } else {
    for (int64_t i = i_first; i < nfl; i += stride) {
      const float v = p[i];
      const int a = (int)(i % 3);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (a == c) { mn[c] = fminf(mn[c], v); mx[c] = fmaxf(mx[c], v); }
      }
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mn[a] = fminf(mn[a], __shfl_xor(mn[a], o));
      mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o));
    }
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { s_mn[w][a] = mn[a]; s_mx[w][a] = mx[a]; }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int a = threadIdx.x;
    float lo = s_mn[0][a], hi = s_mx[0][a];
#pragma unroll
    for (int k = 1; k < BLOCK / 64; ++k) { lo = fminf(lo, s_mn[k][a]); hi = fmaxf(hi, s_mx[k][a]); }
    atomicMin(&keys[n * 8 + a], f2key(lo));
    atomicMax(&keys[n * 8 + 4 + a], f2key(hi));
  }
}

// decode the keys in place: out[n] = [min xyz, 0, max xyz, 0] as floats
__global__ void k_bbox_decode(unsigned* __restrict__ keys, const int64_t* __restrict__ lengths,
                              int64_t p_stride, int n_clouds) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_clouds) return;
  const int64_t len = lengths ? lengths[n] : p_stride;
  float* f = reinterpret_cast<float*>(keys + n * 8);
  for (int a = 0; a < 3; ++a) {
    float lo = key2f(keys[n * 8 + a]), hi = key2f(keys[n * 8 + 4 + a]);
    if (len <= 0) { lo = 0.f; hi = 0.f; }
    f[a] = lo; f[4 + a] = hi;
  }
  f[3] = 0.f; f[7] = 0.f;
}

__global__ void k_grid_finalize(float* __restrict__ params,
                                const int64_t* __restrict__ lengths,
                                int64_t p_stride,
                                const float* __restrict__ radius, int n_clouds, int max_res,
                                float cells_per_point) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_clouds) return;
  unsigned* keys = reinterpret_cast<unsigned*>(params + n * 8);
  const int64_t len = lengths ? lengths[n] : p_stride;
  float mn[3], mx[3];
  for (int a = 0; a < 3; ++a) {
    mn[a] = key2f(keys[a]);
    mx[a] = key2f(keys[4 + a]);
  }
  if (len <= 0) {
    for (int a = 0; a < 3; ++a) { mn[a] = 0.f; mx[a] = 0.f; }
  }
  float ext[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
  float emax = fmaxf(ext[0], fmaxf(ext[1], ext[2]));
  float r = radius[n];
  // density-driven resolution: ~1/cells_per_point (default 8) points per occupied cell on a 2-manifold
  float nres = ceilf(sqrtf((float)len * cells_per_point));
  nres = fminf(fmaxf(nres, 1.f), (float)max_res);
  float cell = emax / nres;
  float half_r = 0.5f * r;
  if (half_r < cell) cell = fmaxf(half_r, emax / (float)max_res);
  if (!(cell > 1e-12f)) cell = 1.0f;  // degenerate cloud: one cell
  float res[3];
  float total = 1.f;
  for (int a = 0; a < 3; ++a) {
    res[a] = floorf(ext[a] / cell) + 1.f;
    res[a] = fminf(res[a], (float)(max_res + 1));
    total *= res[a];
  }
  params[n * 8 + 0] = mn[0];
  params[n * 8 + 1] = mn[1];
  params[n * 8 + 2] = mn[2];
  params[n * 8 + 3] = 1.0f / cell;
  params[n * 8 + 4] = res[0];
  params[n * 8 + 5] = res[1];
  params[n * 8 + 6] = res[2];
  params[n * 8 + 7] = total;
}

// ---- cell coordinate ------------------------------------------------------
__device__ __forceinline__ int cell_coord(float p, float mn, float delta, int res) {
  int c = (int)floorf((p - mn) * delta);
  return c < 0 ? 0 : (c >= res ? res - 1 : c);
}

template <int DIM>
__global__ void k_insert(const float* __restrict__ pts,
                         const int64_t* __restrict__ lengths,
                         const float* __restrict__ params,
                         int32_t* __restrict__ cnt, int32_t* __restrict__ cell,
                         int32_t* __restrict__ slot, int64_t p_stride,
                         int64_t g_stride) {
  constexpr int NP = (DIM == 3) ? ISO_GRID3_PARAMS : ISO_GRID2_PARAMS;
  const int n = blockIdx.y;
  const int64_t len = lengths ? lengths[n] : p_stride;
  const float* gp = params + n * NP;
  const float delta = gp[DIM];
  int res[DIM];
  float mn[DIM];
#pragma unroll
  for (int a = 0; a < DIM; ++a) { mn[a] = gp[a]; res[a] = (int)gp[DIM + 1 + a]; }
  const float* p = pts + (int64_t)n * p_stride * DIM;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len;
       i += (int64_t)gridDim.x * blockDim.x) {
    int c = 0;
#pragma unroll
    for (int a = 0; a < DIM; ++a)
      c = c * res[a] + cell_coord(p[i * DIM + a], mn[a], delta, res[a]);
    cell[n * p_stride + i] = c;
    slot[n * p_stride + i] = atomicAdd(&cnt[n * g_stride + c], 1);
  }
}

template <int DIM>
__global__ void k_counting_sort(const float* __restrict__ pts,
                                const int64_t* __restrict__ lengths,
                                const int32_t* __restrict__ cell,
                                const int32_t* __restrict__ slot,
                                const int32_t* __restrict__ off,
                                float* __restrict__ sorted,
                                int32_t* __restrict__ sorted_idx,
                                int64_t p_stride, int64_t g_stride) {
  const int n = blockIdx.y;
  const int64_t len = lengths ? lengths[n] : p_stride;
  const float* p = pts + (int64_t)n * p_stride * DIM;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len;
       i += (int64_t)gridDim.x * blockDim.x) {
    int c = cell[n * p_stride + i];
    int64_t dst = (int64_t)off[n * g_stride + c] + slot[n * p_stride + i];
#pragma unroll
    for (int a = 0; a < DIM; ++a)
      sorted[(n * p_stride + dst) * DIM + a] = p[i * DIM + a];
    sorted_idx[n * p_stride + dst] = (int32_t)i;
  }
}

// ---- exclusive scan -------------------------------------------------------
// inclusive scan across the 64 lanes of a wave
__device__ __forceinline__ int wave_incl_scan(int v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(v, o);
    if (lane >= o) v += t;
  }
  return v;
}

constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_CHUNK = SCAN_BLOCK * SCAN_ITEMS;  // 2048 counters per block

// block-wide exclusive scan of one int per thread; returns exclusive prefix,
// total in `total` (valid in all threads)
__device__ __forceinline__ int block_excl_scan(int v, int& total, int* lds /*>=5*/) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = wave_incl_scan(v);
  if (lane == 63) lds[w] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < SCAN_BLOCK / 64; ++i) {
    int s = lds[i];
    if (i < w) base += s;
    tot += s;
  }
  total = tot;
  __syncthreads();
  return base + inc - v;
}

__device__ __forceinline__ int64_t row_len(const float* params, int n, int dim,
                                           int64_t n_host, int64_t g_stride) {
  if (!params) return n_host;
  const int np = (dim == 3) ? ISO_GRID3_PARAMS : ISO_GRID2_PARAMS;
  int64_t t = (int64_t)params[n * np + np - 1];
  return t < g_stride ? t : g_stride;
}

// phase 1: per-chunk totals
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_sums(
    const int32_t* __restrict__ in, int32_t* __restrict__ sums,
    const float* __restrict__ params, int dim, int64_t n_host,
    int64_t row_stride, int chunks_per_row) {
  __shared__ int lds[8];
  const int n = blockIdx.y;
  const int64_t len = row_len(params, n, dim, n_host, row_stride);
  const int64_t c0 = (int64_t)blockIdx.x * SCAN_CHUNK;
  int v = 0;
  if (c0 < len) {
    const int32_t* row = in + (int64_t)n * row_stride;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
      int64_t i = c0 + threadIdx.x * SCAN_ITEMS + k;
      if (i < len) v += row[i];
    }
  }
  int tot;
  block_excl_scan(v, tot, lds);
  if (threadIdx.x == 0) sums[n * chunks_per_row + blockIdx.x] = tot;
}

// phase 2: exclusive scan of the chunk totals of one row (one block per row)
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_mid(int32_t* __restrict__ sums,
                                                        int chunks_per_row) {
  __shared__ int lds[8];
  int32_t* row = sums + blockIdx.x * chunks_per_row;
  int carry = 0;
  for (int c0 = 0; c0 < chunks_per_row; c0 += SCAN_BLOCK) {
    int i = c0 + threadIdx.x;
    int v = (i < chunks_per_row) ? row[i] : 0;
    int tot;
    int ex = block_excl_scan(v, tot, lds);
    if (i < chunks_per_row) row[i] = carry + ex;
    carry += tot;
  }
}

// phase 3: scan each chunk with its base
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_final(
    const int32_t* __restrict__ in, int32_t* __restrict__ out,
    const int32_t* __restrict__ sums, const float* __restrict__ params, int dim,
    int64_t n_host, int64_t row_stride, int chunks_per_row) {
  __shared__ int lds[8];
  const int n = blockIdx.y;
  const int64_t len = row_len(params, n, dim, n_host, row_stride);
  const int64_t c0 = (int64_t)blockIdx.x * SCAN_CHUNK;
  if (c0 >= len) return;
  const int32_t* row = in + (int64_t)n * row_stride;
  int32_t* orow = out + (int64_t)n * row_stride;
  int vals[SCAN_ITEMS];
  int v = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    int64_t i = c0 + threadIdx.x * SCAN_ITEMS + k;
    vals[k] = (i < len) ? row[i] : 0;
    v += vals[k];
  }
  int tot;
  int ex = block_excl_scan(v, tot, lds) + sums[n * chunks_per_row + blockIdx.x];
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    int64_t i = c0 + threadIdx.x * SCAN_ITEMS + k;
    if (i < len) orow[i] = ex;
    ex += vals[k];
  }
}

// ---- query ----------------------------------------------------------------
constexpr int kRingCap = 2;

__device__ __forceinline__ bool pair_lt(float d1, int i1, float d2, int i2) {
  return d1 < d2 || (d1 == d2 && i1 < i2);
}

template <int KMAX>
struct TopK {
  float d[KMAX];
  int id[KMAX];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int j = 0; j < KMAX; ++j) { d[j] = FLT_MAX; id[j] = 0x7fffffff; }
  }
  // keep the K smallest (d, id) pairs in ascending order
  __device__ __forceinline__ void push(float cd, int ci, int K) {
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      if (j < K && pair_lt(cd, ci, d[j], id[j])) {
        float td = d[j]; int ti = id[j];
        d[j] = cd; id[j] = ci;
        cd = td; ci = ti;
      }
    }
  }
  __device__ __forceinline__ float worst(int K) const {
    float w = FLT_MAX;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) if (j == K - 1) w = d[j];
    return w;
  }
  __device__ __forceinline__ int worst_id(int K) const {
    int w = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) if (j == K - 1) w = id[j];
    return w;
  }
};

template <int KMAX>
__global__ __launch_bounds__(256) void k_query(
    const float* __restrict__ points1, const int64_t* __restrict__ lengths1,
    const float* __restrict__ points2, const float* __restrict__ sorted2,
    const int32_t* __restrict__ sorted_idx2,
    const int64_t* __restrict__ lengths2, const int32_t* __restrict__ off,
    const float* __restrict__ params, const float* __restrict__ radius, int K,
    float* __restrict__ dists_out, int64_t* __restrict__ idxs_out,
    float* __restrict__ nn_out, int64_t p1_stride, int64_t p2_stride,
    int64_t g_stride, int32_t* __restrict__ tail_list, int32_t* __restrict__ tail_count,
    const float4* __restrict__ xyzi) {
  const int n = blockIdx.y;
  const bool self = (points1 == nullptr);
  const float4* s4 = xyzi + (int64_t)blockIdx.y * p2_stride;
  const int64_t len2 = lengths2 ? lengths2[n] : p2_stride;
  const int64_t len1 = self ? len2 : (lengths1 ? lengths1[n] : p1_stride);
  const float* gp = params + n * ISO_GRID3_PARAMS;
  const float mnx = gp[0], mny = gp[1], mnz = gp[2], delta = gp[3];
  const int rx = (int)gp[4], ry = (int)gp[5], rz = (int)gp[6];
  const int total = (int)gp[7];
  const float r = radius[n];
  const float r2 = r * r;
  const float cell = 1.0f / delta;
  const float* s2 = sorted2 + (int64_t)n * p2_stride * 3;
  const int32_t* sidx = sorted_idx2 + (int64_t)n * p2_stride;
  const int32_t* offn = off + (int64_t)n * g_stride;

  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < len1;
       t += (int64_t)gridDim.x * blockDim.x) {
    float qx, qy, qz;
    int64_t row;
    if (self) {
      qx = s2[t * 3]; qy = s2[t * 3 + 1]; qz = s2[t * 3 + 2];
      row = sidx[t];
    } else {
      const float* q = points1 + ((int64_t)n * p1_stride + t) * 3;
      qx = q[0]; qy = q[1]; qz = q[2];
      row = t;
    }
    TopK<KMAX> best;
    best.init();
    float wd = FLT_MAX;      // current K-th best (d2, idx)
    int wi = 0x7fffffff;
    bool unfinished = false;
    if (len2 > 0 && r > 0.f && qx == qx && qy == qy && qz == qz) {
      // unclamped integer cell of the query (may lie outside the grid)
      float fx = floorf((qx - mnx) * delta), fy = floorf((qy - mny) * delta),
            fz = floorf((qz - mnz) * delta);
      const float lim = 1.0e6f;
      int cx = (int)fminf(fmaxf(fx, -lim), lim);
      int cy = (int)fminf(fmaxf(fy, -lim), lim);
      int cz = (int)fminf(fmaxf(fz, -lim), lim);
      // rings needed so that rho*cell*(1-1e-3) >= r
      float rho_f = ceilf(r * delta * 1.0011f);
      // distance (in cells) from the query's cell to the grid box: rings
      // below that are empty
      int gapx = cx < 0 ? -cx : (cx >= rx ? cx - rx + 1 : 0);
      int gapy = cy < 0 ? -cy : (cy >= ry ? cy - ry + 1 : 0);
      int gapz = cz < 0 ? -cz : (cz >= rz ? cz - rz + 1 : 0);
      int rho0 = max(gapx, max(gapy, gapz));
      int span = max(rx, max(ry, rz)) + rho0;  // beyond this no cell exists
      int rho_max = (rho_f < (float)span) ? (int)rho_f : span;
      // A lane walks at most kRingCap rings itself; the rare query that is still open after
      // that (an isolated point, a huge radius) is handed to k_query_tail, where a whole
      // wave sweeps the remaining cell columns -- one slow lane would otherwise hold up its
      // wave for thousands of dependent loads.
      const int rho_stop = min(rho_max, rho0 + kRingCap);
      unfinished = rho_stop < rho_max;
      auto scan = [&](int64_t i0, int64_t i1) {
        // two candidates per trip: two independent 16-B loads in flight per lane
        for (int64_t i = i0; i < i1; i += 2) {
          const bool two = i + 1 < i1;
          const float4 ca = s4[i];
          const float4 cb = s4[two ? i + 1 : i];
          {
            float dx = qx - ca.x, dy = qy - ca.y, dz = qz - ca.z;
            float d2 = (dx * dx + dy * dy) + dz * dz;
            if (d2 < r2 && d2 <= wd) {
              int oi = __float_as_int(ca.w);
              if (pair_lt(d2, oi, wd, wi)) {
                best.push(d2, oi, K);
                wd = best.worst(K);
                wi = best.worst_id(K);
              }
            }
          }
          if (two) {
            float dx = qx - cb.x, dy = qy - cb.y, dz = qz - cb.z;
            float d2 = (dx * dx + dy * dy) + dz * dz;
            if (d2 < r2 && d2 <= wd) {
              int oi = __float_as_int(cb.w);
              if (pair_lt(d2, oi, wd, wi)) {
                best.push(d2, oi, K);
                wd = best.worst(K);
                wi = best.worst_id(K);
              }
            }
          }
        }
      };
      int rho_first = rho0;
      bool done = false;
      if (rho0 == 0 && rho_stop >= 1) {
        // rings 0 and 1 together: the 3x3x3 block is nine z-runs, each one contiguous range of the
        // sorted array (the common case ends here)
        const int za = max(cz - 1, 0), zb = min(cz + 1, rz - 1);
        for (int x = max(cx - 1, 0); x <= min(cx + 1, rx - 1); ++x)
          for (int y = max(cy - 1, 0); y <= min(cy + 1, ry - 1); ++y) {
            const int c0 = (x * ry + y) * rz + za, c1 = (x * ry + y) * rz + zb;
            scan(offn[c0], (c1 + 1 < total) ? (int64_t)offn[c1 + 1] : len2);
          }
        const float g = cell * 0.999f;
        if (g >= r || (wd < FLT_MAX && wd <= g * g)) { done = true; unfinished = false; }
        rho_first = 2;
      }
      for (int rho = rho_first; rho <= rho_stop && !done; ++rho) {
        const int x0 = max(cx - rho, 0), x1 = min(cx + rho, rx - 1);
        const int y0 = max(cy - rho, 0), y1 = min(cy + rho, ry - 1);
        for (int x = x0; x <= x1; ++x) {
          const bool ex = (x == cx - rho) || (x == cx + rho);
          for (int y = y0; y <= y1; ++y) {
            const bool edge = ex || (y == cy - rho) || (y == cy + rho);
            const int zlo = cz - rho, zhi = cz + rho;
            // edge columns take the whole z run, interior ones the two caps
            const int nseg = edge ? 1 : (rho == 0 ? 1 : 2);
            for (int sgm = 0; sgm < nseg; ++sgm) {
              int za, zb;
              if (edge) { za = zlo; zb = zhi; }
              else if (sgm == 0) { za = zlo; zb = zlo; }
              else { za = zhi; zb = zhi; }
              za = max(za, 0); zb = min(zb, rz - 1);
              if (za > zb) continue;
              const int c0 = (x * ry + y) * rz + za;
              const int c1 = (x * ry + y) * rz + zb;
              const int64_t i0 = offn[c0];
              const int64_t i1 = (c1 + 1 < total) ? (int64_t)offn[c1 + 1] : len2;
              scan(i0, i1);
            }
          }
        }
        if (rho >= 1) {
          float g = (float)rho * cell * 0.999f;
          if (g >= r) { unfinished = false; break; }
          if (wd < FLT_MAX && wd <= g * g) { unfinished = false; break; }
        }
      }
    }
    if (unfinished) {
      // row left for the tail kernel (it recomputes the query from scratch)
      const int slot = atomicAdd(tail_count + n, 1);
      tail_list[(int64_t)n * p1_stride + slot] = (int32_t)t;
      continue;
    }
    // write the row
    float* drow = dists_out + ((int64_t)n * p1_stride + row) * K;
    int64_t* irow = idxs_out + ((int64_t)n * p1_stride + row) * K;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      if (j < K) {
        bool ok = best.d[j] < FLT_MAX;
        drow[j] = ok ? best.d[j] : -1.0f;
        irow[j] = ok ? (int64_t)best.id[j] : (int64_t)-1;
        if (nn_out) {
          float* o = nn_out + (((int64_t)n * p1_stride + row) * K + j) * 3;
          if (ok) {
            const float* s = points2 + ((int64_t)n * p2_stride + best.id[j]) * 3;
            o[0] = s[0]; o[1] = s[1]; o[2] = s[2];
          } else {
            o[0] = 0.f; o[1] = 0.f; o[2] = 0.f;
          }
        }
      }
    }
  }
}


// One WAVE per unfinished query.  The wave walks the same Chebyshev shells as k_query, but the
// lanes split each shell's cell columns (an edge column is one contiguous z-run, an interior
// column its two cap cells) and keep private top-K lists.  After every shell the global K-th
// distance is obtained by K rounds of a wave-wide arg-min over the lanes' list heads, and the
// same stopping rule as k_query is applied; the final K results come out of the same merge --
// identical (d2, idx) order, so the output does not depend on which kernel served a query.
template <int KMAX>
__device__ __forceinline__ void wave_merge(const TopK<KMAX>& best, int K, float& kth, float* drow,
                                           int64_t* irow, float* nnrow, const float* pts2, int lane) {
  // kth = K-th merged distance (FLT_MAX if fewer than K); when drow != null lane 0 also writes
  // the merged row (dists, idxs, optional nn)
  int head = 0;
  kth = FLT_MAX;
  for (int k = 0; k < K; ++k) {
    float hd = FLT_MAX;
    int hi = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) if (j == head) { hd = best.d[j]; hi = best.id[j]; }
    float md = hd;
    int mi = hi;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      float od = __shfl_xor(md, o);
      int oi = __shfl_xor(mi, o);
      if (pair_lt(od, oi, md, mi)) { md = od; mi = oi; }
    }
    if (hd == md && hi == mi && md < FLT_MAX) ++head;   // the winner pops its head
    if (k == K - 1) kth = md;
    if (drow && lane == 0) {
      const bool ok = md < FLT_MAX;
      drow[k] = ok ? md : -1.0f;
      irow[k] = ok ? (int64_t)mi : (int64_t)-1;
      if (nnrow) {
        if (ok) {
          const float* sp = pts2 + (int64_t)mi * 3;
          nnrow[k * 3] = sp[0]; nnrow[k * 3 + 1] = sp[1]; nnrow[k * 3 + 2] = sp[2];
        } else {
          nnrow[k * 3] = 0.f; nnrow[k * 3 + 1] = 0.f; nnrow[k * 3 + 2] = 0.f;
        }
      }
    }
  }
}

template <int KMAX>
__global__ __launch_bounds__(64) void k_query_tail(
    const float* __restrict__ points1, const int64_t* __restrict__ lengths1,
    const float* __restrict__ points2, const float* __restrict__ sorted2,
    const int32_t* __restrict__ sorted_idx2, const int64_t* __restrict__ lengths2,
    const int32_t* __restrict__ off, const float* __restrict__ params,
    const float* __restrict__ radius, int K, float* __restrict__ dists_out,
    int64_t* __restrict__ idxs_out, float* __restrict__ nn_out, int64_t p1_stride,
    int64_t p2_stride, int64_t g_stride, const int32_t* __restrict__ tail_list,
    const int32_t* __restrict__ tail_count, const float4* __restrict__ xyzi) {
  const int n = blockIdx.y;
  const int lane = threadIdx.x;
  const float4* s4 = xyzi + (int64_t)blockIdx.y * p2_stride;
  const bool self = (points1 == nullptr);
  const int64_t len2 = lengths2 ? lengths2[n] : p2_stride;
  const float* gp = params + n * ISO_GRID3_PARAMS;
  const float mnx = gp[0], mny = gp[1], mnz = gp[2], delta = gp[3];
  const int rx = (int)gp[4], ry = (int)gp[5], rz = (int)gp[6];
  const int total = (int)gp[7];
  const float r = radius[n];
  const float r2 = r * r;
  const float cell = 1.0f / delta;
  const float* s2 = sorted2 + (int64_t)n * p2_stride * 3;
  const int32_t* sidx = sorted_idx2 + (int64_t)n * p2_stride;
  const int32_t* offn = off + (int64_t)n * g_stride;
  const int count = tail_count[n];
  for (int w = blockIdx.x; w < count; w += gridDim.x) {
    const int64_t t = tail_list[(int64_t)n * p1_stride + w];
    float qx, qy, qz;
    int64_t row;
    if (self) {
      qx = s2[t * 3]; qy = s2[t * 3 + 1]; qz = s2[t * 3 + 2];
      row = sidx[t];
    } else {
      const float* q = points1 + ((int64_t)n * p1_stride + t) * 3;
      qx = q[0]; qy = q[1]; qz = q[2];
      row = t;
    }
    const float lim = 1.0e6f;
    const int cx = (int)fminf(fmaxf(floorf((qx - mnx) * delta), -lim), lim);
    const int cy = (int)fminf(fmaxf(floorf((qy - mny) * delta), -lim), lim);
    const int cz = (int)fminf(fmaxf(floorf((qz - mnz) * delta), -lim), lim);
    const float rho_f = ceilf(r * delta * 1.0011f);
    const int gapx = cx < 0 ? -cx : (cx >= rx ? cx - rx + 1 : 0);
    const int gapy = cy < 0 ? -cy : (cy >= ry ? cy - ry + 1 : 0);
    const int gapz = cz < 0 ? -cz : (cz >= rz ? cz - rz + 1 : 0);
    const int rho0 = max(gapx, max(gapy, gapz));
    const int span = max(rx, max(ry, rz)) + rho0;
    const int rho_max = (rho_f < (float)span) ? (int)rho_f : span;
    TopK<KMAX> best;
    best.init();
    float wd = FLT_MAX;
    int wi = 0x7fffffff;
    int found = 0;   // candidates within r seen by this lane (capped: only >= K matters)
    for (int rho = rho0; rho <= rho_max; ++rho) {
      const int x0 = max(cx - rho, 0), x1 = min(cx + rho, rx - 1);
      const int y0 = max(cy - rho, 0), y1 = min(cy + rho, ry - 1);
      if (x0 <= x1 && y0 <= y1) {
        const int ny = y1 - y0 + 1;
        const int ncols = (x1 - x0 + 1) * ny;
        for (int col = lane; col < ncols; col += 64) {
          const int x = x0 + col / ny, y = y0 + col % ny;
          const bool edge = (x == cx - rho) || (x == cx + rho) || (y == cy - rho) || (y == cy + rho);
          const int nseg = edge ? 1 : (rho == 0 ? 1 : 2);
          for (int sgm = 0; sgm < nseg; ++sgm) {
            int za, zb;
            if (edge) { za = cz - rho; zb = cz + rho; }
            else if (sgm == 0) { za = cz - rho; zb = cz - rho; }
            else { za = cz + rho; zb = cz + rho; }
            za = max(za, 0); zb = min(zb, rz - 1);
            if (za > zb) continue;
            const int c0 = (x * ry + y) * rz + za, c1 = (x * ry + y) * rz + zb;
            const int64_t i0 = offn[c0];
            const int64_t i1 = (c1 + 1 < total) ? (int64_t)offn[c1 + 1] : len2;
            for (int64_t i = i0; i < i1; ++i) {
              const float4 ca = s4[i];
              float dx = qx - ca.x, dy = qy - ca.y, dz = qz - ca.z;
              float d2 = (dx * dx + dy * dy) + dz * dz;
              if (d2 < r2) {
                if (found < KMAX) ++found;
                if (d2 <= wd) {
                  int oi = __float_as_int(ca.w);
                  if (pair_lt(d2, oi, wd, wi)) {
                    best.push(d2, oi, K);
                    wd = best.worst(K);
                    wi = best.worst_id(K);
                  }
                }
              }
            }
          }
        }
      }
      if (rho >= 1) {
        const float g = (float)rho * cell * 0.999f;
        if (g >= r) break;
        int tot = found;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
        if (tot >= K) {                       // wave-uniform
          float kth;
          wave_merge<KMAX>(best, K, kth, nullptr, nullptr, nullptr, nullptr, lane);
          if (kth < FLT_MAX && kth <= g * g) break;
        }
      }
    }
    float kth;
    wave_merge<KMAX>(best, K, kth, dists_out + ((int64_t)n * p1_stride + row) * K,
                     idxs_out + ((int64_t)n * p1_stride + row) * K,
                     nn_out ? nn_out + ((int64_t)n * p1_stride + row) * K * 3 : nullptr,
                     points2 ? points2 + (int64_t)n * p2_stride * 3 : nullptr, lane);
  }
}

// candidate records for the query kernels: (x, y, z, original index as bits) -- one 16-B load per
// candidate instead of three strided dword loads plus the index load
__global__ void k_pack_xyzi(const float* __restrict__ sorted, const int32_t* __restrict__ sorted_idx,
                            const int64_t* __restrict__ lengths, int64_t p_stride, float4* __restrict__ out) {
  const int n = blockIdx.y;
  const int64_t len = lengths ? lengths[n] : p_stride;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float* q = sorted + ((int64_t)n * p_stride + i) * 3;
    out[(int64_t)n * p_stride + i] = make_float4(q[0], q[1], q[2], __int_as_float(sorted_idx[(int64_t)n * p_stride + i]));
  }
}

__global__ void k_zero_i32(int32_t* p, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0;
}

// nn gather for the query result: nn[n,i,k,:] = points2[n, idx[n,i,k], :]
__global__ void k_gather(const float* __restrict__ x,
                         const int64_t* __restrict__ idx, float* __restrict__ out,
                         int64_t p1, int64_t p2, int K, int U, int64_t total) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    int64_t n = e / (p1 * K);
    int64_t j = idx[e];
    float* o = out + e * U;
    if (j < 0) {
      for (int u = 0; u < U; ++u) o[u] = 0.f;
    } else {
      const float* s = x + (n * p2 + j) * U;
      for (int u = 0; u < U; ++u) o[u] = s[u];
    }
  }
}

}  // namespace

// ---------------------------------------------------------------------------
// workgroups of k_bbox: four loads per thread and round, a stride of a multiple of 3 floats
static int bbox_grid(int64_t p_stride) {
  int gx = iso_div_up(p_stride * 3, 256 * 4);
  if (gx > 255) gx = 255;              // six same-address atomics per workgroup: more workgroups cost more than they read
  if (gx >= 3) gx -= gx % 3;
  return gx < 1 ? 1 : gx;
}

extern "C" int iso_frnn_make_grid_density(const float* points, const int64_t* lengths,
                                          const float* radius, int n_clouds, int64_t p_stride,
                                          int max_res, float points_per_cell, float* grid_params,
                                          void* stream) {
  ISO_REQUIRE(n_clouds >= 0 && p_stride >= 0, ISO_ERR_INVALID, "iso_frnn_make_grid: bad sizes");
  ISO_REQUIRE(points_per_cell >= 1.0f && points_per_cell <= 4096.0f, ISO_ERR_INVALID,
              "iso_frnn_make_grid: points_per_cell must be in [1,4096], got %g", (double)points_per_cell);
  ISO_REQUIRE(max_res >= 1 && max_res <= ISO_GRID_MAX_RES, ISO_ERR_INVALID,
              "iso_frnn_make_grid: max_res must be in [1,%d], got %d", ISO_GRID_MAX_RES, max_res);
  if (n_clouds == 0) return ISO_OK;
  ISO_REQUIRE(radius && grid_params && (points || p_stride == 0), ISO_ERR_INVALID,
              "iso_frnn_make_grid: null pointer");
  hipStream_t s = (hipStream_t)stream;
  unsigned* keys = reinterpret_cast<unsigned*>(grid_params);
  hipLaunchKernelGGL(k_bbox_init, dim3(iso_div_up(n_clouds * 8, 256)), dim3(256), 0, s, keys, n_clouds);
  if (p_stride > 0) {
    int gx = bbox_grid(p_stride);
    hipLaunchKernelGGL(k_bbox<256>, dim3(gx, n_clouds), dim3(256), 0, s, points, lengths, p_stride, keys);
  }
  hipLaunchKernelGGL(k_grid_finalize, dim3(iso_div_up(n_clouds, 64)), dim3(64), 0, s,
                     grid_params, lengths, p_stride, radius, n_clouds, max_res, 1.0f / points_per_cell);
  ISO_CHECK_LAUNCH("iso_frnn_make_grid");
  return ISO_OK;
}

extern "C" int iso_frnn_make_grid(const float* points, const int64_t* lengths,
                                  const float* radius, int n_clouds,
                                  int64_t p_stride, int max_res, float* grid_params,
                                  void* stream) {
  return iso_frnn_make_grid_density(points, lengths, radius, n_clouds, p_stride, max_res, 8.0f,
                                    grid_params, stream);
}

extern "C" int iso_points_bbox(const float* points, const int64_t* lengths, int n_clouds,
                               int64_t p_stride, float* minmax, void* stream) {
  ISO_REQUIRE(n_clouds >= 0 && p_stride >= 0, ISO_ERR_INVALID, "iso_points_bbox: bad sizes");
  if (n_clouds == 0) return ISO_OK;
  ISO_REQUIRE(minmax && (points || p_stride == 0), ISO_ERR_INVALID, "iso_points_bbox: null pointer");
  hipStream_t s = (hipStream_t)stream;
  unsigned* keys = reinterpret_cast<unsigned*>(minmax);
  hipLaunchKernelGGL(k_bbox_init, dim3(iso_div_up(n_clouds * 8, 256)), dim3(256), 0, s, keys, n_clouds);
  if (p_stride > 0) {
    int gx = bbox_grid(p_stride);
    hipLaunchKernelGGL(k_bbox<256>, dim3(gx, n_clouds), dim3(256), 0, s, points, lengths, p_stride, keys);
  }
  hipLaunchKernelGGL(k_bbox_decode, dim3(iso_div_up(n_clouds, 64)), dim3(64), 0, s, keys, lengths,
                     p_stride, n_clouds);
  ISO_CHECK_LAUNCH("iso_points_bbox");
  return ISO_OK;
}

extern "C" int iso_frnn_insert_points(const float* points, const int64_t* lengths,
                                      const float* grid_params, int32_t* cnt,
                                      int32_t* cell, int32_t* idx_in_cell,
                                      int n_clouds, int64_t p_stride,
                                      int64_t g_stride, int dim, void* stream) {
  ISO_REQUIRE(dim == 2 || dim == 3, ISO_ERR_INVALID, "iso_frnn_insert_points: dim must be 2 or 3");
  ISO_REQUIRE(n_clouds >= 0 && p_stride >= 0 && g_stride >= 0, ISO_ERR_INVALID, "iso_frnn_insert_points: bad sizes");
  if (n_clouds == 0 || p_stride == 0) return ISO_OK;
  ISO_REQUIRE(points && grid_params && cnt && cell && idx_in_cell, ISO_ERR_INVALID,
              "iso_frnn_insert_points: null pointer");
  int gx = iso_div_up(p_stride, 256);
  if (gx > 4096) gx = 4096;
  hipStream_t s = (hipStream_t)stream;
  if (dim == 3)
    hipLaunchKernelGGL(k_insert<3>, dim3(gx, n_clouds), dim3(256), 0, s, points, lengths, grid_params, cnt, cell, idx_in_cell, p_stride, g_stride);
  else
    hipLaunchKernelGGL(k_insert<2>, dim3(gx, n_clouds), dim3(256), 0, s, points, lengths, grid_params, cnt, cell, idx_in_cell, p_stride, g_stride);
  ISO_CHECK_LAUNCH("iso_frnn_insert_points");
  return ISO_OK;
}

static int scan_impl(const int32_t* in, int32_t* out, const float* params, int dim,
                     int64_t n_host, int batch, int64_t row_stride, void* ws,
                     int64_t ws_bytes, hipStream_t s, const char* who) {
  int64_t max_len = params ? row_stride : n_host;
  int chunks = iso_div_up(max_len, SCAN_CHUNK);
  if (chunks < 1) chunks = 1;
  ISO_REQUIRE(ws && ws_bytes >= (int64_t)batch * chunks * 4, ISO_ERR_WORKSPACE,
              "%s: workspace too small (%lld < %lld)", who, (long long)ws_bytes,
              (long long)batch * chunks * 4);
  int32_t* sums = (int32_t*)ws;
  hipLaunchKernelGGL(k_scan_sums, dim3(chunks, batch), dim3(SCAN_BLOCK), 0, s, in, sums, params, dim, n_host, row_stride, chunks);
  hipLaunchKernelGGL(k_scan_mid, dim3(batch), dim3(SCAN_BLOCK), 0, s, sums, chunks);
  hipLaunchKernelGGL(k_scan_final, dim3(chunks, batch), dim3(SCAN_BLOCK), 0, s, in, out, sums, params, dim, n_host, row_stride, chunks);
  return ISO_OK;
}

extern "C" int64_t iso_prefix_sum_workspace_bytes(int64_t n, int batch) {
  int64_t chunks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
  if (chunks < 1) chunks = 1;
  return chunks * 4 * (batch > 0 ? batch : 1);
}

extern "C" int iso_prefix_sum(const int32_t* in, int32_t* out, int64_t n, int batch,
                              int64_t row_stride, void* workspace,
                              int64_t workspace_bytes, void* stream) {
  ISO_REQUIRE(n >= 0 && batch >= 0 && row_stride >= n, ISO_ERR_INVALID, "iso_prefix_sum: bad sizes");
  if (n == 0 || batch == 0) return ISO_OK;
  ISO_REQUIRE(in && out, ISO_ERR_INVALID, "iso_prefix_sum: null pointer");
  int rc = scan_impl(in, out, nullptr, 3, n, batch, row_stride, workspace, workspace_bytes, (hipStream_t)stream, "iso_prefix_sum");
  if (rc != ISO_OK) return rc;
  ISO_CHECK_LAUNCH("iso_prefix_sum");
  return ISO_OK;
}

extern "C" int iso_frnn_scan_cells(const int32_t* cnt, int32_t* off,
                                   const float* grid_params, int n_clouds,
                                   int64_t g_stride, int dim, void* workspace,
                                   int64_t workspace_bytes, void* stream) {
  ISO_REQUIRE(dim == 2 || dim == 3, ISO_ERR_INVALID, "iso_frnn_scan_cells: dim must be 2 or 3");
  ISO_REQUIRE(n_clouds >= 0 && g_stride >= 0, ISO_ERR_INVALID, "iso_frnn_scan_cells: bad sizes");
  if (n_clouds == 0 || g_stride == 0) return ISO_OK;
  ISO_REQUIRE(cnt && off && grid_params, ISO_ERR_INVALID, "iso_frnn_scan_cells: null pointer");
  int rc = scan_impl(cnt, off, grid_params, dim, 0, n_clouds, g_stride, workspace, workspace_bytes, (hipStream_t)stream, "iso_frnn_scan_cells");
  if (rc != ISO_OK) return rc;
  ISO_CHECK_LAUNCH("iso_frnn_scan_cells");
  return ISO_OK;
}

extern "C" int iso_frnn_counting_sort(const float* points, const int64_t* lengths,
                                      const int32_t* cell, const int32_t* idx_in_cell,
                                      const int32_t* off, float* sorted_points,
                                      int32_t* sorted_idx, int n_clouds,
                                      int64_t p_stride, int64_t g_stride, int dim,
                                      void* stream) {
  ISO_REQUIRE(dim == 2 || dim == 3, ISO_ERR_INVALID, "iso_frnn_counting_sort: dim must be 2 or 3");
  ISO_REQUIRE(n_clouds >= 0 && p_stride >= 0, ISO_ERR_INVALID, "iso_frnn_counting_sort: bad sizes");
  if (n_clouds == 0 || p_stride == 0) return ISO_OK;
  ISO_REQUIRE(points && cell && idx_in_cell && off && sorted_points && sorted_idx,
              ISO_ERR_INVALID, "iso_frnn_counting_sort: null pointer");
  int gx = iso_div_up(p_stride, 256);
  if (gx > 4096) gx = 4096;
  hipStream_t s = (hipStream_t)stream;
  if (dim == 3)
    hipLaunchKernelGGL(k_counting_sort<3>, dim3(gx, n_clouds), dim3(256), 0, s, points, lengths, cell, idx_in_cell, off, sorted_points, sorted_idx, p_stride, g_stride);
  else
    hipLaunchKernelGGL(k_counting_sort<2>, dim3(gx, n_clouds), dim3(256), 0, s, points, lengths, cell, idx_in_cell, off, sorted_points, sorted_idx, p_stride, g_stride);
  ISO_CHECK_LAUNCH("iso_frnn_counting_sort");
  return ISO_OK;
}

extern "C" int iso_frnn_gather(const float* x, const int64_t* idx, float* out,
                               int n_clouds, int64_t p1, int64_t p2, int K, int U,
                               void* stream) {
  ISO_REQUIRE(n_clouds >= 0 && p1 >= 0 && p2 >= 0 && K >= 0 && U >= 0, ISO_ERR_INVALID, "iso_frnn_gather: bad sizes");
  int64_t total = (int64_t)n_clouds * p1 * K;
  if (total == 0 || U == 0) return ISO_OK;
  ISO_REQUIRE(x && idx && out, ISO_ERR_INVALID, "iso_frnn_gather: null pointer");
  hipLaunchKernelGGL(k_gather, dim3(iso_stream_grid(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, x, idx, out, p1, p2, K, U, total);
  ISO_CHECK_LAUNCH("iso_frnn_gather");
  return ISO_OK;
}

// workspace: [tail_count: n_clouds ints, padded to 64][tail_list: n_clouds*p1_stride ints]
//            [pad to 16 B][candidate records: n_clouds*p2_stride float4]
static int64_t query_ws_tail_bytes(int n_clouds, int64_t p1_stride) {
  int64_t b = 4 * (64 * (int64_t)((n_clouds + 63) / 64) + (int64_t)n_clouds * p1_stride);
  return (b + 15) / 16 * 16;
}

extern "C" int64_t iso_frnn_query_workspace_bytes(int n_clouds, int64_t p1_stride, int64_t p2_stride) {
  if (n_clouds < 0) n_clouds = 0;
  if (p1_stride < 0) p1_stride = 0;
  if (p2_stride < 0) p2_stride = 0;
  return query_ws_tail_bytes(n_clouds, p1_stride) + 16 * (int64_t)n_clouds * p2_stride;
}

extern "C" int iso_frnn_query(const float* points1, const int64_t* lengths1,
                              const float* points2, const float* sorted2,
                              const int32_t* sorted_idx2,
                              const int64_t* lengths2, const int32_t* off,
                              const float* grid_params, const float* radius, int K,
                              float* dists_out, int64_t* idxs_out, float* nn_out,
                              int n_clouds, int64_t p1_stride, int64_t p2_stride,
                              int64_t g_stride, void* workspace, int64_t workspace_bytes,
                              void* stream) {
  ISO_REQUIRE(K >= 1 && K <= 32, ISO_ERR_UNSUPPORTED, "iso_frnn_query: K must be in [1,32], got %d", K);
  ISO_REQUIRE(n_clouds >= 0 && p1_stride >= 0 && p2_stride >= 0, ISO_ERR_INVALID, "iso_frnn_query: bad sizes");
  if (n_clouds == 0 || p1_stride == 0) return ISO_OK;
  ISO_REQUIRE(sorted2 && sorted_idx2 && off && grid_params && radius && dists_out && idxs_out,
              ISO_ERR_INVALID, "iso_frnn_query: null pointer");
  ISO_REQUIRE(points1 || p1_stride == p2_stride, ISO_ERR_INVALID,
              "iso_frnn_query: self query needs p1_stride == p2_stride");
  ISO_REQUIRE(!nn_out || points2, ISO_ERR_INVALID, "iso_frnn_query: nn_out needs points2");
  ISO_REQUIRE(workspace && workspace_bytes >= iso_frnn_query_workspace_bytes(n_clouds, p1_stride, p2_stride),
              ISO_ERR_WORKSPACE, "iso_frnn_query: workspace too small");
  ISO_REQUIRE(((uintptr_t)workspace & 15) == 0, ISO_ERR_INVALID, "iso_frnn_query: workspace must be 16-B aligned");
  hipStream_t s = (hipStream_t)stream;
  int32_t* tail_count = (int32_t*)workspace;                 // [n_clouds] (padded to 64 ints)
  int32_t* tail_list = tail_count + 64 * ((n_clouds + 63) / 64);
  float4* xyzi = reinterpret_cast<float4*>((char*)workspace + query_ws_tail_bytes(n_clouds, p1_stride));
  hipLaunchKernelGGL(k_zero_i32, dim3(iso_div_up(n_clouds, 64)), dim3(64), 0, s, tail_count, n_clouds);
  if (p2_stride > 0) {
    int gp = iso_div_up(p2_stride, 256);
    if (gp > 4096) gp = 4096;
    hipLaunchKernelGGL(k_pack_xyzi, dim3(gp, n_clouds), dim3(256), 0, s, sorted2, sorted_idx2, lengths2, p2_stride, xyzi);
  }
  int gx = iso_div_up(p1_stride, 256);
  if (gx > 65535) gx = 65535;
#define ISO_LAUNCH_Q(KM)                                                          \
  hipLaunchKernelGGL(k_query<KM>, dim3(gx, n_clouds), dim3(256), 0, s, points1,   \
                     lengths1, points2, sorted2, sorted_idx2, lengths2, off, grid_params,  \
                     radius, K, dists_out, idxs_out, nn_out, p1_stride, p2_stride, \
                     g_stride, tail_list, tail_count, xyzi);                      \
  hipLaunchKernelGGL(k_query_tail<KM>, dim3(tail_blocks, n_clouds), dim3(64), 0, s, points1,      \
                     lengths1, points2, sorted2, sorted_idx2, lengths2, off, grid_params,         \
                     radius, K, dists_out, idxs_out, nn_out, p1_stride, p2_stride, g_stride,      \
                     tail_list, tail_count, xyzi)
  int tail_blocks = (int)(p1_stride < 2048 ? p1_stride : 2048);
  if (tail_blocks < 1) tail_blocks = 1;
  // the K-best list is register-resident and an insertion walks all KMAX slots: keep KMAX tight
  // for the K values the path uses (K+1 = 9 neighbour trees, K = 7 splat bandwidth)
  if (K <= 4) { ISO_LAUNCH_Q(4); }
  else if (K <= 8) { ISO_LAUNCH_Q(8); }
  else if (K <= 10) { ISO_LAUNCH_Q(10); }
  else if (K <= 16) { ISO_LAUNCH_Q(16); }
  else { ISO_LAUNCH_Q(32); }
#undef ISO_LAUNCH_Q
  ISO_CHECK_LAUNCH("iso_frnn_query");
  return ISO_OK;
}
