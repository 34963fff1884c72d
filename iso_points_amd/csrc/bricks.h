// Two-level neighbour grid for surface clouds on gfx950 ("bricks").
//
// Level 1 (global memory): coarse BRICKS of 4x4x4 fine cells, x-major brick id
// ((bx*nby + by)*nbz + bz), a dense table of brick offsets (a few hundred thousand entries for a
// 1 M-point surface -- the fine 256^3 table of the classic FRNN grid is 17 M entries, 99 % empty)
// and the points as 2 x 16-byte records in brick order: rec0 = (x, y, z, global id), rec1 =
// (nx, ny, nz, payload).  A contiguous range of brick ids is an x-slab of space: this is the unit
// the multi-GPU path shards by.
//
// Level 2 (LDS): a workgroup that owns a brick stages the brick plus a one-fine-cell halo (6x6x6
// fine cells) into LDS, counting-sorted by local fine cell.  Every query of the brick then walks
// the 3x3x3 fine cells around it entirely in LDS.  All points within g = 0.999 f (f = fine cell
// size) of a query are in that block, so a K-nearest result whose K-th distance is <= g (or any
// result when g >= r) is exact; the rare query that is not certified goes to a tail kernel that
// walks rings of bricks.
#pragma once
#include <float.h>
#include "iso_common.h"

struct BrickHdr {                      // device resident, 32 dwords
  float mn[3]; float inv_f;            //  0..3   origin of the fine grid, 1 / fine cell size
  float nbx_f, nby_f, nbz_f, total_f;  //  4..7   ISO_GRID3_PARAMS layout for iso_frnn_scan_cells ([7] = 8 * bricks + 1)
  float f, r, r2, g2;                  //  8..11  fine cell, search radius, r^2, (0.999 f)^2
  float inv_sigma, diag, spacing, pad0;  // 12..15  P / diag (levelset_sampling.py:256), |bbox diagonal|, sqrt(diag / P)
  int nb[3]; int n_bricks;             // 16..19
  int nf[3]; int n;                    // 20..23  fine cells per axis (4 nb), points in the grid (own + imported)
  int n_own; int id_base; int g_covers_r; int n_total;  // 24..27
  float x_lo, x_hi;                    // 28..29  N ranks: x-range inside which this rank holds EVERY point of the cloud
  int pad1[2];                         // 30..31
};

// counters per brick (a power of two <= 8; a brick's records are the union of its counters' ranges).  Eight spread the
// same-address atomics of the counting pass when it issued one per point; since it aggregates per workgroup in LDS
// one is faster in every case measured (scan / zero / list walk an eighth of the table: 1.876 -> 1.833 ms per
// cfg-3a cycle; unsorted 1 M-point cloud: build 0.137 -> 0.123 ms).
constexpr int BK_CPB = 1;
constexpr int BK_CAP = 1024;           // staged candidates per brick (10-bit slot field of the selection keys)
constexpr int BK_THREADS = 256;
constexpr int BK_NB_MAX = 160;         // bricks per axis, hard cap

static inline int bricks_nb_cap(int64_t n_max) {
  // bricks per axis the table is sized for: ~sqrt(n)/8 (a 1 M-point unit sphere wants ~60)
  int64_t c = 8;
  while (c * c * 64 < n_max && c < BK_NB_MAX) ++c;
  if (c < 8) c = 8;
  return (int)c;
}

struct BrickWs {      // carved out of one caller-owned workspace
  BrickHdr* hdr;
  int32_t* counters;  // [0] list_count  [1] tail_count  [2] overflow bricks  [3] tail2_count  [4] export overflow
                      // [5] import overflow  [6] tail queries whose search left the imported halo  (16 ints)
  int32_t* cnt;       // [G]  G = 8 * cap^3 + 1: eight counters per brick
  int32_t* off;       // [G]
  int32_t* slot;      // [n_max]
  float4* rec0;       // [n_max]
  float4* rec1;       // [n_max]
  int32_t* list;      // [min(G, n_max)]
  int32_t* tail;      // [8 n_max]
  void* scan_ws;
  int64_t scan_ws_bytes;
  int nb_cap;
  int64_t G;
  int64_t bytes;
};

BrickWs bricks_carve(void* ws, int64_t n_max);

__device__ __forceinline__ int bk_fine(float p, float mn, float inv_f, int nf) {
  const int c = (int)floorf((p - mn) * inv_f);
  return c < 0 ? 0 : (c >= nf ? nf - 1 : c);
}

// index of the sub-cell (SUB per fine cell and axis) of p: floor(SUB * t) with t as in bk_fine, so that
// bk_sub<SUB>(p) / SUB == bk_fine(p) (doubling t is exact in f32; same clamps)
template <int SUB>
__device__ __forceinline__ int bk_sub(float p, float mn, float inv_f, int nf) {
  const int c = (int)floorf(((p - mn) * inv_f) * (float)SUB);
  return c < 0 ? 0 : (c >= nf * SUB ? nf * SUB - 1 : c);
}

__device__ __forceinline__ unsigned bk_med3u(unsigned a, unsigned b, unsigned c) {
  unsigned r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// x and y as a packed pair (v_pk_add_f32 / v_pk_mul_f32: two IEEE f32 operations per instruction, same results):
// six instructions instead of eight in the innermost loop of kernels that are bound by their instruction count
typedef float bk_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float bk_d2(float ax, float ay, float az, float bx, float by, float bz) {
  const bk_f2 a = {ax, ay}, b = {bx, by};
  const bk_f2 d = a - b;
  const bk_f2 s = d * d;
  const float dz = az - bz;
  return (s.x + s.y) + dz * dz;                // contraction is off: the oracle's expression
}
