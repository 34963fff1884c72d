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
  int own_bx_lo, own_bx_hi;            // 30..31  N ranks: brick columns (x) that can hold points of this rank -- the others hold imported records only
};

// Counters per brick: one per SUB-BRICK of 2x2x2 fine cells (sub = (sx * 2 + sy) * 2 + sz, sx = bit 1 of the fine x cell
// inside the brick, ...): the records of brick b are [off[8 b], off[8 (b + 1)]), those of its sub-brick s
// [off[8 b + s], off[8 b + s + 1]).  A workgroup that stages brick b + one fine cell of halo then loads, of a neighbouring
// brick, only the sub-bricks that touch b -- half of a face neighbour, a quarter of an edge neighbour, an eighth of a
// corner neighbour: 4 bricks' worth of records on a surface instead of 9 (the loads and cell counts of the staging were
// 35 % of the bandwidth kernel and 18 % of the resample kernel).
constexpr int BK_CPB = 8;
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
  unsigned* scan1;    // [kScan1MaxWords] chunk totals of the one-launch scan (k_brick_offsets1): zero on entry, left zero;
                      // at a FIXED offset (right behind the counter block) -- a caller may carve with n < the n_max of
                      // iso_bricks_workspace_init (iso_bricks_build_whole does) and must still find the zeroed words
  void* scan_ws;
  int64_t scan_ws_bytes;
  int nb_cap;
  int64_t G;
  int64_t bytes;
};

BrickWs bricks_carve(void* ws, int64_t n_max);

// ---- counter block of the workspace (kCounterInts ints) ---------------------------------------------
// [0..15]   the grid's counters (BrickWs::counters; reset by every header write)
// [16..31]  their sums over all earlier grids on this workspace ("sticky": a header write adds the counters it is about
//           to reset, so that overflows / uncertified queries of a whole cycle -- several grids -- can be read once,
//           afterwards, without an accumulation pass per grid)
// [38]      the init mark
// [40..56]  arrival words (bk_last_block; k_brick_offsets1)
// [64.. ]   the PENDING BOX: bounding-box accumulators, kBoxCopies copies of 8 words, ONE 128-BYTE LINE EACH (min xyz as
//           max of the complemented key, max xyz as key: all-zero = empty); a workgroup adds to copy blockIdx %
//           kBoxCopies.  Atomics on one line retire one after the other (~10 ns each) wherever in the line they land:
//           six per workgroup on one copy were the whole cost of a box pass at a few thousand workgroups, and sixteen
//           copies packed into four lines still cost 7 us of a 12 us projection, 256 lines 5 us (23 k atomics).  So a
//           workgroup first LOOKS at its copy (a plain load: possibly stale, i.e. smaller, which only costs an atomic
//           that was not needed) and adds only what would grow it -- a few hundred atomics per pass instead of 23 k.
//           Filled by k_brick_bbox or by a projection launch (follow.h), turned into the header by the count pass of
//           the build that follows (every workgroup decodes it for itself: no launch, no grid-wide wait in between),
//           cleared by that build's offsets pass.
constexpr int kStickyAt = 16, kMagicAt = 38, kArriveAt = 40, kBoxAt = 64, kBoxCopies = 16, kBoxStride = 32,
              kCounterInts = kBoxAt + kBoxCopies * kBoxStride;
constexpr int kBrickMagic = 0x1b71c5;

// what the header is made from besides the box
struct BrickParams {
  int64_t n_total, n_own, id_base;
  float radius; int knn_k; float cell_scale; int nb_cap;
};

#ifdef __HIPCC__
__device__ __forceinline__ unsigned bk_f2key(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float bk_key2f(unsigned k) {
  const unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

// Per-thread bounding box, joined over the workgroup and added to the pending box (all threads of the workgroup call
// commit(); s_box: one row of 6 floats per wave).
struct BkBox {
  float lo[3], hi[3];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int a = 0; a < 3; ++a) { lo[a] = FLT_MAX; hi[a] = -FLT_MAX; }
  }
  __device__ __forceinline__ void add(float x, float y, float z) {
    lo[0] = fminf(lo[0], x); hi[0] = fmaxf(hi[0], x);
    lo[1] = fminf(lo[1], y); hi[1] = fmaxf(hi[1], y);
    lo[2] = fminf(lo[2], z); hi[2] = fmaxf(hi[2], z);
  }
  // what this thread's word of the workgroup's copy holds NOW (threads 0..5; call it early -- at kernel start -- and hand
  // the value to commit(): a stale, i.e. smaller, value only costs an atomic that was not needed, while a load at the end
  // of the workgroup's life is a memory round trip on its critical path: 6 us of a 12 us projection)
  static __device__ __forceinline__ unsigned peek(const int32_t* __restrict__ counters) {
    const unsigned* acc = reinterpret_cast<const unsigned*>(counters) + kBoxAt + kBoxStride * (blockIdx.x % kBoxCopies);
    return threadIdx.x < 6 ? acc[threadIdx.x] : 0u;
  }
  __device__ __forceinline__ void commit(int32_t* __restrict__ counters, float (*s_box)[6], unsigned seen = 0u) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        lo[a] = fminf(lo[a], __shfl_xor(lo[a], o));
        hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o));
      }
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
      for (int a = 0; a < 3; ++a) { s_box[w][a] = lo[a]; s_box[w][3 + a] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
      const int a = threadIdx.x;                                // 0..2: min of axis a, 3..5: max of axis a - 3
      float v = s_box[0][a];
      for (int k = 1; k < (int)(blockDim.x >> 6); ++k) v = a < 3 ? fminf(v, s_box[k][a]) : fmaxf(v, s_box[k][a]);
      unsigned* acc = reinterpret_cast<unsigned*>(counters) + kBoxAt + kBoxStride * (blockIdx.x % kBoxCopies);
      const bool any = a < 3 ? v < FLT_MAX : v > -FLT_MAX;      // (a workgroup without a point: nothing)
      const unsigned key = a < 3 ? ~bk_f2key(v) : bk_f2key(v);
      if (any && key > seen) atomicMax(&acc[a], key);
    }
  }
};

// The pending box, read by a LATER launch than the ones that filled it: all threads of a workgroup (>= 64) call this
// (lane t < kBoxCopies of wave 0 reads copy t; s_red: 8 words); the result is valid in every thread.  An empty box
// decodes to [0, 0].  The accumulators are left as they are (bk_box_clear).
__device__ __forceinline__ void bk_box_read(const int32_t* __restrict__ counters, unsigned* s_red, float* mn, float* mx) {
  const unsigned* acc = reinterpret_cast<const unsigned*>(counters) + kBoxAt;
  static_assert(kBoxCopies <= 64, "one copy per lane of the first wave");
  const int t = threadIdx.x;
  if (t < 64) {
    unsigned m[6] = {0u, 0u, 0u, 0u, 0u, 0u};
    if (t < kBoxCopies) {
      const uint4 lo = *reinterpret_cast<const uint4*>(acc + kBoxStride * t);
      const uint2 hi = *reinterpret_cast<const uint2*>(acc + kBoxStride * t + 4);
      m[0] = lo.x; m[1] = lo.y; m[2] = lo.z; m[3] = lo.w; m[4] = hi.x; m[5] = hi.y;
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m[a] = max(m[a], (unsigned)__shfl_xor((int)m[a], o));
    }
    if (t == 0) {
#pragma unroll
      for (int a = 0; a < 6; ++a) s_red[a] = m[a];
    }
  }
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const unsigned l = s_red[a], u = s_red[3 + a];
    mn[a] = l ? bk_key2f(~l) : 0.f;
    mx[a] = u ? bk_key2f(u) : 0.f;
  }
}
// lanes 0..kBoxCopies-1 of ONE workgroup, in a launch after every reader of the pending box
__device__ __forceinline__ void bk_box_clear(int32_t* __restrict__ counters) {
  unsigned* acc = reinterpret_cast<unsigned*>(counters) + kBoxAt;
  if (threadIdx.x < kBoxCopies) {
    *reinterpret_cast<uint4*>(acc + kBoxStride * threadIdx.x) = make_uint4(0u, 0u, 0u, 0u);
    *reinterpret_cast<uint2*>(acc + kBoxStride * threadIdx.x + 4) = make_uint2(0u, 0u);
  }
}

// "The last workgroup to arrive does the follow-up": every workgroup calls this after its device-scope atomics / its
// write-through (agent-scope atomic) stores; true in exactly one workgroup, on all its threads -- the one whose arrival
// completes the grid.  What the others published with agent-scope atomics or atomic stores is visible to its agent-scope
// atomic loads (no fences: nothing plain is handed over).  Two levels (a thousand arrivals on ONE word retire ~12-30 ns
// apart): workgroup b arrives at word 1 + b % 16, the last of each of those sixteen groups at word 0.  All seventeen
// words are back at zero when the launch ends.  (Costs the launch a serial tail of ~7 us -- arrival, then whatever the
// last workgroup reads through agent-scope loads at ~1 us a round trip: measured on the box pass, which therefore
// leaves its result PENDING for the next launch instead.)
__device__ __forceinline__ bool bk_last_block(int32_t* __restrict__ counters, int* s_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's atomics / stores have reached the coherence point
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned* arrive = reinterpret_cast<unsigned*>(counters) + kArriveAt;
    const unsigned total = gridDim.x * gridDim.y, b = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned g = b % 16u, gsize = (total - g + 15u) / 16u, groups = total < 16u ? total : 16u;
    int last = 0;
    if (__hip_atomic_fetch_add(&arrive[1 + g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1u) {
      __hip_atomic_store(&arrive[1 + g], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__hip_atomic_fetch_add(&arrive[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == groups - 1u) {
        __hip_atomic_store(&arrive[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = 1;
      }
    }
    *s_flag = last;
  }
  __syncthreads();
  return *s_flag != 0;
}

// header for the box [mn, mx] (pure: every field is set)
__device__ inline BrickHdr bricks_header(const float* mn_in, const float* mx_in, const BrickParams& q) {
  BrickHdr h;
  float mn[3], ext[3];
  for (int a = 0; a < 3; ++a) {
    mn[a] = mn_in[a]; ext[a] = mx_in[a] - mn_in[a]; if (!(ext[a] >= 0.f)) ext[a] = 0.f;
  }
  const float diag = sqrtf((ext[0] * ext[0] + ext[1] * ext[1]) + ext[2] * ext[2]);
  const float np = (float)(q.n_total > 0 ? q.n_total : 1);
  const float spacing = sqrtf(diag / np);
  const float r = q.radius > 0.f ? q.radius : spacing * (float)q.knn_k;      // levelset_sampling.py:129-131
  float f = q.cell_scale * spacing;
  // K-nearest grids (radius derived from knn_k): the reference's radius knn_k sqrt(diag / n) (levelset_sampling.py:129-131)
  // is not scale-free -- on a surface cloud it holds ~55 neighbours when diag = 3.5 (the unit sphere of configs[2]) and
  // ~97 when diag = 2.2 (configs[3]: 4 M points on a sphere of radius 0.6), and a cell of 0.8 r then stages 650 records
  // per brick neighbourhood on average: most bricks overflow the 1024 LDS slots and their queries take the tail kernel
  // (35 ms of a 4 M-point resample).  What a query needs is its K nearest, ~1.7 mean spacings away: the cell is capped at
  // 3.4 mean spacings of a surface of area ~1.05 diag^2 (the unit-sphere tuning: 0.8 r there).  Exactness does not
  // depend on the cell: a query is certified against it or goes to the tail.
  if (!(q.radius > 0.f) && q.knn_k > 0) {
    const float fd = 3.47f * diag / sqrtf(np);
    if (fd > 0.f && f > fd) f = fd;
  }
  if (r > 0.f && f > r * 1.002f) f = r * 1.002f;                        // g = 0.999 f >= r: nothing to gain beyond
  const float emax = fmaxf(ext[0], fmaxf(ext[1], ext[2]));
  const float fmin = emax / (4.0f * (float)(q.nb_cap - 1)) * 1.0001f;
  if (f < fmin) f = fmin;
  if (!(f > 1e-20f)) f = 1.0f;                                          // degenerate cloud: one brick
  const float inv_f = 1.0f / f;
  int nb[3];
  for (int a = 0; a < 3; ++a) {
    int nf = (int)floorf(ext[a] * inv_f) + 1;
    nb[a] = (nf + 3) / 4;
    if (nb[a] > q.nb_cap) nb[a] = q.nb_cap;
    if (nb[a] < 1) nb[a] = 1;
    h.nb[a] = nb[a];
    h.nf[a] = 4 * nb[a];
    h.mn[a] = mn[a];
  }
  const int nbr = nb[0] * nb[1] * nb[2];
  h.inv_f = inv_f;
  h.nbx_f = (float)nb[0]; h.nby_f = (float)nb[1]; h.nbz_f = (float)nb[2];
  h.total_f = (float)(BK_CPB * nbr + 1);
  h.f = f; h.r = r; h.r2 = r * r;
  const float g = 0.999f * f;
  h.g2 = g * g;
  h.inv_sigma = np / diag;                                               // levelset_sampling.py:256
  h.diag = diag; h.spacing = spacing; h.pad0 = 0.f;
  h.n_bricks = nbr;
  h.n = (int)q.n_own;                                                    // + imported, added by k_brick_count_recs
  h.n_own = (int)q.n_own; h.id_base = (int)q.id_base;
  h.g_covers_r = g >= r ? 1 : 0;
  h.n_total = (int)q.n_total;
  h.x_lo = -FLT_MAX; h.x_hi = FLT_MAX;
  h.own_bx_lo = 0; h.own_bx_hi = 0x7fffffff;
  return h;
}
// one thread, once per grid: store the header, move the counters of the previous grid to the sticky block
__device__ inline void bricks_store_header(const BrickHdr& hv, BrickHdr* __restrict__ h, int32_t* __restrict__ counters) {
  *h = hv;
  for (int i = 0; i < 16; ++i) { counters[kStickyAt + i] += counters[i]; counters[i] = 0; }
}
#endif  // __HIPCC__

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
