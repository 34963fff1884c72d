// Farthest-point sampling (exact, sequential by nature).  Stands in for torch_cluster.fps as the
// reference's wlop calls it (DSS/utils/point_processing.py:473-499, :51).  Every iteration updates
// the running min-distance of all points to the sample set and takes the arg-max (ties -> lowest
// index).  Four kernels, same arithmetic, same result:
//   * k_fps: one 1024-lane workgroup per cloud, min-distances in a caller workspace; an iteration costs
//     ~16 B x len through ONE CU (3 us per sample at 5 000 points).  Kept as the plain statement of the
//     algorithm the others are tested against (ISO_FPS_ONE_WORKGROUP=1) and for clouds beyond 4 M points.
//   * k_fps_reg (clouds below 4 k points): one workgroup, points and
//     min-distances in registers, two barriers and no memory access per sample: 1.3 us per sample at
//     5 000 points.
//   * k_fps_grid (clouds of >= 4 k points): a cooperative launch of up to 256 workgroups; every
//     thread keeps its <= 16 points AND their min-distances in registers (nothing is read from
//     memory inside the loop except the winner's coordinates); the workgroup maxima -- 64-bit keys
//     (distance bits, ~index) -- meet through one store per workgroup and one polling load per lane of
//     wave 0 (no atomics, no counter barrier: see the comment at the kernel).  One device-wide exchange
//     (~2.2 us) per sample: 3.5-3.7 us per sample at 500 k points.
//   * k_fps_lazy (the default for 4 k .. 2 M points): the same layout, but a workgroup publishes its FOUR
//     largest keys and every workgroup replays the selection on the published lists for as long as its
//     outcome is certain -- 35 samples per exchange on average at 500 k points: 0.8 us per sample
//     (5 000 of 500 k: 18.3 -> 4.5 ms), the sequence identical sample for sample.
#include <float.h>
#include <stdlib.h>
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

// ---- grid-wide form ----------------------------------------------------------------------------
// The workgroups agree on an iteration's winner through ONE round trip of plain memory operations: workgroup b stores its
// best key to slots[s & 1][b] (one agent-scope 8-byte store), wave 0 of every workgroup polls all nb slots (a slot per
// lane and trip) until each carries the iteration's tag, and reduces them with shuffles -- every workgroup computes the
// same maximum.  Until round 5 this was an atomicMax on one word + an arrival counter + a poll of the counter + a load of
// the maximum: four dependent device-scope round trips (5.0 us per sample at 500 k points, of which the distance update
// is 0.7).  Key = (distance bits : ~index); the index is < 2^31, so bit 31 of ~index is always set and carries the tag
// ((s >> 1) & 1) ^ 1 instead: a slot is rewritten every second iteration, its previous content has the other tag, and the
// zeroed control block matches neither iteration 0 nor 1.
constexpr int kFpsMaxGrid = 256;
struct FpsCtl { unsigned long long slot[2][kFpsMaxGrid]; };

template <int PPT>
__global__ __launch_bounds__(FPS_BLOCK) void k_fps_grid(const float* __restrict__ p, const int64_t* __restrict__ lengths,
                                                        const int64_t* __restrict__ n_samples,
                                                        const int64_t* __restrict__ start, int n, int64_t p_stride,
                                                        FpsCtl* ctl, int64_t* __restrict__ out) {
  __shared__ unsigned long long s_key[FPS_BLOCK / 64];
  __shared__ int s_cur;
  const int64_t len = lengths ? lengths[n] : p_stride;
  const int64_t ns = n_samples[n] < len ? n_samples[n] : len;
  if (len <= 0 || ns <= 0) return;                      // uniform over the grid
  const int t = threadIdx.x;
  const int lane = t & 63;
  const unsigned nb = gridDim.x;
  const int64_t stride = (int64_t)nb * FPS_BLOCK;
  const int64_t first = (int64_t)blockIdx.x * FPS_BLOCK + t;
  float px[PPT], py[PPT], pz[PPT], mind[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int64_t i = first + k * stride;
    px[k] = py[k] = pz[k] = 0.f;
    mind[k] = -1.0f;                                    // beyond the cloud: never the maximum
    if (i < len) { px[k] = p[i * 3]; py[k] = p[i * 3 + 1]; pz[k] = p[i * 3 + 2]; mind[k] = FLT_MAX; }
  }
  int cur = (int)(start[n] < len ? (start[n] < 0 ? 0 : start[n]) : len - 1);
  for (int64_t s = 0; s < ns; ++s) {
    if (blockIdx.x == 0 && t == 0) out[s] = cur;
    if (s + 1 == ns) break;
    const float cx = p[(int64_t)cur * 3], cy = p[(int64_t)cur * 3 + 1], cz = p[(int64_t)cur * 3 + 2];
    // a thread's points are in ascending index order: the first strict maximum is its (largest distance, lowest index)
    float bm = -1.0f;
    int bk = 0;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const float dx = px[k] - cx, dy = py[k] - cy, dz = pz[k] - cz;
      const float d = (dx * dx + dy * dy) + dz * dz;
      const float m = fminf(mind[k], d);
      mind[k] = m;
      if (m > bm) { bm = m; bk = k; }
    }
    // (distance bits, ~index): the largest key is the largest distance, lowest index among equals; 0 = no point
    unsigned long long key = bm < 0.f ? 0ull
        : (((unsigned long long)__float_as_uint(bm) << 32) | (unsigned)(0xffffffffu - (unsigned)(first + bk * stride)));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long ok = __shfl_xor(key, o);
      key = ok > key ? ok : key;
    }
    if (lane == 0) s_key[t >> 6] = key;
    __syncthreads();
    if (t < 64) {                                       // wave 0: the workgroup's maximum, the exchange, the winner
      unsigned long long best = lane < FPS_BLOCK / 64 ? s_key[lane] : 0ull;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {
        const unsigned long long ok = __shfl_xor(best, o);
        best = ok > best ? ok : best;
      }
      const unsigned long long tag = (unsigned long long)((((unsigned)s >> 1) & 1u) ^ 1u) << 31;
      const unsigned long long tmask = 1ull << 31;
      unsigned long long* slots = ctl->slot[s & 1];
      if (lane == 0)
        __hip_atomic_store(&slots[blockIdx.x], (best & ~tmask) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned long long win = 0;
      for (unsigned b0 = 0; b0 < nb; b0 += 64) {
        const unsigned b = b0 + lane;
        unsigned long long v = tag;                    // lanes beyond the grid: a matching tag, the smallest key
        if (b < nb) {
          v = __hip_atomic_load(&slots[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          while ((v & tmask) != tag) {
            __builtin_amdgcn_s_sleep(1);
            v = __hip_atomic_load(&slots[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        win = v > win ? v : win;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long ok = __shfl_xor(win, o);
        win = ok > win ? ok : win;
      }
      // every key of the iteration carries the same tag bit, so the order of (distance, ~index) is untouched; the bit is
      // part of ~index (always 1 there): put it back
      if (lane == 0) s_cur = (int)(0xffffffffu - ((unsigned)(win & 0xffffffffull) | 0x80000000u));
    }
    __syncthreads();
    cur = s_cur;
  }
}

// wave-wide maximum of a 64-bit key through the DPP network (row shifts inside the 16-lane rows, then the two row
// broadcasts of gfx9; ~6 dependent VALU operations instead of six LDS-crossbar round trips of __shfl_xor); every lane
// gets the result
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#define FPS_DPP_STEP(CTRL, ROWS)                                                                        \
  {                                                                                                     \
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, ROWS, 0xf, false); \
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, ROWS, 0xf, false); \
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;                                    \
    v = o > v ? o : v;                                                                                  \
  }
  FPS_DPP_STEP(0x111, 0xf)      // row_shr:1
  FPS_DPP_STEP(0x112, 0xf)      // row_shr:2
  FPS_DPP_STEP(0x114, 0xf)      // row_shr:4
  FPS_DPP_STEP(0x118, 0xf)      // row_shr:8   -> lane 15 of every row holds the row's maximum
  FPS_DPP_STEP(0x142, 0xa)      // row_bcast:15 into rows 1 and 3
  FPS_DPP_STEP(0x143, 0xc)      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's maximum
#undef FPS_DPP_STEP
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
  return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#define FPS_DPP_STEP32(CTRL, ROWS) { const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWS, 0xf, false); v = o > v ? o : v; }
  FPS_DPP_STEP32(0x111, 0xf) FPS_DPP_STEP32(0x112, 0xf) FPS_DPP_STEP32(0x114, 0xf) FPS_DPP_STEP32(0x118, 0xf)
  FPS_DPP_STEP32(0x142, 0xa) FPS_DPP_STEP32(0x143, 0xc)
#undef FPS_DPP_STEP32
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// ---- one workgroup, points in registers (clouds below 8 k points: the reference's own sizes) --------------------
// k_fps walks the cloud in memory every sample (3 us per sample at 5 000 points: five dependent L2 round trips per thread,
// a serial final reduction).  Here a thread keeps its <= 8 points and their min-distances in registers, the workgroup's
// maximum is one DPP reduction per wave + sixteen LDS words read by everybody, and the winner's position comes from its owner
// through LDS: two barriers and no memory access per sample.
template <int PPT>
__global__ __launch_bounds__(FPS_BLOCK) void k_fps_reg(const float* __restrict__ pts, const int64_t* __restrict__ lengths,
                                                       const int64_t* __restrict__ n_samples,
                                                       const int64_t* __restrict__ start, int64_t p_stride,
                                                       int64_t out_stride, int64_t* __restrict__ out_idx) {
  __shared__ unsigned long long s_key[2][FPS_BLOCK / 64];
  __shared__ float s_pos[2][3];
  const int n = blockIdx.x;
  const int64_t len = lengths ? lengths[n] : p_stride;
  const int64_t ns = n_samples[n] < len ? n_samples[n] : len;
  if (len <= 0 || ns <= 0) return;
  const float* p = pts + (int64_t)n * p_stride * 3;
  int64_t* out = out_idx + (int64_t)n * out_stride;
  const int t = threadIdx.x, lane = t & 63;
  float px[PPT], py[PPT], pz[PPT], mind[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int64_t i = t + (int64_t)k * FPS_BLOCK;
    px[k] = py[k] = pz[k] = 0.f;
    mind[k] = -1.0f;
    if (i < len) { px[k] = p[i * 3]; py[k] = p[i * 3 + 1]; pz[k] = p[i * 3 + 2]; mind[k] = FLT_MAX; }
  }
  const int cur0 = (int)(start[n] < len ? (start[n] < 0 ? 0 : start[n]) : len - 1);
  float cx = p[(int64_t)cur0 * 3], cy = p[(int64_t)cur0 * 3 + 1], cz = p[(int64_t)cur0 * 3 + 2];
  if (t == 0) out[0] = cur0;
  for (int64_t s = 1; s < ns; ++s) {
    float bm = -1.0f;
    int bk = 0;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const float dx = px[k] - cx, dy = py[k] - cy, dz = pz[k] - cz;
      const float d = (dx * dx + dy * dy) + dz * dz;
      const float m = fminf(mind[k], d);
      mind[k] = m;
      if (m > bm) { bm = m; bk = k; }                     // ascending index per thread: first strict maximum
    }
    const unsigned long long mine = bm < 0.f ? 0ull
        : (((unsigned long long)__float_as_uint(bm) << 32) | (unsigned)(0xffffffffu - (unsigned)(t + bk * FPS_BLOCK)));
    const unsigned long long wk = wave_max_u64(mine);
    const int b = (int)(s & 1);
    if (lane == 0) s_key[b][t >> 6] = wk;
    __syncthreads();
    unsigned long long best = 0;
#pragma unroll
    for (int w = 0; w < FPS_BLOCK / 64; ++w) { const unsigned long long v = s_key[b][w]; best = v > best ? v : best; }
    if (mine == best && best != 0ull) {                   // the owner (keys are unique): the winner's position
      float wx = px[0], wy = py[0], wz = pz[0];
#pragma unroll
      for (int k = 1; k < PPT; ++k) { const bool h = bk == k; wx = h ? px[k] : wx; wy = h ? py[k] : wy; wz = h ? pz[k] : wz; }
      s_pos[b][0] = wx; s_pos[b][1] = wy; s_pos[b][2] = wz;
      out[s] = (int64_t)(0xffffffffu - (unsigned)(best & 0xffffffffull));
    }
    __syncthreads();
    cx = s_pos[b][0]; cy = s_pos[b][1]; cz = s_pos[b][2];
  }
}

// ---- grid-wide form, several samples per exchange ------------------------------------------------
// k_fps_grid pays one device-wide exchange (~2.2 us) per sample.  Here a workgroup publishes its T LARGEST keys instead of
// one, and every workgroup replays the selection on the published lists until the outcome stops being certain:
//   * the listed points of workgroup j are known exactly -- index, position (read from p), current min-distance -- so every
//     workgroup can update them for each sample it decides, with the arithmetic their owner uses;
//   * the unlisted points of j lie below B_j = j's T-th published key for the rest of the round (distances only shrink);
//   * the largest listed key C is a real point's exact key; it is THE arg-max iff C > B_j for every workgroup whose own best
//     listed key has fallen below B_j (for the others B_j <= their best listed key <= C).
// The sequence is the sequential one, sample for sample (same keys, same (distance, lowest index) order); only the number
// of samples per exchange varies -- and every workgroup computes the same number from the same lists.  Then all threads
// apply the decided samples to their own points and the next lists are drawn.
constexpr int kFpsT = 4;            // listed keys per workgroup
constexpr int kFpsMaxRun = 128;     // samples decided per exchange at most
constexpr int kFpsLazyGrid = 128;   // workgroups at most (their lists live in wave 0's registers: 2 workgroups x T entries per lane)
struct FpsCtlLazy { unsigned long long slot[2][kFpsLazyGrid * kFpsT]; };

template <int PPT>
__global__ __launch_bounds__(FPS_BLOCK) void k_fps_lazy(const float* __restrict__ p, const int64_t* __restrict__ lengths,
                                                        const int64_t* __restrict__ n_samples,
                                                        const int64_t* __restrict__ start, int n, int64_t p_stride,
                                                        FpsCtlLazy* ctl, int64_t* __restrict__ out) {
  constexpr int T = kFpsT, NWL = kFpsLazyGrid / 64;
  __shared__ unsigned long long s_key[2][FPS_BLOCK / 64];
  __shared__ unsigned long long s_ent[kFpsLazyGrid * T];
  __shared__ float s_ex[kFpsLazyGrid * T], s_ey[kFpsLazyGrid * T], s_ez[kFpsLazyGrid * T];
  __shared__ float s_sx[kFpsMaxRun], s_sy[kFpsMaxRun], s_sz[kFpsMaxRun];
  __shared__ int s_m;
  const int64_t len = lengths ? lengths[n] : p_stride;
  const int64_t ns = n_samples[n] < len ? n_samples[n] : len;
  if (len <= 0 || ns <= 0) return;                      // uniform over the grid
  const int t = threadIdx.x;
  const int lane = t & 63;
  const unsigned nb = gridDim.x;
  const int64_t stride = (int64_t)nb * FPS_BLOCK;
  const int64_t first = (int64_t)blockIdx.x * FPS_BLOCK + t;
  float px[PPT], py[PPT], pz[PPT], mind[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int64_t i = first + k * stride;
    px[k] = py[k] = pz[k] = 0.f;
    mind[k] = -1.0f;                                    // beyond the cloud: never listed
    if (i < len) { px[k] = p[i * 3]; py[k] = p[i * 3 + 1]; pz[k] = p[i * 3 + 2]; mind[k] = FLT_MAX; }
  }
  if (t == 0) {
    const int cur = (int)(start[n] < len ? (start[n] < 0 ? 0 : start[n]) : len - 1);
    s_sx[0] = p[(int64_t)cur * 3]; s_sy[0] = p[(int64_t)cur * 3 + 1]; s_sz[0] = p[(int64_t)cur * 3 + 2];
    s_m = 1;
    if (blockIdx.x == 0) out[0] = cur;
  }
  __syncthreads();
  const unsigned long long tmask = 1ull << 31;
  int64_t done = 0;
  for (unsigned r = 0;; ++r) {
    const int m = s_m;
    for (int i = 0; i < m; ++i) {
      const float cx = s_sx[i], cy = s_sy[i], cz = s_sz[i];
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        const float dx = px[k] - cx, dy = py[k] - cy, dz = pz[k] - cz;
        const float d = (dx * dx + dy * dy) + dz * dz;
        mind[k] = fminf(mind[k], d);
      }
    }
    done += m;
    if (done >= ns) break;
    // the workgroup's T largest keys (a thread's points are in ascending index order: the first strict maximum among those
    // not listed yet is its (largest distance, lowest index))
    unsigned excl = 0;
    unsigned long long top[T];
#pragma unroll
    for (int q = 0; q < T; ++q) {
      float bm = -1.0f;
      int bk = 0;
#pragma unroll
      for (int k = 0; k < PPT; ++k)
        if (!((excl >> k) & 1u) && mind[k] > bm) { bm = mind[k]; bk = k; }
      const unsigned long long mine = bm < 0.f ? 0ull
          : (((unsigned long long)__float_as_uint(bm) << 32) | (unsigned)(0xffffffffu - (unsigned)(first + bk * stride)));
      const unsigned long long key = wave_max_u64(mine);
      if (lane == 0) s_key[q & 1][t >> 6] = key;
      __syncthreads();
      unsigned long long best = 0;
#pragma unroll
      for (int w = 0; w < FPS_BLOCK / 64; ++w) { const unsigned long long v = s_key[q & 1][w]; best = v > best ? v : best; }
      top[q] = best;
      if (best != 0 && mine == best) excl |= 1u << bk;
    }
    const unsigned long long tag = (unsigned long long)(((r >> 1) & 1u) ^ 1u) << 31;
    unsigned long long* slots = ctl->slot[r & 1];
    if (t < T) {
      unsigned long long v = top[0];
#pragma unroll
      for (int q = 1; q < T; ++q) v = t == q ? top[q] : v;
      __hip_atomic_store(&slots[blockIdx.x * T + t], (v & ~tmask) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // every thread fetches one published key and that point's position
    if ((unsigned)t < nb * T) {
      unsigned long long v = __hip_atomic_load(&slots[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while ((v & tmask) != tag) {
        __builtin_amdgcn_s_sleep(1);
        v = __hip_atomic_load(&slots[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if ((v & ~tmask) == 0ull) {
        s_ent[t] = 0ull;
      } else {
        v |= tmask;                                       // the bit is part of ~index (always 1 there)
        const int64_t idx = (int64_t)(0xffffffffu - (unsigned)(v & 0xffffffffull));
        s_ent[t] = v;
        s_ex[t] = p[idx * 3]; s_ey[t] = p[idx * 3 + 1]; s_ez[t] = p[idx * 3 + 2];
      }
    }
    __syncthreads();
    if (t < 64) {                                         // wave 0: the selection replayed on the lists
      unsigned long long ek[NWL][T], eb[NWL];
      float ex[NWL][T], ey[NWL][T], ez[NWL][T];
#pragma unroll
      for (int i = 0; i < NWL; ++i) {
        const unsigned g = (unsigned)lane + 64u * i;
#pragma unroll
        for (int q = 0; q < T; ++q) {
          ek[i][q] = 0ull; ex[i][q] = ey[i][q] = ez[i][q] = 0.f;
          if (g < nb) { ek[i][q] = s_ent[g * T + q]; ex[i][q] = s_ex[g * T + q]; ey[i][q] = s_ey[g * T + q]; ez[i][q] = s_ez[g * T + q]; }
        }
        eb[i] = ek[i][T - 1];
      }
      const int64_t left = ns - done;
      const int lim = left < kFpsMaxRun ? (int)left : kFpsMaxRun;
      int mm = 0;
      while (mm < lim) {
        // per lane: its best listed key with that entry's position, and the bound of its uncertain workgroups
        unsigned long long mine = 0ull, blk = 0ull;
        float wx = 0.f, wy = 0.f, wz = 0.f;
#pragma unroll
        for (int i = 0; i < NWL; ++i) {
          unsigned long long bl = ek[i][0];
          float bx = ex[i][0], by = ey[i][0], bz = ez[i][0];
#pragma unroll
          for (int q = 1; q < T; ++q) {
            const bool gt = ek[i][q] > bl;
            bl = gt ? ek[i][q] : bl; bx = gt ? ex[i][q] : bx; by = gt ? ey[i][q] : by; bz = gt ? ez[i][q] : bz;
          }
          if (bl < eb[i]) blk = eb[i] > blk ? eb[i] : blk;
          const bool gt = bl > mine;
          mine = gt ? bl : mine; wx = gt ? bx : wx; wy = gt ? by : wy; wz = gt ? bz : wz;
        }
        // the wave's maximum C.  Fast path: the largest DISTANCE word sits in one lane only (then that lane's key is C);
        // equal distances in several lanes take the full 64-bit reduction -- same result either way
        const unsigned cd = wave_max_u32((unsigned)(mine >> 32));
        const unsigned long long at = (unsigned long long)__ballot((unsigned)(mine >> 32) == cd);
        unsigned long long c;
        int src;
        if (__popcll(at) == 1) {
          src = __builtin_ctzll(at);
          c = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(mine >> 32), src) << 32)
              | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)mine, src);
        } else {
          c = wave_max_u64(mine);
          src = __builtin_ctzll((unsigned long long)__ballot(mine == c));
        }
        if (c == 0ull) break;
        // C must exceed the bounds of the uncertain workgroups: decided on the distance words unless they are equal
        const unsigned bd = wave_max_u32((unsigned)(blk >> 32));
        if (bd > cd) break;
        if (bd == cd && !(c > wave_max_u64(blk))) break;
        wx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wx), src));
        wy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wy), src));
        wz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wz), src));
        if (lane == 0) {
          s_sx[mm] = wx; s_sy[mm] = wy; s_sz[mm] = wz;
          if (blockIdx.x == 0) out[done + mm] = (int64_t)(0xffffffffu - (unsigned)(c & 0xffffffffull));
        }
#pragma unroll
        for (int i = 0; i < NWL; ++i)
#pragma unroll
          for (int q = 0; q < T; ++q) {
            const float dx = ex[i][q] - wx, dy = ey[i][q] - wy, dz = ez[i][q] - wz;
            const float d = (dx * dx + dy * dy) + dz * dz;
            const unsigned od = (unsigned)(ek[i][q] >> 32);
            const unsigned nd = __float_as_uint(fminf(__uint_as_float(od), d));
            // an absent entry (key 0) stays 0: its distance word is 0 and min(0, d) = 0 for d >= 0
            ek[i][q] = ((unsigned long long)nd << 32) | (ek[i][q] & 0xffffffffull);
          }
        ++mm;
      }
      if (lane == 0) s_m = mm;
    }
    __syncthreads();
  }
}

template <int PPT>
hipError_t launch_fps_lazy(int nb, const float* p, const int64_t* lengths, const int64_t* n_samples,
                           const int64_t* start, int n, int64_t p_stride, FpsCtlLazy* ctl, int64_t* out, hipStream_t s) {
  void* args[] = {(void*)&p, (void*)&lengths, (void*)&n_samples, (void*)&start, (void*)&n, (void*)&p_stride,
                  (void*)&ctl, (void*)&out};
  return hipLaunchCooperativeKernel(reinterpret_cast<const void*>(&k_fps_lazy<PPT>), dim3(nb), dim3(FPS_BLOCK), args,
                                    0, s);
}

template <int PPT>
hipError_t launch_fps_grid(int nb, const float* p, const int64_t* lengths, const int64_t* n_samples,
                           const int64_t* start, int n, int64_t p_stride, FpsCtl* ctl, int64_t* out, hipStream_t s) {
  void* args[] = {(void*)&p, (void*)&lengths, (void*)&n_samples, (void*)&start, (void*)&n, (void*)&p_stride,
                  (void*)&ctl, (void*)&out};
  return hipLaunchCooperativeKernel(reinterpret_cast<const void*>(&k_fps_grid<PPT>), dim3(nb), dim3(FPS_BLOCK), args,
                                    0, s);
}

constexpr int64_t kFpsGridMin = 4096;     // below: one workgroup (k_fps_reg) is faster (2 500 points: 1.1 against 1.4 us per sample; 5 000: 1.3 against 1.2)
constexpr int kFpsCtlFloats = (sizeof(FpsCtlLazy) > sizeof(FpsCtl) ? sizeof(FpsCtlLazy) : sizeof(FpsCtl)) / 4;   // control block at the end of the workspace

}  // namespace

extern "C" int64_t iso_farthest_point_sampling_work_floats(int n_clouds, int64_t p_stride) {
  if (n_clouds < 0 || p_stride < 0) return 0;
  return (int64_t)n_clouds * p_stride + kFpsCtlFloats + 2;
}

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
  hipStream_t st = (hipStream_t)stream;
  static int64_t grid_min = 0;                       // ISO_FPS_GRID_MIN: development override of the size the grid-wide forms start at
  if (grid_min == 0) { const char* e = getenv("ISO_FPS_GRID_MIN"); grid_min = e ? atoll(e) : kFpsGridMin; if (grid_min < 2048) grid_min = 2048; }
  if (p_stride >= grid_min && p_stride <= (int64_t)256 * FPS_BLOCK * 16 && !getenv("ISO_FPS_ONE_WORKGROUP")) {
    // grid-wide form, cloud after cloud; the smallest grid that keeps <= 8 points per thread
    // (measured: 2.8 us per sample up to 50 k points, 5.0 at 500 k, 7.1 at 1 M -- the barrier's atomic
    // round trips, not the arithmetic; fewer, fatter workgroups are not faster)
    static int lazy = -1;                           // ISO_FPS_LAZY=0: one sample per exchange (k_fps_grid)
    if (lazy < 0) { const char* e = getenv("ISO_FPS_LAZY"); lazy = e ? atoi(e) != 0 : 1; }
    // points per thread the grid is sized for (ISO_FPS_PPT: development override).  k_fps_grid: 8 (fewer, fatter workgroups
    // shorten the exchange); k_fps_lazy: 1, i.e. as many workgroups as the cloud fills, up to 128 -- the exchange is shared by
    // many samples, more lists make longer runs (24 k points: 13.9 ms at 4 points per thread, 10.1 at 1)
    static int ppt_target = 0;
    if (ppt_target == 0) { const char* e = getenv("ISO_FPS_PPT"); ppt_target = e ? atoi(e) : (lazy ? 1 : 8); if (ppt_target < 1 || ppt_target > 16) ppt_target = lazy ? 1 : 8; }
    const bool use_lazy = lazy && p_stride <= (int64_t)kFpsLazyGrid * FPS_BLOCK * 16;
    const int64_t nb_max = use_lazy ? kFpsLazyGrid : 256;
    int64_t nb = (p_stride + FPS_BLOCK * ppt_target - 1) / (FPS_BLOCK * ppt_target);
    nb = nb < 2 ? 2 : (nb > nb_max ? nb_max : nb);
    const int64_t ppt = (p_stride + nb * FPS_BLOCK - 1) / (nb * FPS_BLOCK);
    // (8-byte slots: the control block starts at the next 8-byte boundary; kFpsCtlFloats leaves room for it)
    FpsCtl* ctl = reinterpret_cast<FpsCtl*>(((uintptr_t)(work + (int64_t)n_clouds * p_stride) + 7) & ~(uintptr_t)7);
    bool refused = false;
    for (int n = 0; n < n_clouds; ++n) {
      (void)hipMemsetAsync(ctl, 0, kFpsCtlFloats * 4, st);
      const float* p = points + (int64_t)n * p_stride * 3;
      int64_t* out = out_idx + (int64_t)n * out_stride;
      hipError_t e;
      if (use_lazy) {
        FpsCtlLazy* cl = reinterpret_cast<FpsCtlLazy*>(ctl);
        if (ppt <= 1) e = launch_fps_lazy<1>((int)nb, p, lengths, n_samples, start, n, p_stride, cl, out, st);
        else if (ppt <= 2) e = launch_fps_lazy<2>((int)nb, p, lengths, n_samples, start, n, p_stride, cl, out, st);
        else if (ppt <= 4) e = launch_fps_lazy<4>((int)nb, p, lengths, n_samples, start, n, p_stride, cl, out, st);
        else if (ppt <= 8) e = launch_fps_lazy<8>((int)nb, p, lengths, n_samples, start, n, p_stride, cl, out, st);
        else e = launch_fps_lazy<16>((int)nb, p, lengths, n_samples, start, n, p_stride, cl, out, st);
      } else
      if (ppt <= 1) e = launch_fps_grid<1>((int)nb, p, lengths, n_samples, start, n, p_stride, ctl, out, st);
      else if (ppt <= 2) e = launch_fps_grid<2>((int)nb, p, lengths, n_samples, start, n, p_stride, ctl, out, st);
      else if (ppt <= 4) e = launch_fps_grid<4>((int)nb, p, lengths, n_samples, start, n, p_stride, ctl, out, st);
      else if (ppt <= 8) e = launch_fps_grid<8>((int)nb, p, lengths, n_samples, start, n, p_stride, ctl, out, st);
      else e = launch_fps_grid<16>((int)nb, p, lengths, n_samples, start, n, p_stride, ctl, out, st);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        ISO_REQUIRE(n == 0, ISO_ERR_LAUNCH, "iso_farthest_point_sampling: cooperative launch failed: %s",
                    hipGetErrorString(e));
        refused = true;
        break;
      }
    }
    if (!refused) { ISO_CHECK_LAUNCH("iso_farthest_point_sampling"); return ISO_OK; }
    // the device cannot co-schedule the grid (first cloud refused): the one-workgroup form below
  }
  if (p_stride <= (int64_t)FPS_BLOCK * 8 && !getenv("ISO_FPS_ONE_WORKGROUP")) {
    // the cloud fits one workgroup's registers (ISO_FPS_ONE_WORKGROUP keeps the memory-walking kernel for the tests)
    const int64_t ppt = (p_stride + FPS_BLOCK - 1) / FPS_BLOCK;
    if (ppt <= 1) hipLaunchKernelGGL(k_fps_reg<1>, dim3(n_clouds), dim3(FPS_BLOCK), 0, st, points, lengths, n_samples, start, p_stride, out_stride, out_idx);
    else if (ppt <= 2) hipLaunchKernelGGL(k_fps_reg<2>, dim3(n_clouds), dim3(FPS_BLOCK), 0, st, points, lengths, n_samples, start, p_stride, out_stride, out_idx);
    else if (ppt <= 4) hipLaunchKernelGGL(k_fps_reg<4>, dim3(n_clouds), dim3(FPS_BLOCK), 0, st, points, lengths, n_samples, start, p_stride, out_stride, out_idx);
    else hipLaunchKernelGGL(k_fps_reg<8>, dim3(n_clouds), dim3(FPS_BLOCK), 0, st, points, lengths, n_samples, start, p_stride, out_stride, out_idx);
    ISO_CHECK_LAUNCH("iso_farthest_point_sampling");
    return ISO_OK;
  }
  hipLaunchKernelGGL(k_fps, dim3(n_clouds), dim3(FPS_BLOCK), 0, st, points, lengths,
                     n_samples, start, p_stride, out_stride, work, out_idx);
  ISO_CHECK_LAUNCH("iso_farthest_point_sampling");
  return ISO_OK;
}
