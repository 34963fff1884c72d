// Brick grid (bricks.h): build, the fused resample step (FRNN K+1 query + tangent-plane
// repulsion, neighbour data staged in LDS) and the fused per-view splat bandwidth h
// (K = 7 query of every view's visible cloud in ONE pass over the whole cloud).
//
// Reference semantics
//   neighbour tree  : UniformProjection._create_tree, DSS/models/levelset_sampling.py:110-140
//                     (r = sqrt(diag/P) * knn_k, K+1 self-inclusive FRNN query, column 0 dropped)
//   repulsion       : UniformProjection.resample, levelset_sampling.py:254-284
//   splat bandwidth : SurfaceSplatting._get_per_point_info, DSS/core/rasterizer.py:367-386
//                     (K = 7 self query of the filtered view cloud, h = clamp(max d2 / 2, 5e-5, 0.01))
//   renderable flags: SurfaceSplatting._filter_points_with_invalid_depth / backface culling,
//                     rasterizer.py:184-254
// The stand-alone iso_frnn_* entry points (frnn.hip) stay the general API (any K, any radius,
// foreign query sets); these kernels are the hot path of the iso-point cycle.  Results are the same
// exact "K nearest within r, ties to the lower index" lists; tests pin one against the other.
//
// Selection without per-candidate branches: a lane keeps the M = K + 3 smallest 32-bit keys
// (d2 bits with the low 10 bits replaced by the candidate's LDS slot) in a sorted register list,
// one v_med3_u32 per slot and candidate.  Truncated keys order like d2 up to 2^-13 relative, so the
// exact (d2, id) order is restored afterwards on the M survivors; the list is certified complete
// when the first key beyond the K-th differs from it in the kept bits (else: tail kernel).
#include <type_traits>
#include "bricks.h"

#pragma clang fp contract(off)

// ------------------------------------------------------------------------------------------------
constexpr int kScan1MaxWords = 2048;        // = kScan1Max (k_brick_offsets1): chunk totals the one-launch scan keeps zeroed (BrickWs::scan1)
BrickWs bricks_carve(void* ws, int64_t n_max) {
  BrickWs w;
  if (n_max < 1) n_max = 1;
  w.nb_cap = bricks_nb_cap(n_max);
  w.G = BK_CPB * (int64_t)w.nb_cap * w.nb_cap * w.nb_cap + 1;   // BK_CPB counters per brick (bricks.h)
  auto al = [](int64_t b) { return (b + 255) / 256 * 256; };
  char* p = (char*)ws;
  int64_t o = 0;
  w.hdr = (BrickHdr*)(p + o); o += al(sizeof(BrickHdr));
  w.counters = (int32_t*)(p + o); o += al(4 * kCounterInts);
  w.scan1 = (unsigned*)(p + o); o += al(4 * kScan1MaxWords);          // fixed offset: independent of n_max
  w.cnt = (int32_t*)(p + o); o += al(4 * w.G);
  w.off = (int32_t*)(p + o); o += al(4 * w.G);
  w.slot = (int32_t*)(p + o); o += al(4 * n_max);
  w.rec0 = (float4*)(p + o); o += al(16 * n_max);
  w.rec1 = (float4*)(p + o); o += al(16 * n_max);
  w.list = (int32_t*)(p + o); o += al(4 * (w.G < n_max ? w.G : n_max));
  w.tail = (int32_t*)(p + o); o += al(4 * 8 * n_max);
  w.scan_ws_bytes = iso_prefix_sum_workspace_bytes(w.G, 1) + 4 * (kScan1MaxWords + (w.G + 1023) / 1024);   // (sums of the two-launch scan)
  w.scan_ws = (void*)(p + o); o += al(w.scan_ws_bytes);
  w.bytes = o;
  return w;
}

namespace {

__device__ __forceinline__ int wave_incl_scan_i(int v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(v, o);
    if (lane >= o) v += t;
  }
  return v;
}

// ---- header ------------------------------------------------------------------------------------
// bbox: [min xyz, 0, max xyz, 0] (iso_points_bbox layout; for N ranks the caller reduces it first)
__global__ void k_bricks_params(const float* __restrict__ bbox, int n_boxes, BrickParams q, BrickHdr* __restrict__ h,
                                int32_t* __restrict__ counters) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float mn[3], mx[3];
  for (int a = 0; a < 3; ++a) {                     // the union of the n_boxes boxes (N ranks: every rank's local box)
    float lo = bbox[a], hi = bbox[4 + a];
    for (int k = 1; k < n_boxes; ++k) { lo = fminf(lo, bbox[k * 8 + a]); hi = fmaxf(hi, bbox[k * 8 + 4 + a]); }
    mn[a] = lo; mx[a] = hi;
  }
  bricks_store_header(bricks_header(mn, mx, q), h, counters);
}

// Bounding box of a packed (n,3) cloud into the workspace's PENDING BOX (bricks.h): flat float index (consecutive lanes
// read consecutive dwords; the grid stride is a multiple of 3 floats, so a thread stays on one axis and keeps four loads
// in flight), registers -> wave shuffles -> LDS -> six atomics per workgroup on one of kBoxCopies copies.  The header is
// made from it by the count pass that follows.
__global__ __launch_bounds__(256) void k_brick_bbox(const float* __restrict__ p, int64_t n, int32_t* __restrict__ counters) {
  __shared__ float s_box[4][6];
  const int64_t nfl = n * 3;
  const int64_t stride = (int64_t)gridDim.x * 256;            // a multiple of 3 (launcher)
  const int64_t i_first = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float lo = FLT_MAX, hi = -FLT_MAX;
  int64_t i = i_first;
  for (; i + 3 * stride < nfl; i += 4 * stride) {
    const float v0 = p[i], v1 = p[i + stride], v2 = p[i + 2 * stride], v3 = p[i + 3 * stride];
    lo = fminf(fminf(lo, v0), fminf(v1, fminf(v2, v3)));
    hi = fmaxf(fmaxf(hi, v0), fmaxf(v1, fmaxf(v2, v3)));
  }
  for (; i < nfl; i += stride) { const float v = p[i]; lo = fminf(lo, v); hi = fmaxf(hi, v); }
  const int ax = (int)(i_first % 3);
  BkBox box;
#pragma unroll
  for (int c = 0; c < 3; ++c) { box.lo[c] = ax == c ? lo : FLT_MAX; box.hi[c] = ax == c ? hi : -FLT_MAX; }
  box.commit(counters, s_box);
}

// the pending box as 8 floats (iso_points_bbox layout): what N ranks all-gather before iso_bricks_params; cleared
__global__ __launch_bounds__(256) void k_brick_box_take(int32_t* __restrict__ counters, float* __restrict__ box_out) {
  __shared__ unsigned s_red[8];
  float mn[3], mx[3];
  bk_box_read(counters, s_red, mn, mx);
  __syncthreads();
  bk_box_clear(counters);
  if (threadIdx.x == 0) {
    for (int c = 0; c < 3; ++c) { box_out[c] = mn[c]; box_out[4 + c] = mx[c]; }
    box_out[3] = 0.f; box_out[7] = 0.f;
  }
}

// zero the whole table once (iso_bricks_workspace_init); afterwards the offsets pass leaves it zeroed
__global__ void k_bricks_init(int32_t* __restrict__ counters, int32_t* __restrict__ cnt, int64_t G, unsigned* __restrict__ scan_ws) {
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i0 < kScan1MaxWords) scan_ws[i0] = 0u;
  for (int64_t i = i0; i < G; i += (int64_t)gridDim.x * blockDim.x) cnt[i] = 0;
  for (int64_t i = i0; i < kCounterInts; i += (int64_t)gridDim.x * blockDim.x) counters[i] = i == kMagicAt ? kBrickMagic : 0;
}

// F.normalize(n, dim=-1): n / max(|n|, 1e-12) (levelset_sampling.py:258), stored in rec1 in place of the raw normal
__device__ __forceinline__ void bk_unit_normal(float& ux, float& uy, float& uz) {
  float un = sqrtf((ux * ux + uy * uy) + uz * uz);
  un = un > 1e-12f ? un : 1e-12f;
  ux = ux / un; uy = uy / un; uz = uz / un;
}

__device__ __forceinline__ int brick_of(const BrickHdr& h, float x, float y, float z) {
  const int fx = bk_fine(x, h.mn[0], h.inv_f, h.nf[0]);
  const int fy = bk_fine(y, h.mn[1], h.inv_f, h.nf[1]);
  const int fz = bk_fine(z, h.mn[2], h.inv_f, h.nf[2]);
  return ((fx >> 2) * h.nb[1] + (fy >> 2)) * h.nb[2] + (fz >> 2);
}
// counter of a point: BK_CPB * brick + sub-brick (bricks.h)
__device__ __forceinline__ int counter_of(const BrickHdr& h, float x, float y, float z) {
  const int fx = bk_fine(x, h.mn[0], h.inv_f, h.nf[0]);
  const int fy = bk_fine(y, h.mn[1], h.inv_f, h.nf[1]);
  const int fz = bk_fine(z, h.mn[2], h.inv_f, h.nf[2]);
  const int brick = ((fx >> 2) * h.nb[1] + (fy >> 2)) * h.nb[2] + (fz >> 2);
  return BK_CPB * brick + ((((fx >> 1) & 1) * 2 + ((fy >> 1) & 1)) * 2 + ((fz >> 1) & 1));
}

// Arrival slot of a record inside its brick (own points: packed (n,3) f32; imported halo records: float4).  The records
// of brick b are [off[BK_CPB b], off[BK_CPB (b + 1)]) whatever the order inside.  A workgroup first counts its 1024
// records per brick in an LDS hash table and then reserves each brick's range with ONE returning global atomic (a
// record order with any spatial coherence -- the cycle's clouds run along a z-order curve -- puts many of a round's
// records in the same brick; one returning atomic per record was the whole cost of this pass).  All records of a
// workgroup that fall into one counter (sub-brick) get consecutive ranks; slot = rank; a record that finds no table
// entry within kCntProbe probes takes its rank from global memory directly.
constexpr int kCntTab = 2048, kCntProbe = 16;

// one round of 1024 records of a workgroup: pos(i, x, y, z) fetches record i, its slot goes to slot_out[i]
template <class Pos>
__device__ __forceinline__ void brick_count_round(const BrickHdr& h, int64_t base, int64_t n, int32_t* __restrict__ cnt,
                                                  int32_t* __restrict__ slot_out, int* t_key, int* t_cnt, Pos&& pos) {
  for (int j = threadIdx.x; j < kCntTab; j += 256) { t_key[j] = -1; t_cnt[j] = 0; }
  __syncthreads();
  int e[4], rk[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t i = base + k * 256 + threadIdx.x;
    e[k] = -1; rk[k] = 0;
    if (i < n) {
      float x, y, z;
      pos(i, k, x, y, z);
      const int key = counter_of(h, x, y, z);
      unsigned at = ((unsigned)key * 2654435761u) >> 21;          // 11 bits
      bool found = false;
      for (int t = 0; t < kCntProbe && !found; ++t) {
        const int prev = atomicCAS(&t_key[at], -1, key);
        if (prev == -1 || prev == key) found = true;
        else at = (at + 1) & (kCntTab - 1);
      }
      if (found) { e[k] = (int)at; rk[k] = atomicAdd(&t_cnt[at], 1); }
      else rk[k] = atomicAdd(&cnt[key], 1);
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < kCntTab; j += 256) {
    const int c = t_cnt[j];
    if (c > 0) t_cnt[j] = atomicAdd(&cnt[t_key[j]], c);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t i = base + k * 256 + threadIdx.x;
    if (i < n) slot_out[i] = (e[k] >= 0 ? t_cnt[e[k]] : 0) + rk[k];       // rank inside the point's counter (sub-brick)
  }
  __syncthreads();
}

// pending != 0: the header is not written yet -- it is made HERE from the workspace's pending box, by every workgroup
// for itself (sixteen 32-byte words from L2 + a few dozen scalar operations; a launch of its own cost 5 us, a "last workgroup
// writes it" tail in the pass that took the box 7 us); the last workgroup also stores it for the launches that follow and
// retires the counters of the previous grid.
// job != null: workgroup 0 of the grid does not count -- it runs the chunk scan of the splat front end beside
// the others (ChunkScanJob, below).
struct ChunkScanJob {
  const int32_t* tile_cnt;   // (n_views, n_tiles) renderable points per 256-point tile (follow.h) -- or null
  int32_t* chunk;            // (n_views, n_chunks) out: exclusive offsets per 1024-point chunk (k_mask_chunk_scan's table)
  int n_tiles, n_chunks, n_views;   // n_tiles: row stride of tile_cnt (= 4 n_chunks)
  int n_real;                       // tiles the cloud really has: ceil(n / 256)
  int64_t* first; int64_t* num; int32_t* view_total;
};
__device__ void chunk_scan_job(const ChunkScanJob& j);

// imported halo records (N ranks): the LAST `blocks` workgroups of the count / scatter launch take them (launches of their
// own were 8 + 6 us per grid at the size of a rank's share of eight); blocks == 0: none
struct ImportJob { const float4* rec0; const float4* rec1; const int32_t* count; int64_t max; int blocks; };

__global__ __launch_bounds__(256) void k_brick_count(const float* __restrict__ pts, int64_t n,
                                                     BrickHdr* __restrict__ hp, int32_t* __restrict__ cnt,
                                                     int32_t* __restrict__ slot, int pending, BrickParams q,
                                                     int32_t* __restrict__ counters, ChunkScanJob job, ImportJob imp) {
  __shared__ int t_key[kCntTab], t_cnt[kCntTab];
  __shared__ BrickHdr s_h;
  if (imp.blocks && (int)blockIdx.x >= (int)gridDim.x - imp.blocks) {      // (only with a header that is already written)
    const int bid = (int)blockIdx.x - ((int)gridDim.x - imp.blocks);
    const BrickHdr h = *hp;
    int64_t m = *imp.count;
    if (m > imp.max) m = imp.max;
    if (bid == 0 && threadIdx.x == 0) hp->n = h.n_own + (int)m;
    for (int64_t base = (int64_t)bid * 1024; base < m; base += (int64_t)imp.blocks * 1024)
      brick_count_round(h, base, m, cnt, slot + h.n_own, t_key, t_cnt,
                        [&](int64_t j, int, float& x, float& y, float& z) { const float4 p = imp.rec0[j]; x = p.x; y = p.y; z = p.z; });
    return;
  }
  const bool scan_wg = job.chunk && blockIdx.x == 0;
  if (scan_wg && gridDim.x > 1) { chunk_scan_job(job); return; }        // (needs no header; another workgroup stores it)
  if (pending) {
    __shared__ unsigned s_red[8];
    float mn[3], mx[3];
    bk_box_read(counters, s_red, mn, mx);
    if (threadIdx.x == 0) {
      s_h = bricks_header(mn, mx, q);
      if (blockIdx.x == gridDim.x - 1) bricks_store_header(s_h, hp, counters);
    }
    __syncthreads();
  }
  if (scan_wg) { chunk_scan_job(job); return; }
  const BrickHdr h = pending ? s_h : *hp;
  const int n_wg = (int)gridDim.x - imp.blocks - (job.chunk ? 1 : 0);
  // (the scan job, if any, is workgroup 0: it has the longest chain of dependent steps of the launch and starts first)
  for (int64_t base = (int64_t)(blockIdx.x - (job.chunk ? 1 : 0)) * 1024; base < n; base += (int64_t)n_wg * 1024)
    brick_count_round(h, base, n, cnt, slot, t_key, t_cnt,
                      [&](int64_t i, int, float& x, float& y, float& z) { x = pts[i * 3]; y = pts[i * 3 + 1]; z = pts[i * 3 + 2]; });
}

// imported halo records (count on the device); same aggregation
__global__ __launch_bounds__(256) void k_brick_count_recs(const float4* __restrict__ imp0, const int32_t* __restrict__ imp_count,
                                                          int64_t imp_max, BrickHdr* __restrict__ hp,
                                                          int32_t* __restrict__ cnt, int32_t* __restrict__ slot) {
  __shared__ int t_key[kCntTab], t_cnt[kCntTab];
  const BrickHdr h = *hp;
  int64_t m = *imp_count;
  if (m > imp_max) m = imp_max;
  if (blockIdx.x == 0 && threadIdx.x == 0) hp->n = h.n_own + (int)m;
  for (int64_t base = (int64_t)blockIdx.x * 1024; base < m; base += (int64_t)gridDim.x * 1024)
    brick_count_round(h, base, m, cnt, slot + h.n_own, t_key, t_cnt,
                      [&](int64_t j, int, float& x, float& y, float& z) { const float4 p = imp0[j]; x = p.x; y = p.y; z = p.z; });
}

__global__ __launch_bounds__(256) void k_brick_scatter(const float* __restrict__ pts, const float* __restrict__ nrm,
                                                       const int32_t* __restrict__ payload, int64_t n,
                                                       const BrickHdr* __restrict__ hp, const int32_t* __restrict__ off,
                                                       const int32_t* __restrict__ slot, float4* __restrict__ rec0,
                                                       float4* __restrict__ rec1, ImportJob imp) {
  const BrickHdr h = *hp;
  if (imp.blocks && (int)blockIdx.x >= (int)gridDim.x - imp.blocks) {
    const int bid = (int)blockIdx.x - ((int)gridDim.x - imp.blocks);
    const int64_t m = h.n - h.n_own;
    for (int64_t j = (int64_t)bid * blockDim.x + threadIdx.x; j < m; j += (int64_t)imp.blocks * blockDim.x) {
      const float4 p = imp.rec0[j];
      const int64_t dst = (int64_t)off[counter_of(h, p.x, p.y, p.z)] + slot[h.n_own + j];
      rec0[dst] = p;
      float4 u = imp.rec1[j];                              // an exported record carries the raw normal
      bk_unit_normal(u.x, u.y, u.z);
      rec1[dst] = u;
    }
    return;
  }
  const int n_wg = (int)gridDim.x - imp.blocks;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)n_wg * blockDim.x) {
    const float x = pts[i * 3], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
    const int64_t dst = (int64_t)off[counter_of(h, x, y, z)] + slot[i];            // slot: rank inside the counter (k_brick_count)
    rec0[dst] = make_float4(x, y, z, __int_as_float(h.id_base + (int)i));
    float4 u = make_float4(0.f, 0.f, 0.f, __int_as_float(payload ? payload[i] : 0));
    if (nrm) { u.x = nrm[i * 3]; u.y = nrm[i * 3 + 1]; u.z = nrm[i * 3 + 2]; bk_unit_normal(u.x, u.y, u.z); }
    rec1[dst] = u;
  }
}

__global__ __launch_bounds__(256) void k_brick_scatter_recs(const float4* __restrict__ imp0, const float4* __restrict__ imp1,
                                                            const BrickHdr* __restrict__ hp, const int32_t* __restrict__ off,
                                                            const int32_t* __restrict__ slot, float4* __restrict__ rec0,
                                                            float4* __restrict__ rec1) {
  const BrickHdr h = *hp;
  const int64_t m = h.n - h.n_own;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (int64_t)gridDim.x * blockDim.x) {
    const float4 p = imp0[j];
    const int64_t dst = (int64_t)off[counter_of(h, p.x, p.y, p.z)] + slot[h.n_own + j];
    rec0[dst] = p;
    float4 u = imp1[j];                                  // an exported record carries the raw normal
    bk_unit_normal(u.x, u.y, u.z);
    rec1[dst] = u;
  }
}

// ---- offsets of the bricks: exclusive scan of the counters in two launches ---------------------------
// (the general 3-phase scan of frnn.hip, specialised: a table of <= ~2000 chunks needs no middle pass -- every
// block of the second pass adds up the totals of the chunks before it -- and the second pass also does what two
// more launches did: it leaves the counters ZEROED for the next build and appends the occupied bricks to the work
// list, one returning atomic per 2048 bricks; the order of the list only affects scheduling.)
// The unit of the scan is the BRICK: a chunk is BS_CHUNK bricks (a thread takes BS_ITEMS of them, BK_CPB counters each, as
// 16-byte loads), so that the number of chunks -- and of workgroups that wait for one another in the one-launch form -- stays
// <= 2001 whatever BK_CPB is.
constexpr int BS_ITEMS = 4, BS_CHUNK = 256 * BS_ITEMS;
static_assert(BK_CPB == 8, "a brick's counters are two 16-byte words");

__device__ __forceinline__ int block_excl_scan_256(int v, int& total, int* lds /*>= 4*/) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int inc = wave_incl_scan_i(v);
  if (lane == 63) lds[w] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int t = lds[i]; if (i < w) base += t; tot += t; }
  total = tot;
  __syncthreads();
  return base + inc - v;
}

// a thread's BS_ITEMS bricks of chunk `chunk`: their counters (zeroed behind the read when ZERO) and the brick totals
template <bool ZERO>
__device__ __forceinline__ int brick_counters_load(int32_t* __restrict__ cnt, int chunk, int nb, int (&c)[BS_ITEMS][BK_CPB],
                                                   int (&tot)[BS_ITEMS]) {
  int v = 0;
#pragma unroll
  for (int k = 0; k < BS_ITEMS; ++k) {
    const int brick = chunk * BS_CHUNK + threadIdx.x * BS_ITEMS + k;
    tot[k] = 0;
#pragma unroll
    for (int q = 0; q < BK_CPB; ++q) c[k][q] = 0;
    if (brick < nb) {
      int4* p = reinterpret_cast<int4*>(cnt + (int64_t)BK_CPB * brick);
      const int4 a = p[0], b = p[1];
      c[k][0] = a.x; c[k][1] = a.y; c[k][2] = a.z; c[k][3] = a.w; c[k][4] = b.x; c[k][5] = b.y; c[k][6] = b.z; c[k][7] = b.w;
      if (ZERO) { p[0] = make_int4(0, 0, 0, 0); p[1] = make_int4(0, 0, 0, 0); }
#pragma unroll
      for (int q = 0; q < BK_CPB; ++q) tot[k] += c[k][q];
    }
    v += tot[k];
  }
  return v;
}
// offsets of those bricks' counters from the exclusive prefix `ex` of the thread, occupied bricks appended at list[at..]
__device__ __forceinline__ void brick_offsets_store(int32_t* __restrict__ off, int32_t* __restrict__ list, int chunk, int nb,
                                                    const int (&c)[BS_ITEMS][BK_CPB], const int (&tot)[BS_ITEMS], int ex, int at,
                                                    const bool (&listed)[BS_ITEMS]) {
#pragma unroll
  for (int k = 0; k < BS_ITEMS; ++k) {
    const int brick = chunk * BS_CHUNK + threadIdx.x * BS_ITEMS + k;
    if (brick < nb) {
      int o[BK_CPB];
#pragma unroll
      for (int q = 0; q < BK_CPB; ++q) { o[q] = ex; ex += c[k][q]; }
      int4* p = reinterpret_cast<int4*>(off + (int64_t)BK_CPB * brick);
      p[0] = make_int4(o[0], o[1], o[2], o[3]);
      p[1] = make_int4(o[4], o[5], o[6], o[7]);
      if (listed[k]) list[at++] = brick;
    } else if (brick == nb) {
      off[(int64_t)BK_CPB * nb] = ex;                    // the sentinel behind the last brick
    }
  }
}

// bricks of the work list: occupied, and in a brick column (x-major brick ids) that can hold points of this rank
__device__ __forceinline__ void brick_listed(const BrickHdr& h, int chunk, const int (&tot)[BS_ITEMS], bool (&listed)[BS_ITEMS]) {
  const int per_col = h.nb[1] * h.nb[2];
#pragma unroll
  for (int k = 0; k < BS_ITEMS; ++k) {
    const int brick = chunk * BS_CHUNK + threadIdx.x * BS_ITEMS + k;
    const int bx = brick / per_col;
    listed[k] = tot[k] > 0 && bx >= h.own_bx_lo && bx <= h.own_bx_hi;
  }
}

__global__ __launch_bounds__(256) void k_brick_sums(const BrickHdr* __restrict__ hp, int32_t* __restrict__ cnt,
                                                    int32_t* __restrict__ sums) {
  __shared__ int lds[4];
  const int nb = hp->n_bricks;
  if ((int)blockIdx.x * BS_CHUNK > nb) return;
  int c[BS_ITEMS][BK_CPB], t[BS_ITEMS];
  const int v = brick_counters_load<false>(cnt, blockIdx.x, nb, c, t);
  int tot;
  block_excl_scan_256(v, tot, lds);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void k_brick_offsets(const BrickHdr* __restrict__ hp, int32_t* __restrict__ cnt,
                                                       int32_t* __restrict__ off, const int32_t* __restrict__ sums,
                                                       int32_t* __restrict__ list, int32_t* __restrict__ counters) {
  __shared__ int lds[4], s_base;
  const int nb = hp->n_bricks;
  if ((int)blockIdx.x * BS_CHUNK > nb) return;           // (chunk nb / BS_CHUNK also holds the sentinel)
  if (blockIdx.x == 0) bk_box_clear(counters);
  int before = 0;
  for (int i = threadIdx.x; i < (int)blockIdx.x; i += 256) before += sums[i];
  int base;
  block_excl_scan_256(before, base, lds);                       // base = total of the chunks before this one
  int c[BS_ITEMS][BK_CPB], t[BS_ITEMS];
  const int v = brick_counters_load<true>(cnt, blockIdx.x, nb, c, t);
  int occ = 0;
  bool listed[BS_ITEMS];
  brick_listed(*hp, blockIdx.x, t, listed);
#pragma unroll
  for (int k = 0; k < BS_ITEMS; ++k) occ += listed[k] ? 1 : 0;
  int tot, occ_tot;
  const int ex = base + block_excl_scan_256(v, tot, lds);
  int at = block_excl_scan_256(occ, occ_tot, lds);
  if (threadIdx.x == 0) s_base = occ_tot ? atomicAdd(&counters[0], occ_tot) : 0;
  __syncthreads();
  brick_offsets_store(off, list, blockIdx.x, nb, c, t, ex, at + s_base, listed);
}

// The two passes above as ONE launch (2048 workgroups of this size are resident at once -- 8 per CU --, so a workgroup may
// wait for the ones before it; BK_NB_MAX bricks per axis are <= 4001 chunks of which the first <= 2048 do the waiting: the
// host takes the two-launch form beyond): a workgroup publishes its chunk total (bit 31 = "there") with an agent-scope
// store, adds up the totals of the chunks before it as they appear (polled by 256 lanes), and goes on as k_brick_offsets
// does.  The workgroup that finishes last clears the totals: `sums` is zero on entry and zero again on exit
// (iso_bricks_workspace_init clears it once).
constexpr int kScan1Max = 2048;
static_assert(kScan1Max == kScan1MaxWords, "bricks_carve sizes the zeroed words");
__global__ __launch_bounds__(256) void k_brick_offsets1(const BrickHdr* __restrict__ hp, int32_t* __restrict__ cnt,
                                                        int32_t* __restrict__ off, unsigned* __restrict__ sums,
                                                        int32_t* __restrict__ list, int32_t* __restrict__ counters) {
  __shared__ int lds[4], s_base, s_last;
  const int nb = hp->n_bricks;
  if ((int)blockIdx.x * BS_CHUNK > nb) return;
  if (blockIdx.x == 0) bk_box_clear(counters);      // every reader of the pending box (the count pass) is done
  const int n_active = nb / BS_CHUNK + 1;
  int c[BS_ITEMS][BK_CPB], t[BS_ITEMS];
  const int v = brick_counters_load<true>(cnt, blockIdx.x, nb, c, t);
  int occ = 0;
  bool listed[BS_ITEMS];
  brick_listed(*hp, blockIdx.x, t, listed);
#pragma unroll
  for (int k = 0; k < BS_ITEMS; ++k) occ += listed[k] ? 1 : 0;
  int tot, occ_tot;
  int ex = block_excl_scan_256(v, tot, lds);
  if (threadIdx.x == 0) __hip_atomic_store(&sums[blockIdx.x], 0x80000000u | (unsigned)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int before = 0;
  for (int i = threadIdx.x; i < (int)blockIdx.x; i += 256) {
    unsigned q;
    while (!((q = __hip_atomic_load(&sums[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 0x80000000u)) __builtin_amdgcn_s_sleep(1);
    before += (int)(q & 0x7fffffffu);
  }
  int base;
  block_excl_scan_256(before, base, lds);                       // base = total of the chunks before this one
  ex += base;
  int at = block_excl_scan_256(occ, occ_tot, lds);
  if (threadIdx.x == 0) s_base = occ_tot ? atomicAdd(&counters[0], occ_tot) : 0;
  __syncthreads();
  brick_offsets_store(off, list, blockIdx.x, nb, c, t, ex, at + s_base, listed);
  // everyone has read the totals it needs once it is here; the last one clears them
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned* arrive = reinterpret_cast<unsigned*>(counters) + kArriveAt;
    const unsigned prev = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = prev == (unsigned)n_active - 1u;
    if (s_last) __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (s_last)
    for (int i = threadIdx.x; i < n_active; i += 256) __hip_atomic_store(&sums[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The chunk scan of the splat front end (k_mask_chunk_scan in splat.hip: exclusive scan of every view's per-chunk counts,
// first_idx / num_points as the packed layout of _C.splat_points takes them, view totals) by ONE workgroup of 256 that
// rides in the count launch of the grid the front end needs anyway.  The counts come per 256-point tile (four tiles per
// chunk) from the launch that took the mask (follow.h).
__device__ void chunk_scan_job(const ChunkScanJob& j) {
  __shared__ int s_tot[8];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // a WAVE per view (no workgroup barrier inside the scan): a lane takes four consecutive chunks per step -- sixteen tile
  // counts, four 16-byte loads, all requested before the first is used -- and the wave scans 256 chunks per step
  for (int v = w; v < j.n_views; v += 4) {
    const int32_t* tc = j.tile_cnt + (int64_t)v * j.n_tiles;
    int32_t* row = j.chunk + (int64_t)v * j.n_chunks;
    int carry = 0;
    for (int c0 = 0; c0 < j.n_chunks; c0 += 256) {
      int val[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = c0 + 4 * lane + q;
        val[q] = 0;
        if (i < j.n_chunks) {
          if (4 * i + 3 < j.n_real) { const int4 c = *reinterpret_cast<const int4*>(tc + 4 * i); val[q] = (c.x + c.y) + (c.z + c.w); }
          else for (int k = 0; 4 * i + k < j.n_real; ++k) val[q] += tc[4 * i + k];
        }
      }
      const int mine = (val[0] + val[1]) + (val[2] + val[3]);
      int inc = mine;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
      int ex = carry + inc - mine;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = c0 + 4 * lane + q;
        if (i < j.n_chunks) row[i] = ex;
        ex += val[q];
      }
      carry += __shfl(inc, 63);
    }
    if (lane == 0) s_tot[v] = carry;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t run = 0;
    for (int v = 0; v < j.n_views; ++v) { j.first[v] = run; j.num[v] = s_tot[v]; j.view_total[v] = s_tot[v]; run += s_tot[v]; }
    for (int v = j.n_views; v < 8; ++v) j.view_total[v] = 0;
  }
}

// ---- staging of one brick + halo into LDS -------------------------------------------------------
// WITH_NRM: rec1 (normals) staged next to rec0 (resample); otherwise one record per candidate whose
// w is rec1's payload (the view mask) with bit 31 = "imported, not a query" (bandwidth kernel).
// SUB: the LDS sort runs on SUB x SUB x SUB sub-cells per fine cell (the global grid and everything outside the
// workgroup know fine cells only).  SUB = 2 (resample): a query walks a 4 x 4 x 4 window of HALF cells picked by the
// half of its own half cell it lies in -- every point within 0.75 fine cells of the query is inside (1.5 half cells to
// each face), 1.6 r wide at the 0.8 r cell against the 2.4 r of a 3 x 3 x 3 walk of whole cells: 44 % of the candidates.
// Runs of the brick-sorted arrays a workgroup stages: per (x slot, y slot, neighbour in z) the sub-bricks that touch the
// brick -- x slot 0..3 = (brick x - 1, sub x 1), (x, 0), (x, 1), (x + 1, 0); same in y; in z the whole sub-brick column of
// the brick itself, the upper sub-bricks of the one below, the lower ones of the one above.
constexpr int kStageRuns = 48;
template <bool WITH_NRM, int SUB = 1, int CAP = BK_CAP>
struct BrickStage {
  static constexpr int NL = 6 * SUB;                // local (sub-)cells per axis: the brick + one fine cell of halo
  static constexpr int NCELL = NL * NL * NL;
  static constexpr int NQRUN = 16 * SUB * SUB;      // z-runs of the brick's own (sub-)cells
  float4 rec0[CAP];
  int src[WITH_NRM ? CAP : 1];     // WITH_NRM: position of the staged record in the brick-sorted arrays (its rec1 is read
                                      // from there by the few that need it: staging it cost 16 KB of LDS = three workgroups per CU)
  int gid[WITH_NRM ? 1 : CAP];
  int cstart[NCELL + 4];   // [NCELL + 1] used: local (sub-)cell -> first staged slot
  int ccur[NCELL];
  int run_i0[kStageRuns];
  int run_pre[kStageRuns + 1];
  int qbeg[NQRUN];
  int qpre[NQRUN + 1];
};

// -DBK_DBG_PHASES (timing experiment, tools/diag/brick_phases.py): thread 0 of every workgroup adds the shader-clock time
// between consecutive marks (barrier waits included) to bk_phase[i]; iso_dbg_brick_phases() reads and clears them.
#ifdef BK_DBG_PHASES
__device__ unsigned long long bk_phase[16384 * 8];
__shared__ unsigned long long s_bk_t, s_bk_acc[8];
#define BK_PH(i) do { if (threadIdx.x == 0) { const unsigned long long t_ = clock64(); \
  if ((i) >= 0) s_bk_acc[(i) < 0 ? 0 : (i)] += t_ - s_bk_t; \
  else for (int q_ = 0; q_ < 8; ++q_) s_bk_acc[q_] = 0; \
  s_bk_t = t_; } } while (0)
#define BK_PH_FLUSH() do { if (threadIdx.x == 0 && blockIdx.x < 16384) for (int q_ = 0; q_ < 8; ++q_) \
  bk_phase[blockIdx.x * 8 + q_] += s_bk_acc[q_]; } while (0)
#else
#define BK_PH(i) do {} while (0)
#define BK_PH_FLUSH() do {} while (0)
#endif

struct BrickGeo { int bx, by, bz, ox, oy, oz; };

// All threads of the workgroup.  Returns the number of staged records, or -1 when they do not fit.
// The records of the 27 bricks are nine contiguous runs of the brick-sorted arrays; a thread takes
// every BK_THREADS-th record of their concatenation and requests all of its records before it uses
// the first (BK_RAW in flight: the staging is a chain of global-memory latencies otherwise).
constexpr int BK_RAW = 8;

// VS (bandwidth kernel only): bit stride of the view mask in the staged record -- 1: as stored; 8: view v at bit
// 8 v (up to four views), so that masked records add up to four 8-bit per-view counters in one register.
template <bool WITH_NRM, int VS = 1, int SUB = 1, int CAP = BK_CAP>
__device__ int stage_brick(BrickStage<WITH_NRM, SUB, CAP>& S, const BrickHdr& h, const int32_t* __restrict__ off,
                           const float4* __restrict__ rec0, const float4* __restrict__ rec1, int b, BrickGeo& g) {
  typedef BrickStage<WITH_NRM, SUB, CAP> St;
  constexpr int NL = St::NL, NCELL = St::NCELL;
  const int tid = threadIdx.x;
  const int nbx = h.nb[0], nby = h.nb[1], nbz = h.nb[2];
  g.bz = b % nbz; g.by = (b / nbz) % nby; g.bx = b / (nbz * nby);
  g.ox = 4 * g.bx - 1; g.oy = 4 * g.by - 1; g.oz = 4 * g.bz - 1;
  __syncthreads();                                   // the previous brick's readers are done
  BK_PH(6);
  for (int c = tid; c < NCELL; c += BK_THREADS) S.ccur[c] = 0;
  if (tid < kStageRuns) {
    // (rt: the thread index behind an opaque move -- the run's offsets below depend on the thread only, so the compiler
    // hoists them out of the caller's brick loop and then SPILLS them across it: five scratch registers in
    // k_brick_resample, eight in k_brick_h, reloaded once per brick; recomputing 48 lanes' worth per brick is free)
    int rt = tid;
    asm volatile("" : "+v"(rt));
    const int xs = rt / 12, ys = (rt / 3) % 4, dz = rt % 3 - 1;
    const int x = g.bx + (xs + 1) / 2 - 1, sx = xs == 0 ? 1 : (xs == 3 ? 0 : xs - 1);      // xs 0..3 -> (dx, sub x) = (-1,1) (0,0) (0,1) (+1,0)
    const int y = g.by + (ys + 1) / 2 - 1, sy = ys == 0 ? 1 : (ys == 3 ? 0 : ys - 1);
    const int z = g.bz + dz;
    int i0 = 0, len = 0;
    if (x >= 0 && x < nbx && y >= 0 && y < nby && z >= 0 && z < nbz) {
      const int c = BK_CPB * ((x * nby + y) * nbz + z) + (sx * 2 + sy) * 2;
      const int lo = dz < 0 ? 1 : 0, hi = dz > 0 ? 0 : 1;                // sub z: both of the own brick, the near one of a neighbour
      i0 = off[c + lo];
      len = off[c + hi + 1] - i0;
    }
    S.run_i0[tid] = i0;
    S.run_pre[tid + 1] = len;
  }
  __syncthreads();
  if (tid < 64) {                                    // exclusive prefix of the run lengths (one wave)
    const int len = tid < kStageRuns ? S.run_pre[tid + 1] : 0;
    const int inc = wave_incl_scan_i(len);
    if (tid < kStageRuns) S.run_pre[tid + 1] = inc;
    if (tid == 0) S.run_pre[0] = 0;
  }
  __syncthreads();
  BK_PH(0);
  const int total_raw = S.run_pre[kStageRuns];
  auto index_of = [&](int j) {                       // the run r with run_pre[r] <= j < run_pre[r + 1]
    int lo = 0, hi = kStageRuns;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (S.run_pre[mid] <= j) lo = mid; else hi = mid;
    }
    return S.run_i0[lo] + (j - S.run_pre[lo]);
  };
  auto cell_of = [&](const float4& p) {
    const int lx = bk_sub<SUB>(p.x, h.mn[0], h.inv_f, h.nf[0]) - SUB * g.ox;
    const int ly = bk_sub<SUB>(p.y, h.mn[1], h.inv_f, h.nf[1]) - SUB * g.oy;
    const int lz = bk_sub<SUB>(p.z, h.mn[2], h.inv_f, h.nf[2]) - SUB * g.oz;
    return ((unsigned)lx < (unsigned)NL && (unsigned)ly < (unsigned)NL && (unsigned)lz < (unsigned)NL) ? (lx * NL + ly) * NL + lz : -1;
  };
  auto store = [&](int pos, const float4& p, int at, float uw) {
    if (WITH_NRM) {
      S.rec0[pos] = p;
      S.src[pos] = at;
    } else {
      const int gid = __float_as_int(p.w);
      const bool own = gid >= h.id_base && gid < h.id_base + h.n_own;
      int m = __float_as_int(uw) & 0xff;
      if (VS == 8) m = (m & 1) | ((m & 2) << 7) | ((m & 4) << 14) | ((m & 8) << 21);
      m |= own ? 0 : (int)0x80000000;
      S.rec0[pos] = make_float4(p.x, p.y, p.z, __int_as_float(m));
      S.gid[pos] = gid;
    }
  };
  auto scan_cells = [&]() {
    if (tid < 64) {                                  // exclusive scan of the NCELL counters by one wave
      constexpr int PER = (NCELL + 63) / 64;
      int sum = 0;
      for (int k = 0; k < PER; ++k) { const int c = tid * PER + k; sum += c < NCELL ? S.ccur[c] : 0; }
      int ex = wave_incl_scan_i(sum) - sum;
      for (int k = 0; k < PER; ++k) {
        const int c = tid * PER + k;
        if (c < NCELL) { const int v = S.ccur[c]; S.cstart[c] = ex; S.ccur[c] = ex; ex += v; }
      }
      if (tid == 63) S.cstart[NCELL] = ex;
    }
  };
  if (total_raw <= BK_THREADS * BK_RAW) {
    int idx[BK_RAW], cell[BK_RAW];
    float4 p[BK_RAW];
    // slot k of this WAVE holds records only when live(k) -- a scalar test, so the empty slots (on a surface half of the
    // eight) cost a branch instead of their share of ~1000 predicated-off instructions per wave
    const int wbase = __builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u));
    auto live = [&](int k) { return wbase + k * BK_THREADS < total_raw; };
#pragma unroll
    for (int k = 0; k < BK_RAW; ++k) {
      idx[k] = -1; cell[k] = -1;
      p[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live(k)) {
        const int j = tid + k * BK_THREADS;
        idx[k] = j < total_raw ? index_of(j) : -1;
      }
    }
#pragma unroll
    for (int k = 0; k < BK_RAW; ++k)
      if (live(k)) p[k] = idx[k] >= 0 ? rec0[idx[k]] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < BK_RAW; ++k) {
      if (live(k)) {
        cell[k] = idx[k] >= 0 ? cell_of(p[k]) : -1;
        if (cell[k] >= 0) atomicAdd(&S.ccur[cell[k]], 1);
      }
    }
    __syncthreads();
    BK_PH(1);
    scan_cells();
    __syncthreads();
    BK_PH(2);
    if (S.cstart[NCELL] > CAP) return -1;
    const float* __restrict__ rec1w = reinterpret_cast<const float*>(rec1) + 3;     // the payload word of a second record
    float uw[BK_RAW];
#pragma unroll
    for (int k = 0; k < BK_RAW; ++k) {               // (bandwidth kernel) the payload words of the accepted candidates, all in flight
      uw[k] = 0.f;
      if (!WITH_NRM && live(k)) uw[k] = cell[k] >= 0 ? rec1w[4 * (int64_t)idx[k]] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < BK_RAW; ++k)
      if (live(k)) { if (cell[k] >= 0) store(atomicAdd(&S.ccur[cell[k]], 1), p[k], idx[k], uw[k]); }
  } else {                                           // very dense neighbourhood: two passes over global memory
    for (int j = tid; j < total_raw; j += BK_THREADS) {
      const int c = cell_of(rec0[index_of(j)]);
      if (c >= 0) atomicAdd(&S.ccur[c], 1);
    }
    __syncthreads();
    scan_cells();
    __syncthreads();
    if (S.cstart[NCELL] > CAP) return -1;
    for (int j = tid; j < total_raw; j += BK_THREADS) {
      const int i = index_of(j);
      const float4 p = rec0[i];
      const int c = cell_of(p);
      if (c >= 0) store(atomicAdd(&S.ccur[c], 1), p, i, WITH_NRM ? 0.f : rec1[i].w);
    }
  }
  constexpr int NQ = St::NQRUN, QA = 4 * SUB;        // the brick's own (sub-)cells: QA x QA contiguous z-runs of QA
  int qlen = 0;
  if (tid < NQ) {
    // (cstart is final since the scan; ccur is being advanced by the scatter above; qt: opaque as rt above)
    int qt = tid;
    asm volatile("" : "+v"(qt));
    const int c = ((SUB + qt / QA) * NL + (SUB + qt % QA)) * NL + SUB;
    S.qbeg[qt] = S.cstart[c];
    qlen = S.cstart[c + QA] - S.cstart[c];
  }
  if (tid < 64) {                                    // exclusive prefix of the run lengths (NQ <= 64: one wave)
    const int inc = wave_incl_scan_i(qlen);
    if (tid < NQ) S.qpre[tid + 1] = inc;
    if (tid == 0) S.qpre[0] = 0;
  }
  __syncthreads();
  BK_PH(3);
  return S.cstart[NCELL];
}

template <bool WITH_NRM, int SUB, int CAP>
__device__ __forceinline__ int query_slot(const BrickStage<WITH_NRM, SUB, CAP>& S, int t) {
  int lo = 0, hi = BrickStage<WITH_NRM, SUB, CAP>::NQRUN;         // the run r with qpre[r] <= t < qpre[r + 1]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (S.qpre[mid] <= t) lo = mid; else hi = mid;
  }
  return S.qbeg[lo] + (t - S.qpre[lo]);
}

// The 3x3x3 fine cells around local cell (lx,ly,lz) = nine contiguous slot ranges of the staged block, walked as ONE
// loop per lane: two candidates per trip (both LDS reads issued before either is used; v0 / v1 say which of them
// exist), and a lane that reaches the end of a range moves on to the next one in the same trip -- the bounds of the
// range after that are requested then and are there when they are needed.  The first form ran the nine ranges as nine
// loops: a wave then pays the LONGEST of its lanes' ranges nine times (measured: ~125 trips per wave for ~46 per
// lane); as one loop it pays the longest total.  An empty range costs its lane one idle trip.
template <bool WITH_NRM, int CAP, class Body>
__device__ __forceinline__ void walk_candidates(const BrickStage<WITH_NRM, 1, CAP>& S, int lx, int ly, int lz, Body&& body) {
  constexpr int NL = BrickStage<WITH_NRM, 1, CAP>::NL;
  int cell = ((lx - 1) * NL + (ly - 1)) * NL + (lz - 1);            // range r: cell + (r / 3) NL^2 + (r % 3) NL, three cells
  int i = S.cstart[cell], e = S.cstart[cell + 3];
  cell += NL;
  int ni = S.cstart[cell], ne = S.cstart[cell + 3];                 // range 1
  int run = 0, m = 1;                                               // m = (index of the prefetched range) % 3
  while (run < 9) {
    const bool v0 = i < e, v1 = i + 1 < e;
    const int i0 = v0 ? i : 0, i1 = v1 ? i + 1 : i0;
    float4 c0 = S.rec0[i0];
    float4 c1 = S.rec0[i1];
    // both records are consumed HERE as far as the compiler can tell: left to itself it sinks the second read into the
    // `v1` branch and its payload word into the distance test (three LDS round trips per trip, one after the other)
    asm volatile("" : "+v"(c0.x), "+v"(c0.y), "+v"(c0.z), "+v"(c0.w), "+v"(c1.x), "+v"(c1.y), "+v"(c1.z), "+v"(c1.w));
    body(c0, c1, v0, v1);
    i += 2;
    if (i >= e) {
      ++run;
      i = ni; e = ne;
      m = m == 2 ? 0 : m + 1;
      cell += m == 0 ? NL * NL - 2 * NL : NL;
      if (run < 8) { ni = S.cstart[cell]; ne = S.cstart[cell + 3]; }
    }
  }
}

// The same nine ranges NEAREST FIRST, with an early exit: the column of the query's own cell, then the columns on the side
// of the cell the query lies on (sx, sy = +-1), the far side last.  done() is asked at the end of every range; a lane that
// is done leaves the loop (the wave pays the longest of its lanes' walks).  The bandwidth kernel's counting pass only has
// to find SEVEN renderable neighbours within 0.01 per view -- in a dense cloud the first or second column holds them, so
// most of the ~9 ranges are never read (k_brick_h 163 -> see DESIGN.md 3.2).
template <bool WITH_NRM, int CAP, class Body, class Done>
__device__ __forceinline__ void walk_candidates_near_first(const BrickStage<WITH_NRM, 1, CAP>& S, int lx, int ly, int lz,
                                                           int sx, int sy, Body&& body, Done&& done) {
  constexpr int NL = BrickStage<WITH_NRM, 1, CAP>::NL;
  // step k visits column (a_k sx, b_k sy): a = 0 1 0 1 -1 0 -1 1 -1, b = 0 0 1 1 0 -1 1 -1 -1 (stored + 1, two bits each)
  constexpr unsigned KA = 1u | 2u << 2 | 1u << 4 | 2u << 6 | 0u << 8 | 1u << 10 | 0u << 12 | 2u << 14 | 0u << 16;
  constexpr unsigned KB = 1u | 1u << 2 | 2u << 4 | 2u << 6 | 1u << 8 | 0u << 10 | 2u << 12 | 0u << 14 | 0u << 16;
  const int base = (lx * NL + ly) * NL + (lz - 1);
  const int dxs = sx * NL * NL, dys = sy * NL;
  auto cell_at = [&](int k) {
    const int a = (int)((KA >> (2 * k)) & 3u) - 1, b = (int)((KB >> (2 * k)) & 3u) - 1;
    return base + a * dxs + b * dys;
  };
  int i = S.cstart[base], e = S.cstart[base + 3];
  int c = cell_at(1);
  int ni = S.cstart[c], ne = S.cstart[c + 3];
  int run = 0;
  while (run < 9) {
    const bool v0 = i < e, v1 = i + 1 < e;
    const int i0 = v0 ? i : 0, i1 = v1 ? i + 1 : i0;
    float4 c0 = S.rec0[i0];
    float4 c1 = S.rec0[i1];
    asm volatile("" : "+v"(c0.x), "+v"(c0.y), "+v"(c0.z), "+v"(c0.w), "+v"(c1.x), "+v"(c1.y), "+v"(c1.z), "+v"(c1.w));
    body(c0, c1, v0, v1);
    i += 2;
    if (i >= e) {
      ++run;
      if (done()) run = 9;
      i = ni; e = ne;
      if (run < 8) { c = cell_at(run + 1); ni = S.cstart[c]; ne = S.cstart[c + 3]; }
    }
  }
}

// SUB = 2: the NW x NW x NW window of half cells whose first cell is (wx, wy, wz) = NW^2 contiguous slot ranges of NW
// half cells each (NW = 4: the query's own window; NW = 6: the 3 x 3 x 3 fine cells around its cell); same
// two-candidates-per-trip protocol.
#ifndef BK_RESAMPLE_FLAT
#define BK_RESAMPLE_FLAT 1
#endif
template <int NW, bool WITH_NRM, class Body>
__device__ __forceinline__ void walk_window(const BrickStage<WITH_NRM, 2>& S, int wx, int wy, int wz, Body&& body) {
  constexpr int NL = BrickStage<WITH_NRM, 2>::NL;
  auto cell0 = [&](int c) { return ((wx + c / NW) * NL + (wy + c % NW)) * NL + wz; };
  int a = S.cstart[cell0(0)], e = S.cstart[cell0(0) + NW];
  for (int c = 0; c < NW * NW; ++c) {
    int na = 0, ne = 0;
    if (c < NW * NW - 1) { na = S.cstart[cell0(c + 1)]; ne = S.cstart[cell0(c + 1) + NW]; }
    for (int i = a; i < e; i += 2) {
      const bool two = i + 1 < e;
      const int i1 = two ? i + 1 : i;
      const float4 c0 = S.rec0[i];
      const float4 c1 = S.rec0[i1];
      body(c0, i, c1, i1, true, two);
    }
    a = na; e = ne;
  }
}

// The same window as ONE loop per lane (see walk_candidates): a wave pays the longest total of its lanes' NW^2 ranges
// instead of the longest range NW^2 times.  body(c0, i0, c1, i1, v0, v1): v0 / v1 say which of the two records exist.
template <int NW, bool WITH_NRM, class Body>
__device__ __forceinline__ void walk_window_flat(const BrickStage<WITH_NRM, 2>& S, int wx, int wy, int wz, Body&& body) {
  constexpr int NL = BrickStage<WITH_NRM, 2>::NL;
  int cell = (wx * NL + wy) * NL + wz;                              // range c: cell + (c / NW) NL^2 + (c % NW) NL, NW half cells
  int i = S.cstart[cell], e = S.cstart[cell + NW];
  cell += NL;
  int ni = S.cstart[cell], ne = S.cstart[cell + NW];                // range 1
  int run = 0, m = 1;                                               // m = (index of the prefetched range) % NW
  while (run < NW * NW) {
    const bool v0 = i < e, v1 = i + 1 < e;
    const int i0 = v0 ? i : 0, i1 = v1 ? i + 1 : i0;
    const float4 c0 = S.rec0[i0];
    const float4 c1 = S.rec0[i1];
    body(c0, i0, c1, i1, v0, v1);
    i += 2;
    if (i >= e) {
      ++run;
      i = ni; e = ne;
      m = m == NW - 1 ? 0 : m + 1;
      cell += m == 0 ? NL * NL - (NW - 1) * NL : NL;
      if (run < NW * NW - 1) { ni = S.cstart[cell]; ne = S.cstart[cell + NW]; }
    }
  }
}

// a brick whose neighbourhood does not fit LDS: its own points go to the tail kernel
__device__ void brick_to_tail(const BrickHdr& h, const int32_t* __restrict__ off, const float4* __restrict__ rec0, int b,
                              int32_t* __restrict__ tail, int32_t* __restrict__ counters, int tail_slot, int per_query) {
  for (int i = off[BK_CPB * b] + threadIdx.x; i < off[BK_CPB * (b + 1)]; i += BK_THREADS) {
    const int gid = __float_as_int(rec0[i].w);
    if (gid >= h.id_base && gid < h.id_base + h.n_own) {
      const int at = atomicAdd(&counters[tail_slot], per_query);
      for (int v = 0; v < per_query; ++v) tail[at + v] = per_query > 1 ? (gid - h.id_base) * 8 + v : gid - h.id_base;
    }
  }
  if (threadIdx.x == 0) atomicAdd(&counters[2], 1);
}

// ---- the repulsion of one point given its sorted neighbours (levelset_sampling.py:268-284) ------
struct Repulse {
  float px, py, pz, inv_sigma;
  float sw = 0.f, mx = 0.f, my = 0.f, mz = 0.f;
  // (ux, uy, uz): the neighbour's UNIT normal -- F.normalize of levelset_sampling.py:258 is applied once per point when
  // the records are written (bk_unit_normal in the scatter kernels) instead of once per (query, neighbour) pair here:
  // the same operations on the same operands, so the same bits, and three IEEE divisions + a square root fewer per pair
  __device__ __forceinline__ void add(float qx, float qy, float qz, float ux, float uy, float uz) {
    const float dx = px - qx, dy = py - qy, dz = pz - qz;
    const float d2 = (dx * dx + dy * dy) + dz * dz;
    const float w = expf(-d2 * inv_sigma);
    const float dn = (dx * ux + dy * uy) + dz * uz;
    const float tx = dx - dn * ux, ty = dy - dn * uy, tz = dz - dn * uz;
    sw += w;
    mx += w * tx; my += w * ty; mz += w * tz;
  }
  __device__ __forceinline__ void finish(float* __restrict__ o) const {
    const float dens = sw + 1.0f;
    const float den = iso_eps_denom(sw, 1.0e-17f);
    o[0] = px + dens * mx / den;
    o[1] = py + dens * my / den;
    o[2] = pz + dens * mz / den;
  }
};

__device__ __forceinline__ bool pair_lt(float d1, int i1, float d2, int i2) {
  return d1 < d2 || (d1 == d2 && i1 < i2);
}

// Which brick of the list a workgroup takes.  Workgroup i of a launch runs on XCD i % 8 (tools/probes/simd_map.hip prints
// it), and each XCD has its own L2: dealt out in list order, the eight neighbours of a brick -- whose records it stages,
// and whose queries write into the same cache lines of the row-ordered outputs -- sit in eight different L2s.  With
// BK_XCD_CHUNKS = 8 the list (x-major brick order: consecutive entries are neighbours along z, then y) is cut into eight
// contiguous chunks and XCD k walks chunk k.
#ifndef BK_XCD_CHUNKS
#define BK_XCD_CHUNKS 8
#endif
__device__ __forceinline__ int bk_xcd_span(int n_list) {
  return BK_XCD_CHUNKS <= 1 ? n_list : BK_XCD_CHUNKS * ((n_list + BK_XCD_CHUNKS - 1) / BK_XCD_CHUNKS);
}
__device__ __forceinline__ int bk_xcd_item(int lj, int n_list) {
  if (BK_XCD_CHUNKS <= 1) return lj;
  const int chunk = (n_list + BK_XCD_CHUNKS - 1) / BK_XCD_CHUNKS;
  return (lj % BK_XCD_CHUNKS) * chunk + lj / BK_XCD_CHUNKS;
}

// ---- fused resample step -------------------------------------------------------------------------
// K = knn_k + 1 (self included), M = list length (>= K + 1).
template <int M>
__global__ __launch_bounds__(BK_THREADS, 4) void k_brick_resample(
    const BrickHdr* __restrict__ hp, const int32_t* __restrict__ off, const int32_t* __restrict__ list,
    const float4* __restrict__ rec0, const float4* __restrict__ rec1, int K, float* __restrict__ out,
    int64_t* __restrict__ idx_out, float* __restrict__ d2_out, int32_t* __restrict__ tail,
    int32_t* __restrict__ counters) {
  __shared__ BrickStage<true, 2> S;
  BK_PH(-1);
  __shared__ int s_unc[BK_THREADS], s_nunc;
  const BrickHdr h = *hp;
  const int n_list = counters[0];
  for (int lj = blockIdx.x; lj < bk_xcd_span(n_list); lj += gridDim.x) {
    const int li = bk_xcd_item(lj, n_list);
    if (li >= n_list) continue;
    const int b = list[li];
    BrickGeo g;
    if (threadIdx.x == 0) s_nunc = 0;                  // (stage_brick starts with a barrier)
    const int C = stage_brick<true, 1, 2>(S, h, off, rec0, rec1, b, g);
    if (C < 0) { brick_to_tail(h, off, rec0, b, tail, counters, 1, 1); continue; }
#ifdef BK_DBG_NOQUERY          // timing experiment (tools/build_variant.sh): staging only
    const int nq = 0;
#else
    const int nq = S.qpre[BrickStage<true, 2>::NQRUN];
#endif
    // One query, start to finish.  WIDE = false: the 4 x 4 x 4 window of half cells picked by the half of its own half
    // cell the query lies in: two below and one above on the axes where it lies in the lower half, one below and two
    // above otherwise -- every point within (2 - m) half cells is inside, m = the largest distance of the query from
    // the middle plane of its half cell over the axes (0 <= m <= 1/2): 0.75 .. 1 fine cells.  WIDE = true: the 3 x 3 x 3
    // fine cells around its cell (6^3 half cells; everything within one fine cell), for the few queries the first
    // window cannot certify.  Returns false when the result could not be certified (nothing is written then).
    // The wide pass runs with a list two entries longer when M = K + 1: a query whose (K + 1)-th key ties with its K-th
    // fails the list test whatever the window, and would otherwise end in the tail kernel (a wave per query).
    auto one_query = [&](int pos, auto wide_tag) -> bool {
      constexpr bool WIDE = decltype(wide_tag)::value;
      constexpr int ML = (WIDE && M == 10) ? 12 : M;
      const float4 q = S.rec0[pos];
      const int gid = __float_as_int(q.w);
      int wx, wy, wz;
      float mfrac = 0.f;
      {
        auto win = [&](float pc, float mn, int nf, int o) {
          const float t2 = ((pc - mn) * h.inv_f) * 2.0f;
          int sc = (int)floorf(t2);
          sc = sc < 0 ? 0 : (sc >= 2 * nf ? 2 * nf - 1 : sc);
          if (WIDE) return 2 * (sc >> 1) - 2 * o - 2;                 // the fine cell below the query's
          const float fr = t2 - (float)sc;
          const bool low = fr < 0.5f;
          mfrac = fmaxf(mfrac, low ? fr : 1.0f - fr);
          return sc - 2 * o - (low ? 2 : 1);
        };
        wx = win(q.x, h.mn[0], h.nf[0], g.ox);
        wy = win(q.y, h.mn[1], h.nf[1], g.oy);
        wz = win(q.z, h.mn[2], h.nf[2], g.oz);
      }
      float gq2 = h.g2;                                               // (0.999 f)^2: the wide window
      if (!WIDE) { const float gq = (2.0f - mfrac) * 0.5f; gq2 = (gq * gq) * h.g2; }
      unsigned key[ML];
#pragma unroll
      for (int j = 0; j < ML; ++j) key[j] = 0xffffffffu;
      auto visit = [&](const float4& c0, int i0, const float4& c1, int i1, bool v0, bool v1) {
        const float da = bk_d2(q.x, q.y, q.z, c0.x, c0.y, c0.z);
        const float db = bk_d2(q.x, q.y, q.z, c1.x, c1.y, c1.z);
        const unsigned ka = v0 ? ((__float_as_uint(da) & ~1023u) | (unsigned)i0) : 0xffffffffu;
        const unsigned kb = v1 ? ((__float_as_uint(db) & ~1023u) | (unsigned)i1) : 0xffffffffu;
#pragma unroll
        for (int j = ML - 1; j >= 1; --j) key[j] = bk_med3u(key[j - 1], ka, key[j]);
        key[0] = min(key[0], ka);
#pragma unroll
        for (int j = ML - 1; j >= 1; --j) key[j] = bk_med3u(key[j - 1], kb, key[j]);
        key[0] = min(key[0], kb);
      };
      if (WIDE) walk_window<6>(S, wx, wy, wz, visit);
      else if (BK_RESAMPLE_FLAT) walk_window_flat<4>(S, wx, wy, wz, visit);
      else walk_window<4>(S, wx, wy, wz, visit);
      // exact (d2, id) order of the survivors
      float d[ML];
      int id[ML], ps[ML];
#pragma unroll
      for (int j = 0; j < ML; ++j) {
        const bool valid = key[j] != 0xffffffffu;
        ps[j] = valid ? (int)(key[j] & 1023u) : pos;
        const float4 c = S.rec0[ps[j]];
        const float d2 = bk_d2(q.x, q.y, q.z, c.x, c.y, c.z);
        const bool ok = valid && d2 < h.r2;
        d[j] = ok ? d2 : FLT_MAX;
        id[j] = ok ? __float_as_int(c.w) : 0x7fffffff;
      }
      bool swapped;
      do {
        swapped = false;
#pragma unroll
        for (int j = 0; j + 1 < ML; ++j) {
          if (pair_lt(d[j + 1], id[j + 1], d[j], id[j])) {
            const float td = d[j]; d[j] = d[j + 1]; d[j + 1] = td;
            const int ti = id[j]; id[j] = id[j + 1]; id[j + 1] = ti;
            const int tp = ps[j]; ps[j] = ps[j + 1]; ps[j + 1] = tp;
            swapped = true;
          }
        }
      } while (swapped);
      unsigned kK = 0xffffffffu;
      float dK = FLT_MAX;
#pragma unroll
      for (int j = 0; j < ML; ++j) if (j == K - 1) { kK = key[j]; dK = d[j]; }
      const bool cert_list = key[ML - 1] == 0xffffffffu || (key[ML - 1] >> 10) > (kK >> 10);
      const bool cert_geo = gq2 >= h.r2 || (dK < FLT_MAX && dK <= gq2);
      if (!(cert_list && cert_geo)) return false;
      const int row = gid - h.id_base;
      Repulse R;
      R.px = q.x; R.py = q.y; R.pz = q.z; R.inv_sigma = h.inv_sigma;
      // the normals of the K - 1 neighbours come from the brick-sorted array (four requests in flight)
#pragma unroll
      for (int j0 = 1; j0 < ML; j0 += 4) {
        float4 u[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int j = j0 + k;
          u[k] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (j < ML && j < K && d[j < ML ? j : 0] < FLT_MAX) u[k] = rec1[S.src[ps[j < ML ? j : 0]]];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int j = j0 + k;
          if (j < ML && j < K && d[j < ML ? j : 0] < FLT_MAX) {
            const float4 c = S.rec0[ps[j < ML ? j : 0]];
            R.add(c.x, c.y, c.z, u[k].x, u[k].y, u[k].z);
          }
        }
      }
      R.finish(out + (int64_t)row * 3);
      if (idx_out) {
#pragma unroll
        for (int j = 1; j < ML; ++j)
          if (j < K) {
            idx_out[(int64_t)row * (K - 1) + j - 1] = d[j] < FLT_MAX ? (int64_t)id[j] : (int64_t)-1;
            if (d2_out) d2_out[(int64_t)row * (K - 1) + j - 1] = d[j] < FLT_MAX ? d[j] : -1.0f;
          }
      }
      return true;
    };
    auto to_tail = [&](int pos) { tail[atomicAdd(&counters[1], 1)] = __float_as_int(S.rec0[pos].w) - h.id_base; };
    for (int t = threadIdx.x; t < nq; t += BK_THREADS) {
      const int pos = query_slot(S, t);
      const int gid = __float_as_int(S.rec0[pos].w);
      if (gid < h.id_base || gid >= h.id_base + h.n_own) continue;          // imported halo point
      if (!one_query(pos, std::false_type())) {
        // (~0.1 % of the queries of a uniform cloud) queued for the wide window: done by densely packed lanes below
        const int at = atomicAdd(&s_nunc, 1);
        if (at < BK_THREADS) s_unc[at] = pos; else to_tail(pos);
      }
    }
    BK_PH(4);
    __syncthreads();
    BK_PH(5);
    const int n_unc = min(s_nunc, BK_THREADS);
    for (int u = threadIdx.x; u < n_unc; u += BK_THREADS) {
      const int pos = s_unc[u];
      if (!one_query(pos, std::true_type())) to_tail(pos);                   // beyond one fine cell: rings of bricks
    }
    BK_PH(7);
  }
  BK_PH_FLUSH();
}

// ---- rings of bricks around a query (tail kernels: one wave per query) ---------------------------
// The record ranges of up to 64 columns of bricks are fetched by 64 lanes at once (a ring then costs one
// global-memory latency for its bounds instead of one per column); the records of those columns are one index
// space split over the lanes.  bound() (wave-uniform, asked once per 64 columns) is the squared distance beyond
// which a record cannot matter any more -- the radius, or the K-th distance found so far: bricks whose box lies
// farther than that from the query are dropped before their records are read (a stray point of the SIREN level
// set 0.15 off the surface would otherwise scan the whole shell of bricks that reaches the surface anywhere).
struct WalkGeo { float qx, qy, qz, mnx, mny, mnz, B; };
#ifndef BK_WALK_ITEMS
#define BK_WALK_ITEMS 8     // record loads a lane keeps in flight (4 until round 3: a stray query's few thousand records were a
#endif                      // chain of ~40 batch latencies)

// bk_walk_shell: the bricks at Chebyshev distance rin + 1 .. rho from the query's brick (rin = rho - 1: one ring;
// rin = -1: the whole cube).  A query that found too little in rings 0 and 1 -- a stray point -- takes all the remaining
// rings up to the radius as ONE shell: a ring is a chain of two dependent global latencies (brick bounds, records), five
// rings one after the other were the critical path of the bandwidth tail kernel (150 us in the SIREN cycle for a few
// thousand stray queries).
// part / nparts: the trips of a batch of columns are dealt out to nparts waves that walk the same shell (each of them
// fetches the bounds and forms the prefix sums itself); 0 / 1 = one wave does everything.
// Two bounds.  bound(), asked once per 64 columns, decides which runs enter the index space the trips are dealt from: with
// nparts > 1 it must return the SAME value in every cooperating wave (else the waves would number the records
// differently and visit some twice and others never).  local(), asked before every trip, is the wave's own, sharper
// knowledge (its lanes' K-th distance so far): runs whose box lies beyond it stay in the index space but are not loaded.
template <class Fetch, class Body, class Bound, class Local>
__device__ __forceinline__ void bk_walk_shell(int rin, int rho, int qbx, int qby, int qbz, int nbx, int nby, int nbz,
                                              const int32_t* __restrict__ off, int lane, const WalkGeo& G,
                                              Fetch&& fetch, Body&& body, Bound&& bound, Local&& local,
                                              int part = 0, int nparts = 1) {
  const int x0 = max(qbx - rho, 0), x1 = min(qbx + rho, nbx - 1);
  const int y0 = max(qby - rho, 0), y1 = min(qby + rho, nby - 1);
  if (x0 > x1 || y0 > y1) return;
  const int ny = y1 - y0 + 1, ncols = (x1 - x0 + 1) * ny;
  for (int c0 = 0; c0 < ncols; c0 += 64) {
    int s0 = 0, e0 = 0, s1 = 0, e1 = 0;                  // this lane's column: one z-run (edge) or two caps
    float dm0 = FLT_MAX, dm1 = FLT_MAX;                  // squared distance from the query to the boxes of the runs
    const int col = c0 + lane;
    const float far2 = bound();
    // squared distance from the query to bricks [b0, b1] of one axis, shortened by a margin that covers the
    // rounding of the brick assignment (so that "farther than far2" is never claimed wrongly)
    auto gap = [&](float q, float mn, int b0, int b1) {
      const float lo = mn + (float)b0 * G.B, hi = mn + (float)(b1 + 1) * G.B;
      return fmaxf(0.f, fmaxf(lo - q, q - hi) - 1e-3f * G.B);
    };
    if (col < ncols) {
      const int x = x0 + col / ny, y = y0 + col % ny;
      const bool edge = x < qbx - rin || x > qbx + rin || y < qby - rin || y > qby + rin;   // outside the inner square
      const int cb = (x * nby + y) * nbz;
      const float gx = gap(G.qx, G.mnx, x, x), gy = gap(G.qy, G.mny, y, y);
      const float gxy2 = gx * gx + gy * gy;
      auto box2 = [&](int za, int zb) { const float gz = gap(G.qz, G.mnz, za, zb); return gxy2 + gz * gz; };
      // the bricks of the column that a ball of far2 around the query can reach (same margin as gap(): never too few)
      const float dz = sqrtf(fmaxf(far2 - gxy2, 0.f)) + 1e-3f * G.B;
      const int zc0 = (int)fmaxf(floorf((G.qz - dz - G.mnz) / G.B), 0.f);
      const int zc1 = (int)fminf(floorf((G.qz + dz - G.mnz) / G.B), (float)(nbz - 1));
      if (edge) {
        const int za = max(max(qbz - rho, 0), zc0), zb = min(min(qbz + rho, nbz - 1), zc1);
        if (za <= zb && (dm0 = box2(za, zb)) <= far2) { s0 = off[BK_CPB * (cb + za)]; e0 = off[BK_CPB * (cb + zb + 1)]; }
      } else {                                            // two caps: below and above the inner cube
        const int a0 = max(max(qbz - rho, 0), zc0), a1 = min(qbz - rin - 1, zc1);
        const int b0 = max(qbz + rin + 1, zc0), b1 = min(min(qbz + rho, nbz - 1), zc1);
        if (a0 <= a1 && (dm0 = box2(a0, a1)) <= far2) { s0 = off[BK_CPB * (cb + a0)]; e0 = off[BK_CPB * (cb + a1 + 1)]; }
        if (b0 <= b1 && (dm1 = box2(b0, b1)) <= far2) { s1 = off[BK_CPB * (cb + b0)]; e1 = off[BK_CPB * (cb + b1 + 1)]; }
      }
    }
    // a lane finds the column of its item in the prefix sums of the column lengths
    const int len0 = e0 - s0, len = len0 + (e1 - s1);
    int inc = len;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    const int ex = inc - len, total = __shfl(inc, 63);
    constexpr int NU = BK_WALK_ITEMS;                        // items per lane and trip: their loads overlap
    for (int j0 = part * 64 * NU; j0 < total; j0 += nparts * 64 * NU) {
      const float near2 = local();
      const int live = (dm0 <= near2 ? 1 : 0) | (dm1 <= near2 ? 2 : 0);
      int at[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int j = j0 + u * 64 + lane;
        int c = 0;
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1) {         // the last column whose prefix is <= j holds item j
          const int cand = c + step;
          const int v = __shfl(ex, cand & 63);
          if (cand < 64 && v <= j) c = cand;
        }
        const int cs0 = __shfl(s0, c), cl0 = __shfl(len0, c), cs1 = __shfl(s1, c), cex = __shfl(ex, c);
        const int clive = __shfl(live, c);
        const int r = j - cex;
        at[u] = j < total ? (r < cl0 ? cs0 + r : cs1 + (r - cl0)) : -1;
        if (!(clive & (r < cl0 ? 1 : 2))) at[u] = -1;
      }
      decltype(fetch(0)) rec[NU];                            // loads first (no control flow in between), then the work
#pragma unroll
      for (int u = 0; u < NU; ++u) rec[u] = fetch(at[u] >= 0 ? at[u] : 0);
#pragma unroll
      for (int u = 0; u < NU; ++u)
        if (at[u] >= 0) body(at[u], rec[u]);
    }
  }
}

template <class Fetch, class Body, class Bound, class Local>
__device__ __forceinline__ void bk_walk_ring(int rho, int qbx, int qby, int qbz, int nbx, int nby, int nbz,
                                             const int32_t* __restrict__ off, int lane, const WalkGeo& G,
                                             Fetch&& fetch, Body&& body, Bound&& bound, Local&& local,
                                             int part = 0, int nparts = 1) {
  bk_walk_shell(rho - 1, rho, qbx, qby, qbz, nbx, nby, nbz, off, lane, G, fetch, body, bound, local, part, nparts);
}

template <int KMAX>
struct Top3 {          // K smallest (d, id) with the sorted position of the record alongside
  float d[KMAX];
  int id[KMAX], at[KMAX];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int j = 0; j < KMAX; ++j) { d[j] = FLT_MAX; id[j] = 0x7fffffff; at[j] = 0; }
  }
  __device__ __forceinline__ void push(float cd, int ci, int ca, int K) {
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      if (j < K && pair_lt(cd, ci, d[j], id[j])) {
        const float td = d[j]; const int ti = id[j], ta = at[j];
        d[j] = cd; id[j] = ci; at[j] = ca;
        cd = td; ci = ti; ca = ta;
      }
    }
  }
};

// K rounds of a wave-wide arg-min over the lanes' list heads.  emit(k, d, id, at) is called with
// wave-uniform arguments for k = 0..K-1 (d == FLT_MAX: no such neighbour); returns the K-th distance.
template <int KMAX, class Emit>
__device__ __forceinline__ float wave_merge3(const Top3<KMAX>& best, int K, Emit&& emit) {
  int head = 0;
  float kth = FLT_MAX;
  for (int k = 0; k < K; ++k) {
    float hd = FLT_MAX;
    int hi = 0x7fffffff, ha = 0;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) if (j == head) { hd = best.d[j]; hi = best.id[j]; ha = best.at[j]; }
    float md = hd;
    int mi = hi, ma = ha;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float od = __shfl_xor(md, o);
      const int oi = __shfl_xor(mi, o), oa = __shfl_xor(ma, o);
      if (pair_lt(od, oi, md, mi)) { md = od; mi = oi; ma = oa; }
    }
    if (hd == md && hi == mi && md < FLT_MAX) ++head;
    if (k == K - 1) kth = md;
    emit(k, md, mi, ma);
  }
  return kth;
}

template <int KMAX>
__global__ __launch_bounds__(64) void k_brick_resample_tail(
    const BrickHdr* __restrict__ hp, const int32_t* __restrict__ off, const float4* __restrict__ rec0,
    const float4* __restrict__ rec1, const float* __restrict__ pts, int K, float* __restrict__ out,
    int64_t* __restrict__ idx_out, float* __restrict__ d2_out, const int32_t* __restrict__ tail,
    const int32_t* __restrict__ counters) {
  const BrickHdr h = *hp;
  const int lane = threadIdx.x;
  const int count = counters[1];
  const float B = 4.0f * h.f;
  const int rho_max = max(h.nb[0], max(h.nb[1], h.nb[2]));
  for (int w = blockIdx.x; w < count; w += gridDim.x) {
    const int row = tail[w];
    const float qx = pts[(int64_t)row * 3], qy = pts[(int64_t)row * 3 + 1], qz = pts[(int64_t)row * 3 + 2];
    const int qbx = bk_fine(qx, h.mn[0], h.inv_f, h.nf[0]) >> 2;
    const int qby = bk_fine(qy, h.mn[1], h.inv_f, h.nf[1]) >> 2;
    const int qbz = bk_fine(qz, h.mn[2], h.inv_f, h.nf[2]) >> 2;
    Top3<KMAX> best;
    best.init();
    float wd = FLT_MAX;
    int wi = 0x7fffffff;
    int found = 0;
    const WalkGeo geo = {qx, qy, qz, h.mn[0], h.mn[1], h.mn[2], B};
    float kth_seen = FLT_MAX;                               // K-th distance of the wave at the last ring end
    // a lane that holds K records within wd proves that the K-th distance of the query is <= wd
    auto far2 = [&]() {
      float m = wd;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o));
      return fminf(h.r2, fminf(m, kth_seen));
    };
    for (int rho = 0; rho <= rho_max; ++rho) {
      bk_walk_ring(rho, qbx, qby, qbz, h.nb[0], h.nb[1], h.nb[2], off, lane, geo, [&](int i) { return rec0[i]; },
                   [&](int i, const float4& c) {
        const float d2 = bk_d2(qx, qy, qz, c.x, c.y, c.z);
        if (d2 < h.r2) {
          if (found < KMAX) ++found;
          const int ci = __float_as_int(c.w);
          if (pair_lt(d2, ci, wd, wi)) {
            best.push(d2, ci, i, K);
#pragma unroll
            for (int j = 0; j < KMAX; ++j) if (j == K - 1) { wd = best.d[j]; wi = best.id[j]; }
          }
        }
      }, far2, far2);
      if (rho >= 1) {
        const float gg = (float)rho * B * 0.999f;
        if (gg >= h.r) break;
        int tot = found;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
        if (tot >= K) {
          const float kth = wave_merge3<KMAX>(best, K, [](int, float, int, int) {});
          kth_seen = fminf(kth_seen, kth);
          if (kth < FLT_MAX && kth <= gg * gg) break;
        }
      }
    }
    Repulse R;
    R.px = qx; R.py = qy; R.pz = qz; R.inv_sigma = h.inv_sigma;
    const float kthf = wave_merge3<KMAX>(best, K, [&](int k, float md, int mi, int ma) {
      if (k == 0) return;                                  // column 0 is dropped (levelset_sampling.py:136-138)
      if (md < FLT_MAX) {
        const float4 c = rec0[ma];
        const float4 u = rec1[ma];
        R.add(c.x, c.y, c.z, u.x, u.y, u.z);
      }
      if (idx_out && lane == 0) {
        idx_out[(int64_t)row * (K - 1) + k - 1] = md < FLT_MAX ? (int64_t)mi : (int64_t)-1;
        if (d2_out) d2_out[(int64_t)row * (K - 1) + k - 1] = md < FLT_MAX ? md : -1.0f;
      }
    });
    if (lane == 0) {
      R.finish(out + (int64_t)row * 3);
      const float need = kthf < FLT_MAX ? sqrtf(kthf) : h.r;         // how far this query had to look
      if (qx - need < h.x_lo || qx + need >= h.x_hi) atomicAdd(const_cast<int32_t*>(&counters[6]), 1);
    }
  }
}

// ---- renderable flags of every point for up to 8 views (rasterizer.py:184-254) --------------------
__global__ __launch_bounds__(256) void k_view_mask(const float* __restrict__ pts, const float* __restrict__ nrm,
                                                   const float* __restrict__ views, int n_views, int64_t n,
                                                   float znear, float zfar, int backface,
                                                   int32_t* __restrict__ mask, int32_t* __restrict__ view_count) {
  __shared__ int s_cnt[8];
  if (threadIdx.x < 8) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  int local[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float x = pts[i * 3], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
    float nx = 0.f, ny = 0.f, nz = 0.f;
    if (backface) { nx = nrm[i * 3]; ny = nrm[i * 3 + 1]; nz = nrm[i * 3 + 2]; }
    int m = 0;
#pragma unroll
    for (int v = 0; v < 8; ++v) {
      if (v < n_views) {
        const float* V = views + v * 16;                    // row-vector convention: p_view = [p,1] @ V
        const float zv = ((x * V[2] + y * V[6]) + z * V[10]) + V[14];
        bool ok = (zv >= znear) && (zv <= zfar);
        if (backface) ok = ok && (((nx * V[2] + ny * V[6]) + nz * V[10]) < 0.f);
        if (ok) { m |= 1 << v; ++local[v]; }
      }
    }
    mask[i] = m;
  }
#pragma unroll
  for (int v = 0; v < 8; ++v) {
    if (v < n_views) {
      int c = local[v];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
      if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_cnt[v], c);
    }
  }
  __syncthreads();
  if (threadIdx.x < n_views && s_cnt[threadIdx.x]) atomicAdd(&view_count[threadIdx.x], s_cnt[threadIdx.x]);
}

// ---- fused splat bandwidth -----------------------------------------------------------------------
// h of a point in view v from the sorted 7 smallest d2 among the view's points within r (self
// included): the stand-alone path takes max(dists[1..6]) of a -1-padded row (iso_splat_vrk_h).
__device__ __forceinline__ float h_from_list(const float* d /*7 ascending, FLT_MAX = none*/, bool small_cloud) {
  float m = -FLT_MAX;
#pragma unroll
  for (int k = 1; k < 7; ++k) {
    const float v = small_cloud ? 1e-3f : (d[k] < FLT_MAX ? d[k] : -1.0f);
    m = fmaxf(m, v);
  }
  return fminf(fmaxf(0.5f * m, 5e-5f), 0.01f);
}

// (five waves per SIMD: 277 -> 257 us with a persistent grid; six with a workgroup per brick, see BK_H_WGS and h_grid();
// the resample kernel above is bound by its instruction count and gains nothing)
#ifndef BK_H_CAP
#define BK_H_CAP BK_CAP
#endif
#ifndef BK_H_WGS
#define BK_H_WGS 6     // workgroups per CU (5 / 6 / 7: 189 / 181 / 196 us with a workgroup per brick)
#endif
template <int NV>
__global__ __launch_bounds__(BK_THREADS, NV <= 4 ? BK_H_WGS : 4) void k_brick_h(
    const BrickHdr* __restrict__ hp, const int32_t* __restrict__ off, const int32_t* __restrict__ list,
    const float4* __restrict__ rec0, const float4* __restrict__ rec1, const int32_t* __restrict__ view_total,
    int n_views, float* __restrict__ h_out /*(n_views, n_own)*/, int32_t* __restrict__ tail,
    int32_t* __restrict__ counters) {
  __shared__ BrickStage<false, 1, BK_H_CAP> S;
  BK_PH(-1);
  const BrickHdr h = *hp;
  const int n_list = counters[0];
  bool small_cloud[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) small_cloud[v] = v < n_views && view_total[v] < 7;
  // 0.5 d2 <= 5e-5 (the lower clamp of h) <=> d2 <= kT: seven such points settle h without their order
  const float kT = 2.0f * 5e-5f;
  const float t2 = fminf(kT, h.r2 > 0.f ? __uint_as_float(__float_as_uint(h.r2) - 1u) : 0.f);   // and d2 < r2
  for (int lj = blockIdx.x; lj < bk_xcd_span(n_list); lj += gridDim.x) {
    const int li = bk_xcd_item(lj, n_list);
    if (li >= n_list) continue;
    const int b = list[li];
    BrickGeo g;
    constexpr int VS = NV <= 4 ? 8 : 1;                 // staged view masks: one byte per view when they fit a word
    const int C = stage_brick<false, VS, 1, BK_H_CAP>(S, h, off, rec0, rec1, b, g);
    if (C < 0) { brick_to_tail(h, off, rec0, b, tail, counters, 3, 8); continue; }
#ifdef BK_DBG_NOQUERY
    const int nq = 0;
#else
    const int nq = S.qpre[16];
#endif
    for (int t0 = 0; t0 < nq; t0 += BK_THREADS) {
      const int t = t0 + threadIdx.x;
      float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
      int qmask = 0, lx = 1, ly = 1, lz = 1, row = 0, sx = 1, sy = 1;
      if (t < nq) {
        const int pos = query_slot(S, t);
        q = S.rec0[pos];
        qmask = __float_as_int(q.w);
        if (qmask < 0) qmask = 0;                                    // imported halo point: not a query
        row = S.gid[pos] - h.id_base;
        const int fx = bk_fine(q.x, h.mn[0], h.inv_f, h.nf[0]), fy = bk_fine(q.y, h.mn[1], h.inv_f, h.nf[1]);
        lx = fx - g.ox;
        ly = fy - g.oy;
        lz = bk_fine(q.z, h.mn[2], h.inv_f, h.nf[2]) - g.oz;
        // the half of its cell the query lies in (only the ORDER of the walk depends on it)
        sx = ((q.x - h.mn[0]) * h.inv_f - (float)fx) >= 0.5f ? 1 : -1;
        sy = ((q.y - h.mn[1]) * h.inv_f - (float)fy) >= 0.5f ? 1 : -1;
      }
      // pass 1: renderable neighbours within kT per view, counted until every view of the query has seven (cnt[v] is
      // the full count only where it stays below seven -- all the rest of the kernel asks)
      int cnt[NV];
#pragma unroll
      for (int v = 0; v < NV; ++v) cnt[v] = 0;
      if (VS == 8 && C <= 255) {                                     // (workgroup-uniform) byte counters cannot overflow
        unsigned acc = 0;
        const unsigned need = ((unsigned)qmask & 0x01010101u) << 7;
        if (qmask)
          walk_candidates_near_first(S, lx, ly, lz, sx, sy, [&](const float4& c0, const float4& c1, bool v0, bool v1) {
            const float da = bk_d2(q.x, q.y, q.z, c0.x, c0.y, c0.z);
            const float db = bk_d2(q.x, q.y, q.z, c1.x, c1.y, c1.z);
            const unsigned ma = (v0 && da <= t2) ? (__float_as_uint(c0.w) & 0x01010101u) : 0u;
            const unsigned mb = (v1 && db <= t2) ? (__float_as_uint(c1.w) & 0x01010101u) : 0u;
            acc += ma + mb;
          }, [&]() {      // bit 7 of every byte: that counter is >= 7 (no carry between bytes: 0x7f + 0x79 < 0x100)
            const unsigned ge7 = (((acc & 0x7f7f7f7fu) + 0x79797979u) | acc) & 0x80808080u;
            return (ge7 & need) == need;
          });
#pragma unroll
        for (int v = 0; v < NV; ++v) cnt[v] = (int)((acc >> (8 * (v & 3))) & 0xffu);
      } else if (qmask)
        walk_candidates_near_first(S, lx, ly, lz, sx, sy, [&](const float4& c0, const float4& c1, bool v0, bool v1) {
          const float da = bk_d2(q.x, q.y, q.z, c0.x, c0.y, c0.z);
          const float db = bk_d2(q.x, q.y, q.z, c1.x, c1.y, c1.z);
          const int ma = (v0 && da <= t2) ? __float_as_int(c0.w) : 0;
          const int mb = (v1 && db <= t2) ? __float_as_int(c1.w) : 0;
#pragma unroll
          for (int v = 0; v < NV; ++v) cnt[v] += ((ma >> (VS * v)) & 1) + ((mb >> (VS * v)) & 1);
        }, [&]() {
          bool all = true;
#pragma unroll
          for (int v = 0; v < NV; ++v) all = all && (!((qmask >> (VS * v)) & 1) || cnt[v] >= 7);
          return all;
        });
      BK_PH(4);
      bool open_ = false;
#pragma unroll
      for (int v = 0; v < NV; ++v) open_ = open_ || (((qmask >> (VS * v)) & 1) && !small_cloud[v] && cnt[v] < 7);
      // pass 2 (only the lanes with an open view; rare away from the terminator of a dense cloud):
      // the 7 smallest d2 per view
      float d[NV][7];
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int j = 0; j < 7; ++j) d[v][j] = FLT_MAX;
      if (open_)
        walk_candidates(S, lx, ly, lz, [&](const float4& c0, const float4& c1, bool v0, bool v1) {
          float da = bk_d2(q.x, q.y, q.z, c0.x, c0.y, c0.z);
          float db = bk_d2(q.x, q.y, q.z, c1.x, c1.y, c1.z);
          da = (v0 && da < h.r2) ? da : FLT_MAX;
          db = (v1 && db < h.r2) ? db : FLT_MAX;
          const int ma = __float_as_int(c0.w), mb = __float_as_int(c1.w);
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            const float ka = (ma >> (VS * v)) & 1 ? da : FLT_MAX;
            const float kb = (mb >> (VS * v)) & 1 ? db : FLT_MAX;
#pragma unroll
            for (int j = 6; j >= 1; --j) d[v][j] = __builtin_amdgcn_fmed3f(d[v][j - 1], ka, d[v][j]);
            d[v][0] = fminf(d[v][0], ka);
#pragma unroll
            for (int j = 6; j >= 1; --j) d[v][j] = __builtin_amdgcn_fmed3f(d[v][j - 1], kb, d[v][j]);
            d[v][0] = fminf(d[v][0], kb);
          }
        });
      BK_PH(5);
      if (qmask) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          if (v < n_views && ((qmask >> (VS * v)) & 1)) {
            float hv;
            bool cert = true;
            if (small_cloud[v]) hv = fminf(fmaxf(0.5f * 1e-3f, 5e-5f), 0.01f);
            else if (cnt[v] >= 7) hv = 5e-5f;
            else {
              cert = h.g_covers_r || (d[v][6] < FLT_MAX && d[v][6] <= h.g2);
              hv = h_from_list(d[v], false);
            }
            if (cert) h_out[(int64_t)v * h.n_own + row] = hv;
            else tail[atomicAdd(&counters[3], 1)] = row * 8 + v;
          }
        }
      }
    }
  }
  BK_PH_FLUSH();
}

template <int KMAX>
struct TopF {
  float d[KMAX];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int j = 0; j < KMAX; ++j) d[j] = FLT_MAX;
  }
  __device__ __forceinline__ void push(float k) {
#pragma unroll
    for (int j = KMAX - 1; j >= 1; --j) d[j] = __builtin_amdgcn_fmed3f(d[j - 1], k, d[j]);
    d[0] = fminf(d[0], k);
  }
};

// the 7 smallest values of the union of the lanes' sorted lists, wave-uniform
__device__ __forceinline__ void wave_merge_f7(const TopF<7>& best, float* out7) {
  int head = 0;
  for (int k = 0; k < 7; ++k) {
    float hd = FLT_MAX;
#pragma unroll
    for (int j = 0; j < 7; ++j) if (j == head) hd = best.d[j];
    float md = hd;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) md = fminf(md, __shfl_xor(md, o));
    // exactly one of the lanes holding the minimum pops (the lowest lane)
    const unsigned long long holders = __ballot(hd == md && md < FLT_MAX);
    if (holders && (int)(__ffsll((long long)holders) - 1) == (int)(threadIdx.x & 63)) ++head;
    out7[k] = md;
  }
}

// What the clamp of h_from_list makes of the search: 0.5 d2 >= 0.01 for every d2 >= kHSat, so a list that holds ANY
// distance in [kHSat, r2) among its entries 1..6 gives h = 0.01 whatever the others are.  The tail search therefore keeps
// exact distances only below kHSat and a flag "some renderable point lies in [kHSat, r2)": once the flag is up nothing
// beyond sqrt(kHSat) = 0.141 can change the result (r = 0.2 in the reference's settings: half the shell's volume).
// (0.5f * 0.02f == 0.01f exactly: the two constants share their mantissa.)
constexpr float kHSat = 0.02f;

// One WORKGROUP of four waves per uncertified (query, view): a stray point of the SIREN level set finds nothing nearby and
// scans the whole shell of bricks up to the radius -- ~10 k records, a chain of ~20 batches of loads for one wave, and the
// longest such chain is what the launch takes (120 us in the SIREN cycle however many workgroups share the few thousand
// entries).  The four waves deal the batches out among themselves and pool their seven smallest distances through LDS at
// the ring ends.
#ifndef BK_TAIL_WAVES
#define BK_TAIL_WAVES 4
#endif
// (kTailWaves <= 9: the pooled selection runs in one wave.  Eight waves instead of four: a rank's share of eight 51 -> 40 us,
// the 1 M-point cycle 72-75 -> 83-86 -- the launcher takes eight for clouds below kTailWideBelow points.)
constexpr int64_t kTailWideBelow = 300000;
template <int kTailWaves>
__global__ __launch_bounds__(64 * kTailWaves) void k_brick_h_tail(
    const BrickHdr* __restrict__ hp, const int32_t* __restrict__ off, const float4* __restrict__ rec0,
    const float4* __restrict__ rec1, const float* __restrict__ pts, const int32_t* __restrict__ mask,
    const int32_t* __restrict__ view_total, int n_views, float* __restrict__ h_out,
    const int32_t* __restrict__ tail, const int32_t* __restrict__ counters) {
  __shared__ float s_m7[2][kTailWaves][8];     // [.][.][7]: the wave saw a distance in [kHSat, r2)
  const BrickHdr h = *hp;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int count = counters[3];
  const float B = 4.0f * h.f;
  const int rho_max = max(h.nb[0], max(h.nb[1], h.nb[2]));
  int pool_buf = 0;
  for (int w = blockIdx.x; w < count; w += gridDim.x) {
    const int row = tail[w] >> 3, v = tail[w] & 7;
    if (v >= n_views || !((mask[row] >> v) & 1)) continue;           // (a whole overflowing brick was queued)
    const float qx = pts[(int64_t)row * 3], qy = pts[(int64_t)row * 3 + 1], qz = pts[(int64_t)row * 3 + 2];
    const int qbx = bk_fine(qx, h.mn[0], h.inv_f, h.nf[0]) >> 2;
    const int qby = bk_fine(qy, h.mn[1], h.inv_f, h.nf[1]) >> 2;
    const int qbz = bk_fine(qz, h.mn[2], h.inv_f, h.nf[2]) >> 2;
    const bool small_cloud = view_total[v] < 7;
    TopF<7> best;
    best.init();
    float m7[7];
    const WalkGeo geo = {qx, qy, qz, h.mn[0], h.mn[1], h.mn[2], B};
    float kth_seen = FLT_MAX;
    bool sat = false;                                       // this lane met a distance in [kHSat, r2)
    bool sat_any = false;                                   // ... some lane of the workgroup did (as far as this wave knows)
    bool sat_all = false;                                   // sat_any as of the last pool(): the same in every wave
    auto far2 = [&]() {                                     // a lane's own 7th distance bounds the query's (this wave's view)
      float m = best.d[6];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o));
      sat_any = sat_any || __any(sat);
      return fminf(sat_any ? kHSat : h.r2, fminf(m, kth_seen));
    };
    auto far2_all = [&]() { return fminf(sat_all ? kHSat : h.r2, kth_seen); };    // what all four waves agree on
    // the seven smallest distances of the workgroup: every wave's seven through LDS, then one more selection
    auto pool = [&]() {
      wave_merge_f7(best, m7);
      sat_any = sat_any || __any(sat);
      if (lane < 8) {
        float mine = m7[0];
#pragma unroll
        for (int k = 1; k < 7; ++k) mine = lane == k ? m7[k] : mine;
        s_m7[pool_buf][wave][lane] = lane == 7 ? (sat_any ? 1.f : 0.f) : mine;
      }
      __syncthreads();
      TopF<7> all;
      all.init();
      if (lane < 7 * kTailWaves) all.push(s_m7[pool_buf][lane / 7][lane % 7]);
      wave_merge_f7(all, m7);
#pragma unroll
      for (int k = 0; k < kTailWaves; ++k) sat_any = sat_any || s_m7[pool_buf][k][7] != 0.f;
      sat_all = sat_any;
      pool_buf ^= 1;                                        // (the next pool writes the other buffer: no second barrier)
    };
    auto fetch = [&](int i) { float4 c = rec0[i]; c.w = rec1[i].w; return c; };          // position + view mask
    auto visit = [&](int, const float4& c) {
      if (!((__float_as_int(c.w) >> v) & 1)) return;
      const float d2 = bk_d2(qx, qy, qz, c.x, c.y, c.z);
      if (d2 < kHSat) { if (d2 < h.r2) best.push(d2); }
      else if (d2 < h.r2) sat = true;
    };
    // rings 0 and 1 one after the other (most queries end there); a query that is still open -- a stray point -- takes
    // every remaining ring up to the radius as one shell (bk_walk_shell)
    int rho_r = 1;                                          // first ring whose guarantee rho * B covers the radius
    while (rho_r < rho_max && (float)rho_r * B * 0.999f < h.r) ++rho_r;
    for (int rho = 0; rho <= min(1, rho_max); ++rho) {
      bk_walk_ring(rho, qbx, qby, qbz, h.nb[0], h.nb[1], h.nb[2], off, lane, geo, fetch, visit, far2_all, far2, wave, kTailWaves);
      if (rho >= 1) {
        const float gg = (float)rho * B * 0.999f;
        if (gg >= h.r) { rho_r = 1; break; }
        pool();
        kth_seen = fminf(kth_seen, m7[6]);
        if (m7[6] < FLT_MAX && m7[6] <= gg * gg) { rho_r = 1; break; }
        if (sat_any && kHSat <= gg * gg) { rho_r = 1; break; }          // everything below kHSat was seen
      }
    }
    if (rho_r > 1)
      bk_walk_shell(1, rho_r, qbx, qby, qbz, h.nb[0], h.nb[1], h.nb[2], off, lane, geo, fetch, visit, far2_all, far2, wave, kTailWaves);
    pool();
    if (threadIdx.x == 0) {
      // fewer than seven below kHSat and something in [kHSat, r2): entry 1..6 of the full list holds a value >= kHSat
      const bool saturated = m7[6] == FLT_MAX && sat_any && !small_cloud;
      h_out[(int64_t)v * h.n_own + row] = saturated ? 0.01f : h_from_list(m7, small_cloud);
      const float need = m7[6] < FLT_MAX ? sqrtf(m7[6]) : (sat_any ? fminf(sqrtf(kHSat), h.r) : h.r);
      if (!small_cloud && (qx - need < h.x_lo || qx + need >= h.x_hi)) atomicAdd(const_cast<int32_t*>(&counters[6]), 1);
    }
  }
}

// ---- halo exchange between x-slabs (one rank per GPU) ---------------------------------------------
// Every rank's LOCAL bounding box is known to all (ranges: (world, 8) floats, iso_points_bbox layout).
// Rank k's queries live in fine x-cells [fx(min_k), fx(max_k)]; their 3x3x3 neighbourhoods reach one
// fine cell further, so rank j exports exactly its points whose fine x-cell lies in that widened range
// of some other rank, and rank k imports, from what all ranks exported, those in its own widened range.
// Buffers: float4[2][cap + 1]; word 0 of record 0 is the record count (int bits).
__device__ __forceinline__ void wave_append(bool take, int32_t* counter, int cap, int& slot) {
  const int lane = threadIdx.x & 63;
  const unsigned long long bal = __ballot(take);
  int base = 0;
  if (lane == 0 && bal) base = atomicAdd(counter, __popcll(bal));
  base = __shfl(base, 0);
  slot = take ? base + __popcll(bal & ((1ull << lane) - 1ull)) : -1;
  if (slot >= cap) slot = -1;
}

__global__ __launch_bounds__(256) void k_halo_export(const float* __restrict__ pts, const float* __restrict__ nrm,
                                                     const int32_t* __restrict__ payload, int64_t n_own,
                                                     const BrickHdr* __restrict__ hp, const float* __restrict__ ranges,
                                                     int world, int rank, int halo, float4* __restrict__ out, int cap) {
  const BrickHdr h = *hp;
  __shared__ int s_app[17];
  int32_t* counter = reinterpret_cast<int32_t*>(out);
  const int64_t span = (int64_t)gridDim.x * 1024;
  for (int64_t i0 = (int64_t)blockIdx.x * 1024; i0 < n_own; i0 += span) {
    bool take[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t i = i0 + q * 256 + threadIdx.x;
      take[q] = false;
      if (i < n_own) {
        const int fx = bk_fine(pts[i * 3], h.mn[0], h.inv_f, h.nf[0]);
        for (int k = 0; k < world; ++k) {
          if (k == rank) continue;
          const int lo = bk_fine(ranges[k * 8], h.mn[0], h.inv_f, h.nf[0]) - halo;
          const int hi = bk_fine(ranges[k * 8 + 4], h.mn[0], h.inv_f, h.nf[0]) + halo;
          take[q] = take[q] || (fx >= lo && fx <= hi);
        }
      }
    }
    int slot[4];
    iso_block_append4(take, counter, s_app, slot);      // (the count may pass cap: the importer clamps and reports it)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (slot[q] >= 0 && slot[q] < cap) {
        const int64_t i = i0 + q * 256 + threadIdx.x;
        out[1 + slot[q]] = make_float4(pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2], __int_as_float(h.id_base + (int)i));
        float4 u = make_float4(0.f, 0.f, 0.f, __int_as_float(payload ? payload[i] : 0));
        if (nrm) { u.x = nrm[i * 3]; u.y = nrm[i * 3 + 1]; u.z = nrm[i * 3 + 2]; }
        out[(int64_t)cap + 1 + 1 + slot[q]] = u;
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_halo_import(const float4* __restrict__ gathered, int world, int rank, int cap,
                                                     int halo, BrickHdr* __restrict__ hp, const float* __restrict__ ranges,
                                                     float4* __restrict__ imp0, float4* __restrict__ imp1,
                                                     int32_t* __restrict__ imp_count, int imp_cap,
                                                     int32_t* __restrict__ counters) {
  const int src = blockIdx.y;
  const BrickHdr h = *hp;
  const int lo = bk_fine(ranges[rank * 8], h.mn[0], h.inv_f, h.nf[0]) - halo;
  const int hi = bk_fine(ranges[rank * 8 + 4], h.mn[0], h.inv_f, h.nf[0]) + halo;
  if (src == rank) {
    // everything the cloud holds in fine x-cells [lo, hi] is on this rank now: the tail kernels check their
    // search balls against this range (a query that needs more is counted, IsoCycle.check reports it)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      hp->x_lo = lo <= 0 ? -FLT_MAX : h.mn[0] + (float)lo * h.f;
      hp->x_hi = hi >= h.nf[0] - 1 ? FLT_MAX : h.mn[0] + (float)(hi + 1) * h.f;
      // the work list of the fused kernels (k_brick_offsets*) skips the bricks outside the rank's own columns: their
      // records are candidates only, and a workgroup that staged one found no query (half the list at N = 8)
      hp->own_bx_lo = (lo + halo) >> 2;
      hp->own_bx_hi = (hi - halo) >> 2;
    }
    return;
  }
  const float4* blk = gathered + (int64_t)src * 2 * (cap + 1);
  int cnt = __float_as_int(blk[0].x);
  if (cnt > cap) { cnt = cap; if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(&counters[4], 1); }   // the exporter overflowed
  __shared__ int s_app[17];
  const int span = gridDim.x * 1024;
  for (int j0 = blockIdx.x * 1024; j0 < cnt; j0 += span) {
    bool take[4];
    float4 p[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = j0 + q * 256 + threadIdx.x;
      take[q] = false;
      p[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < cnt) {
        p[q] = blk[1 + j];
        const int fx = bk_fine(p[q].x, h.mn[0], h.inv_f, h.nf[0]);
        take[q] = fx >= lo && fx <= hi;
      }
    }
    int slot[4];
    iso_block_append4(take, imp_count, s_app, slot);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (slot[q] < 0) continue;
      if (slot[q] < imp_cap) { imp0[slot[q]] = p[q]; imp1[slot[q]] = blk[(int64_t)cap + 1 + 1 + j0 + q * 256 + threadIdx.x]; }
      else atomicAdd(&counters[5], 1);                                                                // import overflow
    }
  }
}

}  // namespace

// ================================================================================================
extern "C" int64_t iso_bricks_workspace_bytes(int64_t n_max) {
  return bricks_carve(nullptr, n_max).bytes;
}

extern "C" int iso_bricks_workspace_init(void* workspace, int64_t n_max, void* stream) {
  ISO_REQUIRE(workspace && n_max >= 0, ISO_ERR_INVALID, "iso_bricks_workspace_init: bad arguments");
  ISO_REQUIRE(((uintptr_t)workspace & 255) == 0, ISO_ERR_INVALID, "iso_bricks_workspace_init: workspace must be 256-B aligned");
  const BrickWs w = bricks_carve(workspace, n_max);
  hipLaunchKernelGGL(k_bricks_init, dim3(iso_stream_grid(w.G, 256)), dim3(256), 0, (hipStream_t)stream, w.counters, w.cnt, w.G, w.scan1);
  ISO_CHECK_LAUNCH("iso_bricks_workspace_init");
  return ISO_OK;
}

// Diagnostic (synchronises the stream): did iso_bricks_workspace_init ever run on this workspace?  A build on a workspace
// that was never initialised counts into whatever the memory held and produces a wrong grid without any other sign.
extern "C" int iso_bricks_workspace_check(const void* workspace, int64_t n_max, void* stream) {
  ISO_REQUIRE(workspace && n_max >= 0, ISO_ERR_INVALID, "iso_bricks_workspace_check: bad arguments");
  ISO_REQUIRE(((uintptr_t)workspace & 255) == 0, ISO_ERR_INVALID, "iso_bricks_workspace_check: workspace must be 256-B aligned");
  const BrickWs w = bricks_carve(const_cast<void*>(workspace), n_max);
  // the counter block up to the pending box + the chunk totals of the one-launch scan (both at fixed offsets)
  int32_t host[kBoxAt + kScan1MaxWords];                  // per call (a static one raced between host threads: ADVICE r5)
  ISO_REQUIRE(hipMemcpyAsync(host, w.counters, 4 * kBoxAt, hipMemcpyDeviceToHost, (hipStream_t)stream) == hipSuccess &&
                  hipMemcpyAsync(host + kBoxAt, w.scan1, 4 * kScan1MaxWords, hipMemcpyDeviceToHost, (hipStream_t)stream) == hipSuccess &&
                  hipStreamSynchronize((hipStream_t)stream) == hipSuccess,
              ISO_ERR_LAUNCH, "iso_bricks_workspace_check: could not read the workspace");
  ISO_REQUIRE(host[kMagicAt] == kBrickMagic, ISO_ERR_INVALID,
              "iso_bricks_workspace_check: the workspace was never initialised (iso_bricks_workspace_init)");
  // what the builds rely on finding zero (an aborted launch, or a build on dirty memory, leaves them set and every later
  // build on the workspace then computes wrong offsets): the arrival words and the scan's chunk totals
  for (int i = kArriveAt; i < kArriveAt + 17; ++i)
    ISO_REQUIRE(host[i] == 0, ISO_ERR_INVALID, "iso_bricks_workspace_check: arrival word %d is not zero (an aborted build?): "
                "re-initialise the workspace", i - kArriveAt);
  for (int i = 0; i < kScan1MaxWords; ++i)
    ISO_REQUIRE(host[kBoxAt + i] == 0, ISO_ERR_INVALID, "iso_bricks_workspace_check: chunk total %d of the brick scan is not zero "
                "(an aborted build?): re-initialise the workspace", i);
  return ISO_OK;
}

// count -> offsets (+ work list, counters left zeroed) -> scatter, on a header that is already written
static int bricks_fill(const BrickWs& w, const float* points, const float* normals, const int32_t* payload, int64_t n_own,
                       const float* import_rec0, const float* import_rec1, const int32_t* import_count,
                       int64_t import_max, hipStream_t s, bool pending = false, BrickParams q = BrickParams{0, 0, 0, 0.f, 0, 0.f, 0},
                       ChunkScanJob job = ChunkScanJob{nullptr, nullptr, 0, 0, 0, 0, nullptr, nullptr, nullptr}) {
  // imported records ride in the launches of the own points when the header is there already (the N-rank path: it is
  // written before the halo exchange)
  const bool ride = import_max > 0 && n_own > 0 && !pending && !job.chunk;
  const ImportJob none{nullptr, nullptr, nullptr, 0, 0};
  const ImportJob imp_c{(const float4*)import_rec0, (const float4*)import_rec1, import_count, import_max,
                        ride ? iso_stream_grid(import_max, 1024) : 0};
  const ImportJob imp_s{(const float4*)import_rec0, (const float4*)import_rec1, import_count, import_max,
                        ride ? iso_stream_grid(import_max, 256) : 0};
  if (n_own > 0 || pending || job.chunk)        // (a pending header is written by this pass, whatever the cloud holds)
    hipLaunchKernelGGL(k_brick_count, dim3(iso_stream_grid(n_own, 1024) + (job.chunk ? 1 : 0) + imp_c.blocks), dim3(256), 0, s,
                       points, n_own, w.hdr, w.cnt, w.slot, pending ? 1 : 0, q, w.counters, job, ride ? imp_c : none);
  if (import_max > 0 && !ride)
    hipLaunchKernelGGL(k_brick_count_recs, dim3(iso_stream_grid(import_max, 1024)), dim3(256), 0, s,
                       (const float4*)import_rec0, import_count, import_max, w.hdr, w.cnt, w.slot);
  const int chunks = (int)(((w.G - 1) / BK_CPB + 1 + BS_CHUNK - 1) / BS_CHUNK);      // chunks of BS_CHUNK bricks (+ the sentinel)
  ISO_REQUIRE(w.scan_ws_bytes >= (int64_t)(chunks + kScan1Max) * 4, ISO_ERR_WORKSPACE, "iso_bricks_build: scan workspace too small");
  if (chunks <= kScan1Max) {            // one launch (the totals are zero on entry and left zero)
    hipLaunchKernelGGL(k_brick_offsets1, dim3(chunks), dim3(256), 0, s, w.hdr, w.cnt, w.off, w.scan1, w.list,
                       w.counters);
  } else {
    int32_t* sums2 = (int32_t*)w.scan_ws + kScan1Max;       // (behind the words the one-launch form keeps zeroed)
    hipLaunchKernelGGL(k_brick_sums, dim3(chunks), dim3(256), 0, s, w.hdr, w.cnt, sums2);
    hipLaunchKernelGGL(k_brick_offsets, dim3(chunks), dim3(256), 0, s, w.hdr, w.cnt, w.off, (const int32_t*)sums2, w.list,
                       w.counters);
  }
  if (n_own > 0)
    hipLaunchKernelGGL(k_brick_scatter, dim3(iso_stream_grid(n_own, 256) + imp_s.blocks), dim3(256), 0, s, points, normals, payload,
                       n_own, w.hdr, w.off, w.slot, w.rec0, w.rec1, ride ? imp_s : none);
  if (import_max > 0 && !ride)
    hipLaunchKernelGGL(k_brick_scatter_recs, dim3(iso_stream_grid(import_max, 256)), dim3(256), 0, s,
                       (const float4*)import_rec0, (const float4*)import_rec1, w.hdr, w.off, w.slot, w.rec0, w.rec1);
  return ISO_OK;
}

extern "C" int iso_bricks_build(const float* points, const float* normals, const int32_t* payload,
                                int64_t n_own, int64_t id_base, const float* import_rec0,
                                const float* import_rec1, const int32_t* import_count, int64_t import_max,
                                const float* bbox, int64_t n_total, float radius, int knn_k,
                                float cell_scale, void* workspace, int64_t workspace_bytes, void* stream) {
  ISO_REQUIRE(n_own >= 0 && import_max >= 0 && n_total >= 0, ISO_ERR_INVALID, "iso_bricks_build: bad sizes");
  ISO_REQUIRE(workspace && (points || n_own == 0), ISO_ERR_INVALID, "iso_bricks_build: null pointer");
  ISO_REQUIRE(import_max == 0 || (import_rec0 && import_rec1 && import_count), ISO_ERR_INVALID,
              "iso_bricks_build: import buffers missing");
  ISO_REQUIRE(cell_scale > 0.f && (radius > 0.f || knn_k > 0), ISO_ERR_INVALID,
              "iso_bricks_build: cell_scale and radius / knn_k must be positive");
  ISO_REQUIRE(((uintptr_t)workspace & 255) == 0, ISO_ERR_INVALID, "iso_bricks_build: workspace must be 256-B aligned");
  const int64_t n_max = n_own + import_max;
  ISO_REQUIRE(id_base + n_max < (1ll << 28), ISO_ERR_UNSUPPORTED, "iso_bricks_build: ids must stay below 2^28");
  const BrickWs w = bricks_carve(workspace, n_max);
  ISO_REQUIRE(workspace_bytes >= w.bytes, ISO_ERR_WORKSPACE, "iso_bricks_build: workspace too small (%lld < %lld)",
              (long long)workspace_bytes, (long long)w.bytes);
  hipStream_t s = (hipStream_t)stream;
  if (bbox)        // NULL: the header was written by iso_bricks_params (N ranks: between it and here the halo exchange)
    hipLaunchKernelGGL(k_bricks_params, dim3(1), dim3(64), 0, s, bbox, 1,
                       BrickParams{n_total, n_own, id_base, radius, knn_k, cell_scale, w.nb_cap}, w.hdr, w.counters);
  int rc = bricks_fill(w, points, normals, payload, n_own, import_rec0, import_rec1, import_count, import_max, s);
  if (rc != ISO_OK) return rc;
  ISO_CHECK_LAUNCH("iso_bricks_build");
  return ISO_OK;
}

static int build_whole_checks(const char* who, const float* points, int64_t n, float radius, int knn_k, float cell_scale,
                              void* workspace, int64_t workspace_bytes, BrickWs& w) {
  ISO_REQUIRE(n >= 0 && workspace && (points || n == 0), ISO_ERR_INVALID, "%s: bad arguments", who);
  ISO_REQUIRE(cell_scale > 0.f && (radius > 0.f || knn_k > 0), ISO_ERR_INVALID,
              "%s: cell_scale and radius / knn_k must be positive", who);
  ISO_REQUIRE(((uintptr_t)workspace & 255) == 0, ISO_ERR_INVALID, "%s: workspace must be 256-B aligned", who);
  ISO_REQUIRE(n < (1ll << 28), ISO_ERR_UNSUPPORTED, "%s: ids must stay below 2^28", who);
  w = bricks_carve(workspace, n);
  ISO_REQUIRE(workspace_bytes >= w.bytes, ISO_ERR_WORKSPACE, "%s: workspace too small (%lld < %lld)", who,
              (long long)workspace_bytes, (long long)w.bytes);
  return ISO_OK;
}

extern "C" int iso_bricks_build_whole(const float* points, const float* normals, const int32_t* payload, int64_t n,
                                      float radius, int knn_k, float cell_scale, void* workspace,
                                      int64_t workspace_bytes, void* stream) {
  BrickWs w;
  int rc = build_whole_checks("iso_bricks_build_whole", points, n, radius, knn_k, cell_scale, workspace, workspace_bytes, w);
  if (rc != ISO_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (n > 0) {
    int gx = iso_div_up(n * 3, 256 * 4);       // four loads per thread and round; a stride of a multiple of 3 floats
    if (gx > 1023) gx = 1023;
    if (gx >= 3) gx -= gx % 3; else gx = 3;
    hipLaunchKernelGGL(k_brick_bbox, dim3(gx), dim3(256), 0, s, points, n, w.counters);
  }
  rc = bricks_fill(w, points, normals, payload, n, nullptr, nullptr, nullptr, 0, s, true,
                   BrickParams{n, n, 0, radius, knn_k, cell_scale, w.nb_cap});
  if (rc != ISO_OK) return rc;
  ISO_CHECK_LAUNCH("iso_bricks_build_whole");
  return ISO_OK;
}

// front workspace (iso_splat_front_workspace_bytes): [chunk table 8 x (n_chunks + 1) ints][tile table 8 x (4 n_chunks + 4) ints]
static inline int32_t* front_tile_table(void* front_ws, int64_t n_points) {
  const int64_t n_chunks = (n_points + 1023) / 1024;
  return (int32_t*)front_ws + 8 * (n_chunks + 1);
}

extern "C" int iso_bricks_build_pending(const float* points, const float* normals, const int32_t* payload, int64_t n,
                                        float radius, int knn_k, float cell_scale, void* workspace,
                                        int64_t workspace_bytes, const iso_follow* f, void* stream) {
  BrickWs w;
  int rc = build_whole_checks("iso_bricks_build_pending", points, n, radius, knn_k, cell_scale, workspace, workspace_bytes, w);
  if (rc != ISO_OK) return rc;
  ChunkScanJob job{nullptr, nullptr, 0, 0, 0, 0, nullptr, nullptr, nullptr};
  if (f && f->views) {
    ISO_REQUIRE(f->n_views >= 1 && f->n_views <= 8 && f->front_ws && f->first_idx_out && f->num_pts_out && f->view_total_out,
                ISO_ERR_INVALID, "iso_bricks_build_pending: follow: mask part incomplete");
    ISO_REQUIRE(f->front_ws_bytes >= iso_splat_front_workspace_bytes(n), ISO_ERR_WORKSPACE,
                "iso_bricks_build_pending: follow.front_ws too small");
    job.n_chunks = (int)((n + 1023) / 1024);
    job.n_tiles = 4 * job.n_chunks;                     // row stride of the tile table (follow.h)
    job.n_real = (int)((n + 255) / 256);
    job.n_views = f->n_views;
    job.tile_cnt = front_tile_table(f->front_ws, n);
    job.chunk = (int32_t*)f->front_ws;
    job.first = f->first_idx_out; job.num = f->num_pts_out; job.view_total = f->view_total_out;
  }
  rc = bricks_fill(w, points, normals, payload, n, nullptr, nullptr, nullptr, 0, (hipStream_t)stream, true,
                   BrickParams{n, n, 0, radius, knn_k, cell_scale, w.nb_cap}, job);
  if (rc != ISO_OK) return rc;
  ISO_CHECK_LAUNCH("iso_bricks_build_pending");
  return ISO_OK;
}

extern "C" int iso_bricks_box_take(void* workspace, int64_t n_max, float* box_out, void* stream) {
  ISO_REQUIRE(workspace && box_out && n_max >= 0, ISO_ERR_INVALID, "iso_bricks_box_take: bad arguments");
  const BrickWs w = bricks_carve(workspace, n_max);
  hipLaunchKernelGGL(k_brick_box_take, dim3(1), dim3(256), 0, (hipStream_t)stream, w.counters, box_out);
  ISO_CHECK_LAUNCH("iso_bricks_box_take");
  return ISO_OK;
}

extern "C" int iso_bricks_params(const float* boxes, int n_boxes, int64_t n_total, int64_t n_own, int64_t id_base,
                                 float radius, int knn_k, float cell_scale, void* workspace, int64_t n_max, void* stream) {
  ISO_REQUIRE(boxes && n_boxes >= 1 && workspace && n_max >= n_own && n_own >= 0 && n_total >= 0, ISO_ERR_INVALID,
              "iso_bricks_params: bad arguments");
  ISO_REQUIRE(cell_scale > 0.f && (radius > 0.f || knn_k > 0), ISO_ERR_INVALID,
              "iso_bricks_params: cell_scale and radius / knn_k must be positive");
  const BrickWs w = bricks_carve(workspace, n_max);
  hipLaunchKernelGGL(k_bricks_params, dim3(1), dim3(64), 0, (hipStream_t)stream, boxes, n_boxes,
                     BrickParams{n_total, n_own, id_base, radius, knn_k, cell_scale, w.nb_cap}, w.hdr, w.counters);
  ISO_CHECK_LAUNCH("iso_bricks_params");
  return ISO_OK;
}

extern "C" int iso_halo_export(void* workspace, const float* points, const float* normals, const int32_t* payload,
                               int64_t n_own, const float* rank_boxes, int world, int rank, int halo_cells,
                               float* export_buf, int64_t capacity, void* stream) {
  ISO_REQUIRE(workspace && rank_boxes && export_buf && world >= 1 && rank >= 0 && rank < world && capacity >= 0 &&
                  capacity < (1ll << 30) && (points || n_own == 0),
              ISO_ERR_INVALID, "iso_halo_export: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  iso_zero_words(export_buf, 4, s);
  if (n_own > 0)
    hipLaunchKernelGGL(k_halo_export, dim3(iso_stream_grid(n_own, 1024)), dim3(256), 0, s, points, normals, payload, n_own,
                       (const BrickHdr*)workspace, rank_boxes, world, rank, halo_cells, (float4*)export_buf, (int)capacity);
  ISO_CHECK_LAUNCH("iso_halo_export");
  return ISO_OK;
}

extern "C" int iso_halo_import(void* workspace, int64_t n_max, const float* gathered, const float* rank_boxes, int world,
                               int rank, int halo_cells, int64_t capacity, float* import_rec0, float* import_rec1,
                               int32_t* import_count, int64_t import_capacity, void* stream) {
  ISO_REQUIRE(workspace && gathered && rank_boxes && import_rec0 && import_rec1 && import_count && world >= 1 &&
                  rank >= 0 && rank < world && capacity >= 0 && import_capacity >= 0 && halo_cells >= 1,
              ISO_ERR_INVALID, "iso_halo_import: bad arguments");
  const BrickWs w = bricks_carve(workspace, n_max);
  hipStream_t s = (hipStream_t)stream;
  iso_zero_words(import_count, 1, s);
  if (capacity > 0 && world > 1)
    hipLaunchKernelGGL(k_halo_import, dim3(iso_stream_grid(capacity, 256) > 64 ? 64 : iso_stream_grid(capacity, 256), world),
                       dim3(256), 0, s, (const float4*)gathered, world, rank, (int)capacity, halo_cells, w.hdr, rank_boxes,
                       (float4*)import_rec0, (float4*)import_rec1, import_count, (int)import_capacity, w.counters);
  ISO_CHECK_LAUNCH("iso_halo_import");
  return ISO_OK;
}

static int env_grid(const char* name, int dflt) {
  const char* e = getenv(name);
  const int v = e ? atoi(e) : 0;
  return v > 0 ? v : dflt;
}
// Grids of the two fused kernels: a workgroup per brick or two, handed out by the dispatcher as CUs free up.  Bricks differ
// widely in cost (points per brick, lanes with an open view) and a persistent grid that strides over the list ends with a
// long tail: measured on the cfg-3a cycle, k_brick_h 241 us at 2048 workgroups, 211 at 3840, 186 at 10240 (flat beyond);
// k_brick_resample 291 / 287 / 270 us at 1024 / 2048 / >= 4096.  A workgroup past the end of the list reads one counter
// and leaves.  (ISO_BK_H_GRID / ISO_BK_RESAMPLE_GRID override, tools/sweep_grid.sh)
static int brick_grid(int forced, int64_t n_own, int cap) {
  if (forced > 0) return forced;
  int64_t g = n_own / 48;
  if (g < 1024) g = 1024;
  if (g > cap) g = cap;
  return (int)g;
}
static bool resample_m10() { static const int v = env_grid("ISO_BK_RESAMPLE_M10", 1); return v == 1; }   // 2: off (A/B)
static int h_grid(int64_t n_own) { static const int f = env_grid("ISO_BK_H_GRID", 0); return brick_grid(f, n_own, 10240); }
static int resample_grid(int64_t n_own) { static const int f = env_grid("ISO_BK_RESAMPLE_GRID", 0); return brick_grid(f, n_own, 8192); }

#ifdef BK_DBG_PHASES
extern "C" int iso_dbg_brick_phases(double* out16) {
  static unsigned long long h[16384 * 8];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(bk_phase), sizeof(h)) != hipSuccess) return -1;
  for (int i = 0; i < 16; ++i) out16[i] = 0.0;
  for (int w = 0; w < 16384; ++w) {
    bool any = false;
    for (int i = 0; i < 8; ++i) { out16[i] += (double)h[w * 8 + i]; any = any || h[w * 8 + i]; }
    if (any) out16[8] += 1.0;                       // workgroups that did any work
  }
  void* dp = nullptr;
  (void)hipGetSymbolAddress(&dp, HIP_SYMBOL(bk_phase));
  (void)hipMemset(dp, 0, sizeof(h));
  return 0;
}
#endif

extern "C" int iso_resample_fused(void* workspace, int64_t n_max, const float* points, int64_t n_own,
                                  int k_plus_one, float* points_out, int64_t* idx_out, float* d2_out,
                                  void* stream) {
  ISO_REQUIRE(workspace && n_own >= 0 && n_max >= n_own, ISO_ERR_INVALID, "iso_resample_fused: bad arguments");
  ISO_REQUIRE(k_plus_one >= 2 && k_plus_one <= 13, ISO_ERR_UNSUPPORTED,
              "iso_resample_fused: K + 1 must be in [2,13], got %d", k_plus_one);
  if (n_own == 0) return ISO_OK;
  ISO_REQUIRE(points && points_out && points != points_out, ISO_ERR_INVALID, "iso_resample_fused: null / aliased pointer");
  const BrickWs w = bricks_carve(workspace, n_max);
  hipStream_t s = (hipStream_t)stream;
  const int K = k_plus_one;
#define ISO_RS(MM)                                                                                           \
  hipLaunchKernelGGL(k_brick_resample<MM>, dim3(resample_grid(n_own)), dim3(BK_THREADS), 0, s, w.hdr, w.off, w.list, \
                     w.rec0, w.rec1, K, points_out, idx_out, d2_out, w.tail, w.counters)
  // (list length K + 1 is the shortest that can certify: the 10-entry list of the default K = 9 fails the list test for
  // ~0.1 % of the queries -- they take the wide window -- and saves two of twelve insertion steps per candidate)
  if (K <= 5) ISO_RS(8);
  else if (K == 9 && resample_m10()) ISO_RS(10);
  else if (K <= 9) ISO_RS(12);
  else ISO_RS(16);
#undef ISO_RS
  int tb = (int)(n_own < 4096 ? n_own : 4096);
  hipLaunchKernelGGL(k_brick_resample_tail<16>, dim3(tb), dim3(64), 0, s, w.hdr, w.off, w.rec0, w.rec1, points, K,
                     points_out, idx_out, d2_out, w.tail, w.counters);
  ISO_CHECK_LAUNCH("iso_resample_fused");
  return ISO_OK;
}

extern "C" int iso_splat_view_mask(const float* points, const float* normals, const float* views, int n_views,
                                   int64_t n, float znear, float zfar, int backface_culling, int32_t* mask_out,
                                   int32_t* view_count_out, void* stream) {
  ISO_REQUIRE(n >= 0 && n_views >= 1 && n_views <= 8, ISO_ERR_UNSUPPORTED, "iso_splat_view_mask: 1..8 views per call");
  ISO_REQUIRE(views && mask_out && view_count_out && (points || n == 0) && (normals || !backface_culling),
              ISO_ERR_INVALID, "iso_splat_view_mask: null pointer");
  hipStream_t s = (hipStream_t)stream;
  iso_zero_words(view_count_out, 8, s);
  if (n > 0)
    hipLaunchKernelGGL(k_view_mask, dim3(iso_stream_grid(n, 256 * 4)), dim3(256), 0, s, points, normals, views, n_views,
                       n, znear, zfar, backface_culling, mask_out, view_count_out);
  ISO_CHECK_LAUNCH("iso_splat_view_mask");
  return ISO_OK;
}

extern "C" int iso_splat_h_fused(void* workspace, int64_t n_max, const float* points, const int32_t* mask,
                                 int64_t n_own, const int32_t* view_total, int n_views, float* h_out,
                                 void* stream) {
  ISO_REQUIRE(workspace && n_own >= 0 && n_max >= n_own, ISO_ERR_INVALID, "iso_splat_h_fused: bad arguments");
  ISO_REQUIRE(n_views >= 1 && n_views <= 8, ISO_ERR_UNSUPPORTED, "iso_splat_h_fused: 1..8 views per call");
  if (n_own == 0) return ISO_OK;
  ISO_REQUIRE(points && mask && view_total && h_out, ISO_ERR_INVALID, "iso_splat_h_fused: null pointer");
  const BrickWs w = bricks_carve(workspace, n_max);
  hipStream_t s = (hipStream_t)stream;
#define ISO_H(NV)                                                                                            \
  hipLaunchKernelGGL(k_brick_h<NV>, dim3(h_grid(n_own)), dim3(BK_THREADS), 0, s, w.hdr, w.off, w.list, w.rec0, \
                     w.rec1, view_total, n_views, h_out, w.tail, w.counters)
  if (n_views <= 1) ISO_H(1);
  else if (n_views <= 2) ISO_H(2);
  else if (n_views <= 4) ISO_H(4);
  else ISO_H(8);
#undef ISO_H
  if (n_own < kTailWideBelow)
    hipLaunchKernelGGL((k_brick_h_tail<2 * BK_TAIL_WAVES>), dim3(4096), dim3(64 * 2 * BK_TAIL_WAVES), 0, s, w.hdr, w.off, w.rec0, w.rec1, points, mask, view_total,
                     n_views, h_out, w.tail, w.counters);
  else
    hipLaunchKernelGGL((k_brick_h_tail<BK_TAIL_WAVES>), dim3(4096), dim3(64 * BK_TAIL_WAVES), 0, s, w.hdr, w.off, w.rec0, w.rec1, points, mask, view_total,
                     n_views, h_out, w.tail, w.counters);
  ISO_CHECK_LAUNCH("iso_splat_h_fused");
  return ISO_OK;
}
