// N ranks, splat stage: the fragment-side exchange (the reference has no distributed layer).
//
// Every rank runs the fused front end on its OWN points (an x-slab of the cloud) and rasterises a BAND of 16-pixel
// tile rows of every view.  A packed row (view v, own point j) is needed by exactly the bands its bounding box
// touches -- the test the binning pass of that band makes (splat_frame.h) -- so each row is sent to those ranks only:
//
//   iso_splat_band_export   own rows -> per-destination segments of ONE send buffer (stable: rows keep their order,
//                           so a segment is view-major / ascending like the packed layout itself), 13 floats per
//                           record: ndc 3, ellipse 3, radii 2, scaler 1, features 3, and the row's GLOBAL id (its
//                           position in the single-GPU packed layout: view-major, ranks in order);
//                           an all-to-all (equal splits, device-side counts in the segment headers) moves them;
//   iso_splat_band_import   the received segments -> this rank's local packed arrays, view-major and, inside a
//                           view, source ranks in order: ascending global id, so the rasteriser's (z, id) tie rule
//                           picks what the single-GPU run picks;
//   iso_splat_band_remap    local row ids of the band's per-pixel lists -> global ids (the single-GPU lists);
//   iso_splat_band_return   the band's per-record results of the backward pass (fixed-point z sum, visible flag)
//                           back into the slots the records arrived in; the reverse all-to-all; the owner adds them
//                           to its rows (iso_splat_band_merge) -- integer adds: order-independent.
//
// Nothing here reads a size on the host: capacities are fixed (cap_pair records per ordered pair of ranks), counts
// travel in the headers, an overflow sets a flag the caller checks when convenient.
#include "splat_frame.h"

#pragma clang fp contract(off)

namespace {

constexpr int kRec = 13;             // floats per record on the wire
constexpr int kHdr = 16;             // ints in front of a segment: [0..7] records per view, [8] records wanted (unclipped)
#ifndef BAND_ROWS_PER_LANE
#define BAND_ROWS_PER_LANE 2      // (4 / 2 / 1: count + scan + fill of a rank of 8 = 35.6 / 29.4 / 30.9 us -- the scan walks capacity / chunk entries)
#endif
constexpr int kRowsPerLane = BAND_ROWS_PER_LANE;
constexpr int kChunkRows = 256 * kRowsPerLane;     // rows per workgroup: 256 lanes x kRowsPerLane consecutive rows
constexpr int kMaxWorld = 64;

struct BandGeo { int world, rank, n_views, ty; };

// tile rows [begin, end) of rank d: the balanced split dist.shard_bounds makes of the Ty tile rows
__device__ __forceinline__ void band_of(int d, int world, int ty, int& b, int& e) {
  const int base = ty / world, rem = ty % world;
  b = d * base + (d < rem ? d : rem);
  e = b + base + (d < rem ? 1 : 0);
}

// bit d set: band d lists this row (the binning pass's test: splat.hip k_bin_lds)
__device__ __forceinline__ unsigned long long band_mask(float x, float y, float z, float rx, float ry, const Frame& F, int world) {
  if (!(z >= 0.f)) return 0ull;
  int x0, x1, y0, y1;
  if (!pixel_range(x, rx, F.W, F.ex, F.m, x0, x1)) return 0ull;
  if (!pixel_range(y, ry, F.H, F.ey, F.m, y0, y1)) return 0ull;
  const int t0 = y0 / TILE, t1 = y1 / TILE;
  unsigned long long m = 0ull;
  for (int d = 0; d < world; ++d) {
    int b, e;
    band_of(d, world, F.Ty, b, e);
    if (t0 < e && t1 >= b) m |= 1ull << d;
  }
  return m;
}

// pass 1: rows per (view, chunk, destination)
__global__ __launch_bounds__(256) void k_band_count(const float* __restrict__ ndc, const float* __restrict__ radii,
                                                    const int64_t* __restrict__ first, const int64_t* __restrict__ num,
                                                    Frame F, int world, int n_chunks, int32_t* __restrict__ chunk_cnt) {
  __shared__ int s_cnt[kMaxWorld];
  const int v = blockIdx.y, c = blockIdx.x;
  if (threadIdx.x < kMaxWorld) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const int64_t len = num[v], base = first[v];
  const int64_t i0 = (int64_t)c * kChunkRows + threadIdx.x * kRowsPerLane;
  int local[8];                                     // (world <= 8 fast path; more destinations go straight to LDS)
#pragma unroll
  for (int d = 0; d < 8; ++d) local[d] = 0;
  for (int k = 0; k < kRowsPerLane; ++k) {
    const int64_t i = i0 + k;
    if (i >= len) break;
    const int64_t p = base + i;
    unsigned long long m = band_mask(ndc[p * 3], ndc[p * 3 + 1], ndc[p * 3 + 2], radii[p * 2], radii[p * 2 + 1], F, world);
#pragma unroll
    for (int d = 0; d < 8; ++d) local[d] += (int)((m >> d) & 1ull);
    for (int d = 8; d < world; ++d) if ((m >> d) & 1ull) atomicAdd(&s_cnt[d], 1);
  }
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    if (d < world) {
      int t = local[d];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
      if ((threadIdx.x & 63) == 0 && t) atomicAdd(&s_cnt[d], t);
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < world) chunk_cnt[((int64_t)v * n_chunks + c) * world + threadIdx.x] = s_cnt[threadIdx.x];
}

// pass 2 (one workgroup): per (destination, view) the exclusive prefix over the chunks in place, the segment headers
__global__ __launch_bounds__(1024) void k_band_scan(int32_t* __restrict__ chunk_cnt, int world, int n_views, int n_chunks,
                                                    int64_t cap_pair, int64_t seg_floats, float* __restrict__ send,
                                                    int32_t* __restrict__ seg_base /*[world][n_views + 1]*/,
                                                    int32_t* __restrict__ flags) {
  __shared__ int s_cnt[kMaxWorld * 8];
  const int t = threadIdx.x;
  // a wave per (destination, view), 64 chunks per trip (a thread per pair walking the chunks one after the other was a
  // chain of n_chunks dependent loads: 20 us of a rank's cycle)
  {
    const int lane = t & 63, n_waves = (int)blockDim.x >> 6;
    for (int pair = t >> 6; pair < world * n_views; pair += n_waves) {
      const int d = pair / n_views, v = pair % n_views;
      int run = 0;
      for (int c0 = 0; c0 < n_chunks; c0 += 64) {
        const int c = c0 + lane;
        int32_t* e = chunk_cnt + ((int64_t)v * n_chunks + c) * world + d;
        const int x = c < n_chunks ? *e : 0;
        int inc = x;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o); if (lane >= o) inc += u; }
        if (c < n_chunks) *e = run + inc - x;
        run += __shfl(inc, 63);
      }
      if (lane == 0) s_cnt[d * 8 + v] = run;
    }
  }
  __syncthreads();
  if (t < world) {
    int run = 0;
    int32_t* hdr = reinterpret_cast<int32_t*>(send + (int64_t)t * seg_floats);
    for (int v = 0; v < 8; ++v) {
      const int c = v < n_views ? s_cnt[t * 8 + v] : 0;
      if (v < n_views) seg_base[t * (n_views + 1) + v] = run;
      int64_t clipped = (int64_t)run + c <= cap_pair ? c : (cap_pair > run ? cap_pair - run : 0);
      hdr[v] = (int)clipped;
      run += c;
    }
    seg_base[t * (n_views + 1) + n_views] = run;
    hdr[8] = run;
    for (int k = 9; k < kHdr; ++k) hdr[k] = 0;
    if (run > cap_pair) atomicOr(&flags[0], 1);          // a segment ran out of capacity: its tail is dropped, flagged
  }
}

// pass 3: the records into their slots; the owner keeps slot -> own row for the way back and clears the per-row
// results that the return path accumulates into
__global__ __launch_bounds__(256) void k_band_fill(const float* __restrict__ ndc, const float* __restrict__ ellipse,
                                                   const float* __restrict__ radii, const float* __restrict__ scaler,
                                                   const float* __restrict__ feat, const int64_t* __restrict__ first,
                                                   const int64_t* __restrict__ num, const int64_t* __restrict__ gid_first,
                                                   Frame F, int world, int n_views, int n_chunks,
                                                   const int32_t* __restrict__ chunk_off, const int32_t* __restrict__ seg_base,
                                                   int64_t cap_pair, int64_t seg_floats, float* __restrict__ send,
                                                   int32_t* __restrict__ sent_row, long long* __restrict__ acc_own,
                                                   uint8_t* __restrict__ vis_own) {
  __shared__ int s_w[4][kMaxWorld];
  const int v = blockIdx.y, c = blockIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t len = num[v], base = first[v];
  const int64_t i0 = (int64_t)c * kChunkRows + threadIdx.x * kRowsPerLane;
  unsigned long long m[kRowsPerLane];
  for (int k = 0; k < kRowsPerLane; ++k) {
    const int64_t i = i0 + k;
    m[k] = 0ull;
    if (i < len) {
      const int64_t p = base + i;
      m[k] = band_mask(ndc[p * 3], ndc[p * 3 + 1], ndc[p * 3 + 2], radii[p * 2], radii[p * 2 + 1], F, world);
      if (acc_own) acc_own[p] = 0;
      if (vis_own) vis_own[p] = 0;
    }
  }
  // rank of each of this thread's rows among the chunk's rows for a destination, in row order: nine destinations per
  // round -- three 10-bit counters to a word, three wave scans, ONE barrier (a scan and two barriers per destination
  // were 21 us of a rank's cycle at world 8)
  for (int d0 = 0; d0 < world; d0 += 9) {
    unsigned mine[3], inc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      mine[j] = 0u;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int d = d0 + 3 * j + q;
        if (d < world) {
          unsigned cnt = 0u;
#pragma unroll
          for (int k = 0; k < kRowsPerLane; ++k) cnt += (unsigned)((m[k] >> d) & 1ull);
          mine[j] |= cnt << (10 * q);
        }
      }
      inc[j] = mine[j];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(inc[j], o); if (lane >= o) inc[j] += t; }
      if (lane == 63) {
#pragma unroll
        for (int q = 0; q < 3; ++q) if (d0 + 3 * j + q < world) s_w[w][d0 + 3 * j + q] = (int)((inc[j] >> (10 * q)) & 1023u);
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 3; ++j) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int d = d0 + 3 * j + q;
        if (d >= world || ((mine[j] >> (10 * q)) & 1023u) == 0u) continue;
        int before = (int)(((inc[j] - mine[j]) >> (10 * q)) & 1023u);
        for (int ww = 0; ww < w; ++ww) before += s_w[ww][d];
        int slot = seg_base[d * (n_views + 1) + v] + chunk_off[((int64_t)v * n_chunks + c) * world + d] + before;
        float* seg = send + (int64_t)d * seg_floats + kHdr;
#pragma unroll
        for (int k = 0; k < kRowsPerLane; ++k) {
          if (!((m[k] >> d) & 1ull)) continue;
          if (slot < cap_pair) {
            const int64_t p = base + i0 + k;
            float* r = seg + (int64_t)slot * kRec;
            r[0] = ndc[p * 3]; r[1] = ndc[p * 3 + 1]; r[2] = ndc[p * 3 + 2];
            r[3] = ellipse[p * 3]; r[4] = ellipse[p * 3 + 1]; r[5] = ellipse[p * 3 + 2];
            r[6] = radii[p * 2]; r[7] = radii[p * 2 + 1];
            r[8] = scaler[p];
            r[9] = feat[p * 3]; r[10] = feat[p * 3 + 1]; r[11] = feat[p * 3 + 2];
            r[12] = __int_as_float((int)(gid_first[v] + i0 + k));
            sent_row[(int64_t)d * cap_pair + slot] = (int32_t)p;
          }
          ++slot;
        }
      }
    }
    if (d0 + 9 < world) __syncthreads();                  // (s_w is written again in the next round)
  }
}

// ---- receiving side -------------------------------------------------------------------------------------------
// layout of the local packed arrays from the headers of the received segments (one workgroup)
__global__ void k_band_layout(const float* __restrict__ recv, int world, int n_views, int64_t seg_floats, int64_t cap_pair,
                              int64_t cap_local, int64_t* __restrict__ first_l, int64_t* __restrict__ num_l,
                              int32_t* __restrict__ place /*[world][n_views][2]: local row, slot of the first record*/,
                              int32_t* __restrict__ flags) {
  // the nine header words of every segment at once (one thread reading them as it went was a chain of dependent loads)
  __shared__ int s_hdr[kMaxWorld][9];
  if (blockIdx.x != 0) return;
  for (int i = threadIdx.x; i < world * 9; i += blockDim.x)
    s_hdr[i / 9][i % 9] = reinterpret_cast<const int32_t*>(recv + (int64_t)(i / 9) * seg_floats)[i % 9];
  __syncthreads();
  if (threadIdx.x != 0) return;
  int64_t run = 0;
  int over = 0;
  for (int s = 0; s < world; ++s) {
    const int* hdr = s_hdr[s];
    if (hdr[8] > cap_pair) over = 1;                    // the sender clipped this segment (it flagged it too)
  }
  for (int v = 0; v < n_views; ++v) {
    first_l[v] = run;
    for (int s = 0; s < world; ++s) {
      const int* hdr = s_hdr[s];
      int slot0 = 0;
      for (int vv = 0; vv < v; ++vv) slot0 += hdr[vv];
      int64_t c = hdr[v];
      if (run + c > cap_local) { c = cap_local > run ? cap_local - run : 0; over = 1; }
      place[((int64_t)s * n_views + v) * 2] = (int32_t)run;
      place[((int64_t)s * n_views + v) * 2 + 1] = slot0;
      place[(int64_t)world * n_views * 2 + (int64_t)s * n_views + v] = (int32_t)c;
      run += c;
    }
    num_l[v] = run - first_l[v];
  }
  if (over) atomicOr(&flags[0], 2);
}

// grid (x, source rank, view): the records of (s, v) -> local rows place.. ; per-row results of the backward cleared
__global__ __launch_bounds__(256) void k_band_arrange(const float* __restrict__ recv, int world, int n_views, int64_t seg_floats,
                                                      int64_t cap_pair, const int32_t* __restrict__ place, float cutoffC,
                                                      float* __restrict__ ndc, float* __restrict__ ellipse,
                                                      float* __restrict__ cutoff, float* __restrict__ radii,
                                                      float* __restrict__ scaler, float* __restrict__ feat,
                                                      int32_t* __restrict__ gid, int32_t* __restrict__ origin,
                                                      long long* __restrict__ acc_l, uint8_t* __restrict__ vis_l) {
  const int s = blockIdx.y, v = blockIdx.z;
  const int64_t row0 = place[((int64_t)s * n_views + v) * 2], slot0 = place[((int64_t)s * n_views + v) * 2 + 1];
  const int64_t n = place[(int64_t)world * n_views * 2 + (int64_t)s * n_views + v];
  const float* seg = recv + (int64_t)s * seg_floats + kHdr;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
    const float* r = seg + (slot0 + j) * kRec;
    const int64_t p = row0 + j;
    ndc[p * 3] = r[0]; ndc[p * 3 + 1] = r[1]; ndc[p * 3 + 2] = r[2];
    ellipse[p * 3] = r[3]; ellipse[p * 3 + 1] = r[4]; ellipse[p * 3 + 2] = r[5];
    radii[p * 2] = r[6]; radii[p * 2 + 1] = r[7];
    scaler[p] = r[8];
    feat[p * 3] = r[9]; feat[p * 3 + 1] = r[10]; feat[p * 3 + 2] = r[11];
    cutoff[p] = cutoffC;
    gid[p] = __float_as_int(r[12]);
    origin[p] = (int32_t)((int64_t)s * cap_pair + slot0 + j);
    acc_l[p] = 0;
    vis_l[p] = 0;
  }
}

// local row ids of a band's lists -> global ids (grid.y = view; the band's pixel rows of every view)
__global__ void k_band_remap(const int32_t* __restrict__ idx_l, const int32_t* __restrict__ gid, int64_t band_entries,
                             int64_t view_stride, int32_t* __restrict__ idx_g) {
  const int64_t o = (int64_t)blockIdx.y * view_stride;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < band_entries; i += (int64_t)gridDim.x * blockDim.x) {
    const int l = idx_l[o + i];
    idx_g[o + i] = l >= 0 ? gid[l] : l;
  }
}

// the band's per-record results into the slots the records came in
__global__ void k_band_return(const long long* __restrict__ acc_l, const uint8_t* __restrict__ vis_l,
                              const int32_t* __restrict__ origin, const int64_t* __restrict__ first_l,
                              const int64_t* __restrict__ num_l, int n_views, long long* __restrict__ ret /*[world][cap_pair][2]*/) {
  const int64_t total = first_l[n_views - 1] + num_l[n_views - 1];
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
    const int64_t o = origin[p];
    ret[o * 2] = acc_l[p];
    ret[o * 2 + 1] = vis_l[p];
  }
}

// the owner: what the bands found for the records it sent them (grid.y = band)
__global__ void k_band_merge(const long long* __restrict__ back, const int32_t* __restrict__ sent_row,
                             const int32_t* __restrict__ seg_base, int n_views, int64_t cap_pair,
                             long long* __restrict__ acc_own, uint8_t* __restrict__ vis_own) {
  const int d = blockIdx.y;
  int64_t n = seg_base[d * (n_views + 1) + n_views];
  if (n > cap_pair) n = cap_pair;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
    const int64_t o = (int64_t)d * cap_pair + j;
    const int64_t p = sent_row[o];
    const long long a = back[o * 2];
    if (a) atomicAdd(reinterpret_cast<unsigned long long*>(&acc_own[p]), (unsigned long long)a);
    if (back[o * 2 + 1]) vis_own[p] = 1;
  }
}

// global row id of the first own row of every view: view-major, ranks in order (counts: (world, 8) rows per view)
__global__ void k_band_gid_first(const int32_t* __restrict__ counts, int world, int n_views, int rank,
                                 int64_t* __restrict__ gid_first, int64_t* __restrict__ first_g, int64_t* __restrict__ num_g) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int64_t run = 0;
  for (int v = 0; v < n_views; ++v) {
    first_g[v] = run;
    for (int s = 0; s < world; ++s) {
      if (s == rank) gid_first[v] = run;
      run += counts[s * 8 + v];
    }
    num_g[v] = run - first_g[v];
  }
}

}  // namespace

static Frame frame_of(int image_size, int image_width) { return make_frame(image_size, image_width > 0 ? image_width : image_size); }

extern "C" int64_t iso_splat_band_segment_floats(int64_t cap_pair) { return kHdr + (cap_pair < 0 ? 0 : cap_pair) * kRec; }

extern "C" int64_t iso_splat_band_export_workspace_bytes(int64_t max_rows_per_view, int n_views, int world) {
  if (max_rows_per_view < 0) max_rows_per_view = 0;
  const int64_t n_chunks = (max_rows_per_view + kChunkRows - 1) / kChunkRows + 1;
  return 4 * (n_chunks * n_views * world + (int64_t)world * (n_views + 1)) + 8 * 3 * 8 + 64;
}

// ndc / ellipse / radii / scaler / features: the rank's own packed rows (iso_splat_front); first_idx / num_points: their
// layout (device, int64); counts: (world, 8) int32 rows per view of every rank (as all-gathered).  Outputs: send
// (world segments of iso_splat_band_segment_floats(cap_pair) floats), sent_row (world * cap_pair) int32, and -- in the
// workspace -- what iso_splat_band_merge needs later; first_global / num_global (n_views int64): the layout of the
// single-GPU packed arrays, gid_first (n_views int64): where this rank's rows of every view start in it.
extern "C" int iso_splat_band_export(const float* ndc, const float* ellipse, const float* radii, const float* scaler,
                                     const float* features, const int64_t* first_idx, const int64_t* num_points,
                                     int n_views, int64_t max_rows_per_view, const int32_t* counts, int world, int rank,
                                     int image_size, int image_width, int64_t cap_pair, float* send, int32_t* sent_row,
                                     int64_t* acc_own, uint8_t* vis_own, int64_t* gid_first, int64_t* first_global,
                                     int64_t* num_global, int32_t* flags, void* workspace, int64_t workspace_bytes,
                                     void* stream) {
  ISO_REQUIRE(n_views >= 1 && n_views <= 8 && world >= 1 && world <= kMaxWorld && rank >= 0 && rank < world &&
                  max_rows_per_view >= 0 && cap_pair >= 0 && image_size > 0,
              ISO_ERR_INVALID, "iso_splat_band_export: bad sizes (1..8 views, world <= 64)");
  ISO_REQUIRE(ndc && ellipse && radii && scaler && features && first_idx && num_points && counts && send && sent_row &&
                  gid_first && first_global && num_global && flags && workspace,
              ISO_ERR_INVALID, "iso_splat_band_export: null pointer");
  ISO_REQUIRE(workspace_bytes >= iso_splat_band_export_workspace_bytes(max_rows_per_view, n_views, world), ISO_ERR_WORKSPACE,
              "iso_splat_band_export: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const Frame F = frame_of(image_size, image_width);
  const int n_chunks = (int)((max_rows_per_view + kChunkRows - 1) / kChunkRows) + 1;
  int32_t* chunk = (int32_t*)workspace;
  int32_t* seg_base = chunk + (int64_t)n_chunks * n_views * world;
  const int64_t seg_floats = iso_splat_band_segment_floats(cap_pair);
  hipLaunchKernelGGL(k_band_gid_first, dim3(1), dim3(64), 0, s, counts, world, n_views, rank, gid_first, first_global, num_global);
  hipLaunchKernelGGL(k_band_count, dim3(n_chunks, n_views), dim3(256), 0, s, ndc, radii, first_idx, num_points, F, world,
                     n_chunks, chunk);
  hipLaunchKernelGGL(k_band_scan, dim3(1), dim3(1024), 0, s, chunk, world, n_views, n_chunks, cap_pair, seg_floats, send,
                     seg_base, flags);
  hipLaunchKernelGGL(k_band_fill, dim3(n_chunks, n_views), dim3(256), 0, s, ndc, ellipse, radii, scaler, features, first_idx,
                     num_points, gid_first, F, world, n_views, n_chunks, chunk, seg_base, cap_pair, seg_floats, send, sent_row,
                     reinterpret_cast<long long*>(acc_own), vis_own);
  ISO_CHECK_LAUNCH("iso_splat_band_export");
  return ISO_OK;
}

// recv: the world segments this rank received.  Local packed arrays (capacity cap_local rows) + their layout
// first_local / num_local (n_views int64) + gid / origin (cap_local int32) + the cleared per-row results acc_local
// (int64) / vis_local (u8).  place: int32 scratch of 3 * world * n_views.
extern "C" int iso_splat_band_import(const float* recv, int world, int n_views, int64_t cap_pair, int64_t cap_local,
                                     float cutoff, float* ndc, float* ellipse, float* cutoff_out, float* radii,
                                     float* scaler, float* features, int32_t* gid, int32_t* origin, int64_t* acc_local,
                                     uint8_t* vis_local, int64_t* first_local, int64_t* num_local, int32_t* place,
                                     int32_t* flags, void* stream) {
  ISO_REQUIRE(n_views >= 1 && n_views <= 8 && world >= 1 && world <= kMaxWorld && cap_pair >= 0 && cap_local >= 0,
              ISO_ERR_INVALID, "iso_splat_band_import: bad sizes");
  ISO_REQUIRE(recv && ndc && ellipse && cutoff_out && radii && scaler && features && gid && origin && acc_local &&
                  vis_local && first_local && num_local && place && flags,
              ISO_ERR_INVALID, "iso_splat_band_import: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const int64_t seg_floats = iso_splat_band_segment_floats(cap_pair);
  hipLaunchKernelGGL(k_band_layout, dim3(1), dim3(64), 0, s, recv, world, n_views, seg_floats, cap_pair, cap_local,
                     first_local, num_local, place, flags);
  int gx = iso_stream_grid(cap_pair > 0 ? cap_pair : 1, 256);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(k_band_arrange, dim3(gx, world, n_views), dim3(256), 0, s, recv, world, n_views, seg_floats, cap_pair,
                     place, cutoff, ndc, ellipse, cutoff_out, radii, scaler, features, gid, origin,
                     reinterpret_cast<long long*>(acc_local), vis_local);
  ISO_CHECK_LAUNCH("iso_splat_band_import");
  return ISO_OK;
}

// idx_local / idx_global: (N, H, W, K) arrays; the band = pixel rows [y0, y1) of every view
extern "C" int iso_splat_band_remap(const int32_t* idx_local, const int32_t* gid, int n_views, int64_t view_pixels,
                                    int64_t band_pixel0, int64_t band_pixels, int points_per_pixel, int32_t* idx_global,
                                    void* stream) {
  ISO_REQUIRE(n_views >= 0 && view_pixels >= 0 && band_pixels >= 0 && band_pixel0 >= 0 && points_per_pixel >= 1,
              ISO_ERR_INVALID, "iso_splat_band_remap: bad sizes");
  if (n_views == 0 || band_pixels == 0) return ISO_OK;
  ISO_REQUIRE(idx_local && gid && idx_global, ISO_ERR_INVALID, "iso_splat_band_remap: null pointer");
  const int64_t K = points_per_pixel;
  hipLaunchKernelGGL(k_band_remap, dim3(iso_stream_grid(band_pixels * K, 256), n_views), dim3(256), 0, (hipStream_t)stream,
                     idx_local + band_pixel0 * K, gid, band_pixels * K, view_pixels * K, idx_global + band_pixel0 * K);
  ISO_CHECK_LAUNCH("iso_splat_band_remap");
  return ISO_OK;
}

extern "C" int iso_splat_band_return(const int64_t* acc_local, const uint8_t* vis_local, const int32_t* origin,
                                     const int64_t* first_local, const int64_t* num_local, int n_views, int64_t cap_local,
                                     int64_t* ret, void* stream) {
  ISO_REQUIRE(acc_local && vis_local && origin && first_local && num_local && ret && n_views >= 1 && cap_local >= 0,
              ISO_ERR_INVALID, "iso_splat_band_return: bad arguments");
  if (cap_local == 0) return ISO_OK;
  hipLaunchKernelGGL(k_band_return, dim3(iso_stream_grid(cap_local, 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const long long*>(acc_local), vis_local, origin, first_local, num_local, n_views,
                     reinterpret_cast<long long*>(ret));
  ISO_CHECK_LAUNCH("iso_splat_band_return");
  return ISO_OK;
}

// back: the world segments (cap_pair x 2 int64 each) that came back; workspace: the one of iso_splat_band_export
extern "C" int iso_splat_band_merge(const int64_t* back, const int32_t* sent_row, int world, int n_views,
                                    int64_t max_rows_per_view, int64_t cap_pair, int64_t* acc_own, uint8_t* vis_own,
                                    const void* workspace, void* stream) {
  ISO_REQUIRE(back && sent_row && acc_own && vis_own && workspace && world >= 1 && n_views >= 1 && cap_pair >= 0,
              ISO_ERR_INVALID, "iso_splat_band_merge: bad arguments");
  if (cap_pair == 0) return ISO_OK;
  const int n_chunks = (int)((max_rows_per_view + kChunkRows - 1) / kChunkRows) + 1;
  const int32_t* seg_base = (const int32_t*)workspace + (int64_t)n_chunks * n_views * world;
  int gx = iso_stream_grid(cap_pair, 256);
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL(k_band_merge, dim3(gx, world), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const long long*>(back), sent_row, seg_base, n_views, cap_pair,
                     reinterpret_cast<long long*>(acc_own), vis_own);
  ISO_CHECK_LAUNCH("iso_splat_band_merge");
  return ISO_OK;
}
