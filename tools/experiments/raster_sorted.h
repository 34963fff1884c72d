// EXPERIMENT (round 4, measured, not adopted; write-up in profiles/HISTORY.md): the raster pass on depth-sorted candidates.
// Built as k_raster_sorted<KMAX> inside csrc/splat.hip (it needs that file's helpers: Frame, pixel_range, CompositeArgs,
// RS_PH); bit-identical to k_raster on the whole GPU suite.  cfg-3a cycle: 214 us (4-byte stores per append) / 233 us
// (lists in registers) against 219 us for k_raster: the bitonic sort of an item's <= 1024 64-bit keys costs what the
// insertions it saves cost (26 % of a work item's time; ~1.5-3.5 k wave instructions of the ~22 k an item takes), and 96
// VGPRs + 27 KB of LDS put five workgroups on a CU instead of seven.
// ---- the raster pass on DEPTH-SORTED candidates ----------------------------------------------------------------------
// k_raster above keeps a K-entry list per pixel and INSERTS every hit (a pixel of the cfg-3a cycle is hit 21 times for
// its 8 entries: 25 % of the kernel inserting + 10 % waiting for the longest list of the wave), then gathers q, scaler and
// features of the survivors from global memory (another 25 %).  Here the item's candidates are first sorted by (z, id) --
// the total order of the K-best rule (rasterize_points_cpu.cpp:85-112) -- in LDS (bitonic, 64-bit keys, <= kSortMax
// candidates: larger items are left to k_raster, launched behind this kernel with only_oversize).  Hits then reach a pixel
// in list order: the first K ARE the result -- they are appended, never inserted; the depth-merging cut is known at the
// second hit (z - z0 > threshold ends the list: everything behind is cut too); q is taken from the candidate's record in
// LDS (the operands of the hit test), and so are scaler and features of the fused compositing (a fourth 16-byte record
// per candidate, C <= 3) -- no epilogue, no gathers.  A chunk of 256 candidates marks its hits as BITS of per-pixel masks
// (atomicOr: no list capacity, ascending bit = list order); wide boxes are tested by the pixel threads themselves.  When
// every pixel of the tile has its K entries (or its cut) the rest of the list is not read.
// Slices of a tile write their raw lists to scratch ([z | q | id][k][pixel], as k_raster's) for k_raster_merge.
__device__ __forceinline__ void bitonic_sort_u64(unsigned long long* __restrict__ key, int n /* power of two */) {
  for (int k = 2; k <= n; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < n / 2; t += 256) {
        const int i = ((t / j) * 2 * j) + (t % j), l = i + j;
        const unsigned long long a = key[i], b = key[l];
        const bool up = (i & k) == 0;
        if ((a > b) == up) { key[i] = b; key[l] = a; }
      }
      __syncthreads();
    }
  }
}

template <int KMAX>
__global__ __launch_bounds__(256, 5) void k_raster_sorted(
    const float* __restrict__ pts, const float* __restrict__ ellipse, const float* __restrict__ cutoff,
    const float* __restrict__ radii, const int4* __restrict__ items, const int32_t* __restrict__ item_count,
    float* __restrict__ scratch, const int32_t* __restrict__ tile_off, const int32_t* __restrict__ pairs, int64_t capacity,
    Frame F, int K, float depth_thres, int32_t* __restrict__ idx_out, float* __restrict__ zbuf_out,
    float* __restrict__ q_out, float* __restrict__ occ_out, CompositeArgs ca) {
  // LDS: the sort keys (8 KB) are dead once the sorted ids are copied out; the pixel masks take their place
  __shared__ __attribute__((aligned(16))) unsigned long long s_key[kSortMax];
  __shared__ int s_ids[kSortMax];
  __shared__ float4 s_r0[256], s_r1[256], s_cm[256];
  __shared__ float2 s_r2[256];
  __shared__ short s_wide[256];
  __shared__ int s_nw;
  unsigned* s_mask = reinterpret_cast<unsigned*>(s_key);          // [256 pixels][8 words]
  static_assert(kSortMax * 8 >= 256 * 8 * 4, "the masks fit the key buffer");
  if ((int)blockIdx.x >= *item_count) return;
  const int4 it = items[blockIdx.x];
  const int tile = it.x, slice = it.y, nslices = it.z, slot = it.w;
  const int64_t off = tile_off[tile];
  int cnt = tile_off[tile + 1] - tile_off[tile];
  if (off + cnt > capacity) cnt = off < capacity ? (int)(capacity - off) : 0;  // overflow guard
  int c_begin = 0;
  if (nslices > 1) {                       // slice s of ns: chunk-aligned share of the list (as k_raster)
    const int chunks = (cnt + 255) / 256;
    c_begin = (int)((int64_t)chunks * slice / nslices) * 256;
    cnt = min(cnt, (int)((int64_t)chunks * (slice + 1) / nslices) * 256);
  }
  const int m_all = cnt - c_begin;
  if (m_all > kSortMax) return;            // k_raster(only_oversize) takes it
  RS_PH(-1);
  const int tx = tile % F.Tx, ty = (tile / F.Tx) % F.Ty, n = tile / (F.Tx * F.Ty);
  const int lx = threadIdx.x % TILE, ly = threadIdx.x / TILE;
  const int xi = tx * TILE + lx, yi = ty * TILE + ly;  // NDC pixel index
  const bool inside = xi < F.W && yi < F.H;
  const float xf = ndc_x(xi, F), yf = ndc_y(yi, F);
  // ---- sort the item's candidates by (z, id) ----
  int nsort = 2;
  while (nsort < m_all) nsort <<= 1;
  for (int t = threadIdx.x; t < nsort; t += 256) {
    unsigned long long k = ~0ull;
    if (t < m_all) {
      const int p = pairs[off + c_begin + t];
      // (z >= 0 for every listed candidate -- the binning pass drops the others --, so its bits order like the value)
      k = ((unsigned long long)__float_as_uint(pts[(int64_t)p * 3 + 2]) << 32) | (unsigned)p;
    }
    s_key[t] = k;
  }
  __syncthreads();
  if (m_all > 1) bitonic_sort_u64(s_key, nsort);
  for (int t = threadIdx.x; t < m_all; t += 256) s_ids[t] = (int)(unsigned)(s_key[t] & 0xffffffffull);
  __syncthreads();
  RS_PH(0);
  for (int t = threadIdx.x; t < 256 * 8; t += 256) s_mask[t] = 0u;       // (the keys are dead)
  if (threadIdx.x == 0) s_nw = 0;
  // ---- the pixel's state ----
  const bool sliced = nslices > 1;
  const int yo = F.H - 1 - yi, xo = F.W - 1 - xi;       // output pixel is flipped in both axes (rasterize_points.cu:577-580)
  const int64_t pix = ((int64_t)n * F.H + yo) * F.W + xo;
  float* const sc = scratch + (int64_t)slot * 3 * KMAX * 256;
  int have = 0;                            // entries written so far
  bool done = !inside;                     // K entries, or the depth-merging cut reached
  float z0 = 0.f, sw = 0.f, acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = 0.f;
  const bool stage_cm = ca.scaler && ca.C <= 3;
  float4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = r0, r2 = r0, rc = r0;
  auto fetch = [&](int c0) {
    if (c0 + (int)threadIdx.x < m_all) {
      const int p = s_ids[c0 + threadIdx.x];
      r0 = make_float4(pts[(int64_t)p * 3], pts[(int64_t)p * 3 + 1], pts[(int64_t)p * 3 + 2], __int_as_float(p));
      r1 = make_float4(ellipse[(int64_t)p * 3], ellipse[(int64_t)p * 3 + 1], ellipse[(int64_t)p * 3 + 2], cutoff[p]);
      r2 = make_float4(radii[(int64_t)p * 2], radii[(int64_t)p * 2 + 1], 0.f, 0.f);
      if (stage_cm) {
        rc.x = ca.scaler[p];
        rc.y = ca.C > 0 ? ca.feat[(int64_t)p * ca.C] : 0.f;
        rc.z = ca.C > 1 ? ca.feat[(int64_t)p * ca.C + 1] : 0.f;
        rc.w = ca.C > 2 ? ca.feat[(int64_t)p * ca.C + 2] : 0.f;
      }
    }
  };
  // the next entry of this pixel's list: candidate k of the chunk (a hit by the exact test).  The list sits in
  // registers (static indices: an append is a chain of selects, at most K of them per pixel) and leaves as 16-byte
  // stores at the end -- a 4-byte store per value and append was 8x the store requests of the whole kernel.
  float lz[KMAX], lq[KMAX];
  int li[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) { lz[j] = sliced ? FLT_MAX : -1.0f; lq[j] = -1.0f; li[j] = sliced ? 0x7fffffff : -1; }
  auto append = [&](int k) {
    const float4 c0v = s_r0[k], c1v = s_r1[k];
    const float dx = xf - c0v.x, dy = yf - c0v.y;
    const float q = c1v.x * dx * dx + c1v.y * dx * dy + c1v.z * dy * dy;      // rasterize_points.cu:94
    const float z = c0v.z;
    const int id = __float_as_int(c0v.w);
    if (!sliced) {                         // (a slice keeps its raw list: k_raster_merge applies the cut)
      if (have == 0) z0 = z;
      if ((z - z0) > depth_thres) { done = true; return; }     // this entry and all behind it are cut (ascending z)
    }
#pragma unroll
    for (int j = 0; j < KMAX; ++j) if (j == have) { lz[j] = z; lq[j] = q; li[j] = id; }
    if (!sliced && ca.scaler) {
      float w;
      if (stage_cm) {
        const float4 cm = s_cm[k];
        w = expf(-0.5f * q) * cm.x;
        acc[0] += w * cm.y; acc[1] += w * cm.z; acc[2] += w * cm.w;
      } else {
        w = expf(-0.5f * q) * ca.scaler[id];
#pragma unroll
        for (int c = 0; c < 8; ++c) if (c < ca.C) acc[c] += w * ca.feat[(int64_t)id * ca.C + c];
      }
      sw += w;
    }
    if (++have == K) done = true;
  };
  __syncthreads();
  fetch(0);
  for (int c0 = 0; c0 < m_all; c0 += 256) {
    const int m = min(256, m_all - c0);
    if ((int)threadIdx.x < m) {
      s_r0[threadIdx.x] = r0; s_r1[threadIdx.x] = r1; s_r2[threadIdx.x] = make_float2(r2.x, r2.y);
      if (stage_cm) s_cm[threadIdx.x] = rc;
    }
    __syncthreads();                                    // records visible; masks zero; s_nw zero
    RS_PH(1);
    fetch(c0 + 256);
    if ((int)threadIdx.x < m) {
      const int k = threadIdx.x;
      const float4 c0v = s_r0[k], c1v = s_r1[k];
      const float2 c2v = s_r2[k];
      int x0 = 0, x1 = -1, y0 = 0, y1 = -1;
      const bool any = pixel_range(c0v.x, c2v.x, F.W, F.ex, F.m, x0, x1) && pixel_range(c0v.y, c2v.y, F.H, F.ey, F.m, y0, y1);
      x0 = max(x0, tx * TILE); x1 = min(x1, tx * TILE + TILE - 1);
      y0 = max(y0, ty * TILE); y1 = min(y1, ty * TILE + TILE - 1);
      if (any && x0 <= x1 && y0 <= y1) {
        if ((x1 - x0 + 1) * (y1 - y0 + 1) > kWideArea) {
          s_wide[atomicAdd(&s_nw, 1)] = (short)k;
        } else {
          const unsigned bit = 1u << (k & 31);
          for (int y = y0; y <= y1; ++y) {
            const float dy = ndc_y(y, F) - c0v.y;
            if (fabsf(dy) > c2v.y) continue;
            for (int x = x0; x <= x1; ++x) {
              const float dx = ndc_x(x, F) - c0v.x;
              if (fabsf(dx) > c2v.x) continue;                                          // rasterize_points.cu:92
              const float q = c1v.x * dx * dx + c1v.y * dx * dy + c1v.z * dy * dy;      // :94
              if (q > c1v.w) continue;                                                  // :96
              const int pl = (y - ty * TILE) * TILE + (x - tx * TILE);
              atomicOr(&s_mask[pl * 8 + (k >> 5)], bit);
            }
          }
        }
      }
    }
    __syncthreads();                                    // masks complete
    RS_PH(2);
    unsigned mk[8];
    {
      const uint4 a = *reinterpret_cast<const uint4*>(s_mask + threadIdx.x * 8);
      const uint4 b = *reinterpret_cast<const uint4*>(s_mask + threadIdx.x * 8 + 4);
      mk[0] = a.x; mk[1] = a.y; mk[2] = a.z; mk[3] = a.w; mk[4] = b.x; mk[5] = b.y; mk[6] = b.z; mk[7] = b.w;
      *reinterpret_cast<uint4*>(s_mask + threadIdx.x * 8) = make_uint4(0u, 0u, 0u, 0u);
      *reinterpret_cast<uint4*>(s_mask + threadIdx.x * 8 + 4) = make_uint4(0u, 0u, 0u, 0u);
    }
    const int nw = s_nw;
    if (!done) {
      for (int i = 0; i < nw; ++i) {                    // wide boxes: this pixel's own test
        const int k = s_wide[i];
        const float4 c0v = s_r0[k], c1v = s_r1[k];
        const float2 c2v = s_r2[k];
        const float dx = xf - c0v.x, dy = yf - c0v.y;
        if (fabsf(dx) > c2v.x || fabsf(dy) > c2v.y) continue;
        const float q = c1v.x * dx * dx + c1v.y * dx * dy + c1v.z * dy * dy;
        if (q > c1v.w) continue;
#pragma unroll
        for (int wd = 0; wd < 8; ++wd) if (wd == (k >> 5)) mk[wd] |= 1u << (k & 31);
      }
#pragma unroll
      for (int wd = 0; wd < 8; ++wd) {
        unsigned b = mk[wd];
        while (b && !done) {
          const int k = wd * 32 + (__ffs((int)b) - 1);
          b &= b - 1;
          append(k);
        }
      }
    }
    // all pixels done: the rest of the list cannot change anything (also the barrier that frees records and s_nw)
    RS_PH(3);
    const int all_done = __syncthreads_and(done ? 1 : 0);
    RS_PH(4);
    if (threadIdx.x == 0) s_nw = 0;
    if (all_done) break;
  }
  if (sliced) {                            // [z | q | id][k][pixel]: all KMAX slots (k_raster_merge reads them all)
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      sc[j * 256 + threadIdx.x] = lz[j];
      sc[(KMAX + j) * 256 + threadIdx.x] = lq[j];
      sc[(2 * KMAX + j) * 256 + threadIdx.x] = __int_as_float(li[j]);
    }
    RS_PH(5);
    RS_PH_FLUSH();
    return;
  }
  if (!inside) return;
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    if (j < K) { idx_out[pix * K + j] = li[j]; zbuf_out[pix * K + j] = lz[j]; q_out[pix * K + j] = lq[j]; }
  }
  const bool hit = have > 0;
  occ_out[pix] = hit ? 1.0f : 0.0f;
  if (ca.scaler) {
    float d = 1.0f;
    if (ca.norm) d = sw > ca.eps ? sw : ca.eps;
#pragma unroll
    for (int c = 0; c < 8; ++c) if (c < ca.C) ca.img[pix * (ca.C + 1) + c] = ca.norm ? acc[c] / d : acc[c];
    ca.img[pix * (ca.C + 1) + ca.C] = hit ? 1.0f : 0.0f;
  }
  RS_PH(5);
  RS_PH_FLUSH();
}

