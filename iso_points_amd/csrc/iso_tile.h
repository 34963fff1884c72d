// HBM <-> LDS movers for (n,3) f32 point arrays.
// A tile of `BLOCK` points is 3*BLOCK consecutive floats; it is copied with
// lane-linear accesses (16 B per lane when the tile base is 16-B aligned and
// the tile is full, 4 B per lane otherwise) so every wave instruction covers a
// contiguous span instead of the 12-B-strided pattern a per-point load has.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// TILE: points per tile (a multiple of BLOCK; all of a thread's loads are independent, so a larger tile keeps more of
// them in flight)
template <int BLOCK, int TILE = BLOCK>
__device__ __forceinline__ void iso_tile_load3(const float* __restrict__ g,
                                               int64_t base_pt, int cnt,
                                               float* __restrict__ lds) {
  const float* src = g + base_pt * 3;
  const int nfl = cnt * 3;
  const int t = threadIdx.x;
  if (cnt == TILE && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
    constexpr int NV = TILE * 3 / 4;
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(lds);
    for (int i = t; i < NV; i += BLOCK) d4[i] = s4[i];
  } else {
    for (int i = t; i < nfl; i += BLOCK) lds[i] = src[i];
  }
}

template <int BLOCK, int TILE = BLOCK>
__device__ __forceinline__ void iso_tile_store3(float* __restrict__ g,
                                                int64_t base_pt, int cnt,
                                                const float* __restrict__ lds) {
  float* dst = g + base_pt * 3;
  const int nfl = cnt * 3;
  const int t = threadIdx.x;
  if (cnt == TILE && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
    constexpr int NV = TILE * 3 / 4;
    float4* d4 = reinterpret_cast<float4*>(dst);
    const float4* s4 = reinterpret_cast<const float4*>(lds);
    for (int i = t; i < NV; i += BLOCK) d4[i] = s4[i];
  } else {
    for (int i = t; i < nfl; i += BLOCK) dst[i] = lds[i];
  }
}
