// Image frame, tile size and the pixel range of a splat's bounding box: shared by the raster pipeline (splat.hip)
// and the band exchange of the N-rank path (band.hip) -- a record is sent to a band exactly when the binning pass of
// that band would list it.
#pragma once
#include <float.h>
#include "iso_common.h"

#pragma clang fp contract(off)

namespace {

constexpr int TILE = 16;  // pixels per tile side; one workgroup = 16x16 lanes

// Image frame: H rows x W columns of square pixels.  NDC follows pytorch3d's non-square convention: the
// shorter side spans [-1, 1], the longer one [-e, e] with e = longer / shorter, i.e. a pixel is 2 / min(H, W)
// wide in both axes; H == W is the reference's square image (rasterizer.py:52 supports nothing else).
struct Frame { int W, H, Tx, Ty, m; float ex, ey; };
static inline Frame make_frame(int H, int W) {
  Frame F;
  F.W = W; F.H = H; F.Tx = (W + 15) / 16; F.Ty = (H + 15) / 16; F.m = H < W ? H : W;
  F.ex = (float)W / (float)F.m; F.ey = (float)H / (float)F.m;
  return F;
}
__device__ __forceinline__ float ndc_x(int i, const Frame& F) { return -F.ex + (2 * i + 1.0f) / F.m; }
__device__ __forceinline__ float ndc_y(int i, const Frame& F) { return -F.ey + (2 * i + 1.0f) / F.m; }


// ---------------------------------------------------------------- tile binning
// NDC-index range of pixels whose centre can lie within [c-r, c+r] (one pixel of slack on
// each side; the exact reference test runs in the raster kernel)
// (n pixels along the axis, half extent e of the axis in NDC, m = min(H, W))
__device__ __forceinline__ bool pixel_range(float c, float r, int n, float e, int m, int& lo, int& hi) {
  if (!(r >= 0.f) || !(c == c)) return false;
  float flo = ((c - r) + e) * 0.5f * (float)m - 0.5f;
  float fhi = ((c + r) + e) * 0.5f * (float)m - 0.5f;
  if (!(flo < 1e9f)) return false;
  if (!(fhi > -1e9f)) return false;
  flo = fmaxf(flo, -4.0f);
  fhi = fminf(fhi, (float)n + 4.0f);
  lo = (int)ceilf(flo) - 1;
  hi = (int)floorf(fhi) + 1;
  if (lo < 0) lo = 0;
  if (hi > n - 1) hi = n - 1;
  return lo <= hi;
}


}  // namespace
