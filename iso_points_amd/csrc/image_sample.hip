// Image values at projected points: get_tensor_values (DSS/utils/__init__.py:325-375), the
// ground-truth mask / colour look-up of every iso-point and pixel sample
// (combined_modeling.py:200,273,292,568,670; implicit_modeling.py:488,536,975).
//
// The reference calls torch.nn.functional.grid_sample(tensor (B,C,H,W), p (B,1,N,2), mode,
// padding_mode='reflection') [align_corners = False], then squeezes and permutes to (B,N,C).
// Here one thread owns a sample (b, n): un-normalise  x = ((p + 1) * size - 1) / 2, reflect
// about the pixel-edge range [-0.5, size - 0.5], clamp to [0, size - 1], then either the nearest
// pixel (round half to even) or the four bilinear corners with the weights
//   nw = (x1 - x)(y1 - y), ne = (x - x0)(y1 - y), sw = (x1 - x)(y - y0), se = (x - x0)(y - y0)
// summed nw, ne, sw, se over the corners inside the image; channels are walked by the same
// thread (C is 1 or 3) and written (B,N,C) directly -- no (B,C,1,N) intermediate, no permute.
// Random 4-B gathers from an image that fits L2: bound by the sample stream, 8 B in + 4 C B out.
// `grid_sample=False` (:357-363) is the same kernel with mode 2: p -> trunc((p + 1)(size - 1)/2).
#include "iso_common.h"

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ float reflect_edge(float x, int size) {
  // reflect_coordinates(x, -1, 2 size - 1): min = -0.5, span = size
  const float mn = -0.5f, span = (float)size;
  x = fabsf(x - mn);
  const float extra = fmodf(x, span);
  const int flips = (int)floorf(x / span);
  return (flips & 1) ? (span - extra) + mn : extra + mn;
}

__device__ __forceinline__ float source_index(float p, int size) {
  float x = ((p + 1.f) * (float)size - 1.f) / 2.f;
  x = reflect_edge(x, size);
  return fminf((float)(size - 1), fmaxf(x, 0.f));      // clip_coordinates
}

__global__ __launch_bounds__(kBlock) void k_image_sample(
    const float* __restrict__ img, int B, int C, int H, int W, const float* __restrict__ p, int64_t N,
    int mode, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= (int64_t)B * N) return;
  const int b = (int)(i / N);
  const float px = p[i * 2], py = p[i * 2 + 1];
  const float* base = img + (int64_t)b * C * H * W;
  const int64_t plane = (int64_t)H * W;
  float* o = out + i * C;
  if (mode == 2) {   // integer indexing (:357-363): .long() truncates toward zero
    const int x = (int)((px + 1.f) * (float)(W - 1) / 2.f), y = (int)((py + 1.f) * (float)(H - 1) / 2.f);
    // python indexing: negative indices wrap once; anything else out of range is the caller's error
    const int xx = x < 0 ? x + W : x, yy = y < 0 ? y + H : y;
    const bool ok = xx >= 0 && xx < W && yy >= 0 && yy < H;
    for (int c = 0; c < C; ++c) o[c] = ok ? base[c * plane + (int64_t)yy * W + xx] : __builtin_nanf("");
    return;
  }
  const float x = source_index(px, W), y = source_index(py, H);
  if (mode == 1) {
    const int xn = (int)nearbyintf(x), yn = (int)nearbyintf(y);
    const bool ok = xn >= 0 && xn < W && yn >= 0 && yn < H;
    for (int c = 0; c < C; ++c) o[c] = ok ? base[c * plane + (int64_t)yn * W + xn] : 0.f;
    return;
  }
  const float x0f = floorf(x), y0f = floorf(y);
  const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
  const float wx1 = (x0f + 1.f) - x, wx0 = x - x0f, wy1 = (y0f + 1.f) - y, wy0 = y - y0f;
  const float nw = wx1 * wy1, ne = wx0 * wy1, sw = wx1 * wy0, se = wx0 * wy0;
  const bool bx0 = x0 >= 0 && x0 < W, bx1 = x1 >= 0 && x1 < W, by0 = y0 >= 0 && y0 < H, by1 = y1 >= 0 && y1 < H;
  for (int c = 0; c < C; ++c) {
    const float* im = base + c * plane;
    float acc = 0.f;
    if (bx0 && by0) acc += im[(int64_t)y0 * W + x0] * nw;
    if (bx1 && by0) acc += im[(int64_t)y0 * W + x1] * ne;
    if (bx0 && by1) acc += im[(int64_t)y1 * W + x0] * sw;
    if (bx1 && by1) acc += im[(int64_t)y1 * W + x1] * se;
    o[c] = acc;
  }
}

}  // namespace

extern "C" int iso_image_sample(const float* image, int batch, int channels, int height, int width,
                                const float* p, int64_t n, int mode, float* out, void* stream) {
  ISO_REQUIRE(batch >= 0 && channels >= 1 && height >= 1 && width >= 1 && n >= 0, ISO_ERR_INVALID,
              "iso_image_sample: bad shape B=%d C=%d H=%d W=%d N=%lld", batch, channels, height, width,
              (long long)n);
  ISO_REQUIRE(mode >= 0 && mode <= 2, ISO_ERR_INVALID, "iso_image_sample: mode %d (0 bilinear, 1 nearest, 2 index)", mode);
  if ((int64_t)batch * n == 0) return ISO_OK;
  ISO_REQUIRE(image && p && out, ISO_ERR_INVALID, "iso_image_sample: null argument");
  hipLaunchKernelGGL(k_image_sample, dim3(iso_div_up((int64_t)batch * n, kBlock)), dim3(kBlock), 0,
                     (hipStream_t)stream, image, batch, channels, height, width, p, n, mode, out);
  ISO_CHECK_LAUNCH("iso_image_sample");
  return ISO_OK;
}
