// Shared between idr.hip (f32-MFMA kernels, packing, C ABI) and idr_x16.hip (split-fp16 kernel):
// network shape, packed-buffer layout, argument block, encoding and activation.
#pragma once
#include "iso_common.h"
#include "mlp_common.h"

namespace {


constexpr int kD0Pad = 64;          // padded encoding width (D0 = 3 + 6F <= 63)
constexpr int kW0Row = 64;          // floats per feature row of the VALU backward image

struct IdrShape {
  int H, n_layers, skip, F, D0;     // skip < 0: no skip connection
};

// packed buffer (floats):
//   [b0 H][FW0 (kD0Pad/16)*NT*256][W0v 4*(H/4)*kW0Row]
//   per l = 1..n_layers-1: [b_l H][FW_l H*H][BW_l H*H]
//   [WLimg H][b_last, pad 4]
__host__ __device__ inline int64_t idr_off_b0() { return 0; }
__host__ __device__ inline int64_t idr_off_fw0(int H) { return H; }
__host__ __device__ inline int64_t idr_off_w0v(int H) { return H + (int64_t)(kD0Pad / 16) * (H / 16) * 256; }
__host__ __device__ inline int64_t idr_off_layer(int H, int l) {   // l >= 1
  return idr_off_w0v(H) + (int64_t)H * kW0Row + (int64_t)(l - 1) * ((int64_t)H + 2 * (int64_t)H * H);
}
__host__ __device__ inline int64_t idr_off_wl(int H, int n_layers) { return idr_off_layer(H, n_layers); }
__host__ __device__ inline int64_t idr_total(int H, int n_layers) { return idr_off_wl(H, n_layers) + H + 4; }

// raw layout (effective weights, torch row-major [out][in]):
//   for l = 0..n_layers: W_l[out_l*in_l] b_l[out_l];
//   in_0 = D0, out_l = H (H - D0 for l == skip-1), last layer in = H, out = 1
__host__ __device__ inline int idr_out_dim(const IdrShape& s, int l) {
  if (l == s.n_layers) return 1;
  return (s.skip >= 1 && l == s.skip - 1) ? s.H - s.D0 : s.H;
}
__host__ __device__ inline int idr_in_dim(const IdrShape& s, int l) { return l == 0 ? s.D0 : s.H; }
__host__ __device__ inline int64_t idr_raw_off(const IdrShape& s, int l) {
  int64_t o = 0;
  for (int k = 0; k < l; ++k) o += (int64_t)idr_out_dim(s, k) * idr_in_dim(s, k) + idr_out_dim(s, k);
  return o;
}

// positional-encoding feature `f` (< D0) of a point, its coordinate and derivative
__device__ __forceinline__ void posenc(int f, float x0, float x1, float x2, float& val, int& coord,
                                       float& dval) {
  if (f < 3) {
    coord = f;
    val = f == 0 ? x0 : (f == 1 ? x1 : x2);
    dval = 1.0f;
    return;
  }
  const int m = f - 3, k = m / 6, r = m % 6;
  coord = r % 3;
  const float xc = coord == 0 ? x0 : (coord == 1 ? x1 : x2);
  const float fr = (float)(1 << k);               // 2^k exactly (get_embedder: 2**linspace(0, F-1, F))
  float sn, cs;
  iso_sincos(xc * fr, sn, cs);
  if (r < 3) { val = sn; dval = fr * cs; }
  else { val = cs; dval = -fr * sn; }
}

__device__ __forceinline__ void softplus_b(float z, float beta, float& y, float& dy) {
  // torch.nn.Softplus(beta, threshold=20) and its derivative (sigmoid)
#ifdef IDR_DBG_NOACT      // timing experiment: results wrong by construction
  y = z; dy = beta; return;
#endif
  const float t = z * beta;
  if (t > 20.0f) { y = z; dy = 1.0f; return; }
  const float e = expf(t);
  y = log1pf(e) / beta;
  dy = e / (e + 1.0f);
}

struct IdrArgs {
  float* pts; float* normals; uint8_t* mask; float* sdf_out; float* grad_out;
  const int32_t* idx_in; const int32_t* count_in; int32_t* idx_out; int32_t* count_out;
  const float* packed; float* stash;
  int64_t n;
  IdrShape s;
  float beta, tol;
  int do_move, eval_only;
  // sphere tracing (iso_trace_idr): unit ray directions (n,3); null = Newton / evaluation
  const float* dirs = nullptr;
  float alpha = 1.f, bound = 0.f, tol_valid = 0.f;
  int fwd_only = 0;          // 1: the gradient is not needed
};

}  // namespace

// ---- idr_x16.hip entry points -------------------------------------------------------------
bool idr_x16_supported(int H, int n_layers, int skip, int F);
int64_t idr_x16_floats(int H, int n_layers);            // size of its section of the packed buffer
int64_t idr_x16_stash_floats(int H, int n_layers);
void idr_x16_pack(const float* raw, float* packed, int H, int n_layers, int skip, int F, hipStream_t s);
int idr_x16_launch(const void* idr_args, int64_t n_upper, hipStream_t s);   // const IdrArgs*
