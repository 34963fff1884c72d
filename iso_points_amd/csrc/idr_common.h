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

// softplus(beta, threshold 20) and its derivative (the sigmoid) in ~40 issue slots instead of the
// ~150 of expf + log1pf + two IEEE divisions (the activation is what bounds the split-fp16 kernel:
// PMC, VALU busy 55 % vs MFMA 24 %):
//   u = exp(-|t|)            two-term log2(e) reduction + v_exp_f32 on [-1/2, 1/2] + exact 2^n
//   softplus(t) = max(t, 0) + log1p(u),   log1p(u) = ln(w) * u / (w - 1),  w = fl(1 + u)   (Kahan)
//   sigmoid(t)  = t >= 0 ? 1/w : u/w      reciprocals by v_rcp_f32 + one Newton step
// every piece is good to 1-2 ulp; -DISO_SOFTPLUS_LIBM restores the libm form.
__device__ __forceinline__ float iso_rcp_nr(float d) {
  float r = __builtin_amdgcn_rcpf(d);
  return __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
}

__device__ __forceinline__ void softplus_b(float z, float beta, float& y, float& dy) {
  // torch.nn.Softplus(beta, threshold=20) and its derivative (sigmoid)
#ifdef IDR_DBG_NOACT      // timing experiment: results wrong by construction
  y = z; dy = beta; return;
#endif
#ifdef ISO_SOFTPLUS_LIBM
  const float t = z * beta;
  if (t > 20.0f) { y = z; dy = 1.0f; return; }
  const float e = expf(t);
  y = log1pf(e) / beta;
  dy = e / (e + 1.0f);
#else
  const float t = z * beta;
  const float inv_beta = 1.0f / beta;                             // beta is uniform: one division per wave, hoisted
  const float at = __builtin_fabsf(t);
  const float L2E_hi = 1.44269502162933349609375f, L2E_lo = 1.925963033500011e-08f;
  const float n = __builtin_rintf(-at * L2E_hi);
  float f = __builtin_fmaf(-at, L2E_hi, -n);
  f = __builtin_fmaf(-at, L2E_lo, f);
  const float u = __builtin_amdgcn_exp2f(f) * __builtin_amdgcn_exp2f(__builtin_fmaxf(n, -126.0f));   // exp(-|t|) in (0, 1]
  const float w = 1.0f + u;
  const float d = w - 1.0f;                                        // exact
  const float rw = iso_rcp_nr(w);
  const float lnw = __builtin_amdgcn_logf(w) * 0.693147180559945309f;
  const float rd = __builtin_amdgcn_rcpf(d);
  float q = u * rd;                                                // u / (w - 1), one Newton step
  q = __builtin_fmaf(__builtin_fmaf(-q, d, u), rd, q);
  const float l1p = d == 0.0f ? u : lnw * q;
  const float sp = (__builtin_fmaxf(t, 0.0f) + l1p) * inv_beta;
  const float sg = t >= 0.0f ? rw : u * rw;
  const bool lin = t > 20.0f;
  y = lin ? z : sp;
  dy = lin ? 1.0f : sg;
#endif
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
  // k_idr_step_x16: a counter, ZERO when the launch starts, the workgroups draw their next tile from (null: every
  // gridDim-th tile -- the launch then ends with the slowest XCD, see x3_step_body in siren_x3.hip)
  int32_t* tile_ctr = nullptr;
};

}  // namespace

// ---- idr_x16.hip entry points -------------------------------------------------------------
bool idr_x16_supported(int H, int n_layers, int skip, int F);
int64_t idr_x16_floats(int H, int n_layers);            // size of its section of the packed buffer
int64_t idr_x16_stash_floats(int H, int n_layers);
void idr_x16_pack(const float* raw, float* packed, int H, int n_layers, int skip, int F, hipStream_t s);
int idr_x16_launch(const void* idr_args, int64_t n_upper, hipStream_t s);   // const IdrArgs*
