// Split-operand MFMA building blocks shared by the fused-MLP step kernels (siren_x3.hip, idr_x16.hip):
// operand cut (two-way fp16 under a power-of-two scale), the weight-fragment pipeline
// and the layer GEMM.  Feature order of all per-feature data: x3_feat() in siren_common.h.
#pragma once
#include "siren_common.h"
#include "mlp_common.h"

namespace {


typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// f32 products from fp16 MFMAs: an f32 number cut into TWO fp16 numbers (11 + 11 significant bits, round-to-nearest
// at each cut: unit roundoff 2^-11 each) is represented to 2^-22 relative in the worst case, and W.x is formed from
// THREE partial products (W_h x_l + W_l x_h + W_h x_h; the dropped W_l x_l is <= 2^-22 relative).  Worst-case bounds,
// that is: two bits above the f32 unit roundoff 2^-24; the errors are not correlated over the 256 terms of a dot
// product, and against float64 the kernel's gradient error is the same as torch's f32 autograd (0.89e-6 vs 0.81e-6
// mean, 4.9e-6 vs 4.6e-6 max: tests/test_projection_gpu.py::test_siren_grad_accuracy_vs_float64, the evidence this
// rests on).  fp16 has a 5-bit exponent, so both operands are brought into range by exact power-of-two scales:
// activations (|sin| <= 1) by 2^12 (the low part then stays a normal number down to contributions of 2^-26), the
// weights of layer l by 2^s_l with max|2^s_l W| in [512, 1024); the bias enters the accumulator scaled by
// 2^(s_l + 12) and the scale is taken out again, exactly, in the multiplication by omega that follows.
// The adjoint of the reverse sweep has no a-priori range, so every POINT carries its own power-of-two scale.
// max_f |a_l[f][p]| is exchanged between the waves through LDS (it rides on the barrier that already ends the
// stage), and the scale of the next adjoint is taken from the rigorous bound
//   |a_{l-1}[f][p]| <= omega * (max_f sum_k |W_l[k][f]|) * max_k |a_l[k][p]|
// (column sums prepared at pack time), scaled to below 2^14: no overflow whatever the weights, and since every
// element keeps 22 significant bits of its own, the ~20x slack of the bound only moves the subnormal floor
// (elements below 2^-13 of the largest one) -- contributions at the f32 rounding level of the dot product.
// (An exact three-way bf16 cut with six products was the first form of these kernels: tools/experiments/, history.)
// parts per (K-step, point tile) entry of the activation buffer in LDS
constexpr int kAP = 2;

// 2^E with bound * 2^E in [2^13, 2^14) (1 for a zero / non-finite bound)
__device__ __forceinline__ float x3_scale_for(float bound) {
  if (!(bound > 0.f) || !(bound < 3.0e38f)) return 1.0f;
  int e;
  (void)frexpf(bound, &e);
  int k = 14 - e;
  k = k > 120 ? 120 : (k < -120 ? -120 : k);
  return ldexpf(1.0f, k);
}
constexpr float kActScale = 4096.0f;             // 2^12

// two-way fp16 cut of scale * v, step-major over the four pairs (a packed op whose result feeds the next
// instruction costs a wait state, mlp_common.h); scale is a power of two.
// The low part l = f16(x - f32(h)) is ONE instruction per value: v_fma_mixlo_f16 / v_fma_mixhi_f16 compute 1.0 * x - h in
// f32 from the fp16 half of h in place (the difference is exact) and round it to fp16 into the low / high half of the
// destination -- bit for bit what the conversion back, the subtraction and the second conversion gave (three half-rate
// instructions per value before; tools/probes/mix_split.hip: no mismatch in 2^20 values), 16 instead of 24 instructions per cut.
#ifndef X3_MIX_SPLIT
#define X3_MIX_SPLIT 1
#endif
__device__ __forceinline__ unsigned x3_low_half_pair(float x0, float x1, unsigned h) {
  unsigned l;
  asm("v_fma_mixlo_f16 %0, 1.0, %1, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "v"(h));
  asm("v_fma_mixhi_f16 %0, 1.0, %1, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "v"(h));
  return l;
}
__device__ __forceinline__ void split8_f16(const float (&v)[8], u32x4& hi, u32x4& lo, float scale = kActScale) {
  f32x2 x[4];
  f16x2 h[4];
  const f32x2 sc = {scale, scale};
  ISO_X4(x[p] = ((f32x2){v[2 * p], v[2 * p + 1]}) * sc);
  ISO_X4(h[p] = __builtin_convertvector(x[p], f16x2));
#if X3_MIX_SPLIT
  unsigned l[4];
  ISO_X4(asm("v_fma_mixlo_f16 %0, 1.0, %1, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l[p]) : "v"(x[p].x), "v"(__builtin_bit_cast(unsigned, h[p]))));
  ISO_X4(asm("v_fma_mixhi_f16 %0, 1.0, %1, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l[p]) : "v"(x[p].y), "v"(__builtin_bit_cast(unsigned, h[p]))));
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    hi[d] = __builtin_bit_cast(unsigned, h[d]);
    lo[d] = l[d];
  }
#else
  f32x2 f[4];
  f16x2 l[4];
  ISO_X4(f[p] = __builtin_convertvector(h[p], f32x2));
  ISO_X4(x[p] = x[p] - f[p]);
  ISO_X4(l[p] = __builtin_convertvector(x[p], f16x2));
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    hi[d] = __builtin_bit_cast(unsigned, h[d]);
    lo[d] = __builtin_bit_cast(unsigned, l[d]);
  }
#endif
}

// Timing experiments only (tools/build_variant.sh): -DX3_DBG_NOSINCOS / NOMMA / NOSTASH knock out
// one ingredient each; results are then wrong by construction.
__device__ __forceinline__ f32x4 as_f32x4(const u32x4& v) { return __builtin_bit_cast(f32x4, v); }
__device__ __forceinline__ u32x4 as_u32x4(const f32x4& v) { return __builtin_bit_cast(u32x4, v); }

// ---- the layer GEMM --------------------------------------------------------------------------
// acc[t][n] (32 features x 32 points each) += W[tiles of this wave] . act, K-steps 0..NS-1.
// imgw = image + (TW*w*3)*64 + lane ;  actl = act + lane ;  bias_h = bias_k (K-order, wave-uniform)

// Weight-fragment pipeline: 4 register sets, requested kAD = 3 K-steps ahead (an L2 hit under
// load takes longer than one K-step of MFMAs).  The fragments of the first kAD K-steps are
// expected in A[0..kAD-1] on entry (requested by the previous stage, so no L2 latency is exposed
// after a barrier); on exit A[0..kAD-1] hold the first fragments of the next GEMM stage (image
// next_imgw, K-steps next_s..).  Activation fragments (LDS) run one K-step ahead.
#ifndef X3_KAD
#define X3_KAD 3
#endif
constexpr int kAD = X3_KAD;

// imgw is WAVE-UNIFORM (no lane term): the loads take the scalar-base + 32-bit lane-offset form,
// so no 64-bit per-lane address registers are needed.
// PARTS = 2: the split-fp16 image (imgw points TW*w*PARTS*64 into it)
// IP: `const u32x4*`, or the same with an explicit global address space (siren_pp.hip)
typedef const __attribute__((address_space(1))) u32x4* gimg_t;
template <class IP> struct x3_byte_ptr { typedef const char* type; };
template <> struct x3_byte_ptr<gimg_t> { typedef const __attribute__((address_space(1))) char* type; };

template <int TW, int NTO, int PARTS, class IP>
__device__ __forceinline__ void x3_load_a(u32x4 (&Ar)[TW][3], IP imgw, int s, unsigned lane) {
  typedef typename x3_byte_ptr<IP>::type BP;
  const BP p = (BP)(imgw + (int64_t)s * (NTO * PARTS * 64));
  const unsigned lane_off = lane * 16u;      // 32-bit byte offset: keeps the scalar-base form
#pragma unroll
  for (int t = 0; t < TW; ++t) {
    const BP pt = p + t * (PARTS * 1024);   // scalar; the parts are immediate offsets
#pragma unroll
    for (int c = 0; c < PARTS; ++c) Ar[t][c] = *(IP)(pt + lane_off + c * 1024);
  }
}

template <int TW, int NTO, int PARTS, class IP>
__device__ __forceinline__ void x3_prefetch_a(u32x4 (&A)[4][TW][3], IP imgw, int s, unsigned lane) {
#pragma unroll
  for (int d = 0; d < kAD; ++d) x3_load_a<TW, NTO, PARTS>(A[d], imgw, s + d, lane);
}

enum { kAccumulate = 0, kZero = 1, kBias = 2 };

// Keeps an operand set live (a register use the compiler cannot remove or move above the preceding
// sched_barrier).  The hardware does NOT protect the source registers of an MFMA in flight against
// a later LDS / global load that writes them: when the operand requests of the next K-steps are
// scheduled between the MFMAs, the allocator would otherwise hand a just-read fragment register to
// the very next load (seen: results changing from run to run).
template <int N, int PARTS>
__device__ __forceinline__ void x3_keep_alive(const u32x4 (&X)[N][3]) {
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int c = 0; c < PARTS; ++c) asm volatile("" ::"v"(X[i][c]));
}

// PARTS / NEXT_PARTS: operand parts of this stage / of the stage whose first fragments are requested at the end (2).
// bias_scale multiplies the bias (the accumulator scale of a split-fp16 stage; 1 otherwise).
struct x3_no_hook { __device__ __forceinline__ void operator()() const {} };

// `hook` runs once, in front of the first request for the NEXT stage's fragments (K-step KS - kAD): memory requests
// return in order per wave, so a request for data that is far away (the derivative stash) belongs behind the last
// fragment this stage still waits for and in front of those nobody needs before the next stage.
template <int TW, int NB, int NTO, int KS, int INIT, bool IL, int PARTS = 2, int NEXT_PARTS = 2, class IP = const u32x4*,
          class Hook = x3_no_hook>
__device__ __forceinline__ void gemm_x3(IP imgw, const float* __restrict__ bias_h,
                                        const u32x4* actl, f32x16 (&acc)[TW][NB], int w, int s0,
                                        u32x4 (&A)[4][TW][3], IP next_imgw, int next_s,
                                        unsigned lane, float bias_scale = 1.0f,
                                        const float* bias_scale_n = nullptr,     // per point tile, on top of bias_scale
                                        Hook&& hook = Hook()) {
  static_assert(KS % 4 == 0, "K-steps are processed in groups of four");
  if constexpr (INIT != kAccumulate) {
#pragma unroll
    for (int t = 0; t < TW; ++t) {
      f32x16 init;
      if constexpr (INIT == kBias) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const float* bp = bias_h + (2 * (TW * w + t) + p) * 16 + 8 * (lane >> 5);   // uniform base + lane-half offset
          const f32x4 lo = *reinterpret_cast<const f32x4*>(bp);
          const f32x4 hi = *reinterpret_cast<const f32x4*>(bp + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { init[8 * p + e] = lo[e] * bias_scale; init[8 * p + 4 + e] = hi[e] * bias_scale; }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) init[r] = 0.f;
      }
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        if (INIT == kBias && bias_scale_n) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][n][r] = init[r] * bias_scale_n[n];
        } else {
          acc[t][n] = init;
        }
      }
    }
  }
  u32x4 B[2][NB][3];
  auto ldB = [&](u32x4 (&Br)[NB][3], int s) {
    const u32x4* p = actl + s * (NB * kAP * 64);
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int c = 0; c < PARTS; ++c) Br[n][c] = p[(n * kAP + c) * 64];
  };
  auto mma = [&](const u32x4 (&Ar)[TW][3], const u32x4 (&Br)[NB][3]) {
#ifdef X3_DBG_NOMMA
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
      for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[t][n][c] += __builtin_bit_cast(f32x4, Ar[t][c]).x * __builtin_bit_cast(f32x4, Br[n][c]).y;
    return;
#endif
    static_assert(PARTS == 2, "two fp16 parts per operand");
    // W_l x_h + W_h x_l + W_h x_h
    constexpr int QA[3] = {1, 0, 0};
    constexpr int QB[3] = {0, 1, 0};
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int n = 0; n < NB; ++n)
          acc[t][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, Ar[t][QA[q]]),
                                                             __builtin_bit_cast(f16x8, Br[n][QB[q]]),
                                                             acc[t][n], 0, 0, 0);
  };
  ldB(B[0], s0);
#ifndef X3_GEMM_PRIO
#define X3_GEMM_PRIO 0
#endif
#ifndef X3_STATIC_PRIO
  __builtin_amdgcn_s_setprio(X3_GEMM_PRIO);
#endif
#ifdef X3_KROLLED
#pragma unroll 1
#endif
  for (int i = 0; i < KS; i += 4) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int k = i + jj;                       // K-step of this stage being multiplied
      // set (jj+3)%4 was consumed one K-step ago: refill it with K-step k+3 (or the next stage's)
      if (k + kAD < KS) x3_load_a<TW, NTO, PARTS>(A[(jj + kAD) & 3], imgw, s0 + k + kAD, lane);
      else {
        if (k + kAD == KS) hook();
        x3_load_a<TW, NTO, NEXT_PARTS>(A[(jj + kAD) & 3], next_imgw, next_s + (k + kAD - KS), lane);
      }
      if (k + 1 < KS) ldB(B[(jj + 1) & 1], s0 + k + 1);
#ifndef X3_INTERLEAVE_LOADS
#define X3_INTERLEAVE_LOADS 1
#endif
      if constexpr (IL && X3_INTERLEAVE_LOADS) {
      // the operand requests for the coming K-steps ride in the shadow of this K-step's MFMAs (one
      // memory instruction behind each of the first MFMAs) instead of draining the matrix pipe
      // between K-steps
      mma(A[jj], B[jj & 1]);
#ifndef X3_IL_DS
#define X3_IL_DS 1
#endif
#ifndef X3_IL_VM
#define X3_IL_VM 1
#endif
#if X3_IL_DS
#pragma unroll
      for (int g = 0; g < NB * PARTS; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
#endif
#if X3_IL_VM
#pragma unroll
      for (int g = 0; g < TW * (PARTS > NEXT_PARTS ? PARTS : NEXT_PARTS) && g < TW * NB * 3 - NB * PARTS; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
#ifndef X3_NO_KEEPALIVE
      x3_keep_alive<TW, PARTS>(A[jj]);
      x3_keep_alive<NB, PARTS>(B[jj & 1]);
#endif
#ifdef X3_IL_GUARD_NOP      // experiment: wait states between the end of a K-step and the first MFMA of the next
      asm volatile("s_nop %0" ::"n"(X3_IL_GUARD_NOP));
      __builtin_amdgcn_sched_barrier(0);
#endif
      } else {
      __builtin_amdgcn_sched_barrier(0);
      mma(A[jj], B[jj & 1]);
      __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  // the next stage starts again at set 0: with KS % 4 == 0 the rotation is already aligned
#ifndef X3_STATIC_PRIO
  __builtin_amdgcn_s_setprio(0);
#endif
}

}  // namespace
