// Shared pieces of the fused-MLP kernels (siren.hip, idr.hip): accurate sin/cos and the
// LDS-staged f32 MFMA layer pass.  See siren.hip for the data-layout story.
#pragma once
#include "iso_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- sin/cos ---------------------------------------------------------------
// 3-term Cody-Waite reduction by pi/2 with FMA, cephes-style minimax kernels on
// [-pi/4, pi/4].  Max error ~1 ulp for |x| < 1e4 (tests/test_siren.py checks it
// against float64); larger arguments take the slow libm path.
__device__ __forceinline__ void iso_sincos_core(float x, float& s, float& c) {
  const float two_over_pi = 0.636619772367581343f;
  const float p1 = 1.57079637050628662109375f;        // fl(pi/2)
  const float p2 = -4.37113882867379288655e-8f;       // fl(pi/2 - p1)
  const float p3 = -1.71512451000588187280e-15f;      // fl(pi/2 - p1 - p2)
  float n = rintf(x * two_over_pi);
  float r = __builtin_fmaf(-n, p1, x);
  r = __builtin_fmaf(-n, p2, r);
  r = __builtin_fmaf(-n, p3, r);
  float r2 = r * r;
  float ps = __builtin_fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
  ps = __builtin_fmaf(ps, r2, -1.6666654611e-1f);
  float sr = __builtin_fmaf(ps * r2, r, r);
  float pc = __builtin_fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
  pc = __builtin_fmaf(pc, r2, 4.166664568298827e-2f);
  float cr = __builtin_fmaf(pc, r2 * r2, __builtin_fmaf(-0.5f, r2, 1.0f));
  int q = (int)n;
  float ss = (q & 1) ? cr : sr;
  float cc = (q & 1) ? sr : cr;
  s = (q & 2) ? -ss : ss;
  c = ((q + 1) & 2) ? -cc : cc;
}

__device__ __forceinline__ void iso_sincos(float x, float& s, float& c) {
  if (!(fabsf(x) < 1.0e4f)) {
    sincosf(x, &s, &c);
    return;
  }
  iso_sincos_core(x, s, c);
}

// Two arguments per instruction: on gfx950 a plain wave64 VALU op issues in 4 cycles and the
// packed f32 forms (v_pk_fma_f32 / v_pk_mul_f32) do two values in the same slot, so the
// polynomial part of sin/cos is written on float2.  Same constants, same operation order per
// element as iso_sincos_core (bitwise identical results).
typedef float iso_f32x2 __attribute__((ext_vector_type(2)));
typedef int iso_i32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void iso_sincos_core2(iso_f32x2 x, iso_f32x2& s, iso_f32x2& c) {
  const iso_f32x2 two_over_pi = {0.636619772367581343f, 0.636619772367581343f};
  const iso_f32x2 p1 = {1.57079637050628662109375f, 1.57079637050628662109375f};
  const iso_f32x2 p2 = {-4.37113882867379288655e-8f, -4.37113882867379288655e-8f};
  const iso_f32x2 p3 = {-1.71512451000588187280e-15f, -1.71512451000588187280e-15f};
  auto splat = [](float v) { return (iso_f32x2){v, v}; };
  iso_f32x2 t = x * two_over_pi;
  iso_f32x2 n = {rintf(t.x), rintf(t.y)};
  iso_f32x2 r = __builtin_elementwise_fma(-n, p1, x);
  r = __builtin_elementwise_fma(-n, p2, r);
  r = __builtin_elementwise_fma(-n, p3, r);
  iso_f32x2 r2 = r * r;
  iso_f32x2 ps = __builtin_elementwise_fma(r2, splat(-1.9515295891e-4f), splat(8.3321608736e-3f));
  ps = __builtin_elementwise_fma(ps, r2, splat(-1.6666654611e-1f));
  iso_f32x2 sr = __builtin_elementwise_fma(ps * r2, r, r);
  iso_f32x2 pc = __builtin_elementwise_fma(r2, splat(2.443315711809948e-5f), splat(-1.388731625493765e-3f));
  pc = __builtin_elementwise_fma(pc, r2, splat(4.166664568298827e-2f));
  iso_f32x2 cr = __builtin_elementwise_fma(pc, r2 * r2, __builtin_elementwise_fma(splat(-0.5f), r2, splat(1.0f)));
#ifdef ISO_SINCOS_INT_QUADRANT     // previous form (integer selects), kept for A/B timing
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int q = (int)n[i];
    const float ss = (q & 1) ? cr[i] : sr[i];
    const float cc = (q & 1) ? sr[i] : cr[i];
    s[i] = __uint_as_float(__float_as_uint(ss) ^ (((unsigned)q & 2u) << 30));
    c[i] = __uint_as_float(__float_as_uint(cc) ^ (((unsigned)(q + 1) & 2u) << 30));
  }
  return;
#endif
  // quadrant: rotate (sr, cr) by q*90 degrees with exact 0/+-1 factors, all on packed f32 ops
  //   m = n - 4*rint(n/4) in {-2..2};  a = cos(m pi/2) = 1 - |m|;  b = sin(m pi/2) = m*(1 + a)
  //   sin x = a*sr + b*cr ;  cos x = a*cr - b*sr      (one term of each sum is an exact zero)
  const iso_f32x2 q4 = n * splat(0.25f);
  const iso_f32x2 u = {rintf(q4.x), rintf(q4.y)};
  const iso_f32x2 m = __builtin_elementwise_fma(splat(-4.0f), u, n);
  const iso_f32x2 am = {__builtin_fabsf(m.x), __builtin_fabsf(m.y)};
  const iso_f32x2 a = splat(1.0f) - am;
  const iso_f32x2 b = __builtin_elementwise_fma(m, a, m);
  s = __builtin_elementwise_fma(a, sr, b * cr);
  c = __builtin_elementwise_fma(a, cr, -(b * sr));
}

// Four pairs at once, STEP-major: a packed f32 op whose result feeds the next instruction costs a
// wait state on gfx950 (the compiler pads the chain of one pair with s_nop: 40 per 8 values, 20 %
// of the issue slots of the activation stage), so every step is written across the four
// independent pairs.  Per element exactly the operations of iso_sincos_core2, same order.
#ifndef ISO_SINCOS_STEP_BARRIER
#define ISO_SINCOS_STEP_BARRIER 1
#endif
#if ISO_SINCOS_STEP_BARRIER
#define ISO_STEP() __builtin_amdgcn_sched_barrier(0)
#else
#define ISO_STEP() do {} while (0)
#endif
#define ISO_X4(expr) _Pragma("unroll") for (int p = 0; p < 4; ++p) { expr; } ISO_STEP()
#define ISO_XN(expr) _Pragma("unroll") for (int p = 0; p < NP; ++p) { expr; } ISO_STEP()
template <int NP>
__device__ __forceinline__ void iso_sincos_core2xN(const iso_f32x2* x, iso_f32x2* s, iso_f32x2* c) {
  auto splat = [](float v) { return (iso_f32x2){v, v}; };
  const iso_f32x2 two_over_pi = splat(0.636619772367581343f);
  const iso_f32x2 p1 = splat(1.57079637050628662109375f);
  const iso_f32x2 p2 = splat(-4.37113882867379288655e-8f);
  const iso_f32x2 p3 = splat(-1.71512451000588187280e-15f);
  iso_f32x2 n[NP], r[NP], r2[NP], ps[NP], sr[NP], pc[NP], cr[NP], t[NP], u[NP];
  ISO_XN(t[p] = x[p] * two_over_pi);
  ISO_XN(n[p] = ((iso_f32x2){rintf(t[p].x), rintf(t[p].y)}));
  ISO_XN(r[p] = __builtin_elementwise_fma(-n[p], p1, x[p]));
  ISO_XN(r[p] = __builtin_elementwise_fma(-n[p], p2, r[p]));
  ISO_XN(r[p] = __builtin_elementwise_fma(-n[p], p3, r[p]));
  ISO_XN(r2[p] = r[p] * r[p]);
  ISO_XN(ps[p] = __builtin_elementwise_fma(r2[p], splat(-1.9515295891e-4f), splat(8.3321608736e-3f)));
  ISO_XN(pc[p] = __builtin_elementwise_fma(r2[p], splat(2.443315711809948e-5f), splat(-1.388731625493765e-3f)));
  ISO_XN(ps[p] = __builtin_elementwise_fma(ps[p], r2[p], splat(-1.6666654611e-1f)));
  ISO_XN(pc[p] = __builtin_elementwise_fma(pc[p], r2[p], splat(4.166664568298827e-2f)));
  ISO_XN(t[p] = ps[p] * r2[p]);
  ISO_XN(u[p] = __builtin_elementwise_fma(splat(-0.5f), r2[p], splat(1.0f)));
  ISO_XN(sr[p] = __builtin_elementwise_fma(t[p], r[p], r[p]));
  ISO_XN(t[p] = r2[p] * r2[p]);
  ISO_XN(cr[p] = __builtin_elementwise_fma(pc[p], t[p], u[p]));
  // quadrant rotation (see iso_sincos_core2)
  iso_f32x2 m[NP], a[NP], b[NP];
  ISO_XN(t[p] = n[p] * splat(0.25f));
  ISO_XN(u[p] = ((iso_f32x2){rintf(t[p].x), rintf(t[p].y)}));
  ISO_XN(m[p] = __builtin_elementwise_fma(splat(-4.0f), u[p], n[p]));
  ISO_XN(a[p] = splat(1.0f) - ((iso_f32x2){__builtin_fabsf(m[p].x), __builtin_fabsf(m[p].y)}));
  ISO_XN(b[p] = __builtin_elementwise_fma(m[p], a[p], m[p]));
  ISO_XN(t[p] = b[p] * cr[p]);
  ISO_XN(u[p] = -(b[p] * sr[p]));
  ISO_XN(s[p] = __builtin_elementwise_fma(a[p], sr[p], t[p]));
  ISO_XN(c[p] = __builtin_elementwise_fma(a[p], cr[p], u[p]));
}

#ifndef ISO_SINCOS_HW
#define ISO_SINCOS_HW 1
#endif
// Eight arguments at once: the polynomial path for all, then ONE wave-uniform branch for the
// (practically never taken) large-argument fix-up.  s = sin(w_in*z), c = w*cos(w_in*z)
// (w_in = w except where z carries a power-of-two scale that w_in takes out again).
__device__ __forceinline__ void iso_sin_wcos8(float w_in, float w, const float (&z)[8], float (&s)[8], float (&c)[8]) {
  const iso_f32x2 w2 = {w, w}, wi2 = {w_in, w_in};
  iso_f32x2 x[4], s2[4], c2[4];
  ISO_X4(x[p] = ((iso_f32x2){z[2 * p], z[2 * p + 1]}) * wi2);
#if ISO_SINCOS_HW
  // v_sin_f32 / v_cos_f32 take revolutions and are good to 1.25e-7 absolute on [-1/2, 1/2]
  // (tools/probes/hw_sincos_accuracy.hip: mean error 2.8e-8, a correctly rounded result has 1.5e-8), so
  // only the reduction is done in software: f = x/(2 pi) - rint(x/(2 pi)) with a two-term 1/(2 pi)
  // (the product x * hi is exact inside the fma).  7 issue slots per value instead of 15.5.
  {
    const iso_f32x2 hi = {0.159154936671257019043f, 0.159154936671257019043f};
    const iso_f32x2 lo = {6.4206383167e-9f, 6.4206383167e-9f};
    iso_f32x2 t[4], n[4], f[4];
    ISO_X4(t[p] = x[p] * hi);
    ISO_X4(n[p] = ((iso_f32x2){rintf(t[p].x), rintf(t[p].y)}));
    ISO_X4(f[p] = __builtin_elementwise_fma(x[p], hi, -n[p]));
    ISO_X4(f[p] = __builtin_elementwise_fma(x[p], lo, f[p]));
    ISO_X4(s2[p] = ((iso_f32x2){__builtin_amdgcn_sinf(f[p].x), __builtin_amdgcn_sinf(f[p].y)}));
    ISO_X4(c2[p] = ((iso_f32x2){__builtin_amdgcn_cosf(f[p].x), __builtin_amdgcn_cosf(f[p].y)}));
  }
#else
  // ISO_SINCOS_WIDTH pairs step-major at a time: 4 removes every wait state, 2 keeps the register
  // footprint of the temporaries at half (two chains already separate dependent packed ops)
#ifndef ISO_SINCOS_WIDTH
#define ISO_SINCOS_WIDTH 2
#endif
#pragma unroll
  for (int q = 0; q < 4; q += ISO_SINCOS_WIDTH) iso_sincos_core2xN<ISO_SINCOS_WIDTH>(x + q, s2 + q, c2 + q);
#endif
  ISO_X4(c2[p] = c2[p] * w2);
  float amax = 0.f;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    s[2 * p] = s2[p].x; s[2 * p + 1] = s2[p].y;
    c[2 * p] = c2[p].x; c[2 * p + 1] = c2[p].y;
    amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(x[p].x), __builtin_fabsf(x[p].y)));   // one v_max3
  }
  const bool big = !(amax < 1.0e4f);                 // also true for NaN arguments
  if (__builtin_expect(__any(big), 0)) {
    // |x| >= 1e4 (never seen with trained SIRENs): libm's Payne-Hanek path, one rolled copy
    float xs[8], ss[8], cs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { xs[e] = w_in * z[e]; ss[e] = s[e]; cs[e] = c[e]; }
#pragma unroll 1
    for (int e = 0; e < 8; ++e) {
      float x = xs[0], s0, c0;
      sincosf(x, &s0, &c0);
      const bool fix = !(fabsf(x) < 1.0e4f);
      const float sn = fix ? s0 : ss[0], cn = fix ? w * c0 : cs[0];
      // rotate so that the loop body only ever touches element 0 (static register indices)
#pragma unroll
      for (int i = 0; i < 7; ++i) { xs[i] = xs[i + 1]; ss[i] = ss[i + 1]; cs[i] = cs[i + 1]; }
      xs[7] = x; ss[7] = sn; cs[7] = cn;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = ss[e]; c[e] = cs[e]; }
  }
}

// c = w*cos(w_in*z) alone, bit for bit the c of iso_sin_wcos8 (same reduction, same instructions): the reverse sweep
// of the SIREN step recomputes layer 0's derivative from the point instead of reading it back from a stash.
__device__ __forceinline__ void iso_wcos8(float w_in, float w, const float (&z)[8], float (&c)[8]) {
#if ISO_SINCOS_HW
  const iso_f32x2 w2 = {w, w}, wi2 = {w_in, w_in};
  iso_f32x2 x[4], c2[4];
  ISO_X4(x[p] = ((iso_f32x2){z[2 * p], z[2 * p + 1]}) * wi2);
  {
    const iso_f32x2 hi = {0.159154936671257019043f, 0.159154936671257019043f};
    const iso_f32x2 lo = {6.4206383167e-9f, 6.4206383167e-9f};
    iso_f32x2 t[4], n[4], f[4];
    ISO_X4(t[p] = x[p] * hi);
    ISO_X4(n[p] = ((iso_f32x2){rintf(t[p].x), rintf(t[p].y)}));
    ISO_X4(f[p] = __builtin_elementwise_fma(x[p], hi, -n[p]));
    ISO_X4(f[p] = __builtin_elementwise_fma(x[p], lo, f[p]));
    ISO_X4(c2[p] = ((iso_f32x2){__builtin_amdgcn_cosf(f[p].x), __builtin_amdgcn_cosf(f[p].y)}));
  }
  ISO_X4(c2[p] = c2[p] * w2);
  float amax = 0.f;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    c[2 * p] = c2[p].x; c[2 * p + 1] = c2[p].y;
    amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(x[p].x), __builtin_fabsf(x[p].y)));
  }
#ifndef ISO_WCOS_NOFIX
  if (__builtin_expect(__any(!(amax < 1.0e4f)), 0)) {
    float s[8];
    iso_sin_wcos8(w_in, w, z, s, c);          // the large-argument path: rare, take the full routine
  }
#endif
#else
  float s[8];
  iso_sin_wcos8(w_in, w, z, s, c);
#endif
}

template <int NT, bool HAS_BIAS>
__device__ __forceinline__ void gemm_pass(const float* __restrict__ img,
                                          const float* __restrict__ bias,
                                          const float* __restrict__ hL,
                                          float* __restrict__ wbuf, f32x4 (&acc)[NT],
                                          int lane, int g, int nq = NT) {
  // One pass = nq q-chunks (nq = NT for a square layer); each q-chunk is staged in two halves of TC = NT/2 tiles so
  // that the LDS stage is 2 x (NT/2) KiB and two workgroups fit on a CU.  All A
  // fragments of a half are requested up front (TC ds_read_b128 in flight) and the MFMAs
  // consume them as they land.
  constexpr int TC = NT / 2;              // tiles per staged half
  constexpr int CH = TC * 256;            // floats per half-chunk
  constexpr int NV = CH / 4;              // float4 per half-chunk
  constexpr int PER = (NV + 255) / 256;   // float4 per thread per half-chunk
  static_assert(NT % 2 == 0, "NT must be even");
  const int tid = threadIdx.x;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if constexpr (HAS_BIAS) {
      acc[t] = *reinterpret_cast<const f32x4*>(bias + 16 * t + 4 * g);
    } else {
      acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  constexpr bool FULL = (NV % 256) == 0;  // every thread moves PER float4 (no tail guard)
  // Two register sets: the global loads of half-chunk c+2 are issued while chunk c is
  // multiplied and chunk c+1 (loaded one stage earlier) is written to the other LDS buffer,
  // so a load has two MFMA stages (~2k cycles) to land before it is needed.
  f32x4 sx[PER], sy[PER];
  auto gload = [&](f32x4 (&r)[PER], int c) {
    const f32x4* src = reinterpret_cast<const f32x4*>(img + (int64_t)c * CH);
#pragma unroll
    for (int k = 0; k < PER; ++k)
      if (FULL || tid + 256 * k < NV) r[k] = src[tid + 256 * k];
  };
  auto lwrite = [&](const f32x4 (&r)[PER], int buf) {
    f32x4* dst = reinterpret_cast<f32x4*>(wbuf + buf * CH);
#pragma unroll
    for (int k = 0; k < PER; ++k)
      if (FULL || tid + 256 * k < NV) dst[tid + 256 * k] = r[k];
  };
  auto stage = [&](int half, const f32x4& b4) {
    const f32x4* wa = reinterpret_cast<const f32x4*>(wbuf + half * CH);
    f32x4 a4[TC];
#pragma unroll
    for (int t = 0; t < TC; ++t) a4[t] = wa[t * 64 + lane];
    // keep the TC reads ahead of the MFMA block: the scheduler otherwise sinks each read
    // next to its consumer (2 in flight, full LDS latency exposed every 8 MFMAs)
    __builtin_amdgcn_sched_barrier(0);
    // k-major: consecutive MFMAs hit different accumulators (32 cycles issue, 40 dependent latency)
#pragma unroll
    for (int t = 0; t < TC; ++t) acc[half * TC + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[t].x, b4.x, acc[half * TC + t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < TC; ++t) acc[half * TC + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[t].y, b4.y, acc[half * TC + t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < TC; ++t) acc[half * TC + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[t].z, b4.z, acc[half * TC + t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < TC; ++t) acc[half * TC + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[t].w, b4.w, acc[half * TC + t], 0, 0, 0);
    // ... and the LDS write + barrier of the next chunk BEHIND it (hoisted, they make the wave
    // drain all of its reads with only a few MFMAs in flight)
    __builtin_amdgcn_sched_barrier(0);
  };
  gload(sx, 0);
  lwrite(sx, 0);
  gload(sx, 1);
  __syncthreads();
  // the wave in its MFMA phase outranks a co-resident wave that is in a VALU (sin/cos) phase:
  // +2 % measured (98.2 -> 100.2 TFLOP/s); the two workgroups of a CU stay better interleaved
  __builtin_amdgcn_s_setprio(1);
  for (int q = 0; q < nq; ++q) {
    const f32x4 b4 = reinterpret_cast<const f32x4*>(hL)[q * 64 + lane];
    // ---- half 0: chunk 2q in buffer 0; sx holds chunk 2q+1
    if (2 * q + 2 < 2 * nq) gload(sy, 2 * q + 2);
    stage(0, b4);
    lwrite(sx, 1);
    __syncthreads();
    // ---- half 1: chunk 2q+1 in buffer 1; sy holds chunk 2q+2
    if (2 * q + 3 < 2 * nq) gload(sx, 2 * q + 3);
    stage(1, b4);
    if (2 * q + 2 < 2 * nq) lwrite(sy, 0);
    __syncthreads();
  }
  __builtin_amdgcn_s_setprio(0);
}



// Register-pipelined form of gemm_pass (used by idr.hip: one workgroup per CU, so nothing else hides
// an LDS round trip).  Same staging protocol -- half q-chunks through two LDS buffers, one barrier
// per half -- but the A fragments of half-chunk c+1 are read from LDS into a second register set
// WHILE half-chunk c is multiplied (they were published by the previous barrier), the global loads
// of chunk c+3 and the LDS store of chunk c+2 ride behind the same MFMAs (sched_group_barrier), and
// the four k-steps of a tile are issued k-major so that consecutive MFMAs never hit the same
// accumulator (v_mfma_f32_16x16x4_f32: 32 cycles issue but 40 cycles dependent latency).
template <int NT, bool HAS_BIAS>
__device__ __forceinline__ void gemm_pass_pipe(const float* __restrict__ img,
                                               const float* __restrict__ bias,
                                               const float* __restrict__ hL,
                                               float* __restrict__ wbuf, f32x4 (&acc)[NT],
                                               int lane, int g, int nq = NT) {
  constexpr int TC = NT / 2;              // tiles per staged half
  constexpr int CH = TC * 256;            // floats per half-chunk
  constexpr int NV = CH / 4;              // float4 per half-chunk
  constexpr int PER = (NV + 255) / 256;   // float4 per thread per half-chunk
  constexpr bool FULL = (NV % 256) == 0;
  static_assert(NT % 2 == 0, "NT must be even");
  const int tid = threadIdx.x;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if constexpr (HAS_BIAS) {
      acc[t] = *reinterpret_cast<const f32x4*>(bias + 16 * t + 4 * g);
    } else {
      acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  f32x4 F[2][TC], G[2][PER];
  auto gload = [&](f32x4 (&r)[PER], int c) {
    const f32x4* src = reinterpret_cast<const f32x4*>(img + (int64_t)c * CH);
#pragma unroll
    for (int k = 0; k < PER; ++k)
      if (FULL || tid + 256 * k < NV) r[k] = src[tid + 256 * k];
  };
  auto lwrite = [&](const f32x4 (&r)[PER], int buf) {
    f32x4* dst = reinterpret_cast<f32x4*>(wbuf + buf * CH);
#pragma unroll
    for (int k = 0; k < PER; ++k)
      if (FULL || tid + 256 * k < NV) dst[tid + 256 * k] = r[k];
  };
  auto fread = [&](f32x4 (&f)[TC], int buf) {
    const f32x4* wa = reinterpret_cast<const f32x4*>(wbuf + buf * CH);
#pragma unroll
    for (int t = 0; t < TC; ++t) f[t] = wa[t * 64 + lane];
  };
  auto mma = [&](const f32x4 (&f)[TC], int half, const f32x4& b4) {
#pragma unroll
    for (int t = 0; t < TC; ++t) acc[half * TC + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[t].x, b4.x, acc[half * TC + t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < TC; ++t) acc[half * TC + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[t].y, b4.y, acc[half * TC + t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < TC; ++t) acc[half * TC + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[t].z, b4.z, acc[half * TC + t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < TC; ++t) acc[half * TC + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[t].w, b4.w, acc[half * TC + t], 0, 0, 0);
  };
  auto pattern = [&]() {
    // [2 MFMA, 1 LDS read] x TC, [MFMA, global load] x PER, [MFMA, LDS store] x PER, rest MFMAs
#pragma unroll
    for (int i = 0; i < TC; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    }
  };
  const int nc = 2 * nq;
  // prologue: chunk 0 -> buffer 0 -> F[0]; chunk 1 -> buffer 1; chunk 2 in flight in G[0]
  gload(G[0], 0);
  lwrite(G[0], 0);
  if (nc > 1) gload(G[1], 1);
  __syncthreads();
  fread(F[0], 0);
  if (nc > 1) lwrite(G[1], 1);
  if (nc > 2) gload(G[0], 2);
  f32x4 b4 = reinterpret_cast<const f32x4*>(hL)[lane];
  __syncthreads();
  for (int q = 0; q < nq; ++q) {
    const int c = 2 * q;
    // ---- half 0: multiply chunk c (F[0]); fetch chunk c+1 fragments, request chunk c+3, store chunk c+2
    if (c + 1 < nc) fread(F[1], 1);
    if (c + 3 < nc) gload(G[1], c + 3);
    mma(F[0], 0, b4);
    if (c + 2 < nc) lwrite(G[0], 0);
    pattern();
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    // ---- half 1: multiply chunk c+1 (F[1]); fetch chunk c+2 fragments, request chunk c+4, store chunk c+3
    f32x4 b4n = b4;
    if (q + 1 < nq) b4n = reinterpret_cast<const f32x4*>(hL)[(q + 1) * 64 + lane];
    if (c + 2 < nc) fread(F[0], 0);
    if (c + 4 < nc) gload(G[0], c + 4);
    mma(F[1], 1, b4);
    if (c + 3 < nc) lwrite(G[1], 1);
    pattern();
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    b4 = b4n;
  }
}

// EXPERIMENT (not the default; build siren.hip with -DISO_SIREN_DIRECT): measured 84 TFLOP/s vs
// 98 TFLOP/s for the LDS-staged pass on the bench workload -- the 8 waves of a CU re-reading the
// image through L1/L2 cost more than the barriers they save.
// Variant without the shared LDS stage: every wave streams its own A fragments straight from
// the (L2-resident, lane-linear) weight image with 16-B loads, double-buffered in registers
// (set X = even half-chunks, set Y = odd ones; a set is re-loaded right after the MFMAs that
// read it have been issued, so a load has a whole half-chunk of MFMAs to land).  No workgroup
// barrier at all: the waves of a CU drift freely and fill each other's stalls.
template <int NT, bool HAS_BIAS>
__device__ __forceinline__ void gemm_pass_direct(const float* __restrict__ img,
                                                 const float* __restrict__ bias,
                                                 const float* __restrict__ hL, f32x4 (&acc)[NT],
                                                 int lane, int g, int nq = NT) {
  constexpr int TC = NT / 2;
  constexpr int CH = TC * 256;
  static_assert(NT % 2 == 0, "NT must be even");
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if constexpr (HAS_BIAS) {
      acc[t] = *reinterpret_cast<const f32x4*>(bias + 16 * t + 4 * g);
    } else {
      acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  f32x4 ax[TC], ay[TC];
  auto gload = [&](f32x4 (&r)[TC], int c) {
    const f32x4* src = reinterpret_cast<const f32x4*>(img + (int64_t)c * CH) + lane;
#pragma unroll
    for (int t = 0; t < TC; ++t) r[t] = src[t * 64];
  };
  auto mm = [&](const f32x4 (&a4)[TC], int half, const f32x4& b4) {
#pragma unroll
    for (int t = 0; t < TC; ++t) {
      f32x4& d = acc[half * TC + t];
      d = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[t].x, b4.x, d, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[t].y, b4.y, d, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[t].z, b4.z, d, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[t].w, b4.w, d, 0, 0, 0);
    }
  };
  gload(ax, 0);
  gload(ay, 1);
  for (int q = 0; q < nq; ++q) {
    const f32x4 b4 = reinterpret_cast<const f32x4*>(hL)[q * 64 + lane];
    mm(ax, 0, b4);
    __builtin_amdgcn_sched_barrier(0);
    if (2 * q + 2 < 2 * nq) gload(ax, 2 * q + 2);
    __builtin_amdgcn_sched_barrier(0);
    mm(ay, 1, b4);
    __builtin_amdgcn_sched_barrier(0);
    if (2 * q + 3 < 2 * nq) gload(ay, 2 * q + 3);
    __builtin_amdgcn_sched_barrier(0);
  }
}
