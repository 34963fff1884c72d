// Fused SIREN SDF + gradient Newton step, "point-stationary" form (H = 256, gfx950).
//
// Same arithmetic as siren_x3.hip (split-fp16 operands, three fp16 MFMA products per f32 product, f32 accumulate; the
// reference semantics are Siren.forward DSS/models/common.py:140-165 under autograd.grad in
// UniformProjection._compute_sdf_and_grad, levelset_sampling.py:142-170) -- every accumulator and every sum sees the same
// operations in the same order, so results are BIT-IDENTICAL to k_siren_step_x3 (tests/test_siren_ps_gpu.py) and the two
// kernels can serve parts of one list -- but the work is cut the other way round:
//   * siren_x3.hip splits the OUTPUT FEATURES of a layer over eight waves; the activations of a tile live in LDS and
//     every stage is [GEMM | barrier | sin/cos/split | barrier]: matrix and vector pipes take turns (MFMA busy 40 %).
//   * here a wave owns 32 POINTS and all 256 features of them.  The input of a layer sits in the wave's registers as B
//     operands (the D layout of v_mfma_f32_32x32x16_f16 is the B layout of the next layer, siren_common.h); a layer is
//     computed output-tile pair by output-tile pair, and the vector work of pair t - 1 (range reduction, v_sin / v_cos,
//     the fp16 cuts, the stash) is issued by the SAME wave between the MFMAs of pair t: plain f32 VALU instructions
//     co-issue behind a 32x32x16 MFMA of their own wave (tools/probes/coissue.hip: 104 of them behind 24 MFMAs cost
//     5-7 %), which the two waves of a SIMD do not do for each other.  No barrier between GEMM and activation.
//   * the weight fragments come straight from L2 / L1 into registers, PS_PD K-steps ahead; the four waves of a workgroup
//     (one per SIMD, 512 registers each) walk the same image in step -- one barrier per stage keeps them together -- so
//     three of four requests are L1 hits.  (First form: fragments shared through a double-buffered LDS ring, one barrier
//     per four K-steps: 6 % slower, tools/experiments/siren_ps_ring.hip and profiles/HISTORY.md.)
// LDS: per wave 24 KiB staging of the next layer's input (K-steps 0..11; 12..15 go straight to registers) | W0 / head /
// bias vectors in K-order.
#include <stdlib.h>
#include <type_traits>
#include "siren_common.h"
#include "iso_newton.h"
#include "mlp_common.h"
#include "mfma_split.h"

namespace {

#ifndef PS_PD
#define PS_PD 3                               // K-steps a weight fragment is requested ahead of its MFMAs (3 or 7)
#endif
#ifndef PS_STAGE_BARRIER
#define PS_STAGE_BARRIER 1                    // the waves of a workgroup start every stage together (L1 reuse of the fragments)
#endif
constexpr int PS_W = 4;                       // waves per workgroup
constexpr int PS_P = 32 * PS_W;               // points per workgroup
constexpr int PS_H = 256;
constexpr int kPsYU4 = 12 * 2 * 64;           // u32x4 entries of one wave's staging region
constexpr int kPsYBytes = PS_W * kPsYU4 * 16;
constexpr int kPsConstFloats = (5 + 8) * PS_H;
#ifndef PS_LDS_STASH
#define PS_LDS_STASH 6                        // groups (of 16) of stash slot 0 that stay in LDS: 2 KiB per group and wave
#endif
constexpr size_t kPsLds = (size_t)kPsYBytes + (size_t)kPsConstFloats * 4 + (size_t)PS_W * PS_LDS_STASH * 2048;
static_assert(kPsLds <= 163840, "LDS of a gfx950 CU");

typedef const __attribute__((address_space(1))) u32x4* ps_gimg;
typedef __attribute__((address_space(1))) f32x4* ps_gf4;

// -DPS_DBG_TIMES (timing experiment, tools/ps_stage_times.py): the waves of workgroup 0 stamp the shader clock at the stage
// boundaries of their second tile
#ifdef PS_DBG_TIMES
__device__ long long ps_dbg[4 * 64];
#define PS_STAMP() do { if (dbg_on && lane == 0 && dbg_i < 64) ps_dbg[w * 64 + dbg_i] = (long long)__builtin_amdgcn_s_memtime(); ++dbg_i; } while (0)
#ifdef PS_DBG_INNER
#define PS_STAMP2() PS_STAMP()
#else
#define PS_STAMP2() do {} while (0)
#endif
#else
#define PS_STAMP() do {} while (0)
#define PS_STAMP2() do {} while (0)
#endif

static_assert(16 % (PS_PD + 1) == 0, "the register sets of the fragments rotate with the K-steps of a pair");
enum { PK_NONE = -1, PK_FWD_MID = 0, PK_FWD_TOP = 1, PK_REV_MID = 2, PK_REV0 = 3 };

// registers of the activation group in flight (8 values per lane) and the per-lane context of a stage
struct PsR {
  float x[8], t[8], n[8], f[8], s[8], c[8], r[8], b[8];
  unsigned h[4], l[4];
};
struct PsC {
  float w_in, w;          // s = sin(w_in z), c = w cos(w_in z)
  float scale;            // scale of the fp16 cut
  float inv;              // reverse stages: takes the scales of the accumulator out
  float amax;             // max |adjoint| over this lane's features
  float fp;               // head partial of the current output tile
  float gx, gy, gz;       // gradient partials of the current output tile
  float qx, qy, qz;       // the point
  f32x4 wl0, wl1;         // head weights of the group
  f32x4 sv0, sv1;         // derivative stash of the group
  f32x4 wv[8];            // rows of W0 of the group
};

// ---- the activation programs: one instruction-sized operation per index, step-major over the eight values -----------------
template <int KIND> struct PsCnt;
template <> struct PsCnt<PK_FWD_MID> { static constexpr int NS = 12; static constexpr int c[12] = {8, 8, 8, 8, 8, 8, 8, 8, 8, 4, 4, 4}; };
template <> struct PsCnt<PK_FWD_TOP> { static constexpr int NS = 18; static constexpr int c[18] = {8, 8, 8, 8, 8, 8, 8, 8, 8, 4, 2, 2, 8, 4, 8, 4, 4, 4}; };
template <> struct PsCnt<PK_REV_MID> { static constexpr int NS = 7; static constexpr int c[7] = {8, 8, 4, 8, 4, 4, 4}; };
template <> struct PsCnt<PK_REV0> { static constexpr int NS = 21; static constexpr int c[21] = {8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8}; };
template <int KIND> constexpr int ps_total() { int t = 0; for (int s = 0; s < PsCnt<KIND>::NS; ++s) t += PsCnt<KIND>::c[s]; return t; }
template <int KIND> constexpr int ps_step_of(int i) { int s = 0; while (i >= PsCnt<KIND>::c[s]) { i -= PsCnt<KIND>::c[s]; ++s; } return s; }
template <int KIND> constexpr int ps_elem_of(int i) { int s = 0; while (i >= PsCnt<KIND>::c[s]) { i -= PsCnt<KIND>::c[s]; ++s; } return i; }
// first operation after the sin / cos chain (what the large-argument path runs again)
template <int KIND> constexpr int ps_post() { return KIND == PK_REV0 ? 104 : 64; }

// Every operation's result is pinned where it is produced (an empty asm the compiler may not move or drop): left to
// itself LLVM sinks the whole chain behind the large-argument branch of group_end, i.e. out from between the MFMAs.
#define PS_PIN(v) asm volatile("" : "+v"(v))
constexpr float kRevHi = 0.159154936671257019043f, kRevLo = 6.4206383167e-9f;     // 1 / (2 pi), two terms (iso_sin_wcos8)

// the two-way fp16 cut of split8_f16 (mfma_split.h), operation by operation: v -> (h, l) under `scale`
template <int S, int E>
__device__ __forceinline__ void ps_split_op(PsR& R, const float (&v)[8], float scale) {
  if constexpr (S == 0) { R.r[E] = v[E] * scale; PS_PIN(R.r[E]); }
  if constexpr (S == 1) { R.h[E] = __builtin_bit_cast(unsigned, __builtin_convertvector(((f32x2){R.r[2 * E], R.r[2 * E + 1]}), f16x2)); PS_PIN(R.h[E]); }
  if constexpr (S == 2) asm volatile("v_fma_mixlo_f16 %0, 1.0, %1, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(R.l[E]) : "v"(R.r[2 * E]), "v"(R.h[E]));
  if constexpr (S == 3) asm volatile("v_fma_mixhi_f16 %0, 1.0, %1, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(R.l[E]) : "v"(R.r[2 * E + 1]), "v"(R.h[E]));
}
// range reduction + hardware sin / cos of iso_sin_wcos8 / iso_wcos8 (mlp_common.h), operation by operation
template <int S, int E, bool WITH_SIN>
__device__ __forceinline__ void ps_sincos_op(PsR& R, float zin, const PsC& cx) {
  if constexpr (S == 0) { R.x[E] = zin * cx.w_in; PS_PIN(R.x[E]); }
  if constexpr (S == 1) { R.t[E] = R.x[E] * kRevHi; PS_PIN(R.t[E]); }
  if constexpr (S == 2) { R.n[E] = rintf(R.t[E]); PS_PIN(R.n[E]); }
  if constexpr (S == 3) { R.f[E] = __builtin_fmaf(R.x[E], kRevHi, -R.n[E]); PS_PIN(R.f[E]); }
  if constexpr (S == 4) { R.f[E] = __builtin_fmaf(R.x[E], kRevLo, R.f[E]); PS_PIN(R.f[E]); }
  if constexpr (S == 5 && WITH_SIN) { R.s[E] = __builtin_amdgcn_sinf(R.f[E]); PS_PIN(R.s[E]); }
  if constexpr (S == 6) { R.c[E] = __builtin_amdgcn_cosf(R.f[E]); PS_PIN(R.c[E]); }
  if constexpr (S == 7) { R.c[E] = R.c[E] * cx.w; PS_PIN(R.c[E]); }
}

template <int KIND, int I>
__device__ __forceinline__ void ps_op(PsR& R, PsC& cx, const float (&z)[8]) {
  constexpr int S = ps_step_of<KIND>(I), E = ps_elem_of<KIND>(I);
  if constexpr (KIND == PK_FWD_MID) {
    if constexpr (S <= 7) ps_sincos_op<S, E, true>(R, z[E], cx);
    else ps_split_op<S - 8, E>(R, R.s, cx.scale);
  }
  if constexpr (KIND == PK_FWD_TOP) {
    if constexpr (S <= 7) ps_sincos_op<S, E, true>(R, z[E], cx);
    // head dot product (k_siren_step_x3: f0 = (w0 h0 + w1 h1) + (w2 h2 + w3 h3), f1 likewise, fp += f0 + f1) ...
    if constexpr (S == 8) { R.r[E] = (E < 4 ? cx.wl0[E & 3] : cx.wl1[E & 3]) * R.s[E]; PS_PIN(R.r[E]); }
    if constexpr (S == 9) { R.t[E] = R.r[2 * E] + R.r[2 * E + 1]; PS_PIN(R.t[E]); }
    if constexpr (S == 10) { R.n[E] = R.t[2 * E] + R.t[2 * E + 1]; PS_PIN(R.n[E]); }
    if constexpr (S == 11) { if constexpr (E == 0) { R.n[2] = R.n[0] + R.n[1]; PS_PIN(R.n[2]); } else { cx.fp = cx.fp + R.n[2]; PS_PIN(cx.fp); } }
    // ... and the seed of the adjoint: head weight * w cos(w z)
    if constexpr (S == 12) { R.s[E] = (E < 4 ? cx.wl0[E & 3] : cx.wl1[E & 3]) * R.c[E]; PS_PIN(R.s[E]); }
    if constexpr (S == 13) { cx.amax = __builtin_fmaxf(cx.amax, __builtin_fmaxf(__builtin_fabsf(R.s[2 * E]), __builtin_fabsf(R.s[2 * E + 1]))); PS_PIN(cx.amax); }
    if constexpr (S >= 14) ps_split_op<S - 14, E>(R, R.s, cx.scale);
  }
  if constexpr (KIND == PK_REV_MID) {
    if constexpr (S == 0) { R.t[E] = z[E] * cx.inv; PS_PIN(R.t[E]); }
    if constexpr (S == 1) { R.s[E] = R.t[E] * (E < 4 ? cx.sv0[E & 3] : cx.sv1[E & 3]); PS_PIN(R.s[E]); }
    if constexpr (S == 2) { cx.amax = __builtin_fmaxf(cx.amax, __builtin_fmaxf(__builtin_fabsf(R.s[2 * E]), __builtin_fabsf(R.s[2 * E + 1]))); PS_PIN(cx.amax); }
    if constexpr (S >= 3) ps_split_op<S - 3, E>(R, R.s, cx.scale);
  }
  if constexpr (KIND == PK_REV0) {
    // z0 = W0 x + b0 again (the expression of the forward sweep), its w cos, then d sdf / d x
    if constexpr (S == 0) { R.t[E] = cx.wv[E].x * cx.qx; PS_PIN(R.t[E]); }
    if constexpr (S == 1) { R.n[E] = cx.wv[E].y * cx.qy; PS_PIN(R.n[E]); }
    if constexpr (S == 2) { R.t[E] = R.t[E] + R.n[E]; PS_PIN(R.t[E]); }
    if constexpr (S == 3) { R.n[E] = cx.wv[E].z * cx.qz; PS_PIN(R.n[E]); }
    if constexpr (S == 4) { R.t[E] = R.t[E] + R.n[E]; PS_PIN(R.t[E]); }
    if constexpr (S == 5) { R.b[E] = R.t[E] + cx.wv[E].w; PS_PIN(R.b[E]); }
    if constexpr (S >= 6 && S <= 12) ps_sincos_op<(S - 6 < 5 ? S - 6 : S - 5), E, false>(R, R.b[E], cx);
    if constexpr (S == 13) { R.t[E] = z[E] * cx.inv; PS_PIN(R.t[E]); }
    if constexpr (S == 14) { R.s[E] = R.t[E] * R.c[E]; PS_PIN(R.s[E]); }
    if constexpr (S == 15) { R.r[E] = cx.wv[E].x * R.s[E]; PS_PIN(R.r[E]); }
    if constexpr (S == 16) { R.f[E] = cx.wv[E].y * R.s[E]; PS_PIN(R.f[E]); }
    if constexpr (S == 17) { R.n[E] = cx.wv[E].z * R.s[E]; PS_PIN(R.n[E]); }
    if constexpr (S == 18) { cx.gx = cx.gx + R.r[E]; PS_PIN(cx.gx); }
    if constexpr (S == 19) { cx.gy = cx.gy + R.f[E]; PS_PIN(cx.gy); }
    if constexpr (S == 20) { cx.gz = cx.gz + R.n[E]; PS_PIN(cx.gz); }
  }
}
template <int KIND, int LO, int HI>
__device__ __forceinline__ void ps_ops(PsR& R, PsC& cx, const float (&z)[8]) {
  if constexpr (LO < HI) { ps_op<KIND, LO>(R, cx, z); ps_ops<KIND, LO + 1, HI>(R, cx, z); }
}

template <int N, class F>
__device__ __forceinline__ void ps_for(F&& f) {
  if constexpr (N > 0) { ps_for<N - 1>(f); f(std::integral_constant<int, N - 1>()); }
}

// ---- the step ----------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ps_step_body(const SirenArgs& a, const int bid, const int nblk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* cst = reinterpret_cast<float*>(smem_raw + kPsYBytes);                       // [W0k 4H][WLk H][bias_k H] x L
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, h = lane >> 5, j = lane & 31, h8 = h * 8;
  u32x4* ybase = reinterpret_cast<u32x4*>(smem_raw) + w * kPsYU4 + lane;             // [12][2][64]
  const int L = a.L;
  const int NST = 2 * L;                                                             // GEMM stages per tile
  const f32x4* W0s = reinterpret_cast<const f32x4*>(cst) + h8;                       // + s * 16 + e
  const float* WLs = cst + 4 * PS_H + h8;                                            // + s * 16
  const float bL = a.packed[off_bl(PS_H)];
  ps_gf4 stash = (ps_gf4)(a.stash) + ((int64_t)bid * PS_W + w) * (int64_t)(L > 1 ? L : 1) * (16 * 2 * 64) + lane;
  const float* hdr = a.packed + x16_base(PS_H, L);
  f32x4* lst = reinterpret_cast<f32x4*>(smem_raw + kPsYBytes + kPsConstFloats * 4) + w * (PS_LDS_STASH * 2 * 64) + lane;
  auto stash_get = [&](int stl, int g, f32x4& v0, f32x4& v1) {
#ifdef PS_DBG_NOSTASH      // timing experiment: no derivative stash at all (results wrong by construction)
    v0 = v1 = (f32x4){1.f, 1.f, 1.f, (float)g}; return;
#endif
#ifdef PS_DBG_HALFSTASH    // timing experiment: slot 0 costs nothing (the upper bound of keeping it on chip)
    if (stl == 0) { v0 = v1 = (f32x4){1.f, 1.f, 1.f, (float)g}; return; }
#endif
    if (PS_LDS_STASH > 0 && stl == 0 && g < PS_LDS_STASH) { v0 = lst[(g * 2 + 0) * 64]; v1 = lst[(g * 2 + 1) * 64]; }
    else { v0 = stash[((stl * 16 + g) * 2 + 0) * 64]; v1 = stash[((stl * 16 + g) * 2 + 1) * 64]; }
  };
  auto stash_put = [&](int stl, int g, const f32x4& v0, const f32x4& v1) {
#ifdef PS_DBG_NOSTASH
    return;
#endif
#ifdef PS_DBG_HALFSTASH
    if (stl == 0) return;
#endif
    if (PS_LDS_STASH > 0 && stl == 0 && g < PS_LDS_STASH) { lst[(g * 2 + 0) * 64] = v0; lst[(g * 2 + 1) * 64] = v1; }
    else { stash[((stl * 16 + g) * 2 + 0) * 64] = v0; stash[((stl * 16 + g) * 2 + 1) * 64] = v1; }
  };
  // images of the GEMM stages: forward layers 0 .. L-1, then the transposed ones L-1 .. 0 (contiguous: siren_common.h)
  const ps_gimg img0 = (ps_gimg)(a.packed + x16_off_layer(PS_H, L, 0)) + lane;
  auto img_of = [&](int st) { return img0 + (st < L ? st : 3 * L - 1 - st) * (PS_H * PS_H / 4); };

  // K-order vectors into LDS (all waves; visible after the first barrier below)
  {
    const float* X = a.packed + x3_base(PS_H, L);
    for (int i = tid; i < (5 + L) * PS_H; i += 64 * PS_W) cst[i] = X[i];
  }

  const int64_t total = a.count_in ? (int64_t)__hip_atomic_load(a.count_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : a.n;
  if (!siren_ps_takes(total, L, hdr, a.wh)) return;        // the launch that follows (ps_guard) does this list
  const int64_t n_tiles = (total + PS_P - 1) / PS_P;
  if (bid >= n_tiles) return;

#ifdef PS_DBG_TIMES
  bool dbg_on = false;
  int dbg_i = 0;
#endif
  __syncthreads();                                         // the K-order vectors are in LDS

  // K-step K of a tile = stage K / 64, output-tile pair (K / 16) % 4, step K % 16: four fragments (two tiles x two
  // parts), contiguous in the image.  Requested PS_PD K-steps ahead into a rotation of PS_PD + 1 register sets.
  u32x4 AD[PS_PD + 1][4];
  auto load_a = [&](u32x4 (&Ar)[4], int K) {
    if (K >= 64 * NST) K -= 64 * NST;                      // the next tile's first K-steps
    const ps_gimg pa = img_of(K >> 6) + (16 * (K & 15) + 4 * ((K >> 4) & 3)) * 64;
#pragma unroll
    for (int f = 0; f < 4; ++f) Ar[f] = pa[f * 64];
  };
#pragma unroll
  for (int d = 0; d < PS_PD; ++d) load_a(AD[d], d);
  int q = 0;                                               // chunk (four K-steps) being multiplied (uniform)

  u32x4 Xh[16], Xl[16];                                    // input of the current stage: B operands, high / low parts
  f32x16 acc[2], accP[2];
  PsR R;
  PsC cx;
  f32x4 sv_n0, sv_n1;                                      // the next group's stash (reverse stages)
  float ftot = 0.f, gtx = 0.f, gty = 0.f, gtz = 0.f;
  float bscale = 1.f;
  bool unsafe0 = false;                                    // this tile has a point whose layer-0 arguments may be large

  // One chunk: K-steps 4 C .. 4 C + 3 of the pair being multiplied (6 MFMAs each), group C of the pair before it activated
  // between them, one slice of its program behind every MFMA.
  auto chunk = [&](auto kind_c, auto c_c, const float (&z)[8]) {
    constexpr int KIND = decltype(kind_c)::value, C = decltype(c_c)::value;
    ps_for<4>([&](auto k_c) {
      constexpr int k = decltype(k_c)::value, s = 4 * C + k;
      load_a(AD[(s + PS_PD) % (PS_PD + 1)], 4 * q + k + PS_PD);
      __builtin_amdgcn_sched_barrier(0);
      ps_for<6>([&](auto m_c) {
        constexpr int m = decltype(m_c)::value, u = m & 1;
        constexpr int f = u * 2 + (m < 2 ? 1 : 0);         // W_l x_h, W_h x_l, W_h x_h: the order of gemm_x3
        const u32x4& Bv = (m >= 2 && m < 4) ? Xl[s] : Xh[s];
        acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, AD[s % (PS_PD + 1)][f]), __builtin_bit_cast(f16x8, Bv), acc[u], 0, 0, 0);
        if constexpr (KIND != PK_NONE) {
          constexpr int T = ps_total<KIND>(), i = 6 * k + m;
          ps_ops<KIND, (i * T) / 24, ((i + 1) * T) / 24>(R, cx, z);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    ++q;
  };

  // the group's inputs that come from memory; results of a finished group
  auto group_begin = [&](auto kind_c, int sg /* K-step of the next layer this group makes: 4 pair + g */, int stl) {
    constexpr int KIND = decltype(kind_c)::value;
    if constexpr (KIND == PK_FWD_TOP) {
      if ((sg & 1) == 0) cx.fp = 0.f;
      cx.wl0 = *reinterpret_cast<const f32x4*>(WLs + sg * 16);
      cx.wl1 = *reinterpret_cast<const f32x4*>(WLs + sg * 16 + 4);
    }
    if constexpr (KIND == PK_REV_MID) {
      cx.sv0 = sv_n0; cx.sv1 = sv_n1;
      const int nx = sg + 1 < 16 ? sg + 1 : 15;
      stash_get(stl, nx, sv_n0, sv_n1);
    }
    if constexpr (KIND == PK_REV0) {
      if ((sg & 1) == 0) { cx.gx = 0.f; cx.gy = 0.f; cx.gz = 0.f; }
#pragma unroll
      for (int e = 0; e < 8; ++e) cx.wv[e] = W0s[sg * 16 + e];
    }
  };
  // ends a group: large arguments (|w z| >= 1e4: libm's reduction, as iso_sin_wcos8 does), the stash, per-tile sums;
  // returns with R.h / R.l = the group's entry of the next stage's input
  auto group_end = [&](auto kind_c, const float (&z)[8], int sg, int stl, float g0x, float g0y, float g0z) {
    constexpr int KIND = decltype(kind_c)::value;
    // Large arguments (|w z| >= 1e4: libm's reduction, as iso_sin_wcos8 does).  Hidden layers: none by siren_ps_takes.
    // Layer 0: only in waves whose points are far enough out for it (unsafe0, from the bound of siren_common.h) -- they
    // take cos again through iso_wcos8, which repeats the fast path bit for bit where that one applies.
    if constexpr (KIND == PK_REV0) {
      if (__builtin_expect(unsafe0, 0)) {
        float zz[8], cc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) zz[e] = R.b[e];
        iso_wcos8(cx.w_in, cx.w, zz, cc);
#pragma unroll
        for (int e = 0; e < 8; ++e) R.c[e] = cc[e];
        cx.gx = g0x; cx.gy = g0y; cx.gz = g0z;
        ps_ops<KIND, ps_post<KIND>(), ps_total<KIND>()>(R, cx, z);
      }
    }
    if constexpr (KIND == PK_FWD_MID) {
      stash_put(stl, sg, (f32x4){R.c[0], R.c[1], R.c[2], R.c[3]}, (f32x4){R.c[4], R.c[5], R.c[6], R.c[7]});
    }
    if constexpr (KIND == PK_FWD_TOP) {
      if (sg & 1) {                                          // the output tile is complete: the order of k_siren_step_x3's reduction
        const float sT = cx.fp + __shfl_xor(cx.fp, 32);
        ftot = sg == 1 ? sT : ftot + sT;
      }
    }
    if constexpr (KIND == PK_REV0) {
      if (sg & 1) {
        const float sx = cx.gx + __shfl_xor(cx.gx, 32), sy = cx.gy + __shfl_xor(cx.gy, 32), sz = cx.gz + __shfl_xor(cx.gz, 32);
        gtx = sg == 1 ? sx : gtx + sx;
        gty = sg == 1 ? sy : gty + sy;
        gtz = sg == 1 ? sz : gtz + sz;
      }
    }
  };
  auto hl_hi = [&]() { return (u32x4){R.h[0], R.h[1], R.h[2], R.h[3]}; };
  auto hl_lo = [&]() { return (u32x4){R.l[0], R.l[1], R.l[2], R.l[3]}; };

  // One GEMM stage + the activation of its outputs.  stl: stash slot written (forward) / read (reverse); bias: K-order bias
  // in LDS or null; zscale multiplies it.
  auto stage = [&](auto kind_c, int stl, const float* bias, float zscale) {
    constexpr int KIND = decltype(kind_c)::value;
    constexpr bool MAKES_INPUT = KIND != PK_REV0;
    auto init_acc = [&](int Tp) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (bias) {
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const float* bp = bias + (2 * (2 * Tp + u) + p) * 16 + h8;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(bp), hi = *reinterpret_cast<const f32x4*>(bp + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[u][8 * p + e] = lo[e] * zscale; acc[u][8 * p + 4 + e] = hi[e] * zscale; }
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
        }
      }
    };
#if PS_STAGE_BARRIER
    asm volatile("s_barrier" ::: "memory");
#endif
    float zdummy[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    using none_t = std::integral_constant<int, PK_NONE>;
    // pair 0: nothing to activate yet
    init_acc(0);
    if constexpr (KIND == PK_REV_MID) {                      // the first group's stash
      stash_get(stl, 0, sv_n0, sv_n1);
    }
    ps_for<4>([&](auto c_c) { chunk(none_t(), c_c, zdummy); });
    PS_STAMP2();
#pragma unroll 1
    for (int Tp = 1; Tp < 4; ++Tp) {
      accP[0] = acc[0]; accP[1] = acc[1];
      init_acc(Tp);
      ps_for<4>([&](auto g_c) {
        constexpr int g = decltype(g_c)::value;
        float z[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = accP[g >> 1][8 * (g & 1) + e];
        const int sg = 4 * (Tp - 1) + g;
        const float g0x = cx.gx, g0y = cx.gy, g0z = cx.gz;
        group_begin(kind_c, sg, stl);
        chunk(kind_c, g_c, z);
        group_end(kind_c, z, sg, stl, (sg & 1) ? g0x : 0.f, (sg & 1) ? g0y : 0.f, (sg & 1) ? g0z : 0.f);
        if constexpr (MAKES_INPUT) {
          ybase[(sg * 2 + 0) * 64] = hl_hi();
          ybase[(sg * 2 + 1) * 64] = hl_lo();
        }
      });
      PS_STAMP2();
    }
    // the last pair: no GEMM left in this stage to hide behind; its entries go straight to registers
    accP[0] = acc[0]; accP[1] = acc[1];
    ps_for<4>([&](auto g_c) {
      constexpr int g = decltype(g_c)::value;
      float z[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) z[e] = accP[g >> 1][8 * (g & 1) + e];
      const int sg = 12 + g;
      const float g0x = cx.gx, g0y = cx.gy, g0z = cx.gz;
      group_begin(kind_c, sg, stl);
#ifndef PS_DBG_NOEPI
      ps_ops<KIND, 0, ps_total<KIND>()>(R, cx, z);
#endif
      group_end(kind_c, z, sg, stl, (sg & 1) ? g0x : 0.f, (sg & 1) ? g0y : 0.f, (sg & 1) ? g0z : 0.f);
      if constexpr (MAKES_INPUT) { Xh[12 + g] = hl_hi(); Xl[12 + g] = hl_lo(); }
    });
    PS_STAMP2();
    if constexpr (MAKES_INPUT) {
#pragma unroll
      for (int s = 0; s < 12; ++s) { Xh[s] = ybase[(s * 2 + 0) * 64]; Xl[s] = ybase[(s * 2 + 1) * 64]; }
    }
  };

  // scale of the adjoint seed (uniform): |W_head[f] * w cos| <= max |W_head| * w
  const float seed_scale = x3_scale_for(hdr[16] * a.wh * 1.01f);

  for (int64_t tile = bid; tile < n_tiles; tile += nblk) {
    q = 0;
#ifdef PS_DBG_TIMES
    dbg_on = bid == 0 && tile == (int64_t)nblk;
    dbg_i = 0;
#endif
    PS_STAMP();
    // ---- the tile's points: lane (h, j) holds point 32 w + j -------------------------------------------------------------
    int64_t idx = -1;
    {
      const int64_t slot = tile * PS_P + 32 * w + j;
      if (slot < total) idx = a.idx_in ? (int64_t)a.idx_in[slot] : slot;
    }
    float px = 0.f, py = 0.f, pz = 0.f;
    if (idx >= 0) { px = a.pts[idx * 3]; py = a.pts[idx * 3 + 1]; pz = a.pts[idx * 3 + 2]; }
    {
      const float pm = __builtin_fmaxf(__builtin_fabsf(px), __builtin_fmaxf(__builtin_fabsf(py), __builtin_fabsf(pz)));
      const bool fin = (px == px) && (py == py) && (pz == pz);
      unsafe0 = __any(!fin || !((a.w0 * (hdr[22] * pm + hdr[23])) * 1.01f < 1.0e4f));
    }
    // ---- layer 0 (3 -> H) on the VALU: K-steps 0..11 of hidden layer 0's input through the staging region (one rolled
    // copy of the code), 12..15 straight into registers ----------------------------------------------------------------
    auto layer0 = [&](int s, u32x4& oh, u32x4& ol) {
      f32x4 wv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) wv[e] = W0s[s * 16 + e];
      float zz[8], hv[8], sv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) zz[e] = ((wv[e].x * px + wv[e].y * py) + wv[e].z * pz) + wv[e].w;
      iso_sin_wcos8(a.w0, a.w0, zz, hv, sv);
      (void)sv;
      split8_f16(hv, oh, ol);
    };
#pragma unroll 1
    for (int s = 0; s < 12; ++s) {
      u32x4 oh, ol;
      layer0(s, oh, ol);
      ybase[(s * 2 + 0) * 64] = oh; ybase[(s * 2 + 1) * 64] = ol;
    }
    ps_for<4>([&](auto g_c) { constexpr int g = decltype(g_c)::value; layer0(12 + g, Xh[12 + g], Xl[12 + g]); });
#pragma unroll
    for (int s = 0; s < 12; ++s) { Xh[s] = ybase[(s * 2 + 0) * 64]; Xl[s] = ybase[(s * 2 + 1) * 64]; }
    PS_STAMP();
    // ---- hidden layers, forward ------------------------------------------------------------------------------------------
    for (int l = 0; l < L; ++l) {
      const float zscale = kActScale * hdr[l];
      cx.w_in = a.wh / zscale; cx.w = a.wh;
      const float* bias = cst + (5 + l) * PS_H;
      if (l + 1 < L) {
        cx.scale = kActScale;
        stage(std::integral_constant<int, PK_FWD_MID>(), l, bias, zscale);
      } else {
        cx.scale = seed_scale; cx.amax = 0.f; cx.fp = 0.f;
        stage(std::integral_constant<int, PK_FWD_TOP>(), l, bias, zscale);
        bscale = seed_scale;
      }
      PS_STAMP();
    }
    // ---- hidden layers, reverse --------------------------------------------------------------------------------------------
    for (int l = L - 1; l >= 1; --l) {
      const float M = __builtin_fmaxf(cx.amax, __shfl_xor(cx.amax, 32));       // max over all features of the point
      const float iw = 1.0f / hdr[l];
      const float grow = hdr[8 + l] * a.wh * 1.01f;
      cx.inv = iw / bscale;
      const float nscale = x3_scale_for(M * grow);
      cx.scale = nscale; cx.amax = 0.f;
      stage(std::integral_constant<int, PK_REV_MID>(), l - 1, nullptr, 1.0f);
      bscale = nscale;
      PS_STAMP();
    }
    {
      const float iw = 1.0f / hdr[0];
      cx.inv = iw / bscale;
      cx.w_in = a.w0; cx.w = a.w0;
      cx.qx = px; cx.qy = py; cx.qz = pz;
      cx.gx = cx.gy = cx.gz = 0.f;
      stage(std::integral_constant<int, PK_REV0>(), 0, nullptr, 1.0f);
      PS_STAMP();
    }
    // ---- epilogue: lane (0, j) finishes point j ----------------------------------------------------------------------------
    bool survive = false;
    if (h == 0 && idx >= 0) survive = iso_step_finish<SirenArgs, true>(a, idx, ftot + bL, gtx, gty, gtz, px, py, pz);
    if (!a.eval_only && a.do_move) {
      const unsigned long long bal = __ballot(survive);
      if (bal) {
        int base = 0;
        const int leader = __ffsll((long long)bal) - 1;
        if (lane == leader) base = atomicAdd(a.count_out, __popcll(bal));
        base = __shfl(base, leader);
        if (survive) a.idx_out[base + __popcll(bal & ((1ull << lane) - 1ull))] = (int32_t)idx;
      }
    }
    PS_STAMP();
  }
}

__global__ __launch_bounds__(64 * PS_W, 1) void k_siren_step_ps(SirenArgs a) {
  ps_step_body(a, (int)blockIdx.x, (int)gridDim.x);
}

}  // namespace

#ifdef PS_DBG_TIMES
extern "C" int iso_dbg_ps_times(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(ps_dbg), sizeof(long long) * 4 * 64) == hipSuccess ? 0 : -1;
}
#endif

bool siren_ps_supported(int H, int L) { return H == 256 && L >= 2 && L <= 5; }   // (the bounds of siren_ps_takes: five slots)

int siren_ps_launch(const SirenArgs& a, int64_t n_upper, hipStream_t s) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_siren_step_ps), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPsLds);
    attr_done = true;
  }
  const int64_t tiles = (n_upper + PS_P - 1) / PS_P;
  const int blocks = (int)(tiles < 256 ? (tiles < 1 ? 1 : tiles) : 256);
  hipLaunchKernelGGL(k_siren_step_ps, dim3(blocks), dim3(64 * PS_W), kPsLds, s, a);
  return 0;
}
