// EXPERIMENT (round 2), not compiled into the library: second form of the two-set kernel (one activation buffer,
// results parked in registers, buffer instructions for every global access, X3W_NW = 8 or 4 waves).  Numerically right
// to the last bits (the cross-wave sums are grouped differently with 4 waves; 8 waves reproduce k_siren_step_x3), but
// the register file does not carry it: 600+ spilled VGPRs in either shape (only 256 of a wave's 512 registers are
// architectural ones the VALU can address), 4.4 ms per 1 M evaluations as it stands.  See DESIGN.md 3.1.
// Fused SIREN SDF + gradient Newton step, H = 256: the split-fp16 kernel of siren_x3.hip with the matrix and the
// vector stages running SIDE BY SIDE.  Same reference semantics (Siren.forward DSS/models/common.py:140-165 under
// autograd.grad, levelset_sampling.py:142-170, iterated by _project_points :313-342), same per-point arithmetic,
// bit for bit (every operation of a point is the one k_siren_step_x3 performs, in the same order), so either kernel
// can serve any launch.
//
// In k_siren_step_x3 a tile alternates between GEMM stages (matrix pipe busy, VALU idle) and activation stages
// (sin / cos, fp16 cuts: VALU busy, matrix pipe idle) of about equal length.  tools/probes/coissue.hip: the 104
// plain-f32 VALU instructions of one 8-value activation group issued between 24 MFMAs of the same wave cost 7 % on
// top of the MFMAs alone (826 against 770 cycles) -- but only as plain ops: packed f32 ops collide with the matrix
// pipe (1082 cycles).  So a workgroup carries TWO sets of 32 NB points, one stage apart: while the GEMM of one set
// runs, every wave issues the activation work of the other set between its own MFMAs (sched_group_barrier).  This
// file is compiled with -fno-slp-vectorize so that the vector stages stay plain f32.
//
// What shapes the rest (measured on siren_x3d.hip, the first form of this idea -- 8 waves, two 64-point sets, both
// sets' activations in LDS: correct, bit-identical, and SLOWER than the one-set kernel, 3.3 against 3.0 ms per 1 M
// evaluations): (1) the weight stream L2 -> CU sustains ~30 B / clock / CU; a fragment that feeds two point tiles
// instead of three makes the GEMMs weight-bound (1.8 ms per launch with the vector work removed, MFMA floor 0.94);
// (2) loads, stores and their waits retire in order per wave, so a stash store in front of a weight-fragment load
// stalls the MFMAs that wait for the fragment.  Hence: FOUR waves (one per SIMD, 512 registers each), every wave owns
// 64 output features, a weight fragment feeds NB = 3 point tiles as before -- and only ONE activation buffer in LDS:
// the buffer holds the input of the set that is in its GEMM; the vector stage of the other set leaves its results in
// the registers its accumulators occupied ("parked") and writes them to the buffer after the GEMM has read it for the
// last time (barrier, write, barrier).  A K-step is 18 MFMAs = 576 cycles, so a fragment requested three K-steps
// ahead has 1.7 k cycles to arrive, stash stores included.
//
//   per set (L hidden layers):  V0  G1 V1  G2 V2 ... G_L V_L(top)  G_{L+1} V_{L+1} ... G_{2L}   (4L stages)
//     V0     = reverse activation of layer 0 of the set's PREVIOUS tile (gradient, partial sums -> LDS),
//              then layer 0 (3 -> H) of its next tile
//     G_k    = hidden GEMM k (forward layers 0..L-1, then the transposed ones L-1..0)
//     V_k    = sin / w cos / split of the GEMM before it (top layer: head + adjoint seed; reverse: stash * adjoint)
//     V1     also carries the epilogue (cross-wave sum, Newton move, survivor list) of the previous tile
//   the two sets are one stage apart, every slot is [G of one set || V of the other] [barrier] [write] [barrier].
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "siren_common.h"
#include "iso_newton.h"
#include "mlp_common.h"
#include "mfma_split.h"

namespace {

#ifndef X3W_NW
#define X3W_NW 8                       // waves per workgroup: 8 (32 output features each) or 4 (64 each)
#endif
constexpr int WH = 256, WNW = X3W_NW, WNS = 16, WNTO = 8, WTW = WNTO / WNW, WSL = 2 * WTW;

template <int NB>
struct X3wShape {
  static constexpr int NG = WSL * NB;                              // 8-value groups per lane
  static constexpr int P = 32 * NB;                                // points per set
  static constexpr size_t kActBytes = (size_t)WNS * NB * 2 * 1024; // [K-step][point tile][part][lane] u32x4
  static constexpr size_t kRedBytes = (size_t)WNW * P * 16;        // [wave][point] {f, gx, gy, gz}
  static constexpr size_t kMaxBytes = (size_t)2 * P * WNW * 4;     // [buffer][point][wave] max |adjoint|
  static constexpr size_t kSetBytes = kRedBytes + kMaxBytes;
  static constexpr size_t kLds = kActBytes + 2 * kSetBytes;
  static_assert(kLds <= 160 * 1024, "must fit the LDS of a CU");
  static constexpr int64_t stash_per_wg(int L) { return (int64_t)2 * WNW * (L + 1) * NG * 512; }   // floats
};

typedef __attribute__((address_space(1))) f32x4* gf4_t;
typedef const __attribute__((address_space(1))) f32x4* gcf4_t;
typedef const __attribute__((address_space(1))) float* gcf_t;

template <class F, int... I>
__device__ __forceinline__ void x3w_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void x3w_for(F&& f) { x3w_for_impl(f, std::make_integer_sequence<int, N>{}); }

// a wave-uniform pointer the compiler can keep in SGPRs (loads then take the scalar-base + lane-offset form)
template <class T>
__device__ __forceinline__ T x3w_uniform(T p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (T)(((unsigned long long)hi << 32) | lo);
}

#define X3W_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#ifndef X3W_VALU_PER_MFMA
#define X3W_VALU_PER_MFMA 5
#endif
#ifndef X3W_KAD
#define X3W_KAD 3                      // weight fragments are requested this many K-steps ahead (4 rotating sets)
#endif

// ---- the activation of eight values, plain f32 ops, in three phases (iso_sin_wcos8 element by element) --------
struct X3wSc { float x[8], f[8], s[8], c[8]; };

template <int PH>
__device__ __forceinline__ void x3w_sc(X3wSc& T, const float (&z)[8], float w_in, float w) {
  const float hi = 0.159154936671257019043f, lo = 6.4206383167e-9f;
  if constexpr (PH == 0) {
    float t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) T.x[e] = z[e] * w_in;
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = T.x[e] * hi;
#pragma unroll
    for (int e = 0; e < 8; ++e) T.c[e] = __builtin_rintf(t[e]);
  }
  if constexpr (PH == 1) {
#pragma unroll
    for (int e = 0; e < 8; ++e) T.f[e] = __builtin_fmaf(T.x[e], hi, -T.c[e]);
#pragma unroll
    for (int e = 0; e < 8; ++e) T.f[e] = __builtin_fmaf(T.x[e], lo, T.f[e]);
#pragma unroll
    for (int e = 0; e < 8; ++e) T.s[e] = __builtin_amdgcn_sinf(T.f[e]);
  }
  if constexpr (PH == 2) {
#pragma unroll
    for (int e = 0; e < 8; ++e) T.c[e] = __builtin_amdgcn_cosf(T.f[e]);
#pragma unroll
    for (int e = 0; e < 8; ++e) T.c[e] = T.c[e] * w;
  }
}

// split8_f16 (mfma_split.h) element by element
__device__ __forceinline__ void x3w_split8(const float (&v)[8], float scale, u32x4& hi, u32x4& lo) {
  float y[8];
  f16x2 hh[4], ll[4];
#pragma unroll
  for (int e = 0; e < 8; ++e) y[e] = v[e] * scale;
#pragma unroll
  for (int p = 0; p < 4; ++p) hh[p] = __builtin_convertvector(((f32x2){y[2 * p], y[2 * p + 1]}), f16x2);
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    y[2 * p] = y[2 * p] - (float)hh[p].x;
    y[2 * p + 1] = y[2 * p + 1] - (float)hh[p].y;
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) ll[p] = __builtin_convertvector(((f32x2){y[2 * p], y[2 * p + 1]}), f16x2);
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    hi[d] = __builtin_bit_cast(unsigned, hh[d]);
    lo[d] = __builtin_bit_cast(unsigned, ll[d]);
  }
}

// DBG (timing experiments, results wrong): bit 0 = no vector pieces, bit 2 = no stash stores, bit 3 = no stash loads,
// bit 4 = slot stamps
template <int NB, int DBG>
__global__ __launch_bounds__(64 * WNW, 1) void k_siren_step_x3w(SirenArgs a) {
  using S = X3wShape<NB>;
  constexpr int NG = S::NG, P = S::P;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, h = lane >> 5, j = lane & 31, h8 = h * 8;
  const int L = a.L;
  const int64_t count = a.count_in ? (int64_t)(*a.count_in) : a.n;
  if (count <= a.cnt_lo || count > a.cnt_hi) return;       // another tile shape serves this list (uniform)
  const int64_t n_tiles = (count + P - 1) / P;

  const float* X = a.packed + x3_base(WH, L);
  // Every global access of the hot loops is a buffer instruction: resource in SGPRs, a wave-uniform byte offset in
  // an SGPR (plus an immediate), a 32-bit lane offset in one VGPR -- no 64-bit per-lane addresses to compute, keep
  // alive or spill between the MFMAs.
  const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc((void*)a.packed, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc((void*)a.stash, 0, 0x7fffffff, 0x00020000);
  const int lane16 = lane * 16;
  const int w0_off = __builtin_amdgcn_readfirstlane((int)((x3_base(WH, L) + (int64_t)WSL * w * 64) * 4));         // + (sl*16 + h8 + e)*16
  const int wl_off = __builtin_amdgcn_readfirstlane((int)((x3_base(WH, L) + 4 * WH + (int64_t)WSL * w * 16) * 4)); // + (sl*16 + h8 + e)*4
  auto ld_w0 = [&](int sl, int e) {
    return as_f32x4(__builtin_amdgcn_raw_buffer_load_b128(rP, h8 * 16, w0_off + (sl * 16 + e) * 16, 0));
  };
  auto ld_wl = [&](int sl, int half) {
    return as_f32x4(__builtin_amdgcn_raw_buffer_load_b128(rP, h8 * 4, wl_off + (sl * 16 + 4 * half) * 4, 0));
  };
  const float bL = a.packed[off_bl(WH)];
  const float* hdr = a.packed + x16_base(WH, L);
  const float seed_scale = x3_scale_for(hdr[16] * a.wh * 1.01f);
  const float w0 = a.w0, wh = a.wh;
  // |layer-0 argument| <= w0 * (max_f (|Wx|+|Wy|+|Wz|) * max |coordinate| + max |b|)
  const float z0_w = hdr[17], z0_b = hdr[18];

  // G-stage kg: forward layer kg (kg < L) or the transposed layer 2L-1-kg; byte offset of this wave's share of the image
  auto g_img = [&](int kg) -> int {
    const int l = kg < L ? kg : 2 * L - 1 - kg;
    const int64_t o = (kg < L ? x16_off_layer(WH, L, l) : x16_off_bw(WH, L, l)) * 4 + (int64_t)(WTW * w * 2) * 1024;
    return __builtin_amdgcn_readfirstlane((int)o);
  };
  auto ld_a = [&](u32x4 (&Ar)[WTW][3], int img, int s) {
#pragma unroll
    for (int t = 0; t < WTW; ++t)
#pragma unroll
      for (int c = 0; c < 2; ++c)
        Ar[t][c] = __builtin_amdgcn_raw_buffer_load_b128(rP, lane16, img + s * (WNTO * 2 * 1024) + t * 2048 + c * 1024, 0);
  };
  u32x4* const act = reinterpret_cast<u32x4*>(smem_raw);
  u32x4* const own = act + (size_t)(WSL * w) * NB * 2 * 64 + lane;          // this wave's K-steps of the next layer
  const u32x4* const actl = act + lane;
  auto red_of = [&](int s) { return reinterpret_cast<f32x4*>(smem_raw + S::kActBytes + (size_t)s * S::kSetBytes); };
  auto max_of = [&](int s) { return reinterpret_cast<float*>(smem_raw + S::kActBytes + (size_t)s * S::kSetBytes + S::kRedBytes); };
  // byte offset of (set s, this wave) in the stash: [layer][group][half][lane] f32x4
  auto stash_of = [&](int s) {
    return __builtin_amdgcn_readfirstlane((int)(((((int64_t)blockIdx.x * 2 + s) * WNW + w) * (int64_t)(L + 1) * NG * 128) * 16));
  };
  auto st_ld = [&](int base, int g, int half) {
    return as_f32x4(__builtin_amdgcn_raw_buffer_load_b128(rS, lane16, base + (g * 2 + half) * 1024, 0));
  };
  auto st_st = [&](int base, int g, int half, const f32x4& v) {
    __builtin_amdgcn_raw_buffer_store_b128(as_u32x4(v), rS, lane16, base + (g * 2 + half) * 1024, 0);
  };
  auto tile_of = [&](int s, int64_t c) { return (int64_t)blockIdx.x + (2 * c + s) * (int64_t)gridDim.x; };

  // accumulators of the set in its GEMM; accumulators / parked results of the set in its vector stage
  f32x16 accG[WTW][NB], accV[WTW][NB];
#pragma unroll
  for (int t = 0; t < WTW; ++t)
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) accV[t][n][r] = 0.f;
  // per-set lane state, by role (V = the set in its vector stage this slot, G = the other one); swapped every slot
  float fpartV[NB], bscaleV[NB], amaxV[NB], fpartG[NB], bscaleG[NB], amaxG[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n) { fpartV[n] = fpartG[n] = 0.f; bscaleV[n] = bscaleG[n] = 1.f; amaxV[n] = amaxG[n] = 0.f; }
  int mbufV = 0, mbufG = 0;
  int64_t tileV = -1, tileG = -1;            // tile whose stages are running
  int64_t doneV = -1, doneG = -1;            // tile whose partial sums wait for the epilogue

  u32x4 A[4][WTW][3];                        // weight-fragment pipeline, carried across slots
#pragma unroll
  for (int d = 0; d < X3W_KAD; ++d) ld_a(A[d], g_img(2 * L - 1), d);

  // group g of a lane = K-step sl = g / NB of this wave's share of the next layer, point tile n = g % NB;
  // its eight values are registers 8p..8p+7 of accumulator tile (t, n), sl = 2t + p
  auto park = [&](auto gc, const u32x4& p0, const u32x4& p1) {
    constexpr int g = decltype(gc)::value, sl = g / NB, n = g % NB, t = sl >> 1, p = sl & 1;
    accV[t][n][8 * p + 0] = __uint_as_float(p0.x); accV[t][n][8 * p + 1] = __uint_as_float(p0.y);
    accV[t][n][8 * p + 2] = __uint_as_float(p0.z); accV[t][n][8 * p + 3] = __uint_as_float(p0.w);
    accV[t][n][8 * p + 4] = __uint_as_float(p1.x); accV[t][n][8 * p + 5] = __uint_as_float(p1.y);
    accV[t][n][8 * p + 6] = __uint_as_float(p1.z); accV[t][n][8 * p + 7] = __uint_as_float(p1.w);
  };

  // ---- epilogue of a finished tile (thread = point of the tile) ----------------------------------------------
  auto epilogue = [&](int64_t tile, const f32x4* red) {
    if (tid >= P || tile < 0 || tile >= n_tiles) return;
    bool survive = false;
    int64_t idx = -1;
    const int64_t slot = tile * P + tid;
    if (slot < count) {
      f32x4 r = red[tid];
#pragma unroll
      for (int ww = 1; ww < WNW; ++ww) {
        const f32x4 q = red[ww * P + tid];
        r.x += q.x; r.y += q.y; r.z += q.z; r.w += q.w;
      }
      const float f = r.x + bL;
      idx = a.idx_in ? (int64_t)a.idx_in[slot] : slot;
      survive = iso_step_finish(a, idx, f, r.y, r.z, r.w);
    }
    if (!a.eval_only && a.do_move) {
      const unsigned long long bal = __ballot(survive);
      if (bal) {
        int base = 0;
        const int leader = __ffsll((long long)bal) - 1;
        if (lane == leader) base = atomicAdd(a.count_out, __popcll(bal));
        base = __shfl(base, leader);
        if (survive) {
          const int rank = __popcll(bal & ((1ull << lane) - 1ull));
          a.idx_out[base + rank] = (int32_t)idx;
        }
      }
    }
  };

  // ---- one GEMM stage; piece(kc) is the vector work issued behind the MFMAs of K-step kc ------------------------
  auto gemm = [&](int kg, int kg_next, auto&& piece) {
    const int img = g_img(kg), nimg = g_img(kg_next);
    if (kg < L) {
      const int lay = __builtin_amdgcn_readfirstlane((int)(x3_off_layer(WH, L, kg) * 4));
      const float zscale = kActScale * hdr[kg];
#pragma unroll
      for (int t = 0; t < WTW; ++t) {
        f32x16 init;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int bo = __builtin_amdgcn_readfirstlane(lay + (2 * (WTW * w + t) + p) * 64);
          const f32x4 b0 = as_f32x4(__builtin_amdgcn_raw_buffer_load_b128(rP, h8 * 4, bo, 0));
          const f32x4 b1 = as_f32x4(__builtin_amdgcn_raw_buffer_load_b128(rP, h8 * 4, bo + 16, 0));
#pragma unroll
          for (int e = 0; e < 4; ++e) { init[8 * p + e] = b0[e] * zscale; init[8 * p + 4 + e] = b1[e] * zscale; }
        }
#pragma unroll
        for (int n = 0; n < NB; ++n) accG[t][n] = init;
      }
    } else {
#pragma unroll
      for (int t = 0; t < WTW; ++t)
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
          for (int r = 0; r < 16; ++r) accG[t][n][r] = 0.f;
    }
    u32x4 B[2][NB][3];
    auto ldB = [&](u32x4 (&Br)[NB][3], int s) {
      const u32x4* p = actl + s * (NB * 2 * 64);
#pragma unroll
      for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int c = 0; c < 2; ++c) Br[n][c] = p[(n * 2 + c) * 64];
    };
    ldB(B[0], 0);
    x3w_for<WNS>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      constexpr int jj = k & 3;
      if constexpr (k + X3W_KAD < WNS) ld_a(A[(k + X3W_KAD) & 3], img, k + X3W_KAD);
      else ld_a(A[(k + X3W_KAD) & 3], nimg, k + X3W_KAD - WNS);
      if constexpr (k + 1 < WNS) ldB(B[(k + 1) & 1], k + 1);
      if constexpr (DBG & 128) __builtin_amdgcn_sched_barrier(0);
      // W_l x_h + W_h x_l + W_h x_h, the order of gemm_x3
      constexpr int QA[3] = {1, 0, 0};
      constexpr int QB[3] = {0, 1, 0};
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int t = 0; t < WTW; ++t)
#pragma unroll
          for (int n = 0; n < NB; ++n)
            accG[t][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[jj][t][QA[q]]),
                                                                __builtin_bit_cast(f16x8, B[k & 1][n][QB[q]]), accG[t][n], 0, 0, 0);
      if constexpr (!(DBG & 1)) piece(kc);
      if constexpr (!(DBG & 128))
#pragma unroll
      for (int g = 0; g < 3 * WTW * NB; ++g) {
        X3W_SGB(0x008, 1);
        if (g < 2 * NB) X3W_SGB(0x100, 1);
        else if (g < 2 * NB + 2 * WTW) X3W_SGB(0x020, 1);
        X3W_SGB(0x002, X3W_VALU_PER_MFMA);
      }
      __builtin_amdgcn_sched_barrier(0);
      x3_keep_alive<WTW, 2>(A[jj]);
      x3_keep_alive<NB, 2>(B[k & 1]);
    });
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (DBG & 1) {
#pragma unroll
      for (int t = 0; t < WTW; ++t)
#pragma unroll
        for (int n = 0; n < NB; ++n) asm volatile("" ::"v"(accG[t][n]));
    }
#endif
  };
  auto no_piece = [](auto) {};
  // DBG bit 5: the vector stage first, then the GEMM on its own (same results, nothing overlaps)
  auto run = [&](int kg, int kg_next, auto&& piece) {
    if constexpr (DBG & 32) { x3w_for<WNS>(piece); gemm(kg, kg_next, no_piece); }
    else gemm(kg, kg_next, piece);
  };

  int sV = 0, kV = 0;
  int64_t c = 0;
  bool last = false;
  int dbg_t = 0;
  long long* dbg_buf = reinterpret_cast<long long*>(reinterpret_cast<f32x4*>(a.stash) + ((int64_t)(1 * WNW + 3) * (L + 1) + L) * NG * 128);
  while (true) {
    if constexpr (DBG & 16) {
      if (blockIdx.x == 0 && tid == 0 && dbg_t < 250) dbg_buf[dbg_t] = (long long)__builtin_amdgcn_s_memtime();
      ++dbg_t;
    }
    const int kg = sV == 0 ? (kV == 0 ? 2 * L - 1 : kV - 1) : kV;
    // the slot after [V set0 at k] is [V set1 at k], whose GEMM is set0's stage k; after [V set1 at k] comes
    // [V set0 at k+1], whose GEMM is set1's stage k: either way the next GEMM stage is kV
    const int kg_after = kV;
    float* const maxV = max_of(sV);
    f32x4* const redV = red_of(sV);
    const int stashV = stash_of(sV);

    auto put_amax = [&](int buf) {
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        const float m = __builtin_fmaxf(amaxV[n], __shfl_xor(amaxV[n], 32));
        if (h == 0) maxV[(buf * P + 32 * n + j) * WNW + w] = m;
      }
    };

    if (kV == 1) epilogue(doneV, redV);

    if (kV == 0) {
      // ---- V0: reverse activation of layer 0 of the finished tile, layer 0 of the next one -------------------
      const int64_t tile_new = tile_of(sV, c);
      if (sV == 0) last = tile_new >= n_tiles;
      const float iw = 1.0f / hdr[0];
      float inv[NB], gx[NB], gy[NB], gz[NB], px[NB], py[NB], pz[NB];
      float pmax = 0.f;
#pragma unroll
      for (int n = 0; n < NB; ++n) { inv[n] = iw / bscaleV[n]; gx[n] = gy[n] = gz[n] = 0.f; }
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        const int64_t slot = tile_new * P + 32 * n + j;
        px[n] = py[n] = pz[n] = 0.f;
        if (slot < count) {
          const int64_t idx = a.idx_in ? (int64_t)a.idx_in[slot] : slot;
          px[n] = a.pts[idx * 3]; py[n] = a.pts[idx * 3 + 1]; pz[n] = a.pts[idx * 3 + 2];
        }
        pmax = __builtin_fmaxf(pmax, __builtin_fmaxf(__builtin_fabsf(px[n]), __builtin_fmaxf(__builtin_fabsf(py[n]), __builtin_fabsf(pz[n]))));
      }
      // an argument beyond the range of the fast sin / cos reduction is impossible unless this bound says otherwise
      const bool slow = (DBG & 64) ? true : __any(!(w0 * (z0_w * pmax + z0_b) < 1.0e4f));
      f32x4 wv[8], sv[NB][2];
      X3wSc T;
      float zz[NB][8];
      auto ld_block = [&](int sl) {          // W0 rows of K-step sl of this wave, w cos(w z) of layer 0 of its groups
#pragma unroll
        for (int e = 0; e < 8; ++e) wv[e] = ld_w0(sl, e);
#pragma unroll
        for (int n = 0; n < NB; ++n) {
          if constexpr (DBG & 8) { sv[n][0] = sv[n][1] = (f32x4){1.f, 1.f, 1.f, (float)sl}; continue; }
          sv[n][0] = st_ld(stashV, sl * NB + n, 0);
          sv[n][1] = st_ld(stashV, sl * NB + n, 1);
        }
      };
      // the gradient of the finished tile for K-step sl (all point tiles)
      auto rev0 = [&](auto slc) {
        constexpr int sl = decltype(slc)::value, t = sl >> 1, p = sl & 1;
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float av = (accV[t][n][8 * p + e] * inv[n]) * sv[n][e >> 2][e & 3];
            gx[n] += wv[e].x * av;
            gy[n] += wv[e].y * av;
            gz[n] += wv[e].z * av;
          }
      };
      auto zz0 = [&]() {
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
          for (int e = 0; e < 8; ++e) zz[n][e] = ((wv[e].x * px[n] + wv[e].y * py[n]) + wv[e].z * pz[n]) + wv[e].w;
      };
      // phase ph (0..3) of the activation of group (sl, n) of the next tile
      auto act0 = [&](auto slc, auto nc, auto phc) {
        constexpr int sl = decltype(slc)::value, n = decltype(nc)::value, ph = decltype(phc)::value, g = sl * NB + n;
        if constexpr (ph == 0) x3w_sc<0>(T, zz[n], w0, w0);
        if constexpr (ph == 1) x3w_sc<1>(T, zz[n], w0, w0);
        if constexpr (ph == 2) {
          x3w_sc<2>(T, zz[n], w0, w0);
          if constexpr (!(DBG & 4)) {
            st_st(stashV, g, 0, (f32x4){T.c[0], T.c[1], T.c[2], T.c[3]});
            st_st(stashV, g, 1, (f32x4){T.c[4], T.c[5], T.c[6], T.c[7]});
          }
        }
        if constexpr (ph == 3) {
          u32x4 p0, p1;
          x3w_split8(T.s, kActScale, p0, p1);
          park(std::integral_constant<int, g>{}, p0, p1);
        }
      };
      ld_block(0);
      if (__builtin_expect(!slow, 1)) {
        auto piece = [&](auto kc) {
          constexpr int k = decltype(kc)::value;
          constexpr int BL = WNS / WSL;                     // K-steps per K-step sl of this wave's share
          constexpr int sl = k / BL, kk = k % BL;
          constexpr int NPH = 4 * NB;
          if constexpr (kk == 0) rev0(std::integral_constant<int, sl>{});
          if constexpr (kk == 1) zz0();
          constexpr int lo = kk >= 1 ? ((kk - 1) * NPH) / (BL - 1) : 0, hi = kk >= 1 ? (kk * NPH) / (BL - 1) : 0;
          if constexpr (kk >= 1)
            x3w_for<hi - lo>([&](auto ic) {
              constexpr int q = lo + decltype(ic)::value;
              act0(std::integral_constant<int, sl>{}, std::integral_constant<int, q / 4>{}, std::integral_constant<int, q % 4>{});
            });
          if constexpr (kk == 2 && sl + 1 < WSL) ld_block(sl + 1);
        };
        run(kg, kg_after, piece);
      } else {
        // layer 0 with the library fix-up of iso_sin_wcos8 for huge arguments, then the GEMM on its own
#pragma unroll
        for (int sl = 0; sl < WSL; ++sl) {
          if (sl > 0) ld_block(sl);
#pragma unroll
          for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float av = (accV[sl >> 1][n][8 * (sl & 1) + e] * inv[n]) * sv[n][e >> 2][e & 3];
              gx[n] += wv[e].x * av;
              gy[n] += wv[e].y * av;
              gz[n] += wv[e].z * av;
            }
          zz0();
#pragma unroll
          for (int n = 0; n < NB; ++n) {
            const int g = sl * NB + n;
            float hv[8], cv[8];
            iso_sin_wcos8(w0, w0, zz[n], hv, cv);
            st_st(stashV, g, 0, (f32x4){cv[0], cv[1], cv[2], cv[3]});
            st_st(stashV, g, 1, (f32x4){cv[4], cv[5], cv[6], cv[7]});
            u32x4 p0, p1;
            split8_f16(hv, p0, p1);
            {
              accV[sl >> 1][n][8 * (sl & 1) + 0] = __uint_as_float(p0.x); accV[sl >> 1][n][8 * (sl & 1) + 1] = __uint_as_float(p0.y);
              accV[sl >> 1][n][8 * (sl & 1) + 2] = __uint_as_float(p0.z); accV[sl >> 1][n][8 * (sl & 1) + 3] = __uint_as_float(p0.w);
              accV[sl >> 1][n][8 * (sl & 1) + 4] = __uint_as_float(p1.x); accV[sl >> 1][n][8 * (sl & 1) + 5] = __uint_as_float(p1.y);
              accV[sl >> 1][n][8 * (sl & 1) + 6] = __uint_as_float(p1.z); accV[sl >> 1][n][8 * (sl & 1) + 7] = __uint_as_float(p1.w);
            }
          }
        }
        gemm(kg, kg_after, no_piece);
      }
      __syncthreads();                       // the GEMM has read the buffer for the last time
      // partial sums of the finished tile
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        const float f = fpartV[n] + __shfl_xor(fpartV[n], 32);
        const float x = gx[n] + __shfl_xor(gx[n], 32);
        const float y = gy[n] + __shfl_xor(gy[n], 32);
        const float z = gz[n] + __shfl_xor(gz[n], 32);
        if (h == 0) redV[w * P + 32 * n + j] = (f32x4){f, x, y, z};
        fpartV[n] = 0.f;
      }
      doneV = tileV;
      tileV = tile_new;
    } else if (kV <= L) {
      // ---- forward activation of layer kV-1 (top layer: head dot product and adjoint seed) ---------------------
      const int l = kV - 1;
      const bool top = kV == L;
      const float w_in = wh / (kActScale * hdr[l]);
      const int st_l = stashV + (l + 1) * NG * 2048;
      float zmax = 0.f;
#pragma unroll
      for (int t = 0; t < WTW; ++t)
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
          for (int r = 0; r < 16; r += 2)
            zmax = __builtin_fmaxf(zmax, __builtin_fmaxf(__builtin_fabsf(accV[t][n][r]), __builtin_fabsf(accV[t][n][r + 1])));
      const bool slow = (DBG & 64) ? true : __any(!(zmax * __builtin_fabsf(w_in) < 1.0e4f));
      if (top) {
#pragma unroll
        for (int n = 0; n < NB; ++n) { amaxV[n] = 0.f; bscaleV[n] = seed_scale; }
      }
      X3wSc T;
      f32x4 wl0, wl1;
      if (__builtin_expect(slow, 0)) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const int sl = g / NB, n = g % NB, t = sl >> 1, p = sl & 1;
          float z8[8], hv[8], cv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) z8[e] = accV[t][n][8 * p + e];
          iso_sin_wcos8(w_in, wh, z8, hv, cv);
          u32x4 p0, p1;
          if (top) {
            const f32x4 v0 = ld_wl(sl, 0);
            const f32x4 v1 = ld_wl(sl, 1);
            const float f0 = (v0.x * hv[0] + v0.y * hv[1]) + (v0.z * hv[2] + v0.w * hv[3]);
            const float f1 = (v1.x * hv[4] + v1.y * hv[5]) + (v1.z * hv[6] + v1.w * hv[7]);
            fpartV[n] += f0 + f1;
#pragma unroll
            for (int e = 0; e < 4; ++e) { hv[e] = v0[e] * cv[e]; hv[4 + e] = v1[e] * cv[4 + e]; }
            float m = amaxV[n];
#pragma unroll
            for (int e = 0; e < 8; e += 2) m = __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(hv[e]), __builtin_fabsf(hv[e + 1])));
            amaxV[n] = m;
            split8_f16(hv, p0, p1, seed_scale);
          } else {
            st_st(st_l, g, 0, (f32x4){cv[0], cv[1], cv[2], cv[3]});
            st_st(st_l, g, 1, (f32x4){cv[4], cv[5], cv[6], cv[7]});
            split8_f16(hv, p0, p1);
          }
          accV[t][n][8 * p + 0] = __uint_as_float(p0.x); accV[t][n][8 * p + 1] = __uint_as_float(p0.y);
          accV[t][n][8 * p + 2] = __uint_as_float(p0.z); accV[t][n][8 * p + 3] = __uint_as_float(p0.w);
          accV[t][n][8 * p + 4] = __uint_as_float(p1.x); accV[t][n][8 * p + 5] = __uint_as_float(p1.y);
          accV[t][n][8 * p + 6] = __uint_as_float(p1.z); accV[t][n][8 * p + 7] = __uint_as_float(p1.w);
        }
        gemm(kg, kg_after, no_piece);
      } else if (!top) {
        auto piece = [&](auto kc) {
          constexpr int k = decltype(kc)::value;
          constexpr int QLO = (k * NG * 4) / WNS, QHI = ((k + 1) * NG * 4) / WNS;
          x3w_for<QHI - QLO>([&](auto ic) {
            constexpr int q = QLO + decltype(ic)::value, g = q >> 2, ph = q & 3;
            constexpr int sl = g / NB, n = g % NB, t = sl >> 1, p = sl & 1;
            float zz[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) zz[e] = accV[t][n][8 * p + e];
            if constexpr (ph == 0) x3w_sc<0>(T, zz, w_in, wh);
            if constexpr (ph == 1) x3w_sc<1>(T, zz, w_in, wh);
            if constexpr (ph == 2) {
              x3w_sc<2>(T, zz, w_in, wh);
              if constexpr (!(DBG & 4)) {
                st_st(st_l, g, 0, (f32x4){T.c[0], T.c[1], T.c[2], T.c[3]});
                st_st(st_l, g, 1, (f32x4){T.c[4], T.c[5], T.c[6], T.c[7]});
              }
            }
            if constexpr (ph == 3) {
              u32x4 p0, p1;
              x3w_split8(T.s, kActScale, p0, p1);
              park(std::integral_constant<int, g>{}, p0, p1);
            }
          });
        };
        run(kg, kg_after, piece);
      } else {
        auto piece = [&](auto kc) {
          constexpr int k = decltype(kc)::value;
          constexpr int QLO = (k * NG * 4) / WNS, QHI = ((k + 1) * NG * 4) / WNS;
          x3w_for<QHI - QLO>([&](auto ic) {
            constexpr int q = QLO + decltype(ic)::value, g = q >> 2, ph = q & 3;
            constexpr int sl = g / NB, n = g % NB, t = sl >> 1, p = sl & 1;
            float zz[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) zz[e] = accV[t][n][8 * p + e];
            if constexpr (ph == 0) {
              wl0 = ld_wl(sl, 0);
              wl1 = ld_wl(sl, 1);
              x3w_sc<0>(T, zz, w_in, wh);
            }
            if constexpr (ph == 1) x3w_sc<1>(T, zz, w_in, wh);
            if constexpr (ph == 2) {
              x3w_sc<2>(T, zz, w_in, wh);
              const float f0 = (wl0.x * T.s[0] + wl0.y * T.s[1]) + (wl0.z * T.s[2] + wl0.w * T.s[3]);
              const float f1 = (wl1.x * T.s[4] + wl1.y * T.s[5]) + (wl1.z * T.s[6] + wl1.w * T.s[7]);
              fpartV[n] += f0 + f1;
            }
            if constexpr (ph == 3) {
              float hv[8];
#pragma unroll
              for (int e = 0; e < 4; ++e) { hv[e] = wl0[e] * T.c[e]; hv[4 + e] = wl1[e] * T.c[4 + e]; }
              float m = amaxV[n];
#pragma unroll
              for (int e = 0; e < 8; e += 2) m = __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(hv[e]), __builtin_fabsf(hv[e + 1])));
              amaxV[n] = m;
              u32x4 p0, p1;
              x3w_split8(hv, seed_scale, p0, p1);
              park(std::integral_constant<int, g>{}, p0, p1);
            }
          });
        };
        run(kg, kg_after, piece);
      }
      __syncthreads();                       // the GEMM has read the buffer for the last time
      if (top) { put_amax(0); mbufV = 0; }
    } else {
      // ---- reverse activation of layer l = 2L - kV >= 1: adjoint * w cos(w z), per-point scale, split ------------
      const int l = 2 * L - kV;
      const int st_l = stashV + l * NG * 2048;
      const float iw = 1.0f / hdr[l];
      const float grow = hdr[8 + l] * wh * 1.01f;
      float inv[NB], nscale[NB];
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        float m = 0.f;
#pragma unroll
        for (int ww = 0; ww < WNW; ++ww) m = __builtin_fmaxf(m, maxV[(mbufV * P + 32 * n + j) * WNW + ww]);
        inv[n] = iw / bscaleV[n];
        nscale[n] = x3_scale_for(m * grow);
        amaxV[n] = 0.f;
      }
      // w cos(w z) of the layer below, requested kRA groups ahead
      constexpr int kRA = 3;
      f32x4 sv[kRA + 1][2];
      auto ld_sv = [&](int g) {
        if constexpr (DBG & 8) { sv[g % (kRA + 1)][0] = sv[g % (kRA + 1)][1] = (f32x4){1.f, 1.f, 1.f, (float)l}; return; }
        sv[g % (kRA + 1)][0] = st_ld(st_l, g, 0);
        sv[g % (kRA + 1)][1] = st_ld(st_l, g, 1);
      };
#pragma unroll
      for (int g = 0; g < kRA; ++g) ld_sv(g);
      float av[8];
      auto piece = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int QLO = (k * NG * 4) / WNS, QHI = ((k + 1) * NG * 4) / WNS;
        x3w_for<QHI - QLO>([&](auto ic) {
          constexpr int q = QLO + decltype(ic)::value, g = q >> 2, ph = q & 3;
          constexpr int sl = g / NB, n = g % NB, t = sl >> 1, p = sl & 1;
          if constexpr (ph == 0 && g + kRA < NG) ld_sv(g + kRA);
          if constexpr (ph == 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) av[e] = (accV[t][n][8 * p + e] * inv[n]) * sv[g % (kRA + 1)][e >> 2][e & 3];
            float m = amaxV[n];
#pragma unroll
            for (int e = 0; e < 8; e += 2) m = __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(av[e]), __builtin_fabsf(av[e + 1])));
            amaxV[n] = m;
          }
          if constexpr (ph == 3) {
            u32x4 p0, p1;
            x3w_split8(av, nscale[n], p0, p1);
            park(std::integral_constant<int, g>{}, p0, p1);
          }
        });
      };
      run(kg, kg_after, piece);
      __syncthreads();                       // the GEMM has read the buffer for the last time
#pragma unroll
      for (int n = 0; n < NB; ++n) bscaleV[n] = nscale[n];
      put_amax(mbufV ^ 1);
      mbufV ^= 1;
    }
    // ---- the parked results become the buffer's contents: the input of this set's next GEMM --------------------
    x3w_for<NG>([&](auto gc) {
      constexpr int g = decltype(gc)::value, sl = g / NB, n = g % NB, t = sl >> 1, p = sl & 1;
      const u32x4 p0 = {__float_as_uint(accV[t][n][8 * p + 0]), __float_as_uint(accV[t][n][8 * p + 1]),
                        __float_as_uint(accV[t][n][8 * p + 2]), __float_as_uint(accV[t][n][8 * p + 3])};
      const u32x4 p1 = {__float_as_uint(accV[t][n][8 * p + 4]), __float_as_uint(accV[t][n][8 * p + 5]),
                        __float_as_uint(accV[t][n][8 * p + 6]), __float_as_uint(accV[t][n][8 * p + 7])};
      own[(g * 2 + 0) * 64] = p0; own[(g * 2 + 1) * 64] = p1;
    });
    __syncthreads();
    if constexpr (DBG == 256) {
      if (dbg_t == 0 && blockIdx.x == 0) {
        u32x4* dst = reinterpret_cast<u32x4*>(a.stash) + (int64_t)100 * 2 * WNW * (L + 1) * NG * 128;
        for (int i = tid; i < (int)(S::kActBytes / 16); i += 256) dst[i] = act[i];
      }
      ++dbg_t;
    }
    if (last && kV == 1 && sV == 1) break;
    // ---- the sets change roles: the fresh accumulators are the next vector stage's input -------------------
#pragma unroll
    for (int t = 0; t < WTW; ++t)
#pragma unroll
      for (int n = 0; n < NB; ++n) accV[t][n] = accG[t][n];
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      float u;
      u = fpartV[n]; fpartV[n] = fpartG[n]; fpartG[n] = u;
      u = bscaleV[n]; bscaleV[n] = bscaleG[n]; bscaleG[n] = u;
      u = amaxV[n]; amaxV[n] = amaxG[n]; amaxG[n] = u;
    }
    { const int u = mbufV; mbufV = mbufG; mbufG = u; }
    { const int64_t u = tileV; tileV = tileG; tileG = u; }
    { const int64_t u = doneV; doneV = doneG; doneG = u; }
    if (sV == 1) { if (++kV == 2 * L) { kV = 0; ++c; } }
    sV ^= 1;
  }
}

#ifndef X3W_NB
#define X3W_NB 3
#endif

}  // namespace

int64_t siren_x3w_stash_floats(int L) { return 256 * X3wShape<X3W_NB>::stash_per_wg(L); }

int siren_x3w_launch(const SirenArgs& a, int64_t n_upper, hipStream_t s) {
  using S = X3wShape<X3W_NB>;
  static int dbg = -1;
  if (dbg < 0) {
    const char* e = getenv("ISO_X3D_DBG");
    dbg = e ? atoi(e) : 0;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_siren_step_x3w<X3W_NB, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::kLds);
#ifdef X3D_EXPERIMENTS
#define X3W_ATTR(D) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_siren_step_x3w<X3W_NB, D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::kLds);
    X3W_ATTR(1) X3W_ATTR(4) X3W_ATTR(8) X3W_ATTR(12) X3W_ATTR(16) X3W_ATTR(17) X3W_ATTR(28) X3W_ATTR(32) X3W_ATTR(64) X3W_ATTR(160) X3W_ATTR(256)
#endif
  }
  const int64_t tiles = (n_upper + S::P - 1) / S::P;
  int blocks = (int)(tiles < 256 ? (tiles < 1 ? 1 : tiles) : 256);
#ifdef X3D_EXPERIMENTS
  { const char* e = getenv("ISO_X3D_BLOCKS"); if (e && atoi(e) > 0 && atoi(e) < blocks) blocks = atoi(e); }
#define X3W_RUN(D) if (dbg == D) { hipLaunchKernelGGL((k_siren_step_x3w<X3W_NB, D>), dim3(blocks), dim3(64 * WNW), S::kLds, s, a); return 0; }
  X3W_RUN(1) X3W_RUN(4) X3W_RUN(8) X3W_RUN(12) X3W_RUN(16) X3W_RUN(17) X3W_RUN(28) X3W_RUN(32) X3W_RUN(64) X3W_RUN(160) X3W_RUN(256)
#endif
  hipLaunchKernelGGL((k_siren_step_x3w<X3W_NB, 0>), dim3(blocks), dim3(64 * WNW), S::kLds, s, a);
  return 0;
}
