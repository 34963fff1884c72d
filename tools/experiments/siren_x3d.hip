// EXPERIMENT (round 2), not compiled into the library -- kept because it is correct (bit-identical to k_siren_step_x3 on
// every case of tools/siren_dual_check-style comparisons, L = 1..3, ragged and chaotic-weight clouds) and because its
// measurements decide the design space ("two sets side by side"): 3.33 ms per 1 M evaluations against
// 2.99 ms for the one-set kernel.  To build it again: copy to iso_points_amd/csrc/, compile with -fno-slp-vectorize,
// declare siren_x3d_stash_floats / siren_x3d_launch in siren_common.h and call them from siren_x3_launch /
// siren_x3_stash_floats.  -DX3D_EXPERIMENTS adds the timing variants (ISO_X3D_DBG, ISO_X3D_BLOCKS).
// Fused SIREN SDF + gradient Newton step, H = 256: the split-fp16 kernel of siren_x3.hip with the matrix and the
// vector stages running SIDE BY SIDE.  Same reference semantics (Siren.forward DSS/models/common.py:140-165 under
// autograd.grad, levelset_sampling.py:142-170, iterated by _project_points :313-342), same per-point arithmetic,
// bit for bit (every operation of a point is the one k_siren_step_x3 performs, in the same order), so either kernel
// can serve any launch.
//
// In k_siren_step_x3 a tile alternates between GEMM stages (matrix pipe busy, VALU idle) and activation stages
// (sin / cos, fp16 cuts: VALU busy, matrix pipe idle) of about equal length.  tools/probes/coissue.hip: with two
// waves per SIMD, the 104 plain-f32 VALU instructions of one 8-value activation group issued between 24 MFMAs
// cost 5 % on top of the MFMAs alone (1615 against 1538 cycles; 2438 when run one after the other) -- but only
// as plain ops: packed f32 ops collide with the matrix pipe (2115 cycles).  So here a workgroup carries TWO sets of
// 64 points, one stage apart: while the GEMM of one set runs, every wave issues the activation work of the other
// set between its own MFMAs, five VALU slots behind each MFMA (sched_group_barrier).  This file is compiled with
// -fno-slp-vectorize so that the vector stages stay plain f32.
//
//   per set (L hidden layers):  V0  G1 V1  G2 V2 ... G_L V_L(top)  G_{L+1} V_{L+1} ... G_{2L}   (4L stages)
//     V0     = reverse activation of layer 0 of the set's PREVIOUS tile (gradient, partial sums -> LDS),
//              then layer 0 (3 -> H) of its next tile
//     G_k    = hidden GEMM k (forward layers 0..L-1, then the transposed ones L-1..0)
//     V_k    = sin / w cos / split of the GEMM before it (top layer: head + adjoint seed; reverse: stash * adjoint)
//     V1     also carries the epilogue (cross-wave sum, Newton move, survivor list) of the previous tile
//   the two sets are one stage apart, every slot is [G of one set || V of the other], one barrier per slot:
//   a V stage overwrites its set's activation buffer in place while only the other set's buffer is read.
//
// LDS: 2 x (64 KiB activations + 8 KiB partial sums + 4 KiB adjoint maxima) = 152 KiB, one workgroup of 8 waves
// per CU; a weight fragment feeds two point tiles (three in k_siren_step_x3): 24 KiB of weights per point
// evaluation cross L2 -> CU.
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "siren_common.h"
#include "iso_newton.h"
#include "mlp_common.h"
#include "mfma_split.h"

namespace {

constexpr int DH = 256, DNW = 8, DNB = 2, DNS = 16, DNTO = 8, DSL = 2, DNG = 4, DP = 64;
constexpr size_t kDActBytes = (size_t)DNS * DNB * 2 * 1024;     // [K-step][point tile][part][lane] u32x4
constexpr size_t kDRedBytes = (size_t)DNW * DP * 16;            // [wave][point] {f, gx, gy, gz}
constexpr size_t kDMaxBytes = (size_t)2 * DP * DNW * 4;         // [buffer][point][wave] max |adjoint|
constexpr size_t kDSetBytes = kDActBytes + kDRedBytes + kDMaxBytes;
constexpr size_t kDLds = 2 * kDSetBytes;
static_assert(kDLds <= 160 * 1024, "two sets must fit the LDS of a CU");
__host__ __device__ constexpr int64_t x3d_stash_per_wg(int L) { return (int64_t)2 * DNW * (L + 1) * DNG * 512; }   // floats

typedef __attribute__((address_space(1))) f32x4* gf4_t;
typedef const __attribute__((address_space(1))) f32x4* gcf4_t;
typedef const __attribute__((address_space(1))) float* gcf_t;

template <class F, int... I>
__device__ __forceinline__ void x3d_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void x3d_for(F&& f) { x3d_for_impl(f, std::make_integer_sequence<int, N>{}); }

#define X3D_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#ifndef X3D_VALU_PER_MFMA
#define X3D_VALU_PER_MFMA 5
#endif
// weight-fragment pipeline: X3D_NSETS rotating register sets, requested X3D_KAD K-steps ahead
#ifndef X3D_NSETS
#define X3D_NSETS 4
#endif
#ifndef X3D_KAD
#define X3D_KAD 3
#endif
static_assert(X3D_KAD < X3D_NSETS && 16 % X3D_NSETS == 0, "the rotation must close over the 16 K-steps of a stage");

// ---- the activation of eight values, plain f32 ops, in three phases (iso_sin_wcos8 element by element) --------
struct X3dSc { float x[8], f[8], s[8], c[8]; };

template <int PH>
__device__ __forceinline__ void x3d_sc(X3dSc& T, const float (&z)[8], float w_in, float w, float& big) {
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
    for (int e = 0; e < 8; e += 2) big = __builtin_fmaxf(big, __builtin_fmaxf(__builtin_fabsf(T.x[e]), __builtin_fabsf(T.x[e + 1])));
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
__device__ __forceinline__ void x3d_split8(const float (&v)[8], float scale, u32x4& hi, u32x4& lo) {
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

enum { kVV0 = 0, kVFwd = 1, kVTop = 2, kVRev = 3 };

// DBG (timing experiments, results wrong): bit 0 = no vector pieces behind the MFMAs, bit 1 = no MFMAs
template <int DBG>
__global__ __launch_bounds__(64 * DNW, 1) void k_siren_step_x3d(SirenArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, h = lane >> 5, j = lane & 31, h8 = h * 8;
  const int L = a.L;
  const int64_t count = a.count_in ? (int64_t)(*a.count_in) : a.n;
  if (count <= a.cnt_lo || count > a.cnt_hi) return;       // another tile shape serves this list (uniform)
  const int64_t n_tiles = (count + DP - 1) / DP;

  const float* X = a.packed + x3_base(DH, L);
  const gcf4_t W0u = (gcf4_t)(reinterpret_cast<const f32x4*>(X) + DSL * w * 16);     // + sl*16 + h8 + e
  const gcf_t WLu = (gcf_t)(X + 4 * DH + DSL * w * 16);                              // + sl*16 + h8 + e
  const float bL = a.packed[off_bl(DH)];
  const float* hdr = a.packed + x16_base(DH, L);
  const float seed_scale = x3_scale_for(hdr[16] * a.wh * 1.01f);
  const float w0 = a.w0, wh = a.wh;

  // G-stage kg: forward layer kg (kg < L) or the transposed layer 2L-1-kg
  auto g_img = [&](int kg) -> gimg_t {
    const int l = kg < L ? kg : 2 * L - 1 - kg;
    const float* p = a.packed + (kg < L ? x16_off_layer(DH, L, l) : x16_off_bw(DH, L, l));
    return (gimg_t)(reinterpret_cast<const u32x4*>(p) + (w * 2) * 64);
  };
  // LDS regions / stash of set s
  auto act_of = [&](int s) { return reinterpret_cast<u32x4*>(smem_raw + (size_t)s * kDSetBytes); };
  auto red_of = [&](int s) { return reinterpret_cast<f32x4*>(smem_raw + (size_t)s * kDSetBytes + kDActBytes); };
  auto max_of = [&](int s) { return reinterpret_cast<float*>(smem_raw + (size_t)s * kDSetBytes + kDActBytes + kDRedBytes); };
  auto stash_of = [&](int s) {
    return (gf4_t)(reinterpret_cast<f32x4*>(a.stash) +
                   (((int64_t)blockIdx.x * 2 + s) * DNW + w) * (int64_t)(L + 1) * DNG * 128) + lane;
  };
  auto tile_of = [&](int s, int64_t c) { return (int64_t)blockIdx.x + (2 * c + s) * (int64_t)gridDim.x; };

  f32x16 accG[DNB], accV[DNB];
#pragma unroll
  for (int n = 0; n < DNB; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) accV[n][r] = 0.f;
  // per-set lane state, by role (V = the set in its vector stage this slot, G = the other one); swapped every slot
  float fpartV[DNB], bscaleV[DNB], amaxV[DNB], fpartG[DNB], bscaleG[DNB], amaxG[DNB];
#pragma unroll
  for (int n = 0; n < DNB; ++n) { fpartV[n] = fpartG[n] = 0.f; bscaleV[n] = bscaleG[n] = 1.f; amaxV[n] = amaxG[n] = 0.f; }
  int mbufV = 0, mbufG = 0;
  int64_t tileV = -1, tileG = -1;            // tile whose stages are running
  int64_t doneV = -1, doneG = -1;            // tile whose partial sums wait for the epilogue
  float px[DNB], py[DNB], pz[DNB];

  u32x4 A[X3D_NSETS][1][3];                  // weight-fragment pipeline, carried across slots
#pragma unroll
  for (int d = 0; d < X3D_KAD; ++d) x3_load_a<1, DNTO, 2>(A[d], g_img(2 * L - 1), d, lane);

  // ---- epilogue of a finished tile (wave 0: lane = point of the tile) ----------------------------------------
  auto epilogue = [&](int64_t tile, const f32x4* red) {
    if (w != 0 || tile < 0 || tile >= n_tiles) return;
    bool survive = false;
    int64_t idx = -1;
    const int64_t slot = tile * DP + lane;
    if (slot < count) {
      f32x4 r = red[lane];
#pragma unroll
      for (int ww = 1; ww < DNW; ++ww) {
        const f32x4 q = red[ww * DP + lane];
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

  // ---- one slot: GEMM kg of the G set || vector stage KIND of the V set ---------------------------------------
  // piece(kc) is the vector work issued behind the MFMAs of K-step kc
  auto gemm = [&](int kg, int kg_next, const u32x4* actl, auto&& piece) {
    const gimg_t img = g_img(kg), nimg = g_img(kg_next);
    if (kg < L) {
      const float* lay = a.packed + x3_off_layer(DH, L, kg);
      const float zscale = kActScale * hdr[kg];
      f32x16 init;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const float* bp = lay + (2 * w + p) * 16 + h8;
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(bp + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { init[8 * p + e] = b0[e] * zscale; init[8 * p + 4 + e] = b1[e] * zscale; }
      }
#pragma unroll
      for (int n = 0; n < DNB; ++n) accG[n] = init;
    } else {
#pragma unroll
      for (int n = 0; n < DNB; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) accG[n][r] = 0.f;
    }
    u32x4 B[2][DNB][3];
    auto ldB = [&](u32x4 (&Br)[DNB][3], int s) {
      const u32x4* p = actl + s * (DNB * 2 * 64);
#pragma unroll
      for (int n = 0; n < DNB; ++n)
#pragma unroll
        for (int c = 0; c < 2; ++c) Br[n][c] = p[(n * 2 + c) * 64];
    };
    ldB(B[0], 0);
    x3d_for<DNS>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      constexpr int jj = k % X3D_NSETS;
      if constexpr (k + X3D_KAD < DNS) x3_load_a<1, DNTO, 2>(A[(k + X3D_KAD) % X3D_NSETS], img, k + X3D_KAD, lane);
      else x3_load_a<1, DNTO, 2>(A[(k + X3D_KAD) % X3D_NSETS], nimg, k + X3D_KAD - DNS, lane);
      if constexpr (k + 1 < DNS) ldB(B[(k + 1) & 1], k + 1);
      // W_l x_h + W_h x_l + W_h x_h, the order of gemm_x3
      constexpr int QA[3] = {1, 0, 0};
      constexpr int QB[3] = {0, 1, 0};
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int n = 0; n < DNB; ++n) {
          if constexpr (DBG & 2) accG[n][q] += __builtin_bit_cast(f32x4, A[jj][0][QA[q]]).x * __builtin_bit_cast(f32x4, B[k & 1][n][QB[q]]).y;
          else
          accG[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[jj][0][QA[q]]),
                                                           __builtin_bit_cast(f16x8, B[k & 1][n][QB[q]]), accG[n], 0, 0, 0);
        }
      if constexpr (!(DBG & 1)) piece(kc);
#pragma unroll
      for (int g = 0; g < 3 * DNB; ++g) {
        X3D_SGB(0x008, 1);
        if (g < 2 * DNB) X3D_SGB(0x100, 1); else X3D_SGB(0x020, 1);
        X3D_SGB(0x002, X3D_VALU_PER_MFMA);
      }
      __builtin_amdgcn_sched_barrier(0);
      x3_keep_alive<1, 2>(A[jj]);
      x3_keep_alive<DNB, 2>(B[k & 1]);
    });
    if constexpr (DBG & 1) {
#pragma unroll
      for (int n = 0; n < DNB; ++n) asm volatile("" ::"v"(accG[n]));
    }
  };

  int sV = 0, kV = 0;
  int64_t c = 0;
  bool last = false;
  int dbg_t = 0;
  long long* dbg_buf = reinterpret_cast<long long*>(reinterpret_cast<f32x4*>(a.stash) + ((int64_t)(1 * DNW + 7) * (L + 1) + L) * DNG * 128);
  while (true) {
    if constexpr (DBG & 16) {
      if (blockIdx.x == 0 && tid == 0 && dbg_t < 250) dbg_buf[dbg_t] = (long long)__builtin_amdgcn_s_memtime();
      ++dbg_t;
    }
    const int kg = sV == 0 ? (kV == 0 ? 2 * L - 1 : kV - 1) : kV;
    // the slot after [V set0 at k] is [V set1 at k], whose GEMM is set0's stage k; after [V set1 at k] comes
    // [V set0 at k+1], whose GEMM is set1's stage k: either way the next GEMM stage is kV
    const int kg_after = kV;
    u32x4* const actV = act_of(sV);
    u32x4* const ownV = actV + (size_t)(DSL * w) * DNB * 2 * 64 + lane;
    const u32x4* const actlG = act_of(sV ^ 1) + lane;
    float* const maxV = max_of(sV);
    f32x4* const redV = red_of(sV);
    const gf4_t stashV = stash_of(sV);
    float big = 0.f;

    auto put_amax = [&](int buf) {
#pragma unroll
      for (int n = 0; n < DNB; ++n) {
        const float m = __builtin_fmaxf(amaxV[n], __shfl_xor(amaxV[n], 32));
        if (h == 0) maxV[(buf * DP + 32 * n + j) * DNW + w] = m;
      }
    };

    if (kV == 1) epilogue(doneV, redV);

    if (kV == 0) {
      // ---- V0: reverse activation of layer 0 of the finished tile, layer 0 of the next one -------------------
      const int64_t tile_new = tile_of(sV, c);
      if (sV == 0) last = tile_new >= n_tiles;
      const float iw = 1.0f / hdr[0];
      float inv[DNB], gx[DNB], gy[DNB], gz[DNB];
#pragma unroll
      for (int n = 0; n < DNB; ++n) { inv[n] = iw / bscaleV[n]; gx[n] = gy[n] = gz[n] = 0.f; }
#pragma unroll
      for (int n = 0; n < DNB; ++n) {
        const int64_t slot = tile_new * DP + 32 * n + j;
        px[n] = py[n] = pz[n] = 0.f;
        if (slot < count) {
          const int64_t idx = a.idx_in ? (int64_t)a.idx_in[slot] : slot;
          px[n] = a.pts[idx * 3]; py[n] = a.pts[idx * 3 + 1]; pz[n] = a.pts[idx * 3 + 2];
        }
      }
      f32x4 wv[8], sv[DNB][2];
      X3dSc T;
      float zz[DNB][8];
      auto ld_block = [&](int sl) {          // W0 rows of K-step sl of this wave, w cos(w z) of layer 0 of its groups
#pragma unroll
        for (int e = 0; e < 8; ++e) wv[e] = W0u[sl * 16 + h8 + e];
#pragma unroll
        for (int n = 0; n < DNB; ++n) {
          if constexpr (DBG & 8) { sv[n][0] = sv[n][1] = (f32x4){1.f, 1.f, 1.f, (float)sl}; continue; }
          sv[n][0] = stashV[((sl * DNB + n) * 2 + 0) * 64];
          sv[n][1] = stashV[((sl * DNB + n) * 2 + 1) * 64];
        }
      };
      ld_block(0);
      auto piece = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int sl = k >> 3, ph = k & 7;
        if constexpr (ph == 0 || ph == 1) {            // gradient of the finished tile, point tile n = ph
          constexpr int n = ph;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float av = (accV[n][8 * sl + e] * inv[n]) * sv[n][e >> 2][e & 3];
            gx[n] += wv[e].x * av;
            gy[n] += wv[e].y * av;
            gz[n] += wv[e].z * av;
          }
        }
        if constexpr (ph == 2) {                       // layer 0 of the next tile: both point tiles, then the rows are dead
#pragma unroll
          for (int n = 0; n < DNB; ++n)
#pragma unroll
            for (int e = 0; e < 8; ++e) zz[n][e] = ((wv[e].x * px[n] + wv[e].y * py[n]) + wv[e].z * pz[n]) + wv[e].w;
        }
        if constexpr (ph == 3) { x3d_sc<0>(T, zz[0], w0, w0, big); x3d_sc<1>(T, zz[0], w0, w0, big); }
        if constexpr (ph == 4 || ph == 7) {
          constexpr int n = ph == 4 ? 0 : 1;
          constexpr int g = sl * DNB + n;
          if constexpr (ph == 4) x3d_sc<2>(T, zz[0], w0, w0, big);
          u32x4 p0, p1;
          x3d_split8(T.s, kActScale, p0, p1);
          ownV[(g * 2 + 0) * 64] = p0; ownV[(g * 2 + 1) * 64] = p1;
          if constexpr (!(DBG & 4)) {
          stashV[(g * 2 + 0) * 64] = (f32x4){T.c[0], T.c[1], T.c[2], T.c[3]};
          stashV[(g * 2 + 1) * 64] = (f32x4){T.c[4], T.c[5], T.c[6], T.c[7]};
          }
        }
        if constexpr (ph == 5) {
          x3d_sc<0>(T, zz[1], w0, w0, big); x3d_sc<1>(T, zz[1], w0, w0, big);
          if constexpr (sl + 1 < DSL) ld_block(sl + 1);
        }
        if constexpr (ph == 6) x3d_sc<2>(T, zz[1], w0, w0, big);
      };
      gemm(kg, kg_after, actlG, piece);
      // partial sums of the finished tile
#pragma unroll
      for (int n = 0; n < DNB; ++n) {
        const float f = fpartV[n] + __shfl_xor(fpartV[n], 32);
        const float x = gx[n] + __shfl_xor(gx[n], 32);
        const float y = gy[n] + __shfl_xor(gy[n], 32);
        const float z = gz[n] + __shfl_xor(gz[n], 32);
        if (h == 0) redV[w * DP + 32 * n + j] = (f32x4){f, x, y, z};
        fpartV[n] = 0.f;
      }
      doneV = tileV;
      tileV = tile_new;
      if (__builtin_expect(__any(!(big < 1.0e4f)), 0)) {
        // an argument beyond the range of the fast reduction (never seen with trained SIRENs): layer 0 again, with
        // the library fix-up of iso_sin_wcos8
#pragma unroll 1
        for (int g = 0; g < DNG; ++g) {
          const int sl = g / DNB, n = g % DNB;
          float z8[8], hv[8], cv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const f32x4 r = W0u[sl * 16 + h8 + e];
            const float qx = n ? px[1] : px[0], qy = n ? py[1] : py[0], qz = n ? pz[1] : pz[0];
            z8[e] = ((r.x * qx + r.y * qy) + r.z * qz) + r.w;
          }
          iso_sin_wcos8(w0, w0, z8, hv, cv);
          u32x4 p0, p1;
          split8_f16(hv, p0, p1);
          ownV[(g * 2 + 0) * 64] = p0; ownV[(g * 2 + 1) * 64] = p1;
          stashV[(g * 2 + 0) * 64] = (f32x4){cv[0], cv[1], cv[2], cv[3]};
          stashV[(g * 2 + 1) * 64] = (f32x4){cv[4], cv[5], cv[6], cv[7]};
        }
      }
    } else if (kV <= L) {
      // ---- forward activation of layer kV-1 (top layer: head dot product and adjoint seed) ---------------------
      const int l = kV - 1;
      const bool top = kV == L;
      const float w_in = wh / (kActScale * hdr[l]);
      const gf4_t st_l = stashV + (int64_t)(l + 1) * DNG * 128;
      float fsave[DNB];
#pragma unroll
      for (int n = 0; n < DNB; ++n) fsave[n] = fpartV[n];
      if (top) {
#pragma unroll
        for (int n = 0; n < DNB; ++n) { amaxV[n] = 0.f; bscaleV[n] = seed_scale; }
      }
      X3dSc T;
      f32x4 wl0, wl1;
      if (!top) {
        auto piece = [&](auto kc) {
          constexpr int k = decltype(kc)::value;
          constexpr int g = k >> 2, ph = k & 3, p = g / DNB, n = g % DNB;
          float zz[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) zz[e] = accV[n][8 * p + e];
          if constexpr (ph == 0) x3d_sc<0>(T, zz, w_in, wh, big);
          if constexpr (ph == 1) x3d_sc<1>(T, zz, w_in, wh, big);
          if constexpr (ph == 2) {
            x3d_sc<2>(T, zz, w_in, wh, big);
            if constexpr (!(DBG & 4)) {
            st_l[(g * 2 + 0) * 64] = (f32x4){T.c[0], T.c[1], T.c[2], T.c[3]};
            st_l[(g * 2 + 1) * 64] = (f32x4){T.c[4], T.c[5], T.c[6], T.c[7]};
            }
          }
          if constexpr (ph == 3) {
            u32x4 p0, p1;
            x3d_split8(T.s, kActScale, p0, p1);
            ownV[(g * 2 + 0) * 64] = p0; ownV[(g * 2 + 1) * 64] = p1;
          }
        };
        gemm(kg, kg_after, actlG, piece);
      } else {
        auto piece = [&](auto kc) {
          constexpr int k = decltype(kc)::value;
          constexpr int g = k >> 2, ph = k & 3, p = g / DNB, n = g % DNB;
          float zz[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) zz[e] = accV[n][8 * p + e];
          if constexpr (ph == 0) {
            wl0 = *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(WLu + p * 16 + h8);
            wl1 = *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(WLu + p * 16 + h8 + 4);
            x3d_sc<0>(T, zz, w_in, wh, big);
          }
          if constexpr (ph == 1) x3d_sc<1>(T, zz, w_in, wh, big);
          if constexpr (ph == 2) {
            x3d_sc<2>(T, zz, w_in, wh, big);
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
            x3d_split8(hv, seed_scale, p0, p1);
            ownV[(g * 2 + 0) * 64] = p0; ownV[(g * 2 + 1) * 64] = p1;
          }
        };
        gemm(kg, kg_after, actlG, piece);
      }
      if (__builtin_expect(__any(!(big < 1.0e4f)), 0)) {
        // the stage again with the library fix-up for huge arguments (the accumulators are still intact)
#pragma unroll
        for (int n = 0; n < DNB; ++n) { fpartV[n] = fsave[n]; if (top) amaxV[n] = 0.f; }
#pragma unroll
        for (int g = 0; g < DNG; ++g) {
          const int p = g / DNB, n = g % DNB;
          float z8[8], hv[8], cv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) z8[e] = accV[n][8 * p + e];
          iso_sin_wcos8(w_in, wh, z8, hv, cv);
          u32x4 p0, p1;
          if (top) {
            const f32x4 v0 = *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(WLu + p * 16 + h8);
            const f32x4 v1 = *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(WLu + p * 16 + h8 + 4);
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
            st_l[(g * 2 + 0) * 64] = (f32x4){cv[0], cv[1], cv[2], cv[3]};
            st_l[(g * 2 + 1) * 64] = (f32x4){cv[4], cv[5], cv[6], cv[7]};
            split8_f16(hv, p0, p1);
          }
          ownV[(g * 2 + 0) * 64] = p0; ownV[(g * 2 + 1) * 64] = p1;
        }
      }
      if (top) { put_amax(0); mbufV = 0; }
    } else {
      // ---- reverse activation of layer l = 2L - kV >= 1: adjoint * w cos(w z), per-point scale, split ------------
      const int l = 2 * L - kV;
      const gcf4_t st_l = (gcf4_t)(stashV + (int64_t)l * DNG * 128);
      const float iw = 1.0f / hdr[l];
      const float grow = hdr[8 + l] * wh * 1.01f;
      float inv[DNB], nscale[DNB];
#pragma unroll
      for (int n = 0; n < DNB; ++n) {
        float m = 0.f;
#pragma unroll
        for (int ww = 0; ww < DNW; ++ww) m = __builtin_fmaxf(m, maxV[(mbufV * DP + 32 * n + j) * DNW + ww]);
        inv[n] = iw / bscaleV[n];
        nscale[n] = x3_scale_for(m * grow);
        amaxV[n] = 0.f;
      }
      f32x4 sv[2][2];
      if constexpr (DBG & 8) { sv[0][0] = sv[0][1] = sv[1][0] = sv[1][1] = (f32x4){1.f, 1.f, 1.f, (float)l}; }
      else { sv[0][0] = st_l[0]; sv[0][1] = st_l[64]; }
      float av[8];
      auto piece = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int g = k >> 2, ph = k & 3, p = g / DNB, n = g % DNB;
        if constexpr (ph == 0 && g + 1 < DNG && !(DBG & 8)) {
          sv[(g + 1) & 1][0] = st_l[((g + 1) * 2 + 0) * 64];
          sv[(g + 1) & 1][1] = st_l[((g + 1) * 2 + 1) * 64];
        }
        if constexpr (ph == 2) {
#pragma unroll
          for (int e = 0; e < 8; ++e) av[e] = (accV[n][8 * p + e] * inv[n]) * sv[g & 1][e >> 2][e & 3];
          float m = amaxV[n];
#pragma unroll
          for (int e = 0; e < 8; e += 2) m = __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(av[e]), __builtin_fabsf(av[e + 1])));
          amaxV[n] = m;
        }
        if constexpr (ph == 3) {
          u32x4 p0, p1;
          x3d_split8(av, nscale[n], p0, p1);
          ownV[(g * 2 + 0) * 64] = p0; ownV[(g * 2 + 1) * 64] = p1;
        }
      };
      gemm(kg, kg_after, actlG, piece);
#pragma unroll
      for (int n = 0; n < DNB; ++n) bscaleV[n] = nscale[n];
      put_amax(mbufV ^ 1);
      mbufV ^= 1;
    }
    __syncthreads();
    if (last && kV == 1 && sV == 1) break;
    // ---- the sets change roles: the fresh accumulators are the next vector stage's input -------------------
#pragma unroll
    for (int n = 0; n < DNB; ++n) {
      accV[n] = accG[n];
      float t;
      t = fpartV[n]; fpartV[n] = fpartG[n]; fpartG[n] = t;
      t = bscaleV[n]; bscaleV[n] = bscaleG[n]; bscaleG[n] = t;
      t = amaxV[n]; amaxV[n] = amaxG[n]; amaxG[n] = t;
    }
    { const int t = mbufV; mbufV = mbufG; mbufG = t; }
    { const int64_t t = tileV; tileV = tileG; tileG = t; }
    { const int64_t t = doneV; doneV = doneG; doneG = t; }
    if (sV == 1) { if (++kV == 2 * L) { kV = 0; ++c; } }
    sV ^= 1;
  }
}

}  // namespace

int64_t siren_x3d_stash_floats(int L) { return 256 * x3d_stash_per_wg(L); }

int siren_x3d_launch(const SirenArgs& a, int64_t n_upper, hipStream_t s) {
  static int dbg = -1;
  if (dbg < 0) {
    const char* e = getenv("ISO_X3D_DBG");
    dbg = e ? atoi(e) : 0;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_siren_step_x3d<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDLds);
#ifdef X3D_EXPERIMENTS
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_siren_step_x3d<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDLds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_siren_step_x3d<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDLds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_siren_step_x3d<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDLds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_siren_step_x3d<12>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDLds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_siren_step_x3d<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDLds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_siren_step_x3d<17>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDLds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_siren_step_x3d<28>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDLds);
#endif
  }
  const int64_t tiles = (n_upper + DP - 1) / DP;
  int blocks = (int)(tiles < 256 ? (tiles < 1 ? 1 : tiles) : 256);
#ifdef X3D_EXPERIMENTS
  { const char* e = getenv("ISO_X3D_BLOCKS"); if (e && atoi(e) > 0 && atoi(e) < blocks) blocks = atoi(e); }
  if (dbg == 1) { hipLaunchKernelGGL(k_siren_step_x3d<1>, dim3(blocks), dim3(64 * DNW), kDLds, s, a); return 0; }
  if (dbg == 4) { hipLaunchKernelGGL(k_siren_step_x3d<4>, dim3(blocks), dim3(64 * DNW), kDLds, s, a); return 0; }
  if (dbg == 8) { hipLaunchKernelGGL(k_siren_step_x3d<8>, dim3(blocks), dim3(64 * DNW), kDLds, s, a); return 0; }
  if (dbg == 16) { hipLaunchKernelGGL(k_siren_step_x3d<16>, dim3(blocks), dim3(64 * DNW), kDLds, s, a); return 0; }
  if (dbg == 17) { hipLaunchKernelGGL(k_siren_step_x3d<17>, dim3(blocks), dim3(64 * DNW), kDLds, s, a); return 0; }
  if (dbg == 28) { hipLaunchKernelGGL(k_siren_step_x3d<28>, dim3(blocks), dim3(64 * DNW), kDLds, s, a); return 0; }
  if (dbg == 12) { hipLaunchKernelGGL(k_siren_step_x3d<12>, dim3(blocks), dim3(64 * DNW), kDLds, s, a); return 0; }
#endif
  hipLaunchKernelGGL(k_siren_step_x3d<0>, dim3(blocks), dim3(64 * DNW), kDLds, s, a);
  return 0;
}
