// Fused SIREN SDF + gradient evaluation and Newton step with the hidden-layer GEMMs on the
// 16-bit matrix cores at f32 accuracy (v_mfma_f32_32x32x16_f16 / _bf16, gfx950).
//
// Reference semantics are those of siren.hip (Siren.forward DSS/models/common.py:140-165 under
// autograd.grad in UniformProjection._compute_sdf_and_grad, levelset_sampling.py:142-170, iterated
// by _project_points :313-342); this file only changes how the H x H products are formed.
//
// f32 product from 16-bit MFMAs (mfma_split.h): every f32
// operand is cut into TWO fp16 numbers (11 + 11 significant bits, round-to-nearest at each cut:
// <= 2^-22 relative in the worst case) under an exact power-of-two scale -- per layer for the weights, 2^12 for
// the activations, per point for the adjoint of the reverse sweep -- and W.x is accumulated in f32
// from three products  Wl.xh + Wh.xl + Wh.xh  (fp16 x fp16 is exact in f32; the dropped Wl.xl is
// <= 2^-22 relative; see mfma_split.h for what that means against the 2^-24 of f32).  The fp16 pipe runs 16x the f32 MFMA rate, so three passes are 5.3x faster than
// one f32 pass; measured against float64 in tests/test_projection_gpu.py next to the f32-MFMA
// kernel.  (The file started with an exact three-way bf16 cut, six products -- hence the x3 in the names; that
// form and the overlap experiments built on it are in tools/experiments/ and in the history.)
//
// Work decomposition (differs from siren.hip: weights are NOT staged through LDS)
//   * one workgroup = P = 32*NB points (NB = 3), NW = 8 waves (two per SIMD; NW = 4 also builds).
//   * the OUTPUT features of a layer are split across the waves (wave w owns tiles
//     TW*w .. TW*w+TW-1 of 32 features): each weight element is needed by exactly one wave,
//     which streams it from L2 straight into registers (lane-linear pre-split image, one 16-B
//     load per lane per (K-step, tile, part)), three K-steps ahead through four rotating register
//     sets that run on across layers and tiles; with 8 waves the operand requests are pinned one
//     behind each of the first MFMAs of a K-step (gemm_x3).
//   * the activations of all P points live in LDS, already split (2 x 8 fp16 per lane entry,
//     [K-step][point tile][part][lane]: every B operand is one conflict-free ds_read_b128).
//     Each wave reads all of them, and after the GEMM writes the K-steps made of its own output
//     features.  Two workgroup barriers per layer (readers done / writers done) replace the
//     per-chunk staging barriers of the f32 kernel.
//   * sin/cos: software reduction to [-1/2, 1/2] revolutions on packed f32 ops, then v_sin_f32 /
//     v_cos_f32 (mlp_common.h); a lane walks its accumulators with static indices, eight values at a
//     time (shapes with more than 8 groups per lane park them in LDS first).
//   * reverse sweep: same GEMM on the transposed image; w*cos(w z) comes back from the
//     per-lane global stash (written in the forward sweep by the same lane).
#include <stdlib.h>
#include <type_traits>
#include "siren_common.h"
#include "iso_newton.h"
#include "mlp_common.h"

#include "mfma_split.h"

#ifdef ISO_WITH_SIREN_PS      // tools/experiments/siren_ps: the point-stationary step, not part of the product library
static bool siren_ps_enabled();
#else
static inline bool siren_ps_enabled() { return false; }
#endif

namespace {

// s = sin(w_in * z), c = w * cos(w_in * z); w_in = w / (accumulator scale), an exact power-of-two quotient
__device__ __forceinline__ void x3_sin_wcos8(float w_in, float w, const float (&z)[8], float (&s)[8], float (&c)[8]) {
#ifdef X3_DBG_NOSINCOS
#pragma unroll
  for (int e = 0; e < 8; ++e) { s[e] = w_in * z[e]; c[e] = w; }
#else
  iso_sin_wcos8(w_in, w, z, s, c);
#endif
}


// ---- packing -------------------------------------------------------------------------------
// raw layout: W0[H*3] b0[H] {Wi[H*H] bi[H]}*L WL[H] bL[1]
// the K-order vectors: W0k, WLk, the biases of the hidden layers
__global__ void k_siren_pack_x3(const float* __restrict__ raw, float* __restrict__ packed, int H, int L) {
  const int64_t base = x3_base(H, L), total = x16_base(H, L);
  const int64_t HH = (int64_t)H * H;
  const float* b0 = raw + (int64_t)H * 3;
  const float* WL = raw + (int64_t)H * 4 + (int64_t)L * (HH + H);
  for (int64_t o = base + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total;
       o += (int64_t)gridDim.x * blockDim.x) {
    const int64_t rel = o - base;
    auto feat_of = [](int64_t ko) { return x3_feat((int)(ko >> 4), 8 * (int)((ko >> 3) & 1) + (int)(ko & 7)); };
    if (rel < 4 * (int64_t)H) {
      const int f = feat_of(rel >> 2), c = (int)(rel & 3);
      packed[o] = (c < 3) ? raw[f * 3 + c] : b0[f];
    } else if (rel < 5 * (int64_t)H) {
      packed[o] = WL[feat_of(rel - 4 * (int64_t)H)];
    } else {
      const int64_t r2 = rel - 5 * (int64_t)H;          // bias of hidden layer l, K-order
      const int l = (int)(r2 / H);
      const float* Wl = raw + (int64_t)H * 4 + (int64_t)l * (HH + H);
      packed[o] = Wl[HH + feat_of(r2 % H)];
    }
  }
}

// power-of-two scale of every hidden layer: max |2^s W| in [512, 1024)
__global__ void k_siren_wscale(const float* __restrict__ raw, float* __restrict__ packed, int H, int L) {
  __shared__ float s_m[256];
  const int l = blockIdx.x;
  const int64_t HH = (int64_t)H * H;
  const float* Wl = raw + (int64_t)H * 4 + (int64_t)l * (HH + H);
  // ONE pass over the layer (column f of thread f, rows in order: coalesced across the workgroup, 32 loads in flight --
  // the loop is a latency chain otherwise): the maximum for the scale and the column sums for c_l below.  (Two passes
  // with 16 loads in flight were 17 us per cycle; the weights change every step, so this runs every cycle.)
  float m = 0.f, cs = 0.f;
  for (int f = threadIdx.x; f < H; f += 256) {
    float t = 0.f;
#pragma unroll 32
    for (int k = 0; k < H; ++k) {
      const float a = fabsf(Wl[(int64_t)k * H + f]);
      m = (a == a && a > m) ? a : m;
      t += a;
    }
    cs = fmaxf(cs, t);
  }
  s_m[threadIdx.x] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) s_m[threadIdx.x] = fmaxf(s_m[threadIdx.x], s_m[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float mx = s_m[0];
    int e = 0;
    float sc = 1.0f;
    if (mx > 0.f && mx < 3.0e38f) {
      (void)frexpf(mx, &e);                    // mx = f * 2^e, f in [0.5, 1)
      int k = 10 - e;                          // 2^k * mx in [512, 1024)
      k = k > 100 ? 100 : (k < -100 ? -100 : k);
      sc = ldexpf(1.0f, k);
    }
    packed[x16_base(H, L) + l] = sc;
  }
  __syncthreads();
  // c_l = max_f sum_k |W_l[k][f]|: |(W_l^T a)[f]| <= c_l max|a|
  s_m[threadIdx.x] = cs;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) s_m[threadIdx.x] = fmaxf(s_m[threadIdx.x], s_m[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) packed[x16_base(H, L) + 8 + l] = s_m[0];
  if (l == 0) {
    __syncthreads();
    const float* WLh = raw + (int64_t)H * 4 + (int64_t)L * (HH + H);
    float mh = 0.f;
    for (int f = threadIdx.x; f < H; f += 256) mh = fmaxf(mh, fabsf(WLh[f]));
    s_m[threadIdx.x] = mh;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) s_m[threadIdx.x] = fmaxf(s_m[threadIdx.x], s_m[threadIdx.x + o]);
      __syncthreads();
    }
    if (threadIdx.x == 0) packed[x16_base(H, L) + 16] = s_m[0];
  }
}

// Bounds of the sine arguments for k_siren_step_ps (siren_ps_takes, siren_common.h; header slots 17..23); one workgroup per
// hidden layer.  Only launched when that kernel is enabled.
__global__ void k_siren_ps_bounds(const float* __restrict__ raw, float* __restrict__ packed, int H, int L) {
  __shared__ float s_m[256];
  const int l = blockIdx.x;
  const int64_t HH = (int64_t)H * H;
  const float* Wl = raw + (int64_t)H * 4 + (int64_t)l * (HH + H);
  // r_l = max_f (sum_k |W_l[f][k]| + |b_l[f]|): |W_l h + b_l| <= r_l for |h| <= 1 -- with it a kernel knows that no argument
  // of a hidden sine can reach the large-argument path (siren_ps_takes, siren_common.h); layers 0..4 have a slot
  if (l < 5) {
    float rs = 0.f;
    for (int f = threadIdx.x; f < H; f += 256) {
      float t = fabsf(Wl[HH + f]);
#pragma unroll 16
      for (int k = 0; k < H; ++k) t += fabsf(Wl[(int64_t)f * H + k]);
      rs = (t == t && t > rs) ? t : (t == t ? rs : 3.0e38f);
    }
    s_m[threadIdx.x] = rs;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) s_m[threadIdx.x] = fmaxf(s_m[threadIdx.x], s_m[threadIdx.x + o]);
      __syncthreads();
    }
    if (threadIdx.x == 0) packed[x16_base(H, L) + 17 + l] = s_m[0];
  }
  if (l == 0) {
    // layer 0: |W_0 x + b_0| <= (max_f sum_c |W_0[f][c]|) max|x_c| + max_f |b_0[f]|
    __syncthreads();
    float a0 = 0.f, b0m = 0.f;
    for (int f = threadIdx.x; f < H; f += 256) {
      const float t = (fabsf(raw[f * 3]) + fabsf(raw[f * 3 + 1])) + fabsf(raw[f * 3 + 2]);
      const float b = fabsf(raw[(int64_t)H * 3 + f]);
      a0 = (t == t) ? fmaxf(a0, t) : 3.0e38f;
      b0m = (b == b) ? fmaxf(b0m, b) : 3.0e38f;
    }
    s_m[threadIdx.x] = a0;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) s_m[threadIdx.x] = fmaxf(s_m[threadIdx.x], s_m[threadIdx.x + o]);
      __syncthreads();
    }
    if (threadIdx.x == 0) packed[x16_base(H, L) + 22] = s_m[0];
    __syncthreads();
    s_m[threadIdx.x] = b0m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) s_m[threadIdx.x] = fmaxf(s_m[threadIdx.x], s_m[threadIdx.x + o]);
      __syncthreads();
    }
    if (threadIdx.x == 0) packed[x16_base(H, L) + 23] = s_m[0];
  }
}

__global__ void k_siren_pack_f16(const float* __restrict__ raw, float* __restrict__ packed, int H, int L) {
  const int64_t HH = (int64_t)H * H;
  const int NTO = H / 32;
  const int64_t total = 2 * (int64_t)L * HH;       // forward images, then transposed ones
  for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
    const bool bwd = o >= (int64_t)L * HH;
    const int l = (int)((o / HH) % L);
    const int64_t q = o % HH;
    const float* Wl = raw + (int64_t)H * 4 + (int64_t)l * (HH + H);
    const float sc = packed[x16_base(H, L) + l];
    const int d = (int)(q & 3), lane = (int)((q >> 2) & 63);
    const int64_t blk = q >> 8;                 // (s*NTO + To)*2 + part
    const int part = (int)(blk & 1);
    const int To = (int)((blk >> 1) % NTO), s = (int)(blk / (2 * NTO));
    const int fo = 32 * To + (lane & 31);
    f32x2 v;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int fi = x3_feat(s, 8 * (lane >> 5) + 2 * d + u);
      v[u] = (bwd ? Wl[(int64_t)fi * H + fo] : Wl[(int64_t)fo * H + fi]) * sc;
    }
    const f16x2 h = __builtin_convertvector(v, f16x2);
    const f16x2 lo = __builtin_convertvector(v - __builtin_convertvector(h, f32x2), f16x2);
    reinterpret_cast<unsigned*>(packed + x16_off_layer(H, L, 0))[o] =
        part == 0 ? __builtin_bit_cast(unsigned, h) : __builtin_bit_cast(unsigned, lo);
  }
}

// ---- the step kernel -------------------------------------------------------------------------
#ifndef X3_LDS_STASH
#define X3_LDS_STASH 1
#endif
#ifndef X3_TILE_PIPE
#define X3_TILE_PIPE 1     // the tile boundary without its two load chains and its closing barrier (k_siren_step_x3)
#endif
#ifndef X3_LDS_SLOT
#define X3_LDS_SLOT 0      // which stash slot keeps its first LG groups in LDS (0: the longest-lived one)
#endif
template <int H, int NW, int NB>
struct X3Shape {
  static constexpr int NS = H / 16;        // K-steps of a hidden layer
  static constexpr int NTO = H / 32;       // output tiles
  static constexpr int TW = NTO / NW;      // output tiles per wave
  static constexpr int SL = 2 * TW;        // K-steps of the next layer produced by one wave
  static constexpr int NG = SL * NB;       // 8-value groups per lane
  static constexpr int P = 32 * NB;        // points per workgroup
  static constexpr size_t kActBytes = (size_t)NS * NB * kAP * 1024;
  static constexpr size_t kRedBytes = (size_t)NW * P * 16;
  static constexpr size_t kPtsBytes = (size_t)P * 16;                 // the tile's points, for the reverse sweep's layer 0
  // LDS-resident part of the derivative stash: the first LG of the NG groups per lane of stash slot 0 (written by
  // hidden layer 0, read back last of all by reverse stage 1 -- the longest-lived slot) stay in the CU; 2 KiB per
  // group and wave, as many groups as the 160 KiB of a gfx950 CU leave room for
  static constexpr int kLdsGroupsFit = (int)((163840 - (kActBytes + kRedBytes + kPtsBytes)) / ((size_t)NW * 2048));
  static constexpr int LG = X3_LDS_STASH ? (kLdsGroupsFit < NG ? kLdsGroupsFit : NG) : 0;
  static constexpr size_t kLds = kActBytes + kRedBytes + kPtsBytes + (size_t)NW * LG * 2048;
  // slot l of the global stash = w cos(w z) of hidden layer l's output (layer 0's is recomputed, the top layer's is
  // consumed on the spot): L - 1 slots in use, one kept for L = 1
  static constexpr int64_t kStashPerWg(int L) { return (int64_t)NW * (L > 1 ? L : 1) * NG * 512; }  // floats
  static_assert(NTO % NW == 0 && TW >= 1, "features must split evenly over the waves");
};

// -DX3_DBG_TIMES (timing experiment): waves of workgroup 0 stamp the shader clock at every stage
// boundary of their second tile into the tail of the stash workspace (tools/siren_stage_times.py).
#ifdef X3_DBG_TIMES
#define X3_STAMP()                                                                         \
  do {                                                                                     \
    if (dbg_on && lane == 0) dbg[w * 128 + (dbg_i)] = (long long)__builtin_amdgcn_s_memtime(); \
    ++dbg_i;                                                                               \
  } while (0)
#else
#define X3_STAMP() do {} while (0)
#endif

// FWD: value only (iso_siren_sdf, sphere tracing) -- no stash, no reverse sweep, no w cos(w z)
// bid / nblk: this workgroup's index among the nblk workgroups that share the list (the kernels below)
// sid: index of the workgroup's stash region (= bid unless the caller runs private lists, see k_siren_tail_x3)
template <int H, int NW, int NB, bool FWD>
__device__ __forceinline__ void x3_step_body(const SirenArgs& a, const int bid, const int nblk, const int sid) {
  using S = X3Shape<H, NW, NB>;
  constexpr int NS = S::NS, NTO = S::NTO, TW = S::TW, SL = S::SL, NG = S::NG, P = S::P;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* act = reinterpret_cast<u32x4*>(smem_raw);
  f32x4* red = reinterpret_cast<f32x4*>(smem_raw + S::kActBytes);   // [NW][P] {f,gx,gy,gz}
  f32x4* ptl = reinterpret_cast<f32x4*>(smem_raw + S::kActBytes + S::kRedBytes);                  // [P] {x,y,z,-}
  constexpr int LG = S::LG;
  const int tid = threadIdx.x;
  // the wave index is wave-uniform: say so, and everything derived from it (weight-image and
  // stash bases) lives in SGPRs; loads then use the scalar-base + lane-offset form
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, h = lane >> 5, j = lane & 31;
  u32x4* own = act + (size_t)(SL * w) * NB * kAP * 64 + lane;    // this wave's K-steps (+lane)
  u32x4* park = own + (size_t)(kAP - 2) * NG * 64;               // last NG*2 KiB of the region
  // Operand loads pinned behind the MFMAs (gemm_x3): used for the 8-wave shape only.  The 4-wave,
  // two-workgroups-per-CU build of H = 128 loses run-to-run repeatability as soon as the loads may
  // be scheduled between the MFMAs (tools/siren_stress.py: every repeat differs; the same source
  // with one workgroup per CU, and the 8-wave H = 256 build over 60 chaotic-weight repeats, are
  // bit-stable).  Cause not found yet, so that shape keeps the loads in front of the MFMA block.
#ifdef X3_FORCE_IL
  constexpr bool IL = true;
#else
  constexpr bool IL = (NW == 8);
#endif
  const int L = a.L;
  const float* X = a.packed + x3_base(H, L);
  // wave-uniform bases (SGPRs) + small per-lane offsets: no 64-bit per-lane pointers are kept live
  const f32x4* W0u = reinterpret_cast<const f32x4*>(X) + SL * w * 16;       // + h*8 + ...
  const float* WLu = X + 4 * H + SL * w * 16;                               // + h*8 + ...
  const int h8 = h * 8;
  const float bL = a.packed[off_bl(H)];
  f32x4* stash = reinterpret_cast<f32x4*>(a.stash) +
                 ((int64_t)sid * NW + w) * (int64_t)(L > 1 ? L : 1) * NG * 128;   // + lane
  f32x4* lst = reinterpret_cast<f32x4*>(smem_raw + S::kActBytes + S::kRedBytes + S::kPtsBytes) + w * (LG * 128) + lane;

  // weight images of this wave: forward / transposed image of hidden layer l (two fp16 parts)
  constexpr int FP = 2, BP = 2;
  auto fwd_img = [&](int l) { return reinterpret_cast<const u32x4*>(a.packed + x16_off_layer(H, L, l)) + (TW * w * 2) * 64; };
  auto rev_img = [&](int l) { return reinterpret_cast<const u32x4*>(a.packed + x16_off_bw(H, L, l)) + (TW * w * 2) * 64; };
  // accumulator scale of forward layer l: 2^12 (activations) * 2^s_l (weights)
  auto fwd_scale = [&](int l) { return kActScale * a.packed[x16_base(H, L) + l]; };
  // per-point maxima of the adjoint, exchanged between the waves: [2][P][NW] floats in the (otherwise
  // idle until the final reduction) `red` region
  float* redm = reinterpret_cast<float*>(red);
  float bscale[NB];                           // scale carried by the adjoint that sits in LDS, per point of this lane
  float amax[NB];                             // max |adjoint| over this lane's features, per point
  auto put_amax = [&](int buf) {              // before the barrier that ends the stage
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      const float m = __builtin_fmaxf(amax[n], __shfl_xor(amax[n], 32));
      if (h == 0) redm[(buf * P + 32 * n + j) * NW + w] = m;
    }
  };
#ifdef X3_STATIC_PRIO      // experiment (MI355X guide, "static priority for the younger half"): waves NW/2.. at a fixed higher priority
  if (w >= NW / 2) __builtin_amdgcn_s_setprio(X3_STATIC_PRIO);
#endif
  u32x4 A[4][TW][3];                     // weight-fragment pipeline, carried across stages
  x3_prefetch_a<TW, NTO, FP>(A, fwd_img(0), 0, lane);

  // (agent-scope load: in the Newton tail the count was written by this workgroup's own atomics a moment ago)
  const int64_t total = a.count_in ? (int64_t)__hip_atomic_load(a.count_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : a.n;
  if (total <= a.cnt_lo || total > a.cnt_hi) return;       // the other tile shape serves this list (uniform)
  if (a.ps_guard && H == 256 && siren_ps_takes(total, L, a.packed + x16_base(H, L), a.wh)) return;   // k_siren_step_ps has done it
  // this launch's share of the list: slots [slot0, count)  (SirenArgs::split)
  const int64_t cut = a.split ? siren_split_point(total) : total;
  const int64_t slot0 = a.split == 2 ? cut : 0;
  const int64_t count = a.split == 1 ? cut : total;
  const int64_t n_tiles = (count - slot0 + P - 1) / P;
#ifdef X3_DBG_TIMES
  long long* dbg = reinterpret_cast<long long*>(a.stash + (int64_t)nblk * S::kStashPerWg(L)) - NW * 128;
#endif
  // Tile boundary (X3_TILE_PIPE).  A tile used to end with two chains of dependent loads during which the whole CU
  // idled: the epilogue (list entry -> position -> returning atomic of the survivor list; 4.6 k cycles, one and a
  // half waves busy, the others parked at a closing barrier) and then the next tile's points (list entry -> position,
  // 2.3 k).  Now the next tile's list entries are requested during reverse stage 0 and its positions before the
  // barrier that precedes the epilogue; the epilogue takes the position from LDS (`ptl`) and its list entry from a
  // load issued before that barrier; and the closing barrier is gone -- waves 2..7 start layer 0 of the next tile
  // while waves 0-1 finish the epilogue (no LDS hazard: `red` is next written after three more barriers, `ptl` by
  // wave 0 itself, the activation regions were last read before the barrier that ends reverse stage 0's GEMM).
  __shared__ int s_next_tile, s_first_tile;
  int nidx[NB];
  float npx[NB], npy[NB], npz[NB];
  auto fetch_idx = [&](int64_t t) {                      // list entries of tile t (-1: beyond the list)
    int j_e = j;
    asm volatile("" : "+v"(j_e));
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      const int64_t slot = slot0 + t * P + 32 * n + j_e;
      nidx[n] = -1;
      if (t < n_tiles && slot < count) nidx[n] = a.idx_in ? a.idx_in[slot] : (int)slot;
    }
  };
  auto fetch_pts = [&]() {
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      npx[n] = npy[n] = npz[n] = 0.f;
      if (nidx[n] >= 0) {
        const int64_t idx = nidx[n];
        npx[n] = a.pts[idx * 3]; npy[n] = a.pts[idx * 3 + 1]; npz[n] = a.pts[idx * 3 + 2];
      }
    }
  };
  int64_t first_tile = bid;
  if (!FWD && X3_TILE_PIPE && a.tile_ctr != nullptr && a.draw_first) {       // (uniform) the first tile is drawn as well
    if (tid == 0) s_first_tile = atomicAdd(a.tile_ctr, 1);
    __syncthreads();
    first_tile = s_first_tile;
  }
  if constexpr (X3_TILE_PIPE) { fetch_idx(first_tile); fetch_pts(); }
  // Which tile comes next.  Static: bid + k nblk -- every CU takes the same number of tiles, and the launch ends with the
  // slowest XCD (under the power cap the eight XCDs of one MI355X ran this kernel 6.8 % apart, tools/diag/x3_end_times.py).
  // Dynamic (a.tile_ctr, reverse kernels): thread 0 draws the next tile from a counter (zero when the launch starts) while
  // stage 0's GEMM runs and hands it to the workgroup through LDS at the barrier that ends that GEMM.  A point's result
  // does not depend on the tile it sits in or on the workgroup that takes the tile.
  const bool dyn = !FWD && X3_TILE_PIPE && a.tile_ctr != nullptr;
  const int draw_base = a.draw_first ? 0 : nblk;
  int64_t next_tile = 0;
  for (int64_t tile = first_tile; tile < n_tiles; tile = next_tile) {
#ifdef X3_DBG_TIMES
    const bool dbg_on = bid == 0 && tile == (int64_t)nblk;
    int dbg_i = 0;
#endif
    X3_STAMP();
    float px[NB], py[NB], pz[NB];
    int j_e = j;
    asm volatile("" : "+v"(j_e));
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      if constexpr (X3_TILE_PIPE) {
        px[n] = npx[n]; py[n] = npy[n]; pz[n] = npz[n];
      } else {
        const int64_t slot = slot0 + tile * P + 32 * n + j_e;
        px[n] = py[n] = pz[n] = 0.f;
        if (slot < count) {
          const int64_t idx = a.idx_in ? (int64_t)a.idx_in[slot] : slot;
          px[n] = a.pts[idx * 3]; py[n] = a.pts[idx * 3 + 1]; pz[n] = a.pts[idx * 3 + 2];
        }
      }
      if constexpr (!FWD) {                // kept for the reverse sweep's layer 0 (visible after the barrier below)
        if (w == 0 && h == 0) ptl[32 * n + j_e] = (f32x4){px[n], py[n], pz[n], 0.f};
      }
    }
    X3_STAMP();
    // ---- layer 0 (3 -> H) on the VALU: this wave's H/NW features of all P points ------------
    for (int sl = 0; sl < SL; ++sl) {
      f32x4 wv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) wv[e] = W0u[sl * 16 + h8 + e];
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        float zz[8], hv[8], sv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) zz[e] = ((wv[e].x * px[n] + wv[e].y * py[n]) + wv[e].z * pz[n]) + wv[e].w;
        x3_sin_wcos8(a.w0, a.w0, zz, hv, sv);
        const int k = sl * NB + n;
        {
          u32x4 p0, p1;
          split8_f16(hv, p0, p1);
          own[(k * kAP + 0) * 64] = p0; own[(k * kAP + 1) * 64] = p1;
        }
        // w cos(w z0) is NOT stashed: z0 = W0 x + b0 is three FMAs to form again in the reverse sweep
        (void)sv;
      }
    }
    X3_STAMP();
    __syncthreads();
    X3_STAMP();

    f32x16 acc[TW][NB];
    float fpart[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) fpart[n] = 0.f;
    // ---- hidden layers, forward --------------------------------------------------------------
    for (int l = 0; l < L; ++l) {
      const float* lay = a.packed + x3_off_layer(H, L, l);
      const u32x4* img = fwd_img(l);
      const float zscale = fwd_scale(l);             // the accumulators hold zscale * (W h + b)
      const float w_in = a.wh / zscale;              // exact: zscale is a power of two
      if (l + 1 < L || FWD) {
        // the next stage is a forward one again (layer l+1, or layer 0 of the next tile)
        gemm_x3<TW, NB, NTO, NS, kBias, IL, FP, FP>(img, lay, act + lane, acc, w, 0, A,
                                                    l + 1 < L ? fwd_img(l + 1) : fwd_img(0), 0, lane, zscale);
      } else {
        gemm_x3<TW, NB, NTO, NS, kBias, IL, FP, BP>(img, lay, act + lane, acc, w, 0, A, rev_img(L - 1), 0, lane,
                                                    zscale);
      }
      X3_STAMP();
      __syncthreads();                      // both teams have read this team's K-half
      X3_STAMP();
      const bool top = (l == L - 1);
      f32x4* st_l = stash + (int64_t)l * NG * 128;
      const bool lds_slot = (l == X3_LDS_SLOT);        // the first LG groups of this slot stay in LDS
      // one 8-value group: sin / w cos, head or stash, split, store as the next layer's B entry
      // scale of the adjoint seed (uniform): |W_head[f] * w cos| <= max|W_head| * w
      const float seed_scale = x3_scale_for(a.packed[x16_base(H, L) + 16] * a.wh * 1.01f);
      if (top) {
#pragma unroll
        for (int n = 0; n < NB; ++n) { amax[n] = 0.f; bscale[n] = seed_scale; }
      }
      auto act_group = [&](int k, int n, int sl, const float (&zz)[8], float& fp) {
        float hv[8], sv[8];
        x3_sin_wcos8(w_in, a.wh, zz, hv, sv);
        if (top) {
          // adjoint seed of the top sine layer = head weight * w cos(w z); head dot product here
          const f32x4 wl0 = *reinterpret_cast<const f32x4*>(WLu + sl * 16 + h8);
          const f32x4 wl1 = *reinterpret_cast<const f32x4*>(WLu + sl * 16 + h8 + 4);
          const float f0 = (wl0.x * hv[0] + wl0.y * hv[1]) + (wl0.z * hv[2] + wl0.w * hv[3]);
          const float f1 = (wl1.x * hv[4] + wl1.y * hv[5]) + (wl1.z * hv[6] + wl1.w * hv[7]);
          fp += f0 + f1;
          if constexpr (FWD) return;
#pragma unroll
          for (int e = 0; e < 4; ++e) { hv[e] = wl0[e] * sv[e]; hv[4 + e] = wl1[e] * sv[4 + e]; }
          float m = amax[n];
#pragma unroll
          for (int e = 0; e < 8; e += 2) m = __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(hv[e]), __builtin_fabsf(hv[e + 1])));
          amax[n] = m;
          u32x4 p0, p1;
          split8_f16(hv, p0, p1, seed_scale);
          own[(k * kAP + 0) * 64] = p0; own[(k * kAP + 1) * 64] = p1;
          return;
        } else {
#ifndef X3_DBG_NOSTASH
          if constexpr (!FWD) {
            if (k < LG && lds_slot) {
              lst[(k * 2 + 0) * 64] = (f32x4){sv[0], sv[1], sv[2], sv[3]};
              lst[(k * 2 + 1) * 64] = (f32x4){sv[4], sv[5], sv[6], sv[7]};
            } else {
              st_l[(k * 2 + 0) * 64 + lane] = (f32x4){sv[0], sv[1], sv[2], sv[3]};
              st_l[(k * 2 + 1) * 64 + lane] = (f32x4){sv[4], sv[5], sv[6], sv[7]};
            }
          }
#endif
        }
        u32x4 p0, p1;                           // input of the next forward layer
        split8_f16(hv, p0, p1);
        own[(k * kAP + 0) * 64] = p0; own[(k * kAP + 1) * 64] = p1;
      };
#ifndef X3_DIRECT_ACT
#define X3_DIRECT_ACT 1
#endif
#ifndef X3_EARLY_STASH
#define X3_EARLY_STASH 0
#endif
#ifndef X3_DIRECT_NG
#define X3_DIRECT_NG 8
#endif
      if constexpr (NG <= X3_DIRECT_NG && X3_DIRECT_ACT) {
        // few enough values per lane to walk the accumulators with static indices
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
          for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int n = 0; n < NB; ++n) {
              float zz[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) zz[e] = acc[t][n][8 * p + e];
              act_group((2 * t + p) * NB + n, n, 2 * t + p, zz, fpart[n]);
              __builtin_amdgcn_sched_barrier(0);     // one group at a time: bounds register pressure
            }
      } else {
        // park the accumulators (f32) in the tail of this wave's own, now dead, LDS region and
        // walk them with a rolled loop; results overwrite the region front to back
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
          for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
              const int k = (2 * t + p) * NB + n;
              const f32x16& v = acc[t][n];
              park[(k * 2 + 0) * 64] = as_u32x4((f32x4){v[8 * p], v[8 * p + 1], v[8 * p + 2], v[8 * p + 3]});
              park[(k * 2 + 1) * 64] = as_u32x4((f32x4){v[8 * p + 4], v[8 * p + 5], v[8 * p + 6], v[8 * p + 7]});
            }
        for (int sl = 0; sl < SL; ++sl) {
          f32x4 z[NB][2];
#pragma unroll
          for (int n = 0; n < NB; ++n) {
            z[n][0] = as_f32x4(park[((sl * NB + n) * 2 + 0) * 64]);
            z[n][1] = as_f32x4(park[((sl * NB + n) * 2 + 1) * 64]);
          }
#pragma unroll
          for (int n = 0; n < NB; ++n) {
            float zz[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) zz[e] = z[n][e >> 2][e & 3];
            act_group(sl * NB + n, n, sl, zz, fpart[n]);
          }
        }
      }
      if constexpr (!FWD) { if (top) put_amax(0); }
      X3_STAMP();
      __syncthreads();                      // the next layer's inputs are complete
      X3_STAMP();
    }
    // ---- hidden layers, reverse --------------------------------------------------------------
    float gx[NB], gy[NB], gz[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) gx[n] = gy[n] = gz[n] = 0.f;
    int mbuf = 0;                             // exchange buffer that holds max |a_l| of the adjoint in LDS
    // max_k |a_l[k][p]| of this lane's points (written before the barrier that ended the last stage)
    auto get_amax = [&](float (&Mp)[NB]) {
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        float m = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) m = __builtin_fmaxf(m, redm[(mbuf * P + 32 * n + j) * NW + ww]);
        Mp[n] = m;
      }
    };
    // stages l = L-1 .. 1: a_{l-1} = (W_l^T a_l) * w cos(w z_{l-1}), the derivative from the stash
    for (int l = FWD ? 0 : L - 1; l >= 1; --l) {
      const u32x4* img = rev_img(l);
      const f32x4* st_l = stash + (int64_t)(l - 1) * NG * 128;      // reverse stage l reads slot l - 1
      f32x4 sv[NG][2];
      auto ld_stash = [&]() {
#pragma unroll
        for (int k = 0; k < NG; ++k) {
#ifdef X3_DBG_NOSTASH
          sv[k][0] = sv[k][1] = (f32x4){1.f, 1.f, 1.f, (float)l};
#else
          if (k < LG && l == X3_LDS_SLOT + 1) {
            sv[k][0] = lst[(k * 2) * 64]; sv[k][1] = lst[(k * 2 + 1) * 64];
          } else {
            sv[k][0] = st_l[(k * 2) * 64 + lane]; sv[k][1] = st_l[(k * 2 + 1) * 64 + lane];
          }
#endif
        }
      };
      float Mp[NB];
      get_amax(Mp);
      // w cos(w z) of the layer below: requested before the GEMM when the registers allow it
      // X3_EARLY_STASH 1: before the GEMM; 2: inside it, behind the stage's last own fragment request (gemm_x3's hook)
      if constexpr (NG <= 6 && X3_EARLY_STASH == 1) ld_stash();
      if constexpr (NG <= 6 && X3_EARLY_STASH == 2) {
        gemm_x3<TW, NB, NTO, NS, kZero, IL, BP, BP>(img, nullptr, act + lane, acc, w, 0, A, rev_img(l - 1), 0, lane, 1.0f,
                                                    nullptr, ld_stash);
      } else {
        gemm_x3<TW, NB, NTO, NS, kZero, IL, BP, BP>(img, nullptr, act + lane, acc, w, 0, A, rev_img(l - 1), 0, lane);
      }
      if constexpr (NG > 6 || !X3_EARLY_STASH) ld_stash();
      X3_STAMP();
      __syncthreads();
      X3_STAMP();
      // split-fp16 reverse: the accumulators hold 2^s_l * bscale[p] * (W_l^T a_l); the scale comes out
      // exactly, the next adjoint gets the scale its bound allows
      float inv[NB], nscale[NB];
      {
        const float iw = 1.0f / a.packed[x16_base(H, L) + l];
        const float grow = a.packed[x16_base(H, L) + 8 + l] * a.wh * 1.01f;
#pragma unroll
        for (int n = 0; n < NB; ++n) {
          inv[n] = iw / bscale[n];
          nscale[n] = x3_scale_for(Mp[n] * grow);
          amax[n] = 0.f;
        }
      }
#pragma unroll
      for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
          for (int n = 0; n < NB; ++n) {
            const int k = (2 * t + p) * NB + n;
            float av[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) av[e] = (acc[t][n][8 * p + e] * inv[n]) * sv[k][e >> 2][e & 3];
            float m = amax[n];
#pragma unroll
            for (int e = 0; e < 8; e += 2) m = __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(av[e]), __builtin_fabsf(av[e + 1])));
            amax[n] = m;
            u32x4 p0, p1;
            split8_f16(av, p0, p1, nscale[n]);
            own[(k * kAP + 0) * 64] = p0; own[(k * kAP + 1) * 64] = p1;
          }
        }
#pragma unroll
      for (int n = 0; n < NB; ++n) bscale[n] = nscale[n];
      put_amax(mbuf ^ 1);
      mbuf ^= 1;
      X3_STAMP();
      __syncthreads();
      X3_STAMP();
    }
    // stage 0: d sdf / d x = W_0^T [ (W_1^T a_1) * w0 cos(w0 z_0) ] -- z_0 = W_0 x + b_0 is formed again from the point
    // (three FMAs; the same expression on the same operands as in the forward sweep) instead of being stashed
    if constexpr (!FWD) {
      int drawn = 0;
      if (dyn && tid == 0) drawn = draw_base + atomicAdd(a.tile_ctr, 1);
      gemm_x3<TW, NB, NTO, NS, kZero, IL, BP, FP>(rev_img(0), nullptr, act + lane, acc, w, 0, A, fwd_img(0), 0, lane);
      if (dyn && tid == 0) s_next_tile = drawn;
      X3_STAMP();
      __syncthreads();
      X3_STAMP();
      next_tile = dyn ? (int64_t)s_next_tile : tile + nblk;
      if constexpr (X3_TILE_PIPE) fetch_idx(next_tile);
      float inv[NB];
      {
        const float iw = 1.0f / a.packed[x16_base(H, L)];
#pragma unroll
        for (int n = 0; n < NB; ++n) inv[n] = iw / bscale[n];
      }
      f32x4 q[NB];
#pragma unroll
      for (int n = 0; n < NB; ++n) q[n] = ptl[32 * n + j];
#pragma unroll
      for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          f32x4 wv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) wv[e] = W0u[(2 * t + p) * 16 + h8 + e];
#pragma unroll
          for (int n = 0; n < NB; ++n) {
            float zz[8], cv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) zz[e] = ((wv[e].x * q[n].x + wv[e].y * q[n].y) + wv[e].z * q[n].z) + wv[e].w;
#ifdef X3_DBG_NOSTASH
#pragma unroll
            for (int e = 0; e < 8; ++e) cv[e] = zz[e];
#else
            iso_wcos8(a.w0, a.w0, zz, cv);
#endif
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float av = (acc[t][n][8 * p + e] * inv[n]) * cv[e];
              gx[n] += wv[e].x * av;
              gy[n] += wv[e].y * av;
              gz[n] += wv[e].z * av;
            }
            __builtin_amdgcn_sched_barrier(0);     // one group at a time: bounds register pressure
          }
        }
      X3_STAMP();
      // (no barrier here with X3_TILE_PIPE: between the one that ends this stage's GEMM and the one after the
      // reduction nothing reads what the reduction writes)
      if constexpr (!X3_TILE_PIPE) __syncthreads();
      X3_STAMP();
    } else {
      next_tile = tile + nblk;
      if constexpr (X3_TILE_PIPE) fetch_idx(next_tile);
    }
    // ---- reduce head + gradient over the lane halves and the waves -----------------------------
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      float f = fpart[n] + __shfl_xor(fpart[n], 32);
      float x = gx[n] + __shfl_xor(gx[n], 32);
      float y = gy[n] + __shfl_xor(gy[n], 32);
      float z = gz[n] + __shfl_xor(gz[n], 32);
      if (h == 0) red[w * P + 32 * n + j] = (f32x4){f, x, y, z};
    }
    // (the thread / lane ids are made opaque here: everything derived from them -- LDS addresses,
    // the rank mask of the compaction -- is recomputed per tile instead of being kept live, and
    // spilled, across the whole tile)
    int tid_e = tid, lane_e = lane;
    asm volatile("" : "+v"(tid_e), "+v"(lane_e));
    int eidx = -1;                              // X3_TILE_PIPE: the epilogue's list entry and position, requested
    f32x4 qe = {0.f, 0.f, 0.f, 0.f};            // before the barrier (the position while `ptl` still holds this tile)
    if constexpr (X3_TILE_PIPE) {
      const int64_t slot = slot0 + tile * P + tid_e;
      if (tid_e < P && slot < count) {
        eidx = a.idx_in ? a.idx_in[slot] : (int)slot;
        if constexpr (!FWD) qe = ptl[tid_e];
      }
      fetch_pts();
    }
    X3_STAMP();
    __syncthreads();
    X3_STAMP();
    // ---- epilogue: thread tid handles point `tid` of the tile ----------------------------------
    bool survive = false;
    int64_t idx = -1;
    {
      const int64_t slot = slot0 + tile * P + tid_e;
      if (tid_e < P && slot < count) {
        f32x4 r = red[tid_e];
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) {
          const f32x4 q = red[ww * P + tid_e];
          r.x += q.x; r.y += q.y; r.z += q.z; r.w += q.w;
        }
        const float f = r.x + bL;
        if constexpr (X3_TILE_PIPE) {
          idx = eidx;
          if constexpr (!FWD) survive = iso_step_finish<SirenArgs, true>(a, idx, f, r.y, r.z, r.w, qe.x, qe.y, qe.z);
          else survive = iso_step_finish(a, idx, f, r.y, r.z, r.w);
        } else {
          idx = a.idx_in ? (int64_t)a.idx_in[slot] : slot;
          survive = iso_step_finish(a, idx, f, r.y, r.z, r.w);
        }
      }
    }
    if (!a.eval_only && a.do_move) {
      const unsigned long long bal = __ballot(survive);
      if (bal) {
        int base = 0;
        const int leader = __ffsll((long long)bal) - 1;
        if (lane_e == leader) base = atomicAdd(a.count_out, __popcll(bal));
        base = __shfl(base, leader);
        if (survive) {
          const int rank = __popcll(bal & ((1ull << lane_e) - 1ull));
          a.idx_out[base + rank] = (int32_t)idx;
        }
      }
    }
    X3_STAMP();
    if constexpr (!X3_TILE_PIPE) __syncthreads();
    X3_STAMP();
  }
}

// -DX3_DBG_END (timing experiment, tools/diag/x3_end_times.py): thread 0 of every workgroup of k_siren_step_x3 leaves the
// constant-rate clock at its start and end and the XCD it ran on
#ifdef X3_DBG_END
__device__ unsigned long long x3_end[2048 * 3];
extern "C" int iso_dbg_x3_end(unsigned long long* out, int n_blocks) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(x3_end), sizeof(unsigned long long) * 3 * n_blocks) == hipSuccess ? 0 : -1;
}
#endif
template <int H, int NW, int NB, int MINB, bool FWD>
__global__ __launch_bounds__(64 * NW, MINB) void k_siren_step_x3(SirenArgs a) {
#ifdef X3_DBG_END
  const unsigned long long t0 = wall_clock64();
#endif
  x3_step_body<H, NW, NB, FWD>(a, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.x);
#ifdef X3_DBG_END
  if (threadIdx.x == 0 && blockIdx.x < 2048) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    x3_end[blockIdx.x * 3] = t0; x3_end[blockIdx.x * 3 + 1] = wall_clock64(); x3_end[blockIdx.x * 3 + 2] = xcc;
  }
#endif
}

// Both tile shapes of a split list (SirenArgs::split) in ONE launch: workgroups [0, big_blocks) serve the slots below
// siren_split_point(count) on NB-tile workgroups, the others the rest on one-tile workgroups (their own stash region
// behind the first group's).  The dispatcher places workgroups in index order, one per CU, so the small tiles start
// on the CUs that finish their large ones first; a launch that finds nothing to do for one of the shapes costs nothing
// (issued separately, the idle one of the two launches took ~4.4 us, 15 times per headline cycle).
#ifndef X3_BIG_TAKES_SMALL
#define X3_BIG_TAKES_SMALL 1
#endif
template <int H, int NW, int NB, int MINB>
__global__ __launch_bounds__(64 * NW, MINB) void k_siren_step_x3_both(SirenArgs a) {
#ifdef X3_DBG_END
  const unsigned long long t0 = wall_clock64();
  struct EndStamp { unsigned long long t0; __device__ ~EndStamp() {
    if (threadIdx.x == 0 && blockIdx.x < 2048) {
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      x3_end[blockIdx.x * 3] = t0; x3_end[blockIdx.x * 3 + 1] = wall_clock64(); x3_end[blockIdx.x * 3 + 2] = xcc;
    } } } stamp{t0};
#endif
  const bool big = (int)blockIdx.x < a.big_blocks;
  const int small_blocks = (int)gridDim.x - a.big_blocks;
  if (big) {
    SirenArgs b = a;
    b.split = 1;
    x3_step_body<H, NW, NB, false>(b, (int)blockIdx.x, a.big_blocks, (int)blockIdx.x);
  }
  // Drawn tiles (a.tile_ctr): a workgroup of the large shape that finds no large tile left goes on with the small ones
  // (its own stash region, participant small_blocks + blockIdx of small_blocks + big_blocks) -- the large workgroups end
  // within one tile time (68 us) of each other, and the small tiles are what the early ones fill that time with
  // (headline cycle 13.33-13.38 -> 13.22-13.25 ms; with half a round to one and a half rounds of the large tiles' points
  // handed to the small ones on top: 13.26-13.28).
  if (!big || (X3_BIG_TAKES_SMALL && a.tile_ctr != nullptr)) {
    if (big) __syncthreads();              // (the last large tile's epilogue reads LDS the other shape lays out differently)
    SirenArgs b = a;
    b.split = 2;
    if (a.tile_ctr) b.tile_ctr = a.tile_ctr + 1;
    const bool joint = X3_BIG_TAKES_SMALL && a.tile_ctr != nullptr;
    b.draw_first = joint ? 1 : 0;          // (a workgroup that joins late must not own a tile nobody else may take)
    const int bid = big ? small_blocks + (int)blockIdx.x : (int)blockIdx.x - a.big_blocks;
    const int nblk = joint ? small_blocks + a.big_blocks : small_blocks;
    if (big) b.stash = a.stash + (int64_t)blockIdx.x * X3Shape<H, NW, NB>::kStashPerWg(a.L);
    else b.stash = a.stash + (int64_t)a.big_blocks * X3Shape<H, NW, NB>::kStashPerWg(a.L);
    x3_step_body<H, NW, 1, false>(b, bid, nblk, big ? 0 : bid);
  }
}

// The Newton tail in ONE launch (levelset_sampling.py:306-341: the reference leaves its loop when nothing is active; a
// launch per iteration issued all T + 1 of them, the late ones for a few hundred points or none).  Iteration it_first
// is taken from the list the previous launch left, dealt out tile by tile (32 points) as always; but a workgroup puts
// the survivors of ITS tiles on a private list and goes on with them alone -- evaluate, move, keep the survivors --
// until none is left or iteration it_last (the evaluation without a move) is done.  A point's result does not depend
// on the tile it sits in, so the results are those of the launch-per-iteration form, bit for bit.  Between two rounds
// one workgroup barrier (the epilogue's atomics and list entries of the slower waves must have landed).
template <int H, int NW, int MINB>
__global__ __launch_bounds__(64 * NW, MINB) void k_siren_tail_x3(SirenArgs a) {
  const int b = blockIdx.x, nblk = gridDim.x;
  int32_t* lists = a.tail_lists + (int64_t)b * 2 * a.tail_cap;
  int32_t* cnts = a.tail_counts + b * 2;
  SirenArgs r = a;
  r.split = 0; r.small_tiles = 1; r.cnt_lo = -1; r.cnt_hi = INT64_MAX; r.tile_ctr = nullptr;
  // round 0: this workgroup's tiles of the global list
  r.idx_out = lists; r.count_out = cnts;
  r.do_move = a.it_first < a.it_last ? 1 : 0;
  x3_step_body<H, NW, 1, false>(r, b, nblk, b);
  int cur = 0;
  for (int it = a.it_first + 1; it <= a.it_last; ++it) {
    __syncthreads();
    const int m = __hip_atomic_load(&cnts[cur], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (m == 0) break;                                     // (workgroup-uniform)
    if (threadIdx.x == 0) {
      __hip_atomic_store(&cnts[cur ^ 1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a.iter_counts) atomicAdd(&a.iter_counts[it], m);
    }
    __syncthreads();
    r.idx_in = lists + (int64_t)cur * a.tail_cap; r.count_in = cnts + cur;
    r.idx_out = lists + (int64_t)(cur ^ 1) * a.tail_cap; r.count_out = cnts + (cur ^ 1);
    r.do_move = it < a.it_last ? 1 : 0;
    x3_step_body<H, NW, 1, false>(r, 0, 1, b);
    cur ^= 1;
  }
  // leave the private counters at zero for the next launch on this workspace
  __syncthreads();
  if (threadIdx.x == 0) { cnts[0] = 0; cnts[1] = 0; }
}

template <int H, int NW, int NB, int MINB, bool FWD>
int launch_x3(const SirenArgs& a, int64_t n_upper, hipStream_t s) {
  using S = X3Shape<H, NW, NB>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_siren_step_x3<H, NW, NB, MINB, FWD>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::kLds);
    attr_done = true;
  }
  const int64_t tiles = (n_upper + S::P - 1) / S::P;
#ifdef X3_CAP_BLOCKS
  const int64_t cap = X3_CAP_BLOCKS;
#else
  const int64_t cap = 256 * MINB;
#endif
  const int blocks = (int)(tiles < cap ? (tiles < 1 ? 1 : tiles) : cap);
  hipLaunchKernelGGL((k_siren_step_x3<H, NW, NB, MINB, FWD>), dim3(blocks), dim3(64 * NW), S::kLds, s, a);
  return 0;
}

template <int H, int NW, int NB, int MINB>
int launch_x3_both(SirenArgs a, int64_t n_upper, hipStream_t s) {
  using SB = X3Shape<H, NW, NB>;
  using SS = X3Shape<H, NW, 1>;
  constexpr size_t lds = SB::kLds > SS::kLds ? SB::kLds : SS::kLds;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_siren_step_x3_both<H, NW, NB, MINB>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  const int64_t cap = 256 * MINB;
  const int64_t tb = (n_upper + SB::P - 1) / SB::P, ts = (n_upper + SS::P - 1) / SS::P;
  a.big_blocks = (int)(tb < cap ? (tb < 1 ? 1 : tb) : cap);
  const int small_blocks = (int)(ts < cap ? (ts < 1 ? 1 : ts) : cap);
  hipLaunchKernelGGL((k_siren_step_x3_both<H, NW, NB, MINB>), dim3(a.big_blocks + small_blocks), dim3(64 * NW), lds, s, a);
  return 0;
}

}  // namespace

#ifndef X3_NW
#define X3_NW 8
#endif
#ifndef X3_MINB256
#define X3_MINB256 1   // workgroups per CU at H = 256
#endif
#ifndef X3_NB256
#define X3_NB256 3   // point tiles of 32 per workgroup at H = 256 (4 would fit LDS with the two-part layout, but needs
                     // 64 + 64 + 32 accumulator / operand registers: 101 spilled VGPRs, 3.63 instead of 3.26 ms)
#endif
#ifndef X3_MINB128
#define X3_MINB128 1   // workgroups per CU for H = 128: 2 would fit (78 KiB LDS each) but the 256-VGPR budget then forces
                       // scratch spills, and that build is not repeatable from run to run (tools/siren_repeat_check.py)
#endif
bool siren_x3_supported(int H, int L) { return (H == 256 || H == 128) && L >= 1 && L <= 8; }

int64_t siren_x3_stash_floats(int H, int L) {
  if (H == 256) {     // both tile shapes of a split list run in one launch: a region each
    return 256 * X3_MINB256 * (X3Shape<256, X3_NW, X3_NB256>::kStashPerWg(L) + X3Shape<256, X3_NW, 1>::kStashPerWg(L));
  }
  if (H == 128) return 256 * X3_MINB128 * X3Shape<128, 4, 3>::kStashPerWg(L);
  return 0;
}

void siren_x3_pack(const float* raw, float* packed, int H, int L, hipStream_t s) {
  const int64_t words = x16_base(H, L) - x3_base(H, L);
  hipLaunchKernelGGL(k_siren_pack_x3, dim3(iso_stream_grid(words, 256)), dim3(256), 0, s, raw, packed, H, L);
  if (L > 0) {
    hipLaunchKernelGGL(k_siren_wscale, dim3(L), dim3(256), 0, s, raw, packed, H, L);
#ifdef ISO_WITH_SIREN_PS
    if (siren_ps_enabled() && siren_ps_supported(H, L)) hipLaunchKernelGGL(k_siren_ps_bounds, dim3(L), dim3(256), 0, s, raw, packed, H, L);
#endif
    hipLaunchKernelGGL(k_siren_pack_f16, dim3(iso_stream_grid(2 * (int64_t)L * H * H, 256)), dim3(256), 0, s, raw, packed, H, L);
  }
}

int siren_x3_tail_blocks() { return 256 * X3_MINB256; }

int siren_x3_launch_tail(const SirenArgs& a, int H, hipStream_t s) {
  if (H != 256) return -1;
  using SS = X3Shape<256, X3_NW, 1>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_siren_tail_x3<256, X3_NW, X3_MINB256>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)SS::kLds);
    attr_done = true;
  }
  hipLaunchKernelGGL((k_siren_tail_x3<256, X3_NW, X3_MINB256>), dim3(siren_x3_tail_blocks()), dim3(64 * X3_NW), SS::kLds, s, a);
  return 0;
}

// ISO_SIREN_PS=1: lists the point-stationary kernel takes (siren_ps_takes) are served by it; the launch of this file's
// kernels that follows carries ps_guard and returns at once for them
#ifdef ISO_WITH_SIREN_PS
static bool siren_ps_enabled() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("ISO_SIREN_PS"); v = (e && e[0] == '1') ? 1 : 0; }
  return v == 1;
}
#endif

int siren_x3_launch(const SirenArgs& a, int H, int64_t n_upper, hipStream_t s) {
#ifdef ISO_WITH_SIREN_PS
  if (siren_ps_enabled() && H == 256 && !a.fwd_only && !a.dirs && siren_ps_supported(H, a.L) && n_upper >= kPsMinList &&
      (a.split == 3 || (a.split == 0 && !a.small_tiles)) && a.cnt_lo < 0 && a.cnt_hi == INT64_MAX) {
    siren_ps_launch(a, n_upper, s);
    SirenArgs g = a;
    g.ps_guard = 1;
    if (g.split == 3) return launch_x3_both<256, X3_NW, X3_NB256, X3_MINB256>(g, n_upper, s);
    return launch_x3<256, X3_NW, X3_NB256, X3_MINB256, false>(g, n_upper, s);
  }
#endif
  if (a.split == 3 && H == 256 && !a.fwd_only) return launch_x3_both<256, X3_NW, X3_NB256, X3_MINB256>(a, n_upper, s);
  if (a.small_tiles && H == 256 && !a.fwd_only) return launch_x3<256, X3_NW, 1, X3_MINB256, false>(a, n_upper, s);
  if (a.fwd_only) {
    if (H == 256) return launch_x3<256, X3_NW, X3_NB256, X3_MINB256, true>(a, n_upper, s);
    if (H == 128) return launch_x3<128, 4, 3, X3_MINB128, true>(a, n_upper, s);
  }
  if (H == 256) return launch_x3<256, X3_NW, X3_NB256, X3_MINB256, false>(a, n_upper, s);
  if (H == 128) return launch_x3<128, 4, 3, X3_MINB128, false>(a, n_upper, s);
  return -1;
}
