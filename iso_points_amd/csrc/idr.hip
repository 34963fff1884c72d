// Fused IDR-style SDF + gradient evaluation and Newton step (f32 MFMA), gfx950.
//
// Reference network: SDF.forward, DSS/models/common.py:220-310 (based on IDR / SAL):
//   e(x)   = [x, sin(2^k x), cos(2^k x)]_{k<F}                       (get_embedder :205-217)
//   h_0    = softplus_b(W_0 e + b_0)
//   h_l    = softplus_b(W_l in_l + b_l),  in_l = [h_{l-1}, e]/sqrt(2) when l == skip  (:297-298)
//   sdf    = tanh(W_n h_{n-1} + b_n)                                  (:305)
// with weight-normalised linears (:277-278; the effective weights g*v/|v| are formed by the
// caller), beta = 100, torch's softplus threshold 20.  Layer skip-1 outputs H - D0 features so
// that the concatenation is H wide (:251-252).
//
// Same machinery as siren.hip (one wave = 16 points, activations per wave in LDS in the MFMA
// B-operand layout, weight images shared by the 4 waves through a double-buffered LDS stage,
// reverse-mode gradient with a stash of the activation derivatives, one launch per Newton
// iteration over a device-side active list).  Differences:
//   * layer 0 is an MFMA pass with nq = D0pad/16 chunks: its B operand is the positional
//     encoding, which every lane evaluates for the slots it owns;
//   * the narrow layer is zero-padded to H rows; after its activation the owned encoding slots
//     overwrite the padding and everything is divided by sqrt(2);
//   * backward of layer 0 (H -> D0) and the encoding Jacobian run on the VALU;
//   * tanh' is a per-point scalar: the adjoint seed is W_n * s_{n-1} and the final gradient is
//     multiplied by 1 - tanh^2.
#include <stdlib.h>
#include "iso_common.h"
#include "iso_newton.h"
#include "mlp_common.h"

#ifdef ISO_IDR_PLAIN_GEMM          // the shared, non-pipelined pass (A/B timing)
#define IDR_GEMM gemm_pass
#else
#define IDR_GEMM gemm_pass_pipe
#endif

#include "idr_common.h"

namespace {

__global__ void k_idr_pack(const float* __restrict__ raw, float* __restrict__ packed, IdrShape s) {
  const int H = s.H, NT = H / 16;
  const int64_t total = idr_total(H, s.n_layers);
  const int64_t HH = (int64_t)H * H;
  for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total;
       o += (int64_t)gridDim.x * blockDim.x) {
    float v = 0.f;
    const int od0 = idr_out_dim(s, 0);                           // H, or H - D0 when layer 0 is the narrow one
    if (o < idr_off_fw0(H)) {                                    // b0
      v = (o < od0) ? raw[idr_raw_off(s, 0) + (int64_t)od0 * s.D0 + o] : 0.f;
    } else if (o < idr_off_w0v(H)) {                             // FW0 [q][t][lane][i], in = padded encoding
      int64_t w = o - idr_off_fw0(H);
      int i = (int)(w & 3), lane = (int)((w >> 2) & 63);
      int t = (int)((w >> 8) % NT), q = (int)((w >> 8) / NT);
      int a = 16 * t + (lane & 15), b = 16 * q + 4 * (lane >> 4) + i;
      v = (b < s.D0 && a < od0) ? raw[idr_raw_off(s, 0) + (int64_t)a * s.D0 + b] : 0.f;
    } else if (o < idr_off_layer(H, 1)) {                        // W0v [g][k][e]: input columns for the VALU reverse
      int64_t w = o - idr_off_w0v(H);
      int e = (int)(w % (H / 4));
      int k = (int)((w / (H / 4)) % kW0Row), g = (int)((w / (H / 4)) / kW0Row);
      int f = 16 * (e >> 2) + 4 * g + (e & 3);
      v = (k < s.D0 && f < od0) ? raw[idr_raw_off(s, 0) + (int64_t)f * s.D0 + k] : 0.f;
    } else if (o < idr_off_wl(H, s.n_layers)) {
      int64_t k = o - idr_off_layer(H, 1);
      const int64_t per = H + 2 * HH;
      int l = 1 + (int)(k / per);
      int64_t w = k % per;
      const int od = idr_out_dim(s, l);
      const float* Wl = raw + idr_raw_off(s, l);
      const float* bl = Wl + (int64_t)od * H;
      if (w < H) {
        v = (w < od) ? bl[w] : 0.f;
      } else {
        w -= H;
        bool bwd = w >= HH;
        if (bwd) w -= HH;
        int i = (int)(w & 3), lane = (int)((w >> 2) & 63);
        int t = (int)((w >> 8) % NT), q = (int)((w >> 8) / NT);
        int a = 16 * t + (lane & 15), b = 16 * q + 4 * (lane >> 4) + i;
        if (bwd) v = (b < od) ? Wl[(int64_t)b * H + a] : 0.f;   // contracted index = out row
        else v = (a < od) ? Wl[(int64_t)a * H + b] : 0.f;       // tile row = out row
      }
    } else {
      int64_t k = o - idr_off_wl(H, s.n_layers);
      const float* Wn = raw + idr_raw_off(s, s.n_layers);
      if (k < H) {
        int e = (int)(k % (H / 4)), g = (int)(k / (H / 4));
        v = Wn[16 * (e >> 2) + 4 * g + (e & 3)];
      } else {
        v = (k == H) ? Wn[H] : 0.f;
      }
    }
    packed[o] = v;
  }
}


template <int NT>
__global__ __launch_bounds__(256, 1) void k_idr_step(IdrArgs a) {
  constexpr int H = NT * 16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, j = lane & 15;
  float* hL = smem + wave * (NT * 256);
  float* wbuf = smem + 4 * NT * 256;
  const IdrShape s = a.s;
  const int nL = s.n_layers;
  const float inv_sqrt2_den = 1.41421356237309515f;     // x / np.sqrt(2) as float32
  const float* WLimg = a.packed + idr_off_wl(H, nL);
  const float b_last = a.packed[idr_off_wl(H, nL) + H];
  float* stash = a.stash + ((int64_t)blockIdx.x * 4 + wave) * (int64_t)nL * NT * 256;
  const int first_enc_slot = H - s.D0;                  // concat: slots [H-D0, H) hold e(x)

  const int64_t count = a.count_in ? (int64_t)(*a.count_in) : a.n;
  const int64_t n_tiles = (count + 63) / 64;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t slot = tile * 64 + wave * 16 + j;
    const bool valid = slot < count;
    int64_t idx = -1;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (valid) {
      idx = a.idx_in ? (int64_t)a.idx_in[slot] : slot;
      px = a.pts[idx * 3]; py = a.pts[idx * 3 + 1]; pz = a.pts[idx * 3 + 2];
    }
    // ---- encoding -> B operand of layer 0 (slots 16q+4g+i, q < kD0Pad/16)
    for (int q = 0; q < kD0Pad / 16; ++q) {
      f32x4 e4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = 16 * q + 4 * g + i;
        float v = 0.f, dv; int c;
        if (f < s.D0) posenc(f, px, py, pz, v, c, dv);
        e4[i] = v;
      }
      reinterpret_cast<f32x4*>(hL)[q * 64 + lane] = e4;
    }
    f32x4 acc[NT];
    float fsum = 0.f;
    // ---- forward
    for (int l = 0; l < nL; ++l) {
      if (l == 0) {
        IDR_GEMM<NT, true>(a.packed + idr_off_fw0(H), a.packed + idr_off_b0(), hL, wbuf, acc, lane, g,
                            kD0Pad / 16);
      } else {
        const float* base = a.packed + idr_off_layer(H, l);
        IDR_GEMM<NT, true>(base + H, base, hL, wbuf, acc, lane, g);
      }
      float* st_l = stash + (int64_t)l * NT * 256;
      const bool top = (l == nL - 1);
      const bool narrow = (s.skip >= 1 && l == s.skip - 1);
#pragma unroll
      for (int t = 0; t < NT; ++t) reinterpret_cast<f32x4*>(hL)[t * 64 + lane] = acc[t];
      for (int e4i = 0; e4i < NT; ++e4i) {
        const f32x4 z4 = reinterpret_cast<const f32x4*>(hL)[e4i * 64 + lane];
        f32x4 h4, s4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float y, dy;
          softplus_b(z4[i], a.beta, y, dy);
          h4[i] = y; s4[i] = dy;
        }
        if (top) {
          const f32x4 w4 = reinterpret_cast<const f32x4*>(WLimg)[g * NT + e4i];
          fsum += (w4.x * h4.x + w4.y * h4.y) + (w4.z * h4.z + w4.w * h4.w);
          h4 = (f32x4){w4.x * s4.x, w4.y * s4.y, w4.z * s4.z, w4.w * s4.w};
        } else {
          if (narrow) {
            // concat [h, e(x)] / sqrt(2): owned encoding slots replace the zero-padded rows
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int f = 16 * e4i + 4 * g + i;
              if (f >= first_enc_slot) {
                float v, dv; int c;
                posenc(f - first_enc_slot, px, py, pz, v, c, dv);
                h4[i] = v;
                s4[i] = 0.f;            // no adjoint flows into the padded rows of the narrow layer
              }
              h4[i] = h4[i] / inv_sqrt2_den;
            }
          }
          reinterpret_cast<f32x4*>(st_l)[e4i * 64 + lane] = s4;
        }
        reinterpret_cast<f32x4*>(hL)[e4i * 64 + lane] = h4;
      }
    }
    fsum += __shfl_xor(fsum, 16);
    fsum += __shfl_xor(fsum, 32);
    const float fval = tanhf(fsum + b_last);
    const float dtanh = 1.0f - fval * fval;
    // ---- reverse (seed W_n * s_top is already in hL); encoding adjoint -> gradient on the fly
    float gx = 0.f, gy = 0.f, gz = 0.f;
    for (int l = nL - 1; l >= 1; --l) {
      const float* base = a.packed + idr_off_layer(H, l);
      IDR_GEMM<NT, false>(base + H + (int64_t)H * H, nullptr, hL, wbuf, acc, lane, g);
      const float* st_p = stash + (int64_t)(l - 1) * NT * 256;
      const bool cat = (l == s.skip);
#pragma unroll
      for (int t = 0; t < NT; ++t) reinterpret_cast<f32x4*>(hL)[t * 64 + lane] = acc[t];
      for (int t = 0; t < NT; ++t) {
        const f32x4 s4 = reinterpret_cast<const f32x4*>(st_p)[t * 64 + lane];
        f32x4 av = reinterpret_cast<const f32x4*>(hL)[t * 64 + lane];
        if (cat) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            av[i] = av[i] / inv_sqrt2_den;
            const int f = 16 * t + 4 * g + i;
            if (f >= first_enc_slot) {       // adjoint of an encoding slot -> d/dx
              float v, dv; int c;
              posenc(f - first_enc_slot, px, py, pz, v, c, dv);
              const float contrib = av[i] * dv;
              gx += c == 0 ? contrib : 0.f;
              gy += c == 1 ? contrib : 0.f;
              gz += c == 2 ? contrib : 0.f;
            }
          }
        }
        f32x4 gs = {av.x * s4.x, av.y * s4.y, av.z * s4.z, av.w * s4.w};
        reinterpret_cast<f32x4*>(hL)[t * 64 + lane] = gs;
      }
    }
    // ---- layer 0 reverse on the VALU: p_k = sum_f W0[f][k] gs0[f] over this lane's features,
    //      folded with the encoding Jacobian (rolled over the D0 inputs)
    {
      const float* W0v = a.packed + idr_off_w0v(H) + (int64_t)g * kW0Row * (H / 4);
#ifdef IDR_DBG_NOL0REV
      for (int k = 0; k < 1; ++k) {
#else
      for (int k = 0; k < s.D0; ++k) {
#endif
        const f32x4* col = reinterpret_cast<const f32x4*>(W0v + (int64_t)k * (H / 4));
        float pk = 0.f;
        for (int e4i = 0; e4i < NT; ++e4i) {
          const f32x4 a4 = reinterpret_cast<const f32x4*>(hL)[e4i * 64 + lane];
          const f32x4 w = col[e4i];
          pk += (w.x * a4.x + w.y * a4.y) + (w.z * a4.z + w.w * a4.w);
        }
        float v, dv; int c;
        posenc(k, px, py, pz, v, c, dv);
        const float contrib = pk * dv;
        gx += c == 0 ? contrib : 0.f;
        gy += c == 1 ? contrib : 0.f;
        gz += c == 2 ? contrib : 0.f;
      }
    }
    gx += __shfl_xor(gx, 16); gx += __shfl_xor(gx, 32);
    gy += __shfl_xor(gy, 16); gy += __shfl_xor(gy, 32);
    gz += __shfl_xor(gz, 16); gz += __shfl_xor(gz, 32);
    gx *= dtanh; gy *= dtanh; gz *= dtanh;
    const float f = fval;

    bool survive = false;
    if (valid && g == 0) survive = iso_step_finish(a, idx, f, gx, gy, gz);
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
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// Feature-split form (the default; -DISO_IDR_STAGED selects the staged kernel above).
//
// Same idea as siren_x3.hip, with the f32 matrix cores: a workgroup holds the activations of
// P = 64 points (NB = 4 point tiles of 16) in LDS in the B-operand layout, act[q][n][lane][4]
// (= 128 KiB at H = 512), every wave OWNS the output tiles TW*w .. TW*w+TW-1 of a layer and streams
// exactly those rows of the (unchanged) weight image FW[q][t][lane][4] from L2 / Infinity Cache
// straight into registers -- no LDS staging, two workgroup barriers per layer instead of one per 64
// MFMAs.  A q-chunk is TW*NB*4 = 128 MFMAs (4096 cycles) per wave for 8 weight loads and 4 LDS reads
// per lane, which ride behind the MFMAs.  The positional encoding and its derivative are
// evaluated once per tile into a small LDS table.
template <int NT>
struct IdrFs {
  static constexpr int NB = 4, P = 16 * NB, NW = 4, TW = NT / NW;
  static constexpr size_t kActBytes = (size_t)NT * NB * 1024;
  static constexpr int kEncRows = 48;                    // D0 <= 48 (F <= 7) here; wider encodings use the staged kernel
  static constexpr size_t kEncBytes = (size_t)2 * kEncRows * P * sizeof(float);     // value + derivative
  static constexpr size_t kRedBytes = (size_t)NW * P * 16;
  static constexpr size_t kLds = kActBytes + kEncBytes + kRedBytes;
  static_assert(NT % NW == 0, "tiles must split evenly over the waves");
};

// FWD: value only (no stash, no reverse sweep).
template <int NT, bool FWD>
__global__ __launch_bounds__(256, 1) void k_idr_step_fs(IdrArgs a) {
  using S = IdrFs<NT>;
  constexpr int H = NT * 16, NB = S::NB, P = S::P, NW = S::NW, TW = S::TW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  f32x4* act = reinterpret_cast<f32x4*>(smem_raw);                               // [q][n][lane]
  float* encv = reinterpret_cast<float*>(smem_raw + S::kActBytes);               // [k][P]
  float* encd = encv + S::kEncRows * P;
  f32x4* red = reinterpret_cast<f32x4*>(smem_raw + S::kActBytes + S::kEncBytes); // [w][P]
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, g = lane >> 4, j = lane & 15;
  const IdrShape s = a.s;
  const int nL = s.n_layers;
  const float inv_sqrt2_den = 1.41421356237309515f;
  const float* WLimg = a.packed + idr_off_wl(H, nL);
  const float b_last = a.packed[idr_off_wl(H, nL) + H];
  f32x4* stash = reinterpret_cast<f32x4*>(a.stash) +
                 ((int64_t)blockIdx.x * NW + w) * (int64_t)nL * TW * NB * 64 + lane;   // [l][t][n][lane]
  const int first_enc_slot = H - s.D0;
  const unsigned lane_off = (unsigned)lane * 16u;

  // one layer GEMM over q-chunks [0, nq): acc[t][n] (+)= W[own tiles] . act
  f32x4 acc[TW][NB];
  auto gemm = [&](const float* img, const float* bias, int nq) {
#pragma unroll
    for (int t = 0; t < TW; ++t) {
      f32x4 init = {0.f, 0.f, 0.f, 0.f};
      if (bias) init = *reinterpret_cast<const f32x4*>(bias + 16 * (TW * w + t) + 4 * g);
#pragma unroll
      for (int n = 0; n < NB; ++n) acc[t][n] = init;
    }
    f32x4 A[2][TW], B[2][NB];
    auto ldA = [&](f32x4 (&Ar)[TW], int q) {
      const char* pq = reinterpret_cast<const char*>(img) + ((int64_t)q * NT + TW * w) * 1024;
#pragma unroll
      for (int t = 0; t < TW; ++t) Ar[t] = *reinterpret_cast<const f32x4*>(pq + lane_off + t * 1024);
    };
    auto ldB = [&](f32x4 (&Br)[NB], int q) {
#pragma unroll
      for (int n = 0; n < NB; ++n) Br[n] = act[(q * NB + n) * 64 + lane];
    };
    auto mma = [&](const f32x4 (&Ar)[TW], const f32x4 (&Br)[NB]) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
          for (int n = 0; n < NB; ++n)
            acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ar[t][i], Br[n][i], acc[t][n], 0, 0, 0);
    };
    auto pattern = [&]() {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
#pragma unroll
      for (int i = 0; i < TW; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    };
    ldA(A[0], 0);
    ldB(B[0], 0);
    for (int q = 0; q < nq; q += 2) {
      if (q + 1 < nq) { ldA(A[1], q + 1); ldB(B[1], q + 1); }
      mma(A[0], B[0]);
      pattern();
      __builtin_amdgcn_sched_barrier(0);
      if (q + 1 < nq) {
        if (q + 2 < nq) { ldA(A[0], q + 2); ldB(B[0], q + 2); }
        mma(A[1], B[1]);
        pattern();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  const int64_t count = a.count_in ? (int64_t)(*a.count_in) : a.n;
  const int64_t n_tiles = (count + P - 1) / P;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    // ---- encoding table: thread (k-row, point) pairs, [k][P] value and d/dx_c
    {
      const int pt = tid & (P - 1);
      const int64_t slot = tile * P + pt;
      float qx = 0.f, qy = 0.f, qz = 0.f;
      if (slot < count) {
        const int64_t id = a.idx_in ? (int64_t)a.idx_in[slot] : slot;
        qx = a.pts[id * 3]; qy = a.pts[id * 3 + 1]; qz = a.pts[id * 3 + 2];
      }
      for (int k = tid / P; k < S::kEncRows; k += 256 / P) {
        float v = 0.f, dv = 0.f; int c;
        if (k < s.D0) posenc(k, qx, qy, qz, v, c, dv);
        encv[k * P + pt] = v;
        encd[k * P + pt] = dv;
      }
    }
    __syncthreads();
    // B operand of layer 0: chunk q = w (kD0Pad / 16 = 4 chunks, one per wave)
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      f32x4 e4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = 16 * w + 4 * g + i;
        e4[i] = f < S::kEncRows ? encv[f * P + 16 * n + j] : 0.f;
      }
      act[(w * NB + n) * 64 + lane] = e4;
    }
    __syncthreads();
    float fsum[NB], gx[NB], gy[NB], gz[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) fsum[n] = gx[n] = gy[n] = gz[n] = 0.f;
    // ---- forward
    for (int l = 0; l < nL; ++l) {
      if (l == 0) {
        gemm(a.packed + idr_off_fw0(H), a.packed + idr_off_b0(), kD0Pad / 16);
      } else {
        const float* base = a.packed + idr_off_layer(H, l);
        gemm(base + H, base, NT);
      }
      __syncthreads();                               // every wave has read the activations
      const bool top = (l == nL - 1);
      const bool narrow = (s.skip >= 1 && l == s.skip - 1);
      f32x4* st_l = stash + (int64_t)l * TW * NB * 64;
      for (int t = 0; t < TW; ++t) {
        const int tg = TW * w + t;                  // global tile = q-chunk of the next layer
        f32x4 w4 = {0.f, 0.f, 0.f, 0.f};
        if (top) w4 = reinterpret_cast<const f32x4*>(WLimg)[g * NT + tg];
#pragma unroll
        for (int n = 0; n < NB; ++n) {
          // acc[t][n] with a rolled t: select through a static switch (TW <= 8)
          f32x4 z4 = acc[0][n];
#pragma unroll
          for (int tt = 1; tt < TW; ++tt) if (t == tt) z4 = acc[tt][n];
          f32x4 h4, s4;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float y, dy;
            softplus_b(z4[i], a.beta, y, dy);
            h4[i] = y; s4[i] = dy;
          }
          if (top) {
            fsum[n] += (w4.x * h4.x + w4.y * h4.y) + (w4.z * h4.z + w4.w * h4.w);
            if constexpr (FWD) continue;
            h4 = (f32x4){w4.x * s4.x, w4.y * s4.y, w4.z * s4.z, w4.w * s4.w};
          } else {
            if (narrow) {
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const int f = 16 * tg + 4 * g + i;
                if (f >= first_enc_slot) {
                  h4[i] = encv[(f - first_enc_slot) * P + 16 * n + j];
                  s4[i] = 0.f;
                }
                h4[i] = h4[i] / inv_sqrt2_den;
              }
            }
            if constexpr (!FWD) st_l[(t * NB + n) * 64] = s4;
          }
          act[(tg * NB + n) * 64 + lane] = h4;
        }
      }
      __syncthreads();                               // the next layer's inputs are complete
    }
    // ---- reverse (the seed W_n * s_top is in act)
    for (int l = FWD ? 0 : nL - 1; l >= 1; --l) {
      const float* base = a.packed + idr_off_layer(H, l);
      gemm(base + H + (int64_t)H * H, nullptr, NT);
      __syncthreads();
      const f32x4* st_p = stash + (int64_t)(l - 1) * TW * NB * 64;
      const bool cat = (l == s.skip);
      for (int t = 0; t < TW; ++t) {
        const int tg = TW * w + t;
#pragma unroll
        for (int n = 0; n < NB; ++n) {
          f32x4 av = acc[0][n];
#pragma unroll
          for (int tt = 1; tt < TW; ++tt) if (t == tt) av = acc[tt][n];
          const f32x4 s4 = st_p[(t * NB + n) * 64];
          if (cat) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              av[i] = av[i] / inv_sqrt2_den;
              const int f = 16 * tg + 4 * g + i;
              if (f >= first_enc_slot) {             // adjoint of an encoding slot -> d/dx
                const int k = f - first_enc_slot;
                const float contrib = av[i] * encd[k * P + 16 * n + j];
                const int c = k < 3 ? k : (k - 3) % 3;
                gx[n] += c == 0 ? contrib : 0.f;
                gy[n] += c == 1 ? contrib : 0.f;
                gz[n] += c == 2 ? contrib : 0.f;
              }
            }
          }
          const f32x4 gs = {av.x * s4.x, av.y * s4.y, av.z * s4.z, av.w * s4.w};
          act[(tg * NB + n) * 64 + lane] = gs;
        }
      }
      __syncthreads();
    }
    // ---- layer 0 reverse on the VALU over this wave's features: p_k = sum_f W0[f][k] gs0[f]
    if constexpr (!FWD) {
      const float* W0v = a.packed + idr_off_w0v(H) + (int64_t)g * kW0Row * (H / 4);
      for (int k = 0; k < s.D0; ++k) {
        const f32x4* col = reinterpret_cast<const f32x4*>(W0v + (int64_t)k * (H / 4));
        float pk[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) pk[n] = 0.f;
        for (int t = 0; t < TW; ++t) {
          const int tg = TW * w + t;
          const f32x4 wv = col[tg];
#pragma unroll
          for (int n = 0; n < NB; ++n) {
            const f32x4 a4 = act[(tg * NB + n) * 64 + lane];
            pk[n] += (wv.x * a4.x + wv.y * a4.y) + (wv.z * a4.z + wv.w * a4.w);
          }
        }
        const int c = k < 3 ? k : (k - 3) % 3;
#pragma unroll
        for (int n = 0; n < NB; ++n) {
          const float contrib = pk[n] * encd[k * P + 16 * n + j];
          gx[n] += c == 0 ? contrib : 0.f;
          gy[n] += c == 1 ? contrib : 0.f;
          gz[n] += c == 2 ? contrib : 0.f;
        }
      }
    }
    // ---- reduce over the lane groups and the waves
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      float f = fsum[n], x = gx[n], y = gy[n], z = gz[n];
      f += __shfl_xor(f, 16); f += __shfl_xor(f, 32);
      x += __shfl_xor(x, 16); x += __shfl_xor(x, 32);
      y += __shfl_xor(y, 16); y += __shfl_xor(y, 32);
      z += __shfl_xor(z, 16); z += __shfl_xor(z, 32);
      if (g == 0) red[w * P + 16 * n + j] = (f32x4){f, x, y, z};
    }
    __syncthreads();
    bool survive = false;
    int64_t idx = -1;
    {
      const int64_t slot = tile * P + tid;
      if (tid < P && slot < count) {
        f32x4 r = red[tid];
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) {
          const f32x4 q = red[ww * P + tid];
          r.x += q.x; r.y += q.y; r.z += q.z; r.w += q.w;
        }
        const float f = tanhf(r.x + b_last);
        const float dtanh = 1.0f - f * f;
        const float nx = r.y * dtanh, ny = r.z * dtanh, nz = r.w * dtanh;
        idx = a.idx_in ? (int64_t)a.idx_in[slot] : slot;
        survive = iso_step_finish(a, idx, f, nx, ny, nz);
      }
    }
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
    __syncthreads();
  }
}

constexpr int kIdrBlocks = 256;   // one 160 KiB workgroup per CU at H = 512

inline int64_t idr_stash_floats(int H, int n_layers) {
  const int64_t f32k = (int64_t)kIdrBlocks * 4 * n_layers * H * 16;
  const int64_t x16 = idr_x16_stash_floats(H, n_layers);
  return f32k > x16 ? f32k : x16;
}

template <int NT, bool FWD>
void idr_launch_fs(const IdrArgs& a, int blocks, hipStream_t st) {
  const size_t lds = IdrFs<NT>::kLds;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_idr_step_fs<NT, FWD>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  hipLaunchKernelGGL((k_idr_step_fs<NT, FWD>), dim3(blocks), dim3(256), lds, st, a);
}

template <int NT>
void idr_launch(const IdrArgs& a, int blocks, hipStream_t st) {
#ifndef ISO_IDR_STAGED
  // H = 128 leaves only 32 MFMAs per q-chunk and wave: the staged kernel is 7 % faster there
  // (value-only evaluations always take it where it exists: the staged kernel has no forward-only form)
  if ((NT >= 16 || a.fwd_only) && a.s.D0 <= IdrFs<NT>::kEncRows) {
    if (a.fwd_only) idr_launch_fs<NT, true>(a, blocks, st);
    else idr_launch_fs<NT, false>(a, blocks, st);
    return;
  }
#endif
  const size_t lds = (size_t)(5 * NT * 256) * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_idr_step<NT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  hipLaunchKernelGGL(k_idr_step<NT>, dim3(blocks), dim3(256), lds, st, a);
}

int idr_dispatch(const IdrArgs& a, int blocks, hipStream_t st) {
  switch (a.s.H / 16) {
    case 8: idr_launch<8>(a, blocks, st); return 0;
    case 16: idr_launch<16>(a, blocks, st); return 0;
    case 32: idr_launch<32>(a, blocks, st); return 0;
    default: return -1;
  }
}

bool idr_shape_ok(int H, int n_layers, int skip, int F) {
  const int D0 = 3 + 6 * F;
  return (H == 128 || H == 256 || H == 512) && n_layers >= 2 && n_layers <= 12 && F >= 0 && F <= 10 &&
         D0 <= 63 && (skip < 0 || (skip >= 1 && skip < n_layers)) && H > D0;
}

IdrShape mk_shape(int H, int n_layers, int skip, int F) { return IdrShape{H, n_layers, skip, F, 3 + 6 * F}; }

__global__ void k_idr_zero(int32_t* c, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) c[i] = 0;
}

}  // namespace

extern "C" int64_t iso_idr_raw_floats(int hidden, int n_layers, int skip_layer, int n_freq) {
  IdrShape s = mk_shape(hidden, n_layers, skip_layer, n_freq);
  return idr_raw_off(s, n_layers + 1);
}
// The split-fp16 kernel (idr_x16.hip) serves H = 256 / 512 with encodings of <= 39 slots; ISO_IDR_GEMM=f32 in
// the environment keeps the f32-MFMA kernels of this file for every shape.
static bool idr_use_x16(int H, int n_layers, int skip, int F) {
  static int forced = -1;
  if (forced < 0) {
    const char* e = getenv("ISO_IDR_GEMM");
    forced = (e && e[0] == 'f') ? 1 : 0;
  }
  return !forced && idr_x16_supported(H, n_layers, skip, F);
}

// (the packed buffer always has room for the split-fp16 images: its size may not depend on the environment)
extern "C" int64_t iso_idr_packed_floats(int hidden, int n_layers) {
  return idr_total(hidden, n_layers) + idr_x16_floats(hidden, n_layers);
}

extern "C" int iso_idr_pack_weights(const float* raw, float* packed, int hidden, int n_layers,
                                    int skip_layer, int n_freq, void* stream) {
  ISO_REQUIRE(idr_shape_ok(hidden, n_layers, skip_layer, n_freq), ISO_ERR_UNSUPPORTED,
              "iso_idr_pack_weights: unsupported shape (hidden 128/256/512, 2..12 layers, <=10 frequencies)");
  ISO_REQUIRE(raw && packed, ISO_ERR_INVALID, "iso_idr_pack_weights: null pointer");
  IdrShape s = mk_shape(hidden, n_layers, skip_layer, n_freq);
  hipLaunchKernelGGL(k_idr_pack, dim3(iso_stream_grid(idr_total(hidden, n_layers), 256)), dim3(256), 0,
                     (hipStream_t)stream, raw, packed, s);
  if (idr_x16_supported(hidden, n_layers, skip_layer, n_freq))
    idr_x16_pack(raw, packed, hidden, n_layers, skip_layer, n_freq, (hipStream_t)stream);
  ISO_CHECK_LAUNCH("iso_idr_pack_weights");
  return ISO_OK;
}

extern "C" int64_t iso_project_idr_workspace_bytes(int64_t n, int hidden, int n_layers) {
  if (n < 0) n = 0;
  return idr_stash_floats(hidden, n_layers) * 4 + 2 * n * 4 + 128 * 4 + 64;      // [stash][idx A][idx B][counts 64][tile counters 64]
}

struct IdrTrace { const float* dirs; float alpha, bound; };

static int g_idr_dyn_tiles = -1;                 // -1: from ISO_IDR_DYN_TILES (default on), 0 / 1: iso_idr_set_drawn_tiles
static bool idr_dynamic_tiles_enabled() {        // off: every gridDim-th tile (A/B, tests)
  if (g_idr_dyn_tiles < 0) { const char* e = getenv("ISO_IDR_DYN_TILES"); g_idr_dyn_tiles = (e && e[0] == '0') ? 0 : 1; }
  return g_idr_dyn_tiles == 1;
}
extern "C" int iso_idr_set_drawn_tiles(int on) {
  ISO_REQUIRE(on >= -1 && on <= 1, ISO_ERR_INVALID, "iso_idr_set_drawn_tiles: -1 (environment / default), 0 or 1");
  g_idr_dyn_tiles = on;
  return ISO_OK;
}

static int idr_run(const float* pts_in, float* pts_out, float* normals_out, uint8_t* mask_out,
                   float* sdf_out, float* grad_out, int64_t n, const float* packed, int hidden,
                   int n_layers, int skip_layer, int n_freq, float beta, int max_iters, float tol,
                   void* workspace, int64_t workspace_bytes, void* stream, bool eval_only, const char* who,
                   const IdrTrace* trace = nullptr) {
  ISO_REQUIRE(idr_shape_ok(hidden, n_layers, skip_layer, n_freq), ISO_ERR_UNSUPPORTED,
              "%s: unsupported shape (hidden 128/256/512, 2..12 layers, <=10 frequencies)", who);
  ISO_REQUIRE(n >= 0 && max_iters >= 0 && max_iters <= 60, ISO_ERR_INVALID, "%s: bad n / max_iters", who);
  if (n == 0) return ISO_OK;
  ISO_REQUIRE(n < (1ll << 31), ISO_ERR_UNSUPPORTED, "%s: n must fit int32", who);
  ISO_REQUIRE(packed && workspace, ISO_ERR_INVALID, "%s: null pointer", who);
  ISO_REQUIRE(workspace_bytes >= iso_project_idr_workspace_bytes(n, hidden, n_layers), ISO_ERR_WORKSPACE,
              "%s: workspace too small", who);
  hipStream_t st = (hipStream_t)stream;
  float* stash = (float*)workspace;
  int32_t* idxA = (int32_t*)(stash + idr_stash_floats(hidden, n_layers));
  int32_t* idxB = idxA + n;
  int32_t* counts = idxB + n;
  int64_t tiles = (n + 63) / 64;
  int blocks = (int)(tiles < kIdrBlocks ? tiles : kIdrBlocks);
  const bool x16 = idr_use_x16(hidden, n_layers, skip_layer, n_freq);
  auto run = [&](const IdrArgs& args) { return x16 ? idr_x16_launch(&args, n, st) : idr_dispatch(args, blocks, st); };
  IdrArgs a;
  a.packed = packed; a.stash = stash; a.n = n; a.s = mk_shape(hidden, n_layers, skip_layer, n_freq);
  a.beta = beta; a.tol = tol;
  if (eval_only) {
    a.pts = const_cast<float*>(pts_in); a.normals = nullptr; a.mask = nullptr;
    a.sdf_out = sdf_out; a.grad_out = grad_out;
    a.idx_in = nullptr; a.count_in = nullptr; a.idx_out = nullptr; a.count_out = nullptr;
    a.do_move = 0; a.eval_only = 1;
    a.fwd_only = grad_out ? 0 : 1;
    if (x16 && idr_dynamic_tiles_enabled()) {            // one tile counter, zeroed ahead of the launch
      hipLaunchKernelGGL(k_idr_zero, dim3(1), dim3(64), 0, st, counts + 64, 1);
      a.tile_ctr = counts + 64;
    }
    ISO_REQUIRE(run(a) == 0, ISO_ERR_UNSUPPORTED, "%s: unsupported hidden size", who);
  } else {
    if (pts_out != pts_in) (void)hipMemcpyAsync(pts_out, pts_in, (size_t)n * 12, hipMemcpyDeviceToDevice, st);
    hipLaunchKernelGGL(k_idr_zero, dim3(2), dim3(64), 0, st, counts, 128);
    if (trace) {                         // levelset_sampling.py:764,790: active above 0.1 tol, valid up to tol
      a.dirs = trace->dirs; a.alpha = trace->alpha; a.bound = trace->bound;
      a.tol = 0.1f * tol; a.tol_valid = tol; a.fwd_only = 1;
    }
    for (int it = 0; it <= max_iters; ++it) {
      a.pts = pts_out; a.normals = normals_out; a.mask = mask_out; a.sdf_out = trace ? sdf_out : nullptr; a.grad_out = nullptr;
      a.idx_in = (it == 0) ? nullptr : ((it & 1) ? idxA : idxB);
      a.count_in = (it == 0) ? nullptr : counts + it;
      a.idx_out = (it & 1) ? idxB : idxA;
      a.count_out = counts + it + 1;
      a.do_move = (it < max_iters) ? 1 : 0;
      a.eval_only = 0;
      a.tile_ctr = (x16 && it < 64 && idr_dynamic_tiles_enabled()) ? counts + 64 + it : nullptr;
      ISO_REQUIRE(run(a) == 0, ISO_ERR_UNSUPPORTED, "%s: unsupported hidden size", who);
    }
  }
  ISO_CHECK_LAUNCH(who);
  return ISO_OK;
}

extern "C" int iso_project_idr(const float* pts_in, float* pts_out, float* normals_out,
                               uint8_t* mask_out, int64_t n, const float* packed, int hidden,
                               int n_layers, int skip_layer, int n_freq, float beta, int max_iters,
                               float tol, void* workspace, int64_t workspace_bytes, void* stream) {
  ISO_REQUIRE(n == 0 || (pts_in && pts_out && normals_out && mask_out), ISO_ERR_INVALID,
              "iso_project_idr: null pointer");
  return idr_run(pts_in, pts_out, normals_out, mask_out, nullptr, nullptr, n, packed, hidden, n_layers,
                 skip_layer, n_freq, beta, max_iters, tol, workspace, workspace_bytes, stream, false,
                 "iso_project_idr");
}

extern "C" int iso_idr_sdf_grad(const float* pts, float* sdf_out, float* grad_out, int64_t n,
                                const float* packed, int hidden, int n_layers, int skip_layer,
                                int n_freq, float beta, void* workspace, int64_t workspace_bytes,
                                void* stream) {
  ISO_REQUIRE(n == 0 || (pts && sdf_out), ISO_ERR_INVALID, "iso_idr_sdf_grad: null pointer");
  return idr_run(pts, nullptr, nullptr, nullptr, sdf_out, grad_out, n, packed, hidden, n_layers, skip_layer,
                 n_freq, beta, 0, 0.f, workspace, workspace_bytes, stream, true, "iso_idr_sdf_grad");
}

extern "C" int iso_trace_idr(const float* ray0, const float* dirs, float* pts_out, float* sdf_out,
                             uint8_t* mask_out, int64_t n, const float* packed, int hidden,
                             int n_layers, int skip_layer, int n_freq, float beta, float alpha,
                             float bound, int max_iters, float tol, void* workspace,
                             int64_t workspace_bytes, void* stream) {
  ISO_REQUIRE(n == 0 || (ray0 && dirs && pts_out && sdf_out && mask_out), ISO_ERR_INVALID,
              "iso_trace_idr: null pointer");
  const IdrTrace tr = {dirs, alpha, bound};
  return idr_run(ray0, pts_out, nullptr, mask_out, sdf_out, nullptr, n, packed, hidden, n_layers, skip_layer,
                 n_freq, beta, max_iters, tol, workspace, workspace_bytes, stream, false, "iso_trace_idr", &tr);
}
