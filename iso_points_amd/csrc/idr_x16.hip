// IDR-style SDF (DSS/models/common.py:220-310) on the fp16 matrix cores at f32 accuracy: the
// structure of siren_x3.hip (activations of one point tile set in LDS already cut, every wave owns
// 64 output features and streams exactly those weight rows, two barriers per layer) with the
// split-fp16 products of mfma_split.h.
//
// What differs from the SIREN kernel:
//  * layer 0 is a GEMM too: its B operand is the positional encoding (kD0Pad = 64 slots = four
//    K-steps), built from the per-tile encoding table in LDS;
//  * softplus outputs have no a-priori range, so the FORWARD operand carries a per-point
//    power-of-two scale as well: |h_l[f][p]| <= (max_f sum_k |W_l[f][k]|) max_k |h_{l-1}[k][p]| +
//    max|b_l| + ln2/beta, with the exact per-point maximum of the previous layer exchanged between
//    the waves through LDS (as for the adjoint in siren_x3.hip);
//  * skip connection: the narrow layer's padded output slots take the encoding, everything / sqrt(2)
//    (the image layout of idr.hip: zero rows in the narrow layer, concat folded into the slots);
//  * layer 0 reverse on the VALU straight from the registers: p_k = sum_f W0[f][k] a0[f] is
//    multiplied by d e_k / d x_c as it is formed, only three sums per point leave the wave;
//  * tanh on the head.
// Two fp16 parts are exactly the bytes of one f32: LDS layout (64 points x 512 x 4 B) and weight
// stream (1 MB per layer and direction) equal those of the f32 feature-split kernel of idr.hip
// while the matrix work drops 5.3x.
#include <float.h>
#include "idr_common.h"
#include "iso_newton.h"
// weight fragments requested TWO K-steps ahead here (three in the SIREN kernel): with two output tiles per wave a set is
// 16 registers, and the fourth set was paid for in spills (124 -> 100 spilled VGPRs; 1 M evaluations 29.1 -> 28.4 ms;
// one K-step ahead: 27.5-28.5, not steadier)
#ifndef X3_KAD
#define X3_KAD 2
#endif
#include "mfma_split.h"

static_assert(kAP == 2, "idr_x16.hip is written for the two-part fp16 layout");

namespace {

constexpr int kHdr16 = 128;          // header floats: see x16i_* below
constexpr int kEncRows16 = 40;       // D0 <= 39 (F <= 6); wider encodings use the f32 kernels
constexpr int kMaxL = 16;

// section layout (floats), appended to the f32 images at idr_total(H, nL):
//   header: [0..15] 2^s_l | [16..31] rowsum_l = max_f sum_k |W_l[f][k]| | [32..47] max|b_l| |
//           [48..63] colsum_l = max_k sum_f |W_l[f][k]| | [64] max |W_head|
//   bias_k  nL*H (K-order) | W0k kD0Pad*H: W0[feat(ko)][k] | WLk H | FW16_0 kD0Pad*H |
//   per l = 1..nL-1: FW16_l H*H, BW16_l H*H
__host__ __device__ inline int64_t x16i_base(int H, int nL) { return idr_total(H, nL); }
__host__ __device__ inline int64_t x16i_bias(int H, int nL, int l) { return x16i_base(H, nL) + kHdr16 + (int64_t)l * H; }
__host__ __device__ inline int64_t x16i_w0k(int H, int nL) { return x16i_bias(H, nL, nL); }
__host__ __device__ inline int64_t x16i_wlk(int H, int nL) { return x16i_w0k(H, nL) + (int64_t)kD0Pad * H; }
__host__ __device__ inline int64_t x16i_fw0(int H, int nL) { return x16i_wlk(H, nL) + H; }
__host__ __device__ inline int64_t x16i_fw(int H, int nL, int l) {   // l >= 1
  return x16i_fw0(H, nL) + (int64_t)kD0Pad * H + (int64_t)(l - 1) * 2 * H * H;
}
__host__ __device__ inline int64_t x16i_bw(int H, int nL, int l) { return x16i_fw(H, nL, l) + (int64_t)H * H; }
__host__ __device__ inline int64_t x16i_end(int H, int nL) { return x16i_fw(H, nL, nL); }

__host__ __device__ inline int feat_of_ko(int64_t ko) { return x3_feat((int)(ko >> 4), 8 * (int)((ko >> 3) & 1) + (int)(ko & 7)); }

// effective (zero-padded) weight of hidden layer l: out slot a, in slot b
__device__ __forceinline__ float idr_weff(const float* raw, const IdrShape& s, int l, int a, int b) {
  const int od = idr_out_dim(s, l);
  if (a >= od) return 0.f;
  if (l == 0) return b < s.D0 ? raw[idr_raw_off(s, 0) + (int64_t)a * s.D0 + b] : 0.f;
  return raw[idr_raw_off(s, l) + (int64_t)a * s.H + b];
}

__global__ void k_idr16_stats(const float* __restrict__ raw, float* __restrict__ packed, IdrShape s) {
  __shared__ float s_m[256];
  const int H = s.H, nL = s.n_layers, l = blockIdx.x;
  float* hdr = packed + x16i_base(H, nL);
  auto block_max = [&](float v) {
    s_m[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) s_m[threadIdx.x] = fmaxf(s_m[threadIdx.x], s_m[threadIdx.x + o]);
      __syncthreads();
    }
    const float r = s_m[0];
    __syncthreads();
    return r;
  };
  if (l == nL) {
    const float* Wn = raw + idr_raw_off(s, nL);
    float m = 0.f;
    for (int f = threadIdx.x; f < H; f += 256) m = fmaxf(m, fabsf(Wn[f]));
    m = block_max(m);
    if (threadIdx.x == 0) hdr[64] = m;
    return;
  }
  const int K = (l == 0) ? kD0Pad : H;
  const int od = idr_out_dim(s, l);
  float mx = 0.f, rs = 0.f, bm = 0.f, cs = 0.f;
  // the layer's weights as stored: od rows of nb values (everything else of the padded H x K image is zero).  Batches
  // of 32 unconditional loads per thread, summed in order (one load at a time this kernel took 0.94 ms per packing, and
  // the weights change every step)
  const float* __restrict__ Wl = raw + idr_raw_off(s, l);
  const int nb = l == 0 ? s.D0 : H;
  for (int a = threadIdx.x; a < H; a += 256) {
    float t = 0.f;
    if (a < od) {
      const float* __restrict__ row = Wl + (int64_t)a * nb;
      for (int b0 = 0; b0 < nb; b0 += 32) {
        float v[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) v[q] = fabsf(row[min(b0 + q, nb - 1)]);
#pragma unroll
        for (int q = 0; q < 32; ++q) { const float u = b0 + q < nb ? v[q] : 0.f; t += u; mx = fmaxf(mx, u); }
      }
      bm = fmaxf(bm, fabsf(raw[idr_raw_off(s, l) + (int64_t)od * idr_in_dim(s, l) + a]));
    }
    rs = fmaxf(rs, t);
  }
  for (int b = threadIdx.x; b < nb; b += 256) {
    float t = 0.f;
    for (int a0 = 0; a0 < od; a0 += 32) {
      float v[32];
#pragma unroll
      for (int q = 0; q < 32; ++q) v[q] = fabsf(Wl[(int64_t)min(a0 + q, od - 1) * nb + b]);
#pragma unroll
      for (int q = 0; q < 32; ++q) t += a0 + q < od ? v[q] : 0.f;
    }
    cs = fmaxf(cs, t);
  }
  mx = block_max(mx); rs = block_max(rs); bm = block_max(bm); cs = block_max(cs);
  if (threadIdx.x == 0) {
    float sc = 1.0f;
    if (mx > 0.f && mx < 3.0e38f) {
      int e;
      (void)frexpf(mx, &e);
      int k = 10 - e;                          // 2^k * mx in [512, 1024)
      k = k > 100 ? 100 : (k < -100 ? -100 : k);
      sc = ldexpf(1.0f, k);
    }
    hdr[l] = sc; hdr[16 + l] = rs; hdr[32 + l] = bm; hdr[48 + l] = cs;
  }
}

__global__ void k_idr16_pack(const float* __restrict__ raw, float* __restrict__ packed, IdrShape s) {
  const int H = s.H, nL = s.n_layers, NTO = H / 32;
  const int64_t HH = (int64_t)H * H;
  const int64_t b0 = x16i_bias(H, nL, 0), e0 = x16i_end(H, nL);
  const float* hdr = packed + x16i_base(H, nL);
  for (int64_t o = b0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < e0; o += (int64_t)gridDim.x * blockDim.x) {
    if (o < x16i_w0k(H, nL)) {                                   // biases, K-order
      const int l = (int)((o - b0) / H);
      const int f = feat_of_ko((o - b0) % H);
      const int od = idr_out_dim(s, l);
      packed[o] = f < od ? raw[idr_raw_off(s, l) + (int64_t)od * idr_in_dim(s, l) + f] : 0.f;
    } else if (o < x16i_wlk(H, nL)) {                            // W0k[k][ko] = W0[feat(ko)][k]
      const int64_t q = o - x16i_w0k(H, nL);
      const int k = (int)(q / H), f = feat_of_ko(q % H);
      packed[o] = idr_weff(raw, s, 0, f, k);
    } else if (o < x16i_fw0(H, nL)) {                            // head, K-order
      packed[o] = raw[idr_raw_off(s, nL) + feat_of_ko(o - x16i_wlk(H, nL))];
    } else {                                                     // fp16 images: one 32-bit word = two halves
      int l;
      int64_t q;
      bool bwd = false;
      if (o < x16i_fw(H, nL, 1)) { l = 0; q = o - x16i_fw0(H, nL); }
      else {
        const int64_t r = o - x16i_fw(H, nL, 1);
        l = 1 + (int)(r / (2 * HH));
        q = r % (2 * HH);
        bwd = q >= HH;
        if (bwd) q -= HH;
      }
      const float sc = hdr[l];
      const int d = (int)(q & 3), lane = (int)((q >> 2) & 63);
      const int64_t blk = q >> 8;                                // (s*NTO + To)*2 + part
      const int part = (int)(blk & 1);
      const int To = (int)((blk >> 1) % NTO), ks = (int)(blk / (2 * NTO));
      const int fo = 32 * To + (lane & 31);
      f32x2 v;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int fi = x3_feat(ks, 8 * (lane >> 5) + 2 * d + u);
        v[u] = (bwd ? idr_weff(raw, s, l, fi, fo) : idr_weff(raw, s, l, fo, fi)) * sc;
      }
      const f16x2 h = __builtin_convertvector(v, f16x2);
      const f16x2 lo = __builtin_convertvector(v - __builtin_convertvector(h, f32x2), f16x2);
      reinterpret_cast<unsigned*>(packed)[o] = part == 0 ? __builtin_bit_cast(unsigned, h) : __builtin_bit_cast(unsigned, lo);
    }
  }
}

// ---- the step kernel -------------------------------------------------------------------------
template <int H, int NB>
struct I16Shape {
  static constexpr int NW = 8;
  static constexpr int NS = H / 16, NTO = H / 32, TW = NTO / NW, SL = 2 * TW, NG = SL * NB, P = 32 * NB;
  static constexpr size_t kActBytes = (size_t)NS * NB * kAP * 1024;
  static constexpr size_t kEncBytes = (size_t)2 * kEncRows16 * P * sizeof(float);
  static constexpr size_t kRedBytes = (size_t)NW * P * 16;
  static constexpr size_t kLds = kActBytes + kEncBytes + kRedBytes;
  static constexpr int64_t kStashPerWg(int nL) { return (int64_t)NW * nL * NG * 512; }   // floats
  static_assert(NTO % NW == 0 && TW >= 1, "features must split evenly over the waves");
  static_assert(kLds <= 160 * 1024, "LDS budget");
};

// -DI16_DBG_TIMES (timing experiment): waves 0 and 4 of workgroup 0 stamp the shader clock at every stage
// boundary of their second tile into the tail of the stash workspace (tools/idr_stage_times.py).
#ifdef I16_DBG_TIMES
#define I16_STAMP() do { if (dbg_on && lane == 0 && dbg_i < 128) dbg[w * 128 + dbg_i] = (long long)__builtin_amdgcn_s_memtime(); ++dbg_i; } while (0)
#else
#define I16_STAMP() do {} while (0)
#endif

template <int H, int NB, bool FWD>
__global__ __launch_bounds__(512, 1) void k_idr_step_x16(IdrArgs a) {
  using S = I16Shape<H, NB>;
  constexpr int NW = S::NW, NS = S::NS, NTO = S::NTO, TW = S::TW, SL = S::SL, NG = S::NG, P = S::P;
  constexpr int KS0 = kD0Pad / 16;                         // K-steps of layer 0
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* act = reinterpret_cast<u32x4*>(smem_raw);
  float* encv = reinterpret_cast<float*>(smem_raw + S::kActBytes);          // [k][P]
  float* encd = encv + kEncRows16 * P;
  f32x4* red = reinterpret_cast<f32x4*>(smem_raw + S::kActBytes + S::kEncBytes);   // [NW][P]
  float* redm = reinterpret_cast<float*>(red);                               // [2][P][NW] maxima exchange
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, h = lane >> 5, j = lane & 31, h8 = h * 8;
  const IdrShape s = a.s;
  const int nL = s.n_layers;
  const float* hdr = a.packed + x16i_base(H, nL);
  const float inv_sqrt2_den = 1.41421356237309515f;       // x / np.sqrt(2) as float32
  const int first_enc_slot = H - s.D0;
  u32x4* own = act + (size_t)(SL * w) * NB * kAP * 64 + lane;
  const float* WLu = a.packed + x16i_wlk(H, nL) + SL * w * 16;       // + h8 + ...
  const float* W0u = a.packed + x16i_w0k(H, nL) + SL * w * 16;       // + k*H + h8 + ...
  const float b_last = a.packed[idr_off_wl(H, nL) + H];
  f32x4* stash = reinterpret_cast<f32x4*>(a.stash) + ((int64_t)blockIdx.x * NW + w) * (int64_t)nL * NG * 128;   // + lane
  const float ln2_beta = 0.6931472f / a.beta;

  auto fwd_img = [&](int l) {
#ifdef I16_DBG_ONEIMG      // timing experiment (results wrong): every hidden layer streams the SAME image (L2 resident)
    if (l > 1) l = 1;
#endif
    const float* base = a.packed + (l == 0 ? x16i_fw0(H, nL) : x16i_fw(H, nL, l));
    return reinterpret_cast<const u32x4*>(base) + (TW * w * 2) * 64;
  };
  auto rev_img = [&](int l) {
#ifdef I16_DBG_ONEIMG
    l = 1;
#endif
    return reinterpret_cast<const u32x4*>(a.packed + x16i_bw(H, nL, l)) + (TW * w * 2) * 64;
  };
  u32x4 A[4][TW][3];
  x3_prefetch_a<TW, NTO, 2>(A, fwd_img(0), 0, lane);

  float bscale[NB], amax[NB];
  auto put_amax = [&](int buf) {
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      const float m = __builtin_fmaxf(amax[n], __shfl_xor(amax[n], 32));
      if (h == 0) redm[(buf * P + 32 * n + j) * NW + w] = m;
    }
  };
  auto get_max = [&](int buf, float (&Mp)[NB]) {
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      float m = 0.f;
#pragma unroll
      for (int ww = 0; ww < NW; ++ww) m = __builtin_fmaxf(m, redm[(buf * P + 32 * n + j) * NW + ww]);
      Mp[n] = m;
    }
  };

  const int64_t count = a.count_in ? (int64_t)(*a.count_in) : a.n;
  const int64_t n_tiles = (count + P - 1) / P;
#ifdef I16_DBG_TIMES
  long long* dbg = reinterpret_cast<long long*>(a.stash + (int64_t)gridDim.x * S::kStashPerWg(nL)) - NW * 128;
#endif
  // (the next tile: every gridDim-th, or drawn from a.tile_ctr by thread 0 at the top of this one and handed over
  // through LDS at the tile's last two barriers)
  __shared__ int s_next_tile;
  int64_t next_tile = 0;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile = next_tile) {
    int drawn = 0;
    if (a.tile_ctr && tid == 0) drawn = (int)gridDim.x + atomicAdd(a.tile_ctr, 1);
#ifdef I16_DBG_TIMES
    const bool dbg_on = blockIdx.x == 0 && tile == (int64_t)gridDim.x;
    int dbg_i = 0;
#endif
    I16_STAMP();
    // ---- encoding table: thread (k-row, point) pairs, value and d/dx_c -------------------------
    if (tid < (512 / P) * P) {
      const int pt = tid % P;
      const int64_t slot = tile * P + pt;
      float qx = 0.f, qy = 0.f, qz = 0.f;
      if (slot < count) {
        const int64_t id = a.idx_in ? (int64_t)a.idx_in[slot] : slot;
        qx = a.pts[id * 3]; qy = a.pts[id * 3 + 1]; qz = a.pts[id * 3 + 2];
      }
      for (int k = tid / P; k < kEncRows16; k += 512 / P) {
        float v = 0.f, dv = 0.f; int c;
        if (k < s.D0) posenc(k, qx, qy, qz, v, c, dv);
        encv[k * P + pt] = v;
        encd[k * P + pt] = dv;
      }
    }
    __syncthreads();
    // the largest encoding entry of this lane's points (rows 0..2 are the coordinates, the rest <= 1)
    float Mp[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      const int pt = 32 * n + j;
      Mp[n] = __builtin_fmaxf(1.0f, __builtin_fmaxf(__builtin_fabsf(encv[pt]),
                                  __builtin_fmaxf(__builtin_fabsf(encv[P + pt]), __builtin_fabsf(encv[2 * P + pt]))));
      bscale[n] = x3_scale_for(Mp[n] * 1.01f);
    }
    float emax[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) emax[n] = Mp[n];
    // B operand of layer 0: the KS0*NB entries are spread over the waves
    for (int ent = w; ent < KS0 * NB; ent += NW) {
      const int ks = ent / NB, n = ent % NB;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int f = x3_feat(ks, h8 + e);
        v[e] = f < kEncRows16 ? encv[f * P + 32 * n + j] : 0.f;
      }
      float sc = bscale[0];
#pragma unroll
      for (int nn = 1; nn < NB; ++nn) sc = (n == nn) ? bscale[nn] : sc;
      u32x4 p0, p1;
      split8_f16(v, p0, p1, sc);
      act[((ks * NB + n) * kAP + 0) * 64 + lane] = p0;
      act[((ks * NB + n) * kAP + 1) * 64 + lane] = p1;
    }
    __syncthreads();

    I16_STAMP();
    f32x16 acc[TW][NB];
    float fsum[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) fsum[n] = 0.f;
    int mbuf = 0;                              // exchange buffer written last
    // ---- forward -------------------------------------------------------------------------------
    for (int l = 0; l < nL; ++l) {
      const u32x4* img = fwd_img(l);
      const float* bias = a.packed + x16i_bias(H, nL, l);
      const float wsc = hdr[l];
      float zs[NB];
#pragma unroll
      for (int n = 0; n < NB; ++n) zs[n] = wsc * bscale[n];
      const bool top = (l == nL - 1);
      if (l == 0) {
        gemm_x3<TW, NB, NTO, KS0, kBias, true, 2, 2>(img, bias, act + lane, acc, w, 0, A, fwd_img(1), 0, lane, 1.0f, zs);
      } else if (!top || FWD) {
        gemm_x3<TW, NB, NTO, NS, kBias, true, 2, 2>(img, bias, act + lane, acc, w, 0, A, top ? fwd_img(0) : fwd_img(l + 1),
                                                    0, lane, 1.0f, zs);
      } else {
        gemm_x3<TW, NB, NTO, NS, kBias, true, 2, 2>(img, bias, act + lane, acc, w, 0, A, rev_img(nL - 1), 0, lane, 1.0f, zs);
      }
      I16_STAMP();
      __syncthreads();                                   // every wave has read the activations
      I16_STAMP();
      // bound of this layer's output per point -> scale of the next operand
      const bool narrow = (s.skip >= 1 && l == s.skip - 1);
      float nscale[NB], iz[NB];
      const float seed_scale = x3_scale_for(hdr[64] * 1.01f);       // |W_head * sigma'| <= max|W_head|
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        float bound = hdr[16 + l] * Mp[n] + hdr[32 + l] + ln2_beta;
        if (narrow) bound = __builtin_fmaxf(bound, emax[n]);          // (/ sqrt(2) ignored: looser)
        nscale[n] = top ? seed_scale : x3_scale_for(bound * 1.01f);
        iz[n] = 1.0f / zs[n];
        amax[n] = 0.f;
      }
      f32x4* st_l = stash + (int64_t)l * NG * 128;
#pragma unroll
      for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int n = 0; n < NB; ++n) {
            const int k = (2 * t + p) * NB + n;
            const int sl = 2 * t + p;
            float hv[8], sv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) softplus_b(acc[t][n][8 * p + e] * iz[n], a.beta, hv[e], sv[e]);
            if (top) {
              const f32x4 wl0 = *reinterpret_cast<const f32x4*>(WLu + sl * 16 + h8);
              const f32x4 wl1 = *reinterpret_cast<const f32x4*>(WLu + sl * 16 + h8 + 4);
              fsum[n] += ((wl0.x * hv[0] + wl0.y * hv[1]) + (wl0.z * hv[2] + wl0.w * hv[3])) +
                         ((wl1.x * hv[4] + wl1.y * hv[5]) + (wl1.z * hv[6] + wl1.w * hv[7]));
              if constexpr (FWD) continue;
#pragma unroll
              for (int e = 0; e < 4; ++e) { hv[e] = wl0[e] * sv[e]; hv[4 + e] = wl1[e] * sv[4 + e]; }
            } else {
              if (narrow) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                  const int f = 32 * (TW * w + t) + 16 * p + 8 * (e >> 2) + 4 * h + (e & 3);
                  if (f >= first_enc_slot) { hv[e] = encv[(f - first_enc_slot) * P + 32 * n + j]; sv[e] = 0.f; }
                  hv[e] = hv[e] / inv_sqrt2_den;
                }
              }
              if constexpr (!FWD) {
                st_l[(k * 2 + 0) * 64 + lane] = (f32x4){sv[0], sv[1], sv[2], sv[3]};
                st_l[(k * 2 + 1) * 64 + lane] = (f32x4){sv[4], sv[5], sv[6], sv[7]};
              }
            }
            float m = amax[n];
#pragma unroll
            for (int e = 0; e < 8; e += 2) m = __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(hv[e]), __builtin_fabsf(hv[e + 1])));
            amax[n] = m;
            u32x4 p0, p1;
            split8_f16(hv, p0, p1, nscale[n]);
            own[(k * kAP + 0) * 64] = p0; own[(k * kAP + 1) * 64] = p1;
#ifndef I16_NO_GROUP_BARRIER
            __builtin_amdgcn_sched_barrier(0);     // one group at a time: bounds register pressure
#endif
          }
#pragma unroll
      for (int n = 0; n < NB; ++n) bscale[n] = nscale[n];
      if (!(top && FWD)) { mbuf ^= 1; put_amax(mbuf); }
      I16_STAMP();
      __syncthreads();                                   // the next stage's inputs (and maxima) are complete
      I16_STAMP();
      if (!(top && FWD)) get_max(mbuf, Mp);
    }
    // ---- reverse (the seed W_n * sigma'_top is in LDS) -----------------------------------------
    float gx[NB], gy[NB], gz[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) gx[n] = gy[n] = gz[n] = 0.f;
    for (int l = FWD ? 0 : nL - 1; l >= 1; --l) {
      const u32x4* img = rev_img(l);
      if (l > 1) gemm_x3<TW, NB, NTO, NS, kZero, true, 2, 2>(img, nullptr, act + lane, acc, w, 0, A, rev_img(l - 1), 0, lane);
      else gemm_x3<TW, NB, NTO, NS, kZero, true, 2, 2>(img, nullptr, act + lane, acc, w, 0, A, fwd_img(0), 0, lane);
      const f32x4* st_p = stash + (int64_t)(l - 1) * NG * 128;
      // sigma' of the layer below: two groups in flight (all NG at once would cost 64 registers)
      f32x4 svq[2][2];
      svq[0][0] = st_p[lane]; svq[0][1] = st_p[64 + lane];
      I16_STAMP();
      __syncthreads();
      I16_STAMP();
      const bool cat = (l == s.skip);
      float inv[NB], nscale[NB];
      const float iw = 1.0f / hdr[l];
      const float grow = hdr[48 + l] * 1.01f;              // sigma' <= 1
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        inv[n] = iw / bscale[n];
        nscale[n] = x3_scale_for(Mp[n] * grow);
        amax[n] = 0.f;
      }
#pragma unroll
      for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int n = 0; n < NB; ++n) {
            const int k = (2 * t + p) * NB + n;
            if (k + 1 < NG) {
              svq[(k + 1) & 1][0] = st_p[((k + 1) * 2) * 64 + lane];
              svq[(k + 1) & 1][1] = st_p[((k + 1) * 2 + 1) * 64 + lane];
            }
            const f32x4 (&svk)[2] = svq[k & 1];
            float av[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float v = acc[t][n][8 * p + e] * inv[n];
              if (cat) {
                v = v / inv_sqrt2_den;
                const int f = 32 * (TW * w + t) + 16 * p + 8 * (e >> 2) + 4 * h + (e & 3);
                if (f >= first_enc_slot) {               // adjoint of an encoding slot -> d/dx
                  const int kk = f - first_enc_slot;
                  const float contrib = v * encd[kk * P + 32 * n + j];
                  const int c = kk < 3 ? kk : (kk - 3) % 3;
                  gx[n] += c == 0 ? contrib : 0.f;
                  gy[n] += c == 1 ? contrib : 0.f;
                  gz[n] += c == 2 ? contrib : 0.f;
                }
              }
              av[e] = v * svk[e >> 2][e & 3];
              acc[t][n][8 * p + e] = av[e];              // kept for the layer-0 reverse (l == 1)
            }
            if (l > 1) {
              float m = amax[n];
#pragma unroll
              for (int e = 0; e < 8; e += 2) m = __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(av[e]), __builtin_fabsf(av[e + 1])));
              amax[n] = m;
              u32x4 p0, p1;
              split8_f16(av, p0, p1, nscale[n]);
              own[(k * kAP + 0) * 64] = p0; own[(k * kAP + 1) * 64] = p1;
            }
          }
      if (l > 1) {
#pragma unroll
        for (int n = 0; n < NB; ++n) bscale[n] = nscale[n];
        mbuf ^= 1;
        put_amax(mbuf);
      }
      I16_STAMP();
      __syncthreads();
      I16_STAMP();
      if (l > 1) get_max(mbuf, Mp);
    }
    // ---- layer 0 reverse on the VALU: acc holds a0 = adjoint of z0 for this lane's features -------
#ifdef I16_DBG_NOL0REV
    if (false) {
#else
    if constexpr (!FWD) {
#endif
      // weight rows of k+1 are requested while row k is multiplied (39 dependent L2 round trips otherwise)
      f32x4 wq[2][TW * 2][2];
      auto ldw = [&](f32x4 (&dst)[TW * 2][2], int k) {
#pragma unroll
        for (int g2 = 0; g2 < TW * 2; ++g2) {
          const float* wp = W0u + (int64_t)k * H + g2 * 16 + h8;
          dst[g2][0] = *reinterpret_cast<const f32x4*>(wp);
          dst[g2][1] = *reinterpret_cast<const f32x4*>(wp + 4);
        }
      };
      auto row = [&](const f32x4 (&wr)[TW * 2][2], int k) {
        const int c = k < 3 ? k : (k - 3) % 3;
        float pk[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) pk[n] = 0.f;
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const f32x4 w0 = wr[2 * t + p][0], w1 = wr[2 * t + p][1];
#pragma unroll
            for (int n = 0; n < NB; ++n) {
              const f32x16& v = acc[t][n];
              pk[n] += ((w0.x * v[8 * p] + w0.y * v[8 * p + 1]) + (w0.z * v[8 * p + 2] + w0.w * v[8 * p + 3])) +
                       ((w1.x * v[8 * p + 4] + w1.y * v[8 * p + 5]) + (w1.z * v[8 * p + 6] + w1.w * v[8 * p + 7]));
            }
          }
#pragma unroll
        for (int n = 0; n < NB; ++n) {
          const float contrib = pk[n] * encd[k * P + 32 * n + j];
          gx[n] += c == 0 ? contrib : 0.f;
          gy[n] += c == 1 ? contrib : 0.f;
          gz[n] += c == 2 ? contrib : 0.f;
        }
      };
      if constexpr (TW == 1) {
        ldw(wq[0], 0);
        for (int k = 0; k < s.D0; k += 2) {
          if (k + 1 < s.D0) ldw(wq[1], k + 1);
          row(wq[0], k);
          if (k + 1 < s.D0) {
            if (k + 2 < s.D0) ldw(wq[0], k + 2);
            row(wq[1], k + 1);
          }
        }
      } else {               // two tiles per wave: the second register set would be spilled
        for (int k = 0; k < s.D0; ++k) { ldw(wq[0], k); row(wq[0], k); }
      }
    }
    I16_STAMP();
    // ---- reduce over the lane halves and the waves ---------------------------------------------
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      float f = fsum[n] + __shfl_xor(fsum[n], 32);
      float x = gx[n] + __shfl_xor(gx[n], 32);
      float y = gy[n] + __shfl_xor(gy[n], 32);
      float z = gz[n] + __shfl_xor(gz[n], 32);
      if (h == 0) red[w * P + 32 * n + j] = (f32x4){f, x, y, z};
    }
    if (a.tile_ctr && tid == 0) s_next_tile = drawn;
    __syncthreads();
    next_tile = a.tile_ctr ? (int64_t)s_next_tile : tile + gridDim.x;
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
        idx = a.idx_in ? (int64_t)a.idx_in[slot] : slot;
        survive = iso_step_finish(a, idx, f, r.y * dtanh, r.z * dtanh, r.w * dtanh);
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

template <int H, int NB, bool FWD>
int launch_i16(const IdrArgs& a, int64_t n_upper, hipStream_t st) {
  using S = I16Shape<H, NB>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_idr_step_x16<H, NB, FWD>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::kLds);
    attr_done = true;
  }
  const int64_t tiles = (n_upper + S::P - 1) / S::P;
  const int blocks = (int)(tiles < 256 ? (tiles < 1 ? 1 : tiles) : 256);
  hipLaunchKernelGGL((k_idr_step_x16<H, NB, FWD>), dim3(blocks), dim3(512), S::kLds, st, a);
  return 0;
}

}  // namespace

// H = 512: 16 tiles of 32 rows, two per wave, 64 points per workgroup (128 KiB of activations);
// H = 256: one tile per wave, 96 points.
bool idr_x16_supported(int H, int n_layers, int skip, int F) {
  (void)skip;
  return (H == 512 || H == 256) && n_layers >= 2 && n_layers <= kMaxL && 3 + 6 * F <= kEncRows16;
}
int64_t idr_x16_floats(int H, int n_layers) { return x16i_end(H, n_layers) - x16i_base(H, n_layers); }
int64_t idr_x16_stash_floats(int H, int n_layers) {
  if (H == 512) return 256 * I16Shape<512, 2>::kStashPerWg(n_layers);
  if (H == 256) return 256 * I16Shape<256, 3>::kStashPerWg(n_layers);
  return 0;
}
void idr_x16_pack(const float* raw, float* packed, int H, int n_layers, int skip, int F, hipStream_t st) {
  IdrShape s{H, n_layers, skip, F, 3 + 6 * F};
  hipLaunchKernelGGL(k_idr16_stats, dim3(n_layers + 1), dim3(256), 0, st, raw, packed, s);
  hipLaunchKernelGGL(k_idr16_pack, dim3(iso_stream_grid(idr_x16_floats(H, n_layers), 256)), dim3(256), 0, st, raw, packed, s);
}
int idr_x16_launch(const void* idr_args, int64_t n_upper, hipStream_t st) {
  const IdrArgs& a = *static_cast<const IdrArgs*>(idr_args);
  const int H = a.s.H;
  if (a.fwd_only) {
    if (H == 512) return launch_i16<512, 2, true>(a, n_upper, st);
    if (H == 256) return launch_i16<256, 3, true>(a, n_upper, st);
  }
  if (H == 512) return launch_i16<512, 2, false>(a, n_upper, st);
  if (H == 256) return launch_i16<256, 3, false>(a, n_upper, st);
  return -1;
}
