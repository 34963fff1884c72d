// Ping-pong form of the split-fp16 SIREN step kernel (H = 256): matrix and vector work of a CU run
// SIDE BY SIDE instead of alternating.
//
// In siren_x3.hip every wave multiplies (GEMM stage) and then evaluates sin / cos and cuts the result
// (activation stage); with the split-fp16 products the two kinds of stage are equally long, so the
// matrix pipe idles half of the time.  Here the eight waves of a workgroup are specialised -- waves
// 0..3 (one per SIMD) only issue MFMAs, waves 4..7 only do the activation work -- and a workgroup
// carries TWO sets of 64 points, one phase apart:
//     phase 2k   : MFMA waves  GEMM(set B, stage k-1)      VALU waves  activation(set A, stage k)
//     phase 2k+1 : MFMA waves  GEMM(set A, stage k)        VALU waves  activation(set B, stage k)
// A GEMM wave hands its accumulators to its partner THROUGH THE ACTIVATION BUFFER ITSELF: after the
// barrier that ends a phase (all reads of the set's operands are done) it writes the eight f32 of a
// (K-step, point tile, lane) entry over the two fp16 parts of that entry -- the same 32 bytes --, and
// the partner turns them into the next layer's operand in place, lane for lane.  Two workgroup
// barriers per phase; both roles execute the same barriers in the same loop, so their counts cannot
// drift apart.  Per set: 2L GEMM stages and 2L + 1 activation stages (layer 0 is a vector stage),
// 4L + 2 phases per pair of sets.
// Arithmetic, scales and results are those of siren_x3.hip (same images, same order of operations).
#include <float.h>
#include <type_traits>
#include "siren_common.h"
#include "iso_newton.h"
#include "mfma_split.h"

static_assert(X3_FWD_F16 && X3_BWD_F16 && kAP == 2, "siren_pp.hip is written for the two-part fp16 layout");

#ifndef PP_IL
#define PP_IL false   // see DESIGN 3.1: operand loads pinned between the MFMAs corrupt lanes 16..31 of the B operand when ONE wave per SIMD issues the MFMAs back to back
#endif

namespace {

constexpr int PH = 256, PNS = PH / 16, PNTO = PH / 32, PTW = 2, PNB = 2, PSL = 2 * PTW, PNG = PSL * PNB;
constexpr int PPS = 32 * PNB, PPP = 2 * PPS;                   // points per set / per pair of sets
constexpr int kSetU4 = PNS * PNB * 2 * 64;                     // u32x4 per set: 64 KiB
constexpr size_t kPpAct = (size_t)2 * kSetU4 * 16;
constexpr size_t kPpRed = (size_t)4 * PPP * 16;               // [4 vector waves][128 points] f32x4
constexpr size_t kPpRedm = (size_t)2 * 2 * PPS * 4 * 4;        // [set][buffer][64 points][4 waves] floats
constexpr size_t kPpLds = kPpAct + kPpRed + kPpRedm;
__host__ __device__ constexpr int64_t pp_stash_per_wg(int L) { return (int64_t)4 * 2 * (L + 1) * PNG * 512; }   // floats

// The generic lambdas below capture the argument block by reference, which hides from the compiler that
// its pointers are global memory: without these casts every access becomes a FLAT instruction (which
// counts on both the vector-memory and the LDS counter and forces a full drain before every K-step).
template <class T>
__device__ __forceinline__ T* as_global(T* p) {
  return (T*)(__attribute__((address_space(1))) T*)p;
}

template <bool FWD>
__global__ __launch_bounds__(512, 1) void k_siren_step_pp(SirenArgs a_in) {
  SirenArgs a = a_in;
  a.pts = as_global(a_in.pts); a.normals = as_global(a_in.normals); a.mask = as_global(a_in.mask);
  a.sdf_out = as_global(a_in.sdf_out); a.grad_out = as_global(a_in.grad_out);
  a.idx_in = as_global(a_in.idx_in); a.count_in = as_global(a_in.count_in);
  a.idx_out = as_global(a_in.idx_out); a.count_out = as_global(a_in.count_out);
  a.packed = as_global(a_in.packed); a.stash = as_global(a_in.stash); a.dirs = as_global(a_in.dirs);
  const float* const packed_g = as_global(a_in.packed);
  float* const stash_g = as_global(a_in.stash);
  float* const pts_g = as_global(a_in.pts);
  const int32_t* const idxin_g = as_global(a_in.idx_in);
  const float w0_s = a_in.w0, wh_s = a_in.wh;
  constexpr int H = PH, NS = PNS, NTO = PNTO, TW = PTW, NB = PNB, SL = PSL, NG = PNG, PS = PPS, P = PPP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* act = reinterpret_cast<u32x4*>(smem_raw);
  f32x4* red = reinterpret_cast<f32x4*>(smem_raw + kPpAct);
  float* redm = reinterpret_cast<float*>(smem_raw + kPpAct + kPpRed);
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool mfma_role = w < 4;
  const int m = w & 3;                                         // SIMD / tile-pair index of this wave
  const int lane = tid & 63, h = lane >> 5, j = lane & 31, h8 = h * 8;
  const int L = a.L;
  const int NPH = FWD ? 2 * L + 2 : 4 * L + 2;                 // phases per pair of sets
  const float* X = packed_g + x3_base(H, L);
  const f32x4* W0u = reinterpret_cast<const f32x4*>(X) + SL * m * 16;
  const float* WLu = X + 4 * H + SL * m * 16;
  const float bL = packed_g[off_bl(H)];
  const float* hdr = packed_g + x16_base(H, L);
  // this lane's entries of set S: K-steps SL*m .. SL*m+SL-1, entry k = sl*NB + n, parts 0/1
  u32x4* own0 = act + (size_t)(SL * m) * NB * 2 * 64 + lane;
  f32x4* stash = reinterpret_cast<f32x4*>(stash_g) + ((int64_t)blockIdx.x * 4 + m) * (int64_t)2 * (L + 1) * NG * 128;   // + lane

  auto fwd_img = [&](int l) { return (gimg_t)(packed_g + x16_off_layer(H, L, l)) + (TW * m * 2) * 64; };
  auto rev_img = [&](int l) { return (gimg_t)(packed_g + x16_off_bw(H, L, l)) + (TW * m * 2) * 64; };
  // image of GEMM stage g of a set: forward layers 0..L-1, then reverse layers L-1..0
  auto stage_img = [&](int g) { return g < L ? fwd_img(g) : rev_img(2 * L - 1 - g); };
  const int NGS = FWD ? L : 2 * L;                             // GEMM stages per set
  u32x4 A[4][TW][3];
  if (mfma_role) x3_prefetch_a<TW, NTO, 2>(A, fwd_img(0), 0, lane);
  f32x16 acc[TW][NB];

  const int64_t count = a.count_in ? (int64_t)(*a.count_in) : a.n;
  const int64_t n_tiles = (count + P - 1) / P;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    // state of the vector role, per set
    float fpart[2][NB], gx[2][NB], gy[2][NB], gz[2][NB], bscale[2][NB];
    int mbuf[2] = {0, 0};
#pragma unroll
    for (int S = 0; S < 2; ++S)
#pragma unroll
      for (int n = 0; n < NB; ++n) { fpart[S][n] = gx[S][n] = gy[S][n] = gz[S][n] = 0.f; bscale[S][n] = 1.f; }

    // ---- one activation stage of set S (vector role) -------------------------------------------
    auto valu_stage = [&](auto SC, int v) __attribute__((always_inline)) {
      constexpr int S = decltype(SC)::value;
      u32x4* own = own0 + S * kSetU4;
      f32x4* stS = stash + (int64_t)S * (L + 1) * NG * 128;
      float amax[NB];
#pragma unroll
      for (int n = 0; n < NB; ++n) amax[n] = 0.f;
      auto put_amax = [&]() __attribute__((always_inline)) {
        const int buf = mbuf[S] ^ 1;
#pragma unroll
        for (int n = 0; n < NB; ++n) {
          const float mm = __builtin_fmaxf(amax[n], __shfl_xor(amax[n], 32));
          if (h == 0) redm[(((S * 2 + buf) * PS) + 32 * n + j) * 4 + m] = mm;
        }
        mbuf[S] = buf;
      };
      if (v == 0) {
        // layer 0 (3 -> H) for this wave's 64 features of the 64 points of the set
        float px[NB], py[NB], pz[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) {
          const int64_t slot = tile * P + S * PS + 32 * n + j;
          px[n] = py[n] = pz[n] = 0.f;
          if (slot < count) {
            const int64_t idx = idxin_g ? (int64_t)idxin_g[slot] : slot;
            px[n] = pts_g[idx * 3]; py[n] = pts_g[idx * 3 + 1]; pz[n] = pts_g[idx * 3 + 2];
          }
        }
        for (int sl = 0; sl < SL; ++sl) {
          f32x4 wv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) wv[e] = W0u[sl * 16 + h8 + e];
#pragma unroll
          for (int n = 0; n < NB; ++n) {
            float zz[8], hv[8], sv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) zz[e] = ((wv[e].x * px[n] + wv[e].y * py[n]) + wv[e].z * pz[n]) + wv[e].w;
            iso_sin_wcos8(w0_s, w0_s, zz, hv, sv);
            const int k = sl * NB + n;
            u32x4 p0, p1;
            split8_f16(hv, p0, p1);
            own[(k * 2 + 0) * 64] = p0; own[(k * 2 + 1) * 64] = p1;
            if constexpr (!FWD) {
              stS[(k * 2 + 0) * 64 + lane] = (f32x4){sv[0], sv[1], sv[2], sv[3]};
              stS[(k * 2 + 1) * 64 + lane] = (f32x4){sv[4], sv[5], sv[6], sv[7]};
            }
          }
        }
        return;
      }
      if (v <= L) {
        // forward activation of hidden layer l: the partner's accumulators are in the entries
        const int l = v - 1;
        const bool top = (l == L - 1);
        const float zscale = kActScale * hdr[l];
        const float w_in = wh_s / zscale;
        const float seed_scale = x3_scale_for(hdr[16] * wh_s * 1.01f);
        f32x4* st_l = stS + (int64_t)(l + 1) * NG * 128;
        for (int sl = 0; sl < SL; ++sl) {
#pragma unroll
          for (int n = 0; n < NB; ++n) {
            const int k = sl * NB + n;
            const f32x4 z0 = as_f32x4(own[(k * 2 + 0) * 64]), z1 = as_f32x4(own[(k * 2 + 1) * 64]);
            const float zz[8] = {z0.x, z0.y, z0.z, z0.w, z1.x, z1.y, z1.z, z1.w};
            float hv[8], sv[8];
            iso_sin_wcos8(w_in, wh_s, zz, hv, sv);
            if (top) {
              const f32x4 wl0 = *reinterpret_cast<const f32x4*>(WLu + sl * 16 + h8);
              const f32x4 wl1 = *reinterpret_cast<const f32x4*>(WLu + sl * 16 + h8 + 4);
              const float f0 = (wl0.x * hv[0] + wl0.y * hv[1]) + (wl0.z * hv[2] + wl0.w * hv[3]);
              const float f1 = (wl1.x * hv[4] + wl1.y * hv[5]) + (wl1.z * hv[6] + wl1.w * hv[7]);
              fpart[S][n] += f0 + f1;
              if constexpr (FWD) continue;
#pragma unroll
              for (int e = 0; e < 4; ++e) { hv[e] = wl0[e] * sv[e]; hv[4 + e] = wl1[e] * sv[4 + e]; }
              float mm = amax[n];
#pragma unroll
              for (int e = 0; e < 8; e += 2) mm = __builtin_fmaxf(mm, __builtin_fmaxf(__builtin_fabsf(hv[e]), __builtin_fabsf(hv[e + 1])));
              amax[n] = mm;
              u32x4 p0, p1;
              split8_f16(hv, p0, p1, seed_scale);
              own[(k * 2 + 0) * 64] = p0; own[(k * 2 + 1) * 64] = p1;
            } else {
              if constexpr (!FWD) {
                st_l[(k * 2 + 0) * 64 + lane] = (f32x4){sv[0], sv[1], sv[2], sv[3]};
                st_l[(k * 2 + 1) * 64 + lane] = (f32x4){sv[4], sv[5], sv[6], sv[7]};
              }
              u32x4 p0, p1;
              split8_f16(hv, p0, p1);
              own[(k * 2 + 0) * 64] = p0; own[(k * 2 + 1) * 64] = p1;
            }
          }
        }
        if (top && !FWD) {
#pragma unroll
          for (int n = 0; n < NB; ++n) bscale[S][n] = seed_scale;
          put_amax();
        }
        return;
      }
      // reverse activation of layer lr: adjoint of the layer below, or the gradient for lr == 0
      const int lr = 2 * L - v;
      float Mp[NB], inv[NB], nscale[NB];
      {
        const float iw = 1.0f / hdr[lr];
        const float grow = hdr[8 + lr] * wh_s * 1.01f;
#pragma unroll
        for (int n = 0; n < NB; ++n) {
          const f32x4 mm = *reinterpret_cast<const f32x4*>(redm + (((S * 2 + mbuf[S]) * PS) + 32 * n + j) * 4);
          Mp[n] = __builtin_fmaxf(__builtin_fmaxf(mm.x, mm.y), __builtin_fmaxf(mm.z, mm.w));
          inv[n] = iw / bscale[S][n];
          nscale[n] = lr > 0 ? x3_scale_for(Mp[n] * grow) : 1.0f;
        }
      }
      const f32x4* st_l = stS + (int64_t)lr * NG * 128;
      for (int sl = 0; sl < SL; ++sl) {
        f32x4 wv[8];
        if (lr == 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) wv[e] = W0u[sl * 16 + h8 + e];
        }
#pragma unroll
        for (int n = 0; n < NB; ++n) {
          const int k = sl * NB + n;
          const f32x4 z0 = as_f32x4(own[(k * 2 + 0) * 64]), z1 = as_f32x4(own[(k * 2 + 1) * 64]);
          const f32x4 s0 = st_l[(k * 2 + 0) * 64 + lane], s1 = st_l[(k * 2 + 1) * 64 + lane];
          float av[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) { av[e] = (z0[e] * inv[n]) * s0[e]; av[4 + e] = (z1[e] * inv[n]) * s1[e]; }
          if (lr > 0) {
            float mm = amax[n];
#pragma unroll
            for (int e = 0; e < 8; e += 2) mm = __builtin_fmaxf(mm, __builtin_fmaxf(__builtin_fabsf(av[e]), __builtin_fabsf(av[e + 1])));
            amax[n] = mm;
            u32x4 p0, p1;
            split8_f16(av, p0, p1, nscale[n]);
            own[(k * 2 + 0) * 64] = p0; own[(k * 2 + 1) * 64] = p1;
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              gx[S][n] += wv[e].x * av[e];
              gy[S][n] += wv[e].y * av[e];
              gz[S][n] += wv[e].z * av[e];
            }
          }
        }
      }
      if (lr > 0) {
#pragma unroll
        for (int n = 0; n < NB; ++n) bscale[S][n] = nscale[n];
        put_amax();
      }
    };

    // ---- one GEMM stage of set S (matrix role); the accumulators stay in registers ----------------
    auto mfma_stage = [&](auto SC, int g, int gi_next_valid) __attribute__((always_inline)) {
      constexpr int S = decltype(SC)::value;
      const u32x4* actS = act + S * kSetU4 + lane;
      const gimg_t img = stage_img(g);
      // the stage that follows in this wave's sequence: the other set on the same image, or the next image
      const int gn = (S == 0) ? g : g + 1;
      const gimg_t nxt = (gi_next_valid && gn < NGS) ? stage_img(gn) : fwd_img(0);
      if (g < L) {
        const float* bias = packed_g + x3_off_layer(H, L, g);
        gemm_x3<TW, NB, NTO, NS, kBias, PP_IL, 2, 2>(img, bias, actS, acc, m, 0, A, nxt, 0, lane, kActScale * hdr[g]);
      } else {
        gemm_x3<TW, NB, NTO, NS, kZero, PP_IL, 2, 2>(img, nullptr, actS, acc, m, 0, A, nxt, 0, lane);
      }
    };
    auto write_acc = [&](auto SC) __attribute__((always_inline)) {
      constexpr int S = decltype(SC)::value;
      u32x4* own = own0 + S * kSetU4;
      // the results of the last MFMAs must have landed before an LDS store reads them: the barrier in
      // between is no guarantee when this wave is the last to arrive, and the hazard recogniser does
      // not look across it
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
      for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int n = 0; n < NB; ++n) {
            const int k = (2 * t + p) * NB + n;
            const f32x16& v = acc[t][n];
            own[(k * 2 + 0) * 64] = as_u32x4((f32x4){v[8 * p], v[8 * p + 1], v[8 * p + 2], v[8 * p + 3]});
            own[(k * 2 + 1) * 64] = as_u32x4((f32x4){v[8 * p + 4], v[8 * p + 5], v[8 * p + 6], v[8 * p + 7]});
          }
    };

    // ---- the phases: vector role works on set (ph & 1), matrix role on the other one ---------------
    // matrix stage index gi = ph - 1 (set gi & 1, stage gi >> 1), vector stage index vi = ph
    auto phase = [&](auto QC, int ph) __attribute__((always_inline)) {
      constexpr int Q = decltype(QC)::value;                  // ph & 1
      using SV = std::integral_constant<int, Q>;
      using SG = std::integral_constant<int, 1 - Q>;
      const bool g_on = ph >= 1 && ph <= 2 * NGS;
      const bool v_on = ph <= 2 * NGS + 1;
      if (mfma_role) {
        if (g_on) mfma_stage(SG{}, (ph - 1) >> 1, ph < 2 * NGS);
      } else {
        if (v_on) valu_stage(SV{}, ph >> 1);
      }
      __syncthreads();                                         // every read of this phase is done
      if (mfma_role && g_on) write_acc(SG{});
      __syncthreads();                                         // accumulators / operands are in place
    };
    for (int ph = 0; ph < NPH; ph += 2) {
      phase(std::integral_constant<int, 0>{}, ph);
      phase(std::integral_constant<int, 1>{}, ph + 1);
    }

    // ---- reduce head + gradient over the lane halves and the vector waves -------------------------
    if (!mfma_role) {
#pragma unroll
      for (int S = 0; S < 2; ++S)
#pragma unroll
        for (int n = 0; n < NB; ++n) {
          float f = fpart[S][n] + __shfl_xor(fpart[S][n], 32);
          float x = gx[S][n] + __shfl_xor(gx[S][n], 32);
          float y = gy[S][n] + __shfl_xor(gy[S][n], 32);
          float z = gz[S][n] + __shfl_xor(gz[S][n], 32);
          if (h == 0) red[m * P + S * PS + 32 * n + j] = (f32x4){f, x, y, z};
        }
    }
    __syncthreads();
    bool survive = false;
    int64_t idx = -1;
    {
      const int64_t slot = tile * P + tid;
      if (tid < P && slot < count) {
        f32x4 r = red[tid];
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) {
          const f32x4 q = red[ww * P + tid];
          r.x += q.x; r.y += q.y; r.z += q.z; r.w += q.w;
        }
        const float f = r.x + bL;
        idx = idxin_g ? (int64_t)idxin_g[slot] : slot;
        survive = iso_step_finish(a, idx, f, r.y, r.z, r.w);
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

template <bool FWD>
int launch_pp(const SirenArgs& a, int64_t n_upper, hipStream_t s) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_siren_step_pp<FWD>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPpLds);
    attr_done = true;
  }
  const int64_t tiles = (n_upper + PPP - 1) / PPP;
  const int blocks = (int)(tiles < 256 ? (tiles < 1 ? 1 : tiles) : 256);
  hipLaunchKernelGGL((k_siren_step_pp<FWD>), dim3(blocks), dim3(512), kPpLds, s, a);
  return 0;
}

}  // namespace

bool siren_pp_supported(int H, int L) { return H == 256 && L >= 1 && L <= 8; }
int64_t siren_pp_stash_floats(int L) { return 256 * pp_stash_per_wg(L); }
int siren_pp_launch(const SirenArgs& a, int64_t n_upper, hipStream_t s) {
  return a.fwd_only ? launch_pp<true>(a, n_upper, s) : launch_pp<false>(a, n_upper, s);
}
