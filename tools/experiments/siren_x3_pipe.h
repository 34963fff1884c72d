// NOTE: historical experiment.  Builds with -DX3_PIPE=1 -DX3_FWD_F16=0 -DX3_BWD_F16=0 -fno-slp-vectorize; it was
// last validated and measured on the split-bf16 form of the kernel (profiles v8), before the fp16 products.
// Included by siren_x3.hip (inside its anonymous namespace): the software-pipelined variant of
// the split-bf16 SIREN step kernel.  EXPERIMENT, not the default (build siren_x3.hip with
// -DX3_PIPE=1 -fno-slp-vectorize): it passes the same tests and runs as fast as the plain kernel
// (4.46-4.59 vs 4.55 ms per 1 M evaluations), so the simpler kernel stays the default.
//
// One wave per SIMD (NW = 4).  Measured on the plain kernel (tools/siren_stage_times.py): per
// 96-point tile the matrix pipe is busy 51 % of the time and the sin/cos + split stages (VALU,
// 4 cycles per wave instruction) take 36 % -- strictly one after the other.  Here the activation
// of layer l is fused with the GEMM of layer l+1 and both advance in SL = 2*TW rounds,
//
//   round r of the GEMM consumes K-steps { SL*w' + r : w' = 0..NW-1 }  = group r of every wave,
//   while the wave's VALU produces its group r+1 (sin/cos, split, LDS store) between those MFMAs,
//   pair by pair, spread evenly over the K-steps of the round (sched_group_barrier: 1 MFMA, 7 VALU).
//
// A stage = [park accumulators in the dead tail of the own LDS region] [produce group 0]
// barrier { [MFMA round r || produce group r+1] barrier } x SL.  The barrier that ends a round
// publishes group r+1 and retires the readers of group r; the last one also tells every wave
// that the whole input vector is dead (the next stage may park / overwrite).
//
// What the hardware does with it (tools/probes/mfma_filler.hip, one wave per SIMD, cycles per
// v_mfma_f32_32x32x16_bf16 with N independent fillers behind it): plain v_fma_f32 N=2/4/6/8 ->
// 34.0/36.7/38.4/47.7 (32.0 bare), i.e. ~6 plain VALU ops per MFMA are nearly free; a dependent
// chain 6 -> 55; v_pk_fma_f32 2/4/8 -> 46/59/85 (!): packed f32 ops must not sit beside MFMAs, so
// the in-round producers use plain f32 ops (and SLP vectorisation has to be off, it re-packs them).
// With 7 fillers per MFMA a fused round takes 7.5 k cycles against 5.9 k for an MFMA-only round:
// the 960 VALU ops of a round cost 1.6 k instead of 4.0 k.  That gain (about 3.5 k per forward
// stage) is eaten by the five barriers per stage instead of two, the exposed first group and the
// B-operand prologue after every barrier; the reverse stages, which have little VALU work, get
// slower.  A version of this that pays would need the rounds to be barrier-free.
//
// sin/cos here is the branch-free Cody-Waite + minimax path only (|w z| < 1e5: max abs error
// 1.1e-7 up to 1e6); a larger argument -- impossible for a SIREN with finite weights of sane
// size -- turns that point's SDF into NaN instead of silently using an inaccurate value.

// PACKED: two values per v_pk_* instruction (fewest issue slots: right when no MFMA is in flight);
// !PACKED: plain f32 ops -- beside MFMAs of the same wave a packed f32 op costs ~12 cycles more
// than the two plain ops it replaces (MI355X guide, "price of one filler"), so the groups that
// are produced in the shadow of a GEMM round use the scalar form.
template <bool PACKED>
__device__ __forceinline__ void x3p_sin_wcos8(float w, const float (&z)[8], float (&s)[8], float (&c)[8],
                                              float& amax) {
#ifdef X3_DBG_NOSINCOS
#pragma unroll
  for (int e = 0; e < 8; ++e) { s[e] = w * z[e]; c[e] = w; }
#else
  if constexpr (PACKED) {
    const iso_f32x2 w2 = {w, w};
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      const iso_f32x2 x = (iso_f32x2){z[e], z[e + 1]} * w2;
      iso_f32x2 s2, c2;
      iso_sincos_core2(x, s2, c2);
      c2 = c2 * w2;
      s[e] = s2.x; s[e + 1] = s2.y;
      c[e] = c2.x; c[e + 1] = c2.y;
      amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(x.x), __builtin_fabsf(x.y)));
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = w * z[e];
      float s1, c1;
      iso_sincos_core(x, s1, c1);
      s[e] = s1;
      c[e] = w * c1;
      amax = __builtin_fmaxf(amax, __builtin_fabsf(x));
    }
  }
#endif
}

// two values with plain f32 ops (see above: packed ops are poison beside the wave's own MFMAs)
__device__ __forceinline__ void x3p_sin_wcos2(float w, float z0, float z1, float& s0, float& s1, float& c0,
                                              float& c1, float& amax) {
#ifdef X3_DBG_NOSINCOS
  s0 = w * z0; s1 = w * z1; c0 = w; c1 = w;
#else
  const float x0 = w * z0, x1 = w * z1;
  iso_sincos_core(x0, s0, c0);
  iso_sincos_core(x1, s1, c1);
  c0 *= w; c1 *= w;
  amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(x0), __builtin_fabsf(x1)));
#endif
}

template <int H, int NW, int NB, int MINB>
__global__ __launch_bounds__(64 * NW, MINB) void k_siren_step_x3p(SirenArgs a) {
  using S = X3Shape<H, NW, NB>;
  constexpr int NS = S::NS, NTO = S::NTO, TW = S::TW, SL = S::SL, NG = S::NG, P = S::P;
  static_assert(NW % 4 == 0 && NB <= NW, "round structure: NW K-steps per round, NB chunks of VALU work");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* act = reinterpret_cast<u32x4*>(smem_raw);
  f32x4* red = reinterpret_cast<f32x4*>(smem_raw + S::kActBytes);   // [NW][P] {f,gx,gy,gz}
  const int tid = threadIdx.x;
  // the wave index is wave-uniform: say so, and everything derived from it (weight-image and
  // stash bases) lives in SGPRs; loads then use the scalar-base + lane-offset form
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, h = lane >> 5, j = lane & 31;
  u32x4* own = act + (size_t)(SL * w) * NB * 3 * 64 + lane;      // this wave's K-steps (+lane)
  u32x4* park = own + (size_t)NG * 64;                           // last NG*2 KiB of the region
  const u32x4* actl = act + lane;
  const int L = a.L;
  const float* X = a.packed + x3_base(H, L);
  const f32x4* W0k = reinterpret_cast<const f32x4*>(X) + (SL * w * 2 + h) * 8;
  const float* WLk = X + 4 * H + (SL * w * 2 + h) * 8;
  const float bL = a.packed[off_bl(H)];
  f32x4* stash = reinterpret_cast<f32x4*>(a.stash) +
                 ((int64_t)blockIdx.x * NW + w) * (int64_t)(L + 1) * NG * 128 + lane;
  auto fw_img = [&](int l) {
    return reinterpret_cast<const u32x4*>(a.packed + x3_off_layer(H, L, l) + H) + (TW * w * 3) * 64;
  };
  auto bw_img = [&](int l) {
    return reinterpret_cast<const u32x4*>(a.packed + x3_off_layer(H, L, l) + H + 3 * (H * H / 2)) + (TW * w * 3) * 64;
  };
  // K-step visited at position q of a stage (round-major)
  auto s_of = [](int q) { return SL * (q % NW) + q / NW; };

#ifndef X3P_AD
#define X3P_AD 1
#endif
  constexpr int kPD = X3P_AD, kSets = kPD + 1;     // weight fragments kPD K-steps ahead
  static_assert(NW % kSets == 0 && NS % kSets == 0, "register-set rotation must align with rounds");
  u32x4 A[kSets][TW][3];                 // weight-fragment pipeline, carried across stages
#pragma unroll
  for (int d = 0; d < kPD; ++d) x3_load_a<TW, NTO, 3>(A[d], fw_img(0), s_of(d), lane);

  const int64_t count = a.count_in ? (int64_t)(*a.count_in) : a.n;
  const int64_t n_tiles = (count + P - 1) / P;
#ifdef X3_DBG_TIMES
  long long* dbg = reinterpret_cast<long long*>(a.stash + (int64_t)gridDim.x * S::kStashPerWg(L)) - NW * 128;
#endif
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
#ifdef X3_DBG_TIMES
    const bool dbg_on = blockIdx.x == 0 && tile == (int64_t)gridDim.x;
    int dbg_i = 0;
#endif
    X3_STAMP();
    float px[NB], py[NB], pz[NB], amax[NB], fpart[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      const int64_t slot = tile * P + 32 * n + j;
      px[n] = py[n] = pz[n] = 0.f;
      amax[n] = fpart[n] = 0.f;
      if (slot < count) {
        const int64_t idx = a.idx_in ? (int64_t)a.idx_in[slot] : slot;
        px[n] = a.pts[idx * 3]; py[n] = a.pts[idx * 3 + 1]; pz[n] = a.pts[idx * 3 + 2];
      }
    }
    f32x16 acc[TW][NB];
    u32x4 B[2][NB][3];
    float gq[8], gh[8], gs[8];           // the group being produced pair by pair inside a GEMM round (gz also: adjoint)

    auto ldB = [&](u32x4 (&Br)[NB][3], int s) {
      const u32x4* p = actl + s * (NB * 3 * 64);
#pragma unroll
      for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int c = 0; c < 3; ++c) Br[n][c] = p[(n * 3 + c) * 64];
    };
    auto mma = [&](const u32x4 (&Ar)[TW][3], const u32x4 (&Br)[NB][3]) {
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
      constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
          for (int n = 0; n < NB; ++n)
            acc[t][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Ar[t][PA[q]]),
                                                                __builtin_bit_cast(bf16x8, Br[n][PB[q]]),
                                                                acc[t][n], 0, 0, 0);
    };
    auto store_group = [&](int k, const float (&v)[8]) {
      u32x4 p0, p1, p2;
      split8(v, p0, p1, p2);
      own[(k * 3 + 0) * 64] = p0; own[(k * 3 + 1) * 64] = p1; own[(k * 3 + 2) * 64] = p2;
    };
    auto load_park = [&](int k, float (&z)[8]) {
      const f32x4 z0 = as_f32x4(park[(k * 2 + 0) * 64]);
      const f32x4 z1 = as_f32x4(park[(k * 2 + 1) * 64]);
#pragma unroll
      for (int e = 0; e < 4; ++e) { z[e] = z0[e]; z[4 + e] = z1[e]; }
    };
    // accumulators (optionally times the stash of layer slot `mul_slot`) -> park area.  Only
    // legal after the barrier that ended the previous stage (the tail of the own region is dead).
    auto park_acc = [&](int mul_slot) {
      f32x4 sv[NG][2];
      if (mul_slot >= 0) {
        const f32x4* st = stash + (int64_t)mul_slot * NG * 128;
#pragma unroll
        for (int k = 0; k < NG; ++k) { sv[k][0] = st[(k * 2) * 64]; sv[k][1] = st[(k * 2 + 1) * 64]; }
      }
#pragma unroll
      for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int n = 0; n < NB; ++n) {
            const int k = (2 * t + p) * NB + n;
            f32x4 v0, v1;
#pragma unroll
            for (int e = 0; e < 4; ++e) { v0[e] = acc[t][n][8 * p + e]; v1[e] = acc[t][n][8 * p + 4 + e]; }
            if (mul_slot >= 0) { v0 = v0 * sv[k][0]; v1 = v1 * sv[k][1]; }
            park[(k * 2 + 0) * 64] = as_u32x4(v0);
            park[(k * 2 + 1) * 64] = as_u32x4(v1);
          }
    };
    // One fused stage: `prod(r, n)` produces group (r, n) of this wave's part of the GEMM input.
    // `pair(r, pi)` produces values 2*(pi%4), +1 of group (r, pi/4) with plain f32 ops and finishes the
    // group with its 4th pair; the 4*NB pairs of round r+1 are spread evenly over the NW K-steps of
    // GEMM round r.
    constexpr int PPK = 4 * NB / NW;          // pairs per K-step
    static_assert(PPK * NW == 4 * NB, "pairs must divide evenly over the K-steps of a round");
    auto stage = [&](const u32x4* img, const u32x4* nxt, const float* bias_h, auto&& prod, auto&& pair) {
#pragma unroll
      for (int n = 0; n < NB; ++n) prod(0, n, std::true_type{});
      X3_STAMP();
      __syncthreads();
      X3_STAMP();
#pragma unroll
      for (int t = 0; t < TW; ++t) {
        f32x16 init;
        if (bias_h) {
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const float* bp = bias_h + (2 * (TW * w + t) + p) * 16;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(bp);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(bp + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { init[8 * p + e] = lo[e]; init[8 * p + 4 + e] = hi[e]; }
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) init[r] = 0.f;
        }
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[t][n] = init;
      }
      auto round = [&](int r, auto with_prod) {
        ldB(B[0], s_of(r * NW));
#pragma unroll
        for (int jj = 0; jj < NW; ++jj) {
          const int q = r * NW + jj;
          if (q + kPD < NS) x3_load_a<TW, NTO, 3>(A[(jj + kPD) % kSets], img, s_of(q + kPD), lane);
          else x3_load_a<TW, NTO, 3>(A[(jj + kPD) % kSets], nxt, s_of(q + kPD - NS), lane);
          if (jj + 1 < NW) ldB(B[(jj + 1) & 1], s_of(q + 1));
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (decltype(with_prod)::value) {
            // VALU work issued in the shadow of the MFMAs below
#pragma unroll
            for (int pi = PPK * jj; pi < PPK * (jj + 1); ++pi) pair(r + 1, pi);
          }
          mma(A[jj % kSets], B[jj & 1]);
          if constexpr (decltype(with_prod)::value) {
            {
              // ask for  MFMA, kFill VALU, MFMA, kFill VALU, ...  (plain f32 VALU ops hide behind
              // an MFMA of the same wave up to ~6 per MFMA: tools/probes/mfma_filler.hip)
#ifndef X3P_FILL
#define X3P_FILL 7
#endif
#pragma unroll
              for (int g = 0; g < TW * NB * 6; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, X3P_FILL, 0);
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        X3_STAMP();
        __syncthreads();
        X3_STAMP();
      };
#pragma unroll 1
      for (int r = 0; r < SL - 1; ++r) round(r, std::true_type{});
      round(SL - 1, std::false_type{});
    };

    // ---- S_0: layer 0 (3 -> H, VALU) fused with the GEMM of hidden layer 0 ---------------------
    {
      auto prod0 = [&](int r, int n, auto packed) {
        float zz[8], hv[8], sv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const f32x4 wv = W0k[r * 16 + e];
          zz[e] = ((wv.x * px[n] + wv.y * py[n]) + wv.z * pz[n]) + wv.w;
        }
        x3p_sin_wcos8<decltype(packed)::value>(a.w0, zz, hv, sv, amax[n]);
        const int k = r * NB + n;
        store_group(k, hv);
#ifndef X3_DBG_NOSTASH
        stash[(k * 2 + 0) * 64] = (f32x4){sv[0], sv[1], sv[2], sv[3]};
        stash[(k * 2 + 1) * 64] = (f32x4){sv[4], sv[5], sv[6], sv[7]};
#endif
      };
      auto pair0 = [&](int r, int pi) {
        const int n = pi >> 2, e0 = 2 * (pi & 3), k = r * NB + n;
        const f32x4 w0 = W0k[r * 16 + e0], w1 = W0k[r * 16 + e0 + 1];
        const float z0 = ((w0.x * px[n] + w0.y * py[n]) + w0.z * pz[n]) + w0.w;
        const float z1 = ((w1.x * px[n] + w1.y * py[n]) + w1.z * pz[n]) + w1.w;
        x3p_sin_wcos2(a.w0, z0, z1, gh[e0], gh[e0 + 1], gs[e0], gs[e0 + 1], amax[n]);
        if ((pi & 3) == 3) {
          store_group(k, gh);
#ifndef X3_DBG_NOSTASH
          stash[(k * 2 + 0) * 64] = (f32x4){gs[0], gs[1], gs[2], gs[3]};
          stash[(k * 2 + 1) * 64] = (f32x4){gs[4], gs[5], gs[6], gs[7]};
#endif
        }
      };
      const float* lay = a.packed + x3_off_layer(H, L, 0);
      stage(fw_img(0), L > 1 ? fw_img(1) : bw_img(L - 1), lay + h * 8, prod0, pair0);
    }
    // ---- S_l, l = 1..L-1: sin/cos of hidden layer l-1 fused with the GEMM of hidden layer l --------
    for (int l = 1; l < L; ++l) {
      park_acc(-1);
      f32x4* st_l = stash + (int64_t)l * NG * 128;
      auto prod = [&](int r, int n, auto packed) {
        const int k = r * NB + n;
        float zz[8], hv[8], sv[8];
        load_park(k, zz);
        x3p_sin_wcos8<decltype(packed)::value>(a.wh, zz, hv, sv, amax[n]);
        store_group(k, hv);
#ifndef X3_DBG_NOSTASH
        st_l[(k * 2 + 0) * 64] = (f32x4){sv[0], sv[1], sv[2], sv[3]};
        st_l[(k * 2 + 1) * 64] = (f32x4){sv[4], sv[5], sv[6], sv[7]};
#endif
      };
      auto pair = [&](int r, int pi) {
        const int n = pi >> 2, e0 = 2 * (pi & 3), k = r * NB + n;
        if ((pi & 3) == 0) load_park(k, gq);
        x3p_sin_wcos2(a.wh, gq[e0], gq[e0 + 1], gh[e0], gh[e0 + 1], gs[e0], gs[e0 + 1], amax[n]);
        if ((pi & 3) == 3) {
          store_group(k, gh);
#ifndef X3_DBG_NOSTASH
          st_l[(k * 2 + 0) * 64] = (f32x4){gs[0], gs[1], gs[2], gs[3]};
          st_l[(k * 2 + 1) * 64] = (f32x4){gs[4], gs[5], gs[6], gs[7]};
#endif
        }
      };
      const float* lay = a.packed + x3_off_layer(H, L, l);
      stage(fw_img(l), l + 1 < L ? fw_img(l + 1) : bw_img(L - 1), lay + h * 8, prod, pair);
    }
    // ---- S_L: top sine layer (head dot product, adjoint seed) fused with the first reverse GEMM ----
    {
      park_acc(-1);
      auto prod = [&](int r, int n, auto packed) {
        const int k = r * NB + n;
        float zz[8], hv[8], sv[8];
        load_park(k, zz);
        x3p_sin_wcos8<decltype(packed)::value>(a.wh, zz, hv, sv, amax[n]);
        const f32x4 wl0 = *reinterpret_cast<const f32x4*>(WLk + r * 16);
        const f32x4 wl1 = *reinterpret_cast<const f32x4*>(WLk + r * 16 + 4);
        const float f0 = (wl0.x * hv[0] + wl0.y * hv[1]) + (wl0.z * hv[2] + wl0.w * hv[3]);
        const float f1 = (wl1.x * hv[4] + wl1.y * hv[5]) + (wl1.z * hv[6] + wl1.w * hv[7]);
        fpart[n] += f0 + f1;
#pragma unroll
        for (int e = 0; e < 4; ++e) { hv[e] = wl0[e] * sv[e]; hv[4 + e] = wl1[e] * sv[4 + e]; }
        store_group(k, hv);
      };
      auto pair = [&](int r, int pi) {
        const int n = pi >> 2, e0 = 2 * (pi & 3), k = r * NB + n;
        if ((pi & 3) == 0) load_park(k, gq);
        x3p_sin_wcos2(a.wh, gq[e0], gq[e0 + 1], gh[e0], gh[e0 + 1], gs[e0], gs[e0 + 1], amax[n]);
        if ((pi & 3) == 3) {
          const f32x4 wl0 = *reinterpret_cast<const f32x4*>(WLk + r * 16);
          const f32x4 wl1 = *reinterpret_cast<const f32x4*>(WLk + r * 16 + 4);
          const float f0 = (wl0.x * gh[0] + wl0.y * gh[1]) + (wl0.z * gh[2] + wl0.w * gh[3]);
          const float f1 = (wl1.x * gh[4] + wl1.y * gh[5]) + (wl1.z * gh[6] + wl1.w * gh[7]);
          fpart[n] += f0 + f1;
#pragma unroll
          for (int e = 0; e < 4; ++e) { gh[e] = wl0[e] * gs[e]; gh[4 + e] = wl1[e] * gs[4 + e]; }
          store_group(k, gh);
        }
      };
      stage(bw_img(L - 1), L > 1 ? bw_img(L - 2) : fw_img(0), nullptr, prod, pair);
    }
    // ---- reverse sweep: adjoint * w cos(w z) of the layer below, fused with the next reverse GEMM --
    for (int jl = L - 2; jl >= 0; --jl) {
      park_acc(jl + 1);                 // we hold the adjoint w.r.t. the output of hidden layer jl
      auto prod = [&](int r, int n, auto packed) {
        const int k = r * NB + n;
        float av[8];
        load_park(k, av);
        store_group(k, av);
      };
      auto pair = [&](int r, int pi) {
        const int k = r * NB + (pi >> 2);
        if ((pi & 3) == 0) load_park(k, gq);
        if ((pi & 3) == 3) store_group(k, gq);
      };
      stage(bw_img(jl), jl > 0 ? bw_img(jl - 1) : fw_img(0), nullptr, prod, pair);
    }
    // ---- layer 0 reverse: grad = W0^T (adjoint . w0 cos) -------------------------------------------
    float gx[NB], gy[NB], gz[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) gx[n] = gy[n] = gz[n] = 0.f;
    {
      f32x4 sv[NG][2];
#pragma unroll
      for (int k = 0; k < NG; ++k) { sv[k][0] = stash[(k * 2) * 64]; sv[k][1] = stash[(k * 2 + 1) * 64]; }
#pragma unroll
      for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          f32x4 wv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) wv[e] = W0k[(2 * t + p) * 16 + e];
#pragma unroll
          for (int n = 0; n < NB; ++n) {
            const int k = (2 * t + p) * NB + n;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float av = acc[t][n][8 * p + e] * sv[k][e >> 2][e & 3];
              gx[n] += wv[e].x * av;
              gy[n] += wv[e].y * av;
              gz[n] += wv[e].z * av;
            }
          }
        }
    }
    X3_STAMP();
    // ---- reduce head + gradient over the lane halves and the waves -----------------------------
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      float f = fpart[n] + __shfl_xor(fpart[n], 32);
      float x = gx[n] + __shfl_xor(gx[n], 32);
      float y = gy[n] + __shfl_xor(gy[n], 32);
      float z = gz[n] + __shfl_xor(gz[n], 32);
      const float am = __builtin_fmaxf(amax[n], __shfl_xor(amax[n], 32));
      if (!(am < 1.0e5f)) f = __builtin_nanf("");      // argument outside the validated range
      if (h == 0) red[w * P + 32 * n + j] = (f32x4){f, x, y, z};
    }
    __syncthreads();
    // ---- epilogue: thread tid handles point `tid` of the tile ----------------------------------
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
        const float f = r.x + bL;
        idx = a.idx_in ? (int64_t)a.idx_in[slot] : slot;
        if (a.eval_only) {
          a.sdf_out[idx] = f;
          a.grad_out[idx * 3] = r.y; a.grad_out[idx * 3 + 1] = r.z; a.grad_out[idx * 3 + 2] = r.w;
        } else {
          a.normals[idx * 3] = r.y; a.normals[idx * 3 + 1] = r.z; a.normals[idx * 3 + 2] = r.w;
          const bool active = fabsf(f) > a.tol;
          a.mask[idx] = active ? 0 : 1;
          if (active && a.do_move) {
            float qx = a.pts[idx * 3], qy = a.pts[idx * 3 + 1], qz = a.pts[idx * 3 + 2];
            iso_newton_move(f, r.y, r.z, r.w, qx, qy, qz);
            a.pts[idx * 3] = qx; a.pts[idx * 3 + 1] = qy; a.pts[idx * 3 + 2] = qz;
            survive = true;
          }
        }
      }
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
    X3_STAMP();
    __syncthreads();
  }
}
