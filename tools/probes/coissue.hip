// Probe: does the VALU work of one activation group (8 values: range reduction, v_sin / v_cos, two-way
// fp16 cut) hide behind the MFMAs of a GEMM stage when both are issued by the same waves, two waves per
// SIMD (the shape of k_siren_step_x3)?  One block = 24 x v_mfma_f32_32x32x16_f16 (four K-steps of a
// 64-point tile pair) with the group's instructions spread evenly behind them, in program order (asm volatile).
// hipcc --offload-arch=gfx950 -O2 coissue.hip -o coissue && ./coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct Regs {
  float z[8], x[8], t[8], n[8], f[8], s[8], c[8], r[8];
  f32x2 z2[4], x2[4], t2[4], f2[4], c2[4], r2[4];
  unsigned h[4], l[4];
  float mx;
};

// MODE 0: plain f32 ops only; 1: packed f32 where the kernel uses them today; 2: plain, v_sin/v_cos replaced by fma;
// 3: plain, conversions replaced by fma
template <int MODE>
struct Prog {
  // steps (element counts): mul, mul, rndne, fma, fma, sin, cos, mul, max3(4), mul, cvtpk(4), cvtback(8), sub, cvtpk(4)
  static constexpr int kPlainCnt[14] = {8, 8, 8, 8, 8, 8, 8, 8, 4, 8, 4, 8, 8, 4};
  static constexpr int kPackCnt[14] = {4, 4, 8, 4, 4, 8, 8, 4, 4, 4, 4, 8, 4, 4};
  static constexpr int cnt(int s) { return MODE == 1 ? kPackCnt[s] : kPlainCnt[s]; }
  static constexpr int total() { int t = 0; for (int s = 0; s < 14; ++s) t += cnt(s); return t; }
  static constexpr int step_of(int i) { int s = 0; while (i >= cnt(s)) { i -= cnt(s); ++s; } return s; }
  static constexpr int elem_of(int i) { int s = 0; while (i >= cnt(s)) { i -= cnt(s); ++s; } return i; }
};

template <int MODE, int I>
__device__ __forceinline__ void op(Regs& R, float k1, float k2) {
  constexpr int s = Prog<MODE>::step_of(I), e = Prog<MODE>::elem_of(I);
  constexpr bool PK = MODE == 1;
  if constexpr (s == 0) { if constexpr (PK) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(R.x2[e]) : "v"(R.z2[e]), "v"(R.c2[0])); else asm volatile("v_mul_f32 %0, %1, %2" : "=v"(R.x[e]) : "v"(R.z[e]), "v"(k1)); }
  if constexpr (s == 1) { if constexpr (PK) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(R.t2[e]) : "v"(R.x2[e]), "v"(R.c2[1])); else asm volatile("v_mul_f32 %0, %1, %2" : "=v"(R.t[e]) : "v"(R.x[e]), "v"(k2)); }
  if constexpr (s == 2) { if constexpr (PK) asm volatile("v_rndne_f32 %0, %1" : "=v"(R.n[e]) : "v"(R.t2[e >> 1][e & 1])); else asm volatile("v_rndne_f32 %0, %1" : "=v"(R.n[e]) : "v"(R.t[e])); }
  if constexpr (s == 3) { if constexpr (PK) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(R.f2[e]) : "v"(R.x2[e]), "v"(R.c2[1]), "v"(R.t2[e])); else asm volatile("v_fma_f32 %0, %1, %2, -%3" : "=v"(R.f[e]) : "v"(R.x[e]), "v"(k2), "v"(R.n[e])); }
  if constexpr (s == 4) { if constexpr (PK) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(R.f2[e]) : "v"(R.x2[e]), "v"(R.c2[2])); else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(R.f[e]) : "v"(R.x[e]), "v"(k1)); }
  if constexpr (s == 5) {
    if constexpr (MODE == 2) asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(R.s[e]) : "v"(R.f[e]), "v"(k1));
    else if constexpr (PK) asm volatile("v_sin_f32 %0, %1" : "=v"(R.s[e]) : "v"(R.f2[e >> 1][e & 1]));
    else asm volatile("v_sin_f32 %0, %1" : "=v"(R.s[e]) : "v"(R.f[e]));
  }
  if constexpr (s == 6) {
    if constexpr (MODE == 2) asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(R.c[e]) : "v"(R.f[e]), "v"(k2));
    else if constexpr (PK) asm volatile("v_cos_f32 %0, %1" : "=v"(R.c[e]) : "v"(R.f2[e >> 1][e & 1]));
    else asm volatile("v_cos_f32 %0, %1" : "=v"(R.c[e]) : "v"(R.f[e]));
  }
  if constexpr (s == 7) { if constexpr (PK) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(R.c2[e]) : "v"(R.x2[e]), "v"(R.t2[0])); else asm volatile("v_mul_f32 %0, %0, %1" : "+v"(R.c[e]) : "v"(k1)); }
  if constexpr (s == 8) asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(R.mx) : "v"(R.x[2 * e]), "v"(R.x[2 * e + 1]));
  if constexpr (s == 9) { if constexpr (PK) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(R.r2[e]) : "v"(R.x2[e]), "v"(R.t2[1])); else asm volatile("v_mul_f32 %0, %1, %2" : "=v"(R.r[e]) : "v"(R.s[e]), "v"(k2)); }
  if constexpr (s == 10) {
    if constexpr (MODE == 3) asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(R.h[e]) : "v"(R.r[2 * e]), "v"(R.r[2 * e + 1]));
    else asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(R.h[e]) : "v"(R.r[2 * e]), "v"(R.r[2 * e + 1]));
  }
  if constexpr (s == 11) {
    if constexpr (MODE == 3) asm volatile("v_fma_f32 %0, %1, %1, %1" : "=v"(R.t[e]) : "v"(R.h[e >> 1]));
    else if constexpr ((e & 1) == 0) asm volatile("v_cvt_f32_f16_e32 %0, %1" : "=v"(R.t[e]) : "v"(R.h[e >> 1]));
    else asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(R.t[e]) : "v"(R.h[e >> 1]));
  }
  if constexpr (s == 12) { if constexpr (PK) asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(R.r2[e]) : "v"(R.r2[e]), "v"(R.t2[e])); else asm volatile("v_sub_f32 %0, %0, %1" : "+v"(R.r[e]) : "v"(R.t[e])); }
  if constexpr (s == 13) {
    if constexpr (MODE == 3) asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(R.l[e]) : "v"(R.r[2 * e]), "v"(R.r[2 * e + 1]));
    else asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(R.l[e]) : "v"(R.r[2 * e]), "v"(R.r[2 * e + 1]));
  }
}

template <int MODE, int LO, int HI>
__device__ __forceinline__ void ops(Regs& R, float k1, float k2) {
  if constexpr (LO < HI) { op<MODE, LO>(R, k1, k2); ops<MODE, LO + 1, HI>(R, k1, k2); }
}

// GROUPS: activation groups spread over the 24 MFMAs (1 = the ratio of the hidden layers, 5.7 issue slots per MFMA)
template <int MODE, bool MFMA, bool FILL, int GROUPS, int M, int ORDER = 0>
__device__ __forceinline__ void block(Regs& R, f32x16 (&acc)[3], f16x8 a, f16x8 b, float k1, float k2) {
  if constexpr (M < 24) {
    if constexpr (MFMA) {
      // ORDER 0: three accumulators round robin; 1: two accumulators alternating; 2: two accumulators, three dependent MFMAs in a row
      constexpr int AI = ORDER == 0 ? M % 3 : (ORDER == 1 ? M % 2 : (M / 3) % 2);
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[AI]) : "v"(a), "v"(b));
    }
    if constexpr (FILL) {
      constexpr int T = Prog<MODE>::total();
#define GR(G) if constexpr (GROUPS > G) ops<MODE, (M * T) / 24, ((M + 1) * T) / 24>(R, k1, k2);
      GR(0) GR(1) GR(2)
#undef GR
    }
    block<MODE, MFMA, FILL, GROUPS, M + 1, ORDER>(R, acc, a, b, k1, k2);
  }
}

template <int MODE, bool MFMA, bool FILL, int GROUPS, int ORDER = 0>
__global__ __launch_bounds__(512, 1) void k(long long* out, float* sink, int iters) {
  f32x16 acc[3] = {};
  f16x8 a = {}, b = {};
  Regs R;
#pragma unroll
  for (int i = 0; i < 8; ++i) { R.z[i] = threadIdx.x * 0.001f + i; R.x[i] = R.t[i] = R.n[i] = R.f[i] = R.s[i] = R.c[i] = R.r[i] = R.z[i]; }
#pragma unroll
  for (int i = 0; i < 4; ++i) { R.z2[i] = (f32x2){R.z[i], R.z[i + 4]}; R.x2[i] = R.t2[i] = R.f2[i] = R.c2[i] = R.r2[i] = R.z2[i]; R.h[i] = R.l[i] = i; }
  R.mx = 0.f;
  const float k1 = 1.0001f, k2 = 0.159f;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) block<MODE, MFMA, FILL, GROUPS, 0, ORDER>(R, acc, a, b, k1, k2);
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = R.mx;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += R.x[i] + R.t[i] + R.n[i] + R.f[i] + R.s[i] + R.c[i] + R.r[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += R.x2[i].x + R.t2[i].y + R.f2[i].x + R.c2[i].y + R.r2[i].x + (float)R.h[i] + (float)R.l[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) s += acc[i][0];
  if (s == 123.456f) sink[0] = s;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE, bool MFMA, bool FILL, int GROUPS, int ORDER = 0>
void run(const char* name, int threads, long long* d_out, float* d_sink) {
  const int iters = 500;
  hipLaunchKernelGGL((k<MODE, MFMA, FILL, GROUPS, ORDER>), dim3(256), dim3(threads), 0, 0, d_out, d_sink, iters);
  hipLaunchKernelGGL((k<MODE, MFMA, FILL, GROUPS, ORDER>), dim3(256), dim3(threads), 0, 0, d_out, d_sink, iters);
  hipDeviceSynchronize();
  long long h[8];
  hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-44s waves/SIMD=%d groups=%d : %7.1f cycles per 24-MFMA block (wave0), %7.1f (last wave)\n", name, threads / 256,
         GROUPS, (double)h[0] / iters, (double)h[threads / 64 - 1] / iters);
}

int main() {
  long long* d_out; float* d_sink;
  hipMalloc((void**)&d_out, 256 * 8 * 8); hipMalloc((void**)&d_sink, 64);
  for (int threads : {256, 512}) {
    run<0, true, false, 1>("mfma only", threads, d_out, d_sink);
    run<0, false, true, 1>("fill only, plain", threads, d_out, d_sink);
    run<1, false, true, 1>("fill only, packed", threads, d_out, d_sink);
    run<0, true, true, 1>("mfma + plain", threads, d_out, d_sink);
    run<1, true, true, 1>("mfma + packed", threads, d_out, d_sink);
    run<2, true, true, 1>("mfma + plain, sin/cos -> fma", threads, d_out, d_sink);
    run<3, true, true, 1>("mfma + plain, cvt -> fma", threads, d_out, d_sink);
    run<0, false, true, 2>("fill only, plain", threads, d_out, d_sink);
    run<0, true, true, 2>("mfma + plain", threads, d_out, d_sink);
    run<1, true, true, 2>("mfma + packed", threads, d_out, d_sink);
    run<0, true, false, 1, 1>("mfma only, 2 acc alternating", threads, d_out, d_sink);
    run<0, true, true, 1, 1>("mfma + plain, 2 acc alternating", threads, d_out, d_sink);
    run<0, true, false, 1, 2>("mfma only, 2 acc, 3 dependent in a row", threads, d_out, d_sink);
    run<0, true, true, 1, 2>("mfma + plain, 2 acc, 3 dependent in a row", threads, d_out, d_sink);
  }
  return 0;
}
