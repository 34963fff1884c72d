// Probe: how many VALU instructions of which kind hide behind a v_mfma_f32_32x32x16_bf16 issued
// by the SAME wave (one wave per SIMD), and what two co-resident waves do to each other.
// hipcc --offload-arch=gfx950 -O2 mfma_filler.hip -o mfma_filler && ./mfma_filler
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// KIND: 0 none, 1 v_fma_f32 (independent chains), 2 v_pk_fma_f32, 3 v_cndmask, 4 v_cvt_pk_bf16_f32, 5 v_xor/v_and int, 6 v_fma dependent chain
template <int KIND, int NFILL, bool MFMA>
__global__ __launch_bounds__(512, 1) void k(long long* out, float* sink, int iters) {
  f32x16 acc[3] = {};
  bf16x8 a = {}, b = {};
  float f[8];
  f32x2 p[8];
  unsigned u[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { f[i] = threadIdx.x * 0.001f + i; p[i] = (f32x2){f[i], f[i] + 1.f}; u[i] = threadIdx.x + i; }
  const float c1 = 1.0001f, c2 = 0.5f;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 12; ++m) {
      if (MFMA) acc[m % 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m % 3], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NFILL; ++q) {
        const int r = (m * NFILL + q) & 7;
        if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[r]) : "v"(c1), "v"(c2));
        if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[r]) : "v"(p[(r + 4) & 7]));
        if (KIND == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[r]) : "v"(c1));
        if (KIND == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[r]) : "v"(f[r]), "v"(f[(r + 1) & 7]));
        if (KIND == 5) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[r]) : "v"(u[(r + 3) & 7]));
        if (KIND == 6) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[0]) : "v"(c1), "v"(c2));
        if (KIND == 7) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[(m * NFILL + q) % 6]) : "v"(c1), "v"(c2));
        if (KIND == 8) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[(m * NFILL + q) % 3]) : "v"(c1), "v"(c2));
      }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += f[i] + p[i].x + p[i].y + (float)u[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) s += acc[i][0];
  if (s == 123.456f) sink[0] = s;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int NFILL, bool MFMA>
void run(const char* name, int threads, long long* d_out, float* d_sink) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<KIND, NFILL, MFMA>), dim3(256), dim3(threads), 0, 0, d_out, d_sink, iters);
  hipLaunchKernelGGL((k<KIND, NFILL, MFMA>), dim3(256), dim3(threads), 0, 0, d_out, d_sink, iters);
  hipDeviceSynchronize();
  long long h[8];
  hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-34s waves/SIMD=%d fill=%d : %.1f cycles per MFMA slot (wave0), %.1f (last wave)\n", name, threads / 256,
         NFILL, (double)h[0] / (iters * 12.0), (double)h[threads / 64 - 1] / (iters * 12.0));
}

int main() {
  long long* d_out; float* d_sink;
  hipMalloc((void**)&d_out, 256 * 8 * 8); hipMalloc((void**)&d_sink, 64);
  run<0, 0, true>("mfma only", 256, d_out, d_sink);
  run<0, 0, true>("mfma only", 512, d_out, d_sink);
#define ROW(K, NAME) \
  run<K, 2, true>(NAME, 256, d_out, d_sink); run<K, 4, true>(NAME, 256, d_out, d_sink); \
  run<K, 6, true>(NAME, 256, d_out, d_sink); run<K, 8, true>(NAME, 256, d_out, d_sink); \
  run<K, 8, false>(NAME " (no mfma)", 256, d_out, d_sink);
  ROW(1, "v_fma_f32 independent")
  ROW(6, "v_fma_f32 dependent chain")
  ROW(2, "v_pk_fma_f32")
  ROW(3, "v_cndmask_b32")
  ROW(4, "v_cvt_pk_bf16_f32")
  ROW(5, "v_xor_b32")
  ROW(7, "v_fma_f32, 6 chains")
  ROW(8, "v_fma_f32, 3 chains")
  run<7, 7, true>("v_fma_f32, 6 chains", 256, d_out, d_sink);
  run<1, 7, true>("v_fma_f32 independent", 256, d_out, d_sink);
  run<1, 5, true>("v_fma_f32 independent", 256, d_out, d_sink);
  return 0;
}
