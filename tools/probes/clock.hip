// Probe: the shader clock a kernel actually runs at.  A wave issues N dependent-free v_mfma_f32_32x32x16_f16
// (32 cycles each on its SIMD, one wave per SIMD); cycles / wall time (HIP events) = clock.  Run with 8 and with
// 256 workgroups, and with a VALU-only loop for comparison.
// hipcc --offload-arch=gfx950 -O2 clock.hip -o clock && ./clock
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool MFMA>
__global__ __launch_bounds__(256, 1) void k(float* sink, int iters) {
  f32x16 acc[3] = {};
  f16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) {       // non-trivial operands: the datapath toggles as in a real GEMM
    a[i] = (_Float16)(0.37f + 0.011f * ((threadIdx.x * 7 + i * 3) % 13));
    b[i] = (_Float16)(0.0021f * ((threadIdx.x * 5 + i) % 11) - 0.01f);
  }
  float f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 24; ++m) {
      if (MFMA) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[m % 3]) : "v"(a), "v"(b));
      else {
#pragma unroll
        for (int q = 0; q < 8; ++q) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[q]) : "v"(f[(q + 3) & 7]));
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) s += acc[i][0];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += f[i];
  if (s == 123.456f) sink[0] = s;
}

template <bool MFMA>
void run(const char* name, int blocks, float* sink) {
  const int iters = 200000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MFMA>, dim3(blocks), dim3(256), 0, 0, sink, 1000);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MFMA>, dim3(blocks), dim3(256), 0, 0, sink, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double cycles = (double)iters * 24 * 32;       // per SIMD: 24 MFMAs of 32 cycles, or 24 x 8 VALU ops of 4 cycles
  printf("%-10s %3d workgroups: %.1f ms for %.3g cycles per SIMD -> %.2f GHz\n", name, blocks, ms, cycles, cycles / (ms * 1e6));
}

int main() {
  float* sink; hipMalloc((void**)&sink, 64);
  for (int rep = 0; rep < 2; ++rep) {
    run<true>("mfma", 8, sink);
    run<true>("mfma", 256, sink);
    run<false>("valu", 8, sink);
    run<false>("valu", 256, sink);
  }
  return 0;
}
