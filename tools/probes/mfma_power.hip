// Probe: what dense v_mfma_f32_32x32x16_f16 sustains on the whole chip as a function of the OPERAND DATA.
// tools/probes/clock.hip multiplies the same two (smooth) operand registers over and over: 1.66 GHz on 256 workgroups.  A
// real GEMM feeds new fragments to every MFMA, and the split-fp16 SIREN step feeds it noise-like low-order halves.
// Here a wave cycles through NSET different A and B register sets per MFMA:
//   mode 0: the clock probe's operands (one smooth set, never changing)
//   mode 1: NSET smooth sets (values of a trained-network scale, few mantissa bits set), changing every MFMA
//   mode 2: NSET sets of random fp16 numbers (full mantissas, exponents spread over 2^-6 .. 2^9), changing every MFMA
//   mode 3: all-zero operands
// Reports the rate as the clock the 32-cycle MFMAs would imply (cycles / wall time), for 8 and 256 workgroups, one and
// two waves per SIMD.
// hipcc --offload-arch=gfx950 -O2 mfma_power.hip -o mfma_power && ./mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int NSET = 8;

__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int MODE>
__global__ __launch_bounds__(512, 1) void k(float* sink, int iters) {
  f32x16 acc[3] = {};
  f16x8 a[NSET], b[NSET];
#pragma unroll
  for (int s = 0; s < NSET; ++s) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0 || MODE == 1) {
        const int ss = MODE == 0 ? 0 : s;
        a[s][i] = (_Float16)(0.37f + 0.011f * ((threadIdx.x * 7 + i * 3 + ss * 5) % 13));
        b[s][i] = (_Float16)(0.0021f * ((threadIdx.x * 5 + i + ss * 3) % 11) - 0.01f);
      } else if (MODE == 2) {
        const unsigned ha = hash(threadIdx.x * 131u + i * 17u + s * 1009u + blockIdx.x * 7919u), hb = hash(ha + 0x9e3779b9u);
        // sign | exponent 9..24 (2^-6 .. 2^9) | 10 random mantissa bits
        const unsigned short ua = (unsigned short)(((ha >> 31) << 15) | ((9u + ((ha >> 10) & 15u)) << 10) | (ha & 1023u));
        const unsigned short ub = (unsigned short)(((hb >> 31) << 15) | ((9u + ((hb >> 10) & 15u)) << 10) | (hb & 1023u));
        a[s][i] = __builtin_bit_cast(_Float16, ua);
        b[s][i] = __builtin_bit_cast(_Float16, ub);
      } else {
        a[s][i] = (_Float16)0.f; b[s][i] = (_Float16)0.f;
      }
    }
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 24; ++m)
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[m % 3]) : "v"(a[m % NSET]), "v"(b[(m * 3) % NSET]));
    if (MODE == 2 && (it & 1023) == 1023) {           // keep the accumulators finite
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = acc[q][r] * 1.0e-30f;
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) s += acc[i][0];
  if (s == 123.456f) sink[0] = s;
}

template <int MODE>
void run(const char* name, int blocks, int threads, float* sink) {
  const int iters = 100000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, sink, 1000);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, sink, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double waves_per_simd = threads / 256.0;
  const double cycles = (double)iters * 24 * 32 * waves_per_simd;       // matrix-pipe cycles per SIMD
  const double tflops = (double)blocks * 4 * cycles / 32 * 32768.0 / (ms * 1e-3) / 1e12;
  printf("%-34s %3d workgroups x %d waves/SIMD: %7.1f ms -> %.2f GHz-equivalent, %6.0f TFLOP/s fp16 (%.0f f32-equivalent at 3 passes)\n",
         name, blocks, threads / 256, ms, cycles / (ms * 1e6), tflops, tflops / 3);
}

// mfma_power long MODE SECONDS: 256 workgroups x 1 wave/SIMD of mode MODE back to back for SECONDS (for tools/power_probe.py)
template <int MODE>
void run_long(const char* name, float seconds, float* sink) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, sink, 1000);
  hipDeviceSynchronize();
  const int iters = 100000;
  double total = 0; int n = 0;
  while (total < seconds * 1e3) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, sink, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    total += ms; ++n;
  }
  const double cycles = (double)iters * 24 * 32;
  printf("%s: %d launches, %.1f ms each -> %.2f GHz-equivalent, %.0f TFLOP/s fp16\n", name, n, total / n, cycles / (total / n * 1e6),
         256.0 * 4 * cycles / 32 * 32768.0 / (total / n * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
  float* sink; hipMalloc((void**)&sink, 64);
  if (argc >= 4 && argv[1][0] == 'l') {
    const int mode = atoi(argv[2]); const float sec = atof(argv[3]);
    if (mode == 0) run_long<0>("one smooth set", sec, sink);
    else if (mode == 1) run_long<1>("8 smooth sets", sec, sink);
    else if (mode == 2) run_long<2>("8 random sets", sec, sink);
    else run_long<3>("zeros", sec, sink);
    return 0;
  }
  for (int threads : {256, 512}) {
    for (int blocks : {8, 256}) {
      run<3>("zeros", blocks, threads, sink);
      run<0>("one smooth set (clock.hip)", blocks, threads, sink);
      run<1>("8 smooth sets, changing", blocks, threads, sink);
      run<2>("8 random sets, changing", blocks, threads, sink);
    }
  }
  return 0;
}
