// Probe: is a register that an issued-but-possibly-queued MFMA reads as SrcB protected against a
// following ds_read that overwrites it?  One asm block per trip:
//     v_mfma  acc0 += A * Bv        (Bv holds pattern X)
//     [D filler MFMAs on other accumulators that do not touch Bv]
//     ds_read_b128 Bv <- pattern Y  (write-after-read, D MFMAs after the reader)
//     [8 more filler MFMAs] ; s_waitcnt ; ds_read_b128 Bv <- pattern X ; s_waitcnt
// acc0 must equal trips * (A.X).  Run with 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int D, bool HOG>
__global__ __launch_bounds__(512, 1) void k(float* out, int trips) {
  __shared__ u32x4 lds[2][64];
  const int lane = threadIdx.x & 63;
  if (threadIdx.x < 64) {
    const unsigned one = 0x3f803f80u;   // bf16 1.0, 1.0
    const unsigned two = 0x40004000u;   // bf16 2.0, 2.0
    lds[0][lane] = (u32x4){one, one, one, one};
    lds[1][lane] = (u32x4){two, two, two, two};
  }
  __syncthreads();
  u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  f32x16 acc0 = {}, f1 = {}, f2 = {}, f3 = {};
  u32x4 bv = lds[0][lane];
  u32x4 other = lds[0][lane];
  const unsigned ax = (unsigned)(size_t)&lds[0][lane], ay = (unsigned)(size_t)&lds[1][lane];
  if (HOG && (threadIdx.x >> 6) >= 4) {
    // the co-resident wave of every SIMD only streams MFMAs (an independent GEMM phase)
    for (int t = 0; t < trips * 12; ++t) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n" : "+v"(f1) : "v"(a), "v"(other));
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n" : "+v"(f2) : "v"(a), "v"(other));
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n" : "+v"(f3) : "v"(a), "v"(other));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = trips * 16.0f + 1e-30f * (f1[0] + f2[0] + f3[0]);
    return;
  }
  for (int t = 0; t < trips; ++t) {
    asm volatile("s_waitcnt lgkmcnt(0)\n v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n" : "+v"(acc0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(a), "v"(bv));
#pragma unroll
    for (int d = 0; d < D; ++d)
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n" : "+v"(f1) : "v"(a), "v"(other));
    asm volatile("ds_read_b128 %0, %1\n" : "=v"(bv) : "v"(ay) : "memory");
#pragma unroll
    for (int d = 0; d < 8; ++d)
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n" : "+v"(f2) : "v"(a), "v"(other));
    asm volatile("s_waitcnt lgkmcnt(0)\n ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)\n" : "=v"(bv) : "v"(ax) : "memory");
  }
  float s = acc0[0] + 1e-30f * (f1[0] + f2[0] + f3[0]);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int D, bool HOG>
void run(int threads) {
  const int trips = 2000, blocks = 256;
  float* d; hipMalloc((void**)&d, blocks * threads * 4);
  hipLaunchKernelGGL((k<D, HOG>), dim3(blocks), dim3(threads), 0, 0, d, trips);
  hipDeviceSynchronize();
  std::vector<float> h(blocks * threads);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  const float want = trips * 16.0f;      // 16 k-values of 1*1 per trip
  long bad = 0; float worst = want;
  for (float v : h) if (v != want) { ++bad; if (fabsf(v - want) > fabsf(worst - want)) worst = v; }
  printf("%s waves/SIMD=%d  reader->load distance %d MFMAs: %ld of %zu lanes wrong (expected %.0f, worst %.0f)\n",
         HOG ? "partner streams MFMAs:" : "partner runs the same code:", threads / 256, D, bad, h.size(), want, worst);
  hipFree(d);
}

int main() {
  run<1, false>(256); run<0, false>(256); run<0, false>(512); run<1, false>(512);
  run<0, true>(512); run<1, true>(512); run<2, true>(512); run<4, true>(512); run<8, true>(512);
  return 0;
}
