// Shared host/device helpers for the iso-points MI355X (gfx950) library.
// Everything here is private to csrc/; the public surface is include/isopoints.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/isopoints.h"

// ---- error plumbing -------------------------------------------------------
void iso_set_error(const char* fmt, ...);

#define ISO_REQUIRE(cond, code, ...)            \
  do {                                          \
    if (!(cond)) {                              \
      iso_set_error(__VA_ARGS__);               \
      return (code);                            \
    }                                           \
  } while (0)

#define ISO_CHECK_LAUNCH(name)                                          \
  do {                                                                  \
    hipError_t e__ = hipGetLastError();                                 \
    if (e__ != hipSuccess) {                                            \
      iso_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return ISO_ERR_LAUNCH;                                            \
    }                                                                   \
  } while (0)

static inline int iso_div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Grid for an HBM-bound element-wise kernel: enough workgroups to fill 256 CUs
// several times over, grid-stride the rest (guide: cap ~2048 blocks).
static inline int iso_stream_grid(int64_t n, int block) {
  int64_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > 256 * 16) g = 256 * 16;
  return (int)g;
}

// ---- device helpers -------------------------------------------------------
#define ISO_WAVE 64

// Sign-preserving clamp of |x| to >= eps (reference: DSS/utils/mathHelper.py:14-18).
__device__ __forceinline__ float iso_eps_denom(float x, float eps) {
  float s = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 1.f);  // sign(x) + (x==0)
  float a = fabsf(x);
  return s * (a < eps ? eps : a);
}

// Append the flagged ones of 4 items per thread (256-thread workgroup, all threads call) to a list whose length is
// *counter: ONE returning atomic per call.  Same-address returning atomics retire a few ns apart whatever issues
// them, so one per wave makes the atomics the whole duration of a compaction kernel at a few thousand waves.
// slot[k] = position of item k (item k of thread t is element base + k * 256 + t of the round), -1 if not taken.
// smem: 17 ints of LDS, free again after the call returns on every thread's next barrier.
__device__ __forceinline__ void iso_block_append4(const bool (&take)[4], int32_t* counter, int* smem, int (&slot)[4]) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned long long bal[4];
  __syncthreads();                                  // the previous round's readers of smem are done
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    bal[k] = __ballot(take[k]);
    if (lane == 0) smem[k * 4 + wv] = __popcll(bal[k]);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) tot += smem[q];
    smem[16] = tot ? atomicAdd(counter, tot) : 0;
  }
  __syncthreads();
  int at = smem[16];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    slot[k] = -1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q == wv && take[k]) slot[k] = at + __popcll(bal[k] & ((1ull << lane) - 1ull));
      at += smem[k * 4 + q];
    }
  }
}

__device__ __forceinline__ float iso_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Zero a few words with a kernel instead of hipMemsetAsync: memset nodes of a few bytes captured into a
// HIP graph were observed to leave garbage on replay (ROCm 7.2), a kernel node is exact.
static __global__ void iso_k_zero_words(uint32_t* p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0u;
}
static inline void iso_zero_words(void* p, int64_t n_words, hipStream_t s) {
  int64_t g = (n_words + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(iso_k_zero_words, dim3((int)g), dim3(256), 0, s, (uint32_t*)p, n_words);
}
