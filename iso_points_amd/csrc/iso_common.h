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
