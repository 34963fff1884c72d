// Probe: on which SIMD of its CU does wave w of a 256-lane workgroup run?  Kernels whose workgroups keep only their
// FIRST waves busy in a phase (k_brick_resample: ~120 queries per brick = waves 0 and 1; k_brick_h: 64 + 46 lanes) load the
// SIMDs unevenly if wave w of every workgroup lands on SIMD w % 4.  Every wave records HW_REG_HW_ID (gfx9 layout: wave_id
// [3:0], simd_id [5:4], cu_id [11:8], sh_id [12], se_id [15:13]) and HW_REG_XCC_ID; the host prints the histogram of
// (wave index in workgroup) x (simd_id), and how many distinct (wave 0 .. 3) -> simd patterns occur.
// hipcc --offload-arch=gfx950 -O2 simd_map.hip -o simd_map && ./simd_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256, 4) void k(unsigned* out, int spin) {
  __shared__ float pad[9000];                 // 36 KB: four workgroups per CU, as the brick kernels
  pad[threadIdx.x] = threadIdx.x;
  __syncthreads();
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  float f = pad[threadIdx.x ^ 1];
  for (int i = 0; i < spin; ++i) f = f * 1.0001f + 0.5f;     // keep the workgroup resident while the others arrive
  if (f == 1.2345f) out[0] = 1;
  if ((threadIdx.x & 63) == 0) {
    out[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 0] = hw;
    out[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = xcc;
  }
}

int main() {
  const int blocks = 8192;
  unsigned* d; hipMalloc((void**)&d, blocks * 8 * sizeof(unsigned));
  std::vector<unsigned> h(blocks * 8);
  for (int spin : {2000, 20000}) {
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, spin);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    int hist[4][4] = {};
    int pattern[256] = {};
    int same_cu_wave0_simd[4] = {};
    for (int b = 0; b < blocks; ++b) {
      int pat = 0;
      for (int w = 0; w < 4; ++w) {
        const unsigned hw = h[2 * (b * 4 + w)];
        const int simd = (hw >> 4) & 3;
        hist[w][simd]++;
        pat |= simd << (2 * w);
      }
      pattern[pat]++;
      same_cu_wave0_simd[(h[2 * (b * 4)] >> 4) & 3]++;
    }
    printf("spin %d: wave index (rows) x simd_id (columns)\n", spin);
    for (int w = 0; w < 4; ++w) printf("  wave %d: %6d %6d %6d %6d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    printf("  patterns (simd of waves 0,1,2,3 : workgroups):");
    for (int p = 0; p < 256; ++p) if (pattern[p]) printf("  %d%d%d%d:%d", p & 3, (p >> 2) & 3, (p >> 4) & 3, (p >> 6) & 3, pattern[p]);
    printf("\n");
    // first 16 workgroups: xcc, se, cu, simd of wave 0
    for (int b = 0; b < 16; ++b) {
      const unsigned hw = h[2 * (b * 4)], xcc = h[2 * (b * 4) + 1];
      printf("  wg %2d: xcc %u se %u sh %u cu %2u | simd of waves: %u %u %u %u\n", b, xcc & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15,
             (h[2 * (b * 4)] >> 4) & 3, (h[2 * (b * 4 + 1)] >> 4) & 3, (h[2 * (b * 4 + 2)] >> 4) & 3, (h[2 * (b * 4 + 3)] >> 4) & 3);
    }
  }
  return 0;
}
