// Probe: operand/result lane layouts of v_mfma_f32_32x32x16_bf16 on gfx950 (used by siren_x3.hip).
// hipcc --offload-arch=gfx950 -O2 mfma_bf16_layout.hip -o mfma_bf16_layout && ./mfma_bf16_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const u32x4* a, const u32x4* b, float* d) {
  int lane = threadIdx.x;
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[lane]),
                                                __builtin_bit_cast(bf16x8, b[lane]), acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) d[r * 64 + lane] = acc[r];
}
static uint16_t bf(float x) { uint32_t u; memcpy(&u, &x, 4); return (uint16_t)(u >> 16); }
int main() {
  float A[32][16], B[16][32];
  for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A[i][k] = (float)((i * 7 + k * 3) % 13 - 6);
  for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) B[k][j] = (float)((k * 5 + j * 11) % 17 - 8);
  std::vector<uint16_t> ha(64 * 8), hb(64 * 8);
  for (int l = 0; l < 64; ++l) for (int e = 0; e < 8; ++e) {
    ha[l * 8 + e] = bf(A[l % 32][8 * (l / 32) + e]);
    hb[l * 8 + e] = bf(B[8 * (l / 32) + e][l % 32]);
  }
  void *da, *db; float* dd;
  hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc((void**)&dd, 4096);
  hipMemcpy(da, ha.data(), 1024, hipMemcpyHostToDevice);
  hipMemcpy(db, hb.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, (const u32x4*)da, (const u32x4*)db, dd);
  std::vector<float> hd(1024);
  hipMemcpy(hd.data(), dd, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
    int row = 8 * (r / 4) + 4 * (l / 32) + (r % 4), col = l % 32;
    float want = 0; for (int k = 0; k < 16; ++k) want += A[row][k] * B[k][col];
    if (hd[r * 64 + l] != want) ++bad;
  }
  printf("mfma_f32_32x32x16_bf16 layout hypothesis: %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);
  return bad != 0;
}
