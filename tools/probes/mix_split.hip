#include <hip/hip_runtime.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, unsigned* out, float sc) {
  f32x2 x = {in[threadIdx.x * 2] * sc, in[threadIdx.x * 2 + 1] * sc};
  unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(x, f16x2));
  unsigned l;
  asm("v_fma_mixlo_f16 %0, 1.0, %1, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x[0]), "v"(h));
  asm("v_fma_mixhi_f16 %0, 1.0, %1, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x[1]), "v"(h));
  out[threadIdx.x * 2] = h;
  out[threadIdx.x * 2 + 1] = l;
}
__global__ void kref(const float* in, unsigned* out, float sc) {
  f32x2 x = {in[threadIdx.x * 2] * sc, in[threadIdx.x * 2 + 1] * sc};
  f16x2 h = __builtin_convertvector(x, f16x2);
  f32x2 f = __builtin_convertvector(h, f32x2);
  f16x2 l = __builtin_convertvector(x - f, f16x2);
  out[threadIdx.x * 2] = __builtin_bit_cast(unsigned, h);
  out[threadIdx.x * 2 + 1] = __builtin_bit_cast(unsigned, l);
}
int main() {
  const int N = 1 << 20;
  float* in; unsigned *o1, *o2;
  hipMalloc(&in, N * 4); hipMalloc(&o1, N * 4); hipMalloc(&o2, N * 4);
  float* hin = new float[N];
  unsigned s = 12345;
  for (int i = 0; i < N; ++i) { s = s * 1664525u + 1013904223u; unsigned b = (s & 0x807fffffu) | ((100u + (s >> 23) % 40u) << 23); hin[i] = *(float*)&b; if (i % 97 == 0) hin[i] = 0.f; }
  hipMemcpy(in, hin, N * 4, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 256; ++rep) {
    hipLaunchKernelGGL(k, dim3(N / 512), dim3(256), 0, 0, in, o1, 4096.0f);
    hipLaunchKernelGGL(kref, dim3(N / 512), dim3(256), 0, 0, in, o2, 4096.0f);
  }
  unsigned *h1 = new unsigned[N], *h2 = new unsigned[N];
  hipMemcpy(h1, o1, N * 4, hipMemcpyDeviceToHost); hipMemcpy(h2, o2, N * 4, hipMemcpyDeviceToHost);
  long bad = 0;
  for (int i = 0; i < N; ++i) if (h1[i] != h2[i]) { if (bad < 5) printf("diff %d: %08x %08x in %g %g\n", i, h1[i], h2[i], hin[i & ~1], hin[i | 1]); ++bad; }
  printf("mismatches: %ld of %d\n", bad, N);
  return 0;
}
