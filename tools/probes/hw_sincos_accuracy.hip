// Probe: accuracy of v_sin_f32 / v_cos_f32 (argument in revolutions) on [-0.5, 0.5] against double.
// hipcc --offload-arch=gfx950 -O2 hw_sincos_accuracy.hip -o hw_sincos_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float* x, float* s, float* c, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { s[i] = __builtin_amdgcn_sinf(x[i]); c[i] = __builtin_amdgcn_cosf(x[i]); }
}
int main() {
  const int n = 1 << 22;
  std::vector<float> x(n), s(n), c(n);
  for (int i = 0; i < n; ++i) x[i] = -0.5f + (float)i / (float)n;
  float *dx, *ds, *dc;
  hipMalloc((void**)&dx, n * 4); hipMalloc((void**)&ds, n * 4); hipMalloc((void**)&dc, n * 4);
  hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, ds, dc, n);
  hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
  double es = 0, ec = 0, ms = 0, mc = 0;
  for (int i = 0; i < n; ++i) {
    const double t = 2.0 * M_PI * (double)x[i];
    const double a = fabs((double)s[i] - sin(t)), b = fabs((double)c[i] - cos(t));
    es = fmax(es, a); ec = fmax(ec, b); ms += a; mc += b;
  }
  printf("v_sin_f32: max abs err %.3g mean %.3g | v_cos_f32: max %.3g mean %.3g (f32 half-ulp at 1: 6e-8)\n", es, ms / n, ec, mc / n);
  return 0;
}
