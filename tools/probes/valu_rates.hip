// Probe: issue cost (cycles per instruction, one wave per SIMD, 8 independent chains) of the VALU
// instructions the activation stages are made of.  hipcc --offload-arch=gfx950 -O2 valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define BODY(NAME, ASM_TEXT, ...)                                                              \
  __global__ __launch_bounds__(256, 1) void k_##NAME(long long* out, float* sink, int iters) { \
    float f[8]; f32x2 p[8]; unsigned u[8];                                                     \
    for (int i = 0; i < 8; ++i) { f[i] = threadIdx.x * 0.001f + i + 1.f; p[i] = (f32x2){f[i], f[i] + 1.f}; u[i] = threadIdx.x + i; } \
    const float c1 = 1.0001f, c2 = 0.5f; (void)c1; (void)c2;                                   \
    unsigned long long sm = blockIdx.x & 1 ? 0xaaaaaaaaaaaaaaaaull : 0x5555555555555555ull; (void)sm; \
    long long t0 = __builtin_amdgcn_s_memtime();                                               \
    for (int it = 0; it < iters; ++it) {                                                       \
      _Pragma("unroll") for (int m = 0; m < 12; ++m) {                                         \
        _Pragma("unroll") for (int q = 0; q < 8; ++q) {                                        \
          const int r = q; (void)r;                                                            \
          asm volatile(ASM_TEXT : __VA_ARGS__);                                                \
        }                                                                                      \
      }                                                                                        \
    }                                                                                          \
    long long t1 = __builtin_amdgcn_s_memtime();                                               \
    float s = 0; for (int i = 0; i < 8; ++i) s += f[i] + p[i].x + p[i].y + (float)u[i];        \
    if (s == 123.456f) sink[0] = s;                                                            \
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;          \
  }

BODY(fma, "v_fma_f32 %0, %0, %1, %2", "+v"(f[r]) : "v"(c1), "v"(c2))
BODY(mul, "v_mul_f32 %0, %0, %1", "+v"(f[r]) : "v"(c1))
BODY(pk_fma, "v_pk_fma_f32 %0, %0, %1, %1", "+v"(p[r]) : "v"(p[(r + 4) & 7]))
BODY(pk_mul, "v_pk_mul_f32 %0, %0, %1", "+v"(p[r]) : "v"(p[(r + 4) & 7]))
BODY(pk_add, "v_pk_add_f32 %0, %0, %1", "+v"(p[r]) : "v"(p[(r + 4) & 7]))
BODY(cvt_pk_bf16, "v_cvt_pk_bf16_f32 %0, %1, %2", "=v"(u[r]) : "v"(f[r]), "v"(f[(r + 1) & 7]))
BODY(rndne, "v_rndne_f32 %0, %0", "+v"(f[r]) :)
BODY(and_b32, "v_and_b32 %0, 0xffff0000, %0", "+v"(u[r]) :)
BODY(lshl, "v_lshlrev_b32 %0, 16, %0", "+v"(u[r]) :)
BODY(perm, "v_perm_b32 %0, %0, %1, %2", "+v"(u[r]) : "v"(u[(r + 1) & 7]), "v"(u[(r + 2) & 7]))
BODY(max3, "v_max3_f32 %0, %0, %1, %2", "+v"(f[r]) : "v"(c1), "v"(c2))
BODY(cndmask_vcc, "v_cndmask_b32 %0, %0, %1, vcc", "+v"(f[r]) : "v"(c1))
BODY(cndmask_sgpr, "v_cndmask_b32_e64 %0, %0, %1, %2", "+v"(f[r]) : "v"(c1), "s"(sm))
BODY(cmp, "v_cmp_gt_f32 vcc, %0, %1", : "v"(f[r]), "v"(c1) : "vcc")
BODY(cmp_cnd, "v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc", "+v"(f[r]) : "v"(c1) : "vcc")
BODY(exp, "v_exp_f32 %0, %0", "+v"(f[r]) :)
BODY(log, "v_log_f32 %0, %0", "+v"(f[r]) :)
BODY(rcp, "v_rcp_f32 %0, %0", "+v"(f[r]) :)
BODY(sin, "v_sin_f32 %0, %0", "+v"(f[r]) :)
BODY(add3, "v_add3_u32 %0, %0, %1, %2", "+v"(u[r]) : "v"(u[(r + 1) & 7]), "v"(u[(r + 2) & 7]))
BODY(bfe, "v_bfe_u32 %0, %0, 16, 1", "+v"(u[r]) :)
BODY(cvt_i32, "v_cvt_i32_f32 %0, %1", "=v"(u[r]) : "v"(f[r]))
BODY(med3, "v_med3_f32 %0, %0, %1, %2", "+v"(f[r]) : "v"(c1), "v"(c2))
BODY(ldexp, "v_ldexp_f32 %0, %0, %1", "+v"(f[r]) : "v"(u[(r + 1) & 7]))
BODY(mov, "v_mov_b32 %0, %1", "=v"(f[r]) : "v"(c1))
BODY(mov_self, "v_mov_b32 %0, %1", "=v"(f[r]) : "v"(f[(r + 1) & 7]))
BODY(cmp_u64, "v_cmp_lt_u64 vcc, %0, %1", : "v"(p[r]), "v"(p[(r + 1) & 7]) : "vcc")
BODY(cmp_u64_s, "v_cmp_lt_u64 %0, %1, %2", "=s"(sm) : "v"(p[r]), "v"(p[(r + 1) & 7]))
BODY(cmp_f32_s, "v_cmp_lt_f32 %0, %1, %2", "=s"(sm) : "v"(f[r]), "v"(f[(r + 1) & 7]))
BODY(cmp_u32, "v_cmp_lt_u32 vcc, %0, %1", : "v"(u[r]), "v"(u[(r + 1) & 7]) : "vcc")
BODY(min_f32, "v_min_f32 %0, %0, %1", "+v"(f[r]) : "v"(c1))
BODY(min_u32, "v_min_u32 %0, %0, %1", "+v"(u[r]) : "v"(u[(r + 1) & 7]))
BODY(swap, "v_swap_b32 %0, %1", "+v"(f[r]), "+v"(f[(r + 4) & 7]) :)
BODY(mov_b64, "v_mov_b64 %0, %1", "=v"(p[r]) : "v"(p[(r + 1) & 7]))

template <class K> void run(const char* name, K kern, long long* d_out, float* d_sink) {
  const int iters = 2000;
  hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, d_out, d_sink, iters);
  hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, d_out, d_sink, iters);
  hipDeviceSynchronize();
  long long h[8];
  hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-16s %.2f cycles per instruction (s_memtime ticks / %d)\n", name, (double)h[0] / (iters * 96.0), 96 * iters);
}

int main() {
  long long* d_out; float* d_sink;
  hipMalloc((void**)&d_out, 256 * 8 * 8); hipMalloc((void**)&d_sink, 64);
#define R(N) run(#N, k_##N, d_out, d_sink);
  R(fma) R(mul) R(pk_fma) R(pk_mul) R(pk_add) R(cvt_pk_bf16) R(rndne) R(and_b32) R(lshl) R(perm) R(max3)
  R(cndmask_vcc) R(cndmask_sgpr) R(cmp) R(cmp_cnd) R(exp) R(log) R(rcp) R(sin) R(add3) R(bfe) R(cvt_i32) R(med3) R(ldexp) R(mov) R(mov_self) R(cmp_u64) R(cmp_u64_s) R(cmp_f32_s) R(cmp_u32) R(min_f32) R(min_u32) R(swap) R(mov_b64)
  return 0;
}
