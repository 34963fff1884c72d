// Probe: v_swap_b32 under a partial EXEC mask, with the mask restored right behind it.
//
// The failing builds of the round-5 "spilled registers" case (tools/probes/spill_kit) keep part of a pixel's K-best list
// in memory and implement the (z, id) swap chain of PixK::push with `v_swap_b32 id[j], ci` inside an
// `s_and_saveexec_b64 ... s_or_b64 exec, exec, ...` region that only the lanes with a depth TIE enter; what comes out
// wrong is exactly that swap on a tie: the slot keeps its old id while the carried id moves on.  v_swap_b32 writes BOTH its
// operands; this probe checks whether the second write (or either) can be lost / executed under the wrong mask when the
// region is a few instructions long.  Every lane swaps (a, b) when its bit of a wave-uniform mask is set, in the exact
// instruction pattern of the failing ISA, thousands of times with changing masks, several waves per SIMD; the result is
// compared with a v_cndmask reference.
//   hipcc --offload-arch=gfx950 -O2 vswap_exec.hip -o vswap_exec && ./vswap_exec
#include <hip/hip_runtime.h>
#include <cstdio>

// MODE 0: saveexec, v_swap, restore            1: saveexec, v_mov fillers, v_swap, v_mov_b64, restore (the failing ISA's shape)
//      2: the mask computed by v_cmp_* right in front (VALU writes SGPR -> s_and_saveexec reads it)
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned* bad, int iters, unsigned seed) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned a = tid * 2654435761u + seed, b = ~a + 12345u;
  unsigned errs = 0;
  for (int it = 0; it < iters; ++it) {
    const unsigned h = (a ^ (b >> 3) ^ (unsigned)it * 40503u);
    const bool take = (h % 7u) == 0u || ((unsigned)it & 15u) == 0u && (h & 1u);      // ~15-20 % of the lanes, changing
    const unsigned ea = take ? b : a, eb = take ? a : b;                                // what the swap must give
    unsigned long long sv;
    unsigned f0 = a + 1, f1 = b + 2;
    unsigned long long f2 = ((unsigned long long)a << 32) | b, f3 = 0;
    if (MODE == 0) {
      const unsigned long long m = __ballot(take);
      asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                   "v_swap_b32 %[a], %[b]\n\t"
                   "s_or_b64 exec, exec, %[sv]"
                   : [a] "+v"(a), [b] "+v"(b), [sv] "=&s"(sv) : [m] "s"(m) : "scc");
    } else if (MODE == 1) {
      const unsigned long long m = __ballot(take);
      asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                   "v_mov_b32 %[f0], %[b]\n\t"
                   "v_mov_b32 %[f1], %[a]\n\t"
                   "v_swap_b32 %[a], %[b]\n\t"
                   "v_mov_b64 %[f3], %[f2]\n\t"
                   "s_or_b64 exec, exec, %[sv]"
                   : [a] "+v"(a), [b] "+v"(b), [sv] "=&s"(sv), [f0] "+v"(f0), [f1] "+v"(f1), [f3] "+v"(f3) : [m] "s"(m), [f2] "v"(f2) : "scc");
    } else {
      const unsigned tk = take ? 1u : 0u;
      unsigned long long m;
      asm volatile("v_cmp_ne_u32_e64 %[m], 0, %[tk]\n\t"
                   "s_and_saveexec_b64 %[sv], %[m]\n\t"
                   "v_mov_b32 %[f0], %[b]\n\t"
                   "v_swap_b32 %[a], %[b]\n\t"
                   "v_mov_b64 %[f3], %[f2]\n\t"
                   "s_or_b64 exec, exec, %[sv]"
                   : [a] "+v"(a), [b] "+v"(b), [sv] "=&s"(sv), [m] "=&s"(m), [f0] "+v"(f0), [f3] "+v"(f3) : [tk] "v"(tk), [f2] "v"(f2) : "scc");
    }
    errs += (a != ea) + (b != eb);
    // the 64-bit move in front of the mask's restore (v_mov_b64 is a two-pass instruction): BOTH halves of its destination
    // must be written in the lanes of the mask and NEITHER outside it
    if (MODE >= 1) errs += take ? (f3 != f2) : (f3 != 0ull);
    a = ea * 1664525u + 1013904223u + (f0 & 1u);      // keep going from the CORRECT values (errors do not cascade)
    b = eb ^ (a >> 7) ^ (unsigned)(f3 & 1ull) ^ (f1 & 2u);
  }
  if (errs) atomicAdd(bad, errs);
}

template <int MODE>
void run(const char* what, unsigned* bad) {
  unsigned tot = 0, h;
  for (int r = 0; r < 10; ++r) {
    hipMemset(bad, 0, 4);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * 8), dim3(256), 0, 0, bad, 4000, 77u * r + 5u);
    hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    tot += h;
  }
  printf("%-78s wrong results: %u of %.3g lane-swaps\n", what, tot, 10.0 * 256 * 8 * 256 * 4000);
}

int main() {
  unsigned* bad; hipMalloc((void**)&bad, 64);
  run<0>("s_and_saveexec / v_swap_b32 / s_or exec", bad);
  run<1>("s_and_saveexec / v_mov x2 / v_swap_b32 / v_mov_b64 / s_or exec (the failing ISA's shape)", bad);
  run<2>("v_cmp -> s_and_saveexec / v_mov / v_swap_b32 / v_mov_b64 / s_or exec", bad);
  return 0;
}
