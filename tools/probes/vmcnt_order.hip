// Probe: does vmcnt on gfx950 retire in ISSUE ORDER when a wave mixes loads and stores?
//
// LLVM's gfx9 waitcnt model (no separate vscnt) treats every vector-memory operation as one in-order stream: after
// [load A, load B, load C, store S] an `s_waitcnt vmcnt(1)` is taken to mean "A, B and C have returned".  If the store (or a
// later load that hits a nearer cache) can retire BEFORE an older, slower load, that wait passes early and the wave reads C's
// destination register before the data is there -- a stale value, different from run to run.  That is what round 5's
// k_raster merge variant showed (stale ids read from other slices' lists: twelve sc1 loads in flight, scratch stores of a
// memory-resident pair of list entries issued among them, partial vmcnt waits), and what the round-1 MFMA kernels with
// spills + pinned loads showed.
//
// Each lane issues three slow loads (cold lines of a 2 GB buffer, optionally sc1 = agent scope), then ONE more operation
// X, then `s_waitcnt vmcnt(1)`, and snapshots the third load's destination.  In-order retirement => the snapshot always
// holds the loaded value.  X is: a plain global store / an sc1 global store / a scratch store / a load of a hot line.
//   hipcc --offload-arch=gfx950 -O2 vmcnt_order.hip -o vmcnt_order && ./vmcnt_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr unsigned kSentinel = 0xdeadbeefu;
__host__ __device__ inline unsigned value_at(uint64_t i) { return (unsigned)(i * 2654435761u) ^ 0x5a5a5a5au; }

__global__ void k_fill(unsigned* buf, uint64_t n) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) buf[i] = value_at(i);
}

// MODE 0: X = global_store_dword (plain)   1: X = global_store_dword sc1   2: X = scratch_store_dword
//      3: X = global_load_dword of a hot line (plain)    4: X = scratch_load_dword
// SC1: the three slow loads carry sc1
template <int MODE, bool SC1>
__global__ __launch_bounds__(256) void k_probe(const unsigned* __restrict__ buf, uint64_t n_lines, unsigned* __restrict__ sink,
                                               const unsigned* __restrict__ hot, unsigned* __restrict__ bad, unsigned salt) {
  volatile unsigned priv[4];                       // gives the kernel a private segment (MODE 2 / 4 use its first dword)
  priv[0] = threadIdx.x;
  const uint64_t gid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  // three lines far apart, different for every lane and every launch (cold)
  const uint64_t l0 = (gid * 7919u + salt * 104729ull) % n_lines, l1 = (l0 + n_lines / 3 + 17) % n_lines, l2 = (l0 + 2 * (n_lines / 3) + 39) % n_lines;
  const unsigned* a0 = buf + l0 * 32, *a1 = buf + l1 * 32, *a2 = buf + l2 * 32;
  unsigned* st = sink + gid;
  const unsigned* ht = hot + (threadIdx.x & 63);
  unsigned hv = *ht;                               // warm the hot line
  unsigned d0, d1, d2, snap, x = hv;
#define LOADS(SUF)                                            \
    "v_mov_b32 %0, %9\n v_mov_b32 %1, %9\n v_mov_b32 %2, %9\n"  \
    "s_waitcnt vmcnt(0) lgkmcnt(0)\n"                          \
    "global_load_dword %0, %5, off" SUF "\n"                  \
    "global_load_dword %1, %6, off" SUF "\n"                  \
    "global_load_dword %2, %7, off" SUF "\n"
#define TAIL                                                  \
    "s_waitcnt vmcnt(1)\n"                                    \
    "v_mov_b32 %3, %2\n"                                      \
    "s_waitcnt vmcnt(0)\n"
#define OPS : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(snap), "+v"(x) : "v"(a0), "v"(a1), "v"(a2), "v"(st), "v"(kSentinel), "v"(ht) : "memory"
  if (MODE == 0) { if (SC1) asm volatile(LOADS(" sc1") "global_store_dword %8, %4, off\n" TAIL OPS); else asm volatile(LOADS("") "global_store_dword %8, %4, off\n" TAIL OPS); }
  if (MODE == 1) { if (SC1) asm volatile(LOADS(" sc1") "global_store_dword %8, %4, off sc1\n" TAIL OPS); else asm volatile(LOADS("") "global_store_dword %8, %4, off sc1\n" TAIL OPS); }
  if (MODE == 2) { if (SC1) asm volatile(LOADS(" sc1") "scratch_store_dword off, %4, off\n" TAIL OPS); else asm volatile(LOADS("") "scratch_store_dword off, %4, off\n" TAIL OPS); }
  if (MODE == 3) { if (SC1) asm volatile(LOADS(" sc1") "global_load_dword %4, %10, off\n" TAIL OPS); else asm volatile(LOADS("") "global_load_dword %4, %10, off\n" TAIL OPS); }
  if (MODE == 4) { if (SC1) asm volatile(LOADS(" sc1") "scratch_load_dword %4, off, off\n" TAIL OPS); else asm volatile(LOADS("") "scratch_load_dword %4, off, off\n" TAIL OPS); }
  const bool ok = d2 == value_at(l2 * 32) && d1 == value_at(l1 * 32) && d0 == value_at(l0 * 32);
  if (!ok) atomicAdd(&bad[2], 1u);                 // the loads themselves (must never happen)
  if (snap != d2) atomicAdd(&bad[0], 1u);          // the wait passed before load 2 had returned
  if (snap == kSentinel) atomicAdd(&bad[1], 1u);
  if (x == 0x12345u + priv[1]) sink[0] = x;        // keep x / priv alive
}

template <int MODE, bool SC1>
void run(const char* what, const unsigned* buf, uint64_t n_lines, unsigned* sink, const unsigned* hot, unsigned* bad) {
  unsigned h[3], tot[3] = {0, 0, 0};
  const int blocks = 4096, reps = 20;
  for (int r = 0; r < reps; ++r) {
    hipMemset(bad, 0, 12);
    hipLaunchKernelGGL((k_probe<MODE, SC1>), dim3(blocks), dim3(256), 0, 0, buf, n_lines, sink, hot, bad, (unsigned)(r * 5 + MODE * 1000 + SC1 * 77));
    hipMemcpy(h, bad, 12, hipMemcpyDeviceToHost);
    for (int i = 0; i < 3; ++i) tot[i] += h[i];
  }
  printf("%-52s slow loads %-5s: early snapshots %8u of %u (sentinel seen %u), wrong final loads %u\n", what, SC1 ? "sc1" : "plain",
         tot[0], blocks * 256 * reps, tot[1], tot[2]);
}

int main() {
  const uint64_t n = 512ull << 20;                 // 2 GB of dwords
  unsigned *buf, *sink, *hot, *bad;
  hipMalloc((void**)&buf, n * 4); hipMalloc((void**)&sink, 4096 * 256 * 4); hipMalloc((void**)&hot, 4096); hipMalloc((void**)&bad, 64);
  hipMemset(hot, 0, 4096);
  hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, buf, n);
  hipDeviceSynchronize();
  const uint64_t n_lines = n / 32;
  run<0, false>("X = global_store_dword", buf, n_lines, sink, hot, bad);
  run<0, true>("X = global_store_dword", buf, n_lines, sink, hot, bad);
  run<1, false>("X = global_store_dword sc1", buf, n_lines, sink, hot, bad);
  run<1, true>("X = global_store_dword sc1", buf, n_lines, sink, hot, bad);
  run<2, false>("X = scratch_store_dword", buf, n_lines, sink, hot, bad);
  run<2, true>("X = scratch_store_dword", buf, n_lines, sink, hot, bad);
  run<3, false>("X = global_load_dword (hot line)", buf, n_lines, sink, hot, bad);
  run<3, true>("X = global_load_dword (hot line)", buf, n_lines, sink, hot, bad);
  run<4, false>("X = scratch_load_dword", buf, n_lines, sink, hot, bad);
  run<4, true>("X = scratch_load_dword", buf, n_lines, sink, hot, bad);
  return 0;
}
