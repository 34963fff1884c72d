// Stand-alone form of the round-5 "stale ids on depth ties" case (tools/probes/spill_kit): a K-best list kept in registers,
// other sorted lists inserted into it with the (z, id) swap chain, four entries requested together -- the merge loop of the
// failing k_raster variant, cut out.  Lists with many equal depths; every lane's result is compared with a host merge.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tie_merge.hip -o tie_merge && ./tie_merge
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

template <int KMAX>
struct PixK {
  float z[KMAX];
  float q[KMAX];
  int id[KMAX];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int j = 0; j < KMAX; ++j) { z[j] = FLT_MAX; q[j] = -1.f; id[j] = 0x7fffffff; }
  }
  __device__ __forceinline__ void push(float cz, int ci, float cq, int K) {
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      if (j < K && (cz < z[j] || (cz == z[j] && ci < id[j]))) {
        float tz = z[j], tq = q[j]; int ti = id[j];
        z[j] = cz; q[j] = cq; id[j] = ci;
        cz = tz; cq = tq; ci = ti;
      }
    }
  }
};

constexpr int KMAX = 8;
// lists: [slice][z | q | id][k][256 lanes] per tile; the merging slice is `slice`
// FORM: how the own list gets into the registers; ATOMIC: agent-scope atomic loads (as in the raster kernel) or plain ones;
// GROUPS: 2 = the whole list in two groups of four, 1 = only the first four entries of every list; KCONST: K = KMAX at
// compile time.  `slice` is the list that starts in the registers and is skipped by the loop; n_lists lists per tile, the loop
// visits the first nslices of them (slice == nslices: nothing is skipped, the own list is an extra one)
template <int FORM, bool ATOMIC, int GROUPS, bool KCONST>
__global__ __launch_bounds__(256) void k_merge(const float* __restrict__ scratch, int nslices, int n_lists, int slice, int Krt,
                                               int* __restrict__ out_id, float* __restrict__ out_q) {
  const int K = KCONST ? KMAX : Krt;
  const float* base = scratch + (int64_t)blockIdx.x * n_lists * 3 * KMAX * 256;
  PixK<KMAX> best;
  if (FORM == 0) {                      // the own list straight into the registers
    const float* so = base + (int64_t)slice * 3 * KMAX * 256;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      best.z[j] = so[j * 256 + threadIdx.x];
      best.q[j] = so[(KMAX + j) * 256 + threadIdx.x];
      best.id[j] = __float_as_int(so[(2 * KMAX + j) * 256 + threadIdx.x]);
    }
  } else {                              // ... or built by insertion, as the raster phase does
    best.init();
    const float* so = base + (int64_t)slice * 3 * KMAX * 256;
    for (int j = KMAX - 1; j >= 0; --j) {
      const float zz = so[j * 256 + threadIdx.x];
      if (zz < FLT_MAX) best.push(zz, __float_as_int(so[(2 * KMAX + j) * 256 + threadIdx.x]), so[(KMAX + j) * 256 + threadIdx.x], K);
    }
  }
  for (int sI = 0; sI < nslices; ++sI) {
    if (sI == slice) continue;
    const float* so = base + (int64_t)sI * 3 * KMAX * 256;
#pragma unroll
    for (int j0 = 0; j0 < 4 * GROUPS; j0 += 4) {
      float zz[4], qq[4];
      int ii[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (ATOMIC) {
          zz[j] = __hip_atomic_load(so + (j0 + j) * 256 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          qq[j] = __hip_atomic_load(so + (KMAX + j0 + j) * 256 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ii[j] = __float_as_int(__hip_atomic_load(so + (2 * KMAX + j0 + j) * 256 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        } else {
          zz[j] = so[(j0 + j) * 256 + threadIdx.x];
          qq[j] = so[(KMAX + j0 + j) * 256 + threadIdx.x];
          ii[j] = __float_as_int(so[(2 * KMAX + j0 + j) * 256 + threadIdx.x]);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j0 + j < K && zz[j] < FLT_MAX) best.push(zz[j], ii[j], qq[j], K);
    }
  }
  const int64_t o = ((int64_t)blockIdx.x * 256 + threadIdx.x) * KMAX;
#pragma unroll
  for (int j = 0; j < KMAX; ++j) { out_id[o + j] = best.id[j]; out_q[o + j] = best.q[j]; }
}

struct E { float z, q; int id; };
static bool lessE(const E& a, const E& b) { return a.z < b.z || (a.z == b.z && a.id < b.id); }

template <int FORM, bool ATOMIC, int GROUPS, bool KCONST>
int run(const char* what, int tiles, int nslices, int K, bool skip) {
  const int n_lists = skip ? nslices : nslices + 1;
  const size_t per = (size_t)n_lists * 3 * KMAX * 256;
  std::vector<float> h(per * tiles);
  unsigned rng = 12345u;
  auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
  int next_id = 1;
  std::vector<std::vector<std::vector<E>>> lists((size_t)tiles * 256);
  for (int t = 0; t < tiles; ++t)
    for (int lane = 0; lane < 256; ++lane) {
      auto& ll = lists[(size_t)t * 256 + lane];
      for (int s = 0; s < n_lists; ++s) {
        const int n = (rnd() % 10 == 0) ? (int)(rnd() % (KMAX + 1)) : KMAX;       // some short lists
        std::vector<E> l;
        for (int j = 0; j < n; ++j) l.push_back({4.0f + (float)(rnd() % 24) * 0.03125f, (float)(rnd() % 1000) * 0.001f, (int)((unsigned)(next_id++) * 7919u % 1000003u)});
        std::sort(l.begin(), l.end(), lessE);
        float* so = h.data() + per * t + (size_t)s * 3 * KMAX * 256;
        for (int j = 0; j < KMAX; ++j) {
          const bool ok = j < n;
          so[j * 256 + lane] = ok ? l[j].z : FLT_MAX;
          so[(KMAX + j) * 256 + lane] = ok ? l[j].q : -1.f;
          int idv = ok ? l[j].id : 0x7fffffff;
          so[(2 * KMAX + j) * 256 + lane] = *reinterpret_cast<float*>(&idv);
        }
        ll.push_back(l);
      }
    }
  float* d; int* oid; float* oq;
  (void)hipMalloc((void**)&d, h.size() * 4); (void)hipMalloc((void**)&oid, (size_t)tiles * 256 * KMAX * 4); (void)hipMalloc((void**)&oq, (size_t)tiles * 256 * KMAX * 4);
  (void)hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  int bad_total = 0, bad = 0, badq = 0;
  for (int slice = skip ? 0 : nslices; slice < n_lists; ++slice) {
    hipLaunchKernelGGL((k_merge<FORM, ATOMIC, GROUPS, KCONST>), dim3(tiles), dim3(256), 0, 0, d, nslices, n_lists, slice, K, oid, oq);
    std::vector<int> gi((size_t)tiles * 256 * KMAX); std::vector<float> gq(gi.size());
    (void)hipMemcpy(gi.data(), oid, gi.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(gq.data(), oq, gq.size() * 4, hipMemcpyDeviceToHost);
    for (size_t p = 0; p < lists.size(); ++p) {
      // what the kernel is asked for: the own list (FORM 0: all of it, FORM 1: what K insertions keep), then the first
      // 4 * GROUPS entries (below K) of every other visited list
      std::vector<E> all = lists[p][slice];
      if (FORM == 1 && (int)all.size() > K) all.resize(K);
      for (int s2 = 0; s2 < nslices; ++s2) {
        if (s2 == slice) continue;
        for (int j = 0; j < (int)lists[p][s2].size() && j < K && j < 4 * GROUPS; ++j) all.push_back(lists[p][s2][j]);
      }
      std::sort(all.begin(), all.end(), lessE);
      if ((int)all.size() > K) all.resize(K);
      bool b = false, bq = false;
      for (size_t j = 0; j < all.size(); ++j) { b = b || gi[p * KMAX + j] != all[j].id; bq = bq || gq[p * KMAX + j] != all[j].q; }
      bad += b; badq += (!b && bq);
    }
  }
  printf("%-86s K = %d, %d lists: wrong ids %6d, right ids but wrong q %6d (of %zu)\n", what, K, nslices, bad, badq, lists.size() * (skip ? nslices : 1));
  bad_total = bad + badq;
  (void)hipFree(d); (void)hipFree(oid); (void)hipFree(oq);
  return bad_total;
}

int main() {
  int bad = 0;
  bad += run<0, true, 2, false>("as in the raster kernel: own list loaded, atomic loads, 2 groups, K at run time, skip", 64, 4, 8, true);
  bad += run<1, true, 2, false>("own list by insertion", 64, 4, 8, true);
  bad += run<0, false, 2, false>("plain loads", 64, 4, 8, true);
  bad += run<0, false, 2, true>("plain loads, K = KMAX at compile time", 64, 4, 8, true);
  bad += run<0, false, 1, false>("plain loads, only the first four entries of a list", 64, 4, 8, true);
  bad += run<0, false, 2, false>("plain loads, nothing skipped (own list is an extra one)", 64, 4, 8, false);
  bad += run<0, false, 1, true>("plain loads, first four, K at compile time, nothing skipped", 64, 4, 8, false);
  bad += run<0, true, 2, false>("as in the raster kernel, K = 7", 64, 3, 7, true);
  bad += run<0, true, 1, false>("atomic loads, only the first four entries of a list", 64, 4, 8, true);
  bad += run<0, true, 2, true>("atomic loads, K = KMAX at compile time", 64, 4, 8, true);
  bad += run<0, true, 2, false>("atomic loads, nothing skipped (own list is an extra one)", 64, 4, 8, false);
  bad += run<0, true, 1, true>("atomic loads, first four, K at compile time, nothing skipped", 64, 4, 8, false);
  printf(bad ? "WRONG RESULTS\n" : "all correct\n");
  return 0;
}
