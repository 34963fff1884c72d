// Minimal forms of the (z, id, q) swap-chain insertion, each checked lane by lane against the host on lists full of equal
// depths (see tie_merge.hip for the loop this was cut from).
//   hipcc -w --offload-arch=gfx950 -O3 -ffp-contract=off tie_push_min.hip -o tie_push_min && ./tie_push_min
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdio>
#include <vector>
#include <algorithm>
constexpr int KMAX = 8;
struct L { float z[KMAX], q[KMAX]; int id[KMAX]; };

// FORM 0: the source form (one `if` with a short-circuit condition, three swaps)
// FORM 1: the condition computed into a bool first
// FORM 2: selects instead of the `if`
template <int FORM>
__device__ __forceinline__ void push(L& b, float cz, int ci, float cq, int K) {
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    if (FORM == 0) {
      if (j < K && (cz < b.z[j] || (cz == b.z[j] && ci < b.id[j]))) {
        float tz = b.z[j], tq = b.q[j]; int ti = b.id[j];
        b.z[j] = cz; b.q[j] = cq; b.id[j] = ci;
        cz = tz; cq = tq; ci = ti;
      }
    } else if (FORM == 1) {
      const bool sw = j < K && (cz < b.z[j] || (cz == b.z[j] && ci < b.id[j]));
      if (sw) {
        float tz = b.z[j], tq = b.q[j]; int ti = b.id[j];
        b.z[j] = cz; b.q[j] = cq; b.id[j] = ci;
        cz = tz; cq = tq; ci = ti;
      }
    } else {
      const bool sw = j < K && (cz < b.z[j] || (cz == b.z[j] && ci < b.id[j]));
      const float tz = b.z[j], tq = b.q[j]; const int ti = b.id[j];
      b.z[j] = sw ? cz : tz; b.q[j] = sw ? cq : tq; b.id[j] = sw ? ci : ti;
      cz = sw ? tz : cz; cq = sw ? tq : cq; ci = sw ? ti : ci;
    }
  }
}

// list [z | q | id][k][lane]; candidates [z | q | id][c][lane]; NC candidates pushed one after the other
template <int FORM, int NC, bool KCONST>
__global__ __launch_bounds__(256) void k_push(const float* __restrict__ lists, const float* __restrict__ cands, int Krt, int n_lanes,
                                              int* __restrict__ out_id, float* __restrict__ out_q) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int K = KCONST ? KMAX : Krt;
  L b;
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    b.z[j] = lists[(size_t)j * n_lanes + t];
    b.q[j] = lists[(size_t)(KMAX + j) * n_lanes + t];
    b.id[j] = __float_as_int(lists[(size_t)(2 * KMAX + j) * n_lanes + t]);
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const float cz = cands[(size_t)c * n_lanes + t], cq = cands[(size_t)(NC + c) * n_lanes + t];
    const int ci = __float_as_int(cands[(size_t)(2 * NC + c) * n_lanes + t]);
    if (cz < FLT_MAX) push<FORM>(b, cz, ci, cq, K);
  }
#pragma unroll
  for (int j = 0; j < KMAX; ++j) { out_id[(size_t)j * n_lanes + t] = b.id[j]; out_q[(size_t)j * n_lanes + t] = b.q[j]; }
}

struct E { float z, q; int id; };
static bool lessE(const E& a, const E& b) { return a.z < b.z || (a.z == b.z && a.id < b.id); }
static float asf(int v) { return *reinterpret_cast<float*>(&v); }

template <int FORM, int NC, bool KCONST>
int run(const char* what, int K) {
  const int n = 256 * 64;
  std::vector<float> hl((size_t)3 * KMAX * n), hc((size_t)3 * NC * n);
  std::vector<std::vector<E>> want(n);
  unsigned rng = 777u;
  auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
  int next_id = 1;
  auto mk = [&]() { return E{4.0f + (float)(rnd() % 12) * 0.0625f, (float)(rnd() % 1000) * 0.001f, (int)((unsigned)(next_id++) * 7919u % 1000003u)}; };
  for (int t = 0; t < n; ++t) {
    std::vector<E> l;
    for (int j = 0; j < KMAX; ++j) l.push_back(mk());
    std::sort(l.begin(), l.end(), lessE);
    for (int j = 0; j < KMAX; ++j) { hl[(size_t)j * n + t] = l[j].z; hl[(size_t)(KMAX + j) * n + t] = l[j].q; hl[(size_t)(2 * KMAX + j) * n + t] = asf(l[j].id); }
    l.resize(K);
    for (int c = 0; c < NC; ++c) {
      E e = mk();
      hc[(size_t)c * n + t] = e.z; hc[(size_t)(NC + c) * n + t] = e.q; hc[(size_t)(2 * NC + c) * n + t] = asf(e.id);
      l.push_back(e); std::sort(l.begin(), l.end(), lessE); l.resize(K);
    }
    want[t] = l;
  }
  float *dl, *dc, *oq; int* oi;
  (void)hipMalloc((void**)&dl, hl.size() * 4); (void)hipMalloc((void**)&dc, hc.size() * 4); (void)hipMalloc((void**)&oi, (size_t)KMAX * n * 4); (void)hipMalloc((void**)&oq, (size_t)KMAX * n * 4);
  (void)hipMemcpy(dl, hl.data(), hl.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dc, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((k_push<FORM, NC, KCONST>), dim3(n / 256), dim3(256), 0, 0, dl, dc, K, n, oi, oq);
  std::vector<int> gi((size_t)KMAX * n); std::vector<float> gq(gi.size());
  (void)hipMemcpy(gi.data(), oi, gi.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(gq.data(), oq, gq.size() * 4, hipMemcpyDeviceToHost);
  int bad_id = 0, bad_q = 0;
  for (int t = 0; t < n; ++t) {
    bool b = false, bq = false;
    for (int j = 0; j < K; ++j) { b = b || gi[(size_t)j * n + t] != want[t][j].id; bq = bq || gq[(size_t)j * n + t] != want[t][j].q; }
    bad_id += b; bad_q += (!b && bq);
  }
  printf("%-64s K = %d: lanes with wrong ids %5d, right ids but wrong q %5d (of %d)\n", what, K, bad_id, bad_q, n);
  (void)hipFree(dl); (void)hipFree(dc); (void)hipFree(oi); (void)hipFree(oq);
  return bad_id + bad_q;
}

int main() {
  int bad = 0;
  bad += run<0, 1, false>("source form, 1 candidate, K at run time", 8);
  bad += run<0, 1, true>("source form, 1 candidate, K = KMAX at compile time", 8);
  bad += run<0, 4, false>("source form, 4 candidates, K at run time", 8);
  bad += run<0, 4, true>("source form, 4 candidates, K = KMAX at compile time", 8);
  bad += run<0, 4, false>("source form, 4 candidates, K at run time", 6);
  bad += run<1, 4, false>("condition in a bool first, 4 candidates, K at run time", 8);
  bad += run<2, 4, false>("selects, 4 candidates, K at run time", 8);
  printf(bad ? "WRONG RESULTS\n" : "all correct\n");
  return 0;
}
