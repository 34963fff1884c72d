#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
static int g_tid, g_bid;
int fake_tid(void) { return g_tid; }
int fake_bid(void) { return g_bid; }
static int g_karg[80] __attribute__((aligned(16)));
void* fake_kernarg(void) { return g_karg; }
void k_before(const float*, int, int, int, int, int*, float*);
void k_test(const float*, int, int, int, int, int*, float*);
#define KMAX 8
int main(int argc, char** argv) {
  const int nsl = 4, K = 8;
  const size_t per = (size_t)nsl * 3 * KMAX * 256;
  float* h = malloc(per * 4);
  unsigned rng = 12345u;
  #define RND() (rng = rng * 1664525u + 1013904223u, rng >> 8)
  int next_id = 1, diff = 0;
  for (int lane = 0; lane < 256; ++lane)
    for (int s = 0; s < nsl; ++s) {
      int n = (RND() % 10 == 0) ? (int)(RND() % (KMAX + 1)) : KMAX;
      float z[KMAX], q[KMAX]; int id[KMAX];
      for (int j = 0; j < n; ++j) { z[j] = 4.0f + (float)(RND() % 24) * 0.03125f; q[j] = (float)(RND() % 1000) * 0.001f; id[j] = (int)((unsigned)(next_id++) * 7919u % 1000003u); }
      for (int a = 0; a < n; ++a) for (int b = a + 1; b < n; ++b) if (z[b] < z[a] || (z[b] == z[a] && id[b] < id[a])) { float t = z[a]; z[a] = z[b]; z[b] = t; t = q[a]; q[a] = q[b]; q[b] = t; int u = id[a]; id[a] = id[b]; id[b] = u; }
      float* so = h + (size_t)s * 3 * KMAX * 256;
      for (int j = 0; j < KMAX; ++j) { so[j * 256 + lane] = j < n ? z[j] : FLT_MAX; so[(KMAX + j) * 256 + lane] = j < n ? q[j] : -1.f; int v = j < n ? id[j] : 0x7fffffff; memcpy(&so[(2 * KMAX + j) * 256 + lane], &v, 4); }
    }
  int* ia = malloc(256 * KMAX * 4), *ib = malloc(256 * KMAX * 4); float* qa = malloc(256 * KMAX * 4), *qb = malloc(256 * KMAX * 4);
  for (int slice = 0; slice < nsl; ++slice) {
    for (g_tid = 0; g_tid < 256; ++g_tid) { g_bid = 0; g_karg[2] = nsl; g_karg[3] = nsl; g_karg[4] = slice; g_karg[5] = K; k_before(h, nsl, nsl, slice, K, ia, qa); k_test(h, nsl, nsl, slice, K, ib, qb); }
    for (int t = 0; t < 256; ++t) diff += memcmp(ia + t * KMAX, ib + t * KMAX, KMAX * 4) != 0 || memcmp(qa + t * KMAX, qb + t * KMAX, KMAX * 4) != 0;
  }
  printf("%-40s lanes that differ from the source IR: %4d of %d\n", argv[1], diff, 256 * nsl);
  return 0;
}
