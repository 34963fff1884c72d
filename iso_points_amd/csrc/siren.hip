// Fused SIREN SDF + gradient evaluation and Newton step on the f32 matrix cores
// (v_mfma_f32_16x16x4_f32), gfx950.
//
// Reference semantics: Siren.forward (DSS/models/common.py:140-165, SineLayer
// :86-87) evaluated with autograd.grad inside
// UniformProjection._compute_sdf_and_grad (levelset_sampling.py:142-170) and
// iterated by _project_points (:313-342).
//
// Work decomposition
//   * one wave = 16 points.  A hidden layer H->H is the GEMM
//       Z[H x 16pts] = W[H x H] . Hin[H x 16pts]
//     tiled as NT = H/16 output tiles of 16 rows; each 16x16x4 MFMA contracts 4
//     input features.  With the D-layout of that instruction (lane = 16g + j
//     holds rows 4g..4g+3 of column j) the register i of output tile t in lane
//     (g,j) is feature 16t+4g+i of point j -- which is exactly what the NEXT
//     layer's B operand wants for k-step (q=t, i) (B[k=g][j]).  So activations
//     never change lanes between layers: they are kept per wave in LDS as
//     hL[q][lane][0..3] (16 B per lane, lane-linear, conflict-free) and the B
//     operands of four consecutive k-steps are one ds_read_b128.
//   * the A operand (weights) is pre-arranged by iso_siren_pack_weights into a
//     lane-linear image FW[q][t][lane][0..3] (and its transpose BW for the
//     reverse sweep), so one q-chunk of all NT tiles is a contiguous NT KiB
//     block.  The 4 waves of a workgroup share it through a double-buffered
//     LDS stage (global -> registers -> LDS, one barrier per chunk).
//   * reverse-mode gradient: the forward sweep stashes s = w*cos(w*z) of every
//     sine layer in a per-wave scratch (L2 / Infinity-Cache resident, 1 KiB
//     coalesced rows); the reverse sweep multiplies the running adjoint by it
//     and runs the same GEMM loop on the transposed image.
//   * Newton: one launch = one evaluation (+ move) over the list of still
//     active points; survivors are appended to the next list with a wave
//     aggregated atomic, so converged points cost nothing in later launches
//     (the reference's boolean-mask compaction, without host syncs).
//
// Arithmetic: 2*H*H MAC per hidden layer and point (forward + reverse) on the
// matrix cores; layer 0 (3->H), the head (H->1) and sin/cos on the VALU.
#include <stdlib.h>
#include "siren_common.h"
#include "iso_newton.h"
#include "mlp_common.h"

namespace {

// raw layout: W0[H*3] b0[H] {Wi[H*H] bi[H]}*L WL[H] bL[1]
__global__ void k_siren_pack(const float* __restrict__ raw, float* __restrict__ packed,
                             int H, int L) {
  const int64_t total = off_hidden(H, L);
  const int64_t HH = (int64_t)H * H;
  const int NT = H / 16;
  for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total;
       o += (int64_t)gridDim.x * blockDim.x) {
    float v = 0.f;
    if (o < off_wl(H)) {
      // W0img[g][e][c]: c<3 -> W0[f][c], c==3 -> b0[f];  f = 16*(e>>2)+4g+(e&3)
      int64_t k = o;
      int c = (int)(k & 3);
      int e = (int)((k >> 2) % (H / 4));
      int g = (int)((k >> 2) / (H / 4));
      int f = 16 * (e >> 2) + 4 * g + (e & 3);
      v = (c < 3) ? raw[f * 3 + c] : raw[(int64_t)H * 3 + f];
    } else if (o < off_bl(H)) {
      int64_t k = o - off_wl(H);
      int e = (int)(k % (H / 4));
      int g = (int)(k / (H / 4));
      int f = 16 * (e >> 2) + 4 * g + (e & 3);
      const int64_t raw_wl = (int64_t)H * 4 + (int64_t)L * (HH + H);
      v = raw[raw_wl + f];
    } else if (o < off_hidden(H, 0)) {
      const int64_t raw_wl = (int64_t)H * 4 + (int64_t)L * (HH + H);
      v = (o == off_bl(H)) ? raw[raw_wl + H] : 0.f;
    } else {
      int64_t k = o - off_hidden(H, 0);
      const int64_t per = H + 2 * HH;
      int l = (int)(k / per);
      int64_t w = k % per;
      const float* Wl = raw + (int64_t)H * 4 + (int64_t)l * (HH + H);
      const float* bl = Wl + HH;
      if (w < H) {
        v = bl[w];
      } else {
        w -= H;
        bool bwd = w >= HH;
        if (bwd) w -= HH;
        // image index: [q][t][lane][i]
        int i = (int)(w & 3);
        int lane = (int)((w >> 2) & 63);
        int t = (int)((w >> 8) % NT);
        int q = (int)((w >> 8) / NT);
        int a = 16 * t + (lane & 15);            // tile row
        int b = 16 * q + 4 * (lane >> 4) + i;    // contraction index
        v = bwd ? Wl[(int64_t)b * H + a]         // BW: out=b (contracted), in=a
                : Wl[(int64_t)a * H + b];        // FW: out=a, in=b (contracted)
      }
    }
    packed[o] = v;
  }
}

// ---- the step kernel -------------------------------------------------------
#ifdef ISO_SIREN_DIRECT
#define ISO_GEMM_FWD(img, bias) gemm_pass_direct<NT, true>(img, bias, hL, acc, lane, g)
#define ISO_GEMM_BWD(img) gemm_pass_direct<NT, false>(img, nullptr, hL, acc, lane, g)
#else
#define ISO_GEMM_FWD(img, bias) gemm_pass<NT, true>(img, bias, hL, wbuf, acc, lane, g)
#define ISO_GEMM_BWD(img) gemm_pass<NT, false>(img, nullptr, hL, wbuf, acc, lane, g)
#endif

template <int NT>
__global__ __launch_bounds__(256, 2) void k_siren_step(SirenArgs a) {
  constexpr int H = NT * 16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int g = lane >> 4, j = lane & 15;
  float* hL = smem + wave * (NT * 256);
  float* wbuf = smem + 4 * NT * 256;   // 2 x (NT/2) KiB stage
  const float* W0img = a.packed + off_w0(H);
  const float* WLimg = a.packed + off_wl(H);
  const float bL = a.packed[off_bl(H)];
  float* stash = a.stash + ((int64_t)blockIdx.x * 4 + wave) * (int64_t)(a.L + 1) * NT * 256;

  const int64_t count = a.count_in ? (int64_t)(*a.count_in) : a.n;
  const int64_t n_tiles = (count + 63) / 64;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t slot = tile * 64 + wave * 16 + j;
    const bool valid = slot < count;
    int64_t idx = -1;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (valid) {
      idx = a.idx_in ? (int64_t)a.idx_in[slot] : slot;
      px = a.pts[idx * 3]; py = a.pts[idx * 3 + 1]; pz = a.pts[idx * 3 + 2];
    }
    // ---- layer 0 (3 -> H) on the VALU -------------------------------------
    // The TOP sine layer needs no stash: its adjoint seed is the head weight, so
    // gs = WL * w cos(w z) is formed on the spot (and the head dot product with it).
    float f = 0.f;
    const bool top0 = (a.L == 0);
    for (int e4 = 0; e4 < NT; ++e4) {
      f32x4 h4, s4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 w = reinterpret_cast<const f32x4*>(W0img)[g * (H / 4) + e4 * 4 + i];
        float z = ((w.x * px + w.y * py) + w.z * pz) + w.w;
        float s, c;
        iso_sincos(a.w0 * z, s, c);
        h4[i] = s;
        s4[i] = a.w0 * c;
      }
      if (top0) {
        const f32x4 w4 = reinterpret_cast<const f32x4*>(WLimg)[g * NT + e4];
        f += (w4.x * h4.x + w4.y * h4.y) + (w4.z * h4.z + w4.w * h4.w);
        h4 = (f32x4){w4.x * s4.x, w4.y * s4.y, w4.z * s4.z, w4.w * s4.w};
      }
      reinterpret_cast<f32x4*>(hL)[e4 * 64 + lane] = h4;
      if (!top0) reinterpret_cast<f32x4*>(stash)[e4 * 64 + lane] = s4;
    }
    // ---- hidden layers, forward -------------------------------------------
    f32x4 acc[NT];
    for (int l = 0; l < a.L; ++l) {
      const float* base = a.packed + off_hidden(H, l);
      ISO_GEMM_FWD(base + H, base);
      float* st_l = stash + (int64_t)(l + 1) * NT * 256;
      const bool top = (l == a.L - 1);
      // pre-activations go back to this wave's LDS slab (static register
      // indices), then a rolled loop applies sin / stashes w*cos
#pragma unroll
      for (int t = 0; t < NT; ++t) reinterpret_cast<f32x4*>(hL)[t * 64 + lane] = acc[t];
      for (int e4 = 0; e4 < NT; ++e4) {
        const f32x4 z4 = reinterpret_cast<const f32x4*>(hL)[e4 * 64 + lane];
        f32x4 h4, s4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float s, c;
          iso_sincos(a.wh * z4[i], s, c);
          h4[i] = s;
          s4[i] = a.wh * c;
        }
        if (top) {
          const f32x4 w4 = reinterpret_cast<const f32x4*>(WLimg)[g * NT + e4];
          f += (w4.x * h4.x + w4.y * h4.y) + (w4.z * h4.z + w4.w * h4.w);
          h4 = (f32x4){w4.x * s4.x, w4.y * s4.y, w4.z * s4.z, w4.w * s4.w};
          reinterpret_cast<f32x4*>(hL)[e4 * 64 + lane] = h4;
        } else {
          reinterpret_cast<f32x4*>(hL)[e4 * 64 + lane] = h4;
          reinterpret_cast<f32x4*>(st_l)[e4 * 64 + lane] = s4;
        }
      }
    }
    // ---- head: finish the dot product across the 4 lane groups ---------------
    f += __shfl_xor(f, 16);
    f += __shfl_xor(f, 32);
    f += bL;
    // ---- hidden layers, reverse -------------------------------------------
    for (int l = a.L - 1; l >= 0; --l) {
      const float* base = a.packed + off_hidden(H, l);
      ISO_GEMM_BWD(base + H + (int64_t)H * H);
      const float* st_l = stash + (int64_t)l * NT * 256;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const f32x4 s4 = reinterpret_cast<const f32x4*>(st_l)[t * 64 + lane];
        f32x4 gs = {acc[t].x * s4.x, acc[t].y * s4.y, acc[t].z * s4.z, acc[t].w * s4.w};
        reinterpret_cast<f32x4*>(hL)[t * 64 + lane] = gs;
      }
    }
    // ---- layer 0 reverse: grad = W0^T (adjoint . s0) ------------------------
    float gx = 0.f, gy = 0.f, gz = 0.f;
    for (int e4 = 0; e4 < NT; ++e4) {
      const f32x4 a4 = reinterpret_cast<const f32x4*>(hL)[e4 * 64 + lane];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 w = reinterpret_cast<const f32x4*>(W0img)[g * (H / 4) + e4 * 4 + i];
        gx += w.x * a4[i];
        gy += w.y * a4[i];
        gz += w.z * a4[i];
      }
    }
    gx += __shfl_xor(gx, 16); gx += __shfl_xor(gx, 32);
    gy += __shfl_xor(gy, 16); gy += __shfl_xor(gy, 32);
    gz += __shfl_xor(gz, 16); gz += __shfl_xor(gz, 32);

    // ---- epilogue ----------------------------------------------------------
    bool survive = false;
    if (valid && g == 0) survive = iso_step_finish(a, idx, f, gx, gy, gz);
    if (!a.eval_only && a.do_move) {
      const unsigned long long bal = __ballot(survive);
      if (bal) {
        int base = 0;
        const int leader = __ffsll((long long)bal) - 1;
        if (lane == leader) base = atomicAdd(a.count_out, __popcll(bal));
        base = __shfl(base, leader);
        if (survive) {
          const int rank = __popcll(bal & ((1ull << lane) - 1ull));
          a.idx_out[base + rank] = (int32_t)idx;
        }
      }
    }
    __syncthreads();
  }
}

constexpr int kSirenBlocks = 512;  // persistent: two workgroups per CU (80 KiB LDS each)

inline int64_t stash_floats(int H, int L) { return (int64_t)kSirenBlocks * 4 * (L + 1) * H * 16; }

template <int NT>
int launch_step(const SirenArgs& a, int blocks, hipStream_t s) {
  const size_t lds = (size_t)(5 * NT * 256) * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_siren_step<NT>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  hipLaunchKernelGGL(k_siren_step<NT>, dim3(blocks), dim3(256), lds, s, a);
  return 0;
}

int dispatch_step(const SirenArgs& a, int H, int blocks, hipStream_t s) {
  switch (H / 16) {
    case 4: return launch_step<4>(a, blocks, s);
    case 8: return launch_step<8>(a, blocks, s);
    case 16: return launch_step<16>(a, blocks, s);
    default: return -1;
  }
}

__global__ void k_zero_counts(int32_t* c, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) c[i] = 0;
}

bool siren_shape_ok(int H, int L) {
  return (H == 64 || H == 128 || H == 256) && L >= 0 && L <= 8;
}

// 0 = f32 MFMA kernel (this file), 1 = split-fp16 MFMA kernel (siren_x3.hip) where it applies.
// Default 1; the environment variable ISO_SIREN_GEMM=f32 selects 0 at load time.
int g_gemm_mode = -1;
int gemm_mode() {
  if (g_gemm_mode < 0) {
    const char* e = getenv("ISO_SIREN_GEMM");
    g_gemm_mode = (e && (!strcmp(e, "f32") || !strcmp(e, "0"))) ? 0 : 1;
  }
  return g_gemm_mode;
}
bool use_x3(int H, int L) { return gemm_mode() == 1 && siren_x3_supported(H, L); }

int64_t stash_floats_any(int H, int L) {
  int64_t a = stash_floats(H, L);
  int64_t b = siren_x3_supported(H, L) ? siren_x3_stash_floats(H, L) : 0;
  return a > b ? a : b;
}

int run_step(const SirenArgs& a, int H, int64_t n, hipStream_t s) {
  if (use_x3(H, a.L)) return siren_x3_launch(a, H, n, s);
  int64_t tiles = (n + 63) / 64;
  int blocks = (int)(tiles < kSirenBlocks ? tiles : kSirenBlocks);
  return dispatch_step(a, H, blocks, s);
}

}  // namespace

extern "C" int64_t iso_siren_raw_floats(int hidden, int n_hidden) {
  int64_t H = hidden;
  return H * 3 + H + (int64_t)n_hidden * (H * H + H) + H + 1;
}

extern "C" int64_t iso_siren_packed_floats(int hidden, int n_hidden) {
  return siren_packed_total(hidden, n_hidden);
}

extern "C" int iso_siren_set_gemm_mode(int mode) {
  ISO_REQUIRE(mode == 0 || mode == 1, ISO_ERR_INVALID, "iso_siren_set_gemm_mode: mode must be 0 (f32 MFMA) or 1 (split fp16)");
  g_gemm_mode = mode;
  return ISO_OK;
}

extern "C" int iso_siren_get_gemm_mode(void) { return gemm_mode(); }

extern "C" int iso_siren_pack_weights(const float* raw, float* packed, int hidden,
                                      int n_hidden, void* stream) {
  ISO_REQUIRE(siren_shape_ok(hidden, n_hidden), ISO_ERR_UNSUPPORTED,
              "iso_siren_pack_weights: hidden must be 64/128/256 and 0<=n_hidden<=8 (got %d, %d)",
              hidden, n_hidden);
  ISO_REQUIRE(raw && packed, ISO_ERR_INVALID, "iso_siren_pack_weights: null pointer");
  int64_t total = off_hidden(hidden, n_hidden);
  hipLaunchKernelGGL(k_siren_pack, dim3(iso_stream_grid(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, raw, packed, hidden, n_hidden);
  if (hidden % 32 == 0) siren_x3_pack(raw, packed, hidden, n_hidden, (hipStream_t)stream);
  ISO_CHECK_LAUNCH("iso_siren_pack_weights");
  return ISO_OK;
}

// Newton tail: private survivor lists of the tail launch's workgroups, two of tail_cap entries each (a workgroup's share of
// a list is at most ceil(tiles / blocks) tiles of 32) + their two counters
static int64_t tail_cap_of(int64_t n) {
  const int64_t blocks = siren_x3_tail_blocks(), tiles = (n + 31) / 32;
  return ((tiles + blocks - 1) / blocks + 1) * 32;
}
static int64_t tail_ints(int64_t n) { return siren_x3_tail_blocks() * (2 * tail_cap_of(n) + 2); }

// workspace: [stash floats][idx A n][idx B n][tail lists + counters][counts 64][tile counters: two per launch]
constexpr int kTileCtrInts = 128;     // (max_iters <= 60: 61 launches)
extern "C" int64_t iso_project_siren_workspace_bytes(int64_t n, int hidden, int n_hidden) {
  if (n < 0) n = 0;
  return stash_floats_any(hidden, n_hidden) * 4 + 2 * n * 4 + tail_ints(n) * 4 + (64 + kTileCtrInts) * 4 + 64;
}

extern "C" int64_t iso_project_siren_counts_offset(int64_t n, int hidden, int n_hidden) {
  if (n < 0) n = 0;
  return stash_floats_any(hidden, n_hidden) * 4 + 2 * n * 4 + tail_ints(n) * 4;
}

static bool siren_small_tiles_enabled() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("ISO_SIREN_SMALL_TILES"); v = (e && e[0] == '0') ? 0 : 1; }
  return v == 1;
}

static int g_siren_dyn_tiles = -1;              // -1: from ISO_SIREN_DYN_TILES (default on), 0 / 1: iso_siren_set_drawn_tiles
static bool siren_dynamic_tiles_enabled() {     // off: static tile assignment (A/B, tests)
  if (g_siren_dyn_tiles < 0) { const char* e = getenv("ISO_SIREN_DYN_TILES"); g_siren_dyn_tiles = (e && e[0] == '0') ? 0 : 1; }
  return g_siren_dyn_tiles == 1;
}
extern "C" int iso_siren_set_drawn_tiles(int on) {
  ISO_REQUIRE(on >= -1 && on <= 1, ISO_ERR_INVALID, "iso_siren_set_drawn_tiles: -1 (environment / default), 0 or 1");
  g_siren_dyn_tiles = on;
  return ISO_OK;
}

static bool siren_merged_shapes_enabled() {      // ISO_SIREN_MERGED=0: the two tile shapes as two launches (A/B)
  static int v = -1;
  if (v < 0) { const char* e = getenv("ISO_SIREN_MERGED"); v = (e && e[0] == '0') ? 0 : 1; }
  return v == 1;
}

// One evaluation of a list whose length only the device knows: for the H = 256 gradient kernels the list is cut at
// siren_split_point into a 96-point-tile launch and a 32-point-tile launch (per-point results do not depend on the
// tile shape); everything else is one launch.
static int run_step_split(SirenArgs a, int hidden, int64_t n, hipStream_t s) {
  const bool split = hidden == 256 && !a.fwd_only && use_x3(hidden, a.L) && siren_small_tiles_enabled();
  if (split && siren_merged_shapes_enabled()) {
    a.small_tiles = 0; a.split = 3;
    return run_step(a, hidden, n, s);
  }
  a.small_tiles = 0; a.split = split ? 1 : 0;
  a.tile_ctr = nullptr;                      // (the counters are per launch of k_siren_step_x3_both)
  int rc = run_step(a, hidden, n, s);
  if (rc != 0 || !split) return rc;
  a.small_tiles = 1; a.split = 2;
  return run_step(a, hidden, n, s);
}

// The iteration driver shared by the Newton projection and sphere tracing: launch `it` evaluates
// the list launch it-1 left (device-side counts, no host read), the last one does not move.
// From which iteration on the Newton projection runs as ONE tail launch (k_siren_tail_x3): the lists of the first
// iterations are long (launch-per-iteration, 96-point tiles), the late ones a few hundred points or none.  T >= 6: after
// four launches; shorter projections (the T = 3 re-projection of resample): after two.  ISO_SIREN_TAIL_FROM overrides
// (0: never, the launch-per-iteration form).
static int g_tail_from = -2;       // -1: default policy, 0: never, k > 0: from iteration k
static int siren_tail_from(int max_iters) {
  if (g_tail_from == -2) { const char* e = getenv("ISO_SIREN_TAIL_FROM"); g_tail_from = e ? atoi(e) : -1; }
  if (g_tail_from == 0) return max_iters + 1;
  if (g_tail_from > 0) return g_tail_from;
  return max_iters >= 6 ? 4 : 2;
}
extern "C" int iso_siren_step_launches(int hidden, int n_hidden, int max_iters) {
  if (max_iters < 0) return 0;
  const bool can_tail = hidden == 256 && use_x3(hidden, n_hidden) && siren_small_tiles_enabled();
  const int from = can_tail ? siren_tail_from(max_iters) : max_iters + 1;
  return from <= max_iters && from > 0 ? from + 1 : max_iters + 1;
}
extern "C" int iso_siren_set_tail_from(int first_tail_iteration) {
  ISO_REQUIRE(first_tail_iteration >= -1, ISO_ERR_INVALID, "iso_siren_set_tail_from: -1 (default), 0 (never) or an iteration >= 1");
  g_tail_from = first_tail_iteration;
  return ISO_OK;
}

__global__ void k_zero_tail(int32_t* c, int n, int32_t* tc, int m) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) c[i] = 0;
  if (i < m) tc[i] = 0;
}

// The iteration driver shared by the Newton projection and sphere tracing: launch `it` evaluates
// the list launch it-1 left (device-side counts, no host read), the last one does not move.
static int run_iterations(SirenArgs a, int hidden, int64_t n, int max_iters, void* workspace,
                          hipStream_t s, const char* who) {
  float* stash = (float*)workspace;
  int32_t* idxA = (int32_t*)(stash + stash_floats_any(hidden, a.L));
  int32_t* idxB = idxA + n;
  int32_t* tail = idxB + n;
  int32_t* counts = tail + tail_ints(n);  // counts[it] = size of the list consumed by launch `it`
  const int blocks = siren_x3_tail_blocks();
  const int64_t cap = tail_cap_of(n);
  int32_t* tail_counts = tail + (int64_t)blocks * 2 * cap;
  const bool can_tail = hidden == 256 && !a.dirs && !a.fwd_only && use_x3(hidden, a.L) && siren_small_tiles_enabled();
  const int tail_from = can_tail ? siren_tail_from(max_iters) : max_iters + 1;
  int32_t* tile_ctr = counts + 64;          // [launch][2]: drawn from by the workgroups of k_siren_step_x3_both (zeroed with the counts)
  hipLaunchKernelGGL(k_zero_tail, dim3((2 * blocks + 255) / 256), dim3(256), 0, s, counts, 64 + kTileCtrInts, tail_counts,
                     tail_from <= max_iters ? 2 * blocks : 0);
  const bool dyn_tiles = siren_dynamic_tiles_enabled() && 2 * (max_iters + 1) <= kTileCtrInts;
  a.stash = stash; a.n = n; a.eval_only = 0; a.sdf_out = a.dirs ? a.sdf_out : nullptr; a.grad_out = nullptr;
  for (int it = 0; it <= max_iters; ++it) {
    a.idx_in = (it == 0) ? nullptr : ((it & 1) ? idxA : idxB);
    a.count_in = (it == 0) ? nullptr : counts + it;
    a.idx_out = (it & 1) ? idxB : idxA;
    a.count_out = counts + it + 1;
    a.do_move = (it < max_iters) ? 1 : 0;
    a.tile_ctr = dyn_tiles ? tile_ctr + 2 * it : nullptr;
    if (it >= tail_from && it > 0) {        // iterations it .. max_iters in one launch
      a.tail_lists = tail; a.tail_counts = tail_counts; a.iter_counts = counts; a.tail_cap = (int)cap;
      a.it_first = it; a.it_last = max_iters;
      ISO_REQUIRE(siren_x3_launch_tail(a, hidden, s) == 0, ISO_ERR_UNSUPPORTED, "%s: no tail kernel for hidden size %d", who, hidden);
      break;
    }
    ISO_REQUIRE(run_step_split(a, hidden, n, s) == 0, ISO_ERR_UNSUPPORTED, "%s: unsupported hidden size %d", who, hidden);
  }
  return ISO_OK;
}

extern "C" int iso_project_siren(const float* pts_in, float* pts_out, float* normals_out,
                                 uint8_t* mask_out, int64_t n, const float* packed,
                                 int hidden, int n_hidden, float omega_first,
                                 float omega_hidden, int max_iters, float tol,
                                 void* workspace, int64_t workspace_bytes, void* stream) {
  ISO_REQUIRE(siren_shape_ok(hidden, n_hidden), ISO_ERR_UNSUPPORTED,
              "iso_project_siren: hidden must be 64/128/256 and 0<=n_hidden<=8 (got %d, %d)",
              hidden, n_hidden);
  ISO_REQUIRE(n >= 0 && max_iters >= 0 && max_iters <= 60, ISO_ERR_INVALID,
              "iso_project_siren: bad n / max_iters (max 60)");
  if (n == 0) return ISO_OK;
  ISO_REQUIRE(n < (1ll << 31), ISO_ERR_UNSUPPORTED, "iso_project_siren: n must fit int32");
  ISO_REQUIRE(pts_in && pts_out && normals_out && mask_out && packed && workspace,
              ISO_ERR_INVALID, "iso_project_siren: null pointer");
  ISO_REQUIRE(workspace_bytes >= iso_project_siren_workspace_bytes(n, hidden, n_hidden),
              ISO_ERR_WORKSPACE, "iso_project_siren: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  if (pts_out != pts_in)
    (void)hipMemcpyAsync(pts_out, pts_in, (size_t)n * 12, hipMemcpyDeviceToDevice, s);
  SirenArgs a;
  a.pts = pts_out; a.normals = normals_out; a.mask = mask_out;
  a.packed = packed; a.L = n_hidden;
  a.w0 = omega_first; a.wh = omega_hidden; a.tol = tol;
  int rc = run_iterations(a, hidden, n, max_iters, workspace, s, "iso_project_siren");
  if (rc != ISO_OK) return rc;
  ISO_CHECK_LAUNCH("iso_project_siren");
  return ISO_OK;
}

extern "C" int iso_trace_siren(const float* ray0, const float* dirs, float* pts_out, float* sdf_out,
                               uint8_t* mask_out, int64_t n, const float* packed, int hidden,
                               int n_hidden, float omega_first, float omega_hidden, float alpha,
                               float bound, int max_iters, float tol, void* workspace,
                               int64_t workspace_bytes, void* stream) {
  ISO_REQUIRE(siren_shape_ok(hidden, n_hidden), ISO_ERR_UNSUPPORTED,
              "iso_trace_siren: hidden must be 64/128/256 and 0<=n_hidden<=8 (got %d, %d)",
              hidden, n_hidden);
  ISO_REQUIRE(n >= 0 && max_iters >= 0 && max_iters <= 60, ISO_ERR_INVALID,
              "iso_trace_siren: bad n / max_iters (max 60)");
  if (n == 0) return ISO_OK;
  ISO_REQUIRE(n < (1ll << 31), ISO_ERR_UNSUPPORTED, "iso_trace_siren: n must fit int32");
  ISO_REQUIRE(ray0 && dirs && pts_out && sdf_out && mask_out && packed && workspace,
              ISO_ERR_INVALID, "iso_trace_siren: null pointer");
  ISO_REQUIRE(workspace_bytes >= iso_project_siren_workspace_bytes(n, hidden, n_hidden),
              ISO_ERR_WORKSPACE, "iso_trace_siren: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  if (pts_out != ray0)
    (void)hipMemcpyAsync(pts_out, ray0, (size_t)n * 12, hipMemcpyDeviceToDevice, s);
  SirenArgs a;
  a.pts = pts_out; a.normals = nullptr; a.mask = mask_out; a.sdf_out = sdf_out;
  a.packed = packed; a.L = n_hidden;
  a.w0 = omega_first; a.wh = omega_hidden;
  a.dirs = dirs; a.alpha = alpha; a.bound = bound;
  a.tol = 0.1f * tol;                  // levelset_sampling.py:764: still active above 1e-1 * tolerance
  a.tol_valid = tol;                   // :790: valid projection = |sdf| <= tolerance
  a.fwd_only = 1;
  int rc = run_iterations(a, hidden, n, max_iters, workspace, s, "iso_trace_siren");
  if (rc != ISO_OK) return rc;
  ISO_CHECK_LAUNCH("iso_trace_siren");
  return ISO_OK;
}

extern "C" int iso_siren_sdf_grad(const float* pts, float* sdf_out, float* grad_out,
                                  int64_t n, const float* packed, int hidden,
                                  int n_hidden, float omega_first, float omega_hidden,
                                  void* workspace, int64_t workspace_bytes, void* stream) {
  ISO_REQUIRE(siren_shape_ok(hidden, n_hidden), ISO_ERR_UNSUPPORTED,
              "iso_siren_sdf_grad: hidden must be 64/128/256 and 0<=n_hidden<=8 (got %d, %d)",
              hidden, n_hidden);
  ISO_REQUIRE(n >= 0, ISO_ERR_INVALID, "iso_siren_sdf_grad: n < 0");
  ISO_REQUIRE(n < (1ll << 31), ISO_ERR_UNSUPPORTED, "iso_siren_sdf_grad: n must fit int32");
  if (n == 0) return ISO_OK;
  ISO_REQUIRE(pts && sdf_out && packed && workspace, ISO_ERR_INVALID,
              "iso_siren_sdf_grad: null pointer");
  ISO_REQUIRE(workspace_bytes >= stash_floats_any(hidden, n_hidden) * 4, ISO_ERR_WORKSPACE,
              "iso_siren_sdf_grad: workspace too small");
  SirenArgs a;
  a.pts = const_cast<float*>(pts); a.normals = nullptr; a.mask = nullptr;
  a.sdf_out = sdf_out; a.grad_out = grad_out;
  a.idx_in = nullptr; a.count_in = nullptr; a.idx_out = nullptr; a.count_out = nullptr;
  a.packed = packed; a.stash = (float*)workspace; a.n = n; a.L = n_hidden;
  a.w0 = omega_first; a.wh = omega_hidden; a.tol = 0.f; a.do_move = 0; a.eval_only = 1;
  a.fwd_only = grad_out ? 0 : 1;       // value only: forward sweep only where the kernel has one
  // a workspace of the projection's size (what the Python side allocates) has room for the two tile counters
  if (hidden == 256 && !a.fwd_only && siren_dynamic_tiles_enabled() &&
      workspace_bytes >= iso_project_siren_workspace_bytes(n, hidden, n_hidden)) {
    int32_t* ctr = (int32_t*)((char*)workspace + iso_project_siren_counts_offset(n, hidden, n_hidden)) + 64;
    hipLaunchKernelGGL(k_zero_counts, dim3(1), dim3(64), 0, (hipStream_t)stream, ctr, 2);
    a.tile_ctr = ctr;
  }
  ISO_REQUIRE(run_step_split(a, hidden, n, (hipStream_t)stream) == 0, ISO_ERR_UNSUPPORTED,
              "iso_siren_sdf_grad: unsupported hidden size %d", hidden);
  ISO_CHECK_LAUNCH("iso_siren_sdf_grad");
  return ISO_OK;
}
