// Shared between siren.hip (f32-MFMA step kernel, packing, C ABI) and siren_x3.hip (the
// split-fp16 MFMA step kernel): argument block and the layout of the packed weight buffer.
#pragma once
#include "iso_common.h"

struct SirenArgs {
  float* pts;                // (n,3) in/out
  float* normals;            // (n,3) out (may be null in eval mode)
  uint8_t* mask;             // (n) out
  float* sdf_out;            // eval mode
  float* grad_out;           // eval mode
  const int32_t* idx_in;     // active list (null = identity)
  const int32_t* count_in;   // device count of idx_in (null = n)
  int32_t* idx_out;          // survivors
  int32_t* count_out;
  const float* packed;
  float* stash;              // per-wave scratch
  int64_t n;
  int L;                     // hidden layers
  float w0, wh, tol;
  int do_move;               // 0: last evaluation (no move)
  int eval_only;
  // sphere tracing (iso_trace_siren): unit ray directions (n,3); null = Newton / evaluation
  const float* dirs = nullptr;
  float alpha = 1.f, bound = 0.f, tol_valid = 0.f;
  int fwd_only = 0;          // 1: the gradient is not needed (evaluation of the value / tracing)
  // a launch only works when cnt_lo < (device-side) count <= cnt_hi: the late Newton iterations are issued
  // twice, as the 96-point-tile kernel for long lists and as the 32-point-tile kernel for short ones
  // (a short list costs one tile time per launch whatever its length: 54 us against 25 us)
  int64_t cnt_lo = -1, cnt_hi = INT64_MAX;
  int small_tiles = 0;
  // split != 0 (H = 256 step kernels): the list is served by TWO launches, 96-point tiles for the slots
  // [0, siren_split_point(count)) (split = 1) and 32-point tiles for the rest (split = 2), so that the last, partly
  // filled round of the persistent grid costs a 32-point tile time instead of a 96-point one
  int split = 0;             // 3: both in one launch (k_siren_step_x3_both)
  int big_blocks = 0;        // split = 3: workgroups of the 96-point-tile shape (set by the launcher)
  // Newton tail (k_siren_tail_x3): ONE launch runs iterations it_first .. it_last; a workgroup keeps the survivors of
  // its own tiles in a private pair of lists and iterates them alone -- no launch, no grid-wide step per iteration
  int32_t* tail_lists = nullptr;   // [blocks][2][tail_cap] private survivor lists
  int32_t* tail_counts = nullptr;  // [blocks][2] their lengths (zero on entry of the launch: cleared by k_zero_counts' sibling)
  int32_t* iter_counts = nullptr;  // counts[it] of the projection (diagnostics: points evaluated by iteration it)
  int tail_cap = 0, it_first = 0, it_last = 0;
  // k_siren_step_x3_both: two counters (96-point tiles, 32-point tiles), ZERO when the launch starts; null: static tile
  // assignment.  A workgroup draws its next tile from them (x3_step_body) instead of taking every nblk-th.
  int32_t* tile_ctr = nullptr;
  int draw_first = 0;        // 1: the first tile of a workgroup comes from the counter too (no tile is anyone's by index)
  int ps_guard = 0;          // 1: this launch follows one of k_siren_step_ps on the same list and only works where that one declines
};

// Slots served by the 96-point-tile launch of a split list.  A round of the persistent grid is 256 tiles: 24 576
// points in 71 us (96-point tiles) or 8 192 points in 25 us (32-point tiles); the remainder after the full rounds goes
// to the small tiles when it fits two of their rounds, else it gets one more round of the large ones.
__host__ __device__ inline int64_t siren_split_point(int64_t count) {
  const int64_t big = 256 * 96, small = 256 * 32;
  const int64_t full = count / big * big, rem = count - full;
  if (rem == 0) return count;
  return (rem + small - 1) / small <= 2 ? full : count;
}

// ---- packed weight buffer, f32 section (siren.hip) ---------------------------------------
// [W0img 4*H][WLimg H][bL,pad 4][ per hidden layer: bias H | FW H*H | BW H*H ]
__host__ __device__ inline int64_t off_w0(int H) { (void)H; return 0; }
__host__ __device__ inline int64_t off_wl(int H) { return 4 * (int64_t)H; }
__host__ __device__ inline int64_t off_bl(int H) { return 5 * (int64_t)H; }
__host__ __device__ inline int64_t off_hidden(int H, int l) {
  return 5 * (int64_t)H + 4 + (int64_t)l * ((int64_t)H + 2 * (int64_t)H * H);
}

// ---- packed weight buffer, K-order section (siren_x3.hip), appended to the f32 section ----
// All per-feature vectors are in "K-order": position ko = (s*2 + h)*8 + e  <->  feature
//   f = x3_feat(s, 8h+e),  s = K-step of 16 features, h = lane half, e = element of the lane's
// 16-B B-operand entry.  With the D layout of v_mfma_f32_32x32x16_f16 (lane (h,j), register
// r of output tile T is row 8(r/4)+4h+(r%4)) registers 8p..8p+7 of tile T are exactly the
// lane's entry for K-step s = 2T+p, so activations never change lanes between layers.
//   [W0k 4*H][WLk H][ per hidden layer: bias_k H ]   (float units)
__host__ __device__ inline int x3_feat(int s, int kappa) {
  return 32 * (s >> 1) + 16 * (s & 1) + 8 * ((kappa & 7) >> 2) + 4 * (kappa >> 3) + (kappa & 3);
}
__host__ __device__ inline int64_t x3_base(int H, int L) { return off_hidden(H, L); }
__host__ __device__ inline int64_t x3_off_w0(int H, int L) { return x3_base(H, L); }
__host__ __device__ inline int64_t x3_off_wl(int H, int L) { return x3_base(H, L) + 4 * (int64_t)H; }
__host__ __device__ inline int64_t x3_off_layer(int H, int L, int l) {
  return x3_base(H, L) + 5 * (int64_t)H + (int64_t)l * (int64_t)H;
}
// ---- images in split fp16 (siren_x3.hip), appended to the K-order section --------------------
//   [24 floats: 0..7  2^s_l, the power-of-two scale of hidden layer l
//               8..15 c_l = max over input features f of sum_k |W_l[k][f]|  (growth bound of the adjoint)
//               16    max |W_head|
//               17..21 r_l = max_f (sum_k |W_l[f][k]| + |b_l[f]|), hidden layers 0..4 (bound of a hidden layer's pre-activation)
//               22, 23 max_f sum_c |W_0[f][c]|, max_f |b_0[f]|                      (the same for layer 0, per unit of max |x_c|) ]
//   [ per hidden layer: FW16 H*H ][ per hidden layer: BW16 H*H ]                       (float units)
// FW16 / BW16: uint4 index ((s*NTO + To)*2 + part)*64 + lane, 8 fp16 each (part 0/1 = high / low
// 11+11 bits of 2^s_l * W resp. its transpose), lane = 32h'+row:  W[32To+row][x3_feat(s,8h'+e)].
constexpr int kX16Header = 24;
__host__ __device__ inline int64_t x16_base(int H, int L) { return x3_off_layer(H, L, L); }
__host__ __device__ inline int64_t x16_off_layer(int H, int L, int l) { return x16_base(H, L) + kX16Header + (int64_t)l * H * H; }
__host__ __device__ inline int64_t x16_off_bw(int H, int L, int l) { return x16_off_layer(H, L, L) + (int64_t)l * H * H; }
__host__ __device__ inline int64_t siren_packed_total(int H, int L) { return x16_off_bw(H, L, L); }

// ---- siren_x3.hip entry points -----------------------------------------------------------
bool siren_x3_supported(int H, int L);
int64_t siren_x3_stash_floats(int H, int L);
void siren_x3_pack(const float* raw, float* packed, int H, int L, hipStream_t s);
int siren_x3_launch(const SirenArgs& a, int H, int64_t n_upper, hipStream_t s);
int siren_x3_tail_blocks();                                          // workgroups of the Newton-tail launch
int siren_x3_launch_tail(const SirenArgs& a, int H, hipStream_t s);  // H = 256 only
// ---- siren_ps.hip: the point-stationary form of the H = 256 step (bit-identical results) -------
// Which lists it serves (decided on the device by both kernels from the same data, so exactly one of them works): at
// least kPsMinList points -- a list is dealt out in tiles of 128 points over 256 workgroups, the last round is partly
// filled -- and hidden layers whose sine arguments provably stay below the large-argument threshold of iso_sin_wcos8
// (|w z| < 1e4: the kernel has no such path for them; true for every trained SIREN, r_l w ~ 10..100).
constexpr int64_t kPsMinList = 3 * 256 * 128;
__host__ __device__ inline bool siren_ps_takes(int64_t count, int L, const float* hdr /* packed + x16_base */, float wh) {
  if (count < kPsMinList) return false;
  float r = 0.f;
  for (int l = 0; l < L; ++l) r = hdr[17 + l] > r ? hdr[17 + l] : r;
  return (wh * r) * 1.01f < 1.0e4f;
}
bool siren_ps_supported(int H, int L);
int siren_ps_launch(const SirenArgs& a, int64_t n_upper, hipStream_t s);

