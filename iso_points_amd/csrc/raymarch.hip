// State updates of IDR's two-ended ray marcher between two network evaluations.
//
// Reference: RayTracing.sphere_tracing / .secant, DSS/models/levelset_sampling.py:920-1032, :1114-1133.
// Every ray carries two marching ends (e = 0 walks forward from the bounding-sphere entry, e = 1
// backward from the exit).  Per iteration the reference runs ~40 masked tensor statements and
// several `.sum() > 0` host reads around its two network calls; here the statements between two
// network evaluations are ONE kernel that also writes the next evaluation's input -- the points of
// the rows that still need a value, compacted into a list (wave-aggregated slot allocation, one
// atomic per wave) -- and the number of such rows, which is the only thing the host reads.
//
//   settle     (:962-990, :1027-1030)  cur = live ? nxt : 0;  cur <= thr -> 0;  live &= cur > thr;
//                                      z += +-cur;  list <- cam + z d  of the live ends
//   overshoot  (:993-1025)             nxt <- values;  over = nxt < 0;
//                                      z -= +-(back * cur);  list <- cam + z d  of the overshot ends
//   secant     (:1114-1133)            bracket update by the sign of f(mid), next false-position depth
//
// One thread owns a ray (both ends: the ends meet in the `z0 < z1` test).  Arithmetic is the
// reference's statement for statement in f32 (the library is built with -ffp-contract=off), so a
// ray's trajectory does not depend on which other rays are still alive.  HBM-bound and tiny
// (<= 60 B per ray and launch); what it removes is launch count, not bytes.
#include "iso_common.h"

namespace {

constexpr int kBlock = 256;

// Append up to two entries per lane (flags a0, a1) to the list: all "end 0" entries of the wave
// first, then its "end 1" entries.  Returns the slots (or -1).
__device__ __forceinline__ void wave_append(bool a0, bool a1, int32_t* count, int& s0, int& s1) {
  const unsigned long long m0 = __ballot(a0), m1 = __ballot(a1);
  const int lane = threadIdx.x & (ISO_WAVE - 1);
  const int n0 = __popcll(m0), n1 = __popcll(m1);
  int base = 0;
  if (lane == 0 && n0 + n1 > 0) base = atomicAdd(count, n0 + n1);
  base = __shfl(base, 0);
  const unsigned long long below = (1ull << lane) - 1ull;
  s0 = a0 ? base + __popcll(m0 & below) : -1;
  s1 = a1 ? base + n0 + __popcll(m1 & below) : -1;
}

__device__ __forceinline__ void put_point(float* list, int slot, const float* cam, const float* dir,
                                          int64_t r, float z) {
  if (slot < 0) return;
  list[(int64_t)slot * 3] = cam[r * 3] + z * dir[r * 3];
  list[(int64_t)slot * 3 + 1] = cam[r * 3 + 1] + z * dir[r * 3 + 1];
  list[(int64_t)slot * 3 + 2] = cam[r * 3 + 2] + z * dir[r * 3 + 2];
}

__global__ __launch_bounds__(kBlock) void k_march_settle(
    const float* __restrict__ cam, const float* __restrict__ dirs, int64_t R, float* __restrict__ z,
    float* __restrict__ cur, const float* __restrict__ nxt, uint8_t* __restrict__ live, float thr,
    int check_order, int step, int32_t* __restrict__ slot, float* __restrict__ list,
    int32_t* __restrict__ count) {
  const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool in = r < R;
  bool l0 = false, l1 = false;
  float z0 = 0.f, z1 = 0.f;
  if (in) {
    z0 = z[r]; z1 = z[R + r];
    l0 = live[r] != 0; l1 = live[R + r] != 0;
    if (check_order) { const bool ordered = z0 < z1; l0 = l0 && ordered; l1 = l1 && ordered; }
    float c0 = l0 ? nxt[r] : 0.f, c1 = l1 ? nxt[R + r] : 0.f;
    if (c0 <= thr) c0 = 0.f;
    if (c1 <= thr) c1 = 0.f;
    l0 = l0 && (c0 > thr); l1 = l1 && (c1 > thr);
    cur[r] = c0; cur[R + r] = c1;
    live[r] = l0; live[R + r] = l1;
    if (step) {
      z0 = z0 + c0; z1 = z1 - c1;
      z[r] = z0; z[R + r] = z1;
    }
  }
  int s0, s1;
  wave_append(l0, l1, count, s0, s1);   // *count = unfinished ends (also when the iteration cap forbids the step)
  if (in && step) {
    slot[r] = s0; slot[R + r] = s1;
    put_point(list, s0, cam, dirs, r, z0);
    put_point(list, s1, cam, dirs, r, z1);
  }
}

__global__ __launch_bounds__(kBlock) void k_march_overshoot(
    const float* __restrict__ cam, const float* __restrict__ dirs, int64_t R, float* __restrict__ z,
    const float* __restrict__ cur, float* __restrict__ nxt, const float* __restrict__ values, int first,
    int may_backstep, float back, int32_t* __restrict__ slot, float* __restrict__ list,
    int32_t* __restrict__ count) {
  const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool in = r < R;
  bool o0 = false, o1 = false;
  float z0 = 0.f, z1 = 0.f;
  if (in) {
    const int p0 = slot[r], p1 = slot[R + r];
    float n0 = first ? 0.f : nxt[r], n1 = first ? 0.f : nxt[R + r];
    if (p0 >= 0) n0 = values[p0];
    if (p1 >= 0) n1 = values[p1];
    nxt[r] = n0; nxt[R + r] = n1;
    o0 = may_backstep && (n0 < 0.f); o1 = may_backstep && (n1 < 0.f);
    z0 = z[r]; z1 = z[R + r];
    if (o0) { z0 = z0 - back * cur[r]; z[r] = z0; }
    if (o1) { z1 = z1 + back * cur[R + r]; z[R + r] = z1; }
  }
  int s0, s1;
  wave_append(o0, o1, count, s0, s1);
  if (in) {
    slot[r] = s0; slot[R + r] = s1;
    put_point(list, s0, cam, dirs, r, z0);
    put_point(list, s1, cam, dirs, r, z1);
  }
}

__global__ __launch_bounds__(kBlock) void k_secant_next(
    const float* __restrict__ cam, const float* __restrict__ dirs, int64_t n, float* __restrict__ f_lo,
    float* __restrict__ f_hi, float* __restrict__ z_lo, float* __restrict__ z_hi, float* __restrict__ zp,
    const float* __restrict__ f_mid, float* __restrict__ pts) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  float fl = f_lo[i], fh = f_hi[i], zl = z_lo[i], zh = z_hi[i];
  if (f_mid) {
    const float fm = f_mid[i], zm = zp[i];
    if (fm > 0.f) { zl = zm; fl = fm; }
    if (fm < 0.f) { zh = zm; fh = fm; }
    f_lo[i] = fl; f_hi[i] = fh; z_lo[i] = zl; z_hi[i] = zh;
  }
  const float zn = (-fl) * (zh - zl) / (fh - fl) + zl;
  zp[i] = zn;
  pts[i * 3] = cam[i * 3] + zn * dirs[i * 3];
  pts[i * 3 + 1] = cam[i * 3 + 1] + zn * dirs[i * 3 + 1];
  pts[i * 3 + 2] = cam[i * 3 + 2] + zn * dirs[i * 3 + 2];
}

}  // namespace

extern "C" int iso_raymarch_settle(const float* cam, const float* dirs, int64_t n_rays, float* z,
                                   float* cur, const float* nxt, uint8_t* live, float sdf_threshold,
                                   int check_order, int step, int32_t* slot, float* list,
                                   int32_t* count, void* stream) {
  ISO_REQUIRE(n_rays >= 0 && n_rays < (1ll << 30), ISO_ERR_INVALID, "iso_raymarch_settle: bad ray count");
  ISO_REQUIRE(count, ISO_ERR_INVALID, "iso_raymarch_settle: null count");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(count, 0, sizeof(int32_t), st) != hipSuccess) {
    iso_set_error("iso_raymarch_settle: memset failed");
    return ISO_ERR_LAUNCH;
  }
  if (n_rays == 0) return ISO_OK;
  ISO_REQUIRE(cam && dirs && z && cur && nxt && live && slot && list, ISO_ERR_INVALID,
              "iso_raymarch_settle: null argument");
  hipLaunchKernelGGL(k_march_settle, dim3(iso_div_up(n_rays, kBlock)), dim3(kBlock), 0, st, cam, dirs,
                     n_rays, z, cur, nxt, live, sdf_threshold, check_order, step, slot, list, count);
  ISO_CHECK_LAUNCH("iso_raymarch_settle");
  return ISO_OK;
}

extern "C" int iso_raymarch_overshoot(const float* cam, const float* dirs, int64_t n_rays, float* z,
                                      const float* cur, float* nxt, const float* values, int first,
                                      int may_backstep, float back, int32_t* slot, float* list,
                                      int32_t* count, void* stream) {
  ISO_REQUIRE(n_rays >= 0 && n_rays < (1ll << 30), ISO_ERR_INVALID, "iso_raymarch_overshoot: bad ray count");
  ISO_REQUIRE(count, ISO_ERR_INVALID, "iso_raymarch_overshoot: null count");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(count, 0, sizeof(int32_t), st) != hipSuccess) {
    iso_set_error("iso_raymarch_overshoot: memset failed");
    return ISO_ERR_LAUNCH;
  }
  if (n_rays == 0) return ISO_OK;
  ISO_REQUIRE(cam && dirs && z && cur && nxt && slot && list, ISO_ERR_INVALID,
              "iso_raymarch_overshoot: null argument");
  hipLaunchKernelGGL(k_march_overshoot, dim3(iso_div_up(n_rays, kBlock)), dim3(kBlock), 0, st, cam,
                     dirs, n_rays, z, cur, nxt, values, first, may_backstep, back, slot, list, count);
  ISO_CHECK_LAUNCH("iso_raymarch_overshoot");
  return ISO_OK;
}

extern "C" int iso_raymarch_secant(const float* cam, const float* dirs, int64_t n, float* f_lo,
                                   float* f_hi, float* z_lo, float* z_hi, float* z_pred,
                                   const float* f_mid, float* pts_out, void* stream) {
  ISO_REQUIRE(n >= 0, ISO_ERR_INVALID, "iso_raymarch_secant: bad count");
  if (n == 0) return ISO_OK;
  ISO_REQUIRE(cam && dirs && f_lo && f_hi && z_lo && z_hi && z_pred && pts_out, ISO_ERR_INVALID,
              "iso_raymarch_secant: null argument");
  hipLaunchKernelGGL(k_secant_next, dim3(iso_div_up(n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                     cam, dirs, n, f_lo, f_hi, z_lo, z_hi, z_pred, f_mid, pts_out);
  ISO_CHECK_LAUNCH("iso_raymarch_secant");
  return ISO_OK;
}
