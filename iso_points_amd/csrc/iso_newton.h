// One clamped Newton move toward the zero level set, shared by every SDF
// variant.  Mirrors DSS/models/levelset_sampling.py:336-342 operation by
// operation (f32, no FMA contraction):
//   ssg  = gx^2+gy^2+gz^2
//   m    = f * (g / sdeno(ssg, 1e-17))
//   m    = m / max(|m|, 1e-15) * min(|m|, 0.1)        (F.normalize * clamp_max)
//   p   -= m
#pragma once
#include "iso_common.h"

__device__ __forceinline__ void iso_newton_move(float f, float gx, float gy,
                                                float gz, float& px, float& py,
                                                float& pz) {
  float ssg = (gx * gx + gy * gy) + gz * gz;
  float den = iso_eps_denom(ssg, 1.0e-17f);
  float mx = f * (gx / den), my = f * (gy / den), mz = f * (gz / den);
  float mn = sqrtf((mx * mx + my * my) + mz * mz);
  float nd = mn > 1e-15f ? mn : 1e-15f;
  float sc = mn < 0.1f ? mn : 0.1f;
  px = px - (mx / nd) * sc;
  py = py - (my / nd) * sc;
  pz = pz - (mz / nd) * sc;
}

// One clamped sphere-tracing advance along a unit ray direction.  Mirrors
// DSS/models/levelset_sampling.py:768-776 (SphereTracing.project_points):
//   m = (alpha * f) * d;  m = m / max(|m|, 1e-15) * min(|m|, 0.1);  q = p + m
//   the advance is kept only while |q| < radius + padding (`bound`)
// Returns true when the advanced point is still inside the bounding sphere.
__device__ __forceinline__ bool iso_trace_move(float f, float dx, float dy, float dz, float alpha,
                                               float bound, float& px, float& py, float& pz) {
  const float s = alpha * f;
  float mx = s * dx, my = s * dy, mz = s * dz;
  float mn = sqrtf((mx * mx + my * my) + mz * mz);
  float nd = mn > 1e-15f ? mn : 1e-15f;
  float sc = mn < 0.1f ? mn : 0.1f;
  const float qx = px + (mx / nd) * sc, qy = py + (my / nd) * sc, qz = pz + (mz / nd) * sc;
  const bool inside = sqrtf((qx * qx + qy * qy) + qz * qz) < bound;
  if (inside) { px = qx; py = qy; pz = qz; }
  return inside;
}

// What every step kernel does with the value f and gradient g of point `idx` (Args = SirenArgs /
// IdrArgs: same field names).  Returns true when the point stays on the active list.
//   eval_only : store sdf (+ gradient when grad_out is given)
//   dirs      : sphere tracing (levelset_sampling.py:735-779): store the value, mask = |f| <= tol_valid,
//               still active = |f| > tol (the caller passes 0.1 * proj_tolerance) and inside the sphere
//   otherwise : Newton projection (levelset_sampling.py:309-344): normals = g, mask = |f| <= tol
// HAVE_POS: the caller still holds the point's position (q = a.pts[idx], the values it evaluated at) and passes it in
// instead of having it loaded again here.
template <class Args, bool HAVE_POS = false>
__device__ __forceinline__ bool iso_step_finish(const Args& a, int64_t idx, float f, float gx,
                                                float gy, float gz, float qx = 0.f, float qy = 0.f, float qz = 0.f) {
  if (a.eval_only) {
    a.sdf_out[idx] = f;
    if (a.grad_out) { a.grad_out[idx * 3] = gx; a.grad_out[idx * 3 + 1] = gy; a.grad_out[idx * 3 + 2] = gz; }
    return false;
  }
  if (a.dirs) {
    a.sdf_out[idx] = f;
    a.mask[idx] = fabsf(f) <= a.tol_valid ? 1 : 0;
    if (fabsf(f) > a.tol && a.do_move) {
      if (!HAVE_POS) { qx = a.pts[idx * 3]; qy = a.pts[idx * 3 + 1]; qz = a.pts[idx * 3 + 2]; }
      if (iso_trace_move(f, a.dirs[idx * 3], a.dirs[idx * 3 + 1], a.dirs[idx * 3 + 2], a.alpha, a.bound,
                         qx, qy, qz)) {
        a.pts[idx * 3] = qx; a.pts[idx * 3 + 1] = qy; a.pts[idx * 3 + 2] = qz;
        return true;
      }
    }
    return false;
  }
  a.normals[idx * 3] = gx; a.normals[idx * 3 + 1] = gy; a.normals[idx * 3 + 2] = gz;
  const bool active = fabsf(f) > a.tol;
  a.mask[idx] = active ? 0 : 1;
  if (active && a.do_move) {
    if (!HAVE_POS) { qx = a.pts[idx * 3]; qy = a.pts[idx * 3 + 1]; qz = a.pts[idx * 3 + 2]; }
    iso_newton_move(f, gx, gy, gz, qx, qy, qz);
    a.pts[idx * 3] = qx; a.pts[idx * 3 + 1] = qy; a.pts[idx * 3 + 2] = qz;
    return true;
  }
  return false;
}
