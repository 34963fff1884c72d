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
