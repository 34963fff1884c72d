/*
 * oracle_splat.c -- CPU restatement of the reference's point-splat rasteriser.
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.
 *
 * Pinned against the reference's own DSS/csrc/rasterize_points_cpu.cpp compiled as-is
 * into oracle/_ref/ (see oracle/Makefile, tests/test_oracle_ref_splat.py).
 *
 * Plain C, single thread, host pointers.  Arithmetic is written in the same order as
 * the reference's CPU code so the results are bit-identical to it
 * (build with -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* rasterize_points_cpu.cpp:10-14 / rasterization_utils.cuh:8-11 */
static float pix_to_ndc(int i, int S) { return -1 + (2 * i + 1.0f) / S; }

/* rasterize_points_cpu.cpp:22-25 */
static float qvalue(float dx, float dy, float a, float b, float c) {
  return a * dx * dx + b * dx * dy + c * dy * dy;
}

/*
 * Forward (RasterizePointsNaiveCpu, rasterize_points_cpu.cpp:27-144; CUDA twin
 * rasterize_points.cu:65-211).
 *   bbox_or = 0 : CPU reject rule  `|dx|>rx && |dy|>ry`  (rasterize_points_cpu.cpp:99)
 *   bbox_or = 1 : CUDA reject rule `|dx|>rx || |dy|>ry`  (rasterize_points.cu:92) -- canonical
 * Per pixel the K smallest (z, idx) pairs are kept (the reference's max-heap of
 * (z, idx, q) tuples, :85-112), written in ascending order, then entries with
 * z - z0 > depth_thres are reset to -1 (:125-139).  occupancy = 1 when slot 0 is filled.
 */
void oracle_splat_forward(const float* pts, const float* ellipse, const float* cutoff,
                          const float* radii, const int64_t* first_idx,
                          const int64_t* num_pts, int N, float depth_thres, int S, int K,
                          int bbox_or, int32_t* idx, float* zbuf, float* qv, float* occ) {
  float* kz = (float*)malloc(sizeof(float) * (K + 1));
  float* kq = (float*)malloc(sizeof(float) * (K + 1));
  int32_t* ki = (int32_t*)malloc(sizeof(int32_t) * (K + 1));
  for (int n = 0; n < N; ++n) {
    const int p0 = (int)first_idx[n], p1 = p0 + (int)num_pts[n];
    for (int yi = 0; yi < S; ++yi) {
      const float yf = pix_to_ndc(S - 1 - yi, S);
      for (int xi = 0; xi < S; ++xi) {
        const float xf = pix_to_ndc(S - 1 - xi, S);
        int cnt = 0;
        for (int p = p0; p < p1; ++p) {
          const float px = pts[p * 3], py = pts[p * 3 + 1], pz = pts[p * 3 + 2];
          if (pz < 0) continue;
          const float dx = xf - px, dy = yf - py;
          const float rx = radii[p * 2], ry = radii[p * 2 + 1];
          if (bbox_or) {
            if (fabsf(dx) > rx || fabsf(dy) > ry) continue;
          } else {
            if (fabsf(dx) > rx && fabsf(dy) > ry) continue;
          }
          const float q = qvalue(dx, dy, ellipse[p * 3], ellipse[p * 3 + 1], ellipse[p * 3 + 2]);
          if (q > cutoff[p]) continue;
          /* sorted insert by (z, idx); drop the largest when more than K */
          int j = cnt;
          while (j > 0 && (kz[j - 1] > pz || (kz[j - 1] == pz && ki[j - 1] > p))) {
            kz[j] = kz[j - 1]; ki[j] = ki[j - 1]; kq[j] = kq[j - 1];
            --j;
          }
          kz[j] = pz; ki[j] = p; kq[j] = q;
          if (cnt < K) ++cnt;
        }
        const int64_t o = (((int64_t)n * S + yi) * S + xi) * K;
        for (int k = 0; k < K; ++k) {
          idx[o + k] = k < cnt ? ki[k] : -1;
          zbuf[o + k] = k < cnt ? kz[k] : -1.0f;
          qv[o + k] = k < cnt ? kq[k] : -1.0f;
        }
        occ[((int64_t)n * S + yi) * S + xi] = 0.0f;
        if (idx[o] >= 0 && zbuf[o] >= 0) {
          occ[((int64_t)n * S + yi) * S + xi] = 1.0f;
          const float z0 = zbuf[o];
          for (int k = 1; k < K; ++k) {
            if ((zbuf[o + k] - z0) > depth_thres) {
              idx[o + k] = -1; zbuf[o + k] = -1.0f; qv[o + k] = -1.0f;
            }
          }
        }
      }
    }
  }
  free(kz); free(kq); free(ki);
}

/* sign-preserving clamp, rasterization_utils.cuh:38-43 */
static float eps_denomf(float x, float eps) {
  float s = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 1.f);
  float a = fabsf(x);
  return s * (a < eps ? eps : a);
}

/*
 * Occupancy backward.  grad_xy[p] += d / denom(|d|^2) * grad_occ[pixel], d = pixel_ndc - p_xy,
 * summed in pixel order (n, yi, xi) like the reference's CPU loop.
 *   mode 0: RasterizePointsOccBackwardCpu      (rasterize_points_cpu.cpp:380-477)
 *           support: reject when |dx|>rx*s && |dy|>ry*s ; denom = max(|d|^2, 1e-8)
 *   mode 1: RasterizePointsOccBackwardCudaKernel (rasterize_points.cu:673-760)
 *           support: reject when |dx|>rx*s || |dy|>ry*s ; denom = eps_denom(|d|^2, 1e-10)
 *   mode 2: RasterizePointsBackwardCudaFastKernel (rasterize_points_backward.cu:85-178), the
 *           default training path: support = disc |d|^2 <= rs[n]^2 ; denom as mode 1.
 *           `visible` (P) selects the points that take part (rasterizer.py:850-862); NULL = all.
 * In every mode: skip points with z<0 or |x|>1 or |y|>1; skip when grad>0 and the pixel lies
 * outside the un-scaled rect (|dx|>rx || |dy|>ry).
 */
void oracle_occ_backward(const float* pts, const float* radii, const float* grad_occ,
                         const int64_t* first_idx, const int64_t* num_pts, int N, int S,
                         float radii_s, const float* rs, const uint8_t* visible, int mode,
                         float* grad_xy /* (P,2), zeroed by the caller */) {
  for (int n = 0; n < N; ++n) {
    const int p0 = (int)first_idx[n], p1 = p0 + (int)num_pts[n];
    const float r2 = (mode == 2) ? rs[n] * rs[n] : 0.f;
    for (int yi = 0; yi < S; ++yi) {
      const float yf = pix_to_ndc(S - 1 - yi, S);
      for (int xi = 0; xi < S; ++xi) {
        const float xf = pix_to_ndc(S - 1 - xi, S);
        const float g = grad_occ[((int64_t)n * S + yi) * S + xi];
        if (g == 0.0f) continue;
        for (int p = p0; p < p1; ++p) {
          if (visible && !visible[p]) continue;
          const float px = pts[p * 3], py = pts[p * 3 + 1], pz = pts[p * 3 + 2];
          if (pz < 0 || fabsf(py) > 1.0 || fabsf(px) > 1.0) continue;
          const float dx = xf - px, dy = yf - py;
          const float dist2 = dx * dx + dy * dy;
          float denom;
          if (mode == 2) {
            const float rx = radii[p * 2], ry = radii[p * 2 + 1];
            if (dist2 > r2) continue;
            const int outside = (fabsf(dx) > rx) || (fabsf(dy) > ry);
            if (g > 0.0f && outside) continue;
            denom = eps_denomf(dist2, 1e-10f);
          } else {
            const float rx = radii[p * 2] * radii_s, ry = radii[p * 2 + 1] * radii_s;
            const int outside = (fabsf(dx) > rx / radii_s) || (fabsf(dy) > ry / radii_s);
            if (mode == 0) {
              if (g > 0.0f && outside) continue;
              if (fabsf(dx) > rx && fabsf(dy) > ry) continue;
              denom = dist2 > 1e-8f ? dist2 : 1e-8f;
            } else {
              if (fabsf(dx) > rx || fabsf(dy) > ry) continue;
              if (g > 0.0f && outside) continue;
              denom = eps_denomf(dist2, 1e-10f);
            }
          }
          grad_xy[p * 2] += dx / denom * g;
          grad_xy[p * 2 + 1] += dy / denom * g;
        }
      }
    }
  }
}

/* RasterizeZbufBackwardCpu (rasterize_points_cpu.cpp:479-514): z_grad[idx] += grad_zbuf,
 * zeros skipped, stop at the first idx < 0; pixel order (n, y, x, k). */
void oracle_zbuf_backward(const int32_t* idx, const float* grad_zbuf, int N, int S, int K,
                          float* z_grad /* (P), zeroed by the caller */) {
  const int64_t npix = (int64_t)N * S * S;
  for (int64_t i = 0; i < npix; ++i) {
    for (int k = 0; k < K; ++k) {
      const float g = grad_zbuf[i * K + k];
      if (g == 0.0f) continue;
      const int32_t p = idx[i * K + k];
      if (p < 0) break;
      z_grad[p] += g;
    }
  }
}
