// C-ABI harness around the REFERENCE's own CPU rasteriser.  This file is ours; the
// reference translation unit (DSS/csrc/rasterize_points_cpu.cpp) is compiled where it lies
// under /root/reference by oracle/Makefile and linked with this harness into
// oracle/_ref/libdss_ref_cpu.so (git-ignored, never copied into the repo).
// TEST INFRASTRUCTURE: used to pin oracle_splat.c and, optionally, as bench.py's
// cpu_baseline of kind "reference".
//
// The prototypes below are the public signatures declared in the reference's
// DSS/csrc/rasterize_points.h:18-27, :354-362 (OccBackwardCpu) and :389 (ZbufBackwardCpu).
#include <torch/torch.h>
#include <cstring>
#include <tuple>

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor> RasterizePointsNaiveCpu(
    const torch::Tensor& points, const torch::Tensor& ellipse_params,
    const torch::Tensor& cutoff_thres, const torch::Tensor& radii,
    const torch::Tensor& cloud_to_packed_first_idx, const torch::Tensor& num_points_per_cloud,
    const float depth_merging_thres, const int image_size, const int points_per_pixel);

torch::Tensor RasterizePointsOccBackwardCpu(const torch::Tensor& points, const torch::Tensor& radii,
                                            const torch::Tensor& grad_occ,
                                            const torch::Tensor& cloud_to_packed_first_idx,
                                            const torch::Tensor& num_points_per_cloud,
                                            const float radii_s, const float depth_merging_thres);

void RasterizeZbufBackwardCpu(const at::Tensor& idx, const at::Tensor& grad_zbuf,
                              at::Tensor& point_z_grad);

static torch::Tensor f32(const float* p, std::initializer_list<int64_t> shape) {
  return torch::from_blob(const_cast<float*>(p), shape, torch::kFloat32);
}
static torch::Tensor i64(const int64_t* p, std::initializer_list<int64_t> shape) {
  return torch::from_blob(const_cast<int64_t*>(p), shape, torch::kInt64);
}

extern "C" int ref_splat_forward(const float* pts, const float* ellipse, const float* cutoff,
                                 const float* radii, const int64_t* first_idx,
                                 const int64_t* num_pts, int64_t P, int N, float depth_thres,
                                 int S, int K, int32_t* idx, float* zbuf, float* qv, float* occ) {
  try {
    auto r = RasterizePointsNaiveCpu(f32(pts, {P, 3}), f32(ellipse, {P, 3}), f32(cutoff, {P}),
                                     f32(radii, {P, 2}), i64(first_idx, {N}), i64(num_pts, {N}),
                                     depth_thres, S, K);
    const int64_t npix = (int64_t)N * S * S;
    std::memcpy(idx, std::get<0>(r).contiguous().data_ptr<int32_t>(), npix * K * 4);
    std::memcpy(zbuf, std::get<1>(r).contiguous().data_ptr<float>(), npix * K * 4);
    std::memcpy(qv, std::get<2>(r).contiguous().data_ptr<float>(), npix * K * 4);
    std::memcpy(occ, std::get<3>(r).contiguous().data_ptr<float>(), npix * 4);
    return 0;
  } catch (...) {
    return -1;
  }
}

extern "C" int ref_occ_backward(const float* pts, const float* radii, const float* grad_occ,
                                const int64_t* first_idx, const int64_t* num_pts, int64_t P, int N,
                                int S, float radii_s, float depth_thres, float* grad_xy) {
  try {
    auto g = RasterizePointsOccBackwardCpu(f32(pts, {P, 3}), f32(radii, {P, 2}),
                                           f32(grad_occ, {N, S, S}), i64(first_idx, {N}),
                                           i64(num_pts, {N}), radii_s, depth_thres);
    std::memcpy(grad_xy, g.contiguous().data_ptr<float>(), P * 2 * 4);
    return 0;
  } catch (...) {
    return -1;
  }
}

extern "C" int ref_zbuf_backward(const int32_t* idx, const float* grad_zbuf, int64_t P, int N, int S,
                                 int K, float* z_grad) {
  try {
    auto idx_t = torch::from_blob(const_cast<int32_t*>(idx), {N, S, S, K}, torch::kInt32);
    auto out = torch::zeros({P, 1}, torch::kFloat32);
    RasterizeZbufBackwardCpu(idx_t, f32(grad_zbuf, {N, S, S, K}), out);
    std::memcpy(z_grad, out.data_ptr<float>(), P * 4);
    return 0;
  } catch (...) {
    return -1;
  }
}
