// EWA surface splatting for gfx950: per-point set-up, tile binning, per-tile
// rasterisation, compositing and the (deterministic) backward pass.
//
// Reference semantics
//   set-up   : SurfaceSplatting._get_per_point_info and helpers,
//              DSS/core/rasterizer.py:344-563; filter_renderable :184-254
//   forward  : _C.splat_points -> RasterizePoints{Naive,Coarse,Fine}
//              DSS/csrc/rasterize_points.cu:65-211,293-430,506-596 (CPU twin
//              rasterize_points_cpu.cpp:27-144)
//   composite: SurfaceSplattingRenderer.forward, DSS/core/renderer.py:53-78
//   backward : EllipticalRasterizer.backward (rasterizer.py:841-968) +
//              RasterizePointsBackwardCudaFastKernel
//              (rasterize_points_backward.cu:85-178) + ZbufBackwardKernel
//              (rasterize_points.cu:823-846)
//
// Design (MI355X): the reference bins 512-point chunks into <=21x21 bins with 64
// workgroups and then lets every pixel scan all M = max(1e4,P) slots of its
// bin (an N*B*B*M int table: 4 GB at 1M points).  Here points are binned into
// 16x16-pixel tiles by a count -> scan -> fill counting sort (one int per
// point-tile pair), and one 256-lane workgroup per tile streams its candidate
// list through LDS in 256-entry chunks (every lane reads the same LDS word:
// broadcast, conflict free) while each lane keeps the K front-most (z, idx)
// hits of its own pixel in registers.  4096 tiles at 512^2 x 4 views fill the
// 256 CUs 16 times over.  The K-best rule is the reference CPU's
// lexicographic (z, idx) order, so per-pixel index lists are deterministic and
// bit-identical to the compiled reference no matter in which order the atomics
// of the fill pass land.
//
// Backward is point-major: every visible point walks the pixels of its
// support in image order and accumulates in registers -- no float atomics, so
// gradients are bit-stable (the reference's gpuAtomicAdd order is not).
#include <float.h>
#include "iso_common.h"
#include "splat_frame.h"

#pragma clang fp contract(off)

// Riders (workgroups past the kernel's own grid, see iso_splat_backward): work of the z pass that depends on nothing the
// host kernel of the launch does, only on launches before it.
struct ZScale;
struct ZRider {                  // k_splat_backward's: max |grad_zbuf| (k_z_absmax's pass); k_splat_backward_heavy's: the
  const float* gz; int64_t n;    // conversion of the fixed-point sums (k_z_finish_clouds's pass)
  ZScale* zs; int blocks;        // blocks == 0: no riders
  const long long* acc; float* grad; int terms_log2;
};
__device__ void z_absmax_body(const float* __restrict__ gz, int64_t n, ZScale* __restrict__ zs, int block, int nblocks);
__device__ void z_finish_body(const long long* __restrict__ acc, const ZScale* __restrict__ zs, const int64_t* __restrict__ first,
                              const int64_t* __restrict__ num, int n_clouds, float* __restrict__ grad, int terms_log2,
                              int block, int nblocks);

namespace {

__device__ __forceinline__ float esqrt_arg(float x) {  // eps_sqrt, mathHelper.py:20-25
  float a = fabsf(x);
  return a < 1e-17f ? 1e-17f : a;
}

// ---------------------------------------------------------------- visibility
// flags[n*P+i] = 1 when point i is renderable in view n (rasterizer.py:184-254)
__global__ void k_view_flags(const float* __restrict__ pts, const float* __restrict__ nrm,
                             const float* __restrict__ views, int32_t* __restrict__ flags,
                             int64_t P, float znear, float zfar, int backface) {
  const int n = blockIdx.y;
  const float* V = views + n * 16;  // row-vector convention: p_view = [p,1] @ V
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P;
       i += (int64_t)gridDim.x * blockDim.x) {
    float x = pts[i * 3], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
    float zv = ((x * V[2] + y * V[6]) + z * V[10]) + V[14];
    bool ok = (zv >= znear) && (zv <= zfar);
    if (backface) {
      float nx = nrm[i * 3], ny = nrm[i * 3 + 1], nz = nrm[i * 3 + 2];
      float nzv = (nx * V[2] + ny * V[6]) + nz * V[10];
      ok = ok && (nzv < 0.f);
    }
    flags[(int64_t)n * P + i] = ok ? 1 : 0;
  }
}

// out[off[e]] = in[e % P] for flagged e (stable: packed order = view-major, point ascending)
__global__ void k_compact_rows(const float* __restrict__ in, const int32_t* __restrict__ flags,
                               const int32_t* __restrict__ off, float* __restrict__ out,
                               int64_t P, int64_t total, int U) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    if (flags[e]) {
      const float* s = in + (e % P) * U;
      float* d = out + (int64_t)off[e] * U;
      for (int u = 0; u < U; ++u) d[u] = s[u];
    }
  }
}

// h = clamp(0.5 * max_{6 nn} d2, 5e-5, 0.01)   (rasterizer.py:375-386)
__global__ void k_vrk_h(const float* __restrict__ dists /*(N,Pmax,7)*/,
                        const int64_t* __restrict__ first, const int64_t* __restrict__ num,
                        const int64_t* __restrict__ cloud_num, float* __restrict__ h, int64_t pmax) {
  const int n = blockIdx.y;
  const int64_t len = num[n];
  const int64_t clen = cloud_num ? cloud_num[n] : len;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float* d = dists + ((int64_t)n * pmax + i) * 7;
    float m = -FLT_MAX;
#pragma unroll
    for (int k = 1; k < 7; ++k) {
      float v = (clen < 7) ? 1e-3f : d[k];
      m = fmaxf(m, v);
    }
    float hv = 0.5f * m;
    hv = fminf(fmaxf(hv, 5e-5f), 0.01f);
    h[first[n] + i] = hv;
  }
}

// ---------------------------------------------------------------- per-point EWA set-up
struct SetupOut {
  float* ndc;      // (P,3)
  float* ellipse;  // (P,3)
  float* cutoff;   // (P)
  float* radii;    // (P,2)
  float* scaler;   // (P)
};

// One point in one view (rasterizer.py:344-563; the arithmetic both set-up kernels share).
struct SetupRes { float ndc[3], el[3], rad[2], scaler; };

__device__ __forceinline__ SetupRes splat_setup_point(float x, float y, float z, float nx, float ny, float nz,
                                                      float hk, const float* __restrict__ V,
                                                      const float* __restrict__ M, int S, float sigma,
                                                      float cutoffC) {
  // [p,1] @ M columns 0,1,3 and view depth
  const float xv = ((x * M[0] + y * M[4]) + z * M[8]) + M[12];
  const float yv = ((x * M[1] + y * M[5]) + z * M[9]) + M[13];
  const float t = ((x * M[3] + y * M[7]) + z * M[11]) + M[15];
  const float zv = ((x * V[2] + y * V[6]) + z * V[10]) + V[14];
  const float t2 = iso_eps_denom(t * t, 1e-17f);
  const float td = iso_eps_denom(t, 1e-17f);
  const float j00 = 1.0f / td;
  const float j30 = -1.0f / t2 * xv, j31 = -1.0f / t2 * yv;
  // WJk = M[:3,:] @ Jk  (3x2)
  float w0[3], w1[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    w0[r] = M[r * 4 + 0] * j00 + M[r * 4 + 3] * j30;
    w1[r] = M[r * 4 + 1] * j00 + M[r * 4 + 3] * j31;
  }
  // tangent frame: u0 = normalize(n x (n + e)), u1 = normalize(n x u0), e = axis least
  // aligned with n (a deterministic instance of rasterizer.py:395-397)
  float ex = 0.f, ey = 0.f, ez = 0.f;
  const float ax = fabsf(nx), ay = fabsf(ny), az = fabsf(nz);
  if (ax <= ay && ax <= az) ex = 1.f; else if (ay <= az) ey = 1.f; else ez = 1.f;
  const float mx = nx + ex, my = ny + ey, mz = nz + ez;
  float ux = ny * mz - nz * my, uy = nz * mx - nx * mz, uz = nx * my - ny * mx;
  float un = sqrtf((ux * ux + uy * uy) + uz * uz);
  un = un > 1e-12f ? un : 1e-12f;
  ux /= un; uy /= un; uz /= un;
  float vx = ny * uz - nz * uy, vy = nz * ux - nx * uz, vz = nx * uy - ny * ux;
  float vn = sqrtf((vx * vx + vy * vy) + vz * vz);
  vn = vn > 1e-12f ? vn : 1e-12f;
  vx /= vn; vy /= vn; vz /= vn;
  // Mk = Sk @ WJk (2x2);  Vk = h * Mk^T Mk
  const float m00 = (ux * w0[0] + uy * w0[1]) + uz * w0[2];
  const float m01 = (ux * w1[0] + uy * w1[1]) + uz * w1[2];
  const float m10 = (vx * w0[0] + vy * w0[1]) + vz * w0[2];
  const float m11 = (vx * w1[0] + vy * w1[1]) + vz * w1[2];
  const float ps = 2.0f / (float)S;
  const float lp = sigma * (ps * ps);
  const float g00 = hk * (m00 * m00 + m10 * m10) + lp;
  const float g01 = hk * (m00 * m01 + m10 * m11);
  const float g11 = hk * (m01 * m01 + m11 * m11) + lp;
  const float detM = m00 * m11 - m01 * m10;
  // det(h M^T M + lp I) = h^2 det(M)^2 + lp h |M|_F^2 + lp^2: all terms positive, so no
  // cancellation (the textbook g00*g11 - g01^2 loses ~cond(G) digits on grazing splats)
  const float fro = (m00 * m00 + m10 * m10) + (m01 * m01 + m11 * m11);
  const float detG = (hk * hk) * (detM * detM) + (lp * hk * fro + lp * lp);
  const float a = g11 / detG, c = g00 / detG, b = (-g01 / detG) + (-g01 / detG);
  const float den = iso_eps_denom(4.0f * a * c - b * b, 1e-17f);
  SetupRes o;
  o.rad[1] = sqrtf(esqrt_arg(4.0f * a * cutoffC / den));
  o.rad[0] = sqrtf(esqrt_arg(4.0f * c * cutoffC / den));
  o.scaler = fabsf(detM) / iso_eps_denom(sqrtf(esqrt_arg(detG * 4.0f * 3.14159265358979323846f * 3.14159265358979323846f)), 1e-17f);
  o.ndc[0] = xv / t; o.ndc[1] = yv / t; o.ndc[2] = zv;
  o.el[0] = a; o.el[1] = b; o.el[2] = c;
  return o;
}

__global__ void k_splat_setup(const float* __restrict__ pts, const float* __restrict__ nrm,
                              const float* __restrict__ h, const int64_t* __restrict__ first,
                              const int64_t* __restrict__ num, const float* __restrict__ views,
                              const float* __restrict__ projs, int S, float sigma, float cutoffC,
                              SetupOut o) {
  const int n = blockIdx.y;
  const float* V = views + n * 16;
  const float* M = projs + n * 16;  // full world->NDC, row-vector convention
  const int64_t len = num[n], base = first[n];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = base + i;
    const SetupRes r = splat_setup_point(pts[p * 3], pts[p * 3 + 1], pts[p * 3 + 2], nrm[p * 3], nrm[p * 3 + 1],
                                         nrm[p * 3 + 2], h[p], V, M, S, sigma, cutoffC);
    o.ndc[p * 3] = r.ndc[0]; o.ndc[p * 3 + 1] = r.ndc[1]; o.ndc[p * 3 + 2] = r.ndc[2];
    o.ellipse[p * 3] = r.el[0]; o.ellipse[p * 3 + 1] = r.el[1]; o.ellipse[p * 3 + 2] = r.el[2];
    o.cutoff[p] = cutoffC;
    o.radii[p * 2] = r.rad[0]; o.radii[p * 2 + 1] = r.rad[1];
    o.scaler[p] = r.scaler;
  }
}

// ---------------------------------------------------------------- fused filter + compaction + set-up
// The cycle's front end works on the UNFILTERED cloud: a point's renderable views are bits of
// mask[i] (iso_splat_view_mask), its bandwidths h[v*P+i] (iso_splat_h_fused).  The packed order of
// the reference (view-major, points ascending: rasterizer.py:597-618) needs, for point i in view v,
// the number of renderable points before it: chunk counts (k_mask_chunk_count) -> one scan
// (k_mask_chunk_scan, also first_idx / num_points on the device: no host read anywhere) -> ranks
// inside the chunk from wave ballots here.
constexpr int kChunk = 1024;   // points per workgroup: 256 lanes x 4 consecutive points

__global__ __launch_bounds__(256) void k_mask_chunk_count(const int32_t* __restrict__ mask, int64_t P, int n_views,
                                                          int n_chunks, int32_t* __restrict__ chunk_cnt /*(8, n_chunks)*/) {
  __shared__ int s_cnt[8];
  if (threadIdx.x < 8) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const int64_t i0 = (int64_t)blockIdx.x * kChunk + threadIdx.x * 4;
  int m[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) m[k] = i0 + k < P ? mask[i0 + k] : 0;
#pragma unroll
  for (int v = 0; v < 8; ++v) {
    if (v < n_views) {
      int c = ((m[0] >> v) & 1) + ((m[1] >> v) & 1) + ((m[2] >> v) & 1) + ((m[3] >> v) & 1);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
      if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_cnt[v], c);
    }
  }
  __syncthreads();
  if (threadIdx.x < n_views) chunk_cnt[threadIdx.x * n_chunks + blockIdx.x] = s_cnt[threadIdx.x];
}

// The renderable mask of rasterizer.py:184-254 (k_view_mask in bricks.hip: same expressions) AND the chunk counts
// in one pass: a workgroup owns the kChunk consecutive points the front end's workgroup of the same index will own.
__global__ __launch_bounds__(256) void k_view_mask_chunks(const float* __restrict__ pts, const float* __restrict__ nrm,
                                                          const float* __restrict__ views, int n_views, int64_t P,
                                                          float znear, float zfar, int backface, int n_chunks,
                                                          int32_t* __restrict__ mask, int32_t* __restrict__ chunk_cnt) {
  __shared__ int s_cnt[8];
  if (threadIdx.x < 8) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const int64_t i0 = (int64_t)blockIdx.x * kChunk + threadIdx.x * 4;
  int local[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t i = i0 + k;
    if (i >= P) break;
    const float x = pts[i * 3], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
    float nx = 0.f, ny = 0.f, nz = 0.f;
    if (backface) { nx = nrm[i * 3]; ny = nrm[i * 3 + 1]; nz = nrm[i * 3 + 2]; }
    int m = 0;
#pragma unroll
    for (int v = 0; v < 8; ++v) {
      if (v < n_views) {
        const float* V = views + v * 16;                    // row-vector convention: p_view = [p,1] @ V
        const float zv = ((x * V[2] + y * V[6]) + z * V[10]) + V[14];
        bool ok = (zv >= znear) && (zv <= zfar);
        if (backface) ok = ok && (((nx * V[2] + ny * V[6]) + nz * V[10]) < 0.f);
        if (ok) { m |= 1 << v; ++local[v]; }
      }
    }
    mask[i] = m;
  }
#pragma unroll
  for (int v = 0; v < 8; ++v) {
    if (v < n_views) {
      int c = local[v];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
      if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_cnt[v], c);
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < n_views) chunk_cnt[threadIdx.x * n_chunks + blockIdx.x] = s_cnt[threadIdx.x];
}

// one workgroup: exclusive scan of every view's chunk counts in place; first_idx / num_points (i64,
// the layout _C.splat_points takes) and the int32 view totals
__global__ __launch_bounds__(1024) void k_mask_chunk_scan(int32_t* __restrict__ chunk_cnt, int n_chunks, int n_views,
                                                         int64_t* __restrict__ first, int64_t* __restrict__ num,
                                                         int32_t* __restrict__ view_total) {
  __shared__ int s_w[16];
  __shared__ int s_tot[8];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int v = 0; v < n_views; ++v) {
    int carry = 0;
    int32_t* row = chunk_cnt + (int64_t)v * n_chunks;
    for (int c0 = 0; c0 < n_chunks; c0 += 1024) {
      const int i = c0 + threadIdx.x;
      const int val = i < n_chunks ? row[i] : 0;
      int inc = val;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
      if (lane == 63) s_w[w] = inc;
      __syncthreads();
      int base = 0, tot = 0;
#pragma unroll
      for (int k = 0; k < 16; ++k) { const int sv = s_w[k]; if (k < w) base += sv; tot += sv; }
      if (i < n_chunks) row[i] = carry + base + inc - val;
      carry += tot;
      __syncthreads();
    }
    if (threadIdx.x == 0) s_tot[v] = carry;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t run = 0;
    for (int v = 0; v < n_views; ++v) { first[v] = run; num[v] = s_tot[v]; view_total[v] = s_tot[v]; run += s_tot[v]; }
    for (int v = n_views; v < 8; ++v) view_total[v] = 0;
  }
}

struct FrontOut {
  float* ndc; float* ellipse; float* cutoff; float* radii; float* scaler;
  float* feat;      // (cap, C) packed features or null
  int32_t* src;     // (cap) original point of a packed row, or null
  uint8_t* vis0;    // (cap) "visible" flags of the backward pass, cleared here row by row (or null): no fill launch
  int64_t cap;      // rows the arrays hold (<= 0: whatever comes); rows beyond are dropped and *overflow is set
  int32_t* overflow;
};

__global__ __launch_bounds__(256) void k_splat_front(const float* __restrict__ pts, const float* __restrict__ nrm,
                                                     const float* __restrict__ feat_in, int C,
                                                     const int32_t* __restrict__ mask, const float* __restrict__ h,
                                                     int64_t P, const int32_t* __restrict__ chunk_off, int n_chunks,
                                                     const int64_t* __restrict__ first, const float* __restrict__ views,
                                                     const float* __restrict__ projs, int n_views, int S, float sigma,
                                                     float cutoffC, int feat_from_normal, FrontOut o) {
  __shared__ int s_w[8][4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t i0 = (int64_t)blockIdx.x * kChunk + threadIdx.x * 4;
  int m[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) m[k] = i0 + k < P ? mask[i0 + k] : 0;
  // rank of my first point among the chunk's renderable points, per view
  int rank[8];
#pragma unroll
  for (int v = 0; v < 8; ++v) {
    rank[v] = 0;
    if (v < n_views) {
      const int c = ((m[0] >> v) & 1) + ((m[1] >> v) & 1) + ((m[2] >> v) & 1) + ((m[3] >> v) & 1);
      int inc = c;
#pragma unroll
      for (int o2 = 1; o2 < 64; o2 <<= 1) { const int t = __shfl_up(inc, o2); if (lane >= o2) inc += t; }
      if (lane == 63) s_w[v][w] = inc;
      rank[v] = inc - c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int v = 0; v < 8; ++v)
    if (v < n_views)
      for (int k = 0; k < w; ++k) rank[v] += s_w[v][k];
  // The rows of a chunk in one view are contiguous in the packed arrays: they are staged in LDS view by view and
  // written out as full lines (a lane storing its own 12-byte rows touches every 128-byte line three times: 350 MB
  // of partial writes for 168 MB of records at 1 M points x 4 views).
  __shared__ float s_ndc[kChunk * 3], s_el[kChunk * 3], s_rad[kChunk * 2], s_sc[kChunk], s_ft[kChunk * 3];
  __shared__ int s_src[kChunk];
  const bool any = (m[0] | m[1] | m[2] | m[3]) != 0;
  float x[4], y[4], z[4], nx[4], ny[4], nz[4], f[4][3];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    x[k] = y[k] = z[k] = nx[k] = ny[k] = nz[k] = 0.f;
    f[k][0] = f[k][1] = f[k][2] = 0.f;
    if (!m[k]) continue;
    const int64_t i = i0 + k;
    x[k] = pts[i * 3]; y[k] = pts[i * 3 + 1]; z[k] = pts[i * 3 + 2];
    nx[k] = nrm[i * 3]; ny[k] = nrm[i * 3 + 1]; nz[k] = nrm[i * 3 + 2];
    if (o.feat && feat_from_normal) {               // 0.5 (normalize(n) + 1): the cycle's shading-free features
      float nn = sqrtf((nx[k] * nx[k] + ny[k] * ny[k]) + nz[k] * nz[k]);
      nn = nn > 1e-12f ? nn : 1e-12f;
      f[k][0] = 0.5f * (nx[k] / nn + 1.0f); f[k][1] = 0.5f * (ny[k] / nn + 1.0f); f[k][2] = 0.5f * (nz[k] / nn + 1.0f);
    } else if (o.feat && C <= 3) {
      for (int c = 0; c < C; ++c) f[k][c] = feat_in[i * C + c];
    }
  }
  const bool stage_feat = o.feat && C <= 3;         // wider feature rows are written directly
  const int Cs = C;
  for (int v = 0; v < n_views; ++v) {
    const int cnt = s_w[v][0] + s_w[v][1] + s_w[v][2] + s_w[v][3];
    if (cnt == 0) continue;                                                      // workgroup-uniform
    const int64_t p0 = first[v] + chunk_off[(int64_t)v * n_chunks + blockIdx.x];
    int wcnt = cnt;                                       // rows of this chunk and view that fit the arrays
    if (o.cap > 0 && p0 + cnt > o.cap) {
      wcnt = p0 < o.cap ? (int)(o.cap - p0) : 0;
      if (threadIdx.x == 0 && o.overflow) *o.overflow = 1;
    }
    int lr = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) if (q == v) lr = rank[q];
    if (any) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (!((m[k] >> v) & 1)) continue;
        const int64_t i = i0 + k;
        const SetupRes r = splat_setup_point(x[k], y[k], z[k], nx[k], ny[k], nz[k], h[(int64_t)v * P + i], views + v * 16,
                                             projs + v * 16, S, sigma, cutoffC);
        s_ndc[lr * 3] = r.ndc[0]; s_ndc[lr * 3 + 1] = r.ndc[1]; s_ndc[lr * 3 + 2] = r.ndc[2];
        s_el[lr * 3] = r.el[0]; s_el[lr * 3 + 1] = r.el[1]; s_el[lr * 3 + 2] = r.el[2];
        s_rad[lr * 2] = r.rad[0]; s_rad[lr * 2 + 1] = r.rad[1];
        s_sc[lr] = r.scaler;
        s_src[lr] = (int32_t)i;
        if (stage_feat)
          for (int c = 0; c < Cs; ++c) s_ft[lr * Cs + c] = f[k][c];
        else if (o.feat && lr < wcnt)
          for (int c = 0; c < C; ++c)
            o.feat[(p0 + lr) * C + c] = feat_from_normal ? (c < 3 ? f[k][c] : 0.f) : feat_in[i * C + c];
        ++lr;
      }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < wcnt * 3; j += 256) {
      o.ndc[p0 * 3 + j] = s_ndc[j];
      o.ellipse[p0 * 3 + j] = s_el[j];
    }
    for (int j = threadIdx.x; j < wcnt * 2; j += 256) o.radii[p0 * 2 + j] = s_rad[j];
    for (int j = threadIdx.x; j < wcnt; j += 256) {
      o.cutoff[p0 + j] = cutoffC;
      o.scaler[p0 + j] = s_sc[j];
      if (o.src) o.src[p0 + j] = s_src[j];
      if (o.vis0) o.vis0[p0 + j] = 0;
    }
    if (stage_feat)
      for (int j = threadIdx.x; j < wcnt * Cs; j += 256) o.feat[p0 * Cs + j] = s_ft[j];
    __syncthreads();
  }
}

template <bool FILL>
__global__ void k_bin(const float* __restrict__ pts, const float* __restrict__ radii,
                      const int64_t* __restrict__ first, const int64_t* __restrict__ num, Frame F,
                      int ty_begin, int ty_end,
                      int32_t* __restrict__ tile_cnt,
                      const int32_t* __restrict__ tile_off, int32_t* __restrict__ pairs,
                      int64_t capacity, int32_t* __restrict__ overflow) {
  const int n = blockIdx.y;
  const int64_t len = num[n], base = first[n];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = base + i;
    const float z = pts[p * 3 + 2];
    if (!(z >= 0.f)) continue;  // behind the camera (rasterize_points.cu:87-88) or NaN
    int x0, x1, y0, y1;
    if (!pixel_range(pts[p * 3], radii[p * 2], F.W, F.ex, F.m, x0, x1)) continue;
    if (!pixel_range(pts[p * 3 + 1], radii[p * 2 + 1], F.H, F.ey, F.m, y0, y1)) continue;
    for (int ty = max(y0 / TILE, ty_begin); ty <= min(y1 / TILE, ty_end - 1); ++ty)
      for (int tx = x0 / TILE; tx <= x1 / TILE; ++tx) {
        const int tile = (n * F.Ty + ty) * F.Tx + tx;
        const int slot = atomicAdd(&tile_cnt[tile], 1);
        if (FILL) {
          const int64_t dst = (int64_t)tile_off[tile] + slot;
          if (dst < capacity) pairs[dst] = (int32_t)p;
          else *overflow = 1;
        }
      }
  }
}

// Same binning with the per-tile counters privatised in LDS (one view's T*T tiles fit for S <= 1024):
// a workgroup walks kBinChunk consecutive points, counts them per tile in LDS, reserves one range per
// touched tile with a single global atomic and (FILL) hands out the slots from LDS.  The hot tiles on
// a silhouette otherwise take thousands of same-address global atomics.
// The work items of the raster pass (tile_items_body, below) only need the tile offsets, not the filled pair list: the
// FILL launch carries them as one more workgroup (blockIdx = (gridDim.x - 1, 0)) instead of a single-workgroup launch
// of its own between the fill and the raster (14 us of the cfg-3a cycle).
struct TileItemsJob {          // what k_tile_items takes; items == null: no job
  int ty_rows, n_clouds, max_slots, target_items;
  int4* items; int4* heavy; int32_t* counters;
};
__device__ void tile_items_body(const int32_t* __restrict__ tile_off, Frame F, int ty_begin, int ty_rows, int n_clouds,
                                int max_slots, int target_items, int4* __restrict__ items, int4* __restrict__ heavy,
                                int32_t* __restrict__ counters);
constexpr int kBinChunk = 2048;      // rows per workgroup: 8 per thread -- with 8192 a 512^2 x 4 job was 245 workgroups of 32 sequential rows per thread, latency bound
template <bool FILL>
__global__ __launch_bounds__(256) void k_bin_lds(const float* __restrict__ pts, const float* __restrict__ radii,
                                                 const int64_t* __restrict__ first, const int64_t* __restrict__ num,
                                                 Frame F, int ty_begin, int ty_end,
                                                 int32_t* __restrict__ tile_cnt, const int32_t* __restrict__ tile_off,
                                                 int32_t* __restrict__ pairs, int64_t capacity,
                                                 int32_t* __restrict__ overflow, TileItemsJob job) {
  extern __shared__ int lh[];          // Tx*Ty
  if (FILL && job.items && blockIdx.x == gridDim.x - 1) {           // the extra workgroup(s) of the launch
    if (blockIdx.y == 0)
      tile_items_body(tile_off, F, ty_begin, job.ty_rows, job.n_clouds, job.max_slots, job.target_items, job.items,
                      job.heavy, job.counters);
    return;
  }
  const int TT = F.Tx * F.Ty;
  const int n = blockIdx.y;
  const int64_t len = num[n], base = first[n];
  const int64_t i0 = (int64_t)blockIdx.x * kBinChunk;
  if (i0 >= len) return;
  const int64_t i1 = min(len, i0 + kBinChunk);
  for (int t = threadIdx.x; t < TT; t += blockDim.x) lh[t] = 0;
  __syncthreads();
  auto walk = [&](auto&& visit) {
    for (int64_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
      const int64_t p = base + i;
      const float z = pts[p * 3 + 2];
      if (!(z >= 0.f)) continue;  // behind the camera (rasterize_points.cu:87-88) or NaN
      int x0, x1, y0, y1;
      if (!pixel_range(pts[p * 3], radii[p * 2], F.W, F.ex, F.m, x0, x1)) continue;
      if (!pixel_range(pts[p * 3 + 1], radii[p * 2 + 1], F.H, F.ey, F.m, y0, y1)) continue;
      for (int ty = max(y0 / TILE, ty_begin); ty <= min(y1 / TILE, ty_end - 1); ++ty)
        for (int tx = x0 / TILE; tx <= x1 / TILE; ++tx) visit(ty * F.Tx + tx, p);
    }
  };
  walk([&](int t, int64_t) { atomicAdd(&lh[t], 1); });
  __syncthreads();
  for (int t = threadIdx.x; t < TT; t += blockDim.x) {
    const int c = lh[t];
    if (c) {
      const int b = atomicAdd(&tile_cnt[n * TT + t], c);
      if (FILL) lh[t] = b;
    }
  }
  if (!FILL) return;
  __syncthreads();
  walk([&](int t, int64_t p) {
    const int slot = atomicAdd(&lh[t], 1);
    const int64_t dst = (int64_t)tile_off[n * TT + t] + slot;
    if (dst < capacity) pairs[dst] = (int32_t)p;
    else *overflow = 1;
  });
}

// returns true when the launch carried `job` (the caller then issues no k_tile_items launch)
template <bool FILL>
bool launch_bin(const float* points, const float* radii, const int64_t* first_idx, const int64_t* num_pts,
                int n_clouds, int64_t max_pts, Frame F, int ty0, int ty1, int32_t* tile_cnt,
                const int32_t* tile_off, int32_t* pairs, int64_t capacity, int32_t* overflow, hipStream_t s,
                TileItemsJob job = TileItemsJob{0, 0, 0, 0, nullptr, nullptr, nullptr}) {
  if (F.Tx * F.Ty <= 4096) {
    const bool carry = FILL && job.items != nullptr;
    hipLaunchKernelGGL(k_bin_lds<FILL>, dim3(iso_div_up(max_pts, kBinChunk) + (carry ? 1 : 0), n_clouds), dim3(256),
                       (size_t)F.Tx * F.Ty * sizeof(int), s, points, radii, first_idx, num_pts, F, ty0, ty1, tile_cnt,
                       tile_off, pairs, capacity, overflow, job);
    return carry;
  } else {
    int gx = iso_div_up(max_pts, 256); if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(k_bin<FILL>, dim3(gx, n_clouds), dim3(256), 0, s, points, radii, first_idx, num_pts, F,
                       ty0, ty1, tile_cnt, tile_off, pairs, capacity, overflow);
    return false;
  }
}

// ---------------------------------------------------------------- raster
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
  // The same insertion in the form of push_zi (below): depths >= +0 and never NaN, masks of integer compares straight
  // into the selects.  For lists that are merged (k_raster_merge): entries come from lists made by push_zi.
  __device__ __forceinline__ void push_q(float cz, int ci, float cq, int K) {
    const unsigned cu = __float_as_uint(cz);
    bool m[KMAX];
    const bool full = K == KMAX;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      const unsigned zu = __float_as_uint(z[j]);
      m[j] = (full | (j < K)) & ((cu < zu) | ((cu == zu) & (ci < id[j])));
    }
#pragma unroll
    for (int j = KMAX - 1; j >= 1; --j) {
      z[j] = m[j - 1] ? z[j - 1] : (m[j] ? cz : z[j]);
      q[j] = m[j - 1] ? q[j - 1] : (m[j] ? cq : q[j]);
      id[j] = m[j - 1] ? id[j - 1] : (m[j] ? ci : id[j]);
    }
    z[0] = m[0] ? cz : z[0];
    q[0] = m[0] ? cq : q[0];
    id[0] = m[0] ? ci : id[0];
  }
  // The same insertion for the (z, id) part only (the caller recomputes q for the K survivors at the end).
  // Compares are half-rate instructions here and the swap chain above spends three per level plus six
  // selects; this form takes one `<` and one `==` per level (the id compare only when some lane of the wave
  // meets an equal depth), the depths move with one v_med3_f32 per level and the ids with two selects.
  __device__ __forceinline__ void push_zi(float cz, int ci, int K) {
#ifdef RS_OLD_PUSH
    bool lt[KMAX], eq[KMAX];
    bool tie = false;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      lt[j] = j < K && cz < z[j];
      eq[j] = j < K && cz == z[j];
      tie = tie || eq[j];
    }
    if (__any(tie)) {
#pragma unroll
      for (int j = 0; j < KMAX; ++j) lt[j] = lt[j] || (eq[j] && ci < id[j]);
    }
#pragma unroll
    for (int j = KMAX - 1; j >= 1; --j) {
      id[j] = lt[j - 1] ? id[j - 1] : (lt[j] ? ci : id[j]);
      if (K == KMAX) z[j] = __builtin_amdgcn_fmed3f(z[j - 1], cz, z[j]);     // sorted list: the middle one
      else z[j] = lt[j - 1] ? z[j - 1] : (lt[j] ? cz : z[j]);
    }
    id[0] = lt[0] ? ci : id[0];
    z[0] = lt[0] ? cz : z[0];
#else
    // Depths here are >= +0 and never NaN (binning drops z < 0 and NaN, the caller adds +0.0f so that -0 is +0): their
    // float order is the order of their bit patterns as unsigned integers, and (z, id) "less" is three integer compares
    // whose masks go straight into the selects -- no tie pass, no ballot, no boolean arrays in vector registers (the
    // first form compiled to ~200 instructions per insertion; knock-outs of round 5: the insertions were 98 of the
    // kernel's 207 us).
    const unsigned cu = __float_as_uint(cz);
    bool m[KMAX];
    const bool full = K == KMAX;                        // (uniform: the usual case drops the slot test)
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      const unsigned zu = __float_as_uint(z[j]);
      // `|` and `&`, not `||` and `&&`: three compares and two mask operations, no branch per level
      m[j] = (full | (j < K)) & ((cu < zu) | ((cu == zu) & (ci < id[j])));
    }
#pragma unroll
    for (int j = KMAX - 1; j >= 1; --j) {
      // (sorted list, all slots live: the new z[j] is the middle one of z[j-1], cz, z[j])
#ifdef RS_MED3
      z[j] = full ? __builtin_amdgcn_fmed3f(z[j - 1], cz, z[j]) : (m[j - 1] ? z[j - 1] : (m[j] ? cz : z[j]));
#else
      z[j] = m[j - 1] ? z[j - 1] : (m[j] ? cz : z[j]);
#endif
      id[j] = m[j - 1] ? id[j - 1] : (m[j] ? ci : id[j]);
    }
    z[0] = m[0] ? cz : z[0];
    id[0] = m[0] ? ci : id[0];
#endif
  }
};

struct Cand {  // one LDS record per candidate (SoA in LDS)
  float px, py, pz, a, b, c, rx, ry, cut;
  int id;
};

// optional epilogue of the raster kernels: the compositing of k_composite for the pixel just finished
// (same arithmetic, same order; saves re-reading the K-deep lists)
struct CompositeArgs {
  const float* scaler;   // null: no compositing
  const float* feat;
  float* img;            // (N,S,S,C+1)
  int C, norm;
  float eps;
  uint8_t* vis;          // null, or (P,) flags: every point written to a pixel's list is marked 1 (k_mark_visible's pass)
};

template <int KMAX>
__device__ __forceinline__ void composite_pixel(const PixK<KMAX>& best, int K, float z0, float depth_thres, bool hit,
                                                const CompositeArgs& ca, int64_t pix) {
  float sw = 0.f;
  float acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = 0.f;
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    if (j < K) {
      const bool ok = best.z[j] < FLT_MAX && !((best.z[j] - z0) > depth_thres);
      if (ok) {
        const int p = best.id[j];
        const float w = expf(-0.5f * best.q[j]) * ca.scaler[p];
#pragma unroll
        for (int c = 0; c < 8; ++c) if (c < ca.C) acc[c] += w * ca.feat[(int64_t)p * ca.C + c];
        sw += w;
      }
    }
  }
  float d = 1.0f;
  if (ca.norm) d = sw > ca.eps ? sw : ca.eps;
#pragma unroll
  for (int c = 0; c < 8; ++c) if (c < ca.C) ca.img[pix * (ca.C + 1) + c] = ca.norm ? acc[c] / d : acc[c];
  ca.img[pix * (ca.C + 1) + ca.C] = hit ? 1.0f : 0.0f;
}

// -DRS_DBG_PHASES (timing experiment, tools/diag/raster_phases.py): thread 0 of every workgroup of k_raster adds the
// shader-clock time between consecutive marks (barrier waits included) to its slot of rs_phase.
#ifdef RS_DBG_PHASES
__device__ unsigned long long rs_phase[32768 * 8];
__shared__ unsigned long long s_rs_t, s_rs_acc[8];
#define RS_PH(i) do { if (threadIdx.x == 0) { const unsigned long long t_ = clock64(); \
  if ((i) >= 0) s_rs_acc[(i) < 0 ? 0 : (i)] += t_ - s_rs_t; \
  else for (int q_ = 0; q_ < 8; ++q_) s_rs_acc[q_] = 0; \
  s_rs_t = t_; } } while (0)
#define RS_PH_FLUSH() do { if (threadIdx.x == 0 && blockIdx.x < 32768) for (int q_ = 0; q_ < 8; ++q_) \
  rs_phase[blockIdx.x * 8 + q_] += s_rs_acc[q_]; } while (0)
extern "C" int iso_dbg_raster_phases(double* out16) {
  static unsigned long long h[32768 * 8];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(rs_phase), sizeof(h)) != hipSuccess) return -1;
  for (int i = 0; i < 16; ++i) out16[i] = 0.0;
  for (int w = 0; w < 32768; ++w) {
    bool any = false;
    for (int i = 0; i < 8; ++i) { out16[i] += (double)h[w * 8 + i]; any = any || h[w * 8 + i]; }
    if (any) out16[8] += 1.0;
  }
  void* dp = nullptr;
  (void)hipGetSymbolAddress(&dp, HIP_SYMBOL(rs_phase));
  (void)hipMemset(dp, 0, sizeof(h));
  return 0;
}
#else
#define RS_PH(i) do {} while (0)
#define RS_PH_FLUSH() do {} while (0)
#endif

// CP ("candidate parallel"): how the hits of a chunk of 256 candidates reach the pixels.
//   false: every pixel thread walks the candidates that reach its wave's four rows and tests each one (the first
//          form: the K-best insertion, ~50 instructions, runs for the whole wave whenever ANY of its lanes is hit --
//          with splats a few pixels wide that is nearly every candidate, for a handful of lanes each time);
//   true:  thread t walks the few pixels of candidate t's bounding box inside the tile, tests them exactly as the
//          pixel thread would (same expressions on the same operands) and appends t to the hit list of every pixel
//          it covers (LDS counters); the pixel threads then insert only their own hits.  The K-best rule is a total
//          order on (z, id), so the arrival order in the lists does not matter.  A pixel's first kHitList hits of a
//          chunk go to its list, the others set the candidate's bit in the pixel's mask (s_more: 256 bits), which the
//          pixel thread walks after its list.  Up to kWideCap candidates per chunk whose box covers more than
//          kWideArea pixels take the first form (tested by every pixel thread).
// (Round 4, when a pixel with more hits than list entries tested ALL 256 candidates: 32 / 24 / 16-entry lists 316 / 333 /
// 429 us against 302 with 40; wide boxes from 24 / 48 / 96 / 192 pixels: 367 / 302 / 294 / 298 us.  Round 5 knock-outs at the
// scale of one rank's band of 8, SIREN surface: that overfull path was 43 of the kernel's 98 us -- a silhouette tile
// holds pixels with more than 40 hits in nearly every chunk, and each of them kept its whole wave for 256 tests.)
constexpr int kHitList = 32, kWideArea = 96, kWideCap = 32;

// KFULL: points_per_pixel == KMAX (4 / 8 / 16 / 32: the usual settings) -- K is then a compile-time constant and the
// per-slot `j < K` tests, the selection of the list's last live entry and their scalar branches fold away (a third of
// the instructions of an insertion).  (Seven workgroups per CU leave 72 vector registers; the runtime-K form needs one
// more and takes six -- a raster kernel must not spill, tests/test_abi.py.)
template <int KMAX, bool CP, bool KFULL = false>
__global__ __launch_bounds__(256, (CP && KMAX <= 8) ? (KFULL ? 7 : 6) : 1) void k_raster(
    const float* __restrict__ pts, const float* __restrict__ ellipse,
    const float* __restrict__ cutoff, const float* __restrict__ radii,
    const int32_t* __restrict__ tile_order, const int4* __restrict__ items, const int32_t* __restrict__ item_count,
    float* __restrict__ scratch, const int32_t* __restrict__ tile_off,
    const int32_t* __restrict__ pairs, int64_t capacity, Frame F,
    int K_arg, float depth_thres, int32_t* __restrict__ idx_out, float* __restrict__ zbuf_out, float* __restrict__ q_out,
    float* __restrict__ occ_out, CompositeArgs ca) {
  const int K = KFULL ? KMAX : K_arg;
  constexpr int NSOA = CP ? 1 : 256;
  __shared__ float s_px[NSOA], s_py[NSOA], s_pz[NSOA], s_a[NSOA], s_b[NSOA], s_c[NSOA], s_rx[NSOA], s_ry[NSOA], s_cut[NSOA];
  __shared__ int s_id[NSOA];
  // per-wave candidate lists: wave w owns pixel rows 4w..4w+3 of the tile and only walks the
  // candidates whose y-extent reaches those rows (splats are a few pixels wide: ~40 % of them)
  __shared__ short s_list[4][NSOA];
  __shared__ int s_cntw[4][4];       // [source wave][target wave]
  __shared__ int s_hits[CP ? 256 : 1];                      // CP: hits of the chunk per pixel
  __shared__ __attribute__((aligned(16))) unsigned char s_hit[CP ? 256 : 1][CP ? kHitList : 1];   // the first kHitList of them
  __shared__ unsigned s_more[CP ? 8 : 1][CP ? 256 : 1];     // the others: bit k % 32 of word [k / 32][pixel]
  // workgroups take the tiles of the band heaviest first (k_tile_order): a tile on the sphere's
  // silhouette holds 8x the mean number of candidates and would otherwise finish long after the rest
  // a work item = a tile, or -- for the tiles that hold many times the mean number of candidates (a
  // silhouette) -- one slice of its candidate list; the K-best lists of the slices are merged afterwards
  // (k_raster_merge), so the longest item is a slice, not the heaviest tile
  int tile, slice = 0, nslices = 1, slot = 0;
  if (items) {
    if ((int)blockIdx.x >= *item_count) return;
    const int4 it = items[blockIdx.x];
    tile = it.x; slice = it.y; nslices = it.z; slot = it.w;
  } else {
    tile = tile_order[blockIdx.x];
  }
  const int tx = tile % F.Tx, ty = (tile / F.Tx) % F.Ty, n = tile / (F.Tx * F.Ty);
  const int lx = threadIdx.x % TILE, ly = threadIdx.x / TILE;
  const int xi = tx * TILE + lx, yi = ty * TILE + ly;  // NDC pixel index
  const bool inside = xi < F.W && yi < F.H;
  const float xf = ndc_x(xi, F), yf = ndc_y(yi, F);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // NDC y-range of the four row bands (pix_to_ndc is increasing), widened a little so that the
  // band test is a strict superset of the exact per-pixel test below
  float band_lo[4], band_hi[4];
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    band_lo[w] = ndc_y(ty * TILE + 4 * w, F) - 1.0e-6f;
    band_hi[w] = ndc_y(ty * TILE + 4 * w + 3, F) + 1.0e-6f;
  }
  PixK<KMAX> best;
  best.init();
  float wz = FLT_MAX;
  int wi = 0x7fffffff;
  const int64_t off = tile_off[tile];
  int cnt = tile_off[tile + 1] - tile_off[tile];
  if (off + cnt > capacity) cnt = off < capacity ? (int)(capacity - off) : 0;  // overflow guard
  int c_begin = 0;
  if (nslices > 1) {                       // slice s of ns: chunk-aligned share of the list
    const int chunks = (cnt + 255) / 256;
    c_begin = (int)((int64_t)chunks * slice / nslices) * 256;
    cnt = min(cnt, (int)((int64_t)chunks * (slice + 1) / nslices) * 256);
  }
  if constexpr (CP) {
    // records of a chunk: {x, y, z, id} in LDS (the pixel threads read depth and id of their hits); {a, b, c, cutoff}
    // {rx, ry} stay in the registers of the thread that walks the candidate's box -- only the few wide candidates put
    // theirs in a small table for the pixel threads.  The next chunk's records are requested (into registers) while
    // this one is worked on.
    // ONE record buffer (a third barrier per chunk): 22.5 KB of LDS instead of 37 put seven workgroups on a CU
    // instead of four -- 387 -> 302 us; the kernel is bound by what the resident waves can overlap
    __shared__ float4 s_r0[256];
    __shared__ short s_wide[2][kWideCap];
    __shared__ float s_wrec[2][kWideCap][6];              // {a, b, c, cutoff, rx, ry}
    __shared__ int s_nw[2];
    float4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = r0;
    float2 r2 = {0.f, 0.f};
    auto fetch = [&](int c0) {
      if (c0 + (int)threadIdx.x < cnt) {
        const int p = pairs[off + c0 + threadIdx.x];
        r0 = make_float4(pts[(int64_t)p * 3], pts[(int64_t)p * 3 + 1], pts[(int64_t)p * 3 + 2], __int_as_float(p));
        r1 = make_float4(ellipse[(int64_t)p * 3], ellipse[(int64_t)p * 3 + 1], ellipse[(int64_t)p * 3 + 2], cutoff[p]);
        r2 = make_float2(radii[(int64_t)p * 2], radii[(int64_t)p * 2 + 1]);
      }
    };
    int par = 0;
    auto push_if_better = [&](const float4& c0v) {
      const float pz = c0v.z + 0.0f;                     // (-0 -> +0: push_zi orders depths by their bit patterns)
      const int id = __float_as_int(c0v.w);
      if (pz < wz || (pz == wz && id < wi)) {
        best.push_zi(pz, id, K);
        if (K == KMAX) { wz = best.z[KMAX - 1]; wi = best.id[KMAX - 1]; }
        else {
#pragma unroll
          for (int j = 0; j < KMAX; ++j) if (j == K - 1) { wz = best.z[j]; wi = best.id[j]; }
        }
      }
    };
    auto hit_push = [&](int k) { push_if_better(s_r0[k]); };                      // a listed hit: it passed the tests already
    auto wide_push = [&](int i) {                                                  // entry i of the wide table, tested here
      const float4 c0v = s_r0[s_wide[par][i]];
      const float* wr = s_wrec[par][i];
      const float dx = xf - c0v.x, dy = yf - c0v.y;
      if (fabsf(dx) > wr[4] || fabsf(dy) > wr[5]) return;                          // rasterize_points.cu:92
      const float q = wr[0] * dx * dx + wr[1] * dx * dy + wr[2] * dy * dy;          // :94
      if (q > wr[3]) return;                                                       // :96
      push_if_better(c0v);
    };
    s_hits[threadIdx.x] = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) s_more[w][threadIdx.x] = 0u;
    if (threadIdx.x < 2) s_nw[threadIdx.x] = 0;
    RS_PH(-1);
    fetch(c_begin);
    for (int c0 = c_begin; c0 < cnt; c0 += 256, par ^= 1) {
      const int m = min(256, cnt - c0);
      const float4 c0v = r0, c1v = r1;                    // this thread's candidate of the chunk
      const float2 c2v = r2;
      if ((int)threadIdx.x < m) s_r0[threadIdx.x] = r0;
      __syncthreads();                                    // records visible; the hit counters are zero
      RS_PH(0);
      fetch(c0 + 256);
#ifdef RS_KO_A        // timing experiment (results wrong): no candidate phase
      if (false) {
#else
      if ((int)threadIdx.x < m) {
#endif
        const int k = threadIdx.x;
        int x0 = 0, x1 = -1, y0 = 0, y1 = -1;
        const bool any = pixel_range(c0v.x, c2v.x, F.W, F.ex, F.m, x0, x1) && pixel_range(c0v.y, c2v.y, F.H, F.ey, F.m, y0, y1);
        x0 = max(x0, tx * TILE); x1 = min(x1, tx * TILE + TILE - 1);
        y0 = max(y0, ty * TILE); y1 = min(y1, ty * TILE + TILE - 1);
        bool walk = any && x0 <= x1 && y0 <= y1;
        if (walk && (x1 - x0 + 1) * (y1 - y0 + 1) > kWideArea) {
          const int ws = atomicAdd(&s_nw[par], 1);
          if (ws < kWideCap) {                            // (a full table: the box is walked like the others)
            s_wide[par][ws] = (short)k;
            float* wr = s_wrec[par][ws];
            wr[0] = c1v.x; wr[1] = c1v.y; wr[2] = c1v.z; wr[3] = c1v.w; wr[4] = c2v.x; wr[5] = c2v.y;
            walk = false;
          }
        }
        if (walk) {
          for (int y = y0; y <= y1; ++y) {
            const float dy = ndc_y(y, F) - c0v.y;
            if (fabsf(dy) > c2v.y) continue;
            for (int x = x0; x <= x1; ++x) {
              const float dx = ndc_x(x, F) - c0v.x;
              if (fabsf(dx) > c2v.x) continue;
              const float q = c1v.x * dx * dx + c1v.y * dx * dy + c1v.z * dy * dy;
              if (q > c1v.w) continue;
              const int pl = (y - ty * TILE) * TILE + (x - tx * TILE);
              const int slot = atomicAdd(&s_hits[pl], 1);
              if (slot < kHitList) s_hit[pl][slot] = (unsigned char)k;
              else atomicOr(&s_more[k >> 5][pl], 1u << (k & 31));         // beyond the list: the pixel's bit mask
            }
          }
        }
      }
      __syncthreads();                                    // hit lists complete
      RS_PH(1);
      const int nh = s_hits[threadIdx.x];
      s_hits[threadIdx.x] = 0;
      if (threadIdx.x == 0) s_nw[par ^ 1] = 0;
#ifdef RS_KO_B          // timing experiment (results wrong): no insertions
      if (false) {
#else
      if (inside) {
#endif
        const int nl = min(nh, kHitList);
        for (int i = 0; i < nl; ++i) hit_push(s_hit[threadIdx.x][i]);
        if (nh > kHitList) {
#pragma unroll 1
          for (int w = 0; w < 8; ++w) {
            unsigned mm = s_more[w][threadIdx.x];
            s_more[w][threadIdx.x] = 0u;
            while (mm) {
              const int b = __ffs((int)mm) - 1;
              mm &= mm - 1u;
              hit_push(w * 32 + b);
            }
          }
        }
        const int nw = min(s_nw[par], kWideCap);
        for (int i = 0; i < nw; ++i) wide_push(i);
      }
      RS_PH(2);
      __syncthreads();                                    // one record buffer: all reads done before the next chunk lands
      RS_PH(3);
    }
    // q of the K survivors (:94; the same expression on the same operands as the hit test): their records are
    // re-read once per tile instead of carrying q through every insertion
#ifdef RS_KO_EPI        // timing experiment (results wrong): no epilogue
    if (best.z[0] == 12345.f) occ_out[0] = 1.f;
    return;
#endif
    if (inside) {
#pragma unroll
      for (int j = 0; j < KMAX; ++j) {
        if (j < K && best.z[j] < FLT_MAX) {
          const int64_t p = best.id[j];
          const float dx = xf - pts[p * 3], dy = yf - pts[p * 3 + 1];
          best.q[j] = ellipse[p * 3] * dx * dx + ellipse[p * 3 + 1] * dx * dy + ellipse[p * 3 + 2] * dy * dy;
        }
      }
    }
    RS_PH(4);
  } else {
  for (int c0 = c_begin; c0 < cnt; c0 += 256) {
      const int m = min(256, cnt - c0);
      __syncthreads();
      bool hit_band[4] = {false, false, false, false};
      if ((int)threadIdx.x < m) {
        const int p = pairs[off + c0 + threadIdx.x];
        const float py = pts[(int64_t)p * 3 + 1], ry = radii[(int64_t)p * 2 + 1];
        s_px[threadIdx.x] = pts[(int64_t)p * 3];
        s_py[threadIdx.x] = py;
        s_pz[threadIdx.x] = pts[(int64_t)p * 3 + 2];
        s_a[threadIdx.x] = ellipse[(int64_t)p * 3];
        s_b[threadIdx.x] = ellipse[(int64_t)p * 3 + 1];
        s_c[threadIdx.x] = ellipse[(int64_t)p * 3 + 2];
        s_rx[threadIdx.x] = radii[(int64_t)p * 2];
        s_ry[threadIdx.x] = ry;
        s_cut[threadIdx.x] = cutoff[p];
        s_id[threadIdx.x] = p;
        const float ylo = py - ry * 1.000001f - 1.0e-6f, yhi = py + ry * 1.000001f + 1.0e-6f;
  #pragma unroll
        for (int w = 0; w < 4; ++w) hit_band[w] = !(yhi < band_lo[w]) && !(ylo > band_hi[w]);   // NaN -> kept
      }
      int rank[4];
  #pragma unroll
      for (int w = 0; w < 4; ++w) {
        const unsigned long long bal = __ballot(hit_band[w]);
        rank[w] = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) s_cntw[wave][w] = __popcll(bal);
      }
      __syncthreads();
  #pragma unroll
      for (int w = 0; w < 4; ++w) {
        if (hit_band[w]) {
          int base = 0;
  #pragma unroll
          for (int sw = 0; sw < 4; ++sw) base += (sw < wave) ? s_cntw[sw][w] : 0;
          s_list[w][base + rank[w]] = (short)threadIdx.x;
        }
      }
      const int mine = s_cntw[0][wave] + s_cntw[1][wave] + s_cntw[2][wave] + s_cntw[3][wave];
      __syncthreads();
      if (inside) {
        for (int i = 0; i < mine; ++i) {
          const int k = s_list[wave][i];
          const float dx = xf - s_px[k], dy = yf - s_py[k];
          if (fabsf(dx) > s_rx[k] || fabsf(dy) > s_ry[k]) continue;  // rasterize_points.cu:92
          const float q = s_a[k] * dx * dx + s_b[k] * dx * dy + s_c[k] * dy * dy;  // :94
          if (q > s_cut[k]) continue;                                              // :96
          const float pz = s_pz[k] + 0.0f;            // (-0 -> +0: the slice merge orders depths by their bit patterns)
          const int id = s_id[k];
          if (pz < wz || (pz == wz && id < wi)) {
            best.push(pz, id, q, K);
  #pragma unroll
            for (int j = 0; j < KMAX; ++j) if (j == K - 1) { wz = best.z[j]; wi = best.id[j]; }
          }
        }
      }
    }
  }
  if (nslices > 1) {                       // partial result: the raw K-best of this slice, [z | q | id][k][pixel]
    float* sc = scratch + (int64_t)slot * 3 * KMAX * 256;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      sc[j * 256 + threadIdx.x] = best.z[j];
      sc[(KMAX + j) * 256 + threadIdx.x] = best.q[j];
      sc[(2 * KMAX + j) * 256 + threadIdx.x] = __int_as_float(best.id[j]);
    }
    RS_PH(5);
    RS_PH_FLUSH();
    return;
  }
  if (!inside) return;
  // output pixel is flipped in both axes (+X left, +Y up; rasterize_points.cu:577-580)
  const int yo = F.H - 1 - yi, xo = F.W - 1 - xi;
  const int64_t pix = ((int64_t)n * F.H + yo) * F.W + xo;
  const float z0 = best.z[0];
  const bool hit = z0 < FLT_MAX;
  occ_out[pix] = hit ? 1.0f : 0.0f;
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    if (j < K) {
      const bool ok = best.z[j] < FLT_MAX && !((best.z[j] - z0) > depth_thres);
      idx_out[pix * K + j] = ok ? best.id[j] : -1;
      zbuf_out[pix * K + j] = ok ? best.z[j] : -1.0f;
      q_out[pix * K + j] = ok ? best.q[j] : -1.0f;
      if (ok && ca.vis) ca.vis[best.id[j]] = 1;
    }
  }
  if (ca.scaler) composite_pixel<KMAX>(best, K, z0, depth_thres, hit, ca, pix);
  RS_PH(5);
  RS_PH_FLUSH();
}

// points_per_pixel above 32 (the reference allows 150, rasterization_utils.cuh:18): the K-best list of a pixel does
// not fit its thread's registers, so it lives where it ends up anyway -- in the pixel's rows of the output arrays --
// and a hit is inserted by shifting the tail in global memory.  Same candidate walk, same (z, id) order, same
// outputs as k_raster; built for completeness, not for speed (a list this deep is a debugging setting).
__global__ __launch_bounds__(256) void k_raster_deep(
    const float* __restrict__ pts, const float* __restrict__ ellipse, const float* __restrict__ cutoff,
    const float* __restrict__ radii, const int32_t* __restrict__ tile_order, const int32_t* __restrict__ tile_off,
    const int32_t* __restrict__ pairs, int64_t capacity, Frame F, int K, float depth_thres,
    int32_t* __restrict__ idx_out, float* __restrict__ zbuf_out, float* __restrict__ q_out, float* __restrict__ occ_out,
    CompositeArgs ca) {
  __shared__ float s_px[256], s_py[256], s_pz[256], s_a[256], s_b[256], s_c[256], s_rx[256], s_ry[256], s_cut[256];
  __shared__ int s_id[256];
  const int tile = tile_order[blockIdx.x];
  const int tx = tile % F.Tx, ty = (tile / F.Tx) % F.Ty, n = tile / (F.Tx * F.Ty);
  const int lx = threadIdx.x % TILE, ly = threadIdx.x / TILE;
  const int xi = tx * TILE + lx, yi = ty * TILE + ly;
  const bool inside = xi < F.W && yi < F.H;
  const float xf = ndc_x(xi, F), yf = ndc_y(yi, F);
  const int yo = F.H - 1 - yi, xo = F.W - 1 - xi;
  const int64_t pix = ((int64_t)n * F.H + yo) * F.W + xo;
  int32_t* const li = idx_out + pix * K;      // the pixel's lists (only touched when inside)
  float* const lz = zbuf_out + pix * K;
  float* const lq = q_out + pix * K;
  int have = 0;
  float wz = FLT_MAX;                         // worst entry of a full list
  int wi = 0x7fffffff;
  const int64_t off = tile_off[tile];
  int cnt = tile_off[tile + 1] - tile_off[tile];
  if (off + cnt > capacity) cnt = off < capacity ? (int)(capacity - off) : 0;
  for (int c0 = 0; c0 < cnt; c0 += 256) {
    const int m = min(256, cnt - c0);
    __syncthreads();
    if ((int)threadIdx.x < m) {
      const int p = pairs[off + c0 + threadIdx.x];
      s_px[threadIdx.x] = pts[(int64_t)p * 3]; s_py[threadIdx.x] = pts[(int64_t)p * 3 + 1]; s_pz[threadIdx.x] = pts[(int64_t)p * 3 + 2];
      s_a[threadIdx.x] = ellipse[(int64_t)p * 3]; s_b[threadIdx.x] = ellipse[(int64_t)p * 3 + 1]; s_c[threadIdx.x] = ellipse[(int64_t)p * 3 + 2];
      s_rx[threadIdx.x] = radii[(int64_t)p * 2]; s_ry[threadIdx.x] = radii[(int64_t)p * 2 + 1];
      s_cut[threadIdx.x] = cutoff[p]; s_id[threadIdx.x] = p;
    }
    __syncthreads();
    if (!inside) continue;
    for (int k = 0; k < m; ++k) {
      const float dx = xf - s_px[k], dy = yf - s_py[k];
      if (fabsf(dx) > s_rx[k] || fabsf(dy) > s_ry[k]) continue;                  // rasterize_points.cu:92
      const float q = s_a[k] * dx * dx + s_b[k] * dx * dy + s_c[k] * dy * dy;      // :94
      if (q > s_cut[k]) continue;                                                // :96
      const float pz = s_pz[k];
      const int id = s_id[k];
      if (have == K && !(pz < wz || (pz == wz && id < wi))) continue;
      int j = have < K ? have : K - 1;        // the slot that opens (the worst entry drops out of a full list)
      while (j > 0) {
        const float zj = lz[j - 1];
        const int ij = li[j - 1];
        if (!(pz < zj || (pz == zj && id < ij))) break;
        lz[j] = zj; li[j] = ij; lq[j] = lq[j - 1];
        --j;
      }
      lz[j] = pz; li[j] = id; lq[j] = q;
      if (have < K) ++have;
      if (have == K) { wz = lz[K - 1]; wi = li[K - 1]; }
    }
  }
  if (!inside) return;
  const bool hit = have > 0;
  const float z0 = hit ? lz[0] : FLT_MAX;
  occ_out[pix] = hit ? 1.0f : 0.0f;
  float sw = 0.f, acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = 0.f;
  for (int j = 0; j < K; ++j) {
    const bool ok = j < have && !((lz[j] - z0) > depth_thres);
    if (ok && ca.scaler) {
      const int p = li[j];
      const float w = expf(-0.5f * lq[j]) * ca.scaler[p];
#pragma unroll
      for (int c = 0; c < 8; ++c) if (c < ca.C) acc[c] += w * ca.feat[(int64_t)p * ca.C + c];
      sw += w;
    }
    if (ok && ca.vis) ca.vis[li[j]] = 1;
    if (!ok) { li[j] = -1; lz[j] = -1.0f; lq[j] = -1.0f; }
  }
  if (ca.scaler) {
    float d = 1.0f;
    if (ca.norm) d = sw > ca.eps ? sw : ca.eps;
#pragma unroll
    for (int c = 0; c < 8; ++c) if (c < ca.C) ca.img[pix * (ca.C + 1) + c] = ca.norm ? acc[c] / d : acc[c];
    ca.img[pix * (ca.C + 1) + ca.C] = hit ? 1.0f : 0.0f;
  }
}

// Tiles of the band [ty_begin, ty_begin+ty_rows) of every cloud, heaviest first (64 buckets of the
// candidate count; the order inside a bucket is arbitrary -- it only affects scheduling).
// ---- the reference's two-stage interface: DSS._C._rasterize_coarse / _rasterize_fine (ext.cpp:11-12) --------------
// coarse (RasterizePointsCoarseCudaKernel, rasterize_points.cu:293-441): bin (by, bx) of cloud n -- bin_size x bin_size
// pixels -- lists the PACKED indices of the cloud's points with z >= 0 whose box [p - r, p + r] overlaps the bin's NDC
// extent (PixToNdc of its first / last pixel -+ half a pixel, both comparisons non-strict), -1 behind them.  The
// reference's order inside a bin is the arrival order of its atomics; here: ascending index (the order of the
// reference's CPU path, rasterize_points_cpu.cpp:190-222).  A bin that would hold more than M entries is cut at M and
// *overflow is set (the CUDA reference writes past the bin, its CPU path raises "Got too many points per bin").
// One workgroup per (bin, cloud): it walks the cloud in index order, 256 points a round, and compacts the hits with
// ballots -- the table is a compatibility surface, not the cycle's path (that is the tile lists of k_bin_lds).
__global__ __launch_bounds__(256) void k_coarse_bins(const float* __restrict__ pts, const float* __restrict__ radii,
                                                     const int64_t* __restrict__ first, const int64_t* __restrict__ num,
                                                     int S, int bin_size, int B, int M, int32_t* __restrict__ bin_points,
                                                     int32_t* __restrict__ overflow) {
  __shared__ int s_w[4];
  const int n = blockIdx.y, by = blockIdx.x / B, bx = blockIdx.x % B;
  const float half_pix = 1.0f / (float)S;
  auto pix2ndc = [&](int i) { return -1 + (2 * i + 1.0f) / S; };          // rasterization_utils.cuh:8-11
  const float bx0 = pix2ndc(bx * bin_size) - half_pix, bx1 = pix2ndc((bx + 1) * bin_size - 1) + half_pix;
  const float by0 = pix2ndc(by * bin_size) - half_pix, by1 = pix2ndc((by + 1) * bin_size - 1) + half_pix;
  int32_t* out = bin_points + (((int64_t)n * B + by) * B + bx) * M;
  const int64_t len = num[n], base = first[n];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int64_t filled = 0;
  for (int64_t i0 = 0; i0 < len; i0 += 256) {
    const int64_t i = i0 + threadIdx.x;
    bool hit = false;
    if (i < len) {
      const int64_t p = base + i;
      const float px = pts[p * 3], py = pts[p * 3 + 1], pz = pts[p * 3 + 2];
      if (!(pz < 0)) {                                                    // :349 (a NaN depth is not "behind")
        const float rx = radii[p * 2], ry = radii[p * 2 + 1];
        const float px0 = px - rx, px1 = px + rx, py0 = py - ry, py1 = py + ry;
        hit = (py0 <= by1) && (by0 <= py1) && (px0 <= bx1) && (bx0 <= px1);   // :366,:378
      }
    }
    const unsigned long long bal = __ballot(hit);
    if (lane == 0) s_w[w] = __popcll(bal);
    __syncthreads();
    int before = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (k < w) before += s_w[k]; tot += s_w[k]; }
    if (hit) {
      const int64_t slot = filled + before + __popcll(bal & ((1ull << lane) - 1ull));
      if (slot < M) out[slot] = (int32_t)(base + i);
      else *overflow = 1;
    }
    filled += tot;
    __syncthreads();
  }
  for (int64_t k = (filled < M ? filled : M) + threadIdx.x; k < M; k += 256) out[k] = -1;
}

// fine (RasterizePointsFineCudaKernel, rasterize_points.cu:503-596): every pixel tests the entries of ITS bin (negative
// entries skipped wherever they stand, :567-571) exactly as the naive kernel tests every point (CheckPixelInsidePoint
// :79-97) and keeps the K front-most hits -- by (z, index), the total order of every raster path here; then the
// depth-merging cut and the outputs of k_raster.  One thread per pixel.
template <int KMAX>
__global__ __launch_bounds__(256) void k_fine_bins(const float* __restrict__ pts, const float* __restrict__ ellipse,
                                                   const float* __restrict__ cutoff, const float* __restrict__ radii,
                                                   const int32_t* __restrict__ bin_points, int N, int B, int M, int bin_size,
                                                   Frame F, int K, float depth_thres, int64_t n_points,
                                                   int32_t* __restrict__ idx_out, float* __restrict__ zbuf_out,
                                                   float* __restrict__ q_out, float* __restrict__ occ_out) {
  const int64_t npix = (int64_t)N * F.H * F.W;
  for (int64_t pid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pid < npix; pid += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(pid / ((int64_t)F.H * F.W));
    const int yi = (int)((pid / F.W) % F.H), xi = (int)(pid % F.W);
    const int by = yi / bin_size, bx = xi / bin_size;
    const float xf = ndc_x(xi, F), yf = ndc_y(yi, F);
    const int32_t* bins = bin_points + (((int64_t)n * B + by) * B + bx) * M;
    PixK<KMAX> best;
    best.init();
    float wz = FLT_MAX;
    int wi = 0x7fffffff;
    for (int m = 0; m < M; ++m) {
      const int p = bins[m];
      if (p < 0 || p >= n_points) continue;
      const float pz = pts[(int64_t)p * 3 + 2];
      if (!(pz >= 0.f)) continue;                          // behind the camera (:87-88) or NaN, as every raster path here
      const float dx = xf - pts[(int64_t)p * 3], dy = yf - pts[(int64_t)p * 3 + 1];
      if (fabsf(dx) > radii[(int64_t)p * 2] || fabsf(dy) > radii[(int64_t)p * 2 + 1]) continue;                 // :92
      const float q = ellipse[(int64_t)p * 3] * dx * dx + ellipse[(int64_t)p * 3 + 1] * dx * dy + ellipse[(int64_t)p * 3 + 2] * dy * dy;   // :94
      if (q > cutoff[p]) continue;                                                                                // :96
      if (pz < wz || (pz == wz && p < wi)) {
        best.push(pz, p, q, K);
#pragma unroll
        for (int j = 0; j < KMAX; ++j) if (j == K - 1) { wz = best.z[j]; wi = best.id[j]; }
      }
    }
    const int yo = F.H - 1 - yi, xo = F.W - 1 - xi;        // :577-580
    const int64_t pix = ((int64_t)n * F.H + yo) * F.W + xo;
    const float z0 = best.z[0];
    occ_out[pix] = z0 < FLT_MAX ? 1.0f : 0.0f;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      if (j < K) {
        const bool ok = best.z[j] < FLT_MAX && !((best.z[j] - z0) > depth_thres);
        idx_out[pix * K + j] = ok ? best.id[j] : -1;
        zbuf_out[pix * K + j] = ok ? best.z[j] : -1.0f;
        q_out[pix * K + j] = ok ? best.q[j] : -1.0f;
      }
    }
  }
}

// k_fine_bins for points_per_pixel above 32 (the reference's bound is 150, rasterization_utils.cuh:18, checked at
// rasterize_points.cu:246-251): as in k_raster_deep the pixel's list lives in its rows of the output arrays and a hit is
// inserted by shifting the tail there.  Same walk over the bin, same tests, same (z, index) order, same cut.
__global__ __launch_bounds__(256) void k_fine_bins_deep(const float* __restrict__ pts, const float* __restrict__ ellipse,
                                                        const float* __restrict__ cutoff, const float* __restrict__ radii,
                                                        const int32_t* __restrict__ bin_points, int N, int B, int M,
                                                        int bin_size, Frame F, int K, float depth_thres, int64_t n_points,
                                                        int32_t* __restrict__ idx_out, float* __restrict__ zbuf_out,
                                                        float* __restrict__ q_out, float* __restrict__ occ_out) {
  const int64_t npix = (int64_t)N * F.H * F.W;
  for (int64_t pid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pid < npix; pid += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(pid / ((int64_t)F.H * F.W));
    const int yi = (int)((pid / F.W) % F.H), xi = (int)(pid % F.W);
    const int by = yi / bin_size, bx = xi / bin_size;
    const float xf = ndc_x(xi, F), yf = ndc_y(yi, F);
    const int32_t* bins = bin_points + (((int64_t)n * B + by) * B + bx) * M;
    const int yo = F.H - 1 - yi, xo = F.W - 1 - xi;        // :577-580
    const int64_t pix = ((int64_t)n * F.H + yo) * F.W + xo;
    int32_t* const li = idx_out + pix * K;
    float* const lz = zbuf_out + pix * K;
    float* const lq = q_out + pix * K;
    int have = 0;
    float wz = FLT_MAX;
    int wi = 0x7fffffff;
    for (int m = 0; m < M; ++m) {
      const int p = bins[m];
      if (p < 0 || p >= n_points) continue;
      const float pz = pts[(int64_t)p * 3 + 2];
      if (!(pz >= 0.f)) continue;
      const float dx = xf - pts[(int64_t)p * 3], dy = yf - pts[(int64_t)p * 3 + 1];
      if (fabsf(dx) > radii[(int64_t)p * 2] || fabsf(dy) > radii[(int64_t)p * 2 + 1]) continue;                 // :92
      const float q = ellipse[(int64_t)p * 3] * dx * dx + ellipse[(int64_t)p * 3 + 1] * dx * dy + ellipse[(int64_t)p * 3 + 2] * dy * dy;   // :94
      if (q > cutoff[p]) continue;                                                                                // :96
      if (have == K && !(pz < wz || (pz == wz && p < wi))) continue;
      int j = have < K ? have : K - 1;
      while (j > 0) {
        const float zj = lz[j - 1];
        const int ij = li[j - 1];
        if (!(pz < zj || (pz == zj && p < ij))) break;
        lz[j] = zj; li[j] = ij; lq[j] = lq[j - 1];
        --j;
      }
      lz[j] = pz; li[j] = p; lq[j] = q;
      if (have < K) ++have;
      if (have == K) { wz = lz[K - 1]; wi = li[K - 1]; }
    }
    const float z0 = have > 0 ? lz[0] : FLT_MAX;
    occ_out[pix] = have > 0 ? 1.0f : 0.0f;
    for (int j = 0; j < K; ++j) {
      const bool ok = j < have && !((lz[j] - z0) > depth_thres);
      if (!ok) { li[j] = -1; lz[j] = -1.0f; lq[j] = -1.0f; }
    }
  }
}

__global__ __launch_bounds__(1024) void k_tile_order(const int32_t* __restrict__ tile_off, Frame F, int ty_begin,
                                                     int ty_rows, int n_clouds, int32_t* __restrict__ order) {
  const int T = F.Tx, TY = F.Ty;
  __shared__ int hist[64], base[64];
  const int tiles = n_clouds * T * ty_rows;
  if (threadIdx.x < 64) hist[threadIdx.x] = 0;
  __syncthreads();
  auto tile_of = [&](int i) { return ((i / (T * ty_rows)) * TY + ty_begin + (i / T) % ty_rows) * T + i % T; };
  auto bucket_of = [&](int tile) {
    const int c = tile_off[tile + 1] - tile_off[tile];
    return min(c >> 6, 63);
  };
  for (int i = threadIdx.x; i < tiles; i += blockDim.x) atomicAdd(&hist[bucket_of(tile_of(i))], 1);
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int b = 63; b >= 0; --b) { base[b] = run; run += hist[b]; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < tiles; i += blockDim.x) {
    const int tile = tile_of(i);
    order[atomicAdd(&base[bucket_of(tile)], 1)] = tile;
  }
}

// Work items of a band (see k_raster): tiles with more than 1.5 slices of candidates are cut into slices
// (<= kSlice candidates each) as long as the scratch slots last; slices first, then the whole tiles heaviest
// first.  heavy[h] = (tile, first scratch slot, slices).  counters: [0] items, [1] heavy tiles.
// (kTargetItems, cfg-3a cycle, 1.4 M candidates over 4096 tiles: slices of 1024 / 512 / 256 candidates -- targets 2048 / 3072..5120 /
// >= 6144 -- raster + merge 288 + 29 / 224 + 37 / 206 + 63 us)
constexpr int kSlice = 1024, kTargetItems = 4096;
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// one workgroup of any size that is a multiple of 64
__device__ void tile_items_body(const int32_t* __restrict__ tile_off, Frame F, int ty_begin,
                                int ty_rows, int n_clouds, int max_slots, int target_items,
                                int4* __restrict__ items, int4* __restrict__ heavy,
                                int32_t* __restrict__ counters) {
  __shared__ int hist[64], base[64];
  __shared__ int s_slots, s_heavy, s_slice;
  const int T = F.Tx, TY = F.Ty;
  const int tiles = n_clouds * T * ty_rows;
  auto tile_of = [&](int i) { return ((i / (T * ty_rows)) * TY + ty_begin + (i / T) % ty_rows) * T + i % T; };
  auto count_of = [&](int tile) { return tile_off[tile + 1] - tile_off[tile]; };
  // slice size: total candidates / kTargetItems (so that a small band still yields enough work items to
  // fill the chip), within [256, kSlice], doubled until the slices of all cut tiles fit the scratch slots.
  // A thread's first four tiles (all of them up to 4096 tiles) are read once and kept in registers: every
  // pass below would otherwise start with the same global-memory latency.
  int cc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = threadIdx.x + k * (int)blockDim.x;
    cc[k] = i < tiles ? count_of(tile_of(i)) : 0;
  }
  // visit(f): f(i, tile, count) for every tile of this thread
  auto visit = [&](auto&& f) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = threadIdx.x + k * (int)blockDim.x;
      if (i < tiles) f(i, tile_of(i), cc[k]);
    }
    for (int i = threadIdx.x + 4 * (int)blockDim.x; i < tiles; i += blockDim.x) f(i, tile_of(i), count_of(tile_of(i)));
  };
  if (threadIdx.x == 0) s_slots = 0;
  __syncthreads();
  {
    int mine = 0;
    visit([&](int, int, int c) { mine += c; });
    mine = wave_sum_i(mine);                              // (a thousand same-address LDS atomics are ~4 us of this single-workgroup kernel)
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&s_slots, mine);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int sl0 = (s_slots / target_items + 255) / 256 * 256;
    s_slice = sl0 < 256 ? 256 : (sl0 > kSlice ? kSlice : sl0);
  }
  __syncthreads();
  for (int round = 0; round < 16; ++round) {
    if (threadIdx.x == 0) s_slots = 0;
    __syncthreads();
    const int sl = s_slice;
    int mine = 0;
    visit([&](int, int, int c) { if (2 * c > 3 * sl) mine += (c + sl - 1) / sl; });
    mine = wave_sum_i(mine);                              // (a thousand same-address LDS atomics are ~4 us of this single-workgroup kernel)
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&s_slots, mine);
    __syncthreads();
    const bool fits = s_slots <= max_slots;
    __syncthreads();
    if (fits) break;
    if (threadIdx.x == 0) s_slice = sl * 2;
    __syncthreads();
  }
  const int sl = s_slice;
  auto slices_of = [&](int c) { return (2 * c > 3 * sl && max_slots > 0) ? (c + sl - 1) / sl : 1; };
  if (threadIdx.x < 64) hist[threadIdx.x] = 0;
  if (threadIdx.x == 0) { s_slots = 0; s_heavy = 0; }
  __syncthreads();
  auto bucket_of = [&](int c) { return min(c >> 6, 63); };
  visit([&](int, int tile, int c) {
    const int ns = slices_of(c);
    if (ns > 1) {
      const int b0 = atomicAdd(&s_slots, ns);
      heavy[atomicAdd(&s_heavy, 1)] = make_int4(tile, b0, ns, 0);
      for (int k = 0; k < ns; ++k) items[b0 + k] = make_int4(tile, k, ns, b0 + k);
    } else {
      atomicAdd(&hist[bucket_of(c)], 1);
    }
  });
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = s_slots;
    for (int b = 63; b >= 0; --b) { base[b] = run; run += hist[b]; }
    counters[0] = run;
    counters[1] = s_heavy;
  }
  __syncthreads();
  visit([&](int, int tile, int c) {
    if (slices_of(c) == 1) items[atomicAdd(&base[bucket_of(c)], 1)] = make_int4(tile, 0, 1, 0);
  });
}
__global__ __launch_bounds__(1024) void k_tile_items(const int32_t* __restrict__ tile_off, Frame F, int ty_begin,
                                                     int ty_rows, int n_clouds, int max_slots, int target_items,
                                                     int4* __restrict__ items, int4* __restrict__ heavy,
                                                     int32_t* __restrict__ counters) {
  tile_items_body(tile_off, F, ty_begin, ty_rows, n_clouds, max_slots, target_items, items, heavy, counters);
}

// K-best of a heavy tile's pixels from the K-best lists of its slices (same (z, idx) order: the result
// does not depend on how the list was cut), then the depth-merging cut and the outputs of k_raster.
template <int KMAX, bool KFULL = false>
__global__ __launch_bounds__(256) void k_raster_merge(const int4* __restrict__ heavy, const int32_t* __restrict__ counters,
                                                      const float* __restrict__ scratch, Frame F, int K_arg,
                                                      float depth_thres, int32_t* __restrict__ idx_out,
                                                      float* __restrict__ zbuf_out, float* __restrict__ q_out,
                                                      float* __restrict__ occ_out, CompositeArgs ca) {
  const int K = KFULL ? KMAX : K_arg;
  const int nh = counters[1];
  for (int hI = blockIdx.x; hI < nh; hI += gridDim.x) {
    const int4 hv = heavy[hI];
    const int tile = hv.x;
    const int tx = tile % F.Tx, ty = (tile / F.Tx) % F.Ty, n = tile / (F.Tx * F.Ty);
    const int lx = threadIdx.x % TILE, ly = threadIdx.x / TILE;
    const int xi = tx * TILE + lx, yi = ty * TILE + ly;
    PixK<KMAX> best;
    best.init();
    for (int sI = 0; sI < hv.z; ++sI) {
      // a slice's list is requested whole before any of it is used (the id / q loads sat behind the depth test);
      // the first slice's list IS the K-best so far
      const float* sc = scratch + (int64_t)(hv.y + sI) * 3 * KMAX * 256;
      float zz[KMAX], qq[KMAX];
      int ii[KMAX];
#pragma unroll
      for (int j = 0; j < KMAX; ++j) {
        zz[j] = sc[j * 256 + threadIdx.x];
        qq[j] = sc[(KMAX + j) * 256 + threadIdx.x];
        ii[j] = __float_as_int(sc[(2 * KMAX + j) * 256 + threadIdx.x]);
      }
      if (sI == 0) {
#pragma unroll
        for (int j = 0; j < KMAX; ++j) { best.z[j] = zz[j]; best.q[j] = qq[j]; best.id[j] = ii[j]; }
      } else {
#pragma unroll
        for (int j = 0; j < KMAX; ++j)
          if (j < K && zz[j] < FLT_MAX) best.push_q(zz[j], ii[j], qq[j], K);   // (slice lists hold depths >= +0)
      }
    }
    if (xi >= F.W || yi >= F.H) continue;
    const int yo = F.H - 1 - yi, xo = F.W - 1 - xi;
    const int64_t pix = ((int64_t)n * F.H + yo) * F.W + xo;
    const float z0 = best.z[0];
    const bool hit = z0 < FLT_MAX;
    occ_out[pix] = hit ? 1.0f : 0.0f;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      if (j < K) {
        const bool ok = best.z[j] < FLT_MAX && !((best.z[j] - z0) > depth_thres);
        idx_out[pix * K + j] = ok ? best.id[j] : -1;
        zbuf_out[pix * K + j] = ok ? best.z[j] : -1.0f;
        q_out[pix * K + j] = ok ? best.q[j] : -1.0f;
        if (ok && ca.vis) ca.vis[best.id[j]] = 1;
      }
    }
    if (ca.scaler) composite_pixel<KMAX>(best, K, z0, depth_thres, hit, ca, pix);
  }
}

// ---------------------------------------------------------------- compositing
// w = exp(-0.5 q) * scaler[idx]; out[...,c] = sum w f / max(sum w, eps) ; out[...,C] = occupancy
// The same compositing with the K fragments of a pixel in registers (K = 4 or 8, C <= 3: the cycle's shapes): the lists are
// read with 16-byte loads and ALL the gathers of a pixel (scaler + features of K points) are requested before the first is
// used -- the generic loop below waits for each fragment's gathers in turn (119 -> see profiles/HISTORY.md, round 5).
// Same operations in the same order per pixel: bit-identical to k_composite.
template <int K>
__global__ __launch_bounds__(256) void k_composite_k(const int32_t* __restrict__ idx, const float* __restrict__ qv,
                                                     const float* __restrict__ occ, const float* __restrict__ scaler,
                                                     const float* __restrict__ feat, int C, int norm, float eps,
                                                     int64_t npix, float* __restrict__ frag_scaler, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix) return;
  int p[K];
  float q[K];
#pragma unroll
  for (int v = 0; v < K / 4; ++v) {
    const int4 a = reinterpret_cast<const int4*>(idx + i * K)[v];
    const float4 b = reinterpret_cast<const float4*>(qv + i * K)[v];
    p[4 * v] = a.x; p[4 * v + 1] = a.y; p[4 * v + 2] = a.z; p[4 * v + 3] = a.w;
    q[4 * v] = b.x; q[4 * v + 1] = b.y; q[4 * v + 2] = b.z; q[4 * v + 3] = b.w;
  }
  float s[K], f[K][3];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int64_t pc = p[k] >= 0 ? p[k] : 0;               // (a point always exists when any fragment does; row 0 otherwise unused)
    s[k] = p[k] >= 0 ? scaler[pc] : 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) f[k][c] = (c < C && p[k] >= 0) ? feat[pc * C + c] : 0.f;
  }
  float sw = 0.f, acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (p[k] >= 0) {
      const float w = expf(-0.5f * q[k]) * s[k];
#pragma unroll
      for (int c = 0; c < 3; ++c) if (c < C) acc[c] += w * f[k][c];
      sw += w;
    }
  }
  if (frag_scaler) {
#pragma unroll
    for (int v = 0; v < K / 4; ++v)
      reinterpret_cast<float4*>(frag_scaler + i * K)[v] = make_float4(s[4 * v], s[4 * v + 1], s[4 * v + 2], s[4 * v + 3]);
  }
  float d = 1.0f;
  if (norm) d = sw > eps ? sw : eps;
#pragma unroll
  for (int c = 0; c < 3; ++c) if (c < C) out[i * (C + 1) + c] = norm ? acc[c] / d : acc[c];
  out[i * (C + 1) + C] = occ[i];
}

__global__ void k_composite(const int32_t* __restrict__ idx, const float* __restrict__ qv,
                            const float* __restrict__ occ, const float* __restrict__ scaler,
                            const float* __restrict__ feat, int K, int C, int norm, float eps,
                            int64_t npix, float* __restrict__ frag_scaler /* may be null */,
                            float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix;
       i += (int64_t)gridDim.x * blockDim.x) {
    float sw = 0.f;
    float acc[8];
    for (int c = 0; c < C; ++c) acc[c] = 0.f;
    for (int k = 0; k < K; ++k) {
      const int p = idx[i * K + k];
      float s = 0.f, w = 0.f;
      if (p >= 0) {
        s = scaler[p];
        w = expf(-0.5f * qv[i * K + k]) * s;
        for (int c = 0; c < C; ++c) acc[c] += w * feat[(int64_t)p * C + c];
        sw += w;
      }
      if (frag_scaler) frag_scaler[i * K + k] = s;
    }
    float d = 1.0f;
    if (norm) d = sw > eps ? sw : eps;
    for (int c = 0; c < C; ++c) out[i * (C + 1) + c] = norm ? acc[c] / d : acc[c];
    out[i * (C + 1) + C] = occ[i];
  }
}

// Backward of k_composite (SurfaceSplattingRenderer.forward, renderer.py:53-78, is differentiable with
// respect to the features and the fragment weights through pytorch3d's compositor):
//   w_k = exp(-q_k / 2) s_k,  A_c = sum_k w_k f_c[p_k],  d = max(sum w, eps),  out_c = A_c / d (norm) or A_c
//   dL/df_c[p_k] += g_c w_k / d
//   dL/dw_k       = sum_c g_c (f_c[p_k] - [sum w > eps] out_c) / d            (norm; d = 1 and no second term otherwise)
//   dL/dq_k = -w_k / 2 dL/dw_k ,  dL/ds[p_k] += exp(-q_k / 2) dL/dw_k ,  dL/docc = g_alpha
// Feature and scaler gradients are scattered with float atomics (as pytorch3d's compositor does).
__global__ void k_composite_backward(const int32_t* __restrict__ idx, const float* __restrict__ qv,
                                     const float* __restrict__ scaler, const float* __restrict__ feat,
                                     const float* __restrict__ grad_img, int K, int C, int norm, float eps, int64_t npix,
                                     float* __restrict__ grad_feat, float* __restrict__ grad_q,
                                     float* __restrict__ grad_scaler, float* __restrict__ grad_occ) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
    float g[8], acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { g[c] = c < C ? grad_img[i * (C + 1) + c] : 0.f; acc[c] = 0.f; }
    if (grad_occ) grad_occ[i] = grad_img[i * (C + 1) + C];
    float sw = 0.f;
    for (int k = 0; k < K; ++k) {
      const int p = idx[i * K + k];
      if (p >= 0) {
        const float w = expf(-0.5f * qv[i * K + k]) * scaler[p];
#pragma unroll
        for (int c = 0; c < 8; ++c) if (c < C) acc[c] += w * feat[(int64_t)p * C + c];
        sw += w;
      }
    }
    const float d = norm ? (sw > eps ? sw : eps) : 1.0f;
    const bool through = norm && sw > eps;
    float go = 0.f;                               // sum_c g_c out_c
#pragma unroll
    for (int c = 0; c < 8; ++c) if (c < C) go += g[c] * (acc[c] / d);
    for (int k = 0; k < K; ++k) {
      const int p = idx[i * K + k];
      float gq = 0.f;
      if (p >= 0) {
        const float e = expf(-0.5f * qv[i * K + k]);
        const float sc = scaler[p];
        const float w = e * sc;
        float gw = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (c < C) {
            const float f = feat[(int64_t)p * C + c];
            gw += g[c] * f;
            if (grad_feat && g[c] != 0.f) atomicAdd(&grad_feat[(int64_t)p * C + c], g[c] * w / d);
          }
        gw = (gw - (through ? go : 0.f)) / d;
        gq = -0.5f * w * gw;
        if (grad_scaler && gw != 0.f) atomicAdd(&grad_scaler[p], e * gw);
      }
      if (grad_q) grad_q[i * K + k] = gq;
    }
  }
}

// ---------------------------------------------------------------- backward
// visible[p] = 1 for every point listed in a pixel whose first slot is filled
// (rasterizer.py:850-856)
__global__ void k_mark_visible(const int32_t* __restrict__ idx0, int K, int64_t npix,
                               uint8_t* __restrict__ visible, int64_t view_stride = 0) {
  const int32_t* __restrict__ idx = idx0 + (int64_t)blockIdx.y * view_stride * K;   // (N views of a band: grid.y)
  if ((K & 3) == 0 && ((uintptr_t)idx & 15) == 0) {                    // four list entries per 16-byte load
    const int4* __restrict__ idx4 = reinterpret_cast<const int4*>(idx);
    const int64_t nq = npix * (K / 4);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += (int64_t)gridDim.x * blockDim.x) {
      const int4 p = idx4[i];
      if (p.x >= 0) visible[p.x] = 1;
      if (p.y >= 0) visible[p.y] = 1;
      if (p.z >= 0) visible[p.z] = 1;
      if (p.w >= 0) visible[p.w] = 1;
    }
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix;
       i += (int64_t)gridDim.x * blockDim.x) {
    if (idx[i * K] < 0) continue;
    for (int k = 0; k < K; ++k) {
      const int p = idx[i * K + k];
      if (p >= 0) visible[p] = 1;
    }
  }
}

// ZbufBackwardKernel (rasterize_points.cu:823-846): z_grad[idx] += grad_zbuf, zeros skipped,
// stop at the first idx < 0.  Atomic scatter (order-dependent like the reference's).
__global__ void k_zbuf_scatter(const int32_t* __restrict__ idx, const float* __restrict__ gz,
                               int K, int64_t npix, float* __restrict__ z_grad) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix;
       i += (int64_t)gridDim.x * blockDim.x) {
    for (int k = 0; k < K; ++k) {
      const float g = gz[i * K + k];
      if (g == 0.0f) continue;
      const int p = idx[i * K + k];
      if (p < 0) break;
      atomicAdd(&z_grad[p], g);
    }
  }
}

// coarse map of 8x8 pixel blocks that hold a non-zero occupancy gradient
constexpr int GB = 8;
struct BlkGeo { int NBx, NBy, NB2x, NB2y; };
static inline BlkGeo make_blk(int H, int W) {
  BlkGeo g;
  g.NBx = (W + GB - 1) / GB; g.NBy = (H + GB - 1) / GB; g.NB2x = (g.NBx + 7) / 8; g.NB2y = (g.NBy + 7) / 8;
  return g;
}

// Gradient maps of the occupancy image, three levels in one launch: one WAVE per 64x64-pixel super block, lane = one
// of its 8x8 blocks (lane % 8 = block column).  A lane reads the 8 rows of its block as 2 x 16 bytes each when the row
// is aligned and writes the block's pixel mask (bit 8 ry + rx: that pixel carries a gradient); the flags of a row of
// eight blocks are one byte (the ballot), the super block's flag says whether any of them is set.  Workgroup 0 also
// clears the few words the later passes count into (heavy-point list length, z scale): no clearing launches.
__global__ __launch_bounds__(256) void k_grad_maps(const float* __restrict__ grad_occ, Frame F, BlkGeo G, int N,
                                                   unsigned long long* __restrict__ pixmask /*(N, NBy, NBx)*/,
                                                   uint8_t* __restrict__ rowbytes /*(N, NBy, NB2x)*/,
                                                   uint8_t* __restrict__ blk2 /*(N, NB2y, NB2x)*/,
                                                   int32_t* __restrict__ clear_a, int n_clear_a,
                                                   int32_t* __restrict__ clear_b, int n_clear_b) {
  if (blockIdx.x == 0) {
    if ((int)threadIdx.x < n_clear_a) clear_a[threadIdx.x] = 0;
    if (clear_b && (int)threadIdx.x < n_clear_b) clear_b[threadIdx.x] = 0;
  }
  const int lane = threadIdx.x & 63;
  const int64_t total = (int64_t)N * G.NB2x * G.NB2y;
  for (int64_t sb = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); sb < total; sb += (int64_t)gridDim.x * 4) {
    const int sx = sb % G.NB2x, sy = (sb / G.NB2x) % G.NB2y, n = sb / ((int64_t)G.NB2x * G.NB2y);
    const int bx = sx * 8 + (lane & 7), by = sy * 8 + (lane >> 3);
    unsigned long long m = 0ull;
    if (bx < G.NBx && by < G.NBy) {
      const int x0 = bx * GB, x1 = min(F.W, x0 + GB);
      for (int y = by * GB; y < min(F.H, (by + 1) * GB); ++y) {
        const float* row = grad_occ + ((int64_t)n * F.H + y) * F.W;
        unsigned bits = 0u;
        if (x1 - x0 == GB && (((uintptr_t)(row + x0)) & 15) == 0) {
          const float4 a = *reinterpret_cast<const float4*>(row + x0), c = *reinterpret_cast<const float4*>(row + x0 + 4);
          bits = (unsigned)(a.x != 0.f) | ((unsigned)(a.y != 0.f) << 1) | ((unsigned)(a.z != 0.f) << 2) | ((unsigned)(a.w != 0.f) << 3) |
                 ((unsigned)(c.x != 0.f) << 4) | ((unsigned)(c.y != 0.f) << 5) | ((unsigned)(c.z != 0.f) << 6) | ((unsigned)(c.w != 0.f) << 7);
        } else {
          for (int x = x0; x < x1; ++x) bits |= (unsigned)(row[x] != 0.0f) << (x - x0);
        }
        m |= (unsigned long long)bits << (8 * (y - by * GB));
      }
      pixmask[((int64_t)n * G.NBy + by) * G.NBx + bx] = m;
    }
    const unsigned long long bal = __ballot(m != 0ull);
    if ((lane & 7) == 0 && by < G.NBy) rowbytes[((int64_t)n * G.NBy + by) * G.NB2x + sx] = (uint8_t)((bal >> (8 * (lane >> 3))) & 0xffull);
    if (lane == 0) blk2[sb] = bal ? 1 : 0;
  }
}



// output-pixel range [lo,hi] (after the axis flip) whose centres may lie within c +- r
__device__ __forceinline__ bool out_range(float c, float r, int n, float e, int m, int& lo, int& hi) {
  int a, b;
  if (!pixel_range(c, r, n, e, m, a, b)) return false;
  lo = n - 1 - b;
  hi = n - 1 - a;
  return true;
}

// xy contribution of one pixel to one point (shared by both backward kernels)
__device__ __forceinline__ void occ_term(float g, float dx, float dy, float rx, float ry, float sx,
                                         float sy, float r2, int rect_mode, float radii_s, float& gx,
                                         float& gy) {
  if (g == 0.0f) return;
  const float dist2 = dx * dx + dy * dy;
  bool outside;
  if (rect_mode) {  // rasterize_points.cu:726-746 (slow CUDA kernel)
    if (fabsf(dx) > sx || fabsf(dy) > sy) return;
    outside = (fabsf(dx) > sx / radii_s) || (fabsf(dy) > sy / radii_s);
  } else {          // rasterize_points_backward.cu:156-161 (fast kernel)
    if (dist2 > r2) return;
    outside = (fabsf(dx) > rx) || (fabsf(dy) > ry);
  }
  if (g > 0.0f && outside) return;
  const float denom = iso_eps_denom(dist2, 1e-10f);
  gx += dx / denom * g;
  gy += dy / denom * g;
}

// Pass 1, one lane per point, the cheap part of the xy gradient: a point whose support touches no 64x64
// super block with a gradient is done (xy = 0); the others are appended to the heavy list.
__global__ __launch_bounds__(256) void k_splat_backward(
    const float* __restrict__ pts, const float* __restrict__ radii,
    const uint8_t* __restrict__ visible, const float* __restrict__ rs,
    const int64_t* __restrict__ first, const int64_t* __restrict__ num,
    const uint8_t* __restrict__ blk2, BlkGeo G, Frame F, int rect_mode, float radii_s,
    int32_t* __restrict__ heavy, int32_t* __restrict__ heavy_count, float* __restrict__ grad,
    long long* __restrict__ zacc /* fixed-point z accumulators, cleared here row by row (NULL: none) */,
    ZRider rider, int own_blocks) {
  if ((int)blockIdx.x >= own_blocks) {                 // riders: blockIdx.y == 0 only
    if (blockIdx.y == 0) z_absmax_body(rider.gz, rider.n, rider.zs, blockIdx.x - own_blocks, rider.blocks);
    return;
  }
  const int n = blockIdx.y;
  const int64_t len = num[n], base = first[n];
  const float r = rect_mode ? 0.f : rs[n];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // 1024 points per round and workgroup: the heavy list is appended to with ONE returning atomic per round
  // (one per 256 points made ~8 k same-address atomics the critical path of the kernel)
  __shared__ int s_wcnt[4][4], s_base;
  const int64_t span = (int64_t)own_blocks * 1024;
  for (int64_t i0 = (int64_t)blockIdx.x * 1024; i0 < len; i0 += span) {
    bool is_heavy[4];
    int64_t pp[4];
    unsigned long long bal[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t i = i0 + k * 256 + threadIdx.x;
      is_heavy[k] = false;
      pp[k] = -1;
      if (i < len) {
        const int64_t p = base + i;
        pp[k] = p;
        const float px = pts[p * 3], py = pts[p * 3 + 1], pz = pts[p * 3 + 2];
        const float rx = radii[p * 2], ry = radii[p * 2 + 1];
        const bool vis = (!visible || visible[p]);
        grad[p * 3] = 0.f; grad[p * 3 + 1] = 0.f; grad[p * 3 + 2] = 0.f;   // z: k_z_scatter / k_z_finish
        if (zacc) zacc[p] = 0;
        const float sx = rect_mode ? rx * radii_s : r, sy = rect_mode ? ry * radii_s : r;
        // (the fast kernel skips points outside the image: rasterize_points_backward.cu:101-103; here: outside the frame)
        if (vis && !(pz < 0.f || fabsf(py) > F.ey || fabsf(px) > F.ex) && sx > 0.f && sy > 0.f) {
          int x0, x1, y0, y1;
          if (out_range(px, sx, F.W, F.ex, F.m, x0, x1) && out_range(py, sy, F.H, F.ey, F.m, y0, y1)) {
            for (int sy2 = y0 / 64; sy2 <= y1 / 64 && !is_heavy[k]; ++sy2)
              for (int sx2 = x0 / 64; sx2 <= x1 / 64; ++sx2)
                if (blk2[((int64_t)n * G.NB2y + sy2) * G.NB2x + sx2]) { is_heavy[k] = true; break; }
          }
        }
      }
      bal[k] = __ballot(is_heavy[k]);
      if (lane == 0) s_wcnt[k][wv] = __popcll(bal[k]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int tot = 0;
#pragma unroll
      for (int q = 0; q < 16; ++q) tot += s_wcnt[q >> 2][q & 3];
      s_base = tot ? atomicAdd(heavy_count, tot) : 0;
    }
    __syncthreads();
    int b0 = s_base;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q == wv && is_heavy[k]) heavy[b0 + __popcll(bal[k] & ((1ull << lane) - 1ull))] = (int32_t)pp[k];
        b0 += s_wcnt[k][q];
      }
    }
    __syncthreads();
  }
}

// Pass 2, eight LANES per heavy point.  Measured on the cfg-3a cycle (tools/diag/heavy_count.py): 154 k heavy points whose
// discs (29.5 px radius: up to 81 blocks) hold 3.7 flagged blocks and 18 pixels with a gradient each -- the first form
// spent a wave per point (one pixel per lane, a block per round, a butterfly at the end): ~350 wave instructions for 18
// useful terms, the kernel was bound by exactly that (54 M VALU instructions per launch = 100 % of the issue rate).
// Here the eight lanes of a point read the row bytes of its disc (k_grad_maps; 4 block rows x 32 blocks per round, the
// words requested together) and walk its flagged blocks together; of a block's pixels that carry a gradient (pixel
// masks) lane j takes those on the diagonals (rx + ry) % 8 = j -- a silhouette arc crosses a block along a row, a column
// or the other diagonal, and dealing out block rows (or blocks) left such an arc to one or two lanes: 100 -> 43 us on
// the sparse gradient image of cfg 3a but 159 -> 228 us on the denser one of the SIREN cycle.  A lane adds its terms in
// image order, the eight partial sums meet in a fixed tree -> bit-stable, no atomics, independent of the launch geometry.
// (one lane per point: 100 -> 64 us, bound by the lane's chain of dependent loads at 2.4 waves per SIMD)
#ifndef SB_RY
#define SB_RY 4      // block rows whose flag words are requested together (1 / 2 / 4 / 8 / 16: 56 / 53 / 54 / 59 / 66 us on cfg 3a)
#endif
__global__ __launch_bounds__(256) void k_splat_backward_heavy(
    const float* __restrict__ pts, const float* __restrict__ radii, const float* __restrict__ rs,
    const int64_t* __restrict__ first, const int64_t* __restrict__ num, int n_clouds,
    const float* __restrict__ grad_occ, const unsigned long long* __restrict__ pixmask,
    const uint8_t* __restrict__ rowbytes, BlkGeo G, Frame F,
    int rect_mode, float radii_s, const int32_t* __restrict__ heavy,
    const int32_t* __restrict__ heavy_count, float* __restrict__ grad, ZRider rider, int own_blocks) {
  if ((int)blockIdx.x >= own_blocks) {
    z_finish_body(rider.acc, rider.zs, first, num, n_clouds, rider.grad, rider.terms_log2, blockIdx.x - own_blocks, rider.blocks);
    return;
  }
  const int count = *heavy_count;
  constexpr int LP = 8;                                   // lanes per point
  const int j = threadIdx.x & (LP - 1);
  unsigned long long diag = 0ull;                         // the pixels of an 8x8 block this lane takes: (rx + ry) % 8 = j
  for (int ry = 0; ry < 8; ++ry) diag |= 1ull << (8 * ry + ((j - ry) & 7));
  // (count rounded up: the lanes of a group leave the loop together, the shuffles below need all of them)
  for (int w = (blockIdx.x * blockDim.x + threadIdx.x) / LP; w < (count + 63) / 64 * 64; w += own_blocks * blockDim.x / LP) {
    const bool live = w < count;
    const int64_t p = live ? heavy[w] : 0;
    int n = 0;
    for (int c = 1; c < n_clouds; ++c) if (p >= first[c]) n = c;     // clouds are packed in order
    const float r = rect_mode ? 0.f : rs[n], r2 = r * r;
    const float px = pts[p * 3], py = pts[p * 3 + 1];
    const float rx = radii[p * 2], ry = radii[p * 2 + 1];
    const float sx = rect_mode ? rx * radii_s : r, sy = rect_mode ? ry * radii_s : r;
    int x0, x1, y0, y1;
    float gx = 0.f, gy = 0.f;
    if (live && out_range(px, sx, F.W, F.ex, F.m, x0, x1) && out_range(py, sy, F.H, F.ey, F.m, y0, y1)) {
      const float* __restrict__ gimg = grad_occ + (int64_t)n * F.H * F.W;
      const int bx0 = x0 / GB, bx1 = x1 / GB, by0 = y0 / GB, by1 = y1 / GB;
      constexpr int RY = SB_RY;                           // block rows per round: their flag words are requested together
      for (int cx = bx0 >> 3; cx <= bx1 >> 3; cx += 4) {  // 32 blocks (four row bytes = one unaligned word) per round
        // blocks of the window in this word, and of those the ones on this lane's diagonals
        const int ka = max(bx0 - cx * 8, 0), kb = min(bx1 - cx * 8, 31);
        const unsigned colwin = (0xffffffffu >> (31 - kb)) & (0xffffffffu << ka);
        for (int byc = by0; byc <= by1; byc += RY) {
          unsigned fl[RY];
#pragma unroll
          for (int t = 0; t < RY; ++t) {
            fl[t] = 0u;
            if (byc + t <= by1) {
              const uint8_t* rp = rowbytes + ((int64_t)n * G.NBy + byc + t) * G.NB2x + cx;
              unsigned wv;
              __builtin_memcpy(&wv, rp, 4);               // (bytes past the row's end belong to blocks outside the window)
              fl[t] = wv & colwin;
            }
          }
#pragma unroll
          for (int t = 0; t < RY; ++t) {
            const int by = byc + t;
            unsigned flags = fl[t];
            if (!flags) continue;
            // rows of the block inside the window, as a mask of whole bytes
            const int ra = max(y0 - by * GB, 0), rb = min(y1 - by * GB, GB - 1);
            const unsigned long long rowsel = (rb == 7 ? ~0ull : ((1ull << (8 * (rb + 1))) - 1ull)) & ~((1ull << (8 * ra)) - 1ull);
            while (flags) {
              const int k = __ffs((int)flags) - 1;
              flags &= flags - 1;
              const int bx = cx * 8 + k;
              // columns of the block inside the window, the same byte in every row
              const int ca = max(x0 - bx * GB, 0), cb = min(x1 - bx * GB, GB - 1);
              const unsigned long long colsel = (unsigned long long)((0xffu >> (7 - cb)) & (0xffu << ca)) * 0x0101010101010101ull;
              const unsigned long long win = rowsel & colsel;
              unsigned long long m = pixmask[((int64_t)n * G.NBy + by) * G.NBx + bx] & win;
              if (m == win && bx * GB + GB <= F.W && (F.W & 3) == 0 && (((uintptr_t)gimg) & 15) == 0) {
                // every pixel of the window's part of this block carries a gradient (a dense gradient image): lane j
                // takes pixel row j as two 16-byte loads, no bit scans
                if (j >= ra && j <= rb) {
                  const int yo = by * GB + j;
                  const float4* rowp = reinterpret_cast<const float4*>(gimg + (int64_t)yo * F.W + bx * GB);
                  const float4 ga = rowp[0], gb = rowp[1];
                  const float gv[8] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
                  const float dy = ndc_y(F.H - 1 - yo, F) - py;
#pragma unroll
                  for (int rxx = 0; rxx < GB; ++rxx)
                    if (rxx >= ca && rxx <= cb)
                      occ_term(gv[rxx], ndc_x(F.W - 1 - (bx * GB + rxx), F) - px, dy, rx, ry, sx, sy, r2, rect_mode, radii_s, gx, gy);
                }
                continue;
              }
              m &= diag;
              while (m) {
                const int b = __ffsll((long long)m) - 1;
                m &= m - 1;
                const int yo = by * GB + (b >> 3), xo = bx * GB + (b & 7);
                const float g = gimg[(int64_t)yo * F.W + xo];
                occ_term(g, ndc_x(F.W - 1 - xo, F) - px, ndc_y(F.H - 1 - yo, F) - py, rx, ry, sx, sy, r2, rect_mode, radii_s, gx, gy);
              }
            }
          }
        }
      }
    }
#pragma unroll
    for (int o = 1; o < LP; o <<= 1) {                    // fixed tree over the LP partial sums
      gx += __shfl_xor(gx, o);
      gy += __shfl_xor(gy, o);
    }
    if (live && j == 0) { grad[p * 3] = gx; grad[p * 3 + 1] = gy; }
  }
}

// ---------------------------------------------------------------- median radius (radix select)
// r_n = lower-median(radii of the visible points of cloud n, both columns) * radii_s
// (rasterizer.py:884; torch.median of the flattened (n,2) tensor).  Radii are >= 0, so their
// f32 bit patterns order like unsigned integers: three histogram passes over digits of 11 + 11 + 10 bits pin
// the k-th key.  Four launches: every pass resolves the digits decided so far from the earlier histograms itself
// (a 2048-bin prefix per workgroup -- cheaper than a launch in between), the last launch resolves all three,
// writes the result and leaves the histograms zeroed, which is the state the workspace must be in on entry.
constexpr int kRselBins = 2048;
constexpr int kRselShift[3] = {21, 10, 0};
constexpr int kRselBits[3] = {11, 11, 10};

// Workgroup-wide (256 lanes): prefix and remaining rank after the first `passes` digits of cloud-local histograms
// h[3][kRselBins].  Returns false when the cloud has no visible value.  All lanes get the same result.
// (pass-major layout hist[3][n_clouds][kRselBins]: one pass's histograms are contiguous -- the N-rank path sums
// exactly that slice over the ranks between two passes)
__device__ bool rsel_resolve(const unsigned* __restrict__ hist, int n, int n_clouds, int passes, unsigned& prefix,
                             long long& k, long long& cnt, long long* s_w /*[4]*/, long long* s_pub /*[2]*/) {
  prefix = 0u; k = 0; cnt = 0;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  for (int q = 0; q < passes; ++q) {
    const unsigned* hq = hist + ((int64_t)q * n_clouds + n) * kRselBins;
    long long v[8], sum = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) { v[e] = hq[t * 8 + e]; sum += v[e]; }
    long long inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const long long u = __shfl_up(inc, o); if (lane >= o) inc += u; }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    long long base = 0, tot = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { if (j < w) base += s_w[j]; tot += s_w[j]; }
    if (q == 0) { cnt = tot; k = tot > 0 ? (tot - 1) / 2 : 0; }
    if (cnt == 0) { __syncthreads(); return false; }
    long long ex = base + inc - sum;                    // values in the bins before this lane's eight
    if (k >= ex && k < ex + sum) {                      // exactly one lane holds rank k
      int bsel = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) { if (k >= ex + v[e] && e < 7 && bsel == e) { ex += v[e]; bsel = e + 1; } }
      s_pub[0] = t * 8 + bsel; s_pub[1] = ex;
    }
    __syncthreads();
    prefix |= ((unsigned)s_pub[0]) << kRselShift[q];
    k -= s_pub[1];
    __syncthreads();
  }
  return true;
}

template <int PASS>
__global__ __launch_bounds__(256) void k_rsel_hist(const float* __restrict__ radii,
                                                   const uint8_t* __restrict__ visible,
                                                   const int64_t* __restrict__ first,
                                                   const int64_t* __restrict__ num,
                                                   unsigned* __restrict__ hist /*[3][n_clouds][kRselBins]*/) {
  __shared__ unsigned lh[kRselBins];
  __shared__ long long s_w[4], s_pub[2];
  const int n = blockIdx.y, n_clouds = gridDim.y;
  unsigned prefix = 0u;
  long long k, cnt;
  if (PASS > 0 && !rsel_resolve(hist, n, n_clouds, PASS, prefix, k, cnt, s_w, s_pub)) return;
  for (int j = threadIdx.x; j < kRselBins; j += 256) lh[j] = 0u;
  __syncthreads();
  constexpr int shift = kRselShift[PASS];
  constexpr unsigned dmask = (1u << kRselBits[PASS]) - 1u;
  const unsigned hi_mask = PASS == 0 ? 0u : (0xffffffffu << (shift + kRselBits[PASS]));
  const int64_t len = num[n], base = first[n];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = base + i;
    if (!visible[p]) continue;
    const float2 r2 = *reinterpret_cast<const float2*>(radii + p * 2);
    const unsigned k0 = __float_as_uint(r2.x), k1 = __float_as_uint(r2.y);
    if ((k0 & hi_mask) == prefix) atomicAdd(&lh[(k0 >> shift) & dmask], 1u);
    if ((k1 & hi_mask) == prefix) atomicAdd(&lh[(k1 >> shift) & dmask], 1u);
  }
  __syncthreads();
  unsigned* hp = hist + ((int64_t)PASS * n_clouds + n) * kRselBins;
  for (int j = threadIdx.x; j < kRselBins; j += 256)
    if (lh[j]) atomicAdd(&hp[j], lh[j]);
}

// one 256-lane workgroup per cloud: all three digits, the result, and the histograms back to zero
__global__ __launch_bounds__(256) void k_rsel_final(unsigned* __restrict__ hist, int n_clouds, float radii_s,
                                                    float* __restrict__ out) {
  __shared__ long long s_w[4], s_pub[2];
  const int n = blockIdx.x;
  if (n >= n_clouds) return;
  unsigned prefix;
  long long k, cnt;
  const bool any = rsel_resolve(hist, n, n_clouds, 3, prefix, k, cnt, s_w, s_pub);
  if (threadIdx.x == 0) out[n] = any ? __uint_as_float(prefix) * radii_s : 0.0f;
  __syncthreads();
  for (int q = 0; q < 3; ++q)
    for (int j = threadIdx.x; j < kRselBins; j += 256) hist[((int64_t)q * n_clouds + n) * kRselBins + j] = 0u;
}

}  // namespace

// ===========================================================================
extern "C" int iso_splat_view_flags(const float* points, const float* normals,
                                    const float* views, int32_t* flags, int64_t P, int n_views,
                                    float znear, float zfar, int backface_culling, void* stream) {
  ISO_REQUIRE(P >= 0 && n_views >= 0, ISO_ERR_INVALID, "iso_splat_view_flags: bad sizes");
  if (P == 0 || n_views == 0) return ISO_OK;
  ISO_REQUIRE(points && views && flags && (normals || !backface_culling), ISO_ERR_INVALID,
              "iso_splat_view_flags: null pointer");
  int gx = iso_div_up(P, 256); if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(k_view_flags, dim3(gx, n_views), dim3(256), 0, (hipStream_t)stream, points,
                     normals, views, flags, P, znear, zfar, backface_culling);
  ISO_CHECK_LAUNCH("iso_splat_view_flags");
  return ISO_OK;
}

extern "C" int iso_compact_rows(const float* in, const int32_t* flags, const int32_t* offsets,
                                float* out, int64_t P, int64_t total, int U, void* stream) {
  ISO_REQUIRE(P >= 0 && total >= 0 && U >= 0, ISO_ERR_INVALID, "iso_compact_rows: bad sizes");
  if (total == 0 || U == 0) return ISO_OK;
  ISO_REQUIRE(in && flags && offsets && out && P > 0, ISO_ERR_INVALID, "iso_compact_rows: null pointer");
  hipLaunchKernelGGL(k_compact_rows, dim3(iso_stream_grid(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, in, flags, offsets, out, P, total, U);
  ISO_CHECK_LAUNCH("iso_compact_rows");
  return ISO_OK;
}

extern "C" int iso_splat_vrk_h(const float* dists, const int64_t* first_idx, const int64_t* num_pts,
                               const int64_t* cloud_num_pts, float* h, int n_clouds,
                               int64_t p_stride, void* stream) {
  ISO_REQUIRE(n_clouds >= 0 && p_stride >= 0, ISO_ERR_INVALID, "iso_splat_vrk_h: bad sizes");
  if (n_clouds == 0 || p_stride == 0) return ISO_OK;
  ISO_REQUIRE(dists && first_idx && num_pts && h, ISO_ERR_INVALID, "iso_splat_vrk_h: null pointer");
  int gx = iso_div_up(p_stride, 256); if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(k_vrk_h, dim3(gx, n_clouds), dim3(256), 0, (hipStream_t)stream, dists,
                     first_idx, num_pts, cloud_num_pts, h, p_stride);
  ISO_CHECK_LAUNCH("iso_splat_vrk_h");
  return ISO_OK;
}

extern "C" int iso_splat_setup(const float* points, const float* normals, const float* h,
                               const int64_t* first_idx, const int64_t* num_pts,
                               const float* views, const float* projs, int n_views,
                               int64_t max_pts, int image_size, float sigma, float cutoff,
                               float* ndc_out, float* ellipse_out, float* cutoff_out,
                               float* radii_out, float* scaler_out, void* stream) {
  ISO_REQUIRE(n_views >= 0 && max_pts >= 0 && image_size > 0, ISO_ERR_INVALID, "iso_splat_setup: bad sizes");
  if (n_views == 0 || max_pts == 0) return ISO_OK;
  ISO_REQUIRE(points && normals && h && first_idx && num_pts && views && projs && ndc_out &&
                  ellipse_out && cutoff_out && radii_out && scaler_out,
              ISO_ERR_INVALID, "iso_splat_setup: null pointer");
  SetupOut o{ndc_out, ellipse_out, cutoff_out, radii_out, scaler_out};
  int gx = iso_div_up(max_pts, 256); if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(k_splat_setup, dim3(gx, n_views), dim3(256), 0, (hipStream_t)stream, points,
                     normals, h, first_idx, num_pts, views, projs, image_size, sigma, cutoff, o);
  ISO_CHECK_LAUNCH("iso_splat_setup");
  return ISO_OK;
}

extern "C" int64_t iso_splat_front_workspace_bytes(int64_t n_points) {
  if (n_points < 0) n_points = 0;
  // [chunk table 8 x (n_chunks + 1) ints][per-tile counts 8 x (4 n_chunks + 4) ints: filled by a projection launch with
  //  iso_follow, scanned into the chunk table by iso_bricks_build_pending]
  const int64_t n_chunks = (n_points + kChunk - 1) / kChunk;
  return 8 * 4 * (n_chunks + 1) + 8 * 4 * (4 * n_chunks + 4);
}

// ---- gradient of the packed NDC rows w.r.t. the world points ---------------------------------------
// Row (v, i) of the front end is ndc = (xv / t, yv / t, zv) with [xv, yv, ., t] = [p, 1] M_v and zv = [p, 1] V_v[:, 2]
// (splat_setup_point; SurfaceSplatting.transform, rasterizer.py:565-582 -> cameras.transform_points), so
//   d L / d p_j = sum_v  gx (M[j][0] - ndc_x M[j][3]) / t + gy (M[j][1] - ndc_y M[j][3]) / t + gz V[j][2].
// Point-major: thread = point, views in ascending order (deterministic); the row of (v, i) is found by binary
// search in the view's ascending src list.
__global__ void k_splat_points_backward(const float* __restrict__ pts, int64_t n, const float* __restrict__ views,
                                        const float* __restrict__ projs, int n_views, const int32_t* __restrict__ mask,
                                        const int32_t* __restrict__ src, const int64_t* __restrict__ first,
                                        const int64_t* __restrict__ num, const float* __restrict__ grad_ndc,
                                        float* __restrict__ grad_pts) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float x = pts[i * 3], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
    const int m = mask[i];
    float gx = 0.f, gy = 0.f, gz = 0.f;
    for (int v = 0; v < n_views; ++v) {
      if (!((m >> v) & 1)) continue;
      const int32_t* sv = src + first[v];
      int64_t lo = 0, hi = num[v];
      while (lo < hi) {                       // first row whose source point is >= i
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)sv[mid] < i) lo = mid + 1; else hi = mid;
      }
      if (lo >= num[v] || (int64_t)sv[lo] != i) continue;      // (cannot happen for a mask / src pair of one front end)
      const int64_t row = first[v] + lo;
      const float* M = projs + v * 16;
      const float* V = views + v * 16;
      const float xv = ((x * M[0] + y * M[4]) + z * M[8]) + M[12];
      const float yv = ((x * M[1] + y * M[5]) + z * M[9]) + M[13];
      const float t = ((x * M[3] + y * M[7]) + z * M[11]) + M[15];
      const float it = 1.0f / iso_eps_denom(t, 1e-17f);
      const float nx = xv * it, ny = yv * it;
      const float g0 = grad_ndc[row * 3] * it, g1 = grad_ndc[row * 3 + 1] * it, g2 = grad_ndc[row * 3 + 2];
      gx += (g0 * (M[0] - nx * M[3]) + g1 * (M[1] - ny * M[3])) + g2 * V[2];
      gy += (g0 * (M[4] - nx * M[7]) + g1 * (M[5] - ny * M[7])) + g2 * V[6];
      gz += (g0 * (M[8] - nx * M[11]) + g1 * (M[9] - ny * M[11])) + g2 * V[10];
    }
    grad_pts[i * 3] = gx; grad_pts[i * 3 + 1] = gy; grad_pts[i * 3 + 2] = gz;
  }
}

extern "C" int iso_splat_points_backward(const float* points, int64_t n_points, const float* views, const float* projs,
                                         int n_views, const int32_t* mask, const int32_t* src, const int64_t* first_idx,
                                         const int64_t* num_points, const float* grad_ndc, float* grad_points,
                                         void* stream) {
  ISO_REQUIRE(n_points >= 0 && n_views >= 1 && n_views <= 8, ISO_ERR_INVALID, "iso_splat_points_backward: bad sizes (1..8 views)");
  if (n_points == 0) return ISO_OK;
  ISO_REQUIRE(points && views && projs && mask && src && first_idx && num_points && grad_ndc && grad_points, ISO_ERR_INVALID,
              "iso_splat_points_backward: null pointer");
  hipLaunchKernelGGL(k_splat_points_backward, dim3(iso_stream_grid(n_points, 256)), dim3(256), 0, (hipStream_t)stream, points,
                     n_points, views, projs, n_views, mask, src, first_idx, num_points, grad_ndc, grad_points);
  ISO_CHECK_LAUNCH("iso_splat_points_backward");
  return ISO_OK;
}

extern "C" int iso_splat_front(const float* points, const float* normals, const float* features, int channels,
                               int features_from_normals, const int32_t* mask, const float* h, int64_t n_points,
                               const float* views, const float* projs, int n_views, int image_size, float sigma,
                               float cutoff, void* workspace, int64_t workspace_bytes, int64_t* first_idx_out,
                               int64_t* num_pts_out, int32_t* view_total_out, float* ndc_out, float* ellipse_out,
                               float* cutoff_out, float* radii_out, float* scaler_out, float* features_out,
                               int32_t* src_out, void* stream) {
  ISO_REQUIRE(n_points >= 0 && n_views >= 1 && n_views <= 8 && image_size > 0, ISO_ERR_INVALID,
              "iso_splat_front: bad sizes (1..8 views per call)");
  ISO_REQUIRE(channels >= 0 && channels <= 8, ISO_ERR_UNSUPPORTED, "iso_splat_front: channels must be <= 8");
  ISO_REQUIRE(!features_from_normals || channels == 3, ISO_ERR_INVALID, "iso_splat_front: features_from_normals needs 3 channels");
  ISO_REQUIRE(first_idx_out && num_pts_out && view_total_out && workspace, ISO_ERR_INVALID, "iso_splat_front: null pointer");
  ISO_REQUIRE(workspace_bytes >= iso_splat_front_workspace_bytes(n_points), ISO_ERR_WORKSPACE,
              "iso_splat_front: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int n_chunks = (int)((n_points + kChunk - 1) / kChunk);
  int32_t* chunk = (int32_t*)workspace;
  if (n_points > 0) {
    ISO_REQUIRE(points && normals && mask && h && views && projs && ndc_out && ellipse_out && cutoff_out && radii_out &&
                    scaler_out && (!features_out || features || features_from_normals),
                ISO_ERR_INVALID, "iso_splat_front: null pointer");
    hipLaunchKernelGGL(k_mask_chunk_count, dim3(n_chunks), dim3(256), 0, s, mask, n_points, n_views, n_chunks, chunk);
  }
  hipLaunchKernelGGL(k_mask_chunk_scan, dim3(1), dim3(1024), 0, s, chunk, n_chunks, n_views, first_idx_out, num_pts_out,
                     view_total_out);
  if (n_points > 0) {
    FrontOut o{ndc_out, ellipse_out, cutoff_out, radii_out, scaler_out, features_out, src_out, nullptr, 0, nullptr};
    hipLaunchKernelGGL(k_splat_front, dim3(n_chunks), dim3(256), 0, s, points, normals, features, channels, mask, h,
                       n_points, chunk, n_chunks, first_idx_out, views, projs, n_views, image_size, sigma, cutoff,
                       features_from_normals, o);
  }
  ISO_CHECK_LAUNCH("iso_splat_front");
  return ISO_OK;
}

// The renderable mask, its per-chunk counts and their scan in two launches: what iso_splat_view_mask +
// the first two launches of iso_splat_front do in five.  The workspace (iso_splat_front_workspace_bytes) then holds
// the scanned chunk table iso_splat_front_rows expects; first_idx / num_points / view_total are final.
extern "C" int iso_splat_view_mask_scan(const float* points, const float* normals, const float* views, int n_views,
                                        int64_t n_points, float znear, float zfar, int backface_culling,
                                        int32_t* mask_out, void* workspace, int64_t workspace_bytes,
                                        int64_t* first_idx_out, int64_t* num_pts_out, int32_t* view_total_out,
                                        void* stream) {
  ISO_REQUIRE(n_points >= 0 && n_views >= 1 && n_views <= 8, ISO_ERR_UNSUPPORTED, "iso_splat_view_mask_scan: 1..8 views per call");
  ISO_REQUIRE(views && workspace && first_idx_out && num_pts_out && view_total_out &&
                  (n_points == 0 || (points && mask_out && (normals || !backface_culling))),
              ISO_ERR_INVALID, "iso_splat_view_mask_scan: null pointer");
  ISO_REQUIRE(workspace_bytes >= iso_splat_front_workspace_bytes(n_points), ISO_ERR_WORKSPACE,
              "iso_splat_view_mask_scan: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int n_chunks = (int)((n_points + kChunk - 1) / kChunk);
  int32_t* chunk = (int32_t*)workspace;
  if (n_points > 0)
    hipLaunchKernelGGL(k_view_mask_chunks, dim3(n_chunks), dim3(256), 0, s, points, normals, views, n_views, n_points, znear,
                       zfar, backface_culling, n_chunks, mask_out, chunk);
  hipLaunchKernelGGL(k_mask_chunk_scan, dim3(1), dim3(1024), 0, s, chunk, n_chunks, n_views, first_idx_out, num_pts_out,
                     view_total_out);
  ISO_CHECK_LAUNCH("iso_splat_view_mask_scan");
  return ISO_OK;
}

// iso_splat_front after iso_splat_view_mask_scan: the compaction + set-up pass alone (one launch); workspace,
// first_idx and mask are the ones that call produced.
extern "C" int iso_splat_front_rows(const float* points, const float* normals, const float* features, int channels,
                                    int features_from_normals, const int32_t* mask, const float* h, int64_t n_points,
                                    const float* views, const float* projs, int n_views, int image_size, float sigma,
                                    float cutoff, const void* workspace, int64_t workspace_bytes, const int64_t* first_idx,
                                    float* ndc_out, float* ellipse_out, float* cutoff_out, float* radii_out,
                                    float* scaler_out, float* features_out, int32_t* src_out, uint8_t* visible_zero_out,
                                    int64_t row_capacity, int32_t* overflow_out, void* stream) {
  ISO_REQUIRE(n_points >= 0 && n_views >= 1 && n_views <= 8 && image_size > 0, ISO_ERR_INVALID,
              "iso_splat_front_rows: bad sizes (1..8 views per call)");
  ISO_REQUIRE(channels >= 0 && channels <= 8, ISO_ERR_UNSUPPORTED, "iso_splat_front_rows: channels must be <= 8");
  ISO_REQUIRE(!features_from_normals || channels == 3, ISO_ERR_INVALID, "iso_splat_front_rows: features_from_normals needs 3 channels");
  ISO_REQUIRE(first_idx && workspace && workspace_bytes >= iso_splat_front_workspace_bytes(n_points), ISO_ERR_INVALID,
              "iso_splat_front_rows: workspace / first_idx missing");
  if (n_points == 0) return ISO_OK;
  ISO_REQUIRE(points && normals && mask && h && views && projs && ndc_out && ellipse_out && cutoff_out && radii_out &&
                  scaler_out && (!features_out || features || features_from_normals),
              ISO_ERR_INVALID, "iso_splat_front_rows: null pointer");
  const int n_chunks = (int)((n_points + kChunk - 1) / kChunk);
  FrontOut o{ndc_out, ellipse_out, cutoff_out, radii_out, scaler_out, features_out, src_out, visible_zero_out, row_capacity, overflow_out};
  hipLaunchKernelGGL(k_splat_front, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, points, normals, features, channels,
                     mask, h, n_points, (const int32_t*)workspace, n_chunks, first_idx, views, projs, n_views, image_size,
                     sigma, cutoff, features_from_normals, o);
  ISO_CHECK_LAUNCH("iso_splat_front_rows");
  return ISO_OK;
}

extern "C" int iso_splat_tiles_per_side(int image_size) { return (image_size + TILE - 1) / TILE; }

extern "C" int iso_splat_bin_count(const float* points, const float* radii,
                                   const int64_t* first_idx, const int64_t* num_pts, int n_clouds,
                                   int64_t max_pts, int image_size, int image_width, int tile_row_begin,
                                   int tile_row_end, int32_t* tile_cnt, void* stream) {
  ISO_REQUIRE(n_clouds >= 0 && max_pts >= 0 && image_size > 0, ISO_ERR_INVALID, "iso_splat_bin_count: bad sizes");
  if (n_clouds == 0 || max_pts == 0) return ISO_OK;
  ISO_REQUIRE(points && radii && first_idx && num_pts && tile_cnt, ISO_ERR_INVALID,
              "iso_splat_bin_count: null pointer");
  const Frame F = make_frame(image_size, image_width > 0 ? image_width : image_size);
  ISO_REQUIRE(tile_row_begin >= 0 && tile_row_begin <= tile_row_end && tile_row_end <= F.Ty,
              ISO_ERR_INVALID, "iso_splat_bin_count: bad tile row band");
  launch_bin<false>(points, radii, first_idx, num_pts, n_clouds, max_pts, F, tile_row_begin,
                    tile_row_end, tile_cnt, nullptr, nullptr, 0, nullptr, (hipStream_t)stream);
  ISO_CHECK_LAUNCH("iso_splat_bin_count");
  return ISO_OK;
}

namespace {
// exclusive scan of the n tile counters by ONE workgroup (n = views x tiles + 1: a few thousand entries -- three
// launches of a general multi-block scan for them were three launch latencies); the counters are cleared as they
// are read (the next binning pass finds them zero) and the fill cursors start at zero
__global__ __launch_bounds__(1024) void k_tile_offsets(int32_t* __restrict__ cnt, int32_t* __restrict__ off,
                                                       int32_t* __restrict__ cursor, int64_t n) {
  __shared__ int s_w[16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int carry = 0;
  for (int64_t c0 = 0; c0 < n; c0 += 4096) {
    const int64_t i0 = c0 + (int64_t)threadIdx.x * 4;
    int v[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] = 0;
      if (i0 + k < n) { v[k] = cnt[i0 + k]; cnt[i0 + k] = 0; cursor[i0 + k] = 0; }
      sum += v[k];
    }
    int inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { const int sv = s_w[k]; if (k < w) base += sv; tot += sv; }
    int ex = carry + base + inc - sum;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (i0 + k < n) off[i0 + k] = ex;
      ex += v[k];
    }
    carry += tot;
    __syncthreads();
  }
}
}  // namespace

extern "C" int iso_splat_tile_offsets(int32_t* tile_cnt, int32_t* tile_off, int32_t* tile_cursor, int64_t n, void* stream) {
  ISO_REQUIRE(n >= 0 && (n == 0 || (tile_cnt && tile_off && tile_cursor)), ISO_ERR_INVALID, "iso_splat_tile_offsets: bad arguments");
  if (n == 0) return ISO_OK;
  hipLaunchKernelGGL(k_tile_offsets, dim3(1), dim3(1024), 0, (hipStream_t)stream, tile_cnt, tile_off, tile_cursor, n);
  ISO_CHECK_LAUNCH("iso_splat_tile_offsets");
  return ISO_OK;
}

extern "C" int64_t iso_splat_forward_workspace_bytes(int64_t n_tiles, int points_per_pixel) {
  const int K = points_per_pixel;
  const int KM = K <= 4 ? 4 : (K <= 8 ? 8 : (K <= 16 ? 16 : 32));
  if (n_tiles < 0) n_tiles = 0;
  int64_t slots = n_tiles / 2 + 256;               // room to cut every other tile once; k_tile_items adapts the slice size
  if (slots < 2048) slots = 2048;                  // a small band is cut finer (enough work items to fill the chip)
  if (slots > 8192) slots = 8192;
  return 64 + 32 * n_tiles + slots * (16 + (int64_t)3 * KM * 256 * 4);
}

// ISO_RASTER_CP=0 in the environment selects the first form of the hit distribution (k_raster)
static bool raster_cp() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("ISO_RASTER_CP"); v = (e && e[0] == '0') ? 0 : 1; }
  return v == 1;
}

static int splat_forward_impl(CompositeArgs ca, const float* points, const float* ellipse, const float* cutoff,
                                 const float* radii, const int64_t* first_idx,
                                 const int64_t* num_pts, int n_clouds, int64_t max_pts,
                                 float depth_merging_thres, int image_size, int image_width, int points_per_pixel,
                                 int tile_row_begin, int tile_row_end, int32_t* tile_cursor,
                                 const int32_t* tile_off, int32_t* pairs,
                                 int64_t pair_capacity, int32_t* overflow_flag, int32_t* idx_out,
                                 float* zbuf_out, float* qvalue_out, float* occ_out, void* workspace,
                                 int64_t workspace_bytes, void* stream) {
  ISO_REQUIRE(points_per_pixel >= 1 && points_per_pixel <= 150, ISO_ERR_UNSUPPORTED,
              "iso_splat_forward: points_per_pixel must be in [1,150] (rasterization_utils.cuh:18), got %d", points_per_pixel);
  ISO_REQUIRE(n_clouds >= 0 && max_pts >= 0 && image_size > 0, ISO_ERR_INVALID, "iso_splat_forward: bad sizes");
  if (n_clouds == 0) return ISO_OK;
  ISO_REQUIRE(first_idx && num_pts && tile_cursor && tile_off && overflow_flag && idx_out &&
                  zbuf_out && qvalue_out && occ_out && (pairs || pair_capacity == 0),
              ISO_ERR_INVALID, "iso_splat_forward: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const Frame F = make_frame(image_size, image_width > 0 ? image_width : image_size);
  const int T = F.Tx;
  ISO_REQUIRE(tile_row_begin >= 0 && tile_row_begin <= tile_row_end && tile_row_end <= F.Ty,
              ISO_ERR_INVALID, "iso_splat_forward: bad tile row band");
  if (tile_row_begin == tile_row_end) return ISO_OK;
  const int ty_rows = tile_row_end - tile_row_begin;
  const int tiles = n_clouds * T * ty_rows;
  const int K = points_per_pixel;
  const int KM = K <= 4 ? 4 : (K <= 8 ? 8 : (K <= 16 ? 16 : 32));
  // workspace (optional): work items with the heavy tiles cut into slices
  //   [counters 64 B][items int4 (tiles + slots)][heavy int4 (tiles)][scratch slots * 3 * KM * 256 floats]
  int max_slots = 0;
  int4* items = nullptr; int4* heavy = nullptr; int32_t* counters = nullptr; float* scratch = nullptr;
  if (workspace && tiles > 0 && K <= 32) {
    const int64_t per_slot = 16 + (int64_t)3 * KM * 256 * 4;
    const int64_t fixed = 64 + (int64_t)32 * tiles;
    int64_t slots = workspace_bytes > fixed ? (workspace_bytes - fixed) / per_slot : 0;
    if (slots > 65536) slots = 65536;
    max_slots = (int)slots;
    counters = (int32_t*)workspace;
    items = (int4*)((char*)workspace + 64);
    heavy = items + tiles + max_slots;
    scratch = (float*)(heavy + tiles);
  }
  static int target = -1;               // ISO_RASTER_ITEMS: development override of the work-item target (sweeps)
  if (target < 0) { const char* e = getenv("ISO_RASTER_ITEMS"); target = e ? atoi(e) : kTargetItems; if (target < 1) target = kTargetItems; }
  bool items_done = false;
  if (max_pts > 0) {
    ISO_REQUIRE(points && ellipse && cutoff && radii, ISO_ERR_INVALID, "iso_splat_forward: null pointer");
    // the fill launch also makes the raster's work items (one more workgroup: they only need the offsets)
    items_done = launch_bin<true>(points, radii, first_idx, num_pts, n_clouds, max_pts, F, tile_row_begin,
                                  tile_row_end, tile_cursor, tile_off, pairs, pair_capacity, overflow_flag, s,
                                  TileItemsJob{ty_rows, n_clouds, max_slots, target, items, heavy, counters});
  }
  if (K > 32) {                               // lists too deep for registers: one workgroup per tile, lists in the outputs
    hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(1024), 0, s, tile_off, F, tile_row_begin, ty_rows, n_clouds, tile_cursor);
    hipLaunchKernelGGL(k_raster_deep, dim3(tiles), dim3(256), 0, s, points, ellipse, cutoff, radii, tile_cursor, tile_off, pairs,
                       pair_capacity, F, K, depth_merging_thres, idx_out, zbuf_out, qvalue_out, occ_out, ca);
    ISO_CHECK_LAUNCH("iso_splat_forward");
    return ISO_OK;
  }
  if (items) {
    if (!items_done)
      hipLaunchKernelGGL(k_tile_items, dim3(1), dim3(1024), 0, s, tile_off, F, tile_row_begin, ty_rows, n_clouds, max_slots,
                         target, items, heavy, counters);
  } else {
    // the fill cursors are dead now: their array takes the heaviest-first tile order
    hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(1024), 0, s, tile_off, F, tile_row_begin, ty_rows, n_clouds,
                       tile_cursor);
  }
#define ISO_LAUNCH_R(KM_)                                                                            \
  if (raster_cp() && K == KM_)                                                                   \
    hipLaunchKernelGGL((k_raster<KM_, true, true>), dim3(tiles + max_slots), dim3(256), 0, s, points, ellipse, cutoff, radii, \
                       tile_cursor, items, counters, scratch, tile_off, pairs, pair_capacity, F,                \
                       K, depth_merging_thres, idx_out, zbuf_out, qvalue_out, occ_out, ca);             \
  else if (raster_cp())                                                                                   \
    hipLaunchKernelGGL((k_raster<KM_, true>), dim3(tiles + max_slots), dim3(256), 0, s, points, ellipse, cutoff, radii, \
                       tile_cursor, items, counters, scratch, tile_off, pairs, pair_capacity, F,                \
                       K, depth_merging_thres, idx_out, zbuf_out, qvalue_out, occ_out, ca);             \
  else                                                                                               \
    hipLaunchKernelGGL((k_raster<KM_, false>), dim3(tiles + max_slots), dim3(256), 0, s, points, ellipse, cutoff, radii, \
                       tile_cursor, items, counters, scratch, tile_off, pairs, pair_capacity, F,                \
                       K, depth_merging_thres, idx_out, zbuf_out, qvalue_out, occ_out, ca);             \
  if (items && K == KM_)                                                                              \
    hipLaunchKernelGGL((k_raster_merge<KM_, true>), dim3(tiles < 1024 ? tiles : 1024), dim3(256), 0, s, heavy, counters, scratch, \
                       F, K, depth_merging_thres, idx_out, zbuf_out, qvalue_out, occ_out, ca);              \
  else if (items)                                                                                     \
    hipLaunchKernelGGL(k_raster_merge<KM_>, dim3(tiles < 1024 ? tiles : 1024), dim3(256), 0, s, heavy, counters, scratch, \
                       F, K, depth_merging_thres, idx_out, zbuf_out, qvalue_out, occ_out, ca)
  if (K <= 4) { ISO_LAUNCH_R(4); }
  else if (K <= 8) { ISO_LAUNCH_R(8); }
  else if (K <= 16) { ISO_LAUNCH_R(16); }
  else { ISO_LAUNCH_R(32); }
#undef ISO_LAUNCH_R
  ISO_CHECK_LAUNCH("iso_splat_forward");
  return ISO_OK;
}

extern "C" int iso_splat_forward(const float* points, const float* ellipse, const float* cutoff,
                                 const float* radii, const int64_t* first_idx,
                                 const int64_t* num_pts, int n_clouds, int64_t max_pts,
                                 float depth_merging_thres, int image_size, int image_width, int points_per_pixel,
                                 int tile_row_begin, int tile_row_end, int32_t* tile_cursor,
                                 const int32_t* tile_off, int32_t* pairs,
                                 int64_t pair_capacity, int32_t* overflow_flag, int32_t* idx_out,
                                 float* zbuf_out, float* qvalue_out, float* occ_out, void* workspace,
                                 int64_t workspace_bytes, void* stream) {
  CompositeArgs ca{nullptr, nullptr, nullptr, 0, 0, 0.f, nullptr};
  return splat_forward_impl(ca, points, ellipse, cutoff, radii, first_idx, num_pts, n_clouds, max_pts, depth_merging_thres,
                            image_size, image_width, points_per_pixel, tile_row_begin, tile_row_end, tile_cursor, tile_off, pairs,
                            pair_capacity, overflow_flag, idx_out, zbuf_out, qvalue_out, occ_out, workspace,
                            workspace_bytes, stream);
}

// iso_splat_forward + iso_splat_composite in one pass: the pixels' images are composited from the K-best
// lists while they are still in registers (same arithmetic and order as iso_splat_composite).
extern "C" int iso_splat_render(const float* points, const float* ellipse, const float* cutoff,
                                const float* radii, const int64_t* first_idx,
                                const int64_t* num_pts, int n_clouds, int64_t max_pts,
                                float depth_merging_thres, int image_size, int image_width, int points_per_pixel,
                                int tile_row_begin, int tile_row_end, int32_t* tile_cursor,
                                const int32_t* tile_off, int32_t* pairs,
                                int64_t pair_capacity, int32_t* overflow_flag, int32_t* idx_out,
                                float* zbuf_out, float* qvalue_out, float* occ_out, void* workspace,
                                int64_t workspace_bytes, const float* scaler, const float* features, int channels,
                                int norm_weighted, float eps, float* image_out, void* stream) {
  return iso_splat_render_visible(points, ellipse, cutoff, radii, first_idx, num_pts, n_clouds, max_pts, depth_merging_thres,
                                  image_size, image_width, points_per_pixel, tile_row_begin, tile_row_end, tile_cursor, tile_off,
                                  pairs, pair_capacity, overflow_flag, idx_out, zbuf_out, qvalue_out, occ_out, workspace,
                                  workspace_bytes, scaler, features, channels, norm_weighted, eps, image_out, nullptr, stream);
}

// ... and the visible flags of the backward pass (iso_splat_mark_visible) while the lists are written: visible_out (P,)
// uint8, ZERO on entry over the rows of the clouds (the front end clears them), 1 afterwards for every point that a
// pixel's list holds.
extern "C" int iso_splat_render_visible(const float* points, const float* ellipse, const float* cutoff,
                                        const float* radii, const int64_t* first_idx,
                                        const int64_t* num_pts, int n_clouds, int64_t max_pts,
                                        float depth_merging_thres, int image_size, int image_width, int points_per_pixel,
                                        int tile_row_begin, int tile_row_end, int32_t* tile_cursor,
                                        const int32_t* tile_off, int32_t* pairs,
                                        int64_t pair_capacity, int32_t* overflow_flag, int32_t* idx_out,
                                        float* zbuf_out, float* qvalue_out, float* occ_out, void* workspace,
                                        int64_t workspace_bytes, const float* scaler, const float* features, int channels,
                                        int norm_weighted, float eps, float* image_out, uint8_t* visible_out, void* stream) {
  ISO_REQUIRE(channels >= 0 && channels <= 8, ISO_ERR_UNSUPPORTED, "iso_splat_render: channels must be <= 8");
  ISO_REQUIRE(scaler && image_out && (features || channels == 0), ISO_ERR_INVALID, "iso_splat_render: null pointer");
  CompositeArgs ca{scaler, features, image_out, channels, norm_weighted, eps, visible_out};
  return splat_forward_impl(ca, points, ellipse, cutoff, radii, first_idx, num_pts, n_clouds, max_pts, depth_merging_thres,
                            image_size, image_width, points_per_pixel, tile_row_begin, tile_row_end, tile_cursor, tile_off, pairs,
                            pair_capacity, overflow_flag, idx_out, zbuf_out, qvalue_out, occ_out, workspace,
                            workspace_bytes, stream);
}

extern "C" int iso_rasterize_coarse(const float* points, const float* radii, const int64_t* first_idx,
                                    const int64_t* num_pts, int n_clouds, int image_size, int bin_size,
                                    int max_points_per_bin, int32_t* bin_points_out, int32_t* overflow_out, void* stream) {
  ISO_REQUIRE(n_clouds >= 0 && image_size > 0 && bin_size > 0 && max_points_per_bin >= 0, ISO_ERR_INVALID,
              "iso_rasterize_coarse: bad sizes");
  const int B = 1 + (image_size - 1) / bin_size;
  if (n_clouds == 0 || max_points_per_bin == 0) return ISO_OK;
  ISO_REQUIRE(points && radii && first_idx && num_pts && bin_points_out && overflow_out, ISO_ERR_INVALID,
              "iso_rasterize_coarse: null pointer");
  hipLaunchKernelGGL(k_coarse_bins, dim3(B * B, n_clouds), dim3(256), 0, (hipStream_t)stream, points, radii, first_idx,
                     num_pts, image_size, bin_size, B, max_points_per_bin, bin_points_out, overflow_out);
  ISO_CHECK_LAUNCH("iso_rasterize_coarse");
  return ISO_OK;
}

extern "C" int iso_rasterize_fine(const float* points, const float* ellipse, const float* cutoff, const float* radii,
                                  int64_t n_points, const int32_t* bin_points, int n_clouds, int max_points_per_bin,
                                  float depth_merging_thres, int image_size, int bin_size, int points_per_pixel,
                                  int32_t* idx_out, float* zbuf_out, float* qvalue_out, float* occ_out, void* stream) {
  ISO_REQUIRE(n_clouds >= 0 && image_size > 0 && bin_size > 0 && max_points_per_bin >= 0 && n_points >= 0, ISO_ERR_INVALID,
              "iso_rasterize_fine: bad sizes");
  ISO_REQUIRE(points_per_pixel >= 1 && points_per_pixel <= 150, ISO_ERR_UNSUPPORTED,
              "iso_rasterize_fine: points_per_pixel must be in [1,150] (rasterization_utils.cuh:18), got %d", points_per_pixel);
  if (n_clouds == 0) return ISO_OK;
  ISO_REQUIRE(idx_out && zbuf_out && qvalue_out && occ_out && (max_points_per_bin == 0 || bin_points) &&
                  (n_points == 0 || (points && ellipse && cutoff && radii)),
              ISO_ERR_INVALID, "iso_rasterize_fine: null pointer");
  const Frame F = make_frame(image_size, image_size);
  const int B = 1 + (image_size - 1) / bin_size;
  const int64_t npix = (int64_t)n_clouds * image_size * image_size;
  const int K = points_per_pixel;
  hipStream_t s = (hipStream_t)stream;
#define ISO_FINE(KM_)                                                                                               \
  hipLaunchKernelGGL(k_fine_bins<KM_>, dim3(iso_stream_grid(npix, 256)), dim3(256), 0, s, points, ellipse, cutoff, radii, \
                     bin_points, n_clouds, B, max_points_per_bin, bin_size, F, K, depth_merging_thres, n_points, idx_out,  \
                     zbuf_out, qvalue_out, occ_out)
  if (K <= 4) ISO_FINE(4);
  else if (K <= 8) ISO_FINE(8);
  else if (K <= 16) ISO_FINE(16);
  else if (K <= 32) ISO_FINE(32);
  else                                          // lists too deep for registers: kept in the output arrays
    hipLaunchKernelGGL(k_fine_bins_deep, dim3(iso_stream_grid(npix, 256)), dim3(256), 0, s, points, ellipse, cutoff, radii,
                       bin_points, n_clouds, B, max_points_per_bin, bin_size, F, K, depth_merging_thres, n_points, idx_out,
                       zbuf_out, qvalue_out, occ_out);
#undef ISO_FINE
  ISO_CHECK_LAUNCH("iso_rasterize_fine");
  return ISO_OK;
}

namespace {
// DSS/utils/__init__.py:172-185 (gather_with_neg_idx) for one float per point: out[i] = idx[i] >= 0 ? values[idx[i]] : 0.
// Four entries per thread (16-byte loads of idx, 16-byte stores).
__global__ __launch_bounds__(256) void k_gather_neg_idx(const float* __restrict__ values, const int32_t* __restrict__ idx,
                                                        int64_t n, float* __restrict__ out) {
  const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i0 + 3 < n) {
    const int4 k = *reinterpret_cast<const int4*>(idx + i0);
    float4 r;
    r.x = k.x >= 0 ? values[k.x] : 0.f;
    r.y = k.y >= 0 ? values[k.y] : 0.f;
    r.z = k.z >= 0 ? values[k.z] : 0.f;
    r.w = k.w >= 0 ? values[k.w] : 0.f;
    *reinterpret_cast<float4*>(out + i0) = r;
  } else {
    for (int64_t i = i0; i < n; ++i) { const int k = idx[i]; out[i] = k >= 0 ? values[k] : 0.f; }
  }
}
}  // namespace

extern "C" int iso_gather_neg_idx(const float* values, const int32_t* idx, int64_t n, float* out, void* stream) {
  ISO_REQUIRE(n >= 0, ISO_ERR_INVALID, "iso_gather_neg_idx: bad size");
  if (n == 0) return ISO_OK;
  ISO_REQUIRE(values && idx && out, ISO_ERR_INVALID, "iso_gather_neg_idx: null pointer");
  ISO_REQUIRE((((uintptr_t)idx | (uintptr_t)out) & 15) == 0, ISO_ERR_INVALID, "iso_gather_neg_idx: idx / out must be 16-byte aligned");
  hipLaunchKernelGGL(k_gather_neg_idx, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, values, idx, n, out);
  ISO_CHECK_LAUNCH("iso_gather_neg_idx");
  return ISO_OK;
}

extern "C" int iso_splat_composite(const int32_t* idx, const float* qvalue, const float* occ,
                                   const float* scaler, const float* features, int64_t n_pixels,
                                   int points_per_pixel, int channels, int norm_weighted, float eps,
                                   float* frag_scaler_out, float* image_out, void* stream) {
  ISO_REQUIRE(channels >= 0 && channels <= 8, ISO_ERR_UNSUPPORTED, "iso_splat_composite: channels must be <= 8");
  ISO_REQUIRE(n_pixels >= 0 && points_per_pixel >= 1, ISO_ERR_INVALID, "iso_splat_composite: bad sizes");
  if (n_pixels == 0) return ISO_OK;
  ISO_REQUIRE(idx && qvalue && occ && scaler && image_out && (features || channels == 0),
              ISO_ERR_INVALID, "iso_splat_composite: null pointer");
  const bool aligned = (((uintptr_t)idx | (uintptr_t)qvalue | (uintptr_t)frag_scaler_out) & 15) == 0;
  if (aligned && channels <= 3 && points_per_pixel == 8)
    hipLaunchKernelGGL(k_composite_k<8>, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, (hipStream_t)stream, idx, qvalue,
                       occ, scaler, features, channels, norm_weighted, eps, n_pixels, frag_scaler_out, image_out);
  else if (aligned && channels <= 3 && points_per_pixel == 4)
    hipLaunchKernelGGL(k_composite_k<4>, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, (hipStream_t)stream, idx, qvalue,
                       occ, scaler, features, channels, norm_weighted, eps, n_pixels, frag_scaler_out, image_out);
  else
  hipLaunchKernelGGL(k_composite, dim3(iso_stream_grid(n_pixels, 256)), dim3(256), 0,
                     (hipStream_t)stream, idx, qvalue, occ, scaler, features, points_per_pixel,
                     channels, norm_weighted, eps, n_pixels, frag_scaler_out, image_out);
  ISO_CHECK_LAUNCH("iso_splat_composite");
  return ISO_OK;
}

extern "C" int iso_splat_composite_backward(const int32_t* idx, const float* qvalue, const float* scaler,
                                            const float* features, const float* grad_image, int64_t n_pixels,
                                            int points_per_pixel, int channels, int norm_weighted, float eps,
                                            float* grad_features, float* grad_qvalue, float* grad_scaler,
                                            float* grad_occ, void* stream) {
  ISO_REQUIRE(channels >= 0 && channels <= 8, ISO_ERR_UNSUPPORTED, "iso_splat_composite_backward: channels must be <= 8");
  ISO_REQUIRE(n_pixels >= 0 && points_per_pixel >= 1, ISO_ERR_INVALID, "iso_splat_composite_backward: bad sizes");
  if (n_pixels == 0) return ISO_OK;
  ISO_REQUIRE(idx && qvalue && scaler && grad_image && (features || channels == 0), ISO_ERR_INVALID,
              "iso_splat_composite_backward: null pointer");
  hipLaunchKernelGGL(k_composite_backward, dim3(iso_stream_grid(n_pixels, 256)), dim3(256), 0, (hipStream_t)stream, idx,
                     qvalue, scaler, features, grad_image, points_per_pixel, channels, norm_weighted, eps, n_pixels,
                     grad_features, grad_qvalue, grad_scaler, grad_occ);
  ISO_CHECK_LAUNCH("iso_splat_composite_backward");
  return ISO_OK;
}

extern "C" int iso_splat_mark_visible(const int32_t* idx, int64_t n_pixels, int points_per_pixel,
                                      uint8_t* visible, void* stream) {
  ISO_REQUIRE(n_pixels >= 0 && points_per_pixel >= 1, ISO_ERR_INVALID, "iso_splat_mark_visible: bad sizes");
  if (n_pixels == 0) return ISO_OK;
  ISO_REQUIRE(idx && visible, ISO_ERR_INVALID, "iso_splat_mark_visible: null pointer");
  hipLaunchKernelGGL(k_mark_visible, dim3(iso_stream_grid(n_pixels, 256)), dim3(256), 0,
                     (hipStream_t)stream, idx, points_per_pixel, n_pixels, visible);
  ISO_CHECK_LAUNCH("iso_splat_mark_visible");
  return ISO_OK;
}

extern "C" int64_t iso_splat_median_radius_workspace_bytes(int n_clouds) {
  if (n_clouds < 0) n_clouds = 0;
  return (int64_t)n_clouds * 3 * kRselBins * 4 + 64;
}

extern "C" int iso_splat_median_radius(const float* radii, const uint8_t* visible,
                                       const int64_t* first_idx, const int64_t* num_pts,
                                       int n_clouds, int64_t max_pts, float radii_s,
                                       void* workspace, int64_t workspace_bytes,
                                       float* search_radius_out, void* stream) {
  ISO_REQUIRE(n_clouds >= 0 && max_pts >= 0, ISO_ERR_INVALID, "iso_splat_median_radius: bad sizes");
  if (n_clouds == 0) return ISO_OK;
  ISO_REQUIRE(first_idx && num_pts && workspace && search_radius_out && (max_pts == 0 || (radii && visible)),
              ISO_ERR_INVALID, "iso_splat_median_radius: null pointer");
  ISO_REQUIRE(workspace_bytes >= iso_splat_median_radius_workspace_bytes(n_clouds), ISO_ERR_WORKSPACE,
              "iso_splat_median_radius: workspace too small");
  ISO_REQUIRE(((uintptr_t)radii & 7) == 0, ISO_ERR_INVALID, "iso_splat_median_radius: radii must be 8-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  unsigned* hist = (unsigned*)workspace;      // zero on entry (contract), zero again on exit (k_rsel_final)
  int gx = iso_div_up(max_pts > 0 ? max_pts : 1, 256 * 8);
  if (gx > 512) gx = 512;
  if (max_pts > 0) {
    hipLaunchKernelGGL(k_rsel_hist<0>, dim3(gx, n_clouds), dim3(256), 0, s, radii, visible, first_idx, num_pts, hist);
    hipLaunchKernelGGL(k_rsel_hist<1>, dim3(gx, n_clouds), dim3(256), 0, s, radii, visible, first_idx, num_pts, hist);
    hipLaunchKernelGGL(k_rsel_hist<2>, dim3(gx, n_clouds), dim3(256), 0, s, radii, visible, first_idx, num_pts, hist);
  }
  hipLaunchKernelGGL(k_rsel_final, dim3(n_clouds), dim3(256), 0, s, hist, n_clouds, radii_s, search_radius_out);
  ISO_CHECK_LAUNCH("iso_splat_median_radius");
  return ISO_OK;
}

// The same select in pieces (N ranks: every rank counts its OWN visible rows; the pass's histograms -- the slice
// workspace + pass * n_clouds * 2048 words, n_clouds * 2048 words long -- are summed over the ranks before the next
// piece runs).  pass = 0, 1, 2, then iso_splat_median_final.  Same workspace contract (zero on entry / on exit).
extern "C" int64_t iso_splat_median_pass_words(int n_clouds) { return (int64_t)(n_clouds < 0 ? 0 : n_clouds) * kRselBins; }

extern "C" int iso_splat_median_pass(int pass, const float* radii, const uint8_t* visible, const int64_t* first_idx,
                                     const int64_t* num_pts, int n_clouds, int64_t max_pts, void* workspace,
                                     int64_t workspace_bytes, void* stream) {
  ISO_REQUIRE(pass >= 0 && pass <= 2 && n_clouds >= 0 && max_pts >= 0, ISO_ERR_INVALID, "iso_splat_median_pass: bad arguments");
  if (n_clouds == 0 || max_pts == 0) return ISO_OK;
  ISO_REQUIRE(radii && visible && first_idx && num_pts && workspace, ISO_ERR_INVALID, "iso_splat_median_pass: null pointer");
  ISO_REQUIRE(workspace_bytes >= iso_splat_median_radius_workspace_bytes(n_clouds), ISO_ERR_WORKSPACE,
              "iso_splat_median_pass: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  unsigned* hist = (unsigned*)workspace;
  int gx = iso_div_up(max_pts, 256 * 8);
  if (gx > 512) gx = 512;
  if (pass == 0) hipLaunchKernelGGL(k_rsel_hist<0>, dim3(gx, n_clouds), dim3(256), 0, s, radii, visible, first_idx, num_pts, hist);
  else if (pass == 1) hipLaunchKernelGGL(k_rsel_hist<1>, dim3(gx, n_clouds), dim3(256), 0, s, radii, visible, first_idx, num_pts, hist);
  else hipLaunchKernelGGL(k_rsel_hist<2>, dim3(gx, n_clouds), dim3(256), 0, s, radii, visible, first_idx, num_pts, hist);
  ISO_CHECK_LAUNCH("iso_splat_median_pass");
  return ISO_OK;
}

extern "C" int iso_splat_median_final(void* workspace, int n_clouds, float radii_s, float* search_radius_out, void* stream) {
  ISO_REQUIRE(n_clouds >= 0 && (n_clouds == 0 || (workspace && search_radius_out)), ISO_ERR_INVALID,
              "iso_splat_median_final: bad arguments");
  if (n_clouds == 0) return ISO_OK;
  hipLaunchKernelGGL(k_rsel_final, dim3(n_clouds), dim3(256), 0, (hipStream_t)stream, (unsigned*)workspace, n_clouds, radii_s,
                     search_radius_out);
  ISO_CHECK_LAUNCH("iso_splat_median_final");
  return ISO_OK;
}

extern "C" int iso_splat_zbuf_backward(const int32_t* idx, const float* grad_zbuf,
                                       int64_t n_pixels, int points_per_pixel, float* z_grad,
                                       void* stream) {
  ISO_REQUIRE(n_pixels >= 0 && points_per_pixel >= 1, ISO_ERR_INVALID, "iso_splat_zbuf_backward: bad sizes");
  if (n_pixels == 0) return ISO_OK;
  ISO_REQUIRE(idx && grad_zbuf && z_grad, ISO_ERR_INVALID, "iso_splat_zbuf_backward: null pointer");
  hipLaunchKernelGGL(k_zbuf_scatter, dim3(iso_stream_grid(n_pixels, 256)), dim3(256), 0,
                     (hipStream_t)stream, idx, grad_zbuf, points_per_pixel, n_pixels, z_grad);
  ISO_CHECK_LAUNCH("iso_splat_zbuf_backward");
  return ISO_OK;
}

// ---- z gradient by pixel-major scatter in 64-bit fixed point -------------------------------------
// z_grad[p] = sum of grad_zbuf over the slots that list p (ZbufBackwardKernel, rasterize_points.cu:
// 823-846: zeros skipped, a row ends at the first idx < 0).  The reference scatters with float
// atomics (order-dependent).  Here every contribution is converted exactly to a 64-bit integer
// multiple of q = 2^-e (e chosen from max|grad| so that 2^44 * q >= max|grad|), added with integer
// atomics -- associative, so the result does not depend on the order -- and converted back once:
// the exactly rounded sum (error <= 2^-44 max|grad| per term), bit-stable from run to run.
struct ZScale { unsigned max_bits; int exp2; };

__device__ void z_absmax_body(const float* __restrict__ gz, int64_t n, ZScale* __restrict__ zs, int block, int nblocks) {
  __shared__ unsigned sm[4];
  unsigned m = 0u;
  const int64_t n4 = n / 4;
  const uint4* g4 = reinterpret_cast<const uint4*>(gz);
  const int64_t stride = (int64_t)nblocks * blockDim.x;
  int64_t i = (int64_t)block * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {                    // four requests in flight
    const uint4 v0 = g4[i], v1 = g4[i + stride], v2 = g4[i + 2 * stride], v3 = g4[i + 3 * stride];
    const unsigned a = max(max(v0.x & 0x7fffffffu, v0.y & 0x7fffffffu), max(v0.z & 0x7fffffffu, v0.w & 0x7fffffffu));
    const unsigned b = max(max(v1.x & 0x7fffffffu, v1.y & 0x7fffffffu), max(v1.z & 0x7fffffffu, v1.w & 0x7fffffffu));
    const unsigned c = max(max(v2.x & 0x7fffffffu, v2.y & 0x7fffffffu), max(v2.z & 0x7fffffffu, v2.w & 0x7fffffffu));
    const unsigned d = max(max(v3.x & 0x7fffffffu, v3.y & 0x7fffffffu), max(v3.z & 0x7fffffffu, v3.w & 0x7fffffffu));
    m = max(m, max(max(a, b), max(c, d)));
  }
  for (; i < n4; i += stride) {
    const uint4 v = g4[i];                                          // |x| as ordered bits (NaN/Inf on top)
    const unsigned a = max(v.x & 0x7fffffffu, v.y & 0x7fffffffu), b = max(v.z & 0x7fffffffu, v.w & 0x7fffffffu);
    m = max(m, max(a, b));
  }
  for (int64_t t = n4 * 4 + (int64_t)block * blockDim.x + threadIdx.x; t < n; t += stride)
    m = max(m, __float_as_uint(gz[t]) & 0x7fffffffu);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(sm[0], sm[1]), max(sm[2], sm[3]));
    if (m) atomicMax(&zs->max_bits, m);
  }
}

__global__ __launch_bounds__(256) void k_z_absmax(const float* __restrict__ gz0, int64_t n, ZScale* __restrict__ zs,
                                                  int64_t view_stride = 0) {
  z_absmax_body(gz0 + (int64_t)blockIdx.y * view_stride, n, zs, blockIdx.x, gridDim.x);
}

// terms_log2 = ceil(log2(max number of slots that can list one point)) = ceil(log2(S*S)): every term is
// < 2^(61 - terms_log2) in magnitude after scaling, so no sum of them reaches 2^61 -- the range kept for
// the NaN / Inf poison (a point that received one stays at >= 2^61 whatever else is added).
__device__ __forceinline__ int z_exponent(unsigned max_bits, int terms_log2) {
  int e = 0;
  if (max_bits != 0u && max_bits < 0x7f800000u) {
    int ex;
    (void)frexpf(__uint_as_float(max_bits), &ex);      // max = f * 2^ex, f in [0.5,1)
    e = 61 - terms_log2 - ex;
  }
  return e;
}
__global__ void k_z_scale(ZScale* zs, int terms_log2) { zs->exp2 = z_exponent(zs->max_bits, terms_log2); }

__device__ __forceinline__ void z_scatter_one(int p, float g, int e, long long* __restrict__ acc) {
  if (g == 0.0f) return;
  if (g != g || fabsf(g) > 3.0e38f) {                                // poison: the point ends up NaN
    atomicMax(&acc[p], 1ll << 62);
    return;
  }
  const long long q = __double2ll_rn(ldexp((double)g, e));
  atomicAdd(reinterpret_cast<unsigned long long*>(&acc[p]), (unsigned long long)q);
}

// terms_log2 >= 0: the exponent is derived here from zs->max_bits (no k_z_scale launch in between); < 0: zs->exp2
__global__ void k_z_scatter(const int32_t* __restrict__ idx0, const float* __restrict__ gz0, int K, int64_t npix,
                            const ZScale* __restrict__ zs, long long* __restrict__ acc, int64_t view_stride = 0,
                            int terms_log2 = -1) {
  const int32_t* __restrict__ idx = idx0 + (int64_t)blockIdx.y * view_stride * K;
  const float* __restrict__ gz = gz0 + (int64_t)blockIdx.y * view_stride * K;
  const int e = terms_log2 >= 0 ? z_exponent(zs->max_bits, terms_log2) : zs->exp2;
  if ((K & 3) == 0 && (((uintptr_t)idx | (uintptr_t)gz) & 15) == 0) {   // a pixel's lists as 16-byte loads
    const int4* __restrict__ idx4 = reinterpret_cast<const int4*>(idx);
    const float4* __restrict__ gz4 = reinterpret_cast<const float4*>(gz);
    const int64_t nq = npix * (K / 4);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += (int64_t)gridDim.x * blockDim.x) {
      // the gradient first: four slots without one (a loss on the front-most depth: every second quad) need no index read,
      // and the two loads of a quad that has one are independent
      const float4 g = gz4[i];
      if (g.x == 0.0f && g.y == 0.0f && g.z == 0.0f && g.w == 0.0f) continue;
      const int4 p = idx4[i];
      if (p.x < 0) continue;                                         // -1 padding is a suffix of a pixel's list
      z_scatter_one(p.x, g.x, e, acc);
      if (p.y >= 0) z_scatter_one(p.y, g.y, e, acc);
      if (p.z >= 0) z_scatter_one(p.z, g.z, e, acc);
      if (p.w >= 0) z_scatter_one(p.w, g.w, e, acc);
    }
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
    for (int k = 0; k < K; ++k) {
      const int p = idx[i * K + k];
      if (p < 0) break;
      z_scatter_one(p, gz[i * K + k], e, acc);
    }
  }
}

__global__ void k_z_finish(const long long* __restrict__ acc, const ZScale* __restrict__ zs, int64_t n,
                           float* __restrict__ grad, int terms_log2 = -1) {
  const int e = terms_log2 >= 0 ? z_exponent(zs->max_bits, terms_log2) : zs->exp2;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
    const long long a = acc[p];
    float z = (float)ldexp((double)a, -e);
    if (a >= (1ll << 61) || a <= -(1ll << 61)) z = __builtin_nanf("");
    grad[p * 3 + 2] = z;
  }
}

// the same over the rows of the clouds only (grid.y = cloud): rows of the packed arrays that belong to no cloud keep
// what the caller put there, as in the xy pass
__device__ void z_finish_body(const long long* __restrict__ acc, const ZScale* __restrict__ zs, const int64_t* __restrict__ first,
                              const int64_t* __restrict__ num, int n_clouds, float* __restrict__ grad, int terms_log2,
                              int block, int nblocks) {
  const int e = z_exponent(zs->max_bits, terms_log2);
  for (int c = 0; c < n_clouds; ++c) {
    const int64_t len = num[c], base = first[c];
    for (int64_t i = (int64_t)block * blockDim.x + threadIdx.x; i < len; i += (int64_t)nblocks * blockDim.x) {
      const int64_t p = base + i;
      const long long a = acc[p];
      float z = (float)ldexp((double)a, -e);
      if (a >= (1ll << 61) || a <= -(1ll << 61)) z = __builtin_nanf("");
      grad[p * 3 + 2] = z;
    }
  }
}

__global__ void k_z_finish_clouds(const long long* __restrict__ acc, const ZScale* __restrict__ zs,
                                  const int64_t* __restrict__ first, const int64_t* __restrict__ num,
                                  float* __restrict__ grad, int terms_log2) {
  z_finish_body(acc, zs, first + blockIdx.y, num + blockIdx.y, 1, grad, terms_log2, blockIdx.x, gridDim.x);
}

static int64_t bwd_maps_bytes(int n_clouds, int image_size, int image_width) {
  const BlkGeo G = make_blk(image_size, image_width > 0 ? image_width : image_size);
  // pixel masks (8 B per 8x8 block), row bytes (one per eight blocks of a block row), super-block flags
  return (((int64_t)n_clouds * ((int64_t)G.NBx * G.NBy * 8 + (int64_t)G.NB2x * G.NBy + (int64_t)G.NB2x * G.NB2y)) + 63) / 64 * 64;
}
extern "C" int64_t iso_splat_backward_workspace_bytes(int n_clouds, int image_size, int image_width,
                                                      int64_t total_points) {
  if (total_points < 0) total_points = 0;
  // [block maps][heavy count (64 B) + heavy list (4 B/pt)][pad 16][ZScale 16 B][z accumulators 8 B/pt]
  return bwd_maps_bytes(n_clouds, image_size, image_width) + 64 + (4 * total_points + 15) / 16 * 16 + 16 + 8 * total_points;
}

extern "C" int iso_splat_backward(const float* points, const float* radii, const uint8_t* visible,
                                  const float* search_radius, const int64_t* first_idx,
                                  const int64_t* num_pts, int n_clouds, int64_t max_pts,
                                  const float* grad_occ, const int32_t* idx,
                                  const float* grad_zbuf, int image_size, int image_width, int points_per_pixel,
                                  int rect_mode, float radii_s, int64_t total_points,
                                  void* workspace, int64_t workspace_bytes, float* grad_points,
                                  void* stream) {
  ISO_REQUIRE(n_clouds >= 0 && max_pts >= 0 && image_size > 0, ISO_ERR_INVALID, "iso_splat_backward: bad sizes");
  if (n_clouds == 0 || max_pts == 0) return ISO_OK;
  ISO_REQUIRE(points && radii && (search_radius || rect_mode) && first_idx && num_pts && grad_occ &&
                  grad_points && workspace && (!grad_zbuf || idx),
              ISO_ERR_INVALID, "iso_splat_backward: null pointer");
  ISO_REQUIRE(workspace_bytes >= iso_splat_backward_workspace_bytes(n_clouds, image_size, image_width, total_points),
              ISO_ERR_WORKSPACE, "iso_splat_backward: workspace too small");
  const Frame F = make_frame(image_size, image_width > 0 ? image_width : image_size);
  const BlkGeo G = make_blk(F.H, F.W);
  ISO_REQUIRE(F.H <= 2048 && F.W <= 2048, ISO_ERR_UNSUPPORTED, "iso_splat_backward: image sides must be <= 2048");
  hipStream_t s = (hipStream_t)stream;
  unsigned long long* pixmask = (unsigned long long*)workspace;            // (caller buffers are 16-byte aligned)
  uint8_t* rowbytes = (uint8_t*)(pixmask + (int64_t)n_clouds * G.NBx * G.NBy);
  uint8_t* blk2 = rowbytes + (int64_t)n_clouds * G.NB2x * G.NBy;
  int32_t* heavy_count = (int32_t*)((uint8_t*)workspace + bwd_maps_bytes(n_clouds, image_size, image_width));
  int32_t* heavy = heavy_count + 16;
  const bool with_z = grad_zbuf && total_points > 0;
  ZScale* zs = reinterpret_cast<ZScale*>((char*)(heavy_count) + 64 + (4 * total_points + 15) / 16 * 16);
  long long* zacc = reinterpret_cast<long long*>((char*)zs + 16);
  // block maps of the image gradient (both levels, one launch); it also clears the heavy-list length and the z scale
  {
    int gmaps = iso_div_up((int64_t)n_clouds * G.NB2x * G.NB2y, 4);
    hipLaunchKernelGGL(k_grad_maps, dim3(gmaps < 1 ? 1 : gmaps), dim3(256), 0, s, grad_occ, F, G, n_clouds, pixmask, rowbytes, blk2,
                       heavy_count, 16, with_z ? reinterpret_cast<int32_t*>(zs) : nullptr, 4);
  }
  int gx = iso_div_up(max_pts, 1024); if (gx > 8192) gx = 8192;
  // The z part (pixel-major, fixed point) is three passes -- max |grad_zbuf|, scatter, conversion -- of which only the
  // scatter is a launch of its own: the maximum rides in the xy launch below (it reads nothing that launch writes; the
  // scale was cleared by k_grad_maps), the conversion in the heavy-point launch at the end (which writes x and y only).
  const int64_t npix = (int64_t)n_clouds * F.H * F.W;
  int terms_log2 = 0;
  while ((1ll << terms_log2) < (int64_t)F.H * F.W) ++terms_log2;
  ZRider r1{nullptr, 0, zs, 0, nullptr, nullptr, 0}, r2 = r1;
  if (with_z) {
    int gm = iso_div_up(npix * points_per_pixel, 256 * 16); if (gm > 1024) gm = 1024; if (gm < 1) gm = 1;
    r1 = ZRider{grad_zbuf, npix * points_per_pixel, zs, gm, nullptr, nullptr, 0};
    r2 = ZRider{nullptr, 0, zs, iso_stream_grid(total_points, 256), zacc, grad_points, terms_log2};
  }
  // xy part point-major (z written as 0, the z accumulators of the rows cleared)
  hipLaunchKernelGGL(k_splat_backward, dim3(gx + r1.blocks, n_clouds), dim3(256), 0, s, points, radii, visible,
                     search_radius, first_idx, num_pts, blk2, G, F, rect_mode, radii_s, heavy,
                     heavy_count, grad_points, with_z ? zacc : nullptr, r1, gx);
  if (with_z)      // the exponent is derived from the maximum inside the kernels that need it (no scale launch in between)
    hipLaunchKernelGGL(k_z_scatter, dim3(iso_stream_grid(npix, 256)), dim3(256), 0, s, idx, grad_zbuf,
                       points_per_pixel, npix, zs, zacc, (int64_t)0, terms_log2);
  // eight lanes per heavy point; the cost of a point varies with the gradient pixels under its disc, so the list is spread
  // over short-lived workgroups (a workgroup past the end of the list reads the count and leaves; ISO_HEAVY_GRID overrides)
  static const int heavy_grid = []() { const char* e = getenv("ISO_HEAVY_GRID"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 4096; }();
  hipLaunchKernelGGL(k_splat_backward_heavy, dim3(heavy_grid + r2.blocks), dim3(256), 0, s, points, radii, search_radius,
                     first_idx, num_pts, n_clouds, grad_occ, pixmask, rowbytes, G, F, rect_mode, radii_s,
                     heavy, heavy_count, grad_points, r2, heavy_grid);
  ISO_CHECK_LAUNCH("iso_splat_backward");
  return ISO_OK;
}

// ---- the z gradient in pieces (N ranks: every rank scatters the pixels of its own tile rows; the
// 64-bit accumulators are summed over the ranks before the finish; the scale comes from the global
// max |grad|).  zscale: 2 ints [max |grad| bits, exponent].
extern "C" int iso_splat_z_absmax(const float* grad_zbuf, int64_t n, int32_t* zscale, void* stream) {
  ISO_REQUIRE(zscale && n >= 0 && (grad_zbuf || n == 0), ISO_ERR_INVALID, "iso_splat_z_absmax: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  iso_zero_words(zscale, 2, s);
  if (n > 0) {
    int gm = iso_div_up(n, 256 * 16); if (gm > 1024) gm = 1024; if (gm < 1) gm = 1;
    hipLaunchKernelGGL(k_z_absmax, dim3(gm), dim3(256), 0, s, grad_zbuf, n, reinterpret_cast<ZScale*>(zscale));
  }
  ISO_CHECK_LAUNCH("iso_splat_z_absmax");
  return ISO_OK;
}

extern "C" int iso_splat_z_scatter(const int32_t* idx, const float* grad_zbuf, int64_t n_pixels, int points_per_pixel,
                                   int64_t pixels_per_view, int32_t* zscale, int64_t* acc, void* stream) {
  ISO_REQUIRE(zscale && acc && n_pixels >= 0 && points_per_pixel >= 1 && pixels_per_view > 0 && ((idx && grad_zbuf) || n_pixels == 0),
              ISO_ERR_INVALID, "iso_splat_z_scatter: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  int terms_log2 = 0;
  while ((1ll << terms_log2) < pixels_per_view) ++terms_log2;
  hipLaunchKernelGGL(k_z_scale, dim3(1), dim3(1), 0, s, reinterpret_cast<ZScale*>(zscale), terms_log2);
  if (n_pixels > 0)
    hipLaunchKernelGGL(k_z_scatter, dim3(iso_stream_grid(n_pixels, 256)), dim3(256), 0, s, idx, grad_zbuf, points_per_pixel,
                       n_pixels, reinterpret_cast<const ZScale*>(zscale), reinterpret_cast<long long*>(acc));
  ISO_CHECK_LAUNCH("iso_splat_z_scatter");
  return ISO_OK;
}

// The three band calls of a cycle for all views at once (grid.y = view): the same kernels on
// idx / grad_zbuf + v * view_pixels * K for v < n_views (the band's rows of every view of an (N,H,W,K) array).
extern "C" int iso_splat_band_marks(const int32_t* idx, const float* grad_zbuf, int n_views, int64_t view_pixels,
                                    int64_t band_pixels, int points_per_pixel, uint8_t* visible, int32_t* zscale,
                                    void* stream) {
  ISO_REQUIRE(zscale && n_views >= 0 && view_pixels >= band_pixels && band_pixels >= 0 && points_per_pixel >= 1,
              ISO_ERR_INVALID, "iso_splat_band_marks: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  iso_zero_words(zscale, 2, s);
  if (band_pixels == 0 || n_views == 0) return ISO_OK;
  ISO_REQUIRE(idx && grad_zbuf && visible, ISO_ERR_INVALID, "iso_splat_band_marks: null pointer");
  ISO_REQUIRE((((uintptr_t)grad_zbuf) & 15) == 0 && ((view_pixels * points_per_pixel) & 3) == 0 &&
                  ((band_pixels * points_per_pixel) & 3) == 0,
              ISO_ERR_UNSUPPORTED, "iso_splat_band_marks: grad_zbuf slices must be 16-byte aligned");
  hipLaunchKernelGGL(k_mark_visible, dim3(iso_stream_grid(band_pixels, 256), n_views), dim3(256), 0, s, idx,
                     points_per_pixel, band_pixels, visible, view_pixels);
  const int64_t n = band_pixels * points_per_pixel;
  int gm = iso_div_up(n, 256 * 16); if (gm > 1024) gm = 1024; if (gm < 1) gm = 1;
  hipLaunchKernelGGL(k_z_absmax, dim3(gm, n_views), dim3(256), 0, s, grad_zbuf, n, reinterpret_cast<ZScale*>(zscale),
                     view_pixels * points_per_pixel);
  ISO_CHECK_LAUNCH("iso_splat_band_marks");
  return ISO_OK;
}

extern "C" int iso_splat_band_z_scatter(const int32_t* idx, const float* grad_zbuf, int n_views, int64_t view_pixels,
                                        int64_t band_pixels, int points_per_pixel, int32_t* zscale, int64_t* acc,
                                        void* stream) {
  ISO_REQUIRE(zscale && acc && n_views >= 0 && view_pixels > 0 && band_pixels >= 0 && points_per_pixel >= 1 &&
                  ((idx && grad_zbuf) || band_pixels == 0),
              ISO_ERR_INVALID, "iso_splat_band_z_scatter: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  int terms_log2 = 0;
  while ((1ll << terms_log2) < view_pixels) ++terms_log2;
  hipLaunchKernelGGL(k_z_scale, dim3(1), dim3(1), 0, s, reinterpret_cast<ZScale*>(zscale), terms_log2);
  if (band_pixels > 0 && n_views > 0)
    hipLaunchKernelGGL(k_z_scatter, dim3(iso_stream_grid(band_pixels, 256), n_views), dim3(256), 0, s, idx, grad_zbuf,
                       points_per_pixel, band_pixels, reinterpret_cast<const ZScale*>(zscale),
                       reinterpret_cast<long long*>(acc), view_pixels);
  ISO_CHECK_LAUNCH("iso_splat_band_z_scatter");
  return ISO_OK;
}

namespace {
__global__ void k_z_finish_rows(const long long* __restrict__ acc, const ZScale* __restrict__ zs, int64_t row0, int64_t n,
                                float* __restrict__ grad) {
  const int e = zs->exp2;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
    const long long a = acc[row0 + j];
    float z = (float)ldexp((double)a, -e);
    if (a >= (1ll << 61) || a <= -(1ll << 61)) z = __builtin_nanf("");
    grad[(row0 + j) * 3 + 2] = z;
  }
}

// packed global rows (view-major, then rank, then the rank's own order) from the all-gathered per-rank
// blocks: block s = 12 arrays of `cap` rows (ndc 3, ellipse 3, radii 2, scaler 1, features 3) as written
// by iso_splat_front into one buffer; counts (world, 8) = the ranks' num_points per view.
__global__ __launch_bounds__(256) void k_repack_records(const float* __restrict__ gathered, int64_t cap, int world,
                                                        int n_views, const int32_t* __restrict__ counts, float cutoffC,
                                                        int64_t max_rows, float* __restrict__ ndc, float* __restrict__ ellipse,
                                                        float* __restrict__ cutoff, float* __restrict__ radii,
                                                        float* __restrict__ scaler, float* __restrict__ feat) {
  const int s = blockIdx.y, v = blockIdx.z;
  int64_t g0 = 0, l0 = 0;                       // first global row of (v, s); first local row of view v in block s
  for (int vv = 0; vv < n_views; ++vv)
    for (int ss = 0; ss < world; ++ss) {
      const int c = counts[ss * 8 + vv];
      if (vv < v || (vv == v && ss < s)) g0 += c;
      if (ss == s && vv < v) l0 += c;
    }
  int64_t n = counts[s * 8 + v];
  if (l0 + n > cap) n = cap > l0 ? cap - l0 : 0;      // the sender's buffer overflowed (flagged by the host side)
  if (g0 + n > max_rows) n = max_rows > g0 ? max_rows - g0 : 0;   // ... and so the packed arrays would
  const float* blk = gathered + (int64_t)s * 12 * cap;
  // every field of the (view, rank) block is one contiguous run in both layouts: flat, coalesced copies
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = t0; j < 3 * n; j += step) {
    ndc[g0 * 3 + j] = blk[l0 * 3 + j];
    ellipse[g0 * 3 + j] = blk[3 * cap + l0 * 3 + j];
    feat[g0 * 3 + j] = blk[9 * cap + l0 * 3 + j];
  }
  for (int64_t j = t0; j < 2 * n; j += step) radii[g0 * 2 + j] = blk[6 * cap + l0 * 2 + j];
  for (int64_t j = t0; j < n; j += step) {
    scaler[g0 + j] = blk[8 * cap + l0 + j];
    cutoff[g0 + j] = cutoffC;
  }
}

// first_idx / num_points of the global packed layout and of this rank's own rows in it
__global__ void k_repack_offsets(const int32_t* __restrict__ counts, int world, int n_views, int rank, int64_t max_rows,
                                 int64_t* __restrict__ first_g, int64_t* __restrict__ num_g,
                                 int64_t* __restrict__ first_own, int64_t* __restrict__ num_own) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int64_t run = 0;
  for (int v = 0; v < n_views; ++v) {
    first_g[v] = run < max_rows ? run : max_rows;
    for (int s = 0; s < world; ++s) {
      const int64_t c = counts[s * 8 + v];
      if (s == rank) { first_own[v] = run < max_rows ? run : max_rows; num_own[v] = run + c <= max_rows ? c : (max_rows > run ? max_rows - run : 0); }
      run += c;
    }
    num_g[v] = (run < max_rows ? run : max_rows) - first_g[v];      // rows beyond max_rows do not exist (overflow: flagged)
  }
}
}  // namespace

extern "C" int iso_splat_z_finish(const int64_t* acc, const int32_t* zscale, int64_t row0, int64_t n_rows,
                                  float* grad_points, void* stream) {
  ISO_REQUIRE(acc && zscale && grad_points && row0 >= 0 && n_rows >= 0, ISO_ERR_INVALID, "iso_splat_z_finish: bad arguments");
  if (n_rows > 0)
    hipLaunchKernelGGL(k_z_finish_rows, dim3(iso_stream_grid(n_rows, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const long long*>(acc), reinterpret_cast<const ZScale*>(zscale), row0, n_rows,
                       grad_points);
  ISO_CHECK_LAUNCH("iso_splat_z_finish");
  return ISO_OK;
}

extern "C" int iso_splat_repack(const float* gathered, int64_t capacity, int world, int rank, int n_views,
                                const int32_t* counts, float cutoff, int64_t max_rows, float* ndc_out,
                                float* ellipse_out, float* cutoff_out, float* radii_out, float* scaler_out,
                                float* features_out, int64_t* first_idx_out, int64_t* num_pts_out,
                                int64_t* own_first_out, int64_t* own_num_out, void* stream) {
  ISO_REQUIRE(gathered && counts && world >= 1 && rank >= 0 && rank < world && n_views >= 1 && n_views <= 8 &&
                  capacity >= 0 && ndc_out && ellipse_out && cutoff_out && radii_out && scaler_out && features_out &&
                  first_idx_out && num_pts_out && own_first_out && own_num_out,
              ISO_ERR_INVALID, "iso_splat_repack: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_repack_offsets, dim3(1), dim3(64), 0, s, counts, world, n_views, rank, max_rows, first_idx_out,
                     num_pts_out, own_first_out, own_num_out);
  int gx = iso_div_up(max_rows > 0 ? 3 * max_rows / (world * n_views) + 1 : 1, 256); if (gx > 256) gx = 256;
  hipLaunchKernelGGL(k_repack_records, dim3(gx, world, n_views), dim3(256), 0, s, gathered, capacity, world, n_views, counts,
                     cutoff, max_rows, ndc_out, ellipse_out, cutoff_out, radii_out, scaler_out, features_out);
  ISO_CHECK_LAUNCH("iso_splat_repack");
  return ISO_OK;
}
