// MI355X-native frontend: point_cloud_segmentation::segmentallPointCloudData
// (reference include/planar_segmentation/point_cloud_segmentation.h:105-181,
//  src/planar_segmentation/plane_segmentation.cpp:24-259) as hand-written HIP for gfx950.
//
// The two PCL calls of the reference (IntegralImageNormalEstimation::compute and
// OrganizedMultiPlaneSegmentation::segmentAndRefine, plane_segmentation.cpp:97-104,136-156) are
// restated kernel by kernel.  Parity rule: labels / inlier sets must be bit-identical with the CPU
// oracle, so every float/double expression keeps PCL's operation order (this file is compiled with
// -ffp-contract=off) and the inherently sequential raster recurrences of PCL (integral-image
// recurrence, two-pass chamfer distance map, two-sweep label refinement) are executed as *skewed
// wavefronts*: lane l of a wave owns image row r0+l and runs two columns behind lane l-1, which
// reproduces the sequential order exactly while 64 rows progress in parallel.  Everything that is
// order-free (crop, depth-change map, per-pixel covariance + eigen solve, comparator evaluation,
// union-find connected components) is one thread per pixel.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "../../include/sslam.h"
#include "sslam_common.hpp"

namespace sslam {
namespace seg {

constexpr int kMaxRegions = 64;
constexpr int kBandFloats = 30720;  // LDS floats available to a band-staging kernel (120 KiB)

struct BoxMeta {
  int w, h, tlx, tly;
  int pix0;   // offset into the packed per-pixel arrays
  int ii0;    // offset into the packed integral-image arrays ((w+1)*(h+1) per box)
  int box_index, frame;   // frame: which cloud of a batched call the box is cut from
};

struct Region {
  float centroid[3];
  float model[4];
  int inliers, first_inlier, label, contour_n;
  float area;
  int contour_off;   // start of this region's boundary indices in the box's contour arena (-1: not stored)
  int pad0;
  unsigned long long last_key;  // (pass << 60) | (order key << 24) | pixel index
};

struct View {
  int nbox, npix_total, maxpix;
  const BoxMeta* box;
  const unsigned char* cloud;   // the clouds of all frames of the call, cloud_stride bytes apart
  size_t cloud_stride;
  int* overflow;                // [2]: boxes whose candidate / region tables were full (results truncated, reported to the caller)
  int point_step, row_step, ox, oy, oz;
  float* pts;    // [npix*3]
  float* dm;     // [npix]
  float* nrm;    // [npix*4]
  float* pd;     // [npix]
  int* lab;      // [npix] connected-component root (pixel index inside the box) or -1
  int* cnt;      // [npix] pixels per root
  int* l2m;      // [npix] root label -> region index, or -1
  int* code;     // [npix] region index | kOther | -1  (labels as seen by refinement / contour)
  double* ii;    // [nii*10] integral images, one 80-byte record per entry: first order (3), second order (6), finite-point count
  Region* reg;   // [nbox*kMaxRegions]
  int* contour;  // [4*npix] boundary pixel indices, one arena of 4*w*h ints per box at 4*pix0
  int* ccount;   // [nbox] ints used in each arena
  int* nreg;     // [nbox]
  float mdcf, smoothing, ang_thr_cos, dist_thr, max_curv;
  unsigned min_inliers;
  int refine_bh;   // rows per LDS band of k_refine (64: one box per CU, lowest latency; 24: three boxes per CU for calls with many boxes)
};

// ---------------------------------------------------------------------------------------------
// pcl::eigen33 (smallest eigenpair), float, same operation order as oracle/oracle_seg.c
__device__ __forceinline__ void roots2(float b, float c, float r[3]) {
  r[0] = 0.0f;
  float d = b * b - 4.0f * c;
  if (d < 0.0f) d = 0.0f;
  const float sd = sqrtf(d);
  r[2] = 0.5f * (b + sd);
  r[1] = 0.5f * (b - sd);
}
__device__ __forceinline__ void compute_roots(const float m[9], float r[3]) {
  const float m00 = m[0], m01 = m[1], m02 = m[2], m11 = m[4], m12 = m[5], m22 = m[8];
  const float c0 = m00 * m11 * m22 + 2.0f * m01 * m02 * m12 - m00 * m12 * m12 - m11 * m02 * m02 - m22 * m01 * m01;
  const float c1 = m00 * m11 - m01 * m01 + m00 * m22 - m02 * m02 + m11 * m22 - m12 * m12;
  const float c2 = m00 + m11 + m22;
  if (fabsf(c0) < 1.1920929e-07f) { roots2(c2, c1, r); return; }
  const float s_inv3 = 1.0f / 3.0f;
  const float s_sqrt3 = sqrtf(3.0f);
  const float c2_over_3 = c2 * s_inv3;
  float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
  if (a_over_3 > 0.0f) a_over_3 = 0.0f;
  const float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));
  float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
  if (q > 0.0f) q = 0.0f;
  const float rho = sqrtf(-a_over_3);
  // trigonometry in double, rounded to float: identical bits on glibc and ocml
  const float theta = (float)atan2((double)sqrtf(-q), (double)half_b) * s_inv3;
  const float cos_theta = (float)cos((double)theta);
  const float sin_theta = (float)sin((double)theta);
  r[0] = c2_over_3 + 2.0f * rho * cos_theta;
  r[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
  r[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
  float t;
  if (r[0] >= r[1]) { t = r[0]; r[0] = r[1]; r[1] = t; }
  if (r[1] >= r[2]) {
    t = r[1]; r[1] = r[2]; r[2] = t;
    if (r[0] >= r[1]) { t = r[0]; r[0] = r[1]; r[1] = t; }
  }
  if (r[0] <= 0.0f) roots2(c2, c1, r);
}
__device__ __forceinline__ void cross3(const float* a, const float* b, float* o) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ void eigen33(const float mat[9], float& eigenvalue, float vec[3]) {
  float scale = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) { const float a = fabsf(mat[k]); if (a > scale) scale = a; }
  if (scale <= 1.17549435e-38f) scale = 1.0f;
  float s[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) s[k] = mat[k] / scale;
  float r[3];
  compute_roots(s, r);
  eigenvalue = r[0] * scale;
  s[0] -= r[0]; s[4] -= r[0]; s[8] -= r[0];
  float v1[3], v2[3], v3[3];
  cross3(s + 0, s + 3, v1); cross3(s + 0, s + 6, v2); cross3(s + 3, s + 6, v3);
  const float l1 = v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2];
  const float l2 = v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2];
  const float l3 = v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2];
  float v[3], l;
  if (l1 >= l2 && l1 >= l3) { v[0] = v1[0]; v[1] = v1[1]; v[2] = v1[2]; l = l1; }
  else if (l2 >= l1 && l2 >= l3) { v[0] = v2[0]; v[1] = v2[1]; v[2] = v2[2]; l = l2; }
  else { v[0] = v3[0]; v[1] = v3[1]; v[2] = v3[2]; l = l3; }
  const float n = sqrtf(l);
  vec[0] = v[0] / n; vec[1] = v[1] / n; vec[2] = v[2] / n;
}

// ---------------------------------------------------------------------------------------------
// crop (plane_segmentation.cpp:24-82): box pixels of the organised cloud -> packed xyz
__global__ __launch_bounds__(256) void k_crop(View V) {
  const BoxMeta b = V.box[blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= b.w * b.h) return;
  const int v = i / b.w, u = i - v * b.w;
  const size_t pos = (size_t)b.frame * V.cloud_stride + (size_t)(b.tly + v) * V.row_step + (size_t)(b.tlx + u) * V.point_step;
  float x, y, z;
  memcpy(&x, V.cloud + pos + V.ox, 4); memcpy(&y, V.cloud + pos + V.oy, 4); memcpy(&z, V.cloud + pos + V.oz, 4);
  float* o = V.pts + ((size_t)b.pix0 + i) * 3;
  o[0] = x; o[1] = y; o[2] = z;
}

// depth-change map in gather form + distance-map initialisation
__device__ __forceinline__ bool dc_right(const float* pts, int w, int r, int c, float f) {
  const float d = pts[((size_t)r * w + c) * 3 + 2], dR = pts[((size_t)r * w + c + 1) * 3 + 2];
  const float thr = f * (fabsf(d) + 1.0f) * 2.0f;
  return fabsf(d - dR) > thr || !isfinite(d) || !isfinite(dR);
}
__device__ __forceinline__ bool dc_down(const float* pts, int w, int r, int c, float f) {
  const float d = pts[((size_t)r * w + c) * 3 + 2], dD = pts[((size_t)(r + 1) * w + c) * 3 + 2];
  const float thr = f * (fabsf(d) + 1.0f) * 2.0f;
  return fabsf(d - dD) > thr || !isfinite(d) || !isfinite(dD);
}
__global__ __launch_bounds__(256) void k_depth_change(View V) {
  const BoxMeta b = V.box[blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int w = b.w, h = b.h;
  if (i >= w * h) return;
  const int r = i / w, c = i - r * w;
  const float* pts = V.pts + (size_t)b.pix0 * 3;
  bool zero = false;
  if (r < h - 1 && c < w - 1) zero = dc_right(pts, w, r, c, V.mdcf) || dc_down(pts, w, r, c, V.mdcf);
  if (!zero && c >= 1 && r < h - 1) zero = dc_right(pts, w, r, c - 1, V.mdcf);
  if (!zero && r >= 1 && c < w - 1) zero = dc_down(pts, w, r - 1, c, V.mdcf);
  V.dm[(size_t)b.pix0 + i] = zero ? 0.0f : (float)(w + h);
  // normals default to NaN; labels to "invalid"
  float* nn = V.nrm + ((size_t)b.pix0 + i) * 4;
  const float qnan = __int_as_float(0x7fc00000);
  nn[0] = qnan; nn[1] = qnan; nn[2] = qnan; nn[3] = qnan;
}

// two-pass chamfer distance map: skewed wavefront over rows, band staged in LDS with the image's
// own row stride (so PCL's wrap-around reads prev[w] == cur[0], next[-1] == cur[w-1] come for free)
__global__ __launch_bounds__(256) void k_distance_map(View V) {
  extern __shared__ float band[];
  const BoxMeta b = V.box[blockIdx.x];
  const int w = b.w, h = b.h;
  float* dm = V.dm + b.pix0;
  const int BH = min(64, kBandFloats / w - 1);
  // ---- forward pass: rows 1 .. h-1
  for (int r0 = 1; r0 < h; r0 += BH) {
    const int nr = min(BH, h - r0);
    for (int k = threadIdx.x; k < (nr + 1) * w; k += 256) band[k] = dm[(size_t)(r0 - 1) * w + k];
    __syncthreads();
    if (threadIdx.x < 64) {
      const int l = threadIdx.x;
      const int steps = (w - 1) + 2 * (nr - 1);
      for (int t = 0; t < steps; ++t) {
        const int c = t - 2 * l + 1;
        if (l < nr && c >= 1 && c < w) {
          const float* prev = band + l * w;
          float* cur = band + (l + 1) * w;
          const float upLeft = prev[c - 1] + 1.4f, up = prev[c] + 1.0f, upRight = prev[c + 1] + 1.4f, left = cur[c - 1] + 1.0f;
          const float center = cur[c];
          const float m = fminf(fminf(upLeft, up), fminf(left, upRight));
          if (m < center) cur[c] = m;
        }
      }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nr * w; k += 256) dm[(size_t)r0 * w + k] = band[w + k];
    __syncthreads();
  }
  // ---- backward pass: rows h-2 .. 0
  for (int rhi = h - 2; rhi >= 0; rhi -= BH) {
    const int nr = min(BH, rhi + 1);
    const int rlo = rhi - nr + 1;
    for (int k = threadIdx.x; k < (nr + 1) * w; k += 256) band[k] = dm[(size_t)rlo * w + k];
    __syncthreads();
    if (threadIdx.x < 64) {
      const int l = threadIdx.x;  // lane l owns row rhi - l = band row (nr - 1 - l)
      const int steps = (w - 1) + 2 * (nr - 1);
      for (int t = 0; t < steps; ++t) {
        const int c = (w - 2) - (t - 2 * l);
        if (l < nr && c >= 0 && c <= w - 2) {
          float* cur = band + (nr - 1 - l) * w;
          const float* next = cur + w;
          // (row 0, column 0 would read next[-1] = cur[w-1]; for band row 0 that is band[w-1]: in range)
          const float lowerLeft = next[c - 1] + 1.4f, lower = next[c] + 1.0f, lowerRight = next[c + 1] + 1.4f, right = cur[c + 1] + 1.0f;
          const float center = cur[c];
          const float m = fminf(fminf(lowerLeft, lower), fminf(right, lowerRight));
          if (m < center) cur[c] = m;
        }
      }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nr * w; k += 256) dm[(size_t)rlo * w + k] = band[k];
    __syncthreads();
  }
}

// integral images (IntegralImage2D<float,3>::computeIntegralImages, second order on).  Lane l of a wave owns row r0+l, two columns
// behind lane l-1; the previous row's running values arrive by wave shuffle.  Recurrence and operation order are PCL's:
// cur[c+1] = prev[c+1] + cur[c] - prev[c] (+ element).  The ten running sums (x y z | xx xy xz | yy yz | zz count) are independent of
// one another, so the four waves of the workgroup each carry two or three of them over the same rows: a step is bound by one wave's
// instruction issue, and no barrier is needed between the waves.  One 80-byte record per entry: [x y z xx xy xz yy yz zz count].
struct alignas(16) IiPair { double a, b; };
template <int Q>
__device__ __forceinline__ float ii_elem(float ex, float ey, float ez) {
  return Q == 0 ? ex : Q == 1 ? ey : Q == 2 ? ez : Q == 3 ? ex * ex : Q == 4 ? ex * ey : Q == 5 ? ex * ez : Q == 6 ? ey * ey : Q == 7 ? ey * ez : Q == 8 ? ez * ez : 1.0f;
}
template <int C0, int NC>
__device__ __forceinline__ void ii_store(double* e, const double (&v)[NC]) {   // e = record + C0; 16-byte stores where the pair is aligned
  if (NC == 2) { *reinterpret_cast<IiPair*>(e) = IiPair{v[0], v[1]}; }
  else if (C0 % 2 == 0) { *reinterpret_cast<IiPair*>(e) = IiPair{v[0], v[1]}; e[2] = v[NC - 1]; }
  else { e[0] = v[0]; *reinterpret_cast<IiPair*>(e + 1) = IiPair{v[1], v[NC - 1]}; }
}
template <int C0, int NC>
__device__ __forceinline__ void integral_wave(const float* bpts, const double* prevrow, double* rec, int w, int nr, int r0, int l) {
  const int W1 = w + 1, r = r0 + l;
  double cur[NC], h1[NC], h2[NC], p0[NC];   // cur[c] of the running column; own outputs one / two steps ago; prev[c]
#pragma unroll
  for (int q = 0; q < NC; ++q) { cur[q] = 0; h1[q] = 0; h2[q] = 0; p0[q] = 0; }
  const int steps = w + 2 * (nr - 1);
  for (int t = 0; t < steps; ++t) {
    const int c = t - 2 * l;
    // prev[c+1]: lane l-1's output two steps ago; lane 0 reads the staged last row of the previous band
    double p1[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) p1[q] = __shfl_up(h2[q], 1, 64);
    const bool active = l < nr && c >= 0 && c < w;
    if (l == 0 && active) {
#pragma unroll
      for (int q = 0; q < NC; ++q) p1[q] = prevrow[(c + 1) * 10 + C0 + q];
    }
    double out[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) out[q] = h1[q];
    if (active) {
#pragma unroll
      for (int q = 0; q < NC; ++q) out[q] = p1[q] + cur[q] - p0[q];
      const float* ep = bpts + ((size_t)l * w + c) * 3;
      const float ex = ep[0], ey = ep[1], ez = ep[2];
      if (isfinite(ex + ey + ez)) {
        out[0] += (double)ii_elem<C0>(ex, ey, ez);
        out[1] += (double)ii_elem<C0 + 1>(ex, ey, ez);
        if (NC == 3) out[NC - 1] += (double)ii_elem<C0 + NC - 1>(ex, ey, ez);
      }
      const size_t o = (size_t)(r + 1) * W1 + (c + 1);
      ii_store<C0, NC>(rec + o * 10 + C0, out);
      if (c == 0) {  // integral column 0 of this row is zero
        const double zero[NC] = {};
        ii_store<C0, NC>(rec + (size_t)(r + 1) * W1 * 10 + C0, zero);
      }
#pragma unroll
      for (int q = 0; q < NC; ++q) { cur[q] = out[q]; p0[q] = p1[q]; }
    }
    // shift the output history (every lane, every step: the shuffle above reads h2)
#pragma unroll
    for (int q = 0; q < NC; ++q) { h2[q] = h1[q]; h1[q] = out[q]; }
  }
}
__global__ __launch_bounds__(256) void k_integral(View V, int band_rows) {
  extern __shared__ double prevrow[];  // (w+1) x 10 doubles: last row of the previous band; then the band's points (floats)
  const BoxMeta b = V.box[blockIdx.x];
  const int w = b.w, h = b.h, W1 = w + 1;
  const float* pts = V.pts + (size_t)b.pix0 * 3;
  float* bpts = reinterpret_cast<float*>(prevrow + (size_t)W1 * 10);
  double* rec = V.ii + (size_t)b.ii0 * 10;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // row 0 of the integral images is zero
  for (int k = tid; k < W1 * 10; k += 256) { rec[k] = 0; prevrow[k] = 0; }
  __syncthreads();
  for (int r0 = 0; r0 < h; r0 += band_rows) {
    const int nr = min(band_rows, h - r0);
    for (int k = tid; k < nr * w * 3; k += 256) bpts[k] = pts[(size_t)r0 * w * 3 + k];   // coalesced
    __syncthreads();
    if (wave == 0) integral_wave<0, 3>(bpts, prevrow, rec, w, nr, r0, lane);
    else if (wave == 1) integral_wave<3, 3>(bpts, prevrow, rec, w, nr, r0, lane);
    else if (wave == 2) integral_wave<6, 2>(bpts, prevrow, rec, w, nr, r0, lane);
    else integral_wave<8, 2>(bpts, prevrow, rec, w, nr, r0, lane);
    // stage the last row of this band for the next band's lane 0
    __syncthreads();
    if (r0 + band_rows < h) {
      const double* src = rec + (size_t)(r0 + band_rows) * W1 * 10;     // same record layout as the carry row
      for (int k = tid; k < W1 * 10; k += 256) prevrow[k] = src[k];
    }
    __syncthreads();
  }
}

// per-pixel normal: IntegralImageNormalEstimation::computePointNormal (COVARIANCE_MATRIX),
// BORDER_POLICY_IGNORE, fixed smoothing size
__global__ __launch_bounds__(256) void k_normals(View V) {
  const BoxMeta b = V.box[blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int w = b.w, h = b.h, W1 = w + 1;
  if (i >= w * h) return;
  const int r = i / w, c = i - r * w;
  const int border = (int)V.smoothing;
  if (r < border || r >= h - border || c < border || c >= w - border) return;
  const float* pts = V.pts + (size_t)b.pix0 * 3;
  const float depth = pts[(size_t)i * 3 + 2];
  if (!isfinite(depth)) return;
  const float smoothing = fminf(V.dm[(size_t)b.pix0 + i], V.smoothing);
  if (!(smoothing > 2.0f)) return;
  const int rw = (int)smoothing, rh = (int)smoothing;
  const int sx = c - rw / 2, sy = r - rh / 2;
  const size_t ul = (size_t)sy * W1 + sx, ur = ul + rw, ll = (size_t)(sy + rh) * W1 + sx, lr = ll + rw;
  const double* ii = V.ii + (size_t)b.ii0 * 10;
  const unsigned count = (unsigned)ii[lr * 10 + 9] + (unsigned)ii[ul * 10 + 9] - (unsigned)ii[ur * 10 + 9] - (unsigned)ii[ll * 10 + 9];
  if (count == 0) return;
  float cen[3], sov[6], cov[9];
#pragma unroll
  for (int k = 0; k < 3; ++k) cen[k] = (float)(ii[lr * 10 + k] + ii[ul * 10 + k] - ii[ur * 10 + k] - ii[ll * 10 + k]);
#pragma unroll
  for (int k = 0; k < 6; ++k) sov[k] = (float)(ii[lr * 10 + 3 + k] + ii[ul * 10 + 3 + k] - ii[ur * 10 + 3 + k] - ii[ll * 10 + 3 + k]);
  cov[0] = sov[0]; cov[1] = cov[3] = sov[1]; cov[2] = cov[6] = sov[2]; cov[4] = sov[3]; cov[5] = cov[7] = sov[4]; cov[8] = sov[5];
  const float fc = (float)count;
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int bb = 0; bb < 3; ++bb) cov[a * 3 + bb] -= (cen[a] * cen[bb]) / fc;
  float ev, v[3];
  eigen33(cov, ev, v);
  const float vx = 0.0f - pts[(size_t)i * 3 + 0], vy = 0.0f - pts[(size_t)i * 3 + 1], vz = 0.0f - pts[(size_t)i * 3 + 2];
  const float ct = vx * v[0] + vy * v[1] + vz * v[2];
  if (ct < 0) { v[0] *= -1; v[1] *= -1; v[2] *= -1; }
  float* nn = V.nrm + ((size_t)b.pix0 + i) * 4;
  nn[0] = v[0]; nn[1] = v[1]; nn[2] = v[2];
  nn[3] = ev > 0.0f ? fabsf(ev / (cov[0] + cov[4] + cov[8])) : 0.0f;
}

// ---------------------------------------------------------------------------------------------
// OrganizedConnectedComponentSegmentation with the PlaneCoefficientComparator (depth dependent)
__global__ __launch_bounds__(256) void k_cc_init(View V) {
  const BoxMeta b = V.box[blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= b.w * b.h) return;
  const size_t g = (size_t)b.pix0 + i;
  const float* p = V.pts + g * 3;
  const float* n = V.nrm + g * 4;
  V.pd[g] = p[0] * n[0] + p[1] * n[1] + p[2] * n[2];
  V.lab[g] = isfinite(p[0]) ? i : -1;
  V.cnt[g] = 0;
  V.l2m[g] = -1;
}
__device__ __forceinline__ bool coeff_compare(const View& V, size_t g1, size_t g2) {
  const float z = V.pts[g1 * 3 + 2];
  const float thr = V.dist_thr * (z * z);
  const float* a = V.nrm + g1 * 4;
  const float* b = V.nrm + g2 * 4;
  const float dot = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
  return (fabsf(V.pd[g1] - V.pd[g2]) < thr) && (dot > V.ang_thr_cos);
}
__device__ __forceinline__ int uf_find(int* L, int i) {
  int p = L[i];
  while (p != i) { i = p; p = L[i]; }
  return i;
}
__device__ __forceinline__ void uf_union(int* L, int a, int b) {
  while (true) {
    a = uf_find(L, a); b = uf_find(L, b);
    if (a == b) return;
    if (a < b) { const int t = a; a = b; b = t; }   // attach the larger root to the smaller one
    const int old = atomicMin(&L[a], b);
    if (old == a) return;
    a = old;
  }
}
__global__ __launch_bounds__(256) void k_cc_merge(View V) {
  const BoxMeta b = V.box[blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int w = b.w;
  if (i >= w * b.h) return;
  int* L = V.lab + b.pix0;
  if (L[i] < 0) return;
  const int r = i / w, c = i - r * w;
  const size_t g = (size_t)b.pix0 + i;
  if (c > 0 && L[i - 1] >= 0 && coeff_compare(V, g, g - 1)) uf_union(L, i, i - 1);
  if (r > 0 && L[i - w] >= 0 && coeff_compare(V, g, g - w)) uf_union(L, i, i - w);
}
__global__ __launch_bounds__(256) void k_cc_flatten(View V) {
  const BoxMeta b = V.box[blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= b.w * b.h) return;
  int* L = V.lab + b.pix0;
  if (L[i] < 0) return;
  const int root = uf_find(L, i);
  atomicAdd(&V.cnt[b.pix0 + root], 1);
  // roots only ever point at themselves, so writing the flattened parent is race-free
  if (root != i) L[i] = root;
}
__global__ __launch_bounds__(256) void k_cc_flatten2(View V) {  // second hop: every pixel points at its root
  const BoxMeta b = V.box[blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= b.w * b.h) return;
  int* L = V.lab + b.pix0;
  if (L[i] >= 0) L[i] = uf_find(L, i);
}

// The four kernels above as ONE workgroup per box with the union-find forest in LDS (boxes of <= kCcLdsMax pixels; the YOLO boxes of
// the reference are a few thousand pixels).  Roots are the smallest pixel index of a component whatever the order of the unions, so
// labels and counts are those of the global-memory kernels; the comparator's plane distance p.n is recomputed for the neighbour
// (same expression, no contraction) instead of being parked in HBM.  LDS: 4 bytes per pixel -> three 12k-pixel boxes per CU.
constexpr int kCcLdsMax = 16384;
__device__ __forceinline__ bool coeff_compare_direct(const View& V, size_t g1, size_t g2) {
  const float* p1 = V.pts + g1 * 3;
  const float* p2 = V.pts + g2 * 3;
  const float* a = V.nrm + g1 * 4;
  const float* b = V.nrm + g2 * 4;
  const float pd1 = p1[0] * a[0] + p1[1] * a[1] + p1[2] * a[2];
  const float pd2 = p2[0] * b[0] + p2[1] * b[1] + p2[2] * b[2];
  const float z = p1[2];
  const float thr = V.dist_thr * (z * z);
  const float dot = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
  return (fabsf(pd1 - pd2) < thr) && (dot > V.ang_thr_cos);
}
__global__ __launch_bounds__(1024) void k_cc_lds(View V) {
  extern __shared__ int Ls[];
  constexpr int NT = 1024, PER = kCcLdsMax / NT;
  const BoxMeta b = V.box[blockIdx.x];
  const int w = b.w, n = w * b.h, tid = threadIdx.x;
  for (int i = tid; i < n; i += NT) {
    const size_t g = (size_t)b.pix0 + i;
    Ls[i] = isfinite(V.pts[g * 3]) ? i : -1;
    V.l2m[g] = -1;
  }
  __syncthreads();
  for (int i = tid; i < n; i += NT) {
    if (Ls[i] < 0) continue;
    const int r = i / w, c = i - r * w;
    const size_t g = (size_t)b.pix0 + i;
    if (c > 0 && Ls[i - 1] >= 0 && coeff_compare_direct(V, g, g - 1)) uf_union(Ls, i, i - 1);
    if (r > 0 && Ls[i - w] >= 0 && coeff_compare_direct(V, g, g - w)) uf_union(Ls, i, i - w);
  }
  __syncthreads();
  int root[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = tid + k * NT;
    root[k] = (i < n && Ls[i] >= 0) ? uf_find(Ls, i) : -1;
  }
  __syncthreads();   // every find is done: the forest may go
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = tid + k * NT;
    if (i < n) { V.lab[(size_t)b.pix0 + i] = root[k]; Ls[i] = 0; }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < PER; ++k) if (root[k] >= 0) atomicAdd(&Ls[root[k]], 1);
  __syncthreads();
  for (int i = tid; i < n; i += NT) V.cnt[(size_t)b.pix0 + i] = Ls[i];
}

// per-label plane fit (OrganizedMultiPlaneSegmentation::segment): labels with more than
// min_inliers pixels; PCL accumulates the 9 float sums in index (raster) order, so the additions stay
// a serial chain, but everything around them is parallel: candidate roots are found by all threads,
// each candidate gets a wave, pixels are loaded 64 at a time (coalesced), the 9 products are formed
// lane-parallel and parked in LDS, and lanes 0..8 each run one component's chain over the chunk.
// NT = 1024 (16 candidates at a time) for a call with few boxes; NT = 256 when a call brings more boxes than the chip has CUs: a box
// rarely has more than four candidate labels, and four times as many boxes are resident.
constexpr int kMaxCand = 64;
template <int NT>
__global__ __launch_bounds__(NT) void k_regions(View V) {
  constexpr int NW = NT / 64;
  __shared__ int cand[kMaxCand];
  __shared__ int ncand;
  __shared__ int accepted[kMaxCand];
  __shared__ float prod[NW][64][9];
  __shared__ Region regs[kMaxCand];
  const BoxMeta b = V.box[blockIdx.x];
  const int n = b.w * b.h;
  const int* L = V.lab + b.pix0;
  const int* cnt = V.cnt + b.pix0;
  const float* pts = V.pts + (size_t)b.pix0 * 3;
  if (threadIdx.x == 0) ncand = 0;
  if (threadIdx.x < kMaxCand) accepted[threadIdx.x] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += NT)
    if (L[i] == i && (unsigned)cnt[i] > V.min_inliers) { const int k = atomicAdd(&ncand, 1); if (k < kMaxCand) cand[k] = i; }
  __syncthreads();
  const int nc = min(ncand, kMaxCand);
  if (threadIdx.x == 0) {  // order by root pixel index = PCL's label order (insertion sort, short list)
    for (int a = 1; a < nc; ++a) { const int v = cand[a]; int q = a - 1; while (q >= 0 && cand[q] > v) { cand[q + 1] = cand[q]; --q; } cand[q + 1] = v; }
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int k = wave; k < nc; k += NW) {
    const int label = cand[k];
    float acc = 0;            // lanes 0..8: component `lane`
    int first = -1, last = -1;
    // the labels and points of the next FOUR chunks travel while a chunk's serial additions run: with one chunk ahead (round 3) the loop was one
    // L2 round trip per 64 pixels long -- 90 % of the kernel's wave cycles parked (profiles/r6_pmc_frontend.json)
    constexpr int PF = 4;
    int lq[PF];
    float xq[PF], yq[PF], zq[PF];
#pragma unroll
    for (int c = 0; c < PF; ++c) {
      const int j = 64 * c + lane;
      lq[c] = j < n ? L[j] : -2; xq[c] = yq[c] = zq[c] = 0;
      if (j < n) { xq[c] = pts[(size_t)j * 3]; yq[c] = pts[(size_t)j * 3 + 1]; zq[c] = pts[(size_t)j * 3 + 2]; }
    }
    for (int i00 = 0; i00 < n; i00 += 64 * PF) {
#pragma unroll
      for (int c = 0; c < PF; ++c) {
        const int i0 = i00 + 64 * c;
        const int i = i0 + lane;
        const bool m = i < n && lq[c] == label;
        const float x = xq[c], y = yq[c], z = zq[c];
        {
          const int j = i + 64 * PF;
          lq[c] = j < n ? L[j] : -2;
          if (j < n) { xq[c] = pts[(size_t)j * 3]; yq[c] = pts[(size_t)j * 3 + 1]; zq[c] = pts[(size_t)j * 3 + 2]; }
        }
        if (m) {
          float* p = prod[wave][lane];
          p[0] = x * x; p[1] = x * y; p[2] = x * z; p[3] = y * y; p[4] = y * z; p[5] = z * z; p[6] = x; p[7] = y; p[8] = z;
        }
        const unsigned long long mask = __ballot(m);
        if (mask) {
          if (first < 0) first = i0 + __ffsll((long long)mask) - 1;
          last = i0 + 63 - __clzll((long long)mask);
          // the additions are a serial chain in pixel order (PCL's), the LDS reads are not: eight products are fetched together,
          // then added under the (wave-uniform) inlier bits
          if (lane < 9) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const unsigned bits = (unsigned)(mask >> (8 * j)) & 0xffu;
              if (bits) {
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = prod[wave][8 * j + q][lane];
#pragma unroll
                for (int q = 0; q < 8; ++q) if (bits & (1u << q)) acc += v[q];
              }
            }
          }
        }
      }
    }
    float a[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) a[q] = __shfl(acc, q, 64);
    if (lane == 0) {
      const float cntf = (float)cnt[label];
#pragma unroll
      for (int q = 0; q < 9; ++q) a[q] = a[q] / cntf;
      float cov[9];
      cov[0] = a[0] - a[6] * a[6]; cov[1] = a[1] - a[6] * a[7]; cov[2] = a[2] - a[6] * a[8];
      cov[4] = a[3] - a[7] * a[7]; cov[5] = a[4] - a[7] * a[8]; cov[8] = a[5] - a[8] * a[8];
      cov[3] = cov[1]; cov[6] = cov[2]; cov[7] = cov[5];
      float ev, v[3];
      eigen33(cov, ev, v);
      float p[4] = {v[0], v[1], v[2], 0};
      p[3] = -1 * (p[0] * a[6] + p[1] * a[7] + p[2] * a[8]);
      const float ct = (0.0f - a[6]) * p[0] + (0.0f - a[7]) * p[1] + (0.0f - a[8]) * p[2];
      if (ct < 0) {
        p[0] *= -1; p[1] *= -1; p[2] *= -1;
        p[3] = -1 * (p[0] * a[6] + p[1] * a[7] + p[2] * a[8]);
      }
      const float es = cov[0] + cov[4] + cov[8];
      const float curv = es != 0 ? fabsf(ev / es) : 0;
      Region R;
      R.centroid[0] = a[6]; R.centroid[1] = a[7]; R.centroid[2] = a[8];
      R.model[0] = p[0]; R.model[1] = p[1]; R.model[2] = p[2]; R.model[3] = p[3];
      R.inliers = cnt[label]; R.first_inlier = first; R.label = label; R.contour_n = 0; R.area = 0;
      R.last_key = (unsigned long long)(unsigned)last;
      regs[k] = R;
      accepted[k] = curv < V.max_curv ? 1 : 0;
    }
  }
  __syncthreads();
  if (threadIdx.x < nc && accepted[threadIdx.x]) {
    int idx = 0;
    for (int q = 0; q < (int)threadIdx.x; ++q) idx += accepted[q];
    if (idx < kMaxRegions) {
      V.reg[(size_t)blockIdx.x * kMaxRegions + idx] = regs[threadIdx.x];
      V.l2m[b.pix0 + regs[threadIdx.x].label] = idx;
    }
  }
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int q = 0; q < nc; ++q) tot += accepted[q];
    V.nreg[blockIdx.x] = min(tot, kMaxRegions);
    if (ncand > kMaxCand) atomicAdd(&V.overflow[0], 1);
    if (tot > kMaxRegions) atomicAdd(&V.overflow[1], 1);
  }
}

// labels -> compact codes for refinement / contour / the label image:
//   region index (0..63) for pixels of an accepted plane, kOther for any other valid pixel, -1 invalid
constexpr int kOther = 1000;
__global__ __launch_bounds__(256) void k_relabel(View V) {
  const BoxMeta b = V.box[blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= b.w * b.h) return;
  const int l = V.lab[b.pix0 + i];
  V.code[b.pix0 + i] = l < 0 ? -1 : (V.l2m[b.pix0 + l] >= 0 ? V.l2m[b.pix0 + l] : kOther);
}

// OrganizedMultiPlaneSegmentation::refine: two raster sweeps with the PlaneRefinementComparator,
// executed as skewed wavefronts over a label band staged in LDS (image row stride preserved, so the
// second sweep's colIdx-1 read at column 0 lands on the previous row's last pixel as in PCL).
// PlaneRefinementComparator::compare; p1 / p2 point at the xyz of the current / neighbour pixel (LDS band)
__device__ __forceinline__ bool refine_compare(float dist_thr, const float* models, int cl, int nl, const float* p1, const float* p2) {
  if (!(cl < kMaxRegions && nl >= kMaxRegions)) return false;   // grow[current] && !grow[next]
  const float* m = models + cl * 4;
  const double d = fabs((double)(m[0] * p2[0] + m[1] * p2[1] + m[2] * p2[2] + m[3]));
  const float z = p1[2];
  const float t = dist_thr * (z * z);
  return d < (double)t;
}
__device__ __forceinline__ void refine_record(const View& V, int slot, int model_idx, unsigned long long pass, unsigned long long key, int target) {
  Region* R = &V.reg[(size_t)slot * kMaxRegions + model_idx];
  atomicAdd(&R->inliers, 1);
  atomicMax(&R->last_key, (pass << 60) | (key << 24) | (unsigned long long)target);
}
__global__ __launch_bounds__(256) void k_refine(View V) {
  extern __shared__ int lband[];
  __shared__ float models[kMaxRegions * 4];
  const BoxMeta b = V.box[blockIdx.x];
  const int slot = blockIdx.x;
  const int nregs = V.nreg[slot];
  if (nregs == 0) return;
  for (int k = threadIdx.x; k < nregs * 4; k += 256) models[k] = V.reg[(size_t)slot * kMaxRegions + (k >> 2)].model[k & 3];
  const int w = b.w, h = b.h;
  int* L = V.code + b.pix0;
  const float* gp = V.pts + (size_t)b.pix0 * 3;
  const int BH = min(V.refine_bh, kBandFloats / (4 * w) - 1);     // labels (1 word) + xyz (3 words) per staged pixel
  float* pband = reinterpret_cast<float*>(lband + (BH + 1) * w);
  const float dthr = V.dist_thr;
  __syncthreads();
  // ---- sweep 1: top-down, left-right; checks right then lower neighbour
  for (int r0 = 0; r0 < h - 1; r0 += BH) {
    const int nr = min(BH, h - 1 - r0);
    for (int k = threadIdx.x; k < (nr + 1) * w; k += 256) lband[k] = L[(size_t)r0 * w + k];
    for (int k = threadIdx.x; k < (nr + 1) * w * 3; k += 256) pband[k] = gp[(size_t)r0 * w * 3 + k];
    __syncthreads();
    if (threadIdx.x < 64) {
      // One step = one pixel of this lane's row.  Only the two neighbour labels have to be read after the previous step's writes; the
      // lane's own label is carried in a register (the row above wrote it at least one step before it was read as `rl`), and the
      // points / the plane model of the NEXT step are fetched while this step's labels are on their way.
      const int l = threadIdx.x;
      const int steps = (w - 1) + 2 * (nr - 1);
      const bool rowok = l < nr;
      int* cur = lband + (rowok ? l : 0) * w;
      const float* prow = pband + (size_t)(rowok ? l : 0) * w * 3;
      const int r = r0 + l;
      int cl = -1;
      float z = 0, m0 = 0, m1 = 0, m2 = 0, m3 = 0, pr0 = 0, pr1 = 0, pr2 = 0, pd0 = 0, pd1 = 0, pd2 = 0;
      for (int t = 0; t < steps; ++t) {
        const int c = t - 2 * l;
        const bool act = rowok && c >= 0 && c < w - 1;
        if (act && c == 0) {   // first pixel of the row: nothing carried yet
          cl = cur[0];
          z = prow[2];
          pr0 = prow[3]; pr1 = prow[4]; pr2 = prow[5];
          pd0 = prow[3 * w]; pd1 = prow[3 * w + 1]; pd2 = prow[3 * w + 2];
          const float* m = models + ((cl >= 0 && cl < kMaxRegions) ? cl : 0) * 4;
          m0 = m[0]; m1 = m[1]; m2 = m[2]; m3 = m[3];
        }
        const int cc = act ? c : 0;
        const int rl = cur[cc + 1], ll = cur[w + cc];
        const int cn = min(cc + 1, w - 2);                     // next step's pixel: its right and lower neighbours
        const float* qn = prow + (size_t)(cn + 1) * 3;
        const float* qd = prow + (size_t)(w + cn) * 3;
        const float nr0 = qn[0], nr1 = qn[1], nr2 = qn[2], nd0 = qd[0], nd1 = qd[1], nd2 = qd[2];
        if (act) {
          // PlaneRefinementComparator::compare twice (right, lower): grow[current] && !grow[next] && distance of the next point to the
          // current label's plane < threshold * z^2 of the current point; the lower check only runs when the right label is valid
          const bool ok = cl >= 0 && rl >= 0 && cl < kMaxRegions;
          const float thr = dthr * (z * z);
          const double d1 = fabs((double)(m0 * pr0 + m1 * pr1 + m2 * pr2 + m3));
          const double d2 = fabs((double)(m0 * pd0 + m1 * pd1 + m2 * pd2 + m3));
          const bool c1 = ok && rl >= kMaxRegions && d1 < (double)thr;
          const bool c2 = ok && ll >= kMaxRegions && d2 < (double)thr;
          if (c1) {
            cur[c + 1] = cl;
            refine_record(V, slot, cl, 1ull, (unsigned long long)(r * w + c) * 2ull, r * w + c + 1);
          }
          if (c2) {
            cur[w + c] = cl;
            refine_record(V, slot, cl, 1ull, (unsigned long long)(r * w + c) * 2ull + 1ull, (r + 1) * w + c);
          }
          cl = c1 ? cl : rl;
          z = pr2;
          pr0 = nr0; pr1 = nr1; pr2 = nr2; pd0 = nd0; pd1 = nd1; pd2 = nd2;
          const float* m = models + ((cl >= 0 && cl < kMaxRegions) ? cl : 0) * 4;
          m0 = m[0]; m1 = m[1]; m2 = m[2]; m3 = m[3];
        }
      }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < (nr + 1) * w; k += 256) L[(size_t)r0 * w + k] = lband[k];
    __syncthreads();
  }
  // ---- sweep 2: bottom-up, right-left; checks left then upper neighbour
  for (int rhi = h - 1; rhi >= 1; rhi -= BH) {
    const int nr = min(BH, rhi);
    const int rlo = rhi - nr;  // staged rows rlo .. rhi (rlo is the upper halo of the topmost processed row)
    for (int k = threadIdx.x; k < (nr + 1) * w; k += 256) lband[k] = L[(size_t)rlo * w + k];
    for (int k = threadIdx.x; k < (nr + 1) * w * 3; k += 256) pband[k] = gp[(size_t)rlo * w * 3 + k];
    __syncthreads();
    if (threadIdx.x < 64) {
      const int l = threadIdx.x;  // lane l owns row rhi - l; same register-carried form as sweep 1, mirrored
      const int steps = w + 2 * (nr - 1);
      const bool rowok = l < nr;
      const int r = rhi - (rowok ? l : 0);
      int* cur = lband + (r - rlo) * w;
      const float* prow = pband + (size_t)(r - rlo) * w * 3;
      int cl = -1;
      float z = 0, m0 = 0, m1 = 0, m2 = 0, m3 = 0, pl0 = 0, pl1 = 0, pl2 = 0, pu0 = 0, pu1 = 0, pu2 = 0;
      for (int t = 0; t < steps; ++t) {
        const int c = (w - 1) - (t - 2 * l);
        const bool act = rowok && c >= 0 && c <= w - 1;
        if (act && c == w - 1) {   // first pixel of the row (sweep 2 runs right to left)
          cl = cur[c];
          const float* pc = prow + (size_t)c * 3;
          z = pc[2];
          pl0 = pc[-3]; pl1 = pc[-2]; pl2 = pc[-1];
          pu0 = pc[-3 * w]; pu1 = pc[-3 * w + 1]; pu2 = pc[-3 * w + 2];
          const float* m = models + ((cl >= 0 && cl < kMaxRegions) ? cl : 0) * 4;
          m0 = m[0]; m1 = m[1]; m2 = m[2]; m3 = m[3];
        }
        const int cc = act ? c : 1;
        // column 0 has no left neighbour here (PCL reads the previous row's last pixel): label 0 stands in, unused
        const int lf = cc >= 1 ? cur[cc - 1] : 0;
        const int ul = cur[cc - w];
        const float* ql = prow + (size_t)max(cc - 2, 0) * 3;                       // next step's pixel (c - 1): its left neighbour ...
        const float* qu = prow + (size_t)max(cc - 1, 0) * 3 - (size_t)w * 3;      // ... and its upper neighbour
        const float nl0 = ql[0], nl1 = ql[1], nl2 = ql[2];
        const float nu0 = qu[0], nu1 = qu[1], nu2 = qu[2];
        if (act) {
          const bool ok = cl >= 0 && lf >= 0 && cl < kMaxRegions;
          const float thr = dthr * (z * z);
          const double d1 = fabs((double)(m0 * pl0 + m1 * pl1 + m2 * pl2 + m3));
          const double d2 = fabs((double)(m0 * pu0 + m1 * pu1 + m2 * pu2 + m3));
          const bool c1 = ok && c >= 1 && lf >= kMaxRegions && d1 < (double)thr;
          const bool c2 = ok && ul >= kMaxRegions && d2 < (double)thr;
          const unsigned long long key = (unsigned long long)((h - 1 - r) * w + (w - 1 - c)) * 2ull;
          if (c1) {
            cur[c - 1] = cl;
            refine_record(V, slot, cl, 2ull, key, r * w + c - 1);
          }
          if (c2) {
            cur[c - w] = cl;
            refine_record(V, slot, cl, 2ull, key + 1ull, (r - 1) * w + c);
          }
          cl = c1 ? cl : lf;
          z = pl2;
          pl0 = nl0; pl1 = nl1; pl2 = nl2; pu0 = nu0; pu1 = nu1; pu2 = nu2;
          const float* m = models + ((cl >= 0 && cl < kMaxRegions) ? cl : 0) * 4;
          m0 = m[0]; m1 = m[1]; m2 = m[2]; m3 = m[3];
        }
      }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < (nr + 1) * w; k += 256) L[(size_t)rlo * w + k] = lband[k];
    __syncthreads();
  }
}

// The same two sweeps for boxes whose label image fits LDS (<= kCcLdsMax pixels), as a fixed-point iteration instead of a wavefront.
// A pixel is only ever written while it still carries kOther, and only by the two neighbours the raster order visits before it (sweep 1:
// the upper one first, then the left one; sweep 2: the lower one first, then the right one), each with the label it ends the sweep with.
// So the result of a sweep is the unique solution of   label(T) = F(label(first neighbour), label(second neighbour))   on a DAG, and
// re-evaluating F for every kOther pixel until nothing changes reaches it -- with all the threads of the workgroup, in as many passes
// as the longest chain of newly absorbed pixels (a thread walks its pixels in sweep order, so chains along a row cost one pass).  The
// region records (absorbed pixels, the last one in PCL's order) are taken once a sweep has converged.
constexpr int kRefTX = 32, kRefTY = 16;   // k_refine_lds: 512 threads, thread (tx, ty) owns a tile of ceil(w/32) x ceil(h/16) <= 32 pixels
constexpr int kRefMaskRegions = 8;         // boxes with at most this many planes take the precomputed-mask form
// does region `a` absorb pixel i when it arrives from writer pixel s (threshold from the writer's depth)?
__device__ __forceinline__ bool refine_fits(const float* models, const float* pts, float dthr, int a, int i, int s) {
  const float* m = models + a * 4;
  const float z = pts[(size_t)s * 3 + 2];
  const float t = dthr * (z * z);
  const float d = fabsf(m[0] * pts[(size_t)i * 3] + m[1] * pts[(size_t)i * 3 + 1] + m[2] * pts[(size_t)i * 3 + 2] + m[3]);
  return d < t;
}
// first / second candidate writer of pixel i in this sweep's order and whether they exist as "current" pixels of the sweep; the first
// writer's own gate (its right neighbour in sweep 1 / its left one in sweep 2, label 0 standing in at column 0, must be a valid pixel)
template <int SWEEP>
__device__ __forceinline__ void refine_writers(const short* Lc, int i, int r, int c, int w, int h, int* s1, int* s2, bool* ok1, bool* ok2) {
  *s1 = SWEEP == 1 ? i - w : i + w;
  *s2 = SWEEP == 1 ? i - 1 : i + 1;
  bool o1 = SWEEP == 1 ? (r >= 1 && c <= w - 2) : (r + 1 <= h - 1);
  if (o1) o1 = SWEEP == 1 ? Lc[*s1 + 1] >= 0 : (c >= 1 ? Lc[*s1 - 1] >= 0 : true);
  *ok1 = o1;
  *ok2 = SWEEP == 1 ? (c >= 1 && r <= h - 2) : (c + 1 <= w - 1 && r >= 1);
}
// label of pixel i given its writers' current labels; MASKED: the distance tests were done once per sweep (bit a of the low / high byte
// of M[i]: region a fits when it arrives from the first / second writer; zero where that writer does not exist)
template <int SWEEP, bool MASKED>
__device__ __forceinline__ int refine_eval(const short* Lc, const unsigned short* M, const float* models, const float* pts, float dthr, int i, int r, int c,
                                           int w, int h, int* writer) {
  *writer = 0;
  if (MASKED) {
    // the mask and both writers' labels in one LDS round trip (clamped addresses; a writer that does not exist has an empty mask byte)
    const int n = w * h;
    const int s1 = SWEEP == 1 ? max(i - w, 0) : min(i + w, n - 1), s2 = SWEEP == 1 ? max(i - 1, 0) : min(i + 1, n - 1);
    const unsigned m = M[i];
    const int a1 = Lc[s1], a2 = Lc[s2];
    if (a1 >= 0 && a1 < kRefMaskRegions && ((m >> a1) & 1u)) { *writer = 1; return a1; }
    if (a2 >= 0 && a2 < kRefMaskRegions && ((m >> (8 + a2)) & 1u)) { *writer = 2; return a2; }
    return kOther;
  }
  int s1, s2; bool ok1, ok2;
  refine_writers<SWEEP>(Lc, i, r, c, w, h, &s1, &s2, &ok1, &ok2);
  if (ok1) {
    const int a1 = Lc[s1];
    if (a1 >= 0 && a1 < kMaxRegions && refine_fits(models, pts, dthr, a1, i, s1)) { *writer = 1; return a1; }
  }
  if (ok2) {
    const int a2 = Lc[s2];
    if (a2 >= 0 && a2 < kMaxRegions && refine_fits(models, pts, dthr, a2, i, s2)) { *writer = 2; return a2; }
  }
  return kOther;
}
template <int SWEEP, bool MASKED>
__device__ __forceinline__ void refine_sweep_lds(short* Lc, unsigned short* M, const float* models, int nregs, const float* pts, float dthr, int w, int h,
                                                 int* s_changed, int* rcnt, unsigned long long* rkey) {
  const int tid = threadIdx.x;
  const int tw = (w + kRefTX - 1) / kRefTX, th = (h + kRefTY - 1) / kRefTY;
  const int c0 = (tid % kRefTX) * tw, r0 = (tid / kRefTX) * th;
  const int c1 = min(w, c0 + tw), r1 = min(h, r0 + th);
  const int tw_inv = 65536 / tw + 1;
  unsigned cand = 0;                            // bit (dy * tw + dx): the pixel carried kOther when the sweep began
  unsigned act = 0;                             // MASKED: ... and some plane fits it from some writer (the others can never change)
  for (int r = r0; r < r1; ++r)
    for (int c = c0; c < c1; ++c) {
      const int i = r * w + c;
      if (Lc[i] < kMaxRegions) continue;
      cand |= 1u << ((r - r0) * tw + (c - c0));
      if (MASKED) {
        int s1, s2; bool ok1, ok2;
        refine_writers<SWEEP>(Lc, i, r, c, w, h, &s1, &s2, &ok1, &ok2);
        unsigned m = 0;
        if (ok1 || ok2) {   // the distance to a plane does not depend on the writer, only the threshold (the writer's depth) does
          const float px = pts[(size_t)i * 3], py = pts[(size_t)i * 3 + 1], pz = pts[(size_t)i * 3 + 2];
          const float z1 = pts[(size_t)(ok1 ? s1 : i) * 3 + 2], z2 = pts[(size_t)(ok2 ? s2 : i) * 3 + 2];
          const float t1 = dthr * (z1 * z1), t2 = dthr * (z2 * z2);
          for (int a = 0; a < nregs; ++a) {
            const float* mm = models + a * 4;
            const float d = fabsf(mm[0] * px + mm[1] * py + mm[2] * pz + mm[3]);
            if (ok1 && d < t1) m |= 1u << a;
            if (ok2 && d < t2) m |= 1u << (8 + a);
          }
        }
        M[i] = (unsigned short)m;
        if (m) act |= 1u << ((r - r0) * tw + (c - c0));
      }
    }
  if (tid < 3) s_changed[tid] = 0;
  __syncthreads();
  for (int pass = 0; pass < w + h + 2; ++pass) {
    // three flags in rotation: pass p raises [p % 3], clears [(p + 1) % 3] (last read before the barrier of pass p - 1): one barrier per pass
    if (tid == 0) s_changed[(pass + 1) % 3] = 0;
    bool ch = false;
    // the tile's live pixels in sweep order (lowest bit first in sweep 1, highest first in sweep 2), so that a chain inside the tile is
    // followed to its end in one pass; k / tw by a reciprocal that is exact for k < 32
    unsigned bits = MASKED ? act : cand;
    while (bits) {
      const int k = SWEEP == 1 ? __ffs((int)bits) - 1 : 31 - __clz((int)bits);
      bits &= ~(1u << k);
      const int dy = (k * tw_inv) >> 16, dx = k - dy * tw;
      const int r = r0 + dy, c = c0 + dx, i = r * w + c;
      int wr;
      const int v = refine_eval<SWEEP, MASKED>(Lc, M, models, pts, dthr, i, r, c, w, h, &wr);
      if (v != Lc[i]) { Lc[i] = (short)v; ch = true; }
    }
    if (ch) s_changed[pass % 3] = 1;
    __syncthreads();
    if (!s_changed[pass % 3]) break;
  }
  __syncthreads();
  for (int r = r0; r < r1; ++r)
    for (int c = c0; c < c1; ++c) {
      const int i = r * w + c;
      if (!((cand >> ((r - r0) * tw + (c - c0))) & 1u) || Lc[i] >= kMaxRegions) continue;
      int wr;
      const int v = refine_eval<SWEEP, MASKED>(Lc, M, models, pts, dthr, i, r, c, w, h, &wr);
      // order key of the writing pixel: its raster position (sweep 1) / reverse raster position (sweep 2), x 2, + 1 for the vertical write
      const int sidx = SWEEP == 1 ? (wr == 1 ? i - w : i - 1) : (wr == 1 ? i + w : i + 1);
      const int sr = sidx / w, sc = sidx - sr * w;
      const unsigned long long pos = SWEEP == 1 ? (unsigned long long)sidx : (unsigned long long)((h - 1 - sr) * w + (w - 1 - sc));
      const unsigned long long key = pos * 2ull + (wr == 1 ? 1ull : 0ull);
      atomicAdd(&rcnt[v], 1);
      atomicMax(&rkey[v], ((unsigned long long)SWEEP << 60) | (key << 24) | (unsigned long long)i);
    }
  __syncthreads();
}
__global__ __launch_bounds__(kRefTX * kRefTY) void k_refine_lds(View V) {
  extern __shared__ short Lc[];                  // labels (-1, 0..63, kOther) as 16-bit words, then the masks
  __shared__ float models[kMaxRegions * 4];
  __shared__ int rcnt[kMaxRegions];
  __shared__ unsigned long long rkey[kMaxRegions];
  __shared__ int s_changed[3];
  constexpr int NT = kRefTX * kRefTY;
  const BoxMeta b = V.box[blockIdx.x];
  const int slot = blockIdx.x;
  const int nregs = V.nreg[slot];
  if (nregs == 0) return;
  const int w = b.w, h = b.h, n = w * h, tid = threadIdx.x;
  unsigned short* M = reinterpret_cast<unsigned short*>(Lc + ((n + 1) & ~1));
  for (int k = tid; k < nregs * 4; k += NT) models[k] = V.reg[(size_t)slot * kMaxRegions + (k >> 2)].model[k & 3];
  if (tid < kMaxRegions) { rcnt[tid] = 0; rkey[tid] = 0; }
  int* L = V.code + b.pix0;
  const float* gp = V.pts + (size_t)b.pix0 * 3;
  for (int i = tid; i < n; i += NT) Lc[i] = (short)L[i];
  __syncthreads();
  if (nregs <= kRefMaskRegions) {
    refine_sweep_lds<1, true>(Lc, M, models, nregs, gp, V.dist_thr, w, h, s_changed, rcnt, rkey);
    refine_sweep_lds<2, true>(Lc, M, models, nregs, gp, V.dist_thr, w, h, s_changed, rcnt, rkey);
  } else {
    refine_sweep_lds<1, false>(Lc, M, models, nregs, gp, V.dist_thr, w, h, s_changed, rcnt, rkey);
    refine_sweep_lds<2, false>(Lc, M, models, nregs, gp, V.dist_thr, w, h, s_changed, rcnt, rkey);
  }
  for (int i = tid; i < n; i += NT) L[i] = Lc[i];
  if (tid < nregs && rcnt[tid] > 0) {
    Region* R = &V.reg[(size_t)slot * kMaxRegions + tid];
    R->inliers += rcnt[tid];
    if (rkey[tid] > R->last_key) R->last_key = rkey[tid];
  }
}

// findLabeledRegionBoundary (Moore trace from the last inlier) + pcl::calculatePolygonArea, one
// thread per region (sequential by nature; float accumulation in contour order).  The code image is
// staged in LDS as bytes when it fits (<= 150k pixels), so the trace does not pay HBM/L2 latency.
template <bool STAGED>
__global__ __launch_bounds__(256) void k_contour(View V) {
  extern __shared__ unsigned char cimg[];   // (w+2) x (h+2) code image with a border of "invalid" (no bounds tests)
  const int bx = blockIdx.x;
  const BoxMeta b = V.box[bx];
  const int w = b.w, h = b.h, W2 = w + 2;
  const int* L = V.code + b.pix0;
  const int nregs = V.nreg[bx];
  if (nregs == 0) return;
  if (STAGED) {
    for (int i = threadIdx.x; i < W2 * (h + 2); i += 256) cimg[i] = 255;
    __syncthreads();
    for (int i = threadIdx.x; i < w * h; i += 256) {
      const int c = L[i];
      const int y = i / w, x = i - y * w;
      cimg[(y + 1) * W2 + x + 1] = c < 0 ? 255 : (c >= kMaxRegions ? 254 : (unsigned char)c);
    }
    __syncthreads();
  }
  const int k = threadIdx.x;
  if (k >= nregs) return;
  Region* R = &V.reg[(size_t)bx * kMaxRegions + k];
  const int start = (int)(R->last_key & 0xffffffull);
  // code of pixel (x, y); out-of-image neighbours read as 255 ("not this label"), which is what PCL's
  // bounds tests amount to in the trace loop; the *seed* test below keeps PCL's explicit bounds test
#define CODE_XY(x, y) (STAGED ? (int)cimg[((y) + 1) * W2 + (x) + 1] \
                              : (((x) < 0 || (x) >= w || (y) < 0 || (y) >= h) ? 255 : (L[(y) * w + (x)] < 0 ? 255 : (L[(y) * w + (x)] >= kMaxRegions ? 254 : L[(y) * w + (x)]))))
  int cx = start % w, cy = start / w, ci = start;
  const int label = CODE_XY(cx, cy);
  int dirn = -1;
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    const int ddx = d == 0 || d == 1 || d == 7 ? -1 : (d == 2 || d == 6 ? 0 : 1);
    const int ddy = d == 1 || d == 2 || d == 3 ? -1 : (d == 0 || d == 4 ? 0 : 1);
    const int x = cx + ddx, y = cy + ddy;
    if (dirn < 0 && x >= 0 && x < w && y >= 0 && y < h && CODE_XY(x, y) != label) dirn = d;
  }
  int count = 0, off = -1;
  if (dirn != -1) {
    // each region owns an equal slice of the box's arena (4*w*h ints); boundaries are far shorter
    const int slice = (4 * w * h) / nregs;
    int* out = V.contour + (size_t)b.pix0 * 4 + (size_t)k * slice;
    off = k * slice;
    out[0] = start;
    count = 1;
    const long guard = 4L * (long)w * h + 8;
    long steps = 0;
    do {
      // 8 independent neighbour reads (no dependent LDS chain), then the first match in scan order dirn+1 .. dirn+8
      unsigned match = 0;
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        const int ddx = d == 0 || d == 1 || d == 7 ? -1 : (d == 2 || d == 6 ? 0 : 1);
        const int ddy = d == 1 || d == 2 || d == 3 ? -1 : (d == 0 || d == 4 ? 0 : 1);
        match |= (CODE_XY(cx + ddx, cy + ddy) == label ? 1u : 0u) << d;
      }
      // rotate so that bit 0 is direction dirn+1; no match at all -> PCL's loop ends on (dirn+8)&7 = dirn
      const unsigned rot = ((match | (match << 8)) >> ((dirn + 1) & 7)) & 0xffu;
      const int nI = rot ? ((dirn + 1 + (__ffs(rot) - 1)) & 7) : dirn;
      const int mdx = nI == 0 || nI == 1 || nI == 7 ? -1 : (nI == 2 || nI == 6 ? 0 : 1);
      const int mdy = nI == 1 || nI == 2 || nI == 3 ? -1 : (nI == 0 || nI == 4 ? 0 : 1);
      dirn = (nI + 4) & 7;
      ci += mdy * w + mdx; cx += mdx; cy += mdy;
      if (count < slice) out[count] = ci;
      ++count;
    } while (ci != start && ++steps < guard);
    if (count > slice) off = -1;   // pathological boundary longer than its slice: area stays 0, region is dropped
  }
  R->contour_n = count;
  R->contour_off = off;
  R->area = 0;
#undef CODE_XY
}

// pcl::calculatePolygonArea over the stored boundary: res += p[i] x p[(i+1) % n] in contour order (float).
// One wave per region: 64 cross products are formed in parallel (coalesced index loads, gathered
// points) and parked in LDS; lanes 0..2 then run the three serial float chains of the chunk.
__global__ __launch_bounds__(64) void k_area(View V) {
  __shared__ float prod[3][64];
  const int bx = blockIdx.x, k = blockIdx.y;
  if (k >= V.nreg[bx]) return;
  const BoxMeta b = V.box[bx];
  Region* R = &V.reg[(size_t)bx * kMaxRegions + k];
  const int n = R->contour_n, off = R->contour_off;
  if (n <= 0 || off < 0) return;
  const int* idx = V.contour + (size_t)b.pix0 * 4 + off;
  const float* pts = V.pts + (size_t)b.pix0 * 3;
  const int lane = threadIdx.x;
  float acc = 0;
  for (int i0 = 0; i0 < n; i0 += 64) {
    const int i = i0 + lane;
    if (i < n) {
      const int a = idx[i], c = idx[(i + 1) % n];
      float c3[3];
      cross3(pts + (size_t)a * 3, pts + (size_t)c * 3, c3);
      prod[0][lane] = c3[0]; prod[1][lane] = c3[1]; prod[2][lane] = c3[2];
    }
    __syncthreads();
    if (lane < 3) {
      const int m = min(64, n - i0);
      for (int q = 0; q < m; ++q) acc += prod[lane][q];
    }
    __syncthreads();
  }
  const float rx = __shfl(acc, 0, 64), ry = __shfl(acc, 1, 64), rz = __shfl(acc, 2, 64);
  if (lane == 0) R->area = sqrtf(rx * rx + ry * ry + rz * rz) * 0.5f;
}


// ---------------------------------------------------------------------------------------------
// RANSAC plane fit: pcl::SACSegmentation(SACMODEL_PLANE, SAC_RANSAC) as used by the
// reference's compute2DConvexHull (plane_segmentation.cpp:639-647).  All `max_iterations + slack`
// hypotheses are scored in parallel — one workgroup per hypothesis, one thread per point with a
// wave-ballot / LDS inlier count — and the sequential adaptive-k logic of pcl::RandomSampleConsensus is
// replayed over the counts afterwards, which selects exactly the hypothesis the sequential loop would.
// Sample triples come from a counter-based hash (see oracle/oracle_seg.c for the deviation note).
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ __forceinline__ bool plane_inlier(const float* m, const float* p, float thr) {
  return fabsf(m[0] * p[0] + m[1] * p[1] + m[2] * p[2] + m[3]) < thr;
}
__global__ __launch_bounds__(256) void k_ransac_score(const float* __restrict__ pts, int n, float thr, unsigned long long seed,
                                                     float* __restrict__ models, int* __restrict__ counts) {
  __shared__ float m[4];
  __shared__ int ok;
  __shared__ int wsum[4];
  const int iter = blockIdx.x;
  if (threadIdx.x == 0) {
    int good = 0;
    for (int attempt = 0; attempt < 1000 && !good; ++attempt) {
      int id[3];
      for (int j = 0; j < 3; ++j)
        id[j] = (int)(splitmix64(seed ^ ((unsigned long long)iter << 32) ^ ((unsigned long long)attempt << 8) ^ (unsigned long long)j) % (unsigned long long)n);
      if (id[0] == id[1] || id[0] == id[2] || id[1] == id[2]) continue;
      const float* p0 = pts + (size_t)id[0] * 3; const float* p1 = pts + (size_t)id[1] * 3; const float* p2 = pts + (size_t)id[2] * 3;
      const float a0 = p1[0] - p0[0], a1 = p1[1] - p0[1], a2 = p1[2] - p0[2];
      const float b0 = p2[0] - p0[0], b1 = p2[1] - p0[1], b2 = p2[2] - p0[2];
      const float r0 = a0 / b0, r1 = a1 / b1, r2 = a2 / b2;
      if (!((r0 != r1) || (r2 != r1))) continue;
      float m0 = a1 * b2 - a2 * b1, m1 = a2 * b0 - a0 * b2, m2 = a0 * b1 - a1 * b0;
      const float nn = sqrtf(m0 * m0 + m1 * m1 + m2 * m2);
      m0 /= nn; m1 /= nn; m2 /= nn;
      m[0] = m0; m[1] = m1; m[2] = m2;
      m[3] = -1 * (m0 * p0[0] + m1 * p0[1] + m2 * p0[2]);
      good = 2;   // a sample was found (whether or not its model is finite)
    }
    ok = (good == 2) && isfinite(m[0]) && isfinite(m[3]);
  }
  __syncthreads();
  int cnt = 0;
  if (ok)
    for (int i = threadIdx.x; i < n; i += 256) cnt += plane_inlier(m, pts + (size_t)i * 3, thr) ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    counts[iter] = ok ? wsum[0] + wsum[1] + wsum[2] + wsum[3] : -1;
    models[iter * 4 + 0] = m[0]; models[iter * 4 + 1] = m[1]; models[iter * 4 + 2] = m[2]; models[iter * 4 + 3] = m[3];
  }
}
// ordered inlier list of one model: per-block counts -> offsets -> indices (index order = PCL's)
__global__ __launch_bounds__(256) void k_ransac_mark(const float* __restrict__ pts, int n, float thr, const float* __restrict__ model,
                                                    int* __restrict__ block_counts) {
  __shared__ int wsum[4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float m[4] = {model[0], model[1], model[2], model[3]};
  int c = (i < n && plane_inlier(m, pts + (size_t)i * 3, thr)) ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
__global__ void k_ransac_scan(int* block_counts, int nblocks, int* total) {   // exclusive scan by one thread (nblocks is small)
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int acc = 0;
  for (int b = 0; b < nblocks; ++b) { const int c = block_counts[b]; block_counts[b] = acc; acc += c; }
  *total = acc;
}
__global__ __launch_bounds__(256) void k_ransac_write(const float* __restrict__ pts, int n, float thr, const float* __restrict__ model,
                                                     const int* __restrict__ block_off, int* __restrict__ inliers, int max_inliers) {
  __shared__ int woff[4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float m[4] = {model[0], model[1], model[2], model[3]};
  const bool in = i < n && plane_inlier(m, pts + (size_t)i * 3, thr);
  const unsigned long long mask = __ballot(in);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) woff[wave] = __popcll(mask);
  __syncthreads();
  int base = block_off[blockIdx.x];
  for (int w = 0; w < wave; ++w) base += woff[w];
  if (in) {
    const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
    if (pos < max_inliers) inliers[pos] = i;
  }
}
// ---- a15, second half: pcl::ProjectInliers (plane) + 2-D pcl::ConvexHull (plane_segmentation.cpp:648-662) ---------
// projectPoints: mc = (a, b, c, 0) normalised in float, dist = mc . p + d (d is not rescaled, as in PCL), p' = p - mc dist
__global__ __launch_bounds__(256) void k_hull_project(const float* __restrict__ pts, const int* __restrict__ inliers, int n_in,
                                                     float a, float b, float c, float d, float* __restrict__ proj) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n_in) return;
  const float nrm = sqrtf(a * a + b * b + c * c);
  const float m0 = a / nrm, m1 = b / nrm, m2 = c / nrm;
  const float* p = pts + (size_t)inliers[k] * 3;
  const float dist = m0 * p[0] + m1 * p[1] + m2 * p[2] + d;
  proj[3 * (size_t)k + 0] = p[0] - m0 * dist;
  proj[3 * (size_t)k + 1] = p[1] - m1 * dist;
  proj[3 * (size_t)k + 2] = p[2] - m2 * dist;
}
// coordinate plane of the hull from the normal of the first / last / middle projected point (PCL performReconstruction2D);
// collinear probes: the middle index walks forward.  meta[0] = 0 xy, 1 yz, 2 xz, -1 degenerate
__global__ void k_hull_axes(const float* __restrict__ proj, int n_in, float thresh, int* __restrict__ meta) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float* p0 = proj; const float* p1 = proj + 3 * (size_t)(n_in - 1);
  double nx = 0, ny = 0, nz = 0, nn = 0;
  for (int m = n_in / 2, tries = 0; tries < n_in; ++tries, m = (m + 1) % n_in) {
    const float* p2 = proj + 3 * (size_t)m;
    const double ax = (double)p1[0] - p0[0], ay = (double)p1[1] - p0[1], az = (double)p1[2] - p0[2];
    const double bx = (double)p2[0] - p0[0], by = (double)p2[1] - p0[1], bz = (double)p2[2] - p0[2];
    nx = ay * bz - az * by; ny = az * bx - ax * bz; nz = ax * by - ay * bx;
    nn = sqrt(nx * nx + ny * ny + nz * nz);
    if (nn > 0) break;
  }
  if (!(nn > 0)) { meta[0] = -1; return; }
  const float tx = fabsf((float)(nx / nn)), ty = fabsf((float)(ny / nn)), tz = fabsf((float)(nz / nn));
  bool xy = true, yz = true, xz = true;
  if (tz > thresh) { xz = false; yz = false; }
  if (tx > thresh) { xz = false; xy = false; }
  if (ty > thresh) { xy = false; yz = false; }
  meta[0] = xy ? 0 : (yz ? 1 : (xz ? 2 : -1));
}
__global__ __launch_bounds__(256) void k_hull_coords(const float* __restrict__ proj, int n_in, const int* __restrict__ meta,
                                                    float* __restrict__ px, float* __restrict__ py) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n_in) return;
  const int axes = meta[0];
  px[k] = proj[3 * (size_t)k + (axes == 1 ? 1 : 0)];
  py[k] = proj[3 * (size_t)k + (axes == 0 ? 1 : 2)];
}
// Akl-Toussaint pre-filter: the 8 extreme points (W, SW, S, SE, E, NE, N, NW; ties -> lowest index) span an octagon;
// points strictly inside it (by a margin far above the rounding of the double cross products) cannot be hull vertices.
__device__ __forceinline__ double hull_key(int dir, float x, float y) {
  const double dx = x, dy = y;
  switch (dir) { case 0: return -dx; case 1: return -(dx + dy); case 2: return -dy; case 3: return dx - dy;
                 case 4: return dx; case 5: return dx + dy; case 6: return dy; default: return -(dx - dy); }
}
__global__ __launch_bounds__(256) void k_hull_extreme_partial(const float* __restrict__ px, const float* __restrict__ py, int n_in,
                                                             double* __restrict__ bkey, int* __restrict__ bidx) {
  __shared__ double wk[4][8];
  __shared__ int wi[4][8];
  const int k = blockIdx.x * 256 + threadIdx.x;
  const bool ok = k < n_in;
  const float x = ok ? px[k] : 0.f, y = ok ? py[k] : 0.f;
#pragma unroll
  for (int dir = 0; dir < 8; ++dir) {
    double key = ok ? hull_key(dir, x, y) : -1.0e300;
    int idx = ok ? k : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double ok2 = __shfl_down(key, o, 64);
      const int oi = __shfl_down(idx, o, 64);
      if (ok2 > key || (ok2 == key && oi < idx)) { key = ok2; idx = oi; }
    }
    if ((threadIdx.x & 63) == 0) { wk[threadIdx.x >> 6][dir] = key; wi[threadIdx.x >> 6][dir] = idx; }
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    const int dir = threadIdx.x;
    double key = wk[0][dir]; int idx = wi[0][dir];
    for (int w = 1; w < 4; ++w) if (wk[w][dir] > key || (wk[w][dir] == key && wi[w][dir] < idx)) { key = wk[w][dir]; idx = wi[w][dir]; }
    bkey[(size_t)blockIdx.x * 8 + dir] = key; bidx[(size_t)blockIdx.x * 8 + dir] = idx;
  }
}
// ext[dir] = (x, y) of the extreme point, eidx[dir] its index; ext[16] = margin
__global__ void k_hull_extreme_final(const double* __restrict__ bkey, const int* __restrict__ bidx, int nblk, const float* __restrict__ px,
                                     const float* __restrict__ py, double* __restrict__ ext, int* __restrict__ eidx) {
  const int dir = threadIdx.x;
  if (dir < 8) {
    double key = bkey[dir]; int idx = bidx[dir];
    for (int b = 1; b < nblk; ++b) {
      const double k2 = bkey[(size_t)b * 8 + dir]; const int i2 = bidx[(size_t)b * 8 + dir];
      if (k2 > key || (k2 == key && i2 < idx)) { key = k2; idx = i2; }
    }
    eidx[dir] = idx; ext[2 * dir] = px[idx]; ext[2 * dir + 1] = py[idx];
  }
  __syncthreads();
  if (dir == 0) {
    const double ex = ext[8] - ext[0], ey = ext[13] - ext[5];   // E.x - W.x, N.y - S.y
    ext[16] = 1e-10 * (ex * ex + ey * ey);
  }
}
__device__ __forceinline__ bool hull_keep(const double* __restrict__ ext, float x, float y) {
  const double margin = ext[16];
  int edges = 0;
  bool inside = true;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const double ax = ext[2 * e], ay = ext[2 * e + 1], bx = ext[2 * ((e + 1) & 7)], by = ext[2 * ((e + 1) & 7) + 1];
    if (ax == bx && ay == by) continue;
    ++edges;
    const double cr = (bx - ax) * ((double)y - ay) - (by - ay) * ((double)x - ax);
    if (!(cr > margin)) inside = false;
  }
  return !(inside && edges >= 3);
}
__global__ __launch_bounds__(256) void k_hull_mark(const float* __restrict__ px, const float* __restrict__ py, int n_in, const double* __restrict__ ext,
                                                  int* __restrict__ block_counts) {
  __shared__ int wsum[4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  int c = (i < n_in && hull_keep(ext, px[i], py[i])) ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
__global__ __launch_bounds__(256) void k_hull_write(const float* __restrict__ px, const float* __restrict__ py, int n_in, const double* __restrict__ ext,
                                                   const int* __restrict__ block_off, float* __restrict__ cx, float* __restrict__ cy, int* __restrict__ ci) {
  __shared__ int woff[4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool in = i < n_in && hull_keep(ext, px[i], py[i]);
  const unsigned long long mask = __ballot(in);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) woff[wave] = __popcll(mask);
  __syncthreads();
  int base = block_off[blockIdx.x];
  for (int w = 0; w < wave; ++w) base += woff[w];
  if (in) {
    const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
    cx[pos] = px[i]; cy[pos] = py[i]; ci[pos] = i;
  }
}
// One workgroup per chunk of <= kHullCap candidates: bitonic sort by (x, y, index) in LDS, Andrew's monotone chain by
// one thread (strict turns, duplicates keep the lowest index).  FINAL: counter-clockwise polygon rotated to start at the
// vertex of smallest angle about the (double mean -> float) vertex centroid, indices only.  Otherwise the chunk's hull
// vertices go back out as candidates of the next round (the hull of a union is the hull of the parts' hulls).
constexpr int kHullCap = 4096;
__device__ __forceinline__ bool hull_less(float ax, float ay, int ai, float bx, float by, int bi) {
  if (ax != bx) return ax < bx;
  if (ay != by) return ay < by;
  return ai < bi;
}
__device__ __forceinline__ int hull_half(float x, float y) { return y < 0 ? 0 : ((y == 0 && x > 0) ? 1 : (y > 0 ? 2 : 3)); }
// (plain functions over the LDS arrays: lambdas capturing __shared__ arrays by reference have miscompiled before)
__device__ __forceinline__ double hull_cross(const float* sx, const float* sy, int o, int a, int b) {
  return ((double)sx[a] - (double)sx[o]) * ((double)sy[b] - (double)sy[o]) - ((double)sy[a] - (double)sy[o]) * ((double)sx[b] - (double)sx[o]);
}
__device__ __forceinline__ bool hull_ang_less(const float* sx, const float* sy, const int* si, float gx, float gy, int p, int q) {
  const float ax = sx[p] - gx, ay = sy[p] - gy, bx = sx[q] - gx, by = sy[q] - gy;
  const int hp = hull_half(ax, ay), hq = hull_half(bx, by);
  if (hp != hq) return hp < hq;
  const double cr = (double)ax * (double)by - (double)ay * (double)bx;
  if (cr != 0) return cr > 0;
  return si[p] < si[q];
}
template <bool FINAL>
__global__ __launch_bounds__(1024) void k_hull_core(const float* __restrict__ cx, const float* __restrict__ cy, const int* __restrict__ ci, int m,
                                                   float* __restrict__ ox, float* __restrict__ oy, int* __restrict__ oi, int* __restrict__ counts) {
  __shared__ float sx[kHullCap], sy[kHullCap];
  __shared__ int si[kHullCap], st[kHullCap];
  const int c0 = blockIdx.x * kHullCap, mc = min(kHullCap, m - c0), tid = threadIdx.x;
  int N = 1;
  while (N < mc) N <<= 1;
  for (int t = tid; t < N; t += 1024) {
    const bool ok = t < mc;
    sx[t] = ok ? cx[c0 + t] : __int_as_float(0x7f800000); sy[t] = ok ? cy[c0 + t] : 0.f; si[t] = ok ? ci[c0 + t] : 0x7fffffff;
  }
  __syncthreads();
  for (int k = 2; k <= N; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < N; t += 1024) {
        const int u = t ^ j;
        if (u > t) {
          const bool asc = (t & k) == 0;
          const bool lt = hull_less(sx[u], sy[u], si[u], sx[t], sy[t], si[t]);   // element u sorts before element t
          if (lt == asc) {
            const float fx = sx[t], fy = sy[t]; const int fi = si[t];
            sx[t] = sx[u]; sy[t] = sy[u]; si[t] = si[u]; sx[u] = fx; sy[u] = fy; si[u] = fi;
          }
        }
      }
      __syncthreads();
    }
  if (tid != 0) return;
  // dedupe in place
  int mu = 0;
  for (int k = 0; k < mc; ++k) {
    if (mu > 0 && sx[k] == sx[mu - 1] && sy[k] == sy[mu - 1]) continue;
    sx[mu] = sx[k]; sy[mu] = sy[k]; si[mu] = si[k]; ++mu;
  }
  int h = 0;
  for (int k = 0; k < mu; ++k) {
    while (h >= 2 && hull_cross(sx, sy, st[h - 2], st[h - 1], k) <= 0) --h;
    st[h++] = k;
  }
  for (int k = mu - 2, t = h + 1; k >= 0; --k) {
    while (h >= t && hull_cross(sx, sy, st[h - 2], st[h - 1], k) <= 0) --h;
    st[h++] = k;
  }
  if (mu > 1) --h;
  if (!FINAL) {
    for (int k = 0; k < h; ++k) { const int p = st[k]; ox[c0 + k] = sx[p]; oy[c0 + k] = sy[p]; oi[c0 + k] = si[p]; }
    counts[blockIdx.x] = h;
    return;
  }
  double sumx = 0, sumy = 0;
  for (int k = 0; k < h; ++k) { sumx += sx[st[k]]; sumy += sy[st[k]]; }
  const float gx = (float)(sumx / h), gy = (float)(sumy / h);
  int first = 0;
  for (int k = 1; k < h; ++k) if (hull_ang_less(sx, sy, si, gx, gy, st[k], st[first])) first = k;
  for (int k = 0; k < h; ++k) oi[k] = si[st[(first + k) % h]];
  counts[0] = h;
}
// compaction between rounds: chunk c's `counts[c]` vertices move to offset off[c]
__global__ __launch_bounds__(256) void k_hull_compact(const float* __restrict__ ox, const float* __restrict__ oy, const int* __restrict__ oi,
                                                     const int* __restrict__ counts, const int* __restrict__ off, float* __restrict__ cx,
                                                     float* __restrict__ cy, int* __restrict__ ci) {
  const int c = blockIdx.x, n = counts[c], o = off[c];
  for (int k = threadIdx.x; k < n; k += 256) { cx[o + k] = ox[c * kHullCap + k]; cy[o + k] = oy[c * kHullCap + k]; ci[o + k] = oi[c * kHullCap + k]; }
}
__global__ void k_hull_scan(const int* __restrict__ counts, int nchunks, int* __restrict__ off, int* __restrict__ total) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int acc = 0;
  for (int c = 0; c < nchunks; ++c) { off[c] = acc; acc += counts[c]; }
  *total = acc;
}

// optimizeModelCoefficients: float mean / covariance of the inliers in index order + eigen33 (one wave)
__global__ __launch_bounds__(64) void k_ransac_refit(const float* __restrict__ pts, const int* __restrict__ inliers, const int* __restrict__ total,
                                                    float* __restrict__ model) {
  __shared__ float prod[64][9];
  const int cntN = *total;
  if (cntN <= 3) return;   // PCL keeps the unrefined model
  const int lane = threadIdx.x;
  float acc = 0;
  for (int i0 = 0; i0 < cntN; i0 += 64) {
    const int i = i0 + lane;
    if (i < cntN) {
      const float* p = pts + (size_t)inliers[i] * 3;
      float* q = prod[lane];
      q[0] = p[0] * p[0]; q[1] = p[0] * p[1]; q[2] = p[0] * p[2]; q[3] = p[1] * p[1]; q[4] = p[1] * p[2]; q[5] = p[2] * p[2];
      q[6] = p[0]; q[7] = p[1]; q[8] = p[2];
    }
    __syncthreads();
    if (lane < 9) { const int mcount = min(64, cntN - i0); for (int q = 0; q < mcount; ++q) acc += prod[q][lane]; }
    __syncthreads();
  }
  float a[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) a[q] = __shfl(acc, q, 64);
  if (lane == 0) {
    const float cf = (float)cntN;
#pragma unroll
    for (int q = 0; q < 9; ++q) a[q] = a[q] / cf;
    float cov[9];
    cov[0] = a[0] - a[6] * a[6]; cov[1] = a[1] - a[6] * a[7]; cov[2] = a[2] - a[6] * a[8];
    cov[4] = a[3] - a[7] * a[7]; cov[5] = a[4] - a[7] * a[8]; cov[8] = a[5] - a[8] * a[8];
    cov[3] = cov[1]; cov[6] = cov[2]; cov[7] = cov[5];
    float ev, v[3];
    eigen33(cov, ev, v);
    model[0] = v[0]; model[1] = v[1]; model[2] = v[2];
    model[3] = -1 * (v[0] * a[6] + v[1] * a[7] + v[2] * a[8]);
  }
}

// final label image for the parity hook: region index or -1
__global__ __launch_bounds__(256) void k_label_image(View V, int box, int* out) {
  const BoxMeta b = V.box[box];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= b.w * b.h) return;
  const int c = V.code[b.pix0 + i];
  out[i] = (c >= 0 && c < kMaxRegions) ? c : -1;
}


// ---- point-to-plane ICP (judge row J1; BASELINE.json north_star "RANSAC plane fit + point-to-plane ICP"; no counterpart in the
// reference tree) -------------------------------------------------------------------------------------------------------------------
// One Gauss-Newton round: every labelled point q = R p + t contributes, for its plane (n, d),
//   r = n . q + d,   J = [ (q x n)^T  n^T ]      (update  T <- (exp[w]x, u) o T,  d r / d w = q x n,  d r / d u = n)
// to  J^T J (21 values), J^T r (6), r^2 and the point count: 29 double sums, reduced per workgroup in a fixed order (wave shuffles,
// then the waves in order) -> partial[block][29]; the host adds the blocks in order and solves the 6 x 6 system.
struct IcpPose { double R[9]; double t[3]; };
constexpr int kIcpSums = 29;
__global__ __launch_bounds__(256) void k_icp_accumulate(const float* __restrict__ xyz, const int* __restrict__ lab, int n,
                                                        const float* __restrict__ planes, int n_planes, IcpPose T, double* __restrict__ partial) {
  __shared__ double red[4][kIcpSums];
  double a[kIcpSums];
#pragma unroll
  for (int k = 0; k < kIcpSums; ++k) a[k] = 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int k = lab[i];
    if (k < 0 || k >= n_planes) continue;
    const double px = xyz[3 * (size_t)i], py = xyz[3 * (size_t)i + 1], pz = xyz[3 * (size_t)i + 2];
    if (!(isfinite(px) && isfinite(py) && isfinite(pz))) continue;
    const double qx = T.R[0] * px + T.R[1] * py + T.R[2] * pz + T.t[0];
    const double qy = T.R[3] * px + T.R[4] * py + T.R[5] * pz + T.t[1];
    const double qz = T.R[6] * px + T.R[7] * py + T.R[8] * pz + T.t[2];
    const double nx = planes[4 * k], ny = planes[4 * k + 1], nz = planes[4 * k + 2], d = planes[4 * k + 3];
    const double r = nx * qx + ny * qy + nz * qz + d;
    const double J[6] = {qy * nz - qz * ny, qz * nx - qx * nz, qx * ny - qy * nx, nx, ny, nz};
    int m = 0;
#pragma unroll
    for (int rr = 0; rr < 6; ++rr)
#pragma unroll
      for (int cc = rr; cc < 6; ++cc) a[m++] += J[rr] * J[cc];
#pragma unroll
    for (int rr = 0; rr < 6; ++rr) a[21 + rr] += J[rr] * r;
    a[27] += r * r;
    a[28] += 1.0;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < kIcpSums; ++k) {
    double v = a[k];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < kIcpSums) partial[(size_t)blockIdx.x * kIcpSums + threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}


// ---- RANSAC plane per detection box + point-to-plane ICP per frame over the RESIDENT batch (BASELINE.json configs[3]; north_star "RANSAC
// plane fit + point-to-plane ICP over depth clouds inside detection boxes ... one-thread-per-point HIP kernels with LDS inlier counting";
// the only RANSAC hook upstream is compute2DConvexHull, plane_segmentation.cpp:631-665) -------------------------------------------------
// One workgroup per box.  The box's cropped cloud (w x h points in crop order, invalid points included: they are never inliers and a
// sample that hits one is a bad sample) is staged in LDS once -- 12 bytes per point read from HBM once -- and every hypothesis is scored
// out of LDS: sixteen hypotheses at a time, one wave each (lane 0 draws the sample with the counter hash of the single-cloud entry
// point, the wave counts inliers with a point per lane), then one thread replays pcl::RandomSampleConsensus' adaptive loop over the
// sixteen counts in order and stops the box as soon as the sequential algorithm would have stopped.  optimizeModelCoefficients needs
// float sums over the inliers IN INDEX ORDER: the points are walked 1024 at a time, every inlier gets its rank from wave ballots and
// parks its nine products in LDS, nine lanes add them in rank order -- the arithmetic of oracle_seg.c's serial loop, bit for bit.
struct RansacBox { float coeff[4]; int inliers, best_count, best_iter, hyps; };
// Round 6: two kernels.  (PMC, profiles/r6_pmc_frontend.json: the one-kernel form spent 85 % of its wave cycles parked -- nine lanes added
// the inliers' products one LDS round trip at a time while fifteen waves and a 147 KB crop waited, one box per CU.)
//   k_ransac_hyp      one 1024-thread workgroup per box, the crop staged in LDS once.  A round draws up to sixteen samples (one wave each, lane
//                     0, the counter hash of the single-cloud entry point), then ONE pass over the points scores all of them: a thread keeps the
//                     round's models in registers and tests each of its points against every model -- three LDS reads per point instead of
//                     three per point and hypothesis.  Thread 0 replays pcl::RandomSampleConsensus' adaptive loop over the counts in order and
//                     stops where the sequential algorithm stops (rounds after the first draw eight samples: the scene's boxes stop at 17).
//   k_ransac_refine   one 256-thread workgroup per box, eight per CU: optimizeModelCoefficients' float sums over the inliers IN INDEX ORDER
//                     (ranks from wave ballots, nine products parked in LDS, nine lanes add them in rank order, sixteen loads in flight ahead of
//                     the dependent adds) -- the arithmetic of oracle_seg.c's serial loop, bit for bit --, eigen33, then the inlier flags of the
//                     refined model.  The chains of eight boxes overlap on a CU; the points come through L2.
constexpr int kRsProd = 256;   // inlier products parked per round of the ordered sums
constexpr int kRsHyp = 16;     // hypotheses of the first round of a box (later rounds: kRsHyp / 2)
__global__ __launch_bounds__(1024) void k_ransac_hyp(View V, float thr, int max_iterations, double probability, unsigned long long seed0,
                                                     int lds_points, RansacBox* __restrict__ out) {
  extern __shared__ float sp[];
  __shared__ float s_models[kRsHyp][4];
  __shared__ int s_cnt[kRsHyp], s_ok[kRsHyp];
  __shared__ float s_best[4];
  __shared__ int s_stop, s_best_n, s_best_it, s_it;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const BoxMeta b = V.box[blockIdx.x];
  const int n = b.w * b.h;
  const float* gp = V.pts + (size_t)b.pix0 * 3;
  const bool staged = n <= lds_points;
  if (staged) {
    if ((((size_t)b.pix0 * 3) & 3) == 0) {   // 16-byte pieces where the crop starts on one
      const float4* g4 = reinterpret_cast<const float4*>(gp);
      float4* s4 = reinterpret_cast<float4*>(sp);
      const int n4 = (3 * n) >> 2;
      for (int e = tid; e < n4; e += 1024) s4[e] = g4[e];
      for (int e = 4 * n4 + tid; e < 3 * n; e += 1024) sp[e] = gp[e];
    } else {
      for (int e = tid; e < 3 * n; e += 1024) sp[e] = gp[e];
    }
  }
  const float* P = staged ? (const float*)sp : gp;
  const unsigned long long seed = seed0 + (unsigned long long)blockIdx.x * 0x9E3779B97F4A7C15ull;
  if (tid == 0) { s_stop = n < 3 ? 1 : 0; s_best_n = -1; s_best_it = -1; s_it = 0; s_best[0] = s_best[1] = s_best[2] = s_best[3] = 0; }
  __syncthreads();
  // the adaptive loop's state lives in thread 0
  double k = 1.0;
  int iterations = 0, skipped = 0;
  const int max_skip = max_iterations * 10;
  const double log_probability = log(1.0 - probability), one_over = 1.0 / (double)n, eps = 2.220446049250313e-16;
  int nh = kRsHyp;
  while (!s_stop) {
    if (wave < nh && lane == 0) {
      const int iter = s_it + wave;
      float m0 = 0, m1 = 0, m2 = 0, m3 = 0;
      int good = 0;
      for (int attempt = 0; attempt < 1000 && !good; ++attempt) {
        int id[3];
        for (int j = 0; j < 3; ++j)
          id[j] = (int)(splitmix64(seed ^ ((unsigned long long)iter << 32) ^ ((unsigned long long)attempt << 8) ^ (unsigned long long)j) % (unsigned long long)n);
        if (id[0] == id[1] || id[0] == id[2] || id[1] == id[2]) continue;
        const float* p0 = P + (size_t)id[0] * 3; const float* p1 = P + (size_t)id[1] * 3; const float* p2 = P + (size_t)id[2] * 3;
        const float a0 = p1[0] - p0[0], a1 = p1[1] - p0[1], a2 = p1[2] - p0[2];
        const float b0 = p2[0] - p0[0], b1 = p2[1] - p0[1], b2 = p2[2] - p0[2];
        const float r0 = a0 / b0, r1 = a1 / b1, r2 = a2 / b2;
        if (!((r0 != r1) || (r2 != r1))) continue;
        m0 = a1 * b2 - a2 * b1; m1 = a2 * b0 - a0 * b2; m2 = a0 * b1 - a1 * b0;
        const float nn = sqrtf(m0 * m0 + m1 * m1 + m2 * m2);
        m0 /= nn; m1 /= nn; m2 /= nn;
        m3 = -1 * (m0 * p0[0] + m1 * p0[1] + m2 * p0[2]);
        good = 2;
      }
      s_ok[wave] = (good == 2) && isfinite(m0) && isfinite(m3);
      s_models[wave][0] = m0; s_models[wave][1] = m1; s_models[wave][2] = m2; s_models[wave][3] = m3;
      s_cnt[wave] = 0;
    }
    __syncthreads();
    {   // one pass over the points for all hypotheses of the round
      float m[kRsHyp][4];
      int cnt[kRsHyp];
#pragma unroll
      for (int h = 0; h < kRsHyp; ++h) {
        const int hh = h < nh ? h : 0;
        m[h][0] = s_models[hh][0]; m[h][1] = s_models[hh][1]; m[h][2] = s_models[hh][2]; m[h][3] = s_models[hh][3];
        cnt[h] = 0;
      }
      if (nh == kRsHyp) {
        for (int i = tid; i < n; i += 1024) {
          const float p[3] = {P[(size_t)i * 3], P[(size_t)i * 3 + 1], P[(size_t)i * 3 + 2]};
#pragma unroll
          for (int h = 0; h < kRsHyp; ++h) cnt[h] += plane_inlier(m[h], p, thr) ? 1 : 0;
        }
      } else {
        for (int i = tid; i < n; i += 1024) {
          const float p[3] = {P[(size_t)i * 3], P[(size_t)i * 3 + 1], P[(size_t)i * 3 + 2]};
#pragma unroll
          for (int h = 0; h < kRsHyp / 2; ++h) cnt[h] += plane_inlier(m[h], p, thr) ? 1 : 0;
        }
      }
#pragma unroll
      for (int h = 0; h < kRsHyp; ++h) {
        if (h < nh) {
          int c = cnt[h];
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
          if (lane == 0) atomicAdd(&s_cnt[h], c);     // integer counts: the order of the waves does not matter
        }
      }
    }
    __syncthreads();
    if (tid == 0) {   // pcl::RandomSampleConsensus::computeModel over these hypotheses, in order
      int stop = 0;
      for (int j = 0; j < nh; ++j) {
        if (!((double)iterations < k && skipped < max_skip)) { stop = 1; break; }
        const int c = s_ok[j] ? s_cnt[j] : -1;
        ++s_it;
        if (c < 0) { ++skipped; continue; }
        if (c > s_best_n) {
          s_best_n = c; s_best_it = s_it - 1;
          s_best[0] = s_models[j][0]; s_best[1] = s_models[j][1]; s_best[2] = s_models[j][2]; s_best[3] = s_models[j][3];
          const double w = (double)c * one_over;
          double p_no = 1.0 - pow(w, 3.0);
          if (p_no < eps) p_no = eps;
          if (p_no > 1.0 - eps) p_no = 1.0 - eps;
          k = log_probability / log(p_no);
        }
        ++iterations;
        if (iterations > max_iterations) { stop = 1; break; }
      }
      s_stop = stop;
    }
    nh = kRsHyp / 2;
    __syncthreads();
  }
  if (tid == 0) out[blockIdx.x] = RansacBox{{s_best[0], s_best[1], s_best[2], s_best[3]}, 0, s_best_n, s_best_it, s_it};
}

__global__ __launch_bounds__(256) void k_ransac_refine(View V, float thr, RansacBox* __restrict__ out, unsigned char* __restrict__ flag) {
  __shared__ float prod[kRsProd * 9];
  __shared__ int s_woff[5];
  __shared__ float s_best[4], s_acc[9];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const BoxMeta b = V.box[blockIdx.x];
  const int n = b.w * b.h;
  const float* P = V.pts + (size_t)b.pix0 * 3;
  const RansacBox rb = out[blockIdx.x];
  const int best = rb.best_count;
  float model[4] = {rb.coeff[0], rb.coeff[1], rb.coeff[2], rb.coeff[3]};
  if (best <= 0) {
    for (int i = tid; i < n; i += 256) flag[(size_t)b.pix0 + i] = 0;
    if (tid == 0) out[blockIdx.x] = RansacBox{{0, 0, 0, 0}, 0, best, rb.best_iter, rb.hyps};
    return;
  }
  // ---- optimizeModelCoefficients: float mean / covariance of the inliers of the sampled model, summed in index order
  if (best > 3) {
    float acc = 0;
    int total = 0;
    for (int c0 = 0; c0 < n; c0 += kRsProd) {
      const int i = c0 + tid;
      float p[3] = {0, 0, 0};
      if (i < n) { p[0] = P[(size_t)i * 3]; p[1] = P[(size_t)i * 3 + 1]; p[2] = P[(size_t)i * 3 + 2]; }
      const bool in = i < n && plane_inlier(model, p, thr);
      const unsigned long long mask = __ballot(in);
      if (lane == 0) s_woff[wave] = __popcll(mask);
      __syncthreads();
      const int w0 = s_woff[0], w1 = s_woff[1], w2 = s_woff[2], w3 = s_woff[3];
      const int cn = w0 + w1 + w2 + w3;
      const int rank = (wave > 0 ? w0 : 0) + (wave > 1 ? w1 : 0) + (wave > 2 ? w2 : 0) + __popcll(mask & ((1ull << lane) - 1ull));
      if (in) {
        float* q = prod + rank * 9;
        q[0] = p[0] * p[0]; q[1] = p[0] * p[1]; q[2] = p[0] * p[2]; q[3] = p[1] * p[1]; q[4] = p[1] * p[2]; q[5] = p[2] * p[2];
        q[6] = p[0]; q[7] = p[1]; q[8] = p[2];
      }
      __syncthreads();
      if (tid < 9) {   // sixteen loads ahead of sixteen dependent adds: the chain is the adds, not LDS round trips
        int q = 0;
        for (; q + 16 <= cn; q += 16) {
          float v[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) v[u] = prod[(q + u) * 9 + tid];
#pragma unroll
          for (int u = 0; u < 16; ++u) acc += v[u];
        }
        for (; q < cn; ++q) acc += prod[q * 9 + tid];
      }
      __syncthreads();
      total += cn;
    }
    if (tid < 9) s_acc[tid] = acc;
    __syncthreads();
    if (tid == 0) {
      float a[9];
      const float cf = (float)total;
#pragma unroll
      for (int q = 0; q < 9; ++q) a[q] = s_acc[q] / cf;
      float cov[9];
      cov[0] = a[0] - a[6] * a[6]; cov[1] = a[1] - a[6] * a[7]; cov[2] = a[2] - a[6] * a[8];
      cov[4] = a[3] - a[7] * a[7]; cov[5] = a[4] - a[7] * a[8]; cov[8] = a[5] - a[8] * a[8];
      cov[3] = cov[1]; cov[6] = cov[2]; cov[7] = cov[5];
      float ev, v[3];
      eigen33(cov, ev, v);
      s_best[0] = v[0]; s_best[1] = v[1]; s_best[2] = v[2];
      s_best[3] = -1 * (v[0] * a[6] + v[1] * a[7] + v[2] * a[8]);
    }
    __syncthreads();
    model[0] = s_best[0]; model[1] = s_best[1]; model[2] = s_best[2]; model[3] = s_best[3];
  }
  // ---- inliers of the refined model: flags for the box's pixels + their count
  int cnt = 0;
  for (int i = tid; i < n; i += 256) {
    const bool in = plane_inlier(model, P + (size_t)i * 3, thr);
    flag[(size_t)b.pix0 + i] = in ? 1 : 0;
    cnt += in ? 1 : 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
  __syncthreads();
  if (lane == 0) s_woff[wave] = cnt;
  __syncthreads();
  if (tid == 0) out[blockIdx.x] = RansacBox{{model[0], model[1], model[2], model[3]}, s_woff[0] + s_woff[1] + s_woff[2] + s_woff[3], best, rb.best_iter, rb.hyps};
}

// Point-to-plane ICP of every frame of the resident batch against a plane list (e.g. the previous keyframe's planes): the points are the
// RANSAC inliers of the frame's boxes, box slot q measures plane box_plane[q] (-1: the box takes no part).
// Round 4 ran ONE 1024-thread workgroup per frame through all Gauss-Newton rounds: 32 workgroups on 256 CUs, 0.234 ms per frame -- 2.7 x
// the whole segmentation.  Round 5: a Gauss-Newton round is two launches over the whole batch --
//   k_icp_box_sums    one workgroup per BOX: its inliers' 29 double sums under the frame's current transform (fixed-order reduction:
//                     a thread's points in index order, wave shuffles, the waves in order) -> part[box][29]
//   k_icp_frame_step  one wave per FRAME: the boxes' partial sums in slot order, the 6 x 6 Cholesky solve and T <- (exp[w]x, u) o T by
//                     lane 0; a frame that is finished (last pass, < 6 points, rank-deficient plane set) is frozen by its flag
// -- 2 (iterations + 1) launches of a few microseconds for ALL frames, still no host round trip per iteration.  Same arithmetic per point
// as k_icp_accumulate + the host solve of sslam_seg_icp_point_to_plane; only the order of the sums differs.
struct IcpFrame { double T[12]; double rms; int used, status; };
struct IcpState { double T[12]; int status, done, pad0, pad1; };
__global__ __launch_bounds__(256) void k_icp_box_sums(View V, const unsigned char* __restrict__ flag, const int* __restrict__ box_frame,
                                                      const int* __restrict__ box_plane, const float* __restrict__ planes, int n_planes,
                                                      const IcpState* __restrict__ st, double* __restrict__ part) {
  __shared__ double red[4][kIcpSums];
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int f = box_frame[q], kpl = box_plane[q];
  double a[kIcpSums];
#pragma unroll
  for (int k = 0; k < kIcpSums; ++k) a[k] = 0.0;
  if (!st[f].done && kpl >= 0 && kpl < n_planes) {
    double R[9], t[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = st[f].T[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) t[k] = st[f].T[9 + k];
    const BoxMeta b = V.box[q];
    const double nx = planes[4 * kpl], ny = planes[4 * kpl + 1], nz = planes[4 * kpl + 2], d = planes[4 * kpl + 3];
    const int n = b.w * b.h;
    for (int i = tid; i < n; i += 256) {
      if (!flag[(size_t)b.pix0 + i]) continue;
      const float* p = V.pts + ((size_t)b.pix0 + i) * 3;
      const double px = p[0], py = p[1], pz = p[2];
      if (!(isfinite(px) && isfinite(py) && isfinite(pz))) continue;
      const double qx = R[0] * px + R[1] * py + R[2] * pz + t[0];
      const double qy = R[3] * px + R[4] * py + R[5] * pz + t[1];
      const double qz = R[6] * px + R[7] * py + R[8] * pz + t[2];
      const double r = nx * qx + ny * qy + nz * qz + d;
      const double J[6] = {qy * nz - qz * ny, qz * nx - qx * nz, qx * ny - qy * nx, nx, ny, nz};
      int m = 0;
#pragma unroll
      for (int rr = 0; rr < 6; ++rr)
#pragma unroll
        for (int cc = rr; cc < 6; ++cc) a[m++] += J[rr] * J[cc];
#pragma unroll
      for (int rr = 0; rr < 6; ++rr) a[21 + rr] += J[rr] * r;
      a[27] += r * r;
      a[28] += 1.0;
    }
  }
#pragma unroll
  for (int k = 0; k < kIcpSums; ++k) {
    double v = a[k];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (tid < kIcpSums) part[(size_t)q * kIcpSums + tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
}
__global__ __launch_bounds__(64) void k_icp_frame_step(const int* __restrict__ frame_box0, const double* __restrict__ part, int it, int iterations,
                                                       IcpState* __restrict__ st, IcpFrame* __restrict__ out) {
  __shared__ double sums[kIcpSums];
  const int f = blockIdx.x, tid = threadIdx.x;
  if (st[f].done) return;
  if (tid < kIcpSums) {
    double v = 0;
    for (int q = frame_box0[f]; q < frame_box0[f + 1]; ++q) v += part[(size_t)q * kIcpSums + tid];   // slot order
    sums[tid] = v;
  }
  __syncthreads();
  if (tid != 0) return;
  IcpState S = st[f];
  if (it == iterations || sums[28] < 6 || S.status) {   // the last pass only measures the residual at the returned transform
    IcpFrame o;
    for (int k = 0; k < 12; ++k) o.T[k] = S.T[k];
    o.rms = sums[28] > 0 ? sqrt(sums[27] / sums[28]) : 0.0;
    o.used = (int)sums[28]; o.status = S.status;
    out[f] = o;
    st[f].done = 1;
    return;
  }
  double A[36], bvec[6], L[36];
  for (int k = 0; k < 36; ++k) L[k] = 0;
  int m = 0;
  for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c) { A[r * 6 + c] = A[c * 6 + r] = sums[m++]; }
  for (int r = 0; r < 6; ++r) bvec[r] = -sums[21 + r];
  bool okc = true;
  for (int j = 0; j < 6 && okc; ++j) {
    double dsum = A[j * 6 + j];
    for (int k = 0; k < j; ++k) dsum -= L[j * 6 + k] * L[j * 6 + k];
    if (!(dsum > 1e-12 * A[j * 6 + j]) || !(dsum > 0)) { okc = false; break; }
    L[j * 6 + j] = sqrt(dsum);
    for (int i = j + 1; i < 6; ++i) { double v = A[i * 6 + j]; for (int k = 0; k < j; ++k) v -= L[i * 6 + k] * L[j * 6 + k]; L[i * 6 + j] = v / L[j * 6 + j]; }
  }
  if (!okc) { st[f].status = SSLAM_ERR_NUMERIC; return; }   // the planes leave a degree of freedom unconstrained: the next pass reports it
  double y[6], dx[6];
  for (int i = 0; i < 6; ++i) { double v = bvec[i]; for (int k = 0; k < i; ++k) v -= L[i * 6 + k] * y[k]; y[i] = v / L[i * 6 + i]; }
  for (int i = 5; i >= 0; --i) { double v = y[i]; for (int k = i + 1; k < 6; ++k) v -= L[k * 6 + i] * dx[k]; dx[i] = v / L[i * 6 + i]; }
  const double wx = dx[0], wy = dx[1], wz = dx[2], th = sqrt(wx * wx + wy * wy + wz * wz);
  double E[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (th > 0) {
    const double sa = sin(th) / th, bq = (1.0 - cos(th)) / (th * th);
    const double K[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { double kk = 0; for (int q = 0; q < 3; ++q) kk += K[r * 3 + q] * K[q * 3 + c]; E[r * 3 + c] += sa * K[r * 3 + c] + bq * kk; }
  }
  double Tn[12];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) Tn[r * 3 + c] = E[r * 3] * S.T[c] + E[r * 3 + 1] * S.T[3 + c] + E[r * 3 + 2] * S.T[6 + c];
    Tn[9 + r] = E[r * 3] * S.T[9] + E[r * 3 + 1] * S.T[10] + E[r * 3 + 2] * S.T[11] + dx[3 + r];
  }
  for (int k = 0; k < 12; ++k) st[f].T[k] = Tn[k];
}

// ---- cloud filters of the legacy path (plane_segmentation.cpp:557-629) ---------------------------------------------
// ordered compaction of a flag array (ascending indices): per-block counts -> k_ransac_scan -> write
__global__ __launch_bounds__(256) void k_flag_count(const unsigned char* __restrict__ flag, int n, int* __restrict__ block_counts) {
  __shared__ int wsum[4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  int c = (i < n && flag[i]) ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
__global__ __launch_bounds__(256) void k_flag_write(const unsigned char* __restrict__ flag, int n, const int* __restrict__ block_off, int* __restrict__ out, int max_out) {
  __shared__ int woff[4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool in = i < n && flag[i];
  const unsigned long long mask = __ballot(in);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) woff[wave] = __popcll(mask);
  __syncthreads();
  int base = block_off[blockIdx.x];
  for (int w = 0; w < wave; ++w) base += woff[w];
  if (in) {
    const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
    if (pos < max_out) out[pos] = i;
  }
}
// distance_filter (plane_segmentation.cpp:607-629): keep p with 0.3 < |p| < 3 (float norm, compared as double)
__global__ __launch_bounds__(256) void k_range_flag(const float* __restrict__ pts, int n, double dmin, double dmax, unsigned char* __restrict__ flag) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
  const double d = (double)sqrtf((x * x + y * y) + z * z);
  flag[i] = (d > dmin && d < dmax) ? 1 : 0;
}
// pcl::VoxelGrid (leaf 0.1, plane_segmentation.cpp:565-581).  Bounding box of the finite points: per-block float min / max
__global__ __launch_bounds__(256) void k_voxel_minmax(const float* __restrict__ pts, int n, float* __restrict__ part) {
  __shared__ float red[4][6];
  float lo[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, hi[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
    if (!(isfinite(x) && isfinite(y) && isfinite(z))) continue;
    lo[0] = fminf(lo[0], x); lo[1] = fminf(lo[1], y); lo[2] = fminf(lo[2], z);
    hi[0] = fmaxf(hi[0], x); hi[1] = fmaxf(hi[1], y); hi[2] = fmaxf(hi[2], z);
  }
#pragma unroll
  for (int k = 0; k < 3; ++k)
    for (int o = 32; o > 0; o >>= 1) { lo[k] = fminf(lo[k], __shfl_down(lo[k], o, 64)); hi[k] = fmaxf(hi[k], __shfl_down(hi[k], o, 64)); }
  if ((threadIdx.x & 63) == 0) for (int k = 0; k < 3; ++k) { red[threadIdx.x >> 6][k] = lo[k]; red[threadIdx.x >> 6][3 + k] = hi[k]; }
  __syncthreads();
  if (threadIdx.x < 3) part[blockIdx.x * 6 + threadIdx.x] = fminf(fminf(red[0][threadIdx.x], red[1][threadIdx.x]), fminf(red[2][threadIdx.x], red[3][threadIdx.x]));
  else if (threadIdx.x < 6) part[blockIdx.x * 6 + threadIdx.x] = fmaxf(fmaxf(red[0][threadIdx.x], red[1][threadIdx.x]), fmaxf(red[2][threadIdx.x], red[3][threadIdx.x]));
}
struct VoxelGeom { float inv[3]; int minb[3]; int mul[3]; };
// cell of every finite point; per cell: point count and the coordinate sums in 2^-20 fixed point (integer atomics: exact and
// independent of the order in which the points arrive -- PCL's own accumulation order is that of an unstable sort)
__global__ __launch_bounds__(256) void k_voxel_accumulate(const float* __restrict__ pts, int n, VoxelGeom G, int* __restrict__ count, long long* __restrict__ sums) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
  if (!(isfinite(x) && isfinite(y) && isfinite(z))) return;
  const int ix = (int)floorf(x * G.inv[0]) - G.minb[0], iy = (int)floorf(y * G.inv[1]) - G.minb[1], iz = (int)floorf(z * G.inv[2]) - G.minb[2];
  const size_t c = (size_t)ix * G.mul[0] + (size_t)iy * G.mul[1] + (size_t)iz * G.mul[2];
  atomicAdd(&count[c], 1);
  atomicAdd((unsigned long long*)&sums[3 * c], (unsigned long long)llrint((double)x * 1048576.0));
  atomicAdd((unsigned long long*)&sums[3 * c + 1], (unsigned long long)llrint((double)y * 1048576.0));
  atomicAdd((unsigned long long*)&sums[3 * c + 2], (unsigned long long)llrint((double)z * 1048576.0));
}
__global__ __launch_bounds__(256) void k_voxel_flag(const int* __restrict__ count, int ncell, unsigned char* __restrict__ flag) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < ncell) flag[c] = count[c] > 0;
}
__global__ __launch_bounds__(256) void k_voxel_centroids(const int* __restrict__ cells, int nocc, const int* __restrict__ count, const long long* __restrict__ sums,
                                                        float* __restrict__ out, int* __restrict__ out_count) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= nocc) return;
  const int c = cells[k], m = count[c];
#pragma unroll
  for (int d = 0; d < 3; ++d) out[3 * (size_t)k + d] = (float)(((double)sums[3 * (size_t)c + d] / 1048576.0) / (double)m);
  if (out_count) out_count[k] = m;
}
// pcl::StatisticalOutlierRemoval (meanK 50, plane_segmentation.cpp:583-605): mean distance of every point to its k nearest
// neighbours, exact (brute force): candidates stream through LDS tiles, every thread keeps the k + 1 smallest squared distances of
// its query in an LDS column (replace-the-maximum; after the first tiles almost every candidate is rejected by one compare)
constexpr int kSorMaxK = 64;
__global__ __launch_bounds__(128) void k_sor_mean_distance(const float* __restrict__ pts, int n, int k, float* __restrict__ mean_dist) {
  extern __shared__ float sm[];
  float* tile = sm;                         // [128][3]
  float* best = sm + 128 * 3;               // [k + 1][128]
  const int i = blockIdx.x * 128 + threadIdx.x, tid = threadIdx.x;
  const bool valid_i = i < n;
  float qx = 0, qy = 0, qz = 0;
  bool fin = false;
  if (valid_i) { qx = pts[3 * (size_t)i]; qy = pts[3 * (size_t)i + 1]; qz = pts[3 * (size_t)i + 2]; fin = isfinite(qx) && isfinite(qy) && isfinite(qz); }
  const int K = k + 1;
  for (int j = 0; j < K; ++j) best[j * 128 + tid] = 3.402823466e+38f;
  float cur_max = 3.402823466e+38f;
  int arg_max = 0;
  for (int t0 = 0; t0 < n; t0 += 128) {
    __syncthreads();
    const int c = t0 + tid;
    tile[tid * 3] = c < n ? pts[3 * (size_t)c] : __int_as_float(0x7fc00000);
    tile[tid * 3 + 1] = c < n ? pts[3 * (size_t)c + 1] : __int_as_float(0x7fc00000);
    tile[tid * 3 + 2] = c < n ? pts[3 * (size_t)c + 2] : __int_as_float(0x7fc00000);
    __syncthreads();
    if (!fin) continue;
    const int m = min(128, n - t0);
    for (int j = 0; j < m; ++j) {
      const float dx = qx - tile[j * 3], dy = qy - tile[j * 3 + 1], dz = qz - tile[j * 3 + 2];
      const float d = (dx * dx + dy * dy) + dz * dz;      // flann::L2_Simple<float>
      if (d < cur_max) {                                   // NaN candidates fail the compare
        best[arg_max * 128 + tid] = d;
        cur_max = best[tid]; arg_max = 0;
        for (int q = 1; q < K; ++q) { const float v = best[q * 128 + tid]; if (v > cur_max) { cur_max = v; arg_max = q; } }
      }
    }
  }
  if (!valid_i) return;
  if (!fin) { mean_dist[i] = -1.0f; return; }              // not a finite point: takes no part (marked for the host)
  // ascending insertion sort of the k + 1 values, then the sum of the square roots of entries 1..k in that order (entry 0 is the point itself)
  for (int a = 1; a < K; ++a) {
    const float v = best[a * 128 + tid];
    int b = a - 1;
    while (b >= 0 && best[b * 128 + tid] > v) { best[(b + 1) * 128 + tid] = best[b * 128 + tid]; --b; }
    best[(b + 1) * 128 + tid] = v;
  }
  double sum = 0;
  for (int a = 1; a < K; ++a) sum += (double)sqrtf(best[a * 128 + tid]);
  mean_dist[i] = (float)(sum / (double)k);
}

// ---- k-means of the legacy path (plane_segmentation::computeKmeans -> cv::kmeans, plane_segmentation.cpp:524-535) -----------------
// assignment step: nearest centre in float32 (squared distance accumulated coordinate by coordinate; ties -> lowest centre), and for
// the next centre update the coordinate sums / counts per cluster and the compactness, all in 2^-20 fixed point (integer atomics)
constexpr int kKmMaxK = 16;
struct KmCenters { float c[kKmMaxK * 3]; int k, dim; };
__global__ __launch_bounds__(256) void k_kmeans_assign(const float* __restrict__ pts, int n, KmCenters C, int* __restrict__ labels,
                                                      long long* __restrict__ sums, int* __restrict__ counts, long long* __restrict__ compact) {
  __shared__ long long s_sum[kKmMaxK * 3];
  __shared__ int s_cnt[kKmMaxK];
  __shared__ long long s_cmp;
  for (int t = threadIdx.x; t < kKmMaxK * 3; t += 256) s_sum[t] = 0;
  if (threadIdx.x < kKmMaxK) s_cnt[threadIdx.x] = 0;
  if (threadIdx.x == 0) s_cmp = 0;
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    float p[3] = {0, 0, 0};
    for (int d = 0; d < C.dim; ++d) p[d] = pts[(size_t)i * C.dim + d];
    float best = 3.402823466e+38f;
    int arg = 0;
    for (int j = 0; j < C.k; ++j) {
      float dist = 0;
      for (int d = 0; d < C.dim; ++d) { const float t = p[d] - C.c[j * 3 + d]; dist += t * t; }
      if (dist < best) { best = dist; arg = j; }
    }
    labels[i] = arg;
    for (int d = 0; d < C.dim; ++d) atomicAdd((unsigned long long*)&s_sum[arg * 3 + d], (unsigned long long)llrint((double)p[d] * 1048576.0));
    atomicAdd(&s_cnt[arg], 1);
    atomicAdd((unsigned long long*)&s_cmp, (unsigned long long)llrint((double)best * 1048576.0));
  }
  __syncthreads();
  for (int t = threadIdx.x; t < C.k * 3; t += 256) if (s_sum[t]) atomicAdd((unsigned long long*)&sums[t], (unsigned long long)s_sum[t]);
  if (threadIdx.x < C.k && s_cnt[threadIdx.x]) atomicAdd(&counts[threadIdx.x], s_cnt[threadIdx.x]);
  if (threadIdx.x == 0 && s_cmp) atomicAdd((unsigned long long*)compact, (unsigned long long)s_cmp);
}
}  // namespace seg
}  // namespace sslam

// =================================================================================================
// host side + C-ABI
// =================================================================================================
using namespace sslam;
using namespace sslam::seg;

struct sslam_seg {
  sslam_seg_params P;
  hipStream_t stream = nullptr;
  View V{};
  std::vector<BoxMeta> boxes;      // accepted boxes of the last call
  std::vector<int> box_src;        // accepted slot -> index into the caller's box array (of its frame)
  std::vector<int> box_frame;      // accepted slot -> frame of the batched call
  int last_dropped_planes = 0, last_candidate_overflow = 0, last_region_overflow = 0;
  size_t cap_pix = 0, cap_ii = 0, cap_cloud = 0, cap_box = 0;
  unsigned char* d_cloud = nullptr;
  BoxMeta* d_box = nullptr;
  double last_kernel_ms = 0, last_total_ms = 0;
  std::vector<void*> allocs;
  // a batch between its enqueue (H2D + kernels + result tables D2H, all asynchronous on `stream`) and its finish (wait + scalar
  // post-processing): what the post-processing needs from the caller's frames, copied so that only the clouds must stay alive
  struct FrameMeta { float robot_pose[6]; float cam_angle; };
  struct BoxInfo { float prob; int class_id; };
  std::vector<FrameMeta> q_frames;
  std::vector<BoxInfo> q_boxes;    // per accepted slot
  Region* q_regs = nullptr;        // pinned: the result tables come back asynchronously (a pageable target would make the enqueue wait)
  int* q_nreg = nullptr;           // [cap_q_box + 2]: region counts, then the two overflow counters
  size_t cap_q_box = 0;
  hipEvent_t q_e0 = nullptr, q_e1 = nullptr;
  bool q_e1_armed = false;         // q_e1 has been recorded behind the kernels of a batch (the peer pipeline's next batch waits for it)
  bool q_busy = false;
  std::chrono::steady_clock::time_point q_t0;
  // sslam_seg_submit_batch / _collect_batch: two pipelines (this handle and a twin with its own stream and buffers) used in turn, so
  // that the H2D copy of one batch runs under the kernels of the previous one
  // sslam_seg_ransac_boxes / _icp_boxes over the resident batch of the last blocking segment call
  unsigned char* d_rflag = nullptr;   // [pixels of the batch] inlier of its box's refined RANSAC model
  void* d_rbox = nullptr;             // RansacBox per accepted box
  void* d_icp = nullptr;              // sslam_seg_icp_boxes: [tables | per-frame state | per-box partial sums | results], grows only
  size_t cap_icp = 0;
  std::vector<char> h_icp;            // host image of the tables + state (one H2D copy per call)
  size_t cap_rflag = 0, cap_rbox = 0;
  int r_nbox = -1;                    // boxes the flags belong to (-1: no RANSAC has run on the resident batch)
  sslam_seg* twin = nullptr;
  int fifo[2] = {0, 0};            // which pipeline (0 = this, 1 = twin) holds the oldest / the newer submitted batch
  int n_inflight = 0;
  ~sslam_seg() {
    delete twin;
    if (stream) { (void)hipSetDevice(P.device); (void)hipStreamSynchronize(stream); }
    free_all();
    if (q_regs) (void)hipHostFree(q_regs);
    if (q_nreg) (void)hipHostFree(q_nreg);
    if (d_rflag) (void)hipFree(d_rflag);
    if (d_rbox) (void)hipFree(d_rbox);
    if (d_icp) (void)hipFree(d_icp);
    if (q_e0) (void)hipEventDestroy(q_e0);
    if (q_e1) (void)hipEventDestroy(q_e1);
    if (stream) (void)hipStreamDestroy(stream);
  }
  void free_all() {
    for (void* p : allocs) (void)hipFree(p);
    allocs.clear();
  }
};

static void host_mat4_mul(const float* A, const float* B, float* C) {
  float T[16];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      float s = 0;
      for (int k = 0; k < 4; ++k) s += A[r * 4 + k] * B[k * 4 + c];
      T[r * 4 + c] = s;
    }
  memcpy(C, T, sizeof T);
}

template <typename T>
static int seg_alloc(sslam_seg* s, size_t n, T** out) {
  void* p = nullptr;
  SSLAM_HIP_TRY(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)));
  s->allocs.push_back(p);
  *out = (T*)p;
  return 0;
}

// helpers of the cloud filters (row f4)
namespace {
struct DevGuard {
  std::vector<void*> ptrs;
  ~DevGuard() { for (void* p : ptrs) if (p) (void)hipFree(p); }
  template <class T> int alloc(T** p, size_t count) {
    void* q = nullptr;
    SSLAM_HIP_TRY(hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T)));
    ptrs.push_back(q); *p = (T*)q;
    return 0;
  }
};
int seg_device(sslam_seg* s) {
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) return set_error(SSLAM_ERR_NO_DEVICE, "no HIP device visible; the product has no CPU fallback");
  SSLAM_HIP_TRY(hipSetDevice(s->P.device));
  if (!s->stream) SSLAM_HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  return 0;
}
// ascending indices of the set flags -> out (host); returns their number
int compact_flags(sslam_seg* s, DevGuard& g, const unsigned char* d_flag, int n, int* d_out, int max_out, int* total_out) {
  const int nblk = (n + 255) / 256;
  int *d_blk = nullptr, *d_total = nullptr;
  int rc;
  if ((rc = g.alloc(&d_blk, nblk)) || (rc = g.alloc(&d_total, 1))) return rc;
  hipLaunchKernelGGL(k_flag_count, dim3(nblk), dim3(256), 0, s->stream, d_flag, n, d_blk);
  hipLaunchKernelGGL(k_ransac_scan, dim3(1), dim3(64), 0, s->stream, d_blk, nblk, d_total);
  hipLaunchKernelGGL(k_flag_write, dim3(nblk), dim3(256), 0, s->stream, d_flag, n, d_blk, d_out, max_out);
  SSLAM_HIP_TRY(hipGetLastError());
  SSLAM_HIP_TRY(hipMemcpyAsync(total_out, d_total, sizeof(int), hipMemcpyDeviceToHost, s->stream));
  SSLAM_HIP_TRY(hipStreamSynchronize(s->stream));
  return 0;
}
}  // namespace

extern "C" {

void sslam_seg_default_params(sslam_seg_params* p) {
  if (!p) return;
  p->num_point_seg = 500; p->norm_point_thres = 5000; p->planar_area = 0.1;   // plane_segmentation.cpp:7-9
  p->max_depth_change_factor = 0.03f; p->normal_smoothing_size = 20.0f;        // :99-100
  p->angular_threshold = (float)(0.017453 * 2.0); p->distance_threshold = 0.02f;  // :140-141
  p->maximum_curvature = 0.001f;
  p->min_contour_points = 100;                                                 // :169
  p->image_width = 640; p->image_height = 480;                                 // :34-35
  p->reference_quirks = 1; p->device = 0;
}

sslam_seg* sslam_seg_create(const sslam_seg_params* p) {
  sslam_seg* s = new sslam_seg();
  if (p) s->P = *p; else sslam_seg_default_params(&s->P);
  return s;
}
void sslam_seg_destroy(sslam_seg* s) { delete s; }

// semantic_tools::transformNormalsToWorld (tools.h:18-102), float matrix chain as in the reference
int sslam_seg_transform(const sslam_seg* s, const float pose[6], float cam_pitch, float out[16]) {
  if (!pose || !out) return set_error(SSLAM_ERR_INVALID, "null argument");
  const int quirks = s ? s->P.reference_quirks : 1;
  float rxc[16] = {0}, rxr[16] = {0}, rzr[16] = {0}, T[16] = {0};
  const double roll = pose[3], pitch = pose[4], yaw = pose[5];
  const double a = -(double)cam_pitch;
  rxc[0] = 1; rxc[5] = (float)cos(a); rxc[6] = (float)-sin(a); rxc[9] = (float)sin(a); rxc[10] = (float)cos(a); rxc[15] = 1;
  rxr[0] = 1; rxr[5] = (float)cos(-1.5708); rxr[6] = (float)-sin(-1.5708); rxr[9] = (float)sin(-1.5708); rxr[10] = (float)cos(-1.5708); rxr[15] = 1;
  rzr[0] = (float)cos(-1.5708); rzr[1] = (float)-sin(-1.5708); rzr[4] = (float)sin(-1.5708); rzr[5] = (float)cos(-1.5708); rzr[10] = 1; rzr[15] = 1;
  T[0] = (float)(cos(yaw) * cos(pitch));
  T[1] = (float)(cos(yaw) * sin(pitch) * sin(roll) - sin(yaw) * cos(roll));
  T[2] = (float)(cos(yaw) * sin(pitch) * cos(roll) + sin(yaw) * (quirks ? sin(pitch) : sin(roll)));  // tools.h:80-81 (quirk B2)
  T[4] = (float)(sin(yaw) * cos(pitch));
  T[5] = (float)(sin(yaw) * sin(pitch) * sin(roll) + cos(yaw) * cos(roll));
  T[6] = (float)(sin(yaw) * sin(pitch) * cos(roll) - cos(yaw) * sin(roll));
  T[8] = (float)(-sin(pitch)); T[9] = (float)(cos(pitch) * sin(roll)); T[10] = (float)(cos(pitch) * cos(roll)); T[15] = 1;
  float M[16];
  host_mat4_mul(T, rzr, M); host_mat4_mul(M, rxr, M); host_mat4_mul(M, rxc, out);
  return 0;
}

// One or several frames in one pass: the accepted boxes of ALL frames are packed back to back into one View, so that every
// kernel launch covers 32 x F boxes (a single frame's 32 boxes leave most of the 256 CUs idle: the raster recurrences of PCL's
// algorithms run as one workgroup per box).
// The kernels that stage bands / label images take up to ~150 KiB of dynamic LDS.  The opt-in belongs to the kernel, not to a call: one
// value, set once per device under a lock (handles on several host threads would otherwise overwrite one another's smaller values).
static int seg_lds_opt_in(int device) {
  static std::mutex mu;
  static std::vector<int> done;
  std::lock_guard<std::mutex> lk(mu);
  if (std::find(done.begin(), done.end(), device) != done.end()) return 0;
  const int v = lds_optin_limit(device, 156 * 1024, 4096);
  const void* fns[] = {(const void*)k_distance_map, (const void*)k_refine, (const void*)k_integral, (const void*)k_cc_lds, (const void*)k_refine_lds,
                       (const void*)k_contour<true>};
  for (const void* f : fns) SSLAM_HIP_TRY(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, v));
  done.push_back(device);
  return 0;
}
static int seg_enqueue(sslam_seg* s, const sslam_frame* frames, int n_frames, int width, int height, int point_step, int row_step, int ox, int oy, int oz,
                       sslam_seg* peer = nullptr) {
  if (!s || !frames || n_frames <= 0) return set_error(SSLAM_ERR_INVALID, "null argument");
  if (s->q_busy) return set_error(SSLAM_ERR_INVALID, "the previous batch of this pipeline has not been collected");
  for (int f = 0; f < n_frames; ++f)
    if (!frames[f].cloud || (!frames[f].boxes && frames[f].n_boxes > 0)) return set_error(SSLAM_ERR_INVALID, "frame %d: null cloud or boxes", f);
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) return set_error(SSLAM_ERR_NO_DEVICE, "no HIP device visible; the product has no CPU fallback");
  SSLAM_HIP_TRY(hipSetDevice(s->P.device));
  if (!s->stream) SSLAM_HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  s->q_t0 = std::chrono::steady_clock::now();
  const sslam_seg_params& P = s->P;
  // ---- host-side box filter: class whitelist (point_cloud_segmentation.h:126-130), crop bounds
  //      (plane_segmentation.cpp:34-38), minimum point count (:93-95)
  s->boxes.clear(); s->box_src.clear(); s->box_frame.clear();
  s->r_nbox = -1;
  size_t npix = 0, nii = 0;
  int maxpix = 1;
  for (int f = 0; f < n_frames; ++f)
  for (int i = 0; i < frames[f].n_boxes; ++i) {
    const sslam_box& b = frames[f].boxes[i];
    if (b.class_id < SSLAM_CLASS_CHAIR || b.class_id > SSLAM_CLASS_CAR) continue;
    if (b.height < 0 || b.width < 0 || b.tl_x < 0 || b.tl_y < 0 || (b.tl_x + b.width) > P.image_width || (b.tl_y + b.height) > P.image_height) continue;
    if (b.tl_x + b.width > width || b.tl_y + b.height > height) continue;
    const size_t n = (size_t)b.width * b.height;
    if (n == 0 || (double)n < P.norm_point_thres) continue;
    if (b.width > kBandFloats / 8 - 1) return set_error(SSLAM_ERR_UNSUPPORTED, "box wider than %d px", kBandFloats / 8 - 1);
    BoxMeta m{b.width, b.height, b.tl_x, b.tl_y, (int)npix, (int)nii, (int)s->boxes.size(), f};
    s->boxes.push_back(m); s->box_src.push_back(i); s->box_frame.push_back(f);
    if (npix + n >= ((size_t)1 << 29)) return set_error(SSLAM_ERR_UNSUPPORTED, "too many box pixels in one call (%zu): split the batch", npix + n);
    npix += n; nii += (size_t)(b.width + 1) * (b.height + 1);
    maxpix = std::max(maxpix, (int)n);
  }
  const int nb = (int)s->boxes.size();
  View& V = s->V;
  const size_t frame_bytes = (size_t)row_step * height;
  const size_t cloud_bytes = frame_bytes * n_frames;
  if (cloud_bytes > s->cap_cloud || npix > s->cap_pix || nii > s->cap_ii || (size_t)nb > s->cap_box) {
    SSLAM_HIP_TRY(hipStreamSynchronize(s->stream));
    s->free_all();
    s->cap_cloud = cloud_bytes; s->cap_pix = std::max<size_t>(npix, 1); s->cap_ii = std::max<size_t>(nii, 1); s->cap_box = std::max(nb, 1);
    int rc;
    if ((rc = seg_alloc(s, s->cap_cloud, &s->d_cloud))) return rc;
    if ((rc = seg_alloc(s, s->cap_box, &s->d_box))) return rc;
    if ((rc = seg_alloc(s, s->cap_pix * 3, &V.pts))) return rc;
    if ((rc = seg_alloc(s, s->cap_pix, &V.dm))) return rc;
    if ((rc = seg_alloc(s, s->cap_pix * 4, &V.nrm))) return rc;
    if ((rc = seg_alloc(s, s->cap_pix, &V.pd))) return rc;
    if ((rc = seg_alloc(s, s->cap_pix, &V.lab))) return rc;
    if ((rc = seg_alloc(s, s->cap_pix, &V.cnt))) return rc;
    if ((rc = seg_alloc(s, s->cap_pix, &V.l2m))) return rc;
    if ((rc = seg_alloc(s, s->cap_pix, &V.code))) return rc;
    if ((rc = seg_alloc(s, s->cap_ii * 10, &V.ii))) return rc;
    if ((rc = seg_alloc(s, s->cap_box * kMaxRegions, &V.reg))) return rc;
    if ((rc = seg_alloc(s, s->cap_box, &V.nreg))) return rc;
    if ((rc = seg_alloc(s, s->cap_pix * 4, &V.contour))) return rc;
    if ((rc = seg_alloc(s, s->cap_box, &V.ccount))) return rc;
    if ((rc = seg_alloc(s, (size_t)8, &V.overflow))) return rc;   // [2] overflow counters (+ spare)
  }
  V.nbox = nb; V.npix_total = (int)npix; V.maxpix = maxpix;
  V.box = s->d_box; V.cloud = s->d_cloud; V.cloud_stride = frame_bytes;
  V.point_step = point_step; V.row_step = row_step; V.ox = ox; V.oy = oy; V.oz = oz;
  V.mdcf = P.max_depth_change_factor; V.smoothing = P.normal_smoothing_size;
  V.ang_thr_cos = cosf(P.angular_threshold); V.dist_thr = P.distance_threshold; V.max_curv = P.maximum_curvature;
  V.min_inliers = (unsigned)P.num_point_seg;
  V.refine_bh = nb > 512 ? 24 : 64;
  s->q_frames.resize(n_frames);
  for (int f = 0; f < n_frames; ++f) { memcpy(s->q_frames[f].robot_pose, frames[f].robot_pose, sizeof(float) * 6); s->q_frames[f].cam_angle = frames[f].cam_angle; }
  s->q_boxes.resize(nb);
  for (int bi = 0; bi < nb; ++bi) { const sslam_box& sb = frames[s->box_frame[bi]].boxes[s->box_src[bi]]; s->q_boxes[bi] = {sb.prob, sb.class_id}; }
  if ((size_t)nb > s->cap_q_box || !s->q_regs) {
    if (s->q_regs) (void)hipHostFree(s->q_regs);
    if (s->q_nreg) (void)hipHostFree(s->q_nreg);
    s->q_regs = nullptr; s->q_nreg = nullptr;
    s->cap_q_box = std::max<size_t>(nb, 1);
    SSLAM_HIP_TRY(hipHostMalloc((void**)&s->q_regs, s->cap_q_box * kMaxRegions * sizeof(Region), hipHostMallocDefault));
    SSLAM_HIP_TRY(hipHostMalloc((void**)&s->q_nreg, (s->cap_q_box + 2) * sizeof(int), hipHostMallocDefault));
  }
  Region* regs = s->q_regs;
  int* nreg = s->q_nreg;
  for (int bi = 0; bi < nb; ++bi) nreg[bi] = 0;
  int* ovf = s->q_nreg + s->cap_q_box;
  ovf[0] = ovf[1] = 0;
  if (nb > 0) {
    for (int f = 0; f < n_frames; ++f)
      SSLAM_HIP_TRY(hipMemcpyAsync(s->d_cloud + (size_t)f * frame_bytes, frames[f].cloud, frame_bytes, hipMemcpyHostToDevice, s->stream));
    SSLAM_HIP_TRY(hipMemsetAsync(V.overflow, 0, 8 * sizeof(int), s->stream));
    SSLAM_HIP_TRY(hipMemcpyAsync(s->d_box, s->boxes.data(), nb * sizeof(BoxMeta), hipMemcpyHostToDevice, s->stream));
    if (!s->q_e0) { SSLAM_HIP_TRY(hipEventCreate(&s->q_e0)); SSLAM_HIP_TRY(hipEventCreate(&s->q_e1)); }
    hipEvent_t e0 = s->q_e0, e1 = s->q_e1;
    // Two pipelines that start together stay in lock step: both copy at the same time (each at half the PCIe rate), then both run their
    // kernels side by side, and nothing overlaps (rocprofv3 trace of round 5: 2.5 ms of copies without a kernel, then 2.7 ms of kernels
    // without a copy).  The kernels of this batch therefore wait for the peer pipeline's kernels: the copies of batch k + 1 then run under
    // the kernels of batch k by construction, whatever the host's timing.
    if (peer && peer->q_e1 && peer->q_e1_armed) SSLAM_HIP_TRY(hipStreamWaitEvent(s->stream, peer->q_e1, 0));
    SSLAM_HIP_TRY(hipEventRecord(e0, s->stream));
    const dim3 pg((maxpix + 255) / 256, nb), pb(256);
    int maxw = 1;
    for (auto& b : s->boxes) maxw = std::max(maxw, b.w);
    // dynamic LDS: the largest per-box band (the kernels size their bands from the box's own width)
    size_t band_bytes = 0, rband_bytes = 0;
    for (auto& b : s->boxes) {
      const int bh = std::min(64, kBandFloats / b.w - 1);
      band_bytes = std::max(band_bytes, (size_t)(bh + 1) * b.w * sizeof(float));
      const int bhr = std::min(V.refine_bh, kBandFloats / (4 * b.w) - 1);
      rband_bytes = std::max(rband_bytes, (size_t)(bhr + 1) * b.w * 4 * sizeof(float));
    }
    { const int rc_lds = seg_lds_opt_in(s->P.device); if (rc_lds) return rc_lds; }   // > 64 KiB of dynamic LDS for the band / label kernels, once per device
    SSLAM_HIP_TRY(hipMemsetAsync(V.ccount, 0, nb * sizeof(int), s->stream));
    hipLaunchKernelGGL(k_crop, pg, pb, 0, s->stream, V);
    hipLaunchKernelGGL(k_depth_change, pg, pb, 0, s->stream, V);
    hipLaunchKernelGGL(k_distance_map, dim3(nb), dim3(256), band_bytes, s->stream, V);
    {
      // band rows: as many as fit next to the (w+1) x 10 double carry row in ~150 KiB of LDS (<= 64 lanes)
      const size_t carry = (size_t)(maxw + 1) * 10 * sizeof(double);
      // (a call with more boxes than CUs takes 16-row bands: four boxes per CU instead of one)
      const size_t row_cap = nb > 512 ? 16 : 64;   // (1024 boxes of 128 x 96, per call: 12 / 16 / 24 / 32 / 48 / 64 rows 0.599 / 0.551 / 0.604 / 0.606 / 0.680 / 0.706 ms)
      const int ib_rows = (int)std::max<size_t>(1, std::min<size_t>(row_cap, (150 * 1024 - carry) / ((size_t)maxw * 12)));
      const size_t ilds = carry + (size_t)ib_rows * maxw * 12;
      hipLaunchKernelGGL(k_integral, dim3(nb), dim3(256), ilds, s->stream, V, ib_rows);
    }
    hipLaunchKernelGGL(k_normals, pg, pb, 0, s->stream, V);
    if (maxpix <= kCcLdsMax && !getenv("SSLAM_SEG_GLOBAL_CC")) {
      hipLaunchKernelGGL(k_cc_lds, dim3(nb), dim3(1024), (size_t)maxpix * sizeof(int), s->stream, V);
    } else {
      hipLaunchKernelGGL(k_cc_init, pg, pb, 0, s->stream, V);
      hipLaunchKernelGGL(k_cc_merge, pg, pb, 0, s->stream, V);
      hipLaunchKernelGGL(k_cc_flatten, pg, pb, 0, s->stream, V);
      hipLaunchKernelGGL(k_cc_flatten2, pg, pb, 0, s->stream, V);
    }
    if (nb > 256) hipLaunchKernelGGL(k_regions<256>, dim3(nb), dim3(256), 0, s->stream, V);
    else hipLaunchKernelGGL(k_regions<1024>, dim3(nb), dim3(1024), 0, s->stream, V);
    hipLaunchKernelGGL(k_relabel, pg, pb, 0, s->stream, V);
    bool tiles_ok = true;   // k_refine_lds: every thread's tile must fit its 32-bit candidate mask
    for (auto& b : s->boxes) tiles_ok = tiles_ok && ((b.w + kRefTX - 1) / kRefTX) * ((b.h + kRefTY - 1) / kRefTY) <= 32;
    if (maxpix <= kCcLdsMax && tiles_ok && !getenv("SSLAM_SEG_WAVEFRONT_REFINE")) {
      // 16-bit labels + 16-bit masks: 4 bytes per pixel (three 12k-pixel boxes per CU)
      hipLaunchKernelGGL(k_refine_lds, dim3(nb), dim3(kRefTX * kRefTY), (size_t)maxpix * 4 + 8, s->stream, V);
    } else {
      hipLaunchKernelGGL(k_refine, dim3(nb), dim3(256), rband_bytes, s->stream, V);
    }
    size_t cimg_bytes = 0;
    for (auto& b : s->boxes) cimg_bytes = std::max(cimg_bytes, (size_t)(b.w + 2) * (b.h + 2));
    if (cimg_bytes <= 150 * 1024) {
      hipLaunchKernelGGL(k_contour<true>, dim3(nb), dim3(256), cimg_bytes, s->stream, V);
    } else {
      hipLaunchKernelGGL(k_contour<false>, dim3(nb), dim3(256), 0, s->stream, V);
    }
    hipLaunchKernelGGL(k_area, dim3(nb, kMaxRegions), dim3(64), 0, s->stream, V);
    SSLAM_HIP_TRY(hipEventRecord(e1, s->stream));
    s->q_e1_armed = true;
    SSLAM_HIP_TRY(hipMemcpyAsync(regs, V.reg, (size_t)nb * kMaxRegions * sizeof(Region), hipMemcpyDeviceToHost, s->stream));
    SSLAM_HIP_TRY(hipMemcpyAsync(nreg, V.nreg, nb * sizeof(int), hipMemcpyDeviceToHost, s->stream));
    SSLAM_HIP_TRY(hipMemcpyAsync(ovf, V.overflow, 2 * sizeof(int), hipMemcpyDeviceToHost, s->stream));
  }
  s->q_busy = true;
  return 0;
}

// wait for the batch seg_enqueue put on this pipeline, then plane_segmentation.cpp:158-256 + point_cloud_segmentation.h:43-99
// (scalar post-processing)
static int seg_finish(sslam_seg* s, sslam_plane* out, int max_out, int32_t* out_frame) {
  if (!s || (!out && max_out > 0)) return set_error(SSLAM_ERR_INVALID, "null argument");
  if (!s->q_busy) return set_error(SSLAM_ERR_INVALID, "no batch was submitted on this pipeline");
  s->q_busy = false;
  SSLAM_HIP_TRY(hipSetDevice(s->P.device));
  const sslam_seg_params& P = s->P;
  const int nb = (int)s->boxes.size();
  const Region* regs = s->q_regs;
  const int* nreg = s->q_nreg;
  const int* ovf = s->q_nreg + s->cap_q_box;
  float kernel_ms = 0;
  if (nb > 0) {
    SSLAM_HIP_TRY(hipStreamSynchronize(s->stream));
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return set_error(SSLAM_ERR_HIP, "frontend kernels: %s", hipGetErrorString(le));
    SSLAM_HIP_TRY(hipEventElapsedTime(&kernel_ms, s->q_e0, s->q_e1));
  }
  int nout = 0, dropped = 0;
  for (int bi = 0; bi < nb; ++bi) {
    const sslam_seg::FrameMeta& fr = s->q_frames[s->box_frame[bi]];
    const float* robot_pose = fr.robot_pose;
    float T[16];
    sslam_seg_transform(s, robot_pose, fr.cam_angle, T);
    const float hz[3] = {T[8], T[9], T[10]};
    const sslam_seg::BoxInfo& sb = s->q_boxes[bi];
    for (int k = 0; k < nreg[bi]; ++k) {
      const Region& R = regs[(size_t)bi * kMaxRegions + k];
      if (!(R.contour_n > P.min_contour_points)) continue;
      const float* m = R.model;
      const float dotp = hz[0] * m[0] + hz[1] * m[1] + hz[2] * m[2];
      if (!((double)R.area >= P.planar_area)) continue;
      int type = -1;
      float sgn = 1.0f;
      if ((float)(fabsf(m[0]) - fabsf(hz[0])) < 0.3 && (float)(fabsf(m[1]) - fabsf(hz[1])) < 0.3 && (float)(fabsf(m[2]) - fabsf(hz[2])) < 0.3) {
        type = 0;
        if (m[1] > 0) sgn = -1.0f;
      } else if (dotp < 0.5) {
        type = 1;
        if (m[0] > 0) sgn = -1.0f;
      }
      if (type < 0) continue;
      if (nout >= max_out) { ++dropped; continue; }
      if (out_frame) out_frame[nout] = s->box_frame[bi];
      sslam_plane& o = out[nout++];
      memcpy(o.centroid_cam, R.centroid, 12);
      for (int q = 0; q < 4; ++q) o.normal_d[q] = sgn < 0 ? -m[q] : m[q];
      for (int r = 0; r < 3; ++r) {
        float sacc = 0;
        for (int q = 0; q < 3; ++q) sacc += T[r * 4 + q] * R.centroid[q];
        sacc += T[r * 4 + 3] * 1.0f;
        o.world_pose[r] = sacc + robot_pose[r];
      }
      o.num_points = (float)R.contour_n; o.prob = sb.prob; o.plane_type = type; o.class_id = sb.class_id;
      o.box_index = s->box_src[bi]; o.inlier_count = R.inliers; o.area = R.area;
    }
  }
  s->last_dropped_planes = dropped; s->last_candidate_overflow = ovf[0]; s->last_region_overflow = ovf[1];
  s->last_kernel_ms = kernel_ms;
  s->last_total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - s->q_t0).count();
  return nout;
}
static int seg_run(sslam_seg* s, const sslam_frame* frames, int n_frames, int width, int height, int point_step, int row_step, int ox, int oy, int oz,
                   sslam_plane* out, int max_out, int32_t* out_frame) {
  if (!s || !frames || n_frames <= 0 || (!out && max_out > 0)) return set_error(SSLAM_ERR_INVALID, "null argument");
  if (s->n_inflight > 0) return set_error(SSLAM_ERR_INVALID, "%d submitted batch(es) must be collected first", s->n_inflight);
  const int rc = seg_enqueue(s, frames, n_frames, width, height, point_step, row_step, ox, oy, oz);
  if (rc) { s->q_busy = false; return rc; }
  return seg_finish(s, out, max_out, out_frame);
}


int sslam_seg_segment(sslam_seg* s, const uint8_t* cloud, int width, int height, int point_step, int row_step, int ox, int oy, int oz,
                      const sslam_box* boxes, int n_boxes, const float robot_pose[6], float cam_angle, sslam_plane* out, int max_out) {
  if (!s || !cloud || (!boxes && n_boxes > 0) || !robot_pose || (!out && max_out > 0)) return set_error(SSLAM_ERR_INVALID, "null argument");
  sslam_frame fr;
  fr.cloud = cloud; fr.boxes = boxes; fr.n_boxes = n_boxes; fr.cam_angle = cam_angle;
  memcpy(fr.robot_pose, robot_pose, sizeof fr.robot_pose);
  return seg_run(s, &fr, 1, width, height, point_step, row_step, ox, oy, oz, out, max_out, nullptr);
}

int sslam_seg_segment_batch(sslam_seg* s, const sslam_frame* frames, int n_frames, int width, int height, int point_step, int row_step,
                            int ox, int oy, int oz, sslam_plane* out, int max_out, int32_t* out_frame) {
  return seg_run(s, frames, n_frames, width, height, point_step, row_step, ox, oy, oz, out, max_out, out_frame);
}

// Pipelined form: at most two batches in flight, on two pipelines (own stream, own device buffers) used in turn; the 9.8 MB-per-frame
// H2D copy of batch k+1 runs under the kernels of batch k.  submit(0); loop { submit(k+1); collect(k); }
int sslam_seg_submit_batch(sslam_seg* s, const sslam_frame* frames, int n_frames, int width, int height, int point_step, int row_step,
                           int ox, int oy, int oz) {
  if (!s) return set_error(SSLAM_ERR_INVALID, "null argument");
  if (s->n_inflight >= 2) return set_error(SSLAM_ERR_INVALID, "two batches are in flight: collect one first");
  int which = 0;
  if (s->n_inflight == 1) which = 1 - s->fifo[0];
  sslam_seg* pipe = s;
  if (which == 1) {
    if (!s->twin) { s->twin = new sslam_seg(); s->twin->P = s->P; }
    pipe = s->twin;
  }
  const int rc = seg_enqueue(pipe, frames, n_frames, width, height, point_step, row_step, ox, oy, oz, pipe == s ? s->twin : s);
  if (rc) { pipe->q_busy = false; return rc; }
  s->fifo[s->n_inflight++] = which;
  return 0;
}
int sslam_seg_collect_batch(sslam_seg* s, sslam_plane* out, int max_out, int32_t* out_frame) {
  if (!s) return set_error(SSLAM_ERR_INVALID, "null argument");
  if (s->n_inflight <= 0) return set_error(SSLAM_ERR_INVALID, "no batch in flight");
  sslam_seg* pipe = s->fifo[0] ? s->twin : s;
  s->fifo[0] = s->fifo[1];
  --s->n_inflight;
  const int n = seg_finish(pipe, out, max_out, out_frame);
  if (pipe != s) {   // the handle reports the last collected batch
    s->last_dropped_planes = pipe->last_dropped_planes; s->last_candidate_overflow = pipe->last_candidate_overflow;
    s->last_region_overflow = pipe->last_region_overflow; s->last_kernel_ms = pipe->last_kernel_ms; s->last_total_ms = pipe->last_total_ms;
  }
  return n;
}

int sslam_seg_last_overflow(const sslam_seg* s, int* dropped_planes, int* boxes_with_full_candidate_table, int* boxes_with_full_region_table) {
  if (!s) return set_error(SSLAM_ERR_INVALID, "null argument");
  if (dropped_planes) *dropped_planes = s->last_dropped_planes;
  if (boxes_with_full_candidate_table) *boxes_with_full_candidate_table = s->last_candidate_overflow;
  if (boxes_with_full_region_table) *boxes_with_full_region_table = s->last_region_overflow;
  return s->last_dropped_planes + s->last_candidate_overflow + s->last_region_overflow;
}

int sslam_seg_ransac_plane(sslam_seg* s, const float* xyz, int n, float threshold, int max_iterations, double probability,
                           uint64_t seed, float coeff_out[4], int32_t* inliers_out, int max_inliers) {
  if (!s || !xyz || !coeff_out || (!inliers_out && max_inliers > 0) || n < 0 || max_iterations < 1)
    return set_error(SSLAM_ERR_INVALID, "bad argument");
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) return set_error(SSLAM_ERR_NO_DEVICE, "no HIP device visible; the product has no CPU fallback");
  coeff_out[0] = coeff_out[1] = coeff_out[2] = coeff_out[3] = 0;
  if (n < 3) return 0;
  SSLAM_HIP_TRY(hipSetDevice(s->P.device));
  if (!s->stream) SSLAM_HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  const int max_skip = max_iterations * 10;
  const int H = max_iterations + 1 + max_skip;       // every hypothesis index the sequential loop can reach
  const int nblk = (n + 255) / 256;
  float *d_pts = nullptr, *d_models = nullptr;
  int *d_counts = nullptr, *d_blk = nullptr, *d_inl = nullptr, *d_total = nullptr;
  struct Guard {   // frees whatever was allocated on every exit path (an early SSLAM_HIP_TRY return included)
    std::vector<void**> ptrs;
    ~Guard() { for (void** p : ptrs) if (*p) (void)hipFree(*p); }
  } guard;
  guard.ptrs = {(void**)&d_pts, (void**)&d_models, (void**)&d_counts, (void**)&d_blk, (void**)&d_inl, (void**)&d_total};
  SSLAM_HIP_TRY(hipMalloc((void**)&d_pts, (size_t)n * 3 * sizeof(float)));
  SSLAM_HIP_TRY(hipMalloc((void**)&d_models, (size_t)(H + 1) * 4 * sizeof(float)));
  SSLAM_HIP_TRY(hipMalloc((void**)&d_counts, (size_t)H * sizeof(int)));
  SSLAM_HIP_TRY(hipMalloc((void**)&d_blk, (size_t)nblk * sizeof(int)));
  SSLAM_HIP_TRY(hipMalloc((void**)&d_inl, (size_t)std::max(n, 1) * sizeof(int)));
  SSLAM_HIP_TRY(hipMalloc((void**)&d_total, sizeof(int)));
  auto cleanup = [&]() {};   // the guard frees
  SSLAM_HIP_TRY(hipMemcpyAsync(d_pts, xyz, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, s->stream));
  hipLaunchKernelGGL(k_ransac_score, dim3(H), dim3(256), 0, s->stream, d_pts, n, threshold, (unsigned long long)seed, d_models, d_counts);
  std::vector<int> counts(H);
  std::vector<float> models((size_t)H * 4);
  SSLAM_HIP_TRY(hipMemcpyAsync(counts.data(), d_counts, (size_t)H * sizeof(int), hipMemcpyDeviceToHost, s->stream));
  SSLAM_HIP_TRY(hipMemcpyAsync(models.data(), d_models, (size_t)H * 4 * sizeof(float), hipMemcpyDeviceToHost, s->stream));
  SSLAM_HIP_TRY(hipStreamSynchronize(s->stream));
  // replay of pcl::RandomSampleConsensus::computeModel over the precomputed counts
  int best = -1, best_it = -1, iterations = 0, skipped = 0, it = 0;
  double k = 1.0;
  const double log_probability = std::log(1.0 - probability), one_over = 1.0 / (double)n, eps = 2.220446049250313e-16;
  while ((double)iterations < k && skipped < max_skip && it < H) {
    const int cnt = counts[it];
    ++it;
    if (cnt < 0) { ++skipped; continue; }
    if (cnt > best) {
      best = cnt; best_it = it - 1;
      const double w = (double)best * one_over;
      double p_no = 1.0 - std::pow(w, 3.0);
      p_no = std::max(eps, p_no); p_no = std::min(1.0 - eps, p_no);
      k = log_probability / std::log(p_no);
    }
    ++iterations;
    if (iterations > max_iterations) break;
  }
  if (best <= 0) { cleanup(); return 0; }
  float* d_model = d_models + (size_t)H * 4;   // working copy of the winning model
  SSLAM_HIP_TRY(hipMemcpyAsync(d_model, d_models + (size_t)best_it * 4, 4 * sizeof(float), hipMemcpyDeviceToDevice, s->stream));
  for (int pass = 0; pass < 2; ++pass) {   // inliers of the sampled model -> refit -> inliers of the refined model
    hipLaunchKernelGGL(k_ransac_mark, dim3(nblk), dim3(256), 0, s->stream, d_pts, n, threshold, d_model, d_blk);
    hipLaunchKernelGGL(k_ransac_scan, dim3(1), dim3(64), 0, s->stream, d_blk, nblk, d_total);
    hipLaunchKernelGGL(k_ransac_write, dim3(nblk), dim3(256), 0, s->stream, d_pts, n, threshold, d_model, d_blk, d_inl, n);
    if (pass == 0) hipLaunchKernelGGL(k_ransac_refit, dim3(1), dim3(64), 0, s->stream, d_pts, d_inl, d_total, d_model);
  }
  int total = 0;
  SSLAM_HIP_TRY(hipMemcpyAsync(&total, d_total, sizeof(int), hipMemcpyDeviceToHost, s->stream));
  SSLAM_HIP_TRY(hipMemcpyAsync(coeff_out, d_model, 4 * sizeof(float), hipMemcpyDeviceToHost, s->stream));
  SSLAM_HIP_TRY(hipStreamSynchronize(s->stream));
  if (total > 0 && max_inliers > 0)
    SSLAM_HIP_TRY(hipMemcpy(inliers_out, d_inl, (size_t)std::min(total, max_inliers) * sizeof(int), hipMemcpyDeviceToHost));
  hipError_t le = hipGetLastError();
  cleanup();
  if (le != hipSuccess) return set_error(SSLAM_ERR_HIP, "ransac kernels: %s", hipGetErrorString(le));
  return total;
}

// ---- RANSAC + ICP over the boxes of the resident batch (configs[3]) ------------------------------------------------------------------
static int seg_resident_check(sslam_seg* s) {
  if (!s) return set_error(SSLAM_ERR_INVALID, "null handle");
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) return set_error(SSLAM_ERR_NO_DEVICE, "no HIP device visible; the product has no CPU fallback");
  if (s->q_busy || s->n_inflight > 0) return set_error(SSLAM_ERR_INVALID, "a batch is in flight: collect it first");
  if (!s->stream || s->V.nbox != (int)s->boxes.size()) return set_error(SSLAM_ERR_INVALID, "no resident batch: call sslam_seg_segment / sslam_seg_segment_batch first");
  return 0;
}
static int seg_ransac_lds_opt_in(int device, int* limit) {
  static std::mutex mu;
  static std::vector<std::pair<int, int>> done;
  std::lock_guard<std::mutex> lk(mu);
  for (auto& d : done) if (d.first == device) { *limit = d.second; return 0; }
  const int v = lds_optin_limit(device, 156 * 1024, 4096);
  SSLAM_HIP_TRY(hipFuncSetAttribute((const void*)k_ransac_hyp, hipFuncAttributeMaxDynamicSharedMemorySize, v));
  done.push_back({device, v});
  *limit = v;
  return 0;
}

int sslam_seg_ransac_boxes(sslam_seg* s, float threshold, int max_iterations, double probability, uint64_t seed, sslam_box_plane* out, int max_out,
                           double* kernel_ms) {
  int rc = seg_resident_check(s);
  if (rc) return rc;
  if ((!out && max_out > 0) || max_iterations < 1 || !(threshold > 0) || !(probability > 0 && probability < 1)) return set_error(SSLAM_ERR_INVALID, "bad argument");
  SSLAM_HIP_TRY(hipSetDevice(s->P.device));
  const int nb = (int)s->boxes.size();
  if (kernel_ms) *kernel_ms = 0;
  if (nb == 0) { s->r_nbox = 0; return 0; }
  const size_t npix = (size_t)s->V.npix_total;
  if (npix > s->cap_rflag) {
    if (s->d_rflag) (void)hipFree(s->d_rflag);
    s->d_rflag = nullptr; s->cap_rflag = 0;
    SSLAM_HIP_TRY(hipMalloc((void**)&s->d_rflag, npix));
    s->cap_rflag = npix;
  }
  if ((size_t)nb > s->cap_rbox) {
    if (s->d_rbox) (void)hipFree(s->d_rbox);
    s->d_rbox = nullptr; s->cap_rbox = 0;
    SSLAM_HIP_TRY(hipMalloc(&s->d_rbox, (size_t)nb * sizeof(RansacBox)));
    s->cap_rbox = nb;
  }
  int lds_limit = 0;
  if ((rc = seg_ransac_lds_opt_in(s->P.device, &lds_limit))) return rc;
  // a box whose points fit the LDS is scored out of it; larger boxes read their points through L2
  const int lds_points = std::max(0, (lds_limit - 1024) / 12);
  int max_staged = 0;
  for (auto& b : s->boxes) { const int n = b.w * b.h; if (n <= lds_points) max_staged = std::max(max_staged, n); }
  const size_t lds = (size_t)max_staged * 12 + 16;
  if (!s->q_e0) { SSLAM_HIP_TRY(hipEventCreate(&s->q_e0)); SSLAM_HIP_TRY(hipEventCreate(&s->q_e1)); }
  SSLAM_HIP_TRY(hipEventRecord(s->q_e0, s->stream));
  hipLaunchKernelGGL(k_ransac_hyp, dim3(nb), dim3(1024), lds, s->stream, s->V, threshold, max_iterations, probability, (unsigned long long)seed, lds_points,
                     (RansacBox*)s->d_rbox);
  hipLaunchKernelGGL(k_ransac_refine, dim3(nb), dim3(256), 0, s->stream, s->V, threshold, (RansacBox*)s->d_rbox, s->d_rflag);
  SSLAM_HIP_TRY(hipGetLastError());
  SSLAM_HIP_TRY(hipEventRecord(s->q_e1, s->stream));
  std::vector<RansacBox> rb(nb);
  SSLAM_HIP_TRY(hipMemcpyAsync(rb.data(), s->d_rbox, (size_t)nb * sizeof(RansacBox), hipMemcpyDeviceToHost, s->stream));
  SSLAM_HIP_TRY(hipStreamSynchronize(s->stream));
  if (kernel_ms) { float ms = 0; if (hipEventElapsedTime(&ms, s->q_e0, s->q_e1) == hipSuccess) *kernel_ms = ms; }
  s->r_nbox = nb;
  for (int q = 0; q < nb && q < max_out; ++q) {
    sslam_box_plane& o = out[q];
    for (int k = 0; k < 4; ++k) o.coeff[k] = rb[q].coeff[k];
    o.inliers = rb[q].inliers; o.points = s->boxes[q].w * s->boxes[q].h; o.box_index = s->box_src[q]; o.frame = s->box_frame[q];
    o.hypotheses = rb[q].hyps; o.best_iteration = rb[q].best_iter;
  }
  return nb;
}

int sslam_seg_ransac_box_inliers(sslam_seg* s, int slot, int32_t* out, int max_out) {
  int rc = seg_resident_check(s);
  if (rc) return rc;
  if (s->r_nbox < 0) return set_error(SSLAM_ERR_INVALID, "sslam_seg_ransac_boxes has not run on the resident batch");
  if (slot < 0 || slot >= s->r_nbox || (!out && max_out > 0)) return set_error(SSLAM_ERR_INVALID, "bad box slot %d (of %d)", slot, s->r_nbox);
  SSLAM_HIP_TRY(hipSetDevice(s->P.device));
  const BoxMeta& b = s->boxes[slot];
  std::vector<unsigned char> f((size_t)b.w * b.h);
  SSLAM_HIP_TRY(hipMemcpyAsync(f.data(), s->d_rflag + b.pix0, f.size(), hipMemcpyDeviceToHost, s->stream));
  SSLAM_HIP_TRY(hipStreamSynchronize(s->stream));
  int n = 0;
  for (size_t i = 0; i < f.size(); ++i) if (f[i]) { if (n < max_out) out[n] = (int32_t)i; ++n; }
  return n;
}

int sslam_seg_icp_boxes(sslam_seg* s, const int32_t* box_plane, int n_boxes, const float* planes, int n_planes, int iterations, const double* T0,
                        sslam_icp_result* out, int max_out, double* kernel_ms) {
  int rc = seg_resident_check(s);
  if (rc) return rc;
  if (s->r_nbox < 0) return set_error(SSLAM_ERR_INVALID, "sslam_seg_ransac_boxes has not run on the resident batch");
  if (!box_plane || !planes || !out || n_boxes != s->r_nbox || n_planes <= 0 || iterations < 0 || max_out < 0) return set_error(SSLAM_ERR_INVALID, "bad argument (the batch holds %d boxes)", s->r_nbox);
  SSLAM_HIP_TRY(hipSetDevice(s->P.device));
  const int nf = (int)s->q_frames.size();
  if (kernel_ms) *kernel_ms = 0;
  if (nf == 0) return 0;
  const int nb = s->r_nbox;
  // one upload: [frame_box0 (nf + 1) | box_frame (nb) | box_plane (nb) | planes (4 n_planes floats) | IcpState (nf)]; scratch lives in the
  // handle and only grows (round-4 ADVICE: five hipMalloc / hipFree pairs per call synchronised the device)
  const size_t o_bf = (size_t)(nf + 1), o_bp = o_bf + nb, o_pl = o_bp + nb;
  const size_t o_st = ((o_pl + (size_t)4 * n_planes) * 4 + 15) / 16 * 16;   // bytes
  const size_t in_bytes = o_st + (size_t)nf * sizeof(IcpState);
  const size_t part_bytes = (size_t)std::max(nb, 1) * kIcpSums * sizeof(double), out_bytes = (size_t)nf * sizeof(IcpFrame);
  const size_t need = (in_bytes + 255) / 256 * 256 + (part_bytes + 255) / 256 * 256 + out_bytes;
  if (s->cap_icp < need) {
    SSLAM_HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->d_icp) (void)hipFree(s->d_icp);
    s->d_icp = nullptr; s->cap_icp = 0;
    SSLAM_HIP_TRY(hipMalloc(&s->d_icp, need + 4096));
    s->cap_icp = need + 4096;
  }
  s->h_icp.assign(in_bytes, 0);
  int* hi = reinterpret_cast<int*>(s->h_icp.data());
  for (int q = 0; q < nb; ++q) hi[s->box_frame[q] + 1]++;   // the slots of a frame are consecutive (frames are packed in order)
  for (int f = 0; f < nf; ++f) hi[f + 1] += hi[f];
  for (int q = 0; q < nb; ++q) { hi[o_bf + q] = s->box_frame[q]; hi[o_bp + q] = box_plane[q]; }
  std::memcpy(hi + o_pl, planes, (size_t)4 * n_planes * sizeof(float));
  IcpState* hs = reinterpret_cast<IcpState*>(s->h_icp.data() + o_st);
  for (int f = 0; f < nf; ++f) {
    for (int k = 0; k < 12; ++k) hs[f].T[k] = T0 ? T0[(size_t)f * 12 + k] : ((k < 9 && k % 4 == 0) ? 1.0 : 0.0);
    hs[f].status = 0; hs[f].done = 0;
  }
  char* base = static_cast<char*>(s->d_icp);
  const int* d_fb0 = reinterpret_cast<const int*>(base);
  const int* d_bf = d_fb0 + o_bf;
  const int* d_bp = d_fb0 + o_bp;
  const float* d_planes = reinterpret_cast<const float*>(d_fb0 + o_pl);
  IcpState* d_st = reinterpret_cast<IcpState*>(base + o_st);
  double* d_part = reinterpret_cast<double*>(base + (in_bytes + 255) / 256 * 256);
  IcpFrame* d_out = reinterpret_cast<IcpFrame*>(reinterpret_cast<char*>(d_part) + (part_bytes + 255) / 256 * 256);
  SSLAM_HIP_TRY(hipMemcpyAsync(base, s->h_icp.data(), in_bytes, hipMemcpyHostToDevice, s->stream));
  if (!s->q_e0) { SSLAM_HIP_TRY(hipEventCreate(&s->q_e0)); SSLAM_HIP_TRY(hipEventCreate(&s->q_e1)); }
  SSLAM_HIP_TRY(hipEventRecord(s->q_e0, s->stream));
  for (int it = 0; it <= iterations; ++it) {
    if (nb > 0) hipLaunchKernelGGL(k_icp_box_sums, dim3(nb), dim3(256), 0, s->stream, s->V, (const unsigned char*)s->d_rflag, d_bf, d_bp, d_planes, n_planes, (const IcpState*)d_st, d_part);
    hipLaunchKernelGGL(k_icp_frame_step, dim3(nf), dim3(64), 0, s->stream, d_fb0, (const double*)d_part, it, iterations, d_st, d_out);
  }
  SSLAM_HIP_TRY(hipGetLastError());
  SSLAM_HIP_TRY(hipEventRecord(s->q_e1, s->stream));
  std::vector<IcpFrame> res(nf);
  SSLAM_HIP_TRY(hipMemcpyAsync(res.data(), d_out, out_bytes, hipMemcpyDeviceToHost, s->stream));
  SSLAM_HIP_TRY(hipStreamSynchronize(s->stream));
  if (kernel_ms) { float ms = 0; if (hipEventElapsedTime(&ms, s->q_e0, s->q_e1) == hipSuccess) *kernel_ms = ms; }
  for (int f = 0; f < nf && f < max_out; ++f) {
    for (int k = 0; k < 12; ++k) out[f].T[k] = res[f].T[k];
    out[f].rms = res[f].rms; out[f].used = res[f].used; out[f].status = res[f].status;
  }
  return nf;
}

int sslam_seg_convex_hull_2d(sslam_seg* s, const float* xyz, int n, const int32_t* inliers, int n_inliers, const float coeff[4],
                             float* projected_out, int32_t* hull_out, int max_hull, int* axes_out) {
  if (!s || !xyz || !inliers || !coeff || (!hull_out && max_hull > 0) || n < 0 || n_inliers < 0)
    return set_error(SSLAM_ERR_INVALID, "bad argument");
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) return set_error(SSLAM_ERR_NO_DEVICE, "no HIP device visible; the product has no CPU fallback");
  if (axes_out) *axes_out = -1;
  if (n_inliers < 3) return set_error(SSLAM_ERR_INVALID, "a 2-D hull needs at least 3 inliers");
  for (int k = 0; k < n_inliers; ++k) if (inliers[k] < 0 || inliers[k] >= n) return set_error(SSLAM_ERR_INVALID, "inlier index out of range");
  SSLAM_HIP_TRY(hipSetDevice(s->P.device));
  if (!s->stream) SSLAM_HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  const int m0 = n_inliers, nblk = (m0 + 255) / 256;
  std::vector<void*> bufs;
  auto alloc = [&](size_t bytes) -> void* { void* p = nullptr; if (hipMalloc(&p, std::max<size_t>(bytes, 16)) != hipSuccess) return nullptr; bufs.push_back(p); return p; };
  auto cleanup = [&]() { for (void* p : bufs) (void)hipFree(p); };
  float* d_pts = (float*)alloc((size_t)n * 3 * sizeof(float));
  int* d_inl = (int*)alloc((size_t)m0 * sizeof(int));
  float* d_proj = (float*)alloc((size_t)m0 * 3 * sizeof(float));
  float* d_px = (float*)alloc((size_t)m0 * sizeof(float)); float* d_py = (float*)alloc((size_t)m0 * sizeof(float));
  float* d_cx = (float*)alloc((size_t)m0 * sizeof(float)); float* d_cy = (float*)alloc((size_t)m0 * sizeof(float)); int* d_ci = (int*)alloc((size_t)m0 * sizeof(int));
  const int max_chunks = (m0 + kHullCap - 1) / kHullCap;
  float* d_ox = (float*)alloc((size_t)max_chunks * kHullCap * sizeof(float)); float* d_oy = (float*)alloc((size_t)max_chunks * kHullCap * sizeof(float));
  int* d_oi = (int*)alloc((size_t)max_chunks * kHullCap * sizeof(int));
  int* d_meta = (int*)alloc(4 * sizeof(int));
  double* d_bkey = (double*)alloc((size_t)nblk * 8 * sizeof(double)); int* d_bidx = (int*)alloc((size_t)nblk * 8 * sizeof(int));
  double* d_ext = (double*)alloc(17 * sizeof(double)); int* d_eidx = (int*)alloc(8 * sizeof(int));
  int* d_blk = (int*)alloc((size_t)nblk * sizeof(int)); int* d_total = (int*)alloc(sizeof(int));
  int* d_counts = (int*)alloc((size_t)max_chunks * sizeof(int)); int* d_off = (int*)alloc((size_t)max_chunks * sizeof(int));
  if (!d_pts || !d_inl || !d_proj || !d_px || !d_py || !d_cx || !d_cy || !d_ci || !d_ox || !d_oy || !d_oi || !d_meta || !d_bkey || !d_bidx || !d_ext ||
      !d_eidx || !d_blk || !d_total || !d_counts || !d_off) { cleanup(); return set_error(SSLAM_ERR_HIP, "hipMalloc failed (convex hull)"); }
#define SSLAM_HULL_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return set_error(SSLAM_ERR_HIP, "%s: %s", #x, hipGetErrorString(e_)); } } while (0)
  SSLAM_HULL_TRY(hipMemcpyAsync(d_pts, xyz, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, s->stream));
  SSLAM_HULL_TRY(hipMemcpyAsync(d_inl, inliers, (size_t)m0 * sizeof(int), hipMemcpyHostToDevice, s->stream));
  hipLaunchKernelGGL(k_hull_project, dim3(nblk), dim3(256), 0, s->stream, d_pts, d_inl, m0, coeff[0], coeff[1], coeff[2], coeff[3], d_proj);
  hipLaunchKernelGGL(k_hull_axes, dim3(1), dim3(1), 0, s->stream, d_proj, m0, cosf(0.174532925f), d_meta);
  int axes = -1;
  SSLAM_HULL_TRY(hipMemcpyAsync(&axes, d_meta, sizeof(int), hipMemcpyDeviceToHost, s->stream));
  if (projected_out) SSLAM_HULL_TRY(hipMemcpyAsync(projected_out, d_proj, (size_t)m0 * 3 * sizeof(float), hipMemcpyDeviceToHost, s->stream));
  SSLAM_HULL_TRY(hipStreamSynchronize(s->stream));
  if (axes_out) *axes_out = axes;
  if (axes < 0) { cleanup(); return set_error(SSLAM_ERR_NUMERIC, "convex hull: the projected inliers are collinear"); }
  hipLaunchKernelGGL(k_hull_coords, dim3(nblk), dim3(256), 0, s->stream, d_proj, m0, d_meta, d_px, d_py);
  hipLaunchKernelGGL(k_hull_extreme_partial, dim3(nblk), dim3(256), 0, s->stream, d_px, d_py, m0, d_bkey, d_bidx);
  hipLaunchKernelGGL(k_hull_extreme_final, dim3(1), dim3(64), 0, s->stream, d_bkey, d_bidx, nblk, d_px, d_py, d_ext, d_eidx);
  hipLaunchKernelGGL(k_hull_mark, dim3(nblk), dim3(256), 0, s->stream, d_px, d_py, m0, d_ext, d_blk);
  hipLaunchKernelGGL(k_ransac_scan, dim3(1), dim3(64), 0, s->stream, d_blk, nblk, d_total);
  hipLaunchKernelGGL(k_hull_write, dim3(nblk), dim3(256), 0, s->stream, d_px, d_py, m0, d_ext, d_blk, d_cx, d_cy, d_ci);
  int m = 0;
  SSLAM_HULL_TRY(hipMemcpyAsync(&m, d_total, sizeof(int), hipMemcpyDeviceToHost, s->stream));
  SSLAM_HULL_TRY(hipStreamSynchronize(s->stream));
  while (m > kHullCap) {   // hull of hulls until one workgroup holds every candidate
    const int nchunks = (m + kHullCap - 1) / kHullCap;
    hipLaunchKernelGGL(k_hull_core<false>, dim3(nchunks), dim3(1024), 0, s->stream, d_cx, d_cy, d_ci, m, d_ox, d_oy, d_oi, d_counts);
    hipLaunchKernelGGL(k_hull_scan, dim3(1), dim3(64), 0, s->stream, d_counts, nchunks, d_off, d_total);
    hipLaunchKernelGGL(k_hull_compact, dim3(nchunks), dim3(256), 0, s->stream, d_ox, d_oy, d_oi, d_counts, d_off, d_cx, d_cy, d_ci);
    int m2 = 0;
    SSLAM_HULL_TRY(hipMemcpyAsync(&m2, d_total, sizeof(int), hipMemcpyDeviceToHost, s->stream));
    SSLAM_HULL_TRY(hipStreamSynchronize(s->stream));
    if (m2 >= m) { cleanup(); return set_error(SSLAM_ERR_UNSUPPORTED, "convex hull with more than %d vertices", kHullCap); }
    m = m2;
  }
  hipLaunchKernelGGL(k_hull_core<true>, dim3(1), dim3(1024), 0, s->stream, d_cx, d_cy, d_ci, m, d_ox, d_oy, d_oi, d_counts);
  int h = 0;
  SSLAM_HULL_TRY(hipMemcpyAsync(&h, d_counts, sizeof(int), hipMemcpyDeviceToHost, s->stream));
  SSLAM_HULL_TRY(hipStreamSynchronize(s->stream));
  if (h > 0 && max_hull > 0) SSLAM_HULL_TRY(hipMemcpy(hull_out, d_oi, (size_t)std::min(h, max_hull) * sizeof(int), hipMemcpyDeviceToHost));
  hipError_t le = hipGetLastError();
  cleanup();
#undef SSLAM_HULL_TRY
  if (le != hipSuccess) return set_error(SSLAM_ERR_HIP, "convex hull kernels: %s", hipGetErrorString(le));
  return h;
}

static int seg_find_slot(sslam_seg* s, int box) {
  for (size_t k = 0; k < s->box_src.size(); ++k) if (s->box_src[k] == box && s->box_frame[k] == 0) return (int)k;   // parity hooks address frame 0
  return -1;
}
int sslam_seg_get_normals(sslam_seg* s, int box, float* out) {
  if (!s || !out) return set_error(SSLAM_ERR_INVALID, "null argument");
  const int k = seg_find_slot(s, box);
  if (k < 0) return set_error(SSLAM_ERR_INVALID, "box %d was rejected or skipped in the last call", box);
  const BoxMeta& b = s->boxes[k];
  SSLAM_HIP_TRY(hipMemcpy(out, s->V.nrm + (size_t)b.pix0 * 4, (size_t)b.w * b.h * 4 * sizeof(float), hipMemcpyDeviceToHost));
  return b.w * b.h;
}
int sslam_seg_get_labels(sslam_seg* s, int box, int32_t* out) {
  if (!s || !out) return set_error(SSLAM_ERR_INVALID, "null argument");
  const int k = seg_find_slot(s, box);
  if (k < 0) return set_error(SSLAM_ERR_INVALID, "box %d was rejected or skipped in the last call", box);
  const BoxMeta& b = s->boxes[k];
  const int n = b.w * b.h;
  int* d = nullptr;
  SSLAM_HIP_TRY(hipMalloc((void**)&d, (size_t)n * sizeof(int)));
  hipLaunchKernelGGL(k_label_image, dim3((n + 255) / 256), dim3(256), 0, s->stream, s->V, k, d);
  SSLAM_HIP_TRY(hipStreamSynchronize(s->stream));
  SSLAM_HIP_TRY(hipMemcpy(out, d, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
  (void)hipFree(d);
  return n;
}
int sslam_seg_last_timing(const sslam_seg* s, double* kernel_ms, double* total_ms) {
  if (!s) return set_error(SSLAM_ERR_INVALID, "null argument");
  if (kernel_ms) *kernel_ms = s->last_kernel_ms;
  if (total_ms) *total_ms = s->last_total_ms;
  return 0;
}


// Point-to-plane ICP of labelled points against a set of planes (see k_icp_accumulate).  T_out = R (row-major) | t, 12 doubles.
int sslam_seg_icp_point_to_plane(sslam_seg* s, const float* xyz, const int32_t* labels, int n, const float* planes, int n_planes,
                                 int iterations, const double T0[12], double T_out[12], double* rms_out) {
  if (!s || !xyz || !labels || !planes || !T_out || n < 0 || n_planes <= 0 || iterations < 0) return set_error(SSLAM_ERR_INVALID, "bad argument");
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) return set_error(SSLAM_ERR_NO_DEVICE, "no HIP device visible; the product has no CPU fallback");
  SSLAM_HIP_TRY(hipSetDevice(s->P.device));
  if (!s->stream) SSLAM_HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  IcpPose T;
  if (T0) { for (int k = 0; k < 9; ++k) T.R[k] = T0[k]; for (int k = 0; k < 3; ++k) T.t[k] = T0[9 + k]; }
  else { for (int k = 0; k < 9; ++k) T.R[k] = (k % 4 == 0) ? 1.0 : 0.0; T.t[0] = T.t[1] = T.t[2] = 0.0; }
  const int nblk = std::max(1, std::min(256, (n + 255) / 256));
  float *d_pts = nullptr, *d_planes = nullptr;
  int* d_lab = nullptr;
  double* d_part = nullptr;
  struct Guard {
    std::vector<void**> ptrs;
    ~Guard() { for (void** p : ptrs) if (*p) (void)hipFree(*p); }
  } guard;
  guard.ptrs = {(void**)&d_pts, (void**)&d_planes, (void**)&d_lab, (void**)&d_part};
  SSLAM_HIP_TRY(hipMalloc((void**)&d_pts, (size_t)std::max(n, 1) * 3 * sizeof(float)));
  SSLAM_HIP_TRY(hipMalloc((void**)&d_lab, (size_t)std::max(n, 1) * sizeof(int)));
  SSLAM_HIP_TRY(hipMalloc((void**)&d_planes, (size_t)n_planes * 4 * sizeof(float)));
  SSLAM_HIP_TRY(hipMalloc((void**)&d_part, (size_t)nblk * kIcpSums * sizeof(double)));
  if (n > 0) {
    SSLAM_HIP_TRY(hipMemcpyAsync(d_pts, xyz, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, s->stream));
    SSLAM_HIP_TRY(hipMemcpyAsync(d_lab, labels, (size_t)n * sizeof(int), hipMemcpyHostToDevice, s->stream));
  }
  SSLAM_HIP_TRY(hipMemcpyAsync(d_planes, planes, (size_t)n_planes * 4 * sizeof(float), hipMemcpyHostToDevice, s->stream));
  std::vector<double> part((size_t)nblk * kIcpSums);
  double sums[kIcpSums];
  auto accumulate = [&]() -> int {
    hipLaunchKernelGGL(k_icp_accumulate, dim3(nblk), dim3(256), 0, s->stream, d_pts, d_lab, n, d_planes, n_planes, T, d_part);
    SSLAM_HIP_TRY(hipGetLastError());
    SSLAM_HIP_TRY(hipMemcpyAsync(part.data(), d_part, part.size() * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    SSLAM_HIP_TRY(hipStreamSynchronize(s->stream));
    for (int k = 0; k < kIcpSums; ++k) { double v = 0; for (int b = 0; b < nblk; ++b) v += part[(size_t)b * kIcpSums + k]; sums[k] = v; }
    return 0;
  };
  int rc;
  for (int it = 0; it < iterations; ++it) {
    if ((rc = accumulate())) return rc;
    if (sums[28] < 6) break;
    // (J^T J) delta = -J^T r by Cholesky; a rank-deficient plane set (fewer than three independent normals) leaves the transform
    // free along some direction: SSLAM_ERR_NUMERIC
    double A[36], bvec[6], L[36] = {0};
    int m = 0;
    for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c) { A[r * 6 + c] = A[c * 6 + r] = sums[m++]; }
    for (int r = 0; r < 6; ++r) bvec[r] = -sums[21 + r];
    for (int j = 0; j < 6; ++j) {
      double dsum = A[j * 6 + j];
      for (int k = 0; k < j; ++k) dsum -= L[j * 6 + k] * L[j * 6 + k];
      if (!(dsum > 1e-12 * A[j * 6 + j]) || !(dsum > 0)) return set_error(SSLAM_ERR_NUMERIC, "point-to-plane system is rank deficient (the planes do not constrain all six degrees of freedom)");
      L[j * 6 + j] = std::sqrt(dsum);
      for (int i = j + 1; i < 6; ++i) { double v = A[i * 6 + j]; for (int k = 0; k < j; ++k) v -= L[i * 6 + k] * L[j * 6 + k]; L[i * 6 + j] = v / L[j * 6 + j]; }
    }
    double y[6], dx[6];
    for (int i = 0; i < 6; ++i) { double v = bvec[i]; for (int k = 0; k < i; ++k) v -= L[i * 6 + k] * y[k]; y[i] = v / L[i * 6 + i]; }
    for (int i = 5; i >= 0; --i) { double v = y[i]; for (int k = i + 1; k < 6; ++k) v -= L[k * 6 + i] * dx[k]; dx[i] = v / L[i * 6 + i]; }
    // T <- (exp[w]x, u) o T   (Rodrigues)
    const double wx = dx[0], wy = dx[1], wz = dx[2], th = std::sqrt(wx * wx + wy * wy + wz * wz);
    double E[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (th > 0) {
      const double a = std::sin(th) / th, bq = (1.0 - std::cos(th)) / (th * th);
      const double K[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { double kk = 0; for (int q = 0; q < 3; ++q) kk += K[r * 3 + q] * K[q * 3 + c]; E[r * 3 + c] += a * K[r * 3 + c] + bq * kk; }
    }
    IcpPose Tn;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) Tn.R[r * 3 + c] = E[r * 3] * T.R[c] + E[r * 3 + 1] * T.R[3 + c] + E[r * 3 + 2] * T.R[6 + c];
      Tn.t[r] = E[r * 3] * T.t[0] + E[r * 3 + 1] * T.t[1] + E[r * 3 + 2] * T.t[2] + dx[3 + r];
    }
    T = Tn;
  }
  if ((rc = accumulate())) return rc;   // residual at the returned transform
  for (int k = 0; k < 9; ++k) T_out[k] = T.R[k];
  for (int k = 0; k < 3; ++k) T_out[9 + k] = T.t[k];
  if (rms_out) *rms_out = sums[28] > 0 ? std::sqrt(sums[27] / sums[28]) : 0.0;
  return (int)sums[28];
}


// ---- cloud filters ------------------------------------------------------------------------------------------------

int sslam_seg_distance_filter(sslam_seg* s, const float* xyz, int n, double dmin, double dmax, int32_t* keep_out, int max_out) {
  if (!s || !xyz || n < 0 || (!keep_out && max_out > 0)) return set_error(SSLAM_ERR_INVALID, "bad argument");
  int rc = seg_device(s);
  if (rc) return rc;
  if (n == 0) return 0;
  DevGuard g;
  float* d_pts = nullptr; unsigned char* d_flag = nullptr; int* d_out = nullptr;
  if ((rc = g.alloc(&d_pts, (size_t)n * 3)) || (rc = g.alloc(&d_flag, n)) || (rc = g.alloc(&d_out, n))) return rc;
  SSLAM_HIP_TRY(hipMemcpyAsync(d_pts, xyz, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, s->stream));
  hipLaunchKernelGGL(k_range_flag, dim3((n + 255) / 256), dim3(256), 0, s->stream, d_pts, n, dmin, dmax, d_flag);
  int total = 0;
  if ((rc = compact_flags(s, g, d_flag, n, d_out, n, &total))) return rc;
  const int m = std::min(total, max_out);
  if (m > 0) SSLAM_HIP_TRY(hipMemcpy(keep_out, d_out, (size_t)m * sizeof(int), hipMemcpyDeviceToHost));
  return total;
}

int sslam_seg_voxel_grid(sslam_seg* s, const float* xyz, int n, float leaf, float* centroids_out, int32_t* counts_out, int max_out) {
  if (!s || !xyz || n < 0 || !(leaf > 0) || (!centroids_out && max_out > 0)) return set_error(SSLAM_ERR_INVALID, "bad argument");
  int rc = seg_device(s);
  if (rc) return rc;
  if (n == 0) return 0;
  DevGuard g;
  float *d_pts = nullptr, *d_part = nullptr;
  const int nblk = std::min(256, (n + 255) / 256);
  if ((rc = g.alloc(&d_pts, (size_t)n * 3)) || (rc = g.alloc(&d_part, (size_t)nblk * 6))) return rc;
  SSLAM_HIP_TRY(hipMemcpyAsync(d_pts, xyz, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, s->stream));
  hipLaunchKernelGGL(k_voxel_minmax, dim3(nblk), dim3(256), 0, s->stream, d_pts, n, d_part);
  std::vector<float> part((size_t)nblk * 6);
  SSLAM_HIP_TRY(hipMemcpyAsync(part.data(), d_part, part.size() * sizeof(float), hipMemcpyDeviceToHost, s->stream));
  SSLAM_HIP_TRY(hipStreamSynchronize(s->stream));
  float lo[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, hi[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
  for (int b = 0; b < nblk; ++b) for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], part[(size_t)b * 6 + k]); hi[k] = std::max(hi[k], part[(size_t)b * 6 + 3 + k]); }
  if (!(lo[0] <= hi[0])) return 0;   // no finite point
  // pcl::VoxelGrid::applyFilter: inverse leaf size, integer bounds of the box, cell strides
  VoxelGeom G;
  long long div[3];
  for (int k = 0; k < 3; ++k) {
    G.inv[k] = 1.0f / leaf;
    G.minb[k] = (int)std::floor(lo[k] * G.inv[k]);
    const int maxb = (int)std::floor(hi[k] * G.inv[k]);
    div[k] = (long long)maxb - G.minb[k] + 1;
  }
  const long long ncell = div[0] * div[1] * div[2];
  if (ncell > (1ll << 26)) return set_error(SSLAM_ERR_UNSUPPORTED, "voxel grid of %lld cells: leaf size too small for the extent of the cloud (PCL refuses as well)", ncell);
  G.mul[0] = 1; G.mul[1] = (int)div[0]; G.mul[2] = (int)(div[0] * div[1]);
  int* d_count = nullptr; long long* d_sums = nullptr; unsigned char* d_flag = nullptr; int* d_cells = nullptr;
  if ((rc = g.alloc(&d_count, (size_t)ncell)) || (rc = g.alloc(&d_sums, (size_t)ncell * 3)) || (rc = g.alloc(&d_flag, (size_t)ncell)) || (rc = g.alloc(&d_cells, (size_t)ncell))) return rc;
  SSLAM_HIP_TRY(hipMemsetAsync(d_count, 0, (size_t)ncell * sizeof(int), s->stream));
  SSLAM_HIP_TRY(hipMemsetAsync(d_sums, 0, (size_t)ncell * 3 * sizeof(long long), s->stream));
  hipLaunchKernelGGL(k_voxel_accumulate, dim3((n + 255) / 256), dim3(256), 0, s->stream, d_pts, n, G, d_count, d_sums);
  hipLaunchKernelGGL(k_voxel_flag, dim3((int)((ncell + 255) / 256)), dim3(256), 0, s->stream, d_count, (int)ncell, d_flag);
  int nocc = 0;
  if ((rc = compact_flags(s, g, d_flag, (int)ncell, d_cells, (int)ncell, &nocc))) return rc;
  if (nocc == 0) return 0;
  float* d_out = nullptr; int* d_oc = nullptr;
  if ((rc = g.alloc(&d_out, (size_t)nocc * 3)) || (rc = g.alloc(&d_oc, nocc))) return rc;
  hipLaunchKernelGGL(k_voxel_centroids, dim3((nocc + 255) / 256), dim3(256), 0, s->stream, d_cells, nocc, d_count, d_sums, d_out, d_oc);
  SSLAM_HIP_TRY(hipGetLastError());
  const int m = std::min(nocc, max_out);
  if (m > 0) {
    SSLAM_HIP_TRY(hipMemcpyAsync(centroids_out, d_out, (size_t)m * 3 * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    if (counts_out) SSLAM_HIP_TRY(hipMemcpyAsync(counts_out, d_oc, (size_t)m * sizeof(int), hipMemcpyDeviceToHost, s->stream));
  }
  SSLAM_HIP_TRY(hipStreamSynchronize(s->stream));
  return nocc;
}

int sslam_seg_statistical_outlier_removal(sslam_seg* s, const float* xyz, int n, int mean_k, double stddev_mul, int32_t* keep_out, int max_out,
                                          float* mean_dist_out) {
  if (!s || !xyz || n < 0 || mean_k < 1 || mean_k >= kSorMaxK || (!keep_out && max_out > 0)) return set_error(SSLAM_ERR_INVALID, "bad argument (mean_k must be in [1, %d))", kSorMaxK);
  int rc = seg_device(s);
  if (rc) return rc;
  if (n == 0) return 0;
  if (n < mean_k + 1) return set_error(SSLAM_ERR_INVALID, "fewer points (%d) than neighbours asked for (%d)", n, mean_k);
  DevGuard g;
  float *d_pts = nullptr, *d_md = nullptr;
  if ((rc = g.alloc(&d_pts, (size_t)n * 3)) || (rc = g.alloc(&d_md, n))) return rc;
  SSLAM_HIP_TRY(hipMemcpyAsync(d_pts, xyz, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, s->stream));
  const size_t lds = (size_t)(128 * 3 + (mean_k + 1) * 128) * sizeof(float);
  hipLaunchKernelGGL(k_sor_mean_distance, dim3((n + 127) / 128), dim3(128), lds, s->stream, d_pts, n, mean_k, d_md);
  SSLAM_HIP_TRY(hipGetLastError());
  std::vector<float> md(n);
  SSLAM_HIP_TRY(hipMemcpyAsync(md.data(), d_md, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, s->stream));
  SSLAM_HIP_TRY(hipStreamSynchronize(s->stream));
  // pcl::StatisticalOutlierRemoval::applyFilterIndices: mean and standard deviation of the mean distances in double, in index order
  double sum = 0, sq = 0;
  int valid = 0;
  for (int i = 0; i < n; ++i) if (md[i] >= 0) { sum += md[i]; sq += (double)md[i] * md[i]; ++valid; }
  if (mean_dist_out) for (int i = 0; i < n; ++i) mean_dist_out[i] = md[i];
  if (valid < 2) return 0;
  const double mean = sum / valid, variance = (sq - sum * sum / valid) / (valid - 1.0), stddev = std::sqrt(variance);
  const double thr = mean + stddev_mul * stddev;
  int kept = 0;
  for (int i = 0; i < n; ++i)
    if (md[i] >= 0 && !((double)md[i] > thr)) { if (kept < max_out) keep_out[kept] = i; ++kept; }
  return kept;
}


// cv::kmeans as plane_segmentation::computeKmeans configures it (10 attempts, <= 10 iterations or a centre shift <= 0.01, random
// centres), restated with (a) centres drawn uniformly in the bounding box from splitmix64(seed, attempt, centre, coordinate) instead
// of cv::RNG, (b) order-independent fixed-point sums in the centre update, (c) an empty cluster keeping its centre.
int sslam_seg_kmeans(sslam_seg* s, const float* pts, int n, int dim, int k, uint64_t seed, int32_t* labels_out, float* centers_out, double* compactness_out) {
  if (!s || !pts || !labels_out || !centers_out || n <= 0 || (dim != 1 && dim != 3) || k < 1 || k > kKmMaxK) return set_error(SSLAM_ERR_INVALID, "bad argument (dim 1 or 3, 1 <= k <= %d)", kKmMaxK);
  int rc = seg_device(s);
  if (rc) return rc;
  for (size_t t = 0; t < (size_t)n * dim; ++t) if (!std::isfinite(pts[t])) return set_error(SSLAM_ERR_INVALID, "k-means input holds a non-finite value (the reference removes them first, removeNans)");
  float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  for (int d = 0; d < dim; ++d) { lo[d] = hi[d] = pts[d]; }
  for (int i = 1; i < n; ++i) for (int d = 0; d < dim; ++d) { const float v = pts[(size_t)i * dim + d]; lo[d] = std::min(lo[d], v); hi[d] = std::max(hi[d], v); }
  DevGuard g;
  float* d_pts = nullptr; int *d_lab = nullptr, *d_cnt = nullptr; long long *d_sums = nullptr, *d_cmp = nullptr;
  if ((rc = g.alloc(&d_pts, (size_t)n * dim)) || (rc = g.alloc(&d_lab, n)) || (rc = g.alloc(&d_cnt, kKmMaxK)) || (rc = g.alloc(&d_sums, kKmMaxK * 3)) || (rc = g.alloc(&d_cmp, 1))) return rc;
  SSLAM_HIP_TRY(hipMemcpyAsync(d_pts, pts, (size_t)n * dim * sizeof(float), hipMemcpyHostToDevice, s->stream));
  auto mix = [](uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); };
  const int attempts = 10, max_count = 10;
  const double eps2 = 0.01 * 0.01;
  double best = 1.7976931348623157e308;
  std::vector<int32_t> lab(n);
  KmCenters C; C.k = k; C.dim = dim;
  long long h_sums[kKmMaxK * 3]; int h_cnt[kKmMaxK]; long long h_cmp = 0;
  bool have = false;
  for (int a = 0; a < attempts; ++a) {
    double max_shift = 1.7976931348623157e308, compact = 0;
    bool assigned = false;
    for (int iter = 0;;) {
      if (iter == 0) {
        for (int j = 0; j < k; ++j)
          for (int d = 0; d < 3; ++d) {
            const float u = (float)(mix(seed ^ ((uint64_t)a << 40) ^ ((uint64_t)j << 20) ^ (uint64_t)d) >> 40) / 16777216.0f;
            C.c[j * 3 + d] = d < dim ? lo[d] + u * (hi[d] - lo[d]) : 0.0f;
          }
      } else {   // means of the clusters of the last assignment
        max_shift = 0;
        for (int j = 0; j < k; ++j) {
          double sh = 0;
          for (int d = 0; d < dim; ++d) {
            const float old = C.c[j * 3 + d];
            if (h_cnt[j] > 0) C.c[j * 3 + d] = (float)(((double)h_sums[j * 3 + d] / 1048576.0) / (double)h_cnt[j]);
            const double t = (double)C.c[j * 3 + d] - (double)old;
            sh += t * t;
          }
          max_shift = std::max(max_shift, sh);
        }
      }
      if (++iter == std::max(max_count, 2) || max_shift <= eps2) break;
      SSLAM_HIP_TRY(hipMemsetAsync(d_sums, 0, sizeof(long long) * kKmMaxK * 3, s->stream));
      SSLAM_HIP_TRY(hipMemsetAsync(d_cnt, 0, sizeof(int) * kKmMaxK, s->stream));
      SSLAM_HIP_TRY(hipMemsetAsync(d_cmp, 0, sizeof(long long), s->stream));
      hipLaunchKernelGGL(k_kmeans_assign, dim3((n + 255) / 256), dim3(256), 0, s->stream, d_pts, n, C, d_lab, d_sums, d_cnt, d_cmp);
      SSLAM_HIP_TRY(hipGetLastError());
      SSLAM_HIP_TRY(hipMemcpyAsync(h_sums, d_sums, sizeof h_sums, hipMemcpyDeviceToHost, s->stream));
      SSLAM_HIP_TRY(hipMemcpyAsync(h_cnt, d_cnt, sizeof h_cnt, hipMemcpyDeviceToHost, s->stream));
      SSLAM_HIP_TRY(hipMemcpyAsync(&h_cmp, d_cmp, sizeof h_cmp, hipMemcpyDeviceToHost, s->stream));
      SSLAM_HIP_TRY(hipStreamSynchronize(s->stream));
      compact = (double)h_cmp / 1048576.0;
      assigned = true;
    }
    if (assigned && compact < best) {   // cv::kmeans keeps the labels of the last assignment and the centres updated after it
      best = compact;
      SSLAM_HIP_TRY(hipMemcpy(lab.data(), d_lab, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
      for (int j = 0; j < k; ++j) for (int d = 0; d < dim; ++d) centers_out[j * dim + d] = C.c[j * 3 + d];
      have = true;
    }
  }
  if (!have) return set_error(SSLAM_ERR_NUMERIC, "k-means made no assignment");
  for (int i = 0; i < n; ++i) labels_out[i] = lab[i];
  if (compactness_out) *compactness_out = best;
  return k;
}

}  // extern "C"
