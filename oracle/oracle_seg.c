/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded CPU restatement of the reference's planar-segmentation frontend:
 *   point_cloud_segmentation::segmentallPointCloudData   reference include/planar_segmentation/point_cloud_segmentation.h:105-181
 *   plane_segmentation::segmentPointCloudData (crop)     reference src/planar_segmentation/plane_segmentation.cpp:24-82
 *   plane_segmentation::computeNormalsFromPointCloud     reference src/planar_segmentation/plane_segmentation.cpp:84-106
 *   plane_segmentation::multiPlaneSegmentation           reference src/planar_segmentation/plane_segmentation.cpp:108-259
 *   point_cloud_segmentation::segmentPlanarSurfaces      reference include/planar_segmentation/point_cloud_segmentation.h:26-103
 *   semantic_tools::transformNormalsToWorld              reference include/tools.h:18-102 (incl. the typo at :80-81, quirk B2)
 *
 * PARITY UNPINNED: the arithmetic of the two PCL calls lives in PCL ("PCL 1.7", reference
 * CMakeLists.txt:22-23), which is neither vendored nor installed here, and the reference has no
 * tests.  The PCL algorithms below are restated from their published implementations
 * (pcl/features/integral_image_normal.hpp, pcl/features/integral_image2D.hpp, pcl/common/eigen.hpp,
 * pcl/segmentation/organized_connected_component_segmentation.hpp,
 * pcl/segmentation/organized_multi_plane_segmentation.hpp, plane_coefficient_comparator.h,
 * plane_refinement_comparator.h, pcl/geometry/polygon_operations.h — PCL 1.7 semantics), from memory:
 *   - IntegralImageNormalEstimation, COVARIANCE_MATRIX, BORDER_POLICY_IGNORE, no depth-dependent
 *     smoothing: depth-change map, two-pass chamfer distance map (float, incl. its row wrap-around
 *     reads), double integral images of x,y,z and of the float products, per-pixel float covariance,
 *     pcl::eigen33 (closed-form roots), flip towards the origin, curvature
 *   - OrganizedMultiPlaneSegmentation::segmentAndRefine: plane_d = p.n, PlaneCoefficientComparator
 *     (depth-dependent distance threshold: segment() passes `true`), two-pass connected components,
 *     float mean/covariance per label (> min_inliers), eigen33, curvature gate, PlaneRefinementComparator
 *     two-sweep growth, Moore boundary trace from the LAST inlier index, pcl::calculatePolygonArea
 * One deliberate deviation (documented in DESIGN.md): the three trigonometric calls of
 * pcl::computeRoots are evaluated in double and rounded to float, so that CPU (glibc) and GPU (ocml)
 * agree bit-for-bit; PCL evaluates them in float.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <time.h>

typedef struct {
  double num_point_seg, norm_point_thres, planar_area;
  float max_depth_change_factor, normal_smoothing_size, angular_threshold, distance_threshold, maximum_curvature;
  int min_contour_points, image_width, image_height, reference_quirks, device;
} os_params;

typedef struct { int32_t tl_x, tl_y, width, height, class_id; float prob; } os_box;

typedef struct {
  float centroid_cam[3], normal_d[4], world_pose[3], num_points, prob;
  int32_t plane_type, class_id, box_index, inlier_count;
  float area;
} os_plane;

static const float kNaN = NAN;

/* ---------------------------------------------------------------- tools.h:18-102 */
static void mat4_mul(const float *A, const float *B, float *C) {
  float T[16];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      float s = 0;
      for (int k = 0; k < 4; ++k) s += A[r * 4 + k] * B[k * 4 + c];
      T[r * 4 + c] = s;
    }
  memcpy(C, T, sizeof T);
}
void os_transform_normals_to_world(const float pose[6], float cam_pitch, int quirks, float out[16]) {
  float rxc[16] = {0}, rxr[16] = {0}, rzr[16] = {0}, T[16] = {0};
  const double roll = pose[3], pitch = pose[4], yaw = pose[5];
  const double a = -(double)cam_pitch;
  rxc[0] = 1; rxc[5] = (float)cos(a); rxc[6] = (float)-sin(a); rxc[9] = (float)sin(a); rxc[10] = (float)cos(a); rxc[15] = 1;
  rxr[0] = 1; rxr[5] = (float)cos(-1.5708); rxr[6] = (float)-sin(-1.5708); rxr[9] = (float)sin(-1.5708); rxr[10] = (float)cos(-1.5708); rxr[15] = 1;
  rzr[0] = (float)cos(-1.5708); rzr[1] = (float)-sin(-1.5708); rzr[4] = (float)sin(-1.5708); rzr[5] = (float)cos(-1.5708); rzr[10] = 1; rzr[15] = 1;
  T[0] = (float)(cos(yaw) * cos(pitch));
  T[1] = (float)(cos(yaw) * sin(pitch) * sin(roll) - sin(yaw) * cos(roll));
  T[2] = (float)(cos(yaw) * sin(pitch) * cos(roll) + sin(yaw) * (quirks ? sin(pitch) : sin(roll)));  /* tools.h:80-81 typo */
  T[4] = (float)(sin(yaw) * cos(pitch));
  T[5] = (float)(sin(yaw) * sin(pitch) * sin(roll) + cos(yaw) * cos(roll));
  T[6] = (float)(sin(yaw) * sin(pitch) * cos(roll) - cos(yaw) * sin(roll));
  T[8] = (float)(-sin(pitch)); T[9] = (float)(cos(pitch) * sin(roll)); T[10] = (float)(cos(pitch) * cos(roll)); T[15] = 1;
  float M[16];
  mat4_mul(T, rzr, M); mat4_mul(M, rxr, M); mat4_mul(M, rxc, out);
}

/* ---------------------------------------------------------------- pcl::eigen33 (smallest eigenpair, float) */
static void roots2(float b, float c, float r[3]) {
  r[0] = 0.0f;
  float d = b * b - 4.0f * c;
  if (d < 0.0f) d = 0.0f;
  float sd = sqrtf(d);
  r[2] = 0.5f * (b + sd);
  r[1] = 0.5f * (b - sd);
}
static void compute_roots(const float m[9], float r[3]) {
  const float m00 = m[0], m01 = m[1], m02 = m[2], m11 = m[4], m12 = m[5], m22 = m[8];
  float c0 = m00 * m11 * m22 + 2.0f * m01 * m02 * m12 - m00 * m12 * m12 - m11 * m02 * m02 - m22 * m01 * m01;
  float c1 = m00 * m11 - m01 * m01 + m00 * m22 - m02 * m02 + m11 * m22 - m12 * m12;
  float c2 = m00 + m11 + m22;
  if (fabsf(c0) < 1.1920929e-07f) { roots2(c2, c1, r); return; }
  const float s_inv3 = 1.0f / 3.0f;
  const float s_sqrt3 = sqrtf(3.0f);
  float c2_over_3 = c2 * s_inv3;
  float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
  if (a_over_3 > 0.0f) a_over_3 = 0.0f;
  float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));
  float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
  if (q > 0.0f) q = 0.0f;
  float rho = sqrtf(-a_over_3);
  float theta = (float)atan2((double)sqrtf(-q), (double)half_b) * s_inv3;
  float cos_theta = (float)cos((double)theta);
  float sin_theta = (float)sin((double)theta);
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
static void cross3(const float *a, const float *b, float *o) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
void os_eigen33(const float mat[9], float *eigenvalue, float vec[3]) {
  float scale = 0;
  for (int k = 0; k < 9; ++k) { float a = fabsf(mat[k]); if (a > scale) scale = a; }
  if (scale <= 1.17549435e-38f) scale = 1.0f;
  float s[9];
  for (int k = 0; k < 9; ++k) s[k] = mat[k] / scale;
  float r[3];
  compute_roots(s, r);
  *eigenvalue = r[0] * scale;
  s[0] -= r[0]; s[4] -= r[0]; s[8] -= r[0];
  float v1[3], v2[3], v3[3];
  cross3(s + 0, s + 3, v1); cross3(s + 0, s + 6, v2); cross3(s + 3, s + 6, v3);
  float l1 = v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2];
  float l2 = v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2];
  float l3 = v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2];
  const float *v; float l;
  if (l1 >= l2 && l1 >= l3) { v = v1; l = l1; }
  else if (l2 >= l1 && l2 >= l3) { v = v2; l = l2; }
  else { v = v3; l = l3; }
  float n = sqrtf(l);
  vec[0] = v[0] / n; vec[1] = v[1] / n; vec[2] = v[2] / n;
}

/* ---------------------------------------------------------------- IntegralImageNormalEstimation */
/* pts: w*h*3 floats (x,y,z); normals: w*h*4 floats (nx,ny,nz,curvature), NaN where undefined.
 * dist_out (optional): w*h floats, the chamfer distance map (test hook). */
void os_normals(const float *pts, int w, int h, float max_depth_change_factor, float smoothing_size, float *normals, float *dist_out) {
  const size_t n = (size_t)w * h;
  for (size_t i = 0; i < n * 4; ++i) normals[i] = kNaN;
  /* integral images (w+1) x (h+1): first order (3, double), second order (6, double), finite count */
  const int W1 = w + 1;
  double *fo = (double *)calloc((size_t)W1 * (h + 1) * 3, sizeof(double));
  double *so = (double *)calloc((size_t)W1 * (h + 1) * 6, sizeof(double));
  unsigned *cnt = (unsigned *)calloc((size_t)W1 * (h + 1), sizeof(unsigned));
  for (int r = 0; r < h; ++r) {
    const double *pf = fo + (size_t)r * W1 * 3; double *cf = fo + (size_t)(r + 1) * W1 * 3;
    const double *ps = so + (size_t)r * W1 * 6; double *cs = so + (size_t)(r + 1) * W1 * 6;
    const unsigned *pc = cnt + (size_t)r * W1; unsigned *cc = cnt + (size_t)(r + 1) * W1;
    for (int c = 0; c < w; ++c) {
      for (int k = 0; k < 3; ++k) cf[(c + 1) * 3 + k] = pf[(c + 1) * 3 + k] + cf[c * 3 + k] - pf[c * 3 + k];
      for (int k = 0; k < 6; ++k) cs[(c + 1) * 6 + k] = ps[(c + 1) * 6 + k] + cs[c * 6 + k] - ps[c * 6 + k];
      cc[c + 1] = pc[c + 1] + cc[c] - pc[c];
      const float *e = pts + ((size_t)r * w + c) * 3;
      if (isfinite(e[0] + e[1] + e[2])) {
        for (int k = 0; k < 3; ++k) cf[(c + 1) * 3 + k] += (double)e[k];
        ++cc[c + 1];
        int el = 0;
        for (int a = 0; a < 3; ++a)
          for (int b = a; b < 3; ++b, ++el) cs[(c + 1) * 6 + el] += (double)(e[a] * e[b]);  /* float product */
      }
    }
  }
  /* depth-change map */
  unsigned char *dcm = (unsigned char *)malloc(n);
  memset(dcm, 255, n);
  for (int r = 0; r < h - 1; ++r)
    for (int c = 0; c < w - 1; ++c) {
      const size_t i = (size_t)r * w + c;
      const float d = pts[i * 3 + 2], dR = pts[(i + 1) * 3 + 2], dD = pts[(i + w) * 3 + 2];
      const float thr = max_depth_change_factor * (fabsf(d) + 1.0f) * 2.0f;
      if (fabsf(d - dR) > thr || !isfinite(d) || !isfinite(dR)) { dcm[i] = 0; dcm[i + 1] = 0; }
      if (fabsf(d - dD) > thr || !isfinite(d) || !isfinite(dD)) { dcm[i] = 0; dcm[i + w] = 0; }
    }
  /* two-pass chamfer distance map (float) */
  float *dm = (float *)malloc((n + 1) * sizeof(float));
  for (size_t i = 0; i < n; ++i) dm[i] = dcm[i] == 0 ? 0.0f : (float)(w + h);
  dm[n] = 0;
  {
    float *prev = dm, *cur = dm + w;
    for (int r = 1; r < h; ++r) {
      for (int c = 1; c < w; ++c) {
        const float upLeft = prev[c - 1] + 1.4f, up = prev[c] + 1.0f, upRight = prev[c + 1] + 1.4f, left = cur[c - 1] + 1.0f;
        const float center = cur[c];
        const float m = fminf(fminf(upLeft, up), fminf(left, upRight));
        if (m < center) cur[c] = m;
      }
      prev = cur; cur += w;
    }
    float *next = dm + (size_t)w * (h - 1);
    cur = next - w;
    for (int r = h - 2; r >= 0; --r) {
      for (int c = w - 2; c >= 0; --c) {
        const float lowerLeft = next[c - 1] + 1.4f;  /* c == 0 reads cur[w-1] (PCL's wrap-around) */
        const float lower = next[c] + 1.0f, lowerRight = next[c + 1] + 1.4f, right = cur[c + 1] + 1.0f;
        const float center = cur[c];
        const float m = fminf(fminf(lowerLeft, lower), fminf(right, lowerRight));
        if (m < center) cur[c] = m;
      }
      next = cur; cur -= w;
    }
  }
  if (dist_out) memcpy(dist_out, dm, n * sizeof(float));
  /* per-pixel normals, BORDER_POLICY_IGNORE */
  const int border = (int)smoothing_size;
  for (int r = border; r < h - border; ++r)
    for (int c = border; c < w - border; ++c) {
      const size_t i = (size_t)r * w + c;
      const float depth = pts[i * 3 + 2];
      if (!isfinite(depth)) continue;
      const float smoothing = fminf(dm[i], smoothing_size);
      if (!(smoothing > 2.0f)) continue;
      const int rw = (int)smoothing, rh = (int)smoothing;
      const int sx = c - rw / 2, sy = r - rh / 2;
      const size_t ul = (size_t)sy * W1 + sx, ur = ul + rw, ll = (size_t)(sy + rh) * W1 + sx, lr = ll + rw;
      const unsigned count = cnt[lr] + cnt[ul] - cnt[ur] - cnt[ll];
      if (count == 0) continue;
      float cen[3], cov[9], sov[6];
      for (int k = 0; k < 3; ++k) cen[k] = (float)(fo[lr * 3 + k] + fo[ul * 3 + k] - fo[ur * 3 + k] - fo[ll * 3 + k]);
      for (int k = 0; k < 6; ++k) sov[k] = (float)(so[lr * 6 + k] + so[ul * 6 + k] - so[ur * 6 + k] - so[ll * 6 + k]);
      cov[0] = sov[0]; cov[1] = cov[3] = sov[1]; cov[2] = cov[6] = sov[2]; cov[4] = sov[3]; cov[5] = cov[7] = sov[4]; cov[8] = sov[5];
      const float fc = (float)count;
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) cov[a * 3 + b] -= (cen[a] * cen[b]) / fc;
      float ev, v[3];
      os_eigen33(cov, &ev, v);
      /* flipNormalTowardsViewpoint(point, 0,0,0) */
      const float vx = 0.0f - pts[i * 3 + 0], vy = 0.0f - pts[i * 3 + 1], vz = 0.0f - pts[i * 3 + 2];
      const float ct = vx * v[0] + vy * v[1] + vz * v[2];
      if (ct < 0) { v[0] *= -1; v[1] *= -1; v[2] *= -1; }
      normals[i * 4 + 0] = v[0]; normals[i * 4 + 1] = v[1]; normals[i * 4 + 2] = v[2];
      normals[i * 4 + 3] = ev > 0.0f ? fabsf(ev / (cov[0] + cov[4] + cov[8])) : 0.0f;
    }
  free(fo); free(so); free(cnt); free(dcm); free(dm);
}

/* ---------------------------------------------------------------- OrganizedMultiPlaneSegmentation::segmentAndRefine */
typedef struct {
  float centroid[3], model[4];
  int inliers;        /* inlier count after refinement */
  int last_inlier;    /* last appended inlier index (boundary trace start) */
  int first_inlier;
  int label;          /* compact connected-component label of the region */
} os_region;

static unsigned find_root(const unsigned *runs, unsigned i) {
  while (runs[i] != i) i = runs[i];
  return i;
}

/* PlaneCoefficientComparator::compare(idx1, idx2), depth dependent */
static int coeff_compare(const float *pts, const float *nrm, const float *pd, float dist_thr, float ang_thr, size_t i1, size_t i2) {
  const float z = pts[i1 * 3 + 2];
  const float thr = dist_thr * (z * z);
  const float dot = nrm[i1 * 4 + 0] * nrm[i2 * 4 + 0] + nrm[i1 * 4 + 1] * nrm[i2 * 4 + 1] + nrm[i1 * 4 + 2] * nrm[i2 * 4 + 2];
  return (fabsf(pd[i1] - pd[i2]) < thr) && (dot > ang_thr);
}

/* labels_out: w*h int32, -1 where no accepted plane, else region index (order of `regions`).
 * cc_labels_out (optional): compact connected-component labels before refinement (-1 invalid).
 * contour_out: boundary indices of every region concatenated; contour_ptr[nregions+1]. Returns #regions. */
int os_multi_plane(const float *pts, const float *nrm, int w, int h, unsigned min_inliers, float angular_threshold, float distance_threshold,
                   float maximum_curvature, os_region *regions, int max_regions, int32_t *labels_out, int32_t *cc_labels_out,
                   int32_t *contour_out, int32_t *contour_ptr, int max_contour) {
  const size_t n = (size_t)w * h;
  const float ang_thr = cosf(angular_threshold);
  float *pd = (float *)malloc(n * sizeof(float));
  for (size_t i = 0; i < n; ++i) pd[i] = pts[i * 3] * nrm[i * 4] + pts[i * 3 + 1] * nrm[i * 4 + 1] + pts[i * 3 + 2] * nrm[i * 4 + 2];
  const unsigned INV = 0xffffffffu;
  unsigned *lab = (unsigned *)malloc(n * sizeof(unsigned));
  for (size_t i = 0; i < n; ++i) lab[i] = INV;
  unsigned *runs = (unsigned *)malloc((n + 1) * sizeof(unsigned));
  unsigned clust = 0;
#define CMP(a, b) coeff_compare(pts, nrm, pd, distance_threshold, ang_thr, (a), (b))
  if (isfinite(pts[0])) { lab[0] = clust; runs[clust] = clust; ++clust; }
  for (int c = 1; c < w; ++c) {
    if (!isfinite(pts[(size_t)c * 3])) continue;
    if (CMP((size_t)c, (size_t)c - 1)) lab[c] = lab[c - 1];
    else { lab[c] = clust; runs[clust] = clust; ++clust; }
  }
  for (int r = 1; r < h; ++r) {
    const size_t cur = (size_t)r * w, prev = cur - w;
    if (isfinite(pts[cur * 3])) {
      if (CMP(cur, prev)) lab[cur] = lab[prev];
      else { lab[cur] = clust; runs[clust] = clust; ++clust; }
    }
    for (int c = 1; c < w; ++c) {
      const size_t i = cur + c;
      if (!isfinite(pts[i * 3])) continue;
      if (CMP(i, i - 1)) lab[i] = lab[i - 1];
      if (CMP(i, prev + c)) {
        if (lab[i] == INV) lab[i] = lab[prev + c];
        else if (lab[prev + c] != INV) {
          const unsigned r1 = find_root(runs, lab[i]), r2 = find_root(runs, lab[prev + c]);
          if (r1 < r2) runs[r2] = r1; else runs[r1] = r2;
        }
      }
      if (lab[i] == INV) { lab[i] = clust; runs[clust] = clust; ++clust; }
    }
  }
#undef CMP
  /* second pass: compact ids in order of root provisional id */
  unsigned *map = (unsigned *)malloc((clust + 1) * sizeof(unsigned));
  unsigned max_id = 0;
  for (unsigned k = 0; k < clust; ++k) {
    if (runs[k] == k) map[k] = max_id++;
    else map[k] = map[find_root(runs, k)];
  }
  unsigned *lcount = (unsigned *)calloc(max_id + 1, sizeof(unsigned));
  for (size_t i = 0; i < n; ++i)
    if (lab[i] != INV) { lab[i] = map[lab[i]]; lcount[lab[i]]++; }
  if (cc_labels_out) for (size_t i = 0; i < n; ++i) cc_labels_out[i] = lab[i] == INV ? -1 : (int32_t)lab[i];
  /* per-label plane fit (float accumulation in index order) */
  int *label_to_model = (int *)malloc((max_id + 1) * sizeof(int));
  char *grow = (char *)calloc(max_id + 1, 1);
  float (*acc)[9] = (float (*)[9])calloc(max_id + 1, sizeof(float[9]));
  int *first = (int *)malloc((max_id + 1) * sizeof(int)), *last = (int *)malloc((max_id + 1) * sizeof(int));
  for (unsigned k = 0; k <= max_id; ++k) { first[k] = -1; last[k] = -1; label_to_model[k] = 0; }
  for (size_t i = 0; i < n; ++i) {
    if (lab[i] == INV) continue;
    const unsigned l = lab[i];
    if (!(lcount[l] > min_inliers)) continue;
    const float x = pts[i * 3], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
    float *a = acc[l];
    a[0] += x * x; a[1] += x * y; a[2] += x * z; a[3] += y * y; a[4] += y * z; a[5] += z * z; a[6] += x; a[7] += y; a[8] += z;
    if (first[l] < 0) first[l] = (int)i;
    last[l] = (int)i;
  }
  int nreg = 0;
  for (unsigned l = 0; l < max_id; ++l) {
    if (!(lcount[l] > min_inliers)) continue;
    float a[9];
    const float cntf = (float)lcount[l];
    for (int k = 0; k < 9; ++k) a[k] = acc[l][k] / cntf;
    float cov[9];
    cov[0] = a[0] - a[6] * a[6]; cov[1] = a[1] - a[6] * a[7]; cov[2] = a[2] - a[6] * a[8];
    cov[4] = a[3] - a[7] * a[7]; cov[5] = a[4] - a[7] * a[8]; cov[8] = a[5] - a[8] * a[8];
    cov[3] = cov[1]; cov[6] = cov[2]; cov[7] = cov[5];
    float ev, v[3];
    os_eigen33(cov, &ev, v);
    float p[4] = {v[0], v[1], v[2], 0};
    p[3] = -1 * (p[0] * a[6] + p[1] * a[7] + p[2] * a[8]);
    const float ct = (0.0f - a[6]) * p[0] + (0.0f - a[7]) * p[1] + (0.0f - a[8]) * p[2];
    if (ct < 0) {
      p[0] *= -1; p[1] *= -1; p[2] *= -1;
      p[3] = -1 * (p[0] * a[6] + p[1] * a[7] + p[2] * a[8]);
    }
    const float es = cov[0] + cov[4] + cov[8];
    const float curv = es != 0 ? fabsf(ev / es) : 0;
    if (curv < maximum_curvature && nreg < max_regions) {
      os_region *R = &regions[nreg];
      R->centroid[0] = a[6]; R->centroid[1] = a[7]; R->centroid[2] = a[8];
      memcpy(R->model, p, sizeof p);
      R->inliers = (int)lcount[l]; R->first_inlier = first[l]; R->last_inlier = last[l]; R->label = (int)l;
      label_to_model[l] = nreg; grow[l] = 1;
      ++nreg;
    }
  }
  /* refine(): two sweeps with the PlaneRefinementComparator */
#define RCOMPARE(i1, i2, res)                                                                          \
  do {                                                                                                 \
    const unsigned cl_ = lab[i1], nl_ = lab[i2];                                                       \
    res = 0;                                                                                           \
    if (grow[cl_] && !grow[nl_]) {                                                                     \
      const float *m_ = regions[label_to_model[cl_]].model;                                            \
      const double d_ = fabs((double)(m_[0] * pts[(i2) * 3] + m_[1] * pts[(i2) * 3 + 1] + m_[2] * pts[(i2) * 3 + 2] + m_[3])); \
      const float z_ = pts[(i1) * 3 + 2];                                                              \
      const float t_ = distance_threshold * (z_ * z_);                                                 \
      res = d_ < (double)t_;                                                                           \
    }                                                                                                  \
  } while (0)
  for (int r = 0; r < h - 1; ++r) {
    const size_t cur = (size_t)r * w, nxt = cur + w;
    for (int c = 0; c < w - 1; ++c) {
      const int cl = (int)lab[cur + c], rl = (int)lab[cur + c + 1];
      if (cl < 0 || rl < 0) continue;
      int ok;
      RCOMPARE(cur + c, cur + c + 1, ok);
      if (ok) { lab[cur + c + 1] = (unsigned)cl; os_region *R = &regions[label_to_model[cl]]; R->inliers++; R->last_inlier = (int)(cur + c + 1); }
      const int ll = (int)lab[nxt + c];
      if (ll < 0) continue;
      RCOMPARE(cur + c, nxt + c, ok);
      if (ok) { lab[nxt + c] = (unsigned)cl; os_region *R = &regions[label_to_model[cl]]; R->inliers++; R->last_inlier = (int)(nxt + c); }
    }
  }
  {
    size_t cur = (size_t)w * (h - 1), prv = cur - w;
    for (int r = 0; r < h - 1; ++r, cur = prv, prv -= w) {
      for (int c = w - 1; c >= 0; --c) {
        /* PCL reads colIdx-1 even at column 0 (the previous row's last pixel).  Restated with no left
         * neighbour at column 0: a deliberate, documented deviation that keeps rows independent. */
        const int cl = (int)lab[cur + c];
        if (cl < 0) continue;
        int ok;
        if (c >= 1) {
          const int ll = (int)lab[cur + c - 1];
          if (ll < 0) continue;
          RCOMPARE(cur + c, cur + c - 1, ok);
          if (ok) { lab[cur + c - 1] = (unsigned)cl; os_region *R = &regions[label_to_model[cl]]; R->inliers++; R->last_inlier = (int)(cur + c - 1); }
        }
        const int ul = (int)lab[prv + c];
        if (ul < 0) continue;
        RCOMPARE(cur + c, prv + c, ok);
        if (ok) { lab[prv + c] = (unsigned)cl; os_region *R = &regions[label_to_model[cl]]; R->inliers++; R->last_inlier = (int)(prv + c); }
      }
    }
  }
#undef RCOMPARE
  for (size_t i = 0; i < n; ++i) labels_out[i] = (lab[i] != INV && grow[lab[i]]) ? label_to_model[lab[i]] : -1;
  /* boundary trace (findLabeledRegionBoundary) from the last inlier of every region */
  static const int dxs[8] = {-1, -1, 0, 1, 1, 1, 0, -1}, dys[8] = {0, -1, -1, -1, 0, 1, 1, 1};
  int cp = 0;
  contour_ptr[0] = 0;
  for (int k = 0; k < nreg; ++k) {
    const int start = regions[k].last_inlier;
    const unsigned label = lab[start];
    int dirn = -1, cx = start % w, cy = start / w, ci = start;
    for (int d = 0; d < 8; ++d) {
      const int x = cx + dxs[d], y = cy + dys[d], idx = ci + dys[d] * w + dxs[d];
      if (x >= 0 && x < w && y >= 0 && y < h && lab[idx] != label) { dirn = d; break; }
    }
    if (dirn != -1) {
      if (cp < max_contour) contour_out[cp] = start;
      ++cp;
      const long guard = 4L * (long)n + 8;
      long steps = 0;
      do {
        int nI = 0;
        for (int d = 1; d <= 8; ++d) {
          nI = (dirn + d) & 7;
          const int x = cx + dxs[nI], y = cy + dys[nI], idx = ci + dys[nI] * w + dxs[nI];
          if (x >= 0 && x < w && y >= 0 && y < h && lab[idx] == label) break;
        }
        dirn = (nI + 4) & 7;
        ci += dys[nI] * w + dxs[nI]; cx += dxs[nI]; cy += dys[nI];
        if (cp < max_contour) contour_out[cp] = ci;
        ++cp;
      } while (ci != start && ++steps < guard);
    }
    contour_ptr[k + 1] = cp < max_contour ? cp : max_contour;
  }
  free(pd); free(lab); free(runs); free(map); free(lcount); free(label_to_model); free(grow); free(acc); free(first); free(last);
  return nreg;
}

/* pcl::calculatePolygonArea over the contour points */
float os_polygon_area(const float *pts, const int32_t *idx, int n) {
  float res[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const int j = (i + 1) % n;
    const float *a = pts + (size_t)idx[i] * 3, *b = pts + (size_t)idx[j] * 3;
    float c[3];
    cross3(a, b, c);
    res[0] += c[0]; res[1] += c[1]; res[2] += c[2];
  }
  const float area = sqrtf(res[0] * res[0] + res[1] * res[1] + res[2] * res[2]);
  return area * 0.5f;
}

/* crop (plane_segmentation.cpp:24-82); returns 0 if the box is rejected ("spurious") */
int os_crop(const uint8_t *cloud, int point_step, int row_step, int ox, int oy, int oz, const os_box *b, int img_w, int img_h, float *pts) {
  if (b->height < 0 || b->width < 0 || b->tl_x < 0 || b->tl_y < 0 || (b->tl_x + b->width) > img_w || (b->tl_y + b->height) > img_h) return 0;
  for (int v = 0; v < b->height; ++v)
    for (int u = 0; u < b->width; ++u) {
      const size_t pos = (size_t)(b->tl_y + v) * row_step + (size_t)(b->tl_x + u) * point_step;
      float *o = pts + ((size_t)v * b->width + u) * 3;
      memcpy(o + 0, cloud + pos + ox, 4); memcpy(o + 1, cloud + pos + oy, 4); memcpy(o + 2, cloud + pos + oz, 4);
    }
  return 1;
}

static int class_whitelisted(int c) { return c >= 1 && c <= 7; } /* point_cloud_segmentation.h:126-130 */

/* Full per-frame entry: segmentallPointCloudData.  normals_out / labels_out (optional) receive the
 * per-box products packed back to back in box order (rejected / skipped boxes contribute nothing). */
int os_segment(const os_params *P, const uint8_t *cloud, int width, int height, int point_step, int row_step, int ox, int oy, int oz,
               const os_box *boxes, int nboxes, const float robot_pose[6], float cam_angle, os_plane *out, int max_out,
               float *normals_out, int32_t *labels_out) {
  (void)width; (void)height;
  float T[16];
  os_transform_normals_to_world(robot_pose, cam_angle, P->reference_quirks, T);
  float hz_cam[3] = {T[8], T[9], T[10]};  /* T^T * (0,0,1,0) */
  int nout = 0;
  size_t poff = 0;
  for (int bi = 0; bi < nboxes; ++bi) {
    const os_box *b = &boxes[bi];
    if (!class_whitelisted(b->class_id)) continue;
    const size_t npix = (size_t)(b->width > 0 ? b->width : 0) * (size_t)(b->height > 0 ? b->height : 0);
    float *pts = (float *)malloc((npix + 1) * 3 * sizeof(float));
    if (!os_crop(cloud, point_step, row_step, ox, oy, oz, b, P->image_width, P->image_height, pts)) { free(pts); continue; }
    if (npix == 0 || (double)npix < P->norm_point_thres) { free(pts); continue; }
    const int w = b->width, h = b->height;
    float *nrm = (float *)malloc(npix * 4 * sizeof(float));
    os_normals(pts, w, h, P->max_depth_change_factor, P->normal_smoothing_size, nrm, NULL);
    os_region regs[64];
    int32_t *lab = (int32_t *)malloc(npix * sizeof(int32_t));
    int32_t *contour = (int32_t *)malloc((4 * npix + 16) * sizeof(int32_t));
    int32_t cptr[65];
    const int nreg = os_multi_plane(pts, nrm, w, h, (unsigned)P->num_point_seg, P->angular_threshold, P->distance_threshold,
                                    P->maximum_curvature, regs, 64, lab, NULL, contour, cptr, (int)(4 * npix + 16));
    if (normals_out) memcpy(normals_out + poff * 4, nrm, npix * 4 * sizeof(float));
    if (labels_out) memcpy(labels_out + poff, lab, npix * sizeof(int32_t));
    poff += npix;
    for (int k = 0; k < nreg; ++k) {
      const int nc = cptr[k + 1] - cptr[k];
      if (!(nc > P->min_contour_points)) continue;       /* plane_segmentation.cpp:169 */
      const float *m = regs[k].model;
      const float dotp = hz_cam[0] * m[0] + hz_cam[1] * m[1] + hz_cam[2] * m[2];
      const float area = os_polygon_area(pts, contour + cptr[k], nc);
      if (!((double)area >= P->planar_area)) continue;    /* :195 */
      int type = -1;
      float sgn = 1.0f;
      if ((float)(fabsf(m[0]) - fabsf(hz_cam[0])) < 0.3 && (float)(fabsf(m[1]) - fabsf(hz_cam[1])) < 0.3 &&
          (float)(fabsf(m[2]) - fabsf(hz_cam[2])) < 0.3) {   /* :197-202 (quirk B7) */
        type = 0;
        if (m[1] > 0) sgn = -1.0f;
      } else if (dotp < 0.5) {                                  /* :226 */
        type = 1;
        if (m[0] > 0) sgn = -1.0f;
      }
      if (type < 0 || nout >= max_out) continue;
      os_plane *o = &out[nout++];
      memcpy(o->centroid_cam, regs[k].centroid, 12);
      for (int q = 0; q < 4; ++q) o->normal_d[q] = sgn < 0 ? -m[q] : m[q];
      /* point_cloud_segmentation.h:55-60,91-94 */
      float wp[3];
      for (int r = 0; r < 3; ++r) {
        float s = 0;
        for (int q = 0; q < 3; ++q) s += T[r * 4 + q] * regs[k].centroid[q];
        s += T[r * 4 + 3] * 1.0f;
        wp[r] = s;
      }
      o->world_pose[0] = wp[0] + robot_pose[0]; o->world_pose[1] = wp[1] + robot_pose[1]; o->world_pose[2] = wp[2] + robot_pose[2];
      o->num_points = (float)nc; o->prob = b->prob; o->plane_type = type; o->class_id = b->class_id; o->box_index = bi;
      o->inlier_count = regs[k].inliers; o->area = area;
    }
    free(pts); free(nrm); free(lab); free(contour);
  }
  return nout;
}

/* ---------------------------------------------------------------- RANSAC plane (row a15) */
/* pcl::SACSegmentation (SACMODEL_PLANE, SAC_RANSAC, optimize coefficients) as called by the reference's
 * dead code path plane_segmentation::compute2DConvexHull (src/planar_segmentation/plane_segmentation.cpp:639-647:
 * distance threshold 0.01, default max_iterations 50, probability 0.99).  Restated from
 * pcl/sample_consensus/{ransac.hpp, sac_model_plane.hpp} (from memory): adaptive iteration count k,
 * 3-point plane, |n.p + d| < threshold, refit by mean/covariance + eigen33, inliers re-selected with
 * the refined model.  DEVIATION: PCL draws its samples with rand(); here sample triples come from a
 * counter-based hash (splitmix64 of seed / iteration / attempt) so that the CPU oracle and the GPU
 * evaluate the very same hypotheses. */
static uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
/* returns 1 and fills model[4] if a good (non-collinear, distinct) sample exists for this iteration */
static int ransac_hypothesis(const float *pts, int n, uint64_t seed, int iter, float model[4]) {
  for (int attempt = 0; attempt < 1000; ++attempt) {
    int id[3];
    for (int j = 0; j < 3; ++j) id[j] = (int)(splitmix64(seed ^ ((uint64_t)iter << 32) ^ ((uint64_t)attempt << 8) ^ (uint64_t)j) % (uint64_t)n);
    if (id[0] == id[1] || id[0] == id[2] || id[1] == id[2]) continue;
    const float *p0 = pts + (size_t)id[0] * 3, *p1 = pts + (size_t)id[1] * 3, *p2 = pts + (size_t)id[2] * 3;
    const float a0 = p1[0] - p0[0], a1 = p1[1] - p0[1], a2 = p1[2] - p0[2];
    const float b0 = p2[0] - p0[0], b1 = p2[1] - p0[1], b2 = p2[2] - p0[2];
    const float r0 = a0 / b0, r1 = a1 / b1, r2 = a2 / b2;          /* isSampleGood: dy1dy2 */
    if (!((r0 != r1) || (r2 != r1))) continue;
    float m0 = a1 * b2 - a2 * b1, m1 = a2 * b0 - a0 * b2, m2 = a0 * b1 - a1 * b0;
    const float nn = sqrtf(m0 * m0 + m1 * m1 + m2 * m2);
    m0 /= nn; m1 /= nn; m2 /= nn;
    model[0] = m0; model[1] = m1; model[2] = m2;
    model[3] = -1 * (m0 * p0[0] + m1 * p0[1] + m2 * p0[2]);
    return isfinite(model[0]) && isfinite(model[3]);
  }
  return 0;
}
static int plane_inlier(const float *m, const float *p, float thr) {
  return fabsf(m[0] * p[0] + m[1] * p[1] + m[2] * p[2] + m[3]) < thr;
}
/* counts_out (optional): inlier count of every evaluated hypothesis (test hook). Returns #inliers. */
int os_ransac_plane(const float *pts, int n, float threshold, int max_iterations, double probability, uint64_t seed,
                    float coeff[4], int32_t *inliers, int max_inliers, int32_t *counts_out, int *best_iter_out) {
  coeff[0] = coeff[1] = coeff[2] = coeff[3] = 0;
  if (best_iter_out) *best_iter_out = -1;
  if (n < 3) return 0;
  int best = -1, best_it = -1;
  float best_model[4] = {0, 0, 0, 0};
  double k = 1.0;
  const double log_probability = log(1.0 - probability), one_over = 1.0 / (double)n;
  const double eps = 2.220446049250313e-16;
  int iterations = 0, skipped = 0, it = 0;
  const int max_skip = max_iterations * 10;
  while ((double)iterations < k && skipped < max_skip) {
    float m[4];
    const int ok = ransac_hypothesis(pts, n, seed, it, m);
    int cnt = 0;
    if (ok) for (int i = 0; i < n; ++i) cnt += plane_inlier(m, pts + (size_t)i * 3, threshold);
    if (counts_out && it < max_iterations + max_skip + 2) counts_out[it] = ok ? cnt : -1;
    ++it;
    if (!ok) { ++skipped; continue; }
    if (cnt > best) {
      best = cnt; best_it = it - 1; memcpy(best_model, m, sizeof m);
      const double w = (double)best * one_over;
      double p_no = 1.0 - pow(w, 3.0);
      if (p_no < eps) p_no = eps;
      if (p_no > 1.0 - eps) p_no = 1.0 - eps;
      k = log_probability / log(p_no);
    }
    ++iterations;
    if (iterations > max_iterations) break;
  }
  if (best_iter_out) *best_iter_out = best_it;
  if (best <= 0) return 0;
  /* optimizeModelCoefficients: mean/covariance of the inliers (float, index order) + eigen33 */
  float model[4];
  memcpy(model, best_model, sizeof model);
  if (best > 3) {
    float a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int cnt = 0;
    for (int i = 0; i < n; ++i) {
      const float *p = pts + (size_t)i * 3;
      if (!plane_inlier(best_model, p, threshold)) continue;
      a[0] += p[0] * p[0]; a[1] += p[0] * p[1]; a[2] += p[0] * p[2]; a[3] += p[1] * p[1]; a[4] += p[1] * p[2]; a[5] += p[2] * p[2];
      a[6] += p[0]; a[7] += p[1]; a[8] += p[2];
      ++cnt;
    }
    const float cf = (float)cnt;
    for (int q = 0; q < 9; ++q) a[q] = a[q] / cf;
    float cov[9];
    cov[0] = a[0] - a[6] * a[6]; cov[1] = a[1] - a[6] * a[7]; cov[2] = a[2] - a[6] * a[8];
    cov[4] = a[3] - a[7] * a[7]; cov[5] = a[4] - a[7] * a[8]; cov[8] = a[5] - a[8] * a[8];
    cov[3] = cov[1]; cov[6] = cov[2]; cov[7] = cov[5];
    float ev, v[3];
    os_eigen33(cov, &ev, v);
    model[0] = v[0]; model[1] = v[1]; model[2] = v[2];
    model[3] = -1 * (v[0] * a[6] + v[1] * a[7] + v[2] * a[8]);
  }
  memcpy(coeff, model, sizeof model);
  int ni = 0;
  for (int i = 0; i < n; ++i)
    if (plane_inlier(model, pts + (size_t)i * 3, threshold)) { if (ni < max_inliers) inliers[ni] = i; ++ni; }
  return ni;
}

/* ---------------------------------------------------------------------------------------------------
 * a15, second half: pcl::ProjectInliers (SACMODEL_PLANE) + pcl::ConvexHull (2-D) as compute2DConvexHull
 * configures them (reference plane_segmentation.cpp:648-662).  PCL / qhull are not in /root/reference:
 * restated from the published algorithms (PCL 1.7 sample_consensus/impl/sac_model_plane.hpp projectPoints,
 * surface/impl/convex_hull.hpp performReconstruction2D).  PARITY UNPINNED (no reference vectors).
 *   projectPoints : mc = (a,b,c,0) normalised in float; dist = mc.p + d (d NOT rescaled, as PCL);  p' = p - mc dist
 *   hull          : the coordinate plane is picked from the normal of the first / last / middle projected
 *                   points (|n.axis| > cos(10 deg) forbids the two planes containing that axis; order xy, yz, xz);
 *                   strictly convex vertices of the 2-D point set (qhull drops collinear points), ordered by
 *                   angle around the vertex centroid (comparePoints2D: atan2 ascending, -pi first) = the
 *                   counter-clockwise polygon started at its vertex of smallest angle.
 * Deviations (documented in DESIGN.md): if the three probe points are collinear PCL re-draws random points
 * (rand()); here the middle index walks forward until the triple is not collinear.  The centroid is the
 * double mean of the vertices in counter-clockwise chain order, rounded to float (qhull's vertex order is
 * not reproducible); the angular order uses an exact half-plane / cross-product comparator instead of atan2.
 * hull_out = positions in `inliers` order.  Returns the number of hull vertices (or -1: degenerate input). */
typedef struct { float x, y; int32_t i; } os_p2;
static int os_p2_cmp(const void *a, const void *b) {
  const os_p2 *p = (const os_p2 *)a, *q = (const os_p2 *)b;
  if (p->x != q->x) return p->x < q->x ? -1 : 1;
  if (p->y != q->y) return p->y < q->y ? -1 : 1;
  return p->i < q->i ? -1 : (p->i > q->i ? 1 : 0);
}
static double os_cross2(const os_p2 *o, const os_p2 *a, const os_p2 *b) {
  return ((double)a->x - (double)o->x) * ((double)b->y - (double)o->y) - ((double)a->y - (double)o->y) * ((double)b->x - (double)o->x);
}
static int os_ang_half(float x, float y) { return y < 0 ? 0 : ((y == 0 && x > 0) ? 1 : (y > 0 ? 2 : 3)); }
static float g_cx, g_cy;
static int os_ang_cmp(const void *a, const void *b) {
  const os_p2 *p = (const os_p2 *)a, *q = (const os_p2 *)b;
  const float px = p->x - g_cx, py = p->y - g_cy, qx = q->x - g_cx, qy = q->y - g_cy;
  const int hp = os_ang_half(px, py), hq = os_ang_half(qx, qy);
  if (hp != hq) return hp < hq ? -1 : 1;
  const double cr = (double)px * (double)qy - (double)py * (double)qx;
  if (cr != 0) return cr > 0 ? -1 : 1;
  return p->i < q->i ? -1 : (p->i > q->i ? 1 : 0);
}

void os_project_inliers(const float *pts, const int32_t *inliers, int n_in, const float coeff[4], float *proj) {
  const float nrm = sqrtf(coeff[0] * coeff[0] + coeff[1] * coeff[1] + coeff[2] * coeff[2]);
  const float mc[3] = {coeff[0] / nrm, coeff[1] / nrm, coeff[2] / nrm};
  for (int k = 0; k < n_in; ++k) {
    const float *p = pts + (size_t)inliers[k] * 3;
    const float dist = mc[0] * p[0] + mc[1] * p[1] + mc[2] * p[2] + coeff[3];
    proj[3 * k + 0] = p[0] - mc[0] * dist;
    proj[3 * k + 1] = p[1] - mc[1] * dist;
    proj[3 * k + 2] = p[2] - mc[2] * dist;
  }
}

/* axes_out: 0 = xy, 1 = yz, 2 = xz */
int os_hull_axes(const float *proj, int n_in) {
  if (n_in < 3) return -1;
  const float *p0 = proj, *p1 = proj + 3 * (size_t)(n_in - 1);
  double nx = 0, ny = 0, nz = 0, nn = 0;
  for (int m = n_in / 2, tries = 0; tries < n_in; ++tries, m = (m + 1) % n_in) {
    const float *p2 = proj + 3 * (size_t)m;
    const double ax = (double)p1[0] - p0[0], ay = (double)p1[1] - p0[1], az = (double)p1[2] - p0[2];
    const double bx = (double)p2[0] - p0[0], by = (double)p2[1] - p0[1], bz = (double)p2[2] - p0[2];
    nx = ay * bz - az * by; ny = az * bx - ax * bz; nz = ax * by - ay * bx;
    nn = sqrt(nx * nx + ny * ny + nz * nz);
    if (nn > 0) break;
  }
  if (!(nn > 0)) return -1;
  const float thresh = cosf(0.174532925f);
  const float tx = fabsf((float)(nx / nn)), ty = fabsf((float)(ny / nn)), tz = fabsf((float)(nz / nn));
  int xy = 1, yz = 1, xz = 1;
  if (tz > thresh) { xz = 0; yz = 0; }
  if (tx > thresh) { xz = 0; xy = 0; }
  if (ty > thresh) { xy = 0; yz = 0; }
  return xy ? 0 : (yz ? 1 : (xz ? 2 : -1));
}

int os_convex_hull_2d(const float *proj, int n_in, int32_t *hull_out, int max_hull, int *axes_out) {
  const int axes = os_hull_axes(proj, n_in);
  if (axes_out) *axes_out = axes;
  if (axes < 0) return -1;
  const int ia = axes == 1 ? 1 : 0, ib = axes == 0 ? 1 : 2;
  os_p2 *P = (os_p2 *)malloc(sizeof(os_p2) * (size_t)n_in), *Hh = (os_p2 *)malloc(sizeof(os_p2) * (size_t)(2 * n_in + 2));
  for (int k = 0; k < n_in; ++k) { P[k].x = proj[3 * k + ia]; P[k].y = proj[3 * k + ib]; P[k].i = k; }
  qsort(P, (size_t)n_in, sizeof(os_p2), os_p2_cmp);
  int m = 0;
  for (int k = 0; k < n_in; ++k) {              /* duplicates: the lowest index stays */
    if (m > 0 && P[k].x == P[m - 1].x && P[k].y == P[m - 1].y) continue;
    P[m++] = P[k];
  }
  int h = 0;
  for (int k = 0; k < m; ++k) {                 /* lower chain */
    while (h >= 2 && os_cross2(&Hh[h - 2], &Hh[h - 1], &P[k]) <= 0) --h;
    Hh[h++] = P[k];
  }
  for (int k = m - 2, t = h + 1; k >= 0; --k) { /* upper chain */
    while (h >= t && os_cross2(&Hh[h - 2], &Hh[h - 1], &P[k]) <= 0) --h;
    Hh[h++] = P[k];
  }
  if (m > 1) --h;                               /* the first point closes the loop */
  double sx = 0, sy = 0;
  for (int k = 0; k < h; ++k) { sx += Hh[k].x; sy += Hh[k].y; }
  g_cx = (float)(sx / h); g_cy = (float)(sy / h);
  /* angular order = the counter-clockwise chain rotated to start at the vertex of smallest angle */
  int first = 0;
  for (int k = 1; k < h; ++k) if (os_ang_cmp(&Hh[k], &Hh[first]) < 0) first = k;
  for (int k = 0; k < h && k < max_hull; ++k) hull_out[k] = Hh[(first + k) % h].i;
  free(P); free(Hh);
  return h;
}
