// Fixed-size FP64 geometry for the ps_graph_slam hot path on gfx950 (device + host).
//
// Restates the g2o types the reference instantiates:
//   VertexSE3 / EdgeSE3            <- reference src/ps_graph_slam/graph_slam.cpp:104-115,136-148
//   VertexPointXYZ / EdgeSE3PointXYZ (offset id 0 = identity, :75-83,150-166)
//   VertexPlane / EdgeSE3Plane     <- reference include/g2o/edge_se3_plane.hpp:8-48
// Poses are stored as translation + unit quaternion (x,y,z,w); increments are g2o's "MQT"
// minimal vectors [dt, dq_xyz] applied on the right:  X <- X * fromVectorMQT(d).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#define SSLAM_HD __host__ __device__ __forceinline__

namespace sslam {

struct Vec3 {
  double x, y, z;
};
struct Quat {
  double x, y, z, w;
};
struct Pose {
  Vec3 t;
  Quat q;
};

SSLAM_HD Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
SSLAM_HD Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
SSLAM_HD Vec3 operator*(double s, Vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
SSLAM_HD double dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
SSLAM_HD Vec3 cross(Vec3 a, Vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

SSLAM_HD Quat qmul(Quat a, Quat b) {
  return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
          a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w,
          a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
SSLAM_HD Quat qconj(Quat a) { return {-a.x, -a.y, -a.z, a.w}; }

// 3x3 rotation matrix, row-major m[r*3+c]
struct Mat3 {
  double m[9];
};
SSLAM_HD Mat3 qmat(Quat q) {
  Mat3 R;
  const double xx = q.x * q.x, yy = q.y * q.y, zz = q.z * q.z;
  const double xy = q.x * q.y, xz = q.x * q.z, yz = q.y * q.z;
  const double wx = q.w * q.x, wy = q.w * q.y, wz = q.w * q.z;
  R.m[0] = 1 - 2 * (yy + zz); R.m[1] = 2 * (xy - wz);     R.m[2] = 2 * (xz + wy);
  R.m[3] = 2 * (xy + wz);     R.m[4] = 1 - 2 * (xx + zz); R.m[5] = 2 * (yz - wx);
  R.m[6] = 2 * (xz - wy);     R.m[7] = 2 * (yz + wx);     R.m[8] = 1 - 2 * (xx + yy);
  return R;
}
SSLAM_HD Vec3 mul(const Mat3& R, Vec3 v) {
  return {R.m[0] * v.x + R.m[1] * v.y + R.m[2] * v.z, R.m[3] * v.x + R.m[4] * v.y + R.m[5] * v.z,
          R.m[6] * v.x + R.m[7] * v.y + R.m[8] * v.z};
}
SSLAM_HD Vec3 mulT(const Mat3& R, Vec3 v) {
  return {R.m[0] * v.x + R.m[3] * v.y + R.m[6] * v.z, R.m[1] * v.x + R.m[4] * v.y + R.m[7] * v.z,
          R.m[2] * v.x + R.m[5] * v.y + R.m[8] * v.z};
}
SSLAM_HD Vec3 qrot(Quat q, Vec3 v) { return mul(qmat(q), v); }

// VertexSE3::oplus
SSLAM_HD Pose se3_oplus(Pose X, const double d[6]) {
  const double w2 = 1.0 - (d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
  Quat dq;
  if (w2 < 0) dq = {0, 0, 0, 1};
  else dq = {d[3], d[4], d[5], sqrt(w2)};
  Pose Y;
  Y.t = X.t + qrot(X.q, Vec3{d[0], d[1], d[2]});
  Quat q = qmul(X.q, dq);
  const double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  Y.q = {q.x / n, q.y / n, q.z / n, q.w / n};
  return Y;
}

// ---------------------------------------------------------------------------------------------
// EdgeSE3: e = toVectorMQT(Z^-1 * Xi^-1 * Xj).  Intermediate terms kept for the Jacobians.
struct Se3Lin {
  double e[6];
  Mat3 Ra;   // rotation of Z^-1
  Mat3 Re;   // rotation of E
  Vec3 tb;   // translation of Xi^-1 Xj
  Quat qa;   // Z^-1 rotation
  Quat qb;   // rotation of Xi^-1 Xj
  Quat qe;   // rotation of E (un-normalised sign)
  double s;  // +1 / -1 so that the reported quaternion has w >= 0
};

SSLAM_HD void se3_error(const Pose& Xi, const Pose& Xj, const Pose& Z, Se3Lin& L) {
  L.qa = qconj(Z.q);
  const Quat qii = qconj(Xi.q);
  L.tb = qrot(qii, Xj.t - Xi.t);
  L.qb = qmul(qii, Xj.q);
  L.Ra = qmat(L.qa);
  const Vec3 te = mul(L.Ra, L.tb - Z.t);
  L.qe = qmul(L.qa, L.qb);
  L.s = L.qe.w < 0 ? -1.0 : 1.0;
  L.e[0] = te.x; L.e[1] = te.y; L.e[2] = te.z;
  L.e[3] = L.s * L.qe.x; L.e[4] = L.s * L.qe.y; L.e[5] = L.s * L.qe.z;
}

// Column c (0..5) of d e / d delta_i, written to col[6].
//   translation rows: [-Ra | 2 Ra [tb]x];  rotation rows: [0 | -s * xyz(qa (e_k,0) qb)]
SSLAM_HD double pick3(int k, double a, double b, double c) { return k == 0 ? a : (k == 1 ? b : c); }
SSLAM_HD void se3_Ji_col(const Se3Lin& L, int c, double col[6]) {
  if (c < 3) {  // (no dynamic indexing: keeps the matrices in registers)
    col[0] = -pick3(c, L.Ra.m[0], L.Ra.m[1], L.Ra.m[2]);
    col[1] = -pick3(c, L.Ra.m[3], L.Ra.m[4], L.Ra.m[5]);
    col[2] = -pick3(c, L.Ra.m[6], L.Ra.m[7], L.Ra.m[8]);
    col[3] = col[4] = col[5] = 0;
  } else {
    const int k = c - 3;
    // column k of [tb]x
    Vec3 sk = k == 0 ? Vec3{0, L.tb.z, -L.tb.y} : (k == 1 ? Vec3{-L.tb.z, 0, L.tb.x} : Vec3{L.tb.y, -L.tb.x, 0});
    const Vec3 a = mul(L.Ra, sk);
    col[0] = 2 * a.x; col[1] = 2 * a.y; col[2] = 2 * a.z;
    const Quat vk = {k == 0 ? 1.0 : 0.0, k == 1 ? 1.0 : 0.0, k == 2 ? 1.0 : 0.0, 0.0};
    const Quat t = qmul(qmul(L.qa, vk), L.qb);
    col[3] = -L.s * t.x; col[4] = -L.s * t.y; col[5] = -L.s * t.z;
  }
}
// Column c of d e / d delta_j:  [[Re, 0], [0, s (w I + [q_xyz]x)]]
SSLAM_HD void se3_Jj_col(const Se3Lin& L, int c, double col[6]) {
  if (c < 3) {
    col[0] = pick3(c, L.Re.m[0], L.Re.m[1], L.Re.m[2]);
    col[1] = pick3(c, L.Re.m[3], L.Re.m[4], L.Re.m[5]);
    col[2] = pick3(c, L.Re.m[6], L.Re.m[7], L.Re.m[8]);
    col[3] = col[4] = col[5] = 0;
  } else {
    const int k = c - 3;
    const double w = L.qe.w, x = L.qe.x, y = L.qe.y, z = L.qe.z;
    col[0] = col[1] = col[2] = 0;
    if (k == 0) { col[3] = L.s * w; col[4] = L.s * z; col[5] = -L.s * y; }
    else if (k == 1) { col[3] = -L.s * z; col[4] = L.s * w; col[5] = L.s * x; }
    else { col[3] = L.s * y; col[4] = -L.s * x; col[5] = L.s * w; }
  }
}
SSLAM_HD void se3_full_jacobians(Se3Lin& L, double Ji[36], double Jj[36]) {  // row-major 6x6
  L.Re = qmat(L.qe);
  for (int c = 0; c < 6; ++c) {
    double a[6], b[6];
    se3_Ji_col(L, c, a);
    se3_Jj_col(L, c, b);
    for (int r = 0; r < 6; ++r) { Ji[r * 6 + c] = a[r]; Jj[r * 6 + c] = b[r]; }
  }
}

// ---------------------------------------------------------------------------------------------
// EdgeSE3PointXYZ with identity sensor offset: e = Ri^T (p - ti) - z
struct PointLin {
  double e[3];
  Vec3 pc;
  Mat3 R;  // rotation of Xi
};
SSLAM_HD void point_error(const Pose& Xi, Vec3 p, Vec3 z, PointLin& L) {
  L.R = qmat(Xi.q);
  L.pc = mulT(L.R, p - Xi.t);
  L.e[0] = L.pc.x - z.x; L.e[1] = L.pc.y - z.y; L.e[2] = L.pc.z - z.z;
}
// Ji = [-I | 2[pc]x] (3x6 row-major), Jl = Ri^T (3x3 row-major)
SSLAM_HD void point_jacobians(const PointLin& L, double Ji[18], double Jl[9]) {
  for (int k = 0; k < 18; ++k) Ji[k] = 0;
  Ji[0] = -1; Ji[7] = -1; Ji[14] = -1;
  Ji[4] = -2 * L.pc.z; Ji[5] = 2 * L.pc.y;
  Ji[6 + 3] = 2 * L.pc.z; Ji[6 + 5] = -2 * L.pc.x;
  Ji[12 + 3] = -2 * L.pc.y; Ji[12 + 4] = 2 * L.pc.x;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) Jl[r * 3 + c] = L.R.m[c * 3 + r];
}

// ---------------------------------------------------------------------------------------------
// Plane3D helpers (g2o slam3d_addons): plane = (n, d), n.x + d = 0, distance() = -d
struct Plane {
  Vec3 n;
  double d;
};
SSLAM_HD double pl_azimuth(Vec3 n) { return atan2(n.y, n.x); }
SSLAM_HD double pl_elevation(Vec3 n) { return atan2(n.z, sqrt(n.x * n.x + n.y * n.y)); }
// Rz(azimuth) * Ry(-elevation).  g2o's Plane3D::rotation() goes through the two angles (atan2, then cos / sin of them); the cosines and sines
// of atan2(y, x) and atan2(z, r) ARE x / r, y / r and r / |n|, z / |n|: taken from the components here (round 6) -- the same matrix to an ulp per
// entry, two atan2 and two sincos (700 of the 1,190 FP64 instructions of an error evaluation) less.  A plane edge evaluates its error 19
// times per linearisation (central differences); the rounding noise of such a Jacobian (1e-16 / 2e-9) is the same either way, and is what
// the 2e-5 of the plane parity tests allow for since round 1 (host libm and device ocml differ by an ulp as well).
SSLAM_HD Mat3 pl_rotation(Vec3 n) {
  const double r2 = n.x * n.x + n.y * n.y;
  const double r = sqrt(r2), nn = sqrt(r2 + n.z * n.z);
  const double ca = r > 0 ? n.x / r : 1.0, sa = r > 0 ? n.y / r : 0.0;          // atan2(0, 0) = 0
  const double cb = nn > 0 ? r / nn : 1.0, sb = nn > 0 ? -n.z / nn : 0.0;       // cos(-el) = cos(el), sin(-el) = -sin(el)
  Mat3 R;
  R.m[0] = ca * cb; R.m[1] = -sa; R.m[2] = ca * sb;
  R.m[3] = sa * cb; R.m[4] = ca;  R.m[5] = sa * sb;
  R.m[6] = -sb;     R.m[7] = 0;   R.m[8] = cb;
  return R;
}
SSLAM_HD Plane pl_oplus(Plane p, const double v[3]) {
  const Mat3 R = pl_rotation(p.n);
  const Vec3 s = {cos(v[1]) * cos(v[0]), cos(v[1]) * sin(v[0]), sin(v[1])};
  Vec3 n = mul(R, s);
  double d = -(-p.d + v[2]);
  const double nn = sqrt(dot(n, n));
  return {{n.x / nn, n.y / nn, n.z / nn}, d / nn};
}
// EdgeSE3Plane::computeError: (Xi^-1 ∘ pi_w) ⊖ z
SSLAM_HD void plane_error(const Pose& Xi, Plane pw, Plane z, double e[3]) {
  const Quat qi = qconj(Xi.q);
  const Vec3 ti = -1.0 * qrot(qi, Xi.t);
  const Vec3 n = qrot(qi, pw.n);
  const double d = pw.d - dot(ti, n);
  const Vec3 m = mulT(pl_rotation(n), z.n);
  e[0] = pl_azimuth(m); e[1] = pl_elevation(m); e[2] = -d + z.d;
}
// g2o BaseBinaryEdge numeric Jacobian: central differences, delta = 1e-9
SSLAM_HD void plane_jacobians(const Pose& Xi, Plane pw, Plane z, double Ji[18], double Jl[9]) {
  const double delta = 1e-9, scalar = 1.0 / (2 * delta);
  for (int d = 0; d < 6; ++d) {
    double dv[6] = {0, 0, 0, 0, 0, 0}, ep[3], em[3];
    dv[d] = delta;  plane_error(se3_oplus(Xi, dv), pw, z, ep);
    dv[d] = -delta; plane_error(se3_oplus(Xi, dv), pw, z, em);
    for (int r = 0; r < 3; ++r) Ji[r * 6 + d] = scalar * (ep[r] - em[r]);
  }
  for (int d = 0; d < 3; ++d) {
    double dv[3] = {0, 0, 0}, ep[3], em[3];
    dv[d] = delta;  plane_error(Xi, pl_oplus(pw, dv), z, ep);
    dv[d] = -delta; plane_error(Xi, pl_oplus(pw, dv), z, em);
    for (int r = 0; r < 3; ++r) Jl[r * 3 + d] = scalar * (ep[r] - em[r]);
  }
}

}  // namespace sslam
