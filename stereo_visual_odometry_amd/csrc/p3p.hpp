// p3p.hpp -- the camera pose from three 3D-2D matches in closed form (Grunert's formulation of the perspective
// three-point problem), for the minimal samples of cv::solvePnPRansac (src/solve/estimator.cpp:28-29 calls it with
// rvec = tvec = 0 and useExtrinsicGuess = false: no prior pose).
//
// Unknown: the distances s1, s2, s3 of the three points along their unit bearing vectors f1, f2, f3.  With
//   a = |P2 - P3|, b = |P1 - P3|, c = |P1 - P2|,  cos(alpha) = f2.f3, cos(beta) = f1.f3, cos(gamma) = f1.f2
// the law of cosines gives three quadratics in (s1, s2, s3); s2 = u s1, s3 = v s1 reduces them to a quartic in v
// (coefficients as in Haralick, Lee, Ottenberg, Noelle, "Review and analysis of solutions of the three point
// perspective pose estimation problem", IJCV 1994, eqs. for Grunert's solution), u follows linearly from v and s1
// from either side.  Every real root with positive distances is a pose: the rigid motion that takes the triangle
// P1 P2 P3 onto s1 f1, s2 f2, s3 f3 (two orthonormal triads).  Up to four poses; the caller picks with a fourth match.
#pragma once
#include <hip/hip_runtime.h>

#include "ba_math.hpp"

namespace hipba {

// (a comparison instead of the library's test: one definition for host and device)
__host__ __device__ inline bool is_finite(double x) { return fabs(x) <= 1.7976931348623157e308; }

// largest real root of z^3 + a2 z^2 + a1 z + a0 (trigonometric form with three real roots, Cardano's otherwise),
// polished by two Newton steps
__host__ __device__ inline double cubic_largest_root(double a2, double a1, double a0) {
  const double Q = (a2 * a2 - 3.0 * a1) / 9.0;
  const double R = (2.0 * a2 * a2 * a2 - 9.0 * a2 * a1 + 27.0 * a0) / 54.0;
  const double Q3 = Q * Q * Q;
  double z;
  if (R * R < Q3) {
    const double th = acos(fmin(fmax(R / sqrt(Q3), -1.0), 1.0));
    const double m = -2.0 * sqrt(Q);
    const double z0 = m * cos(th / 3.0), z1 = m * cos((th + 6.283185307179586) / 3.0),
                 z2 = m * cos((th - 6.283185307179586) / 3.0);
    z = fmax(z0, fmax(z1, z2)) - a2 / 3.0;
  } else {
    const double A = -copysign(cbrt(fabs(R) + sqrt(fmax(R * R - Q3, 0.0))), R);
    const double B = A != 0.0 ? Q / A : 0.0;
    z = A + B - a2 / 3.0;
  }
  for (int it = 0; it < 2; ++it) {
    const double f = ((z + a2) * z + a1) * z + a0;
    const double df = (3.0 * z + 2.0 * a2) * z + a1;
    if (df != 0.0 && is_finite(f / df)) z -= f / df;
  }
  return z;
}

// real roots of c4 x^4 + c3 x^3 + c2 x^2 + c1 x + c0 (Ferrari: the depressed quartic split into two quadratics by the
// largest root of its resolvent cubic; every root polished by three Newton steps on the quartic itself).  A
// discriminant within rounding of zero counts as a double root.  Returns the number of roots written (0..4).
__host__ __device__ inline int quartic_real_roots(double c4, double c3, double c2, double c1, double c0, double* x) {
  const double b = c3 / c4, c = c2 / c4, d = c1 / c4, e = c0 / c4;
  if (!(is_finite(b) && is_finite(c) && is_finite(d) && is_finite(e))) return 0;
  const double b2 = b * b;
  const double p = c - 0.375 * b2;
  const double q = d - 0.5 * b * c + 0.125 * b2 * b;
  const double r = e - 0.25 * b * d + 0.0625 * b2 * c - 0.01171875 * b2 * b2;
  double y[4];
  int n = 0;
  const double z = cubic_largest_root(2.0 * p, p * p - 4.0 * r, -q * q);
  const double tiny = 1e-12 * (1.0 + fabs(p));
  if (!(z > tiny)) {
    // q = 0 to rounding: a quadratic in y^2
    double disc = p * p - 4.0 * r;
    if (disc < 0.0 && disc > -1e-12 * (p * p + fabs(4.0 * r))) disc = 0.0;
    if (disc >= 0.0) {
      const double sd = sqrt(disc);
      const double w0 = 0.5 * (-p + sd), w1 = 0.5 * (-p - sd);
      if (w0 >= 0.0) { y[n++] = sqrt(w0); y[n++] = -sqrt(w0); }
      if (w1 >= 0.0) { y[n++] = sqrt(w1); y[n++] = -sqrt(w1); }
    }
  } else {
    const double s = sqrt(z);
    const double t1 = 0.5 * (p + z - q / s), t2 = 0.5 * (p + z + q / s);
    double d1 = z - 4.0 * t1, d2 = z - 4.0 * t2;
    if (d1 < 0.0 && d1 > -1e-12 * (z + fabs(4.0 * t1))) d1 = 0.0;
    if (d2 < 0.0 && d2 > -1e-12 * (z + fabs(4.0 * t2))) d2 = 0.0;
    if (d1 >= 0.0) { const double sd = sqrt(d1); y[n++] = 0.5 * (-s + sd); y[n++] = 0.5 * (-s - sd); }
    if (d2 >= 0.0) { const double sd = sqrt(d2); y[n++] = 0.5 * (s + sd); y[n++] = 0.5 * (s - sd); }
  }
  int m = 0;
  for (int k = 0; k < n; ++k) {
    double v = y[k] - 0.25 * b;
    for (int it = 0; it < 3; ++it) {
      const double f = (((v + b) * v + c) * v + d) * v + e;
      const double df = ((4.0 * v + 3.0 * b) * v + 2.0 * c) * v + d;
      if (df != 0.0 && is_finite(f / df)) v -= f / df;
    }
    if (is_finite(v)) x[m++] = v;
  }
  return m;
}

// unit vector towards the pixel in the camera frame (the inverse of project_residual's projection)
__host__ __device__ inline void pixel_bearing(const CameraK& K, double u, double v, double* f) {
  const double x = (u - K.cx) / K.fx, y = (v - K.cy) / K.fy;
  const double n = 1.0 / sqrt(1.0 + x * x + y * y);
  if (K.model == 0) { f[0] = x * n; f[1] = y * n; f[2] = n; }
  else { f[0] = n; f[1] = -x * n; f[2] = -y * n; }
}

// orthonormal triad of a triangle: e1 along A->B, e3 normal to the triangle, e2 = e3 x e1 (rows of E); false when
// the points are collinear or coincide
__host__ __device__ inline bool triangle_triad(const double* A, const double* B, const double* C, double E[9]) {
  double d1[3], d2[3];
  for (int i = 0; i < 3; ++i) { d1[i] = B[i] - A[i]; d2[i] = C[i] - A[i]; }
  const double n1 = sqrt(d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2]);
  double e3[3] = {d1[1] * d2[2] - d1[2] * d2[1], d1[2] * d2[0] - d1[0] * d2[2], d1[0] * d2[1] - d1[1] * d2[0]};
  const double n3 = sqrt(e3[0] * e3[0] + e3[1] * e3[1] + e3[2] * e3[2]);
  const double n2 = sqrt(d2[0] * d2[0] + d2[1] * d2[1] + d2[2] * d2[2]);
  if (!(n1 > 0.0) || !(n3 > 1e-12 * n1 * n2)) return false;
  for (int i = 0; i < 3; ++i) { E[i] = d1[i] / n1; E[6 + i] = e3[i] / n3; }
  E[3] = E[7] * E[2] - E[8] * E[1];
  E[4] = E[8] * E[0] - E[6] * E[2];
  E[5] = E[6] * E[1] - E[7] * E[0];
  return true;
}

// unit quaternion (x, y, z, w) of a rotation matrix M (row-major), by the largest of the four squared components
__host__ __device__ inline void rotation_to_quat(const double* M, double* q) {
  const double tr = M[0] + M[4] + M[8];
  if (tr > 0.0) {
    const double s = 2.0 * sqrt(1.0 + tr);
    q[3] = 0.25 * s; q[0] = (M[7] - M[5]) / s; q[1] = (M[2] - M[6]) / s; q[2] = (M[3] - M[1]) / s;
  } else if (M[0] > M[4] && M[0] > M[8]) {
    const double s = 2.0 * sqrt(1.0 + M[0] - M[4] - M[8]);
    q[3] = (M[7] - M[5]) / s; q[0] = 0.25 * s; q[1] = (M[1] + M[3]) / s; q[2] = (M[2] + M[6]) / s;
  } else if (M[4] > M[8]) {
    const double s = 2.0 * sqrt(1.0 + M[4] - M[0] - M[8]);
    q[3] = (M[2] - M[6]) / s; q[0] = (M[1] + M[3]) / s; q[1] = 0.25 * s; q[2] = (M[5] + M[7]) / s;
  } else {
    const double s = 2.0 * sqrt(1.0 + M[8] - M[0] - M[4]);
    q[3] = (M[3] - M[1]) / s; q[0] = (M[2] + M[6]) / s; q[1] = (M[5] + M[7]) / s; q[2] = 0.25 * s;
  }
}

// Poses (qx qy qz qw tx ty tz, camera-to-world as ba_math.hpp stores them) that put the three world points P[0..2]
// ([3][3]) on the pixels uv[0..2] ([3][2]).  Returns how many were written to poses ([4][7]).
__host__ __device__ inline int p3p_poses(const CameraK& K, const double* P, const double* uv, double* poses) {
  double f[9];
  for (int i = 0; i < 3; ++i) pixel_bearing(K, uv[2 * i], uv[2 * i + 1], f + 3 * i);
  double Ep[9];
  if (!triangle_triad(P, P + 3, P + 6, Ep)) return 0;
  double a2 = 0.0, b2 = 0.0, c2 = 0.0;
  for (int i = 0; i < 3; ++i) {
    a2 += (P[3 + i] - P[6 + i]) * (P[3 + i] - P[6 + i]);
    b2 += (P[i] - P[6 + i]) * (P[i] - P[6 + i]);
    c2 += (P[i] - P[3 + i]) * (P[i] - P[3 + i]);
  }
  const double ca = f[3] * f[6] + f[4] * f[7] + f[5] * f[8];
  const double cb = f[0] * f[6] + f[1] * f[7] + f[2] * f[8];
  const double cg = f[0] * f[3] + f[1] * f[4] + f[2] * f[5];
  const double k = (a2 - c2) / b2, m = (a2 + c2) / b2, ab = a2 / b2, cbr = c2 / b2;
  const double A4 = (k - 1.0) * (k - 1.0) - 4.0 * cbr * ca * ca;
  const double A3 = 4.0 * (k * (1.0 - k) * cb - (1.0 - m) * ca * cg + 2.0 * cbr * ca * ca * cb);
  const double A2 = 2.0 * (k * k - 1.0 + 2.0 * k * k * cb * cb + 2.0 * (1.0 - cbr) * ca * ca - 4.0 * m * ca * cb * cg +
                           2.0 * (1.0 - ab) * cg * cg);
  const double A1 = 4.0 * (-k * (1.0 + k) * cb + 2.0 * ab * cg * cg * cb - (1.0 - m) * ca * cg);
  const double A0 = (1.0 + k) * (1.0 + k) - 4.0 * ab * cg * cg;
  double roots[4];
  const int nr = quartic_real_roots(A4, A3, A2, A1, A0, roots);
  int n = 0;
  for (int r = 0; r < nr; ++r) {
    const double v = roots[r];
    if (!(v > 0.0)) continue;
    const double u = ((k - 1.0) * v * v - 2.0 * k * cb * v + 1.0 + k) / (2.0 * (cg - v * ca));
    if (!(u > 0.0) || !is_finite(u)) continue;
    const double den = 1.0 + v * v - 2.0 * v * cb;
    if (!(den > 0.0)) continue;
    const double s1 = sqrt(b2 / den), s2 = u * s1, s3 = v * s1;
    double Q[9], Eq[9];
    for (int i = 0; i < 3; ++i) { Q[i] = s1 * f[i]; Q[3 + i] = s2 * f[3 + i]; Q[6 + i] = s3 * f[6 + i]; }
    if (!triangle_triad(Q, Q + 3, Q + 6, Eq)) continue;
    // camera-to-world rotation M = Ep^T Eq (triad rows), translation = P1 - M Q1
    double M[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) M[3 * i + j] = Ep[i] * Eq[j] + Ep[3 + i] * Eq[3 + j] + Ep[6 + i] * Eq[6 + j];
    double* out = poses + 7 * n;
    rotation_to_quat(M, out);
    for (int i = 0; i < 3; ++i) out[4 + i] = P[i] - (M[3 * i] * Q[0] + M[3 * i + 1] * Q[1] + M[3 * i + 2] * Q[2]);
    bool ok = true;
    for (int i = 0; i < 7; ++i) ok = ok && is_finite(out[i]);
    if (ok) ++n;
  }
  return n;
}

}  // namespace hipba
