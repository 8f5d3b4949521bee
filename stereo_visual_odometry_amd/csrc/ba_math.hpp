// ba_math.hpp -- per-observation arithmetic of the reprojection cost, fp64, device side.
//
// The residual follows the reference functor (include/visual_odometry/solve/optimizer.hpp:30-62
// for the z-forward camera, test/ba_test.h:179-211 for the x-forward one): the pose is
// T_world<-camera stored qx qy qz qw tx ty tz, the camera-frame point is
// p_c = R(q)^T p - R(q)^T t with R scaled by 1/|q|^2 (ceres::QuaternionToRotation), and the
// residual is the pinhole projection minus the observed pixel.
//
// The Jacobians are the closed form of what Ceres assembles for the reference from
// NumericDiffCostFunction x Sophus::Manifold<SE3>::PlusJacobian (optimizer.cpp:17,43-46):
//   d r / d(upsilon, omega) = Jproj * [ -I | [p_c]x ],   d r / d p = Jproj * R(q)^T.
#pragma once
#include <hip/hip_runtime.h>

namespace hipba {

struct CameraK {
  double fx, fy, cx, cy;
  int model;  // 0 = z forward, 1 = x forward
};

// rows of the world->camera rotation, normalised by 1/|q|^2
__host__ __device__ __forceinline__ void quat_to_Rt(const double* q, double* Rt) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double ww = w * w, xx = x * x, yy = y * y, zz = z * z;
  const double n = 1.0 / (ww + zz + yy + xx);
  Rt[0] = (ww + xx - yy - zz) * n; Rt[1] = 2.0 * (x * y + w * z) * n; Rt[2] = 2.0 * (x * z - w * y) * n;
  Rt[3] = 2.0 * (x * y - w * z) * n; Rt[4] = (ww - xx + yy - zz) * n; Rt[5] = 2.0 * (y * z + w * x) * n;
  Rt[6] = 2.0 * (x * z + w * y) * n; Rt[7] = 2.0 * (y * z - w * x) * n; Rt[8] = (ww - xx - yy + zz) * n;
}

// p_c = Rt p - Rt t, in the functor's order of operations
__host__ __device__ __forceinline__ void to_camera(const double* Rt, const double* t, const double* p, double* pc) {
  pc[0] = Rt[0] * p[0] + Rt[1] * p[1] + Rt[2] * p[2];
  pc[1] = Rt[3] * p[0] + Rt[4] * p[1] + Rt[5] * p[2];
  pc[2] = Rt[6] * p[0] + Rt[7] * p[1] + Rt[8] * p[2];
  pc[0] -= Rt[0] * t[0] + Rt[1] * t[1] + Rt[2] * t[2];
  pc[1] -= Rt[3] * t[0] + Rt[4] * t[1] + Rt[5] * t[2];
  pc[2] -= Rt[6] * t[0] + Rt[7] * t[1] + Rt[8] * t[2];
}

__host__ __device__ __forceinline__ void project_residual(const CameraK& K, const double* pc, double u, double v, double* r) {
  if (K.model == 0) {
    r[0] = pc[0] / pc[2] * K.fx + K.cx - u;
    r[1] = pc[1] / pc[2] * K.fy + K.cy - v;
  } else {
    r[0] = -pc[1] / pc[0] * K.fx + K.cx - u;
    r[1] = -pc[2] / pc[0] * K.fy + K.cy - v;
  }
}

// residual only
__host__ __device__ __forceinline__ void residual_only(const CameraK& K, const double* pose, const double* p, double u, double v,
                                              double* r) {
  double Rt[9], pc[3];
  quat_to_Rt(pose, Rt);
  to_camera(Rt, pose + 4, p, pc);
  project_residual(K, pc, u, v, r);
}

// tangent-space Jacobians from the camera-frame point: Jc 2x6 row-major, Jp 2x3 row-major
__host__ __device__ __forceinline__ void projection_jacobians(const CameraK& K, const double* Rt, const double* pc, double* Jc,
                                                     double* Jp) {
  double A[6];
  if (K.model == 0) {
    const double iz = 1.0 / pc[2];
    A[0] = K.fx * iz; A[1] = 0.0; A[2] = -K.fx * pc[0] * iz * iz;
    A[3] = 0.0; A[4] = K.fy * iz; A[5] = -K.fy * pc[1] * iz * iz;
  } else {
    const double ix = 1.0 / pc[0];
    A[0] = K.fx * pc[1] * ix * ix; A[1] = -K.fx * ix; A[2] = 0.0;
    A[3] = K.fy * pc[2] * ix * ix; A[4] = 0.0; A[5] = -K.fy * ix;
  }
#pragma unroll
  for (int row = 0; row < 2; ++row) {
    const double a0 = A[3 * row], a1 = A[3 * row + 1], a2 = A[3 * row + 2];
    Jc[6 * row + 0] = -a0; Jc[6 * row + 1] = -a1; Jc[6 * row + 2] = -a2;
    // A * [p_c]x, [p_c]x = [[0,-z,y],[z,0,-x],[-y,x,0]]
    Jc[6 * row + 3] = a1 * pc[2] - a2 * pc[1];
    Jc[6 * row + 4] = a2 * pc[0] - a0 * pc[2];
    Jc[6 * row + 5] = a0 * pc[1] - a1 * pc[0];
    Jp[3 * row + 0] = a0 * Rt[0] + a1 * Rt[3] + a2 * Rt[6];
    Jp[3 * row + 1] = a0 * Rt[1] + a1 * Rt[4] + a2 * Rt[7];
    Jp[3 * row + 2] = a0 * Rt[2] + a1 * Rt[5] + a2 * Rt[8];
  }
}

// The same Jacobians with the products written out without A's zero entries.
// With A = d(projection)/d(p_c) (2x3, two of its six entries zero in either camera convention):
//   Jc = [ -A | A [p_c]x ],  [p_c]x = [[0,-z,y],[z,0,-x],[-y,x,0]];   Jp = A R^T.
// (The compiler may not drop 0 * x on its own.)  The same values for finite operands, a third fewer multiply-adds;
// for the kernels whose time is their arithmetic -- the others are tuned to the register count the form above gives.
__host__ __device__ __forceinline__ void projection_jacobians_sparse(const CameraK& K, const double* Rt, const double* pc, double* Jc,
                                                     double* Jp) {
  if (K.model == 0) {
    const double iz = 1.0 / pc[2];
    const double a0 = K.fx * iz, a2 = -K.fx * pc[0] * iz * iz;   // row 0: (a0, 0, a2)
    const double b1 = K.fy * iz, b2 = -K.fy * pc[1] * iz * iz;   // row 1: (0, b1, b2)
    Jc[0] = -a0; Jc[1] = -0.0; Jc[2] = -a2;
    Jc[3] = -(a2 * pc[1]); Jc[4] = a2 * pc[0] - a0 * pc[2]; Jc[5] = a0 * pc[1];
    Jc[6] = -0.0; Jc[7] = -b1; Jc[8] = -b2;
    Jc[9] = b1 * pc[2] - b2 * pc[1]; Jc[10] = b2 * pc[0]; Jc[11] = -(b1 * pc[0]);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      Jp[c] = a0 * Rt[c] + a2 * Rt[6 + c];
      Jp[3 + c] = b1 * Rt[3 + c] + b2 * Rt[6 + c];
    }
  } else {
    const double ix = 1.0 / pc[0];
    const double a0 = K.fx * pc[1] * ix * ix, a1 = -K.fx * ix;   // row 0: (a0, a1, 0)
    const double b0 = K.fy * pc[2] * ix * ix, b2 = -K.fy * ix;   // row 1: (b0, 0, b2)
    Jc[0] = -a0; Jc[1] = -a1; Jc[2] = -0.0;
    Jc[3] = a1 * pc[2]; Jc[4] = -(a0 * pc[2]); Jc[5] = a0 * pc[1] - a1 * pc[0];
    Jc[6] = -b0; Jc[7] = -0.0; Jc[8] = -b2;
    Jc[9] = -(b2 * pc[1]); Jc[10] = b2 * pc[0] - b0 * pc[2]; Jc[11] = b0 * pc[1];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      Jp[c] = a0 * Rt[c] + a1 * Rt[3 + c];
      Jp[3 + c] = b0 * Rt[c] + b2 * Rt[6 + c];
    }
  }
}

// residual + tangent-space Jacobians
__host__ __device__ __forceinline__ void residual_jacobian(const CameraK& K, const double* pose, const double* p, double u, double v,
                                                  double* r, double* Jc, double* Jp) {
  double Rt[9], pc[3];
  quat_to_Rt(pose, Rt);
  to_camera(Rt, pose + 4, p, pc);
  project_residual(K, pc, u, v, r);
  projection_jacobians(K, Rt, pc, Jc, Jp);
}

// ceres::HuberLoss::Evaluate followed by Corrector: returns rho0 and the sqrt(rho1) factor
// that scales residual and Jacobian (rho2 <= 0 for Huber, so the Triggs term vanishes).
__host__ __device__ __forceinline__ void huber_correct(double a, double s, double* rho0, double* sqrt_rho1) {
  const double b = a * a;
  if (s > b) {
    const double rt = sqrt(s);
    *rho0 = 2.0 * a * rt - b;
    double rho1 = a / rt;
    if (rho1 < 2.2250738585072014e-308) rho1 = 2.2250738585072014e-308;
    *sqrt_rho1 = sqrt(rho1);
  } else {
    *rho0 = s;
    *sqrt_rho1 = 1.0;
  }
}

// ---- Sophus SE3 retraction T * exp(delta), delta = (upsilon, omega) ----------------------------
// (sophus/se3.hpp, so3.hpp; used through Sophus::Manifold<SE3>::Plus at optimizer.cpp:17,27)
__host__ __device__ __forceinline__ void quat_rotate(const double* q, const double* p, double* out) {
  double uvx = q[1] * p[2] - q[2] * p[1];
  double uvy = q[2] * p[0] - q[0] * p[2];
  double uvz = q[0] * p[1] - q[1] * p[0];
  uvx += uvx; uvy += uvy; uvz += uvz;
  out[0] = p[0] + q[3] * uvx + (q[1] * uvz - q[2] * uvy);
  out[1] = p[1] + q[3] * uvy + (q[2] * uvx - q[0] * uvz);
  out[2] = p[2] + q[3] * uvz + (q[0] * uvy - q[1] * uvx);
}

__host__ __device__ __forceinline__ void se3_exp(const double* a, double* out) {
  const double* ups = a;
  const double* om = a + 3;
  const double eps = 1e-10;
  const double theta_sq = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
  double theta, imag_factor, real_factor;
  if (theta_sq < eps * eps) {
    theta = 0.0;
    const double theta_po4 = theta_sq * theta_sq;
    imag_factor = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_po4;
    real_factor = 1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * theta_po4;
  } else {
    theta = sqrt(theta_sq);
    const double half_theta = 0.5 * theta;
    imag_factor = sin(half_theta) / theta;
    real_factor = cos(half_theta);
  }
  out[0] = imag_factor * om[0]; out[1] = imag_factor * om[1]; out[2] = imag_factor * om[2];
  out[3] = real_factor;
  // V * upsilon with V = I + k1 * hat(om) + k2 * hat(om)^2  (V = R when theta < eps)
  if (theta < eps) {
    quat_rotate(out, ups, out + 4);
  } else {
    const double k1 = (1.0 - cos(theta)) / theta_sq;
    const double k2 = (theta - sin(theta)) / (theta_sq * theta);
    // hat(om) u = om x u ; hat(om)^2 u = om x (om x u)
    const double c1x = om[1] * ups[2] - om[2] * ups[1];
    const double c1y = om[2] * ups[0] - om[0] * ups[2];
    const double c1z = om[0] * ups[1] - om[1] * ups[0];
    const double c2x = om[1] * c1z - om[2] * c1y;
    const double c2y = om[2] * c1x - om[0] * c1z;
    const double c2z = om[0] * c1y - om[1] * c1x;
    out[4] = ups[0] + k1 * c1x + k2 * c2x;
    out[5] = ups[1] + k1 * c1y + k2 * c2y;
    out[6] = ups[2] + k1 * c1z + k2 * c2z;
  }
}

__host__ __device__ __forceinline__ void se3_mul(const double* a, const double* b, double* out) {
  const double ax = a[0], ay = a[1], az = a[2], aw = a[3];
  const double bx = b[0], by = b[1], bz = b[2], bw = b[3];
  double w = aw * bw - ax * bx - ay * by - az * bz;
  double x = aw * bx + ax * bw + ay * bz - az * by;
  double y = aw * by + ay * bw + az * bx - ax * bz;
  double z = aw * bz + az * bw + ax * by - ay * bx;
  const double sn = x * x + y * y + z * z + w * w;
  if (sn != 1.0) {
    const double s = 2.0 / (1.0 + sn);
    x *= s; y *= s; z *= s; w *= s;
  }
  double rt[3];
  quat_rotate(a, b + 4, rt);
  out[0] = x; out[1] = y; out[2] = z; out[3] = w;
  out[4] = a[4] + rt[0]; out[5] = a[5] + rt[1]; out[6] = a[6] + rt[2];
}

__host__ __device__ __forceinline__ void se3_plus(const double* pose, const double* delta, double* out) {
  double e[7];
  se3_exp(delta, e);
  se3_mul(pose, e, out);
}

// inverse of a symmetric 3x3 given as (m00 m01 m02 m11 m12 m22); returns the same packing
__host__ __device__ __forceinline__ void inv3_sym(const double* V, double* Vi) {
  const double a = V[0], b = V[1], c = V[2], d = V[3], e = V[4], f = V[5];
  const double A = d * f - e * e, B = c * e - b * f, C = b * e - c * d;
  const double id = 1.0 / (a * A + b * B + c * C);
  Vi[0] = A * id; Vi[1] = B * id; Vi[2] = C * id;
  Vi[3] = (a * f - c * c) * id; Vi[4] = (b * c - a * e) * id; Vi[5] = (a * d - b * b) * id;
}

}  // namespace hipba
