/*
 * ba_oracle.c -- CPU restatement of the reference's bundle-adjustment path.
 *
 * TEST INFRASTRUCTURE ONLY; see ba_oracle.h.  PARITY UNPINNED (no Ceres/Sophus
 * build and no asserted outputs exist to pin it against).
 *
 * What is restated, and from where:
 *   - the reprojection functor, literally, both axis conventions
 *       (include/visual_odometry/solve/optimizer.hpp:30-62, test/ba_test.h:179-211)
 *   - ceres::QuaternionToRotation                     (ceres/rotation.h, Ceres >= 2.1)
 *   - ceres::NumericDiffCostFunction<..., CENTRAL>     (ceres/internal/numeric_diff.h)
 *       as instantiated at src/solve/optimizer.cpp:43-46, test/ba_test.cpp:258-262
 *   - Sophus::Manifold<Sophus::SE3> Plus / PlusJacobian (sophus/ceres_manifold.hpp,
 *       se3.hpp, so3.hpp) as used at src/solve/optimizer.cpp:17,27
 *   - ceres::HuberLoss + Corrector                     (test/ba_test.cpp:241,262-265)
 *   - problem structure and constant-block rule        (src/solve/optimizer.cpp:19-52)
 *   - ceres::Solve with the trust-region Levenberg-Marquardt minimizer and an
 *       exact solve of the damped normal equations     (src/solve/optimizer.cpp:54-61)
 */
#include "ba_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ------------------------------------------------------------------------- */
/* Functor                                                                   */
/* ------------------------------------------------------------------------- */

/* ceres::QuaternionToRotation: q = (w, x, y, z), row-major R, scaled by 1/|q|^2. */
static void quaternion_to_rotation(const double q[4], double R[9]) {
  const double a = q[0], b = q[1], c = q[2], d = q[3];
  const double aa = a * a, ab = a * b, ac = a * c, ad = a * d;
  const double bb = b * b, bc = b * c, bd = b * d;
  const double cc = c * c, cd = c * d, dd = d * d;
  R[0] = aa + bb - cc - dd; R[1] = 2.0 * (bc - ad);   R[2] = 2.0 * (ac + bd);
  R[3] = 2.0 * (ad + bc);   R[4] = aa - bb + cc - dd; R[5] = 2.0 * (cd - ab);
  R[6] = 2.0 * (bd - ac);   R[7] = 2.0 * (ab + cd);   R[8] = aa - bb - cc + dd;
  double normalizer = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  normalizer = 1.0 / normalizer;
  for (int i = 0; i < 9; ++i) R[i] *= normalizer;
}

/* optimizer.hpp:30-62 (model 0) and ba_test.h:179-211 (model 1), line for line. */
void oracle_residual(int camera_model, double fx, double fy, double cx, double cy,
                     const double* extrinsics, const double* p, double u, double v,
                     double* residuals) {
  double q[4];
  q[0] = extrinsics[3];
  q[1] = -extrinsics[2];
  q[2] = -extrinsics[1];
  q[3] = -extrinsics[0];
  double R[9];
  quaternion_to_rotation(q, R);
  double p_w[3] = {p[0], p[1], p[2]};
  double p_c[3];
  p_c[0] = R[8] * p_w[0] + R[5] * p_w[1] + R[2] * p_w[2];
  p_c[1] = R[7] * p_w[0] + R[4] * p_w[1] + R[1] * p_w[2];
  p_c[2] = R[6] * p_w[0] + R[3] * p_w[1] + R[0] * p_w[2];
  p_c[0] -= R[8] * extrinsics[4] + R[5] * extrinsics[5] + R[2] * extrinsics[6];
  p_c[1] -= R[7] * extrinsics[4] + R[4] * extrinsics[5] + R[1] * extrinsics[6];
  p_c[2] -= R[6] * extrinsics[4] + R[3] * extrinsics[5] + R[0] * extrinsics[6];
  double predicted_x, predicted_y;
  if (camera_model == ORACLE_Z_FORWARD) {
    predicted_x = p_c[0] / p_c[2] * fx + cx;
    predicted_y = p_c[1] / p_c[2] * fy + cy;
  } else {
    predicted_x = -p_c[1] / p_c[0] * fx + cx;
    predicted_y = -p_c[2] / p_c[0] * fy + cy;
  }
  residuals[0] = predicted_x - u;
  residuals[1] = predicted_y - v;
}

/* ------------------------------------------------------------------------- */
/* Sophus SE3                                                                */
/* ------------------------------------------------------------------------- */

static const double kSophusEps = 1e-10; /* Sophus::Constants<double>::epsilon() */

/* SO3::operator*(Vector3): p + w*2(qv x p) + qv x 2(qv x p) */
static void quat_rotate(const double* q /*x y z w*/, const double* p, double* out) {
  double uvx = q[1] * p[2] - q[2] * p[1];
  double uvy = q[2] * p[0] - q[0] * p[2];
  double uvz = q[0] * p[1] - q[1] * p[0];
  uvx += uvx; uvy += uvy; uvz += uvz;
  out[0] = p[0] + q[3] * uvx + (q[1] * uvz - q[2] * uvy);
  out[1] = p[1] + q[3] * uvy + (q[2] * uvx - q[0] * uvz);
  out[2] = p[2] + q[3] * uvz + (q[0] * uvy - q[1] * uvx);
}

/* SO3 group product with Sophus' first-order renormalisation. */
static void quat_mul(const double* a, const double* b, double* out) {
  const double ax = a[0], ay = a[1], az = a[2], aw = a[3];
  const double bx = b[0], by = b[1], bz = b[2], bw = b[3];
  double w = aw * bw - ax * bx - ay * by - az * bz;
  double x = aw * bx + ax * bw + ay * bz - az * by;
  double y = aw * by + ay * bw + az * bx - ax * bz;
  double z = aw * bz + az * bw + ax * by - ay * bx;
  const double squared_norm = x * x + y * y + z * z + w * w;
  if (squared_norm != 1.0) {
    const double scale = 2.0 / (1.0 + squared_norm);
    x *= scale; y *= scale; z *= scale; w *= scale;
  }
  out[0] = x; out[1] = y; out[2] = z; out[3] = w;
}

void oracle_se3_mul(const double* a, const double* b, double* out) {
  double q[4], rt[3];
  quat_mul(a, b, q);
  quat_rotate(a, b + 4, rt);
  out[0] = q[0]; out[1] = q[1]; out[2] = q[2]; out[3] = q[3];
  out[4] = a[4] + rt[0]; out[5] = a[5] + rt[1]; out[6] = a[6] + rt[2];
}

/* SE3::exp, tangent = (upsilon, omega). */
void oracle_se3_exp(const double* a, double* out) {
  const double* ups = a;
  const double* om = a + 3;
  const double theta_sq = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
  double theta, imag_factor, real_factor;
  if (theta_sq < kSophusEps * kSophusEps) {
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
  const double qx = imag_factor * om[0], qy = imag_factor * om[1], qz = imag_factor * om[2];
  const double qw = real_factor;
  /* Omega = hat(omega), Omega_sq = Omega*Omega */
  const double O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
  double O2[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s += O[3 * i + k] * O[3 * k + j];
      O2[3 * i + j] = s;
    }
  double V[9];
  if (theta < kSophusEps) {
    /* V = so3.matrix() */
    const double q4[4] = {qx, qy, qz, qw};
    for (int j = 0; j < 3; ++j) {
      double e[3] = {0, 0, 0}, c[3];
      e[j] = 1.0;
      quat_rotate(q4, e, c);
      V[0 + j] = c[0]; V[3 + j] = c[1]; V[6 + j] = c[2];
    }
  } else {
    const double k1 = (1.0 - cos(theta)) / theta_sq;
    const double k2 = (theta - sin(theta)) / (theta_sq * theta);
    for (int i = 0; i < 9; ++i) V[i] = k1 * O[i] + k2 * O2[i];
    V[0] += 1.0; V[4] += 1.0; V[8] += 1.0;
  }
  out[0] = qx; out[1] = qy; out[2] = qz; out[3] = qw;
  out[4] = V[0] * ups[0] + V[1] * ups[1] + V[2] * ups[2];
  out[5] = V[3] * ups[0] + V[4] * ups[1] + V[5] * ups[2];
  out[6] = V[6] * ups[0] + V[7] * ups[1] + V[8] * ups[2];
}

/* Sophus::Manifold<SE3>::Plus: T * exp(delta). */
void oracle_se3_plus(const double* pose7, const double* delta6, double* out7) {
  double e[7];
  oracle_se3_exp(delta6, e);
  oracle_se3_mul(pose7, e, out7);
}

void oracle_se3_inverse(const double* a, double* out) {
  double qi[4] = {-a[0], -a[1], -a[2], a[3]};
  double t[3];
  quat_rotate(qi, a + 4, t);
  out[0] = qi[0]; out[1] = qi[1]; out[2] = qi[2]; out[3] = qi[3];
  out[4] = -t[0]; out[5] = -t[1]; out[6] = -t[2];
}

/* SE3::Dx_this_mul_exp_x_at_0, 7x6 row-major. */
void oracle_se3_plus_jacobian(const double* T, double* J) {
  const double qx = T[0], qy = T[1], qz = T[2], qw = T[3];
  const double c0 = 0.5 * qw, c1 = 0.5 * qz, c2 = -c1, c3 = 0.5 * qy, c4 = 0.5 * qx;
  const double c5 = -c4, c6 = -c3;
  const double c7 = qw * qw, c8 = qx * qx, c9 = qy * qy, c10 = -c9, c11 = qz * qz, c12 = -c11;
  const double c13 = 2.0 * qw, c14 = c13 * qz, c15 = 2.0 * qx, c16 = c15 * qy;
  const double c17 = c13 * qy, c18 = c15 * qz, c19 = c7 - c8, c20 = c13 * qx;
  const double c21 = 2.0 * qy * qz;
  memset(J, 0, 42 * sizeof(double));
  J[0 * 6 + 3] = c0; J[0 * 6 + 4] = c2; J[0 * 6 + 5] = c3;
  J[1 * 6 + 3] = c1; J[1 * 6 + 4] = c0; J[1 * 6 + 5] = c5;
  J[2 * 6 + 3] = c6; J[2 * 6 + 4] = c4; J[2 * 6 + 5] = c0;
  J[3 * 6 + 3] = c5; J[3 * 6 + 4] = c6; J[3 * 6 + 5] = c2;
  J[4 * 6 + 0] = c10 + c12 + c7 + c8; J[4 * 6 + 1] = -c14 + c16; J[4 * 6 + 2] = c17 + c18;
  J[5 * 6 + 0] = c14 + c16; J[5 * 6 + 1] = c12 + c19 + c9; J[5 * 6 + 2] = -c20 + c21;
  J[6 * 6 + 0] = -c17 + c18; J[6 * 6 + 1] = c20 + c21; J[6 * 6 + 2] = c10 + c11 + c19;
}

/* ------------------------------------------------------------------------- */
/* Per-observation evaluation                                                */
/* ------------------------------------------------------------------------- */

/* ceres numeric_diff.h, CENTRAL: h = max(sqrt(eps), |x_j| * 1e-6), one coordinate at a time. */
static void numeric_jacobians(const oracle_problem* P, const double* pose, const double* point,
                              double u, double v, double* J7 /*2x7*/, double* J3 /*2x3*/) {
  const double min_step = sqrt(DBL_EPSILON);
  const double rel = 1e-6;
  double x[7];
  memcpy(x, pose, sizeof(x));
  for (int j = 0; j < 7; ++j) {
    double delta = fabs(pose[j]) * rel;
    if (delta < min_step) delta = min_step;
    double rp[2], rm[2];
    x[j] = pose[j] + delta;
    oracle_residual(P->camera_model, P->fx, P->fy, P->cx, P->cy, x, point, u, v, rp);
    double one_over_delta = 1.0 / delta;
    x[j] = pose[j] - delta;
    oracle_residual(P->camera_model, P->fx, P->fy, P->cx, P->cy, x, point, u, v, rm);
    rp[0] -= rm[0]; rp[1] -= rm[1];
    one_over_delta /= 2;
    x[j] = pose[j];
    J7[0 * 7 + j] = rp[0] * one_over_delta;
    J7[1 * 7 + j] = rp[1] * one_over_delta;
  }
  double y[3];
  memcpy(y, point, sizeof(y));
  for (int j = 0; j < 3; ++j) {
    double delta = fabs(point[j]) * rel;
    if (delta < min_step) delta = min_step;
    double rp[2], rm[2];
    y[j] = point[j] + delta;
    oracle_residual(P->camera_model, P->fx, P->fy, P->cx, P->cy, pose, y, u, v, rp);
    double one_over_delta = 1.0 / delta;
    y[j] = point[j] - delta;
    oracle_residual(P->camera_model, P->fx, P->fy, P->cx, P->cy, pose, y, u, v, rm);
    rp[0] -= rm[0]; rp[1] -= rm[1];
    one_over_delta /= 2;
    y[j] = point[j];
    J3[0 * 3 + j] = rp[0] * one_over_delta;
    J3[1 * 3 + j] = rp[1] * one_over_delta;
  }
}

/* Closed form of (ambient Jacobian) x (plus Jacobian): J_pose = Jproj [-I | [p_c]x], J_point = Jproj R^T. */
static void analytic_jacobians(const oracle_problem* P, const double* e, const double* p,
                               double* J6 /*2x6*/, double* J3 /*2x3*/) {
  double q[4] = {e[3], -e[2], -e[1], -e[0]};
  double R[9];
  quaternion_to_rotation(q, R);
  /* Rt = rows of the world->camera rotation, in the functor's reversed indexing */
  const double Rt[9] = {R[8], R[5], R[2], R[7], R[4], R[1], R[6], R[3], R[0]};
  const double d[3] = {p[0] - e[4], p[1] - e[5], p[2] - e[6]};
  double pc[3];
  for (int i = 0; i < 3; ++i) pc[i] = Rt[3 * i] * d[0] + Rt[3 * i + 1] * d[1] + Rt[3 * i + 2] * d[2];
  double A[6]; /* Jproj 2x3 */
  if (P->camera_model == ORACLE_Z_FORWARD) {
    const double iz = 1.0 / pc[2];
    A[0] = P->fx * iz; A[1] = 0.0; A[2] = -P->fx * pc[0] * iz * iz;
    A[3] = 0.0; A[4] = P->fy * iz; A[5] = -P->fy * pc[1] * iz * iz;
  } else {
    const double ix = 1.0 / pc[0];
    A[0] = P->fx * pc[1] * ix * ix; A[1] = -P->fx * ix; A[2] = 0.0;
    A[3] = P->fy * pc[2] * ix * ix; A[4] = 0.0; A[5] = -P->fy * ix;
  }
  /* [p_c]x */
  const double X[9] = {0, -pc[2], pc[1], pc[2], 0, -pc[0], -pc[1], pc[0], 0};
  for (int r = 0; r < 2; ++r) {
    for (int c = 0; c < 3; ++c) {
      J6[r * 6 + c] = -A[r * 3 + c];
      J6[r * 6 + 3 + c] = A[r * 3 + 0] * X[c] + A[r * 3 + 1] * X[3 + c] + A[r * 3 + 2] * X[6 + c];
      J3[r * 3 + c] = A[r * 3 + 0] * Rt[c] + A[r * 3 + 1] * Rt[3 + c] + A[r * 3 + 2] * Rt[6 + c];
    }
  }
}

/* ceres::HuberLoss::Evaluate */
static void huber(double a, double s, double rho[3]) {
  const double b = a * a;
  if (s > b) {
    const double r = sqrt(s);
    rho[0] = 2.0 * a * r - b;
    rho[1] = a / r;
    if (rho[1] < DBL_MIN) rho[1] = DBL_MIN;
    rho[2] = -rho[1] / (2.0 * s);
  } else {
    rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
  }
}

/*
 * One residual block as ceres::ResidualBlock::Evaluate produces it: residual,
 * tangent Jacobians, loss correction (Corrector with rho[2] <= 0 reduces to a
 * sqrt(rho[1]) scaling), cost contribution.  Returns 0 if anything is non-finite.
 */
static int eval_block(const oracle_problem* P, int jac_mode, int loss, double loss_scale,
                      int apply_loss, const double* pose, const double* point, double u, double v,
                      double* r2, double* J6, double* J3, double* J7amb, double* cost) {
  oracle_residual(P->camera_model, P->fx, P->fy, P->cx, P->cy, pose, point, u, v, r2);
  if (J6 || J3 || J7amb) {
    double j6[12], j3[6], j7[14];
    if (jac_mode == ORACLE_JAC_NUMERIC_CENTRAL) {
      numeric_jacobians(P, pose, point, u, v, j7, j3);
      double PJ[42];
      oracle_se3_plus_jacobian(pose, PJ);
      for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 6; ++c) {
          double s = 0.0;
          for (int k = 0; k < 7; ++k) s += j7[r * 7 + k] * PJ[k * 6 + c];
          j6[r * 6 + c] = s;
        }
      if (J7amb) memcpy(J7amb, j7, sizeof(j7));
    } else {
      analytic_jacobians(P, pose, point, j6, j3);
      if (J7amb) memset(J7amb, 0, 14 * sizeof(double));
    }
    if (J6) memcpy(J6, j6, sizeof(j6));
    if (J3) memcpy(J3, j3, sizeof(j3));
  }
  const double sq = r2[0] * r2[0] + r2[1] * r2[1];
  if (loss == ORACLE_LOSS_HUBER && apply_loss) {
    double rho[3];
    huber(loss_scale, sq, rho);
    *cost = 0.5 * rho[0];
    const double s1 = sqrt(rho[1]);
    /* Corrector: rho[2] <= 0 always for Huber -> residual_scaling = sqrt(rho1), alpha = 0 */
    if (J6) for (int i = 0; i < 12; ++i) J6[i] *= s1;
    if (J3) for (int i = 0; i < 6; ++i) J3[i] *= s1;
    r2[0] *= s1; r2[1] *= s1;
  } else {
    *cost = 0.5 * sq;
  }
  int ok = isfinite(r2[0]) && isfinite(r2[1]) && isfinite(*cost);
  if (J6) for (int i = 0; i < 12; ++i) ok = ok && isfinite(J6[i]);
  if (J3) for (int i = 0; i < 6; ++i) ok = ok && isfinite(J3[i]);
  return ok;
}

int oracle_evaluate(const oracle_problem* p, int jacobian_mode, int loss, double loss_scale,
                    int apply_loss, int num_threads, double* cost, double* residuals,
                    double* jac_pose, double* jac_point, double* jac_pose_ambient) {
  double total = 0.0;
  int all_ok = 1;
  if (num_threads < 1) num_threads = 1;
#pragma omp parallel for num_threads(num_threads) reduction(+ : total) reduction(&& : all_ok) schedule(static)
  for (int64_t k = 0; k < p->n_obs; ++k) {
    double r[2], c;
    int ok = eval_block(p, jacobian_mode, loss, loss_scale, apply_loss,
                        p->poses + 7 * (int64_t)p->obs_pose[k], p->points + 3 * (int64_t)p->obs_point[k],
                        p->obs_uv[2 * k], p->obs_uv[2 * k + 1], r,
                        jac_pose ? jac_pose + 12 * k : NULL, jac_point ? jac_point + 6 * k : NULL,
                        jac_pose_ambient ? jac_pose_ambient + 14 * k : NULL, &c);
    if (residuals) { residuals[2 * k] = r[0]; residuals[2 * k + 1] = r[1]; }
    total += c;
    all_ok = all_ok && ok;
  }
  if (cost) *cost = total;
  return all_ok ? 0 : 1;
}

/* ------------------------------------------------------------------------- */
/* Dense Cholesky (lower, row-major, in place)                               */
/* ------------------------------------------------------------------------- */

static int chol_unblocked(double* A, int64_t lda, int n) {
  for (int j = 0; j < n; ++j) {
    double d = A[j * lda + j];
    for (int k = 0; k < j; ++k) d -= A[j * lda + k] * A[j * lda + k];
    if (!(d > 0.0) || !isfinite(d)) return 1;
    d = sqrt(d);
    A[j * lda + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = A[i * lda + j];
      for (int k = 0; k < j; ++k) s -= A[i * lda + k] * A[j * lda + k];
      A[i * lda + j] = s / d;
    }
  }
  return 0;
}

static int chol_factor_tiled(double* A, int n, int num_threads);

static int chol_factor(double* A, int n, int num_threads) {
  if (n >= 1536) return chol_factor_tiled(A, n, num_threads);
  const int nb = 64;
  const int64_t lda = n;
  for (int k = 0; k < n; k += nb) {
    const int kb = (n - k < nb) ? n - k : nb;
    if (chol_unblocked(A + k * lda + k, lda, kb)) return 1;
    const int rest = n - k - kb;
    if (rest <= 0) break;
    /* panel: L21 = A21 * L11^-T */
#pragma omp parallel for num_threads(num_threads) schedule(static)
    for (int i = k + kb; i < n; ++i) {
      double* row = A + i * lda + k;
      for (int j = 0; j < kb; ++j) {
        double s = row[j];
        const double* lj = A + (k + j) * lda + k;
        for (int t = 0; t < j; ++t) s -= row[t] * lj[t];
        row[j] = s / lj[j];
      }
    }
    /* trailing update (lower triangle): A22 -= L21 L21^T */
#pragma omp parallel for num_threads(num_threads) schedule(dynamic, 8)
    for (int i = k + kb; i < n; ++i) {
      const double* li = A + i * lda + k;
      for (int j = k + kb; j <= i; ++j) {
        const double* lj = A + j * lda + k;
        double s = 0.0;
        for (int t = 0; t < kb; ++t) s += li[t] * lj[t];
        A[i * lda + j] -= s;
      }
    }
  }
  return 0;
}

/* Large matrices (the reduced camera system of thousands of poses): the same right-looking factorisation with
 * the trailing update done on cache-sized tiles by a 4x4 register kernel, four partial sums per entry (t mod 4),
 * so that a full-size CPU baseline finishes in seconds.  Same arithmetic up to the order of the additions. */
typedef double v4d __attribute__((vector_size(32)));
typedef double v4d_unaligned __attribute__((vector_size(32), aligned(8)));
#define ld4(p) ((v4d)(*(const v4d_unaligned*)(p)))

/* A[i][j] -= sum_t A[i][k0+t] A[j][k0+t]  for i in [i0,i1), j in [j0, min(j1, i+1)) */
__attribute__((target_clones("avx2", "default")))
void oracle_syrk_tile(double* A, int64_t lda, int k0, int kb, int i0, int i1, int j0, int j1) {
  const int kb4 = kb & ~3;
  for (int i = i0; i < i1; i += 4) {
    const int ih = (i + 4 <= i1) ? 4 : i1 - i;
    int j = j0;
    if (ih == 4) {
      for (; j + 4 <= j1 && j + 3 <= i; j += 4) {
        const double* a0 = A + (int64_t)i * lda + k0;
        const double* b0 = A + (int64_t)j * lda + k0;
        v4d acc[4][4];
        for (int p = 0; p < 4; ++p)
          for (int q = 0; q < 4; ++q) acc[p][q] = (v4d){0.0, 0.0, 0.0, 0.0};
        for (int t = 0; t < kb4; t += 4) {
          const v4d x0 = ld4(a0 + t), x1 = ld4(a0 + lda + t), x2 = ld4(a0 + 2 * lda + t), x3 = ld4(a0 + 3 * lda + t);
          const v4d y0 = ld4(b0 + t), y1 = ld4(b0 + lda + t), y2 = ld4(b0 + 2 * lda + t), y3 = ld4(b0 + 3 * lda + t);
          acc[0][0] += x0 * y0; acc[0][1] += x0 * y1; acc[0][2] += x0 * y2; acc[0][3] += x0 * y3;
          acc[1][0] += x1 * y0; acc[1][1] += x1 * y1; acc[1][2] += x1 * y2; acc[1][3] += x1 * y3;
          acc[2][0] += x2 * y0; acc[2][1] += x2 * y1; acc[2][2] += x2 * y2; acc[2][3] += x2 * y3;
          acc[3][0] += x3 * y0; acc[3][1] += x3 * y1; acc[3][2] += x3 * y2; acc[3][3] += x3 * y3;
        }
        for (int p = 0; p < 4; ++p)
          for (int q = 0; q < 4; ++q) {
            double sum = (acc[p][q][0] + acc[p][q][1]) + (acc[p][q][2] + acc[p][q][3]);
            for (int t = kb4; t < kb; ++t) sum += a0[p * lda + t] * b0[q * lda + t];
            A[(int64_t)(i + p) * lda + j + q] -= sum;
          }
      }
    }
    /* what the 4x4 kernel leaves: the blocks the diagonal crosses and ragged edges */
    for (int p = 0; p < ih; ++p) {
      const double* a0 = A + (int64_t)(i + p) * lda + k0;
      const int jend = (j1 < i + p + 1) ? j1 : i + p + 1;
      for (int jj = j; jj < jend; ++jj) {
        const double* b0 = A + (int64_t)jj * lda + k0;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        for (int t = 0; t < kb4; t += 4) {
          s0 += a0[t] * b0[t]; s1 += a0[t + 1] * b0[t + 1]; s2 += a0[t + 2] * b0[t + 2]; s3 += a0[t + 3] * b0[t + 3];
        }
        double sum = (s0 + s1) + (s2 + s3);
        for (int t = kb4; t < kb; ++t) sum += a0[t] * b0[t];
        A[(int64_t)(i + p) * lda + jj] -= sum;
      }
    }
  }
}

static int chol_factor_tiled(double* A, int n, int num_threads) {
  const int nb = 128, tile = 192;
  const int64_t lda = n;
  for (int k = 0; k < n; k += nb) {
    const int kb = (n - k < nb) ? n - k : nb;
    if (chol_unblocked(A + k * lda + k, lda, kb)) return 1;
    const int r0 = k + kb;
    if (r0 >= n) break;
#pragma omp parallel for num_threads(num_threads) schedule(static)
    for (int i = r0; i < n; ++i) {
      double* row = A + i * lda + k;
      for (int j = 0; j < kb; ++j) {
        double s = row[j];
        const double* lj = A + (k + j) * lda + k;
        for (int t = 0; t < j; ++t) s -= row[t] * lj[t];
        row[j] = s / lj[j];
      }
    }
    const int nt = (n - r0 + tile - 1) / tile;
    const int n_tiles = nt * (nt + 1) / 2;
#pragma omp parallel for num_threads(num_threads) schedule(dynamic, 1)
    for (int idx = 0; idx < n_tiles; ++idx) {
      /* idx -> (ti, tj), tj <= ti, rows of large ti first so that the long rows are not left for last */
      int ti = nt - 1, rem = idx;
      while (rem > ti) { rem -= ti + 1; --ti; }
      const int tj = rem;
      const int i0 = r0 + ti * tile, j0 = r0 + tj * tile;
      const int i1 = (i0 + tile < n) ? i0 + tile : n, j1 = (j0 + tile < n) ? j0 + tile : n;
      oracle_syrk_tile(A, lda, k, kb, i0, i1, j0, j1);
    }
  }
  return 0;
}

static void chol_solve(const double* L, int n, double* b) {
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    const double* li = L + (int64_t)i * n;
    for (int k = 0; k < i; ++k) s -= li[k] * b[k];
    b[i] = s / li[i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int k = i + 1; k < n; ++k) s -= L[(int64_t)k * n + i] * b[k];
    b[i] = s / L[(int64_t)i * n + i];
  }
}

/* ------------------------------------------------------------------------- */
/* Reduced program                                                           */
/* ------------------------------------------------------------------------- */

typedef struct {
  const oracle_problem* P;
  const oracle_options* O;
  int n_poses, n_points;
  int64_t n_obs;
  uint8_t* pose_active;
  uint8_t* point_active;
  int* pose_col;   /* tangent column of an active pose, else -1 */
  int* point_col;
  int n_ap, n_apt, n_tan;
  int64_t n_act;
  int64_t* act;    /* indices of observations that touch an active block */
  double fixed_cost;
  /* evaluation at x (scaled, loss-corrected) */
  double* r;   /* 2*n_act */
  double* Jc;  /* 12*n_act */
  double* Jp;  /* 6*n_act */
  double* grad;  /* unscaled J^T r */
  double* scale; /* Jacobi scaling */
  double t_eval, t_lin;
} prog;

static void prog_free(prog* g) {
  free(g->pose_active); free(g->point_active); free(g->pose_col); free(g->point_col);
  free(g->act); free(g->r); free(g->Jc); free(g->Jp); free(g->grad); free(g->scale);
}

/* Mirrors Ceres' preprocessing: constant blocks and blocks without residuals
 * leave the program; residual blocks touching only constants become fixed cost. */
static int prog_init(prog* g, const oracle_problem* P, const oracle_options* O) {
  memset(g, 0, sizeof(*g));
  g->P = P; g->O = O;
  g->n_poses = P->n_poses; g->n_points = P->n_points; g->n_obs = P->n_obs;
  g->pose_active = (uint8_t*)calloc((size_t)P->n_poses + 1, 1);
  g->point_active = (uint8_t*)calloc((size_t)P->n_points + 1, 1);
  g->pose_col = (int*)malloc(sizeof(int) * ((size_t)P->n_poses + 1));
  g->point_col = (int*)malloc(sizeof(int) * ((size_t)P->n_points + 1));
  g->act = (int64_t*)malloc(sizeof(int64_t) * ((size_t)P->n_obs + 1));
  for (int64_t k = 0; k < P->n_obs; ++k) {
    const int c = P->obs_pose[k], q = P->obs_point[k];
    if (c < 0 || c >= P->n_poses || q < 0 || q >= P->n_points) return 1;
    const int cc = P->pose_const ? P->pose_const[c] : 0;
    const int qc = P->point_const ? P->point_const[q] : 0;
    if (!cc) g->pose_active[c] = 1;
    if (!qc) g->point_active[q] = 1;
    if (!cc || !qc) g->act[g->n_act++] = k;
  }
  /* columns: landmarks first, then poses (optimizer.cpp:19-27 adds them in that order) */
  int col = 0;
  for (int i = 0; i < P->n_points; ++i) {
    if (g->point_active[i]) { g->point_col[i] = col; col += 3; g->n_apt++; } else g->point_col[i] = -1;
  }
  for (int i = 0; i < P->n_poses; ++i) {
    if (g->pose_active[i]) { g->pose_col[i] = col; col += 6; g->n_ap++; } else g->pose_col[i] = -1;
  }
  g->n_tan = col;
  g->r = (double*)malloc(sizeof(double) * 2 * ((size_t)g->n_act + 1));
  g->Jc = (double*)malloc(sizeof(double) * 12 * ((size_t)g->n_act + 1));
  g->Jp = (double*)malloc(sizeof(double) * 6 * ((size_t)g->n_act + 1));
  g->grad = (double*)calloc((size_t)col + 1, sizeof(double));
  g->scale = (double*)malloc(sizeof(double) * ((size_t)col + 1));
  for (int i = 0; i < col; ++i) g->scale[i] = 1.0;
  return 0;
}

/* cost (+ optionally residuals/Jacobians/gradient) of the reduced program at (poses, points). */
static int prog_evaluate(prog* g, const double* poses, const double* points, int with_jac, double* cost) {
  const oracle_problem* P = g->P;
  const oracle_options* O = g->O;
  const double t0 = now_s();
  double total = 0.0;
  int all_ok = 1;
  int nt = O->num_threads < 1 ? 1 : O->num_threads;
#pragma omp parallel for num_threads(nt) reduction(+ : total) reduction(&& : all_ok) schedule(static)
  for (int64_t a = 0; a < g->n_act; ++a) {
    const int64_t k = g->act[a];
    double r[2], c;
    int ok = eval_block(P, O->jacobian_mode, O->loss, O->loss_scale, 1,
                        poses + 7 * (int64_t)P->obs_pose[k], points + 3 * (int64_t)P->obs_point[k],
                        P->obs_uv[2 * k], P->obs_uv[2 * k + 1], r,
                        with_jac ? g->Jc + 12 * a : NULL, with_jac ? g->Jp + 6 * a : NULL, NULL, &c);
    if (with_jac) { g->r[2 * a] = r[0]; g->r[2 * a + 1] = r[1]; }
    total += c;
    all_ok = all_ok && ok;
  }
  *cost = total;
  if (with_jac && all_ok) {
    memset(g->grad, 0, sizeof(double) * (size_t)g->n_tan);
    for (int64_t a = 0; a < g->n_act; ++a) {
      const int64_t k = g->act[a];
      const int pc = g->pose_col[P->obs_pose[k]];
      const int qc = g->point_col[P->obs_point[k]];
      const double* r = g->r + 2 * a;
      if (pc >= 0) {
        const double* J = g->Jc + 12 * a;
        for (int c = 0; c < 6; ++c) g->grad[pc + c] += J[c] * r[0] + J[6 + c] * r[1];
      }
      if (qc >= 0) {
        const double* J = g->Jp + 6 * a;
        for (int c = 0; c < 3; ++c) g->grad[qc + c] += J[c] * r[0] + J[3 + c] * r[1];
      }
    }
  }
  g->t_eval += now_s() - t0;
  return all_ok ? 0 : 1;
}

/* SquaredColumnNorm of the stored (possibly scaled) Jacobian. */
static void prog_colnorm2(const prog* g, double* out) {
  const oracle_problem* P = g->P;
  memset(out, 0, sizeof(double) * (size_t)g->n_tan);
  for (int64_t a = 0; a < g->n_act; ++a) {
    const int64_t k = g->act[a];
    const int pc = g->pose_col[P->obs_pose[k]];
    const int qc = g->point_col[P->obs_point[k]];
    if (pc >= 0) {
      const double* J = g->Jc + 12 * a;
      for (int c = 0; c < 6; ++c) out[pc + c] += J[c] * J[c] + J[6 + c] * J[6 + c];
    }
    if (qc >= 0) {
      const double* J = g->Jp + 6 * a;
      for (int c = 0; c < 3; ++c) out[qc + c] += J[c] * J[c] + J[3 + c] * J[3 + c];
    }
  }
}

static void prog_scale_columns(prog* g) {
  const oracle_problem* P = g->P;
  for (int64_t a = 0; a < g->n_act; ++a) {
    const int64_t k = g->act[a];
    const int pc = g->pose_col[P->obs_pose[k]];
    const int qc = g->point_col[P->obs_point[k]];
    if (pc >= 0) {
      double* J = g->Jc + 12 * a;
      for (int c = 0; c < 6; ++c) { J[c] *= g->scale[pc + c]; J[6 + c] *= g->scale[pc + c]; }
    }
    if (qc >= 0) {
      double* J = g->Jp + 6 * a;
      for (int c = 0; c < 3; ++c) { J[c] *= g->scale[qc + c]; J[3 + c] *= g->scale[qc + c]; }
    }
  }
}

/* Solve (J^T J + D^2) y = J^T r exactly with the full normal equations. */
static int solve_full(prog* g, const double* D, double* y) {
  const oracle_problem* P = g->P;
  const int n = g->n_tan;
  double* H = (double*)calloc((size_t)n * (size_t)n, sizeof(double));
  if (!H) return 2;
  memset(y, 0, sizeof(double) * (size_t)n);
  for (int64_t a = 0; a < g->n_act; ++a) {
    const int64_t k = g->act[a];
    const int pc = g->pose_col[P->obs_pose[k]];
    const int qc = g->point_col[P->obs_point[k]];
    const double* r = g->r + 2 * a;
    const double* Jc = g->Jc + 12 * a;
    const double* Jp = g->Jp + 6 * a;
    int cols[9]; double j0[9], j1[9]; int m = 0;
    if (qc >= 0) for (int c = 0; c < 3; ++c) { cols[m] = qc + c; j0[m] = Jp[c]; j1[m] = Jp[3 + c]; ++m; }
    if (pc >= 0) for (int c = 0; c < 6; ++c) { cols[m] = pc + c; j0[m] = Jc[c]; j1[m] = Jc[6 + c]; ++m; }
    for (int i = 0; i < m; ++i) {
      y[cols[i]] += j0[i] * r[0] + j1[i] * r[1];
      for (int j = 0; j < m; ++j) {
        if (cols[j] <= cols[i]) H[(int64_t)cols[i] * n + cols[j]] += j0[i] * j0[j] + j1[i] * j1[j];
      }
    }
  }
  for (int i = 0; i < n; ++i) H[(int64_t)i * n + i] += D[i] * D[i];
  int rc = chol_factor(H, n, g->O->num_threads < 1 ? 1 : g->O->num_threads);
  if (!rc) chol_solve(H, n, y);
  free(H);
  return rc;
}

static int inv3_sym(const double* V /*3x3 row-major*/, double* Vi) {
  const double a = V[0], b = V[1], c = V[2], d = V[4], e = V[5], f = V[8];
  const double A = d * f - e * e, B = c * e - b * f, C = b * e - c * d;
  const double det = a * A + b * B + c * C;
  if (!(fabs(det) > 0.0) || !isfinite(det)) return 1;
  const double id = 1.0 / det;
  Vi[0] = A * id; Vi[1] = B * id; Vi[2] = C * id;
  Vi[3] = B * id; Vi[4] = (a * f - c * c) * id; Vi[5] = (b * c - a * e) * id;
  Vi[6] = C * id; Vi[7] = (b * c - a * e) * id; Vi[8] = (a * d - b * b) * id;
  return 0;
}

/* Same system solved by eliminating the landmarks first (Schur complement). */
static int solve_schur(prog* g, const double* D, double* y) {
  const oracle_problem* P = g->P;
  const int nc = 6 * g->n_ap;
  const int poff = 3 * g->n_apt; /* pose columns start here */
  const int npt = g->n_apt;
  double* V = (double*)calloc((size_t)npt * 9 + 1, sizeof(double));
  double* gp = (double*)calloc((size_t)npt * 3 + 1, sizeof(double));
  double* S = (double*)calloc((size_t)nc * (size_t)nc + 1, sizeof(double));
  double* b = (double*)calloc((size_t)nc + 1, sizeof(double));
  if (!V || !gp || !S || !b) { free(V); free(gp); free(S); free(b); return 2; }
  /* point blocks, camera blocks */
  for (int64_t a = 0; a < g->n_act; ++a) {
    const int64_t k = g->act[a];
    const int pc = g->pose_col[P->obs_pose[k]];
    const int qc = g->point_col[P->obs_point[k]];
    const double* r = g->r + 2 * a;
    const double* Jc = g->Jc + 12 * a;
    const double* Jp = g->Jp + 6 * a;
    if (qc >= 0) {
      double* Vq = V + 3 * qc; /* qc is 3*index */
      for (int i = 0; i < 3; ++i) {
        gp[qc + i] += Jp[i] * r[0] + Jp[3 + i] * r[1];
        for (int j = 0; j < 3; ++j) Vq[3 * i + j] += Jp[i] * Jp[j] + Jp[3 + i] * Jp[3 + j];
      }
    }
    if (pc >= 0) {
      const int c0 = pc - poff;
      for (int i = 0; i < 6; ++i) {
        b[c0 + i] += Jc[i] * r[0] + Jc[6 + i] * r[1];
        for (int j = 0; j <= i; ++j) S[(int64_t)(c0 + i) * nc + c0 + j] += Jc[i] * Jc[j] + Jc[6 + i] * Jc[6 + j];
      }
    }
  }
  for (int i = 0; i < nc; ++i) S[(int64_t)i * nc + i] += D[poff + i] * D[poff + i];
  for (int q = 0; q < npt; ++q) {
    double* Vq = V + 9 * q;
    for (int i = 0; i < 3; ++i) Vq[4 * i] += D[3 * q + i] * D[3 * q + i];
    double Vi[9];
    if (inv3_sym(Vq, Vi)) { free(V); free(gp); free(S); free(b); return 1; }
    memcpy(Vq, Vi, sizeof(Vi));
  }
  /* group observations by point to form W V^-1 W^T */
  int64_t* start = (int64_t*)calloc((size_t)npt + 2, sizeof(int64_t));
  int64_t* order = (int64_t*)malloc(sizeof(int64_t) * ((size_t)g->n_act + 1));
  for (int64_t a = 0; a < g->n_act; ++a) {
    const int qc = g->point_col[P->obs_point[g->act[a]]];
    if (qc >= 0) start[qc / 3 + 1]++;
  }
  for (int q = 0; q < npt; ++q) start[q + 1] += start[q];
  {
    int64_t* fill = (int64_t*)malloc(sizeof(int64_t) * ((size_t)npt + 1));
    memcpy(fill, start, sizeof(int64_t) * (size_t)npt);
    for (int64_t a = 0; a < g->n_act; ++a) {
      const int qc = g->point_col[P->obs_point[g->act[a]]];
      if (qc >= 0) order[fill[qc / 3]++] = a;
    }
    free(fill);
  }
  /* Every thread walks all landmarks and forms the products whose ROW block falls into its own range of poses:
   * one writer per word of S and b, and per word the same order of additions whatever the thread count. */
  const int n_workers = (g->O->num_threads > 1 && npt > 50000) ? g->O->num_threads : 1;
#pragma omp parallel for num_threads(n_workers) schedule(static, 1)
  for (int worker = 0; worker < n_workers; ++worker) {
  const int own_lo = poff + 6 * (int)(((int64_t)g->n_ap * worker) / n_workers);
  const int own_hi = poff + 6 * (int)(((int64_t)g->n_ap * (worker + 1)) / n_workers);
  for (int q = 0; q < npt; ++q) {
    const double* Vi = V + 9 * q;
    const int64_t s0 = start[q], s1 = start[q + 1];
    for (int64_t ii = s0; ii < s1; ++ii) {
      const int64_t a = order[ii];
      const int pci = g->pose_col[P->obs_pose[g->act[a]]];
      if (pci < own_lo || pci >= own_hi) continue;  /* (a constant pose has column -1) */
      const double* Jc = g->Jc + 12 * a;
      const double* Jp = g->Jp + 6 * a;
      double W[18], Y[18]; /* W = Jc^T Jp (6x3), Y = W V^-1 */
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 3; ++j) W[3 * i + j] = Jc[i] * Jp[j] + Jc[6 + i] * Jp[3 + j];
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 3; ++j)
          Y[3 * i + j] = W[3 * i] * Vi[j] + W[3 * i + 1] * Vi[3 + j] + W[3 * i + 2] * Vi[6 + j];
      const int ci = pci - poff;
      for (int i = 0; i < 6; ++i)
        b[ci + i] -= Y[3 * i] * gp[3 * q] + Y[3 * i + 1] * gp[3 * q + 1] + Y[3 * i + 2] * gp[3 * q + 2];
      for (int64_t jj = s0; jj < s1; ++jj) {
        const int64_t a2 = order[jj];
        const int pcj = g->pose_col[P->obs_pose[g->act[a2]]];
        if (pcj < 0 || pcj > pci) continue;
        const double* Jc2 = g->Jc + 12 * a2;
        const double* Jp2 = g->Jp + 6 * a2;
        const int cj = pcj - poff;
        for (int i = 0; i < 6; ++i)
          for (int j = 0; j < 6; ++j) {
            if (pcj == pci && j > i) continue;
            /* W2[j][:] = Jc2[:,j]^T Jp2 */
            double s = 0.0;
            for (int t = 0; t < 3; ++t) s += Y[3 * i + t] * (Jc2[j] * Jp2[t] + Jc2[6 + j] * Jp2[3 + t]);
            S[(int64_t)(ci + i) * nc + cj + j] -= s;
          }
      }
    }
  }
  }
  int rc = 0;
  if (nc > 0) {
    rc = chol_factor(S, nc, g->O->num_threads < 1 ? 1 : g->O->num_threads);
    if (!rc) chol_solve(S, nc, b);
  }
  if (!rc) {
    for (int i = 0; i < nc; ++i) y[poff + i] = b[i];
    /* y_p = V^-1 (g_p - W^T y_c) */
    for (int q = 0; q < npt; ++q) {
      double t[3] = {gp[3 * q], gp[3 * q + 1], gp[3 * q + 2]};
      for (int64_t ii = start[q]; ii < start[q + 1]; ++ii) {
        const int64_t a = order[ii];
        const int pci = g->pose_col[P->obs_pose[g->act[a]]];
        if (pci < 0) continue;
        const double* Jc = g->Jc + 12 * a;
        const double* Jp = g->Jp + 6 * a;
        const double* yc = b + (pci - poff);
        double m0 = 0.0, m1 = 0.0;
        for (int i = 0; i < 6; ++i) { m0 += Jc[i] * yc[i]; m1 += Jc[6 + i] * yc[i]; }
        for (int j = 0; j < 3; ++j) t[j] -= Jp[j] * m0 + Jp[3 + j] * m1;
      }
      const double* Vi = V + 9 * q;
      for (int i = 0; i < 3; ++i) y[3 * q + i] = Vi[3 * i] * t[0] + Vi[3 * i + 1] * t[1] + Vi[3 * i + 2] * t[2];
    }
  }
  free(V); free(gp); free(S); free(b); free(start); free(order);
  return rc;
}

static int prog_linear_solve(prog* g, const double* D, double* y) {
  const double t0 = now_s();
  int mode = g->O->linear_solver;
  if (mode == ORACLE_LIN_AUTO) mode = (g->n_tan <= 1500) ? ORACLE_LIN_FULL_NORMAL : ORACLE_LIN_SCHUR;
  int rc = (mode == ORACLE_LIN_FULL_NORMAL) ? solve_full(g, D, y) : solve_schur(g, D, y);
  if (!rc) for (int i = 0; i < g->n_tan; ++i) if (!isfinite(y[i])) { rc = 1; break; }
  g->t_lin += now_s() - t0;
  return rc;
}

/* model_cost_change = -(J s)^T (r + J s / 2) on the stored (scaled) Jacobian. */
static double prog_model_cost_change(const prog* g, const double* step) {
  const oracle_problem* P = g->P;
  double acc = 0.0;
  for (int64_t a = 0; a < g->n_act; ++a) {
    const int64_t k = g->act[a];
    const int pc = g->pose_col[P->obs_pose[k]];
    const int qc = g->point_col[P->obs_point[k]];
    double m0 = 0.0, m1 = 0.0;
    if (pc >= 0) {
      const double* J = g->Jc + 12 * a;
      for (int c = 0; c < 6; ++c) { m0 += J[c] * step[pc + c]; m1 += J[6 + c] * step[pc + c]; }
    }
    if (qc >= 0) {
      const double* J = g->Jp + 6 * a;
      for (int c = 0; c < 3; ++c) { m0 += J[c] * step[qc + c]; m1 += J[3 + c] * step[qc + c]; }
    }
    acc += m0 * (g->r[2 * a] + m0 / 2.0) + m1 * (g->r[2 * a + 1] + m1 / 2.0);
  }
  return -acc;
}

/* evaluator Plus over the active blocks; inactive blocks are copied through. */
static void prog_plus(const prog* g, const double* poses, const double* points, const double* delta,
                      double* poses_out, double* points_out) {
  memcpy(poses_out, poses, sizeof(double) * 7 * (size_t)g->n_poses);
  memcpy(points_out, points, sizeof(double) * 3 * (size_t)g->n_points);
  for (int i = 0; i < g->n_poses; ++i)
    if (g->pose_col[i] >= 0) oracle_se3_plus(poses + 7 * i, delta + g->pose_col[i], poses_out + 7 * i);
  for (int i = 0; i < g->n_points; ++i)
    if (g->point_col[i] >= 0)
      for (int c = 0; c < 3; ++c) points_out[3 * i + c] = points[3 * i + c] + delta[g->point_col[i] + c];
}

static double prog_x_norm(const prog* g, const double* poses, const double* points) {
  double s = 0.0;
  for (int i = 0; i < g->n_points; ++i)
    if (g->point_col[i] >= 0) for (int c = 0; c < 3; ++c) s += points[3 * i + c] * points[3 * i + c];
  for (int i = 0; i < g->n_poses; ++i)
    if (g->pose_col[i] >= 0) for (int c = 0; c < 7; ++c) s += poses[7 * i + c] * poses[7 * i + c];
  return sqrt(s);
}

static void prog_diff_norms(const prog* g, const double* pa, const double* qa, const double* pb,
                            const double* qb, double* l2, double* linf) {
  double s = 0.0, m = 0.0;
  for (int i = 0; i < g->n_points; ++i)
    if (g->point_col[i] >= 0)
      for (int c = 0; c < 3; ++c) { const double d = qa[3 * i + c] - qb[3 * i + c]; s += d * d; if (fabs(d) > m) m = fabs(d); }
  for (int i = 0; i < g->n_poses; ++i)
    if (g->pose_col[i] >= 0)
      for (int c = 0; c < 7; ++c) { const double d = pa[7 * i + c] - pb[7 * i + c]; s += d * d; if (fabs(d) > m) m = fabs(d); }
  *l2 = sqrt(s); *linf = m;
}

/* EvaluateGradientAndJacobian of the trust-region minimizer. */
static int lm_linearize(prog* g, const double* poses, const double* points, int iteration, double* cost,
                        double* gmax, double* gnorm, double* tp, double* tq, double* neg) {
  if (prog_evaluate(g, poses, points, 1, cost)) return 1;
  if (g->O->jacobi_scaling) {
    if (iteration == 0) {
      prog_colnorm2(g, g->scale);
      for (int i = 0; i < g->n_tan; ++i) g->scale[i] = 1.0 / (1.0 + sqrt(g->scale[i]));
    }
    prog_scale_columns(g);
  }
  for (int i = 0; i < g->n_tan; ++i) neg[i] = -g->grad[i];
  prog_plus(g, poses, points, neg, tp, tq);
  prog_diff_norms(g, poses, points, tp, tq, gnorm, gmax);
  return 0;
}

void oracle_options_default(oracle_options* o) {
  memset(o, 0, sizeof(*o));
  o->max_num_iterations = 50;
  o->loss = ORACLE_LOSS_NONE;
  o->loss_scale = 1.0;
  o->jacobian_mode = ORACLE_JAC_NUMERIC_CENTRAL;
  o->linear_solver = ORACLE_LIN_AUTO;
  o->num_threads = 1;
  o->jacobi_scaling = 1;
  o->param_tol_needs_success = 0;
  o->initial_trust_region_radius = 1e4;
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->max_num_consecutive_invalid_steps = 5;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
}

static double fixed_cost_of(const oracle_problem* P, const oracle_options* O) {
  double total = 0.0;
  for (int64_t k = 0; k < P->n_obs; ++k) {
    const int cc = P->pose_const ? P->pose_const[P->obs_pose[k]] : 0;
    const int qc = P->point_const ? P->point_const[P->obs_point[k]] : 0;
    if (cc && qc) {
      double r[2], c;
      eval_block(P, O->jacobian_mode, O->loss, O->loss_scale, 1, P->poses + 7 * (int64_t)P->obs_pose[k],
                 P->points + 3 * (int64_t)P->obs_point[k], P->obs_uv[2 * k], P->obs_uv[2 * k + 1], r, NULL,
                 NULL, NULL, &c);
      total += c;
    }
  }
  return total;
}

int oracle_solve(oracle_problem* P, const oracle_options* O, oracle_summary* S, oracle_iteration* log) {
  const double t_begin = now_s();
  prog g;
  memset(S, 0, sizeof(*S));
  if (prog_init(&g, P, O)) { prog_free(&g); S->termination_type = ORACLE_FAILURE; return -1; }
  S->num_active_poses = g.n_ap; S->num_active_points = g.n_apt; S->num_active_obs = g.n_act;
  S->fixed_cost = fixed_cost_of(P, O);
  const int n = g.n_tan;
  const size_t np7 = 7 * (size_t)P->n_poses + 1, nq3 = 3 * (size_t)P->n_points + 1;
  double* x_p = (double*)malloc(sizeof(double) * np7);
  double* x_q = (double*)malloc(sizeof(double) * nq3);
  double* c_p = (double*)malloc(sizeof(double) * np7);
  double* c_q = (double*)malloc(sizeof(double) * nq3);
  double* t_p = (double*)malloc(sizeof(double) * np7);
  double* t_q = (double*)malloc(sizeof(double) * nq3);
  double* diag = (double*)malloc(sizeof(double) * ((size_t)n + 1));
  double* D = (double*)malloc(sizeof(double) * ((size_t)n + 1));
  double* step = (double*)malloc(sizeof(double) * ((size_t)n + 1));
  double* delta = (double*)malloc(sizeof(double) * ((size_t)n + 1));
  double* neg = (double*)malloc(sizeof(double) * ((size_t)n + 1));
  memcpy(x_p, P->poses, sizeof(double) * 7 * (size_t)P->n_poses);
  memcpy(x_q, P->points, sizeof(double) * 3 * (size_t)P->n_points);

  int term = ORACLE_NO_CONVERGENCE;
  int n_log = 0;
  oracle_iteration it;
  memset(&it, 0, sizeof(it));
  double x_cost = 0.0, x_norm = prog_x_norm(&g, x_p, x_q);
  double radius = O->initial_trust_region_radius, decrease_factor = 2.0;
  int reuse_diagonal = 0, n_invalid = 0, at_least_one_success = 0;
  int solution_usable = 1;

  if (n == 0) {
    /* nothing to optimise: Ceres reports convergence with the fixed cost only */
    S->initial_cost = S->final_cost = S->fixed_cost;
    S->termination_type = ORACLE_CONVERGENCE;
    goto done;
  }

  /* IterationZero */
  if (lm_linearize(&g, x_p, x_q, 0, &x_cost, &it.gradient_max_norm, &it.gradient_norm, t_p, t_q, neg)) {
    term = ORACLE_FAILURE; solution_usable = 0;
    goto finish;
  }
  it.iteration = 0; it.step_is_valid = 1; it.step_is_successful = 1;
  it.cost = x_cost + S->fixed_cost; it.cost_change = 0.0; it.step_norm = 0.0; it.relative_decrease = 0.0;
  S->initial_cost = it.cost;

  for (;;) {
    /* FinalizeIterationAndCheckIfMinimizerCanContinue */
    if (it.step_is_successful) S->num_successful_steps++; else S->num_unsuccessful_steps++;
    it.trust_region_radius = radius;
    if (log) log[n_log] = it;
    n_log++;
    if (it.iteration >= O->max_num_iterations) { term = ORACLE_NO_CONVERGENCE; break; }
    if (it.step_is_successful && it.gradient_max_norm <= O->gradient_tolerance) { term = ORACLE_CONVERGENCE; break; }
    if (radius <= O->min_trust_region_radius) { term = ORACLE_CONVERGENCE; break; }

    const double prev_gnorm = it.gradient_norm, prev_gmax = it.gradient_max_norm;
    const int iter = it.iteration + 1;
    memset(&it, 0, sizeof(it));
    it.iteration = iter;

    /* ComputeTrustRegionStep: LevenbergMarquardtStrategy::ComputeStep */
    if (!reuse_diagonal) {
      prog_colnorm2(&g, diag);
      for (int i = 0; i < n; ++i) {
        double d = diag[i];
        if (d < O->min_lm_diagonal) d = O->min_lm_diagonal;
        if (d > O->max_lm_diagonal) d = O->max_lm_diagonal;
        diag[i] = d;
      }
    }
    for (int i = 0; i < n; ++i) D[i] = sqrt(diag[i] / radius);
    int lin_fail = prog_linear_solve(&g, D, step);
    reuse_diagonal = 1;
    double model_cost_change = 0.0;
    if (!lin_fail) {
      for (int i = 0; i < n; ++i) step[i] = -step[i];
      model_cost_change = prog_model_cost_change(&g, step);
      it.step_is_valid = model_cost_change > 0.0;
    }
    if (!it.step_is_valid) {
      /* HandleInvalidStep */
      if (++n_invalid >= O->max_num_consecutive_invalid_steps) { term = ORACLE_FAILURE; break; }
      radius *= 0.5; reuse_diagonal = 1;
      it.cost = x_cost + S->fixed_cost; it.cost_change = 0.0;
      it.gradient_max_norm = prev_gmax; it.gradient_norm = prev_gnorm;
      it.step_norm = 0.0; it.relative_decrease = 0.0; it.step_is_successful = 0;
      continue;
    }
    n_invalid = 0;
    for (int i = 0; i < n; ++i) delta[i] = step[i] * g.scale[i];

    /* ComputeCandidatePointAndEvaluateCost */
    double cand_cost;
    prog_plus(&g, x_p, x_q, delta, c_p, c_q);
    if (prog_evaluate(&g, c_p, c_q, 0, &cand_cost)) cand_cost = DBL_MAX;

    /* ParameterToleranceReached */
    {
      double linf;
      prog_diff_norms(&g, x_p, x_q, c_p, c_q, &it.step_norm, &linf);
      const double tol = O->parameter_tolerance * (x_norm + O->parameter_tolerance);
      if ((at_least_one_success || !O->param_tol_needs_success) && it.step_norm <= tol) {
        term = ORACLE_CONVERGENCE; break;
      }
    }
    /* FunctionToleranceReached */
    it.cost_change = x_cost - cand_cost;
    if (fabs(it.cost_change) <= O->function_tolerance * x_cost) { term = ORACLE_CONVERGENCE; break; }

    /* IsStepSuccessful (monotonic step evaluator) */
    if (cand_cost >= DBL_MAX) it.relative_decrease = -DBL_MAX;
    else it.relative_decrease = (x_cost - cand_cost) / model_cost_change;
    if (it.relative_decrease > O->min_relative_decrease) {
      /* HandleSuccessfulStep */
      memcpy(x_p, c_p, sizeof(double) * 7 * (size_t)P->n_poses);
      memcpy(x_q, c_q, sizeof(double) * 3 * (size_t)P->n_points);
      x_norm = prog_x_norm(&g, x_p, x_q);
      if (lm_linearize(&g, x_p, x_q, iter, &x_cost, &it.gradient_max_norm, &it.gradient_norm, t_p, t_q, neg)) {
        term = ORACLE_FAILURE; break;
      }
      it.cost = x_cost + S->fixed_cost;
      it.step_is_successful = 1;
      at_least_one_success = 1;
      double q = 1.0 - pow(2.0 * it.relative_decrease - 1.0, 3);
      if (q < 1.0 / 3.0) q = 1.0 / 3.0;
      radius = radius / q;
      if (radius > O->max_trust_region_radius) radius = O->max_trust_region_radius;
      decrease_factor = 2.0; reuse_diagonal = 0;
    } else {
      it.step_is_successful = 0;
      it.cost = cand_cost + S->fixed_cost;
      it.gradient_norm = prev_gnorm; it.gradient_max_norm = prev_gmax;
      radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = 1;
    }
  }

finish:
  S->final_cost = x_cost + S->fixed_cost;
  S->termination_type = term;
  S->num_iterations = n_log;
  /* ceres::Solver: user state is overwritten only when Summary::IsSolutionUsable() */
  if (solution_usable && term != ORACLE_FAILURE) {
    memcpy(P->poses, x_p, sizeof(double) * 7 * (size_t)P->n_poses);
    memcpy(P->points, x_q, sizeof(double) * 3 * (size_t)P->n_points);
  }
done:
  S->time_evaluate_s = g.t_eval; S->time_linear_solve_s = g.t_lin;
  S->time_total_s = now_s() - t_begin;
  free(x_p); free(x_q); free(c_p); free(c_q); free(t_p); free(t_q);
  free(diag); free(D); free(step); free(delta); free(neg);
  prog_free(&g);
  return 0;
}

int oracle_lm_step(const oracle_problem* P, const oracle_options* O, double radius, double* step_pose,
                   double* step_point, double* model_cost_change) {
  prog g;
  if (prog_init(&g, P, O)) { prog_free(&g); return -1; }
  const int n = g.n_tan;
  double* t_p = (double*)malloc(sizeof(double) * (7 * (size_t)P->n_poses + 1));
  double* t_q = (double*)malloc(sizeof(double) * (3 * (size_t)P->n_points + 1));
  double* diag = (double*)malloc(sizeof(double) * ((size_t)n + 1));
  double* D = (double*)malloc(sizeof(double) * ((size_t)n + 1));
  double* step = (double*)malloc(sizeof(double) * ((size_t)n + 1));
  double* neg = (double*)malloc(sizeof(double) * ((size_t)n + 1));
  double cost, gmax, gnorm;
  int rc = lm_linearize(&g, P->poses, P->points, 0, &cost, &gmax, &gnorm, t_p, t_q, neg);
  if (!rc) {
    prog_colnorm2(&g, diag);
    for (int i = 0; i < n; ++i) {
      double d = diag[i];
      if (d < O->min_lm_diagonal) d = O->min_lm_diagonal;
      if (d > O->max_lm_diagonal) d = O->max_lm_diagonal;
      D[i] = sqrt(d / radius);
    }
    rc = prog_linear_solve(&g, D, step);
  }
  if (!rc) {
    for (int i = 0; i < n; ++i) step[i] = -step[i];
    if (model_cost_change) *model_cost_change = prog_model_cost_change(&g, step);
    memset(step_pose, 0, sizeof(double) * 6 * (size_t)P->n_poses);
    memset(step_point, 0, sizeof(double) * 3 * (size_t)P->n_points);
    for (int i = 0; i < P->n_poses; ++i)
      if (g.pose_col[i] >= 0)
        for (int c = 0; c < 6; ++c) step_pose[6 * i + c] = step[g.pose_col[i] + c] * g.scale[g.pose_col[i] + c];
    for (int i = 0; i < P->n_points; ++i)
      if (g.point_col[i] >= 0)
        for (int c = 0; c < 3; ++c) step_point[3 * i + c] = step[g.point_col[i] + c] * g.scale[g.point_col[i] + c];
  }
  free(t_p); free(t_q); free(diag); free(D); free(step); free(neg);
  prog_free(&g);
  return rc;
}

/* ---- stereo triangulation (src/solve/triangulator.cpp:48-61) ---------------------------------- */
/* right singular vector of the smallest singular value of the 4x4 matrix whose COLUMNS are a[0..3]:
 * Hestenes' one-sided Jacobi, the method of cv::SVD for small matrices */
static void smallest_right_singular_vector(double a[4][4], double out[4]) {
  double v[4][4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) v[i][j] = i == j ? 1.0 : 0.0;
  const double eps = 2.0 * 2.220446049250313e-16;
  for (int sweep = 0; sweep < 30; ++sweep) {
    int changed = 0;
    for (int i = 0; i < 3; ++i) {
      for (int j = i + 1; j < 4; ++j) {
        double aa = 0.0, bb = 0.0, p = 0.0;
        for (int k = 0; k < 4; ++k) {
          aa += a[i][k] * a[i][k];
          bb += a[j][k] * a[j][k];
          p += a[i][k] * a[j][k];
        }
        if (fabs(p) <= eps * sqrt(aa * bb)) continue;
        p *= 2.0;
        const double beta = aa - bb, gamma = hypot(p, beta);
        double c, s;
        if (beta < 0.0) {
          const double delta = (gamma - beta) * 0.5;
          s = sqrt(delta / gamma);
          c = p / (gamma * s * 2.0);
        } else {
          c = sqrt((gamma + beta) / (gamma * 2.0));
          s = p / (gamma * c * 2.0);
        }
        for (int k = 0; k < 4; ++k) {
          const double x = a[i][k], y = a[j][k];
          a[i][k] = c * x + s * y;
          a[j][k] = c * y - s * x;
          const double vx = v[i][k], vy = v[j][k];
          v[i][k] = c * vx + s * vy;
          v[j][k] = c * vy - s * vx;
        }
        changed = 1;
      }
    }
    if (!changed) break;
  }
  int best = 0;
  double best_norm = 0.0;
  for (int i = 0; i < 4; ++i) {
    double nn = 0.0;
    for (int k = 0; k < 4; ++k) nn += a[i][k] * a[i][k];
    if (i == 0 || nn < best_norm) { best = i; best_norm = nn; }
  }
  for (int k = 0; k < 4; ++k) out[k] = v[best][k];
}

void oracle_triangulate_stereo(const double* Pl, const double* Pr, int64_t n, const float* luv,
                               const float* ruv, const double* pose7, double* out) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    const double xl = luv[2 * i], yl = luv[2 * i + 1], xr = ruv[2 * i], yr = ruv[2 * i + 1];
    double a[4][4];
    for (int k = 0; k < 4; ++k) {  /* cv::triangulatePoints: rows x P(2,:) - P(0,:), y P(2,:) - P(1,:) per camera */
      a[k][0] = xl * Pl[8 + k] - Pl[k];
      a[k][1] = yl * Pl[8 + k] - Pl[4 + k];
      a[k][2] = xr * Pr[8 + k] - Pr[k];
      a[k][3] = yr * Pr[8 + k] - Pr[4 + k];
    }
    double h[4];
    smallest_right_singular_vector(a, h);
    /* points_4d is CV_32F (triangulator.cpp:43-48 passes Point2f); convertPointsFromHomogeneous divides in float */
    const float X = (float)h[0], Y = (float)h[1], Z = (float)h[2], W = (float)h[3];
    const float scale = W != 0.f ? 1.f / W : 1.f;
    double p[3] = {(double)(X * scale), (double)(Y * scale), (double)(Z * scale)};
    if (pose7) {  /* triangulator.cpp:61: frame_prev_->pose_ * landmark */
      double R[9];
      const double qx = pose7[0], qy = pose7[1], qz = pose7[2], qw = pose7[3];
      R[0] = 1 - 2 * (qy * qy + qz * qz); R[1] = 2 * (qx * qy - qz * qw);     R[2] = 2 * (qx * qz + qy * qw);
      R[3] = 2 * (qx * qy + qz * qw);     R[4] = 1 - 2 * (qx * qx + qz * qz); R[5] = 2 * (qy * qz - qx * qw);
      R[6] = 2 * (qx * qz - qy * qw);     R[7] = 2 * (qy * qz + qx * qw);     R[8] = 1 - 2 * (qx * qx + qy * qy);
      const double x = p[0], y = p[1], z = p[2];
      p[0] = R[0] * x + R[1] * y + R[2] * z + pose7[4];
      p[1] = R[3] * x + R[4] * y + R[5] * z + pose7[5];
      p[2] = R[6] * x + R[7] * y + R[8] * z + pose7[6];
    }
    out[3 * i] = p[0]; out[3 * i + 1] = p[1]; out[3 * i + 2] = p[2];
  }
}
