/*
 * ba_oracle.h -- CPU restatement of the reference's bundle-adjustment path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped HIP library links, loads
 * or calls this code; it exists so that tests and the benchmark's CPU leg have
 * something to compare the GPU path with.
 *
 * PARITY UNPINNED: the reference's arithmetic for this path lives in Ceres
 * Solver (>= 2.1, unpinned) and Sophus (HEAD, unpinned), neither of which is
 * vendored in the reference tree or installed here, and the reference's own
 * ba_test asserts nothing and seeds its noise from std::random_device
 * (test/ba_test.cpp:169-170, 340-341).  This file therefore restates the
 * published algorithms of those libraries around the reference's own call
 * sites (src/solve/optimizer.cpp:14-63, include/visual_odometry/solve/
 * optimizer.hpp:30-62, test/ba_test.h:179-211, test/ba_test.cpp:235-276).
 */
#ifndef BA_ORACLE_H_
#define BA_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORACLE_Z_FORWARD = 0, ORACLE_X_FORWARD = 1 };        /* optimizer.hpp:56-57 / ba_test.h:205-206 */
enum { ORACLE_LOSS_NONE = 0, ORACLE_LOSS_HUBER = 1 };        /* optimizer.cpp:48 / ba_test.cpp:241      */
enum { ORACLE_JAC_NUMERIC_CENTRAL = 0, ORACLE_JAC_ANALYTIC = 1 };
enum { ORACLE_LIN_AUTO = 0, ORACLE_LIN_FULL_NORMAL = 1, ORACLE_LIN_SCHUR = 2 };

/* termination_type values follow ceres::TerminationType */
enum {
  ORACLE_CONVERGENCE = 0,
  ORACLE_NO_CONVERGENCE = 1,
  ORACLE_FAILURE = 2
};

typedef struct {
  int32_t n_poses;
  double* poses;              /* [n_poses*7] qx qy qz qw tx ty tz  (Sophus::SE3d::data(), frame.hpp:28) */
  const uint8_t* pose_const;  /* [n_poses] or NULL */
  int32_t n_points;
  double* points;             /* [n_points*3] (map_point.hpp:20) */
  const uint8_t* point_const; /* [n_points] or NULL */
  int64_t n_obs;
  const int32_t* obs_pose;    /* [n_obs] */
  const int32_t* obs_point;   /* [n_obs] */
  const double* obs_uv;       /* [n_obs*2] */
  double fx, fy, cx, cy;      /* camera.hpp:16 */
  int32_t camera_model;
} oracle_problem;

typedef struct {
  int32_t max_num_iterations;          /* 50 default; optimizer.cpp:58 sets 300 */
  int32_t loss;                        /* ORACLE_LOSS_* */
  double loss_scale;                   /* Huber a */
  int32_t jacobian_mode;               /* ORACLE_JAC_* */
  int32_t linear_solver;               /* ORACLE_LIN_* */
  int32_t num_threads;                 /* OpenMP threads for evaluation */
  int32_t jacobi_scaling;              /* 1 */
  int32_t param_tol_needs_success;     /* 0: upstream TrustRegionMinimizer::ParameterToleranceReached (tested on every
                                          valid step); 1: only after one accepted step (not an upstream rule) */
  double initial_trust_region_radius;  /* 1e4 */
  double max_trust_region_radius;      /* 1e16 */
  double min_trust_region_radius;      /* 1e-32 */
  double min_relative_decrease;        /* 1e-3 */
  double min_lm_diagonal;              /* 1e-6 */
  double max_lm_diagonal;              /* 1e32 */
  int32_t max_num_consecutive_invalid_steps; /* 5 */
  double function_tolerance;           /* 1e-6 */
  double gradient_tolerance;           /* 1e-10 */
  double parameter_tolerance;          /* 1e-8 */
} oracle_options;

typedef struct {
  int32_t iteration;
  int32_t step_is_valid;
  int32_t step_is_successful;
  double cost;                 /* includes fixed cost, as ceres::IterationSummary::cost */
  double cost_change;
  double gradient_max_norm;
  double gradient_norm;
  double step_norm;
  double relative_decrease;
  double trust_region_radius;
} oracle_iteration;

typedef struct {
  int32_t termination_type;
  int32_t num_iterations;       /* entries written to the iteration log */
  int32_t num_successful_steps;
  int32_t num_unsuccessful_steps;
  double initial_cost;
  double final_cost;
  double fixed_cost;
  double time_evaluate_s;
  double time_linear_solve_s;
  double time_total_s;
  int32_t num_active_poses;
  int32_t num_active_points;
  int64_t num_active_obs;
} oracle_summary;

void oracle_options_default(oracle_options* o);

/* One evaluation of the reference's functor (optimizer.hpp:30-62 / ba_test.h:179-211). */
void oracle_residual(int camera_model, double fx, double fy, double cx, double cy,
                     const double* pose7, const double* point3, double u, double v,
                     double* residual2);

/* Sophus pieces (sophus/se3.hpp, so3.hpp, ceres_manifold.hpp). */
void oracle_se3_exp(const double* tangent6, double* pose7);
void oracle_se3_mul(const double* a7, const double* b7, double* out7);
void oracle_se3_plus(const double* pose7, const double* delta6, double* out7);
void oracle_se3_inverse(const double* pose7, double* out7);
void oracle_se3_plus_jacobian(const double* pose7, double* j7x6_rowmajor);

/*
 * Evaluate every residual block at the problem's current state, as
 * ceres::Problem::Evaluate would with the manifold applied:
 *   residuals   [2*n_obs]
 *   jac_pose    [12*n_obs]  2x6 row-major, tangent order (v0 v1 v2 w0 w1 w2); NULL to skip
 *   jac_point   [6*n_obs]   2x3 row-major; NULL to skip
 *   jac_pose_ambient [14*n_obs] 2x7 row-major before the plus-Jacobian; NULL to skip
 * Constant blocks are differentiated too (the flags matter only to the solver).
 * If apply_loss != 0 the Huber corrector is applied to residuals and Jacobians
 * and *cost is 1/2 sum rho(|r|^2); otherwise *cost is 1/2 sum |r|^2.
 */
int oracle_evaluate(const oracle_problem* p, int jacobian_mode, int loss, double loss_scale,
                    int apply_loss, int num_threads, double* cost, double* residuals,
                    double* jac_pose, double* jac_point, double* jac_pose_ambient);

/*
 * The reference's solve: Ceres trust-region Levenberg-Marquardt with an exact
 * linear solve of the damped normal equations.  Poses/points are updated in
 * place when the solution is usable.  iteration_log may be NULL; otherwise it
 * must hold max_num_iterations+1 entries.
 */
int oracle_solve(oracle_problem* p, const oracle_options* o, oracle_summary* s,
                 oracle_iteration* iteration_log);

/*
 * One damped Gauss-Newton step at the current state, exposed so that the
 * full-normal-equation and Schur solves (and the GPU's) can be compared:
 *   step_pose  [6*n_poses], step_point [3*n_points]  (zeros for inactive blocks)
 * in the unscaled tangent space (i.e. the delta handed to Plus).
 */
int oracle_lm_step(const oracle_problem* p, const oracle_options* o, double radius,
                   double* step_pose, double* step_point, double* model_cost_change);

/*
 * Linear triangulation of stereo matches as Triangulator::triangulate does it
 * (src/solve/triangulator.cpp:48-61): cv::triangulatePoints, cv::convertPointsFromHomogeneous,
 * then frame pose * point.  OpenCV is not in this image (PARITY UNPINNED here as well): this restates
 * its published method -- per match the 4x4 system x P(2,:) - P(0,:), y P(2,:) - P(1,:) of both
 * cameras, the right singular vector of the smallest singular value by one-sided Jacobi rotations
 * (what cv::SVD runs on a small matrix), the homogeneous point stored as float like the float
 * pixels it came from, and the division by w in float (w == 0: undivided).
 * Pl, Pr: 3x4 row-major; luv, ruv: [2n] floats; pose7: qx qy qz qw tx ty tz or NULL; out: [3n].
 */
void oracle_triangulate_stereo(const double* Pl, const double* Pr, int64_t n, const float* luv,
                               const float* ruv, const double* pose7, double* out);

#ifdef __cplusplus
}
#endif
#endif
