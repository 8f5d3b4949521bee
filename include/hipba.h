/*
 * hipba.h -- C ABI of the MI355X (gfx950) bundle-adjustment engine.
 *
 * This is the boundary a maintainer of the reference binds to in place of the
 * ceres::Problem / ceres::Solve calls inside VO::Optimizer::optimize
 * (reference: src/solve/optimizer.cpp:14-63).  Plain pointers and sizes only.
 *
 * Conventions shared by every entry point
 *   - poses are 7 doubles each, qx qy qz qw tx ty tz, exactly the memory of
 *     Sophus::SE3d::data() (reference: include/visual_odometry/sensor/frame.hpp:28,
 *     src/solve/optimizer.cpp:27); the pose is T_world<-camera.
 *   - landmarks are 3 doubles each (reference: sensor/map_point.hpp:20,
 *     src/solve/optimizer.cpp:21).
 *   - an observation is (pose index, landmark index, u, v) -- one
 *     AddResidualBlock of src/solve/optimizer.cpp:47-50; pixels are doubles as at
 *     optimizer.cpp:40-41.
 *   - every function returns HIPBA_OK (0) or a negative HIPBA_ERR_* code and
 *     never throws; hipba_last_error() gives the text for the calling thread.
 *   - caller memory is never retained past the call that received it.
 *   - results are written back only when the solution is usable, as Ceres does
 *     (termination CONVERGENCE or NO_CONVERGENCE); otherwise inputs stay untouched.
 */
#ifndef HIPBA_H_
#define HIPBA_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bumped whenever a struct of this header changes size or layout; hipba_version() returns the library's value */
#define HIPBA_VERSION 104

/* status codes */
enum {
  HIPBA_OK = 0,
  HIPBA_ERR_INVALID_ARGUMENT = -1,
  HIPBA_ERR_NO_DEVICE = -2,        /* no usable gfx950 device: there is no CPU fallback */
  HIPBA_ERR_HIP = -3,              /* a HIP runtime call failed */
  HIPBA_ERR_OUT_OF_MEMORY = -4,
  HIPBA_ERR_UNSUPPORTED = -5,      /* e.g. a batched window of more than 32 variable poses */
  HIPBA_ERR_COMM = -6,             /* the caller's all-reduce callback failed */
  HIPBA_ERR_INTERNAL = -7          /* an unexpected C++ exception was caught at the boundary */
};

/* reference: optimizer.hpp:56-57 (z forward, production) vs test/ba_test.h:205-206 (x forward) */
enum { HIPBA_CAMERA_Z_FORWARD = 0, HIPBA_CAMERA_X_FORWARD = 1 };
/* reference: optimizer.cpp:48 passes no loss; test/ba_test.cpp:241 uses ceres::HuberLoss(scale) */
enum { HIPBA_LOSS_NONE = 0, HIPBA_LOSS_HUBER = 1 };
/* how the reduced camera system is solved */
enum { HIPBA_LINEAR_AUTO = 0, HIPBA_LINEAR_DENSE_CHOLESKY = 1, HIPBA_LINEAR_PCG = 2 };
/* ceres::TerminationType */
enum { HIPBA_CONVERGENCE = 0, HIPBA_NO_CONVERGENCE = 1, HIPBA_FAILURE = 2 };

/* A landmark may have any number of observations (the reference adds one residual block per feature,
   optimizer.cpp:37-51) in a single window of more than 20 variable poses; in smaller or batched windows a
   landmark is limited to this many (more: HIPBA_ERR_UNSUPPORTED) -- with at most 32 poses in such a window that
   only excludes a pose observing one landmark many times over. */
#define HIPBA_MAX_POINT_DEGREE 256

/*
 * One window: what VO::Optimizer::optimize walks out of VO::Map
 * (reference: optimizer.cpp:19-52; sensor/map.hpp:21-22).
 */
typedef struct hipba_problem {
  int32_t n_poses;
  double* poses;              /* [7*n_poses] in/out */
  const uint8_t* pose_const;  /* [n_poses] or NULL; 1 = SetParameterBlockConstant (optimizer.cpp:30) */
  int32_t n_points;
  double* points;             /* [3*n_points] in/out */
  const uint8_t* point_const; /* [n_points] or NULL (optimizer.cpp:33) */
  int64_t n_obs;
  const int32_t* obs_pose;    /* [n_obs] */
  const int32_t* obs_point;   /* [n_obs] */
  const double* obs_uv;       /* [2*n_obs] */
  double fx, fy, cx, cy;      /* K(0,0) K(1,1) K(0,2) K(1,2) (sensor/camera.hpp:16) */
  int32_t camera_model;       /* HIPBA_CAMERA_* */
} hipba_problem;

/*
 * ceres::Solver::Options fields the reference relies on, same names, Ceres'
 * defaults (reference: optimizer.cpp:54-58 overrides max_num_iterations = 300;
 * test/ba_test.cpp:269-271 keeps 50).
 */
typedef struct hipba_options {
  int32_t max_num_iterations;               /* 50 */
  int32_t loss;                             /* HIPBA_LOSS_NONE */
  double loss_scale;                        /* 1.0 */
  int32_t linear_solver;                    /* HIPBA_LINEAR_AUTO */
  int32_t jacobi_scaling;                   /* 1 */
  double initial_trust_region_radius;       /* 1e4 */
  double max_trust_region_radius;           /* 1e16 */
  double min_trust_region_radius;           /* 1e-32 */
  double min_relative_decrease;             /* 1e-3 */
  double min_lm_diagonal;                   /* 1e-6 */
  double max_lm_diagonal;                   /* 1e32 */
  int32_t max_num_consecutive_invalid_steps;/* 5 */
  double function_tolerance;                /* 1e-6 */
  double gradient_tolerance;                /* 1e-10 */
  double parameter_tolerance;               /* 1e-8 */
  int32_t pcg_max_iterations;               /* 500 */
  double pcg_tolerance;                     /* 1e-12, relative residual |r|/|b| */
  int32_t minimizer_progress_to_stdout;     /* 0; 1 prints a Ceres-like table (optimizer.cpp:56) */
  /* 0 (default): TrustRegionMinimizer::ParameterToleranceReached as upstream Ceres 2.0-2.2 has it -- the test
     |step| <= parameter_tolerance * (|x| + parameter_tolerance) runs on every valid step, also the first one.
     1: the test is skipped until one step has been accepted.  That is NOT an upstream Ceres rule; it is kept
     only so that the difference can be measured (a window that starts converged runs on until the function
     tolerance or the trust-region radius ends it). */
  int32_t parameter_tolerance_needs_success;
  /* 0; 1 brackets the phases of every iteration with HIP events and fills the time_*_evaluation_s /
     time_linear_solver_s fields of the summary (the split ceres::Solver::Summary::FullReport prints,
     optimizer.cpp:62); costs a few event records per iteration */
  int32_t report_time_breakdown;
  int32_t reserved;
} hipba_options;

/* ceres::IterationSummary, the fields the reference prints */
typedef struct hipba_iteration {
  int32_t iteration;
  int32_t step_is_valid;
  int32_t step_is_successful;
  int32_t linear_solver_iterations;
  double cost;
  double cost_change;
  double gradient_max_norm;
  double gradient_norm;
  double step_norm;
  double relative_decrease;
  double trust_region_radius;
} hipba_iteration;

/* ceres::Solver::Summary, reduced to what is measurable here */
typedef struct hipba_summary {
  int32_t termination_type;        /* HIPBA_CONVERGENCE ... */
  int32_t num_iterations;          /* entries in the iteration log, iteration 0 included */
  int32_t num_successful_steps;
  int32_t num_unsuccessful_steps;
  double initial_cost;
  double final_cost;
  double fixed_cost;
  int32_t num_active_poses;
  int32_t num_active_points;
  int64_t num_active_observations;
  double time_setup_s;             /* host analysis + upload */
  double time_solve_s;             /* LM loop on the device */
  double time_total_s;
  int32_t linear_solver_used;      /* HIPBA_LINEAR_DENSE_CHOLESKY or HIPBA_LINEAR_PCG */
  int32_t num_linear_solves;       /* trust-region steps computed; Ceres does not log the step on which a
                                      parameter/function-tolerance test ends the solve, so this can be
                                      num_iterations (= log entries) rather than num_iterations - 1 */
  /* optional per-iteration log: caller-provided storage for iteration_log_capacity entries.  These two are INPUTS:
     zero the struct (hipba_summary s = {0};) before a call that wants no log */
  hipba_iteration* iteration_log;
  int32_t iteration_log_capacity;
  int32_t reserved2;
  /* bytes this call moved over the host link (for a batched call: the whole call, in every summary) */
  int64_t bytes_host_to_device;
  int64_t bytes_device_to_host;
  /* FullReport's "Time (in seconds)" split, device time summed over the iterations; filled (else 0) when
     options.report_time_breakdown is set.  Residual evaluation = the cost passes; Jacobian evaluation = the
     camera-block pass (the landmark side of J is recomputed inside the elimination, which Ceres books under
     its Schur linear solver as well); linear solver = landmark elimination + reduced solve + back-substitution
     + retraction.  A batched call reports the whole batch in every summary. */
  double time_residual_evaluation_s;
  double time_jacobian_evaluation_s;
  double time_linear_solver_s;
} hipba_summary;

typedef struct hipba_handle_s* hipba_handle;          /* one per calling thread and GPU */
typedef struct hipba_device_problem_s* hipba_device_problem;

/*
 * All-reduce over the ranks of a job, supplied by the caller (RCCL through
 * torch.distributed, or a wrapper around a raw ncclComm_t).  buf is DEVICE memory
 * holding count doubles; op is HIPBA_REDUCE_SUM or HIPBA_REDUCE_MAX; stream is the
 * hipStream_t the call belongs to (the callee must order its collective after
 * work already queued on it, and leave the result visible to later work on it).
 * That is the engine's stream for most calls and a second stream of the handle for
 * the pieces of the reduced matrix, which travel while the factorisation runs; the
 * calls come from one thread, in the same order on every rank, and the callee need
 * not wait for the host.  Return 0 on success.
 */
enum { HIPBA_REDUCE_SUM = 0, HIPBA_REDUCE_MAX = 1 };
typedef int (*hipba_allreduce_fn)(void* ctx, double* buf, size_t count, int op, void* stream);

typedef struct hipba_comm {
  hipba_allreduce_fn allreduce;
  void* ctx;
  int32_t rank;
  int32_t world_size;
} hipba_comm;

/* ---- library ---------------------------------------------------------------------------- */
int hipba_version(void);
const char* hipba_last_error(void);
void hipba_options_default(hipba_options* options);

/* ---- handle: owns the device, one stream and grow-only device buffers ------------------- */
/* number of visible GPUs (for callers that spread handles over them; HIPBA_ERR_NO_DEVICE if there is none) */
int hipba_device_count(int* count);
int hipba_create(int device, hipba_handle* out);
int hipba_destroy(hipba_handle h);
/* hipStream_t of the handle, for callers that order their own work (or a collective) on it */
void* hipba_stream(hipba_handle h);

/*
 * Replaces ceres::Solve on one window (reference: optimizer.cpp:61): analyses the
 * problem on the host, uploads it, runs Levenberg-Marquardt on the GPU, writes
 * poses/points back in place when the solution is usable.
 */
int hipba_solve(hipba_handle h, hipba_problem* problem, const hipba_options* options,
                hipba_summary* summary);

/*
 * Many independent windows in one call (small sliding windows batched on one
 * GPU).  Each window keeps its own trust region, acceptance test and
 * termination; summaries[i] belongs to problems[i].
 */
int hipba_solve_batched(hipba_handle h, int32_t n_windows, hipba_problem* problems,
                        const hipba_options* options, hipba_summary* summaries);

/*
 * ceres::Problem::Evaluate for the same residual blocks: residuals [2*n_obs],
 * jac_pose [12*n_obs] (2x6 row-major in the SE3 tangent space v0 v1 v2 w0 w1 w2,
 * i.e. ambient Jacobian x Sophus::Manifold<SE3>::PlusJacobian), jac_point
 * [6*n_obs] (2x3 row-major).  Any output may be NULL.  With apply_loss != 0 the
 * robust-loss corrector is applied as Ceres' ResidualBlock::Evaluate does and
 * *cost = 1/2 sum rho(|r|^2); otherwise *cost = 1/2 sum |r|^2.  Constant flags
 * are ignored (every block is differentiated).
 */
int hipba_evaluate(hipba_handle h, const hipba_problem* problem, const hipba_options* options,
                   int apply_loss, double* cost, double* residuals, double* jac_pose,
                   double* jac_point);

/*
 * hipba_evaluate for callers that CANNOT ALLOCATE device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  No counterpart in the reference: it exists for those callers only.
 * Every array argument is a DEVICE pointer the caller owns (same shapes and meaning as hipba_evaluate; any
 * output may be NULL; d_cost receives one double); constant flags do not exist here.  The call allocates and
 * frees nothing, on the host link or the device: the one buffer its kernels need beyond inputs and outputs (a
 * partial cost sum per 256 observations) is a fixed array inside the library, 512 KB on every GPU the library
 * runs on, which limits one call to HIPBA_EVALUATE_DEVICE_MAX_OBS observations (more: HIPBA_ERR_UNSUPPORTED).
 * The work is QUEUED on the handle's stream (hipba_stream) and the call returns without waiting; outputs are
 * valid once that stream has reached this point.  All calls on one GPU share the array: calls that overlap on
 * one GPU (two handles, two streams) race on the cost sum -- serialise them.  Indices cannot be checked on the
 * host: an observation whose pose or landmark index is out of range yields NaN outputs and a NaN cost.
 */
#define HIPBA_EVALUATE_DEVICE_MAX_OBS (1 << 24)
int hipba_evaluate_device(hipba_handle h, int32_t n_poses, const double* d_poses, int32_t n_points,
                          const double* d_points, int64_t n_obs, const int32_t* d_obs_pose,
                          const int32_t* d_obs_point, const double* d_obs_uv, double fx, double fy, double cx,
                          double cy, int32_t camera_model, const hipba_options* options, int apply_loss,
                          double* d_cost, double* d_residuals, double* d_jac_pose, double* d_jac_point);

/* ---- device-resident problems: upload once, solve/reset many times ---------------------- */
int hipba_upload(hipba_handle h, int32_t n_windows, const hipba_problem* problems,
                 hipba_device_problem* out);
/*
 * Upload ONE window's shard for hipba_solve_device_sharded: all poses of the window (identical on
 * every rank) and this rank's landmarks with all of their observations.  Unlike hipba_upload every
 * non-constant pose takes part in the reduced system even if this rank holds none of its
 * observations, so that all ranks lay the reduced camera matrix out identically.  pose_order: see
 * hipba_pose_order below.
 */
int hipba_upload_shard(hipba_handle h, const hipba_problem* shard, const int32_t* pose_order,
                       hipba_device_problem* out);
/*
 * The order in which hipba_upload would lay the poses of a one-window problem out in the reduced camera system
 * (what Ceres' ordering step does for SPARSE_NORMAL_CHOLESKY, optimizer.cpp:55): co-visible poses next to each
 * other, so that the dense factorisation can skip the part of the matrix that stays zero.  order [n_poses]
 * receives a permutation of the pose indices.  A caller that shards a problem computes it ONCE on the whole
 * problem and hands the same order to every rank's hipba_upload_shard (pose_order; NULL keeps the natural order):
 * a shard sees the couplings of its own landmarks only and cannot choose for the others.
 */
int hipba_pose_order(const hipba_problem* problem, int32_t* order);
/* restore the parameters uploaded by hipba_upload (device-to-device copy) */
int hipba_reset(hipba_handle h, hipba_device_problem dp);
/* LM on the resident problem; summaries [n_windows] */
int hipba_solve_device(hipba_handle h, hipba_device_problem dp, const hipba_options* options,
                       hipba_summary* summaries);
/*
 * As hipba_solve_device for ONE window whose observations are a shard of a larger
 * problem: every rank holds all poses and a disjoint subset of the landmarks with
 * all of their observations.  The reduced camera matrix, its right-hand side and
 * the per-iteration scalars are summed over ranks through comm.
 */
int hipba_solve_device_sharded(hipba_handle h, hipba_device_problem dp, const hipba_comm* comm,
                               const hipba_options* options, hipba_summary* summary);
/* copy current device parameters of window w back: poses [7*n_poses], points [3*n_points] */
int hipba_download(hipba_handle h, hipba_device_problem dp, int32_t window, double* poses,
                   double* points);
int hipba_free_problem(hipba_handle h, hipba_device_problem dp);

/*
 * The pose parameterisation's Plus on n poses at once: out[i] = poses[i] * exp(deltas[i]), what
 * Sophus::Manifold<Sophus::SE3>::Plus does for the blocks added at src/solve/optimizer.cpp:17,27 (tangent order
 * upsilon then omega; small-angle series below |omega| = 1e-10; Sophus' first-order renormalisation of the product's
 * quaternion).  HOST arrays: poses [7 n], deltas [6 n], out [7 n]; computed on the device by the same function the
 * solver's pose update calls.
 */
int hipba_se3_plus(hipba_handle h, int64_t n, const double* poses, const double* deltas, double* out);

/* ---- the steps either side of bundle adjustment in the per-frame pipeline ------------------- */
/*
 * Linear triangulation of stereo matches, what Triangulator::triangulate does for every keyframe
 * (src/solve/triangulator.cpp:48-61): cv::triangulatePoints(P_left, P_right, left, right) -- per match the null
 * vector of the 4x4 system x P(2,:) - P(0,:), y P(2,:) - P(1,:) of both cameras --, then
 * cv::convertPointsFromHomogeneous (the homogeneous point rounded to float as OpenCV stores it, divided by w in
 * float; w == 0 leaves it undivided), then frame_prev_->pose_ * point.  P_left / P_right: the 3x4 projection
 * matrices, row-major (sensor/camera.hpp: Camera::P_); left_uv / right_uv: n pixel pairs as cv::Point2f lays them
 * out; pose: qx qy qz qw tx ty tz of the frame (Sophus::SE3d::data()), or NULL to stay in the camera frame;
 * points: [3 n] doubles.  Points at negative depth or infinity come out as they are, as in the reference.
 * hipba_triangulate_stereo takes HOST arrays and returns when points is filled; the _device form takes DEVICE
 * pointers for every argument, allocates nothing, queues its one kernel on the handle's stream and returns.
 */
int hipba_triangulate_stereo(hipba_handle h, const double P_left[12], const double P_right[12], int64_t n,
                             const float* left_uv, const float* right_uv, const double* pose, double* points);
int hipba_triangulate_stereo_device(hipba_handle h, const double* d_P_left, const double* d_P_right, int64_t n,
                                    const float* d_left_uv, const float* d_right_uv, const double* d_pose,
                                    double* d_points);
/*
 * Pose-only refinement on 3D-2D matches: the minimisation of the reprojection error over one camera pose that ends
 * cv::solvePnPRansac (src/solve/estimator.cpp:28-29), as a bundle adjustment with every landmark constant -- same
 * residual, same SE3 parameterisation and the same Levenberg-Marquardt as hipba_solve.  The RANSAC hypothesis
 * search stays with the caller: `use` is its inlier mask ([n] or NULL = all).  points: [3 n] world coordinates,
 * uv: [2 n] pixels, pose: in = the hypothesis to refine, out = the refined camera-to-world pose (what
 * estimator.cpp:58-59 calls curr_pose).  HIPBA_ERR_INVALID_ARGUMENT with fewer than 3 matches in use.
 */
int hipba_refine_pose(hipba_handle h, double fx, double fy, double cx, double cy, int32_t camera_model, int64_t n,
                      const double* points, const double* uv, const uint8_t* use, double pose[7],
                      const hipba_options* options, hipba_summary* summary);

/*
 * The whole of cv::solvePnPRansac(points_3d, points_2d, K, coeffs, rvec, tvec, false, iterations, reprojection_error,
 * confidence, inliers) as the reference calls it (src/solve/estimator.cpp:28-29: no extrinsic guess, 1000 iterations,
 * 3.0 px, 0.99).  pose: out = the result, camera-to-world as estimator.cpp:58-59 has it; in = either a quaternion of
 * four zeros -- no prior, as in the reference: every minimal sample (four matches) is solved in closed form, the
 * perspective-three-point problem on its first three matches with the fourth choosing among the up to four
 * solutions, and polished by Gauss-Newton on all four -- or a pose near the answer (frame_prev_->pose_: the frames of
 * a sequence are a fraction of a metre apart), from which the samples are then fitted by Gauss-Newton alone.
 * All `iterations` hypotheses are evaluated at once; the answer is the one OpenCV's sequential loop would end with
 * on the same hypotheses: they are taken in the order of their numbers, one with more matches inside
 * reprojection_error than every earlier one (and at least four) becomes the best and shrinks the number of
 * iterations still to run to log(1 - confidence) / log(1 - (inlier share)^4) (cv::RANSACUpdateNumIters), and
 * hypotheses past that bound do not count.  confidence in [0, 1]; 1 keeps (in practice) every hypothesis in play.
 * *iterations_run (may be NULL) = that bound at the end, what the loop would have run.  `inliers` ([n] bytes, may be
 * NULL) marks the winner's matches, *n_inliers counts them; with refine != 0 the pose is then refined on them
 * (hipba_refine_pose).  The sampling is a counter-based generator: the same seed gives the same hypotheses on any
 * machine.  A match counts as an inlier when its squared reprojection error is <= reprojection_error^2 (OpenCV's
 * test).  When no hypothesis gathers at least four inliers, or the winner is not finite, the call does what
 * cv::solvePnPRansac does when it returns false: pose is left as it came in, the mask is cleared and *n_inliers = 0
 * (the status is still HIPBA_OK: test *n_inliers).  Not result-identical to OpenCV: other random samples, and
 * OpenCV's minimal solver for five and more matches is EPnP on five of them.  HIPBA_ERR_INVALID_ARGUMENT with fewer
 * than 4 matches or a confidence outside [0, 1].
 */
int hipba_pnp_ransac(hipba_handle h, double fx, double fy, double cx, double cy, int32_t camera_model, int64_t n,
                     const double* points, const double* uv, double pose[7], int32_t iterations,
                     double reprojection_error, double confidence, uint64_t seed, int32_t refine, uint8_t* inliers,
                     int64_t* n_inliers, int32_t* iterations_run);

/* ---- native RCCL communicator for sharded solves ------------------------------------------ */
/*
 * A hipba_comm whose all-reduce is ncclAllReduce (RCCL over xGMI) queued on the engine's stream: for C / C++
 * callers of hipba_solve_device_sharded (the reference is C++, src/solve/optimizer.cpp), one process per GPU.
 * Rank 0 asks for an id and hands the 128 bytes to the other ranks by whatever means the job has (MPI, a file, a
 * socket); then every rank calls hipba_rccl_comm_create (collective: it returns when all ranks have called it).
 * RCCL is loaded with dlopen on first use; HIPBA_ERR_COMM if it is not there or a call fails.
 */
#define HIPBA_RCCL_ID_BYTES 128
int hipba_rccl_get_unique_id(uint8_t id[HIPBA_RCCL_ID_BYTES]);
int hipba_rccl_comm_create(int device, const uint8_t id[HIPBA_RCCL_ID_BYTES], int32_t rank, int32_t world_size,
                           hipba_comm* out);
/* all-reduce calls made through comm so far and the bytes they carried */
int hipba_rccl_comm_stats(const hipba_comm* comm, int64_t* calls, int64_t* bytes);
int hipba_rccl_comm_destroy(hipba_comm* comm);

/* ---- measurement hooks (bench / profiling; no counterpart in the reference) ------------- */
enum {
  HIPBA_KERNEL_RESIDUAL = 0,          /* cost pass                                  */
  HIPBA_KERNEL_RESIDUAL_JACOBIAN = 1, /* one thread per observation, r + J to HBM   */
  HIPBA_KERNEL_CAMERA_BLOCKS = 2,     /* 6x6 camera blocks + gradient               */
  HIPBA_KERNEL_POINT_SCHUR = 3,       /* 3x3 inversion + Schur complement           */
  HIPBA_KERNEL_BACKSUB = 4,           /* landmark back-substitution + model change  */
  HIPBA_KERNEL_REDUCED_SOLVE = 5,     /* Cholesky / PCG on the reduced system       */
  HIPBA_KERNEL_COUNT = 6
};
/*
 * Floating-point operations of one Cholesky factorisation of the resident problem's largest reduced camera
 * system: as a dense matrix ((6 n_c)^3 / 3), and inside the envelope the library's pose order leaves (what the
 * factorisation actually has to do; equal to the dense figure for windows solved in LDS or without an envelope).
 */
int hipba_reduced_solve_flops(hipba_device_problem dp, double* flops_dense, double* flops_envelope);
/*
 * Launches one hot-path kernel `repeats` times on the resident problem's current
 * state and returns the average duration of one launch in milliseconds measured
 * with HIP events on the handle's stream.  Does not change the parameters.
 */
int hipba_time_kernel(hipba_handle h, hipba_device_problem dp, const hipba_options* options,
                      int32_t kernel, int32_t repeats, double* avg_ms);

#ifdef __cplusplus
}
#endif
#endif /* HIPBA_H_ */
