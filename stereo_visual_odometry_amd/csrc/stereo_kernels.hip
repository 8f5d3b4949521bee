// stereo_kernels.hip -- the data-parallel step in front of bundle adjustment: linear triangulation of stereo matches.
//
// What the reference does per keyframe on the CPU (src/solve/triangulator.cpp:48-61):
//   cv::triangulatePoints(P_left, P_right, left, right, points_4d)   per match, the null vector of a 4x4 system
//   cv::convertPointsFromHomogeneous(points_4d.t(), points_3d)       divide by w
//   landmark = frame_prev_->pose_ * landmark                         lift into the world frame
// One lane per match here; the three steps never leave registers.
#include <hip/hip_runtime.h>

#include <cfloat>

#include "ba_math.hpp"
#include "hipba_internal.hpp"
#include "p3p.hpp"

namespace hipba {

// The right singular vector of the smallest singular value of a 4x4 matrix by one-sided Jacobi rotations (Hestenes):
// pairs of columns are rotated until all are orthogonal; the rotations accumulate in V; the singular values are the
// columns' norms.  (cv::triangulatePoints takes the last row of V^T of cv::SVD, which for small matrices is this
// very method.)  A is column-major a[col][row].
__device__ __forceinline__ void smallest_right_singular_vector(double a[4][4], double out[4]) {
  double v[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[i][j] = i == j ? 1.0 : 0.0;
  }
  const double eps = 2.0 * 2.220446049250313e-16;
  for (int sweep = 0; sweep < 30; ++sweep) {
    bool changed = false;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = i + 1; j < 4; ++j) {
        double aa = 0.0, bb = 0.0, p = 0.0;
#pragma unroll
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
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const double x = a[i][k], y = a[j][k];
          a[i][k] = c * x + s * y;
          a[j][k] = c * y - s * x;
          const double vx = v[i][k], vy = v[j][k];
          v[i][k] = c * vx + s * vy;
          v[j][k] = c * vy - s * vx;
        }
        changed = true;
      }
    }
    if (!changed) break;
  }
  int best = 0;
  double best_norm = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    double nn = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) nn += a[i][k] * a[i][k];
    if (i == 0 || nn < best_norm) { best = i; best_norm = nn; }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i == best) {
#pragma unroll
      for (int k = 0; k < 4; ++k) out[k] = v[i][k];
    }
  }
}

// pl, pr: the two 3x4 projection matrices, row-major.  pose: qx qy qz qw tx ty tz (camera to world) or null.
__global__ __launch_bounds__(256) void k_triangulate_stereo(const double* __restrict__ pl, const double* __restrict__ pr,
                                                            long long n, const float2* __restrict__ left,
                                                            const float2* __restrict__ right, const double* __restrict__ pose,
                                                            double* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float2 l = left[i], r = right[i];
  // rows of the system: x P(2,:) - P(0,:) and y P(2,:) - P(1,:) for each camera (the pixels are cv::Point2f: float)
  double a[4][4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    a[k][0] = (double)l.x * pl[8 + k] - pl[k];
    a[k][1] = (double)l.y * pl[8 + k] - pl[4 + k];
    a[k][2] = (double)r.x * pr[8 + k] - pr[k];
    a[k][3] = (double)r.y * pr[8 + k] - pr[4 + k];
  }
  double h[4];
  smallest_right_singular_vector(a, h);
  // points_4d has the pixels' type, float; the division is convertPointsFromHomogeneous' (w == 0: left as it is)
  const float X = (float)h[0], Y = (float)h[1], Z = (float)h[2], W = (float)h[3];
  const float scale = W != 0.f ? 1.f / W : 1.f;
  double p[3] = {(double)(X * scale), (double)(Y * scale), (double)(Z * scale)};
  if (pose) {
    // Sophus::SE3d * point: R(q) p + t
    const double qx = pose[0], qy = pose[1], qz = pose[2], qw = pose[3];
    const double tx = 2.0 * (qy * p[2] - qz * p[1]), ty = 2.0 * (qz * p[0] - qx * p[2]), tz = 2.0 * (qx * p[1] - qy * p[0]);
    const double rx = p[0] + qw * tx + (qy * tz - qz * ty);
    const double ry = p[1] + qw * ty + (qz * tx - qx * tz);
    const double rz = p[2] + qw * tz + (qx * ty - qy * tx);
    p[0] = rx + pose[4]; p[1] = ry + pose[5]; p[2] = rz + pose[6];
  }
  out[3 * i] = p[0]; out[3 * i + 1] = p[1]; out[3 * i + 2] = p[2];
}

hipError_t launch_triangulate_stereo(const double* d_pl, const double* d_pr, long long n, const float* d_left,
                                     const float* d_right, const double* d_pose, double* d_out, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_triangulate_stereo, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_pl, d_pr, n,
                     reinterpret_cast<const float2*>(d_left), reinterpret_cast<const float2*>(d_right), d_pose, d_out);
  return hipGetLastError();
}

// ---- the manifold's Plus on arrays of poses -------------------------------------------------------------------------
// Sophus::Manifold<SE3>::Plus(x, delta) = x * SE3::exp(delta) (optimizer.cpp:17,27), one lane per pose: the same
// se3_plus the solver's pose update calls, reachable on its own so that it can be checked value by value.
__global__ __launch_bounds__(256) void k_se3_plus(long long n, const double* __restrict__ poses,
                                                  const double* __restrict__ deltas, double* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double x[7], d[6], y[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) x[k] = poses[7 * i + k];
#pragma unroll
  for (int k = 0; k < 6; ++k) d[k] = deltas[6 * i + k];
  se3_plus(x, d, y);
#pragma unroll
  for (int k = 0; k < 7; ++k) out[7 * i + k] = y[k];
}

hipError_t launch_se3_plus(long long n, const double* d_poses, const double* d_deltas, double* d_out, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_se3_plus, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, d_poses, d_deltas, d_out);
  return hipGetLastError();
}

// ---- PnP by random sample consensus (src/solve/estimator.cpp:28-29: cv::solvePnPRansac) -----------------------------
// One lane per hypothesis: four matches drawn with a counter-based generator (the same draw for the same seed and
// hypothesis number wherever it runs); the pose of the first three in closed form (p3p.hpp), the fourth choosing among
// its up to four solutions -- or, when the caller has a prior (consecutive frames of a sequence: the previous frame's
// pose), that prior --; Gauss-Newton on the four matches' reprojection error from there.  A second launch counts the
// matches every pose reprojects within the threshold.  A third walks the hypotheses in the order of their numbers the
// way OpenCV's sequential loop does (a better one shrinks the number of iterations still to run by the confidence
// bound, hypotheses past the bound do not count), takes the best (the lowest number among equals) and marks its
// inliers.  Every hypothesis is evaluated, all at once; the bound only decides which of them the answer may come from.
// Three launches: the fits (a lane per hypothesis), their support (a workgroup per hypothesis), the choice.
__host__ __device__ __forceinline__ unsigned long long splitmix64(unsigned long long& state) {
  unsigned long long z = (state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

constexpr int kPnpSample = 4;
constexpr int kPnpGaussNewton = 10;

__global__ __launch_bounds__(64) void k_pnp_hypotheses(CameraK K, long long n, const double* __restrict__ pts,
                                                       const double* __restrict__ uv, const double* __restrict__ prior,
                                                       int closed_form, int n_hyp, unsigned long long seed,
                                                       int* __restrict__ count, double* __restrict__ poses) {
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= n_hyp) return;
  unsigned long long state = seed + 0xD1B54A32D192ED03ull * (unsigned long long)(h + 1);
  long long idx[kPnpSample];
  for (int k = 0; k < kPnpSample; ++k) {
    for (int attempt = 0; attempt < 16; ++attempt) {
      idx[k] = (long long)(splitmix64(state) % (unsigned long long)n);
      bool fresh = true;
      for (int j = 0; j < k; ++j) fresh = fresh && idx[j] != idx[k];
      if (fresh) break;
    }
  }
  double pose[7];
  if (closed_form) {
    double P3[9], uv3[6], cand[28];
    for (int k = 0; k < 3; ++k) {
      for (int i = 0; i < 3; ++i) P3[3 * k + i] = pts[3 * idx[k] + i];
      uv3[2 * k] = uv[2 * idx[k]]; uv3[2 * k + 1] = uv[2 * idx[k] + 1];
    }
    const int n_cand = p3p_poses(K, P3, uv3, cand);
    int pick = -1;
    double pick_err = 0.0;
    for (int c = 0; c < n_cand; ++c) {
      double r[2];
      residual_only(K, cand + 7 * c, pts + 3 * idx[3], uv[2 * idx[3]], uv[2 * idx[3] + 1], r);
      const double e = r[0] * r[0] + r[1] * r[1];
      if (e == e && (pick < 0 || e < pick_err)) { pick = c; pick_err = e; }
    }
    if (pick < 0) {
      // no pose puts the three matches on their pixels (or they are collinear): a hypothesis without support
      count[h] = -1;
      for (int i = 0; i < 7; ++i) poses[7ll * h + i] = i == 3 ? 1.0 : 0.0;
      return;
    }
    for (int i = 0; i < 7; ++i) pose[i] = cand[7 * pick + i];
  } else {
#pragma unroll
    for (int i = 0; i < 7; ++i) pose[i] = prior[i];
  }
  for (int it = 0; it < kPnpGaussNewton; ++it) {
    double H[21], g[6];
#pragma unroll
    for (int i = 0; i < 21; ++i) H[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) g[i] = 0.0;
    for (int k = 0; k < kPnpSample; ++k) {
      double r[2], Jc[12], Jp[6];
      residual_jacobian(K, pose, pts + 3 * idx[k], uv[2 * idx[k]], uv[2 * idx[k] + 1], r, Jc, Jp);
      int q = 0;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int j = i; j < 6; ++j) { H[q] += Jc[i] * Jc[j] + Jc[6 + i] * Jc[6 + j]; ++q; }
        g[i] += Jc[i] * r[0] + Jc[6 + i] * r[1];
      }
    }
    // Cholesky of the 6x6 normal matrix (upper triangle packed row by row), a whisker of damping against a
    // degenerate sample; a sample it cannot factor keeps the pose it has.  (Every loop has constant bounds and is
    // unrolled: with a data-dependent exit the matrix lived in scratch memory and the fit took three times as long.)
    double L[6][6];
    bool ok = true;
    {
      int q = 0;
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i; j < 6; ++j) { L[j][i] = H[q] + (i == j ? 1e-9 * (1.0 + H[q]) : 0.0); ++q; }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      double d = L[j][j];
#pragma unroll
      for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
      ok = ok && d > 0.0 && isfinite(d);
      d = sqrt(d);
      L[j][j] = d;
#pragma unroll
      for (int i = j + 1; i < 6; ++i) {
        double v = L[i][j];
#pragma unroll
        for (int k = 0; k < j; ++k) v -= L[i][k] * L[j][k];
        L[i][j] = v / d;
      }
    }
    if (!ok) break;
    double y[6], dlt[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      double v = -g[i];
#pragma unroll
      for (int k = 0; k < i; ++k) v -= L[i][k] * y[k];
      y[i] = v / L[i][i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
      double v = y[i];
#pragma unroll
      for (int k = i + 1; k < 6; ++k) v -= L[k][i] * dlt[k];
      dlt[i] = v / L[i][i];
    }
    double next[7], step2 = 0.0;
    se3_plus(pose, dlt, next);
#pragma unroll
    for (int i = 0; i < 7; ++i) pose[i] = next[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) step2 += dlt[i] * dlt[i];
    if (!(step2 > 1e-24)) break;  // converged (or not a number: the count below is zero then)
  }
  count[h] = 0;  // (k_pnp_count fills it in)
#pragma unroll
  for (int i = 0; i < 7; ++i) poses[7ll * h + i] = pose[i];
}

// The support of every hypothesis: one workgroup per hypothesis, its lanes striding over the matches (a lane per
// hypothesis walking all matches alone took ten times the fit itself).  A hypothesis without a pose (count -1) gets 0.
__global__ __launch_bounds__(256) void k_pnp_count(CameraK K, long long n, const double* __restrict__ pts,
                                                   const double* __restrict__ uv, const double* __restrict__ poses, double thr2,
                                                   int* __restrict__ count) {
  __shared__ int partial[4];
  const int h = blockIdx.x;
  if (count[h] < 0) {  // (uniform over the workgroup)
    __syncthreads();   // every lane has read the mark before lane 0 overwrites it
    if (threadIdx.x == 0) count[h] = 0;
    return;
  }
  double pose[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) pose[i] = poses[7ll * h + i];
  int c = 0;
  for (long long j = threadIdx.x; j < n; j += 256) {
    double r[2];
    residual_only(K, pose, pts + 3 * j, uv[2 * j], uv[2 * j + 1], r);
    c += (r[0] * r[0] + r[1] * r[1] <= thr2) ? 1 : 0;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
  if ((threadIdx.x & 63) == 0) partial[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) count[h] = partial[0] + partial[1] + partial[2] + partial[3];
}

// cv::RANSACUpdateNumIters (OpenCV's modules/calib3d/src/ptsetreg.cpp): iterations after which a sample free of
// outliers has been drawn with probability p when a fraction ep of the matches are outliers; never more than max_iters
__host__ __device__ inline int ransac_update_iterations(double p, double ep, int model_points, int max_iters) {
  p = fmin(fmax(p, 0.0), 1.0);
  ep = fmin(fmax(ep, 0.0), 1.0);
  double num = fmax(1.0 - p, DBL_MIN);
  double denom = 1.0 - pow(1.0 - ep, (double)model_points);
  if (denom < DBL_MIN) return 0;
  num = log(num);
  denom = log(denom);
  return denom >= 0.0 || -num >= max_iters * (-denom) ? max_iters : (int)rint(num / denom);
}

// out: [0..6] the winning hypothesis' pose, [7] its inlier count (as a double; 0 = no hypothesis with a consensus of
// at least the minimal sample), [8] the number of iterations the sequential loop would have run; mask: one byte per
// match
constexpr int kPnpTile = 1024;
__global__ __launch_bounds__(256) void k_pnp_select(CameraK K, long long n, const double* __restrict__ pts,
                                                    const double* __restrict__ uv, int n_hyp, const int* __restrict__ count,
                                                    const double* __restrict__ poses, double thr2, double confidence,
                                                    double* __restrict__ out, unsigned char* __restrict__ mask) {
  __shared__ int tile[kPnpTile];
  __shared__ int s_best, s_best_count, s_limit;
  const int t = threadIdx.x;
  if (t == 0) { s_best = -1; s_best_count = kPnpSample - 1; s_limit = n_hyp; }
  __syncthreads();
  for (int base = 0; base < n_hyp; base += kPnpTile) {
    if (base >= s_limit) break;  // (uniform: read behind a barrier)
    for (int i = t; i < kPnpTile && base + i < n_hyp; i += 256) tile[i] = count[base + i];
    __syncthreads();
    if (t == 0) {
      int best = s_best, bc = s_best_count, limit = s_limit;
      for (int i = 0; i < kPnpTile && base + i < limit; ++i) {
        const int c = tile[i];
        if (c > bc) {
          bc = c;
          best = base + i;
          limit = ransac_update_iterations(confidence, (double)(n - c) / (double)n, kPnpSample, limit);
        }
      }
      s_best = best; s_best_count = bc; s_limit = limit;
    }
    __syncthreads();
  }
  const int best = s_best;
  double pose[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) pose[i] = best >= 0 ? poses[7ll * best + i] : (i == 3 ? 1.0 : 0.0);
  if (t < 7) out[t] = pose[t];
  if (t == 7) out[7] = best >= 0 ? (double)s_best_count : 0.0;
  if (t == 8) out[8] = (double)s_limit;
  for (long long j = t; j < n; j += 256) {
    double r[2];
    residual_only(K, pose, pts + 3 * j, uv[2 * j], uv[2 * j + 1], r);
    mask[j] = (best >= 0 && r[0] * r[0] + r[1] * r[1] <= thr2) ? 1 : 0;
  }
}

hipError_t launch_pnp_ransac(const CameraK& K, long long n, const double* d_pts, const double* d_uv, const double* d_prior,
                             int closed_form, int n_hyp, unsigned long long seed, double reprojection_error,
                             double confidence, int* d_count, double* d_poses, double* d_out, unsigned char* d_mask,
                             hipStream_t s) {
  const double thr2 = reprojection_error * reprojection_error;
  hipLaunchKernelGGL(k_pnp_hypotheses, dim3((unsigned)((n_hyp + 63) / 64)), dim3(64), 0, s, K, n, d_pts, d_uv, d_prior,
                     closed_form, n_hyp, seed, d_count, d_poses);
  hipLaunchKernelGGL(k_pnp_count, dim3((unsigned)n_hyp), dim3(256), 0, s, K, n, d_pts, d_uv, (const double*)d_poses, thr2, d_count);
  hipLaunchKernelGGL(k_pnp_select, dim3(1), dim3(256), 0, s, K, n, d_pts, d_uv, n_hyp, (const int*)d_count,
                     (const double*)d_poses, thr2, confidence, d_out, d_mask);
  return hipGetLastError();
}

}  // namespace hipba
