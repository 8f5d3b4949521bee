// stereo_kernels.hip -- the data-parallel step in front of bundle adjustment: linear triangulation of stereo matches.
//
// What the reference does per keyframe on the CPU (src/solve/triangulator.cpp:48-61):
//   cv::triangulatePoints(P_left, P_right, left, right, points_4d)   per match, the null vector of a 4x4 system
//   cv::convertPointsFromHomogeneous(points_4d.t(), points_3d)       divide by w
//   landmark = frame_prev_->pose_ * landmark                         lift into the world frame
// One lane per match here; the three steps never leave registers.
#include <hip/hip_runtime.h>

#include "hipba_internal.hpp"

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

}  // namespace hipba
