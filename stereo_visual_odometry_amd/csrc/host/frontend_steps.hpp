// frontend_steps.hpp -- drop-ins for the two data-parallel steps either side of bundle adjustment in the reference's
// per-frame pipeline (src/frontend.cpp:34-47), same class names, members and in-place semantics, with the OpenCV
// calls replaced by the HIP engine's C ABI:
//   VO::Triangulator::triangulate(context)   src/solve/triangulator.cpp:14-61, include/.../solve/triangulator.hpp:12-29
//       cv::triangulatePoints + cv::convertPointsFromHomogeneous + the lift into the world frame
//       -> hipba_triangulate_stereo
//   VO::Estimator::estimate(context, K)      src/solve/estimator.cpp:13-73, include/.../solve/estimator.hpp:10-18
//       cv::solvePnPRansac(points_3d, points_2d, K, 0, rvec, tvec, false, 1000, 3.0, 0.99, inliers)
//       -> hipba_pnp_ransac without a prior pose
// The classes are templates over nothing: they use the member names of vo_types.hpp, which are the reference's.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "hipba.h"
#include "kitti.hpp"
#include "vo_types.hpp"

namespace VO {

// ---- the little of Sophus::SE3d the estimator needs (estimator.cpp:58-61) ---------------------------------------
// poses are SE3d storage: qx qy qz qw tx ty tz
inline void se3Rotate(const double* q, const double* v, double* out) {
  const double ux = 2 * (q[1] * v[2] - q[2] * v[1]), uy = 2 * (q[2] * v[0] - q[0] * v[2]), uz = 2 * (q[0] * v[1] - q[1] * v[0]);
  out[0] = v[0] + q[3] * ux + (q[1] * uz - q[2] * uy);
  out[1] = v[1] + q[3] * uy + (q[2] * ux - q[0] * uz);
  out[2] = v[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
}
inline SE3d se3Inverse(const SE3d& a) {
  SE3d r;
  r.p[0] = -a.p[0]; r.p[1] = -a.p[1]; r.p[2] = -a.p[2]; r.p[3] = a.p[3];
  double t[3];
  se3Rotate(r.p, a.p + 4, t);
  r.p[4] = -t[0]; r.p[5] = -t[1]; r.p[6] = -t[2];
  return r;
}
inline SE3d se3Multiply(const SE3d& a, const SE3d& b) {
  SE3d r;
  const double *p = a.p, *q = b.p;
  r.p[0] = p[3] * q[0] + p[0] * q[3] + p[1] * q[2] - p[2] * q[1];
  r.p[1] = p[3] * q[1] - p[0] * q[2] + p[1] * q[3] + p[2] * q[0];
  r.p[2] = p[3] * q[2] + p[0] * q[1] - p[1] * q[0] + p[2] * q[3];
  r.p[3] = p[3] * q[3] - p[0] * q[0] - p[1] * q[1] - p[2] * q[2];
  double t[3];
  se3Rotate(a.p, b.p + 4, t);
  r.p[4] = t[0] + a.p[4]; r.p[5] = t[1] + a.p[5]; r.p[6] = t[2] + a.p[6];
  return r;
}
// |log(T)| of Sophus::SE3d::log(): (upsilon, omega) with omega the rotation vector and upsilon = V^-1 t
inline double se3LogNorm(const SE3d& a) {
  const double* q = a.p;
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
  const double w = q[3];
  double two_atan_nbyw_by_n;  // sophus/so3.hpp, logAndTheta
  if (n < 1e-10) two_atan_nbyw_by_n = 2.0 / w - 2.0 / 3.0 * (n * n) / (w * w * w);
  else if (std::fabs(w) < 1e-10) two_atan_nbyw_by_n = (w > 0 ? M_PI : -M_PI) / n;
  else two_atan_nbyw_by_n = 2.0 * std::atan(n / w) / n;
  const double om[3] = {two_atan_nbyw_by_n * q[0], two_atan_nbyw_by_n * q[1], two_atan_nbyw_by_n * q[2]};
  const double theta = two_atan_nbyw_by_n * n;
  // V^-1 = I - Omega / 2 + c Omega^2, c = (1 - theta cos(theta / 2) / (2 sin(theta / 2))) / theta^2 (1 / 12 near zero)
  const double c = std::fabs(theta) < 1e-10 ? 1.0 / 12.0
                                            : (1.0 - theta * std::cos(0.5 * theta) / (2.0 * std::sin(0.5 * theta))) / (theta * theta);
  const double* t = a.p + 4;
  const double ot[3] = {om[1] * t[2] - om[2] * t[1], om[2] * t[0] - om[0] * t[2], om[0] * t[1] - om[1] * t[0]};
  const double oot[3] = {om[1] * ot[2] - om[2] * ot[1], om[2] * ot[0] - om[0] * ot[2], om[0] * ot[1] - om[1] * ot[0]};
  double s = 0.0;
  for (int i = 0; i < 3; ++i) {
    const double u = t[i] - 0.5 * ot[i] + c * oot[i];
    s += u * u + om[i] * om[i];
  }
  return std::sqrt(s);
}

// One engine handle for the steps of a pipeline (created on first use; the Optimizer keeps its own).
class EngineHandle {
public:
  EngineHandle() = default;
  EngineHandle(const EngineHandle&) = delete;
  EngineHandle& operator=(const EngineHandle&) = delete;
  ~EngineHandle() { if (handle_) hipba_destroy(handle_); }
  hipba_handle get() {
    if (!handle_) {
      if (hipba_version() != HIPBA_VERSION)
        throw std::runtime_error("libhipba.so is ABI version " + std::to_string(hipba_version()) +
                                 ", this adapter was compiled against version " + std::to_string(HIPBA_VERSION));
      if (hipba_create(device_, &handle_) != HIPBA_OK)
        throw std::runtime_error(std::string("hipba_create: ") + hipba_last_error());
    }
    return handle_;
  }
  int device_ = 0;

private:
  hipba_handle handle_ = nullptr;
};

class Triangulator {
public:
  Triangulator() = default;
  Triangulator(const Camera& camera_left, const Camera& camera_right, const std::shared_ptr<Map>& map,
               const std::shared_ptr<EngineHandle>& engine = std::make_shared<EngineHandle>())
      : map_(map), camera_left_(camera_left), camera_right_(camera_right), engine_(engine) {}

  // triangulator.cpp:34-61: the left features of the previous frame that have no landmark yet (they come last) are
  // triangulated against their right matches, lifted into the world frame by the frame's pose and entered in the map
  void triangulate(Context& context) {
    auto& frame = context.frame_prev_;
    const std::size_t start_idx = firstFeatureWithoutLandmark(frame->features_left_);
    if (start_idx == frame->features_left_.size()) return;
    const std::vector<Point2f> matches_2d_left = frame->getPointsLeft2D(start_idx);
    const std::vector<Point2f> matches_2d_right(frame->features_right_.begin() + start_idx, frame->features_right_.end());
    if (matches_2d_left.size() != matches_2d_right.size())
      throw std::runtime_error("Triangulator: left and right matches differ in number");
    std::vector<double> world(3 * matches_2d_left.size());
    if (hipba_triangulate_stereo(engine_->get(), camera_left_.P_.val, camera_right_.P_.val, (int64_t)matches_2d_left.size(),
                                 &matches_2d_left[0].x, &matches_2d_right[0].x, frame->pose_.data(), world.data()) != HIPBA_OK)
      throw std::runtime_error(std::string("hipba_triangulate_stereo: ") + hipba_last_error());
    for (std::size_t i = 0; i < matches_2d_left.size(); ++i) {
      std::shared_ptr<MapPoint>& landmark = frame->features_left_[i + start_idx]->landmark_;
      landmark = std::make_shared<MapPoint>(Vector3d(world[3 * i], world[3 * i + 1], world[3 * i + 2]));
      map_->insertLandmark(landmark);
      landmark->observations_.push_back(frame->features_left_[i + start_idx]);
    }
    if (verbose_) printf("%zu points triangulated\n", matches_2d_left.size());
  }

  std::shared_ptr<Map> map_;
  Camera camera_left_;
  Camera camera_right_;
  bool verbose_ = true;

private:
  static std::size_t firstFeatureWithoutLandmark(const std::vector<std::shared_ptr<Feature>>& features) {
    for (std::size_t i = 0; i < features.size(); ++i)
      if (!features[i]->landmark_) return i;
    return features.size();
  }
  std::shared_ptr<EngineHandle> engine_ = std::make_shared<EngineHandle>();
};

class Estimator {
public:
  Estimator() = default;
  explicit Estimator(const std::shared_ptr<EngineHandle>& engine) : engine_(engine) {}

  // estimator.cpp:13-73.  The current frame's pose from its 3D-2D matches by sample consensus, no prior pose; the
  // features that were not used are dropped; false -- and the frame's pose untouched -- when the motion since the
  // previous frame is implausibly large.  When no consensus is found OpenCV leaves rvec = tvec = 0 and no inliers, and
  // the reference goes on with that: every feature dropped, the identity as the pose to test.
  bool estimate(Context& context, const Matx33d& K) {
    auto& frame = context.frame_curr_;
    const std::vector<Point2f> points_2d = frame->getPointsLeft2D(0);
    const std::vector<Point3f> points_3d = frame->getPointsLeft3D();
    const int64_t n = (int64_t)points_2d.size();
    std::vector<double> pts(3 * (std::size_t)n), uv(2 * (std::size_t)n);
    for (int64_t i = 0; i < n; ++i) {
      pts[3 * i] = points_3d[i].x; pts[3 * i + 1] = points_3d[i].y; pts[3 * i + 2] = points_3d[i].z;
      uv[2 * i] = points_2d[i].x; uv[2 * i + 1] = points_2d[i].y;
    }
    std::vector<uint8_t> mask((std::size_t)n, 0);
    int64_t n_inliers = 0;
    SE3d curr_pose;  // identity: what rvec = tvec = 0 are
    if (n >= 4) {
      double pose[7] = {0, 0, 0, 0, 0, 0, 0};  // a zero quaternion: no extrinsic guess
      if (hipba_pnp_ransac(engine_->get(), K(0, 0), K(1, 1), K(0, 2), K(1, 2), HIPBA_CAMERA_Z_FORWARD, n, pts.data(), uv.data(),
                           pose, iterations_, reprojection_error_, confidence_, seed_++, /*refine=*/1, mask.data(), &n_inliers,
                           &last_iterations_) != HIPBA_OK)
        throw std::runtime_error(std::string("hipba_pnp_ransac: ") + hipba_last_error());
      if (n_inliers > 0)
        for (int i = 0; i < 7; ++i) curr_pose.p[i] = pose[i];  // camera-to-world: estimator.cpp:58-59's T.inverse()
    }
    // remove feature if it was not used to estimate motion (estimator.cpp:31-45)
    std::vector<std::shared_ptr<Feature>> left_feature_curr;
    std::vector<Point2f> right_feature_curr;
    left_feature_curr.reserve((std::size_t)n_inliers);
    right_feature_curr.reserve((std::size_t)n_inliers);
    for (int64_t i = 0; i < n; ++i) {
      if (!mask[(std::size_t)i]) continue;
      left_feature_curr.push_back(frame->features_left_[(std::size_t)i]);
      right_feature_curr.push_back(frame->features_right_[(std::size_t)i]);
    }
    frame->features_left_ = left_feature_curr;
    frame->features_right_ = right_feature_curr;
    if (verbose_) printf("%lld features used to estimate motion\n", (long long)n_inliers);
    const SE3d delta_pose = se3Multiply(curr_pose, se3Inverse(context.frame_prev_->pose_));
    if (se3LogNorm(delta_pose) > max_delta_pose_norm_) return false;
    frame->pose_ = curr_pose;
    return true;
  }

  double max_delta_pose_norm_{50.0};
  // the arguments the reference passes to cv::solvePnPRansac (estimator.cpp:28-29)
  int32_t iterations_ = 1000;
  double reprojection_error_ = 3.0;
  double confidence_ = 0.99;
  uint64_t seed_ = 0;            // one sample stream per call: the same sequence of frames gives the same poses
  int32_t last_iterations_ = 0;  // what the confidence bound let the last call's sequential loop run
  bool verbose_ = true;

private:
  std::shared_ptr<EngineHandle> engine_ = std::make_shared<EngineHandle>();
};

}  // namespace VO
