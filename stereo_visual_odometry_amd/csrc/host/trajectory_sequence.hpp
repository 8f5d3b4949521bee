// trajectory_sequence.hpp -- a stand-in for the image half of the reference's frontend (FAST, optical flow,
// triangulation, PnP: src/frontend.cpp:34-47, all OpenCV) that follows a GIVEN camera trajectory, e.g. the ground
// truth of a KITTI sequence: every "frame" arrives with a dead-reckoned pose estimate and pixel observations of the
// landmarks in view, which is the state the pipeline is in when it reaches the bundle-adjustment stage
// (src/frontend.cpp:65).  Landmarks are scattered in the frustum of every frame of the trajectory; a frame's estimate
// is the previous frame's current pose (optimized or not) composed with the true relative motion plus noise, so that
// the error accumulates unless bundle adjustment removes it.  corridor_sequence.hpp is the special case of a
// straight line.
#pragma once
#include <cmath>
#include <memory>
#include <random>
#include <vector>

#include "vo_types.hpp"

namespace VO {

namespace se3 {
// v' = R(q) v
inline void rotate(const double* q, const double* v, double* out) {
  const double t[3] = {2 * (q[1] * v[2] - q[2] * v[1]), 2 * (q[2] * v[0] - q[0] * v[2]), 2 * (q[0] * v[1] - q[1] * v[0])};
  out[0] = v[0] + q[3] * t[0] + (q[1] * t[2] - q[2] * t[1]);
  out[1] = v[1] + q[3] * t[1] + (q[2] * t[0] - q[0] * t[2]);
  out[2] = v[2] + q[3] * t[2] + (q[0] * t[1] - q[1] * t[0]);
}
inline void quat_mul(const double* a, const double* b, double* out) {
  out[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  out[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
  out[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
  out[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
// a * b for poses stored qx qy qz qw tx ty tz
inline SE3d mul(const SE3d& a, const SE3d& b) {
  SE3d r;
  quat_mul(a.p, b.p, r.p);
  const double n = std::sqrt(r.p[0] * r.p[0] + r.p[1] * r.p[1] + r.p[2] * r.p[2] + r.p[3] * r.p[3]);
  for (int k = 0; k < 4; ++k) r.p[k] /= n;
  double t[3];
  rotate(a.p, b.p + 4, t);
  for (int k = 0; k < 3; ++k) r.p[4 + k] = a.p[4 + k] + t[k];
  return r;
}
inline SE3d inverse(const SE3d& a) {
  SE3d r;
  r.p[0] = -a.p[0]; r.p[1] = -a.p[1]; r.p[2] = -a.p[2]; r.p[3] = a.p[3];
  const double m[3] = {-a.p[4], -a.p[5], -a.p[6]};
  rotate(r.p, m, r.p + 4);
  return r;
}
}  // namespace se3

class TrajectorySequence {
public:
  struct Noise {
    double rotation = 0.002;     // rad per frame, each axis
    double translation = 0.05;   // m per frame, each axis
    double depth = 0.003;        // relative, at triangulation
    double pixel = 0.15;         // px
  };

  TrajectorySequence(const std::vector<SE3d>& truth, const Matx33d& K, unsigned seed, int features_per_frame = 300,
                     double width = 1241.0, double height = 376.0)
      : truth_(truth), fx_(K(0, 0)), fy_(K(1, 1)), cx_(K(0, 2)), cy_(K(1, 2)), width_(width), height_(height),
        features_per_frame_(features_per_frame), gen_(seed) {
    // landmarks in the frustum of every frame, off the optical axis: under forward motion a point straight ahead
    // shows no parallax, and without a robust loss (optimizer.cpp:48 passes none) a landmark of unobservable depth
    // can run away
    const int per_frame = features_per_frame / 4 + 1;
    for (size_t i = 0; i < truth_.size(); ++i) {
      for (int k = 0; k < per_frame; ++k) {
        const double side = u01_(gen_) < 0.5 ? -1.0 : 1.0;
        const double u = cx_ + side * (0.25 + 0.70 * u01_(gen_)) * 0.5 * width_;
        const double v = 0.05 * height_ + 0.9 * height_ * u01_(gen_);
        const double z = 5.0 + 35.0 * u01_(gen_);
        const double c[3] = {(u - cx_) / fx_ * z, (v - cy_) / fy_ * z, z};
        double w[3];
        se3::rotate(truth_[i].p, c, w);
        Landmark lm;
        lm.world = Vector3d(w[0] + truth_[i].p[4], w[1] + truth_[i].p[5], w[2] + truth_[i].p[6]);
        lm.born = (int)i;
        landmarks_.push_back(lm);
      }
    }
  }

  int size() const { return (int)truth_.size(); }

  // Frame i of the trajectory: estimate, features; landmarks seen for the first time are triangulated from the
  // drifted estimate (the error the frame carries, plus depth noise) and inserted into the map.
  std::shared_ptr<Frame> make_frame(int i, const std::shared_ptr<Frame>& prev, const std::shared_ptr<Map>& map) {
    auto frame = std::make_shared<Frame>();
    if (prev) {
      SE3d step = se3::mul(se3::inverse(truth_[i - 1]), truth_[i]);
      SE3d jitter;
      jitter.p[0] = 0.5 * noise.rotation * n01_(gen_); jitter.p[1] = 0.5 * noise.rotation * n01_(gen_);
      jitter.p[2] = 0.5 * noise.rotation * n01_(gen_); jitter.p[3] = 1.0;
      for (int k = 0; k < 3; ++k) jitter.p[4 + k] = noise.translation * n01_(gen_);
      frame->pose_ = se3::mul(prev->pose_, se3::mul(step, jitter));
    } else {
      frame->pose_ = truth_[0];
    }
    const SE3d to_camera = se3::inverse(truth_[i]);
    int seen = 0;
    // landmarks born up to 45 frames back or 10 ahead can be in view
    const size_t per_frame = landmarks_.size() / truth_.size();
    const size_t lo = (size_t)std::max(0, i - 45) * per_frame, hi = std::min(landmarks_.size(), (size_t)(i + 11) * per_frame);
    for (size_t l = lo; l < hi && seen < features_per_frame_; ++l) {
      Landmark& lm = landmarks_[l];
      double c[3];
      se3::rotate(to_camera.p, lm.world.data(), c);
      for (int k = 0; k < 3; ++k) c[k] += to_camera.p[4 + k];
      if (c[2] < 3.0 || c[2] > 45.0) continue;
      const double u = fx_ * c[0] / c[2] + cx_, v = fy_ * c[1] / c[2] + cy_;
      if (u < 0 || u >= width_ || v < 0 || v >= height_) continue;
      if (!lm.point) {
        const double zn = 1.0 + noise.depth * n01_(gen_);
        const double cn[3] = {c[0] * zn, c[1] * zn, c[2] * zn};
        double w[3];
        se3::rotate(frame->pose_.p, cn, w);
        lm.point = std::make_shared<MapPoint>(Vector3d(w[0] + frame->pose_.p[4], w[1] + frame->pose_.p[5], w[2] + frame->pose_.p[6]));
        map->insertLandmark(lm.point);
        ++created;
      }
      Point2f px;
      px.x = (float)(u + noise.pixel * n01_(gen_)); px.y = (float)(v + noise.pixel * n01_(gen_));
      frame->pushFeatureLeftWithLandmark(px, lm.point);
      ++seen;
    }
    return frame;
  }

  Noise noise;
  int created = 0;

private:
  struct Landmark {
    Vector3d world;
    int born = 0;
    std::shared_ptr<MapPoint> point;
  };
  std::vector<SE3d> truth_;
  double fx_, fy_, cx_, cy_, width_, height_;
  int features_per_frame_;
  std::mt19937 gen_;
  std::normal_distribution<double> n01_{0.0, 1.0};
  std::uniform_real_distribution<double> u01_{0.0, 1.0};
  std::vector<Landmark> landmarks_;
};

}  // namespace VO
