// corridor_sequence.hpp -- a synthetic stereo-odometry-like sequence without the OpenCV frontend: every "frame"
// arrives with a drifting pose estimate and pixel observations of the landmarks in view, exactly the state the
// reference's pipeline is in when it reaches the bundle-adjustment stage (reference: src/frontend.cpp:65).  Shared
// by ba_stage_demo (one sequence) and ba_sequences (many at once).
#pragma once
#include <cmath>
#include <memory>
#include <random>
#include <vector>

#include "vo_types.hpp"

namespace VO {

class CorridorSequence {
public:
  static constexpr double fx = 718.856, fy = 718.856, cx = 607.1928, cy = 185.2157, width = 1241.0, height = 376.0;

  CorridorSequence(int n_frames, unsigned seed, int features_per_frame = 300)
      : n_frames_(n_frames), features_per_frame_(features_per_frame), gen_(seed), gt_z(n_frames) {
    // ground truth: forward motion along +z (camera z-forward, identity attitude), ~1 m per frame
    for (int i = 0; i < n_frames; ++i) gt_z[i] = (i ? gt_z[i - 1] : 0.0) + 0.8 + 0.4 * u01_(gen_);
    // landmarks scattered along the corridor
    const int n_landmarks = 60 * n_frames;
    landmarks_.resize(n_landmarks);
    gt_points_.resize(n_landmarks);
    for (int l = 0; l < n_landmarks; ++l) {
      // off the optical axis: under pure forward motion a point straight ahead shows no parallax, and without a
      // robust loss (optimizer.cpp:48 passes none) a two-view landmark of unobservable depth can run away
      const double side = u01_(gen_) < 0.5 ? -1.0 : 1.0;
      gt_points_[l] = Vector3d(side * (2.0 + 10.0 * u01_(gen_)), -3.0 + 6.0 * u01_(gen_), 4.0 + (gt_z.back() + 30.0) * u01_(gen_));
      landmarks_[l] = nullptr;
    }
  }

  Matx33d K() const {
    Matx33d k;
    k(0, 0) = fx; k(1, 1) = fy; k(0, 2) = cx; k(1, 2) = cy;
    return k;
  }

  // Frame i: an odometry-like estimate -- the previous frame's current pose (optimized or not) composed with the
  // true relative motion plus noise, so that errors accumulate unless bundle adjustment removes them -- and
  // features of the landmarks in view; landmarks seen for the first time are triangulated from the drifted estimate
  // and inserted into the map.
  std::shared_ptr<Frame> make_frame(int i, const std::shared_ptr<Frame>& prev, const std::shared_ptr<Map>& map) {
    auto frame = std::make_shared<Frame>();
    double* p = frame->pose_.data();
    p[0] = p[1] = p[2] = 0.0; p[3] = 1.0;
    if (prev) {
      // T_curr = T_prev * [exp(small rotation noise) | true step in the camera frame + noise]
      const double* q = prev->pose_.data();
      const double w[3] = {0.002 * n01_(gen_), 0.002 * n01_(gen_), 0.002 * n01_(gen_)};
      const double dq[4] = {0.5 * w[0], 0.5 * w[1], 0.5 * w[2], 1.0};
      double r[4] = {q[3] * dq[0] + q[0] * dq[3] + q[1] * dq[2] - q[2] * dq[1],
                     q[3] * dq[1] - q[0] * dq[2] + q[1] * dq[3] + q[2] * dq[0],
                     q[3] * dq[2] + q[0] * dq[1] - q[1] * dq[0] + q[2] * dq[3],
                     q[3] * dq[3] - q[0] * dq[0] - q[1] * dq[1] - q[2] * dq[2]};
      const double nr = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
      for (int k = 0; k < 4; ++k) p[k] = r[k] / nr;
      const double d[3] = {0.05 * n01_(gen_), 0.05 * n01_(gen_), (gt_z[i] - gt_z[i - 1]) + 0.05 * n01_(gen_)};
      // rotate d by q (v' = v + qw t + qv x t, t = 2 qv x v)
      const double t[3] = {2 * (q[1] * d[2] - q[2] * d[1]), 2 * (q[2] * d[0] - q[0] * d[2]), 2 * (q[0] * d[1] - q[1] * d[0])};
      p[4] = q[4] + d[0] + q[3] * t[0] + (q[1] * t[2] - q[2] * t[1]);
      p[5] = q[5] + d[1] + q[3] * t[1] + (q[2] * t[0] - q[0] * t[2]);
      p[6] = q[6] + d[2] + q[3] * t[2] + (q[0] * t[1] - q[1] * t[0]);
    } else {
      p[4] = p[5] = 0.0; p[6] = gt_z[0];
    }
    int seen = 0;
    const int n_landmarks = (int)landmarks_.size();
    for (int l = 0; l < n_landmarks && seen < features_per_frame_; ++l) {
      const double z = gt_points_[l].z() - gt_z[i];
      if (z < 4.0 || z > 30.0) continue;
      const double u = fx * gt_points_[l].x() / z + cx, v = fy * gt_points_[l].y() / z + cy;
      if (u < 0 || u >= width || v < 0 || v >= height) continue;
      if (!landmarks_[l]) {
        // triangulated from the drifted estimate: the same error the frame carries, plus depth noise
        const double zn = z * (1.0 + 0.003 * n01_(gen_));
        const double c[3] = {gt_points_[l].x() * zn / z, gt_points_[l].y() * zn / z, zn};
        const double tt[3] = {2 * (p[1] * c[2] - p[2] * c[1]), 2 * (p[2] * c[0] - p[0] * c[2]), 2 * (p[0] * c[1] - p[1] * c[0])};
        landmarks_[l] = std::make_shared<MapPoint>(Vector3d(p[4] + c[0] + p[3] * tt[0] + (p[1] * tt[2] - p[2] * tt[1]),
                                                            p[5] + c[1] + p[3] * tt[1] + (p[2] * tt[0] - p[0] * tt[2]),
                                                            p[6] + c[2] + p[3] * tt[2] + (p[0] * tt[1] - p[1] * tt[0])));
        map->insertLandmark(landmarks_[l]);
        ++created;
      }
      Point2f px;
      px.x = (float)(u + 0.15 * n01_(gen_)); px.y = (float)(v + 0.15 * n01_(gen_));
      frame->pushFeatureLeftWithLandmark(px, landmarks_[l]);
      ++seen;
    }
    return frame;
  }

  int created = 0;

private:
  int n_frames_, features_per_frame_;
  std::mt19937 gen_;
  std::normal_distribution<double> n01_{0.0, 1.0};
  std::uniform_real_distribution<double> u01_{0.0, 1.0};
  std::vector<std::shared_ptr<MapPoint>> landmarks_;
  std::vector<Vector3d> gt_points_;

public:
  std::vector<double> gt_z;
};

}  // namespace VO
