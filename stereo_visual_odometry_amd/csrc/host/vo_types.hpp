// vo_types.hpp -- the slice of the reference's data model that bundle adjustment touches, with
// dependency-free stand-ins for the Sophus / Eigen / OpenCV value types (none of which are installed
// here).  Member names and memory layouts are the reference's, so code written against these types
// compiles unchanged against the real headers:
//   VO::Map       landmarks_, frames_, insertFrame, insertLandmark,
//                 cleanMap                                           (sensor/map.hpp:13-25, map.cpp:8-50)
//   VO::Frame     pose_, features_left_, features_right_, pushFeatureLeft[WithLandmark], getPointsLeft2D / 3D
//                                                                    (sensor/frame.hpp:15-39, frame.cpp:5-50)
//   VO::Context   frame_prev_, frame_curr_                           (sensor/context.hpp:13-27, without the images)
//   VO::Feature   frame_, pixel, landmark_                           (sensor/feature.hpp:15-33)
//   VO::MapPoint  pose_, observations_                               (sensor/map_point.hpp:12-22)
//   SE3d::data()  7 doubles qx qy qz qw tx ty tz                     (Sophus::SE3d as used at optimizer.cpp:27)
#pragma once
#include <array>
#include <cmath>
#include <deque>
#include <memory>
#include <unordered_set>
#include <vector>

namespace VO {

struct Point2f { float x = 0.f, y = 0.f; };                       // cv::Point2f
struct Point3f { float x = 0.f, y = 0.f, z = 0.f; };              // cv::Point3f
struct Matx33d {                                                   // cv::Matx33d (row-major)
  double val[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double operator()(int r, int c) const { return val[3 * r + c]; }
  double& operator()(int r, int c) { return val[3 * r + c]; }
};
struct Vector3d {                                                  // Eigen::Vector3d
  double v[3] = {0, 0, 0};
  Vector3d() = default;
  Vector3d(double x, double y, double z) : v{x, y, z} {}
  double* data() { return v; }
  const double* data() const { return v; }
  double x() const { return v[0]; } double y() const { return v[1]; } double z() const { return v[2]; }
};
struct SE3d {                                                      // Sophus::SE3d storage
  double p[7] = {0, 0, 0, 1, 0, 0, 0};
  static constexpr int num_parameters = 7;
  double* data() { return p; }
  const double* data() const { return p; }
  // T^-1 * point (what the harness needs; rotation by the conjugate quaternion)
  Vector3d inverseApply(const Vector3d& w) const {
    const double d[3] = {w.v[0] - p[4], w.v[1] - p[5], w.v[2] - p[6]};
    const double qx = -p[0], qy = -p[1], qz = -p[2], qw = p[3];
    const double ux = 2 * (qy * d[2] - qz * d[1]), uy = 2 * (qz * d[0] - qx * d[2]), uz = 2 * (qx * d[1] - qy * d[0]);
    return Vector3d(d[0] + qw * ux + (qy * uz - qz * uy), d[1] + qw * uy + (qz * ux - qx * uz),
                    d[2] + qw * uz + (qx * uy - qy * ux));
  }
};

class Feature;
class Frame;

class MapPoint {
public:
  MapPoint() = default;
  explicit MapPoint(const Vector3d& pose) : pose_(pose) {}
  Vector3d pose_;
  std::vector<std::weak_ptr<Feature>> observations_;
};

class Feature {
public:
  Feature() = default;
  Feature(const std::weak_ptr<Frame>& frame, const Point2f& point_2d, const std::shared_ptr<MapPoint>& landmark)
      : frame_(frame), pixel(point_2d), landmark_(landmark) {}
  std::weak_ptr<Frame> frame_;
  Point2f pixel;
  std::shared_ptr<MapPoint> landmark_{nullptr};
};

class Frame : public std::enable_shared_from_this<Frame> {
public:
  std::vector<std::shared_ptr<Feature>> features_left_;
  std::vector<Point2f> features_right_;
  SE3d pose_;
  // frame.cpp:5-8, 27-37: the pixels of the left features from start_idx on
  std::vector<Point2f> getPointsLeft2D(std::size_t start_idx) const {
    std::vector<Point2f> points;
    points.reserve(features_left_.size() - start_idx);
    for (std::size_t i = start_idx; i < features_left_.size(); ++i) points.push_back(features_left_[i]->pixel);
    return points;
  }
  // frame.cpp:10-13, 39-50: the landmarks of the left features, in single precision as cv::Point3f holds them
  std::vector<Point3f> getPointsLeft3D() const {
    std::vector<Point3f> points;
    points.reserve(features_left_.size());
    for (const auto& feature : features_left_) {
      Point3f q;
      q.x = (float)feature->landmark_->pose_.x(); q.y = (float)feature->landmark_->pose_.y(); q.z = (float)feature->landmark_->pose_.z();
      points.push_back(q);
    }
    return points;
  }
  void pushFeatureLeft(const Point2f& point_2d) {  // frame.cpp:15-18
    features_left_.push_back(std::make_shared<Feature>(shared_from_this(), point_2d, nullptr));
  }
  void pushFeatureLeftWithLandmark(const Point2f& point_2d, const std::shared_ptr<MapPoint>& landmark) {
    auto f = std::make_shared<Feature>(shared_from_this(), point_2d, landmark);
    features_left_.push_back(f);
    landmark->observations_.push_back(f);
  }
};

class Context {  // sensor/context.hpp:13-27 (the frames only: images stay with the OpenCV frontend)
public:
  std::shared_ptr<Frame> frame_prev_{nullptr};
  std::shared_ptr<Frame> frame_curr_{nullptr};
  void insertFrame(const std::shared_ptr<Frame>& frame) { frame_prev_ = frame_curr_; frame_curr_ = frame; }
};

class Map {
public:
  void insertFrame(const std::shared_ptr<Frame>& key_frame) {
    frames_.push_back(key_frame);
    if (frames_.size() > num_active_frames_) frames_.pop_front();
  }
  void insertLandmark(const std::shared_ptr<MapPoint>& landmark) { landmarks_.insert(landmark); }
  // erases the landmarks whose observing features have all expired (map.cpp:23-50); returns how many
  int cleanMap() {
    int removed = 0;
    for (auto it = landmarks_.begin(); it != landmarks_.end();) {
      bool all_expired = true;
      for (const std::weak_ptr<Feature>& feature : (*it)->observations_) all_expired = all_expired && feature.expired();
      if (all_expired) { it = landmarks_.erase(it); ++removed; }
      else ++it;
    }
    return removed;
  }
  std::unordered_set<std::shared_ptr<MapPoint>> landmarks_;
  std::deque<std::shared_ptr<Frame>> frames_;
  const unsigned int num_active_frames_{30};
};

}  // namespace VO
