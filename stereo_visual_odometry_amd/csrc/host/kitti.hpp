// kitti.hpp -- the text side of the reference's KITTI odometry reader (reference: src/parse_KITTI.cpp:14-114,
// include/visual_odometry/parse_KITTI.hpp:13-33) and its camera type (include/visual_odometry/sensor/camera.hpp:10-45),
// without OpenCV or Sophus: times.txt -> stereo pair file names, calib.txt -> cameras, <sequence>.txt -> ground-truth
// poses.  Image decoding stays with whoever owns the frontend; a ROS-free runner (reference template:
// test/valgrind_main.cpp:20-42) needs exactly these three loaders around it.
//
// Same member names as the reference.  Ground-truth poses are kept in the 7-double layout the BA interface uses
// (qx qy qz qw tx ty tz); like the reference's Sophus::SE3f they carry single precision (parse_KITTI.cpp:96-103
// explains why the reference reads them as float: KITTI's rotation matrices are orthogonal only to 1e-7).
#pragma once
#include <cmath>
#include <cstdio>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "vo_types.hpp"

namespace VO {

struct Matx34d {  // cv::Matx34d (row-major)
  double val[12] = {0};
  double operator()(int r, int c) const { return val[4 * r + c]; }
  double& operator()(int r, int c) { return val[4 * r + c]; }
};

// reference: sensor/camera.hpp:13-41
class Camera {
public:
  Camera() = default;
  Camera(double fx, double fy, double cx, double cy, double tx, double ty, double tz)
      : fx_(fx), fy_(fy), cx_(cx), cy_(cy), tx_(tx), ty_(ty), tz_(tz) {
    const double k[9] = {fx_, 0, cx_, 0, fy_, cy_, 0, 0, 1};
    const double p[12] = {fx_, 0, cx_, tx_, 0, fy_, cy_, ty_, 0, 0, 1, tz_};
    for (int i = 0; i < 9; ++i) K_.val[i] = k[i];
    for (int i = 0; i < 12; ++i) P_.val[i] = p[i];
  }
  // camera.hpp:22-28 (cv::Point3f: single precision)
  void pixelToWorld(float px, float py, double depth, float out[3]) const {
    out[0] = (float)((px - cx_) * depth / fx_);
    out[1] = (float)((py - cy_) * depth / fy_);
    out[2] = (float)depth;
  }
  Matx33d K_;
  Matx34d P_;

private:
  double fx_{0.0}, fy_{0.0}, cx_{0.0}, cy_{0.0}, tx_{0.0}, ty_{0.0}, tz_{0.0};
};

struct StereoPair {  // parse_KITTI.hpp:13-18
  std::string left_frame_;
  std::string right_frame_;
  double time_ = 0.0;
};

class ParseKITTI {
public:
  // parse_KITTI.cpp:14-37: one stereo pair per line of <dir>/times.txt, images <dir>/image_{0,1}/%06d.png
  void loadFrames(const std::string& frame_filename) {
    const std::string time_file = frame_filename + "/times.txt";
    std::ifstream file(time_file);
    if (!file) throw std::runtime_error("Could not open file: " + time_file);
    frames_.clear();
    int frame_idx = 0;
    std::string line;
    while (std::getline(file, line)) {
      StereoPair pair;
      pair.time_ = std::stod(line);
      pair.left_frame_ = getFrame(frame_filename + "/image_0/", frame_idx);
      pair.right_frame_ = getFrame(frame_filename + "/image_1/", frame_idx);
      frames_.push_back(pair);
      ++frame_idx;
    }
  }

  // parse_KITTI.cpp:39-65: every line is "P<i>:" followed by a row-major 3x4 projection matrix; the camera takes
  // fx = P[0], fy = P[5], cx = P[2], cy = P[6] and the fourth column P[3], P[7], P[11]
  void loadCameras(const std::string& calib_filename) {
    std::ifstream file(calib_filename);
    if (!file) throw std::runtime_error("Could not open file: " + calib_filename);
    std::string line;
    while (std::getline(file, line)) {
      std::istringstream iss(line);
      char tag[3];
      for (int i = 0; i < 3; ++i) iss >> tag[i];
      double P[12];
      for (int i = 0; i < 12; ++i) iss >> P[i];
      if (!iss) continue;  // the reference would push a camera of whatever was read; a malformed line is skipped here
      cameras_.emplace_back(P[0], P[5], P[2], P[6], P[3], P[7], P[11]);
    }
  }

  // parse_KITTI.cpp:74-113: every line is a row-major 3x4 [R | t], read in single precision
  void loadGroundTruth(const std::string& ground_truth_filename) {
    std::ifstream file(ground_truth_filename);
    if (!file) throw std::runtime_error("Could not open file: " + ground_truth_filename);
    std::string line;
    while (std::getline(file, line)) {
      std::istringstream iss(line);
      float T[12];
      for (int i = 0; i < 12; ++i) iss >> T[i];
      if (!iss) continue;
      const float R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
      SE3d pose;
      float q[4];
      rotationToQuaternion(R, q);
      for (int i = 0; i < 4; ++i) pose.p[i] = q[i];
      pose.p[4] = T[3]; pose.p[5] = T[7]; pose.p[6] = T[11];
      ground_truth_poses_.push_back(pose);
    }
  }

  std::vector<StereoPair> frames_;
  std::vector<Camera> cameras_;
  std::vector<SE3d> ground_truth_poses_;

  // unit quaternion (x y z w, w >= 0) of a rotation matrix, single precision like Sophus::SE3f::setRotationMatrix
  static void rotationToQuaternion(const float R[9], float q[4]) {
    const float tr = R[0] + R[4] + R[8];
    if (tr > 0.f) {
      const float s = std::sqrt(tr + 1.f) * 2.f;
      q[3] = 0.25f * s; q[0] = (R[7] - R[5]) / s; q[1] = (R[2] - R[6]) / s; q[2] = (R[3] - R[1]) / s;
    } else if (R[0] > R[4] && R[0] > R[8]) {
      const float s = std::sqrt(1.f + R[0] - R[4] - R[8]) * 2.f;
      q[3] = (R[7] - R[5]) / s; q[0] = 0.25f * s; q[1] = (R[1] + R[3]) / s; q[2] = (R[2] + R[6]) / s;
    } else if (R[4] > R[8]) {
      const float s = std::sqrt(1.f + R[4] - R[0] - R[8]) * 2.f;
      q[3] = (R[2] - R[6]) / s; q[0] = (R[1] + R[3]) / s; q[1] = 0.25f * s; q[2] = (R[5] + R[7]) / s;
    } else {
      const float s = std::sqrt(1.f + R[8] - R[0] - R[4]) * 2.f;
      q[3] = (R[3] - R[1]) / s; q[0] = (R[2] + R[6]) / s; q[1] = (R[5] + R[7]) / s; q[2] = 0.25f * s;
    }
    const float n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const float sgn = q[3] < 0.f ? -1.f : 1.f;
    for (int i = 0; i < 4; ++i) q[i] = sgn * q[i] / n;
  }

private:
  // parse_KITTI.cpp:67-72
  static std::string getFrame(const std::string& path_to_frames, int frame_idx) {
    char name[16];
    snprintf(name, sizeof name, "%06d", frame_idx);
    return path_to_frames + name + ".png";
  }
};

// Absolute trajectory error of estimated poses against ground truth, frame by frame (no alignment beyond the
// shared first frame): RMSE and mean of the translation differences.  What a runner prints at the end.
inline void trajectoryError(const std::vector<SE3d>& estimate, const std::vector<SE3d>& truth, double* rmse, double* mean) {
  const size_t n = estimate.size() < truth.size() ? estimate.size() : truth.size();
  double s2 = 0.0, s1 = 0.0;
  for (size_t i = 0; i < n; ++i) {
    double d2 = 0.0;
    for (int k = 4; k < 7; ++k) { const double d = estimate[i].p[k] - truth[i].p[k]; d2 += d * d; }
    s2 += d2; s1 += std::sqrt(d2);
  }
  *rmse = n ? std::sqrt(s2 / n) : 0.0;
  *mean = n ? s1 / n : 0.0;
}

}  // namespace VO
