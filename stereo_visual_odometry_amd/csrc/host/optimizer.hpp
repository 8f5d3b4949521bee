// optimizer.hpp -- drop-in for the reference's VO::Optimizer (include/visual_odometry/solve/optimizer.hpp:14-19):
// same class name, same optimize(map, K) signature, same in-place semantics, but the ceres::Problem /
// ceres::Solve body of src/solve/optimizer.cpp:16-62 is replaced by one call into the HIP engine's C ABI.
//
// flattenMap / optimizeMap are templates over the map and matrix types: they only use the member names
// the reference's own optimize() uses (frames_, features_left_, pixel, landmark_, landmarks_, pose_.data()),
// so the same header compiles against the reference's real Map (Sophus/Eigen/OpenCV types) or against the
// dependency-free stand-ins of vo_types.hpp.
#pragma once
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "hipba.h"
#include "window_io.hpp"

namespace VO {

struct FlatWindow {
  std::vector<double> poses, points, uv;
  std::vector<int32_t> obs_pose, obs_point;
  std::vector<uint8_t> pose_const, point_const;
  std::vector<double*> pose_storage, point_storage;  // where each block lives in the caller's map
  hipba_problem problem{};
};

// Walks the map the way optimizer.cpp:19-52 does: frames oldest first, one residual per left feature;
// landmarks are numbered by first appearance (the reference's unordered_set order is address dependent),
// landmarks nobody observes come last.  first_frame_constant applies optimizer.cpp:28-36.
template <class MapT, class KT>
void flattenMap(const std::shared_ptr<MapT>& map, const KT& K, int camera_model, bool first_frame_constant,
                FlatWindow* w) {
  // landmark storage -> index: an open-addressing table over the pointers (a window makes tens of thousands of
  // lookups for a few thousand landmarks; std::unordered_map spent a millisecond on them, a third of the call)
  size_t n_features = 0;
  for (const auto& frame : map->frames_) n_features += frame->features_left_.size();
  size_t cap = 64;
  while (cap < 2 * (map->landmarks_.size() + n_features / 4 + 1)) cap <<= 1;
  std::vector<const void*> keys(cap, nullptr);
  std::vector<int32_t> vals(cap, 0);
  size_t used = 0;
  auto slot_of = [&](const void* key) {
    size_t h = (size_t)(((uintptr_t)key >> 4) * 0x9E3779B97F4A7C15ull) & (cap - 1);
    while (keys[h] && keys[h] != key) h = (h + 1) & (cap - 1);
    return h;
  };
  auto landmark_index = [&](double* storage) {
    size_t h = slot_of(storage);
    if (keys[h]) return vals[h];
    if (2 * (used + 1) > cap) {  // (more landmarks than the map lists: grow and re-enter)
      std::vector<const void*> old_keys;
      std::vector<int32_t> old_vals;
      old_keys.swap(keys); old_vals.swap(vals);
      cap <<= 1;
      keys.assign(cap, nullptr); vals.assign(cap, 0);
      for (size_t i = 0; i < old_keys.size(); ++i)
        if (old_keys[i]) { const size_t g = slot_of(old_keys[i]); keys[g] = old_keys[i]; vals[g] = old_vals[i]; }
      h = slot_of(storage);
    }
    const int32_t id = (int32_t)w->point_storage.size();
    keys[h] = storage; vals[h] = id; ++used;
    w->point_storage.push_back(storage);
    w->point_const.push_back(0);
    return id;
  };
  w->obs_pose.reserve(n_features); w->obs_point.reserve(n_features); w->uv.reserve(2 * n_features);
  int32_t fi = 0;
  for (const auto& frame : map->frames_) {
    w->pose_storage.push_back(frame->pose_.data());
    w->pose_const.push_back(first_frame_constant && fi == 0 ? 1 : 0);
    for (const auto& feature : frame->features_left_) {
      const int32_t li = landmark_index(feature->landmark_->pose_.data());
      if (first_frame_constant && fi == 0) w->point_const[li] = 1;
      w->obs_pose.push_back(fi);
      w->obs_point.push_back(li);
      w->uv.push_back(static_cast<double>(feature->pixel.x));  // optimizer.cpp:40-41
      w->uv.push_back(static_cast<double>(feature->pixel.y));
    }
    ++fi;
  }
  for (const auto& landmark : map->landmarks_) landmark_index(landmark->pose_.data());
  w->poses.resize(7 * w->pose_storage.size());
  w->points.resize(3 * w->point_storage.size());
  for (size_t i = 0; i < w->pose_storage.size(); ++i)
    for (int k = 0; k < 7; ++k) w->poses[7 * i + k] = w->pose_storage[i][k];
  for (size_t i = 0; i < w->point_storage.size(); ++i)
    for (int k = 0; k < 3; ++k) w->points[3 * i + k] = w->point_storage[i][k];
  hipba_problem& p = w->problem;
  p.n_poses = (int32_t)w->pose_storage.size();
  p.poses = w->poses.data();
  p.pose_const = w->pose_const.data();
  p.n_points = (int32_t)w->point_storage.size();
  p.points = w->points.data();
  p.point_const = w->point_const.data();
  p.n_obs = (int64_t)w->obs_pose.size();
  p.obs_pose = w->obs_pose.data();
  p.obs_point = w->obs_point.data();
  p.obs_uv = w->uv.data();
  p.fx = K(0, 0); p.fy = K(1, 1); p.cx = K(0, 2); p.cy = K(1, 2);
  p.camera_model = camera_model;
}

// Ceres writes through the raw pointers it was given (optimizer.cpp:21,27); so does this.
inline void writeBack(const FlatWindow& w) {
  for (size_t i = 0; i < w.pose_storage.size(); ++i)
    for (int k = 0; k < 7; ++k) w.pose_storage[i][k] = w.poses[7 * i + k];
  for (size_t i = 0; i < w.point_storage.size(); ++i)
    for (int k = 0; k < 3; ++k) w.point_storage[i][k] = w.points[3 * i + k];
}

template <class MapT, class KT>
hipba_summary optimizeMap(hipba_handle handle, const std::shared_ptr<MapT>& map, const KT& K, int camera_model,
                          bool first_frame_constant, const hipba_options& options) {
  FlatWindow w;
  flattenMap(map, K, camera_model, first_frame_constant, &w);
  hipba_summary summary{};
  const int rc = hipba_solve(handle, &w.problem, &options, &summary);
  if (rc != HIPBA_OK) throw std::runtime_error(std::string("hipba_solve: ") + hipba_last_error());
  if (summary.termination_type != HIPBA_FAILURE) writeBack(w);
  return summary;
}

class Optimizer {
public:
  Optimizer() = default;
  Optimizer(const Optimizer&) = delete;
  Optimizer& operator=(const Optimizer&) = delete;
  ~Optimizer() { if (handle_) hipba_destroy(handle_); }

  // reference: void Optimizer::optimize(const std::shared_ptr<Map>& map, const cv::Matx33d& K)
  template <class MapT, class KT>
  void optimize(const std::shared_ptr<MapT>& map, const KT& K) {
    if (!handle_) {
      if (hipba_version() != HIPBA_VERSION)
        throw std::runtime_error("libhipba.so is ABI version " + std::to_string(hipba_version()) +
                                 ", this adapter was compiled against version " + std::to_string(HIPBA_VERSION));
      if (hipba_create(device_, &handle_) != HIPBA_OK)
        throw std::runtime_error(std::string("hipba_create: ") + hipba_last_error());
    }
    hipba_options options;
    hipba_options_default(&options);
    options.max_num_iterations = 300;          // optimizer.cpp:58
    options.minimizer_progress_to_stdout = progress_to_stdout_ ? 1 : 0;  // optimizer.cpp:56 (+ the report of line 62)
    const auto t0 = std::chrono::steady_clock::now();
    FlatWindow w;
    flattenMap(map, K, HIPBA_CAMERA_Z_FORWARD, /*first_frame_constant=*/true, &w);
    const auto t1 = std::chrono::steady_clock::now();
    if (!dump_directory_.empty()) {
      // capture hook: the window exactly as the solver receives it, for replay on a machine without the frontend
      char name[64];
      snprintf(name, sizeof name, "/window_%06d.%s", num_calls_, dump_binary_ ? "bin" : "txt");
      writeWindow(dump_directory_ + name, w.problem, dump_binary_);
    }
    ++num_calls_;
    hipba_summary summary{};
    if (hipba_solve(handle_, &w.problem, &options, &summary) != HIPBA_OK)
      throw std::runtime_error(std::string("hipba_solve: ") + hipba_last_error());
    const auto t2 = std::chrono::steady_clock::now();
    if (summary.termination_type != HIPBA_FAILURE) writeBack(w);
    last_summary_ = summary;
    last_flatten_ms_ = std::chrono::duration<double, std::milli>(t1 - t0).count();
    last_solve_ms_ = std::chrono::duration<double, std::milli>(t2 - t1).count();
    last_write_back_ms_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t2).count();
  }

  int device_ = 0;
  bool progress_to_stdout_ = true;
  std::string dump_directory_;   // when set, every window is written there before it is solved
  bool dump_binary_ = false;
  int num_calls_ = 0;
  hipba_summary last_summary_{};
  // where the last call's time went: walking the map into arrays, the engine (a window dump included), copying back
  double last_flatten_ms_ = 0.0, last_solve_ms_ = 0.0, last_write_back_ms_ = 0.0;

private:
  hipba_handle handle_ = nullptr;
};

}  // namespace VO
