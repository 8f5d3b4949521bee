// frontend_ba.hpp -- the part of VO::Frontend::visualOdometryPipeline that surrounds bundle adjustment
// (reference: src/frontend.cpp:65-84): insert the current frame into the sliding-window map, drop landmarks
// nobody observes any more, run the optimizer every bundle_adjustment_window_ iterations and splice the
// optimized window poses into the trajectory, otherwise append the previous frame's pose.
//
// Templated on the map / pose / intrinsics types like optimizer.hpp, so it compiles against the reference's
// Map (Sophus::SE3d poses) or the stand-ins of vo_types.hpp.  Quirks are kept: in the non-BA branch it is
// the PREVIOUS frame's pose that is appended (frontend.cpp:81), and the splice clears the whole trajectory
// when it is shorter than the window (frontend.cpp:75).
//
// async_ (off by default; no counterpart in the reference): the window is flattened and solved on a worker
// thread while the caller goes on with the next frames; results are written back, and the trajectory is
// spliced, at the start of the next call (or by finish()).  Poses the frontend reads in between are the
// un-optimized ones, so trajectories differ slightly from the synchronous mode.
#pragma once
#include <future>
#include <memory>
#include <vector>

#include "optimizer.hpp"

namespace VO {

template <class MapT, class FrameT, class PoseT>
class BundleAdjustmentStage {
public:
  // parameters, with the reference's defaults (frontend.hpp:32,35)
  bool do_bundle_adjustment_{false};
  int bundle_adjustment_window_{20};
  bool async_{false};

  std::vector<PoseT> poses_;          // frontend.hpp:29
  unsigned long iterations_{1};       // frontend.hpp:51
  Optimizer optimizer_;
  int num_optimizations_{0};

  // frontend.cpp:65-84; frame_prev / frame_curr are context_.frame_prev_ / frame_curr_
  template <class KT>
  void step(const std::shared_ptr<MapT>& map, const std::shared_ptr<FrameT>& frame_prev,
            const std::shared_ptr<FrameT>& frame_curr, const KT& K) {
    finish();                          // a solve started by an earlier call lands before the map changes
    map->insertFrame(frame_curr);      // frontend.cpp:66
    map->cleanMap();                   // frontend.cpp:68
    if (do_bundle_adjustment_ && iterations_ % (unsigned long)bundle_adjustment_window_ == 0) {
      if (!async_) {
        optimizer_.optimize(map, K);   // frontend.cpp:73
        splice(map->frames_);
      } else {
        start_async(map, K);
        // the un-optimized poses stand in until the solve lands
        splice(map->frames_);
      }
      ++num_optimizations_;
    } else {
      poses_.push_back(frame_prev->pose_);  // frontend.cpp:81
    }
    ++iterations_;
  }

  // waits for an asynchronous solve, writes its result into the map blocks and re-splices the trajectory
  void finish() {
    if (!pending_.valid()) return;
    const hipba_summary summary = pending_.get();
    if (summary.termination_type != HIPBA_FAILURE) {
      writeBack(*flat_);
      // the frames of that window occupy the same trajectory slots they were spliced into
      for (size_t i = 0; i < held_frames_.size(); ++i) {
        const size_t slot = splice_begin_ + i;
        if (slot < poses_.size()) poses_[slot] = held_frames_[i]->pose_;
      }
    }
    optimizer_.last_summary_ = summary;
    held_frames_.clear();
    held_landmarks_.clear();
    flat_.reset();
  }

private:
  template <class FramesT>
  void splice(const FramesT& frames) {
    // frontend.cpp:74-77
    if (poses_.size() < frames.size()) poses_.clear();
    else poses_.erase(poses_.end() - frames.size(), poses_.end());
    splice_begin_ = poses_.size();
    for (const auto& frame : frames) poses_.push_back(frame->pose_);
  }

  template <class KT>
  void start_async(const std::shared_ptr<MapT>& map, const KT& K) {
    flat_ = std::make_unique<FlatWindow>();
    flattenMap(map, K, HIPBA_CAMERA_Z_FORWARD, /*first_frame_constant=*/true, flat_.get());
    // keep every block the write-back will touch alive, whatever the map drops meanwhile
    held_frames_.assign(map->frames_.begin(), map->frames_.end());
    for (const auto& frame : map->frames_)
      for (const auto& feature : frame->features_left_) held_landmarks_.push_back(feature->landmark_);
    for (const auto& landmark : map->landmarks_) held_landmarks_.push_back(landmark);
    hipba_handle handle = async_handle();
    FlatWindow* w = flat_.get();
    pending_ = std::async(std::launch::async, [handle, w]() {
      hipba_options options;
      hipba_options_default(&options);
      options.max_num_iterations = 300;  // optimizer.cpp:58
      hipba_summary summary{};
      if (hipba_solve(handle, &w->problem, &options, &summary) != HIPBA_OK) summary.termination_type = HIPBA_FAILURE;
      return summary;
    });
  }

  hipba_handle async_handle() {
    if (!async_handle_ && hipba_create(optimizer_.device_, &async_handle_) != HIPBA_OK)
      throw std::runtime_error(std::string("hipba_create: ") + hipba_last_error());
    return async_handle_;
  }

public:
  ~BundleAdjustmentStage() {
    if (pending_.valid()) pending_.wait();
    if (async_handle_) hipba_destroy(async_handle_);
  }

private:
  std::future<hipba_summary> pending_;
  std::unique_ptr<FlatWindow> flat_;
  std::vector<std::shared_ptr<FrameT>> held_frames_;
  std::vector<std::shared_ptr<void>> held_landmarks_;
  size_t splice_begin_ = 0;
  hipba_handle async_handle_ = nullptr;
};

}  // namespace VO
