// kitti_runner.cpp -- ROS-free runner over a KITTI-odometry-layout sequence directory, after the reference's
// test/valgrind_main.cpp:20-42: load the sequence (ParseKITTI::loadFrames / loadCameras / loadGroundTruth,
// src/parse_KITTI.cpp:14-114), build the pipeline from the left camera, run it frame by frame, report.
//
// What is real here: the three loaders (times.txt, calib.txt, the ground-truth pose file), the camera intrinsics
// taken from calib.txt's P0 (Camera::K_), the sliding-window map, the bundle-adjustment stage with the reference's
// trigger and trajectory splice (src/frontend.cpp:65-84), the GPU solve behind VO::Optimizer::optimize, and the
// trajectory error against the ground truth.  What is NOT: the image half of the frontend (cv::imread, FAST, optical
// flow, triangulation, PnP need OpenCV, which this build does not have).  It is replaced by TrajectorySequence, which
// follows the ground-truth trajectory of the very sequence that was loaded and hands every frame to the stage with a
// dead-reckoned pose estimate and noisy pixel observations -- stated in the output as "frontend synthetic".
//
//   kitti_runner <sequence dir> [--poses <file>] [--frames N] [--window 20] [--no-ba] [--both] [--paced] [--seed S]
//                [--device D] [--async]
//
//   <sequence dir>   holds times.txt and calib.txt (and image_0/, image_1/, which are not opened)
//   --poses          ground-truth file (12 numbers per line); default <sequence dir>/poses.txt
//   --both           run twice, without and with bundle adjustment, and print both errors
//   --paced          sleep until every frame's time stamp (times.txt) before processing it
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <thread>
#include <vector>

#include "frontend_ba.hpp"
#include "kitti.hpp"
#include "trajectory_sequence.hpp"
#include "vo_types.hpp"

using namespace VO;
using Clock = std::chrono::steady_clock;

struct RunResult {
  double rmse = 0.0, mean = 0.0, wall_s = 0.0, ba_ms_mean = 0.0, ba_ms_max = 0.0;
  int windows = 0, frames = 0, trajectory_entries = 0;
  long landmarks_created = 0;
};

static RunResult run(const ParseKITTI& pk, int n_frames, bool do_ba, int window, bool paced, unsigned seed, int device,
                     bool async) {
  const Matx33d K = pk.cameras_[0].K_;  // main.cpp:45 / valgrind_main.cpp:32: the left camera's intrinsics feed the pipeline
  std::vector<SE3d> truth(pk.ground_truth_poses_.begin(), pk.ground_truth_poses_.begin() + n_frames);
  TrajectorySequence seq(truth, K, seed);
  auto map = std::make_shared<Map>();
  BundleAdjustmentStage<Map, Frame, SE3d> stage;
  stage.do_bundle_adjustment_ = do_ba;
  stage.bundle_adjustment_window_ = window;
  stage.async_ = async;
  stage.optimizer_.progress_to_stdout_ = false;
  stage.optimizer_.device_ = device;
  std::vector<std::shared_ptr<Frame>> all_frames;
  std::shared_ptr<Frame> prev;
  RunResult r;
  double ba_total = 0.0;
  const auto t0 = Clock::now();
  for (int i = 0; i < n_frames; ++i) {
    if (paced) {
      const auto due = t0 + std::chrono::duration_cast<Clock::duration>(std::chrono::duration<double>(pk.frames_[i].time_ - pk.frames_[0].time_));
      std::this_thread::sleep_until(due);
    }
    auto frame = seq.make_frame(i, prev, map);
    all_frames.push_back(frame);
    const int before = stage.num_optimizations_;
    const auto s0 = Clock::now();
    if (prev) stage.step(map, prev, frame, K);
    if (stage.num_optimizations_ > before) {
      const double ms = std::chrono::duration<double, std::milli>(Clock::now() - s0).count();
      ba_total += ms;
      if (ms > r.ba_ms_max) r.ba_ms_max = ms;
    }
    prev = frame;
  }
  stage.finish();
  r.wall_s = std::chrono::duration<double>(Clock::now() - t0).count();
  r.windows = stage.num_optimizations_;
  r.ba_ms_mean = r.windows ? ba_total / r.windows : 0.0;
  r.frames = n_frames;
  r.trajectory_entries = (int)stage.poses_.size();
  r.landmarks_created = seq.created;
  // every frame's final pose (bundle adjustment refines a frame in place for as long as it is in the window)
  std::vector<SE3d> estimate;
  for (const auto& f : all_frames) estimate.push_back(f->pose_);
  trajectoryError(estimate, truth, &r.rmse, &r.mean);
  return r;
}

int main(int argc, char** argv) {
  if (argc < 2) {
    std::cout << "usage: kitti_runner <sequence dir> [--poses file] [--frames N] [--window W] [--no-ba] [--both] [--paced] "
                 "[--seed S] [--device D] [--async]\n";
    return 2;
  }
  const std::string dir = argv[1];
  std::string poses = dir + "/poses.txt";
  int n_frames = 0, window = 20, device = 0;
  bool do_ba = true, both = false, paced = false, async = false;
  unsigned seed = 1000;
  for (int i = 2; i < argc; ++i) {
    if (!strcmp(argv[i], "--poses") && i + 1 < argc) poses = argv[++i];
    else if (!strcmp(argv[i], "--frames") && i + 1 < argc) n_frames = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--window") && i + 1 < argc) window = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--seed") && i + 1 < argc) seed = (unsigned)atoi(argv[++i]);
    else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--no-ba")) do_ba = false;
    else if (!strcmp(argv[i], "--both")) both = true;
    else if (!strcmp(argv[i], "--paced")) paced = true;
    else if (!strcmp(argv[i], "--async")) async = true;
    else { std::cerr << "kitti_runner: unknown argument " << argv[i] << "\n"; return 2; }
  }
  try {
    ParseKITTI pk;
    pk.loadFrames(dir);
    pk.loadCameras(dir + "/calib.txt");
    pk.loadGroundTruth(poses);
    if (pk.cameras_.size() < 2) throw std::runtime_error("calib.txt holds fewer than two cameras");
    if (pk.frames_.empty()) throw std::runtime_error("times.txt holds no frame");
    if (pk.ground_truth_poses_.size() < pk.frames_.size())
      throw std::runtime_error("fewer ground-truth poses than frames");
    if (n_frames <= 0 || n_frames > (int)pk.frames_.size()) n_frames = (int)pk.frames_.size();
    const Matx33d& K = pk.cameras_[0].K_;
    printf("sequence %s frames %d cameras %zu fx %.4f fy %.4f cx %.4f cy %.4f baseline_m %.6f frontend synthetic\n", dir.c_str(),
           n_frames, pk.cameras_.size(), K(0, 0), K(1, 1), K(0, 2), K(1, 2), -pk.cameras_[1].P_(0, 3) / pk.cameras_[1].P_(0, 0));
    if (both || !do_ba) {
      const RunResult r = run(pk, n_frames, false, window, paced, seed, device, false);
      printf("bundle_adjustment off windows %d ate_rmse_m %.6f ate_mean_m %.6f wall_s %.3f trajectory_entries %d\n", r.windows, r.rmse,
             r.mean, r.wall_s, r.trajectory_entries);
    }
    if (both || do_ba) {
      const RunResult r = run(pk, n_frames, true, window, paced, seed, device, async);
      printf("bundle_adjustment on windows %d ate_rmse_m %.6f ate_mean_m %.6f wall_s %.3f trajectory_entries %d ba_frame_ms_mean %.3f "
             "ba_frame_ms_max %.3f landmarks %ld\n", r.windows, r.rmse, r.mean, r.wall_s, r.trajectory_entries, r.ba_ms_mean,
             r.ba_ms_max, r.landmarks_created);
    }
  } catch (const std::exception& ex) {
    std::cerr << "kitti_runner: " << ex.what() << std::endl;
    return 1;
  }
  return 0;
}
