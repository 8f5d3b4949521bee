// vo_pipeline -- the reference's main loop (src/main.cpp:45-58, test/valgrind_main.cpp:30-40: insert a stereo pair,
// run Frontend::visualOdometryPipeline) over a synthetic sequence, with triangulation, pose estimation and bundle
// adjustment on the GPU and the image half replaced by a scene of known landmarks (synthetic_frontend.hpp).
//   vo_pipeline [--frames N] [--kitti-dir <sequence dir> [--poses <file>]] [--no-ba] [--both] [--window 20] [--min-features 500]
//               [--seed S] [--wrong 0.03] [--lost 0.04] [--verbose] [--dump-dir <dir>]
// Without --kitti-dir the camera drives a gentle S-curve at 0.8 m per frame; with it, the trajectory is the
// sequence's ground truth (--poses, default <sequence dir>/poses.txt) and the cameras its calib.txt.  Prints the absolute trajectory error against
// the truth (trajectoryError, kitti.hpp) and what the three GPU steps took.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "synthetic_frontend.hpp"

using namespace VO;

struct Result { double rmse, mean, wall_s, tri_ms, est_ms, ba_ms; int windows, resets; std::size_t poses; double mean_features; };

static bool g_verbose = false;
static std::string g_dump_dir;
static Result run(const std::vector<SE3d>& truth, const Camera& left, const Camera& right, bool do_ba, int window,
                  unsigned long min_features, unsigned seed, const SyntheticScene::Options& so) {
  auto scene = std::make_shared<SyntheticScene>(truth, left, right, seed, so);
  auto map = std::make_shared<Map>();
  Frontend frontend(left, right, map, scene);
  frontend.do_bundle_adjustment_ = do_ba;
  frontend.bundle_adjustment_window_ = window;
  frontend.min_num_features_ = min_features;
  frontend.verbose_ = g_verbose;
  frontend.dump_directory_ = g_dump_dir;
  frontend.initialize();
  std::vector<std::shared_ptr<Frame>> all;
  double features = 0.0;
  const auto t0 = std::chrono::steady_clock::now();
  for (std::size_t i = 0; i < truth.size(); ++i) {
    auto frame = std::make_shared<Frame>();
    all.push_back(frame);
    frontend.context_.insertFrame(frame);  // main.cpp:55 (insertImages, without the images)
    scene->advance();
    frontend.visualOdometryPipeline();
    features += (double)frame->features_left_.size();
  }
  Result r{};
  r.wall_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::vector<SE3d> estimate, rel;
  for (const auto& f : all) estimate.push_back(f->pose_);
  for (std::size_t i = 0; i < truth.size(); ++i) rel.push_back(scene->truth((int)i));
  trajectoryError(estimate, rel, &r.rmse, &r.mean);
  const double nf = (double)truth.size();
  r.tri_ms = 1e3 * frontend.time_triangulate_s_ / nf;
  r.est_ms = 1e3 * frontend.time_estimate_s_ / nf;
  r.ba_ms = frontend.num_optimizations_ ? 1e3 * frontend.time_optimize_s_ / frontend.num_optimizations_ : 0.0;
  r.windows = frontend.num_optimizations_;
  r.resets = frontend.num_resets_;
  r.poses = frontend.poses_.size();
  r.mean_features = features / nf;
  return r;
}

int main(int argc, char** argv) {
  int n_frames = 200, window = 20;
  unsigned long min_features = 500;
  unsigned seed = 1;
  bool do_ba = true, both = false;
  std::string kitti_dir, poses_file;
  SyntheticScene::Options so;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--frames") && i + 1 < argc) n_frames = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--window") && i + 1 < argc) window = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--min-features") && i + 1 < argc) min_features = (unsigned long)atol(argv[++i]);
    else if (!strcmp(argv[i], "--seed") && i + 1 < argc) seed = (unsigned)atoi(argv[++i]);
    else if (!strcmp(argv[i], "--wrong") && i + 1 < argc) so.wrong = atof(argv[++i]);
    else if (!strcmp(argv[i], "--lost") && i + 1 < argc) so.lost = atof(argv[++i]);
    else if (!strcmp(argv[i], "--kitti-dir") && i + 1 < argc) kitti_dir = argv[++i];
    else if (!strcmp(argv[i], "--poses") && i + 1 < argc) poses_file = argv[++i];
    else if (!strcmp(argv[i], "--no-ba")) do_ba = false;
    else if (!strcmp(argv[i], "--verbose")) g_verbose = true;
    else if (!strcmp(argv[i], "--dump-dir") && i + 1 < argc) g_dump_dir = argv[++i];
    else if (!strcmp(argv[i], "--both")) both = true;
    else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
  }
  Camera left(718.856, 718.856, 607.1928, 185.2157, 0.0, 0.0, 0.0), right(718.856, 718.856, 607.1928, 185.2157, -386.1448, 0.0, 0.0);
  std::vector<SE3d> truth;
  try {
    if (!kitti_dir.empty()) {
      ParseKITTI pk;
      pk.loadCameras(kitti_dir + "/calib.txt");
      pk.loadGroundTruth(poses_file.empty() ? kitti_dir + "/poses.txt" : poses_file);
      left = pk.cameras_.at(0); right = pk.cameras_.at(1);
      if (n_frames > (int)pk.ground_truth_poses_.size()) n_frames = (int)pk.ground_truth_poses_.size();
      truth.assign(pk.ground_truth_poses_.begin(), pk.ground_truth_poses_.begin() + n_frames);
    } else {
      // a gentle S-curve in the x-z plane (the camera looks along +z, y points down): 0.8 m per frame
      SE3d pose;
      for (int i = 0; i < n_frames; ++i) {
        truth.push_back(pose);
        const double yaw_rate = 0.012 * std::sin(2.0 * M_PI * i / 120.0);
        SE3d step;
        step.p[1] = std::sin(0.5 * yaw_rate); step.p[3] = std::cos(0.5 * yaw_rate);
        step.p[6] = 0.8;
        pose = se3Multiply(pose, step);
      }
    }
    printf("vo_pipeline: %d frames, frontend synthetic (known landmarks; %.0f %% wrong tracks, %.0f %% lost per frame), "
           "triangulation / pose estimation / bundle adjustment on the GPU\n", n_frames, 100 * so.wrong, 100 * so.lost);
    for (int pass = 0; pass < (both ? 2 : 1); ++pass) {
      const bool ba = both ? pass == 1 : do_ba;
      const Result r = run(truth, left, right, ba, window, min_features, seed, so);
      printf("bundle_adjustment %s ate_rmse_m %.4f ate_mean_m %.4f windows %d resets %d trajectory_entries %zu "
             "mean_features %.0f triangulate_ms_per_frame %.3f estimate_ms_per_frame %.3f ba_ms_per_window %.3f wall_s %.2f\n",
             ba ? "on" : "off", r.rmse, r.mean, r.windows, r.resets, r.poses, r.mean_features, r.tri_ms, r.est_ms, r.ba_ms, r.wall_s);
    }
  } catch (const std::exception& e) {
    fprintf(stderr, "vo_pipeline: %s\n", e.what());
    return 1;
  }
  return 0;
}
