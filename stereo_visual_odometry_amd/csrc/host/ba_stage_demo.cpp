// ba_stage_demo.cpp -- drives the frontend's bundle-adjustment stage (frontend_ba.hpp; reference:
// src/frontend.cpp:65-84 with sensor/map.cpp:8-50) over a synthetic stereo-odometry-like sequence, without the
// OpenCV frontend: every "frame" arrives with a drifting pose estimate and pixel observations of the
// landmarks in view, exactly the state the reference's pipeline is in when it reaches frontend.cpp:65.
//
//   ba_stage_demo [--frames N] [--window W] [--no-ba] [--async] [--dump-dir D] [--seed S]
//
// Prints the trajectory error (mean translation error against ground truth) and how many windows were solved.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <unordered_map>

#include "corridor_sequence.hpp"
#include "frontend_ba.hpp"
#include "vo_types.hpp"

using namespace VO;

int main(int argc, char** argv) {
  int n_frames = 120, window = 20;
  bool do_ba = true, async = false, verbose = false;
  unsigned seed = 7;
  std::string dump_dir;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--frames") && i + 1 < argc) n_frames = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--window") && i + 1 < argc) window = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--seed") && i + 1 < argc) seed = (unsigned)atol(argv[++i]);
    else if (!strcmp(argv[i], "--dump-dir") && i + 1 < argc) dump_dir = argv[++i];
    else if (!strcmp(argv[i], "--no-ba")) do_ba = false;
    else if (!strcmp(argv[i], "--async")) async = true;
    else if (!strcmp(argv[i], "--verbose")) verbose = true;
    else { std::cout << "usage: ba_stage_demo [--frames N] [--window W] [--no-ba] [--async] [--dump-dir D] [--seed S]\n"; return 2; }
  }
  try {
    CorridorSequence seq(n_frames, seed);
    const std::vector<double>& gt_z = seq.gt_z;
    auto map = std::make_shared<Map>();
    BundleAdjustmentStage<Map, Frame, SE3d> stage;
    stage.do_bundle_adjustment_ = do_ba;
    stage.bundle_adjustment_window_ = window;
    stage.async_ = async;
    stage.optimizer_.progress_to_stdout_ = false;
    stage.optimizer_.dump_directory_ = dump_dir;
    const Matx33d K = seq.K();
    std::shared_ptr<Frame> prev;
    std::vector<SE3d> final_pose(n_frames);
    std::unordered_map<const Frame*, int> frame_index;
    bool removed_any = false;
    for (int i = 0; i < n_frames; ++i) {
      auto frame = seq.make_frame(i, prev, map);
      frame_index[frame.get()] = i;
      final_pose[i] = frame->pose_;
      if (async) {  // let a pending solve land while its frames are all still in the window
        stage.finish();
        for (const auto& f : map->frames_) final_pose[frame_index[f.get()]] = f->pose_;
      }
      const int solved_before = stage.num_optimizations_;
      if (prev) stage.step(map, prev, frame, K);  // the first frame only becomes frame_prev (frontend.cpp:25)
      if (verbose && !async && stage.num_optimizations_ > solved_before) {
        const hipba_summary& sm = stage.optimizer_.last_summary_;
        double e = 0.0;
        for (const auto& f : map->frames_) {
          const double* q = f->pose_.data();
          const double dz = q[6] - gt_z[frame_index[f.get()]];
          e += std::sqrt(q[4] * q[4] + q[5] * q[5] + dz * dz);
        }
        printf("frame %d: window of %zu frames, cost %.4f -> %.4f in %d iterations (termination %d), window error %.4f\n", i,
               map->frames_.size(), sm.initial_cost, sm.final_cost, sm.num_iterations - 1, sm.termination_type,
               e / map->frames_.size());
      }
      // frames leave the window (and, with their features, expire) with the pose they have at that moment
      for (const auto& f : map->frames_) final_pose[frame_index[f.get()]] = f->pose_;
      if (stage.num_optimizations_ > 0) removed_any = removed_any || map->landmarks_.size() < (size_t)seq.created;
      prev = frame;
    }
    stage.finish();
    for (const auto& f : map->frames_) final_pose[frame_index[f.get()]] = f->pose_;
    // error of the frames' own poses (the trajectory vector keeps the reference's splice quirks, so it is
    // not index-aligned with the frames)
    double err = 0.0;
    const size_t n = final_pose.size();
    for (size_t i = 0; i < n; ++i) {
      const double* p = final_pose[i].data();
      err += std::sqrt(p[4] * p[4] + p[5] * p[5] + (p[6] - gt_z[i]) * (p[6] - gt_z[i]));
    }
    printf("frames %d trajectory_poses %zu windows_solved %d landmarks_created %d landmarks_in_map %zu mean_translation_error %.6f\n",
           n_frames, stage.poses_.size(), stage.num_optimizations_, seq.created, map->landmarks_.size(), n ? err / n : 0.0);
    (void)removed_any;
  } catch (const std::exception& e) {
    std::cerr << "ba_stage_demo: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
