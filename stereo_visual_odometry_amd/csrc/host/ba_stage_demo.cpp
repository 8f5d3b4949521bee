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
#include <random>
#include <unordered_map>

#include "frontend_ba.hpp"
#include "vo_types.hpp"

using namespace VO;

static const double fx = 718.856, fy = 718.856, cx = 607.1928, cy = 185.2157, width = 1241.0, height = 376.0;

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
    std::mt19937 gen(seed);
    std::normal_distribution<double> n01(0.0, 1.0);
    std::uniform_real_distribution<double> u01(0.0, 1.0);
    // ground truth: forward motion along +z (camera z-forward, identity attitude), ~1 m per frame
    std::vector<double> gt_z(n_frames);
    for (int i = 0; i < n_frames; ++i) gt_z[i] = (i ? gt_z[i - 1] : 0.0) + 0.8 + 0.4 * u01(gen);
    // landmarks scattered along the corridor
    const int n_landmarks = 60 * n_frames;
    std::vector<std::shared_ptr<MapPoint>> landmarks(n_landmarks);
    std::vector<Vector3d> gt_points(n_landmarks);
    for (int l = 0; l < n_landmarks; ++l) {
      // off the optical axis: under pure forward motion a point straight ahead shows no parallax, and without a
      // robust loss (optimizer.cpp:48 passes none) a two-view landmark of unobservable depth can run away
      const double side = u01(gen) < 0.5 ? -1.0 : 1.0;
      gt_points[l] = Vector3d(side * (2.0 + 10.0 * u01(gen)), -3.0 + 6.0 * u01(gen), 4.0 + (gt_z.back() + 30.0) * u01(gen));
      landmarks[l] = nullptr;
    }
    auto map = std::make_shared<Map>();
    BundleAdjustmentStage<Map, Frame, SE3d> stage;
    stage.do_bundle_adjustment_ = do_ba;
    stage.bundle_adjustment_window_ = window;
    stage.async_ = async;
    stage.optimizer_.progress_to_stdout_ = false;
    stage.optimizer_.dump_directory_ = dump_dir;
    Matx33d K;
    K(0, 0) = fx; K(1, 1) = fy; K(0, 2) = cx; K(1, 2) = cy;
    std::shared_ptr<Frame> prev;
    std::vector<SE3d> final_pose(n_frames);
    std::unordered_map<const Frame*, int> frame_index;
    bool removed_any = false;
    int created = 0;
    for (int i = 0; i < n_frames; ++i) {
      auto frame = std::make_shared<Frame>();
      // odometry-like estimate: the previous frame's current pose (optimized or not) composed with the true
      // relative motion plus noise, so that errors accumulate unless bundle adjustment removes them
      double* p = frame->pose_.data();
      p[0] = p[1] = p[2] = 0.0; p[3] = 1.0;
      if (prev) {
        // T_curr = T_prev * [exp(small rotation noise) | true step in the camera frame + noise]
        const double* q = prev->pose_.data();
        const double w[3] = {0.002 * n01(gen), 0.002 * n01(gen), 0.002 * n01(gen)};
        const double dq[4] = {0.5 * w[0], 0.5 * w[1], 0.5 * w[2], 1.0};
        double r[4] = {q[3] * dq[0] + q[0] * dq[3] + q[1] * dq[2] - q[2] * dq[1],
                       q[3] * dq[1] - q[0] * dq[2] + q[1] * dq[3] + q[2] * dq[0],
                       q[3] * dq[2] + q[0] * dq[1] - q[1] * dq[0] + q[2] * dq[3],
                       q[3] * dq[3] - q[0] * dq[0] - q[1] * dq[1] - q[2] * dq[2]};
        const double nr = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
        for (int k = 0; k < 4; ++k) p[k] = r[k] / nr;
        const double d[3] = {0.05 * n01(gen), 0.05 * n01(gen), (gt_z[i] - gt_z[i - 1]) + 0.05 * n01(gen)};
        // rotate d by q (v' = v + qw t + qv x t, t = 2 qv x v)
        const double t[3] = {2 * (q[1] * d[2] - q[2] * d[1]), 2 * (q[2] * d[0] - q[0] * d[2]), 2 * (q[0] * d[1] - q[1] * d[0])};
        p[4] = q[4] + d[0] + q[3] * t[0] + (q[1] * t[2] - q[2] * t[1]);
        p[5] = q[5] + d[1] + q[3] * t[1] + (q[2] * t[0] - q[0] * t[2]);
        p[6] = q[6] + d[2] + q[3] * t[2] + (q[0] * t[1] - q[1] * t[0]);
      } else {
        p[4] = p[5] = 0.0; p[6] = gt_z[0];
      }
      int seen = 0;
      for (int l = 0; l < n_landmarks && seen < 300; ++l) {
        const double z = gt_points[l].z() - gt_z[i];
        if (z < 4.0 || z > 30.0) continue;
        const double u = fx * gt_points[l].x() / z + cx, v = fy * gt_points[l].y() / z + cy;
        if (u < 0 || u >= width || v < 0 || v >= height) continue;
        if (!landmarks[l]) {
          // triangulated from the drifted estimate: the same error the frame carries, plus depth noise
          const double zn = z * (1.0 + 0.003 * n01(gen));
          const double c[3] = {gt_points[l].x() * zn / z, gt_points[l].y() * zn / z, zn};
          const double tt[3] = {2 * (p[1] * c[2] - p[2] * c[1]), 2 * (p[2] * c[0] - p[0] * c[2]), 2 * (p[0] * c[1] - p[1] * c[0])};
          landmarks[l] = std::make_shared<MapPoint>(Vector3d(p[4] + c[0] + p[3] * tt[0] + (p[1] * tt[2] - p[2] * tt[1]),
                                                             p[5] + c[1] + p[3] * tt[1] + (p[2] * tt[0] - p[0] * tt[2]),
                                                             p[6] + c[2] + p[3] * tt[2] + (p[0] * tt[1] - p[1] * tt[0])));
          map->insertLandmark(landmarks[l]);
          ++created;
        }
        Point2f px;
        px.x = (float)(u + 0.15 * n01(gen)); px.y = (float)(v + 0.15 * n01(gen));
        frame->pushFeatureLeftWithLandmark(px, landmarks[l]);
        ++seen;
      }
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
      if (stage.num_optimizations_ > 0) removed_any = removed_any || map->landmarks_.size() < (size_t)created;
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
           n_frames, stage.poses_.size(), stage.num_optimizations_, created, map->landmarks_.size(), n ? err / n : 0.0);
    (void)removed_any;
  } catch (const std::exception& e) {
    std::cerr << "ba_stage_demo: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
