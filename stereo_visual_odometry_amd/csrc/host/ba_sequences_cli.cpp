// ba_sequences.cpp -- several visual-odometry sequences at once, one per GPU (round-robin over the visible devices):
// the shape of "KITTI 00-10 concurrently, CPU frontend + GPU bundle adjustment" without the datasets and without
// OpenCV.  The template is the reference's ROS-free runner (test/valgrind_main.cpp:20-42: load a sequence, then
// Frontend::visualOdometryPipeline() per frame); here every sequence is a host thread that owns a
// BundleAdjustmentStage (frontend_ba.hpp = reference src/frontend.cpp:65-84) with its own engine handle and feeds it
// a synthetic corridor sequence (corridor_sequence.hpp).  What the real frontend would spend per frame on FAST,
// optical flow and PnP is NOT computed: it is replaced by a sleep of --frontend-ms milliseconds (default 30, stated
// in the output), so the latencies below are "stub + map bookkeeping + bundle adjustment when it is due".
//
//   ba_sequences [--sequences 11] [--frames 200] [--window 20] [--frontend-ms 30] [--devices N] [--async]
//
// Prints one line: per-frame end-to-end latency percentiles, latency of the frames on which bundle adjustment ran,
// windows solved and windows per second over all sequences.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <thread>
#include <vector>

#include "corridor_sequence.hpp"
#include "frontend_ba.hpp"
#include "vo_types.hpp"

using namespace VO;
using Clock = std::chrono::steady_clock;

static double percentile(std::vector<double>& v, double q) {
  if (v.empty()) return 0.0;
  std::sort(v.begin(), v.end());
  const double at = q * (v.size() - 1);
  const size_t lo = (size_t)at;
  const size_t hi = std::min(lo + 1, v.size() - 1);
  return v[lo] + (at - lo) * (v[hi] - v[lo]);
}

int main(int argc, char** argv) {
  int n_seq = 11, n_frames = 200, window = 20, devices = 0;
  double frontend_ms = 30.0;
  bool async = false;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--sequences") && i + 1 < argc) n_seq = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--frames") && i + 1 < argc) n_frames = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--window") && i + 1 < argc) window = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--frontend-ms") && i + 1 < argc) frontend_ms = atof(argv[++i]);
    else if (!strcmp(argv[i], "--devices") && i + 1 < argc) devices = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--async")) async = true;
    else { std::cout << "usage: ba_sequences [--sequences N] [--frames N] [--window W] [--frontend-ms T] [--devices N] [--async]\n"; return 2; }
  }
  int visible = 0;
  if (hipba_device_count(&visible) != HIPBA_OK || visible <= 0) {
    std::cerr << "ba_sequences: " << hipba_last_error() << std::endl;
    return 1;
  }
  if (devices <= 0 || devices > visible) devices = visible;
  std::vector<std::vector<double>> frame_ms(n_seq), ba_ms(n_seq);
  std::vector<int> solved(n_seq, 0);
  std::atomic<int> failed{0};
  const auto t0 = Clock::now();
  std::vector<std::thread> threads;
  for (int q = 0; q < n_seq; ++q) {
    threads.emplace_back([&, q] {
      try {
        CorridorSequence seq(n_frames, 1000u + (unsigned)q);
        auto map = std::make_shared<Map>();
        BundleAdjustmentStage<Map, Frame, SE3d> stage;
        stage.do_bundle_adjustment_ = true;
        stage.bundle_adjustment_window_ = window;
        stage.async_ = async;
        stage.optimizer_.progress_to_stdout_ = false;
        stage.optimizer_.device_ = q % devices;  // one sequence per GPU, wrapping around when there are more sequences
        const Matx33d K = seq.K();
        std::shared_ptr<Frame> prev;
        for (int i = 0; i < n_frames; ++i) {
          const auto f0 = Clock::now();
          // the frontend's image work, not computed here
          if (frontend_ms > 0) std::this_thread::sleep_for(std::chrono::duration<double, std::milli>(frontend_ms));
          auto frame = seq.make_frame(i, prev, map);
          const int before = stage.num_optimizations_;
          const auto s0 = Clock::now();
          if (prev) stage.step(map, prev, frame, K);
          const auto f1 = Clock::now();
          frame_ms[q].push_back(std::chrono::duration<double, std::milli>(f1 - f0).count());
          if (stage.num_optimizations_ > before) ba_ms[q].push_back(std::chrono::duration<double, std::milli>(f1 - s0).count());
          prev = frame;
        }
        stage.finish();
        solved[q] = stage.num_optimizations_;
      } catch (const std::exception& ex) {
        std::cerr << "sequence " << q << ": " << ex.what() << std::endl;
        failed++;
      }
    });
  }
  for (auto& th : threads) th.join();
  const double wall = std::chrono::duration<double>(Clock::now() - t0).count();
  if (failed) return 1;
  std::vector<double> all, ba;
  int windows = 0;
  for (int q = 0; q < n_seq; ++q) {
    all.insert(all.end(), frame_ms[q].begin(), frame_ms[q].end());
    ba.insert(ba.end(), ba_ms[q].begin(), ba_ms[q].end());
    windows += solved[q];
  }
  printf("sequences %d devices %d frames_per_sequence %d window %d frontend_stub_ms %.1f async %d windows_solved %d wall_s %.3f "
         "frames_per_s %.1f frame_latency_ms_p50 %.3f frame_latency_ms_p95 %.3f frame_latency_ms_p99 %.3f frame_latency_ms_max %.3f "
         "ba_frame_ms_p50 %.3f ba_frame_ms_p95 %.3f ba_frame_ms_max %.3f\n",
         n_seq, devices, n_frames, window, frontend_ms, async ? 1 : 0, windows, wall, all.size() / wall, percentile(all, 0.5),
         percentile(all, 0.95), percentile(all, 0.99), percentile(all, 1.0), percentile(ba, 0.5), percentile(ba, 0.95),
         percentile(ba, 1.0));
  return 0;
}
