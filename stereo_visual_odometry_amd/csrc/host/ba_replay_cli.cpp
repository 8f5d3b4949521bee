// ba_replay_cli.cpp -- replays captured bundle-adjustment windows (window_io.hpp) on the GPU engine.
//
//   ba_replay <window file>... [--batched] [--huber S] [--max-iterations N] [--progress]
//             [--out-dir D] [--binary] [--repeat R]
//
// Default options are the production ones (reference: src/solve/optimizer.cpp:54-58: 300 iterations, no loss;
// the constant flags and the camera model travel in the file).  One line per window:
//   file poses points observations iterations termination initial_cost final_cost setup_ms solve_ms
// --batched solves all windows with one hipba_solve_batched call (independent windows side by side).
// --out-dir writes every solved window as <D>/<basename>.solved (same encoding flag as --binary).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "window_io.hpp"

static std::string base_name(const std::string& p) {
  const size_t s = p.find_last_of('/');
  return s == std::string::npos ? p : p.substr(s + 1);
}

int main(int argc, char** argv) {
  std::vector<std::string> files;
  bool batched = false, binary = false, progress = false;
  double huber = 0.0;
  int max_iterations = 300, repeat = 1;
  std::string out_dir;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--batched")) batched = true;
    else if (!strcmp(argv[i], "--binary")) binary = true;
    else if (!strcmp(argv[i], "--progress")) progress = true;
    else if (!strcmp(argv[i], "--huber") && i + 1 < argc) huber = atof(argv[++i]);
    else if (!strcmp(argv[i], "--max-iterations") && i + 1 < argc) max_iterations = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--repeat") && i + 1 < argc) repeat = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--out-dir") && i + 1 < argc) out_dir = argv[++i];
    else if (!strcmp(argv[i], "--help")) { files.clear(); break; }
    else files.push_back(argv[i]);
  }
  if (files.empty()) {
    std::cout << "usage: ba_replay <window file>... [--batched] [--huber S] [--max-iterations N] [--progress]\n"
                 "                 [--out-dir D] [--binary] [--repeat R]\n";
    return 2;
  }
  try {
    std::vector<VO::WindowData> windows(files.size());
    for (size_t i = 0; i < files.size(); ++i) VO::readWindow(files[i], &windows[i]);
    hipba_handle h = nullptr;
    if (hipba_create(0, &h) != HIPBA_OK) throw std::runtime_error(hipba_last_error());
    hipba_options o;
    hipba_options_default(&o);
    o.max_num_iterations = max_iterations;
    o.minimizer_progress_to_stdout = progress ? 1 : 0;
    if (huber > 0.0) { o.loss = HIPBA_LOSS_HUBER; o.loss_scale = huber; }
    std::vector<hipba_summary> summaries(files.size());
    std::vector<std::vector<double>> poses0(files.size()), points0(files.size());
    for (size_t i = 0; i < files.size(); ++i) { poses0[i] = windows[i].poses; points0[i] = windows[i].points; }
    double wall_ms = 0.0;
    for (int r = 0; r < repeat; ++r) {
      for (size_t i = 0; i < files.size(); ++i) {  // every repeat starts from the file's values
        windows[i].poses = poses0[i]; windows[i].points = points0[i];
        windows[i].bind();
      }
      const auto t0 = std::chrono::steady_clock::now();
      if (batched) {
        std::vector<hipba_problem> probs(files.size());
        for (size_t i = 0; i < files.size(); ++i) probs[i] = windows[i].problem;
        if (hipba_solve_batched(h, (int32_t)probs.size(), probs.data(), &o, summaries.data()) != HIPBA_OK)
          throw std::runtime_error(hipba_last_error());
      } else {
        for (size_t i = 0; i < files.size(); ++i)
          if (hipba_solve(h, &windows[i].problem, &o, &summaries[i]) != HIPBA_OK) throw std::runtime_error(hipba_last_error());
      }
      wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    static const char* term[] = {"CONVERGENCE", "NO_CONVERGENCE", "FAILURE"};
    for (size_t i = 0; i < files.size(); ++i) {
      const hipba_summary& s = summaries[i];
      const hipba_problem& p = windows[i].problem;
      printf("%s poses %d points %d observations %lld iterations %d %s initial_cost %.17g final_cost %.17g setup_ms %.3f solve_ms %.3f\n",
             base_name(files[i]).c_str(), p.n_poses, p.n_points, (long long)p.n_obs, s.num_iterations - 1,
             term[s.termination_type >= 0 && s.termination_type <= 2 ? s.termination_type : 2], s.initial_cost, s.final_cost,
             1e3 * s.time_setup_s, 1e3 * s.time_solve_s);
      if (!out_dir.empty()) VO::writeWindow(out_dir + "/" + base_name(files[i]) + ".solved", p, binary);
    }
    printf("total %zu windows in %.3f ms (%s)\n", files.size(), wall_ms, batched ? "batched" : "one by one");
    hipba_destroy(h);
  } catch (const std::exception& e) {
    std::cerr << "ba_replay: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
