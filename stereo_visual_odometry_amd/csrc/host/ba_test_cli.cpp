// ba_test_cli.cpp -- the reference's stand-alone BA harness (test/ba_test.cpp) without ROS or RViz:
//   scene file -> map -> observations -> noisy copy -> solve on the GPU engine -> "Total error".
//
//   ba_test <ba_data.txt> [--noisy <file>] [--seed N] [--loss-scale S] [--production]
//
// --noisy reads a perturbed instance ("pose qx qy qz qw tx ty tz" / "landmark x y z" lines, same order as
// the scene) instead of drawing noise, so that runs are comparable across programs; without it noise is
// drawn from std::mt19937(seed) with the reference's standard deviations (ba_test.cpp:172-174).
// --production applies VO::Optimizer's constant-block rule and 300-iteration budget (optimizer.cpp:28-36,58)
// instead of ba_test's Huber loss / no constants / 50 iterations (ba_test.cpp:235-276).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <random>
#include <sstream>
#include <string>

#include "optimizer.hpp"
#include "vo_types.hpp"

using namespace VO;

// copied from KITTI (ba_test.cpp:23-28)
static const double fx = 7.188560000000e+02, fy = 7.188560000000e+02, cx = 6.071928000000e+02, cy = 1.852157000000e+02;
static const double frame_width = 1241.0, frame_height = 376.0;

struct Scene {
  std::shared_ptr<Map> map = std::make_shared<Map>();
  std::vector<std::shared_ptr<MapPoint>> landmarks;  // file order (the set's order is address dependent)
};

static Scene makeMapFromData(const std::string& filename) {
  std::ifstream file(filename);
  if (!file.is_open()) throw std::runtime_error("Could not open file: " + filename);
  Scene s;
  std::string line;
  int mode = 0;
  while (std::getline(file, line)) {
    if (line == "camera_poses") { mode = 1; continue; }
    if (line == "landmarks") { mode = 2; continue; }
    double v[6];
    if (mode == 1 && sscanf(line.c_str(), "%lf %lf %lf %lf %lf %lf", v, v + 1, v + 2, v + 3, v + 4, v + 5) == 6) {
      // R = Rz(yaw) Ry(pitch) Rx(roll) as a quaternion
      const double cr = cos(v[3] / 2), sr = sin(v[3] / 2), cp = cos(v[4] / 2), sp = sin(v[4] / 2);
      const double cyw = cos(v[5] / 2), syw = sin(v[5] / 2);
      auto frame = std::make_shared<Frame>();
      double* p = frame->pose_.data();
      p[0] = sr * cp * cyw - cr * sp * syw;
      p[1] = cr * sp * cyw + sr * cp * syw;
      p[2] = cr * cp * syw - sr * sp * cyw;
      p[3] = cr * cp * cyw + sr * sp * syw;
      p[4] = v[0]; p[5] = v[1]; p[6] = v[2];
      s.map->insertFrame(frame);
    } else if (mode == 2 && sscanf(line.c_str(), "%lf %lf %lf", v, v + 1, v + 2) == 3) {
      auto lm = std::make_shared<MapPoint>(Vector3d(v[0], v[1], v[2]));
      s.map->insertLandmark(lm);
      s.landmarks.push_back(lm);
    }
  }
  return s;
}

static bool project3DtoPixel(const Vector3d& c, Point2f& pixel) {
  pixel.x = static_cast<float>(-c.y() / c.x() * fx + cx);
  pixel.y = static_cast<float>(-c.z() / c.x() * fy + cy);
  return 0.0 <= pixel.x && pixel.x < frame_width && 0.0 <= pixel.y && pixel.y < frame_height;
}

static void makeFeaturesAndCorrespondence(Scene& s) {
  for (auto& frame : s.map->frames_) {
    for (auto& landmark : s.landmarks) {
      const Vector3d c = frame->pose_.inverseApply(landmark->pose_);
      if (c.x() < 0.0) continue;  // landmark is behind camera
      Point2f pixel;
      if (project3DtoPixel(c, pixel)) frame->pushFeatureLeftWithLandmark(pixel, landmark);
    }
  }
}

static void addNoise(Scene& s, unsigned seed) {
  std::mt19937 gen(seed);
  std::normal_distribution<double> nt(0, 0.1), nr(0, 0.05), nl(0, 0.1);
  for (auto& frame : s.map->frames_) {
    double* p = frame->pose_.data();
    p[4] += nt(gen); p[5] += nt(gen); p[6] += nt(gen);
    double q[4] = {p[0] + nr(gen), p[1] + nr(gen), p[2] + nr(gen), p[3] + nr(gen)};
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int k = 0; k < 4; ++k) p[k] = q[k] / n;
  }
  for (auto& lm : s.landmarks) for (int k = 0; k < 3; ++k) lm->pose_.data()[k] += nl(gen);
}

static void loadNoisy(Scene& s, const std::string& filename) {
  std::ifstream file(filename);
  if (!file.is_open()) throw std::runtime_error("Could not open file: " + filename);
  std::string tag;
  size_t fi = 0, li = 0;
  while (file >> tag) {
    if (tag == "pose" && fi < s.map->frames_.size()) {
      double* p = s.map->frames_[fi++]->pose_.data();
      for (int k = 0; k < 7; ++k) file >> p[k];
    } else if (tag == "landmark" && li < s.landmarks.size()) {
      double* p = s.landmarks[li++]->pose_.data();
      for (int k = 0; k < 3; ++k) file >> p[k];
    }
  }
}

static double calculateTotalError(const Scene& s) {
  double total = 0.0;
  for (const auto& frame : s.map->frames_)
    for (const auto& feature : frame->features_left_) {
      Point2f p;
      project3DtoPixel(frame->pose_.inverseApply(feature->landmark_->pose_), p);
      total += std::abs(p.x - feature->pixel.x) + std::abs(p.y - feature->pixel.y);
    }
  return total;
}

int main(int argc, char** argv) {
  if (argc < 2 || !strcmp(argv[1], "--help")) {
    std::cout << "usage: ba_test <ba_data.txt> [--noisy <file>] [--seed N] [--loss-scale S] [--production]\n";
    return argc < 2 ? 2 : 0;
  }
  std::string noisy;
  unsigned seed = 20241008u;
  double loss_scale = 1.0;
  bool production = false;
  for (int i = 2; i < argc; ++i) {
    if (!strcmp(argv[i], "--noisy") && i + 1 < argc) noisy = argv[++i];
    else if (!strcmp(argv[i], "--seed") && i + 1 < argc) seed = (unsigned)atol(argv[++i]);
    else if (!strcmp(argv[i], "--loss-scale") && i + 1 < argc) loss_scale = atof(argv[++i]);
    else if (!strcmp(argv[i], "--production")) production = true;
  }
  try {
    Scene s = makeMapFromData(argv[1]);
    makeFeaturesAndCorrespondence(s);
    size_t n_obs = 0;
    for (auto& f : s.map->frames_) n_obs += f->features_left_.size();
    std::cout << "frames " << s.map->frames_.size() << " landmarks " << s.landmarks.size() << " observations " << n_obs << "\n";
    std::cout << "Total error (ground truth): " << calculateTotalError(s) << std::endl;
    if (!noisy.empty()) loadNoisy(s, noisy); else addNoise(s, seed);
    std::cout << "Total error (noisy): " << calculateTotalError(s) << std::endl;
    Matx33d K;
    K(0, 0) = fx; K(1, 1) = fy; K(0, 2) = cx; K(1, 2) = cy;
    hipba_handle h = nullptr;
    if (hipba_create(0, &h) != HIPBA_OK) throw std::runtime_error(hipba_last_error());
    hipba_options o;
    hipba_options_default(&o);
    o.minimizer_progress_to_stdout = 1;
    if (production) o.max_num_iterations = 300;
    else { o.loss = HIPBA_LOSS_HUBER; o.loss_scale = loss_scale; }
    const hipba_summary sm = optimizeMap(h, s.map, K, HIPBA_CAMERA_X_FORWARD, production, o);
    hipba_destroy(h);
    std::cout << "final cost " << sm.final_cost << " iterations " << sm.num_iterations << std::endl;
    std::cout << "Total error: " << calculateTotalError(s) << std::endl;
  } catch (const std::exception& e) {
    std::cerr << "ba_test: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
