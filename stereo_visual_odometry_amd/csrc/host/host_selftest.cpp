// host_selftest.cpp -- checks of the C++ host side that need no GPU: the map stand-ins, the flattening of a map into
// the C ABI's arrays and back, the stage around the optimizer when bundle adjustment is off, the window files and the
// KITTI text readers.  (The parts that solve are exercised through ba_test, ba_replay, ba_stage_demo and ba_sequences.)
// Usage: host_selftest <scratch directory>; exit status 0 when every check holds.
#include <cmath>
#include <cstdio>
#include <fstream>
#include <string>

#include "frontend_ba.hpp"
#include "kitti.hpp"
#include "optimizer.hpp"
#include "vo_types.hpp"
#include "window_io.hpp"

namespace {

int g_checks = 0, g_failed = 0;
#define CHECK(cond)                                                        \
  do {                                                                     \
    ++g_checks;                                                            \
    if (!(cond)) { ++g_failed; printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); } \
  } while (0)

using namespace VO;

std::shared_ptr<Frame> make_frame(double z) {
  auto f = std::make_shared<Frame>();
  f->pose_.p[3] = 1.0;  // identity rotation (qx qy qz qw)
  f->pose_.p[6] = z;
  return f;
}

// reference: src/sensor/map.cpp:8-50
void test_map() {
  Map map;
  std::vector<std::shared_ptr<MapPoint>> lm;
  for (int i = 0; i < 40; ++i) {
    auto frame = make_frame(i);
    auto p = std::make_shared<MapPoint>(Vector3d(0.1 * i, 0.0, 10.0 + i));
    frame->pushFeatureLeftWithLandmark(Point2f{10.f + i, 20.f}, p);
    if (!lm.empty()) frame->pushFeatureLeftWithLandmark(Point2f{30.f, 40.f + i}, lm.back());  // also sees the previous one
    map.insertLandmark(p);
    lm.push_back(p);
    map.insertFrame(frame);
  }
  CHECK(map.frames_.size() == 30);                       // map.cpp:8-16: the window is capped
  CHECK(map.frames_.front()->pose_.p[6] == 10.0);
  CHECK(map.landmarks_.size() == 40);
  // landmark i is seen by frames i and i + 1: frames 0..9 are gone, so landmarks 0..8 have no live observer
  CHECK(map.cleanMap() == 9);
  CHECK(map.landmarks_.size() == 31);
  CHECK(map.cleanMap() == 0);
}

// reference: src/solve/optimizer.cpp:19-52
void test_flatten_and_write_back() {
  auto map = std::make_shared<Map>();
  auto shared = std::make_shared<MapPoint>(Vector3d(1.0, 2.0, 30.0));
  auto only_second = std::make_shared<MapPoint>(Vector3d(-1.0, 0.5, 12.0));
  auto unobserved = std::make_shared<MapPoint>(Vector3d(9.0, 9.0, 9.0));
  auto f0 = make_frame(0.0), f1 = make_frame(1.0);
  f0->pushFeatureLeftWithLandmark(Point2f{100.5f, 50.25f}, shared);
  f1->pushFeatureLeftWithLandmark(Point2f{101.f, 51.f}, only_second);
  f1->pushFeatureLeftWithLandmark(Point2f{99.f, 49.f}, shared);
  map->insertLandmark(shared); map->insertLandmark(only_second); map->insertLandmark(unobserved);
  map->insertFrame(f0); map->insertFrame(f1);
  Matx33d K;
  K.val[0] = 700.0; K.val[4] = 710.0; K.val[2] = 600.0; K.val[5] = 180.0; K.val[8] = 1.0;
  FlatWindow w;
  flattenMap(map, K, HIPBA_CAMERA_Z_FORWARD, true, &w);
  const hipba_problem& p = w.problem;
  CHECK(p.n_poses == 2 && p.n_points == 3 && p.n_obs == 3);
  CHECK(p.fx == 700.0 && p.fy == 710.0 && p.cx == 600.0 && p.cy == 180.0);
  CHECK(p.pose_const[0] == 1 && p.pose_const[1] == 0);             // optimizer.cpp:28-31
  CHECK(p.obs_pose[0] == 0 && p.obs_pose[1] == 1 && p.obs_pose[2] == 1);
  CHECK(p.obs_point[0] == 0 && p.obs_point[1] == 1 && p.obs_point[2] == 0);  // numbered by first appearance
  CHECK(p.point_const[0] == 1 && p.point_const[1] == 0 && p.point_const[2] == 0);  // optimizer.cpp:32-36
  CHECK(p.obs_uv[0] == 100.5 && p.obs_uv[1] == 50.25);
  CHECK(p.points[3 * 2] == 9.0);                                     // the landmark nobody observes comes last
  CHECK(p.poses[7 * 1 + 6] == 1.0 && p.poses[3] == 1.0);
  // the solver writes through the flat arrays; writeBack carries that into the map's own blocks
  w.poses[7 * 1 + 4] = 0.25;
  w.points[3 * 1 + 2] = 12.5;
  writeBack(w);
  CHECK(f1->pose_.p[4] == 0.25 && only_second->pose_.v[2] == 12.5 && shared->pose_.v[2] == 30.0);
}

// reference: src/frontend.cpp:65-84 with do_bundle_adjustment_ = false (nothing is solved)
void test_stage_without_bundle_adjustment() {
  BundleAdjustmentStage<Map, Frame, SE3d> stage;
  CHECK(!stage.do_bundle_adjustment_ && stage.bundle_adjustment_window_ == 20 && stage.iterations_ == 1);
  auto map = std::make_shared<Map>();
  Matx33d K;
  auto prev = make_frame(0.0);
  for (int i = 1; i <= 45; ++i) {
    auto curr = make_frame(i);
    stage.step(map, prev, curr, K);
    prev = curr;
  }
  CHECK(stage.poses_.size() == 45 && stage.iterations_ == 46 && stage.num_optimizations_ == 0);
  CHECK(stage.poses_.front().p[6] == 0.0 && stage.poses_.back().p[6] == 44.0);  // frontend.cpp:81: the PREVIOUS frame's pose
  CHECK(map->frames_.size() == 30);
}

void test_window_files(const std::string& dir) {
  WindowData a;
  a.poses = {0, 0, 0, 1, 0, 0, 0, 0.1, 0, 0, 0.995, 1, 2, 3};
  a.points = {1, 2, 30, -4, 5, 60.5, 7, 8, 9};
  a.uv = {100.25, 50.5, 640.0, 180.0, 12.125, 300.0};
  a.obs_pose = {0, 1, 1};
  a.obs_point = {0, 0, 2};
  a.pose_const = {1, 0};
  a.point_const = {1, 0, 0};
  a.bind();
  a.problem.fx = 718.856; a.problem.fy = 718.856; a.problem.cx = 607.1928; a.problem.cy = 185.2157;
  a.problem.camera_model = HIPBA_CAMERA_X_FORWARD;
  for (int binary = 0; binary < 2; ++binary) {
    const std::string path = dir + (binary ? "/w.bin" : "/w.txt");
    writeWindow(path, a.problem, binary != 0);
    WindowData b;
    readWindow(path, &b);
    CHECK(b.poses == a.poses && b.points == a.points && b.uv == a.uv);
    CHECK(b.obs_pose == a.obs_pose && b.obs_point == a.obs_point);
    CHECK(b.pose_const == a.pose_const && b.point_const == a.point_const);
    CHECK(b.problem.fx == a.problem.fx && b.problem.cy == a.problem.cy && b.problem.camera_model == HIPBA_CAMERA_X_FORWARD);
    CHECK(b.problem.n_poses == 2 && b.problem.n_points == 3 && b.problem.n_obs == 3);
  }
  // what is not a window file is refused
  { std::ofstream f(dir + "/bad.txt"); f << "hipba-window 7 text\n"; }
  bool threw = false;
  WindowData c;
  try { readWindow(dir + "/bad.txt", &c); } catch (const std::exception&) { threw = true; }
  CHECK(threw);
  threw = false;
  try { readWindow(dir + "/missing.txt", &c); } catch (const std::exception&) { threw = true; }
  CHECK(threw);
}

// reference: src/parse_KITTI.cpp:14-113
void test_kitti_text_readers(const std::string& dir) {
  { std::ofstream f(dir + "/times.txt"); f << "0.000000e+00\n1.036224e-01\n2.072800e-01\n"; }
  {
    std::ofstream f(dir + "/calib.txt");
    f << "P0: 718.856 0 607.1928 0 0 718.856 185.2157 0 0 0 1 0\n";
    f << "P1: 718.856 0 607.1928 -386.1448 0 718.856 185.2157 0 0 0 1 0\n";
  }
  {
    std::ofstream f(dir + "/poses.txt");
    f << "1 0 0 0 0 1 0 0 0 0 1 0\n";
    f << "0 0 1 1.5 0 1 0 -0.25 -1 0 0 2.0\n";  // a quarter turn about y
  }
  ParseKITTI kitti;
  kitti.loadFrames(dir);
  kitti.loadCameras(dir + "/calib.txt");
  kitti.loadGroundTruth(dir + "/poses.txt");
  CHECK(kitti.frames_.size() == 3 && std::fabs(kitti.frames_[1].time_ - 0.1036224) < 1e-12);
  CHECK(kitti.frames_[2].left_frame_ == dir + "/image_0/000002.png" && kitti.frames_[2].right_frame_ == dir + "/image_1/000002.png");
  CHECK(kitti.cameras_.size() == 2 && kitti.cameras_[1].P_(0, 3) == -386.1448 && kitti.cameras_[0].K_.val[2] == 607.1928);
  CHECK(kitti.ground_truth_poses_.size() == 2 && kitti.ground_truth_poses_[0].p[3] == 1.0);
  const SE3d& q = kitti.ground_truth_poses_[1];
  CHECK(std::fabs(q.p[1] - std::sqrt(0.5)) < 1e-6 && std::fabs(q.p[3] - std::sqrt(0.5)) < 1e-6 && q.p[4] == 1.5 && q.p[6] == 2.0);
  double rmse = -1.0, mean = -1.0;
  trajectoryError(kitti.ground_truth_poses_, kitti.ground_truth_poses_, &rmse, &mean);
  CHECK(rmse == 0.0 && mean == 0.0);
  bool threw = false;
  try { ParseKITTI other; other.loadFrames(dir + "/nowhere"); } catch (const std::exception&) { threw = true; }
  CHECK(threw);
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) { printf("usage: host_selftest <scratch directory>\n"); return 2; }
  const std::string dir = argv[1];
  test_map();
  test_flatten_and_write_back();
  test_stage_without_bundle_adjustment();
  test_window_files(dir);
  test_kitti_text_readers(dir);
  printf("%d checks, %d failed\n", g_checks, g_failed);
  return g_failed ? 1 : 0;
}
