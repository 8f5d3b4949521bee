// frontend_steps_cli -- the reference's per-frame steps around bundle adjustment (src/frontend.cpp:34-47:
// triangulator_.triangulate(context_), estimator_.estimate(context_, K)) driven through the drop-in classes of
// frontend_steps.hpp on a synthetic stereo pair of frames: landmarks triangulated from the previous frame's stereo
// matches, the current frame's pose estimated from them with a share of wrong matches among the features, and an
// implausible motion refused.  Checks its own results against the scene's truth; exit code 0 when all hold.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "frontend_steps.hpp"

using namespace VO;

static void project(const Camera& cam, const SE3d& pose, const double* world, float* px) {
  const Vector3d c = pose.inverseApply(Vector3d(world[0], world[1], world[2]));
  const double x = cam.P_(0, 0) * c.x() + cam.P_(0, 2) * c.z() + cam.P_(0, 3);
  const double y = cam.P_(1, 1) * c.y() + cam.P_(1, 2) * c.z() + cam.P_(1, 3);
  px[0] = (float)(x / c.z());
  px[1] = (float)(y / c.z());
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 600;
  const double outlier_share = argc > 2 ? atof(argv[2]) : 0.25;
  std::mt19937_64 rng(7);
  std::uniform_real_distribution<double> uni(0.0, 1.0);
  std::normal_distribution<double> gauss(0.0, 1.0);
  // KITTI sequence 00: fx = fy = 718.856, principal point (607.1928, 185.2157), baseline 0.537 m
  const Camera left(718.856, 718.856, 607.1928, 185.2157, 0.0, 0.0, 0.0);
  const Camera right(718.856, 718.856, 607.1928, 185.2157, -386.1448, 0.0, 0.0);
  auto map = std::make_shared<Map>();
  auto engine = std::make_shared<EngineHandle>();
  Triangulator triangulator(left, right, map, engine);
  Estimator estimator(engine);
  triangulator.verbose_ = estimator.verbose_ = false;

  // previous frame: a few degrees of yaw, somewhere in the world; current frame: 0.8 m ahead, turned a little more
  auto prev = std::make_shared<Frame>(), curr = std::make_shared<Frame>();
  const double yaw0 = 0.05, yaw1 = 0.08;
  prev->pose_.p[1] = std::sin(0.5 * yaw0); prev->pose_.p[3] = std::cos(0.5 * yaw0);
  prev->pose_.p[4] = 3.0; prev->pose_.p[5] = -0.2; prev->pose_.p[6] = 10.0;
  SE3d truth;
  truth.p[1] = std::sin(0.5 * yaw1); truth.p[3] = std::cos(0.5 * yaw1);
  truth.p[4] = 3.05; truth.p[5] = -0.21; truth.p[6] = 10.8;
  std::vector<double> world(3 * (size_t)n);
  for (int i = 0; i < n; ++i) {
    // a point in view of the previous frame, 5 to 40 m ahead
    const double z = 5.0 + 35.0 * uni(rng), u = 1241.0 * uni(rng), v = 376.0 * uni(rng);
    const double pc[3] = {(u - 607.1928) * z / 718.856, (v - 185.2157) * z / 718.856, z};
    double w[3];
    se3Rotate(prev->pose_.p, pc, w);
    for (int k = 0; k < 3; ++k) world[3 * i + k] = w[k] + prev->pose_.p[4 + k];
    float pl[2], pr[2];
    project(left, prev->pose_, &world[3 * i], pl);
    project(right, prev->pose_, &world[3 * i], pr);
    Point2f a, b;
    a.x = pl[0] + (float)(0.1 * gauss(rng)); a.y = pl[1] + (float)(0.1 * gauss(rng));
    b.x = pr[0] + (float)(0.1 * gauss(rng)); b.y = a.y;
    prev->pushFeatureLeft(a);
    prev->features_right_.push_back(b);
  }
  Context context;
  context.insertFrame(prev);
  context.insertFrame(curr);
  int failures = 0;

  // ---- Triangulator::triangulate ------------------------------------------------------------------------------
  triangulator.triangulate(context);
  double worst_rel = 0.0;
  for (int i = 0; i < n; ++i) {
    const auto& lm = prev->features_left_[(size_t)i]->landmark_;
    if (!lm) { ++failures; continue; }
    double d2 = 0.0, r2 = 0.0;
    for (int k = 0; k < 3; ++k) {
      d2 += (lm->pose_.v[k] - world[3 * i + k]) * (lm->pose_.v[k] - world[3 * i + k]);
      r2 += (world[3 * i + k] - prev->pose_.p[4 + k]) * (world[3 * i + k] - prev->pose_.p[4 + k]);
    }
    worst_rel = std::fmax(worst_rel, std::sqrt(d2 / r2));
  }
  printf("triangulated %zu landmarks into the map, worst relative error %.3g (0.1 px of disparity noise)\n", map->landmarks_.size(), worst_rel);
  if ((int)map->landmarks_.size() != n || !(worst_rel < 0.05)) { printf("FAIL: triangulation\n"); ++failures; }
  triangulator.triangulate(context);  // every feature has its landmark now: nothing to do
  if ((int)map->landmarks_.size() != n) { printf("FAIL: second triangulation changed the map\n"); ++failures; }

  // ---- Estimator::estimate --------------------------------------------------------------------------------------
  int n_wrong = 0;
  std::vector<char> wrong((size_t)n, 0);
  for (int i = 0; i < n; ++i) {
    float px[2];
    project(left, truth, &world[3 * i], px);
    Point2f a;
    a.x = px[0] + (float)(0.3 * gauss(rng)); a.y = px[1] + (float)(0.3 * gauss(rng));
    if (uni(rng) < outlier_share) { a.x = (float)(1241.0 * uni(rng)); a.y = (float)(376.0 * uni(rng)); wrong[(size_t)i] = 1; ++n_wrong; }
    curr->pushFeatureLeftWithLandmark(a, prev->features_left_[(size_t)i]->landmark_);
    curr->features_right_.push_back(a);
  }
  const bool ok = estimator.estimate(context, left.K_);
  double dt = 0.0, dq = 0.0;
  const double sign = curr->pose_.p[3] * truth.p[3] + curr->pose_.p[1] * truth.p[1] < 0 ? -1.0 : 1.0;
  for (int k = 0; k < 3; ++k) dt = std::fmax(dt, std::fabs(curr->pose_.p[4 + k] - truth.p[4 + k]));
  for (int k = 0; k < 4; ++k) dq = std::fmax(dq, std::fabs(sign * curr->pose_.p[k] - truth.p[k]));
  printf("estimate: %s, %zu of %d features kept (%d wrong matches planted), pose off by %.3g m / %.3g (quaternion), %d iterations by the confidence bound\n",
         ok ? "accepted" : "refused", curr->features_left_.size(), n, n_wrong, dt, dq, estimator.last_iterations_);
  if (!ok || !(dt < 0.05) || !(dq < 2e-3)) { printf("FAIL: pose\n"); ++failures; }
  if (curr->features_left_.size() != curr->features_right_.size()) { printf("FAIL: feature lists differ\n"); ++failures; }
  if ((int)curr->features_left_.size() > n - n_wrong + n_wrong / 10 || (int)curr->features_left_.size() < (n - n_wrong) * 95 / 100) {
    printf("FAIL: inlier set\n"); ++failures;
  }

  // ---- an implausible motion is refused and leaves the pose alone (estimator.cpp:60-65) -------------------------
  auto next = std::make_shared<Frame>();
  for (size_t i = 0; i < curr->features_left_.size(); ++i) {
    next->pushFeatureLeftWithLandmark(curr->features_left_[i]->pixel, curr->features_left_[i]->landmark_);
    next->features_right_.push_back(curr->features_right_[i]);
  }
  next->pose_.p[4] = 123.0;
  context.insertFrame(next);
  curr->pose_.p[6] += 100.0;  // what the pipeline believes of the previous frame is 100 m from where the matches put this one
  const bool refused = !estimator.estimate(context, left.K_);
  printf("implausible motion (100 m against a bound of %g): %s, pose left at x = %g\n", estimator.max_delta_pose_norm_,
         refused ? "refused" : "accepted", next->pose_.p[4]);
  if (!refused || next->pose_.p[4] != 123.0) { printf("FAIL: implausible motion\n"); ++failures; }
  if (next->features_left_.size() != next->features_right_.size() || next->features_left_.empty()) { printf("FAIL: features after a refusal\n"); ++failures; }

  // ---- se3 helpers: T * T^-1 = identity, |log| of a known motion --------------------------------------------------
  const SE3d id = se3Multiply(truth, se3Inverse(truth));
  if (!(se3LogNorm(id) < 1e-12)) { printf("FAIL: T T^-1\n"); ++failures; }
  SE3d rot;  // a pure rotation by 0.3 rad about y: |log| = 0.3
  rot.p[1] = std::sin(0.15); rot.p[3] = std::cos(0.15);
  if (!(std::fabs(se3LogNorm(rot) - 0.3) < 1e-12)) { printf("FAIL: log of a rotation\n"); ++failures; }
  SE3d tr;  // a pure translation: |log| = |t|
  tr.p[4] = 3.0; tr.p[5] = 4.0;
  if (!(std::fabs(se3LogNorm(tr) - 5.0) < 1e-12)) { printf("FAIL: log of a translation\n"); ++failures; }

  printf(failures ? "FAILED (%d)\n" : "ALL OK\n", failures);
  return failures ? 1 : 0;
}
