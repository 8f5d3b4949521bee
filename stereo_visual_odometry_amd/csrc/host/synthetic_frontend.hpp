// synthetic_frontend.hpp -- the reference's whole per-frame pipeline (src/frontend.cpp:22-103, frontend.hpp:17-63)
// with the three GPU steps in place -- VO::Triangulator, VO::Estimator (frontend_steps.hpp) and VO::Optimizer
// (optimizer.hpp) -- and the image half, which is OpenCV on the CPU in the reference (FAST corners: detector.cpp:21-65;
// pyramidal Lucas-Kanade: matcher.cpp:13-119) and absent from this image, replaced by a scene that knows where its
// landmarks are: SyntheticDetector / SyntheticMatcher do the same bookkeeping on the frames' feature lists as
// Detector::detectFeatures, Matcher::matchStereo and Matcher::matchCircular, but take their pixels from the projection
// of known world points along a known trajectory, with pixel noise, lost tracks and wrong matches.
// VO::Frontend::visualOdometryPipeline below is the reference's control flow line by line.
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <memory>
#include <random>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "frontend_steps.hpp"
#include "kitti.hpp"
#include "optimizer.hpp"
#include "vo_types.hpp"

namespace VO {

class SyntheticScene {
public:
  struct Options {
    int points_per_frame = 60;    // landmarks scattered in the view of every frame of the trajectory
    double depth_min = 5.0, depth_max = 45.0;
    double pixel_sigma = 0.25;    // of a detected or tracked corner
    double stereo_sigma = 0.15;   // of a stereo match along the row
    double lost = 0.04;           // share of tracks the circular match loses per frame
    double wrong = 0.03;          // share of tracks that land on something else
    int width = 1241, height = 376;
  };
  // truth: camera-to-world poses of the left camera; the pipeline's world is the first frame (its pose is the identity,
  // frame.hpp:28), so the trajectory is re-expressed relative to it
  SyntheticScene(const std::vector<SE3d>& truth, const Camera& left, const Camera& right, unsigned seed, const Options& o)
      : left_(left), right_(right), opt_(o), rng_(seed) {
    const SE3d inv0 = se3Inverse(truth.at(0));
    for (const SE3d& t : truth) truth_.push_back(se3Multiply(inv0, t));
    std::uniform_real_distribution<double> uni(0.0, 1.0);
    for (std::size_t i = 0; i < truth_.size(); ++i) {
      for (int k = 0; k < o.points_per_frame; ++k) {
        const double z = o.depth_min + (o.depth_max - o.depth_min) * uni(rng_);
        const double u = o.width * uni(rng_), v = o.height * uni(rng_);
        const double pc[3] = {(u - left.K_(0, 2)) * z / left.K_(0, 0), (v - left.K_(1, 2)) * z / left.K_(1, 1), z};
        double w[3];
        se3Rotate(truth_[i].p, pc, w);
        points_.emplace_back(w[0] + truth_[i].p[4], w[1] + truth_[i].p[5], w[2] + truth_[i].p[6]);
        born_.push_back((int)i);
      }
    }
  }
  void advance() { prev_index_ = curr_index_; ++curr_index_; }  // a new stereo pair has arrived
  int prev_index() const { return prev_index_; }
  int curr_index() const { return curr_index_; }
  std::size_t frames() const { return truth_.size(); }
  const SE3d& truth(int i) const { return truth_[(std::size_t)i]; }

  // pixel of world point `point` in camera `cam` of frame `frame`; false when behind, too far or outside the image
  bool project(const Camera& cam, int frame, int point, Point2f* px) const {
    const Vector3d c = truth_[(std::size_t)frame].inverseApply(points_[(std::size_t)point]);
    if (!(c.z() > 1.0 && c.z() < 2.0 * opt_.depth_max)) return false;
    const double x = (cam.P_(0, 0) * c.x() + cam.P_(0, 2) * c.z() + cam.P_(0, 3)) / c.z();
    const double y = (cam.P_(1, 1) * c.y() + cam.P_(1, 2) * c.z() + cam.P_(1, 3)) / c.z();
    if (!(x >= 0.0 && x < opt_.width && y >= 0.0 && y < opt_.height)) return false;
    px->x = (float)x; px->y = (float)y;
    return true;
  }
  double gauss(double sigma) { return sigma * normal_(rng_); }
  double uniform() { return uniform_(rng_); }
  // which world point a feature shows: by the feature itself while it has no landmark, by its landmark afterwards
  int pointOf(const std::shared_ptr<Feature>& f) const {
    auto it = id_.find(f.get());
    if (it != id_.end()) return it->second;
    if (f->landmark_) { it = id_.find(f->landmark_.get()); if (it != id_.end()) return it->second; }
    return -1;
  }
  void remember(const void* key, int point) { id_[key] = point; }
  void forget() { id_.clear(); }

  Camera left_, right_;
  Options opt_;
  std::vector<Vector3d> points_;
  std::vector<int> born_;  // the frame in whose view a point was scattered

private:
  std::vector<SE3d> truth_;
  int prev_index_ = -1, curr_index_ = -1;
  std::unordered_map<const void*, int> id_;
  std::mt19937_64 rng_;
  std::normal_distribution<double> normal_{0.0, 1.0};
  std::uniform_real_distribution<double> uniform_{0.0, 1.0};
};

// detector.cpp:21-65: new corners of the previous frame's left image, strongest first, one per unoccupied grid cell,
// appended to the frame's left features (without landmarks).  Here: the scene's points in view that the frame does not
// track yet, in a random order.
class SyntheticDetector {
public:
  void detectFeatures(Context& context) {
    SyntheticScene& sc = *scene_;
    const int cols = (int)std::ceil((double)sc.opt_.width / grid_cell_size_), rows = (int)std::ceil((double)sc.opt_.height / grid_cell_size_);
    std::vector<bool> occ_grid((std::size_t)cols * rows, false);
    std::unordered_set<int> tracked;
    auto& features = context.frame_prev_->features_left_;
    for (const auto& feature : features) {
      const std::size_t idx = (std::size_t)((int)feature->pixel.y / grid_cell_size_) * cols + (std::size_t)((int)feature->pixel.x / grid_cell_size_);
      if (idx < occ_grid.size()) occ_grid[idx] = true;
      tracked.insert(sc.pointOf(feature));
    }
    const int frame = sc.prev_index();
    std::vector<std::pair<double, int>> candidates;  // (stand-in for the corner response, point)
    for (std::size_t p = 0; p < sc.points_.size(); ++p) {
      if (std::abs(sc.born_[p] - frame) > 80 || tracked.count((int)p)) continue;
      Point2f px;
      if (sc.project(sc.left_, frame, (int)p, &px)) candidates.emplace_back(sc.uniform(), (int)p);
    }
    std::sort(candidates.begin(), candidates.end());
    int new_feature_cnt = 0;
    for (const auto& c : candidates) {
      Point2f px;
      sc.project(sc.left_, frame, c.second, &px);
      px.x += (float)sc.gauss(sc.opt_.pixel_sigma); px.y += (float)sc.gauss(sc.opt_.pixel_sigma);
      if (px.x < 0 || px.y < 0) continue;
      const std::size_t idx = (std::size_t)((int)px.y / grid_cell_size_) * cols + (std::size_t)((int)px.x / grid_cell_size_);
      if (idx < occ_grid.size() && !occ_grid[idx]) {
        context.frame_prev_->pushFeatureLeft(px);
        sc.remember(context.frame_prev_->features_left_.back().get(), c.second);
        occ_grid[idx] = true;
        ++new_feature_cnt;
      }
    }
    if (verbose_) printf("%d new features added\n", new_feature_cnt);
  }
  int grid_cell_size_{10};
  bool verbose_ = false;
  std::shared_ptr<SyntheticScene> scene_;
};

class SyntheticMatcher {
public:
  // matcher.cpp:13-49: right-image matches for the left features that have none yet; features without a match go
  void matchStereo(Context& context) {
    SyntheticScene& sc = *scene_;
    auto& features_left = context.frame_prev_->features_left_;
    auto& features_right = context.frame_prev_->features_right_;
    const std::size_t features_right_size = features_right.size();
    std::vector<std::shared_ptr<Feature>> new_features_left(features_left.begin() + features_right_size, features_left.end());
    features_left.erase(features_left.begin() + features_right_size, features_left.end());
    for (const auto& f : new_features_left) {
      Point2f pr;
      const int p = sc.pointOf(f);
      if (p < 0 || !sc.project(sc.right_, sc.prev_index(), p, &pr)) continue;  // (status 0: the flow left the image)
      pr.x += (float)sc.gauss(sc.opt_.stereo_sigma);
      pr.y = f->pixel.y;  // rectified pair: the same row
      if (pr.x < 0) continue;
      features_left.push_back(f);
      features_right.push_back(pr);
    }
    features_left.shrink_to_fit();
  }
  // matcher.cpp:51-119: the previous frame's features tracked into the current stereo pair and checked around the
  // circle; what fails goes from the previous frame's lists, what holds is entered in the current frame's with the
  // landmark copied over
  void matchCircular(Context& context) {
    SyntheticScene& sc = *scene_;
    const std::vector<std::shared_ptr<Feature>> features_left_prev = context.frame_prev_->features_left_;
    const std::vector<Point2f> points_right_prev = context.frame_prev_->features_right_;
    context.frame_prev_->features_left_.clear();
    context.frame_prev_->features_right_.clear();
    for (std::size_t i = 0; i < features_left_prev.size(); ++i) {
      const int p = sc.pointOf(features_left_prev[i]);
      Point2f pl, pr;
      if (p < 0 || !sc.project(sc.left_, sc.curr_index(), p, &pl) || !sc.project(sc.right_, sc.curr_index(), p, &pr)) continue;
      if (sc.uniform() < sc.opt_.lost) continue;
      if (sc.uniform() < sc.opt_.wrong) {  // the flow locked onto something else
        pl.x = (float)(sc.opt_.width * sc.uniform()); pl.y = (float)(sc.opt_.height * sc.uniform());
        pr.x = std::fmax(0.0f, pl.x - (float)(60.0 * sc.uniform())); pr.y = pl.y;
      } else {
        pl.x += (float)sc.gauss(sc.opt_.pixel_sigma); pl.y += (float)sc.gauss(sc.opt_.pixel_sigma);
        pr.x += (float)sc.gauss(sc.opt_.pixel_sigma); pr.y = pl.y;
      }
      if (pl.x < 0 || pl.y < 0 || pr.x < 0 || pr.y < 0) continue;
      context.frame_prev_->features_left_.push_back(features_left_prev[i]);
      context.frame_prev_->features_right_.push_back(points_right_prev[i]);
      context.frame_curr_->pushFeatureLeftWithLandmark(pl, features_left_prev[i]->landmark_);
      context.frame_curr_->features_right_.push_back(pr);
      sc.remember(features_left_prev[i]->landmark_.get(), p);
    }
    context.frame_prev_->features_left_.shrink_to_fit();
    context.frame_prev_->features_right_.shrink_to_fit();
    if (verbose_) printf("%zu Features matched\n", context.frame_prev_->features_left_.size());
  }
  bool verbose_ = false;
  std::shared_ptr<SyntheticScene> scene_;
};

// frontend.hpp:17-63 / frontend.cpp:7-103, with the members' types swapped for the ones above
class Frontend {
public:
  Frontend(const Camera& camera_left, const Camera& camera_right, const std::shared_ptr<Map>& map,
           const std::shared_ptr<SyntheticScene>& scene)
      : camera_left_(camera_left), camera_right_(camera_right), map_(map), scene_(scene) {}

  void initialize() {  // frontend.cpp:13-20
    detector_.grid_cell_size_ = grid_cell_size_;
    detector_.scene_ = scene_;
    matcher_.scene_ = scene_;
    triangulator_ = Triangulator(camera_left_, camera_right_, map_, engine_);
    triangulator_.verbose_ = verbose_;
    estimator_ = std::make_unique<Estimator>(engine_);
    estimator_->max_delta_pose_norm_ = max_delta_pose_norm_;
    estimator_->verbose_ = verbose_;
    optimizer_.progress_to_stdout_ = false;
    optimizer_.dump_directory_ = dump_directory_;
    engine_->get();  // (the engine handle is created here, not inside the first frame's triangulation)
  }

  void visualOdometryPipeline() {  // frontend.cpp:22-86
    if (!context_.frame_prev_) return;
    do {
      // if number of features is below the threshold, add new features
      if (context_.frame_prev_->features_left_.size() < min_num_features_) {
        detector_.detectFeatures(context_);
        matcher_.matchStereo(context_);
      }
      // triangulate features in the previous frame
      tick(&time_triangulate_s_, [&] { triangulator_.triangulate(context_); });
      // get optical flow of previous features to current image (copy over the landmarks)
      matcher_.matchCircular(context_);
      // estimate pose of current frame using triangulated features of previous frame
      tick(&time_estimate_s_, [&] { estimation_status_ = estimator_->estimate(context_, camera_left_.K_); });
      if (!estimation_status_ && retried_) {
        throw std::runtime_error("Estimation failed twice.");
      } else if (!estimation_status_ && !retried_) {
        // reset map and context and try again
        resetContext();
        resetMap();
        iterations_ = 1;
        retried_ = true;
        ++num_resets_;
      }
    } while (!estimation_status_);
    retried_ = false;
    // insert current frame into map
    map_->insertFrame(context_.frame_curr_);
    // this function deletes landmarks in the map that have no features pointing to it
    map_->cleanMap();
    // bundle adjustment
    if (do_bundle_adjustment_ && iterations_ % bundle_adjustment_window_ == 0) {
      tick(&time_optimize_s_, [&] { optimizer_.optimize(map_, camera_left_.K_); });
      ++num_optimizations_;
      if (verbose_) {
        const hipba_summary& su = optimizer_.last_summary_;
        printf("bundle adjustment %d: %zu frames, cost %.6g -> %.6g, %d iterations, termination %d; map -> arrays %.2f ms, engine %.2f ms, back %.2f ms\n",
               num_optimizations_, map_->frames_.size(), su.initial_cost, su.final_cost, su.num_iterations, su.termination_type,
               optimizer_.last_flatten_ms_, optimizer_.last_solve_ms_, optimizer_.last_write_back_ms_);
      }
      // erase the poses that will get optimized
      if (poses_.size() < map_->frames_.size()) poses_.clear();
      else poses_.erase(poses_.end() - map_->frames_.size(), poses_.end());
      for (const std::shared_ptr<Frame>& frame : map_->frames_) poses_.push_back(frame->pose_);
    } else {
      poses_.push_back(context_.frame_prev_->pose_);
    }
    ++iterations_;
  }

  Context context_;
  std::vector<SE3d> poses_;
  // parameters (frontend.hpp:29-34)
  bool do_bundle_adjustment_{false};
  unsigned long min_num_features_{500};
  int grid_cell_size_{10};
  int bundle_adjustment_window_{20};
  double max_delta_pose_norm_{10.0};
  bool verbose_ = false;
  std::string dump_directory_;  // when set, every bundle-adjustment window is written there before it is solved
  // what the steps took, wall clock, over the whole run
  double time_triangulate_s_ = 0.0, time_estimate_s_ = 0.0, time_optimize_s_ = 0.0;
  int num_optimizations_ = 0, num_resets_ = 0;

private:
  template <class F>
  static void tick(double* total, F&& f) {
    const auto t0 = std::chrono::steady_clock::now();
    f();
    *total += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  void resetContext() {  // frontend.cpp:88-94
    context_.frame_prev_->features_left_.clear();
    context_.frame_prev_->features_right_.clear();
    context_.frame_curr_->features_left_.clear();
    context_.frame_curr_->features_right_.clear();
    scene_->forget();
  }
  void resetMap() {  // frontend.cpp:96-100
    map_ = std::make_shared<Map>();
    triangulator_.map_ = map_;
  }

  SyntheticDetector detector_;
  SyntheticMatcher matcher_;
  std::shared_ptr<EngineHandle> engine_ = std::make_shared<EngineHandle>();
  Triangulator triangulator_;
  std::unique_ptr<Estimator> estimator_;
  Optimizer optimizer_;
  Camera camera_left_;
  Camera camera_right_;
  std::shared_ptr<Map> map_;
  std::shared_ptr<SyntheticScene> scene_;
  unsigned long iterations_{1};
  bool estimation_status_ = false;
  bool retried_{false};
};

}  // namespace VO
