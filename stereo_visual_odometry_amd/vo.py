"""Host-side mirror of the reference's BA interface, in Python, for tests and scripts.

Same names and meaning as the reference's C++ types so that a test here reads like the
reference's ``test/ba_test.cpp``:

  Map / Frame / Feature / MapPoint   reference: include/visual_odometry/sensor/{map,frame,feature,map_point}.hpp
  Optimizer.optimize(map, K)         reference: include/visual_odometry/solve/optimizer.hpp:18,
                                                src/solve/optimizer.cpp:14-63
  BundleAdjustmentStage.step(...)    reference: src/frontend.cpp:65-84 (window insert, landmark GC, BA trigger,
                                                trajectory splice)

``Optimizer.optimize`` flattens the map exactly as the reference builds its ceres::Problem
(frames oldest first, first frame and the landmarks it observes held constant, one residual
per left feature) and hands the arrays to the HIP engine through the C ABI.  The C++ twin of
this file is csrc/host/optimizer.{hpp,cpp}.
"""
import numpy as np

from . import hipba

NUM_ACTIVE_FRAMES = 30  # reference: sensor/map.hpp:24


class MapPoint:
    """reference: sensor/map_point.hpp:12-22"""

    def __init__(self, pose=(0.0, 0.0, 0.0)):
        self.pose_ = np.array(pose, dtype=np.float64)
        self.observations_ = []


class Feature:
    """reference: sensor/feature.hpp:15-33; pixel is a cv::Point2f, hence float32."""

    def __init__(self, frame, pixel, landmark=None):
        self.frame_ = frame
        self.pixel = np.array(pixel, dtype=np.float32)
        self.landmark_ = landmark


class Frame:
    """reference: sensor/frame.hpp:15-39; pose_ is T_world<-camera as qx qy qz qw tx ty tz."""

    def __init__(self, pose=None):
        self.pose_ = np.array([0, 0, 0, 1, 0, 0, 0], dtype=np.float64) if pose is None else np.array(pose, dtype=np.float64)
        self.features_left_ = []

    def pushFeatureLeftWithLandmark(self, pixel, landmark):
        f = Feature(self, pixel, landmark)
        self.features_left_.append(f)
        landmark.observations_.append(f)
        return f


class Map:
    """reference: sensor/map.hpp:13-25, src/sensor/map.cpp:8-21"""

    def __init__(self):
        self.landmarks_ = []  # insertion-ordered stand-in for the unordered_set
        self.frames_ = []

    def insertFrame(self, frame):
        self.frames_.append(frame)
        if len(self.frames_) > NUM_ACTIVE_FRAMES:
            self.frames_.pop(0)

    def insertLandmark(self, landmark):
        self.landmarks_.append(landmark)

    def cleanMap(self, keep_frames=()):
        """reference: src/sensor/map.cpp:23-50 -- erase the landmarks whose observing features have all expired.
        A feature lives as long as its frame does: in the window, or held by the caller (the reference's
        context_.frame_prev_/frame_curr_, passed here as keep_frames).  Returns the number removed."""
        alive = {id(f) for f in self.frames_} | {id(f) for f in keep_frames}
        kept = [lm for lm in self.landmarks_ if any(id(ft.frame_) in alive for ft in lm.observations_)]
        removed = len(self.landmarks_) - len(kept)
        self.landmarks_ = kept
        return removed


def flatten_map(map_, K, camera_model=hipba.CAMERA_Z_FORWARD, first_frame_constant=True):
    """Map -> hipba.Problem in canonical order (frames oldest first, landmarks by first appearance while
    walking frames_/features_left_, then unobserved ones).  Applies the constant rule of
    optimizer.cpp:28-36 when first_frame_constant.  Returns (problem, landmark list in problem order)."""
    index = {}
    lms = []

    def lm_index(lm):
        k = id(lm)
        if k not in index:
            index[k] = len(lms)
            lms.append(lm)
        return index[k]

    obs_pose, obs_point, obs_uv = [], [], []
    for fi, frame in enumerate(map_.frames_):
        for feat in frame.features_left_:
            obs_pose.append(fi)
            obs_point.append(lm_index(feat.landmark_))
            obs_uv.append((float(feat.pixel[0]), float(feat.pixel[1])))
    for lm in map_.landmarks_:
        lm_index(lm)
    poses = np.array([f.pose_ for f in map_.frames_], dtype=np.float64).reshape(-1, 7)
    points = np.array([lm.pose_ for lm in lms], dtype=np.float64).reshape(-1, 3)
    pose_const = np.zeros(len(map_.frames_), dtype=np.uint8)
    point_const = np.zeros(len(lms), dtype=np.uint8)
    if first_frame_constant and len(map_.frames_) > 0:
        pose_const[0] = 1
        for feat in map_.frames_[0].features_left_:
            point_const[index[id(feat.landmark_)]] = 1
    Kt = (K[0][0], K[1][1], K[0][2], K[1][2]) if np.ndim(K) == 2 else tuple(K)
    prob = hipba.Problem(poses, points, np.array(obs_pose, dtype=np.int32), np.array(obs_point, dtype=np.int32),
                         np.array(obs_uv, dtype=np.float64).reshape(-1, 2), Kt, camera_model, pose_const, point_const)
    return prob, lms


def write_back(map_, lms, prob):
    for f, p in zip(map_.frames_, prob.poses):
        f.pose_[:] = p
    for lm, p in zip(lms, prob.points):
        lm.pose_[:] = p


class Optimizer:
    """reference: VO::Optimizer (optimizer.hpp:14-19).  Stateless apart from the GPU handle."""

    def __init__(self, device=0, progress_to_stdout=False, dump_directory=None, dump_binary=False):
        self.solver = hipba.Solver(device)
        self.progress_to_stdout = progress_to_stdout
        self.last_summary = None
        self.dump_directory = dump_directory  # capture hook: every window is written there before it is solved
        self.dump_binary = dump_binary
        self.num_calls = 0

    def optimize(self, map_, K):
        """Drop-in for ``optimizer_.optimize(map_, camera_left_.K_)`` (reference: src/frontend.cpp:73):
        z-forward camera, first frame and its landmarks constant, no loss, 300 iterations
        (optimizer.cpp:28-36, 48, 58).  Updates the map in place; returns nothing, like the reference."""
        prob, lms = flatten_map(map_, K, hipba.CAMERA_Z_FORWARD, first_frame_constant=True)
        if self.dump_directory:
            from .windows import save_window
            save_window("%s/window_%06d.%s" % (self.dump_directory, self.num_calls, "bin" if self.dump_binary else "txt"),
                        prob, self.dump_binary)
        self.num_calls += 1
        opts = hipba.reference_production_options(
            minimizer_progress_to_stdout=1 if self.progress_to_stdout else 0)
        self.last_summary = self.solver.solve(prob, opts)
        write_back(map_, lms, prob)


class BundleAdjustmentStage:
    """The tail of ``Frontend::visualOdometryPipeline`` (reference: src/frontend.cpp:65-84): window insert, landmark
    GC, the BA trigger and the trajectory splice, quirks included (the non-BA branch appends the PREVIOUS
    frame's pose, frontend.cpp:81; a trajectory shorter than the window is cleared, frontend.cpp:75).
    `optimizer` is anything with ``optimize(map, K)``."""

    def __init__(self, optimizer, do_bundle_adjustment=False, bundle_adjustment_window=20):
        self.optimizer_ = optimizer
        self.do_bundle_adjustment_ = do_bundle_adjustment          # frontend.hpp:32
        self.bundle_adjustment_window_ = bundle_adjustment_window  # frontend.hpp:35
        self.poses_ = []                                           # frontend.hpp:29
        self.iterations_ = 1                                       # frontend.hpp:51
        self.num_optimizations_ = 0

    def step(self, map_, frame_prev, frame_curr, K):
        map_.insertFrame(frame_curr)                       # frontend.cpp:66
        map_.cleanMap(keep_frames=(frame_prev, frame_curr))  # frontend.cpp:68
        if self.do_bundle_adjustment_ and self.iterations_ % self.bundle_adjustment_window_ == 0:
            self.optimizer_.optimize(map_, K)              # frontend.cpp:73
            n = len(map_.frames_)
            if len(self.poses_) < n:                       # frontend.cpp:75-77
                self.poses_.clear()
            else:
                del self.poses_[len(self.poses_) - n:]
            self.poses_.extend(f.pose_.copy() for f in map_.frames_)
            self.num_optimizations_ += 1
        else:
            self.poses_.append(frame_prev.pose_.copy())    # frontend.cpp:81
        self.iterations_ += 1
