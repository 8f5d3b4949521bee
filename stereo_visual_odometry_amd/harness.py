"""The reference's stand-alone BA harness (test/ba_test.cpp) without ROS: scene file ->
map -> observations -> noisy copy -> solve -> L1 reprojection error.

Function names follow the reference's helpers:
  makeMapFromData                ba_test.cpp:30-85
  project3DtoPixel               ba_test.cpp:87-100
  makeFeaturesAndCorrespondence  ba_test.cpp:102-126
  copyMap / copyMapAndAddNoise   ba_test.cpp:128-202  (noise from a seeded numpy generator
                                 instead of std::random_device, so runs are reproducible)
  optimizeLieAlgebra             ba_test.cpp:235-276  (x-forward camera, Huber loss, no constant
                                 blocks, Ceres' default 50 iterations)
  calculateTotalError            ba_test.cpp:278-298
"""
import numpy as np

from . import hipba
from .vo import Frame, Map, MapPoint, flatten_map, write_back

# copied from KITTI (reference: ba_test.cpp:23-28)
FX = 7.188560000000e+02
FY = 7.188560000000e+02
CX = 6.071928000000e+02
CY = 1.852157000000e+02
FRAME_WIDTH = 1241.0
FRAME_HEIGHT = 376.0
INTRINSICS = np.array([[FX, 0.0, CX], [0.0, FY, CY], [0.0, 0.0, 1.0]])


def _quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz])


def _quat_rotate(q, p):
    qv = np.asarray(q[:3])
    uv = 2.0 * np.cross(qv, p)
    return p + q[3] * uv + np.cross(qv, uv)


def pose_inverse_apply(pose, p_w):
    """frame->pose_.inverse() * w_point"""
    qi = np.array([-pose[0], -pose[1], -pose[2], pose[3]])
    return _quat_rotate(qi, np.asarray(p_w) - pose[4:7])


def makeMapFromData(filename):
    """Lines 'x y z roll pitch yaw' under camera_poses (R = Rz(yaw) Ry(pitch) Rx(roll)), 'x y z' under landmarks."""
    map_ = Map()
    mode = None
    with open(filename) as f:
        for line in f:
            line = line.strip()
            if line == "camera_poses":
                mode = "poses"
                continue
            if line == "landmarks":
                mode = "landmarks"
                continue
            vals = line.split()
            if mode == "poses" and len(vals) == 6:
                x, y, z, roll, pitch, yaw = (float(v) for v in vals)
                qz = np.array([0, 0, np.sin(yaw / 2), np.cos(yaw / 2)])
                qy = np.array([0, np.sin(pitch / 2), 0, np.cos(pitch / 2)])
                qx = np.array([np.sin(roll / 2), 0, 0, np.cos(roll / 2)])
                q = _quat_mul(_quat_mul(qz, qy), qx)
                q /= np.linalg.norm(q)
                map_.insertFrame(Frame(np.concatenate([q, [x, y, z]])))
            elif mode == "landmarks" and len(vals) == 3:
                map_.insertLandmark(MapPoint([float(v) for v in vals]))
    return map_


def project3DtoPixel(point_3d, intrinsics=INTRINSICS, frame_width=FRAME_WIDTH, frame_height=FRAME_HEIGHT):
    """x-forward pinhole; the pixel is rounded to float32 like cv::Point2f.  Returns (inside, pixel)."""
    px = -point_3d[1] / point_3d[0] * intrinsics[0, 0] + intrinsics[0, 2]
    py = -point_3d[2] / point_3d[0] * intrinsics[1, 1] + intrinsics[1, 2]
    pixel = np.array([px, py], dtype=np.float32)
    inside = (0.0 <= pixel[0] < frame_width) and (0.0 <= pixel[1] < frame_height)
    return bool(inside), pixel


def makeFeaturesAndCorrespondence(map_, intrinsics=INTRINSICS):
    for frame in map_.frames_:
        for landmark in map_.landmarks_:
            c_point = pose_inverse_apply(frame.pose_, landmark.pose_)
            if c_point[0] < 0.0:
                continue  # landmark is behind camera
            inside, pixel = project3DtoPixel(c_point, intrinsics)
            if inside:
                frame.pushFeatureLeftWithLandmark(pixel, landmark)


def copyMap(map_):
    cpy = Map()
    mp = {}
    for lm in map_.landmarks_:
        c = MapPoint(lm.pose_.copy())
        cpy.insertLandmark(c)
        mp[id(lm)] = c
    for frame in map_.frames_:
        fc = Frame(frame.pose_.copy())
        cpy.insertFrame(fc)
        for feat in frame.features_left_:
            fc.pushFeatureLeftWithLandmark(feat.pixel.copy(), mp[id(feat.landmark_)])
    return cpy


def copyMapAndAddNoise(map_, seed=20241008, pose_translation_stddev=0.1, pose_rotation_stddev=0.05,
                       landmark_stddev=0.1):
    cpy = copyMap(map_)
    rng = np.random.default_rng(seed)
    for frame in cpy.frames_:
        frame.pose_[4:7] += rng.normal(0.0, pose_translation_stddev, 3)
        q = frame.pose_[0:4] + rng.normal(0.0, pose_rotation_stddev, 4)
        frame.pose_[0:4] = q / np.linalg.norm(q)  # SE3::setQuaternion normalises
    for lm in cpy.landmarks_:
        lm.pose_ += rng.normal(0.0, landmark_stddev, 3)
    return cpy


def optimizeLieAlgebra(map_, intrinsics=INTRINSICS, loss_function_scale=1.0, solver=None, options=None):
    """ba_test's solve on the GPU engine.  Returns the summary dict."""
    prob, lms = flatten_map(map_, intrinsics, hipba.CAMERA_X_FORWARD, first_frame_constant=False)
    opts = options if options is not None else hipba.default_options(loss=hipba.LOSS_HUBER,
                                                                     loss_scale=loss_function_scale)
    s = solver if solver is not None else hipba.Solver(0)
    summary = s.solve(prob, opts)
    write_back(map_, lms, prob)
    return summary


def calculateTotalError(map_, intrinsics=INTRINSICS):
    total = 0.0
    for frame in map_.frames_:
        for feat in frame.features_left_:
            c = pose_inverse_apply(frame.pose_, feat.landmark_.pose_)
            _, p = project3DtoPixel(c, intrinsics)
            total += abs(float(p[0]) - float(feat.pixel[0]))
            total += abs(float(p[1]) - float(feat.pixel[1]))
    return total
