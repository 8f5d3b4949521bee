"""The text side of the reference's KITTI odometry reader (reference: src/parse_KITTI.cpp:14-114,
include/visual_odometry/parse_KITTI.hpp:13-33) and its camera type (sensor/camera.hpp:10-45), in Python:
times.txt -> stereo pair file names, calib.txt -> cameras, <sequence>.txt -> ground-truth poses.  Image decoding
stays with the frontend's owner.  The C++ twin is csrc/host/kitti.hpp."""
import numpy as np


class Camera:
    """reference: sensor/camera.hpp:13-41"""

    def __init__(self, fx=0.0, fy=0.0, cx=0.0, cy=0.0, tx=0.0, ty=0.0, tz=0.0):
        self.K_ = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=np.float64)
        self.P_ = np.array([[fx, 0, cx, tx], [0, fy, cy, ty], [0, 0, 1, tz]], dtype=np.float64)

    def pixelToWorld(self, pixel, depth):
        fx, fy, cx, cy = self.K_[0, 0], self.K_[1, 1], self.K_[0, 2], self.K_[1, 2]
        return np.array([(pixel[0] - cx) * depth / fx, (pixel[1] - cy) * depth / fy, depth], dtype=np.float32)


class StereoPair:
    """reference: parse_KITTI.hpp:13-18"""

    def __init__(self, left_frame, right_frame, time):
        self.left_frame_, self.right_frame_, self.time_ = left_frame, right_frame, time


def rotation_to_quaternion(R):
    """Unit quaternion (x y z w, w >= 0) of a rotation matrix, single precision like Sophus::SE3f."""
    R = np.asarray(R, dtype=np.float32)
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    if tr > 0:
        s = np.sqrt(tr + np.float32(1)) * 2
        q = [(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s]
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = np.sqrt(np.float32(1) + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = [0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s, (R[2, 1] - R[1, 2]) / s]
    elif R[1, 1] > R[2, 2]:
        s = np.sqrt(np.float32(1) + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = [(R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s, (R[0, 2] - R[2, 0]) / s]
    else:
        s = np.sqrt(np.float32(1) + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = [(R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s, (R[1, 0] - R[0, 1]) / s]
    q = np.array(q, dtype=np.float32)
    q = q / np.linalg.norm(q)
    return -q if q[3] < 0 else q


class ParseKITTI:
    def __init__(self):
        self.frames_, self.cameras_, self.ground_truth_poses_ = [], [], []

    def loadFrames(self, frame_filename):
        """parse_KITTI.cpp:14-37"""
        with open(frame_filename + "/times.txt") as f:
            self.frames_ = [StereoPair("%s/image_0/%06d.png" % (frame_filename, i), "%s/image_1/%06d.png" % (frame_filename, i),
                                       float(line)) for i, line in enumerate(l for l in f.read().split("\n") if l.strip())]

    def loadCameras(self, calib_filename):
        """parse_KITTI.cpp:39-65: fx = P[0], fy = P[5], cx = P[2], cy = P[6], fourth column P[3], P[7], P[11]"""
        with open(calib_filename) as f:
            for line in f:
                tok = line.split()
                if len(tok) < 13:
                    continue
                P = [float(v) for v in tok[1:13]]
                self.cameras_.append(Camera(P[0], P[5], P[2], P[6], P[3], P[7], P[11]))

    def loadGroundTruth(self, ground_truth_filename):
        """parse_KITTI.cpp:74-113: rows of [R | t], single precision; kept as qx qy qz qw tx ty tz"""
        with open(ground_truth_filename) as f:
            for line in f:
                tok = line.split()
                if len(tok) < 12:
                    continue
                T = np.array([float(v) for v in tok[:12]], dtype=np.float32).reshape(3, 4)
                q = rotation_to_quaternion(T[:, :3])
                self.ground_truth_poses_.append(np.concatenate([q, T[:, 3]]).astype(np.float64))


def trajectory_error(estimate, truth):
    """RMSE and mean of the per-frame translation differences (no alignment beyond the shared first frame)."""
    n = min(len(estimate), len(truth))
    if n == 0:
        return 0.0, 0.0
    d = np.linalg.norm(np.asarray(estimate)[:n, 4:7] - np.asarray(truth)[:n, 4:7], axis=1)
    return float(np.sqrt((d ** 2).mean())), float(d.mean())
