"""A bundle-adjustment window on disk ("hipba-window 1"): what ``VO::Optimizer::optimize`` reads out of the map
(reference: src/solve/optimizer.cpp:19-52) and nothing else -- poses, landmarks, observations, constant flags,
intrinsics and the camera model.  It succeeds the reference's only fixture format, ``test/ba_data.txt``
(test/ba_test.cpp:30-85), which holds a scene but neither observations nor flags.  The C++ twin, with the
format description, is csrc/host/window_io.hpp; files written by either side load on the other.
"""
import struct

import numpy as np

from . import hipba

_MAGIC_TEXT = "hipba-window 1 text"
_MAGIC_BIN = b"hipba-window 1 binary\n"


def save_window(path, prob, binary=False):
    """Writes a hipba.Problem.  Text doubles carry 17 significant digits, so a round trip is exact."""
    n_poses, n_points, n_obs = len(prob.poses), len(prob.points), len(prob.obs_pose)
    pose_const = np.zeros(n_poses, np.uint8) if prob.pose_const is None else np.asarray(prob.pose_const, np.uint8)
    point_const = np.zeros(n_points, np.uint8) if prob.point_const is None else np.asarray(prob.point_const, np.uint8)
    fx, fy, cx, cy = (float(v) for v in prob.K)
    if binary:
        with open(path, "wb") as f:
            f.write(_MAGIC_BIN)
            f.write(struct.pack("<i4diiq", int(prob.camera_model), fx, fy, cx, cy, n_poses, n_points, n_obs))
            f.write(np.ascontiguousarray(prob.poses, "<f8").tobytes())
            f.write(pose_const.tobytes())
            f.write(np.ascontiguousarray(prob.points, "<f8").tobytes())
            f.write(point_const.tobytes())
            f.write(np.ascontiguousarray(prob.obs_pose, "<i4").tobytes())
            f.write(np.ascontiguousarray(prob.obs_point, "<i4").tobytes())
            f.write(np.ascontiguousarray(prob.obs_uv, "<f8").tobytes())
        return
    g = lambda v: "%.17g" % float(v)
    with open(path, "w") as f:
        f.write(_MAGIC_TEXT + "\n")
        f.write("camera %s %s %s %s %s\n" % ("x" if prob.camera_model == hipba.CAMERA_X_FORWARD else "z", g(fx), g(fy), g(cx), g(cy)))
        f.write("poses %d\n" % n_poses)
        for c, p in zip(pose_const, prob.poses):
            f.write("%d %s\n" % (c, " ".join(g(v) for v in p)))
        f.write("points %d\n" % n_points)
        for c, p in zip(point_const, prob.points):
            f.write("%d %s\n" % (c, " ".join(g(v) for v in p)))
        f.write("observations %d\n" % n_obs)
        for ip, iq, uv in zip(prob.obs_pose, prob.obs_point, np.asarray(prob.obs_uv).reshape(-1, 2)):
            f.write("%d %d %s %s\n" % (ip, iq, g(uv[0]), g(uv[1])))
        f.write("end\n")


def load_window(path):
    """Reads either encoding and returns a hipba.Problem."""
    with open(path, "rb") as f:
        header = f.readline()
        if header == _MAGIC_BIN:
            model, fx, fy, cx, cy, n_poses, n_points, n_obs = struct.unpack("<i4diiq", f.read(struct.calcsize("<i4diiq")))
            if min(n_poses, n_points, n_obs) < 0:
                raise ValueError("negative count in %s" % path)

            def take(dtype, count):
                a = np.frombuffer(f.read(np.dtype(dtype).itemsize * count), dtype=dtype)
                if a.size != count:
                    raise ValueError("truncated window file %s" % path)
                return a

            poses = take("<f8", 7 * n_poses).reshape(-1, 7).copy()
            pose_const = take(np.uint8, n_poses).copy()
            points = take("<f8", 3 * n_points).reshape(-1, 3).copy()
            point_const = take(np.uint8, n_points).copy()
            obs_pose = take("<i4", n_obs).copy()
            obs_point = take("<i4", n_obs).copy()
            uv = take("<f8", 2 * n_obs).reshape(-1, 2).copy()
        elif header.decode("ascii", "replace").strip() == _MAGIC_TEXT:
            tok = f.read().decode("ascii").split()
            pos = 0

            def expect(word):
                nonlocal pos
                if pos >= len(tok) or tok[pos] != word:
                    raise ValueError("expected '%s' in %s" % (word, path))
                pos += 1

            expect("camera")
            if tok[pos] not in ("z", "x"):
                raise ValueError("camera model must be z or x in %s" % path)
            model = hipba.CAMERA_X_FORWARD if tok[pos] == "x" else hipba.CAMERA_Z_FORWARD
            fx, fy, cx, cy = (float(v) for v in tok[pos + 1:pos + 5])
            pos += 5
            expect("poses")
            n_poses = int(tok[pos]); pos += 1
            rec = np.array(tok[pos:pos + 8 * n_poses], dtype=np.float64).reshape(-1, 8); pos += 8 * n_poses
            pose_const, poses = rec[:, 0].astype(np.uint8), rec[:, 1:].copy()
            expect("points")
            n_points = int(tok[pos]); pos += 1
            rec = np.array(tok[pos:pos + 4 * n_points], dtype=np.float64).reshape(-1, 4); pos += 4 * n_points
            point_const, points = rec[:, 0].astype(np.uint8), rec[:, 1:].copy()
            expect("observations")
            n_obs = int(tok[pos]); pos += 1
            rec = np.array(tok[pos:pos + 4 * n_obs], dtype=np.float64).reshape(-1, 4); pos += 4 * n_obs
            obs_pose, obs_point, uv = rec[:, 0].astype(np.int32), rec[:, 1].astype(np.int32), rec[:, 2:].copy()
        else:
            raise ValueError("not a hipba-window version 1 file: %s" % path)
    if n_obs and (obs_pose.min() < 0 or obs_pose.max() >= n_poses or obs_point.min() < 0 or obs_point.max() >= n_points):
        raise ValueError("observation index out of range in %s" % path)
    return hipba.Problem(poses, points, obs_pose, obs_point, uv, (fx, fy, cx, cy), model, pose_const, point_const)
