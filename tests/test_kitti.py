"""KITTI odometry reader (text side of the reference's src/parse_KITTI.cpp): Python and C++ twins on a
synthetic sequence directory in KITTI's format.  The calibration values are sequence 00's, the same constants the
reference hard-codes in test/ba_test.cpp:23-28."""
import os
import subprocess

import numpy as np
import pytest

from stereo_visual_odometry_amd import kitti

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(os.path.dirname(HERE), "stereo_visual_odometry_amd", "lib")

CALIB = """P0: 7.188560000000e+02 0.000000000000e+00 6.071928000000e+02 0.000000000000e+00 0.000000000000e+00 7.188560000000e+02 1.852157000000e+02 0.000000000000e+00 0.000000000000e+00 0.000000000000e+00 1.000000000000e+00 0.000000000000e+00
P1: 7.188560000000e+02 0.000000000000e+00 6.071928000000e+02 -3.861448000000e+02 0.000000000000e+00 7.188560000000e+02 1.852157000000e+02 0.000000000000e+00 0.000000000000e+00 0.000000000000e+00 1.000000000000e+00 0.000000000000e+00
"""


def _rot(axis, angle):
    axis = np.asarray(axis, dtype=np.float64) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * K @ K


@pytest.fixture
def sequence(tmp_path):
    d = tmp_path / "00"
    d.mkdir()
    times = [0.0, 0.1036, 0.2073, 0.3109]
    (d / "times.txt").write_text("".join("%.6e\n" % t for t in times))
    (d / "calib.txt").write_text(CALIB)
    rots = [np.eye(3), _rot([0, 1, 0], 0.02), _rot([1, 2, 0.5], 2.5), _rot([0, 0, 1], 3.1)]  # incl. negative-trace cases
    trans = [np.zeros(3), [0.01, -0.02, 0.85], [0.05, -0.03, 1.7], [0.11, -0.05, 2.6]]
    lines = []
    for R, t in zip(rots, trans):
        T = np.hstack([R, np.asarray(t, dtype=np.float64).reshape(3, 1)])
        lines.append(" ".join("%.9e" % v for v in T.reshape(-1)))
    gt = tmp_path / "00.txt"
    gt.write_text("\n".join(lines) + "\n")
    return str(d), str(gt), times, rots, trans


def test_python_reader(sequence):
    d, gt, times, rots, trans = sequence
    pk = kitti.ParseKITTI()
    pk.loadFrames(d)
    pk.loadCameras(d + "/calib.txt")
    pk.loadGroundTruth(gt)
    assert len(pk.frames_) == 4 and pk.frames_[2].left_frame_ == d + "/image_0/000002.png"
    assert pk.frames_[3].right_frame_ == d + "/image_1/000003.png" and pk.frames_[1].time_ == pytest.approx(0.1036)
    assert len(pk.cameras_) == 2
    K = pk.cameras_[0].K_
    assert (K[0, 0], K[1, 1], K[0, 2], K[1, 2]) == (718.856, 718.856, 607.1928, 185.2157)
    assert pk.cameras_[1].P_[0, 3] == -386.1448 and pk.cameras_[0].P_[0, 3] == 0.0
    assert np.allclose(pk.cameras_[0].pixelToWorld((607.1928, 185.2157), 10.0), [0, 0, 10])
    for pose, R, t in zip(pk.ground_truth_poses_, rots, trans):
        x, y, z, w = pose[:4]
        Rq = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                       [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                       [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        assert np.abs(Rq - R).max() < 1e-6 and np.abs(pose[4:] - t).max() < 1e-6  # single precision, like SE3f
        assert abs(np.linalg.norm(pose[:4]) - 1) < 1e-6 and w >= 0
    rmse, mean = kitti.trajectory_error(pk.ground_truth_poses_, pk.ground_truth_poses_)
    assert rmse == 0.0 and mean == 0.0
    shifted = [p + np.array([0, 0, 0, 0, 0.3, 0, 0.4]) for p in pk.ground_truth_poses_]
    rmse, mean = kitti.trajectory_error(shifted, pk.ground_truth_poses_)
    assert rmse == pytest.approx(0.5) and mean == pytest.approx(0.5)


def test_missing_files_raise(tmp_path):
    pk = kitti.ParseKITTI()
    with pytest.raises(OSError):
        pk.loadFrames(str(tmp_path / "nope"))
    with pytest.raises(OSError):
        pk.loadCameras(str(tmp_path / "nope.txt"))


def test_cpp_reader_agrees_with_python(sequence):
    exe = os.path.join(LIB, "kitti_info")
    if not os.path.exists(exe):
        pytest.skip("kitti_info not built")
    d, gt, times, rots, trans = sequence
    r = subprocess.run([exe, d, gt], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = r.stdout.split("\n")
    assert out[0].startswith("frames 4 first %s/image_0/000000.png" % d) and ("last %s/image_1/000003.png" % d) in out[0]
    assert out[1] == "cameras 2"
    f = out[2].split()
    assert [float(f[f.index(k) + 1]) for k in ("fx", "fy", "cx", "cy")] == [718.856, 718.856, 607.1928, 185.2157]
    f = out[3].split()
    assert float(f[f.index("t") + 1]) == -386.1448
    pk = kitti.ParseKITTI()
    pk.loadGroundTruth(gt)
    g = out[4].split()
    assert int(g[1]) == 4
    path = sum(np.linalg.norm(pk.ground_truth_poses_[i][4:] - pk.ground_truth_poses_[i - 1][4:]) for i in range(1, 4))
    assert float(g[3]) == pytest.approx(path, rel=1e-6)
    last = np.array([float(v) for v in out[5].split()[1:]])
    assert np.abs(last - pk.ground_truth_poses_[-1]).max() < 1e-6
    bad = subprocess.run([exe, d + "_missing"], capture_output=True, text=True)
    assert bad.returncode == 1 and "Could not open file" in bad.stderr
