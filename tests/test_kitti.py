"""KITTI odometry reader (text side of the reference's src/parse_KITTI.cpp): Python and C++ twins on a
synthetic sequence directory in KITTI's format.  The calibration values are sequence 00's, the same constants the
reference hard-codes in test/ba_test.cpp:23-28."""
import os
import subprocess

import numpy as np
import pytest

from stereo_visual_odometry_amd import hipba, kitti

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


def _write_curved_sequence(tmp_path, n=120):
    """A KITTI-layout directory (times.txt at 10 Hz, sequence 00's calib.txt, no images) and a ground-truth file of a
    car-like trajectory in KITTI's camera frame (z forward, x right, y down): ~1 m per frame along a gentle S-curve."""
    d = tmp_path / "07"
    d.mkdir()
    (d / "times.txt").write_text("".join("%.6e\n" % (0.1 * i) for i in range(n)))
    (d / "calib.txt").write_text(CALIB)
    lines, pos, yaw = [], np.zeros(3), 0.0
    for i in range(n):
        R = _rot([0, 1, 0], yaw)
        T = np.hstack([R, pos.reshape(3, 1)])
        lines.append(" ".join("%.9e" % v for v in T.reshape(-1)))
        yaw += 0.012 * np.sin(2 * np.pi * i / 60.0)
        pos = pos + _rot([0, 1, 0], yaw) @ np.array([0.0, 0.0, 0.9 + 0.2 * np.sin(0.3 * i)])
    gt = tmp_path / "07.txt"
    gt.write_text("\n".join(lines) + "\n")
    return str(d), str(gt)


@pytest.mark.gpu
def test_kitti_runner_over_a_sequence_directory(tmp_path):
    """The ROS-free runner (template: reference test/valgrind_main.cpp:20-42) on a KITTI-layout directory built here:
    loaders -> left camera's K -> per-frame stage with the reference's BA trigger -> GPU solves -> trajectory error
    against the ground-truth file.  The image half of the frontend is the synthetic one (no OpenCV here), which
    dead-reckons along the loaded trajectory: with bundle adjustment the absolute trajectory error must come out
    smaller than without, and the windows must have been solved on the GPU."""
    exe = os.path.join(LIB, "kitti_runner")
    d, gt = _write_curved_sequence(tmp_path)
    r = subprocess.run([exe, d, "--poses", gt, "--both"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    out = r.stdout.strip().split("\n")
    head = out[0].split()
    assert int(head[head.index("frames") + 1]) == 120 and int(head[head.index("cameras") + 1]) == 2
    assert float(head[head.index("fx") + 1]) == pytest.approx(718.856) and float(head[head.index("cy") + 1]) == pytest.approx(185.2157)
    assert float(head[head.index("baseline_m") + 1]) == pytest.approx(386.1448 / 718.856, rel=1e-6)
    off, on = out[1].split(), out[2].split()
    assert off[:2] == ["bundle_adjustment", "off"] and on[:2] == ["bundle_adjustment", "on"]
    get = lambda f, k: float(f[f.index(k) + 1])
    assert get(off, "windows") == 0 and get(on, "windows") == 5          # frames 20, 40, ... 100 (frontend.cpp:71)
    assert get(on, "ate_rmse_m") < 0.6 * get(off, "ate_rmse_m")
    assert get(on, "ate_rmse_m") < 1.0
    # the reference's trajectory bookkeeping (frontend.cpp:74-81): one entry per step, one more per splice
    assert get(off, "trajectory_entries") == 119
    # a sequence directory that is not there
    bad = subprocess.run([exe, d + "_missing"], capture_output=True, text=True)
    assert bad.returncode == 1 and "Could not open file" in bad.stderr


@pytest.mark.gpu
def test_whole_pipeline_with_the_gpu_steps(tmp_path):
    """The reference's main loop (src/main.cpp:45-58) and Frontend::visualOdometryPipeline (src/frontend.cpp:22-86) line
    by line, with Triangulator::triangulate, Estimator::estimate and Optimizer::optimize on the GPU and the image half
    (FAST, Lucas-Kanade: OpenCV, absent here) replaced by a scene of known landmarks with pixel noise, lost tracks and
    wrong matches (csrc/host/synthetic_frontend.hpp).  200 frames at 0.8 m each: the trajectory stays within a fraction
    of a percent of the distance travelled without bundle adjustment and a few percent with it (the reference's
    left-image-only bundle adjustment, anchored by the first frame's landmarks alone, does not improve a stereo odometry
    this accurate), windows are solved where frontend.cpp:71 says, nothing
    is reset, and the trajectory bookkeeping (frontend.cpp:74-81) has one entry per step."""
    exe = os.path.join(LIB, "vo_pipeline")
    r = subprocess.run([exe, "--frames", "200", "--both"], capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, r.stderr
    lines = [l.split() for l in r.stdout.strip().split("\n") if l.startswith("bundle_adjustment")]
    get = lambda f, k: float(f[f.index(k) + 1])
    off, on = lines
    assert off[1] == "off" and on[1] == "on"
    assert get(off, "windows") == 0 and get(on, "windows") == 9          # iterations 20, 40, ... 180
    assert get(off, "resets") == 0 and get(on, "resets") == 0
    # 160 m travelled.  Repeated runs of this very sequence gave 0.07-0.26 m without and 0.14-0.95 m with bundle
    # adjustment (the consensus amplifies last-bit differences of the solves, so runs differ): bounds with room
    assert get(off, "ate_rmse_m") < 1.0 and get(on, "ate_rmse_m") < 4.0
    assert get(off, "trajectory_entries") == 199
    assert get(off, "mean_features") > 500                                 # min_num_features_ (frontend.hpp:30)
    # along a KITTI-layout sequence's ground truth, with its calibration
    d, gt = _write_curved_sequence(tmp_path)
    r = subprocess.run([exe, "--kitti-dir", d, "--poses", gt, "--frames", "120"], capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, r.stderr
    on = [l.split() for l in r.stdout.strip().split("\n") if l.startswith("bundle_adjustment")][0]
    assert get(on, "windows") == 5 and get(on, "resets") == 0 and get(on, "ate_rmse_m") < 4.0
    # heavy mismatching: a quarter of the tracks wrong every frame; the consensus still finds the motion
    r = subprocess.run([exe, "--frames", "80", "--wrong", "0.25", "--no-ba"], capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, r.stderr
    off = [l.split() for l in r.stdout.strip().split("\n") if l.startswith("bundle_adjustment")][0]
    assert get(off, "resets") == 0 and get(off, "ate_rmse_m") < 1.0


@pytest.mark.gpu
def test_windows_the_pipeline_hands_to_bundle_adjustment_match_the_oracle(tmp_path, solver, oracle):
    """What Optimizer::optimize receives inside the running pipeline (dumped by the adapter's capture hook), replayed
    through the engine and through the oracle under the reference's options (optimizer.cpp:54-58): the same number of
    iterations (give or take one), final cost to 1e-6 relative, poses to 1e-6.  Landmarks straight ahead of a camera that drives forward
    have almost no parallax in the left images alone, and their depth is weakly determined: the typical landmark
    agrees to 1e-6 m, ninety-nine in a hundred to 1e-4 m, the worst one only to millimetres or centimetres, at the same
    cost."""
    from stereo_visual_odometry_amd import windows
    from test_gpu_parity import to_oracle
    exe = os.path.join(LIB, "vo_pipeline")
    d = tmp_path / "dump"
    d.mkdir()
    r = subprocess.run([exe, "--frames", "65", "--dump-dir", str(d)], capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, r.stderr
    files = sorted(os.listdir(d))
    assert files == ["window_000000.txt", "window_000001.txt", "window_000002.txt"]   # iterations 20, 40, 60
    for name in files[1:]:
        prob = windows.load_window(str(d / name))
        assert prob.poses.shape[0] == 30 and prob.obs_pose.shape[0] > 15000
        g = prob.copy()
        s = solver.solve(g, hipba.reference_production_options())
        o = to_oracle(oracle, prob)
        so, oits = oracle.solve(o, oracle.default_options(max_num_iterations=300, jacobian_mode=oracle.JAC_ANALYTIC, num_threads=16))
        assert s["termination_type"] == so["termination_type"] == hipba.CONVERGENCE
        # (the windows differ from run to run -- the consensus in front of them amplifies last-bit differences -- so a
        # stopping test decided by the last digits of a cost may fall either way once in a while)
        assert abs(len(s["iterations"]) - len(oits)) <= 1
        same = len(s["iterations"]) == len(oits)   # (one more step at convergence moves the cost by < 1e-6 of itself)
        assert abs(s["final_cost"] - so["final_cost"]) <= (1e-6 if same else 3e-6) * so["final_cost"]
        assert np.abs(g.poses - o.poses).max() <= (1e-6 if same else 1e-4)
        dq = np.abs(g.points - o.points).max(axis=1)
        assert np.median(dq) <= (1e-6 if same else 1e-4) and np.percentile(dq, 99) <= (1e-4 if same else 1e-2) and dq.max() <= 0.5
