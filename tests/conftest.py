import json
import os
import sys

import numpy as np
import pytest

# torch first: its wheel carries its own copy of the HIP runtime, and a process that has already initialised the
# system copy through libhipba.so may find torch unable to see the GPU afterwards (the tests that need both are
# the collective ones)
try:
    import torch  # noqa: F401
except ImportError:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure); built with gcc on first use."""
    from oracle import oracle as orc
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN_DIR, "ba_golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def solver():
    from stereo_visual_odometry_amd import hipba
    s = hipba.Solver(0)
    yield s
    s.close()


def problem_from_dict(d, cls):
    return cls(np.array(d["poses"]), np.array(d["points"]), np.array(d["obs_pose"], dtype=np.int32),
               np.array(d["obs_point"], dtype=np.int32), np.array(d["obs_uv"]), d["K"], d["camera_model"],
               None if d["pose_const"] is None else np.array(d["pose_const"], dtype=np.uint8),
               None if d["point_const"] is None else np.array(d["point_const"], dtype=np.uint8))


def to_oracle(orc, p):
    return orc.Problem(p.poses, p.points, p.obs_pose, p.obs_point, p.obs_uv, p.K, p.camera_model, p.pose_const,
                       p.point_const)
