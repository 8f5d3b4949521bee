"""A slice of the randomised parity sweep (scripts/fuzz_parity.py: random window shapes, constant patterns, robust loss
with outliers, repeated observations, iteration budgets) as a test: the engine against the oracle, one by one and
batched.  Well-posed problems are compared in full (termination, every logged cost, final cost 1e-6, poses 1e-6 m,
landmarks 1e-5 m); the others -- landmarks seen once, gauge left free -- on the first iterations of the log, where
rounding has had no time to grow."""
import importlib.util
import os

import pytest

from stereo_visual_odometry_amd import hipba

HERE = os.path.dirname(os.path.abspath(__file__))


def _sweep():
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(HERE), "scripts", "fuzz_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.gpu
def test_random_windows_match_oracle(solver, oracle):
    fz = _sweep()
    held, strict_seen = [], 0
    for seed in range(70000, 70060):
        prob, okw, name = fz.random_case(seed)
        if prob.obs_pose.shape[0] == 0:
            continue
        o = oracle.Problem(prob.poses, prob.points, prob.obs_pose, prob.obs_point, prob.obs_uv, prob.K, prob.camera_model,
                           prob.pose_const, prob.point_const)
        so, oits = oracle.solve(o, oracle.default_options(jacobian_mode=oracle.JAC_ANALYTIC, **okw))
        g = prob.copy()
        s = solver.solve(g, hipba.default_options(**okw), log=True)
        strict = fz.well_posed(prob)
        strict_seen += strict
        assert fz.compare(name, seed, s, s["iterations"], g, so, oits, o, strict), (seed, name)
        if okw["loss"] == hipba.LOSS_NONE and okw["max_num_iterations"] == 25 and prob.poses.shape[0] <= 30:
            held.append((seed, name, prob, so, oits, o))
    assert strict_seen >= 5 and len(held) >= 3
    dp = solver.upload([h[2].copy() for h in held])
    res = dp.solve(hipba.default_options(max_num_iterations=25, loss=hipba.LOSS_NONE), log=True)
    for w, (seed, name, prob, so, oits, o) in enumerate(held):
        poses, points = dp.download(w)
        g = prob.copy()
        g.poses[:] = poses
        g.points[:] = points
        assert fz.compare(name + " (batched)", seed, res[w], res[w]["iterations"], g, so, oits, o, fz.well_posed(prob)), seed
    dp.free()
