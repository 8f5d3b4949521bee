"""Randomised parity sweep of the GATHERED elimination (one large window, at least two million 6x6 products: the
landmark records, the pair lists, one lane or one wave per block of the reduced matrix) against the CPU oracle with
analytic Jacobians: camera grids of random shape, depth range, observations per landmark, loss and camera numbering.
Prints every case that leaves the tolerances of the parity tests.

Usage: python scripts/fuzz_gathered.py [n_cases] [first_seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from oracle import oracle
from stereo_visual_odometry_amd import hipba, scenes
from test_gpu_parity import ATOL_STATE, RTOL_COST, assert_logs_match, to_oracle
from test_headline_parity import _shuffled_cameras


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 70000
    solver = hipba.Solver(0)
    bad = 0
    t0 = time.time()
    for seed in range(first, first + n_cases):
        rng = np.random.default_rng(seed)
        gx, gy = int(rng.integers(4, 16)), int(rng.integers(4, 16))
        degree = int(rng.integers(5, 11))
        lo = float(rng.choice([5.0, 10.0, 20.0]))
        hi = lo * float(rng.choice([3.0, 6.0]))
        # (the generator pushes a landmark deeper until `degree` cameras see it: a landmark in a corner of the grid, at
        # the middle of the depth range, must have that many in view or it never ends)
        z_mid = 0.5 * (lo + hi)
        in_view = min(gx, int(0.80 * z_mid / 2.0) + 1) * min(gy, int(0.24 * z_mid / 2.0) + 1)
        degree = max(2, min(degree, in_view))
        # enough landmarks for two million products: degree (degree + 1) / 2 per landmark
        n_points = int(2.2e6 / (degree * (degree + 1) / 2)) + int(rng.integers(0, 20000))
        prob, _, _ = scenes.bal_scale(seed=seed, grid=(gx, gy), n_points=n_points, degree=degree, depth_lo=lo,
                                      depth_hi=hi, sigma_t=float(rng.choice([0.05, 0.2])),
                                      sigma_w=float(rng.choice([0.005, 0.02])))
        if rng.uniform() < 0.5:
            prob = _shuffled_cameras(prob, seed)
        huber = rng.uniform() < 0.4
        kw = dict(loss=hipba.LOSS_HUBER, loss_scale=0.8) if huber else {}
        okw = dict(loss=oracle.LOSS_HUBER, loss_scale=0.8) if huber else {}
        g = prob.copy()
        s = solver.solve(g, hipba.default_options(max_num_iterations=5, **kw))
        o = to_oracle(oracle, prob)
        so, oits = oracle.solve(o, oracle.default_options(max_num_iterations=5, jacobian_mode=oracle.JAC_ANALYTIC, num_threads=16, **okw))
        tag = "seed %d [grid %dx%d, %d landmarks x %d, depth from %g, huber %d]" % (seed, gx, gy, n_points, degree, lo, huber)
        try:
            assert s["termination_type"] == so["termination_type"], "termination %d vs %d" % (s["termination_type"], so["termination_type"])
            assert abs(s["final_cost"] - so["final_cost"]) <= RTOL_COST * so["final_cost"], "final cost"
            assert_logs_match(s["iterations"], oits)
            dp, dq = np.abs(g.poses - o.poses).max(), np.abs(g.points - o.points).max()
            assert dp <= ATOL_STATE and dq <= ATOL_STATE, "state: poses %.2e landmarks %.2e" % (dp, dq)
            print(tag, "ok, %d iterations" % len(oits), flush=True)
        except AssertionError as e:
            bad += 1
            print(tag, "OUT OF TOLERANCE:", e, flush=True)
    print("%d of %d in tolerance, %.0f s" % (n_cases - bad, n_cases, time.time() - t0))


if __name__ == "__main__":
    main()
