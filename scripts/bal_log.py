"""Prints the iteration log of the BAL-scale problem (10 LM iterations)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereo_visual_odometry_amd import hipba, scenes

solver = hipba.Solver(0)
prob, _, _ = scenes.bal_scale(seed=3000, n_points=int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000)
dp = solver.upload(prob)
o = hipba.default_options(max_num_iterations=10, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
for rep in range(2):
    dp.reset()
    t0 = time.perf_counter()
    s = dp.solve(o, log=True)[0]
    dt = time.perf_counter() - t0
    print("run %d: %.1f ms, %d log entries, termination %d" % (rep, 1e3 * dt, s["num_iterations"], s["termination_type"]))
for it in s["iterations"]:
    print(it)
