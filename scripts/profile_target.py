"""Profiling target: the benchmark's BAL-scale problem, each hot-path kernel a few times, then four LM iterations.
Run under rocprofv3 (kernel trace or one --pmc pass at a time)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereo_visual_odometry_amd import hipba, scenes

def main():
    n_points = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    solver = hipba.Solver(0)
    prob, _, _ = scenes.bal_scale_bench(seed=3000, n_points=n_points)
    dp = solver.upload(prob)
    o = hipba.default_options(max_num_iterations=4)
    for k in range(6):
        ms = dp.time_kernel(k, 3 if k != hipba.KERNEL_REDUCED_SOLVE else 1, o)
        print("%-18s %.3f ms" % (hipba.KERNEL_NAMES[k], ms))
    dp.reset()
    s = dp.solve(o)[0]
    print("4 LM iterations: %.1f ms, cost %.6e -> %.6e, %d accepted" % (
        1e3 * s["time_solve_s"], s["initial_cost"], s["final_cost"], s["num_successful_steps"] - 1))
    print("workload: bal_scale %d %d %d world 1" % (prob.poses.shape[0], prob.points.shape[0], prob.obs_pose.shape[0]))

if __name__ == "__main__":
    main()
