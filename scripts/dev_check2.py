"""Developer check: large-window path (MFMA Cholesky, PCG), batched windows, kernel timers."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereo_visual_odometry_amd import hipba, scenes
from oracle import oracle
from dev_check import to_oracle, compare

def main():
    solver = hipba.Solver(0)
    prob, _, _ = scenes.bal_scale(seed=3000, grid=(10, 10), n_points=20000, depth_lo=5.0, depth_hi=30.0)
    print("bal 100 cams: obs", prob.obs_pose.shape[0])
    oo = oracle.default_options(max_num_iterations=8, jacobian_mode=oracle.JAC_ANALYTIC, num_threads=8)
    compare("bal100 cholesky", prob, hipba.default_options(max_num_iterations=8, linear_solver=hipba.LINEAR_DENSE_CHOLESKY), oo, solver)
    compare("bal100 pcg", prob, hipba.default_options(max_num_iterations=8, linear_solver=hipba.LINEAR_PCG, pcg_max_iterations=2000, pcg_tolerance=1e-13), oo, solver)
    # batched small windows
    wins = [scenes.small_window(2000 + w)[0] for w in range(16)]
    t0 = time.time()
    res = solver.solve_batched([w.copy() for w in wins], hipba.default_options(max_num_iterations=300), log=False)
    print("batched 16 windows: %.1f ms" % (1e3 * (time.time() - t0)), [r['num_iterations'] for r in res])
    o = to_oracle(wins[5]); os_, _ = oracle.solve(o, oracle.default_options(max_num_iterations=300, jacobian_mode=oracle.JAC_ANALYTIC, num_threads=8))
    g = [w.copy() for w in wins]
    res = solver.solve_batched(g, hipba.default_options(max_num_iterations=300))
    print("window 5: final cost gpu %.9e oracle %.9e  dpose %.3e dpoint %.3e" % (res[5]['final_cost'], os_['final_cost'], np.abs(g[5].poses - o.poses).max(), np.abs(g[5].points - o.points).max()))
    # kernel timers on a mid-size resident problem
    big, _, _ = scenes.bal_scale(seed=3000, grid=(20, 25), n_points=250000)
    dp = solver.upload(big)
    for k in range(6):
        ms = dp.time_kernel(k, 5)
        print("kernel %-18s %.3f ms" % (hipba.KERNEL_NAMES[k], ms))
    dp.reset()
    t0 = time.time()
    s = dp.solve(hipba.default_options(max_num_iterations=5, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0), log=True)[0]
    print("bal500: 5 iterations %.1f ms; costs" % (1e3 * (time.time() - t0)), ["%.6e" % it['cost'] for it in s['iterations']])

if __name__ == '__main__':
    main()
