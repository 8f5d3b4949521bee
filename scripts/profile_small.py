"""Profiling target: KITTI-like single windows (hipba_solve end to end) and 1024 batched small windows."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereo_visual_odometry_amd import hipba, scenes

def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    solver = hipba.Solver(0)
    if which in ("kitti", "both"):
        wins = [scenes.kitti_like_window(1000 + i)[0] for i in range(4)]
        solver.solve(wins[0].copy(), hipba.reference_production_options(), log=False)
        for w in wins:
            p = w.copy()
            t0 = time.perf_counter()
            s = solver.solve(p, hipba.reference_production_options(), log=False)
            print("kitti-like window: %.2f ms total, setup %.2f ms, solve %.2f ms, %d iterations" % (
                1e3 * (time.perf_counter() - t0), 1e3 * s["time_setup_s"], 1e3 * s["time_solve_s"], s["num_iterations"] - 1))
    if which in ("batch", "both"):
        n_w = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
        small = [scenes.small_window(2000 + i)[0] for i in range(n_w)]
        t0 = time.perf_counter()
        dp = solver.upload(small)
        print("upload of %d windows: %.1f ms" % (n_w, 1e3 * (time.perf_counter() - t0)))
        opts = hipba.reference_production_options()
        dp.solve(opts); dp.reset()
        t0 = time.perf_counter()
        res = dp.solve(opts)
        dt = time.perf_counter() - t0
        print("batched: %.2f ms, %d LM iterations max, %.0f windows/s" % (1e3 * dt, max(r["num_iterations"] for r in res) - 1, n_w / dt))

if __name__ == "__main__":
    main()
