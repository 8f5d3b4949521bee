"""Host-side breakdown of hipba_solve on KITTI-like windows (needs scripts/build_setup_timing.sh): the analysis steps
go to stderr from the library, the call's own split is printed here."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stereo_visual_odometry_amd import hipba, scenes
hipba.LIB_PATH = os.path.join(ROOT, "scripts", "dev", "libhipba_setup.so")
solver = hipba.Solver(0)
wins = [scenes.kitti_like_window(1000 + i)[0] for i in range(3)]
solver.solve(wins[0].copy(), hipba.reference_production_options(), log=False)
for w in wins:
    p = w.copy()
    sys.stderr.write("---- window\n"); sys.stderr.flush()
    t0 = time.perf_counter()
    s = solver.solve(p, hipba.reference_production_options(), log=False)
    dt = time.perf_counter() - t0
    sys.stderr.write("call %.3f ms: setup %.3f ms, solve %.3f ms, total (library) %.3f ms, %d LM iterations, %d observations\n" % (
        1e3 * dt, 1e3 * s["time_setup_s"], 1e3 * s["time_solve_s"], 1e3 * s["time_total_s"], s["num_iterations"] - 1, w.obs_pose.shape[0]))
