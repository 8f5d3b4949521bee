"""Development aid: phase timestamps of k_point_schur workgroups (needs scripts/build_timing.sh)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereo_visual_odometry_amd import hipba, scenes

def main():
    here = os.path.dirname(os.path.abspath(__file__))
    lib = hipba.load_library(os.path.join(here, "dev", "libhipba_timing.so"))
    hipba._lib = lib
    solver = hipba.Solver(0)
    which = sys.argv[1] if len(sys.argv) > 1 else "kitti"
    if which == "kitti":
        probs = [scenes.kitti_like_window(1000)[0]]
    elif which in ("bal", "potrf"):
        probs = [scenes.bal_scale(seed=3000, n_points=250_000)[0]]
    else:
        probs = [scenes.small_window(2000 + i)[0] for i in range(256)]
    dp = solver.upload(probs)
    if which == "potrf":
        ms = dp.time_kernel(hipba.KERNEL_REDUCED_SOLVE, 1)
        n = 4096
        buf = (C.c_ulonglong * (8 * n))()
        lib.hipba_debug_ticks.argtypes = [C.c_void_p, C.c_int]
        lib.hipba_debug_ticks(buf, 8 * n)
        t = np.frombuffer(buf, dtype=np.uint64).astype(np.int64).reshape(n, 8)
        print("reduced_solve %.2f ms" % ms)
        for k in (2, 20, 40, 60, 92):
            f, u, st, wk = t[k], t[2048 + k], t[1024 + k], t[3072 + k]
            print("block %2d (shader cycles): factoring workgroup waits %d, load %d, 8 tile steps %d, store %d, inverse %d" % (
                k, u[1] - u[0], f[1] - f[0], f[2] - f[1], f[3] - f[2], f[4] - f[3]))
            print("   strip 0: solve %d, publish %d, block rows %d, waits for all strips %d, band update %d; strip 3: block rows %d" % (
                st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[7] - st[4], st[6] - st[5]))
            print("   first plain worker: waits for the strips %d, band update %d" % (wk[1] - wk[0], wk[2] - wk[1]))
        return
    ms = dp.time_kernel(hipba.KERNEL_POINT_SCHUR, 5)
    print("point_schur %.1f us" % (1e3 * ms))
    n = 4096
    buf = (C.c_ulonglong * (8 * n))()
    lib.hipba_debug_ticks.argtypes = [C.c_void_p, C.c_int]
    rc = lib.hipba_debug_ticks(buf, 8 * n)
    t = np.frombuffer(buf, dtype=np.uint64).reshape(n, 8).astype(np.int64)
    print('rc', rc, 'nonzero', int((t != 0).sum()), hipba._lib is lib)
    t = t[t[:, 0] > 0]
    base = t[:, 0].min()
    d = np.diff(t, axis=1) * 10e-3  # 100 MHz ticks -> us
    print("rc", rc, "workgroups", len(t))
    print("start offsets (us): min %.1f max %.1f" % (0.0, (t[:, 0].max() - base) * 10e-3))
    names = ["setup", "phase1", "phase2", "stageYW", "phase3", "gpart", "flush"]
    for i, nm in enumerate(names):
        print("%-8s mean %8.2f us  max %8.2f us" % (nm, d[:, i].mean(), d[:, i].max()))
    if which == "kitti":
        ms = dp.time_kernel(hipba.KERNEL_REDUCED_SOLVE, 5)
        lib.hipba_debug_ticks(buf, 8 * n)
        tt = np.frombuffer(buf, dtype=np.uint64).reshape(n, 8).astype(np.int64)
        print("factor loop, shader cycles: wave 0 panel %d, wait %d, diagonal tile %d, wait %d; wave 1 panel %d, wait %d, trailing %d, wait %d" % (
            *tt[1][:4], *tt[2][:4]))
        print("   wave 4: panel %d, wait %d, forward blocks (and tiles) %d, wait %d" % tuple(tt[5][:4]))
        t = tt[0]
        print("   start -> matrix words requested %.2f us, -> gradient test done %.2f us" % ((t[7] - t[0]) * 10e-3, (t[1] - t[7]) * 10e-3))
        print("reduced_solve %.1f us: gradient check %.2f, load %.2f, factor %.2f, back sweep %.2f us" % (
            1e3 * ms, *[(t[i + 1] - t[i]) * 10e-3 for i in range(4)]))
        print("factor phase: %d shader cycles in %.2f us -> %.0f MHz" % (t[6] - t[5], (t[3] - t[2]) * 10e-3,
                                                                       (t[6] - t[5]) / ((t[3] - t[2]) * 10e-3)))
        return
    print("total    mean %8.2f us  max %8.2f us; last end %.1f us" % (
        ((t[:, 7] - t[:, 0]) * 10e-3).mean(), ((t[:, 7] - t[:, 0]) * 10e-3).max(), (t[:, 7].max() - base) * 10e-3))

if __name__ == "__main__":
    main()
