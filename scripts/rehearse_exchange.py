"""Rehearsal of the sharded path's exchange on a one-GPU box: the benchmark's BAL-scale problem through the library's own
RCCL communicator with ONE rank (HIPBA_SHARD_SINGLE_RANK: every all-reduce is the identity, but every launch, stream
and event of the multi-GPU path is there).  Prints ms per LM iteration for the plain solve, for the exchange in one
piece and for the pipelined exchange: what the chunking itself costs, not what a link costs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stereo_visual_odometry_amd import hipba, scenes


def timed(dp, opts, comm=None):
    best = None
    for _ in range(3):
        dp.reset()
        t0 = time.perf_counter()
        s = dp.solve_sharded(comm.c, opts, log=True) if comm else dp.solve(opts, log=True)[0]
        dt = time.perf_counter() - t0
        n = s["num_iterations"] - 1
        best = (1e3 * dt / n, s) if best is None or 1e3 * dt / n < best[0] else best
    return best


def main():
    n_points = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    grid = (40, 50) if n_points >= 500_000 else (20, 25)
    prob, _, _ = scenes.bal_scale_bench(n_points=n_points, grid=grid)
    sv = hipba.Solver(0)
    opts = hipba.default_options(max_num_iterations=10, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
    dp = sv.upload(prob)
    ms, s_ref = timed(dp, opts)
    print("plain solve: %.2f ms per LM iteration, final cost %.9e" % (ms, s_ref["final_cost"]), flush=True)
    print("   costs:", " ".join("%.12e" % it["cost"] for it in s_ref["iterations"][:6]), flush=True)
    dp.free()
    comm = hipba.RcclComm(0, hipba.RcclComm.unique_id(), 0, 1)
    os.environ["HIPBA_SHARD_SINGLE_RANK"] = "1"
    dp = sv.upload_shard(prob, hipba.pose_order(prob))
    for chunks in ("1", "4", "8", "16"):
        os.environ["HIPBA_EXCHANGE_CHUNKS"] = chunks
        before = (comm.calls, comm.bytes)
        ms, s = timed(dp, opts, comm)
        print("one-rank sharded, exchange in %2s chunk(s): %.2f ms per LM iteration, final cost %.9e (rel. diff %.1e), "
              "%d all-reduces, %.0f MB" % (chunks, ms, s["final_cost"], abs(s["final_cost"] - s_ref["final_cost"]) / s_ref["final_cost"],
                                           (comm.calls - before[0]) // 3, (comm.bytes - before[1]) / 3e6), flush=True)
        print("   costs:", " ".join("%.12e" % it["cost"] for it in s["iterations"][:6]), flush=True)
    dp.free()
    comm.close()
    sv.close()


if __name__ == "__main__":
    main()
