"""Randomised sweep of the sharded solve on the GPU box: the random windows of fuzz_parity.py, each cut into 2-4
landmark-aligned shards that are solved by as many ranks (threads on one GPU, every all-reduce staged through host
memory) and compared with the same window solved whole.  Prints every case in which a rank ends differently.

Usage: python scripts/fuzz_sharded.py [n_cases] [first_seed]"""
import ctypes as C
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from stereo_visual_odometry_amd import hipba, scenes
from fuzz_parity import random_case, well_posed

hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]


def solve_in_shards(solvers, prob, opts, world, timeout=120):
    barrier = threading.Barrier(world)
    order = hipba.pose_order(prob)
    slots, results, errors = [None] * world, [None] * world, []

    def make_cb(rank):
        def cb(ctx, buf, count, op, stream):
            try:
                assert hip.hipStreamSynchronize(stream) == 0
                mine = np.empty(count, dtype=np.float64)
                assert hip.hipMemcpy(mine.ctypes.data, buf, 8 * count, 2) == 0
                slots[rank] = mine
                barrier.wait(timeout=timeout)
                tot = np.sum(slots, axis=0) if op == hipba.REDUCE_SUM else np.max(slots, axis=0)
                barrier.wait(timeout=timeout)
                assert hip.hipMemcpy(buf, tot.ctypes.data, 8 * count, 1) == 0
                return 0
            except Exception as e:
                errors.append(repr(e))
                return 1
        return hipba.ALLREDUCE_FN(cb)

    def run(rank):
        try:
            shard, (lo, hi) = scenes.shard_by_point(prob, rank, world)
            dp = solvers[rank].upload_shard(shard, order)
            cb = make_cb(rank)
            s = dp.solve_sharded(hipba.CComm(cb, None, rank, world), opts, log=True)
            poses, pts = dp.download()
            results[rank] = (s, poses, pts, lo, hi)
            dp.free()
        except Exception as e:
            errors.append("rank %d: %r" % (rank, e))
            barrier.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th: t.start()
    for t in th: t.join(timeout=timeout)
    return results, errors


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 70000
    solvers = [hipba.Solver(0) for _ in range(4)]
    ok = n_run = n_strict = 0
    t0 = time.time()
    for seed in range(first, first + n_cases):
        prob, okw, name = random_case(seed)
        if prob.obs_pose.shape[0] == 0 or prob.points.shape[0] < 4:
            continue
        world = int(np.random.default_rng(seed).integers(2, 5))
        opts = hipba.default_options(**okw)
        ref = prob.copy()
        s_ref = solvers[0].solve(ref, opts, log=True)
        results, errors = solve_in_shards(solvers, prob, opts, world)
        n_run += 1
        strict = well_posed(prob)
        n_strict += strict
        bad = list(errors)
        for rank, r in enumerate(results):
            if r is None:
                bad.append("rank %d returned nothing" % rank)
                continue
            s, poses, pts, lo, hi = r
            its, rits = s["iterations"], s_ref["iterations"]
            n_cmp = min(len(its), len(rits)) if strict else min(len(its), len(rits), 3)
            for a, b in list(zip(its, rits))[:n_cmp]:
                if a["step_is_successful"] != b["step_is_successful"] or abs(a["cost"] - b["cost"]) > 1e-8 * max(abs(b["cost"]), 1e-9) + 1e-13:
                    bad.append("rank %d iteration %d: cost %.12e vs %.12e" % (rank, a["iteration"], a["cost"], b["cost"]))
                    break
            if strict:
                if s["termination_type"] != s_ref["termination_type"] or s["num_iterations"] != s_ref["num_iterations"]:
                    bad.append("rank %d: termination %d after %d vs %d after %d" % (rank, s["termination_type"], s["num_iterations"],
                                                                                    s_ref["termination_type"], s_ref["num_iterations"]))
                if abs(s["final_cost"] - s_ref["final_cost"]) > 1e-8 * max(abs(s_ref["final_cost"]), 1e-12) + 1e-13:
                    bad.append("rank %d: final cost %.12e vs %.12e" % (rank, s["final_cost"], s_ref["final_cost"]))
                dq = np.abs(poses - ref.poses).max()
                dl = np.abs(pts - ref.points[lo:hi]).max() if hi > lo else 0.0
                if dq > 1e-6 or dl > 1e-5:
                    bad.append("rank %d: poses differ by %.2e, landmarks by %.2e" % (rank, dq, dl))
        if bad:
            print("seed %d [%s, %d ranks]%s %s" % (seed, name, world, " well-posed:" if strict else " loose:", "; ".join(bad[:4])), flush=True)
        ok += not bad
        if n_run % 20 == 0:
            print("... %d cases, %d agree, %.0f s" % (n_run, ok, time.time() - t0), flush=True)
    print("sharded against whole: %d of %d agree (%d well-posed, compared in full)" % (ok, n_run, n_strict), flush=True)
    for sv in solvers:
        sv.close()


if __name__ == "__main__":
    main()
