#!/usr/bin/env python
"""Benchmark of the MI355X bundle-adjustment engine.

    python bench.py --gpus N --steps K --warmup W

A step is ONE Levenberg-Marquardt iteration (linearise, eliminate landmarks, solve the reduced
camera system, back-substitute, evaluate the candidate, accept/reject) on the BAL-scale synthetic
problem of BASELINE.json (2000 cameras x 1M landmarks x 8M observations), resident in HBM before
the timed region starts.  With N > 1 (launched by torch.distributed.run, one rank per GPU) the same
problem is sharded by landmark and the reduced camera matrix / right-hand side are all-reduced over
RCCL every iteration: strong scaling, value = iterations of the whole job per second.

Rank 0 prints ONE JSON line carrying the metric, the roofline of the dominant kernel (durations
measured here with HIP events on the engine's stream), and a CPU baseline (the restated-Ceres
oracle on the host cores, on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E, spec
FP64_PEAK_TFLOPS = 78.6    # MI355X fp64 matrix / vector, spec


def algorithmic_bytes(kernel, n_c, n_p, n_o, nnz_s):
    """Un-fused algorithmic traffic per launch (fp64, every array touched once)."""
    if kernel == "residual":
        return 24 * n_o + 56 * n_c + 24 * n_p
    if kernel == "residual_jacobian":
        return 184 * n_o + 56 * n_c + 24 * n_p
    if kernel == "camera_blocks":
        return 24 * n_o + 56 * n_c + 24 * n_p + 336 * n_c
    if kernel == "point_schur":
        return 168 * n_o + 168 * n_p + 336 * n_c + 8 * nnz_s
    if kernel == "backsub":
        return 56 * n_o + 120 * n_p + 48 * n_c
    raise KeyError(kernel)


def count_reduced_nonzeros(prob, degree_hint=None):
    """36 x number of co-visible camera pairs (diagonal included, both triangles) of the reduced matrix."""
    from stereo_visual_odometry_amd import hipba  # noqa: F401
    n_c = prob.poses.shape[0]
    order = np.argsort(prob.obs_point, kind="stable")
    pt, cam = prob.obs_point[order], prob.obs_pose[order].astype(np.int64)
    start = np.flatnonzero(np.r_[True, pt[1:] != pt[:-1]])
    deg = np.diff(np.r_[start, pt.shape[0]])
    if deg.min() != deg.max():  # ragged degrees: report against the dense bound
        return 36 * n_c * n_c
    d = int(deg[0])
    c = cam.reshape(-1, d)
    seen = np.zeros((n_c, n_c), dtype=bool)
    for i in range(d):
        for j in range(d):
            seen[c[:, i], c[:, j]] = True
    return 36 * int(seen.sum())


def cpu_baseline(threads):
    """The CPU oracle (restated Ceres path: numeric central-difference Jacobians, exact solve) on a
    1/8-scale sample of the workload."""
    from oracle import oracle
    from stereo_visual_odometry_amd import scenes
    prob, _, _ = scenes.bal_scale(seed=3000, grid=(10, 25), n_points=125_000)
    o = oracle.Problem(prob.poses, prob.points, prob.obs_pose, prob.obs_point, prob.obs_uv, prob.K,
                       prob.camera_model, prob.pose_const, prob.point_const)
    iters = 3
    opts = oracle.default_options(max_num_iterations=iters, num_threads=threads, function_tolerance=0.0,
                                  gradient_tolerance=0.0, parameter_tolerance=0.0)
    t0 = time.time()
    s, its = oracle.solve(o, opts)
    dt = time.time() - t0
    done = max(len(its) - 1, 1)
    return {
        "value": done / dt, "unit": "LM iterations/s", "cores": threads, "kind": "port",
        "sample": "restated-Ceres oracle (numeric central-difference Jacobians, exact Schur+Cholesky solve), "
                  "%d LM iterations on a 1/8-scale problem: 250 cameras x 125k landmarks x 1M observations "
                  "(evaluate %.2fs, linear solve %.2fs); the full-size problem has 8x the observations and a "
                  "512x larger factorisation" % (done, s["time_evaluate_s"], s["time_linear_solve_s"]),
    }


def secondary_measurements(solver):
    """ms per KITTI-like keyframe window (end to end through hipba_solve, upload included) and batched
    small-window throughput -- the other two single-GPU configurations of BASELINE.json."""
    from stereo_visual_odometry_amd import hipba, scenes
    out = {}
    wins = [scenes.kitti_like_window(1000 + i)[0] for i in range(8)]
    solver.solve(wins[0].copy(), hipba.reference_production_options(), log=False)  # warm-up
    t, iters = [], []
    for w in wins:
        p = w.copy()
        t0 = time.perf_counter()
        s = solver.solve(p, hipba.reference_production_options(), log=False)
        t.append(time.perf_counter() - t0)
        iters.append(s["num_iterations"] - 1)
    out["kitti_like_window"] = {
        "ms_per_window": 1e3 * float(np.mean(t)), "lm_iterations_per_window": float(np.mean(iters)),
        "workload": "30 keyframes, ~4000 landmarks, ~25k observations, production semantics "
                    "(first frame constant, 300 iteration budget), host upload included"}
    n_w = 1024
    small = [scenes.small_window(2000 + i)[0] for i in range(n_w)]
    dp = solver.upload(small)
    opts = hipba.reference_production_options()
    dp.solve(opts)
    dp.reset()
    t0 = time.perf_counter()
    res = dp.solve(opts)
    dt = time.perf_counter() - t0
    tot_it = sum(r["num_iterations"] - 1 for r in res)
    out["batched_small_windows"] = {
        "windows": n_w, "windows_per_s": n_w / dt, "lm_iterations_per_s": tot_it / dt, "ms_total": 1e3 * dt,
        "workload": "1024 windows x (10 keyframes, 2000 landmarks, ~14k observations), resident in HBM"}
    dp.free()
    out["concurrent_sequences"] = concurrent_sequences(solver.device)
    return out


def concurrent_sequences(device, n_seq=11, windows_per_seq=8):
    """Eleven sequences (KITTI 00-10) sharing ONE GPU: a host thread and an engine handle per sequence, every thread
    solving its keyframe windows back to back (the reference solves one window per 20 frames, so this is the worst
    case of all sequences asking at once).  Latency of one optimize() call as its caller sees it."""
    import threading
    from stereo_visual_odometry_amd import hipba, scenes
    solvers = [hipba.Solver(device) for _ in range(n_seq)]
    wins = [[scenes.kitti_like_window(1000 + 16 * q + i)[0] for i in range(windows_per_seq)] for q in range(n_seq)]
    opts = hipba.reference_production_options()
    for q in range(n_seq):
        solvers[q].solve(wins[q][0].copy(), opts, log=False)  # warm-up: arenas grown, code loaded
    lat = [[] for _ in range(n_seq)]
    start = threading.Barrier(n_seq + 1)

    def run(q):
        start.wait()
        for w in wins[q]:
            t0 = time.perf_counter()
            solvers[q].solve(w, opts, log=False)
            lat[q].append(time.perf_counter() - t0)

    threads = [threading.Thread(target=run, args=(q,)) for q in range(n_seq)]
    for th in threads:
        th.start()
    start.wait()
    t0 = time.perf_counter()
    for th in threads:
        th.join()
    wall = time.perf_counter() - t0
    for sv in solvers:
        sv.close()
    all_lat = np.sort(np.concatenate([np.asarray(l) for l in lat]))
    return {"sequences": n_seq, "windows": int(all_lat.size), "windows_per_s": all_lat.size / wall,
            "latency_ms": {"p50": 1e3 * float(np.percentile(all_lat, 50)), "p95": 1e3 * float(np.percentile(all_lat, 95)),
                           "max": 1e3 * float(all_lat[-1])},
            "workload": "%d host threads, one engine handle each, on one GPU; %d KITTI-like 30-keyframe windows per "
                        "thread solved back to back, host upload included" % (n_seq, windows_per_seq)}


def replica_windows(solver, rank, world, fence, n_w=1024):
    """Small windows over N GPUs: independent units, a static partition with no exchange on the data path.
    Every rank solves its own n_w windows (weak scaling); returns the windows and LM iterations this rank did and
    its wall time between two fences."""
    from stereo_visual_odometry_amd import hipba, scenes
    small = [scenes.small_window(2000 + rank * n_w + i)[0] for i in range(n_w)]
    dp = solver.upload(small)
    opts = hipba.reference_production_options()
    dp.solve(opts)
    dp.reset()
    fence()
    t0 = time.perf_counter()
    res = dp.solve(opts)
    fence()
    dt = time.perf_counter() - t0
    dp.free()
    return n_w, sum(r["num_iterations"] - 1 for r in res), dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--points", type=int, default=1_000_000, help="landmarks of the BAL-scale problem")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--linear-solver", default="cholesky", choices=["cholesky", "pcg"])
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from stereo_visual_odometry_amd import hipba, scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    # rehearsal on a one-GPU box: every rank on device 0, collectives through gloo (everything but RCCL itself)
    rehearsal = os.environ.get("HIPBA_BENCH_ONE_DEVICE") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    red_dev = "cpu" if rehearsal else "cuda"  # where the bench's own small reductions live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    solver = hipba.Solver(local_rank)
    prob, _, _ = scenes.bal_scale(seed=3000, n_points=args.points)
    n_c, n_p, n_o = prob.poses.shape[0], prob.points.shape[0], prob.obs_pose.shape[0]
    lin = hipba.LINEAR_DENSE_CHOLESKY if args.linear_solver == "cholesky" else hipba.LINEAR_PCG

    def lm_options(iters):
        return hipba.default_options(max_num_iterations=iters, function_tolerance=0.0, gradient_tolerance=0.0,
                                     parameter_tolerance=0.0, linear_solver=lin, pcg_max_iterations=2000,
                                     pcg_tolerance=1e-10)

    if world > 1:
        from stereo_visual_odometry_amd.sharded import ShardedProblem, TorchComm
        comm = TorchComm(torch.device("cuda", local_rank))
        job = ShardedProblem(solver, prob, comm)
        run = lambda iters: job.solve(lm_options(iters))
        reset = job.reset
        dp = job.dp
    else:
        dp = solver.upload(prob)
        run = lambda iters: dp.solve(lm_options(iters))[0]
        reset = dp.reset

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:
        run(args.warmup)
        reset()
    fence()
    t0 = time.perf_counter()
    summary = run(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    # every trust-region step computed counts, also the one a convergence test ends on (which Ceres, and
    # therefore the iteration log, does not list): each is one full pass of the hot path
    steps_done = summary["num_linear_solves"]

    result = None
    if rank == 0:
        nnz = count_reduced_nonzeros(prob)
        # per-kernel durations with HIP events on the engine's stream (single-GPU shard of rank 0)
        kernels = []
        flops_chol = (6.0 * (n_c - 1)) ** 3 / 3.0
        shard_o = dp.problems[0].obs_pose.shape[0]
        shard_p = dp.problems[0].points.shape[0]
        for kid, name in enumerate(hipba.KERNEL_NAMES):
            ms = dp.time_kernel(kid, 5 if name != "reduced_solve" else 2, lm_options(1))
            if name == "reduced_solve":
                if lin == hipba.LINEAR_DENSE_CHOLESKY:
                    ach = flops_chol / (ms * 1e-3) / 1e12
                    kernels.append({"kernel": name, "bound": "mfma", "ms": ms, "achieved": ach, "peak": FP64_PEAK_TFLOPS,
                                    "unit": "TFLOP/s", "frac": ach / FP64_PEAK_TFLOPS, "traffic": None})
                else:
                    kernels.append({"kernel": name, "bound": "hbm", "ms": ms, "achieved": None, "peak": HBM_PEAK_GBS,
                                    "unit": "GB/s", "frac": None, "traffic": None})
            else:
                b = algorithmic_bytes(name, n_c, shard_p, shard_o, nnz)
                ach = b / (ms * 1e-3) / 1e9
                kernels.append({"kernel": name, "bound": "hbm", "ms": ms, "achieved": ach, "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None})
        traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(traffic_file):
            tr = json.load(open(traffic_file))
            for k in kernels:
                # HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE doubled, WRITE_SIZE; KiB units),
                # collected on this workload by scripts/profile_target.py + scripts/summarize_pmc.py
                if k["kernel"] in tr and (n_p, world) == (1_000_000, 1):
                    k["traffic"] = tr[k["kernel"]]["hbm_bytes"]
        in_loop = [k for k in kernels if k["kernel"] != "residual_jacobian"]
        dom = max(in_loop, key=lambda k: k["ms"])
        result = {
            "metric": "BA LM iterations/sec", "value": steps_done / elapsed, "unit": "LM iterations/s",
            "n_gpus": world, "steps": steps_done, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / max(steps_done, 1),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BAL-scale synthetic BA: %d cameras x %d landmarks x %d observations, "
                                   "landmark-aligned observation shards over %d GPU(s), reduced camera system by %s"
                                   % (n_c, n_p, n_o, world, "fp64 matrix-core Cholesky" if lin == 1 else "block-Jacobi PCG"),
                       "cameras": n_c, "landmarks": n_p, "observations": n_o, "linear_solver": args.linear_solver,
                       "final_cost": summary["final_cost"], "initial_cost": summary["initial_cost"]},
            "roofline": {k: dom[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")},
            "roofline_kernel": dom["kernel"],
            "kernels": kernels,
        }
        if world > 1:
            result["allreduce_calls_per_step"] = comm.calls / max(steps_done + args.warmup, 1)
            result["allreduce_bytes_per_step"] = comm.bytes / max(steps_done + args.warmup, 1)
    if world > 1:
        dist.barrier()
        if not args.no_secondary:
            # the other shape of multi-GPU work on this path: replicas over independent small windows
            dp.free()
            n_w, n_it, dt = replica_windows(solver, rank, world, fence)
            acc = torch.tensor([n_w, n_it], dtype=torch.float64, device=red_dev)
            tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
            dist.all_reduce(acc)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            if rank == 0:
                dt = float(tmax.item())
                result["secondary"] = {"batched_small_windows_replicas": {
                    "windows": int(acc[0].item()), "windows_per_s": float(acc[0].item()) / dt,
                    "lm_iterations_per_s": float(acc[1].item()) / dt, "ms_total": 1e3 * dt, "scaling": "weak",
                    "workload": "%d GPUs x 1024 independent windows (10 keyframes, 2000 landmarks, ~14k observations), "
                                "resident in HBM, no exchange on the data path" % world}}
    if rank == 0:
        if world == 1 and not args.no_secondary:
            dp.free()
            result["secondary"] = secondary_measurements(solver)
        if world == 1 and not args.no_cpu_baseline:
            threads = min(16, os.cpu_count() or 1)
            result["cpu_baseline"] = cpu_baseline(threads)
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
