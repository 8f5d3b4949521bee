#!/usr/bin/env python
"""Benchmark of the MI355X bundle-adjustment engine.

    python bench.py --gpus N --steps K --warmup W

A step is ONE Levenberg-Marquardt iteration (linearise, eliminate landmarks, solve the reduced
camera system, back-substitute, evaluate the candidate, accept/reject) on the BAL-scale synthetic
problem of BASELINE.json (2000 cameras x 1M landmarks x 8M observations), resident in HBM before
the timed region starts, solved with Ceres' default tolerances from a poor starting point so that the
K timed iterations are a real descent (accepted and rejected steps are reported).  With N > 1
(launched by torch.distributed.run, one rank per GPU) the same problem is sharded by landmark and
the reduced camera matrix / right-hand side are all-reduced over RCCL every iteration: strong
scaling, value = iterations of the whole job per second.

Rank 0 prints ONE JSON line carrying the metric, the roofline of the dominant kernel (durations
measured here with HIP events on the engine's stream), and a CPU baseline (the restated-Ceres
oracle on the host cores, on a bounded sample); under "secondary" the other single-GPU
configurations of BASELINE.json (ms per KITTI-like keyframe window, batched small windows), each
with its own roofline and its own CPU baseline on the same inputs.
"""
import argparse
import glob
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E, spec
FP64_PEAK_TFLOPS = 78.6    # MI355X fp64 matrix / vector, spec
MFMA_F64_16X16X4_TFLOPS = 48.5  # measured on MI355X: the most v_mfma_f64_16x16x4_f64 delivers chip-wide (register operands)
TERMINATION = ["CONVERGENCE", "NO_CONVERGENCE", "FAILURE"]


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
        # the pass also evaluates the cost at the candidate point: the back-substitution's and the cost pass's figures
        return 56 * n_o + 120 * n_p + 48 * n_c + 24 * n_o + 56 * n_c + 24 * n_p
    raise KeyError(kernel)


def count_reduced_nonzeros(prob):
    """36 x number of co-visible camera pairs (diagonal included, both triangles) of the reduced matrix."""
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


def csrc_fingerprint():
    """sha256 over the kernel sources: profiles/traffic.json carries the fingerprint of the sources its counters
    were collected on, and is ignored when the kernels have changed since."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "stereo_visual_odometry_amd", "csrc", "*.h*"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def load_traffic(workload_key):
    """HBM bytes per launch from rocprofv3 PMC passes (scripts/profile_target.py + scripts/summarize_pmc.py), or {}
    if the file is absent, was measured on another workload or on other kernel sources."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return {}
    tr = json.load(open(path))
    if tr.get("csrc_sha256") != csrc_fingerprint() or tr.get("workload") != workload_key:
        return {}
    return tr.get("kernels", {})


SHADER_CLOCK_GHZ = 2.4   # MI355X peak engine clock
SIMDS = 1024             # 256 compute units x 4
# bench.py's kernel names -> the kernels (as rocprofv3 names them) one launch is made of
ISSUE_MEMBERS = {"residual": ["k_cost"], "residual_jacobian": ["k_residual_jacobian"], "camera_blocks": ["k_camera_blocks"],
                 "point_schur": ["k_point_schur<2>", "k_schur_gather", "k_schur_gather_lanes"], "backsub": ["k_backsub<true>"]}


def load_issue(workload_key):
    """Vector instructions per launch from a rocprofv3 PMC pass of SQ counters (scripts/profile_all.sh +
    scripts/summarize_sq.py), or {} if the file is absent or belongs to another workload or other kernel sources."""
    path = os.path.join(ROOT, "profiles", "issue.json")
    if not os.path.exists(path):
        return {}
    doc = json.load(open(path))
    if doc.get("csrc_sha256") != csrc_fingerprint() or doc.get("workload") != workload_key:
        return {}
    return doc.get("kernels", {})


def add_vector_issue_bound(rows, issue):
    """A second bound for the kernels that recompute their Jacobians in fp64: the time the vector units need just to
    ISSUE the launch's vector instructions (SQ_INSTS_VALU per wave x waves, four cycles per wave64 instruction -- the
    full-rate cost; fp64 divisions and square roots take more -- spread over all SIMDs at the peak clock).  A kernel
    can be no faster than that however little memory it touches; `frac` = that floor / the measured duration."""
    for k in rows:
        members = ISSUE_MEMBERS.get(k["kernel"])
        if not members or not all(m in issue for m in members):
            continue
        insts = sum(issue[m]["valu_insts_per_wave"] * issue[m]["waves_per_call"] for m in members)
        floor_ms = 1e3 * insts * 4.0 / (SIMDS * SHADER_CLOCK_GHZ * 1e9)
        k["vector_issue"] = {"bound": "valu", "wave_instructions": insts, "floor_ms": floor_ms, "frac": floor_ms / k["ms"],
                             "issuing_frac_of_wave_cycles": max(issue[m]["issuing_frac"] for m in members)}


def kernel_table(dp, hipba, opts, n_c, n_p, n_o, nnz, flops_solve, single_window):
    """Average launch duration of every hot-path kernel on the resident problem (HIP events on the engine's
    stream) against its algorithmic bytes / flops."""
    rows = []
    for kid, name in enumerate(hipba.KERNEL_NAMES):
        if name == "residual_jacobian" and not single_window:
            continue  # the materialising twin is not part of the loop and is timed on one-window problems only
        ms = dp.time_kernel(kid, 5 if name != "reduced_solve" else 2, opts)
        if name == "reduced_solve":
            # algorithmic flops: what a tile-level factorisation has to do inside the envelope of the reduced matrix
            # under the engine's pose order -- the multiply-adds the kernels execute, counted per 128x128 tile product
            # (for a batch of small windows: the sum of their dense factorisations).  The dense figure, (6 n_c)^3 / 3,
            # which the reference's sparse Cholesky would not pay either, is listed beside it as what the same time
            # would mean for a solver without the envelope.
            dense, env = dp.reduced_solve_flops()
            flops = env if flops_solve is None else flops_solve
            dense = dense if flops_solve is None else flops_solve
            ach = flops / (ms * 1e-3) / 1e12
            rows.append({"kernel": name, "bound": "mfma", "ms": ms, "achieved": ach, "peak": FP64_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": ach / FP64_PEAK_TFLOPS, "traffic": None,
                         "algorithmic_flops": flops, "dense_flops": dense,
                         "dense_equivalent_tflops": dense / (ms * 1e-3) / 1e12,
                         # what the instruction the factorisation is built on sustains over the whole chip from
                         # registers alone, at any occupancy (scripts/bench_fp64_rates.hip): the nominal peak is only
                         # reached by v_mfma_f64_4x4x4_4b_f64 (75 TFLOP/s), which lost inside the band update
                         "instruction_ceiling": {"instruction": "v_mfma_f64_16x16x4_f64", "tflops": MFMA_F64_16X16X4_TFLOPS,
                                                 "frac": ach / MFMA_F64_16X16X4_TFLOPS}})
        else:
            ach = algorithmic_bytes(name, n_c, n_p, n_o, nnz) / (ms * 1e-3) / 1e9
            rows.append({"kernel": name, "bound": "hbm", "ms": ms, "achieved": ach, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None})
    return rows


def roofline_of(rows):
    in_loop = [k for k in rows if k["kernel"] != "residual_jacobian"]
    dom = max(in_loop, key=lambda k: k["ms"])
    out = {k: dom[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
    for k in ("algorithmic_flops", "dense_equivalent_tflops", "instruction_ceiling"):
        if k in dom:
            out[k] = dom[k]
    out["kernel"] = dom["kernel"]
    return out


def oracle_problem(oracle, p):
    return oracle.Problem(p.poses, p.points, p.obs_pose, p.obs_point, p.obs_uv, p.K, p.camera_model, p.pose_const,
                          p.point_const)


def cpu_baseline(threads, solver=None, full_prob=None, engine_log=None, budget_s=120.0):
    """The CPU oracle (restated Ceres path: numeric central-difference Jacobians as the reference differentiates, exact
    solve) on the host cores.  First on a 1/4-scale sample of the workload (same scene generator, same starting noise),
    with the engine on that same sample beside it; then, if the quarter-scale timing says it fits the budget, on the
    quoted configuration itself for two LM iterations -- that run is the headline baseline, and its first costs are
    set beside the engine's log of the same steps."""
    from oracle import oracle
    from stereo_visual_odometry_amd import hipba, scenes
    prob, _, _ = scenes.bal_scale_bench(grid=(20, 25), n_points=250_000)
    o = oracle_problem(oracle, prob)
    iters = 4
    opts = oracle.default_options(max_num_iterations=iters, num_threads=threads)
    t0 = time.time()
    s, its = oracle.solve(o, opts)
    dt = time.time() - t0
    done = max(len(its) - 1, 1)
    quarter = {
        "value": done / dt, "unit": "LM iterations/s", "cores": threads, "kind": "port",
        "sample": "restated-Ceres oracle (numeric central-difference Jacobians, exact Schur+Cholesky solve), "
                  "%d LM iterations on a 1/4-scale problem from the same generator and start: 500 cameras x 250k landmarks "
                  "x 2M observations (evaluate %.2fs, linear solve %.2fs)" % (done, s["time_evaluate_s"], s["time_linear_solve_s"]),
    }
    if solver is not None:
        dp = solver.upload(prob)
        go = hipba.default_options(max_num_iterations=iters)
        dp.solve(go)  # warm-up
        dp.reset()
        gs = dp.solve(go)[0]
        dp.free()
        quarter["engine_on_same_sample"] = {"value": max(gs["num_linear_solves"], 1) / gs["time_solve_s"],
                                            "unit": "LM iterations/s", "lm_iterations": gs["num_linear_solves"]}
    if full_prob is None:
        return quarter
    # the quoted configuration: 4x the observations (evaluation, elimination) and a 64x larger dense factorisation
    full_iters = 2
    estimate = full_iters * (4.0 * s["time_evaluate_s"] + 64.0 * s["time_linear_solve_s"]) / done
    if estimate > budget_s:
        quarter["full_size_skipped"] = ("estimated %.0f s for %d LM iterations of the oracle on the full problem "
                                        "(budget %.0f s)" % (estimate, full_iters, budget_s))
        return quarter
    of = oracle_problem(oracle, full_prob)
    t0 = time.time()
    sf, fits = oracle.solve(of, oracle.default_options(max_num_iterations=full_iters, num_threads=threads))
    dtf = time.time() - t0
    donef = max(len(fits) - 1, 1)
    n_c, n_p, n_o = full_prob.poses.shape[0], full_prob.points.shape[0], full_prob.obs_pose.shape[0]
    out = {
        "value": donef / dtf, "unit": "LM iterations/s", "cores": threads, "kind": "port",
        "sample": "restated-Ceres oracle (numeric central-difference Jacobians, exact Schur+Cholesky solve), %d LM "
                  "iterations on the quoted configuration itself, %d cameras x %d landmarks x %d observations, same start "
                  "(evaluate %.1fs, linear solve %.1fs)" % (donef, n_c, n_p, n_o, sf["time_evaluate_s"], sf["time_linear_solve_s"]),
        "costs": [it["cost"] for it in fits],
        "quarter_scale": quarter,
    }
    if engine_log:
        # the engine's closed-form Jacobians against the reference's numeric differentiation, step by step
        eng = [it["cost"] for it in engine_log[:len(fits)]]
        out["engine_costs_same_steps"] = eng
        out["max_relative_cost_difference"] = max(abs(a - b) / abs(b) for a, b in zip(eng, out["costs"]))
    return out


def secondary_measurements(solver, with_cpu, threads):
    """ms per KITTI-like keyframe window (end to end through hipba_solve, upload included) and batched
    small-window throughput -- the other two single-GPU configurations of BASELINE.json -- each with the roofline
    of its dominant kernel and the CPU oracle timed on the same inputs."""
    from stereo_visual_odometry_amd import hipba, scenes
    out = {}
    opts = hipba.reference_production_options()
    wins = [scenes.kitti_like_window(1000 + i)[0] for i in range(8)]
    # one untimed pass over the windows first: the handle's device arena and pinned staging buffer grow to the largest
    # window they have seen and stay there, which is the state a sequence is in after its first few windows
    for w in wins:
        solver.solve(w.copy(), opts, log=False)
    t, iters = [], []
    for rep in range(3):
        for w in wins:
            p = w.copy()
            t0 = time.perf_counter()
            s = solver.solve(p, opts, log=False)
            t.append(time.perf_counter() - t0)
            iters.append(s["num_iterations"] - 1)
    kw = {"ms_per_window": 1e3 * float(np.mean(t)), "ms_per_window_median": 1e3 * float(np.median(t)),
          "lm_iterations_per_window": float(np.mean(iters)),
          "lm_iterations_per_s": float(np.sum(iters) / np.sum(t)),
          "workload": "30 keyframes, ~4000 landmarks, ~25k observations, production semantics "
                      "(first frame constant, 300 iteration budget), host upload included; mean of 8 windows x 3 passes "
                      "after one untimed pass (buffers at their high-water mark)"}
    # per-kernel durations on window 0, resident
    w0 = wins[0]
    dp = solver.upload(w0)
    n_c, n_p, n_o = w0.poses.shape[0], w0.points.shape[0], w0.obs_pose.shape[0]
    rows = kernel_table(dp, hipba, opts, n_c, n_p, n_o, 36 * n_c * n_c, None, True)
    dp.free()
    kw["roofline"] = roofline_of(rows)
    kw["kernels"] = rows
    if with_cpu:
        from oracle import oracle
        tc, itc = [], []
        for w in wins:
            o = oracle_problem(oracle, w)
            t0 = time.perf_counter()
            so, its = oracle.solve(o, oracle.default_options(max_num_iterations=300, num_threads=threads))
            tc.append(time.perf_counter() - t0)
            itc.append(len(its) - 1)
        kw["cpu_baseline"] = {"value": 1e3 * float(np.mean(tc)), "unit": "ms per window", "cores": threads, "kind": "port",
                              "lm_iterations_per_window": float(np.mean(itc)),
                              "sample": "restated-Ceres oracle (numeric central-difference Jacobians, full normal "
                                        "equations solved exactly) on the same 8 windows"}
    out["kitti_like_window"] = kw

    n_w = 1024
    small = [scenes.small_window(2000 + i)[0] for i in range(n_w)]
    dp = solver.upload(small)
    dp.solve(opts)
    dp.reset()
    t0 = time.perf_counter()
    res = dp.solve(opts)
    dt = time.perf_counter() - t0
    tot_it = sum(r["num_iterations"] - 1 for r in res)
    bw = {"windows": n_w, "windows_per_s": n_w / dt, "lm_iterations_per_s": tot_it / dt, "ms_total": 1e3 * dt,
          "workload": "1024 windows x (10 keyframes, 2000 landmarks, ~14k observations), resident in HBM"}
    dp.reset()
    n_c = sum(p.poses.shape[0] for p in small)
    n_p = sum(p.points.shape[0] for p in small)
    n_o = sum(p.obs_pose.shape[0] for p in small)
    nnz = sum(36 * p.poses.shape[0] ** 2 for p in small)
    flops = sum((6.0 * (p.poses.shape[0] - 1)) ** 3 / 3.0 for p in small)
    rows = kernel_table(dp, hipba, opts, n_c, n_p, n_o, nnz, flops, False)
    dp.free()
    bw["roofline"] = roofline_of(rows)
    bw["kernels"] = rows
    if with_cpu:
        from oracle import oracle
        n_cpu = 16
        t0 = time.perf_counter()
        itc = 0
        for w in small[:n_cpu]:
            o = oracle_problem(oracle, w)
            so, its = oracle.solve(o, oracle.default_options(max_num_iterations=300, num_threads=threads))
            itc += len(its) - 1
        dtc = time.perf_counter() - t0
        bw["cpu_baseline"] = {"value": n_cpu / dtc, "unit": "windows/s", "cores": threads, "kind": "port",
                              "lm_iterations_per_s": itc / dtc,
                              "sample": "restated-Ceres oracle (numeric central-difference Jacobians, exact solve) on the "
                                        "first %d of the same 1024 windows, one after another" % n_cpu}
    out["batched_small_windows"] = bw
    out["concurrent_sequences"] = concurrent_sequences(solver.device)
    out["frontend_steps"] = frontend_steps(solver, with_cpu, threads)
    out["vo_pipeline"] = vo_pipeline()
    return out


def vo_pipeline():
    """The reference's per-frame pipeline (src/frontend.cpp:22-86) around the three GPU steps on a synthetic image half
    (lib/vo_pipeline, csrc/host/synthetic_frontend.hpp): what triangulation, pose estimation and a bundle-adjustment
    window take per call inside the pipeline.  A separate program with its own handle; its absence or failure leaves a
    note instead of numbers."""
    import subprocess
    exe = os.path.join(ROOT, "stereo_visual_odometry_amd", "lib", "vo_pipeline")
    try:
        r = subprocess.run([exe, "--frames", "200"], capture_output=True, text=True, timeout=120)
        line = [l.split() for l in r.stdout.splitlines() if l.startswith("bundle_adjustment")][0]
        get = lambda k: float(line[line.index(k) + 1])
        return {"frames": 200, "features_per_frame": get("mean_features"), "bundle_adjustment_windows": int(get("windows")),
                "triangulate_ms_per_frame": get("triangulate_ms_per_frame"), "estimate_ms_per_frame": get("estimate_ms_per_frame"),
                "ba_ms_per_window": get("ba_ms_per_window"), "ate_rmse_m": get("ate_rmse_m"), "wall_s": get("wall_s"),
                "workload": "200 frames at 0.8 m, ~1000 tracked features, 3 % wrong and 4 % lost tracks per frame, bundle "
                            "adjustment of the last 30 frames every 20 (frontend.cpp:71); image half synthetic"}
    except Exception as e:  # noqa: BLE001 -- a secondary figure must not cost the bench line
        return {"note": "lib/vo_pipeline did not run: %s" % e}


def frontend_steps(solver, with_cpu, threads):
    """The data-parallel steps either side of the bundle adjustment in the per-frame pipeline: stereo triangulation
    (matches resident in HBM, one kernel) and the pose refinement that ends PnP (host call, upload included)."""
    import torch
    from stereo_visual_odometry_amd import hipba, scenes
    out = {}
    n = 1 << 20
    left, right, _, pose = scenes.stereo_matches(4000, n)
    dev = torch.device("cuda", solver.device)
    d = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in
         (scenes.P_LEFT.reshape(12), scenes.P_RIGHT.reshape(12), left, right, pose)]
    d_out = torch.empty((n, 3), dtype=torch.float64, device=dev)
    stream = torch.cuda.ExternalStream(int(solver.stream), device=dev)
    torch.cuda.synchronize(dev)
    call = lambda: solver.triangulate_stereo_device(d[0].data_ptr(), d[1].data_ptr(), n, d[2].data_ptr(), d[3].data_ptr(),
                                                    d[4].data_ptr(), d_out.data_ptr())
    call()
    stream.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    stream.synchronize()
    dt = (time.perf_counter() - t0) / reps
    algorithmic = 40.0 * n  # two float pixel pairs in, three doubles out
    tri = {"matches": n, "matches_per_s": n / dt, "ms": 1e3 * dt,
           "workload": "2^20 stereo matches of a KITTI-like rectified pair, resident in HBM, lifted to the world frame",
           "roofline": {"bound": "hbm", "achieved": algorithmic / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": algorithmic / dt / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "note": "40 B per match; the kernel is bound by the 4x4 Jacobi SVD's fp64 arithmetic, not by HBM"}}
    if with_cpu:
        from oracle import oracle
        oracle.triangulate_stereo(scenes.P_LEFT, scenes.P_RIGHT, left[:1000], right[:1000], pose)
        t0 = time.perf_counter()
        oracle.triangulate_stereo(scenes.P_LEFT, scenes.P_RIGHT, left, right, pose)
        dtc = time.perf_counter() - t0
        tri["cpu_baseline"] = {"value": n / dtc, "unit": "matches/s", "cores": threads, "kind": "port",
                               "sample": "the oracle's restatement of cv::triangulatePoints on the same matches (OpenMP)"}
    out["stereo_triangulation"] = tri
    pts, uv, start, _, mask = scenes.pnp_case(4001, n=800, outlier_fraction=0.2)
    opts = hipba.default_options(max_num_iterations=50)
    solver.refine_pose(scenes.KITTI_K, pts, uv, start, use=mask, options=opts)
    t, its = [], []
    for k in range(8):
        pts, uv, start, _, mask = scenes.pnp_case(4002 + k, n=800, outlier_fraction=0.2)
        t0 = time.perf_counter()
        _, s = solver.refine_pose(scenes.KITTI_K, pts, uv, start, use=mask, options=opts)
        t.append(time.perf_counter() - t0)
        its.append(s["num_iterations"] - 1)
    out["pose_refinement"] = {"ms_per_call": 1e3 * float(np.mean(t)), "lm_iterations_per_call": float(np.mean(its)),
                              "workload": "800 3D-2D matches, 640 in use, host arrays in, refined pose out; mean of 8"}
    t, found, its = [], [], []
    for k in range(8):
        pts, uv, start, _, mask = scenes.pnp_case(4010 + k, n=800, outlier_fraction=0.3, sigma_t=0.3, sigma_w=0.02)
        t0 = time.perf_counter()
        _, got, run = solver.pnp_ransac(scenes.KITTI_K, pts, uv, None, iterations=1000, reprojection_error=3.0, seed=k,
                                        confidence=0.99, return_iterations=True)
        t.append(time.perf_counter() - t0)
        found.append(float((got == mask).mean()))
        its.append(run)
    out["pnp_ransac"] = {"ms_per_call": 1e3 * float(np.mean(t)), "inlier_mask_agreement": float(np.mean(found)),
                         "sequential_iterations_by_the_confidence_bound": float(np.mean(its)),
                         "workload": "800 3D-2D matches with 30 % gross outliers, no prior pose (minimal samples in closed "
                                     "form), 1000 hypotheses, 3 px, confidence 0.99 as estimator.cpp:28-29 asks, refinement "
                                     "on the consensus set included, host arrays in and out; mean of 8"}
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


def launch_ranks(n):
    """`python bench.py --gpus N` started plainly: run the same command line as N ranks of one node under
    torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1 at a free port) as a CHILD process.
    Called before anything in this process has touched the GPU; returns the child's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if env.get("HIPBA_BENCH_ONE_DEVICE") == "1":
        env.setdefault("HIPBA_COMM_HOST_SYNC", "1")  # every rank on device 0 through gloo: see TorchComm
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--points", type=int, default=1_000_000, help="landmarks of the BAL-scale problem")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--linear-solver", default="cholesky", choices=["cholesky", "pcg"])
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started plainly with --gpus N: launch the N ranks as a child job (one process per GPU) and pass its
        # output and exit code on.  This process has not touched the GPU (torch is not even imported yet) and
        # never will: it only waits for the child.
        sys.exit(launch_ranks(args.gpus))

    # A rank that hangs (a collective that never completes) must not hold the job for ever: after 25 minutes the
    # process says so and leaves.  (A default run takes well under two.)
    import threading

    def give_up():
        print("bench.py: no result after 25 minutes, giving up", file=sys.stderr, flush=True)
        os._exit(3)

    watchdog = threading.Timer(1500.0, give_up)
    watchdog.daemon = True
    watchdog.start()

    import torch
    import torch.distributed as dist
    from stereo_visual_odometry_amd import hipba, scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d: running with %d ranks" % (args.gpus, world, world), file=sys.stderr)
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
    prob, _, _ = scenes.bal_scale_bench(seed=3000, n_points=args.points)
    n_c, n_p, n_o = prob.poses.shape[0], prob.points.shape[0], prob.obs_pose.shape[0]
    lin = hipba.LINEAR_DENSE_CHOLESKY if args.linear_solver == "cholesky" else hipba.LINEAR_PCG

    def lm_options(iters):
        # Ceres' defaults everywhere (function 1e-6, gradient 1e-10, parameter 1e-8): the iteration budget alone
        # is what the reference sets (optimizer.cpp:58)
        return hipba.default_options(max_num_iterations=iters, linear_solver=lin, pcg_max_iterations=2000,
                                     pcg_tolerance=1e-10)

    native_comm = True
    if world > 1:
        from stereo_visual_odometry_amd.sharded import ShardedProblem, TorchComm
        if rehearsal:
            comm = TorchComm(torch.device("cuda", local_rank))   # gloo: everything but RCCL itself
        else:
            # the library's own RCCL communicator: every all-reduce of the solve is an ncclAllReduce issued by the
            # engine on its stream, no Python in the loop; torch.distributed only carries the 128-byte id
            ids = [hipba.RcclComm.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            try:
                comm = hipba.RcclComm(local_rank, ids[0], rank, world)
                made = 1
            except Exception as e:  # (no RCCL to load, or its initialisation refused)
                print("bench.py: rank %d: native RCCL communicator failed (%r)" % (rank, e), file=sys.stderr)
                comm, made = None, 0
            agreed = torch.tensor([made], device="cuda")
            dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
            if int(agreed.item()) == 0:
                # every rank falls back together: the same sums through torch.distributed's RCCL process group, issued
                # from the all-reduce callback on the engine's streams
                comm = TorchComm(torch.device("cuda", local_rank))
                native_comm = False
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
        shard_o = dp.problems[0].obs_pose.shape[0]
        shard_p = dp.problems[0].points.shape[0]
        kernels = kernel_table(dp, hipba, lm_options(1), n_c, shard_p, shard_o, nnz, None, True)
        if lin != hipba.LINEAR_DENSE_CHOLESKY:
            for k in kernels:
                if k["kernel"] == "reduced_solve":
                    k.update({"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None})
        traffic = load_traffic("bal_scale %d %d %d world %d" % (n_c, n_p, n_o, world))
        for k in kernels:
            if k["kernel"] in traffic:
                k["traffic"] = traffic[k["kernel"]]["hbm_bytes"]
        add_vector_issue_bound(kernels, load_issue("bal_scale %d %d %d world %d" % (n_c, n_p, n_o, world)))
        roof = roofline_of(kernels)
        result = {
            "metric": "BA LM iterations/sec", "value": steps_done / elapsed, "unit": "LM iterations/s",
            "n_gpus": world, "steps": steps_done, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / max(steps_done, 1),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BAL-scale synthetic BA: %d cameras x %d landmarks x %d observations, "
                                   "landmark-aligned observation shards over %d GPU(s), reduced camera system by %s; "
                                   "start %.1f m / %.2f rad / %.1f m from the truth, Ceres default tolerances"
                                   % (n_c, n_p, n_o, world, "fp64 matrix-core Cholesky" if lin == 1 else "block-Jacobi PCG",
                                      scenes.BENCH_START["sigma_t"], scenes.BENCH_START["sigma_w"], scenes.BENCH_START["sigma_p"]),
                       "cameras": n_c, "landmarks": n_p, "observations": n_o, "linear_solver": args.linear_solver,
                       "termination_type": TERMINATION[summary["termination_type"]],
                       "num_successful_steps": summary["num_successful_steps"] - 1,  # Ceres counts iteration 0 too
                       "num_unsuccessful_steps": summary["num_unsuccessful_steps"],
                       "final_cost": summary["final_cost"], "initial_cost": summary["initial_cost"]},
            "roofline": {k: roof[k] for k in roof if k != "kernel"},
            "roofline_kernel": roof["kernel"],
            "kernels": kernels,
        }
        if world > 1:
            result["allreduce_calls_per_step"] = comm.calls / max(steps_done + args.warmup, 1)
            result["allreduce_bytes_per_step"] = comm.bytes / max(steps_done + args.warmup, 1)
            result["collective"] = ("gloo through torch.distributed, every rank on device 0 (rehearsal)" if rehearsal
                                    else "ncclAllReduce (RCCL) issued by the engine on its own streams" if native_comm
                                    else "torch.distributed all_reduce (RCCL process group) from the engine's callback")
            result["collective_ranks"] = world
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
        threads = min(16, os.cpu_count() or 1)
        if world == 1 and not args.no_secondary:
            dp.free()
            result["secondary"] = secondary_measurements(solver, not args.no_cpu_baseline, threads)
        if world == 1 and not args.no_cpu_baseline:
            # (the engine's log of the first steps, for the step-by-step comparison with the oracle at full size)
            dp2 = solver.upload(prob)
            first = dp2.solve(lm_options(2), log=True)[0]["iterations"]
            dp2.free()
            full = prob if args.points == 1_000_000 else None
            result["cpu_baseline"] = cpu_baseline(threads, solver, full, first)
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
