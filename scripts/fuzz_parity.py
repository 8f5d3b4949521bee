"""Randomised parity sweep on the GPU box: many small and medium windows with random structure and options through
hipba_solve (one by one and batched) against the CPU oracle with analytic Jacobians.  Prints every case that leaves the
tolerances of the parity tests (final cost 1e-6 relative, poses and landmarks 1e-6, the same accept/reject sequence).

Usage: python scripts/fuzz_parity.py [n_cases] [first_seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle
from stereo_visual_odometry_amd import hipba, scenes


def random_case(seed):
    rng = np.random.default_rng(seed)
    n_frames = int(rng.choice([2, 3, 5, 8, 10, 13, 21, 22, 30, 33, 40, 48, 64]))
    n_tracks = int(rng.choice([6, 40, 300, 300, 1500, 1500]))
    lo = int(rng.choice([1, 2, 3, 4], p=[0.1, 0.1, 0.4, 0.4]))
    lo = min(lo, n_frames)
    hi = int(rng.integers(lo, n_frames + 1))
    gt_p, gt_q, poses, points, op, oq, uv = scenes._window(rng, n_frames, n_tracks, lo, hi,
                                                           sigma_t=float(rng.choice([0.02, 0.1])),
                                                           sigma_w=float(rng.choice([0.002, 0.02])))
    model = hipba.CAMERA_Z_FORWARD
    kind = int(rng.integers(0, 5))
    pose_const = np.zeros(n_frames, dtype=np.uint8)
    point_const = np.zeros(n_tracks, dtype=np.uint8)
    if kind == 0:      # production semantics
        pose_const, point_const = scenes._production_constants(n_frames, n_tracks, op, oq)
    elif kind == 1:    # two constant poses (gauge fixed), free landmarks
        pose_const[:2] = 1
    elif kind == 2:    # random constants
        pose_const[rng.uniform(size=n_frames) < 0.3] = 1
        pose_const[0] = 1
        point_const[rng.uniform(size=n_tracks) < 0.3] = 1
    elif kind == 3:    # everything but one pose constant (PnP-like), or all landmarks constant
        point_const[:] = 1
        if rng.uniform() < 0.5:
            pose_const[:] = 1
            pose_const[int(rng.integers(0, n_frames))] = 0
    else:              # first pose constant only
        pose_const[0] = 1
    # some outliers for the robust loss
    loss = int(rng.choice([hipba.LOSS_NONE, hipba.LOSS_HUBER]))
    if loss == hipba.LOSS_HUBER and uv.shape[0] > 10:
        bad = rng.choice(uv.shape[0], max(1, uv.shape[0] // 20), replace=False)
        uv[bad] += rng.normal(0.0, 30.0, (bad.shape[0], 2))
    # duplicates: a pose that sees a landmark twice
    if rng.uniform() < 0.2 and uv.shape[0] > 4:
        k = rng.choice(uv.shape[0], 3, replace=False)
        op = np.concatenate([op, op[k]]); oq = np.concatenate([oq, oq[k]])
        uv = np.concatenate([uv, uv[k] + rng.normal(0.0, 0.3, (3, 2))])
    if rng.uniform() < 0.3:
        # the harness's camera (x forward, y left, z up: test/ba_test.h:179-211): the same scene with every camera frame
        # turned by the fixed rotation that maps (x, y, z) to (z, -x, -y)
        model = hipba.CAMERA_X_FORWARD
        poses = poses.copy()
        poses[:, :4] = scenes._quat_mul(poses[:, :4], np.array([0.5, -0.5, 0.5, 0.5]))
    prob = hipba.Problem(poses, points, op, oq, uv, scenes.KITTI_K, model, pose_const, point_const)
    opts = dict(max_num_iterations=int(rng.choice([5, 25, 60])), loss=loss, loss_scale=float(rng.choice([1.0, 2.5])))
    return prob, opts, "frames %d tracks %d len %d-%d kind %d loss %d model %d" % (n_frames, n_tracks, lo, hi, kind, loss, model)


def well_posed(prob):
    """Every variable landmark seen by at least three poses, every variable pose sees at least eight landmarks, and the
    gauge is fixed (a constant pose plus either a second one or a constant landmark): only then are the optimum's
    parameters determined well enough to be compared to 1e-6."""
    pc = prob.pose_const.astype(bool) if prob.pose_const is not None else np.zeros(prob.poses.shape[0], bool)
    qc = prob.point_const.astype(bool) if prob.point_const is not None else np.zeros(prob.points.shape[0], bool)
    if pc.sum() == 0 or (pc.sum() < 2 and qc.sum() == 0):
        return False
    deg = np.bincount(prob.obs_point, minlength=prob.points.shape[0])
    seen = np.bincount(prob.obs_pose, minlength=prob.poses.shape[0])
    return bool((deg[~qc] >= 3).all() and (seen[~pc] >= 8).all())


def compare(name, seed, s, its, gp, so, oits, o, strict):
    """strict: a well-posed problem -- everything is compared.  Otherwise only what rounding cannot move much: the log
    up to the first three iterations and the accept/reject pattern there."""
    bad = []
    n_cmp = min(len(its), len(oits)) if strict else min(len(its), len(oits), 4)
    for a, b in list(zip(its, oits))[:n_cmp]:
        if a["step_is_successful"] != b["step_is_successful"]:
            bad.append("accept/reject differs at iteration %d" % a["iteration"])
            break
        if abs(a["cost"] - b["cost"]) > 1e-6 * max(abs(b["cost"]), 1e-9) + 1e-12:
            bad.append("cost at iteration %d: %.12e vs %.12e" % (a["iteration"], a["cost"], b["cost"]))
            break
    if strict:
        if s["termination_type"] != so["termination_type"]:
            bad.append("termination %d vs %d" % (s["termination_type"], so["termination_type"]))
        if len(its) != len(oits):
            bad.append("iterations %d vs %d" % (len(its), len(oits)))
        fc, oc = s["final_cost"], so["final_cost"]
        if abs(fc - oc) > 1e-6 * max(abs(oc), 1e-12) + 1e-12:
            bad.append("final cost %.12e vs %.12e" % (fc, oc))
        dp = np.abs(gp.poses - o.poses).max()
        dq = np.abs(gp.points - o.points).max() if gp.points.size else 0.0
        if dp > 1e-6 or dq > 1e-5:
            bad.append("poses differ by %.2e, landmarks by %.2e" % (dp, dq))
    if bad:
        print("seed %d [%s]%s %s" % (seed, name, " well-posed:" if strict else " loose:", "; ".join(bad)), flush=True)
    return not bad


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
    solver = hipba.Solver(0)
    ok = 0
    n_strict = 0
    t0 = time.time()
    held = []  # for a batched pass: windows of at most 32 variable poses with the same options
    for seed in range(first, first + n_cases):
        prob, okw, name = random_case(seed)
        if prob.obs_pose.shape[0] == 0:
            continue
        o = oracle.Problem(prob.poses, prob.points, prob.obs_pose, prob.obs_point, prob.obs_uv, prob.K, prob.camera_model,
                           prob.pose_const, prob.point_const)
        so, oits = oracle.solve(o, oracle.default_options(jacobian_mode=oracle.JAC_ANALYTIC, **okw))
        g = prob.copy()
        try:
            s = solver.solve(g, hipba.default_options(**okw), log=True)
        except hipba.HipbaError as e:
            print("seed %d [%s] engine error: %s" % (seed, name, e), flush=True)
            continue
        strict = well_posed(prob)
        n_strict += strict
        ok += compare(name, seed, s, s["iterations"], g, so, oits, o, strict)
        if okw["loss"] == hipba.LOSS_NONE and okw["max_num_iterations"] == 25 and prob.poses.shape[0] <= 30:
            held.append((seed, name, prob, so, oits, o))
        if (seed - first) % 25 == 24:
            print("... %d cases, %d in tolerance, %.0f s" % (seed - first + 1, ok, time.time() - t0), flush=True)
    print("one by one: %d of %d in tolerance (%d well-posed, compared in full)" % (ok, n_cases, n_strict), flush=True)
    if held:
        dp = solver.upload([h[2].copy() for h in held])
        res = dp.solve(hipba.default_options(max_num_iterations=25, loss=hipba.LOSS_NONE), log=True)
        okb = 0
        for w, (seed, name, prob, so, oits, o) in enumerate(held):
            poses, points = dp.download(w)
            g = prob.copy(); g.poses[:] = poses; g.points[:] = points
            okb += compare(name + " (batched)", seed, res[w], res[w]["iterations"], g, so, oits, o, well_posed(prob))
        dp.free()
        print("batched: %d of %d in tolerance" % (okb, len(held)), flush=True)
    solver.close()


if __name__ == "__main__":
    main()
