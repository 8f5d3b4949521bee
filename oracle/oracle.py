"""ctypes access to the CPU oracle (oracle/libba_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.  PARITY UNPINNED:
see the header of ba_oracle.h.
"""
import ctypes as C
import math
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libba_oracle.so")

Z_FORWARD, X_FORWARD = 0, 1
LOSS_NONE, LOSS_HUBER = 0, 1
JAC_NUMERIC, JAC_ANALYTIC = 0, 1
LIN_AUTO, LIN_FULL, LIN_SCHUR = 0, 1, 2
CONVERGENCE, NO_CONVERGENCE, FAILURE = 0, 1, 2

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_bp = C.POINTER(C.c_uint8)


class CProblem(C.Structure):
    _fields_ = [
        ("n_poses", C.c_int32), ("poses", _dp), ("pose_const", _bp),
        ("n_points", C.c_int32), ("points", _dp), ("point_const", _bp),
        ("n_obs", C.c_int64), ("obs_pose", _ip), ("obs_point", _ip), ("obs_uv", _dp),
        ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
        ("camera_model", C.c_int32),
    ]


class COptions(C.Structure):
    _fields_ = [
        ("max_num_iterations", C.c_int32), ("loss", C.c_int32), ("loss_scale", C.c_double),
        ("jacobian_mode", C.c_int32), ("linear_solver", C.c_int32), ("num_threads", C.c_int32),
        ("jacobi_scaling", C.c_int32), ("param_tol_needs_success", C.c_int32),
        ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double),
        ("min_trust_region_radius", C.c_double), ("min_relative_decrease", C.c_double),
        ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
        ("max_num_consecutive_invalid_steps", C.c_int32),
        ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
        ("parameter_tolerance", C.c_double),
    ]


class CIteration(C.Structure):
    _fields_ = [
        ("iteration", C.c_int32), ("step_is_valid", C.c_int32), ("step_is_successful", C.c_int32),
        ("cost", C.c_double), ("cost_change", C.c_double), ("gradient_max_norm", C.c_double),
        ("gradient_norm", C.c_double), ("step_norm", C.c_double), ("relative_decrease", C.c_double),
        ("trust_region_radius", C.c_double),
    ]


class CSummary(C.Structure):
    _fields_ = [
        ("termination_type", C.c_int32), ("num_iterations", C.c_int32),
        ("num_successful_steps", C.c_int32), ("num_unsuccessful_steps", C.c_int32),
        ("initial_cost", C.c_double), ("final_cost", C.c_double), ("fixed_cost", C.c_double),
        ("time_evaluate_s", C.c_double), ("time_linear_solve_s", C.c_double), ("time_total_s", C.c_double),
        ("num_active_poses", C.c_int32), ("num_active_points", C.c_int32), ("num_active_obs", C.c_int64),
    ]


_lib = None
_variants = {}


def build():
    """Compile the oracle with gcc (seconds)."""
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib(variant=None):
    """The oracle library.  variant="fma": the same source compiled with fused multiply-adds allowed
    (libba_oracle_fma.so) -- a second rounding of the same arithmetic, which tests use to tell how far apart two
    correct fp64 evaluations of one ill-conditioned step may land; it only serves solve()."""
    global _lib
    if variant is not None:
        if variant not in _variants:
            path = os.path.join(_HERE, "libba_oracle_%s.so" % variant)
            if not os.path.exists(path):
                build()
            v = C.CDLL(path)
            v.oracle_solve.argtypes = [C.POINTER(CProblem), C.POINTER(COptions), C.POINTER(CSummary), C.POINTER(CIteration)]
            v.oracle_solve.restype = C.c_int
            _variants[variant] = v
        return _variants[variant]
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_options_default.argtypes = [C.POINTER(COptions)]
        _lib.oracle_residual.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                         _dp, _dp, C.c_double, C.c_double, _dp]
        _lib.oracle_se3_exp.argtypes = [_dp, _dp]
        _lib.oracle_se3_mul.argtypes = [_dp, _dp, _dp]
        _lib.oracle_se3_plus.argtypes = [_dp, _dp, _dp]
        _lib.oracle_se3_inverse.argtypes = [_dp, _dp]
        _lib.oracle_se3_plus_jacobian.argtypes = [_dp, _dp]
        _lib.oracle_evaluate.argtypes = [C.POINTER(CProblem), C.c_int, C.c_int, C.c_double, C.c_int, C.c_int,
                                         _dp, _dp, _dp, _dp, _dp]
        _lib.oracle_evaluate.restype = C.c_int
        _lib.oracle_solve.argtypes = [C.POINTER(CProblem), C.POINTER(COptions), C.POINTER(CSummary),
                                      C.POINTER(CIteration)]
        _lib.oracle_solve.restype = C.c_int
        _lib.oracle_lm_step.argtypes = [C.POINTER(CProblem), C.POINTER(COptions), C.c_double, _dp, _dp, _dp]
        _lib.oracle_lm_step.restype = C.c_int
        _lib.oracle_triangulate_stereo.argtypes = [_dp, _dp, C.c_int64, C.POINTER(C.c_float), C.POINTER(C.c_float), _dp, _dp]
        _lib.oracle_triangulate_stereo.restype = None
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp)


class Problem:
    """Flat BA problem (copies its inputs; poses/points are updated in place by solve)."""

    def __init__(self, poses, points, obs_pose, obs_point, obs_uv, K, camera_model=Z_FORWARD,
                 pose_const=None, point_const=None):
        self.poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 7).copy()
        self.points = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3).copy()
        self.obs_pose = np.ascontiguousarray(obs_pose, dtype=np.int32).copy()
        self.obs_point = np.ascontiguousarray(obs_point, dtype=np.int32).copy()
        self.obs_uv = np.ascontiguousarray(obs_uv, dtype=np.float64).reshape(-1, 2).copy()
        self.K = tuple(float(k) for k in K)  # fx fy cx cy
        self.camera_model = int(camera_model)
        self.pose_const = None if pose_const is None else np.ascontiguousarray(pose_const, dtype=np.uint8).copy()
        self.point_const = None if point_const is None else np.ascontiguousarray(point_const, dtype=np.uint8).copy()

    def c(self):
        p = CProblem()
        p.n_poses = self.poses.shape[0]
        p.poses = _d(self.poses)
        p.pose_const = None if self.pose_const is None else self.pose_const.ctypes.data_as(_bp)
        p.n_points = self.points.shape[0]
        p.points = _d(self.points)
        p.point_const = None if self.point_const is None else self.point_const.ctypes.data_as(_bp)
        p.n_obs = self.obs_pose.shape[0]
        p.obs_pose = self.obs_pose.ctypes.data_as(_ip)
        p.obs_point = self.obs_point.ctypes.data_as(_ip)
        p.obs_uv = _d(self.obs_uv)
        p.fx, p.fy, p.cx, p.cy = self.K
        p.camera_model = self.camera_model
        return p


def default_options(**kw):
    o = COptions()
    lib().oracle_options_default(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def residual(camera_model, K, pose7, point3, u, v):
    pose7 = np.ascontiguousarray(pose7, dtype=np.float64)
    point3 = np.ascontiguousarray(point3, dtype=np.float64)
    out = np.zeros(2)
    lib().oracle_residual(camera_model, K[0], K[1], K[2], K[3], _d(pose7), _d(point3), float(u), float(v), _d(out))
    return out


def se3_exp(t6):
    t6 = np.ascontiguousarray(t6, dtype=np.float64)
    out = np.zeros(7)
    lib().oracle_se3_exp(_d(t6), _d(out))
    return out


def se3_mul(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    out = np.zeros(7)
    lib().oracle_se3_mul(_d(a), _d(b), _d(out))
    return out


def se3_plus(pose7, d6):
    pose7 = np.ascontiguousarray(pose7, dtype=np.float64)
    d6 = np.ascontiguousarray(d6, dtype=np.float64)
    out = np.zeros(7)
    lib().oracle_se3_plus(_d(pose7), _d(d6), _d(out))
    return out


def se3_inverse(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    out = np.zeros(7)
    lib().oracle_se3_inverse(_d(a), _d(out))
    return out


def se3_plus_jacobian(pose7):
    pose7 = np.ascontiguousarray(pose7, dtype=np.float64)
    out = np.zeros((7, 6))
    lib().oracle_se3_plus_jacobian(_d(pose7), _d(out))
    return out


def evaluate(prob, jacobian_mode=JAC_ANALYTIC, loss=LOSS_NONE, loss_scale=1.0, apply_loss=True,
             num_threads=1, ambient=False):
    """Returns dict(cost, residuals[N,2], jac_pose[N,2,6], jac_point[N,2,3], (jac_pose_ambient[N,2,7]))."""
    n = prob.obs_pose.shape[0]
    res = np.zeros((n, 2))
    jc = np.zeros((n, 2, 6))
    jp = np.zeros((n, 2, 3))
    ja = np.zeros((n, 2, 7)) if ambient else None
    cost = C.c_double(0.0)
    cp = prob.c()
    rc = lib().oracle_evaluate(C.byref(cp), jacobian_mode, loss, float(loss_scale), int(bool(apply_loss)),
                               num_threads, C.cast(C.byref(cost), _dp), _d(res), _d(jc), _d(jp),
                               _d(ja) if ambient else None)
    out = {"rc": rc, "cost": cost.value, "residuals": res, "jac_pose": jc, "jac_point": jp}
    if ambient:
        out["jac_pose_ambient"] = ja
    return out


def solve(prob, options=None, variant=None):
    """Runs the restated Ceres solve in place on prob; returns (summary dict, iteration list)."""
    o = options if options is not None else default_options()
    s = CSummary()
    log = (CIteration * (o.max_num_iterations + 2))()
    cp = prob.c()
    rc = lib(variant).oracle_solve(C.byref(cp), C.byref(o), C.byref(s), log)
    summ = {f[0]: getattr(s, f[0]) for f in CSummary._fields_}
    summ["rc"] = rc
    its = [{f[0]: getattr(log[i], f[0]) for f in CIteration._fields_} for i in range(s.num_iterations)]
    return summ, its


def lm_step(prob, options, radius):
    sp = np.zeros((prob.poses.shape[0], 6))
    sq = np.zeros((prob.points.shape[0], 3))
    mc = C.c_double(0.0)
    cp = prob.c()
    rc = lib().oracle_lm_step(C.byref(cp), C.byref(options), float(radius), _d(sp), _d(sq),
                              C.cast(C.byref(mc), _dp))
    return rc, sp, sq, mc.value


def triangulate_stereo(P_left, P_right, left_uv, right_uv, pose=None):
    """oracle_triangulate_stereo: [n, 3] points (world frame when `pose` -- qx qy qz qw tx ty tz -- is given)."""
    pl = np.ascontiguousarray(P_left, dtype=np.float64).reshape(12)
    pr = np.ascontiguousarray(P_right, dtype=np.float64).reshape(12)
    l = np.ascontiguousarray(left_uv, dtype=np.float32).reshape(-1, 2)
    r = np.ascontiguousarray(right_uv, dtype=np.float32).reshape(-1, 2)
    assert l.shape == r.shape
    out = np.zeros((l.shape[0], 3))
    ps = None if pose is None else np.ascontiguousarray(pose, dtype=np.float64).reshape(7)
    fp = C.POINTER(C.c_float)
    lib().oracle_triangulate_stereo(_d(pl), _d(pr), l.shape[0], l.ctypes.data_as(fp), r.ctypes.data_as(fp),
                                  None if ps is None else _d(ps), _d(out))
    return out


def _splitmix64(state):
    state = (state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return state, z ^ (z >> 31)


def _cubic_largest_root(a2, a1, a0):
    """Largest real root of z^3 + a2 z^2 + a1 z + a0 (trigonometric form / Cardano), two Newton steps."""
    Q = (a2 * a2 - 3.0 * a1) / 9.0
    R = (2.0 * a2 ** 3 - 9.0 * a2 * a1 + 27.0 * a0) / 54.0
    Q3 = Q ** 3
    if R * R < Q3:
        th = math.acos(min(max(R / math.sqrt(Q3), -1.0), 1.0))
        m = -2.0 * math.sqrt(Q)
        z = max(m * math.cos(th / 3.0), m * math.cos((th + 2.0 * math.pi) / 3.0),
                m * math.cos((th - 2.0 * math.pi) / 3.0)) - a2 / 3.0
    else:
        A = -math.copysign((abs(R) + math.sqrt(max(R * R - Q3, 0.0))) ** (1.0 / 3.0), R)
        B = Q / A if A != 0.0 else 0.0
        z = A + B - a2 / 3.0
    for _ in range(2):
        f = ((z + a2) * z + a1) * z + a0
        df = (3.0 * z + 2.0 * a2) * z + a1
        if df != 0.0 and math.isfinite(f / df):
            z -= f / df
    return z


def _quartic_real_roots(c4, c3, c2, c1, c0):
    """Real roots of a quartic by Ferrari's method (resolvent cubic, two quadratics), three Newton steps each."""
    if c4 == 0.0:
        return []
    b, c, d, e = c3 / c4, c2 / c4, c1 / c4, c0 / c4
    if not all(math.isfinite(x) for x in (b, c, d, e)):
        return []
    p = c - 0.375 * b * b
    q = d - 0.5 * b * c + 0.125 * b ** 3
    r = e - 0.25 * b * d + 0.0625 * b * b * c - 0.01171875 * b ** 4
    ys = []
    z = _cubic_largest_root(2.0 * p, p * p - 4.0 * r, -q * q)
    if not z > 1e-12 * (1.0 + abs(p)):
        disc = p * p - 4.0 * r
        if -1e-12 * (p * p + abs(4.0 * r)) < disc < 0.0:
            disc = 0.0
        if disc >= 0.0:
            for w in (0.5 * (-p + math.sqrt(disc)), 0.5 * (-p - math.sqrt(disc))):
                if w >= 0.0:
                    ys += [math.sqrt(w), -math.sqrt(w)]
    else:
        sq = math.sqrt(z)
        t1, t2 = 0.5 * (p + z - q / sq), 0.5 * (p + z + q / sq)
        d1, d2 = z - 4.0 * t1, z - 4.0 * t2
        if -1e-12 * (z + abs(4.0 * t1)) < d1 < 0.0:
            d1 = 0.0
        if -1e-12 * (z + abs(4.0 * t2)) < d2 < 0.0:
            d2 = 0.0
        if d1 >= 0.0:
            ys += [0.5 * (-sq + math.sqrt(d1)), 0.5 * (-sq - math.sqrt(d1))]
        if d2 >= 0.0:
            ys += [0.5 * (sq + math.sqrt(d2)), 0.5 * (sq - math.sqrt(d2))]
    out = []
    for y in ys:
        v = y - 0.25 * b
        for _ in range(3):
            f = (((v + b) * v + c) * v + d) * v + e
            df = ((4.0 * v + 3.0 * b) * v + 2.0 * c) * v + d
            if df != 0.0 and math.isfinite(f / df):
                v -= f / df
        if math.isfinite(v):
            out.append(v)
    return out


def _triad(A, B, Cc):
    d1, d2 = B - A, Cc - A
    e3 = np.cross(d1, d2)
    n1, n2, n3 = np.linalg.norm(d1), np.linalg.norm(d2), np.linalg.norm(e3)
    if not n1 > 0.0 or not n3 > 1e-12 * n1 * n2:
        return None
    e1, e3 = d1 / n1, e3 / n3
    return np.stack([e1, np.cross(e3, e1), e3])


def _quat_of_rotation(M):
    tr = M[0, 0] + M[1, 1] + M[2, 2]
    if tr > 0.0:
        s = 2.0 * math.sqrt(1.0 + tr)
        return np.array([(M[2, 1] - M[1, 2]) / s, (M[0, 2] - M[2, 0]) / s, (M[1, 0] - M[0, 1]) / s, 0.25 * s])
    if M[0, 0] > M[1, 1] and M[0, 0] > M[2, 2]:
        s = 2.0 * math.sqrt(1.0 + M[0, 0] - M[1, 1] - M[2, 2])
        return np.array([0.25 * s, (M[0, 1] + M[1, 0]) / s, (M[0, 2] + M[2, 0]) / s, (M[2, 1] - M[1, 2]) / s])
    if M[1, 1] > M[2, 2]:
        s = 2.0 * math.sqrt(1.0 + M[1, 1] - M[0, 0] - M[2, 2])
        return np.array([(M[0, 1] + M[1, 0]) / s, 0.25 * s, (M[1, 2] + M[2, 1]) / s, (M[0, 2] - M[2, 0]) / s])
    s = 2.0 * math.sqrt(1.0 + M[2, 2] - M[0, 0] - M[1, 1])
    return np.array([(M[0, 2] + M[2, 0]) / s, (M[1, 2] + M[2, 1]) / s, 0.25 * s, (M[1, 0] - M[0, 1]) / s])


def p3p_poses(K, P, uv, camera_model=Z_FORWARD):
    """Camera-to-world poses (qx qy qz qw tx ty tz) that put three world points P [3, 3] on the pixels uv [3, 2]:
    Grunert's perspective-three-point solution (quartic in the ratio of two of the points' distances; coefficients as
    in Haralick et al., IJCV 1994), one pose per real root with positive distances, from the rigid motion between the
    two triangles.  cv::solvePnPRansac (estimator.cpp:28-29) solves its minimal samples in closed form; OpenCV is
    absent here, PARITY UNPINNED."""
    fx, fy, cx, cy = K
    f = []
    for k in range(3):
        x, y = (uv[k][0] - cx) / fx, (uv[k][1] - cy) / fy
        n = 1.0 / math.sqrt(1.0 + x * x + y * y)
        f.append(np.array([x * n, y * n, n]) if camera_model == Z_FORWARD else np.array([n, -x * n, -y * n]))
    P = np.asarray(P, dtype=np.float64)
    Ep = _triad(P[0], P[1], P[2])
    if Ep is None:
        return []
    a2, b2, c2 = ((P[1] - P[2]) ** 2).sum(), ((P[0] - P[2]) ** 2).sum(), ((P[0] - P[1]) ** 2).sum()
    ca, cb, cg = f[1] @ f[2], f[0] @ f[2], f[0] @ f[1]
    k, m, ab, cr = (a2 - c2) / b2, (a2 + c2) / b2, a2 / b2, c2 / b2
    A4 = (k - 1.0) ** 2 - 4.0 * cr * ca * ca
    A3 = 4.0 * (k * (1.0 - k) * cb - (1.0 - m) * ca * cg + 2.0 * cr * ca * ca * cb)
    A2 = 2.0 * (k * k - 1.0 + 2.0 * k * k * cb * cb + 2.0 * (1.0 - cr) * ca * ca - 4.0 * m * ca * cb * cg
                + 2.0 * (1.0 - ab) * cg * cg)
    A1 = 4.0 * (-k * (1.0 + k) * cb + 2.0 * ab * cg * cg * cb - (1.0 - m) * ca * cg)
    A0 = (1.0 + k) ** 2 - 4.0 * ab * cg * cg
    out = []
    with np.errstate(all="ignore"):
        for v in _quartic_real_roots(A4, A3, A2, A1, A0):
            if not v > 0.0:
                continue
            den_u = 2.0 * (cg - v * ca)
            if den_u == 0.0:
                continue
            u = ((k - 1.0) * v * v - 2.0 * k * cb * v + 1.0 + k) / den_u
            den = 1.0 + v * v - 2.0 * v * cb
            if not u > 0.0 or not math.isfinite(u) or not den > 0.0:
                continue
            s1 = math.sqrt(b2 / den)
            Q = np.stack([s1 * f[0], u * s1 * f[1], v * s1 * f[2]])
            Eq = _triad(Q[0], Q[1], Q[2])
            if Eq is None:
                continue
            M = Ep.T @ Eq
            pose = np.concatenate([_quat_of_rotation(M), P[0] - M @ Q[0]])
            if np.isfinite(pose).all():
                out.append(pose)
    return out


def ransac_update_iterations(p, ep, model_points, max_iters):
    """cv::RANSACUpdateNumIters (OpenCV modules/calib3d/src/ptsetreg.cpp), restated."""
    p = min(max(p, 0.0), 1.0)
    ep = min(max(ep, 0.0), 1.0)
    num = max(1.0 - p, sys.float_info.min)
    denom = 1.0 - (1.0 - ep) ** model_points
    if denom < sys.float_info.min:
        return 0
    num, denom = math.log(num), math.log(denom)
    if denom >= 0.0 or -num >= max_iters * (-denom):
        return max_iters
    return int(round(num / denom))  # (Python rounds halves to even, as cvRound does)


def pnp_ransac(K, points, uv, prior=None, iterations=1000, reprojection_error=3.0, seed=0, camera_model=Z_FORWARD,
               confidence=1.0):
    """PnP by random sample consensus the way the engine's hipba_pnp_ransac states it (estimator.cpp:28-29 calls
    cv::solvePnPRansac; OpenCV is absent, PARITY UNPINNED): per hypothesis four matches from a splitmix64 stream seeded
    by (seed, hypothesis number); the start of the fit is `prior`, or with prior=None the closed-form pose of the first
    three matches that reprojects the fourth best (p3p_poses; a sample without one counts no inliers); at most ten
    Gauss-Newton steps on the four matches' reprojection error from there (normal equations with a 1e-9 relative
    damping, solved exactly), then the count of matches within the threshold.  The hypotheses are then walked in
    order as OpenCV's loop walks them: a count above every earlier one (and above 3) becomes the best and shrinks the
    iterations to run (ransac_update_iterations); hypotheses past that bound do not count.
    Returns (pose before any refinement, mask, counts, iterations the loop runs; pose None when nothing reaches 4).
    Python loops over the hypotheses: for the small cases of tests."""
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    px = np.ascontiguousarray(uv, dtype=np.float64).reshape(-1, 2)
    n = pts.shape[0]
    thr2 = reprojection_error * reprojection_error
    ident = np.array([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0])
    everything = Problem(ident.reshape(1, 7).copy(), pts, np.zeros(n, dtype=np.int32),
                         np.arange(n, dtype=np.int32), px, K, camera_model, None, np.ones(n, dtype=np.uint8))
    counts = np.zeros(iterations, dtype=np.int64)
    poses = np.zeros((iterations, 7))
    for h in range(iterations):
        state = (seed + 0xD1B54A32D192ED03 * (h + 1)) & 0xFFFFFFFFFFFFFFFF
        idx = []
        for k in range(4):
            for _ in range(16):
                state, z = _splitmix64(state)
                cand = z % n
                if cand not in idx:
                    break
            idx.append(cand)
        ii = np.array(idx)
        if prior is None:
            best_err, pose = None, None
            for cand_pose in p3p_poses(K, pts[ii[:3]], px[ii[:3]], camera_model):
                one = Problem(cand_pose.reshape(1, 7), pts[ii[3:]], np.zeros(1, dtype=np.int32), np.zeros(1, dtype=np.int32),
                              px[ii[3:]], K, camera_model, None, np.ones(1, dtype=np.uint8))
                r4 = evaluate(one, JAC_ANALYTIC, apply_loss=False)["residuals"].reshape(2)
                err = float(r4 @ r4)
                if err == err and (best_err is None or err < best_err):
                    best_err, pose = err, cand_pose.copy()
            if pose is None:
                poses[h] = ident
                continue
        else:
            pose = np.asarray(prior, dtype=np.float64).reshape(7).copy()
        for _ in range(10):
            mini = Problem(pose.reshape(1, 7), pts[ii], np.zeros(4, dtype=np.int32), np.arange(4, dtype=np.int32), px[ii], K,
                           camera_model, None, np.ones(4, dtype=np.uint8))
            e = evaluate(mini, JAC_ANALYTIC, apply_loss=False)
            J = e["jac_pose"].reshape(8, 6)
            r = e["residuals"].reshape(8)
            H = J.T @ J
            H[np.diag_indices(6)] += 1e-9 * (1.0 + np.diag(H))
            try:
                L = np.linalg.cholesky(H)
            except np.linalg.LinAlgError:
                break
            d = -np.linalg.solve(L.T, np.linalg.solve(L, J.T @ r))
            if not np.isfinite(d).all():
                break
            pose = se3_plus(pose, d)
            if not (d @ d > 1e-24):
                break
        everything.poses[0] = pose
        res = evaluate(everything, JAC_ANALYTIC, apply_loss=False)["residuals"]
        counts[h] = int(((res * res).sum(axis=1) <= thr2).sum())
        poses[h] = pose
    best, best_count, limit, h = -1, 3, iterations, 0
    while h < limit:
        if counts[h] > best_count:
            best, best_count = h, int(counts[h])
            limit = ransac_update_iterations(confidence, (n - best_count) / n, 4, limit)
        h += 1
    if best < 0:
        return None, np.zeros(n, dtype=np.uint8), counts, limit
    everything.poses[0] = poses[best]
    res = evaluate(everything, JAC_ANALYTIC, apply_loss=False)["residuals"]
    return poses[best].copy(), ((res * res).sum(axis=1) <= thr2).astype(np.uint8), counts, limit
