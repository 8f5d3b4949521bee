"""ctypes binding of libhipba.so (include/hipba.h) -- the MI355X bundle-adjustment engine.

The binding mirrors the C ABI one to one; there is no Python or CPU fallback: if the
shared library is missing or no gfx950 device is visible, construction fails loudly.
The reference call this stands in for is ``ceres::Solve`` inside
``VO::Optimizer::optimize`` (reference: src/solve/optimizer.cpp:14-63).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libhipba.so")

OK = 0
ERR_INVALID_ARGUMENT, ERR_NO_DEVICE, ERR_HIP, ERR_OUT_OF_MEMORY, ERR_UNSUPPORTED, ERR_COMM, ERR_INTERNAL = (
    -1, -2, -3, -4, -5, -6, -7)
EVALUATE_DEVICE_MAX_OBS = 1 << 24  # HIPBA_EVALUATE_DEVICE_MAX_OBS
CAMERA_Z_FORWARD, CAMERA_X_FORWARD = 0, 1
ABI_VERSION = 104  # HIPBA_VERSION of the include/hipba.h these ctypes structures mirror
LOSS_NONE, LOSS_HUBER = 0, 1
LINEAR_AUTO, LINEAR_DENSE_CHOLESKY, LINEAR_PCG = 0, 1, 2
CONVERGENCE, NO_CONVERGENCE, FAILURE = 0, 1, 2
REDUCE_SUM, REDUCE_MAX = 0, 1
(KERNEL_RESIDUAL, KERNEL_RESIDUAL_JACOBIAN, KERNEL_CAMERA_BLOCKS, KERNEL_POINT_SCHUR, KERNEL_BACKSUB,
 KERNEL_REDUCED_SOLVE) = range(6)
KERNEL_NAMES = ["residual", "residual_jacobian", "camera_blocks", "point_schur", "backsub", "reduced_solve"]

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_bp = C.POINTER(C.c_uint8)


class CProblem(C.Structure):
    # (the array members are declared as untyped addresses -- the layout is that of the header's typed pointers -- so that
    # Problem.c() can store plain integers: seven typed ctypes casts per call were 20 us of a 1 ms window solve)
    _fields_ = [
        ("n_poses", C.c_int32), ("poses", C.c_void_p), ("pose_const", C.c_void_p),
        ("n_points", C.c_int32), ("points", C.c_void_p), ("point_const", C.c_void_p),
        ("n_obs", C.c_int64), ("obs_pose", C.c_void_p), ("obs_point", C.c_void_p), ("obs_uv", C.c_void_p),
        ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
        ("camera_model", C.c_int32),
    ]


class COptions(C.Structure):
    _fields_ = [
        ("max_num_iterations", C.c_int32), ("loss", C.c_int32), ("loss_scale", C.c_double),
        ("linear_solver", C.c_int32), ("jacobi_scaling", C.c_int32),
        ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double),
        ("min_trust_region_radius", C.c_double), ("min_relative_decrease", C.c_double),
        ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
        ("max_num_consecutive_invalid_steps", C.c_int32),
        ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
        ("parameter_tolerance", C.c_double),
        ("pcg_max_iterations", C.c_int32), ("pcg_tolerance", C.c_double),
        ("minimizer_progress_to_stdout", C.c_int32), ("parameter_tolerance_needs_success", C.c_int32),
        ("report_time_breakdown", C.c_int32), ("reserved", C.c_int32),
    ]


class CIteration(C.Structure):
    _fields_ = [
        ("iteration", C.c_int32), ("step_is_valid", C.c_int32), ("step_is_successful", C.c_int32),
        ("linear_solver_iterations", C.c_int32),
        ("cost", C.c_double), ("cost_change", C.c_double), ("gradient_max_norm", C.c_double),
        ("gradient_norm", C.c_double), ("step_norm", C.c_double), ("relative_decrease", C.c_double),
        ("trust_region_radius", C.c_double),
    ]


class CSummary(C.Structure):
    _fields_ = [
        ("termination_type", C.c_int32), ("num_iterations", C.c_int32),
        ("num_successful_steps", C.c_int32), ("num_unsuccessful_steps", C.c_int32),
        ("initial_cost", C.c_double), ("final_cost", C.c_double), ("fixed_cost", C.c_double),
        ("num_active_poses", C.c_int32), ("num_active_points", C.c_int32),
        ("num_active_observations", C.c_int64),
        ("time_setup_s", C.c_double), ("time_solve_s", C.c_double), ("time_total_s", C.c_double),
        ("linear_solver_used", C.c_int32), ("num_linear_solves", C.c_int32),
        ("iteration_log", C.POINTER(CIteration)), ("iteration_log_capacity", C.c_int32), ("reserved2", C.c_int32),
        ("bytes_host_to_device", C.c_int64), ("bytes_device_to_host", C.c_int64),
        ("time_residual_evaluation_s", C.c_double), ("time_jacobian_evaluation_s", C.c_double),
        ("time_linear_solver_s", C.c_double),
    ]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)


class CComm(C.Structure):
    _fields_ = [("allreduce", ALLREDUCE_FN), ("ctx", C.c_void_p), ("rank", C.c_int32), ("world_size", C.c_int32)]


# every symbol include/hipba.h declares
EXPORTED_SYMBOLS = [
    "hipba_version", "hipba_last_error", "hipba_options_default", "hipba_device_count", "hipba_create", "hipba_destroy", "hipba_stream",
    "hipba_solve", "hipba_solve_batched", "hipba_evaluate", "hipba_evaluate_device", "hipba_upload", "hipba_upload_shard", "hipba_pose_order", "hipba_reset", "hipba_solve_device",
    "hipba_solve_device_sharded", "hipba_download", "hipba_free_problem", "hipba_time_kernel", "hipba_reduced_solve_flops",
    "hipba_rccl_get_unique_id", "hipba_rccl_comm_create", "hipba_rccl_comm_stats", "hipba_rccl_comm_destroy",
    "hipba_triangulate_stereo", "hipba_triangulate_stereo_device", "hipba_refine_pose", "hipba_pnp_ransac", "hipba_se3_plus",
]


RCCL_ID_BYTES = 128


class RcclComm:
    """hipba_comm over a native RCCL communicator (hipba_rccl_comm_create): the all-reduces of a sharded solve are
    ncclAllReduce calls made by the library itself on the engine's stream -- no Python in the loop.  Rank 0 makes the
    id (RcclComm.unique_id()) and the job distributes it (e.g. torch.distributed.broadcast_object_list)."""

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(RCCL_ID_BYTES)
        _check(load_library().hipba_rccl_get_unique_id(buf))
        return buf.raw

    def __init__(self, device, unique_id, rank, world_size):
        self._lib = load_library()
        self.c = CComm()
        self.rank, self.world_size = int(rank), int(world_size)
        rc = self._lib.hipba_rccl_comm_create(int(device), unique_id, self.rank, self.world_size, C.byref(self.c))
        if rc != OK:
            raise HipbaError(rc, "hipba_rccl_comm_create failed (is librccl.so loadable?)")

    def _stat(self, which):
        calls, nbytes = C.c_int64(0), C.c_int64(0)
        self._lib.hipba_rccl_comm_stats(C.byref(self.c), C.byref(calls), C.byref(nbytes))
        return (calls.value, nbytes.value)[which]

    @property
    def calls(self):
        return self._stat(0)

    @property
    def bytes(self):
        return self._stat(1)

    def close(self):
        if self.c.ctx:
            self._lib.hipba_rccl_comm_destroy(C.byref(self.c))


class HipbaError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("hipba error %d: %s" % (code, message))
        self.code = code


_lib = None


def load_library(path=None):
    """dlopen libhipba.so and declare its prototypes.  Raises if it is not there."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise ImportError("libhipba.so not found at %s: run ./build.sh (or __graft_entry__.build()); "
                          "there is no CPU fallback" % p)
    lib = C.CDLL(p)
    lib.hipba_version.restype = C.c_int
    if lib.hipba_version() != ABI_VERSION:
        raise ImportError("%s is ABI version %d, this binding mirrors version %d of include/hipba.h: rebuild with "
                          "./build.sh" % (p, lib.hipba_version(), ABI_VERSION))
    lib.hipba_last_error.restype = C.c_char_p
    lib.hipba_options_default.argtypes = [C.POINTER(COptions)]
    lib.hipba_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.hipba_destroy.argtypes = [C.c_void_p]
    lib.hipba_stream.argtypes = [C.c_void_p]
    lib.hipba_stream.restype = C.c_void_p
    lib.hipba_solve.argtypes = [C.c_void_p, C.POINTER(CProblem), C.POINTER(COptions), C.POINTER(CSummary)]
    lib.hipba_solve_batched.argtypes = [C.c_void_p, C.c_int32, C.POINTER(CProblem), C.POINTER(COptions),
                                        C.POINTER(CSummary)]
    lib.hipba_evaluate.argtypes = [C.c_void_p, C.POINTER(CProblem), C.POINTER(COptions), C.c_int, _dp, _dp, _dp, _dp]
    fp = C.POINTER(C.c_float)
    lib.hipba_triangulate_stereo.argtypes = [C.c_void_p, _dp, _dp, C.c_int64, fp, fp, _dp, _dp]
    lib.hipba_triangulate_stereo_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                                    C.c_void_p, C.c_void_p]
    lib.hipba_se3_plus.argtypes = [C.c_void_p, C.c_int64, _dp, _dp, _dp]
    lib.hipba_refine_pose.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_int64,
                                      _dp, _dp, _bp, _dp, C.POINTER(COptions), C.POINTER(CSummary)]
    lib.hipba_pnp_ransac.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_int64, _dp, _dp,
                                     _dp, C.c_int32, C.c_double, C.c_double, C.c_uint64, C.c_int32, _bp, C.POINTER(C.c_int64),
                                     C.POINTER(C.c_int32)]
    lib.hipba_evaluate_device.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double,
                                          C.c_double, C.c_int32, C.POINTER(COptions), C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p]
    lib.hipba_upload.argtypes = [C.c_void_p, C.c_int32, C.POINTER(CProblem), C.POINTER(C.c_void_p)]
    lib.hipba_upload_shard.argtypes = [C.c_void_p, C.POINTER(CProblem), _ip, C.POINTER(C.c_void_p)]
    lib.hipba_pose_order.argtypes = [C.POINTER(CProblem), _ip]
    lib.hipba_reset.argtypes = [C.c_void_p, C.c_void_p]
    lib.hipba_solve_device.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(COptions), C.POINTER(CSummary)]
    lib.hipba_solve_device_sharded.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(CComm), C.POINTER(COptions),
                                               C.POINTER(CSummary)]
    lib.hipba_download.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, _dp, _dp]
    lib.hipba_free_problem.argtypes = [C.c_void_p, C.c_void_p]
    lib.hipba_time_kernel.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(COptions), C.c_int32, C.c_int32, _dp]
    lib.hipba_reduced_solve_flops.argtypes = [C.c_void_p, _dp, _dp]
    lib.hipba_rccl_get_unique_id.argtypes = [C.c_char_p]
    lib.hipba_rccl_comm_create.argtypes = [C.c_int, C.c_char_p, C.c_int32, C.c_int32, C.POINTER(CComm)]
    lib.hipba_rccl_comm_stats.argtypes = [C.POINTER(CComm), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.hipba_rccl_comm_destroy.argtypes = [C.POINTER(CComm)]
    for name in EXPORTED_SYMBOLS:
        getattr(lib, name)  # AttributeError if the library lacks a declared entry point
    if path is None:
        _lib = lib
    return lib


def _check(rc):
    if rc != OK:
        raise HipbaError(rc, (load_library().hipba_last_error() or b"").decode("utf-8", "replace"))


def default_options(**kw):
    o = COptions()
    load_library().hipba_options_default(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def reference_production_options(**kw):
    """The options VO::Optimizer::optimize sets (reference: src/solve/optimizer.cpp:54-58)."""
    return default_options(max_num_iterations=300, **kw)


class Problem:
    """One window as flat arrays (copies its inputs).  poses [N,7] qx qy qz qw tx ty tz, points [M,3],
    observations (pose index, point index, u, v); K = (fx, fy, cx, cy)."""

    def __init__(self, poses, points, obs_pose, obs_point, obs_uv, K, camera_model=CAMERA_Z_FORWARD,
                 pose_const=None, point_const=None):
        self.poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 7).copy()
        self.points = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3).copy()
        self.obs_pose = np.ascontiguousarray(obs_pose, dtype=np.int32).copy()
        self.obs_point = np.ascontiguousarray(obs_point, dtype=np.int32).copy()
        self.obs_uv = np.ascontiguousarray(obs_uv, dtype=np.float64).reshape(-1, 2).copy()
        self.K = tuple(float(k) for k in K)
        self.camera_model = int(camera_model)
        self.pose_const = None if pose_const is None else np.ascontiguousarray(pose_const, dtype=np.uint8).copy()
        self.point_const = None if point_const is None else np.ascontiguousarray(point_const, dtype=np.uint8).copy()

    def copy(self):
        return Problem(self.poses, self.points, self.obs_pose, self.obs_point, self.obs_uv, self.K,
                       self.camera_model, self.pose_const, self.point_const)

    def c(self):
        """The hipba_problem view of the arrays (which must stay alive, and where they are, while it is in use)."""
        addr = lambda a: None if a is None else a.__array_interface__["data"][0]
        return CProblem(self.poses.shape[0], addr(self.poses), addr(self.pose_const),
                        self.points.shape[0], addr(self.points), addr(self.point_const),
                        self.obs_pose.shape[0], addr(self.obs_pose), addr(self.obs_point), addr(self.obs_uv),
                        self.K[0], self.K[1], self.K[2], self.K[3], self.camera_model)


def _summary_dict(s, log=None):
    d = {f[0]: getattr(s, f[0]) for f in CSummary._fields_ if f[0] not in ("iteration_log", "reserved", "reserved2")}
    if log is not None:
        d["iterations"] = [{f[0]: getattr(log[i], f[0]) for f in CIteration._fields_}
                           for i in range(min(s.num_iterations, len(log)))]
    return d


class DeviceProblem:
    """Windows resident in HBM (hipba_upload)."""

    def __init__(self, solver, problems, shard=False, pose_order=None):
        self.solver = solver
        self.problems = list(problems)
        arr = (CProblem * len(self.problems))(*[p.c() for p in self.problems])
        self._h = C.c_void_p()
        if shard:
            po = None if pose_order is None else np.ascontiguousarray(pose_order, dtype=np.int32)
            _check(solver._lib.hipba_upload_shard(solver._h, arr, None if po is None else po.ctypes.data_as(_ip),
                                                  C.byref(self._h)))
        else:
            _check(solver._lib.hipba_upload(solver._h, len(self.problems), arr, C.byref(self._h)))

    @property
    def n_windows(self):
        return len(self.problems)

    def reset(self):
        _check(self.solver._lib.hipba_reset(self.solver._h, self._h))

    def solve(self, options=None, log=False):
        o = options if options is not None else default_options()
        n = self.n_windows
        summ = (CSummary * n)()
        logs = []
        if log:
            for w in range(n):
                lg = (CIteration * (o.max_num_iterations + 2))()
                logs.append(lg)
                summ[w].iteration_log = lg
                summ[w].iteration_log_capacity = len(lg)
        _check(self.solver._lib.hipba_solve_device(self.solver._h, self._h, C.byref(o), summ))
        return [_summary_dict(summ[w], logs[w] if log else None) for w in range(n)]

    def solve_sharded(self, comm, options=None, log=False):
        o = options if options is not None else default_options()
        summ = CSummary()
        lg = None
        if log:
            lg = (CIteration * (o.max_num_iterations + 2))()
            summ.iteration_log = lg
            summ.iteration_log_capacity = len(lg)
        _check(self.solver._lib.hipba_solve_device_sharded(self.solver._h, self._h, C.byref(comm), C.byref(o),
                                                           C.byref(summ)))
        return _summary_dict(summ, lg)

    def download(self, window=0):
        p = self.problems[window]
        poses = np.zeros_like(p.poses)
        points = np.zeros_like(p.points)
        _check(self.solver._lib.hipba_download(self.solver._h, self._h, window, poses.ctypes.data_as(_dp),
                                               points.ctypes.data_as(_dp)))
        return poses, points

    def time_kernel(self, kernel, repeats=10, options=None):
        o = options if options is not None else default_options()
        ms = C.c_double(0.0)
        _check(self.solver._lib.hipba_time_kernel(self.solver._h, self._h, C.byref(o), kernel, repeats,
                                                  C.cast(C.byref(ms), _dp)))
        return ms.value

    def reduced_solve_flops(self):
        """(dense, inside the envelope): flops of one factorisation of the reduced camera system"""
        a, b = C.c_double(0.0), C.c_double(0.0)
        _check(self.solver._lib.hipba_reduced_solve_flops(self._h, C.cast(C.byref(a), _dp), C.cast(C.byref(b), _dp)))
        return a.value, b.value

    def free(self):
        if self._h:
            self.solver._lib.hipba_free_problem(self.solver._h, self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Solver:
    """One handle = one GPU + one stream (hipba_create)."""

    def __init__(self, device=0):
        self._lib = load_library()
        self._h = C.c_void_p()
        self.device = int(device)
        _check(self._lib.hipba_create(self.device, C.byref(self._h)))

    @property
    def stream(self):
        return self._lib.hipba_stream(self._h)

    def close(self):
        if self._h:
            self._lib.hipba_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def solve(self, problem, options=None, log=True):
        """hipba_solve: optimises problem.poses / problem.points in place."""
        return self.solve_batched([problem], options, log)[0]

    def solve_batched(self, problems, options=None, log=False):
        o = options if options is not None else default_options()
        n = len(problems)
        arr = (CProblem * n)(*[p.c() for p in problems])
        summ = (CSummary * n)()
        logs = []
        if log:
            for w in range(n):
                lg = (CIteration * (o.max_num_iterations + 2))()
                logs.append(lg)
                summ[w].iteration_log = lg
                summ[w].iteration_log_capacity = len(lg)
        _check(self._lib.hipba_solve_batched(self._h, n, arr, C.byref(o), summ))
        return [_summary_dict(summ[w], logs[w] if log else None) for w in range(n)]

    def evaluate(self, problem, options=None, apply_loss=True):
        """hipba_evaluate: dict(cost, residuals [N,2], jac_pose [N,2,6], jac_point [N,2,3])."""
        o = options if options is not None else default_options()
        n = problem.obs_pose.shape[0]
        res = np.zeros((n, 2))
        jc = np.zeros((n, 2, 6))
        jp = np.zeros((n, 2, 3))
        cost = C.c_double(0.0)
        cp = problem.c()
        _check(self._lib.hipba_evaluate(self._h, C.byref(cp), C.byref(o), int(bool(apply_loss)),
                                        C.cast(C.byref(cost), _dp), res.ctypes.data_as(_dp),
                                        jc.ctypes.data_as(_dp), jp.ctypes.data_as(_dp)))
        return {"cost": cost.value, "residuals": res, "jac_pose": jc, "jac_point": jp}

    def evaluate_device(self, n_poses, d_poses, n_points, d_points, n_obs, d_obs_pose, d_obs_point, d_obs_uv, K,
                        camera_model=CAMERA_Z_FORWARD, options=None, apply_loss=True, d_cost=None, d_residuals=None,
                        d_jac_pose=None, d_jac_point=None):
        """hipba_evaluate_device: every d_* is a device address (int) owned by the caller; the work is queued on
        this handle's stream and nothing is allocated.  Synchronise the stream before reading the outputs."""
        o = options if options is not None else default_options()
        _check(self._lib.hipba_evaluate_device(self._h, n_poses, d_poses, n_points, d_points, n_obs, d_obs_pose,
                                               d_obs_point, d_obs_uv, K[0], K[1], K[2], K[3], camera_model,
                                               C.byref(o), int(bool(apply_loss)), d_cost, d_residuals, d_jac_pose,
                                               d_jac_point))

    def triangulate_stereo(self, P_left, P_right, left_uv, right_uv, pose=None):
        """hipba_triangulate_stereo (Triangulator::triangulate, triangulator.cpp:48-61): [n, 3] points from n stereo
        matches (float pixel pairs), in the world frame when the frame's pose (qx qy qz qw tx ty tz) is given."""
        pl = np.ascontiguousarray(P_left, dtype=np.float64).reshape(12)
        pr = np.ascontiguousarray(P_right, dtype=np.float64).reshape(12)
        l = np.ascontiguousarray(left_uv, dtype=np.float32).reshape(-1, 2)
        r = np.ascontiguousarray(right_uv, dtype=np.float32).reshape(-1, 2)
        if l.shape != r.shape:
            raise ValueError("left_uv and right_uv differ in length")
        ps = None if pose is None else np.ascontiguousarray(pose, dtype=np.float64).reshape(7)
        out = np.zeros((l.shape[0], 3))
        fp = C.POINTER(C.c_float)
        _check(self._lib.hipba_triangulate_stereo(self._h, pl.ctypes.data_as(_dp), pr.ctypes.data_as(_dp), l.shape[0],
                                                  l.ctypes.data_as(fp), r.ctypes.data_as(fp),
                                                  None if ps is None else ps.ctypes.data_as(_dp), out.ctypes.data_as(_dp)))
        return out

    def triangulate_stereo_device(self, d_P_left, d_P_right, n, d_left_uv, d_right_uv, d_pose, d_points):
        """hipba_triangulate_stereo_device: device addresses (ints), queued on this handle's stream, nothing allocated."""
        _check(self._lib.hipba_triangulate_stereo_device(self._h, d_P_left, d_P_right, n, d_left_uv, d_right_uv, d_pose,
                                                         d_points))

    def se3_plus(self, poses, deltas):
        """hipba_se3_plus (Sophus::Manifold<SE3>::Plus, optimizer.cpp:17,27): poses [n,7] * exp(deltas [n,6]) -> [n,7]."""
        x = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 7)
        d = np.ascontiguousarray(deltas, dtype=np.float64).reshape(-1, 6)
        if x.shape[0] != d.shape[0]:
            raise ValueError("poses and deltas differ in length")
        out = np.zeros_like(x)
        _check(self._lib.hipba_se3_plus(self._h, x.shape[0], x.ctypes.data_as(_dp), d.ctypes.data_as(_dp),
                                        out.ctypes.data_as(_dp)))
        return out

    def refine_pose(self, K, points, uv, pose, use=None, camera_model=CAMERA_Z_FORWARD, options=None):
        """hipba_refine_pose (the refinement that ends cv::solvePnPRansac, estimator.cpp:28-29): (refined pose, summary)."""
        o = options if options is not None else default_options()
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        px = np.ascontiguousarray(uv, dtype=np.float64).reshape(-1, 2)
        if pts.shape[0] != px.shape[0]:
            raise ValueError("points and uv differ in length")
        ps = np.ascontiguousarray(pose, dtype=np.float64).reshape(7).copy()
        mask = None if use is None else np.ascontiguousarray(use, dtype=np.uint8)
        summ = CSummary()
        _check(self._lib.hipba_refine_pose(self._h, K[0], K[1], K[2], K[3], camera_model, pts.shape[0],
                                           pts.ctypes.data_as(_dp), px.ctypes.data_as(_dp),
                                           None if mask is None else mask.ctypes.data_as(_bp), ps.ctypes.data_as(_dp),
                                           C.byref(o), C.byref(summ)))
        return ps, _summary_dict(summ)

    def pnp_ransac(self, K, points, uv, pose=None, iterations=1000, reprojection_error=3.0, seed=0, refine=True,
                   camera_model=CAMERA_Z_FORWARD, confidence=0.99, return_iterations=False):
        """hipba_pnp_ransac (cv::solvePnPRansac as estimator.cpp:28-29 calls it; pose=None: no prior, minimal samples
        in closed form; a pose: samples fitted from it): (pose, inlier mask[, iterations the sequential loop runs]).
        Without a consensus the pose comes back as it went in (zeros when there was none)."""
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        px = np.ascontiguousarray(uv, dtype=np.float64).reshape(-1, 2)
        if pts.shape[0] != px.shape[0]:
            raise ValueError("points and uv differ in length")
        ps = np.zeros(7) if pose is None else np.ascontiguousarray(pose, dtype=np.float64).reshape(7).copy()
        mask = np.zeros(pts.shape[0], dtype=np.uint8)
        cnt = C.c_int64(0)
        run = C.c_int32(0)
        _check(self._lib.hipba_pnp_ransac(self._h, K[0], K[1], K[2], K[3], camera_model, pts.shape[0],
                                          pts.ctypes.data_as(_dp), px.ctypes.data_as(_dp), ps.ctypes.data_as(_dp),
                                          int(iterations), float(reprojection_error), float(confidence), int(seed),
                                          int(bool(refine)), mask.ctypes.data_as(_bp), C.byref(cnt), C.byref(run)))
        assert cnt.value == int(mask.sum())
        return (ps, mask, run.value) if return_iterations else (ps, mask)

    def upload(self, problems):
        if isinstance(problems, Problem):
            problems = [problems]
        return DeviceProblem(self, problems)

    def upload_shard(self, shard, pose_order=None):
        return DeviceProblem(self, [shard], shard=True, pose_order=pose_order)


def pose_order(problem):
    """hipba_pose_order: the order of the poses in the reduced camera system hipba_upload would choose for this
    (whole) problem; hand it to every rank's upload_shard."""
    out = np.zeros(problem.poses.shape[0], dtype=np.int32)
    cp = problem.c()
    _check(load_library().hipba_pose_order(C.byref(cp), out.ctypes.data_as(_ip)))
    return out
