"""hipba_evaluate_device -- the entry point that allocates nothing -- against hipba_evaluate.

The caller owns every device array (allocated here with plain HIP calls through ctypes, before the calls under
test); the library's only extra buffer is a fixed array inside its code object.  Two calls in a row on the same
GPU, the second with a smaller input, must both reproduce hipba_evaluate: bit for bit on residuals and Jacobians
(the same kernel computes them), to 1e-12 relative on the cost (the same fixed-order sum).
"""
import ctypes as C

import numpy as np
import pytest

from stereo_visual_odometry_amd import hipba, scenes

pytestmark = pytest.mark.gpu


class _Hip:
    def __init__(self):
        self.lib = C.CDLL("libamdhip64.so")
        self.lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.lib.hipFree.argtypes = [C.c_void_p]
        self.lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.lib.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.lib.hipStreamSynchronize.argtypes = [C.c_void_p]
        self.live = []

    def upload(self, a):
        a = np.ascontiguousarray(a)
        p = C.c_void_p()
        assert self.lib.hipMalloc(C.byref(p), max(a.nbytes, 8)) == 0
        assert self.lib.hipMemcpy(p, a.ctypes.data, a.nbytes, 1) == 0
        self.live.append(p)
        return p.value

    def empty(self, nbytes):
        p = C.c_void_p()
        assert self.lib.hipMalloc(C.byref(p), max(nbytes, 8)) == 0
        assert self.lib.hipMemset(p, 0xff, max(nbytes, 8)) == 0   # NaN pattern: stale words would show
        self.live.append(p)
        return p.value

    def download(self, ptr, shape, dtype=np.float64):
        out = np.empty(shape, dtype=dtype)
        assert self.lib.hipMemcpy(out.ctypes.data, ptr, out.nbytes, 2) == 0
        return out

    def free_all(self):
        for p in self.live:
            self.lib.hipFree(p)
        self.live = []


def _run(solver, hip, prob, opts, apply_loss):
    n = prob.obs_pose.shape[0]
    d = dict(poses=hip.upload(prob.poses), points=hip.upload(prob.points), op=hip.upload(prob.obs_pose),
             oq=hip.upload(prob.obs_point), uv=hip.upload(prob.obs_uv), cost=hip.empty(8), res=hip.empty(16 * n),
             jc=hip.empty(96 * n), jp=hip.empty(48 * n))
    # everything above is the caller's allocation; the call under test:
    solver.evaluate_device(prob.poses.shape[0], d["poses"], prob.points.shape[0], d["points"], n, d["op"], d["oq"],
                           d["uv"], prob.K, prob.camera_model, opts, apply_loss, d["cost"], d["res"], d["jc"], d["jp"])
    assert hip.lib.hipStreamSynchronize(solver.stream) == 0
    return {"cost": float(hip.download(d["cost"], (1,))[0]), "residuals": hip.download(d["res"], (n, 2)),
            "jac_pose": hip.download(d["jc"], (n, 2, 6)), "jac_point": hip.download(d["jp"], (n, 2, 3))}


def _same(got, want):
    np.testing.assert_array_equal(got["residuals"], want["residuals"])
    np.testing.assert_array_equal(got["jac_pose"], want["jac_pose"])
    np.testing.assert_array_equal(got["jac_point"], want["jac_point"])
    assert abs(got["cost"] - want["cost"]) <= 1e-12 * abs(want["cost"])


def test_evaluate_device_matches_evaluate_twice_in_a_row(solver):
    hip = _Hip()
    try:
        big, _, _ = scenes.kitti_like_window(1000)                       # ~25k observations: ~100 partial sums
        small, _, _ = scenes.small_window(2000, n_frames=4, n_points=60)  # a few hundred: 1-2 partial sums
        opts = hipba.default_options(loss=hipba.LOSS_HUBER, loss_scale=1.0)
        for prob in (big, small):                                          # second call: smaller input, same scratch
            want = solver.evaluate(prob, opts, apply_loss=True)
            got = _run(solver, hip, prob, opts, True)
            _same(got, want)
        # size limit: rejected with a clear message, before anything is queued
        with pytest.raises(hipba.HipbaError) as ex:
            solver.evaluate_device(1, 1, 1, 1, hipba.EVALUATE_DEVICE_MAX_OBS + 1, 1, 1, 1, scenes.KITTI_K)
        assert ex.value.code == hipba.ERR_UNSUPPORTED and "at most" in str(ex.value)
        # an out-of-range index is caught on the device: NaN for that observation, nothing read out of bounds
        bad = small.copy()
        bad.obs_point[3] = bad.points.shape[0] + 5
        got = _run(solver, hip, bad, opts, True)
        assert np.isnan(got["residuals"][3]).all() and np.isnan(got["cost"])
        assert np.isfinite(np.delete(got["residuals"], 3, axis=0)).all()
    finally:
        hip.free_all()
