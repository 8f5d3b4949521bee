"""CPU-only, world_size 2 over gloo: the multi-rank plumbing of stereo_visual_odometry_amd.sharded --
landmark-aligned shards plus the all-reduce callback the engine calls -- checked with the oracle as
the per-rank arithmetic (no GPU kernel runs here)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ctypes as C
        from oracle import oracle
        from stereo_visual_odometry_amd import hipba, scenes
        from stereo_visual_odometry_amd.sharded import TorchComm

        prob, _, _ = scenes.bal_scale(seed=9, grid=(4, 5), n_points=600, depth_lo=5.0, depth_hi=20.0)
        shard, (lo, hi) = scenes.shard_by_point(prob, rank, world)
        comm = TorchComm("cpu")
        assert comm.c.rank == rank and comm.c.world_size == world

        def grad_and_cost(p):
            o = oracle.Problem(p.poses, p.points, p.obs_pose, p.obs_point, p.obs_uv, p.K, p.camera_model)
            ev = oracle.evaluate(o, oracle.JAC_ANALYTIC, apply_loss=False)
            g = np.zeros((p.poses.shape[0], 6))
            np.add.at(g, p.obs_pose, np.einsum("nrc,nr->nc", ev["jac_pose"], ev["residuals"]))
            return g, ev["cost"]

        g_local, c_local = grad_and_cost(shard)
        buf = np.concatenate([g_local.ravel(), [c_local]])
        # drive the callback exactly as the engine does: raw pointer, count, op, stream
        rc = comm._cb(None, buf.ctypes.data_as(C.c_void_p), buf.size, hipba.REDUCE_SUM, None)
        assert rc == 0
        g_full, c_full = grad_and_cost(prob)
        ok_sum = np.allclose(buf[:-1].reshape(-1, 6), g_full, rtol=1e-10, atol=1e-6) and abs(buf[-1] - c_full) <= 1e-10 * c_full
        mx = np.array([float(rank + 1), -float(rank)])
        rc = comm._cb(None, mx.ctypes.data_as(C.c_void_p), mx.size, hipba.REDUCE_MAX, None)
        ok_max = rc == 0 and mx.tolist() == [float(world), 0.0]
        ret[rank] = (bool(ok_sum), bool(ok_max), comm.calls, hi - lo)
    finally:
        dist.destroy_process_group()


def test_two_rank_shards_and_allreduce_callback_over_gloo():
    world = 2
    port = 29500 + (os.getpid() % 2000)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
        assert len(ret) == world
        for r in range(world):
            ok_sum, ok_max, calls, n_local = ret[r]
            assert ok_sum and ok_max and calls == 2 and n_local == 300
