"""Observation-sharded solves over several GPUs of one node: one process per GPU, the engine's
per-iteration sums completed with ``torch.distributed`` all-reduces (RCCL over xGMI when the
process group's backend is "nccl"; gloo on CPU tensors in tests).

The reference has no counterpart (it is a single-process CPU program); the sharding rule is the
one a bundle-adjustment problem offers naturally: every rank holds all poses and a contiguous block
of the landmarks with all of their observations, so landmark elimination is rank-local and only the
reduced camera matrix, its right-hand side and a handful of scalars are summed.
"""
import ctypes as C
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from . import hipba
from .scenes import shard_by_point


class _DevicePointer:
    """Exposes a raw device address through __cuda_array_interface__ so that torch can alias it."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def tensor_from_pointer(ptr, count, device):
    """A float64 torch tensor aliasing `count` doubles at `ptr` (device memory, or host memory for 'cpu')."""
    if torch.device(device).type == "cpu":
        arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(int(count),))
        return torch.from_numpy(arr)
    return torch.as_tensor(_DevicePointer(ptr, count), device=device)


class TorchComm:
    """hipba_comm whose all-reduce is torch.distributed.all_reduce on the engine's own buffers.

    The collective is issued with the engine's stream made current (torch.cuda.ExternalStream), so it
    is ordered after the kernels already queued there and before the ones queued next -- no host
    synchronisation per call."""

    def __init__(self, device, group=None):
        self.device = torch.device(device)
        self.group = group
        self.rank = dist.get_rank(group)
        self.world_size = dist.get_world_size(group)
        self.calls = 0
        self.bytes = 0
        # HIPBA_COMM_HOST_SYNC: the host waits for the stream before and after every collective (rehearsals of several
        # ranks on ONE device through gloo, whose staging streams otherwise interleave with the engine's two)
        self.host_sync = bool(os.environ.get("HIPBA_COMM_HOST_SYNC"))
        self._cb = hipba.ALLREDUCE_FN(self._allreduce)  # keep a reference: ctypes callbacks are not owned by C
        self.c = hipba.CComm(self._cb, None, self.rank, self.world_size)

    def _allreduce(self, ctx, buf, count, op, stream):
        try:
            if os.environ.get("HIPBA_TRACE_COMM"):
                print("%.3f rank %d all-reduce #%d count %d op %d" % (time.time() % 1000.0, self.rank, self.calls, count, op), flush=True)
            t = tensor_from_pointer(buf, count, self.device)
            rop = dist.ReduceOp.SUM if op == hipba.REDUCE_SUM else dist.ReduceOp.MAX
            if self.device.type == "cuda":
                ext = torch.cuda.ExternalStream(int(stream), device=self.device)
                with torch.cuda.stream(ext):
                    if self.host_sync:
                        ext.synchronize()
                    dist.all_reduce(t, op=rop, group=self.group)
                    if self.host_sync:
                        ext.synchronize()
            else:
                dist.all_reduce(t, op=rop, group=self.group)
            self.calls += 1
            self.bytes += 8 * int(count)
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            print("TorchComm all-reduce failed:", repr(e))
            return 1


class ShardedProblem:
    """One window split over the ranks of the default process group (landmark-aligned shards)."""

    def __init__(self, solver, problem, comm):
        self.solver = solver
        self.comm = comm
        self.full = problem
        self.shard, (self.lo, self.hi) = shard_by_point(problem, comm.rank, comm.world_size)
        # the order of the poses in the reduced system is chosen on the whole problem (every rank has it here and
        # computes the same order); a shard alone only knows its own landmarks' couplings
        self.dp = solver.upload_shard(self.shard, hipba.pose_order(problem))

    def reset(self):
        self.dp.reset()

    def solve(self, options=None):
        return self.dp.solve_sharded(self.comm.c, options)

    def download(self):
        """(poses, this rank's landmark block)"""
        return self.dp.download(0)

    def gather_points(self):
        """All landmarks on every rank (for checks; not part of the timed path)."""
        _, pts = self.download()
        parts = [None] * self.comm.world_size
        dist.all_gather_object(parts, (self.lo, pts), group=self.comm.group)
        out = np.zeros_like(self.full.points)
        for lo, p in parts:
            out[lo:lo + p.shape[0]] = p
        return out
