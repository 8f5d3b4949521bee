"""One rank of a sharded solve on a one-GPU box: N PROCESSES on device 0, the engine's all-reduces carried by gloo on
CUDA tensors through the same TorchComm callback a torch.distributed job uses.  Launched with torch.distributed.run by
tests/test_sharded_processes.py (and by hand: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1
tests/sharded_rank_main.py [n_points] [band]); rank 0 prints the comparison with the same window solved whole."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from stereo_visual_odometry_amd import hipba, scenes
from stereo_visual_odometry_amd.sharded import ShardedProblem, TorchComm

def main():
    import faulthandler
    faulthandler.dump_traceback_later(120, exit=True)  # a mismatch between the ranks' collectives shows as a hang
    rank = int(os.environ["RANK"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    print("rank", rank, "up", flush=True)
    solver = hipba.Solver(0)
    n_points = int(sys.argv[1]) if len(sys.argv) > 1 else 40000  # 160000: every rank in the gathered elimination mode
    if len(sys.argv) > 2 and sys.argv[2] == "band":
        # a band of tiles: gathered elimination into the dense matrix, tile exchange (pipelined unless HIPBA_EXCHANGE_CHUNKS=1)
        prob, _, _ = scenes.bal_scale(seed=37, grid=(60, 4), n_points=n_points, depth_lo=5.0, depth_hi=15.0)
    else:
        prob, _, _ = scenes.bal_scale(seed=5, grid=(8, 10), n_points=n_points, depth_lo=5.0, depth_hi=30.0)
    comm = TorchComm(torch.device("cuda", 0))
    job = ShardedProblem(solver, prob, comm)
    opts = hipba.default_options(max_num_iterations=5)
    t0 = time.perf_counter()
    s = job.solve(opts)
    print("rank", rank, "solved in %.1f ms, %d all-reduces, %.1f MB" % (1e3 * (time.perf_counter() - t0), comm.calls, comm.bytes / 1e6), flush=True)
    pts = job.gather_points()
    poses, _ = job.download()
    if rank == 0:
        ref = prob.copy()
        s_ref = hipba.Solver(0).solve(ref, opts)
        print("final cost sharded %.12e single %.12e; poses differ by %.2e, points by %.2e; iterations %d vs %d" % (
            s["final_cost"], s_ref["final_cost"], np.abs(poses - ref.poses).max(), np.abs(pts - ref.points).max(),
            s["num_iterations"], s_ref["num_iterations"]), flush=True)
    dist.barrier()
    dist.destroy_process_group()

if __name__ == "__main__":
    main()
