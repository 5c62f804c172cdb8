"""Helper of tests/test_gpu_tiled_frame.py (not a test): one rank of a 2-process row-tiled frame solve.  Both ranks use
cuda:0 (the GPU box has one device) and a gloo group; with one GPU per rank and backend "nccl" the code path is the same."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import torch.distributed as dist

    import rsdsfm

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    with torch.cuda.stream(stream):
        d = rsdsfm.synth.make_config(3, rows=96, cols=250)
        rows, cols, K, gamma = d["rows"], d["cols"], d["K"], d["gamma"]
        bounds, per = rsdsfm.dist.slab_bounds(cols, world)
        c0, c1 = bounds[rank]
        solver = rsdsfm.Solver(0, stream=stream.cuda_stream)
        slab = torch.from_numpy(np.ascontiguousarray(d["flow_img"][:, c0:c1, :])).to(dev)  # each rank only holds its slab
        shard = rsdsfm.dist.HipFrameShard(solver, slab, c0, K, gamma, torch)
        drv = rsdsfm.dist.TiledFrameSolve([shard], rows, cols, per, torch, dist)
        r = drv.solve(trials=14, tol=0.002, seed=7, flow_index_mode=int(os.environ.get("RSDSFM_TEST_FLOW_MODE", "1")))
        torch.cuda.synchronize()
        dm = r["depth_map"].cpu().numpy()
        mine = np.concatenate([r["v"], r["w"], [r["k"], r["num_inliers"], r["best_trial"], float((dm != 0).sum()), dm.sum()]])
        t = torch.from_numpy(mine).clone()
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        agree = all(torch.equal(o, outs[0]) for o in outs)
        if rank == 0:
            with open(os.environ["RSDSFM_TILED_OUT"], "w") as f:
                json.dump(dict(world=world, n=r["n"], num_inliers=r["num_inliers"], best_trial=r["best_trial"], v=list(r["v"]), w=list(r["w"]),
                               k=r["k"], depth_nonzero=int((dm != 0).sum()), depth_sum=float(dm.sum()), ranks_agree=bool(agree)), f)
        solver.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
