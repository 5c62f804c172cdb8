"""Helper of tests/test_gpu_tiled_native.py (not a test): one rank of a 2-process run of the NATIVE column-tiled solve
(rsdsfm_solve_frame_tiled_dev).  Both ranks use cuda:0 (the GPU box has one device, RCCL refuses two ranks on one GPU), so the
collectives go through tests/transports.GlooTransport; with one GPU per rank rsdsfm_dist_init (RCCL) replaces it, same call."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist

    import rsdsfm
    from transports import GlooTransport

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", 0)
    d = rsdsfm.synth.make_config(3, rows=96, cols=250)
    rows, cols, K, gamma = d["rows"], d["cols"], d["K"], d["gamma"]
    c0, sc, per = rsdsfm.tiled_slab_bounds(cols, world, rank)
    solver = rsdsfm.Solver(0)
    solver.dist_set_transport(world, rank, *GlooTransport(dist, torch).callbacks())
    slab = torch.from_numpy(np.ascontiguousarray(d["flow_img"][:, c0:c0 + sc, :])).to(dev)  # each rank only holds its slab
    dm = torch.zeros(cols * rows, dtype=torch.float64, device=dev)
    r = solver.solve_frame_tiled_dev(slab.data_ptr(), rows, cols, K, gamma, dm.data_ptr(), trials=14, tol=0.002, seed=7, flow_index_mode=int(os.environ.get("RSDSFM_TEST_FLOW_MODE", "1")))
    torch.cuda.synchronize()
    dmh = dm.cpu().numpy()
    mine = np.concatenate([r["v"], r["w"], [r["k"], r["num_inliers"], r["best_trial"], float((dmh != 0).sum()), dmh.sum()]])
    t = torch.from_numpy(mine).clone()
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    agree = all(torch.equal(o, outs[0]) for o in outs)
    if rank == 0:
        with open(os.environ["RSDSFM_TILED_OUT"], "w") as f:
            json.dump(dict(world=world, n=r["n"], num_inliers=r["num_inliers"], best_trial=r["best_trial"], v=list(r["v"]), w=list(r["w"]),
                           k=r["k"], depth_nonzero=int((dmh != 0).sum()), depth_sum=float(dmh.sum()), ranks_agree=bool(agree), info=r["info"]), f)
    solver.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
