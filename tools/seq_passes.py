"""Per-call times of the sequence solve (32 pairs, 32 data seeds, one context, 3 lanes): how much the calls differ and how many RANSAC runs
started over / how the refinement's iteration counts fall in each.  usage (GPU box): python tools/seq_passes.py [calls]"""
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("rs-aware-differential-sfm_amd")

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 12
pairs = 32
flows, seq = pkg.synth.make_flow_sequence(5, [0x5EED0005 + i for i in range(pairs)])
rows, cols = seq["rows"], seq["cols"]
imgs = [torch.from_numpy(f).cuda() for f in flows]
dms = [torch.empty((cols, rows), dtype=torch.float64, device="cuda") for _ in range(pairs)]
with pkg.Solver(0) as s:
    jobs = [dict(d_flow_img=im.data_ptr(), rows=rows, cols=cols, K=seq["K"], gamma=seq["gamma"], d_depth_map=dm.data_ptr()) for im, dm in zip(imgs, dms)]
    call = s.prepared_frames_solve(jobs, trials=50, tol=0.05)
    call([1 + i for i in range(pairs)])
    for c in range(calls):
        seeds = [1 + pairs * (c + 1) + i for i in range(pairs)]
        t0 = time.perf_counter()
        rs = call(seeds)
        dt = (time.perf_counter() - t0) * 1e3
        its = np.array([int(r.refine_summary.num_iterations) for r in rs])
        print("call %2d: %.3f ms per pair; refinement iterations %s; restarts so far %d" % (c, dt / pairs, dict(zip(*np.unique(its, return_counts=True))), s.ransac_restarts()))
