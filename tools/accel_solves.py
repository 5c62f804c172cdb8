#!/usr/bin/env python3
"""Whole solves in ACCELERATION mode (main.cc:306: k estimated by the minimal solver and refined) on the bench's 1280x720 pair -- for a kernel trace
of that regime (rocprofv3 --kernel-trace --stats -- python3 tools/accel_solves.py [solves]) and its host times / refinement statistics."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rsdsfm  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
d = rsdsfm.synth.make_config(5, seed=0x5EED0005)
rows, cols = d["rows"], d["cols"]
img = torch.from_numpy(d["flow_img"]).cuda()
dm = torch.empty((cols, rows), dtype=torch.float64, device="cuda")
with rsdsfm.Solver(0) as s:
    call = s.prepared_frame_solve(img.data_ptr(), rows, cols, d["K"], d["gamma"], dm.data_ptr(), None, None, trials=50, tol=0.05, use_acceleration_mode=True)
    ts, its = [], []
    for i in range(n + 5):
        t0 = time.perf_counter()
        r = call(1 + i)
        if i >= 5:
            ts.append((time.perf_counter() - t0) * 1e3)
            its.append(int(r.refine_summary.num_iterations))
    print("acceleration mode, %d solves: median %.3f ms, mean %.3f, min %.3f, max %.3f; LM iterations min / median / max %d / %d / %d; refinement %s; lma restarts %s" % (
        n, np.median(ts), np.mean(ts), min(ts), max(ts), min(its), int(np.median(its)), max(its), s.refine_restarts(), s.lma_restarts()))
