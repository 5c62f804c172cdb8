#!/usr/bin/env python3
"""Duration of the analytic pixel pass (ransac_lma_kernel) inside ordinary whole solves, bracketed by the library's own events
(rsdsfm_set_profiling), and the solve's median time:  python tools/lma_time.py [solves]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import rsdsfm  # noqa: E402

solves = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda", 0)
d = rsdsfm.synth.make_config(5, rows=720, cols=1280, seed=1)
rows, cols, K, gamma = d["rows"], d["cols"], d["K"], d["gamma"]
img = torch.from_numpy(d["flow_img"]).to(dev)
dm = torch.empty((cols, rows), dtype=torch.float64, device=dev)
with rsdsfm.Solver(0) as s:
    for T, tol in ((50, 0.05), (50, 0.002), (5, 0.05)):
        s.set_profiling(True)
        ks, mhz, ts = [], [], []
        for i in range(solves):
            t0 = time.perf_counter()
            s.solve_frame_dev(img.data_ptr(), rows, cols, K, gamma, dm.data_ptr(), trials=T, tol=tol, seed=1 + i)
            ts.append((time.perf_counter() - t0) * 1e3)
            ks.append(s.profile_last_ms("ransac_lm_round0"))
            mhz.append(s.profile_last_ms("ransac_lm_round0_clock_mhz"))
        s.set_profiling(False)
        print("T %3d tol %.3f: pixel pass %.1f us (min %.1f) at %.0f MHz; solve median %.3f ms; lma restarts %s" % (
            T, tol, 1e3 * np.median(ks[5:]), 1e3 * min(ks[5:]), np.mean(mhz[5:]), np.median(ts[5:]), s.lma_restarts()))
