#!/usr/bin/env python3
"""Duration of the analytic pixel pass (ransac_lma_kernel) inside ordinary whole solves, bracketed by the library's own events
(rsdsfm_set_profiling), and the solve's median time:  python tools/lma_time.py [solves]
    python tools/lma_time.py sweep [solves]   -- the pass over T in {5, 20, 50, 64, 86, 100, 130, 200, 256} (hypothesis groups keep >= 90 % of the lanes
                                                 busy for every T: ransac_lma_group_size), against T / 50 x the T = 50 time"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import rsdsfm  # noqa: E402

SWEEP = len(sys.argv) > 1 and sys.argv[1] == "sweep"
if SWEEP:
    del sys.argv[1]
solves = int(sys.argv[1]) if len(sys.argv) > 1 else 60


def group_size(T):
    """ransac_lma_group_size (csrc/ransac_lma_kernels.hip) restated: (hypotheses per group, groups)"""
    best_ng, best_u = 1, 0.0
    for ng in range(1, 9):
        if ng > T:
            break
        tg = -(-T // ng)
        u = T * (256 // tg) / (-(-T // tg) * 256)
        if u > best_u + 1e-9:
            best_u, best_ng = u, ng
        if u >= 0.9:
            best_ng = ng
            break
    tg = -(-T // best_ng)
    return tg, -(-T // tg)


dev = torch.device("cuda", 0)
d = rsdsfm.synth.make_config(5, rows=720, cols=1280, seed=1)
rows, cols, K, gamma = d["rows"], d["cols"], d["K"], d["gamma"]
img = torch.from_numpy(d["flow_img"]).to(dev)
dm = torch.empty((cols, rows), dtype=torch.float64, device=dev)
with rsdsfm.Solver(0) as s:
    base = None
    batch128 = 0.0
    for T, tol in (((5, 0.05), (20, 0.05), (50, 0.05), (64, 0.05), (86, 0.05), (100, 0.05), (128, 0.05), (130, 0.05), (200, 0.05), (256, 0.05)) if SWEEP else ((50, 0.05), (50, 0.002), (5, 0.05))):
        s.set_profiling(True)
        ks, mhz, ts = [], [], []
        for i in range(solves):
            t0 = time.perf_counter()
            s.solve_frame_dev(img.data_ptr(), rows, cols, K, gamma, dm.data_ptr(), trials=T, tol=tol, seed=1 + i)
            ts.append((time.perf_counter() - t0) * 1e3)
            ks.append(s.profile_last_ms("ransac_lm_round0"))
            mhz.append(s.profile_last_ms("ransac_lm_round0_clock_mhz"))
        s.set_profiling(False)
        if T > 128:
            # a RANSAC of more than 128 trials runs in hypothesis batches of 128 (kRansacBatch); the profiling record brackets the LAST batch's pass:
            # the full batches cost what the T = 128 row above measured
            full128 = batch128 * (T // 128 if T % 128 else T // 128 - 1)
            ks = [k + full128 * 1e-3 for k in ks]
        med = 1e3 * np.median(ks[5:])
        if T == 50 and base is None:
            base = med
        if T == 128:
            batch128 = med
        tg, groups = group_size(min(T, 128))  # (T > 128: hypothesis batches of 128 + the rest; the groups of a full batch)
        extra = "" if not SWEEP else "  groups %d x %d hypotheses, lanes busy %.1f %%%s" % (groups, tg, 100.0 * min(T, 128) * (256 // tg) / (groups * 256), "" if base is None or T < 50 else ", %.2f x (T / 50 x the T = 50 pass)" % (med / (base * T / 50.0)))
        print("T %3d tol %.3f: pixel pass %.1f us (min %.1f) at %.0f MHz; solve median %.3f ms; lma restarts %s%s" % (
            T, tol, med, 1e3 * min(ks[5:]), np.mean(mhz[5:]), np.median(ts[5:]), s.lma_restarts(), extra))
