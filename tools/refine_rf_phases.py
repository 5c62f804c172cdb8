#!/usr/bin/env python3
"""Where the time of the radius-factorised refinement's pass goes (csrc/refine_rf_kernels.hip), from the opt-in phase stamps of workgroup 0:
state load / stage in the prologue / loop over the inliers / row reduction, per pass, over whole solves of DeepFlow-like 1280x720 pairs.
    RSDSFM_RF_STAMPS=1 python tools/refine_rf_phases.py          (SLOTS_ACCEL=1: acceleration mode)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rsdsfm  # noqa: E402

os.environ.setdefault("RSDSFM_RF_STAMPS", "1")
dev = torch.device("cuda", 0)
ACCEL = bool(int(os.environ.get("SLOTS_ACCEL", "0")))
with rsdsfm.Solver(0) as s:
    frames, meta = rsdsfm.synth.make_flow_sequence(5, [0x5EED0005 + 1000 * sd for sd in range(4)])
    rows, cols = meta["rows"], meta["cols"]
    imgs = [torch.from_numpy(f).to(dev) for f in frames]
    dm = torch.empty((cols, rows), dtype=torch.float64, device=dev)
    its = []
    for rep in range(6):
        for i, img in enumerate(imgs):
            r = s.solve_frame_dev(img.data_ptr(), rows, cols, meta["K"], meta["gamma"], dm.data_ptr(), trials=50, tol=0.05, seed=1 + rep, use_acceleration_mode=ACCEL)
            if rep == 0:
                s.profile_last_ms("refine_rf_phase0")  # (zeroes the stamps behind the warm-up solves)
            else:
                its.append(r["refine_summary"]["num_iterations"])
    ph = [s.profile_last_ms("refine_rf_phase%s" % k) for k in "0123456789abcdef"]
    n = max(ph[4], 1.0)
    print("passes stamped %d over %d solves (LM iterations %s); restarts %s" % (ph[4], len(its), sorted(set(its)), s.refine_restarts()))
    print("per pass that ran its loop: state load %.2f us, stage %.2f us, loop %.2f us, row reduction %.2f us" % tuple(p / n for p in ph[:4]))
    print("   loop end of waves 1 .. 7 behind wave 0's: " + " ".join("%.2f" % (p / n) for p in ph[9:16]) + " us")
