"""How many refinement slots (streaming passes over the inliers) the whole solve of DeepFlow-like pairs consumes, from the iteration traces of 12 data
seeds x 4 sampler seeds: on the radius-factorised path (the default: one slot per LM iteration that evaluated a candidate, + the first pass;
checked against the slot counter the tiled driver reports in tests/test_gpu_tiled_native.py) and what the SAME trajectories cost on the
iterate-by-iterate slot kernels (one more slot behind every step that was rejected, invalid, or accepted with another radius than x 3 --
VERDICT r5 item 4).    usage (GPU box): python tools/refine_slots.py        SLOTS_ACCEL=1: acceleration mode (k refined, 7 parameters)"""
import collections, os, sys
import numpy as np, torch
sys.path.insert(0, ".")
import rsdsfm
dev = torch.device("cuda", 0)
ACCEL = bool(int(os.environ.get("SLOTS_ACCEL", "0")))


def slots_rf(rows):
    return 1 + sum(1 for r in rows if not (r[7] == 2.0 and r[3] == 0.0))


def slots_exact(rows, np_params):
    slots, miss_run = 1, 0
    for i, row in enumerate(rows):
        slots += 1
        if i == len(rows) - 1:
            break
        r_spec = min(row[5] / (1.0 / 3.0), 1e16)
        applies = row[7] == 1.0 and rows[i + 1][5] == r_spec
        spec_on = np_params == 6 and miss_run < 2
        miss_run = 0 if applies else min(miss_run + 1, 2)
        if not (applies and spec_on):
            slots += 1
    return slots


hist = collections.Counter()
tot_it = tot_rf = tot_ex = n = 0
with rsdsfm.Solver(0) as s:
    s.set_refine_trace(60)
    for sd in range(12):
        frames, meta = rsdsfm.synth.make_flow_sequence(5, [0x5EED0005 + 1000 * sd])
        rows, cols = meta["rows"], meta["cols"]
        img = torch.from_numpy(frames[0]).to(dev)
        dm = torch.empty((cols, rows), dtype=torch.float64, device=dev)
        for seed in range(4):
            r = s.solve_frame_dev(img.data_ptr(), rows, cols, meta["K"], meta["gamma"], dm.data_ptr(), trials=50, tol=0.05, seed=1 + seed, use_acceleration_mode=ACCEL)
            tr = s.get_refine_trace()
            valid = tr[~np.isnan(tr[:, 0])]
            a, b = slots_rf(valid), slots_exact(valid, 7 if ACCEL else 6)
            hist[(len(valid), a, b)] += 1
            tot_it += len(valid); tot_rf += a; tot_ex += b; n += 1
    rs = s.refine_restarts()
print("%s: %d solves, LM iterations %.2f per solve; slots per solve: radius-factorised %.2f (= iterations with a candidate + 1), iterate-by-iterate rule on the same trajectories %.2f; guards: %s" % (
    "acceleration mode" if ACCEL else "constant velocity", n, tot_it / n, tot_rf / n, tot_ex / n, rs))
print("(LM iterations, slots radius-factorised, slots iterate-by-iterate) -> solves:", sorted(hist.items()))
