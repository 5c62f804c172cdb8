"""How many refinement slots the whole solve of DeepFlow-like pairs consumes: (LM iterations, steps whose speculated radius did not apply) over
12 data seeds x 4 sampler seeds, with the relative decreases of the first solves (usage, GPU box: python tools/refine_slots.py)."""
import collections, os, sys
import numpy as np, torch
sys.path.insert(0, ".")
import rsdsfm
dev = torch.device("cuda", 0)
ACCEL = bool(int(os.environ.get("SLOTS_ACCEL", "0")))  # SLOTS_ACCEL=1: acceleration mode (k estimated and refined, 7 parameters)
hist = collections.Counter(); rels = []
with rsdsfm.Solver(0) as s:
    s.set_refine_trace(50)
    for sd in range(12):
        frames, meta = rsdsfm.synth.make_flow_sequence(5, [0x5EED0005 + 1000 * sd])
        rows, cols = meta["rows"], meta["cols"]
        img = torch.from_numpy(frames[0]).to(dev)
        dm = torch.empty((cols, rows), dtype=torch.float64, device=dev)
        for seed in range(4):
            r = s.solve_frame_dev(img.data_ptr(), rows, cols, meta["K"], meta["gamma"], dm.data_ptr(), trials=50, tol=0.05, seed=1 + seed, use_acceleration_mode=ACCEL)
            tr = s.get_refine_trace()
            it = r["refine_summary"]["num_iterations"]
            valid = tr[~np.isnan(tr[:, 0])]
            acc = valid[valid[:, 7] == 1.0]
            miss = int((acc[:, 4] < 0.9375).sum()) + int((valid[:, 7] == 0.0).sum()) + int((valid[:, 7] == 2.0).sum())
            hist[(len(valid), miss)] += 1
            rels.append([round(x, 3) for x in valid[:, 4]])
print("(trace rows, speculation misses) -> solves:", sorted(hist.items()))
for r in rels[:12]: print(r)
