#!/usr/bin/env python3
"""Randomised campaign on the one-call FRAME solve (rsdsfm_solve_frame_dev: flatten -> RANSAC -> refinement -> depth map), GPU only:
the three forms of the RANSAC's depth solves must return the same frame, bit for bit --
  mode 1  iterate by iterate (the reference's arithmetic: what tests/fuzz_gpu.py pins against the oracle, case by case),
  mode 0  the analytic LM trajectory, fused error sums or count-only as the context's previous solves say,
  mode 2  the analytic trajectory with the COUNT-ONLY pass forced (error sums fetched lazily for the trials that share the best count).
Random small frames (DeepFlow-like and noise-free), random motions, trial counts, tolerances from selective to permissive (every pixel an
inlier: ties in the count on every solve), acceleration mode, with and without refinement.
    python tests/fuzz_frames.py [cases] [seed]        (run through gpurun; exit code 1 on a mismatch)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    import rsdsfm

    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    dev = torch.device("cuda", 0)
    bad = lazy = count_only = 0
    with rsdsfm.Solver(0) as s:
        s.set_refine_arithmetic(1)  # (the forms of the DEPTH SOLVES are compared bit for bit: one refinement arithmetic behind all of them; tests/fuzz_gpu.py fuzzes the refinement's default)
        for c in range(cases):
            rng = np.random.default_rng(seed0 * 1000003 + c)
            rows, cols = int(rng.integers(12, 120)), int(rng.integers(12, 160))
            cfg = int(rng.choice([1, 3, 3]))
            v = rng.normal(size=3) * np.array([0.03, 0.03, 0.02])
            w = rng.normal(size=3) * 0.004
            k = float(rng.choice([0.0, 0.0, rng.uniform(-0.5, 0.8)]))
            d = rsdsfm.synth.make_config(cfg, seed=int(rng.integers(1 << 30)), v=v, w=w, k=k, rows=rows, cols=cols)
            if not np.all(np.isfinite(d["flow_img"])) or len(d["q"]) < 9:
                continue
            K, gamma = d["K"], d["gamma"]
            img = torch.from_numpy(d["flow_img"]).to(dev)
            kw = dict(trials=int(rng.choice([1, 3, 8, 20, 50])), tol=float(rng.choice([1.0, 0.2, 0.05, 0.01, 0.003, 0.001])), seed=int(rng.integers(1 << 30)),
                      use_acceleration_mode=bool(rng.integers(2)) and k != 0.0, use_refinement=bool(rng.integers(4)))
            tag = "frame case %d (%dx%d cfg %d k %.3f %s)" % (c, rows, cols, cfg, k, kw)
            outs = []
            try:
                for mode in (1, 2, 2, 0):
                    s.set_lm_arithmetic(mode)
                    r0 = s.lma_count_only()
                    dm = torch.zeros((cols, rows), dtype=torch.float64, device=dev)
                    R = torch.zeros((rows, 9), dtype=torch.float64, device=dev)
                    t = torch.zeros((rows, 3), dtype=torch.float64, device=dev)
                    r = s.solve_frame_dev(img.data_ptr(), rows, cols, K, gamma, dm.data_ptr(), R.data_ptr(), t.data_ptr(), **kw)
                    s.synchronize()
                    r1 = s.lma_count_only()
                    count_only += r1[0] - r0[0]
                    lazy += r1[1] - r0[1]
                    outs.append((r, dm.cpu().numpy(), R.cpu().numpy(), t.cpu().numpy()))
                x = outs[0]
                for mode, o in zip((2, 2, 0), outs[1:]):
                    assert o[0]["num_inliers"] == x[0]["num_inliers"] and o[0]["best_trial"] == x[0]["best_trial"], ("winner", mode, o[0]["best_trial"], x[0]["best_trial"], o[0]["num_inliers"], x[0]["num_inliers"])
                    assert o[0]["refine_summary"] == x[0]["refine_summary"], ("refinement", mode)
                    for key in ("v", "w", "k", "ransac_v", "ransac_w", "ransac_k", "flipped"):
                        assert np.array_equal(np.asarray(o[0][key]), np.asarray(x[0][key]), equal_nan=True), (key, mode)
                    for j, what in ((1, "depth map"), (2, "R"), (3, "t")):
                        assert np.array_equal(o[j], x[j], equal_nan=True), (what, mode)
            except AssertionError as e:
                bad += 1
                print("MISMATCH", tag, e.args[0] if e.args else "", flush=True)
            except rsdsfm.RsdsfmError as e:
                # (an error must be every form's: a frame the reference could not solve either, e.g. fewer than 9 points with flow)
                errs = []
                for mode in (1, 2):
                    s.set_lm_arithmetic(mode)
                    try:
                        dm = torch.zeros((cols, rows), dtype=torch.float64, device=dev)
                        s.solve_frame_dev(img.data_ptr(), rows, cols, K, gamma, dm.data_ptr(), **kw)
                        errs.append(None)
                    except rsdsfm.RsdsfmError as e2:
                        errs.append(str(e2))
                if errs[0] is None or errs[1] is None:
                    bad += 1
                    print("ERROR", tag, e, errs, flush=True)
        s.set_lm_arithmetic(0)
        print("analytic LM trajectory: %d RANSAC runs started over (last guards: bit set %d)" % s.lma_restarts())
    print("fuzz_frames: %d cases, %d mismatches; %d RANSACs ran the count-only pass, %d of them fetched error sums for trials sharing the best count" % (cases, bad, count_only, lazy))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
