#!/usr/bin/env python3
"""GPU probe of the radius-factorised refinement (csrc/refine_rf_kernels.hip): for a few scenes, the default path against the oracle's
reference arithmetic (mode 1) and its restatement (mode 2), the restart counters, and the time of rsdsfm_refine_dev per path.
    python tools/refine_rf_probe.py [rows cols]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as O  # noqa: E402
import rsdsfm  # noqa: E402


def main():
    rows, cols = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (180, 320)
    s = rsdsfm.Solver(0)
    for acc, tol in ((False, 0.002), (True, 0.002), (False, 0.05), (True, 0.05)):
        d = rsdsfm.synth.make_config(3, rows=rows, cols=cols)
        q, u, a, ak = d["q"], d["u"], d["alpha"], d["alpha_k"]
        samples = O.sample_indices(len(q), 20, 77)
        r = s.ransac(q, u, a, ak, acc, 20, tol, samples=samples, depth_mode=1)
        args = (u, r["inliers"], r["alpha"], r["alpha_k"], r["v"], r["w"], r["k"], acc)
        kw = dict(flow_index_mode=1, inlier_idx=r["inlier_idx"])
        before = s.refine_restarts()
        s.set_lm_arithmetic(0)
        g0 = s.non_linear_refinement(*args, **kw)
        after = s.refine_restarts()
        s.set_lm_arithmetic(1)
        g1 = s.non_linear_refinement(*args, **kw)
        s.set_lm_arithmetic(0)
        o1 = O.refine(*args, **kw)
        o2 = O.refine(*args, mode=2, **kw)
        print("acc=%d tol=%g m=%d  rf runs +%d restarts +%d resolves +%d guard %d | oracle2 guard %d resolves %d" % (
            acc, tol, len(r["inliers"]), after["runs"] - before["runs"], after["restarts"] - before["restarts"], after["resolves"] - before["resolves"],
            after["last_guard"], o2["guard"], o2["resolves"]))
        for name, x in (("gpu rf", g0), ("gpu exact", g1), ("oracle 2", o2)):
            sm, smo = x["summary"], o1["summary"]
            same = all(sm[k] == smo[k] for k in ("num_iterations", "num_successful_steps", "num_unsuccessful_steps", "termination"))
            print("   %-9s vs oracle 1: decisions %s (%d it, term %d)  v %.2e  w %.2e  z %.2e  cost %.2e" % (
                name, "same" if same else "DIFFER", sm["num_iterations"], sm["termination"], np.max(np.abs(x["v"] - o1["v"])), np.max(np.abs(x["w"] - o1["w"])),
                np.max(np.abs(x["inliers"][:, 2] / o1["inliers"][:, 2] - 1)), abs(sm["final_cost"] / smo["final_cost"] - 1)))
        print("   gpu rf vs oracle 2: v %.2e w %.2e z %.2e" % (np.max(np.abs(g0["v"] - o2["v"])), np.max(np.abs(g0["w"] - o2["w"])),
                                                              np.max(np.abs(g0["inliers"][:, 2] / o2["inliers"][:, 2] - 1))))
        for mode in (0, 1):
            s.set_lm_arithmetic(mode)
            for _ in range(3):
                s.non_linear_refinement(*args, **kw)
            t0 = time.perf_counter()
            for _ in range(10):
                s.non_linear_refinement(*args, **kw)
            print("   host-pointer call, lm_arithmetic %d: %.3f ms" % (mode, (time.perf_counter() - t0) / 10 * 1e3))
        s.set_lm_arithmetic(0)
    s.close()


if __name__ == "__main__":
    main()
