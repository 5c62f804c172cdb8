"""How uneven are the 50 SVDs of one RANSAC?  minimal9_kernel runs one wavefront per hypothesis and the launch -- 18 % of a whole solve,
with ransac_lm_kernel waiting behind it -- lasts as long as its SLOWEST hypothesis.  This tool runs the library's solver on the 50 sampled
9-point sets of N solves (the bench's 1280x720 DeepFlow-like pairs, the library's sampler, seeds 1 .. N over 4 data seeds) through
rsdsfm_minimal9_probe_dev, which leaves per hypothesis the sweeps and rotations of its 9x9 two-sided Jacobi SVD (minimal.cc:98) and the
shader clocks the SVD and the whole hypothesis took, and prints the distribution of max / median per solve.

    python tools/svd_spread.py [solves=1000] > profiles/r04_svd_spread.txt        (on a GPU box)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def hist(x, edges):
    h, _ = np.histogram(x, bins=edges)
    return " ".join("%s:%d" % (("%.2f" % edges[i]), h[i]) for i in range(len(h)))


def main():
    import torch

    import rsdsfm

    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    T = 50
    dev = torch.device("cuda", 0)
    seeds_data = [0x5EED0005 + 7919 * j for j in range(4)]
    flows, meta = rsdsfm.synth.make_flow_sequence(5, seeds_data)
    pts = [rsdsfm.synth.flatten_numpy(f, meta["K"], meta["gamma"]) for f in flows]
    rows = []
    with rsdsfm.Solver(0) as s:
        d_probe = torch.zeros(4 * T, dtype=torch.float64, device=dev)
        d_hyp = torch.zeros(8 * T, dtype=torch.float64, device=dev)
        for cores in (1, 0):
            for i in range(N):
                q, u, a, ak, _ = pts[i % len(pts)]
                smp = rsdsfm.sample_indices(len(a), T, 1 + i)
                tt = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
                dq, du, da, dak = tt(q[smp]), tt(u[smp]), tt(a[smp]), tt(ak[smp])
                s.minimal9_probe_dev(dq.data_ptr(), du.data_ptr(), da.data_ptr(), dak.data_ptr(), T, 0, 0, cores, d_hyp.data_ptr(), d_probe.data_ptr())
                s.synchronize()
                p = d_probe.cpu().numpy().reshape(T, 4).copy()
                rows.append((cores, i, p))
    for cores in (1, 0):
        P = np.stack([p for c2, _, p in rows if c2 == cores])  # [N][T][4]
        sweeps, rot, svd_clk, tot_clk = P[..., 0], P[..., 1], P[..., 2], P[..., 3]
        print("# %d solves x %d hypotheses, SVD through the %s" % (len(P), T, "in-range function cores (as inside a RANSAC)" if cores else "standard functions"))
        print("sweeps per SVD:        min %d  median %d  max %d   histogram %s" % (sweeps.min(), np.median(sweeps), sweeps.max(), hist(sweeps.ravel(), np.arange(sweeps.min(), sweeps.max() + 2))))
        print("rotations per SVD:     min %d  median %d  max %d   (36 pairs per sweep)" % (rot.min(), np.median(rot), rot.max()))
        print("SVD clocks per hypothesis:   median %.0f  (%.1f clocks per rotation);  share of the hypothesis' clocks: %.2f" % (np.median(svd_clk), np.median(svd_clk / np.maximum(rot, 1)), np.median(svd_clk / tot_clk)))
        ratio_svd = svd_clk.max(axis=1) / np.median(svd_clk, axis=1)
        ratio_tot = tot_clk.max(axis=1) / np.median(tot_clk, axis=1)
        ratio_rot = rot.max(axis=1) / np.median(rot, axis=1)
        edges = np.array([1.0, 1.05, 1.1, 1.15, 1.2, 1.3, 1.5, 2.0, 100.0])
        print("per solve, slowest / median hypothesis:")
        print("   rotations        median %.3f  90 %% %.3f  max %.3f   histogram %s" % (np.median(ratio_rot), np.quantile(ratio_rot, 0.9), ratio_rot.max(), hist(ratio_rot, edges)))
        print("   SVD clocks       median %.3f  90 %% %.3f  max %.3f   histogram %s" % (np.median(ratio_svd), np.quantile(ratio_svd, 0.9), ratio_svd.max(), hist(ratio_svd, edges)))
        print("   whole hypothesis median %.3f  90 %% %.3f  max %.3f   histogram %s" % (np.median(ratio_tot), np.quantile(ratio_tot, 0.9), ratio_tot.max(), hist(ratio_tot, edges)))
        print("   the launch waits for max: mean over solves of (max - median) / max of the hypothesis clocks = %.3f" % np.mean((tot_clk.max(axis=1) - np.median(tot_clk, axis=1)) / tot_clk.max(axis=1)))
        print()


if __name__ == "__main__":
    main()
