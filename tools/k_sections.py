"""Where the k estimation of the minimal solver (acceleration mode, minimal.cc:58-80) spends its shader clocks, per section, on the 50 sampled
9-point sets of N solves.  Needs a diagnostic build: RSDSFM_K_SECTIONS=1 in csrc/minimal9_kernels.hip (the probe's four slots then hold the
sections' clocks instead of the SVD's counts).

    python tools/k_sections.py [solves=40]        (on a GPU box)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    import rsdsfm

    N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    T = 50
    dev = torch.device("cuda", 0)
    seeds_data = [0x5EED0005 + 7919 * j for j in range(4)]
    flows, meta = rsdsfm.synth.make_flow_sequence(5, seeds_data)
    pts = [rsdsfm.synth.flatten_numpy(f, meta["K"], meta["gamma"]) for f in flows]
    rows = []
    with rsdsfm.Solver(0) as s:
        d_probe = torch.zeros(4 * T, dtype=torch.float64, device=dev)
        d_hyp = torch.zeros(8 * T, dtype=torch.float64, device=dev)
        for i in range(N):
            q, u, a, ak, _ = pts[i % len(pts)]
            smp = rsdsfm.sample_indices(len(a), T, 1 + i)
            tt = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
            dq, du, da, dak = tt(q[smp]), tt(u[smp]), tt(a[smp]), tt(ak[smp])
            s.minimal9_probe_dev(dq.data_ptr(), du.data_ptr(), da.data_ptr(), dak.data_ptr(), T, 1, 0, 1, d_hyp.data_ptr(), d_probe.data_ptr())
            s.synchronize()
            rows.append(d_probe.cpu().numpy().reshape(T, 4).copy())
    P = np.stack(rows)
    names = ["Z, 3x3 inverse, dga, P, PK", "6x6 inverse", "P PK^-1", "eigenvalues (Hessenberg + Francis)"]
    print("# %d solves x %d hypotheses, shader clocks per hypothesis (median / max of the per-solve maxima)" % (N, T))
    for j, n in enumerate(names):
        print("%-36s median %8.0f   slowest of a solve: median %8.0f  max %8.0f" % (n, np.median(P[..., j]), np.median(P[..., j].max(axis=1)), P[..., j].max()))
    tot = P.sum(axis=2)
    print("%-36s median %8.0f   slowest of a solve: median %8.0f  max %8.0f" % ("k estimation", np.median(tot), np.median(tot.max(axis=1)), tot.max()))


if __name__ == "__main__":
    main()
