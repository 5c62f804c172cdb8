"""Dumps the hypotheses of the minimal solver in acceleration mode (k estimated: minimal.cc:58-80) for many random 9-point sets, through the
wave-per-hypothesis path (T <= 512) and the lane-per-hypothesis path (larger T) -- to compare two builds of the library bit for bit:

    RSDSFM_LIB=<old .so> python tools/k_bits_dump.py /tmp/a.npy [cases];  python tools/k_bits_dump.py /tmp/b.npy [cases];  cmp /tmp/a.npy /tmp/b.npy
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import rsdsfm

    out = sys.argv[1]
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    res = []
    with rsdsfm.Solver(0) as s:
        for c in range(cases):
            rng = np.random.default_rng(991 * c + 5)
            rows, cols = int(rng.integers(20, 120)), int(rng.integers(20, 160))
            v = rng.normal(size=3) * np.array([0.03, 0.03, 0.02])
            w = rng.normal(size=3) * 0.004
            k = float(rng.uniform(-0.5, 0.8))
            d = rsdsfm.synth.make_config(int(rng.choice([1, 3])), seed=int(rng.integers(1 << 30)), v=v, w=w, k=k, rows=rows, cols=cols)
            q, u, a, ak = d["q"], d["u"], d["alpha"], d["alpha_k"]
            if len(q) < 9 or not (np.all(np.isfinite(q)) and np.all(np.isfinite(u))):
                continue
            for T in (50, 1500):
                smp = rsdsfm.sample_indices(len(a), T, 1 + c)
                qq, uu, aa, kk = (np.ascontiguousarray(x[smp], dtype=np.float64) for x in (q, u, a, ak))
                W, V, K = np.zeros((T, 3)), np.zeros((T, 3)), np.zeros(T)
                P = lambda x: x.ctypes.data_as(C.c_void_p)
                # (the call reports "no real eigenvalue" for the set as a whole, like the reference; the other hypotheses are still written)
                rc = s.lib.rsdsfm_calculate_velocities(s._ctx, P(qq), P(uu), P(aa), P(kk), C.c_int32(T), 1, 0, P(W), P(V), P(K))
                res.append(np.concatenate([W.ravel(), V.ravel(), K.ravel(), [float(rc)]]))
    blob = np.concatenate(res)
    np.save(out, blob)
    print("cases %d, values %d, finite %d" % (cases, blob.size, int(np.isfinite(blob).sum())))


if __name__ == "__main__":
    main()
