"""Proves the PLUMBING of the pinning kit without the reference: builds a stand-in fixture with the harness' exact array names and
layouts FROM THE ORACLE (so the comparison is the oracle against itself), writes it through the .pin container and the importer's
packing into a temporary directory, and runs tests/test_reference_golden.py against it (RSDSFM_REFERENCE_GOLDEN).  It also checks
that every committed sample set is reachable by the reference's sampler (what pin_harness.cpp needs to inject it through rand()).
The stand-in is never written under tests/golden and says nothing about parity with the reference.

    python tools/pin_reference/selfcheck.py [-m gpu]
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pinio  # noqa: E402
from export_inputs import CASES, RANSAC_TOL  # noqa: E402

SUMMARY_COLS, TRACE_COLS, TRACE_ROWS = 10, 9, 64  # pin_harness.cpp


def summary_row(sm):
    pushed = sm["num_iterations"] - (1 if sm["termination"] in (1, 2) else 0)  # what summary.iterations would hold behind iteration 0
    return np.array([pushed, sm["num_successful_steps"], sm["num_unsuccessful_steps"], sm["termination"], sm["initial_cost"], sm["final_cost"], 0.0,
                     sm["final_radius"], sm["num_successful_steps"], sm["num_unsuccessful_steps"]])


def harness_trace(sm, tr):
    """the oracle's refinement trace (include/rsdsfm.h layout) as pin_harness.cpp lays out ceres' IterationSummary rows"""
    out = np.full((TRACE_ROWS, TRACE_COLS), np.nan)
    out[0] = [0, sm["initial_cost"], 0, np.nan, 0, 0, 1e4, 1, 1]
    cost = sm["initial_cost"]
    for i in range(sm["num_iterations"] - (1 if sm["termination"] in (1, 2) else 0)):
        ok = tr[i, 7] in (1.0, 5.0)
        cost = tr[i, 2] if ok else cost
        out[i + 1] = [i + 1, cost, tr[i, 1] - tr[i, 2], np.nan, tr[i, 6], tr[i, 4], np.nan, 1.0 if ok else 0.0, 0.0 if tr[i, 7] == 2.0 else 1.0]
    return out


def reachable(n, samples):
    perm = list(range(n))
    for s in samples:
        n_temp = n
        for want in s:
            r = perm.index(int(want), 0, n_temp)  # raises when the wanted index is no longer among the live slots
            perm[n_temp - 1], perm[r] = perm[r], perm[n_temp - 1]
            n_temp -= 1
    return True


def main():
    import oracle_py as O

    g = np.load(os.path.join(ROOT, "tests", "golden", "golden_v1.npz"))
    tmp = tempfile.mkdtemp(prefix="rsdsfm_pin_selfcheck_")
    packed = {}
    for case in CASES:
        q, u, a, ak, samples = (g[case + "/" + k] for k in ("q", "u", "alpha", "alpha_k", "samples"))
        use_k = bool(g[case + "/use_k"])
        n, T = len(q), len(samples)
        assert reachable(n, samples), case
        out = {}
        hyp = [O.calculate_velocities(q[s], u[s], a[s], ak[s], use_k) for s in samples]
        out["hyp_w"] = np.array([h[0] for h in hyp])
        out["hyp_v"] = np.array([h[1] for h in hyp])
        out["hyp_k"] = np.array([h[2] for h in hyp])
        rho, sums = [], []
        for t in range(T):
            r, sm = O.estimate_inverse_depths(q, u, out["hyp_v"][t], out["hyp_w"][t], out["hyp_k"][t], a, ak, mode=1)
            rho.append(r)
            sums.append(summary_row(sm))
        out["depth_rho"], out["depth_summary"] = np.array(rho), np.array(sums)
        out["depth_trace"] = np.full((T, TRACE_ROWS, TRACE_COLS), np.nan)
        # (the harness' one-step dump: the first finite hypothesis, up to 100 pixels at i * n / m)
        t0 = next((t for t in range(T) if np.all(np.isfinite(np.r_[out["hyp_v"][t], out["hyp_w"][t], out["hyp_k"][t]]))), -1)
        m = min(n, 100)
        out["one_step_rho"] = np.array([O.one_lm_step(q[i * n // m], u[i * n // m], a[i * n // m], ak[i * n // m], out["hyp_v"][t0], out["hyp_w"][t0],
                                                      out["hyp_k"][t0]) if t0 >= 0 else 0.0 for i in range(m)])
        out["one_step_hypothesis"] = np.array([float(t0)])
        rr = O.ransac(q, u, a, ak, use_k, T, RANSAC_TOL, samples, depth_mode=1)
        out["ransac_num_inliers"] = np.array([float(rr["num_inliers"])])
        out["ransac_inliers"], out["ransac_alpha"], out["ransac_alpha_k"] = rr["inliers"], rr["alpha"], rr["alpha_k"]
        out["ransac_wvk"] = np.concatenate([rr["w"], rr["v"], [rr["k"]]])
        for mode, name in ((0, "compat"), (1, "gather")):
            rf = O.refine(u, rr["inliers"], rr["alpha"], rr["alpha_k"], rr["v"], rr["w"], rr["k"], use_k, flow_index_mode=mode, inlier_idx=rr["inlier_idx"], trace_rows=64)
            out["refine_%s_wvk" % name] = np.concatenate([rf["w"], rf["v"], [rf["k"]]])
            out["refine_%s_z" % name] = rf["inliers"][:, 2].copy()
            out["refine_%s_summary" % name] = summary_row(rf["summary"])
            out["refine_%s_trace" % name] = harness_trace(rf["summary"], rf["trace"])
        pinio.write(os.path.join(tmp, case + ".pin"), out)  # through the container, as the harness' output would travel
        for k, v in pinio.read(os.path.join(tmp, case + ".pin")).items():
            packed[case + "/" + k] = v
    packed["_meta/versions"] = np.frombuffer(b"STAND-IN built from the oracle by tools/pin_reference/selfcheck.py: not the reference\n", dtype=np.uint8)
    fixture = os.path.join(tmp, "standin_reference.npz")
    np.savez_compressed(fixture, **packed)
    env = dict(os.environ, RSDSFM_REFERENCE_GOLDEN=fixture)
    marker = sys.argv[2] if len(sys.argv) > 2 and sys.argv[1] == "-m" else "not gpu"
    rc = subprocess.call([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_reference_golden.py"), "-q", "-m", marker, "-x"], env=env, cwd=ROOT)
    print("selfcheck:", "plumbing ok" if rc == 0 else "FAILED", "(stand-in fixture: %s)" % fixture)
    return rc


if __name__ == "__main__":
    sys.exit(main())
