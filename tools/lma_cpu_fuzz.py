#!/usr/bin/env python3
"""CPU campaign for the analytic LM trajectory: the oracle's mode 2 (rso_lma_trial: the HIP library's default arithmetic for the dense
depth solves, restated) against its mode 1 (the reference's iterate-by-iterate arithmetic) on the random cases of tests/fuzz_gpu.py.
Every integer must agree (per-trial counts, accepted LM steps, winner, mask); prints the guards' statistics.
    python tools/lma_cpu_fuzz.py [cases] [seed] [study] [wide]
study = 1 also runs rso_lma_trial's study mode on every finite hypothesis (distance between the two arithmetics, in units of guard (b)).
wide = 1 leaves synth.py's one scene family: every case in ACCELERATION mode (k != 0, k estimated per hypothesis), 10 % of the points get a flow
vector of up to +-0.5 in normalised units on top (real DeepFlow outliers are not capped at 30 px), coordinates up to a 110 degree field of view."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    import oracle_py as O
    import rsdsfm

    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    study = len(sys.argv) > 3 and int(sys.argv[3]) != 0
    wide = len(sys.argv) > 4 and int(sys.argv[4]) != 0
    bad = 0
    tot = dict(listed_clamped=0, listed_near=0, fallback=0, trials=0, pxhyp=0)
    reasons = {}
    kappa = rho_d = 0.0
    flips = caught = 0
    for c in range(cases):
        rng = np.random.default_rng(seed0 * 100003 + c)
        rows, cols = int(rng.integers(9, 90)), int(rng.integers(9, 130))
        cfg = int(rng.choice([1, 3]))
        v = rng.normal(size=3) * np.array([0.03, 0.03, 0.02])
        w = rng.normal(size=3) * 0.004
        k = float(rng.choice([0.0, 0.0, rng.uniform(-0.5, 0.8)]))
        if wide:
            k = float(rng.uniform(-0.5, 0.8))
            if abs(k) < 0.02:
                k = 0.05
        d = rsdsfm.synth.make_config(cfg, seed=int(rng.integers(1 << 30)), v=v, w=w, k=k, rows=rows, cols=cols)
        q, u, a, ak = d["q"], d["u"], d["alpha"], d["alpha_k"]
        n = len(q)
        if wide and n >= 9:
            u = u.copy()
            hit = rng.random(n) < 0.1
            u[hit] += rng.uniform(-0.5, 0.5, size=(int(hit.sum()), 2))
            q = q * float(rng.uniform(1.0, 2.2))  # (|x|, |y| up to ~1.4: a 110 degree field of view)
        if n < 9 or not (np.all(np.isfinite(q)) and np.all(np.isfinite(u))):
            continue
        tag = "case %d (%dx%d cfg %d n %d k %.3f)" % (c, rows, cols, cfg, n, k)
        try:
            pv = v + rng.normal(size=3) * 0.005
            pv /= np.linalg.norm(pv)
            pw = w + rng.normal(size=3) * 0.001
            pk = float(rng.choice([0.0, k]))
            r1, s1 = O.estimate_inverse_depths(q, u, pv, pw, pk, a, ak, mode=1)
            r2, s2 = O.estimate_inverse_depths(q, u, pv, pw, pk, a, ak, mode=2)
            for key in ("num_iterations", "num_successful_steps", "num_unsuccessful_steps", "termination"):
                assert s1[key] == s2[key], ("depth " + key, s1, s2)
            assert np.allclose(r1, r2, rtol=1e-9, atol=1e-13, equal_nan=True), "depth values"
            T = int(rng.choice([1, 3, 8, 20, 50]))
            tol = float(rng.choice([0.05, 0.01, 0.003, 0.001]))
            use_k = (bool(rng.integers(2)) or wide) and k != 0.0
            samples = O.sample_indices(n, T, int(rng.integers(1 << 30)))
            o1 = O.ransac(q, u, a, ak, use_k, T, tol, samples, depth_mode=1)
            o2 = O.ransac(q, u, a, ak, use_k, T, tol, samples, depth_mode=2)
            st = O.lma_last_stats()
            tot["trials"] += T
            tot["pxhyp"] += T * n
            for key in ("listed_clamped", "listed_near", "fallback"):
                tot[key] += st[key]
            if st["fallback"]:
                reasons[st["fallback_reason"]] = reasons.get(st["fallback_reason"], 0) + 1
            assert np.array_equal(o1["trial_count"], o2["trial_count"]), ("trial_count", o1["trial_count"], o2["trial_count"])
            assert np.array_equal(o1["trial_steps"], o2["trial_steps"]), ("trial_steps", o1["trial_steps"], o2["trial_steps"])
            assert o1["best_trial"] == o2["best_trial"], ("best_trial", o1["best_trial"], o2["best_trial"], o1["trial_err"], o2["trial_err"])
            assert np.array_equal(o1["mask"], o2["mask"]) and np.array_equal(o1["inlier_idx"], o2["inlier_idx"]), "mask"
            assert np.array_equal(o1["inv_depth"], o2["inv_depth"]), "winner's depths (mode 1's replay on both sides)"
            assert np.allclose(o1["trial_err"], o2["trial_err"], rtol=1e-9, atol=1e-12), ("trial_err", o1["trial_err"], o2["trial_err"])
            if study:
                for t in range(T):
                    tv = o1["trial_vel"][t]
                    if not np.all(np.isfinite(tv)):
                        continue
                    r = O.lma_trial(q, u, a, ak, tv[3:6], tv[0:3], tv[6], tol, study=True)["stats"]
                    kappa = max(kappa, r["margin_use_max"])
                    rho_d = max(rho_d, r["rho_diff_max"])
                    flips += r["flips_unguarded"]
                    caught += r["flips_listed"]
        except AssertionError as e:
            bad += 1
            print("MISMATCH", tag, e.args[0] if e.args else "", flush=True)
    print("lma cpu fuzz: %d cases seed %d, %d mismatches; trials %d, pixel-hypotheses %.3e, clamped pixels %d, near-tolerance pixels %d, "
          "fallbacks %d %s" % (cases, seed0, bad, tot["trials"], tot["pxhyp"], tot["listed_clamped"], tot["listed_near"], tot["fallback"], reasons))
    if study:
        print("study: kappa %.3g (guard b holds while < eta / 2 = 5e-12), rho diff %.3g, unguarded flips %d, flips the guard caught %d" % (kappa, rho_d, flips, caught))
    return 1 if bad or flips else 0


if __name__ == "__main__":
    sys.exit(main())
