"""What "PARITY UNPINNED" risks: the oracle's trust-region loops restate Ceres 1.14 from recollection (Ceres is not in this image).
For every recalled detail with a plausible alternative reading the oracle has a switch (rso_set_variant); this tool runs the
oracle chain (flatten -> RANSAC with T Ceres-LM depth solves -> joint refinement) on BASELINE configs[2] (1920x1080, T = 5, selective
tolerance) and configs[4] data (1280x720, T = 50, main.cc's tolerance) once per switch and reports what changes against the pinned
oracle: integers (winner, inlier-mask bits, per-trial inlier counts and accepted LM steps, refinement iterations) and the largest
relative change of rho (winner's dense 1/depth), v, w.  CPU only; test infrastructure.

    python tools/oracle_sensitivity.py [--small] [--json out.json]      (full sizes: ~1-2 min)
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

SWITCHES = [("ftol", 1, "function tolerance: a candidate that would be accepted IS applied before terminating"),
            ("ftol", 2, "function tolerance tested only after an accepted step"),
            ("ftol_lt", 1, "function tolerance with strict < instead of <="),
            ("jacobi", 1, "no Jacobi column scaling"),
            ("mindiag", 1, "LM diagonal not clamped to [1e-6, 1e32]"),
            ("dsq", 1, "D^2 formed as sqrt(diag / radius)^2 (Ceres' literal form) instead of diag * (1 / radius)"),
            ("radius", 1, "textbook radius rule (x3 above 0.75, / 2 below 0.25) instead of 1 / max(1/3, 1 - (2 rho - 1)^3)"),
            ("svd_sign", 1, "Eigen's JacobiSVD leaves the null vector V.col(8) with the opposite sign (implementation-defined)")]


def chain(O, d, T, tol, seed, accel=False):
    rows, K, gamma = d["rows"], d["K"], d["gamma"]
    q, u, qpx, fpx = O.flatten(d["flow_img"], *K, gamma)
    a, ak = O.get_alpha(fpx, rows, gamma), O.get_alpha_k(qpx, fpx, rows, gamma)
    ro = O.ransac(q, u, a, ak, accel, T, tol, O.sample_indices(len(q), T, seed), depth_mode=1)
    ref = O.refine(u, ro["inliers"], ro["alpha"], ro["alpha_k"], ro["v"], ro["w"], ro["k"], accel, 0, None)  # main.cc:457: rank-indexed flow
    inl, v, flipped = O.canonicalize_sign(ref["inliers"], ref["v"])
    return dict(ransac=ro, refine=ref, v=v, inliers=inl, flipped=flipped)


def rel(a, b):
    """largest element-wise relative change"""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    den = np.maximum(np.abs(b), 1e-300)
    return float(np.max(np.abs(a - b) / den)) if a.size else 0.0


def rel_norm(a, b):
    """|a - b| / |b| of a vector (a pose component near zero does not inflate it)"""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def rel_quantiles(a, b):
    """median / 99.9th percentile / maximum of the element-wise relative change"""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if not a.size:
        return [0.0, 0.0, 0.0]
    r = np.abs(a - b) / np.maximum(np.abs(b), 1e-300)
    return [float(np.median(r)), float(np.quantile(r, 0.999)), float(r.max())]


def compare(base, var):
    rb, rv = base["ransac"], var["ransac"]
    # a flipped null vector flips the RANSAC's v and every inverse depth (main.cc:466-478 undoes it after the refinement): compare the
    # winner's dense 1/depth and the RANSAC pose up to that global sign
    sgn = 1.0 if float(np.dot(rb["v"], rv["v"])) >= 0 else -1.0
    rv = dict(rv, inv_depth=sgn * rv["inv_depth"], v=sgn * rv["v"])
    out = {"winner_same": bool(rb["best_trial"] == rv["best_trial"]),
           "inlier_count": [int(rb["num_inliers"]), int(rv["num_inliers"])],
           "mask_bits_flipped": int(np.count_nonzero(rb["mask"] != rv["mask"])),
           "trial_counts_differ": int(np.count_nonzero(rb["trial_count"] != rv["trial_count"])),
           "max_trial_count_change": int(np.max(np.abs(rb["trial_count"] - rv["trial_count"]))) if len(rb["trial_count"]) else 0,
           "trial_lm_steps_differ": int(np.count_nonzero(rb["trial_steps"] != rv["trial_steps"])),
           "rho_winner_max_rel": rel(rv["inv_depth"], rb["inv_depth"]), "rho_winner_rel_quantiles": rel_quantiles(rv["inv_depth"], rb["inv_depth"]),
           "refine_iterations": [int(base["refine"]["summary"]["num_iterations"]), int(var["refine"]["summary"]["num_iterations"])],
           "refine_termination": [int(base["refine"]["summary"]["termination"]), int(var["refine"]["summary"]["termination"])],
           "v_max_rel": rel_norm(var["v"], base["v"]), "w_max_rel": rel_norm(var["refine"]["w"], base["refine"]["w"]),
           "ransac_v_rel": rel_norm(rv["v"], rb["v"]), "ransac_w_rel": rel_norm(rv["w"], rb["w"])}
    if rb["num_inliers"] == rv["num_inliers"] and out["mask_bits_flipped"] == 0:
        out["depth_refined_max_rel"] = rel(var["inliers"][:, 2], base["inliers"][:, 2])
        out["depth_refined_rel_quantiles"] = rel_quantiles(var["inliers"][:, 2], base["inliers"][:, 2])  # median / 99.9 % / max (pixels with rho ~ 0 dominate the max)
    return out


def study(O, rsdsfm, small=False):
    cases = [("configs[2] 1920x1080, T = 5, tol 0.002", 3, dict(rows=270, cols=480) if small else {}, 5, 0.002, 5),
             ("configs[4] 1280x720, T = %d, tol 0.05" % (12 if small else 50), 5, dict(rows=180, cols=320) if small else {}, 12 if small else 50, 0.05, 1)]
    report = {}
    for name, cfg, size, T, tol, seed in cases:
        d = rsdsfm.synth.make_config(cfg, **size)
        base = chain(O, d, T, tol, seed)
        rows = {}
        for sw, val, note in SWITCHES:
            with O.variant(**{sw: val}):
                var = chain(O, d, T, tol, seed)
            rows["%s=%d" % (sw, val)] = dict(compare(base, var), note=note)
        again = chain(O, d, T, tol, seed)  # the switches are reset: the pinned oracle again, bit for bit
        assert compare(base, again)["rho_winner_max_rel"] == 0.0 and np.array_equal(again["v"], base["v"])
        report[name] = {"n": int(len(base["ransac"]["mask"])), "num_inliers": int(base["ransac"]["num_inliers"]), "trials": T,
                        "lm_steps_per_trial": sorted(set(int(x) for x in base["ransac"]["trial_steps"])), "switches": rows}
    return report


def markdown(report):
    lines = []
    for name, rec in report.items():
        lines.append("**%s** (n = %d, inliers = %d, accepted LM steps per trial: %s)\n" % (name, rec["n"], rec["num_inliers"], rec["lm_steps_per_trial"]))
        lines.append("| if recollection X is wrong | winner | mask bits | trials whose count / LM steps change | rho of the winner: median / 99.9 % / max rel. change | refinement iterations | v (refined) | w (refined) | refined depth: median / 99.9 % |")
        lines.append("|---|---|---|---|---|---|---|---|---|")
        for key, r in rec["switches"].items():
            qs = r["rho_winner_rel_quantiles"]
            dq = r.get("depth_refined_rel_quantiles")
            lines.append("| %s (%s) | %s | %d | %d (max %d) / %d | %.1e / %.1e / %.1e | %d -> %d | %.1e | %.1e | %s |" % (
                r["note"], key, "same" if r["winner_same"] else "CHANGES", r["mask_bits_flipped"], r["trial_counts_differ"], r["max_trial_count_change"],
                r["trial_lm_steps_differ"], qs[0], qs[1], qs[2], r["refine_iterations"][0], r["refine_iterations"][1], r["v_max_rel"], r["w_max_rel"],
                ("%.1e / %.1e" % (dq[0], dq[1])) if dq else "-"))
        lines.append("")
    return "\n".join(lines)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--small", action="store_true")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    import oracle_py as O
    import rsdsfm

    O.lib()
    rep = study(O, rsdsfm, small=args.small)
    if args.json:
        json.dump(rep, open(args.json, "w"), indent=1)
    print(markdown(rep))
