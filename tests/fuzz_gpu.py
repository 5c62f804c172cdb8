#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity campaign (not collected by pytest: run  python tests/fuzz_gpu.py [cases] [seed]  on a GPU box).

Each case draws a frame size, motion, shutter parameter, noise / outlier level, RANSAC tolerance and trial count, then
compares through the C ABI against the CPU oracle: the dense depth solve for a random pose (closed form + LM: depths,
iteration counts, termination), RANSAC (per-trial counts and accepted LM steps, best trial, mask, index list: bit-exact)
and the refinement started from the RANSAC result (iteration counts and termination exact, values 1e-6).  Prints one
line per failing case and a summary; exit code 1 if anything differed.

Environment: FUZZ_ONLY=4200,7707 re-runs selected case numbers of a campaign (same cases / seed arguments); FUZZ_VERBOSE=1 prints, for
every refinement whose decisions split from the oracle's, the first LM iteration where the two iteration traces differ and how close
the deciding quantity sat to its threshold (rsdsfm_set_refine_trace / the oracle's rso_set_refine_trace)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def fuzz_consumers(O, rsdsfm, cases, seed0):
    """SURVEY 8(f) consumers on random scenes: flatten + depth map glue, RS->GS back projection (RS / GS mode, both Q5 modes),
    crack interpolation, 8-bit depth image, ground-truth flow search -- byte / index / fp outputs bit-exact -- and the
    reprojection metric (statistics to summation order, integer counts exact)"""
    bad = 0
    with rsdsfm.Solver(0) as s:
        for c in range(cases):
            rng = np.random.default_rng(seed0 * 7919 + 31 * c + 5)
            rows, cols = int(rng.integers(1, 70)), int(rng.integers(1, 110))
            tag = "consumer case %d (%dx%d)" % (c, rows, cols)
            try:
                d = rsdsfm.synth.make_config(int(rng.choice([1, 3])), seed=int(rng.integers(1 << 30)), rows=max(rows, 3), cols=max(cols, 3))
                rows, cols, K, gamma = d["rows"], d["cols"], d["K"], d["gamma"]
                # flatten glue
                q, u, a, ak = s.flatten(d["flow_img"], K, gamma)  # the GPU glue fuses getAlpha / getAlphaK into the flatten
                qo, uo, qpxo, fpxo = O.flatten(d["flow_img"], *K, gamma)
                assert np.array_equal(q, qo) and np.array_equal(u, uo), "flatten"
                assert np.array_equal(a, O.get_alpha(fpxo, rows, gamma)) and np.array_equal(ak, O.get_alpha_k(qpxo, fpxo, rows, gamma)), "alpha"
                # a scene: image, depth with holes, a random motion
                img = rng.integers(0, 256, size=(rows, cols, 3), dtype=np.uint8)
                img[rng.random((rows, cols)) < 0.05] = (2, 3, 1)
                depth = np.array(d["truth"]["Z"]) * rng.uniform(0.5, 2.0)
                depth[rng.random((rows, cols)) < 0.08] = 0.0
                v = rng.normal(size=3) * np.array([0.15, 0.12, 0.3])
                w = rng.normal(size=3) * 0.05
                k = float(rng.choice([0.0, rng.uniform(-0.4, 0.6)]))
                R, t = O.pose_table(v, w, k, gamma, rows)
                for mode, q5 in ((0, 0), (0, 1), (1, 0), (1, 1)):
                    gs, c3 = s.back_project(img, depth, R, t, K, mode=mode, q5_mode=q5)
                    gs_o, c3_o = O.back_project(img, depth, R, t, *K, mode=mode, q5_mode=q5)
                    assert np.array_equal(gs, gs_o), "back_project image mode %d q5 %d" % (mode, q5)
                    assert np.array_equal(c3.view(np.uint32), c3_o.view(np.uint32)), "back_project coords mode %d q5 %d" % (mode, q5)
                off = int(rng.integers(1, 4))
                assert np.array_equal(s.interpolate_cracky(gs, off), O.interpolate_cracky(gs_o, off)), "interpolate_cracky"
                # depth image of random inliers (collisions, points outside the image, negative depths)
                m = int(rng.integers(0, 3 * rows * cols + 1))
                inl = np.column_stack([rng.uniform(-0.6, 0.6, m), rng.uniform(-0.5, 0.5, m), rng.normal(2.0, 1.5, m)])
                assert np.array_equal(s.depth_preview(inl, K, rows, cols), O.depth_preview(inl, *K, rows, cols)), "depth_preview"
                # ground-truth flow search against a frame 2 with its own number of scanlines
                yy, xx = np.mgrid[0:rows, 0:cols]
                world = np.stack([(xx - K[2]) / K[0], (yy - K[3]) / K[1], np.ones((rows, cols))], axis=2) * depth[:, :, None]
                rows2 = int(rng.integers(1, 90))
                R2, t2 = O.pose_table(v * rng.uniform(0.5, 1.5), w * rng.uniform(0.5, 1.5), k, gamma, rows2)
                t2 = t2 + rng.normal(size=3) * 0.02
                q5 = int(rng.integers(2))
                flow_o, best_o = O.true_flow(world, R2, t2, *K, q5_mode=q5)
                for search in (1, 2):  # the exhaustive loop and the interval-pruned search (forced: these frames have few scanlines)
                    s.set_true_flow_search(search)
                    flow, best = s.true_flow(world, R2, t2, K, q5_mode=q5)
                    assert np.array_equal(best, best_o), "true_flow winners (search mode %d)" % search
                    assert np.array_equal(flow.view(np.uint64), flow_o.view(np.uint64)), "true_flow values (search mode %d)" % search
                s.set_true_flow_search(0)
                # reprojection metric
                est = (c3_o.astype(np.float64) * rng.uniform(0.7, 1.4)).astype(np.float32)
                est += (rng.normal(0, 0.02, est.shape) * (rng.random(est.shape) < 0.5)).astype(np.float32)
                gt = np.array(d["truth"]["Z"])
                gt[rng.random((rows, cols)) < 0.03] = 0.0
                st, eimg = s.reprojection_error(est, gt, depth, R, t, K, max_norm=4.0)
                st_o, eimg_o = O.reprojection_error(est, gt, depth, R, t, *K, max_norm=4.0)
                for key in ("number_outliers", "scale_inliers", "error_inliers"):
                    assert st[key] == st_o[key], ("metric " + key, st, st_o)
                for key in ("scale", "mean_error", "sum_error"):
                    assert np.isclose(st[key], st_o[key], rtol=1e-10, atol=1e-300, equal_nan=True), ("metric " + key, st, st_o)
                assert (eimg != eimg_o).mean() < 1e-3, "error image"
            except AssertionError as e:
                bad += 1
                print("MISMATCH", tag, e.args[0] if e.args else "", flush=True)
            except rsdsfm.RsdsfmError as e:
                bad += 1
                print("ERROR", tag, e, flush=True)
    print("fuzz consumers: %d cases, %d mismatches" % (cases, bad))
    return bad


DECISION_KEYS = ("num_iterations", "num_successful_steps", "num_unsuccessful_steps", "termination")


def _refine_deviation(x, ref):
    """deviation of one refinement result from another, modulo the scale gauge of the problem (the residual only sees rho * v, the
    refinement does not normalise v, and a long trajectory drifts along that flat direction by accumulated rounding)"""
    nx, nr = np.linalg.norm(x["v"]), np.linalg.norm(ref["v"])
    dv = np.abs(x["v"] / nx - ref["v"] / nr) / np.maximum(np.abs(ref["v"] / nr), 1e-2)  # relative, floor 1e-7 / 1e-5
    dw = np.abs(x["w"] - ref["w"]) / np.maximum(np.abs(ref["w"]), 1e-3)               # relative, floor 1e-8 / 1e-5
    zg, zr = x["inliers"][:, 2] / nx, ref["inliers"][:, 2] / nr
    rel = np.abs(zg - zr) / np.maximum(np.abs(zr), 1e-4)                                # relative, floor 1e-9 / 1e-5
    cx, cr = x["summary"]["final_cost"], ref["summary"]["final_cost"]
    return dict(pose=float(max(dv.max(), dw.max())), depth_q995=float(np.quantile(rel, 0.995)) if len(rel) else 0.0,
                depth_max=float(rel.max()) if len(rel) else 0.0, cost=abs(cx - cr) / max(abs(cr), 1e-18))


def _first_divergence(s, O, u, r, ro, use_k):
    """where a split trajectory leaves the oracle's: both sides re-run with their iteration traces (rsdsfm_set_refine_trace / the
    oracle's own); returns the first iteration whose outcome differs, both outcomes, and how close the deciding quantity sat to its
    threshold (relative decrease against 1e-3, |cost change| / cost against 1e-6) -- the margin a summation order has to flip"""
    s.set_refine_trace(50)
    try:
        s.non_linear_refinement(u, r["inliers"], r["alpha"], r["alpha_k"], r["v"], r["w"], r["k"], use_k, flow_index_mode=1, inlier_idx=r["inlier_idx"])
        tg = s.get_refine_trace()
    finally:
        s.set_refine_trace(0)
    to = O.refine(u, ro["inliers"], ro["alpha"], ro["alpha_k"], ro["v"], ro["w"], ro["k"], use_k, 1, ro["inlier_idx"], trace_rows=50)["trace"]
    for i in range(50):
        a, b = tg[i, 7], to[i, 7]
        if np.isnan(a) and np.isnan(b):
            break
        if a != b and not (np.isnan(a) and np.isnan(b)):
            out = dict(iteration=i + 1, gpu_outcome=float(a), oracle_outcome=float(b), radius=float(to[i, 5]))
            for name, t in (("gpu", tg), ("oracle", to)):
                cost, ccost, model = t[i, 1], t[i, 2], t[i, 3]
                out[name + "_rel_decrease"] = float((cost - ccost) / model) if model > 0 else float("nan")
                out[name + "_cost_change_over_cost"] = float(abs(cost - ccost) / cost) if cost > 0 else float("nan")
            prev = slice(0, i)
            with np.errstate(invalid="ignore", divide="ignore"):
                out["max_rel_diff_of_the_sums_before"] = float(np.nanmax(np.abs(tg[prev, 1:5] / to[prev, 1:5] - 1.0))) if i > 0 else 0.0
            return out
    return None


def _oracle_reorderings(O, u, ro, use_k):
    """the oracle's refinement of the same problem with its inlier list re-ordered (reversed + 3 seeded random permutations): its
    own sums added in other orders.  What these runs disagree about is not defined by the reference's arithmetic."""
    m = len(ro["inlier_idx"])
    perms = [np.arange(m)[::-1]] + [np.random.default_rng(100 + t).permutation(m) for t in range(3)]
    outs = []
    for pm in perms:
        o2 = O.refine(u, ro["inliers"][pm], ro["alpha"][pm], ro["alpha_k"][pm], ro["v"], ro["w"], ro["k"], use_k, 1, ro["inlier_idx"][pm])
        inv = np.empty(m, dtype=np.int64)
        inv[pm] = np.arange(m)
        o2["inliers"] = o2["inliers"][inv]  # back in the original order
        outs.append(o2)
    return outs


def main(cases=None, seed0=None):
    import oracle_py as O
    import rsdsfm

    if cases is None:
        cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    if seed0 is None:
        seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    bad = ties = splits = floors = 0
    only = [int(x) for x in os.environ.get("FUZZ_ONLY", "").split(",") if x]  # re-run selected case numbers of a campaign
    with rsdsfm.Solver(0) as s:
        for c in (only or range(cases)):
            rng = np.random.default_rng(seed0 * 100003 + c)
            rows, cols = int(rng.integers(9, 90)), int(rng.integers(9, 130))
            cfg = int(rng.choice([1, 3]))
            v = rng.normal(size=3) * np.array([0.03, 0.03, 0.02])
            w = rng.normal(size=3) * 0.004
            k = float(rng.choice([0.0, 0.0, rng.uniform(-0.5, 0.8)]))
            d = rsdsfm.synth.make_config(cfg, seed=int(rng.integers(1 << 30)), v=v, w=w, k=k, rows=rows, cols=cols)
            q, u, a, ak = d["q"], d["u"], d["alpha"], d["alpha_k"]
            n = len(q)
            tag = "case %d (%dx%d cfg %d n %d k %.3f)" % (c, rows, cols, cfg, n, k)
            try:
                if n < 9:
                    continue
                # dense depth solve at a random (not the true) pose
                pv = v + rng.normal(size=3) * 0.005
                pv /= np.linalg.norm(pv)
                pw = w + rng.normal(size=3) * 0.001
                pk = float(rng.choice([0.0, k]))
                if not (np.all(np.isfinite(q)) and np.all(np.isfinite(u))):
                    continue  # degenerate intrinsics of a sliver-shaped frame: not a parity case (NaN handling has its own test)
                for mode in (0, 1):
                    rho, sm = s.estimate_inverse_depths(q, u, pv, pw, pk, a, ak, mode=mode)
                    rho_o, sm_o = O.estimate_inverse_depths(q, u, pv, pw, pk, a, ak, mode=mode)
                    # (equal_nan: a sliver-shaped frame's degenerate intrinsics can give finite flows of 1e260 whose squares overflow --
                    # NaN depths on both sides)
                    assert np.allclose(rho, rho_o, rtol=1e-9, atol=1e-13, equal_nan=True), "depth values mode %d" % mode
                    if mode == 1:
                        for key in ("num_iterations", "num_successful_steps", "num_unsuccessful_steps", "termination"):
                            assert sm[key] == sm_o[key], ("depth " + key, sm, sm_o)
                # RANSAC
                T = int(rng.choice([1, 3, 8, 20, 50]))
                tol = float(rng.choice([0.05, 0.01, 0.003, 0.001]))
                use_k = bool(rng.integers(2)) and k != 0.0
                samples = O.sample_indices(n, T, int(rng.integers(1 << 30)))
                dm = int(rng.integers(2))
                # (FUZZ_LM_ARITHMETIC=1: the iterate-by-iterate kernels.  Default 0 = the analytic LM trajectory -- set before EVERY case, which
                # also ends a hold: otherwise the noise-free cases, whose ties send a run back to the iterate-by-iterate kernels for the
                # context's next 16 runs, would keep most of the campaign off the analytic path)
                s.set_lm_arithmetic(int(os.environ.get("FUZZ_LM_ARITHMETIC", "0")))
                r = s.ransac(q, u, a, ak, use_k, T, tol, samples=samples, depth_mode=dm)
                ro = O.ransac(q, u, a, ak, use_k, T, tol, samples, depth_mode=dm)
                assert np.array_equal(r["trial_count"], ro["trial_count"]), "trial_count"
                assert np.array_equal(r["trial_steps"], ro["trial_steps"]), "trial_steps"
                if r["best_trial"] != ro["best_trial"]:
                    # minimal.cc:278-285 breaks ties in the inlier count by the SUM of the inlier errors.  With noise-free data every
                    # trial explains every point and each point's error is the rounding noise of the arithmetic itself (1e-14),
                    # which depends on the last bits of the trust-region radius (a ratio of global sums whose rounding depends on
                    # the summation order: sequential on the CPU, a tree on the GPU): the sums of two all-inlier trials differ by
                    # 1e-7 .. 1e-5 relative and their order is arbitrary on both sides.  Counted, not failed.
                    tg, to = r["best_trial"], ro["best_trial"]
                    cnt = max(int(r["trial_count"][to]), 1)
                    tie = (r["trial_count"][tg] == r["trial_count"][to]
                           and (abs(r["trial_err"][tg] - r["trial_err"][to]) <= 1e-3 * max(abs(r["trial_err"][to]), 1e-300)
                                or max(r["trial_err"][tg], r["trial_err"][to]) / cnt < 1e-11))  # per-point error = rounding noise
                    if tie:
                        ties += 1
                        continue
                    raise AssertionError("best trial %d vs %d, counts %s, errs %.17g %.17g" % (tg, to, r["trial_count"][[tg, to]], r["trial_err"][tg], r["trial_err"][to]))
                assert r["num_inliers"] == ro["num_inliers"], "num_inliers"
                assert np.array_equal(r["mask"], ro["mask"]) and np.array_equal(r["inlier_idx"], ro["inlier_idx"]), "mask"
                if not np.allclose(r["inv_depth"], ro["inv_depth"], rtol=1e-9, atol=1e-13, equal_nan=True):
                    rel = np.abs(r["inv_depth"] - ro["inv_depth"]) / np.maximum(np.abs(ro["inv_depth"]), 1e-300)
                    i = int(np.argmax(rel))
                    raise AssertionError("inv_depth: %d of %d points beyond 1e-9, worst %.3e at %d (%.17g vs %.17g), q %s" % (
                        int(np.sum(rel > 1e-9)), n, rel[i], i, r["inv_depth"][i], ro["inv_depth"][i], q[i]))
                if not np.allclose(r["trial_vel"], ro["trial_vel"], rtol=1e-8, atol=1e-10, equal_nan=True):  # degenerate samples give NaN poses on both sides
                    dv = np.nan_to_num(np.abs(r["trial_vel"] - ro["trial_vel"])).reshape(T, -1)
                    t_bad = int(np.argmax(dv.max(axis=1)))
                    raise AssertionError("trial_vel: trial %d max abs diff %.3e, gpu %s oracle %s, count %d of %d" % (
                        t_bad, dv.max(), np.array2string(r["trial_vel"].reshape(T, -1)[t_bad], precision=12),
                        np.array2string(ro["trial_vel"].reshape(T, -1)[t_bad], precision=12), int(r["trial_count"][t_bad]), n))
                # refinement from the RANSAC result
                if r["num_inliers"] >= 12:
                    out = s.non_linear_refinement(u, r["inliers"], r["alpha"], r["alpha_k"], r["v"], r["w"], r["k"], use_k,
                                                  flow_index_mode=1, inlier_idx=r["inlier_idx"])
                    ref = O.refine(u, ro["inliers"], ro["alpha"], ro["alpha_k"], ro["v"], ro["w"], ro["k"], use_k, 1, ro["inlier_idx"])
                    so, sr = out["summary"], ref["summary"]
                    same = all(so[key] == sr[key] for key in DECISION_KEYS)
                    if not same:
                        # The accept / reject / converge decisions differ.  Both sides evaluate the same per-point terms; what differs is
                        # the ORDER in which the global Schur / cost sums are added (sequential on the CPU, tree-shaped on the GPU).
                        # Characterise the case with the oracle alone: re-run it on re-ordered point lists (reversed + 3 random
                        # permutations) -- the same problem, its own sums merely added in another order.  If the oracle reproduces its
                        # decisions every time, the trajectory is stable
                        # under summation order and the GPU's deviation is a real mismatch: FAIL.  If the oracle splits from itself
                        # (an accept / reject test sits within the rounding of a sum: tiny or sliver-shaped frames, acceleration mode,
                        # 15+ iterations), the reference's own result is not defined beyond that spread; the GPU must then end with
                        # the same termination type, inside the oracle's own spread of the final cost.  Counted, not failed.
                        others = _oracle_reorderings(O, u, ro, use_k)
                        oracle_stable = all(all(o2["summary"][key] == sr[key] for key in DECISION_KEYS) for o2 in others)
                        if oracle_stable:
                            # four re-orderings are a sample, not a proof.  One more class is benign by construction: noise-free data
                            # refined down to the rounding floor (final cost <= 1e-12 x initial cost on both sides) -- there the gradient's
                            # max norm IS rounding residue, and whether it lands below the 1e-10 gradient tolerance one iteration earlier or
                            # later cannot matter: the results must then agree at the value bars; anything else stays a mismatch
                            floor = max(so["final_cost"], sr["final_cost"]) <= 1e-12 * sr["initial_cost"]
                            dev = _refine_deviation(out, ref)
                            assert floor and dev["pose"] <= 1e-5 and dev["depth_q995"] <= 1e-5 and dev["depth_max"] <= 1e-2, (
                                "refinement decisions differ although the oracle's trajectory is stable under re-ordering its sums", so, sr, dev,
                                _first_divergence(s, O, u, r, ro, use_k))
                            floors += 1
                            continue
                        spread = max(abs(o2["summary"]["final_cost"] - sr["final_cost"]) for o2 in others)
                        # diagnostics of a failure: how far the results are apart (gauge-free), next to the oracle's own re-ordering spread
                        diag = dict(gpu_vs_oracle=_refine_deviation(out, ref), oracle_reorderings=[_refine_deviation(o2, ref) for o2 in others],
                                    first_divergence=_first_divergence(s, O, u, r, ro, use_k))
                        if os.environ.get("FUZZ_VERBOSE"):
                            print("SPLIT", tag, diag["first_divergence"], flush=True)
                        assert so["termination"] in [sr["termination"]] + [o2["summary"]["termination"] for o2 in others], ("termination type of a split trajectory", so, sr, diag)
                        # (a trajectory cut off by the 50-iteration cap ends wherever its accept / reject pattern took it: one order
                        # of magnitude around the oracle's own spread is the bar, a lower cost than every oracle run is not a defect)
                        assert abs(so["final_cost"] - sr["final_cost"]) <= 10.0 * spread + 1e-9 * abs(sr["final_cost"]), (
                            "split trajectory ends outside the oracle's own spread", so, sr, [o2["summary"] for o2 in others], diag)
                        splits += 1
                        continue
                    # values at the north-star tolerance (1e-5; the committed tests assert 1e-6 on well-conditioned cases), compared
                    # modulo the scale gauge of the problem: the residual only sees rho * v, the refinement does not normalise v,
                    # and a long trajectory drifts along that flat direction by accumulated rounding (|v| differed by 1e-3 in one
                    # 41-iteration case while the cost agreed to 4e-10)
                    assert np.array_equal(out["inliers"][:, :2], ref["inliers"][:, :2]), "refine inlier coordinates"
                    dev = _refine_deviation(out, ref)
                    # bars: pose 1e-5 (north star; absolute floors 1e-7 / 1e-8 as before), 99.5 % of the depths within 1e-5 and all within
                    # 1e-2 (single points whose depth the data barely constrains, 1/depth ~ 0, may differ more after a long run), cost 1e-7
                    bars = dict(pose=1e-5, depth_q995=1e-5, depth_max=1e-2, cost=1e-7)
                    over = [key for key in bars if dev[key] > bars[key]]
                    if over:
                        # same decisions, different values: a long trajectory (tens of iterations, a trust-region radius of 1e15 where the
                        # undamped Schur system is singular along the scale gauge) amplifies the rounding of its sums.  Same
                        # characterisation as above: the oracle on re-ordered point lists.  Accepted only if the oracle's own re-ordering
                        # moves the quantity at least a quarter as far as the GPU is away from it; a deviation the reference's arithmetic
                        # does not exhibit itself is a mismatch.
                        others = [_refine_deviation(o2, ref) for o2 in _oracle_reorderings(O, u, ro, use_k)]
                        for key in over:
                            own = max(o2[key] for o2 in others)
                            assert dev[key] <= 4.0 * own, ("refine " + key, dev[key], "oracle's own spread", own, so, sr)
                        splits += 1
            except AssertionError as e:
                bad += 1
                print("MISMATCH", tag, e.args[0] if e.args else "", flush=True)
            except rsdsfm.RsdsfmError as e:
                bad += 1
                print("ERROR", tag, e, flush=True)
            lma = s.lma_restarts() if c == (only or range(cases))[-1] else None
            if lma is not None:
                # the RANSACs ran on the analytic LM trajectory (the library's default): how often a global guard sent a run back to the
                # iterate-by-iterate kernels (ties on noise-free data, mostly), and which guards tripped last
                print("analytic LM trajectory: %d of the RANSAC runs started over iterate by iterate (last guards: bit set %d)" % lma, flush=True)
                # the refinements ran on radius-factorised Schur sums (the default): how many a guard sent back to the iterate-by-iterate kernels
                print("radius-factorised refinement: %(runs)d ran on it, %(restarts)d were sent back by a guard (last guard %(last_guard)d), %(resolves)d reduced "
                      "systems were solved again from kept sums (rejected / invalid steps)" % s.refine_restarts(), flush=True)
    if not only:
        bad += fuzz_consumers(O, rsdsfm, max(cases // 2, 1), seed0)
    print("fuzz: %d cases, %d mismatches; %d all-inlier ties decided by rounding noise, %d ill-conditioned / split refinement trajectories (outcome compared), "
          "%d refinements to the rounding floor ending one iteration apart (values compared)" % (cases, bad, ties, splits, floors))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
