"""CPU: the oracle's mode 2 -- the analytic LM trajectory, i.e. the HIP library's default arithmetic for the dense depth solves restated
in C (oracle/rsdsfm_oracle.c rso_lma_trial) -- against its mode 1 (the reference's iterate-by-iterate arithmetic).  Every integer must
agree: accepted LM steps, terminations, per-trial inlier counts, winner, mask.  The study mode measures how far the two arithmetics are
apart in units of guard (b)'s margin (csrc/lma_common.hpp)."""
import numpy as np
import pytest

from conftest import GOLDEN_CASES

INT_KEYS = ("num_iterations", "num_successful_steps", "num_unsuccessful_steps", "termination")


def _same_ransac(o1, o2):
    assert np.array_equal(o1["trial_count"], o2["trial_count"])
    assert np.array_equal(o1["trial_steps"], o2["trial_steps"])
    assert o1["best_trial"] == o2["best_trial"] and o1["num_inliers"] == o2["num_inliers"]
    assert np.array_equal(o1["mask"], o2["mask"]) and np.array_equal(o1["inlier_idx"], o2["inlier_idx"])
    assert np.array_equal(o1["inv_depth"], o2["inv_depth"])  # the winner's depths are mode 1's replay on both sides
    assert np.allclose(o1["trial_err"], o2["trial_err"], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_mode2_dense_depth_solve_takes_mode1s_decisions(golden, oracle, case):
    g = lambda k: golden[case + "/" + k]
    q, u, a, ak = g("q"), g("u"), g("alpha"), g("alpha_k")
    for t in range(len(g("hyp_v"))):
        v, w, k = g("hyp_v")[t], g("hyp_w")[t], float(g("hyp_k")[t]) if "%s/hyp_k" % case in golden else 0.0
        r1, s1 = oracle.estimate_inverse_depths(q, u, v, w, k, a, ak, mode=1)
        r2, s2 = oracle.estimate_inverse_depths(q, u, v, w, k, a, ak, mode=2)
        for key in INT_KEYS:
            assert s1[key] == s2[key], (key, s1, s2)
        assert np.allclose(r1, r2, rtol=1e-9, atol=1e-13)
        # (noise-free cases end at the rounding floor of the cost, ~1e-24 of the initial cost: absolute term)
        assert abs(s1["final_cost"] - s2["final_cost"]) <= 1e-11 * abs(s1["final_cost"]) + 1e-20 * s1["initial_cost"]


@pytest.mark.parametrize("case", ["noisy_k0", "deepflow_k0", "noisy_k04"])
def test_mode2_ransac_golden_cases(golden, oracle, case):
    g = lambda k: golden[case + "/" + k]
    q, u, a, ak, samples = g("q"), g("u"), g("alpha"), g("alpha_k"), g("samples")
    use_k = bool(g("use_k"))
    o1 = oracle.ransac(q, u, a, ak, use_k, len(samples), 0.05, samples, depth_mode=1)
    o2 = oracle.ransac(q, u, a, ak, use_k, len(samples), 0.05, samples, depth_mode=2)
    _same_ransac(o1, o2)
    assert oracle.lma_last_stats()["fallback"] == 0
    assert np.array_equal(o2["trial_count"], g("count_lm"))


@pytest.mark.parametrize("tol", [0.05, 0.002, 0.0005])
def test_mode2_selective_tolerances_and_the_study(oracle, rsdsfm, tol):
    """DeepFlow-like data; the study runs mode 1's recurrence for EVERY pixel beside the closed form: no inlier decision differs, and the
    largest difference of the errors stays > 100 x below what guard (b) covers"""
    d = rsdsfm.synth.make_config(3, rows=120, cols=200)
    q, u, a, ak = d["q"], d["u"], d["alpha"], d["alpha_k"]
    T = 12
    samples = oracle.sample_indices(len(q), T, 77)
    o1 = oracle.ransac(q, u, a, ak, False, T, tol, samples, depth_mode=1)
    o2 = oracle.ransac(q, u, a, ak, False, T, tol, samples, depth_mode=2)
    _same_ransac(o1, o2)
    assert oracle.lma_last_stats()["fallback"] == 0
    kappa = 0.0
    for t in range(T):
        tv = o1["trial_vel"][t]
        if not np.all(np.isfinite(tv)):
            continue
        r = oracle.lma_trial(q, u, a, ak, tv[3:6], tv[0:3], tv[6], tol, study=True)
        assert r["stats"]["flips_unguarded"] == 0
        assert r["count"] == o1["trial_count"][t] and r["summary"]["num_successful_steps"] == o1["trial_steps"][t]
        kappa = max(kappa, r["stats"]["margin_use_max"])
    assert kappa < 5e-14  # guard (b) holds while kappa < eta / 2 = 5e-12


def test_mode2_clamped_pixels_walk_the_exact_recurrence(oracle, rsdsfm):
    """forward motion: the focus of expansion lies inside the image, the pixels around it have a Jacobian below the LM diagonal's clamp
    (guard a) -- mode 2 lists them and still takes every decision of mode 1"""
    d = rsdsfm.synth.make_config(3, rows=180, cols=320, v=np.array([0.002, 0.001, 0.03]))
    q, u, a, ak = d["q"], d["u"], d["alpha"], d["alpha_k"]
    T = 10
    samples = oracle.sample_indices(len(q), T, 5)
    o1 = oracle.ransac(q, u, a, ak, False, T, 0.01, samples, depth_mode=1)
    o2 = oracle.ransac(q, u, a, ak, False, T, 0.01, samples, depth_mode=2)
    st = oracle.lma_last_stats()
    _same_ransac(o1, o2)
    assert st["listed_clamped"] > 0 and st["fallback"] == 0


def test_mode2_noise_free_ties_fall_back_to_mode1(oracle, rsdsfm):
    """noise-free data: every good hypothesis explains every pixel and the error sums are rounding noise -- a tie only the reference's own
    arithmetic can break (guard d): mode 2 hands the whole RANSAC to mode 1"""
    d = rsdsfm.synth.make_config(1, rows=48, cols=64)
    q, u, a, ak = d["q"], d["u"], d["alpha"], d["alpha_k"]
    T = 8
    samples = oracle.sample_indices(len(q), T, 11)
    o1 = oracle.ransac(q, u, a, ak, False, T, 0.05, samples, depth_mode=1)
    o2 = oracle.ransac(q, u, a, ak, False, T, 0.05, samples, depth_mode=2)
    st = oracle.lma_last_stats()
    assert st["fallback"] >= 1 and st["fallback_reason"] == 7
    _same_ransac(o1, o2)
    assert np.array_equal(o1["trial_err"], o2["trial_err"])


def test_mode2_nan_hypothesis_and_nan_pixel(oracle, rsdsfm):
    """a NaN pose ends at iteration zero with no inlier (the gradient maximum drops NaNs), a single NaN pixel makes five invalid steps"""
    d = rsdsfm.synth.make_config(3, rows=60, cols=80)
    q, u, a, ak = d["q"], d["u"], d["alpha"], d["alpha_k"]
    v, w = np.array([np.nan, 0.0, 1.0]), np.zeros(3)
    r = oracle.lma_trial(q, u, a, ak, v, w, 0.0, 0.05)
    r1, s1 = oracle.estimate_inverse_depths(q, u, v, w, 0.0, a, ak, mode=1)
    assert r["count"] == 0 and all(r["summary"][k] == s1[k] for k in INT_KEYS)
    u2 = u.copy()
    u2[17, 0] = np.nan
    v = np.array([0.6, 0.8, 0.0])
    r = oracle.lma_trial(q, u2, a, ak, v, w, 0.0, 0.05)
    r1, s1 = oracle.estimate_inverse_depths(q, u2, v, w, 0.0, a, ak, mode=1)
    assert all(r["summary"][k] == s1[k] for k in INT_KEYS) and s1["termination"] == 4
    c1, e1, m1 = oracle.score(q, u2, a, ak, v, w, 0.0, r1, 0.05)
    assert r["count"] == c1 and np.array_equal(r["mask"], m1)


def test_mode2_random_cases(oracle, rsdsfm):
    """the random cases of tests/fuzz_gpu.py, mode 2 against mode 1 (tools/lma_cpu_fuzz.py runs the same generator at length)"""
    bad = 0
    for c in range(120):
        rng = np.random.default_rng(424242 * 100003 + c)
        rows, cols = int(rng.integers(9, 60)), int(rng.integers(9, 90))
        cfg = int(rng.choice([1, 3]))
        v = rng.normal(size=3) * np.array([0.03, 0.03, 0.02])
        w = rng.normal(size=3) * 0.004
        k = float(rng.choice([0.0, 0.0, rng.uniform(-0.5, 0.8)]))
        d = rsdsfm.synth.make_config(cfg, seed=int(rng.integers(1 << 30)), v=v, w=w, k=k, rows=rows, cols=cols)
        q, u, a, ak = d["q"], d["u"], d["alpha"], d["alpha_k"]
        if len(q) < 9 or not (np.all(np.isfinite(q)) and np.all(np.isfinite(u))):
            continue
        T = int(rng.choice([1, 3, 8, 20]))
        tol = float(rng.choice([0.05, 0.01, 0.003, 0.001]))
        use_k = bool(rng.integers(2)) and k != 0.0
        samples = oracle.sample_indices(len(q), T, int(rng.integers(1 << 30)))
        o1 = oracle.ransac(q, u, a, ak, use_k, T, tol, samples, depth_mode=1)
        o2 = oracle.ransac(q, u, a, ak, use_k, T, tol, samples, depth_mode=2)
        try:
            _same_ransac(o1, o2)
        except AssertionError:
            bad += 1
    assert bad == 0
