"""oracle/rsdsfm_cpu_ref.cpp -- the reference-STRUCTURED single-thread CPU baseline (BASELINE.md section 3.1 `cpu_ref`: a problem built per
call from per-pixel heap objects, a Schur ordering pass, dual-number Jacobians behind a virtual call) -- computes what the oracle
computes: same decisions, values to the rounding of automatic vs analytic derivatives.  Only then is its run time a baseline."""
import time

import numpy as np
import pytest

from conftest import GOLDEN_CASES


def test_structured_depth_solve_equals_oracle(oracle, rsdsfm):
    for cfg in (1, 3):
        d = rsdsfm.synth.make_config(cfg, rows=60, cols=90)
        t = d["truth"]
        v = t["v"] / np.linalg.norm(t["v"])
        rho_o, sm_o = oracle.estimate_inverse_depths(d["q"], d["u"], v, t["w"], 0.0, d["alpha"], d["alpha_k"], mode=1)
        rho_r, sm_r = oracle.estimate_inverse_depths_reference_structured(d["q"], d["u"], v, t["w"], 0.0, d["alpha"], d["alpha_k"])
        for key in ("num_iterations", "num_successful_steps", "num_unsuccessful_steps", "termination"):
            assert sm_r[key] == sm_o[key], (cfg, key)
        assert np.allclose(rho_r, rho_o, rtol=1e-10, atol=1e-14)
        assert np.isclose(sm_r["final_cost"], sm_o["final_cost"], rtol=1e-9, atol=1e-25)
    rho_e, sm_e = oracle.estimate_inverse_depths_reference_structured(np.zeros((0, 2)), np.zeros((0, 2)), v, t["w"], 0.0, np.zeros(0), np.zeros(0))
    assert rho_e.shape == (0,) and sm_e["termination"] == 0


@pytest.mark.parametrize("accel", [False, True])
def test_structured_ransac_and_refinement_equal_oracle(oracle, rsdsfm, accel):
    d = rsdsfm.synth.make_config(3, rows=48, cols=80, k=0.3 if accel else 0.0)
    q, u, a, ak = d["q"], d["u"], d["alpha"], d["alpha_k"]
    T, tol = 6, 0.004
    smp = oracle.sample_indices(len(q), T, 5)
    ro = oracle.ransac(q, u, a, ak, accel, T, tol, smp, depth_mode=1)
    rr = oracle.ransac_reference_structured(q, u, a, ak, accel, T, tol, smp)
    assert rr["num_inliers"] == ro["num_inliers"] and rr["best_trial"] == ro["best_trial"]
    assert np.array_equal(rr["trial_count"], ro["trial_count"]) and np.array_equal(rr["trial_steps"], ro["trial_steps"])
    assert np.array_equal(rr["mask"], ro["mask"]) and np.array_equal(rr["inlier_idx"], ro["inlier_idx"])
    assert np.allclose(rr["inliers"], ro["inliers"], rtol=1e-9) and np.array_equal(rr["v"], ro["v"]) and np.array_equal(rr["w"], ro["w"])
    for mode, idx in ((1, ro["inlier_idx"]), (0, None)):
        fo = oracle.refine(u, ro["inliers"], ro["alpha"], ro["alpha_k"], ro["v"], ro["w"], ro["k"], accel, mode, idx)
        fr = oracle.refine_reference_structured(u, ro["inliers"], ro["alpha"], ro["alpha_k"], ro["v"], ro["w"], ro["k"], accel, mode, idx)
        for key in ("num_iterations", "num_successful_steps", "num_unsuccessful_steps", "termination"):
            assert fr["summary"][key] == fo["summary"][key], (mode, key)
        assert np.allclose(fr["v"], fo["v"], rtol=1e-7, atol=1e-12) and np.allclose(fr["w"], fo["w"], rtol=1e-7, atol=1e-12)
        assert np.isclose(fr["k"], fo["k"], rtol=1e-7, atol=1e-12) and np.allclose(fr["inliers"], fo["inliers"], rtol=1e-6)


def test_structure_costs_what_it_should(oracle, rsdsfm):
    """the point of the exercise: the same arithmetic takes several times longer when it is organised as the reference organises it"""
    d = rsdsfm.synth.make_config(5, rows=180, cols=320)
    t = d["truth"]
    v = t["v"] / np.linalg.norm(t["v"])
    args = (d["q"], d["u"], v, t["w"], 0.0, d["alpha"], d["alpha_k"])
    oracle.estimate_inverse_depths(*args, mode=1), oracle.estimate_inverse_depths_reference_structured(*args)
    t0 = time.perf_counter()
    for _ in range(3):
        oracle.estimate_inverse_depths(*args, mode=1)
    t_port = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(3):
        oracle.estimate_inverse_depths_reference_structured(*args)
    t_ref = time.perf_counter() - t0
    assert t_ref > 2.0 * t_port, (t_ref, t_port)
