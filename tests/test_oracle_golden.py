"""CPU: the oracle (oracle/rsdsfm_oracle.c) against the committed golden fixtures, which come from an
independent numpy/scipy transcription (tests/golden/make_golden.py).  Float tolerance stated per check."""
import numpy as np
import pytest

from conftest import GOLDEN_CASES


def canon(v, w):
    """sign-canonical (v, w): the SVD null-vector sign is implementation-defined."""
    i = int(np.argmax(np.abs(v)))
    return (v if v[i] > 0 else -v), w


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_calculate_velocities(golden, oracle, case):
    g = lambda k: golden[case + "/" + k]
    q, u, a, ak, samples = g("q"), g("u"), g("alpha"), g("alpha_k"), g("samples")
    use_k = bool(g("use_k"))
    for t in range(len(samples)):
        s = samples[t]
        w, v, k, rc = oracle.calculate_velocities(q[s], u[s], a[s], ak[s], use_k)
        assert rc == 0
        v_ref, w_ref = canon(g("hyp_v")[t], g("hyp_w")[t])
        v_c, w_c = canon(v, w)
        # tolerance: 1e-8 absolute on unit-norm v and on w (|w| ~ 1e-2); conditioning of a 9-point sample
        assert np.allclose(v_c, v_ref, atol=1e-8), (t, v_c, v_ref)
        assert np.allclose(w_c, w_ref, atol=1e-8), (t, w_c, w_ref)
        assert abs(k - g("hyp_k")[t]) <= 1e-7 * max(1.0, abs(k))


def test_linalg_pieces(golden, oracle):
    for Z, sv_ref, vl_ref in zip(golden["linalg/svd_in"], golden["linalg/svd_sv"], golden["linalg/svd_vlast"]):
        sv, V = oracle.jacobi_svd9(Z)
        assert np.allclose(sv, sv_ref, rtol=1e-12, atol=1e-13)
        vl = V[:, 8]
        assert min(np.abs(vl - vl_ref).max(), np.abs(vl + vl_ref).max()) < 1e-10
        assert np.allclose(V.T @ V, np.eye(9), atol=1e-13)
    for G, ev_ref in zip(golden["linalg/eig_in"], golden["linalg/eig_vals_sorted"]):
        ev = np.sort_complex(oracle.eigvals_general(G))
        assert np.allclose(ev, ev_ref, rtol=1e-10, atol=1e-11)
    for S, lam_ref in zip(golden["linalg/sym_in"], golden["linalg/sym_vals"]):
        lam, V = oracle.eig_sym3(S)
        assert np.allclose(lam, lam_ref, rtol=1e-13, atol=1e-14)
        assert np.allclose(V @ np.diag(lam) @ V.T, S, atol=1e-13)


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_depth_and_score(golden, oracle, case):
    g = lambda k: golden[case + "/" + k]
    q, u, a, ak = g("q"), g("u"), g("alpha"), g("alpha_k")
    W, V, K = g("hyp_w"), g("hyp_v"), g("hyp_k")
    for t in range(len(W)):
        rho0, _ = oracle.estimate_inverse_depths(q, u, V[t], W[t], K[t], a, ak, mode=0)
        rho1, sm = oracle.estimate_inverse_depths(q, u, V[t], W[t], K[t], a, ak, mode=1)
        ref_sm = g("lm_summary")[t]
        # LM trajectory decisions are integers: exact
        assert sm["num_iterations"] == int(ref_sm[0])
        assert sm["num_successful_steps"] == int(ref_sm[1])
        assert sm["num_unsuccessful_steps"] == int(ref_sm[2])
        assert sm["termination"] == int(ref_sm[3])
        assert np.isclose(sm["initial_cost"], ref_sm[4], rtol=1e-12)
        assert np.isclose(sm["final_radius"], ref_sm[6], rtol=1e-15)
        if t < 3:
            # depth: 1e-9 relative (north-star bar is 1e-5)
            assert np.allclose(rho0, g("rho_cf")[t], rtol=1e-9, atol=1e-12)
            assert np.allclose(rho1, g("rho_lm")[t], rtol=1e-9, atol=1e-12)
        c0, e0, _ = oracle.score(q, u, a, ak, V[t], W[t], K[t], rho0, 0.05)
        c1, e1, _ = oracle.score(q, u, a, ak, V[t], W[t], K[t], rho1, 0.05)
        assert c0 == int(g("count_cf")[t]) and c1 == int(g("count_lm")[t])  # integer: bit-exact
        assert np.isclose(e0, g("err_cf")[t], rtol=1e-9) and np.isclose(e1, g("err_lm")[t], rtol=1e-9)


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_ransac(golden, oracle, case):
    g = lambda k: golden[case + "/" + k]
    q, u, a, ak, samples = g("q"), g("u"), g("alpha"), g("alpha_k"), g("samples")
    r = oracle.ransac(q, u, a, ak, bool(g("use_k")), len(samples), 0.05, samples, depth_mode=1)
    assert np.array_equal(r["trial_count"], g("count_lm"))
    cnt, err = g("count_lm"), g("err_lm")
    top = np.nonzero(cnt == cnt.max())[0]
    # the tie-break on the (floating) error sum is only defined when the sums differ by more than rounding
    # noise (noise-free data: all trials have every point as inlier and error sums ~1e-10)
    decisive = len(top) == 1 or np.sort(err[top])[1] - np.sort(err[top])[0] > 1e-9 * max(err[top].max(), 1e-300) + 1e-12
    assert r["best_trial"] in top
    if decisive:
        assert r["best_trial"] == int(g("best"))
        assert np.array_equal(r["mask"], g("best_mask"))
        assert np.array_equal(r["inlier_idx"], np.nonzero(g("best_mask"))[0])
    assert r["num_inliers"] == int(g("best_mask").sum()) == int(r["mask"].sum())
    # the LM trajectory starts at rho = 1, so it is NOT invariant to the (implementation-defined) sign of the
    # SVD null vector: compare accepted-step counts only for trials where oracle and fixture agree on sign(v)
    same_sign = np.einsum("ij,ij->i", r["trial_vel"][:, 3:6], g("hyp_v")) > 0
    assert np.array_equal(r["trial_steps"][same_sign], g("lm_summary")[:, 1].astype(np.int32)[same_sign])
    r0 = oracle.ransac(q, u, a, ak, bool(g("use_k")), len(samples), 0.05, samples, depth_mode=0)
    assert np.array_equal(r0["trial_count"], g("count_cf"))


@pytest.mark.parametrize("case", GOLDEN_CASES)
@pytest.mark.parametrize("mode", ["compat", "gather"])
def test_refine(golden, oracle, case, mode):
    g = lambda k: golden[case + "/" + k]
    q, u, a, ak = g("q"), g("u"), g("alpha"), g("alpha_k")
    use_k = bool(g("use_k"))
    # inputs of the refinement exactly as the fixture had them: the fixture's best hypothesis (its SVD sign),
    # its inlier mask, and the LM depths of that hypothesis (checked to 1e-9 in test_depth_and_score)
    b = int(g("best"))
    v, w, k = g("hyp_v")[b], g("hyp_w")[b], float(g("hyp_k")[b])
    mask = g("best_mask").astype(bool)
    rho, _ = oracle.estimate_inverse_depths(q, u, v, w, k, a, ak, mode=1)
    inl = np.stack([q[mask, 0], q[mask, 1], 1.0 / rho[mask]], axis=1)
    out = oracle.refine(u, inl, a[mask], ak[mask], v, w, k, use_k,
                        flow_index_mode=0 if mode == "compat" else 1, inlier_idx=np.nonzero(mask)[0])
    ref_sm = g("ref_%s_summary" % mode)
    sm = out["summary"]
    assert sm["num_iterations"] == int(ref_sm[0]) and sm["termination"] == int(ref_sm[3])
    assert sm["num_successful_steps"] == int(ref_sm[1])
    # the oracle eliminates rho_i by Schur complement, the fixture solved the full dense normal equations:
    # agreement 1e-7 relative after up to 50 LM iterations (north-star bar is 1e-5)
    sgn = np.sign(out["v"] @ g("ref_%s_v" % mode))
    assert sgn == 1.0
    assert np.allclose(out["v"], g("ref_%s_v" % mode), rtol=1e-7, atol=1e-10)
    assert np.allclose(out["w"], g("ref_%s_w" % mode), rtol=1e-7, atol=1e-10)
    assert np.isclose(out["k"], g("ref_%s_k" % mode), rtol=1e-7, atol=1e-10)
    assert np.allclose(out["inliers"][:, 2], g("ref_%s_z" % mode), rtol=1e-7)
    assert np.isclose(sm["final_cost"], ref_sm[5], rtol=1e-8)
    # the oracle's iteration trace (the checker of the product's rsdsfm_get_refine_trace) is consistent with its own summary,
    # follows Ceres' radius rule, and tracing does not change the result
    tr = oracle.refine(u, inl, a[mask], ak[mask], v, w, k, use_k, flow_index_mode=0 if mode == "compat" else 1,
                       inlier_idx=np.nonzero(mask)[0], trace_rows=50)
    assert tr["summary"] == sm and np.array_equal(tr["inliers"], out["inliers"])
    t = tr["trace"]
    n_it = sm["num_iterations"]
    assert np.array_equal(t[:n_it, 0], np.arange(1, n_it + 1)) and np.isnan(t[n_it:]).all()
    assert int(np.isin(t[:n_it, 7], (1.0, 5.0)).sum()) == sm["num_successful_steps"]
    assert int(np.isin(t[:n_it, 7], (0.0, 2.0)).sum()) == sm["num_unsuccessful_steps"]
    assert t[0, 1] == sm["initial_cost"] and t[0, 5] == 1e4
    for i in range(n_it - 1):
        if t[i, 7] == 1.0:  # accepted: cost moves to the candidate's, radius /= max(1/3, 1 - (2 rel - 1)^3)
            assert t[i + 1, 1] == t[i, 2]
            assert np.isclose(t[i + 1, 5], min(t[i, 5] / max(1.0 / 3.0, 1.0 - (2.0 * t[i, 4] - 1.0) ** 3), 1e16), rtol=1e-14)
            assert t[i, 4] > 1e-3
        elif t[i, 7] == 0.0:
            assert t[i + 1, 1] == t[i, 1] and t[i + 1, 5] < t[i, 5] and not t[i, 4] > 1e-3
