"""GPU parity: batched 9-point minimal solver and hypothesis-batched RANSAC (through the C ABI) against the
CPU oracle on identical inputs and identical injected samples.  Integer outputs (per-trial inlier counts, best
trial, inlier mask, inlier index list, accepted LM steps) bit-exact; floats 1e-9 relative (north-star: 1e-5)."""
import numpy as np
import pytest

from conftest import GOLDEN_CASES

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def solver(rsdsfm):
    s = rsdsfm.Solver(0)
    yield s
    s.close()


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_calculate_velocities_vs_oracle_and_golden(golden, oracle, solver, case):
    g = lambda k: golden[case + "/" + k]
    q, u, a, ak, samples = g("q"), g("u"), g("alpha"), g("alpha_k"), g("samples")
    use_k = bool(g("use_k"))
    W, V, K = solver.calculate_velocities(q[samples], u[samples], a[samples], ak[samples], use_k)
    for t in range(len(samples)):
        s = samples[t]
        wo, vo, ko, rc = oracle.calculate_velocities(q[s], u[s], a[s], ak[s], use_k)
        # same algorithm, same rotation order: the sign of the SVD null vector agrees as well
        assert np.allclose(V[t], vo, atol=1e-10), (t, V[t], vo)
        assert np.allclose(W[t], wo, atol=1e-10), (t, W[t], wo)
        assert abs(K[t] - ko) <= 1e-9 * max(1.0, abs(ko))
        # and the independent numpy fixture up to the null-vector sign
        sgn = np.sign(V[t] @ g("hyp_v")[t])
        assert np.allclose(sgn * V[t], g("hyp_v")[t], atol=1e-8)
        assert np.allclose(W[t], g("hyp_w")[t], atol=1e-8)
    # k_sign_mode fixed = negated k (quirk Q4)
    if use_k:
        W2, V2, K2 = solver.calculate_velocities(q[samples], u[samples], a[samples], ak[samples], True, k_sign_mode=1)
        assert np.allclose(K2, -K, rtol=1e-12)


@pytest.mark.parametrize("case", ["noisy_k0", "noisy_k04"])
def test_minimal9_wave_and_lane_variants_bit_identical(golden, solver, case):
    """few hypotheses run one per WAVE (shared 9x9 SVD), many run one per LANE: same arithmetic, identical bits"""
    g = lambda k: golden[case + "/" + k]
    q, u, a, ak, samples = g("q"), g("u"), g("alpha"), g("alpha_k"), g("samples")
    use_k = bool(g("use_k"))
    few = solver.calculate_velocities(q[samples], u[samples], a[samples], ak[samples], use_k)
    reps = 70  # 10 x 70 = 700 hypotheses > 2 x 256 CUs -> lane variant
    big = np.tile(samples, (reps, 1))
    many = solver.calculate_velocities(q[big], u[big], a[big], ak[big], use_k)
    for f, m in zip(few, many):
        m = np.asarray(m).reshape(reps, len(samples), -1)
        for r in range(reps):
            assert np.array_equal(np.asarray(f).reshape(len(samples), -1), m[r])


def test_minimal_known_answer(solver, rsdsfm):
    """noise-free model data: w_true, +-v_true/|v| recovered"""
    d = rsdsfm.synth.make_config(1, rows=96, cols=128, v=np.array([0.03, 0.02, 0.01]), w=np.array([0.002, -0.003, 0.0087]))
    rng = np.random.default_rng(5)
    idx = np.stack([rng.choice(len(d["q"]), 9, replace=False) for _ in range(130)])  # > 2 workgroups of 64 lanes
    W, V, K = solver.calculate_velocities(d["q"][idx], d["u"][idx], d["alpha"][idx], d["alpha_k"][idx], False)
    t = d["truth"]
    vt = t["v"] / np.linalg.norm(t["v"])
    ok = 0
    for i in range(len(idx)):
        sgn = np.sign(V[i] @ vt)
        if np.allclose(W[i], t["w"], atol=1e-7) and np.allclose(sgn * V[i], vt, atol=1e-6):
            ok += 1
    assert ok >= 120  # a few random 9-point samples are near-degenerate


def _compare_ransac(r, ro, rho_rtol=1e-9):
    assert np.array_equal(r["trial_count"], ro["trial_count"])
    assert np.array_equal(r["trial_steps"], ro["trial_steps"])
    assert np.allclose(r["trial_err"], ro["trial_err"], rtol=1e-9, atol=1e-12)
    assert np.allclose(r["trial_vel"], ro["trial_vel"], rtol=1e-8, atol=1e-10, equal_nan=True)  # (a degenerate sample gives a NaN hypothesis on both sides)
    assert r["best_trial"] == ro["best_trial"]
    assert r["num_inliers"] == ro["num_inliers"]
    assert np.array_equal(r["mask"], ro["mask"])
    assert np.array_equal(r["inlier_idx"], ro["inlier_idx"])
    assert np.allclose(r["inv_depth"], ro["inv_depth"], rtol=rho_rtol, atol=1e-13)
    assert np.allclose(r["inliers"], ro["inliers"], rtol=rho_rtol, atol=1e-13)
    assert np.array_equal(r["alpha"], ro["alpha"]) and np.array_equal(r["alpha_k"], ro["alpha_k"])
    assert np.allclose(r["w"], ro["w"], atol=1e-10) and np.allclose(r["v"], ro["v"], atol=1e-10)
    assert abs(r["k"] - ro["k"]) <= 1e-9 * max(1.0, abs(ro["k"]))


@pytest.mark.parametrize("case", ["noisy_k0", "deepflow_k0", "noisy_k04"])
@pytest.mark.parametrize("mode", [0, 1])
def test_ransac_golden_cases(golden, oracle, solver, case, mode):
    g = lambda k: golden[case + "/" + k]
    q, u, a, ak, samples = g("q"), g("u"), g("alpha"), g("alpha_k"), g("samples")
    use_k = bool(g("use_k"))
    r = solver.ransac(q, u, a, ak, use_k, len(samples), 0.05, samples=samples, depth_mode=mode)
    ro = oracle.ransac(q, u, a, ak, use_k, len(samples), 0.05, samples, depth_mode=mode)
    _compare_ransac(r, ro)
    assert np.array_equal(r["trial_count"], g("count_lm" if mode else "count_cf"))


@pytest.mark.parametrize("T", [1, 5, 50, 130])
def test_ransac_deepflow_like(oracle, solver, rsdsfm, T):
    """DeepFlow-like data (0.3 px noise, 10 % outliers), tight tolerance so that the inlier sets are selective;
    T = 130 exercises more than one hypothesis batch."""
    d = rsdsfm.synth.make_config(3, rows=135, cols=240)
    q, u, a, ak = d["q"], d["u"], d["alpha"], d["alpha_k"]
    samples = oracle.sample_indices(len(q), T, 1234)
    for mode in (0, 1):
        r = solver.ransac(q, u, a, ak, False, T, 0.002, samples=samples, depth_mode=mode)
        ro = oracle.ransac(q, u, a, ak, False, T, 0.002, samples, depth_mode=mode)
        _compare_ransac(r, ro)
        assert 0 < r["num_inliers"] < len(q)


@pytest.mark.parametrize("rows,cols,T", [(24, 32, 50), (40, 64, 17), (64, 100, 128)])
def test_ransac_hypothesis_groups_on_small_frames(oracle, solver, rsdsfm, rows, cols, T):
    """frames of a few pixel tiles (768 points < one 1280-point tile; exactly two tiles; five tiles) with many hypotheses: the
    hypothesis-batched LM kernel then splits the hypotheses over gridDim.y (up to T / 8 groups per tile) and the ragged-tile
    path carries the whole frame -- same trials, counts, best trial and inliers as the oracle"""
    d = rsdsfm.synth.make_config(3, rows=rows, cols=cols)
    q, u, a, ak = d["q"], d["u"], d["alpha"], d["alpha_k"]
    samples = oracle.sample_indices(len(q), T, 99)
    r = solver.ransac(q, u, a, ak, False, T, 0.004, samples=samples, depth_mode=1)
    ro = oracle.ransac(q, u, a, ak, False, T, 0.004, samples, depth_mode=1)
    _compare_ransac(r, ro)


def test_ransac_builtin_sampler_matches_reference_sampler(oracle, solver, rsdsfm):
    """samples=NULL: the library's sampler is the reference's partial Fisher-Yates (minimal.cc:226-244) driven by
    splitmix64(seed) -- identical to the oracle's restatement, so the whole run matches."""
    d = rsdsfm.synth.make_config(3, rows=90, cols=120)
    q, u, a, ak = d["q"], d["u"], d["alpha"], d["alpha_k"]
    T, seed = 12, 0xC0FFEE
    r = solver.ransac(q, u, a, ak, False, T, 0.003, samples=None, seed=seed, depth_mode=1)
    ro = oracle.ransac(q, u, a, ak, False, T, 0.003, oracle.sample_indices(len(q), T, seed), depth_mode=1)
    _compare_ransac(r, ro)


def test_ransac_edge_cases(oracle, solver, rsdsfm):
    d = rsdsfm.synth.make_config(1, rows=24, cols=32)
    q, u, a, ak = d["q"], d["u"], d["alpha"], d["alpha_k"]
    # fewer than 9 points: error, not UB (reference Q8: rand() % 0)
    with pytest.raises(rsdsfm.RsdsfmError):
        solver.ransac(q[:8], u[:8], a[:8], ak[:8], False, 3, 0.05, seed=1)
    # exactly 9 points
    s9 = np.arange(9, dtype=np.int32).reshape(1, 9)
    r = solver.ransac(q[:9], u[:9], a[:9], ak[:9], False, 1, 0.05, samples=s9, depth_mode=0)
    ro = oracle.ransac(q[:9], u[:9], a[:9], ak[:9], False, 1, 0.05, s9, depth_mode=0)
    _compare_ransac(r, ro)
    # zero tolerance: no inliers
    r = solver.ransac(q, u, a, ak, False, 2, 0.0, seed=3, depth_mode=1)
    assert r["num_inliers"] == 0 and r["mask"].sum() == 0 and len(r["inlier_idx"]) == 0
    # out-of-range injected sample index: rejected
    bad = np.full((1, 9), len(q), dtype=np.int32)
    with pytest.raises(rsdsfm.RsdsfmError):
        solver.ransac(q, u, a, ak, False, 1, 0.05, samples=bad)


def test_ransac_full_size_invariants(solver, rsdsfm):
    """1280x720, noise-free model data, T = 8: every trial explains every point (N inliers at tol 0.05,
    SURVEY 8c-1); permutation invariance of the scoring: reversing the point order gives the same counts."""
    d = rsdsfm.synth.make_config(2)
    q, u, a, ak = d["q"], d["u"], d["alpha"], d["alpha_k"]
    n = len(q)
    rng = np.random.default_rng(11)
    samples = np.stack([rng.choice(n, 9, replace=False) for _ in range(8)]).astype(np.int32)
    r = solver.ransac(q, u, a, ak, False, 8, 0.05, samples=samples, depth_mode=1)
    assert np.all(r["trial_count"] == n) and r["num_inliers"] == n
    assert np.array_equal(r["inlier_idx"], np.arange(n))
    rr = solver.ransac(q[::-1], u[::-1], a[::-1], ak[::-1], False, 8, 0.05, samples=(n - 1 - samples), depth_mode=1)
    assert np.array_equal(rr["trial_count"], r["trial_count"])
    assert np.allclose(rr["inv_depth"][::-1], r["inv_depth"], rtol=1e-12)


def test_ransac_multi_round_lm(oracle, solver, rsdsfm):
    """hypotheses whose dense LM solve needs more iterations than one speculative pass covers (10 accepted steps here: a
    pixel next to the epipole has a tiny Jacobian): continuation rounds of the hypothesis-batched kernel, the separate score
    pass, and a mix of finished / running hypotheses in one batch"""
    d = rsdsfm.synth.make_config(1, rows=40, cols=48, v=np.array([0.05, 0.03, 1.0]))
    q, u, a, ak = d["q"].copy(), d["u"].copy(), d["alpha"], d["alpha_k"]
    t = d["truth"]
    v = t["v"] / np.linalg.norm(t["v"])
    q[7] = [v[0] / v[2] + 1e-7, v[1] / v[2] - 2e-7]
    u[7] = [3e-2, -2e-2]
    T = 6
    smp = oracle.sample_indices(len(q), T, 3)
    smp[smp == 7] = 8
    r = solver.ransac(q, u, a, ak, False, T, 0.05, samples=smp, depth_mode=1)
    ro = oracle.ransac(q, u, a, ak, False, T, 0.05, smp, depth_mode=1)
    assert ro["trial_steps"].max() > 3  # more than KMAX accepted steps: at least two more rounds
    _compare_ransac(r, ro, rho_rtol=1e-8)
    # mixed batch: two near-degenerate samples (9 neighbours in one image column) give poor poses whose solves stop one
    # iteration earlier, so finished and still-running hypotheses share a round
    smp2 = smp.copy()
    smp2[1] = np.arange(100, 109)
    smp2[4] = np.arange(400, 409)
    r2 = solver.ransac(q, u, a, ak, False, T, 0.05, samples=smp2, depth_mode=1)
    ro2 = oracle.ransac(q, u, a, ak, False, T, 0.05, smp2, depth_mode=1)
    assert len(set(ro2["trial_steps"].tolist())) > 1
    _compare_ransac(r2, ro2, rho_rtol=1e-8)


@pytest.mark.parametrize("cfg,rows,cols,tol,noise", [(3, 135, 240, 0.002, None), (1, 60, 80, 0.05, None), (3, 90, 160, 0.05, 40.0)])
def test_ransac_speculation_depth_does_not_change_results(oracle, rsdsfm, cfg, rows, cols, tol, noise):
    """rsdsfm_set_ransac_speculation: round 0 of the hypothesis-batched LM solves speculates 3 iterations (+ fused scores of the one-
    and two-step iterates) or 2 (+ the one-step score).  Mixed step counts (DeepFlow-like data: mostly two accepted steps), noise-
    free data (2-3 steps) and outlier-dominated data (one step, decided by either depth in a single pass): every integer output is
    identical for both depths and equal to the oracle's, and so are the bits of every float -- the error sums included"""
    d = rsdsfm.synth.make_config(cfg, rows=rows, cols=cols)
    q, u, a, ak = d["q"], d["u"].copy(), d["alpha"], d["alpha_k"]
    if noise is not None:  # gross errors on a third of the points: the cost is dominated by them, every solve stops after one step
        rng = np.random.default_rng(5)
        idx = rng.choice(len(u), len(u) // 3, replace=False)
        u[idx] += rng.uniform(-noise, noise, size=(len(idx), 2)) * d["gamma"] / d["K"][0]
    T = 24
    samples = oracle.sample_indices(len(q), T, 77)
    ro = oracle.ransac(q, u, a, ak, False, T, tol, samples, depth_mode=1)
    outs = []
    with rsdsfm.Solver(0) as s:
        for k0 in (3, 2, 0):
            s.set_ransac_speculation(k0)
            r = s.ransac(q, u, a, ak, False, T, tol, samples=samples, depth_mode=1)
            _compare_ransac(r, ro)
            # trial_err included: whichever kernel forms the inlier-error sum of a hypothesis (round 0's fused score or the separate
            # scoring pass) adds the same numbers in the same order
            outs.append((r["trial_count"].tobytes(), r["trial_steps"].tobytes(), r["mask"].tobytes(), r["inv_depth"].tobytes(),
                         r["inliers"].tobytes(), r["best_trial"], r["trial_err"].tobytes()))
        with pytest.raises(rsdsfm.RsdsfmError):
            s.set_ransac_speculation(1)
    assert all(o == outs[0] for o in outs[1:])
    if noise is not None:
        assert ro["trial_steps"].max() == 1


def _ransac_bytes(r):
    return (r["trial_count"].tobytes(), r["trial_steps"].tobytes(), r["mask"].tobytes(), r["inv_depth"].tobytes(), r["inliers"].tobytes(),
            r["best_trial"], r["trial_err"].tobytes(), r["trial_vel"].tobytes(), r["inlier_idx"].tobytes())


@pytest.mark.parametrize("cfg,rows,cols,tol", [(3, 135, 240, 0.002), (1, 96, 128, 0.05), (5, 144, 256, 0.05), (5, 360, 640, 0.05)])
def test_ransac_function_cores_do_not_change_results(oracle, rsdsfm, cfg, rows, cols, tol):
    """rsdsfm_set_ransac_math: round 0 of the batched depth solves takes sqrt and the reciprocal through the in-range cores of the
    compiler's expansions (default) or through the standard functions.  Every output is identical bit for bit, equal to the oracle's,
    and on real-valued data no run has to start over."""
    d = rsdsfm.synth.make_config(cfg, rows=rows, cols=cols)
    q, u, a, ak = d["q"], d["u"], d["alpha"], d["alpha_k"]
    T = 24
    samples = oracle.sample_indices(len(q), T, 31337)
    ro = oracle.ransac(q, u, a, ak, False, T, tol, samples, depth_mode=1)
    outs = []
    with rsdsfm.Solver(0) as s:
        for mode in (0, 1, 0):
            s.set_ransac_math(mode)
            r = s.ransac(q, u, a, ak, False, T, tol, samples=samples, depth_mode=1)
            _compare_ransac(r, ro)
            outs.append(_ransac_bytes(r))
        assert s.ransac_restarts() == 0
        with pytest.raises(rsdsfm.RsdsfmError):
            s.set_ransac_math(2)
    assert outs[0] == outs[1] == outs[2]
    if (cfg, rows) == (5, 144):  # this draw holds a degenerate sample: a NaN hypothesis must not cost a restart (asserted above)
        assert np.isnan(ro["trial_vel"]).any()


@pytest.mark.parametrize("T", [16, 150])  # 150: two hypothesis batches (the flag words reach the host by a copy, not with the pick kernel)
@pytest.mark.parametrize("poison", ["zero_jacobian", "nan_flow", "zero_error", "tiny_jacobian"])
def test_ransac_function_cores_restart_on_arguments_out_of_range(oracle, rsdsfm, poison, T):
    """an argument outside the range of the in-range cores -- a non-finite flow, a Jacobian of 1e-160 (its square is a denormal) -- makes
    the run start over with the standard functions: results equal the oracle's and the standard-function setting's bit for bit, the
    restart is counted, and the context keeps the standard functions for its next runs.  An argument of EXACTLY zero -- a pixel whose
    Jacobian vanishes (alpha = alpha_k = 0: beta = 0), an error of exactly zero (ground-truth flow) -- is a select inside the cores since
    round 5 (sqrt_core_z): same bits, no restart."""
    d = rsdsfm.synth.make_config(5, rows=240, cols=320)  # 76800 points: 50 full tiles of 1536, the path the cores run on
    q, u, a, ak = d["q"].copy(), d["u"].copy(), d["alpha"].copy(), d["alpha_k"].copy()
    n = len(q)
    samples = oracle.sample_indices(n, T, 99)
    victim = 1536 * 7 + 100
    assert victim not in samples
    if poison == "zero_jacobian":
        a[victim] = 0.0
        ak[victim] = 0.0
    elif poison == "nan_flow":
        u[victim] = np.nan
    elif poison == "tiny_jacobian":
        a[victim] = 1e-160
        ak[victim] = 0.0
    else:
        # an error of exactly zero under every hypothesis: beta = 0 and u = 0 give e = beta (...) - u = 0 (and a zero Jacobian with it)
        a[victim] = 0.0
        ak[victim] = 0.0
        u[victim] = 0.0
    ro = oracle.ransac(q, u, a, ak, False, T, 0.05, samples, depth_mode=1)
    with rsdsfm.Solver(0) as s:
        expect = 1 if poison in ("nan_flow", "tiny_jacobian") else 0
        s.set_lm_arithmetic(1)  # the iterate-by-iterate kernels: the ones that run the cores
        r0 = s.ransac(q, u, a, ak, False, T, 0.05, samples=samples, depth_mode=1)
        assert s.ransac_restarts() == expect
        _compare_ransac(r0, ro)
        r1 = s.ransac(q, u, a, ak, False, T, 0.05, samples=samples, depth_mode=1)  # standard functions from the start: no second restart
        assert s.ransac_restarts() == expect
        s.set_ransac_math(1)
        r2 = s.ransac(q, u, a, ak, False, T, 0.05, samples=samples, depth_mode=1)
    assert _ransac_bytes(r0) == _ransac_bytes(r1) == _ransac_bytes(r2)
    # the default arithmetic (the analytic LM trajectory, csrc/lma_common.hpp) needs no restart of either kind for any of them: a pixel with a
    # vanishing Jacobian is one of the pixels it walks on the exact recurrence anyway (guard a), an error of exactly zero is a select, and
    # a NaN pixel poisons its sums exactly as it poisons the reference's -- same bits in every output
    with rsdsfm.Solver(0) as s:
        r3 = s.ransac(q, u, a, ak, False, T, 0.05, samples=samples, depth_mode=1)
        assert s.ransac_restarts() == 0 and s.lma_restarts()[0] == 0
    _compare_ransac(r3, ro)
    for key in ("trial_count", "trial_steps", "mask", "inlier_idx", "inv_depth", "inliers"):
        assert np.array_equal(r3[key], r0[key], equal_nan=True), key


def test_minimal_solver_function_cores_restart_on_operands_out_of_range(oracle, rsdsfm):
    """the wave-per-hypothesis SVD of the minimal solver runs its rotations through the in-range cores of division, reciprocal and
    square root; an operand outside their window makes the RANSAC start over with the standard functions.  Forced here with the first
    rotation's t = W(1,1) + W(0,0) = (u_x of the second sampled point - u_y of the first) / scale = 0 exactly; results equal the oracle's
    and the standard-function setting's bit for bit, one restart is counted."""
    d = rsdsfm.synth.make_config(5, rows=240, cols=320)
    q, u, a, ak = d["q"].copy(), d["u"].copy(), d["alpha"].copy(), d["alpha_k"].copy()
    T = 12
    samples = oracle.sample_indices(len(q), T, 4711)
    i0, i1 = int(samples[5][0]), int(samples[5][1])
    u[i1, 0] = u[i0, 1]
    ro = oracle.ransac(q, u, a, ak, False, T, 0.05, samples, depth_mode=1)
    with rsdsfm.Solver(0) as s:
        r0 = s.ransac(q, u, a, ak, False, T, 0.05, samples=samples, depth_mode=1)
        assert s.ransac_restarts() == 1
        _compare_ransac(r0, ro)
        s.set_ransac_math(1)
        r1 = s.ransac(q, u, a, ak, False, T, 0.05, samples=samples, depth_mode=1)
        assert s.ransac_restarts() == 1
    assert _ransac_bytes(r0) == _ransac_bytes(r1)
    # and the same draw without the planted zero does not restart
    with rsdsfm.Solver(0) as s:
        r2 = s.ransac(d["q"], d["u"], a, ak, False, T, 0.05, samples=samples, depth_mode=1)
        assert s.ransac_restarts() == 0
        _compare_ransac(r2, oracle.ransac(d["q"], d["u"], a, ak, False, T, 0.05, samples, depth_mode=1))
