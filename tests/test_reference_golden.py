"""Reference-pinned parity: the oracle (CPU, `-m "not gpu"`) and the HIP path (`-m gpu`) against OUTPUTS OF THE REFERENCE ITSELF --
`tests/golden/reference_v1.npz`, produced by tools/pin_reference (the reference's own minimal.cc / nonlinearRefinement.cc compiled
against Ceres 1.14.0 / Eigen 3.3.4 and run on the committed inputs of tests/golden/golden_v1.npz with the committed sample sets
injected).  That toolchain does not exist in the development image of this repository (no Ceres, Eigen, network), so the fixture is
ABSENT there and every test of this file SKIPS with the message below; DESIGN.md keeps saying "parity unpinned" until someone with
the toolchain runs `REFERENCE_DIR=... sh tools/pin_reference/run.sh` and commits the fixture.  Then this file is the five-minute check
of every recalled detail -- first of all the placement of Ceres' function-tolerance test (profiles/r03_oracle_sensitivity.md).

Each comparison feeds the implementation the REFERENCE's inputs of that stage (its hypotheses for the depth solve, its RANSAC result
for the refinement), so one stage's deviation does not leak into the next.  Bars: integers (iteration / step counts, termination,
inlier counts, inlier coordinates) exact; floats within the north star's 1e-5 relative, written at each assert.

RSDSFM_REFERENCE_GOLDEN=<path> points the tests at another fixture file (tools/pin_reference/selfcheck.py uses that to prove the
plumbing of this file with a stand-in built from the oracle -- never a substitute for the real fixture)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_CASES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = os.environ.get("RSDSFM_REFERENCE_GOLDEN") or os.path.join(ROOT, "tests", "golden", "reference_v1.npz")
SKIP = ("tests/golden/reference_v1.npz is absent: it holds outputs of the reference's own code (Ceres 1.14.0 / Eigen 3.3.4), which cannot be built "
        "in this image -- run `REFERENCE_DIR=<checkout> sh tools/pin_reference/run.sh` where that toolchain exists and commit the file")

pytestmark = pytest.mark.skipif(not os.path.exists(FIXTURE), reason=SKIP)

RTOL = 1e-5  # north star: "depth and (v, w) within 1e-5 relative"
TOL_RANSAC = 0.05  # tools/pin_reference/export_inputs.py (main.cc:305)


@pytest.fixture(scope="module")
def ref():
    return np.load(FIXTURE)


def _inputs(golden, case):
    g = lambda k: golden[case + "/" + k]
    return g("q"), g("u"), g("alpha"), g("alpha_k"), g("samples"), bool(g("use_k"))


def _check_counts(sm, r, ctx):
    """Ceres' decisions: exact.  r = the harness' summary row, whose counts come from summary.iterations behind iteration 0.  Ceres
    leaves the loop from INSIDE on the parameter / function tolerance, so the iteration that met it is not in that vector, while this
    repository's summaries count it (neither successful nor unsuccessful): num_iterations = entries + 1 for terminations 1 and 2."""
    inside = 1 if int(r[3]) in (1, 2) else 0
    assert sm["termination"] == int(r[3]), (ctx, sm, r)
    assert sm["num_successful_steps"] == int(r[1]) and sm["num_unsuccessful_steps"] == int(r[2]), (ctx, sm, r)
    assert sm["num_iterations"] == int(r[0]) + inside, (ctx, sm, r)
    return int(r[0])


def _inlier_idx(q, inl):
    """the inliers keep the order of the point list (minimal.cc:291-303): index of the point each one came from"""
    idx, j = [], 0
    for i in range(len(q)):
        if j < len(inl) and q[i, 0] == inl[j, 0] and q[i, 1] == inl[j, 1]:
            idx.append(i)
            j += 1
    assert j == len(inl)
    return np.array(idx, dtype=np.int64)


# ---------------------------------------------------------------------------------------------------
# the checks, written once; `impl` is the oracle module or a thin adapter over rsdsfm.Solver
# ---------------------------------------------------------------------------------------------------
def _check_minimal(impl, golden, ref, case):
    q, u, a, ak, samples, use_k = _inputs(golden, case)
    W, V, K = ref[case + "/hyp_w"], ref[case + "/hyp_v"], ref[case + "/hyp_k"]
    for t, s in enumerate(samples):
        if not (np.isfinite(W[t]).all() and np.isfinite(V[t]).all()):
            continue  # a degenerate sample in the reference
        w, v, k = impl.calculate_velocities(q[s], u[s], a[s], ak[s], use_k)
        # the SIGN of Eigen's null vector is part of what is pinned (minimal.cc:98-105: v = V.col(8) normalised, no sign rule)
        assert np.sign(v @ V[t]) == 1.0, (case, t, "sign of the SVD null vector differs from Eigen's")
        # 1e-8 absolute on the unit-norm v and on w (|w| ~ 1e-2): the conditioning of a 9-point sample, far inside 1e-5 relative
        assert np.allclose(v, V[t], atol=1e-8) and np.allclose(w, W[t], atol=1e-8), (case, t, v, V[t], w, W[t])
        assert abs(k - K[t]) <= RTOL * max(1.0, abs(K[t])), (case, t, k, K[t])


def _check_depth(impl, golden, ref, case, cost_floor=1e-300):
    """cost_floor (x initial cost): the absolute term of the final-cost comparison -- none for an implementation of the reference's arithmetic
    (the oracle, the HIP path's iterate-by-iterate kernels); the analytic LM trajectory (the HIP default) ends noise-free cases at ITS rounding
    floor, ~1e-24 of the initial cost"""
    q, u, a, ak, samples, use_k = _inputs(golden, case)
    W, V, K = ref[case + "/hyp_w"], ref[case + "/hyp_v"], ref[case + "/hyp_k"]
    rho_ref, sm_ref = ref[case + "/depth_rho"], ref[case + "/depth_summary"]
    worst = 0.0
    for t in range(len(samples)):
        if not (np.isfinite(W[t]).all() and np.isfinite(V[t]).all() and np.isfinite(rho_ref[t]).all()):
            continue
        rho, sm = impl.estimate_inverse_depths(q, u, V[t], W[t], K[t], a, ak)
        r = sm_ref[t]
        # THE lines that settle the function-tolerance question: one accepted step more or less shows here
        _check_counts(sm, r, (case, t))
        assert np.isclose(sm["initial_cost"], r[4], rtol=1e-9) and np.isclose(sm["final_cost"], r[5], rtol=1e-7, atol=cost_floor * r[4]), (case, t, sm, r)
        # inverse depths: 1e-5 relative (pixels with rho ~ 0 have no relative scale: absolute floor 1e-9)
        rel = np.abs(rho - rho_ref[t]) / np.maximum(np.abs(rho_ref[t]), 1e-4)
        worst = max(worst, float(rel.max()))
        assert rel.max() < RTOL, (case, t, rel.max(), int(rel.argmax()))
    return worst


def _check_ransac(impl, golden, ref, case):
    q, u, a, ak, samples, use_k = _inputs(golden, case)
    r = impl.ransac(q, u, a, ak, use_k, len(samples), TOL_RANSAC, samples)
    inl = ref[case + "/ransac_inliers"]
    wvk = ref[case + "/ransac_wvk"]
    assert r["num_inliers"] == int(ref[case + "/ransac_num_inliers"][0]) == len(inl)  # integer: exact
    assert np.array_equal(r["inliers"][:, :2], inl[:, :2])  # WHICH points are inliers: exact (x, y are copies of q)
    assert np.allclose(r["w"], wvk[:3], atol=1e-8) and np.allclose(r["v"], wvk[3:6], atol=1e-8) and abs(r["k"] - wvk[6]) <= RTOL * max(1.0, abs(wvk[6]))
    assert np.allclose(r["inliers"][:, 2], inl[:, 2], rtol=RTOL)  # 1 / rho of the winner
    assert np.array_equal(r["alpha"], ref[case + "/ransac_alpha"]) and np.array_equal(r["alpha_k"], ref[case + "/ransac_alpha_k"])


def _check_refine(impl, golden, ref, case, mode):
    q, u, a, ak, samples, use_k = _inputs(golden, case)
    inl, wvk = ref[case + "/ransac_inliers"], ref[case + "/ransac_wvk"]
    idx = _inlier_idx(q, inl)
    out = impl.refine(u, inl, ref[case + "/ransac_alpha"], ref[case + "/ransac_alpha_k"], wvk[3:6], wvk[:3], float(wvk[6]), use_k,
                      0 if mode == "compat" else 1, idx)
    p = "%s/refine_%s_" % (case, mode)
    r, o = ref[p + "summary"], ref[p + "wvk"]
    sm = out["summary"]
    pushed = _check_counts(sm, r, (case, mode))
    assert np.isclose(sm["initial_cost"], r[4], rtol=1e-9) and np.isclose(sm["final_cost"], r[5], rtol=1e-7), (case, mode, sm, r)
    # refined pose and depths: 1e-5 relative (w against its own norm: single components may vanish)
    assert np.linalg.norm(out["v"] - o[3:6]) <= RTOL * np.linalg.norm(o[3:6]) and np.linalg.norm(out["w"] - o[:3]) <= RTOL * np.linalg.norm(o[:3]), (case, mode)
    assert abs(out["k"] - o[6]) <= RTOL * max(1.0, abs(o[6]))
    assert np.allclose(out["inliers"][:, 2], ref[p + "z"], rtol=RTOL), (case, mode)
    # the cost after every iteration (Ceres' IterationSummary::cost) against the implementation's trace: same trajectory, not just the same end
    tr, tref = out.get("trace"), ref[p + "trace"]
    if tr is not None:
        n_it = pushed  # (the iteration that met the parameter / function tolerance has no row in Ceres' vector)
        cost_after = [tr[i, 2] if tr[i, 7] in (1.0, 5.0) else tr[i, 1] for i in range(n_it)]  # accepted: the candidate's cost; else unchanged
        assert np.allclose(cost_after, tref[1:n_it + 1, 1], rtol=1e-7), (case, mode, cost_after, tref[1:n_it + 1, 1])
        assert np.array_equal([1.0 if tr[i, 7] in (1.0, 5.0) else 0.0 for i in range(n_it)], tref[1:n_it + 1, 7])


# ---------------------------------------------------------------------------------------------------
# CPU: the oracle
# ---------------------------------------------------------------------------------------------------
class _OracleImpl:
    def __init__(self, O):
        self.O = O

    def calculate_velocities(self, q9, u9, a9, ak9, use_k):
        w, v, k, rc = self.O.calculate_velocities(q9, u9, a9, ak9, use_k)
        assert rc == 0
        return w, v, k

    def estimate_inverse_depths(self, q, u, v, w, k, a, ak):
        return self.O.estimate_inverse_depths(q, u, v, w, k, a, ak, mode=1)

    def ransac(self, q, u, a, ak, use_k, T, tol, samples):
        return self.O.ransac(q, u, a, ak, use_k, T, tol, samples, depth_mode=1)

    def refine(self, flow, inl, a, ak, v, w, k, use_k, flow_mode, idx):
        return self.O.refine(flow, inl, a, ak, v, w, k, use_k, flow_index_mode=flow_mode, inlier_idx=idx, trace_rows=64)


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_oracle_minimal_solver_vs_reference(oracle, golden, ref, case):
    _check_minimal(_OracleImpl(oracle), golden, ref, case)


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_oracle_depth_solve_vs_reference(oracle, golden, ref, case):
    _check_depth(_OracleImpl(oracle), golden, ref, case)


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_oracle_ransac_vs_reference(oracle, golden, ref, case):
    _check_ransac(_OracleImpl(oracle), golden, ref, case)


@pytest.mark.parametrize("mode", ["compat", "gather"])
@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_oracle_refinement_vs_reference(oracle, golden, ref, case, mode):
    _check_refine(_OracleImpl(oracle), golden, ref, case, mode)


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_oracle_first_lm_step_vs_ceres_in_ulps(oracle, golden, ref, case):
    """The last-bit question of the 1x1 e-blocks (Weak #1): Ceres' Schur eliminator inverts each e-block through InvertPSDMatrix
    (an LLT solve, then a MULTIPLY by the inverse), the oracle divides.  The harness lets real Ceres take ONE LM step for up to 100 pixels,
    each as a problem of its own; the oracle's one-step rho stands beside it.  Asserted: 1e-12 relative (far inside the 1e-5 bar); REPORTED
    (pytest -rA / -s): how many of the pixels agree bit for bit and the largest distance in ulps -- the number DESIGN.md section 5 quotes
    once the fixture exists."""
    key = case + "/one_step_rho"
    if key not in ref:
        pytest.skip("this fixture was produced by a harness without the one-step dump")
    q, u, a, ak, samples, use_k = _inputs(golden, case)
    t0 = int(ref[case + "/one_step_hypothesis"][0])
    if t0 < 0:
        pytest.skip("no finite hypothesis in this case")
    W, V, K = ref[case + "/hyp_w"], ref[case + "/hyp_v"], ref[case + "/hyp_k"]
    got = ref[key]
    n, m = len(q), len(got)
    mine = np.array([oracle.one_lm_step(q[i * n // m], u[i * n // m], a[i * n // m], ak[i * n // m], V[t0], W[t0], float(K[t0])) for i in range(m)])
    assert np.allclose(mine, got, rtol=1e-12, atol=1e-15), (case, np.abs(mine - got).max())
    ulps = np.abs(mine.view(np.int64) - got.view(np.int64))
    print("%s: first LM step, oracle vs Ceres: %d of %d pixels bit-identical, max distance %d ulp" % (case, int((ulps == 0).sum()), m, int(ulps.max())))


# ---------------------------------------------------------------------------------------------------
# GPU: the HIP path through the C ABI
# ---------------------------------------------------------------------------------------------------
class _HipImpl:
    def __init__(self, pkg, lm_arithmetic=0):
        self.s = pkg.Solver(0)
        self.lm_arithmetic = lm_arithmetic  # 0: the default path (analytic LM trajectory, radius-factorised refinement); 1: iterate by iterate
        self.s.set_lm_arithmetic(lm_arithmetic)

    def close(self):
        self.s.close()

    def calculate_velocities(self, q9, u9, a9, ak9, use_k):
        return self.s.calculate_velocities(q9, u9, a9, ak9, use_k)

    def estimate_inverse_depths(self, q, u, v, w, k, a, ak):
        return self.s.estimate_inverse_depths(q, u, v, w, k, a, ak, mode=1)

    def ransac(self, q, u, a, ak, use_k, T, tol, samples):
        return self.s.ransac(q, u, a, ak, use_k, T, tol, samples=samples, depth_mode=1)

    def refine(self, flow, inl, a, ak, v, w, k, use_k, flow_mode, idx):
        self.s.set_refine_trace(64)
        out = self.s.non_linear_refinement(flow, inl, a, ak, v, w, k, use_k, flow_index_mode=flow_mode, inlier_idx=idx)
        out["trace"] = self.s.get_refine_trace(64)
        self.s.set_refine_trace(0)
        return out


@pytest.fixture(params=[0, 1], ids=["default_arithmetic", "iterate_by_iterate"])
def hip(rsdsfm, request):
    impl = _HipImpl(rsdsfm, request.param)
    yield impl
    impl.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_hip_minimal_solver_vs_reference(hip, golden, ref, case):
    _check_minimal(hip, golden, ref, case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_hip_depth_solve_vs_reference(hip, golden, ref, case):
    _check_depth(hip, golden, ref, case, cost_floor=1e-20 if hip.lm_arithmetic == 0 else 1e-300)


@pytest.mark.gpu
@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_hip_ransac_vs_reference(hip, golden, ref, case):
    _check_ransac(hip, golden, ref, case)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["compat", "gather"])
@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_hip_refinement_vs_reference(hip, golden, ref, case, mode):
    _check_refine(hip, golden, ref, case, mode)
