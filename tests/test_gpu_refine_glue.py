"""GPU parity: joint nonlinear refinement, flatten, depth map against the CPU oracle (through the C ABI).
Refinement: trust-region decisions (iteration / step counters, termination) exact; v, w, k, z within 1e-6 relative
after up to 50 LM iterations (north-star bar 1e-5).  Glue: integer outputs (point count, pixel / scanline indices)
bit-exact, floats bit-exact (same operation order, no FMA contraction)."""
import numpy as np
import pytest

from conftest import GOLDEN_CASES

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def solver(rsdsfm):
    s = rsdsfm.Solver(0)
    yield s
    s.close()


def _check_refine(out, ref, rtol=1e-6):
    sm, smo = out["summary"], ref["summary"]
    for k in ("num_iterations", "num_successful_steps", "num_unsuccessful_steps", "termination"):
        assert sm[k] == smo[k], (k, sm, smo)
    assert np.isclose(sm["initial_cost"], smo["initial_cost"], rtol=1e-11)
    assert np.isclose(sm["final_cost"], smo["final_cost"], rtol=1e-7, atol=1e-25)
    assert np.allclose(out["v"], ref["v"], rtol=rtol, atol=1e-10)
    assert np.allclose(out["w"], ref["w"], rtol=rtol, atol=1e-10)
    assert np.isclose(out["k"], ref["k"], rtol=rtol, atol=1e-10)
    assert np.array_equal(out["inliers"][:, :2], ref["inliers"][:, :2])
    assert np.allclose(out["inliers"][:, 2], ref["inliers"][:, 2], rtol=rtol)


@pytest.mark.parametrize("case", GOLDEN_CASES)
@pytest.mark.parametrize("mode", [0, 1])
def test_refine_golden_cases(golden, oracle, solver, case, mode):
    g = lambda k: golden[case + "/" + k]
    q, u, a, ak = g("q"), g("u"), g("alpha"), g("alpha_k")
    use_k = bool(g("use_k"))
    b = int(g("best"))
    v, w, k = g("hyp_v")[b], g("hyp_w")[b], float(g("hyp_k")[b])
    mask = g("best_mask").astype(bool)
    rho, _ = oracle.estimate_inverse_depths(q, u, v, w, k, a, ak, mode=1)
    inl = np.stack([q[mask, 0], q[mask, 1], 1.0 / rho[mask]], axis=1)
    idx = np.nonzero(mask)[0]
    out = solver.non_linear_refinement(u, inl, a[mask], ak[mask], v, w, k, use_k, flow_index_mode=mode, inlier_idx=idx)
    ref = oracle.refine(u, inl, a[mask], ak[mask], v, w, k, use_k, flow_index_mode=mode, inlier_idx=idx)
    _check_refine(out, ref)
    # and the independent dense-LM fixture
    name = "ref_compat" if mode == 0 else "ref_gather"
    assert np.allclose(out["v"], g(name + "_v"), rtol=1e-6, atol=1e-9)
    assert np.allclose(out["inliers"][:, 2], g(name + "_z"), rtol=1e-6)


@pytest.mark.parametrize("const_acc", [False, True])
def test_refine_after_ransac_deepflow_like(oracle, solver, rsdsfm, const_acc):
    d = rsdsfm.synth.make_config(3, rows=180, cols=320)
    q, u, a, ak = d["q"], d["u"], d["alpha"], d["alpha_k"]
    samples = oracle.sample_indices(len(q), 20, 77)
    r = solver.ransac(q, u, a, ak, const_acc, 20, 0.002, samples=samples, depth_mode=1)
    assert r["num_inliers"] > 1000
    out = solver.non_linear_refinement(u, r["inliers"], r["alpha"], r["alpha_k"], r["v"], r["w"], r["k"], const_acc,
                                       flow_index_mode=1, inlier_idx=r["inlier_idx"])
    ref = oracle.refine(u, r["inliers"], r["alpha"], r["alpha_k"], r["v"], r["w"], r["k"], const_acc, flow_index_mode=1,
                        inlier_idx=r["inlier_idx"])
    _check_refine(out, ref)
    assert out["summary"]["final_cost"] <= out["summary"]["initial_cost"]


@pytest.mark.parametrize("const_acc", [False, True])
def test_refine_trace_matches_the_oracle_iteration_by_iteration(oracle, solver, rsdsfm, const_acc):
    """the whole trust-region trajectory, not only its end: every LM iteration's cost, candidate cost, model cost change, relative
    decrease, radius and step norm agree with the oracle's own trace (per-column bars below; the sums differ in summation order only) and every
    accept / reject / converge outcome is the same; rows past the last iteration stay NaN; switching the trace off costs nothing"""
    d = rsdsfm.synth.make_config(3, rows=180, cols=320)
    q, u, a, ak = d["q"], d["u"], d["alpha"], d["alpha_k"]
    samples = oracle.sample_indices(len(q), 20, 77)
    r = solver.ransac(q, u, a, ak, const_acc, 20, 0.002, samples=samples, depth_mode=1)
    args = (u, r["inliers"], r["alpha"], r["alpha_k"], r["v"], r["w"], r["k"], const_acc)
    plain = solver.non_linear_refinement(*args, flow_index_mode=1, inlier_idx=r["inlier_idx"])
    ROWS = 50
    solver.set_refine_trace(ROWS)
    try:
        out = solver.non_linear_refinement(*args, flow_index_mode=1, inlier_idx=r["inlier_idx"])
        tr = solver.get_refine_trace()
    finally:
        solver.set_refine_trace(0)
    ref = oracle.refine(*args, flow_index_mode=1, inlier_idx=r["inlier_idx"], trace_rows=ROWS)
    tro = ref["trace"]
    n_it = out["summary"]["num_iterations"]
    assert n_it == ref["summary"]["num_iterations"] and 2 <= n_it < ROWS
    assert tr.shape == tro.shape == (ROWS, rsdsfm.REFINE_TRACE_COLS)
    assert np.array_equal(tr[:n_it, 0], np.arange(1, n_it + 1)) and np.array_equal(tro[:n_it, 0], tr[:n_it, 0])
    assert np.array_equal(tr[:n_it, 7], tro[:n_it, 7]), (tr[:n_it, 7], tro[:n_it, 7])  # outcomes
    assert np.array_equal(np.isnan(tr), np.isnan(tro))
    assert np.isnan(tr[n_it:]).all()
    # per column: costs 1e-10 (measured 1e-13 .. 2e-11), model cost change / relative decrease 1e-7 (differences of nearly equal sums:
    # 2e-9), radius 1e-9, step norm 1e-3: the undamped system is flat along the scale gauge (|v| against the depths), so the step's
    # component along it is decided by the rounding of the Schur sums (measured 2e-5 at radius 5e9) while the cost does not see it
    dev = np.abs(tr[:n_it, 1:7] / tro[:n_it, 1:7] - 1.0)
    for col, bar in ((1, 1e-10), (2, 1e-10), (3, 1e-7), (4, 1e-7), (5, 1e-9), (6, 1e-3)):
        assert not (dev[:, col - 1] > bar).any(), (col, np.nanmax(dev[:, col - 1]))
    assert tr[0, 5] == 1e4 and tr[0, 1] == out["summary"]["initial_cost"]
    last = tr[n_it - 1, 7]
    assert last in (rsdsfm.TRACE_FUNCTION_TOL, rsdsfm.TRACE_PARAMETER_TOL, rsdsfm.TRACE_ACCEPTED_GRADIENT_TOL) or n_it == 50
    assert int(np.sum(np.isin(tr[:n_it, 7], (rsdsfm.TRACE_ACCEPTED, rsdsfm.TRACE_ACCEPTED_GRADIENT_TOL)))) == out["summary"]["num_successful_steps"]
    # tracing does not change the result
    assert out["summary"] == plain["summary"] and np.array_equal(out["inliers"], plain["inliers"]) and np.array_equal(out["v"], plain["v"])
    with pytest.raises(rsdsfm.RsdsfmError):
        solver.get_refine_trace(4)  # switched off


def test_refine_fixed_point_and_edge_cases(oracle, solver, rsdsfm):
    """noise-free model data + exact pose: refinement is a fixed point (gradient tolerance at iteration 0)."""
    d = rsdsfm.synth.make_config(1, rows=48, cols=64)
    q, u, a, ak, t = d["q"], d["u"], d["alpha"], d["alpha_k"], d["truth"]
    nv = np.linalg.norm(t["v"])
    v = t["v"] / nv
    z = t["Z"].T.reshape(-1) / nv
    inl = np.stack([q[:, 0], q[:, 1], z], axis=1)
    out = solver.non_linear_refinement(u, inl, a, ak, v, t["w"], 0.0, False)
    assert out["summary"]["termination"] == 0 and out["summary"]["num_iterations"] == 0
    assert np.allclose(out["v"], v, atol=1e-15) and np.allclose(out["inliers"][:, 2], z, rtol=1e-12)
    # m = 0
    out0 = solver.non_linear_refinement(u, np.zeros((0, 3)), np.zeros(0), np.zeros(0), v, t["w"], 0.0, False)
    assert out0["summary"]["termination"] == 0 and len(out0["inliers"]) == 0
    # compat mode with fewer flow columns than inliers: error, not an out-of-bounds read
    with pytest.raises(rsdsfm.RsdsfmError):
        solver.non_linear_refinement(u[:10], inl, a, ak, v, t["w"], 0.0, False)


def test_flatten_matches_reference_glue(oracle, solver, rsdsfm):
    d = rsdsfm.synth.make_config(3, rows=75, cols=130)
    img = d["flow_img"].copy()
    img[10:20, 5:9] = 0.0  # zero-flow pixels are dropped (main.cc:415-417)
    img[0, 0] = [1e-6, 0.0]  # |flow|^2 = 1e-12 <= 1e-10: dropped
    img[1, 0] = [2e-5, 0.0]  # 4e-10 > 1e-10: kept
    K, gamma = d["K"], d["gamma"]
    q, u, a, ak = solver.flatten(img, K, gamma)
    qo, uo, qpx, fpx = oracle.flatten(img, *K, gamma)
    assert len(q) == len(qo) == 75 * 130 - 41
    assert np.array_equal(q, qo) and np.array_equal(u, uo)
    assert np.array_equal(a, oracle.get_alpha(fpx, 75, gamma))
    assert np.array_equal(ak, oracle.get_alpha_k(qpx, fpx, 75, gamma))
    # all-zero image: nothing kept
    q0, _, _, _ = solver.flatten(np.zeros((8, 8, 2)), K, gamma)
    assert len(q0) == 0


@pytest.mark.parametrize("rows,cols", [(1, 1), (1, 9), (63, 7), (64, 8), (65, 9), (130, 17), (200, 2100), (2160, 3840)])
def test_flatten_ragged_tiles_and_sparse_images(oracle, solver, rsdsfm, rows, cols):
    """tile edges (tiles are 8 columns x 64 rows), more than one scan segment (> 2048 cells), sparse images (most pixels below
    the threshold, empty cells) and the 4K size: order and values bit-exact"""
    rng = np.random.default_rng(rows * 7 + cols)
    img = rng.normal(0, 1.0, (rows, cols, 2))
    img[rng.random((rows, cols)) < 0.7] = 0.0  # 70 % of the pixels dropped
    if rows > 3:
        img[rows // 2] = 0.0  # an empty image row
    if cols > 3:
        img[:, cols // 3] = 0.0  # an empty column: empty cells
    K, gamma = (300.0, 310.0, cols / 2.0 + 0.3, rows / 2.0 - 0.2), 0.9
    q, u, a, ak = solver.flatten(img, K, gamma)
    qo, uo, qpx, fpx = oracle.flatten(img, *K, gamma)
    assert len(q) == len(qo)
    assert np.array_equal(q, qo) and np.array_equal(u, uo)
    assert np.array_equal(a, oracle.get_alpha(fpx, rows, gamma)) and np.array_equal(ak, oracle.get_alpha_k(qpx, fpx, rows, gamma))


def test_depth_map_and_scanline_indices(oracle, solver, rsdsfm):
    d = rsdsfm.synth.make_config(1, rows=60, cols=80)
    q, t, K = d["q"], d["truth"], d["K"]
    z = t["Z"].T.reshape(-1)
    rng = np.random.default_rng(3)
    sel = np.sort(rng.choice(len(q), 3000, replace=False))
    for sign in (1.0, -1.0):
        inl = np.stack([q[sel, 0], q[sel, 1], sign * z[sel]], axis=1)
        v = np.array([0.3, -0.2, 0.1])
        out = solver.depth_map(inl, v, K, 60, 80)
        inl_o, v_o, flipped = oracle.canonicalize_sign(inl, v)
        dm_o, xs_o, ys_o = oracle.scatter_depth(inl_o, *K, 60, 80)
        assert out["flipped"] == flipped == (sign < 0)
        assert np.array_equal(out["v"], v_o) and np.array_equal(out["inliers"], inl_o)
        assert np.array_equal(out["xs"], xs_o) and np.array_equal(out["ys"], ys_o)  # scanline indices: bit-exact
        assert np.array_equal(out["depth_map"], dm_o)
        # the pixel indices are those of the original pixels
        assert np.array_equal(out["xs"], (d["pix"][sel] // 60).astype(np.int32))
        assert np.array_equal(out["ys"], (d["pix"][sel] % 60).astype(np.int32))
    # duplicate targets: the highest index wins (sequential last-writer semantics of main.cc:499-508)
    inl = np.array([[q[5, 0], q[5, 1], 1.0], [q[5, 0], q[5, 1], 2.0], [q[9, 0], q[9, 1], 3.0]])
    out = solver.depth_map(inl, np.ones(3), K, 60, 80)
    dm_o, _, _ = oracle.scatter_depth(inl, *K, 60, 80)
    assert np.array_equal(out["depth_map"], dm_o) and out["depth_map"][out["ys"][1], out["xs"][1]] == 2.0
    # out-of-image points are skipped
    inl = np.array([[10.0, 10.0, 1.0], [q[0, 0], q[0, 1], 4.0]])
    out = solver.depth_map(inl, np.ones(3), K, 60, 80)
    assert out["depth_map"].sum() == 4.0
