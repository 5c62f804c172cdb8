"""CPU: the oracle's restatement of the product's default refinement arithmetic (rso_refine_rf: radius-factorised Schur sums, mode 2) against the
reference's arithmetic (rso_refine, mode 1) -- the same Ceres loop stated twice: every integer must agree, the floats to the noise the flat
scale gauge leaves (1e-6 bar, as the GPU tests hold the HIP path to), and the guards must trip where the design says they trip."""
import numpy as np
import pytest

from conftest import GOLDEN_CASES

INTS = ("num_iterations", "num_successful_steps", "num_unsuccessful_steps", "termination")


def _same(r1, r2, rtol=1e-6):
    assert r2["guard"] == 0, r2["guard"]
    for k in INTS:
        assert r1["summary"][k] == r2["summary"][k], (k, r1["summary"], r2["summary"])
    assert np.isclose(r1["summary"]["initial_cost"], r2["summary"]["initial_cost"], rtol=1e-12)
    assert np.isclose(r1["summary"]["final_cost"], r2["summary"]["final_cost"], rtol=1e-8, atol=1e-25)
    assert np.allclose(r1["v"], r2["v"], rtol=rtol, atol=1e-10) and np.allclose(r1["w"], r2["w"], rtol=rtol, atol=1e-10)
    assert np.isclose(r1["k"], r2["k"], rtol=rtol, atol=1e-10)
    assert np.array_equal(r1["inliers"][:, :2], r2["inliers"][:, :2])
    assert np.allclose(r1["inliers"][:, 2], r2["inliers"][:, 2], rtol=rtol)


@pytest.mark.parametrize("case", GOLDEN_CASES)
@pytest.mark.parametrize("mode", [0, 1])
def test_rf_restatement_on_the_golden_cases(golden, oracle, case, mode):
    g = lambda k: golden[case + "/" + k]
    q, u, a, ak = g("q"), g("u"), g("alpha"), g("alpha_k")
    use_k = bool(g("use_k"))
    b = int(g("best"))
    v, w, k = g("hyp_v")[b], g("hyp_w")[b], float(g("hyp_k")[b])
    mask = g("best_mask").astype(bool)
    rho, _ = oracle.estimate_inverse_depths(q, u, v, w, k, a, ak, mode=1)
    inl = np.stack([q[mask, 0], q[mask, 1], 1.0 / rho[mask]], axis=1)
    idx = np.nonzero(mask)[0]
    r1 = oracle.refine(u, inl, a[mask], ak[mask], v, w, k, use_k, flow_index_mode=mode, inlier_idx=idx, trace_rows=50)
    r2 = oracle.refine(u, inl, a[mask], ak[mask], v, w, k, use_k, flow_index_mode=mode, inlier_idx=idx, trace_rows=50, mode=2)
    if r2["guard"]:  # (a golden case that sits inside a band: the product runs it iterate by iterate; nothing to compare)
        pytest.skip("guard %d" % r2["guard"])
    _same(r1, r2)
    n = r1["summary"]["num_iterations"]
    assert np.array_equal(r1["trace"][:n, 7], r2["trace"][:n, 7])  # every accept / reject / invalid / converge outcome


@pytest.mark.parametrize("const_acc", [False, True])
@pytest.mark.parametrize("tol", [0.002, 0.05])
def test_rf_restatement_after_a_ransac(oracle, rsdsfm, const_acc, tol):
    d = rsdsfm.synth.make_config(3, rows=90, cols=160)
    q, u, a, ak = d["q"], d["u"], d["alpha"], d["alpha_k"]
    samples = oracle.sample_indices(len(q), 12, 77)
    r = oracle.ransac(q, u, a, ak, const_acc, 12, tol, samples=samples, depth_mode=1)
    args = (u, r["inliers"], r["alpha"], r["alpha_k"], r["v"], r["w"], r["k"], const_acc)
    r1 = oracle.refine(*args, flow_index_mode=1, inlier_idx=r["inlier_idx"])
    r2 = oracle.refine(*args, flow_index_mode=1, inlier_idx=r["inlier_idx"], mode=2)
    if r2["guard"] == 8:  # k puts beta ~ 0 on a band of scanlines: more than 64 clamped inliers -> the product falls back; a designed exit
        return
    _same(r1, r2)


def test_rf_listed_inliers_focus_of_expansion_inside_the_image(oracle, rsdsfm):
    """forward motion: the focus of expansion lies in the image, the inliers around it have |J_rho| s < 1e-3 (the LM diagonal's clamp is active):
    they go through the exact list, and the result still equals the reference arithmetic's"""
    q0 = rsdsfm.synth.make_config(3, rows=96, cols=128)["q"]
    target = q0[np.argmin(np.hypot(q0[:, 0] - 0.08, q0[:, 1] + 0.06))] + 2e-4  # the focus of expansion 2e-4 beside a pixel centre (the pitch is ~1e-2)
    d = rsdsfm.synth.make_config(3, rows=96, cols=128, v=0.05 * np.array([target[0], target[1], 1.0]), w=np.array([0.001, -0.002, 0.004]))
    q, u, a, ak, t = d["q"], d["u"], d["alpha"], d["alpha_k"], d["truth"]
    v0 = t["v"] / np.linalg.norm(t["v"])
    foe = v0[:2] / v0[2]
    assert (np.hypot(q[:, 0] - foe[0], q[:, 1] - foe[1]) < 8e-4).sum() >= 1  # some pixel is within the clamp radius
    rho, _ = oracle.estimate_inverse_depths(q, u, v0, t["w"], 0.0, a, ak, mode=1)
    keep = np.abs(rho) > 1e-6
    inl = np.stack([q[keep, 0], q[keep, 1], 1.0 / rho[keep]], axis=1)
    idx = np.nonzero(keep)[0]
    # start a little off the truth so that the solve takes steps
    v1 = v0 + np.array([2e-5, -1e-5, 0.0])  # (the focus of expansion stays within the clamp radius of that pixel)
    v1 /= np.linalg.norm(v1)
    r1 = oracle.refine(u, inl, a[keep], ak[keep], v1, t["w"] * 1.02, 0.0, False, flow_index_mode=1, inlier_idx=idx)
    r2 = oracle.refine(u, inl, a[keep], ak[keep], v1, t["w"] * 1.02, 0.0, False, flow_index_mode=1, inlier_idx=idx, mode=2)
    assert r1["summary"]["num_iterations"] >= 2 and r2["listed_max"] >= 1
    _same(r1, r2)


def test_rf_edge_cases(oracle, rsdsfm):
    d = rsdsfm.synth.make_config(1, rows=48, cols=64)
    q, u, a, ak, t = d["q"], d["u"], d["alpha"], d["alpha_k"], d["truth"]
    nv = np.linalg.norm(t["v"])
    v = t["v"] / nv
    z = t["Z"].T.reshape(-1) / nv
    inl = np.stack([q[:, 0], q[:, 1], z], axis=1)
    # the fixed point: gradient tolerance at iteration zero, like the reference arithmetic
    r1 = oracle.refine(u, inl, a, ak, v, t["w"], 0.0, False)
    r2 = oracle.refine(u, inl, a, ak, v, t["w"], 0.0, False, mode=2)
    assert r2["guard"] == 0 and r2["summary"]["termination"] == r1["summary"]["termination"] == 0 and r2["summary"]["num_iterations"] == 0
    # no inliers
    r0 = oracle.refine(u, np.zeros((0, 3)), np.zeros(0), np.zeros(0), v, t["w"], 0.0, False, mode=2)
    assert r0["guard"] == 0 and r0["summary"]["termination"] == 0
    # a NaN in the data: a non-finite sum is a guard (the product then takes the reference's arithmetic through its own failure path)
    bad = inl.copy()
    bad[5, 2] = np.nan
    assert oracle.refine(u, bad, a, ak, v, t["w"], 0.0, False, mode=2)["guard"] == 1
