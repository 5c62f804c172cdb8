"""GPU parity of the DEFAULT joint refinement (csrc/refine_rf_kernels.hip: radius-factorised Schur sums) through the C ABI: every decision of the
trust-region loop equals the oracle's reference arithmetic (mode 1) and its restatement of the new arithmetic (mode 2); v, w, k, z within 1e-6
relative (north star 1e-5); the guards send a solve to the iterate-by-iterate kernels where the design says so; rsdsfm_set_lm_arithmetic(1)
selects those kernels outright."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
INTS = ("num_iterations", "num_successful_steps", "num_unsuccessful_steps", "termination")


@pytest.fixture(scope="module")
def solver(rsdsfm):
    s = rsdsfm.Solver(0)
    yield s
    s.close()


def _close(out, ref, rtol=1e-6):
    for k in INTS:
        assert out["summary"][k] == ref["summary"][k], (k, out["summary"], ref["summary"])
    assert np.isclose(out["summary"]["initial_cost"], ref["summary"]["initial_cost"], rtol=1e-11)
    assert np.isclose(out["summary"]["final_cost"], ref["summary"]["final_cost"], rtol=1e-7, atol=1e-25)
    assert np.allclose(out["v"], ref["v"], rtol=rtol, atol=1e-10) and np.allclose(out["w"], ref["w"], rtol=rtol, atol=1e-10)
    assert np.isclose(out["k"], ref["k"], rtol=rtol, atol=1e-10)
    assert np.array_equal(out["inliers"][:, :2], ref["inliers"][:, :2]) and np.allclose(out["inliers"][:, 2], ref["inliers"][:, 2], rtol=rtol)


@pytest.mark.parametrize("const_acc", [False, True])
@pytest.mark.parametrize("tol", [0.002, 0.05])
@pytest.mark.parametrize("size", [(90, 160), (360, 640)])
def test_default_refinement_equals_both_oracle_arithmetics(oracle, solver, rsdsfm, const_acc, tol, size):
    d = rsdsfm.synth.make_config(3, rows=size[0], cols=size[1])
    q, u, a, ak = d["q"], d["u"], d["alpha"], d["alpha_k"]
    samples = oracle.sample_indices(len(q), 12, 77)
    r = solver.ransac(q, u, a, ak, const_acc, 12, tol, samples=samples, depth_mode=1)
    args = (u, r["inliers"], r["alpha"], r["alpha_k"], r["v"], r["w"], r["k"], const_acc)
    kw = dict(flow_index_mode=1, inlier_idx=r["inlier_idx"])
    o1, o2 = oracle.refine(*args, **kw), oracle.refine(*args, mode=2, **kw)
    before = solver.refine_restarts()
    out = solver.non_linear_refinement(*args, **kw)
    after = solver.refine_restarts()
    assert after["runs"] == before["runs"] + 1
    _close(out, o1)  # whichever path ran: the reference arithmetic's integers, its values to 1e-6
    if o2["guard"] == 0:
        assert after["restarts"] == before["restarts"], after  # no guard in the restatement, none on the GPU (sums differ in order only)
        assert after["resolves"] - before["resolves"] == o2["resolves"]  # rejected / invalid steps were solved again from the kept sums: no pass
        _close(out, o2)
    else:
        assert after["restarts"] == before["restarts"] + 1 and after["last_guard"] == o2["guard"], (after, o2["guard"])
    # the iterate-by-iterate kernels on request: the same integers, the values within the same bar
    solver.set_lm_arithmetic(1)
    try:
        exact = solver.non_linear_refinement(*args, **kw)
        assert solver.refine_restarts()["runs"] == after["runs"]  # (did not run on the radius-factorised path)
    finally:
        solver.set_lm_arithmetic(0)
    _close(exact, o1)
    _close(out, exact)


def test_listed_inliers_focus_of_expansion_inside_the_image(oracle, solver, rsdsfm):
    """forward motion: the inliers next to the focus of expansion have an active LM-diagonal clamp; the pass lists them, the stage adds their
    exact terms for the radius in question -- the result equals the reference arithmetic's, no guard"""
    q0 = rsdsfm.synth.make_config(3, rows=96, cols=128)["q"]
    target = q0[np.argmin(np.hypot(q0[:, 0] - 0.08, q0[:, 1] + 0.06))] + 2e-4
    d = rsdsfm.synth.make_config(3, rows=96, cols=128, v=0.05 * np.array([target[0], target[1], 1.0]), w=np.array([0.001, -0.002, 0.004]))
    q, u, a, ak, t = d["q"], d["u"], d["alpha"], d["alpha_k"], d["truth"]
    v0 = t["v"] / np.linalg.norm(t["v"])
    rho, _ = oracle.estimate_inverse_depths(q, u, v0, t["w"], 0.0, a, ak, mode=1)
    keep = np.abs(rho) > 1e-6
    inl = np.stack([q[keep, 0], q[keep, 1], 1.0 / rho[keep]], axis=1)
    idx = np.nonzero(keep)[0]
    v1 = v0 + np.array([2e-5, -1e-5, 0.0])
    v1 /= np.linalg.norm(v1)
    args = (u, inl, a[keep], ak[keep], v1, t["w"] * 1.02, 0.0, False)
    kw = dict(flow_index_mode=1, inlier_idx=idx)
    o1, o2 = oracle.refine(*args, **kw), oracle.refine(*args, mode=2, **kw)
    assert o2["listed_max"] >= 1 and o2["guard"] == 0 and o1["summary"]["num_iterations"] >= 2
    before = solver.refine_restarts()
    out = solver.non_linear_refinement(*args, **kw)
    assert solver.refine_restarts()["restarts"] == before["restarts"]
    _close(out, o1)
    _close(out, o2)


def test_guards_send_the_solve_to_the_iterate_by_iterate_kernels(oracle, solver, rsdsfm):
    d = rsdsfm.synth.make_config(1, rows=48, cols=64)
    q, u, a, ak, t = d["q"], d["u"], d["alpha"], d["alpha_k"], d["truth"]
    nv = np.linalg.norm(t["v"])
    v = t["v"] / nv
    inl = np.stack([q[:, 0], q[:, 1], t["Z"].T.reshape(-1) / nv], axis=1)
    bad = inl.copy()
    bad[5, 2] = np.nan  # a non-finite sum: guard 1; the reference's arithmetic then walks its own failure path (five invalid steps)
    before = solver.refine_restarts()
    out = solver.non_linear_refinement(u, bad, a, ak, v, t["w"], 0.0, False)
    after = solver.refine_restarts()
    assert after["restarts"] == before["restarts"] + 1 and after["last_guard"] == 1
    ref = oracle.refine(u, bad, a, ak, v, t["w"], 0.0, False)
    for k in INTS:
        assert out["summary"][k] == ref["summary"][k], (k, out["summary"], ref["summary"])
    # m = 0 and the fixed point stay on the path: gradient tolerance at iteration zero
    before = solver.refine_restarts()
    out = solver.non_linear_refinement(u, inl, a, ak, v, t["w"], 0.0, False)
    assert out["summary"]["termination"] == 0 and out["summary"]["num_iterations"] == 0
    out0 = solver.non_linear_refinement(u, np.zeros((0, 3)), np.zeros(0), np.zeros(0), v, t["w"], 0.0, False)
    assert out0["summary"]["termination"] == 0 and solver.refine_restarts()["restarts"] == before["restarts"]


def test_results_do_not_depend_on_the_contexts_history_or_on_tracing(solver, rsdsfm, oracle):
    d = rsdsfm.synth.make_config(3, rows=120, cols=200)
    q, u, a, ak = d["q"], d["u"], d["alpha"], d["alpha_k"]
    samples = oracle.sample_indices(len(q), 10, 5)
    r = solver.ransac(q, u, a, ak, False, 10, 0.05, samples=samples, depth_mode=1)
    args = (u, r["inliers"], r["alpha"], r["alpha_k"], r["v"], r["w"], r["k"], False)
    outs = []
    for i in range(3):
        if i == 2:
            solver.set_refine_trace(32)
        outs.append(solver.non_linear_refinement(*args, flow_index_mode=1, inlier_idx=r["inlier_idx"]))
    solver.set_refine_trace(0)
    for o in outs[1:]:
        assert o["summary"] == outs[0]["summary"] and np.array_equal(o["v"], outs[0]["v"]) and np.array_equal(o["inliers"], outs[0]["inliers"])


def test_refinement_from_the_resident_ransac_outputs_equals_the_uploaded_one(oracle, rsdsfm):
    """rsdsfm_refine_from_ransac (the C++ mirror's nonLinearRefinement on an unmodified RansacValues): the RANSAC's outputs and the flow it was
    given are still on the device, nothing is uploaded again; the result is the uploading call's bit for bit, and everything that makes the
    resident copy unusable (another host-pointer call in between, a stale tag, arrays that no longer hold what was downloaded) falls back"""
    d = rsdsfm.synth.make_config(3, rows=180, cols=320)
    q, u, a, ak = d["q"], d["u"], d["alpha"], d["alpha_k"]
    samples = oracle.sample_indices(len(q), 10, 3)
    with rsdsfm.Solver(0) as s:
        r = s.ransac(q, u, a, ak, False, 10, 0.004, samples=samples, depth_mode=1)
        assert r["tag"] != 0 and 1000 < r["num_inliers"] < len(q)
        args = (u, r["inliers"], r["alpha"], r["alpha_k"], r["v"], r["w"], r["k"], False)
        for mode in (0, 1):  # the reference's rank-indexed flow (quirk Q2) and the gathered one
            kw = dict(flow_index_mode=mode, inlier_idx=r["inlier_idx"])
            h0 = s.refine_cache_hits()
            fast = s.non_linear_refinement(*args, tag=r["tag"], **kw)
            assert s.refine_cache_hits() == h0 + 1
            slow = s.non_linear_refinement(*args, **kw)  # (tag 0: uploads everything; takes the staging buffer -> the resident copy is gone)
            assert s.refine_cache_hits() == h0 + 1
            assert fast["summary"] == slow["summary"] and np.array_equal(fast["v"], slow["v"]) and np.array_equal(fast["w"], slow["w"])
            assert np.array_equal(fast["inliers"], slow["inliers"])
            again = s.non_linear_refinement(*args, tag=r["tag"], **kw)  # the tag is stale now: falls back, same result
            assert s.refine_cache_hits() == h0 + 1 and np.array_equal(again["inliers"], slow["inliers"])
            r = s.ransac(q, u, a, ak, False, 10, 0.004, samples=samples, depth_mode=1)  # a new resident copy for the next round
        # another flow array with the same contents: inliers from the resident copy, the flow uploaded
        h0 = s.refine_cache_hits()
        other = s.non_linear_refinement(u.copy(), r["inliers"], r["alpha"], r["alpha_k"], r["v"], r["w"], r["k"], False, flow_index_mode=1, inlier_idx=r["inlier_idx"], tag=r["tag"])
        assert s.refine_cache_hits() == h0 + 1 and np.array_equal(other["inliers"], slow["inliers"])
        # arrays that no longer hold what was downloaded (rebuilt: scaled depths): the probes notice, everything is uploaded, the answer is theirs
        inl2 = r["inliers"].copy()
        inl2[:, 2] *= 1.25
        h0 = s.refine_cache_hits()
        mod = s.non_linear_refinement(u, inl2, r["alpha"], r["alpha_k"], r["v"], r["w"], r["k"], False, flow_index_mode=1, inlier_idx=r["inlier_idx"], tag=r["tag"])
        assert s.refine_cache_hits() == h0
        ref = oracle.refine(u, inl2, r["alpha"], r["alpha_k"], r["v"], r["w"], r["k"], False, flow_index_mode=1, inlier_idx=r["inlier_idx"])
        _close(mod, ref)


def test_host_transfers_through_the_pinned_ring_are_exact(rsdsfm, oracle):
    """inputs / outputs of several pieces of the ring (2 + 4 + 8 MiB ... up, 8 MiB ... + a tail of 2 MiB down; uploads alternate between two
    streams) and not a multiple of any of them: what comes back is what the device holds"""
    d = rsdsfm.synth.make_config(3, rows=720, cols=1279)
    q, u, a, ak, t = d["q"], d["u"], d["alpha"], d["alpha_k"], d["truth"]
    v = t["v"] / np.linalg.norm(t["v"])
    with rsdsfm.Solver(0) as s:
        rho, sm = s.estimate_inverse_depths(q, u, v, t["w"], 0.0, a, ak, mode=1)
        samples = oracle.sample_indices(len(q), 6, 9)
        r = s.ransac(q, u, a, ak, False, 6, 0.05, samples=samples, depth_mode=1)
        ro = oracle.ransac(q, u, a, ak, False, 6, 0.05, samples=samples, depth_mode=1)
    assert r["num_inliers"] == ro["num_inliers"] and np.array_equal(r["inlier_idx"], ro["inlier_idx"]) and np.array_equal(r["mask"], ro["mask"])
    assert np.array_equal(r["inliers"][:, :2], q[r["inlier_idx"]])  # x, y are copies of q: every chunk landed where it belongs
    assert np.array_equal(r["alpha"], a[r["inlier_idx"]]) and np.array_equal(r["alpha_k"], ak[r["inlier_idx"]])
    assert np.allclose(r["inv_depth"], ro["inv_depth"], rtol=1e-9, atol=1e-13)


def test_host_transfers_of_two_contexts_on_two_threads_share_the_ring(rsdsfm, oracle):
    """the pinned ring belongs to the process and the device: two contexts driven from two threads upload through it at the same time (its
    lock serialises them) and each gets its own answer"""
    import threading

    ds = [rsdsfm.synth.make_config(3, rows=300, cols=1400 + 37 * j, seed=77 + j) for j in range(2)]
    samples = [oracle.sample_indices(len(d["q"]), 6, 5 + j) for j, d in enumerate(ds)]

    def solve(j, reps, out):
        d = ds[j]
        with rsdsfm.Solver(0) as s:
            for _ in range(reps):
                r = s.ransac(d["q"], d["u"], d["alpha"], d["alpha_k"], False, 6, 0.05, samples=samples[j], depth_mode=1)
                out.append((r["num_inliers"], r["best_trial"], r["inlier_idx"].tobytes(), r["inliers"].tobytes(), r["inv_depth"].tobytes()))

    serial = [[], []]
    for j in range(2):
        solve(j, 1, serial[j])
    both = [[], []]
    errs = []

    def work(j):
        try:
            solve(j, 4, both[j])
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=work, args=(j,)) for j in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(120)
    assert not errs, errs
    for j in range(2):
        assert len(both[j]) == 4 and all(x == serial[j][0] for x in both[j])
