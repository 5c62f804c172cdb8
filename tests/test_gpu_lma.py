"""GPU: the RANSAC's depth solves on the ANALYTIC LM TRAJECTORY (csrc/lma_common.hpp, csrc/ransac_lma_kernels.hip; the library's default
since round 5) against the iterate-by-iterate kernels (rsdsfm_set_lm_arithmetic(1): the reference's arithmetic, operation for operation)
and against the CPU oracle.  Integer outputs -- per-trial inlier counts, accepted LM steps, winner, mask, index list -- bit-exact on both
sides; the winner's depths come from the same exact replay and are bit-identical between the two arithmetics; error sums 1e-9."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def solver(rsdsfm):
    s = rsdsfm.Solver(0)
    # this module compares the FORMS OF THE DEPTH SOLVES bit for bit through whole frame solves: the joint refinement behind them is pinned to ONE
    # arithmetic (the iterate-by-iterate kernels); the refinement's own default arithmetic has its tests in test_gpu_refine_rf.py
    s.set_refine_arithmetic(1)
    yield s
    s.close()


def _same(r, ro, exact_depths=True):
    assert np.array_equal(r["trial_count"], ro["trial_count"])
    assert np.array_equal(r["trial_steps"], ro["trial_steps"])
    assert r["best_trial"] == ro["best_trial"] and r["num_inliers"] == ro["num_inliers"]
    assert np.array_equal(r["mask"], ro["mask"]) and np.array_equal(r["inlier_idx"], ro["inlier_idx"])
    assert np.allclose(r["trial_err"], ro["trial_err"], rtol=1e-9, atol=1e-12)
    if exact_depths:
        assert np.array_equal(r["inv_depth"], ro["inv_depth"]) and np.array_equal(r["inliers"], ro["inliers"])
    else:
        assert np.allclose(r["inv_depth"], ro["inv_depth"], rtol=1e-9, atol=1e-13)


def _both(solver, *args, **kw):
    solver.set_lm_arithmetic(0)
    n0 = solver.lma_restarts()[0]
    ra = solver.ransac(*args, **kw)
    restarts = solver.lma_restarts()[0] - n0
    solver.set_lm_arithmetic(1)
    rx = solver.ransac(*args, **kw)
    solver.set_lm_arithmetic(0)
    return ra, rx, restarts


@pytest.mark.parametrize("T", [1, 5, 50, 130])
@pytest.mark.parametrize("tol", [0.05, 0.002])
def test_analytic_equals_iterate_by_iterate_and_the_oracle(oracle, solver, rsdsfm, T, tol):
    d = rsdsfm.synth.make_config(3, rows=135, cols=240)
    q, u, a, ak = d["q"], d["u"], d["alpha"], d["alpha_k"]
    samples = oracle.sample_indices(len(q), T, 1234)
    ra, rx, restarts = _both(solver, q, u, a, ak, False, T, tol, samples=samples)
    assert restarts == 0
    _same(ra, rx)
    ro = oracle.ransac(q, u, a, ak, False, T, tol, samples, depth_mode=1)
    _same(ra, ro, exact_depths=False)
    ro2 = oracle.ransac(q, u, a, ak, False, T, tol, samples, depth_mode=2)  # the analytic arithmetic restated on the CPU
    _same(ra, ro2, exact_depths=False)
    assert np.allclose(ra["trial_err"], ro2["trial_err"], rtol=1e-12, atol=1e-300)


def test_clamped_pixels_around_the_focus_of_expansion(oracle, solver, rsdsfm):
    """forward motion: the pixels around the focus of expansion have a Jacobian below the LM diagonal's clamp (guard a): listed, walked on
    the exact recurrence by the decide stage -- no restart, every integer as iterate by iterate"""
    d = rsdsfm.synth.make_config(3, rows=180, cols=320, v=np.array([0.002, 0.001, 0.03]))
    q, u, a, ak = d["q"], d["u"], d["alpha"], d["alpha_k"]
    T = 24
    samples = oracle.sample_indices(len(q), T, 5)
    ra, rx, restarts = _both(solver, q, u, a, ak, False, T, 0.01, samples=samples)
    assert restarts == 0
    _same(ra, rx)
    ro2 = oracle.ransac(q, u, a, ak, False, T, 0.01, samples, depth_mode=2)
    assert oracle.lma_last_stats()["listed_clamped"] > 0
    _same(ra, ro2, exact_depths=False)


def test_noise_free_tie_starts_over_and_holds(oracle, solver, rsdsfm):
    """noise-free data: a tie in count and error sum that only the reference's own rounding breaks (guard d) -- the run starts over
    iterate by iterate (counted), the context stays there while the data keeps tying, and the results are the iterate-by-iterate ones"""
    d = rsdsfm.synth.make_config(1, rows=96, cols=128)
    q, u, a, ak = d["q"], d["u"], d["alpha"], d["alpha_k"]
    T = 10
    samples = oracle.sample_indices(len(q), T, 3)
    n0 = solver.lma_restarts()[0]
    r1 = solver.ransac(q, u, a, ak, False, T, 0.05, samples=samples)
    n1, guards = solver.lma_restarts()
    assert n1 == n0 + 1 and guards & (1 << 7)
    for _ in range(20):  # (> the hold of 16: a tie seen by the iterate-by-iterate run renews it)
        r2 = solver.ransac(q, u, a, ak, False, T, 0.05, samples=samples)
    assert solver.lma_restarts()[0] == n1
    solver.set_lm_arithmetic(1)
    rx = solver.ransac(q, u, a, ak, False, T, 0.05, samples=samples)
    solver.set_lm_arithmetic(0)
    for r in (r1, r2):
        _same(r, rx)
        assert np.array_equal(r["trial_err"], rx["trial_err"])
    # noisy data ends the hold after 16 runs
    dn = rsdsfm.synth.make_config(3, rows=96, cols=128)
    sn = oracle.sample_indices(len(dn["q"]), T, 3)
    for _ in range(17):
        solver.ransac(dn["q"], dn["u"], dn["alpha"], dn["alpha_k"], False, T, 0.05, samples=sn)
    ra, rx, restarts = _both(solver, dn["q"], dn["u"], dn["alpha"], dn["alpha_k"], False, T, 0.05, samples=sn)
    assert restarts == 0
    _same(ra, rx)


def test_nan_hypotheses_and_acceleration_mode(golden, oracle, solver):
    for case in ("noisy_k0", "deepflow_k0", "noisy_k04"):
        g = lambda k: golden[case + "/" + k]
        q, u, a, ak, samples = g("q"), g("u"), g("alpha"), g("alpha_k"), g("samples")
        use_k = bool(g("use_k"))
        ra, rx, restarts = _both(solver, q, u, a, ak, use_k, len(samples), 0.05, samples=samples)
        assert restarts == 0
        _same(ra, rx)
        assert np.array_equal(ra["trial_count"], g("count_lm"))
    # a non-finite flow at a sampled point: that trial's pose is NaN (it ends at iteration zero, no inlier), and the NaN pixel makes the sums of
    # every other trial NaN (five invalid steps, termination FAILURE, scored at rho = 1 by the scoring pass) -- no restart, same integers
    q, u, a, ak = golden["noisy_k0/q"], golden["noisy_k0/u"].copy(), golden["noisy_k0/alpha"], golden["noisy_k0/alpha_k"]
    samples = oracle.sample_indices(len(q), 6, 9)
    u[samples[2, 0], 0] = np.nan
    ra, rx, restarts = _both(solver, q, u, a, ak, False, 6, 0.05, samples=samples)
    assert restarts == 0
    _same(ra, rx)
    ro = oracle.ransac(q, u, a, ak, False, 6, 0.05, samples, depth_mode=1)
    assert np.array_equal(ra["trial_count"], ro["trial_count"]) and np.array_equal(ra["trial_steps"], ro["trial_steps"]) and ra["best_trial"] == ro["best_trial"]
    assert not ra["trial_steps"].any()


def test_every_pixel_clamped_overflows_the_list_and_starts_over(oracle, solver, rsdsfm):
    """a translation of 1e-5 x the unit vector the solver returns cannot happen -- but a frame whose points all sit at the focus of
    expansion can be built: the Jacobian of every point is below the clamp, the hypothesis' list overflows (guard c) and the run starts
    over iterate by iterate with identical results"""
    rng = np.random.default_rng(1)
    n = 4000
    q = rng.normal(size=(n, 2)) * 1e-5  # all points within 1e-5 of the principal point
    u = rng.normal(size=(n, 2)) * 1e-3
    a, ak = np.ones(n), np.full(n, 0.5)
    T = 4
    samples = oracle.sample_indices(n, T, 1)
    n0 = solver.lma_restarts()[0]
    ra, rx, restarts = _both(solver, q, u, a, ak, False, T, 0.05, samples=samples)
    _same(ra, rx)
    ro = oracle.ransac(q, u, a, ak, False, T, 0.05, samples, depth_mode=1)
    _same(ra, ro, exact_depths=False)


def test_frame_solve_is_the_same_in_both_arithmetics(solver, rsdsfm):
    """the one-call frame solve (flatten -> RANSAC -> refinement -> depth map): analytic pass and iterate-by-iterate kernels give the same
    inliers, the same refinement, the same depth map -- bit for bit (everything behind the RANSAC starts from the same exact replay)"""
    import torch

    dev = torch.device("cuda", 0)
    d = rsdsfm.synth.make_config(3, rows=270, cols=480)
    rows, cols, K, gamma = d["rows"], d["cols"], d["K"], d["gamma"]
    img = torch.from_numpy(d["flow_img"]).to(dev)
    outs = []
    for tol in (0.05, 0.002):
        for mode in (0, 1, 0):
            solver.set_lm_arithmetic(mode)
            dm = torch.zeros((cols, rows), dtype=torch.float64, device=dev)
            r = solver.solve_frame_dev(img.data_ptr(), rows, cols, K, gamma, dm.data_ptr(), trials=50, tol=tol, seed=7)
            solver.synchronize()
            outs.append((r, dm.cpu().numpy()))
        solver.set_lm_arithmetic(0)
        (a, da), (b, db), (c, dc) = outs[-3:]
        for x, dx in ((b, db), (c, dc)):
            assert a["num_inliers"] == x["num_inliers"] and a["best_trial"] == x["best_trial"]
            assert a["refine_summary"] == x["refine_summary"]
            for key in ("v", "w", "ransac_v", "ransac_w"):
                assert np.array_equal(a[key], x[key]), key
            assert np.array_equal(da, dx, equal_nan=True)
    assert solver.lma_restarts()[0] == 0


def _frame(solver, torch, dev, img, rows, cols, K, gamma, **kw):
    dm = torch.zeros((cols, rows), dtype=torch.float64, device=dev)
    r = solver.solve_frame_dev(img.data_ptr(), rows, cols, K, gamma, dm.data_ptr(), **kw)
    solver.synchronize()
    return r, dm.cpu().numpy()


def _same_frame(a, da, x, dx):
    assert a["num_inliers"] == x["num_inliers"] and a["best_trial"] == x["best_trial"]
    assert a["refine_summary"] == x["refine_summary"]
    for key in ("v", "w", "ransac_v", "ransac_w"):
        assert np.array_equal(a[key], x[key], equal_nan=True), key
    assert np.array_equal(da, dx, equal_nan=True)


@pytest.mark.parametrize("cfg", [3, 5, 2])  # (2: noise-free flow -- every good hypothesis explains every pixel)
def test_count_only_pass_and_lazy_error_sums(solver, rsdsfm, cfg):
    """The frame solve's count-only form of the analytic pass (rsdsfm_set_lm_arithmetic(2) forces it): no error sums in the pixel pass; where
    trials share the best inlier count (BASELINE's tolerance 0.05 admits every pixel under any good hypothesis; noise-free flow) exactly those
    are scored by the iterate-by-iterate scoring pass and minimal.cc:278-285's tie rule runs on the reference arithmetic's sums.  Winner,
    inliers, refinement and depth map: bit for bit what the iterate-by-iterate library returns."""
    import torch

    dev = torch.device("cuda", 0)
    d = rsdsfm.synth.make_config(cfg, rows=270, cols=480)
    rows, cols, K, gamma = d["rows"], d["cols"], d["K"], d["gamma"]
    img = torch.from_numpy(d["flow_img"]).to(dev)
    lazy_total = 0
    for tol in (0.5, 0.05, 0.002):  # (0.5: every pixel of this small frame an inlier under any good hypothesis, as 0.05 is at 1280x720)
        for T in (50, 5):
            for seed in (7, 8, 9):
                solver.set_lm_arithmetic(1)
                x, dx = _frame(solver, torch, dev, img, rows, cols, K, gamma, trials=T, tol=tol, seed=seed)
                solver.set_lm_arithmetic(2)
                runs0, lazy0 = solver.lma_count_only()
                a, da = _frame(solver, torch, dev, img, rows, cols, K, gamma, trials=T, tol=tol, seed=seed)
                b, db = _frame(solver, torch, dev, img, rows, cols, K, gamma, trials=T, tol=tol, seed=seed)  # (warm: the speculated stages follow the first)
                runs1, lazy1 = solver.lma_count_only()
                assert runs1 - runs0 == 2
                lazy_total += lazy1 - lazy0
                _same_frame(a, da, x, dx)
                _same_frame(b, db, x, dx)
    solver.set_lm_arithmetic(0)
    assert solver.lma_restarts()[0] == 0
    if cfg == 3:
        assert lazy_total > 0  # (tolerance 0.5: good hypotheses share the count N)


def test_count_only_follows_the_data(solver, rsdsfm):
    """mode 0: the pass turns count-only behind two solves whose best count was unique (selective tolerance) and goes back to fused error sums
    when trials share the best count again (permissive tolerance); results never depend on it"""
    import torch

    dev = torch.device("cuda", 0)
    d = rsdsfm.synth.make_config(5, rows=270, cols=480)
    rows, cols, K, gamma = d["rows"], d["cols"], d["K"], d["gamma"]
    img = torch.from_numpy(d["flow_img"]).to(dev)
    solver.set_lm_arithmetic(0)
    seq = [(0.002, s) for s in range(1, 6)] + [(0.5, s) for s in range(1, 5)] + [(0.002, s) for s in range(6, 10)]
    got = []
    for tol, seed in seq:
        r0 = solver.lma_count_only()
        a, da = _frame(solver, torch, dev, img, rows, cols, K, gamma, trials=50, tol=tol, seed=seed)
        r1 = solver.lma_count_only()
        got.append((r1[0] - r0[0], r1[1] - r0[1]))
        ref = rsdsfm.Solver(0)
        ref.set_lm_arithmetic(1)
        ref.set_refine_arithmetic(1)
        x, dx = _frame(ref, torch, dev, img, rows, cols, K, gamma, trials=50, tol=tol, seed=seed)
        ref.close()
        _same_frame(a, da, x, dx)
    # two unique solves, then count-only; the first permissive solve meets the tie (lazy), the following ones fuse the sums again
    assert [g[0] for g in got[:5]] == [0, 0, 1, 1, 1]
    assert got[5] == (1, 1) and [g[0] for g in got[6:9]] == [0, 0, 0]
    assert [g[0] for g in got[9:]] == [0, 0, 1, 1]


def test_count_only_edge_cases(solver, rsdsfm):
    """count-only form, forced: more trials than one hypothesis batch (the form does not apply: fused error sums, same results), a frame with a
    NaN flow vector (NaN hypotheses count no inlier; a NaN pixel is no inlier of any hypothesis), acceleration mode (hypotheses handed over to
    the iterate-by-iterate rounds have exact sums already), a sequence through the lanes of one context"""
    import torch

    dev = torch.device("cuda", 0)
    d = rsdsfm.synth.make_config(5, rows=135, cols=240)
    rows, cols, K, gamma = d["rows"], d["cols"], d["K"], d["gamma"]
    clean = np.array(d["flow_img"])
    bad = clean.copy()
    bad[40, 100] = np.nan
    for flow, kw, expect_runs in ((clean, dict(trials=130, tol=0.5), 0), (bad, dict(trials=50, tol=0.5), 1), (bad, dict(trials=20, tol=0.003), 1),
                                  (clean, dict(trials=50, tol=0.5, use_acceleration_mode=True), 1), (clean, dict(trials=8, tol=0.01, use_refinement=False), 1)):
        img = torch.from_numpy(flow).to(dev)
        solver.set_lm_arithmetic(1)
        x, dx = _frame(solver, torch, dev, img, rows, cols, K, gamma, seed=5, **kw)
        solver.set_lm_arithmetic(2)
        r0 = solver.lma_count_only()[0]
        a, da = _frame(solver, torch, dev, img, rows, cols, K, gamma, seed=5, **kw)
        assert solver.lma_count_only()[0] - r0 == expect_runs, kw
        _same_frame(a, da, x, dx)
    # a sequence of pairs through the context's lanes (rsdsfm_solve_frames_dev): every lane inherits the form
    imgs = [torch.from_numpy(rsdsfm.synth.make_config(5, rows=rows, cols=cols, seed=100 + i)["flow_img"]).to(dev) for i in range(6)]
    dms = [torch.zeros((cols, rows), dtype=torch.float64, device=dev) for _ in imgs]
    jobs = [dict(d_flow_img=im.data_ptr(), rows=rows, cols=cols, K=K, gamma=gamma, d_depth_map=dm.data_ptr(), d_R=None, d_t=None) for im, dm in zip(imgs, dms)]
    outs = {}
    for mode in (1, 2):
        solver.set_lm_arithmetic(mode)
        for tol in (0.5, 0.003):
            res = solver.solve_frames_dev(jobs, [3 + i for i in range(len(jobs))], trials=50, tol=tol)
            solver.synchronize()
            outs[(mode, tol)] = ([(r["num_inliers"], r["best_trial"], tuple(r["v"]), tuple(r["w"]), str(r["refine_summary"])) for r in res], [dm.cpu().numpy().copy() for dm in dms])
    for tol in (0.5, 0.003):
        assert outs[(1, tol)][0] == outs[(2, tol)][0]
        for a_, b_ in zip(outs[(1, tol)][1], outs[(2, tol)][1]):
            assert np.array_equal(a_, b_, equal_nan=True)
    solver.set_lm_arithmetic(0)


# ---------------------------------------------------------------------------------------------------
# the dense depth solve on the analytic trajectory (depth_lma_kernels.hip)
# ---------------------------------------------------------------------------------------------------
INT_KEYS = ("num_iterations", "num_successful_steps", "num_unsuccessful_steps", "termination")


def _depth_both(solver, *args):
    solver.set_lm_arithmetic(0)
    ra, sa = solver.estimate_inverse_depths(*args, mode=1)
    solver.set_lm_arithmetic(1)
    rx, sx = solver.estimate_inverse_depths(*args, mode=1)
    solver.set_lm_arithmetic(0)
    return ra, sa, rx, sx


@pytest.mark.parametrize("cfg,rows,cols,kw", [(3, 135, 240, {}), (1, 96, 128, {}), (5, 360, 640, {}), (3, 180, 320, dict(v=np.array([0.002, 0.001, 0.03]))),
                                              (2, 720, 1280, {})])
def test_dense_depth_analytic_equals_oracle_mode2_and_iterate_by_iterate(oracle, rsdsfm, cfg, rows, cols, kw):
    """rho of the analytic fast path = the oracle's mode 2 BIT FOR BIT (the same closed form per pixel, the same planned phi; clamped pixels
    -- the forward-motion case has some -- from the same exact recurrence), the LM decisions those of the iterate-by-iterate kernels and of
    the oracle's mode 1, rho within 1e-9 of theirs; repeated solves (predictor warm) give the same bits as the first (predictor cold)"""
    d = rsdsfm.synth.make_config(cfg, rows=rows, cols=cols, **kw)
    q, u, a, ak, t = d["q"], d["u"], d["alpha"], d["alpha_k"], d["truth"]
    v = t["v"] / np.linalg.norm(t["v"]) + np.array([0.01, -0.02, 0.005])  # (not the true pose: a residual is left)
    v /= np.linalg.norm(v)
    w = t["w"] * 1.1
    with rsdsfm.Solver(0) as s:
        n0 = s.lma_restarts()[0]
        ra, sa, rx, sx = _depth_both(s, q, u, v, w, 0.0, a, ak)
        ra2, sa2 = s.estimate_inverse_depths(q, u, v, w, 0.0, a, ak, mode=1)
        assert s.lma_restarts()[0] == n0
    r2, s2 = oracle.estimate_inverse_depths(q, u, v, w, 0.0, a, ak, mode=2)
    r1, s1 = oracle.estimate_inverse_depths(q, u, v, w, 0.0, a, ak, mode=1)
    assert np.array_equal(ra, r2) and np.array_equal(ra, ra2)
    for key in INT_KEYS:
        assert sa[key] == sx[key] == s1[key] == s2[key] == sa2[key], (key, sa, sx, s1)
    assert np.allclose(ra, rx, rtol=1e-9, atol=1e-13) and np.allclose(ra, r1, rtol=1e-9, atol=1e-13)
    assert abs(sa["final_cost"] - s1["final_cost"]) <= 1e-10 * abs(s1["final_cost"]) + 1e-20 * s1["initial_cost"]
    assert abs(sa["initial_cost"] - s1["initial_cost"]) <= 1e-11 * s1["initial_cost"] and sa["final_radius"] == s1["final_radius"]


def test_dense_depth_analytic_batched_entry_point_and_a_guard(oracle, rsdsfm):
    """the batched entry point (two problems per launch: DeepFlow-like and noise-free, i.e. other iterates are final) on the analytic path; and a
    problem whose points all sit at the focus of expansion: the list of clamped pixels overflows (guard c), the solve is left unfinished and
    rsdsfm_depth_finish_dev runs it again iterate by iterate -- same results as rsdsfm_set_lm_arithmetic(1), counted"""
    import torch

    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    da, db = rsdsfm.synth.make_config(5, rows=240, cols=320), rsdsfm.synth.make_config(2, rows=240, cols=320)
    t = da["truth"]
    v = t["v"] / np.linalg.norm(t["v"])
    with torch.cuda.stream(stream):
        solvers = [rsdsfm.Solver(0, stream=stream.cuda_stream) for _ in range(2)]
        data = [(da["q"], da["u"], da["alpha"], da["alpha_k"]), (db["q"], db["u"], db["alpha"], db["alpha_k"])]
        dev_t = [[torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in c] for c in data]
        rhos = [torch.zeros(len(c[2]), dtype=torch.float64, device=dev) for c in data]
        probs = [dict(d_q=t_[0].data_ptr(), d_u=t_[1].data_ptr(), d_alpha=t_[2].data_ptr(), d_alpha_k=t_[3].data_ptr(), d_rho=r.data_ptr(), n=len(c[2]), v=v, w=t["w"], k=0.0)
                 for t_, r, c in zip(dev_t, rhos, data)]
        call = rsdsfm.prepared_depth_batch(solvers, probs)
        for rep in range(3):
            call()
            for i, p in enumerate(probs):
                sm, _ = solvers[i].depth_finish_dev(p["d_q"], p["d_u"], p["n"], v, t["w"], 0.0, p["d_alpha"], p["d_alpha_k"], p["d_rho"])
                r2, s2 = oracle.estimate_inverse_depths(*data[i][:2], v, t["w"], 0.0, *data[i][2:], mode=2)
                for key in INT_KEYS:
                    assert sm[key] == s2[key], (rep, i, key, sm, s2)
                assert np.array_equal(rhos[i].cpu().numpy(), r2), (rep, i)
        assert all(s.lma_restarts()[0] == 0 for s in solvers)
        for s in solvers:
            s.close()
    # a guard: every pixel clamped
    rng = np.random.default_rng(1)
    n = 6000
    q = rng.normal(size=(n, 2)) * 1e-5
    u = rng.normal(size=(n, 2)) * 1e-3
    a, ak = np.ones(n), np.full(n, 0.5)
    vv, ww = np.array([0.0, 0.0, 1.0]), np.array([0.001, -0.002, 0.0005])
    with rsdsfm.Solver(0) as s:
        ra, sa, rx, sx = _depth_both(s, q, u, vv, ww, 0.0, a, ak)
        assert s.lma_restarts()[0] == 1
        assert np.array_equal(ra, rx) and all(sa[k] == sx[k] for k in INT_KEYS)
        rb, sb = s.estimate_inverse_depths(q, u, vv, ww, 0.0, a, ak, mode=1)  # no hold for the dense solve: a result is a function of its inputs alone
        assert s.lma_restarts()[0] == 2 and np.array_equal(rb, rx)
    r1, s1 = oracle.estimate_inverse_depths(q, u, vv, ww, 0.0, a, ak, mode=1)
    assert np.allclose(ra, r1, rtol=1e-9, atol=1e-13) and all(sa[k] == s1[k] for k in INT_KEYS)
