"""GPU parity: the HIP dense depth solve (through the C ABI) against the CPU oracle and the golden fixtures.

Bars: LM decisions / counters are integers -> exact; rho within 1e-9 relative of the oracle on the same
inputs (north-star bar: 1e-5); in practice the kernels reproduce the oracle bit for bit because both are
compiled without FMA contraction and use the reference's operation order."""
import numpy as np
import pytest

from conftest import GOLDEN_CASES

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[0, 1, 2, 3], ids=["default_decide_apply", "ldsdma", "decide_in_launch0_tail", "separate_decide"])
def solver(rsdsfm, request):
    """every test of this module runs on the data-movement variants of the LM kernel and on the variant with the decision fused into launch 0"""
    s = rsdsfm.Solver(0)
    s.set_depth_variant(request.param)
    yield s
    s.close()


def _check_summary(sm, ref):
    for k in ("num_iterations", "num_successful_steps", "num_unsuccessful_steps", "termination"):
        assert sm[k] == ref[k], (k, sm, ref)
    assert np.isclose(sm["initial_cost"], ref["initial_cost"], rtol=1e-12)
    assert np.isclose(sm["final_cost"], ref["final_cost"], rtol=1e-9, atol=1e-25)
    assert np.isclose(sm["final_radius"], ref["final_radius"], rtol=1e-15)


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_depth_vs_golden_and_oracle(golden, oracle, solver, case):
    g = lambda k: golden[case + "/" + k]
    q, u, a, ak = g("q"), g("u"), g("alpha"), g("alpha_k")
    W, V, K = g("hyp_w"), g("hyp_v"), g("hyp_k")
    for t in range(len(W)):
        for mode in (0, 1):
            rho, sm = solver.estimate_inverse_depths(q, u, V[t], W[t], K[t], a, ak, mode=mode)
            rho_o, sm_o = oracle.estimate_inverse_depths(q, u, V[t], W[t], K[t], a, ak, mode=mode)
            assert np.allclose(rho, rho_o, rtol=1e-9, atol=1e-13)
            if mode == 1:
                _check_summary(sm, sm_o)
                ref = g("lm_summary")[t]
                assert sm["num_successful_steps"] == int(ref[1]) and sm["termination"] == int(ref[3])
            if t < 3:
                assert np.allclose(rho, g("rho_lm" if mode else "rho_cf")[t], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("n", [0, 1, 2, 3, 9, 255, 256, 257, 1000, 4097])
def test_ragged_sizes(oracle, solver, rsdsfm, n):
    d = rsdsfm.synth.make_config(1, rows=72, cols=96)
    q, u, a, ak = d["q"][:n], d["u"][:n], d["alpha"][:n], d["alpha_k"][:n]
    v, w, k = np.array([0.6, 0.7, 0.3]), np.array([0.01, -0.02, 0.008]), 0.1
    for mode in (0, 1):
        rho, sm = solver.estimate_inverse_depths(q, u, v, w, k, a, ak, mode=mode)
        rho_o, sm_o = oracle.estimate_inverse_depths(q, u, v, w, k, a, ak, mode=mode)
        assert rho.shape == (n,)
        assert np.allclose(rho, rho_o, rtol=1e-9, atol=1e-13)
        if mode == 1:
            _check_summary(sm, sm_o)


def test_single_pixel_variant(oracle, solver):
    # estimateInverseDepth (nonlinearRefinement.cc:55-106) == the dense solve with n = 1
    v, w = np.array([0.1, 0.9, 0.2]), np.array([0.01, 0.0, -0.01])
    r = solver.estimate_inverse_depth([0.1, -0.2], v, w, [0.02, 0.03], 0.0, 1.05, 0.55)
    ro, _ = oracle.estimate_inverse_depths([[0.1, -0.2]], [[0.02, 0.03]], v, w, 0.0, [1.05], [0.55], mode=1)
    assert np.isclose(r, ro[0], rtol=1e-12)


def test_many_lm_iterations_fallback_path(oracle, solver, rsdsfm):
    """Data built so that the LM needs more iterations than one speculative launch covers: a pixel whose
    Jacobian is tiny (clamped LM diagonal) converges slowly and keeps the step norm above the parameter
    tolerance.  Exercises the replanning / continuation launches."""
    d = rsdsfm.synth.make_config(1, rows=40, cols=48)
    q, u, a, ak = d["q"].copy(), d["u"].copy(), d["alpha"], d["alpha_k"]
    t = d["truth"]
    v = np.array([0.05, 0.03, 1.0])
    v /= np.linalg.norm(v)
    w = t["w"]
    # focus of expansion inside the image: pixels next to it have |J| ~ 0 -> clamped diagonal
    q[7] = [v[0] / v[2] + 1e-7, v[1] / v[2] - 2e-7]
    u[7] = [3e-2, -2e-2]
    rho, sm = solver.estimate_inverse_depths(q, u, v, w, 0.0, a, ak, mode=1)
    rho_o, sm_o = oracle.estimate_inverse_depths(q, u, v, w, 0.0, a, ak, mode=1)
    _check_summary(sm, sm_o)
    assert sm["num_iterations"] > 4
    assert np.allclose(rho, rho_o, rtol=1e-9, atol=1e-13)


def test_full_size_properties(solver, rsdsfm):
    """1280x720 (BASELINE config 2): size-independent properties.  (a) noise-free model data: closed-form
    depth recovers rho_true*|v| to 1e-10 relative; (b) LM iterate obeys rho_lm - rho* = (1 - rho*) * prod eps_t for
    unclamped pixels; (c) linearity: scaling (v -> s v) scales rho by 1/s."""
    d = rsdsfm.synth.make_config(2)
    q, u, a, ak, t = d["q"], d["u"], d["alpha"], d["alpha_k"], d["truth"]
    assert len(q) == 1280 * 720
    nv = np.linalg.norm(t["v"])
    v, w = t["v"] / nv, t["w"]
    rho0, _ = solver.estimate_inverse_depths(q, u, v, w, 0.0, a, ak, mode=0)
    rho_true = (1.0 / t["Z"]).T.reshape(-1) * nv
    assert np.allclose(rho0, rho_true, rtol=1e-10)
    rho1, sm = solver.estimate_inverse_depths(q, u, v, w, 0.0, a, ak, mode=1)
    eps = 1.0
    R = 1e4
    for _ in range(sm["num_successful_steps"]):
        eps *= (1.0 / R) / (1.0 + 1.0 / R)
        R *= 3
    assert np.allclose(rho1 - rho0, (1.0 - rho0) * eps, rtol=1e-6, atol=1e-15)
    rho_s, _ = solver.estimate_inverse_depths(q, u, 2.0 * v, w, 0.0, a, ak, mode=0)
    assert np.allclose(rho_s, rho0 / 2.0, rtol=1e-12)


@pytest.mark.parametrize("cfg", [2, 5])
def test_bench_depth_workload_matches_oracle(oracle, solver, rsdsfm, cfg):
    """exactly what `bench.py` times (1280x720, config 2 = the headline workload; config 5 = the DeepFlow-like pair): the
    oracle takes 0.1 s at this size, so the full-size result is compared directly -- decisions exact, depths to 1e-9"""
    d = rsdsfm.synth.make_config(cfg, seed=0x5EED0000 + cfg)
    q, u, a, ak, t = d["q"], d["u"], d["alpha"], d["alpha_k"], d["truth"]
    v = t["v"] / np.linalg.norm(t["v"])
    for mode in (0, 1):
        rho, sm = solver.estimate_inverse_depths(q, u, v, t["w"], 0.0, a, ak, mode=mode)
        rho_o, sm_o = oracle.estimate_inverse_depths(q, u, v, t["w"], 0.0, a, ak, mode=mode)
        assert np.allclose(rho, rho_o, rtol=1e-9, atol=1e-13)
        for key in ("num_iterations", "num_successful_steps", "num_unsuccessful_steps", "termination"):
            assert sm[key] == sm_o[key], (mode, key)
        if mode == 1:  # the closed-form path streams once and does not evaluate costs
            assert np.isclose(sm["final_cost"], sm_o["final_cost"], rtol=1e-9, atol=1e-18)


def test_alpha_and_pose_table(oracle, solver, rsdsfm):
    d = rsdsfm.synth.make_config(1, rows=60, cols=80)
    q, u, qpx, fpx = oracle.flatten(d["flow_img"], *d["K"], d["gamma"])
    assert np.array_equal(solver.get_alpha(fpx, 60, d["gamma"]), oracle.get_alpha(fpx, 60, d["gamma"]))
    assert np.array_equal(solver.get_alpha_k(qpx, fpx, 60, d["gamma"]), oracle.get_alpha_k(qpx, fpx, 60, d["gamma"]))
    v, w = np.array([0.03, 0.02, 0.01]), np.array([0.002, -0.003, 0.0087])
    for k in (0.0, 0.4):
        R, t = solver.pose_table(v, w, k, 0.8, 720)
        Ro, to = oracle.pose_table(v, w, k, 0.8, 720)
        assert np.array_equal(R, Ro) and np.array_equal(t, to)
        assert np.array_equal(R[0], np.eye(3)) and np.array_equal(t[0], np.zeros(3))


def test_batched_fast_path_equals_single_solves(oracle, rsdsfm):
    """rsdsfm_estimate_inverse_depths_batch_dev: several independent solves of DIFFERENT sizes / poses / data per launch give,
    solve by solve, bit-identical depths and the same summaries as the single-solve entry point, incl. a solve that needs the
    apply pass (cold predictor) and one that needs continuation launches (more LM iterations than one pass speculates)"""
    import torch

    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    with torch.cuda.stream(stream):
        cases = []
        for j, (cfg, rows, cols) in enumerate([(1, 100, 160), (3, 64, 90), (1, 40, 48), (5, 77, 129)]):
            d = rsdsfm.synth.make_config(cfg, rows=rows, cols=cols, v=np.array([0.05, 0.03, 1.0]) if j == 2 else None)
            q, u = d["q"].copy(), d["u"].copy()
            t = d["truth"]
            v = t["v"] / np.linalg.norm(t["v"])
            if j == 2:  # a pixel next to the epipole: many LM iterations -> continuation launches after the fast path
                q[7] = [v[0] / v[2] + 1e-7, v[1] / v[2] - 2e-7]
                u[7] = [3e-2, -2e-2]
            cases.append(dict(q=q, u=u, a=d["alpha"], ak=d["alpha_k"], v=v, w=t["w"], k=0.0 if j != 3 else 0.2))
        solvers = [rsdsfm.Solver(0, stream=stream.cuda_stream) for _ in cases]
        single = rsdsfm.Solver(0, stream=stream.cuda_stream)
        dev_t = [{k2: torch.from_numpy(np.ascontiguousarray(c[k2])).to(dev) for k2 in ("q", "u", "a", "ak")} for c in cases]
        rhos = [torch.zeros(len(c["a"]), dtype=torch.float64, device=dev) for c in cases]
        probs = [dict(d_q=t_["q"].data_ptr(), d_u=t_["u"].data_ptr(), d_alpha=t_["a"].data_ptr(), d_alpha_k=t_["ak"].data_ptr(), d_rho=r.data_ptr(),
                      n=len(c["a"]), v=c["v"], w=c["w"], k=c["k"]) for c, t_, r in zip(cases, dev_t, rhos)]
        call = rsdsfm.prepared_depth_batch(solvers, probs)
        for rep in range(3):  # rep 0: cold predictors (apply pass), later: warm
            call()
            for i, (c, p) in enumerate(zip(cases, probs)):
                sm, extra = solvers[i].depth_finish_dev(p["d_q"], p["d_u"], p["n"], c["v"], c["w"], c["k"], p["d_alpha"], p["d_alpha_k"], p["d_rho"])
                ref = torch.zeros(p["n"], dtype=torch.float64, device=dev)
                single.estimate_inverse_depths_dev(p["d_q"], p["d_u"], p["n"], c["v"], c["w"], c["k"], p["d_alpha"], p["d_alpha_k"], ref.data_ptr(), mode=1)
                sm1, _ = single.depth_finish_dev(p["d_q"], p["d_u"], p["n"], c["v"], c["w"], c["k"], p["d_alpha"], p["d_alpha_k"], ref.data_ptr())
                assert torch.equal(rhos[i], ref), (rep, i)
                for key in ("num_iterations", "num_successful_steps", "num_unsuccessful_steps", "termination"):
                    assert sm[key] == sm1[key], (rep, i, key)
                rho_o, sm_o = oracle.estimate_inverse_depths(c["q"], c["u"], c["v"], c["w"], c["k"], c["a"], c["ak"], mode=1)
                assert sm["num_iterations"] == sm_o["num_iterations"] and np.allclose(rhos[i].cpu().numpy(), rho_o, rtol=1e-9, atol=1e-13)
                if i == 2:
                    assert sm["num_iterations"] > 4 and extra > 0
        # argument checks: a context twice, contexts on different streams
        with pytest.raises(rsdsfm.RsdsfmError):
            rsdsfm.prepared_depth_batch([solvers[0], solvers[0]], probs[:2])()
        other = rsdsfm.Solver(0)
        with pytest.raises(rsdsfm.RsdsfmError):
            rsdsfm.prepared_depth_batch([solvers[0], other], probs[:2])()
        other.close()
        single.close()
        for s in solvers:
            s.close()


def test_nan_input_is_reported_not_hidden(oracle, solver, rsdsfm):
    """a NaN in the flow poisons the global LM sums: Ceres' loop (and the oracle) ends with FAILURE after 5 consecutive invalid
    steps; the HIP path walks the same state machine and reports it as an error instead of returning depths.  The closed-form
    mode has no global coupling: only the poisoned pixel is NaN, all others are bit-identical to the oracle's."""
    d = rsdsfm.synth.make_config(1, rows=40, cols=48)
    q, u, a, ak, t = d["q"], d["u"].copy(), d["alpha"], d["alpha_k"], d["truth"]
    v = t["v"] / np.linalg.norm(t["v"])
    u[5, 0] = np.nan
    _, sm_o = oracle.estimate_inverse_depths(q, u, v, t["w"], 0.0, a, ak, mode=1)
    assert sm_o["termination"] == 4 and sm_o["num_unsuccessful_steps"] == 5
    with pytest.raises(rsdsfm.RsdsfmError, match="LM failure"):
        solver.estimate_inverse_depths(q, u, v, t["w"], 0.0, a, ak, mode=1)
    rho0, _ = solver.estimate_inverse_depths(q, u, v, t["w"], 0.0, a, ak, mode=0)
    rho0_o, _ = oracle.estimate_inverse_depths(q, u, v, t["w"], 0.0, a, ak, mode=0)
    assert np.isnan(rho0[5]) and np.isnan(rho0_o[5]) and np.array_equal(np.delete(rho0, 5), np.delete(rho0_o, 5))
    # the context is still usable afterwards
    rho, sm = solver.estimate_inverse_depths(q, d["u"], v, t["w"], 0.0, a, ak, mode=1)
    rho_o, sm_o2 = oracle.estimate_inverse_depths(q, d["u"], v, t["w"], 0.0, a, ak, mode=1)
    assert sm["termination"] == sm_o2["termination"] and np.allclose(rho, rho_o, rtol=1e-9, atol=1e-13)


# ---------------------------------------------------------------------------------------------------
# launch 0 of the dense depth solve through the in-range function cores (depth_kernels.hip CORE)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg,rows,cols", [(1, 96, 128), (3, 135, 240), (5, 360, 640)])
def test_depth_function_cores_do_not_change_results(oracle, rsdsfm, cfg, rows, cols):
    """rsdsfm_set_ransac_math also governs launch 0 of the dense depth solve: in-range cores of sqrt / reciprocal in the Jacobi scaling
    (default) or the standard functions -- every inverse depth and every LM decision identical bit for bit, equal to the oracle's, and
    on real-valued data no solve has to start over"""
    d = rsdsfm.synth.make_config(cfg, rows=rows, cols=cols)
    q, u, a, ak, t = d["q"], d["u"], d["alpha"], d["alpha_k"], d["truth"]
    v = t["v"] / np.linalg.norm(t["v"])
    rho_o, sm_o = oracle.estimate_inverse_depths(q, u, v, t["w"], 0.0, a, ak, mode=1)
    outs = []
    with rsdsfm.Solver(0) as s:
        for mode in (0, 1, 0):
            s.set_ransac_math(mode)
            rho, sm = s.estimate_inverse_depths(q, u, v, t["w"], 0.0, a, ak, mode=1)
            _check_summary(sm, sm_o)
            assert np.allclose(rho, rho_o, rtol=1e-9, atol=1e-13)
            outs.append((rho.tobytes(), sm["num_iterations"], sm["num_successful_steps"], sm["termination"], sm["final_cost"], sm["final_radius"]))
        assert s.depth_restarts() == 0
    assert outs[0] == outs[1] == outs[2]


@pytest.mark.parametrize("poison", ["zero_jacobian", "zero_jacobian_zero_flow", "tiny_jacobian"])
def test_depth_function_cores_restart_on_arguments_out_of_range(oracle, rsdsfm, poison):
    """an argument outside the cores' range -- a Jacobian of 1e-160, whose square is a denormal -- leaves the fast path unfinished and
    rsdsfm_depth_finish_dev runs the solve again with the standard functions: results equal the oracle's and the standard-function
    setting's bit for bit, the restart is counted, the context keeps the standard functions for its next solves; also through the batched
    entry point.  A Jacobian that vanishes EXACTLY (alpha = alpha_k = 0: beta = 0; sqrt's argument is +0), with and without a residual, is
    a select inside the cores since round 5 (sqrt_core_z): same bits, no restart."""
    import torch

    d = rsdsfm.synth.make_config(5, rows=240, cols=320)
    q, u, a, ak, t = d["q"].copy(), d["u"].copy(), d["alpha"].copy(), d["alpha_k"].copy(), d["truth"]
    v = t["v"] / np.linalg.norm(t["v"])
    victim = 1536 * 7 + 100
    a[victim] = 1e-160 if poison == "tiny_jacobian" else 0.0
    ak[victim] = 0.0
    if poison == "zero_jacobian_zero_flow":
        u[victim] = 0.0
    expect = 1 if poison == "tiny_jacobian" else 0
    rho_o, sm_o = oracle.estimate_inverse_depths(q, u, v, t["w"], 0.0, a, ak, mode=1)
    with rsdsfm.Solver(0) as s:
        s.set_lm_arithmetic(1)  # the iterate-by-iterate kernels: the ones that run the cores
        rho0, sm0 = s.estimate_inverse_depths(q, u, v, t["w"], 0.0, a, ak, mode=1)
        assert s.depth_restarts() == expect
        _check_summary(sm0, sm_o)
        assert np.allclose(rho0, rho_o, rtol=1e-9, atol=1e-13, equal_nan=True)
        rho1, sm1 = s.estimate_inverse_depths(q, u, v, t["w"], 0.0, a, ak, mode=1)  # standard functions from the start: no second restart
        assert s.depth_restarts() == expect
        s.set_ransac_math(1)
        rho2, sm2 = s.estimate_inverse_depths(q, u, v, t["w"], 0.0, a, ak, mode=1)
    assert rho0.tobytes() == rho1.tobytes() == rho2.tobytes() and sm0 == sm1 == sm2
    # the batched entry point: one poisoned and one clean problem in the same launch
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    with torch.cuda.stream(stream):
        solvers = [rsdsfm.Solver(0, stream=stream.cuda_stream) for _ in range(2)]
        for s in solvers:
            s.set_lm_arithmetic(1)
        data = [(q, u, a, ak), (d["q"], d["u"], d["alpha"], d["alpha_k"])]
        dev_t = [[torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in c] for c in data]
        rhos = [torch.zeros(len(a), dtype=torch.float64, device=dev) for _ in data]
        probs = [dict(d_q=t_[0].data_ptr(), d_u=t_[1].data_ptr(), d_alpha=t_[2].data_ptr(), d_alpha_k=t_[3].data_ptr(), d_rho=r.data_ptr(), n=len(a), v=v, w=t["w"], k=0.0)
                 for t_, r in zip(dev_t, rhos)]
        call = rsdsfm.prepared_depth_batch(solvers, probs)
        for rep in range(2):
            call()
            for i, p in enumerate(probs):
                sm, _ = solvers[i].depth_finish_dev(p["d_q"], p["d_u"], p["n"], v, t["w"], 0.0, p["d_alpha"], p["d_alpha_k"], p["d_rho"])
                ro, so = oracle.estimate_inverse_depths(*data[i][:2], v, t["w"], 0.0, *data[i][2:], mode=1)
                _check_summary(sm, so)
                assert np.allclose(rhos[i].cpu().numpy(), ro, rtol=1e-9, atol=1e-13, equal_nan=True), (rep, i)
        assert solvers[0].depth_restarts() == expect and solvers[1].depth_restarts() == 0
        for s in solvers:
            s.close()
