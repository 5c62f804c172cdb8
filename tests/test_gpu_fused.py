"""GPU: what the opt-in FUSED-arithmetic library (librsdsfm_hip_fused.so, explicit fmas in the per-pixel model) changes.

The reference is built without FMA (src/CMakeLists.txt:18: plain -std=c++11), so the pinned target is the UNFUSED oracle,
which the default library matches bit for bit on every integer output (all other test_gpu_* files).  This file measures the
fused kernels against that same UNFUSED oracle on every BASELINE.json config at full size and asserts bounds:
  * dense 1/depth, v, w, k: <= 1e-5 relative (the north-star bar; measured ~1e-12),
  * LM decisions (accepted steps per trial, termination types, refinement iteration counts): identical,
  * inlier-mask flips per RANSAC trial: <= max(2, 1e-6 n) (a flip needs an error within rounding of the tolerance),
  * scanline indices (image row of every inlier both runs share) identical,
and, so that the fused path itself stays verified, compares it bit-exactly with the oracle's matching -DRSO_FUSED build.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL = 1e-5  # north-star tolerance on depth / pose


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))) if a.size else 0.0


def test_library_reports_its_arithmetic(rsdsfm):
    assert rsdsfm.load_library(arith="reference").rsdsfm_fused_arithmetic() == 0
    assert rsdsfm.load_library(arith="fused").rsdsfm_fused_arithmetic() == 1
    assert b"reference arithmetic" in rsdsfm.load_library().rsdsfm_version()


@pytest.mark.parametrize("cfg", [1, 2])
def test_fused_depth_solve_vs_unfused_oracle(oracle, rsdsfm, cfg):
    """BASELINE configs[0] (640x480) and configs[1] (1280x720, depth kernel only), full size"""
    d = rsdsfm.synth.make_config(cfg)
    t = d["truth"]
    v = t["v"] / np.linalg.norm(t["v"])
    rho_o, sm_o = oracle.estimate_inverse_depths(d["q"], d["u"], v, t["w"], 0.0, d["alpha"], d["alpha_k"], mode=1)
    with rsdsfm.Solver(0, arith="fused") as s:
        rho_f, sm_f = s.estimate_inverse_depths(d["q"], d["u"], v, t["w"], 0.0, d["alpha"], d["alpha_k"], mode=1)
    with rsdsfm.Solver(0) as s:
        rho_r, sm_r = s.estimate_inverse_depths(d["q"], d["u"], v, t["w"], 0.0, d["alpha"], d["alpha_k"], mode=1)
    for key in ("num_iterations", "num_successful_steps", "num_unsuccessful_steps", "termination"):
        assert sm_f[key] == sm_o[key] == sm_r[key], key
    dev_f, dev_r = _rel(rho_f, rho_o), _rel(rho_r, rho_o)
    print("config %d: fused vs unfused oracle max rel %.3e; reference-arithmetic library %.3e" % (cfg, dev_f, dev_r))
    assert dev_f <= REL
    assert dev_r <= 1e-9  # the default library: only the trust-region radius (a ratio of global sums) differs in its last bits
    with oracle.arithmetic("fused"):
        rho_of, sm_of = oracle.estimate_inverse_depths(d["q"], d["u"], v, t["w"], 0.0, d["alpha"], d["alpha_k"], mode=1)
    assert sm_of["num_successful_steps"] == sm_f["num_successful_steps"] and _rel(rho_f, rho_of) <= 1e-9


def _oracle_chain(oracle, d, T, tol, seed, accel=False):
    rows, K, gamma = d["rows"], d["K"], d["gamma"]
    q, u, qpx, fpx = oracle.flatten(d["flow_img"], *K, gamma)
    a, ak = oracle.get_alpha(fpx, rows, gamma), oracle.get_alpha_k(qpx, fpx, rows, gamma)
    ro = oracle.ransac(q, u, a, ak, accel, T, tol, oracle.sample_indices(len(q), T, seed), depth_mode=1)
    refo = oracle.refine(u, ro["inliers"], ro["alpha"], ro["alpha_k"], ro["v"], ro["w"], ro["k"], accel, 1, ro["inlier_idx"])
    inl, v, flipped = oracle.canonicalize_sign(refo["inliers"], refo["v"])
    dm, xs, ys = oracle.scatter_depth(inl, *K, d["rows"], d["cols"])
    return dict(q=q, u=u, a=a, ak=ak, ransac=ro, refine=refo, inliers=inl, v=v, flipped=flipped, depth_map=dm, ys=ys)


def _gpu_chain(rsdsfm, arith, d, o, T, tol, seed, accel=False):
    """stage by stage through the host-pointer API on the oracle's flattened arrays (so per-trial records come back)"""
    with rsdsfm.Solver(0, arith=arith) as s:
        rr = s.ransac(o["q"], o["u"], o["a"], o["ak"], accel, T, tol, samples=None, seed=seed, depth_mode=1)
        ref = s.non_linear_refinement(o["u"], rr["inliers"], rr["alpha"], rr["alpha_k"], rr["v"], rr["w"], rr["k"], accel,
                                      flow_index_mode=1, inlier_idx=rr["inlier_idx"])
        dmap = s.depth_map(ref["inliers"], ref["v"], d["K"], d["rows"], d["cols"])
    return rr, ref, dmap


# (config, rows, cols, trials, tol): configs[2] 1920x1080 with main.cc's 5 trials, configs[3] 3840x2160, configs[4] = the bench
# workload (1280x720 DeepFlow-like pair, 50 trials, tol 0.05); the 4K oracle chain takes ~20 s on one host core
@pytest.mark.parametrize("cfg,T,tol,seed", [(3, 5, 0.002, 2024), (4, 5, 0.002, 5), (5, 50, 0.05, 1)])
def test_fused_whole_solve_vs_unfused_oracle(oracle, rsdsfm, big_config, oracle_chain, cfg, T, tol, seed):
    data_seed = 0x5EED0005 if cfg == 5 else None
    d = big_config(cfg, data_seed)
    o = oracle_chain(cfg, T, tol, seed, data_seed=data_seed)
    ro, n = o["ransac"], len(o["q"])
    rr, ref, dmap = _gpu_chain(rsdsfm, "fused", d, o, T, tol, seed)
    # LM decisions of every trial's depth solve: identical
    assert np.array_equal(rr["trial_steps"], ro["trial_steps"])
    # inlier-mask flips: per trial from the counts (lower bound on flips, exact when flips go one way), exact for the winner
    count_diff = np.abs(rr["trial_count"].astype(np.int64) - ro["trial_count"].astype(np.int64))
    bound = max(2, int(1e-6 * n))
    assert rr["best_trial"] == ro["best_trial"]
    flips = int(np.count_nonzero(rr["mask"] != ro["mask"]))
    print("config %d (n = %d, T = %d): per-trial |count difference| max %d, winner-mask flips %d (bound %d)" %
          (cfg, n, T, int(count_diff.max()), flips, bound))
    assert int(count_diff.max()) <= bound and flips <= bound
    # dense 1/depth of the winner, pose after refinement
    assert _rel(rr["inv_depth"], ro["inv_depth"]) <= REL
    for key in ("num_iterations", "num_successful_steps", "termination"):
        assert ref["summary"][key] == o["refine"]["summary"][key], key
    assert dmap["flipped"] == o["flipped"]
    dv, dw = _rel(dmap["v"], o["v"]), float(np.max(np.abs(ref["w"] - o["refine"]["w"])) / np.linalg.norm(o["refine"]["w"]))
    print("   v max rel %.3e, w rel %.3e" % (dv, dw))
    assert dv <= REL and dw <= REL
    # scanline indices: identical for every inlier both runs share; depth values <= 1e-5 there
    common = np.intersect1d(rr["inlier_idx"], ro["inlier_idx"], assume_unique=True)
    ys_f = dmap["ys"][np.searchsorted(rr["inlier_idx"], common)]
    ys_o = o["ys"][np.searchsorted(ro["inlier_idx"], common)]
    assert np.array_equal(ys_f, ys_o)
    both = (dmap["depth_map"] != 0) & (o["depth_map"] != 0)
    assert int(np.count_nonzero((dmap["depth_map"] != 0) != (o["depth_map"] != 0))) <= bound
    assert _rel(dmap["depth_map"][both], o["depth_map"][both]) <= REL


def test_fused_library_matches_fused_oracle_bit_exactly(oracle, rsdsfm):
    """the fused kernels against the oracle build that fuses at the same places: integers bit-exact as in the default pair"""
    d = rsdsfm.synth.make_config(5, rows=360, cols=640)
    T, tol, seed = 20, 0.01, 3
    with oracle.arithmetic("fused"):
        o = _oracle_chain(oracle, d, T, tol, seed)
    rr, ref, dmap = _gpu_chain(rsdsfm, "fused", d, o, T, tol, seed)
    ro = o["ransac"]
    assert np.array_equal(rr["trial_count"], ro["trial_count"]) and np.array_equal(rr["trial_steps"], ro["trial_steps"])
    assert rr["best_trial"] == ro["best_trial"] and np.array_equal(rr["mask"], ro["mask"])
    assert np.array_equal(rr["inlier_idx"], ro["inlier_idx"]) and np.array_equal(dmap["ys"], o["ys"])
    assert np.allclose(rr["inv_depth"], ro["inv_depth"], rtol=1e-9, atol=1e-13)
    assert np.allclose(dmap["v"], o["v"], rtol=1e-6, atol=1e-10)


def test_fused_true_flow_vs_unfused_oracle(oracle, rsdsfm):
    """ground-truth flow search: fused projections against the unfused oracle -- a winning scanline can only
    change where two scanlines are equally close to rounding; flows <= 1e-9 relative wherever the winner is the same"""
    rows, cols = 180, 320
    d = rsdsfm.synth.make_config(1, rows=rows, cols=cols)
    K, t = d["K"], d["truth"]
    Z = t["Z"]
    xi, yj = np.meshgrid(np.arange(cols, dtype=np.float64), np.arange(rows, dtype=np.float64))
    world = np.stack([(xi - K[2]) / K[0] * Z, (yj - K[3]) / K[1] * Z, Z], axis=-1)
    R2, t2 = oracle.pose_table(t["v"] * 3.0, t["w"] * 3.0, 0.0, d["gamma"], rows)
    flow_o, best_o = oracle.true_flow(world, R2, t2, *K)
    with rsdsfm.Solver(0, arith="fused") as s:
        flow_f, best_f = s.true_flow(world, R2, t2, K)
    with rsdsfm.Solver(0) as s:
        flow_r, best_r = s.true_flow(world, R2, t2, K)
    assert np.array_equal(best_r, best_o) and np.array_equal(flow_r, flow_o)  # default library: bit-exact
    changed = int(np.count_nonzero(best_f != best_o))
    print("true flow %dx%d: %d winning scanlines differ under fused arithmetic" % (rows, cols, changed))
    assert changed <= max(2, rows * cols // 10000)
    same = best_f == best_o
    assert np.allclose(flow_f[same], flow_o[same], rtol=1e-9, atol=1e-9)
    with oracle.arithmetic("fused"):
        flow_of, best_of = oracle.true_flow(world, R2, t2, *K)
    assert np.array_equal(best_f, best_of) and np.array_equal(flow_f, flow_of)
