"""GPU: the row-tiled whole-frame solve (dist.TiledFrameSolve over the rsdsfm_tile_* stage entry points) against the
single-context solve (rsdsfm_solve_frame_dev): 1, 2, 3 and 5 column slabs on one device (logical shards) and two
PROCESSES sharing the device over a gloo group.  Integer results (point / inlier counts, winner, per-trial counts,
LM step counts, depth-map support) are bit-exact; floating results agree to 1e-9 relative (the global sums are added
per slab first, so only their summation order differs)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _single(rsdsfm, torch, d, stream, **kw):
    dev = torch.device("cuda", 0)
    rows, cols, K, gamma = d["rows"], d["cols"], d["K"], d["gamma"]
    img = torch.from_numpy(d["flow_img"]).to(dev)
    dm = torch.empty((cols, rows), dtype=torch.float64, device=dev)
    R = torch.empty((rows, 9), dtype=torch.float64, device=dev)
    t = torch.empty((rows, 3), dtype=torch.float64, device=dev)
    kw = dict(kw)
    kw.setdefault("flow_index_mode", rsdsfm.FLOW_GATHERED)  # (tests of the reference's rank-indexed flow pass FLOW_COMPAT_RANK)
    with rsdsfm.Solver(0, stream=stream.cuda_stream) as s:
        r = s.solve_frame_dev(img.data_ptr(), rows, cols, K, gamma, dm.data_ptr(), R.data_ptr(), t.data_ptr(), **kw)
        s.synchronize()
        r["depth_map"] = dm.cpu().numpy().reshape(-1)
        r["R"], r["t"] = R.cpu().numpy(), t.cpu().numpy()
    return r


def _tiled(rsdsfm, torch, d, nslabs, stream, **kw):
    dev = torch.device("cuda", 0)
    rows, cols, K, gamma = d["rows"], d["cols"], d["K"], d["gamma"]
    bounds, per = rsdsfm.dist.slab_bounds(cols, nslabs)
    solvers = [rsdsfm.Solver(0, stream=stream.cuda_stream) for _ in range(nslabs)]
    img = torch.from_numpy(d["flow_img"]).to(dev)
    shards = [rsdsfm.dist.HipFrameShard(s, img[:, c0:c1, :].contiguous(), c0, K, gamma, torch)
              for s, (c0, c1) in zip(solvers, bounds)]
    drv = rsdsfm.dist.TiledFrameSolve(shards, rows, cols, per, torch, None)
    kw = dict(kw)
    kw.setdefault("flow_index_mode", rsdsfm.FLOW_GATHERED)
    r = drv.solve(pose_table=True, **kw)
    torch.cuda.synchronize()
    r["depth_map"] = r["depth_map"].cpu().numpy()
    r["R"], r["t"] = r["R"].cpu().numpy().reshape(rows, 9), r["t"].cpu().numpy().reshape(rows, 3)
    r["inliers"] = np.concatenate([sh.final[: 3 * sh.m].cpu().numpy().reshape(-1, 3) for sh in shards])
    r["ys"] = np.concatenate([sh.ys[: sh.m].cpu().numpy() for sh in shards])
    r["shard_n"] = [sh.n for sh in shards]
    for s in solvers:
        s.close()
    return r


def _compare(a, b, rows, cols, depth_rtol=1e-9):
    assert a["n"] == b["n"] and a["num_inliers"] == b["num_inliers"] and a["best_trial"] == b["best_trial"]
    assert a["flipped"] == b["flipped"]
    # the winner's hypothesis comes from the same 9 points through the same kernel
    assert np.array_equal(a["ransac_w"], b["ransac_w"]) and np.array_equal(a["ransac_v"], b["ransac_v"]) and a["ransac_k"] == b["ransac_k"]
    for key in ("num_iterations", "num_successful_steps", "num_unsuccessful_steps", "termination"):
        assert a["refine_summary"][key] == b["refine_summary"][key], key
    assert np.isclose(a["refine_summary"]["final_cost"], b["refine_summary"]["final_cost"], rtol=1e-9)
    assert np.allclose(a["v"], b["v"], rtol=1e-9, atol=1e-14) and np.allclose(a["w"], b["w"], rtol=1e-9, atol=1e-14)
    assert np.isclose(a["k"], b["k"], rtol=1e-9, atol=1e-14)
    da, db = a["depth_map"], b["depth_map"]
    assert da.shape == db.shape == (rows * cols,)
    assert np.array_equal(da != 0, db != 0)
    nz = db != 0
    rel = np.abs(da[nz] - db[nz]) / np.abs(db[nz])
    assert rel.max() < depth_rtol, rel.max()
    assert np.allclose(a["R"], b["R"], rtol=1e-9, atol=1e-14) and np.allclose(a["t"], b["t"], rtol=1e-9, atol=1e-14)


@pytest.mark.parametrize("cfg,accel", [(3, False), (5, True)])
def test_tiled_frame_matches_single_context(rsdsfm, cfg, accel):
    import torch

    stream = torch.cuda.Stream(torch.device("cuda", 0))
    with torch.cuda.stream(stream):
        d = rsdsfm.synth.make_config(cfg, rows=96, cols=250)
        rows, cols = d["rows"], d["cols"]
        kw = dict(trials=14, tol=0.002 if cfg == 3 else 0.01, seed=7, use_acceleration_mode=accel)
        one = _single(rsdsfm, torch, d, stream, **kw)
        assert one["num_inliers"] > 0.1 * rows * cols
        for nslabs in (1, 2, 3, 5):
            til = _tiled(rsdsfm, torch, d, nslabs, stream, **kw)
            assert sum(til["shard_n"]) == one["n"]
            # the 7-parameter problem (k free) is worse conditioned: summation-order noise is amplified
            _compare(til, one, rows, cols, depth_rtol=1e-6 if accel else 1e-9)
            assert len(til["inliers"]) == one["num_inliers"] and len(til["ys"]) == one["num_inliers"]


def test_tiled_3840x2160_in_8_slabs_matches_single_context_and_oracle(rsdsfm, oracle_chain, big_config):
    """BASELINE configs[3] at full size: the 3840x2160 DeepFlow-like frame as 8 column slabs (8 logical shards = 8 solver
    contexts on the one GPU of the test box, exactly the per-rank call sequence of the 8-GPU run) against the un-tiled solve and
    against the oracle chain with main.cc's 5 trials: every integer (points, inliers, winner, LM decisions, depth-map support,
    scanline indices) exact, floats to 1e-6 vs the oracle and 1e-9 between the two GPU paths"""
    import torch

    stream = torch.cuda.Stream(torch.device("cuda", 0))
    T, tol, seed = 5, 0.002, 5
    with torch.cuda.stream(stream):
        d = big_config(4)
        rows, cols, K, gamma = d["rows"], d["cols"], d["K"], d["gamma"]
        kw = dict(trials=T, tol=tol, seed=seed)
        one = _single(rsdsfm, torch, d, stream, **kw)
        til = _tiled(rsdsfm, torch, d, 8, stream, **kw)
    assert len(til["shard_n"]) == 8 and sum(til["shard_n"]) == one["n"] == rows * cols
    _compare(til, one, rows, cols)
    # oracle chain (gathered flow, as the tiled solve uses it)
    o = oracle_chain(4, T, tol, seed)
    ro, refo = o["ransac"], o["refine"]
    assert til["num_inliers"] == ro["num_inliers"] and til["best_trial"] == ro["best_trial"]
    for key in ("num_iterations", "num_successful_steps", "termination"):
        assert til["refine_summary"][key] == refo["summary"][key], key
    assert til["flipped"] == o["flipped"] and np.allclose(til["v"], o["v"], rtol=1e-6, atol=1e-10) and np.allclose(til["w"], refo["w"], rtol=1e-6, atol=1e-10)
    assert np.array_equal(til["ys"], o["ys"])  # scanline index of every inlier: bit-exact
    got = til["depth_map"].reshape(cols, rows).T
    assert np.array_equal(got != 0, o["depth_map"] != 0) and np.allclose(got, o["depth_map"], rtol=1e-6)


def test_tiled_frame_refinement_trace_is_replicated_and_matches_single_context(rsdsfm):
    """the refinement's iteration log (rsdsfm_set_refine_trace) in the column-tiled solve: every slab's context takes the same
    decisions on the same gathered sum rows, so all logs are bit-identical; against the single-context solve every outcome is equal
    and the costs agree to 1e-10 (the sums are added per slab first)"""
    import torch

    d = rsdsfm.synth.make_config(3, rows=96, cols=250)
    stream = torch.cuda.Stream(torch.device("cuda", 0))
    dev = torch.device("cuda", 0)
    rows, cols, K, gamma = d["rows"], d["cols"], d["K"], d["gamma"]
    kw = dict(trials=12, tol=0.01, seed=5)
    with torch.cuda.stream(stream):
        img = torch.from_numpy(d["flow_img"]).to(dev)
        dm = torch.empty((cols, rows), dtype=torch.float64, device=dev)
        with rsdsfm.Solver(0, stream=stream.cuda_stream) as s:
            s.set_refine_trace(50)
            r1 = s.solve_frame_dev(img.data_ptr(), rows, cols, K, gamma, dm.data_ptr(), 0, 0, flow_index_mode=rsdsfm.FLOW_GATHERED, **kw)
            t1 = s.get_refine_trace()
        bounds, per = rsdsfm.dist.slab_bounds(cols, 3)
        solvers = [rsdsfm.Solver(0, stream=stream.cuda_stream) for _ in range(3)]
        for sv in solvers:
            sv.set_refine_trace(50)
        shards = [rsdsfm.dist.HipFrameShard(sv, img[:, c0:c1, :].contiguous(), c0, K, gamma, torch) for sv, (c0, c1) in zip(solvers, bounds)]
        r3 = rsdsfm.dist.TiledFrameSolve(shards, rows, cols, per, torch, None).solve(flow_index_mode=rsdsfm.FLOW_GATHERED, **kw)
        traces = [sv.get_refine_trace() for sv in solvers]
        for sv in solvers:
            sv.close()
    n_it = r1["refine_summary"]["num_iterations"]
    assert n_it >= 2 and r3["refine_summary"]["num_iterations"] == n_it
    for t in traces[1:]:
        assert np.array_equal(t, traces[0], equal_nan=True)
    t3 = traces[0]
    assert np.array_equal(t3[:, 7], t1[:, 7], equal_nan=True) and np.array_equal(np.isnan(t3), np.isnan(t1))
    assert np.allclose(t3[:n_it, 1:3], t1[:n_it, 1:3], rtol=1e-10, atol=0.0, equal_nan=True)
    assert np.allclose(t3[:n_it, 5], t1[:n_it, 5], rtol=1e-9)


def test_tiled_frame_closed_form_no_refinement_and_many_trials(rsdsfm):
    """closed-form depth mode (no LM rounds, score pass only), no refinement, and more trials than one hypothesis
    batch (> 128) so that the batched rows / decide calls are offset correctly"""
    import torch

    stream = torch.cuda.Stream(torch.device("cuda", 0))
    with torch.cuda.stream(stream):
        d = rsdsfm.synth.make_config(3, rows=64, cols=90)
        rows, cols = d["rows"], d["cols"]
        for mode, T in ((0, 20), (1, 150)):
            kw = dict(trials=T, tol=0.002, seed=3, use_refinement=False, depth_mode=mode)
            one = _single(rsdsfm, torch, d, stream, **kw)
            til = _tiled(rsdsfm, torch, d, 3, stream, **kw)
            assert til["n"] == one["n"] and til["num_inliers"] == one["num_inliers"] and til["best_trial"] == one["best_trial"]
            assert np.array_equal(til["v"], one["v"]) and np.array_equal(til["w"], one["w"])
            # no global sums enter the per-pixel results here: bit-exact
            assert np.array_equal(til["depth_map"], one["depth_map"])
            assert len(til["trial_count"]) == T


def test_tiled_frame_empty_slab_and_too_few_points(rsdsfm):
    """more slabs than the image supports (empty trailing slabs) and a frame whose flow is below the threshold in all
    but a few pixels (the reference would call rand() % 0: refused)"""
    import torch

    stream = torch.cuda.Stream(torch.device("cuda", 0))
    with torch.cuda.stream(stream):
        d = rsdsfm.synth.make_config(1, rows=40, cols=7)
        kw = dict(trials=6, tol=0.01, seed=1)
        one = _single(rsdsfm, torch, d, stream, **kw)
        til = _tiled(rsdsfm, torch, d, 4, stream, **kw)  # per = 2 -> slabs (0,2) (2,4) (4,6) (6,7)
        _compare(til, one, d["rows"], d["cols"])
        d2 = dict(d)
        img = np.zeros_like(d["flow_img"])
        img[:2, :3] = d["flow_img"][:2, :3]  # 6 points < 9
        d2["flow_img"] = img
        with pytest.raises(ValueError):
            _tiled(rsdsfm, torch, d2, 2, stream, **kw)


@pytest.mark.parametrize("cfg,accel", [(3, False), (5, False)])  # (k free on this mismatched problem wanders for ~50 iterations: chaotic)
def test_tiled_frame_rank_indexed_flow_matches_single_context(rsdsfm, cfg, accel):
    """the Python driver with the reference's default flow indexing (quirk Q2) and a selective tolerance: the flow columns a slab's
    inliers read by global RANK are fetched from the slabs in front of it (dist.TiledFrameSolve.rank_indexed_flow)"""
    import torch

    stream = torch.cuda.Stream(torch.device("cuda", 0))
    with torch.cuda.stream(stream):
        d = rsdsfm.synth.make_config(cfg, rows=96, cols=250)
        rows, cols = d["rows"], d["cols"]
        kw = dict(trials=14, tol=0.002 if cfg == 3 else 0.004, seed=7, use_acceleration_mode=accel, flow_index_mode=rsdsfm.FLOW_COMPAT_RANK)
        one = _single(rsdsfm, torch, d, stream, **kw)
        assert 0.1 * rows * cols < one["num_inliers"] < 0.97 * rows * cols
        for nslabs in (1, 2, 3, 5):
            til = _tiled(rsdsfm, torch, d, nslabs, stream, **kw)
            _compare(til, one, rows, cols, depth_rtol=1e-6 if accel else 1e-9)


@pytest.mark.parametrize("flow_mode", [1, 0])
def test_tiled_frame_two_processes_share_the_gpu(rsdsfm, tmp_path, flow_mode):
    """two ranks (gloo rendezvous on 127.0.0.1, both on cuda:0) run tests/mp_tiled_frame.py; rank 0's result equals the
    single-context solve (flow_mode 0: the reference's rank-indexed flow, fetched across the two processes)"""
    import torch

    out = tmp_path / "res.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", RSDSFM_TILED_OUT=str(out), RSDSFM_TEST_FLOW_MODE=str(flow_mode))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(29631 + flow_mode),
           os.path.join(ROOT, "tests", "mp_tiled_frame.py")]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    got = json.loads(out.read_text())
    stream = torch.cuda.Stream(torch.device("cuda", 0))
    with torch.cuda.stream(stream):
        d = rsdsfm.synth.make_config(3, rows=96, cols=250)
        one = _single(rsdsfm, torch, d, stream, trials=14, tol=0.002, seed=7, flow_index_mode=flow_mode)
    assert got["world"] == 2 and got["n"] == one["n"] and got["num_inliers"] == one["num_inliers"] and got["best_trial"] == one["best_trial"]
    assert np.allclose(got["v"], one["v"], rtol=1e-9) and np.allclose(got["w"], one["w"], rtol=1e-9)
    dm = np.asarray(got["depth_nonzero"]), np.asarray(got["depth_sum"])
    assert int(dm[0]) == int((one["depth_map"] != 0).sum()) and np.isclose(float(dm[1]), one["depth_map"].sum(), rtol=1e-9)
    assert got["ranks_agree"]
