"""GPU: the column-tiled whole solve driven from C++ inside the library (rsdsfm_solve_frame_tiled_dev, csrc/dist_host.hip).

The GPU box has ONE device and RCCL refuses two ranks on one GPU, so
  * the RCCL code path (dlopen, ncclCommInitRank from a unique id, ncclAllGather / ncclAllReduce on the context's stream) is
    exercised with a 1-rank communicator, and
  * the multi-rank logic (slab bounds, offsets, rank-ordered rows, padded depth-map gather) with several LOGICAL ranks on the one
    GPU through rsdsfm_dist_set_transport: N host threads (tests/transports.ThreadTransport) and 2 processes over gloo.
Every variant must reproduce the single-context solve: integers exact, floats to the summation order of the per-slab sums."""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from test_gpu_tiled_frame import _compare, _single  # noqa: E402


def _native_threads(rsdsfm, torch, d, nranks, **kw):
    """nranks logical ranks = host threads, one context each, collectives through ThreadTransport"""
    from transports import ThreadTransport

    dev = torch.device("cuda", 0)
    rows, cols, K, gamma = d["rows"], d["cols"], d["K"], d["gamma"]
    img = torch.from_numpy(d["flow_img"]).to(dev)
    tr = ThreadTransport(nranks, timeout=float(__import__("os").environ.get("RSDSFM_TEST_BARRIER_TIMEOUT", "120")))
    outs, errs = [None] * nranks, [None] * nranks
    kw = dict(kw)
    kw.setdefault("flow_index_mode", rsdsfm.FLOW_GATHERED)  # (tests of the reference's rank-indexed flow pass FLOW_COMPAT_RANK)
    refine_arithmetic = kw.pop("refine_arithmetic", 0)

    def work(rank):
        try:
            torch.cuda.set_device(0)
            c0, sc, per = rsdsfm.tiled_slab_bounds(cols, nranks, rank)
            slab = img[:, c0:c0 + sc, :].contiguous()
            dm = torch.zeros(cols * rows, dtype=torch.float64, device=dev)
            R = torch.empty(rows * 9, dtype=torch.float64, device=dev)
            t = torch.empty(rows * 3, dtype=torch.float64, device=dev)
            torch.cuda.synchronize()
            with rsdsfm.Solver(0) as s:
                s.set_refine_arithmetic(refine_arithmetic)
                s.dist_set_transport(nranks, rank, *tr.callbacks(rank))
                r = s.solve_frame_tiled_dev(slab.data_ptr() if sc else 0, rows, cols, K, gamma, dm.data_ptr(), R.data_ptr(), t.data_ptr(), **kw)
                s.synchronize()
                m = r["info"]["shard_inliers"]
                inl = torch.empty(3 * max(m, 1), dtype=torch.float64, device=dev)
                ys = torch.empty(max(m, 1), dtype=torch.int32, device=dev)
                if m:
                    assert tr.hip.hipMemcpy(inl.data_ptr(), r["d_inliers"], 24 * m, 3) == 0
                    assert tr.hip.hipMemcpy(ys.data_ptr(), r["d_scanline"], 4 * m, 3) == 0
                r["inliers"], r["ys"] = inl.cpu().numpy().reshape(-1, 3)[:m], ys.cpu().numpy()[:m]
            r["depth_map"] = dm.cpu().numpy()
            r["R"], r["t"] = R.cpu().numpy().reshape(rows, 9), t.cpu().numpy().reshape(rows, 3)
            outs[rank] = r
        except Exception as e:  # noqa: BLE001
            errs[rank] = e
            tr.barrier.abort()

    ths = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    # (the error that matters is the one that is not "a peer left": a rank that fails breaks the barrier for the others)
    real = [e for e in errs if e is not None and "all-gather failed" not in str(e) and "all-reduce failed" not in str(e)]
    for e in real + [e for e in errs if e is not None]:
        raise e
    r0 = outs[0]
    for r in outs[1:]:  # every rank returns the same pose, counts and the same full depth map
        assert r["n"] == r0["n"] and r["num_inliers"] == r0["num_inliers"] and r["best_trial"] == r0["best_trial"]
        assert np.array_equal(r["v"], r0["v"]) and np.array_equal(r["w"], r0["w"]) and r["k"] == r0["k"]
        assert r["refine_summary"] == r0["refine_summary"] and np.array_equal(r["depth_map"], r0["depth_map"])
    res = dict(r0)
    res["inliers"] = np.concatenate([r["inliers"] for r in outs])
    res["ys"] = np.concatenate([r["ys"] for r in outs])
    res["shard_n"] = [r["info"]["shard_points"] for r in outs]
    res["infos"] = [r["info"] for r in outs]
    return res


def test_native_tiled_warm_path_and_a_frame_that_is_not_dense_after_all(rsdsfm):
    """A sequence on ONE set of contexts: the second solve of a shape goes ahead on the point counts of a dense frame (no wait for the counts
    exchange: one host synchronisation and one collective less), a frame with dropped pixels then breaks that assumption -- every rank
    notices with its first host read, all start over through the counts exchange -- and the frame after it waits again.  Every solve
    equals the single-context solve of its frame."""
    import torch
    from transports import ThreadTransport

    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    d = rsdsfm.synth.make_config(3, rows=96, cols=250)
    rows, cols, K, gamma = d["rows"], d["cols"], d["K"], d["gamma"]
    holed = d["flow_img"].copy()
    holed[20:40, 100:130] = 0.0
    frames = [d["flow_img"], d["flow_img"], d["flow_img"], holed, d["flow_img"], d["flow_img"], d["flow_img"]]
    kw = dict(trials=14, tol=0.002, flow_index_mode=rsdsfm.FLOW_GATHERED)
    singles = []
    with torch.cuda.stream(stream):
        for i, f in enumerate(frames):
            singles.append(_single(rsdsfm, torch, dict(d, flow_img=f), stream, seed=3, **kw))
    nranks = 3
    tr = ThreadTransport(nranks, timeout=float(__import__("os").environ.get("RSDSFM_TEST_BARRIER_TIMEOUT", "120")))
    outs, errs = [[] for _ in range(nranks)], [None] * nranks
    imgs = [torch.from_numpy(f).to(dev) for f in frames]

    def work(rank):
        try:
            torch.cuda.set_device(0)
            c0, sc, per = rsdsfm.tiled_slab_bounds(cols, nranks, rank)
            with rsdsfm.Solver(0) as s:
                s.dist_set_transport(nranks, rank, *tr.callbacks(rank))
                for i, img in enumerate(imgs):
                    slab = img[:, c0:c0 + sc, :].contiguous()
                    dm = torch.zeros(cols * rows, dtype=torch.float64, device=dev)
                    torch.cuda.synchronize()
                    r = s.solve_frame_tiled_dev(slab.data_ptr(), rows, cols, K, gamma, dm.data_ptr(), seed=3, **kw)
                    s.synchronize()
                    r["depth_map"] = dm.cpu().numpy()
                    outs[rank].append(r)
        except Exception as e:  # noqa: BLE001
            errs[rank] = e
            tr.barrier.abort()

    ths = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    for e in errs:
        if e is not None:
            raise e
    syncs = [outs[0][i]["info"]["host_syncs"] for i in range(len(frames))]
    colls = [outs[0][i]["info"]["collectives"] for i in range(len(frames))]
    for i, one in enumerate(singles):
        for rank in range(nranks):
            r = outs[rank][i]
            assert r["n"] == one["n"] and r["num_inliers"] == one["num_inliers"] and r["best_trial"] == one["best_trial"], (i, rank)
            assert np.array_equal(r["ransac_v"], one["ransac_v"]) and r["refine_summary"]["num_iterations"] == one["refine_summary"]["num_iterations"], (i, rank)
            assert np.allclose(r["v"], one["v"], rtol=1e-9, atol=1e-14) and np.allclose(r["w"], one["w"], rtol=1e-9, atol=1e-14)
            assert np.array_equal(r["depth_map"] != 0, one["depth_map"] != 0) and np.allclose(r["depth_map"], one["depth_map"], rtol=1e-9)
    assert singles[3]["n"] < rows * cols == singles[0]["n"]
    # the path every solve took (identical on all ranks): cold / ahead on dense counts / ahead / ahead but the frame has a hole: started over /
    # the previous frame was not dense: cold / ahead / ahead
    for rank in range(nranks):
        assert [o["info"]["path_flags"] & 0xEF for o in outs[rank]] == [0, 1, 1, 3, 0, 1, 1], rank
        # bit 4: the refinement went behind the speculated final stage, from the device-resident winner, and counted -- once that stage has
        # counted in a previous solve (never on the first solve of a communicator, nor behind a frame whose RANSAC started over)
        assert [(o["info"]["path_flags"] >> 4) & 1 for o in outs[rank]][:3] == [0, 1, 1] and (outs[rank][6]["info"]["path_flags"] >> 4) & 1 == 1, rank
    # same frame, same seed, hints settled: going ahead saves exactly one host synchronisation and one collective
    assert syncs[6] == syncs[5] == syncs[2] and colls[6] == colls[5] == colls[2], (syncs, colls)
    assert syncs[3] > syncs[2] and colls[3] > colls[2], (syncs, colls)


def test_native_tiled_function_cores_and_a_restart_on_one_rank(rsdsfm):
    """The tiled RANSAC runs the minimal solver's SVD and round 0 of the LM solves through the in-range function cores like the single-context
    solve.  A pixel with a flow of 1e160 px (alpha_k overflows, beta = alpha + 0 x inf = NaN: a NaN pixel) lies in ONE
    rank's slab: that rank's rows carry the flag, every rank sees it, all start the RANSAC over with the standard functions (path_flags bit
    2) -- and the results equal the single-context solve's and those of a communicator set to the standard functions from the start."""
    import torch
    from transports import ThreadTransport

    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    d = rsdsfm.synth.make_config(5, rows=64, cols=480)
    rows, cols, K = d["rows"], d["cols"], d["K"]
    gamma = 0.5
    clean = np.array(d["flow_img"])
    bad = clean.copy()
    bad[17, 301] = (3.0, 1e160)  # column 301: the second of three slabs
    kw = dict(trials=20, tol=0.05, seed=11, flow_index_mode=rsdsfm.FLOW_GATHERED)
    nranks = 3
    results = {}
    for math in (0, 1, 2):  # 2: the default arithmetic of the depth solves (analytic LM trajectory): no cores to leave, same bits
        tr = ThreadTransport(nranks, timeout=float(__import__("os").environ.get("RSDSFM_TEST_BARRIER_TIMEOUT", "120")))
        outs, errs = [[] for _ in range(nranks)], [None] * nranks
        imgs = [torch.from_numpy(f).to(dev) for f in (clean, bad, clean)]

        def work(rank):
            try:
                torch.cuda.set_device(0)
                c0, sc, per = rsdsfm.tiled_slab_bounds(cols, nranks, rank)
                with rsdsfm.Solver(0) as s:
                    s.set_ransac_math(math & 1)
                    s.set_lm_arithmetic(1 if math < 2 else 0)
                    s.set_refine_arithmetic(1)  # (the same refinement behind the three forms of the depth solves: compared bit for bit)
                    s.dist_set_transport(nranks, rank, *tr.callbacks(rank))
                    for img in imgs:
                        slab = img[:, c0:c0 + sc, :].contiguous()
                        dm = torch.zeros(cols * rows, dtype=torch.float64, device=dev)
                        torch.cuda.synchronize()
                        r = s.solve_frame_tiled_dev(slab.data_ptr(), rows, cols, K, gamma, dm.data_ptr(), **kw)
                        s.synchronize()
                        outs[rank].append((r["n"], r["num_inliers"], r["best_trial"], r["ransac_v"].tobytes(), r["v"].tobytes(), r["w"].tobytes(),
                                           str(r["refine_summary"]), dm.cpu().numpy().tobytes(), r["info"]["path_flags"] & 4))
                    outs[rank].append(s.ransac_restarts())
            except Exception as e:  # noqa: BLE001
                errs[rank] = e
                tr.barrier.abort()

        ths = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        for e in errs:
            if e is not None:
                raise e
        for rank in range(1, nranks):
            assert outs[rank] == outs[0], (math, rank)
        assert [o[8] for o in outs[0][:3]] == ([0, 4, 0] if math == 0 else [0, 0, 0]) and outs[0][3] == (1 if math == 0 else 0)
        results[math] = [o[:8] for o in outs[0][:3]]
    assert results[0] == results[1] == results[2]
    assert results[0][0] == results[0][2] and results[0][0] != results[0][1]
    with torch.cuda.stream(stream):
        for f, got in zip((clean, bad), results[0][:2]):
            one = _single(rsdsfm, torch, dict(d, flow_img=f, gamma=gamma), stream, **kw)
            assert (one["n"], one["num_inliers"], one["best_trial"], one["ransac_v"].tobytes()) == got[:4]
            assert np.allclose(np.frombuffer(got[4]), one["v"], rtol=1e-9, atol=1e-14) and np.allclose(np.frombuffer(got[5]), one["w"], rtol=1e-9, atol=1e-14)


@pytest.mark.parametrize("cfg,accel", [(3, False), (5, True)])
def test_native_tiled_solve_matches_single_context(rsdsfm, cfg, accel):
    import torch

    stream = torch.cuda.Stream(torch.device("cuda", 0))
    d = rsdsfm.synth.make_config(cfg, rows=96, cols=250)
    rows, cols = d["rows"], d["cols"]
    kw = dict(trials=14, tol=0.002 if cfg == 3 else 0.01, seed=7, use_acceleration_mode=accel)
    with torch.cuda.stream(stream):
        one = _single(rsdsfm, torch, d, stream, **kw)
    assert one["num_inliers"] > 0.1 * rows * cols
    for nranks in (1, 2, 3, 5):  # 250 columns over 3 ranks: stride 84, the last slab is narrower (padded gather)
        til = _native_threads(rsdsfm, torch, d, nranks, **kw)
        assert sum(til["shard_n"]) == one["n"]
        _compare(til, one, rows, cols, depth_rtol=1e-6 if accel else 1e-9)
        assert len(til["inliers"]) == one["num_inliers"] and len(til["ys"]) == one["num_inliers"]
        # the driver's host synchronisations do not grow with RANSAC rounds or LM iterations: counts, RANSAC, refinement polls, tail
        assert max(i["host_syncs"] for i in til["infos"]) <= 3 + -(-2 * one["refine_summary"]["num_iterations"] // 5) + 1
        # one exchange per LM iteration: a slot in front of the first iteration, one per iteration, one more behind every iteration whose
        # speculated Schur sums did not apply (at most every iteration: never more than the two exchanges per iteration of the staged protocol)
        slots = [(i["path_flags"] >> 8) & 0xFFFF for i in til["infos"]]
        iters = one["refine_summary"]["num_iterations"]
        assert len(set(slots)) == 1 and iters <= slots[0] <= 2 * iters, (slots, iters)


rsdsfm_trace_accepted = 1.0  # RSDSFM_TRACE_ACCEPTED
rsdsfm_trace_invalid = 2.0  # RSDSFM_TRACE_INVALID


def _predicted_slots(trace, np_params, radius_factorised=True):
    """Slots the refinement consumes by the documented rule.
    Radius-factorised path (the default; DESIGN section 5b, refine_rf_kernels.hip): ONE in front (iteration zero + the Schur sums of iteration
    1) and one per LM iteration that evaluated a candidate -- whatever the step's quality; an iteration whose reduced system did not factor
    (outcome INVALID with no model change recorded) is solved again at half the radius inside the same stage: no slot.
    Iterate-by-iterate slot kernels (rsdsfm_set_refine_arithmetic(1); refine_kernels.hip): one in front (the Schur pass of iteration 1), one per
    LM iteration (its back-substitution + decision), and one plain Schur slot behind every iteration the solve outlives unless that iteration's
    step was accepted with exactly the radius the pass speculated on (radius_accept(radius, 1): x 3, capped) WHILE the pass was speculating -- it
    is while fewer than two decisions in a row failed that test, and never with k refined."""
    rows = trace[~np.isnan(trace[:, 0])]
    if radius_factorised:
        return 1 + int(sum(1 for row in rows if not (row[7] == rsdsfm_trace_invalid and row[3] == 0.0)))
    slots, miss_run = 1, 0
    for i, row in enumerate(rows):
        slots += 1
        if i == len(rows) - 1:
            break  # (the callers only pass solves that ended inside a decision)
        r_spec = min(row[5] / (1.0 / 3.0), 1e16)
        applies = row[7] == rsdsfm_trace_accepted and rows[i + 1][5] == r_spec  # accepted, and the next iteration starts from the speculated radius
        spec_on = np_params == 6 and miss_run < 2
        miss_run = 0 if applies else min(miss_run + 1, 2)
        if not (applies and spec_on):
            slots += 1
    return slots


@pytest.mark.parametrize("cfg,accel,tol", [(5, False, 0.05), (3, False, 0.002), (3, False, 0.01), (5, True, 0.01)])
def test_native_tiled_refinement_consumes_the_documented_number_of_slots(rsdsfm, cfg, accel, tol):
    """the exchanges of the tiled refinement (rsdsfm_tiled_info::path_flags bits 8-23) against the rule applied to the iteration trace of the
    single-context solve of the same frame -- default path: ONE exchange per LM iteration whatever the step's quality (+ the first pass);
    iterate-by-iterate slot kernels: one where the speculation applies, two where it does not"""
    import torch

    dev = torch.device("cuda", 0)
    d = rsdsfm.synth.make_config(cfg, rows=120, cols=208)
    rows, cols = d["rows"], d["cols"]
    img = torch.from_numpy(d["flow_img"]).to(dev)
    dm = torch.empty((cols, rows), dtype=torch.float64, device=dev)
    checked = 0
    for seed in (3, 11, 29):
        for refine_arithmetic in (0, 1):
            kw = dict(trials=12, tol=tol, seed=seed, use_acceleration_mode=accel)
            with rsdsfm.Solver(0) as s:
                s.set_refine_arithmetic(refine_arithmetic)
                s.set_refine_trace(60)
                r0 = s.refine_restarts()["restarts"]
                one = s.solve_frame_dev(img.data_ptr(), rows, cols, d["K"], d["gamma"], dm.data_ptr(), flow_index_mode=rsdsfm.FLOW_GATHERED, **kw)
                trace = s.get_refine_trace()
                restarted = s.refine_restarts()["restarts"] != r0
            if rsdsfm.TERMINATION[one["refine_summary"]["termination"]] not in ("gradient", "parameter", "function") or restarted:
                continue  # (max iterations / minimum radius are found at the top of the loop, in a slot of their own; a guard: two runs)
            want = _predicted_slots(trace, 7 if accel else 6, radius_factorised=refine_arithmetic == 0)
            for nranks in (1, 3):
                til = _native_threads(rsdsfm, torch, d, nranks, refine_arithmetic=refine_arithmetic, **kw)
                assert til["refine_summary"]["num_iterations"] == one["refine_summary"]["num_iterations"]
                slots = {(i["path_flags"] >> 8) & 0xFFFF for i in til["infos"]}
                assert slots == {want}, (seed, nranks, refine_arithmetic, slots, want, trace[~np.isnan(trace[:, 0])][:, [3, 4, 5, 7]])
            checked += 1
    assert checked >= 4


def test_native_tiled_modes(rsdsfm):
    """closed-form depth mode, no refinement, more trials than one hypothesis batch (> 128), empty trailing slabs, global
    shutter mode, too few points"""
    import torch

    stream = torch.cuda.Stream(torch.device("cuda", 0))
    d = rsdsfm.synth.make_config(3, rows=64, cols=90)
    rows, cols = d["rows"], d["cols"]
    with torch.cuda.stream(stream):
        for kw in (dict(trials=20, tol=0.002, seed=3, use_refinement=False, depth_mode=0), dict(trials=150, tol=0.002, seed=3, use_refinement=False),
                   dict(trials=10, tol=0.004, seed=5, use_global_shutter_mode=True)):
            one = _single(rsdsfm, torch, d, stream, **kw)
            til = _native_threads(rsdsfm, torch, d, 3, **kw)
            assert til["n"] == one["n"] and til["num_inliers"] == one["num_inliers"] and til["best_trial"] == one["best_trial"]
            if not kw.get("use_refinement", True):
                assert np.array_equal(til["v"], one["v"]) and np.array_equal(til["depth_map"], one["depth_map"])  # no global float sums involved
            else:
                _compare(til, one, rows, cols)
        d1 = rsdsfm.synth.make_config(1, rows=40, cols=7)
        kw = dict(trials=6, tol=0.01, seed=1)
        one = _single(rsdsfm, torch, d1, stream, **kw)
        til = _native_threads(rsdsfm, torch, d1, 4, **kw)  # stride 2 -> slabs (0,2) (2,4) (4,6) (6,7)
        _compare(til, one, d1["rows"], d1["cols"])
        til = _native_threads(rsdsfm, torch, d1, 5, **kw)  # stride 2 -> the fifth slab is empty
        _compare(til, one, d1["rows"], d1["cols"])
        d2 = dict(d1)
        img = np.zeros_like(d1["flow_img"])
        img[:2, :3] = d1["flow_img"][:2, :3]  # 6 points < 9: refused like the single-context solve
        d2["flow_img"] = img
        with pytest.raises(rsdsfm.RsdsfmError):
            _native_threads(rsdsfm, torch, d2, 2, **kw)


def test_native_tiled_over_rccl_one_rank(rsdsfm):
    """the RCCL path on the one GPU of the box: unique id -> ncclCommInitRank (1 rank) -> ncclAllGather / ncclAllReduce on the
    context's stream; result = the single-context solve bit for bit (one slab: the same sums in the same order)"""
    import torch

    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    d = rsdsfm.synth.make_config(3, rows=120, cols=200)
    rows, cols, K, gamma = d["rows"], d["cols"], d["K"], d["gamma"]
    kw = dict(trials=12, tol=0.002, seed=9)
    with torch.cuda.stream(stream):
        one = _single(rsdsfm, torch, d, stream, **kw)
        img = torch.from_numpy(d["flow_img"]).to(dev)
        dm = torch.zeros(cols * rows, dtype=torch.float64, device=dev)
        R = torch.empty(rows * 9, dtype=torch.float64, device=dev)
        t = torch.empty(rows * 3, dtype=torch.float64, device=dev)
        with rsdsfm.Solver(0, stream=stream.cuda_stream) as s:
            s.dist_init(1, 0, rsdsfm.dist_unique_id())
            for _ in range(2):  # the communicator is reused across solves
                r = s.solve_frame_tiled_dev(img.data_ptr(), rows, cols, K, gamma, dm.data_ptr(), R.data_ptr(), t.data_ptr(), flow_index_mode=rsdsfm.FLOW_GATHERED, **kw)
            s.synchronize()
            assert r["info"]["collectives"] >= 8 and r["info"]["nranks"] == 1
            s.dist_finalize()
    assert r["n"] == one["n"] and r["num_inliers"] == one["num_inliers"] and r["best_trial"] == one["best_trial"]
    assert np.array_equal(r["v"], one["v"]) and np.array_equal(r["w"], one["w"]) and r["k"] == one["k"]
    assert r["refine_summary"] == one["refine_summary"]
    assert np.array_equal(dm.cpu().numpy(), one["depth_map"])
    assert np.array_equal(R.cpu().numpy().reshape(rows, 9), one["R"]) and np.array_equal(t.cpu().numpy().reshape(rows, 3), one["t"])


def test_native_tiled_3840x2160_in_8_ranks(rsdsfm, oracle_chain, big_config):
    """BASELINE configs[3] at full size through the native driver: 8 logical ranks (threads) on the one GPU, against the un-tiled
    solve and the oracle chain (5 trials): integers exact, floats 1e-9 / 1e-6"""
    import torch

    stream = torch.cuda.Stream(torch.device("cuda", 0))
    T, tol, seed = 5, 0.002, 5
    d = big_config(4)
    rows, cols = d["rows"], d["cols"]
    kw = dict(trials=T, tol=tol, seed=seed)
    with torch.cuda.stream(stream):
        one = _single(rsdsfm, torch, d, stream, **kw)
    til = _native_threads(rsdsfm, torch, d, 8, **kw)
    assert len(til["shard_n"]) == 8 and sum(til["shard_n"]) == one["n"] == rows * cols
    _compare(til, one, rows, cols)
    o = oracle_chain(4, T, tol, seed)
    ro, refo = o["ransac"], o["refine"]
    assert til["num_inliers"] == ro["num_inliers"] and til["best_trial"] == ro["best_trial"]
    for key in ("num_iterations", "num_successful_steps", "termination"):
        assert til["refine_summary"][key] == refo["summary"][key], key
    assert til["flipped"] == o["flipped"] and np.allclose(til["v"], o["v"], rtol=1e-6, atol=1e-10) and np.allclose(til["w"], refo["w"], rtol=1e-6, atol=1e-10)
    assert np.array_equal(til["ys"], o["ys"])  # scanline index of every inlier: bit-exact
    got = til["depth_map"].reshape(cols, rows).T
    assert np.array_equal(got != 0, o["depth_map"] != 0) and np.allclose(got, o["depth_map"], rtol=1e-6)


# ---------------------------------------------------------------------------------------------------
# the reference's DEFAULT flow indexing across slabs (quirk Q2: main.cc:457 passes the un-compacted flow, nonlinearRefinement.cc:
# 209-212 reads column i for the i-th inlier): a zero-initialised rsdsfm_frame_params must work tiled, also when M < N
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg,accel", [(3, False), (5, False)])  # (k free on this mismatched problem wanders for ~50 iterations: chaotic)
def test_native_tiled_rank_indexed_flow_matches_single_context(rsdsfm, oracle, cfg, accel):
    """selective tolerance (M < N): inlier ranks and point indices differ, a slab's inliers read flow columns that live on the
    slabs in front of it.  1 / 2 / 3 / 5 logical ranks = the single-context default solve (integers exact, floats to the summation
    order), and the single-context solve = the oracle chain in mode 0"""
    import torch

    stream = torch.cuda.Stream(torch.device("cuda", 0))
    d = rsdsfm.synth.make_config(cfg, rows=96, cols=250)
    rows, cols, K, gamma = d["rows"], d["cols"], d["K"], d["gamma"]
    kw = dict(trials=14, tol=0.002 if cfg == 3 else 0.004, seed=7, use_acceleration_mode=accel, flow_index_mode=rsdsfm.FLOW_COMPAT_RANK)
    with torch.cuda.stream(stream):
        one = _single(rsdsfm, torch, d, stream, **kw)
        gathered = _single(rsdsfm, torch, d, stream, **dict(kw, flow_index_mode=rsdsfm.FLOW_GATHERED))
    assert 0.1 * rows * cols < one["num_inliers"] < 0.97 * rows * cols  # selective: the two indexings are different problems
    assert not np.allclose(one["v"], gathered["v"], rtol=1e-7)
    # oracle chain, rank-indexed
    q, u, qpx, fpx = oracle.flatten(d["flow_img"], *K, gamma)
    a, ak = oracle.get_alpha(fpx, rows, gamma), oracle.get_alpha_k(qpx, fpx, rows, gamma)
    ro = oracle.ransac(q, u, a, ak, accel, kw["trials"], kw["tol"], oracle.sample_indices(len(q), kw["trials"], kw["seed"]), depth_mode=1)
    refo = oracle.refine(u, ro["inliers"], ro["alpha"], ro["alpha_k"], ro["v"], ro["w"], ro["k"], accel, 0, None)
    _, v_o, _ = oracle.canonicalize_sign(refo["inliers"], refo["v"])
    assert one["num_inliers"] == ro["num_inliers"] and one["refine_summary"]["num_iterations"] == refo["summary"]["num_iterations"]
    assert np.allclose(one["v"], v_o, rtol=1e-6, atol=1e-10) and np.allclose(one["w"], refo["w"], rtol=1e-6, atol=1e-10)
    for nranks in (1, 2, 3, 5):
        til = _native_threads(rsdsfm, torch, d, nranks, **kw)
        _compare(til, one, rows, cols, depth_rtol=1e-6 if accel else 1e-9)


def test_native_tiled_rank_indexed_flow_without_remote_columns(rsdsfm):
    """every point an inlier (as main.cc:310's tolerance 0.05 makes them on the 1280x720 bench data; here: model flow): rank ==
    index, every column is local and the exchange is skipped -- the same number of collectives as the gathered solve of what is then
    the same problem; the zero-initialised parameter struct (flow_index_mode 0) is accepted"""
    import torch

    stream = torch.cuda.Stream(torch.device("cuda", 0))
    d = rsdsfm.synth.make_config(2, rows=96, cols=250)
    rows, cols = d["rows"], d["cols"]
    kw = dict(trials=10, tol=0.05, seed=3, flow_index_mode=rsdsfm.FLOW_COMPAT_RANK)
    with torch.cuda.stream(stream):
        one = _single(rsdsfm, torch, d, stream, **kw)
    assert one["num_inliers"] == one["n"] == rows * cols
    til = _native_threads(rsdsfm, torch, d, 3, **kw)
    base = _native_threads(rsdsfm, torch, d, 3, **dict(kw, flow_index_mode=rsdsfm.FLOW_GATHERED))
    _compare(til, one, rows, cols)
    _compare(base, one, rows, cols)
    assert [i["collectives"] for i in til["infos"]] == [i["collectives"] for i in base["infos"]]
    # pixels without flow are dropped by the flatten, not by the RANSAC: every remaining POINT is still an inlier (rank == index)
    for c_lo in (200, 10):
        d2 = dict(d)
        img = d["flow_img"].copy()
        img[20:50, c_lo:c_lo + 30] = 0.0  # pixels without flow are dropped by the flatten
        d2["flow_img"] = img
        with torch.cuda.stream(stream):
            one2 = _single(rsdsfm, torch, d2, stream, **kw)
        assert one2["num_inliers"] == one2["n"] == rows * cols - 30 * 30
        til2 = _native_threads(rsdsfm, torch, d2, 3, **kw)
        _compare(til2, one2, rows, cols)


def test_native_tiled_rank_indexed_flow_3840x2160_in_8_ranks(rsdsfm, oracle_chain, big_config):
    """BASELINE configs[3] at full size with the reference's default flow indexing and a selective tolerance: 8 logical ranks = the
    single-context default solve = the oracle chain in mode 0 (integers exact incl. every inlier's scanline index)"""
    import torch

    stream = torch.cuda.Stream(torch.device("cuda", 0))
    T, tol, seed = 5, 0.002, 5
    d = big_config(4)
    rows, cols = d["rows"], d["cols"]
    kw = dict(trials=T, tol=tol, seed=seed, flow_index_mode=rsdsfm.FLOW_COMPAT_RANK)
    with torch.cuda.stream(stream):
        one = _single(rsdsfm, torch, d, stream, **kw)
    assert one["num_inliers"] < 0.97 * one["n"]
    til = _native_threads(rsdsfm, torch, d, 8, **kw)
    _compare(til, one, rows, cols)
    o = oracle_chain(4, T, tol, seed, flow_mode=0)
    ro, refo = o["ransac"], o["refine"]
    assert til["num_inliers"] == ro["num_inliers"] and til["best_trial"] == ro["best_trial"]
    for key in ("num_iterations", "num_successful_steps", "termination"):
        assert til["refine_summary"][key] == refo["summary"][key], key
    assert til["flipped"] == o["flipped"] and np.allclose(til["v"], o["v"], rtol=1e-6, atol=1e-10) and np.allclose(til["w"], refo["w"], rtol=1e-6, atol=1e-10)
    assert np.array_equal(til["ys"], o["ys"])
    got = til["depth_map"].reshape(cols, rows).T
    assert np.array_equal(got != 0, o["depth_map"] != 0) and np.allclose(got, o["depth_map"], rtol=1e-6)


def test_native_tiled_setup_failure_on_one_rank_ends_every_rank(rsdsfm):
    """a failure only one rank sees (here: a null slab pointer) travels with the point counts in the first exchange: that rank
    reports its own error, the others return RSDSFM_ERR_PEER instead of waiting in the next collective for ever; the contexts stay
    usable"""
    import torch
    from transports import ThreadTransport

    dev = torch.device("cuda", 0)
    d = rsdsfm.synth.make_config(3, rows=64, cols=90)
    rows, cols, K, gamma = d["rows"], d["cols"], d["K"], d["gamma"]
    img = torch.from_numpy(d["flow_img"]).to(dev)
    nranks = 3
    tr = ThreadTransport(nranks, timeout=60.0)
    errs, oks = [None] * nranks, [None] * nranks

    def work(rank):
        torch.cuda.set_device(0)
        c0, sc, per = rsdsfm.tiled_slab_bounds(cols, nranks, rank)
        slab = img[:, c0:c0 + sc, :].contiguous()
        dm = torch.zeros(cols * rows, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        with rsdsfm.Solver(0) as s:
            s.dist_set_transport(nranks, rank, *tr.callbacks(rank))
            try:
                s.solve_frame_tiled_dev(0 if rank == 1 else slab.data_ptr(), rows, cols, K, gamma, dm.data_ptr(), trials=6, tol=0.01, seed=2)
            except rsdsfm.RsdsfmError as e:
                errs[rank] = str(e)
            r = s.solve_frame_tiled_dev(slab.data_ptr(), rows, cols, K, gamma, dm.data_ptr(), trials=6, tol=0.01, seed=2)  # all ranks fine now
            s.synchronize()
            oks[rank] = (r["num_inliers"], r["v"].tobytes())

    ths = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    for th in ths:
        th.start()
    for th in ths:
        th.join(timeout=120)
    assert not any(th.is_alive() for th in ths), "a rank is still waiting in a collective"
    assert errs[1] is not None and "(-1)" in errs[1] and "null slab pointer" in errs[1]
    for r in (0, 2):
        assert errs[r] is not None and "(-6)" in errs[r] and "rank 1 failed" in errs[r], errs[r]
    assert oks[0] is not None and oks[0] == oks[1] == oks[2]


@pytest.mark.parametrize("flow_mode", [1, 0])
def test_native_tiled_two_processes_share_the_gpu(rsdsfm, tmp_path, flow_mode):
    """two ranks = two processes (gloo rendezvous on 127.0.0.1, both on cuda:0, collectives through GlooTransport) run
    tests/mp_tiled_native.py; rank 0's result equals the single-context solve and both ranks agree bit for bit.  flow_mode 0 = the
    reference's rank-indexed flow (rank 1 fetches flow columns of rank 0's slab)"""
    import torch

    out = tmp_path / "res.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", RSDSFM_TILED_OUT=str(out), RSDSFM_TEST_FLOW_MODE=str(flow_mode))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(29641 + flow_mode),
           os.path.join(ROOT, "tests", "mp_tiled_native.py")]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    got = json.loads(out.read_text())
    stream = torch.cuda.Stream(torch.device("cuda", 0))
    with torch.cuda.stream(stream):
        d = rsdsfm.synth.make_config(3, rows=96, cols=250)
        one = _single(rsdsfm, torch, d, stream, trials=14, tol=0.002, seed=7, flow_index_mode=flow_mode)
    assert got["world"] == 2 and got["ranks_agree"] and got["info"]["nranks"] == 2
    assert got["n"] == one["n"] and got["num_inliers"] == one["num_inliers"] and got["best_trial"] == one["best_trial"]
    assert np.allclose(got["v"], one["v"], rtol=1e-9) and np.allclose(got["w"], one["w"], rtol=1e-9)
    assert got["depth_nonzero"] == int((one["depth_map"] != 0).sum()) and np.isclose(got["depth_sum"], one["depth_map"].sum(), rtol=1e-9)


@pytest.mark.parametrize("flow_mode", [0, 1])
def test_native_tiled_over_a_real_two_rank_rccl_communicator(rsdsfm, tmp_path, flow_mode):
    """TWO ranks over RCCL itself on the one GPU of the box (tests/mp_tiled_rccl.py): RCCL identifies a host by NCCL_HOSTID, so one id
    per rank makes it treat the ranks as two single-GPU nodes joined by its socket transport over the loopback interface -- not
    xGMI, but the multi-rank code path for real: ncclCommInitRank(2 ranks) from the broadcast unique id, in-place ncclAllGather
    (ncclChar) / ncclAllReduce on the context's stream, the rank-ordered protocol of the driver (incl. the rank-indexed flow exchange in
    flow_mode 0 and the row-tiled depth solve).  Rank 0's result equals the single-context solve; both ranks agree bit for bit."""
    import torch

    out = tmp_path / "res.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", RSDSFM_TILED_OUT=str(out), RSDSFM_TEST_FLOW_MODE=str(flow_mode), NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1")
    env.pop("NCCL_HOSTID", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(29651 + flow_mode),
           os.path.join(ROOT, "tests", "mp_tiled_rccl.py")]
    try:
        p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    except subprocess.TimeoutExpired:  # (the trick depends on the box's loopback networking: an environment limit, not a product failure)
        pytest.skip("two RCCL ranks over the loopback interface did not connect within 300 s on this box")
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    got = json.loads(out.read_text())
    if got["init"] != "ok":
        pytest.skip("this box's RCCL cannot connect two ranks over the loopback interface: " + got["init"][:300])
    stream = torch.cuda.Stream(torch.device("cuda", 0))
    d = rsdsfm.synth.make_config(3, rows=96, cols=250)
    with torch.cuda.stream(stream):
        one = _single(rsdsfm, torch, d, stream, trials=14, tol=0.002, seed=7, flow_index_mode=flow_mode)
    assert got["world"] == 2 and got["ranks_agree"] and got["info"]["nranks"] == 2
    assert got["n"] == one["n"] and got["num_inliers"] == one["num_inliers"] and got["best_trial"] == one["best_trial"]
    assert got["iterations"] == one["refine_summary"]["num_iterations"]
    assert np.allclose(got["v"], one["v"], rtol=1e-9) and np.allclose(got["w"], one["w"], rtol=1e-9)
    assert got["depth_nonzero"] == int((one["depth_map"] != 0).sum()) and np.isclose(got["depth_sum"], one["depth_map"].sum(), rtol=1e-9)
    # the row-tiled dense depth solve over the same communicator = the single-context solve
    t = d["truth"]
    rho, sm = _depth_single(rsdsfm, torch, d, t["v"] / np.linalg.norm(t["v"]), t["w"], 0.0, 1)
    assert got["depth_lm"]["num_iterations"] == sm["num_iterations"] and got["depth_lm"]["termination"] == sm["termination"]
    assert np.isclose(got["depth_sum_tiled"], rho.sum(), rtol=1e-12) and got["depth_info"]["nranks"] == 2


@pytest.mark.parametrize("lm_arithmetic", [1, 0])
def test_native_tiled_paths_over_a_real_two_rank_rccl_communicator(rsdsfm, tmp_path, lm_arithmetic):
    """The driver's speculative paths over RCCL itself (two processes, tests/mp_tiled_rccl.py with RSDSFM_TEST_SEQUENCE): cold, ahead on the
    dense counts, a frame that is not dense after all (restart through the counts exchange), a function-core miss in ONE rank's slab (every
    rank restarts the RANSAC).  Each path issues another sequence of collectives; the two processes only finish if they pair up.  Every
    solve equals the single-context solve of its frame, both ranks report the same path."""
    import torch

    out = tmp_path / "seq.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", RSDSFM_TILED_OUT=str(out), RSDSFM_TEST_FLOW_MODE="1", RSDSFM_TEST_SEQUENCE="1", NCCL_SOCKET_IFNAME="lo",
               NCCL_IB_DISABLE="1", RSDSFM_TEST_LM_ARITHMETIC=str(lm_arithmetic))
    env.pop("NCCL_HOSTID", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29657",
           os.path.join(ROOT, "tests", "mp_tiled_rccl.py")]
    try:
        p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    except subprocess.TimeoutExpired:  # (the trick depends on the box's loopback networking: an environment limit, not a product failure)
        pytest.skip("two RCCL ranks over the loopback interface did not connect within 300 s on this box")
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    got = json.loads(out.read_text())
    if got["init"] != "ok":
        pytest.skip("this box's RCCL cannot connect two ranks over the loopback interface: " + got["init"][:300])
    seq = got["sequence"]
    assert len(seq) == 2 and len(seq[0]) == 7 and seq[0] == seq[1]  # both ranks: same results, same paths, same numbers of collectives
    # cold / ahead / ahead but the frame has a hole: started over / cold (the previous frame was not dense) / ahead / ahead, and the RANSAC
    # started over with the standard functions / ahead (standard functions: no restart)
    # (lm_arithmetic 0, the default: the analytic pass takes the NaN pixel as the reference takes it -- no cores to leave, no restart)
    assert [s["path_flags"] & 0xEF for s in seq[0]] == ([0, 1, 3, 0, 1, 5, 1] if lm_arithmetic == 1 else [0, 1, 3, 0, 1, 1, 1])  # (bit 4: see the warm-path test)
    assert got["restarts"] == (1 if lm_arithmetic == 1 else 0) and got["lma_restarts"] == 0
    stream = torch.cuda.Stream(torch.device("cuda", 0))
    d = rsdsfm.synth.make_config(5, rows=64, cols=480)
    clean = np.array(d["flow_img"])
    holed = clean.copy()
    holed[10:30, 300:340] = 0.0
    bad = clean.copy()
    bad[17, 301] = (3.0, 1e160)
    ones = {}
    with torch.cuda.stream(stream):
        for name, f in (("clean", clean), ("holed", holed), ("bad", bad)):
            ones[name] = _single(rsdsfm, torch, dict(d, flow_img=f, gamma=0.5), stream, trials=20, tol=0.05, seed=11, flow_index_mode=1)
    assert ones["holed"]["n"] < ones["clean"]["n"]
    for s, name in zip(seq[0], ["clean", "clean", "holed", "clean", "clean", "bad", "clean"]):
        one = ones[name]
        assert (s["n"], s["num_inliers"], s["best_trial"], s["iterations"]) == (one["n"], one["num_inliers"], one["best_trial"], one["refine_summary"]["num_iterations"]), name
        assert np.allclose(s["v"], one["v"], rtol=1e-9, atol=1e-14) and np.allclose(s["w"], one["w"], rtol=1e-9, atol=1e-14), name
        assert s["depth_nonzero"] == int((one["depth_map"] != 0).sum()) and np.isclose(s["depth_sum"], one["depth_map"].sum(), rtol=1e-9), name


def test_native_tiled_random_sequences_over_a_real_two_rank_rccl_communicator(rsdsfm, tmp_path):
    """40 random cases of tests/fuzz_tiled.py (frame size, data kind, tolerance, trials, flow mode, acceleration mode, sequences of clean /
    holed / poisoned frames) over ONE real two-rank RCCL communicator (two processes): the ranks agree bit for bit on every solve and rank 0's
    results equal the single-context solves"""
    out = tmp_path / "fuzz.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", RSDSFM_TILED_OUT=str(out), RSDSFM_TEST_FUZZ="40", NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1")
    env.pop("NCCL_HOSTID", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29659",
           os.path.join(ROOT, "tests", "mp_tiled_rccl.py")]
    try:
        p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=420)
    except subprocess.TimeoutExpired:  # (the trick depends on the box's loopback networking: an environment limit, not a product failure)
        pytest.skip("two RCCL ranks over the loopback interface did not finish within 420 s on this box")
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    got = json.loads(out.read_text())
    if got["init"] != "ok":
        pytest.skip("this box's RCCL cannot connect two ranks over the loopback interface: " + got["init"][:300])
    assert got["fuzz"]["solves"] >= 80 and got["fuzz"]["bad"] == [], got["fuzz"]
    assert len(got["fuzz"]["paths"]) >= 3, got["fuzz"]["paths"]  # cold, ahead, started over ...


# ---------------------------------------------------------------------------------------------------
# the row-tiled DENSE DEPTH solve driven from C++ (rsdsfm_estimate_inverse_depths_tiled_dev)
# ---------------------------------------------------------------------------------------------------
def _depth_single(rsdsfm, torch, d, v, w, k, mode):
    dev = torch.device("cuda", 0)
    n = len(d["alpha"])
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    q, u, a, ak = tt(d["q"]), tt(d["u"]), tt(d["alpha"]), tt(d["alpha_k"])
    rho = torch.zeros(max(n, 2), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    with rsdsfm.Solver(0) as s:
        s.estimate_inverse_depths_dev(q.data_ptr(), u.data_ptr(), n, v, w, k, a.data_ptr(), ak.data_ptr(), rho.data_ptr(), mode=mode)
        sm = s.depth_finish_dev(q.data_ptr(), u.data_ptr(), n, v, w, k, a.data_ptr(), ak.data_ptr(), rho.data_ptr())[0] if mode == 1 else None
        s.synchronize()
    return rho.cpu().numpy()[:n], sm


def _depth_threads(rsdsfm, torch, d, v, w, k, nranks, mode, repeat=1):
    from transports import ThreadTransport

    dev = torch.device("cuda", 0)
    n = len(d["alpha"])
    tr = ThreadTransport(nranks, timeout=float(__import__("os").environ.get("RSDSFM_TEST_BARRIER_TIMEOUT", "120")))
    outs, errs = [None] * nranks, [None] * nranks

    def work(rank):
        try:
            torch.cuda.set_device(0)
            i0, cnt, per = rsdsfm.tiled_shard_bounds(n, nranks, rank)
            tt = lambda a: torch.from_numpy(np.ascontiguousarray(a[i0:i0 + cnt])).to(dev)
            q, u, a, ak = tt(d["q"]), tt(d["u"]), tt(d["alpha"]), tt(d["alpha_k"])
            full = torch.full((max(n, 2),), -7.0, dtype=torch.float64, device=dev)
            torch.cuda.synchronize()
            with rsdsfm.Solver(0) as s:
                s.dist_set_transport(nranks, rank, *tr.callbacks(rank))
                for _ in range(repeat):
                    sm, info = s.estimate_inverse_depths_tiled_dev(q.data_ptr() if cnt else 0, u.data_ptr() if cnt else 0, n, v, w, k,
                                                                   a.data_ptr() if cnt else 0, ak.data_ptr() if cnt else 0, full.data_ptr(), mode=mode)
                s.synchronize()
            outs[rank] = (full.cpu().numpy()[:n], sm, info)
        except Exception as e:  # noqa: BLE001
            errs[rank] = e
            tr.barrier.abort()

    ths = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    for e in errs:
        if e is not None:
            raise e
    for rho, sm, info in outs[1:]:  # every rank holds the same full vector and took the same decisions
        assert np.array_equal(rho, outs[0][0]) and sm == outs[0][1]
    assert sum(o[2]["shard_points"] for o in outs) == n
    return outs[0][0], outs[0][1], [o[2] for o in outs]


@pytest.mark.parametrize("cfg", [1, 3])
def test_native_tiled_depth_matches_single_context_and_oracle(oracle, rsdsfm, cfg):
    """1 / 2 / 5 logical ranks give the single-context solve's depths bit for bit (per-pixel solves; the global decisions come from
    rank-ordered sums) and the oracle's to 1e-9; the common case costs ONE host synchronisation"""
    import torch

    d = rsdsfm.synth.make_config(cfg, rows=150, cols=201)  # 30150 points: 5 ranks -> stride 6030, exact; 4 ranks -> padded
    t = d["truth"]
    v = t["v"] / np.linalg.norm(t["v"])
    w, k = t["w"], 0.0
    for mode in (0, 1):
        one, sm1 = _depth_single(rsdsfm, torch, d, v, w, k, mode)
        rho_o, sm_o = oracle.estimate_inverse_depths(d["q"], d["u"], v, w, k, d["alpha"], d["alpha_k"], mode=mode)
        assert np.allclose(one, rho_o, rtol=1e-9, atol=1e-13)
        for nranks in (1, 2, 4, 5):
            rho, sm, infos = _depth_threads(rsdsfm, torch, d, v, w, k, nranks, mode, repeat=2)
            assert np.array_equal(rho, one), (mode, nranks)
            assert all(i["nranks"] == nranks for i in infos)
            if mode == 1:
                for key in ("num_iterations", "num_successful_steps", "num_unsuccessful_steps", "termination"):
                    assert sm[key] == sm1[key] == sm_o[key], key
                assert np.isclose(sm["final_cost"], sm_o["final_cost"], rtol=1e-9)
                if sm["num_iterations"] <= 2:
                    assert all(i["host_syncs"] == 1 for i in infos)
                assert all(i["collectives"] == i["host_syncs"] + 1 for i in infos)  # one row all-gather per decision + the shards
            else:
                assert all(i["host_syncs"] == 0 and i["collectives"] == 1 for i in infos)


def test_native_tiled_depth_many_iterations_and_empty_shards(oracle, rsdsfm):
    """a pose far from the truth makes the emulated trust-region loop run several iterations (rejected steps included): the
    state machine continues across host polls on every rank alike; 7 points over 5 ranks leave the last rank empty"""
    import torch

    d = rsdsfm.synth.make_config(3, rows=90, cols=120)
    t = d["truth"]
    v = np.array([0.3, -0.2, 0.93])
    v /= np.linalg.norm(v)
    w, k = 3.0 * t["w"] + 0.01, 0.4
    one, sm1 = _depth_single(rsdsfm, torch, d, v, w, k, 1)
    rho_o, sm_o = oracle.estimate_inverse_depths(d["q"], d["u"], v, w, k, d["alpha"], d["alpha_k"], mode=1)
    rho, sm, infos = _depth_threads(rsdsfm, torch, d, v, w, k, 3, 1)
    assert np.array_equal(rho, one)
    for key in ("num_iterations", "num_successful_steps", "num_unsuccessful_steps", "termination"):
        assert sm[key] == sm1[key] == sm_o[key], key
    assert np.allclose(rho, rho_o, rtol=1e-8, atol=1e-12)
    small = {key: d[key][:7] for key in ("q", "u", "alpha", "alpha_k")}
    one, sm1 = _depth_single(rsdsfm, torch, small, v, w, k, 1)
    for nranks in (2, 5):  # stride 4 / 2: ranks (0,2) (2,4) (4,6) (6,7) () -- the fifth shard is empty
        rho, sm, infos = _depth_threads(rsdsfm, torch, small, v, w, k, nranks, 1)
        assert np.array_equal(rho, one) and sm["num_iterations"] == sm1["num_iterations"]
    assert infos[-1]["shard_points"] == 0
    empty = {key: d[key][:0] for key in ("q", "u", "alpha", "alpha_k")}
    rho, sm, infos = _depth_threads(rsdsfm, torch, empty, v, w, k, 2, 0)
    assert rho.shape == (0,)


def test_native_tiled_depth_over_rccl_one_rank(rsdsfm):
    """the RCCL path (1-rank communicator) of the depth solve: same bits as the single-context solve"""
    import torch

    dev = torch.device("cuda", 0)
    d = rsdsfm.synth.make_config(3, rows=120, cols=200)
    t = d["truth"]
    v = t["v"] / np.linalg.norm(t["v"])
    w, k, n = t["w"], 0.0, len(d["alpha"])
    one, sm1 = _depth_single(rsdsfm, torch, d, v, w, k, 1)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    q, u, a, ak = tt(d["q"]), tt(d["u"]), tt(d["alpha"]), tt(d["alpha_k"])
    full = torch.zeros(n, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    with rsdsfm.Solver(0) as s:
        s.dist_init(1, 0, rsdsfm.dist_unique_id())
        for _ in range(2):
            sm, info = s.estimate_inverse_depths_tiled_dev(q.data_ptr(), u.data_ptr(), n, v, w, k, a.data_ptr(), ak.data_ptr(), full.data_ptr())
        s.synchronize()
        s.dist_finalize()
    assert np.array_equal(full.cpu().numpy(), one) and info["collectives"] >= 2
    for key in ("num_iterations", "num_successful_steps", "num_unsuccessful_steps", "termination", "final_radius"):
        assert sm[key] == sm1[key], key
    # the shard's workgroup partials are first reduced to ONE row: the cost sums differ from the single-context order in the last ulp
    assert np.isclose(sm["final_cost"], sm1["final_cost"], rtol=1e-13) and np.isclose(sm["initial_cost"], sm1["initial_cost"], rtol=1e-13)
