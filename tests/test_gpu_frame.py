"""GPU: the one-call frame solve (rsdsfm_solve_frame_dev) equals the stage-by-stage path and the oracle chain."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("flow_mode", [0, 1])
def test_solve_frame_dev_matches_stages_and_oracle(oracle, rsdsfm, flow_mode):
    """flow_mode 0 = RSDSFM_FLOW_COMPAT_RANK, the default of rsdsfm_frame_params and what evaluateSingleRun does (main.cc:457:
    the refinement reads flow column i for the i-th inlier, quirk Q2); 1 = RSDSFM_FLOW_GATHERED (the inlier's own pixel)"""
    import torch

    dev = torch.device("cuda", 0)
    d = rsdsfm.synth.make_config(3, rows=144, cols=256)
    rows, cols, K, gamma = d["rows"], d["cols"], d["K"], d["gamma"]
    img = torch.from_numpy(d["flow_img"]).to(dev)
    dm = torch.empty((cols, rows), dtype=torch.float64, device=dev)
    R = torch.empty((rows, 9), dtype=torch.float64, device=dev)
    t = torch.empty((rows, 3), dtype=torch.float64, device=dev)
    T, tol, seed = 12, 0.002, 99
    with rsdsfm.Solver(0) as s:
        kw = {} if flow_mode == 0 else {"flow_index_mode": rsdsfm.FLOW_GATHERED}  # 0 is the default
        r = s.solve_frame_dev(img.data_ptr(), rows, cols, K, gamma, dm.data_ptr(), R.data_ptr(), t.data_ptr(), trials=T, tol=tol, seed=seed, **kw)
        s.synchronize()
        # stage by stage (host-pointer API)
        q, u, a, ak = s.flatten(d["flow_img"], K, gamma)
        rr = s.ransac(q, u, a, ak, False, T, tol, samples=None, seed=seed, depth_mode=1)
        ref = s.non_linear_refinement(u, rr["inliers"], rr["alpha"], rr["alpha_k"], rr["v"], rr["w"], rr["k"], False, flow_index_mode=flow_mode,
                                      inlier_idx=rr["inlier_idx"] if flow_mode else None)
        dmap = s.depth_map(ref["inliers"], ref["v"], K, rows, cols)
        Rr, tr = s.pose_table(dmap["v"], ref["w"], ref["k"], gamma, rows)
    assert r["n"] == len(q) and r["num_inliers"] == rr["num_inliers"] and r["best_trial"] == rr["best_trial"]
    assert np.array_equal(r["ransac_w"], rr["w"]) and np.array_equal(r["ransac_v"], rr["v"])
    assert np.array_equal(r["w"], ref["w"]) and np.array_equal(r["v"], dmap["v"]) and r["k"] == ref["k"]
    assert r["refine_summary"] == ref["summary"]
    assert np.array_equal(dm.cpu().numpy().T, dmap["depth_map"])
    assert np.array_equal(R.cpu().numpy().reshape(rows, 3, 3), Rr) and np.array_equal(t.cpu().numpy(), tr)
    # oracle chain on the same sampler / seed
    ro = oracle.ransac(q, u, a, ak, False, T, tol, oracle.sample_indices(len(q), T, seed), depth_mode=1)
    assert ro["num_inliers"] == r["num_inliers"] and ro["best_trial"] == r["best_trial"]
    assert 0 < ro["num_inliers"] < len(q)  # selective tolerance: rank and pixel index differ, the two modes are different problems
    refo = oracle.refine(u, ro["inliers"], ro["alpha"], ro["alpha_k"], ro["v"], ro["w"], ro["k"], False, flow_mode, ro["inlier_idx"] if flow_mode else None)
    for key in ("num_iterations", "num_successful_steps", "termination"):
        assert r["refine_summary"][key] == refo["summary"][key], key
    inl_o, v_o, _ = oracle.canonicalize_sign(refo["inliers"], refo["v"])
    assert np.allclose(r["v"], v_o, rtol=1e-6, atol=1e-10) and np.allclose(r["w"], refo["w"], rtol=1e-6, atol=1e-10)
    dm_o, _, ys_o = oracle.scatter_depth(inl_o, *K, rows, cols)
    got = dm.cpu().numpy().T
    assert np.array_equal(got != 0, dm_o != 0)
    if flow_mode == 1:
        assert np.allclose(got, dm_o, rtol=1e-6)
    else:
        # rank-indexed flow with a strict inlier subset pairs points with other pixels' flow (the reference's quirk Q2): a
        # well-defined but physically meaningless problem whose optimum has points with 1/depth ~ 0 -- compare the solver's
        # own variable (1/depth) with an absolute floor instead of the reciprocal
        m = dm_o != 0
        assert np.allclose(1.0 / got[m], 1.0 / dm_o[m], rtol=1e-6, atol=1e-9)


def test_solve_frame_with_pixels_without_flow_equals_stages(rsdsfm):
    """The one-call solve enqueues the RANSAC behind the flatten on the assumption of a dense flow (n = rows * cols) and checks the real
    point count at its first wait; a flow image with zero-flow pixels (dropped by the threshold, main.cc:410) takes the second path --
    everything behind the flatten runs again with the real count -- and must give what the stage-by-stage path gives, bit for bit;
    a following dense frame on the same context takes the fast path again"""
    import torch

    dev = torch.device("cuda", 0)
    d = rsdsfm.synth.make_config(3, rows=96, cols=200)
    rows, cols, K, gamma = d["rows"], d["cols"], d["K"], d["gamma"]
    holes = np.array(d["flow_img"])
    rng = np.random.default_rng(5)
    holes[rng.random((rows, cols)) < 0.07] = 0.0       # scattered pixels without flow
    holes[10:30, 50:90] = 0.0                           # and a block
    T, tol, seed = 10, 0.003, 7
    dm = torch.empty((cols, rows), dtype=torch.float64, device=dev)
    R = torch.empty((rows, 9), dtype=torch.float64, device=dev)
    t = torch.empty((rows, 3), dtype=torch.float64, device=dev)
    with rsdsfm.Solver(0) as s:
        for flow in (holes, np.array(d["flow_img"]), holes):
            img = torch.from_numpy(flow).to(dev)
            r = s.solve_frame_dev(img.data_ptr(), rows, cols, K, gamma, dm.data_ptr(), R.data_ptr(), t.data_ptr(), trials=T, tol=tol, seed=seed,
                                  flow_index_mode=rsdsfm.FLOW_GATHERED)
            s.synchronize()
            q, u, a, ak = s.flatten(flow, K, gamma)
            assert r["n"] == len(q) and (len(q) < rows * cols) == (flow is holes)
            rr = s.ransac(q, u, a, ak, False, T, tol, samples=None, seed=seed, depth_mode=1)
            ref = s.non_linear_refinement(u, rr["inliers"], rr["alpha"], rr["alpha_k"], rr["v"], rr["w"], rr["k"], False, flow_index_mode=1,
                                          inlier_idx=rr["inlier_idx"])
            dmap = s.depth_map(ref["inliers"], ref["v"], K, rows, cols)
            Rr, tr = s.pose_table(dmap["v"], ref["w"], ref["k"], gamma, rows)
            assert r["num_inliers"] == rr["num_inliers"] and r["best_trial"] == rr["best_trial"]
            assert np.array_equal(r["ransac_w"], rr["w"]) and np.array_equal(r["ransac_v"], rr["v"])
            assert np.array_equal(r["w"], ref["w"]) and np.array_equal(r["v"], dmap["v"]) and r["k"] == ref["k"] and r["flipped"] == dmap["flipped"]
            assert r["refine_summary"] == ref["summary"]
            assert np.array_equal(dm.cpu().numpy().T, dmap["depth_map"])
            assert np.array_equal(R.cpu().numpy().reshape(rows, 3, 3), Rr) and np.array_equal(t.cpu().numpy(), tr)


def test_full_solve_4k_frame(rsdsfm, big_config):
    """3840x2160 (BASELINE configs[3] size): the whole solve runs at the largest configured size -- workspace sizing,
    64-bit indexing, compaction over > 2048 workgroups -- and recovers the motion of the DeepFlow-like pair."""
    import torch

    dev = torch.device("cuda", 0)
    d = big_config(4)
    rows, cols, K, gamma = d["rows"], d["cols"], d["K"], d["gamma"]
    img = torch.from_numpy(d["flow_img"]).to(dev)
    dm = torch.empty((cols, rows), dtype=torch.float64, device=dev)
    with rsdsfm.Solver(0) as s:
        r = s.solve_frame_dev(img.data_ptr(), rows, cols, K, gamma, dm.data_ptr(), trials=8, tol=0.002, seed=5, flow_index_mode=rsdsfm.FLOW_GATHERED)
        s.synchronize()
    n = rows * cols
    assert r["n"] == n and 0.5 * n < r["num_inliers"] < 0.95 * n  # 10 % outliers + noise tail rejected
    t = d["truth"]
    vt = t["v"] / np.linalg.norm(t["v"])
    vv = r["v"] / np.linalg.norm(r["v"])
    assert np.linalg.norm(r["w"] - t["w"]) < 2e-4 and abs(float(vv @ vt)) > 0.999
    got = dm.cpu().numpy().T
    assert int((got != 0).sum()) == r["num_inliers"]
    Z = t["Z"] / np.linalg.norm(t["v"]) * np.linalg.norm(r["v"])
    mask = got != 0
    assert np.median(np.abs(got[mask] - Z[mask]) / Z[mask]) < 0.05


def test_solve_frame_is_bit_reproducible(rsdsfm):
    """every floating reduction has a fixed order and the only atomics are integer: repeated solves of the same frame --
    on one context and on a fresh one -- return bit-identical poses, summaries and depth maps"""
    import torch

    dev = torch.device("cuda", 0)
    d = rsdsfm.synth.make_config(5, rows=180, cols=320)
    rows, cols, K, gamma = d["rows"], d["cols"], d["K"], d["gamma"]
    img = torch.from_numpy(d["flow_img"]).to(dev)
    outs = []
    for ctx in range(2):
        with rsdsfm.Solver(0) as s:
            for rep in range(3):
                dm = torch.zeros((cols, rows), dtype=torch.float64, device=dev)
                r = s.solve_frame_dev(img.data_ptr(), rows, cols, K, gamma, dm.data_ptr(), trials=30, tol=0.01, seed=11)
                s.synchronize()
                outs.append((r["num_inliers"], r["best_trial"], r["v"].tobytes(), r["w"].tobytes(), r["k"], r["refine_summary"]["final_cost"],
                             r["refine_summary"]["num_iterations"], dm.cpu().numpy().tobytes()))
    assert all(o == outs[0] for o in outs[1:])


def test_profiling_records_do_not_change_results(rsdsfm):
    """rsdsfm_set_profiling brackets round 0 of the RANSAC's LM solves with two events and lets one workgroup of that launch stamp the
    shader clock: a duration and a plausible clock come back, the solve's bits stay the same, and asking without a record is an error"""
    import torch

    dev = torch.device("cuda", 0)
    d = rsdsfm.synth.make_config(5, rows=180, cols=320)
    rows, cols, K, gamma = d["rows"], d["cols"], d["K"], d["gamma"]
    img = torch.from_numpy(d["flow_img"]).to(dev)

    def solve(s):
        dm = torch.zeros((cols, rows), dtype=torch.float64, device=dev)
        r = s.solve_frame_dev(img.data_ptr(), rows, cols, K, gamma, dm.data_ptr(), trials=30, tol=0.01, seed=11)
        s.synchronize()
        return (r["num_inliers"], r["best_trial"], r["v"].tobytes(), r["w"].tobytes(), r["k"], r["refine_summary"]["final_cost"],
                dm.cpu().numpy().tobytes())

    with rsdsfm.Solver(0) as s:
        plain = solve(s)
        with pytest.raises(rsdsfm.RsdsfmError):
            s.profile_last_ms("ransac_lm_round0")
        s.set_profiling(True)
        for _ in range(3):
            assert solve(s) == plain
            ms = s.profile_last_ms("ransac_lm_round0")
            mhz = s.profile_last_ms("ransac_lm_round0_clock_mhz")
            assert 0.0 < ms < 50.0
            assert 500.0 < mhz < 2600.0, mhz  # MI355X: 2.4 GHz nominal, lower under fp64 load
        with pytest.raises(rsdsfm.RsdsfmError):
            s.profile_last_ms("depth_lm_batch")  # the pending record is the RANSAC's
        with pytest.raises(rsdsfm.RsdsfmError):
            s.profile_last_ms("no_such_record")
        s.set_profiling(False)
        assert solve(s) == plain


def test_refine_stage_placement_does_not_change_results(rsdsfm):
    """rsdsfm_set_refine_stage: the single-workgroup stage of the refinement in the next pass's prologue (1), as a launch of its own (2), or
    chosen automatically (0: prologue for single solves, own launches while a sequence has several pairs in flight) -- the same bits from
    the single solve and from the sequence solve in every mode, with k fixed and with k refined"""
    import torch

    dev = torch.device("cuda", 0)
    frames = [rsdsfm.synth.make_config(cfg, rows=120, cols=200, seed=77 + cfg) for cfg in (5, 3, 5, 1, 3)]
    rows, cols = frames[0]["rows"], frames[0]["cols"]
    imgs = [torch.from_numpy(d["flow_img"]).to(dev) for d in frames]
    dms = [torch.zeros((cols, rows), dtype=torch.float64, device=dev) for _ in frames]
    jobs = [dict(d_flow_img=im.data_ptr(), rows=rows, cols=cols, K=d["K"], gamma=d["gamma"], d_depth_map=dm.data_ptr()) for im, dm, d in zip(imgs, dms, frames)]
    for accel in (False, True):
        kw = dict(trials=16, tol=0.01, use_acceleration_mode=accel, flow_index_mode=rsdsfm.FLOW_GATHERED)
        want = None
        for mode in (0, 1, 2):
            with rsdsfm.Solver(0) as s:
                s.set_refine_stage(mode)
                seq = s.solve_frames_dev(jobs, [3 + i for i in range(len(jobs))], **kw)
                s.synchronize()
                got = [(r["num_inliers"], r["best_trial"], r["v"].tobytes(), r["w"].tobytes(), r["k"], str(r["refine_summary"]), dm.cpu().numpy().tobytes())
                       for r, dm in zip(seq, dms)]
                singles = []
                for i, (im, d) in enumerate(zip(imgs, frames)):
                    r = s.solve_frame_dev(im.data_ptr(), rows, cols, d["K"], d["gamma"], dms[i].data_ptr(), seed=3 + i, **kw)
                    s.synchronize()
                    singles.append((r["num_inliers"], r["best_trial"], r["v"].tobytes(), r["w"].tobytes(), r["k"], str(r["refine_summary"]),
                                    dms[i].cpu().numpy().tobytes()))
            assert got == singles, (accel, mode)
            want = want or got
            assert got == want, (accel, mode)
    with rsdsfm.Solver(0) as s:
        with pytest.raises(rsdsfm.RsdsfmError):
            s.set_refine_stage(3)


def test_solve_does_not_depend_on_the_contexts_history(rsdsfm):
    """a context remembers how its previous solve went -- whether the separate scoring pass was needed (noise-free data), how many
    refinement iterations it took, which depth iterate was right -- only to decide what it enqueues AHEAD of the host's reads; a
    frame solved behind frames of the other kind returns the bits of the same frame on a fresh context"""
    import torch

    dev = torch.device("cuda", 0)
    frames = []
    for cfg in (1, 5, 3):  # noise-free (three accepted steps, one refinement iteration), DeepFlow-like, noisy
        d = rsdsfm.synth.make_config(cfg, rows=150, cols=260)
        frames.append((d, torch.from_numpy(d["flow_img"]).to(dev)))

    def solve(s, k):
        d, img = frames[k]
        dm = torch.zeros((d["cols"], d["rows"]), dtype=torch.float64, device=dev)
        r = s.solve_frame_dev(img.data_ptr(), d["rows"], d["cols"], d["K"], d["gamma"], dm.data_ptr(), trials=20, tol=0.01, seed=5)
        s.synchronize()
        return (r["num_inliers"], r["best_trial"], r["v"].tobytes(), r["w"].tobytes(), r["k"], r["refine_summary"]["final_cost"],
                r["refine_summary"]["num_iterations"], dm.cpu().numpy().tobytes())

    fresh = []
    for k in range(3):
        with rsdsfm.Solver(0) as s:
            fresh.append(solve(s, k))
    assert fresh[0][6] <= 2 < fresh[1][6]  # the regimes differ the way the hints care about
    with rsdsfm.Solver(0) as s:
        for k in (0, 0, 1, 1, 0, 2, 0, 1, 2, 2, 0):
            assert solve(s, k) == fresh[k], k


def test_side_flatten_and_direct_minimal_solver_do_not_change_results(rsdsfm):
    """a dense frame forms the minimal solver's sampled points straight from the flow image (the flatten's own expressions) and runs
    the flatten beside the solver on a second stream; switching that off (flatten first, on the context's stream) returns the same
    bits -- rolling and global shutter, k estimated or not, and a frame with dropped pixels (which takes the fallback either way)"""
    import torch

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(11)
    for cfg, kw in ((5, {}), (3, dict(use_global_shutter_mode=True)), (5, dict(use_acceleration_mode=True)), (1, dict(trials=0)), (3, dict(hole=True))):
        kw = dict(kw)
        d = rsdsfm.synth.make_config(cfg, rows=130, cols=210)
        img_h = d["flow_img"].copy()
        if kw.pop("hole", False):
            img_h[40:70, 100:140] = 0.0
        img = torch.from_numpy(img_h).to(dev)
        rows, cols = d["rows"], d["cols"]
        outs = []
        for side in (3, 2, 1, 0):  # flatten inside the minimal solver's launch (default) / behind it / beside it on a second stream / in front of it
            with rsdsfm.Solver(0) as s:
                s.set_frame_side_flatten(side)
                got = []
                for rep in range(3):  # (the first solve of a context already assumes a dense frame)
                    dm = torch.zeros((cols, rows), dtype=torch.float64, device=dev)
                    R = torch.zeros((rows, 9), dtype=torch.float64, device=dev)
                    r = s.solve_frame_dev(img.data_ptr(), rows, cols, d["K"], d["gamma"], dm.data_ptr(), R.data_ptr(), 0,
                                          **dict(dict(trials=12, tol=0.01, seed=3 + rep), **kw))
                    s.synchronize()
                    got.append((r["n"], r["num_inliers"], r["best_trial"], r["ransac_v"].tobytes(), r["ransac_w"].tobytes(), r["ransac_k"], r["v"].tobytes(),
                                r["w"].tobytes(), r["k"], r["refine_summary"]["final_cost"], dm.cpu().numpy().tobytes(), R.cpu().numpy().tobytes()))
                outs.append(got)
        assert outs[0] == outs[1] == outs[2] == outs[3], (cfg, kw)


def test_sequence_solve_equals_single_solves(rsdsfm):
    """rsdsfm_solve_frames_dev: pairs of different sizes, data kinds and data seeds (dense, with dropped pixels, noise-free, DeepFlow-like)
    pipelined over 1 / 2 / 3 / 4 lanes return, pair by pair, the bits of rsdsfm_solve_frame_dev on a fresh context -- which lane a pair
    runs on, what ran before it there and what runs beside it decides when its kernels run, never what they compute"""
    import torch

    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    frames = []
    shapes = [(150, 260), (96, 250), (150, 260), (200, 120)]
    for i in range(11):
        cfg = (5, 1, 3)[i % 3]
        rows, cols = shapes[i % 4]
        d = rsdsfm.synth.make_config(cfg, rows=rows, cols=cols, seed=0x5EED0100 + i)
        img_h = d["flow_img"].copy()
        if i in (4, 5, 9):
            img_h[10:30, 20 + i:50 + i] = 0.0  # pixels without flow: the flatten drops them, the speculation on a dense frame fails
        frames.append((d, img_h))
    kw = dict(trials=16, tol=0.01)
    seeds = [7 + 3 * i for i in range(len(frames))]

    def record(r, dm, R):
        return (int(r["n"]), int(r["num_inliers"]), int(r["best_trial"]), bool(r["flipped"]), r["v"].tobytes(), r["w"].tobytes(), float(r["k"]),
                r["refine_summary"]["num_iterations"], r["refine_summary"]["final_cost"], dm.cpu().numpy().tobytes(), R.cpu().numpy().tobytes())

    with torch.cuda.stream(stream):
        bufs = []
        for d, img_h in frames:
            rows, cols = d["rows"], d["cols"]
            bufs.append(dict(img=torch.from_numpy(img_h).to(dev), dm=torch.zeros((cols, rows), dtype=torch.float64, device=dev),
                             R=torch.zeros((rows, 9), dtype=torch.float64, device=dev), t=torch.zeros((rows, 3), dtype=torch.float64, device=dev)))
        fresh = []
        for (d, _), b, sd in zip(frames, bufs, seeds):
            with rsdsfm.Solver(0, stream=stream.cuda_stream) as s:
                r = s.solve_frame_dev(b["img"].data_ptr(), d["rows"], d["cols"], d["K"], d["gamma"], b["dm"].data_ptr(), b["R"].data_ptr(), b["t"].data_ptr(), seed=sd, **kw)
                s.synchronize()
                fresh.append(record(r, b["dm"], b["R"]))
        assert len({f[0] for f in fresh}) > 3 and any(f[0] != d["rows"] * d["cols"] for f, (d, _) in zip(fresh, frames))
        jobs = [dict(d_flow_img=b["img"].data_ptr(), rows=d["rows"], cols=d["cols"], K=d["K"], gamma=d["gamma"], d_depth_map=b["dm"].data_ptr(),
                     d_R=b["R"].data_ptr(), d_t=b["t"].data_ptr()) for (d, _), b in zip(frames, bufs)]
        for lanes in (1, 2, 3, 4):
            with rsdsfm.Solver(0, stream=stream.cuda_stream) as s:
                s.set_sequence_lanes(lanes)
                for rep in range(2):  # the second pass meets warm lanes (hints of other pairs)
                    for b in bufs:
                        b["dm"].zero_(), b["R"].zero_()
                    res = s.solve_frames_dev(jobs, seeds, **kw)
                    s.synchronize()
                    got = [record(r, b["dm"], b["R"]) for r, b in zip(res, bufs)]
                    assert got == fresh, (lanes, rep, [i for i in range(len(got)) if got[i] != fresh[i]])
        # errors surface: a pair with fewer than 9 points fails the call like the single solve does
        tiny = torch.zeros((3, 3, 2), dtype=torch.float64, device=dev)
        bad = dict(jobs[0], d_flow_img=tiny.data_ptr(), rows=3, cols=3)
        with rsdsfm.Solver(0, stream=stream.cuda_stream) as s:
            with pytest.raises(rsdsfm.RsdsfmError):
                s.solve_frames_dev([jobs[1], bad, jobs[2]], [1, 2, 3], **kw)
            r = s.solve_frames_dev([jobs[1]], [seeds[1]], **kw)  # the context stays usable
            s.synchronize()
            assert record(r[0], bufs[1]["dm"], bufs[1]["R"]) == fresh[1]


def test_frame_params_struct_bytes_is_checked(rsdsfm):
    """rsdsfm_frame_params_init fills the reference's constants and stamps the struct size; a struct from another header layout
    (round 1's: ransac_tol where struct_bytes now sits) is refused instead of being misread"""
    import ctypes as C

    import torch

    lib = rsdsfm.load_library()
    p = rsdsfm.FrameParams()
    lib.rsdsfm_frame_params_init.restype = None
    lib.rsdsfm_frame_params_init(C.byref(p))
    assert (p.ransac_trials, p.use_refinement, p.flow_index_mode, p.use_global_shutter_mode) == (5, 1, rsdsfm.FLOW_COMPAT_RANK, 0)
    assert p.struct_bytes == C.sizeof(rsdsfm.FrameParams) and p.ransac_tol == 0.05 and p.flow_threshold == 1e-10
    dev = torch.device("cuda", 0)
    d = rsdsfm.synth.make_config(5, rows=64, cols=80)
    img = torch.from_numpy(d["flow_img"]).to(dev)
    dm = torch.zeros((80, 64), dtype=torch.float64, device=dev)
    res = rsdsfm.FrameResult()
    dd = C.c_double
    with rsdsfm.Solver(0) as s:
        args = (s._ctx, C.c_void_p(img.data_ptr()), C.c_int32(64), C.c_int32(80), dd(d["K"][0]), dd(d["K"][1]), dd(d["K"][2]), dd(d["K"][3]), dd(d["gamma"]))
        assert lib.rsdsfm_solve_frame_dev(*args, C.byref(p), C.c_void_p(dm.data_ptr()), None, None, C.byref(res)) == 0
        p.struct_bytes = C.sizeof(rsdsfm.FrameParams) - 8
        assert lib.rsdsfm_solve_frame_dev(*args, C.byref(p), C.c_void_p(dm.data_ptr()), None, None, C.byref(res)) != 0
        assert b"struct_bytes" in lib.rsdsfm_last_error(s._ctx)


def test_concurrent_contexts_do_not_interfere(oracle, rsdsfm):
    """the sequence-throughput mode of bench.py: several contexts on separate streams, (a) one host thread interleaving
    asynchronous depth solves of DIFFERENT problems, (b) one host thread per context running whole-frame solves -- every
    result equals what the same context produces alone (bit-exact) and the oracle"""
    import threading

    import torch

    dev = torch.device("cuda", 0)
    S = 3
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    solvers = [rsdsfm.Solver(0, stream=st.cuda_stream) for st in streams]
    probs = []
    for j in range(S):
        d = rsdsfm.synth.make_config(1 if j != 1 else 3, rows=120 + 16 * j, cols=200)
        t = d["truth"]
        v = t["v"] / np.linalg.norm(t["v"]) * (1.0 if j != 2 else -1.0)
        tens = {k2: torch.from_numpy(d[k2]).to(dev) for k2 in ("q", "u", "alpha", "alpha_k")}
        rho = torch.zeros(len(d["alpha"]), dtype=torch.float64, device=dev)
        probs.append((d, v, t["w"], tens, rho))
    torch.cuda.synchronize()
    for rep in range(20):  # interleaved, asynchronous
        for j, (d, v, w, tens, rho) in enumerate(probs):
            solvers[j].estimate_inverse_depths_dev(tens["q"].data_ptr(), tens["u"].data_ptr(), len(d["alpha"]), v, w, 0.0, tens["alpha"].data_ptr(),
                                                   tens["alpha_k"].data_ptr(), rho.data_ptr(), mode=1)
    for j, (d, v, w, tens, rho) in enumerate(probs):
        sm, _ = solvers[j].depth_finish_dev(tens["q"].data_ptr(), tens["u"].data_ptr(), len(d["alpha"]), v, w, 0.0, tens["alpha"].data_ptr(),
                                            tens["alpha_k"].data_ptr(), rho.data_ptr())
        rho_o, sm_o = oracle.estimate_inverse_depths(d["q"], d["u"], v, w, 0.0, d["alpha"], d["alpha_k"], mode=1)
        assert sm["num_successful_steps"] == sm_o["num_successful_steps"] and sm["termination"] == sm_o["termination"]
        assert np.allclose(rho.cpu().numpy(), rho_o, rtol=1e-9, atol=1e-13)
    # (b) whole-frame solves from one thread per context
    frames = [rsdsfm.synth.make_config(5, rows=150, cols=260, seed=100 + j) for j in range(S)]

    def solve(sv, d, reps):
        img = torch.from_numpy(d["flow_img"]).to(dev)
        dm = torch.zeros((d["cols"], d["rows"]), dtype=torch.float64, device=dev)
        r = None
        for _ in range(reps):
            r = sv.solve_frame_dev(img.data_ptr(), d["rows"], d["cols"], d["K"], d["gamma"], dm.data_ptr(), trials=20, tol=0.01, seed=5)
        sv.synchronize()
        return (r["num_inliers"], r["best_trial"], r["v"].tobytes(), r["w"].tobytes(), dm.cpu().numpy().tobytes())

    alone = []
    for j in range(S):
        with torch.cuda.stream(streams[j]):
            alone.append(solve(solvers[j], frames[j], 1))
    together = [None] * S

    def worker(j):
        with torch.cuda.stream(streams[j]):
            together[j] = solve(solvers[j], frames[j], 6)

    ths = [threading.Thread(target=worker, args=(j,)) for j in range(S)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert together == alone
    for sv in solvers:
        sv.close()


@pytest.mark.parametrize("flow_mode", [0, 1])
def test_full_pipeline_1920x1080_matches_oracle_chain(oracle, rsdsfm, big_config, flow_mode):
    """BASELINE configs[2] ("real_world 1920x1080 full pipeline", 5 RANSAC trials like main.cc:304) end to end against the
    oracle chain on the same sampler: every integer (points, per-trial counts, winner, inlier list, LM step counts, depth-map
    support, scanline indices) bit-exact, floats to 1e-6 (north-star bar 1e-5)"""
    import torch

    dev = torch.device("cuda", 0)
    d = big_config(3)
    rows, cols, K, gamma = d["rows"], d["cols"], d["K"], d["gamma"]
    T, tol, seed = 5, 0.002, 2024
    img = torch.from_numpy(d["flow_img"]).to(dev)
    dm = torch.empty((cols, rows), dtype=torch.float64, device=dev)
    with rsdsfm.Solver(0) as s:
        r = s.solve_frame_dev(img.data_ptr(), rows, cols, K, gamma, dm.data_ptr(), trials=T, tol=tol, seed=seed, flow_index_mode=flow_mode)
        s.synchronize()
        q, u, a, ak = s.flatten(d["flow_img"], K, gamma)
        rr = s.ransac(q, u, a, ak, False, T, tol, samples=None, seed=seed, depth_mode=1)
    assert r["n"] == rows * cols == len(q)
    ro = oracle.ransac(q, u, a, ak, False, T, tol, oracle.sample_indices(len(q), T, seed), depth_mode=1)
    assert np.array_equal(rr["trial_count"], ro["trial_count"]) and np.array_equal(rr["trial_steps"], ro["trial_steps"])
    assert rr["best_trial"] == ro["best_trial"] == r["best_trial"] and r["num_inliers"] == ro["num_inliers"]
    assert np.array_equal(rr["inlier_idx"], ro["inlier_idx"])
    refo = oracle.refine(u, ro["inliers"], ro["alpha"], ro["alpha_k"], ro["v"], ro["w"], ro["k"], False, flow_mode, ro["inlier_idx"] if flow_mode else None)
    for key in ("num_iterations", "num_successful_steps", "termination"):
        assert r["refine_summary"][key] == refo["summary"][key], key
    inl_o, v_o, flipped_o = oracle.canonicalize_sign(refo["inliers"], refo["v"])
    assert r["flipped"] == flipped_o
    assert np.allclose(r["v"], v_o, rtol=1e-6, atol=1e-10) and np.allclose(r["w"], refo["w"], rtol=1e-6, atol=1e-10)
    dm_o, _, _ = oracle.scatter_depth(inl_o, *K, rows, cols)
    got = dm.cpu().numpy().T
    assert np.array_equal(got != 0, dm_o != 0) and np.allclose(got, dm_o, rtol=1e-6)


def test_bench_full_workload_matches_oracle_chain(oracle, rsdsfm):
    """exactly what `bench.py --workload full` times (1280x720 DeepFlow-like pair, 50 trials, tol 0.05, seed 1): integers
    bit-exact against the oracle chain, pose to 1e-6"""
    import torch

    dev = torch.device("cuda", 0)
    d = rsdsfm.synth.make_config(5, seed=0x5EED0005)
    rows, cols, K, gamma = d["rows"], d["cols"], d["K"], d["gamma"]
    T, tol, seed = 50, 0.05, 1
    img = torch.from_numpy(d["flow_img"]).to(dev)
    dm = torch.empty((cols, rows), dtype=torch.float64, device=dev)
    with rsdsfm.Solver(0) as s:
        r = s.solve_frame_dev(img.data_ptr(), rows, cols, K, gamma, dm.data_ptr(), trials=T, tol=tol, seed=seed)
        s.synchronize()
    q, u, qpx, fpx = oracle.flatten(d["flow_img"], *K, gamma)
    a, ak = oracle.get_alpha(fpx, rows, gamma), oracle.get_alpha_k(qpx, fpx, rows, gamma)
    ro = oracle.ransac(q, u, a, ak, False, T, tol, oracle.sample_indices(len(q), T, seed), depth_mode=1)
    assert r["n"] == len(q) and r["best_trial"] == ro["best_trial"] and r["num_inliers"] == ro["num_inliers"]
    refo = oracle.refine(u, ro["inliers"], ro["alpha"], ro["alpha_k"], ro["v"], ro["w"], ro["k"], False, 0, None)  # the default: flow by inlier rank
    for key in ("num_iterations", "num_successful_steps", "termination"):
        assert r["refine_summary"][key] == refo["summary"][key], key
    inl_o, v_o, flipped_o = oracle.canonicalize_sign(refo["inliers"], refo["v"])
    assert r["flipped"] == flipped_o and np.allclose(r["v"], v_o, rtol=1e-6, atol=1e-10) and np.allclose(r["w"], refo["w"], rtol=1e-6, atol=1e-10)
    dm_o, _, _ = oracle.scatter_depth(inl_o, *K, rows, cols)
    got = dm.cpu().numpy().T
    assert np.array_equal(got != 0, dm_o != 0) and np.allclose(got, dm_o, rtol=1e-6)


def test_randomised_parity_campaign_sample(rsdsfm):
    """120 cases of tests/fuzz_gpu.py (random frame sizes, motions, noise, tolerances, trial counts; depth solve, RANSAC and
    refinement against the oracle).  The full campaign (thousands of cases, `python tests/fuzz_gpu.py 400 <seed>`) is how the
    statements of DESIGN.md section 6 about ties and ill-conditioned refinements were established."""
    import fuzz_gpu

    assert fuzz_gpu.main(120, 1) == 0


def test_global_shutter_mode_matches_oracle_chain(oracle, rsdsfm):
    """use_global_shutter_mode (main.cc:305, :441-444): alpha = 1 for every point before RANSAC -- the one-call solve equals the
    oracle chain run on alpha = alpha * 0 + 1"""
    import torch

    dev = torch.device("cuda", 0)
    d = rsdsfm.synth.make_config(3, rows=144, cols=256)
    rows, cols, K, gamma = d["rows"], d["cols"], d["K"], d["gamma"]
    T, tol, seed = 10, 0.004, 21
    img = torch.from_numpy(d["flow_img"]).to(dev)
    dm = torch.empty((cols, rows), dtype=torch.float64, device=dev)
    with rsdsfm.Solver(0) as s:
        r = s.solve_frame_dev(img.data_ptr(), rows, cols, K, gamma, dm.data_ptr(), trials=T, tol=tol, seed=seed, use_global_shutter_mode=True)
        r_rs = s.solve_frame_dev(img.data_ptr(), rows, cols, K, gamma, dm.data_ptr(), trials=T, tol=tol, seed=seed)
        s.synchronize()
    q, u, qpx, fpx = oracle.flatten(d["flow_img"], *K, gamma)
    a, ak = oracle.get_alpha(fpx, rows, gamma) * 0.0 + 1.0, oracle.get_alpha_k(qpx, fpx, rows, gamma)
    ro = oracle.ransac(q, u, a, ak, False, T, tol, oracle.sample_indices(len(q), T, seed), depth_mode=1)
    assert r["num_inliers"] == ro["num_inliers"] and r["best_trial"] == ro["best_trial"]
    assert np.allclose(r["ransac_w"], ro["w"], atol=1e-10) and np.allclose(r["ransac_v"], ro["v"], atol=1e-10)
    assert r["num_inliers"] != r_rs["num_inliers"]  # the rolling-shutter model explains this data differently
    refo = oracle.refine(u, ro["inliers"], ro["alpha"], ro["alpha_k"], ro["v"], ro["w"], ro["k"], False, 0, None)
    _, v_o, _ = oracle.canonicalize_sign(refo["inliers"], refo["v"])
    assert np.allclose(r["v"], v_o, rtol=1e-6, atol=1e-10) and np.allclose(r["w"], refo["w"], rtol=1e-6, atol=1e-10)


def test_context_from_a_fresh_thread_and_device_is_restored(oracle, rsdsfm):
    """every entry point makes the context's device current for the calling thread (allocations follow the CURRENT device) and
    restores the caller's: a context used from a new host thread works, and on a multi-GPU box a context of device 1 leaves the
    caller on device 0"""
    import threading

    import torch

    d = rsdsfm.synth.make_config(1, rows=60, cols=80)
    t = d["truth"]
    v = t["v"] / np.linalg.norm(t["v"])
    rho_o, _ = oracle.estimate_inverse_depths(d["q"], d["u"], v, t["w"], 0.0, d["alpha"], d["alpha_k"], mode=1)
    ndev = torch.cuda.device_count()
    for device in range(min(ndev, 2)):
        torch.cuda.set_device(0)
        s = rsdsfm.Solver(device)
        assert torch.cuda.current_device() == 0
        out = {}

        def work():
            out["rho"], _ = s.estimate_inverse_depths(d["q"], d["u"], v, t["w"], 0.0, d["alpha"], d["alpha_k"], mode=1)  # grows the staging buffers
            out["rr"] = s.ransac(d["q"], d["u"], d["alpha"], d["alpha_k"], False, 4, 0.05, seed=3)["num_inliers"]

        th = threading.Thread(target=work)
        th.start()
        th.join()
        assert np.allclose(out["rho"], rho_o, rtol=1e-9, atol=1e-13) and out["rr"] > 0
        assert torch.cuda.current_device() == 0
        s.close()


@pytest.mark.parametrize("cfg,T,tol", [(5, 10, 0.01), (3, 5, 0.004)])
def test_acceleration_mode_full_size_matches_oracle_chain(oracle, rsdsfm, cfg, T, tol):
    """use_acceleration_mode (main.cc:306: the minimal solver estimates k, the refinement frees it: 7 parameters, 7x7 Schur
    complement) at BASELINE sizes -- 1280x720 and 1920x1080 DeepFlow-like pairs generated with k = 0.4 -- against the oracle
    chain: integers (points, per-trial counts and LM steps, winner, inlier list, refinement iteration counts, depth-map
    support) exact, v, w, k and the depth map within the north-star 1e-5 (measured ~1e-9)"""
    import torch

    dev = torch.device("cuda", 0)
    d = rsdsfm.synth.make_config(cfg, k=0.4)
    rows, cols, K, gamma = d["rows"], d["cols"], d["K"], d["gamma"]
    seed = 31
    img = torch.from_numpy(d["flow_img"]).to(dev)
    dm = torch.empty((cols, rows), dtype=torch.float64, device=dev)
    with rsdsfm.Solver(0) as s:
        r = s.solve_frame_dev(img.data_ptr(), rows, cols, K, gamma, dm.data_ptr(), trials=T, tol=tol, seed=seed, use_acceleration_mode=True,
                              flow_index_mode=rsdsfm.FLOW_GATHERED)
        s.synchronize()
        q, u, a, ak = s.flatten(d["flow_img"], K, gamma)
        rr = s.ransac(q, u, a, ak, True, T, tol, samples=None, seed=seed, depth_mode=1)
    ro = oracle.ransac(q, u, a, ak, True, T, tol, oracle.sample_indices(len(q), T, seed), depth_mode=1)
    assert np.array_equal(rr["trial_count"], ro["trial_count"]) and np.array_equal(rr["trial_steps"], ro["trial_steps"])
    assert rr["best_trial"] == ro["best_trial"] == r["best_trial"] and r["num_inliers"] == ro["num_inliers"]
    assert np.array_equal(rr["inlier_idx"], ro["inlier_idx"])
    assert 0.3 * len(q) < ro["num_inliers"] < len(q) and abs(ro["k"]) > 1e-3  # a selective problem with a real k estimate
    refo = oracle.refine(u, ro["inliers"], ro["alpha"], ro["alpha_k"], ro["v"], ro["w"], ro["k"], True, 1, ro["inlier_idx"])
    for key in ("num_iterations", "num_successful_steps", "num_unsuccessful_steps", "termination"):
        assert r["refine_summary"][key] == refo["summary"][key], key
    assert np.isclose(r["refine_summary"]["final_cost"], refo["summary"]["final_cost"], rtol=1e-9)
    inl_o, v_o, flipped_o = oracle.canonicalize_sign(refo["inliers"], refo["v"])
    assert r["flipped"] == flipped_o
    rel = lambda x, y: float(np.max(np.abs(np.asarray(x) - np.asarray(y))) / np.max(np.abs(np.asarray(y))))
    dv, dw, dk = rel(r["v"], v_o), rel(r["w"], refo["w"]), abs(r["k"] - refo["k"]) / abs(refo["k"])
    print("accel mode %dx%d: v %.2e w %.2e k %.2e (iterations %d)" % (cols, rows, dv, dw, dk, refo["summary"]["num_iterations"]))
    assert dv <= 1e-5 and dw <= 1e-5 and dk <= 1e-5
    dm_o, _, _ = oracle.scatter_depth(inl_o, *K, rows, cols)
    got = dm.cpu().numpy().T
    assert np.array_equal(got != 0, dm_o != 0) and np.allclose(got, dm_o, rtol=1e-5)


@pytest.mark.parametrize("flow_mode", [0, 1])
def test_stage_pipeline_on_device_buffers_equals_one_call_solve(rsdsfm, flow_mode):
    """pipeline.FramePipeline drives the solve stage by stage through the `_dev` entry points on caller-owned HBM buffers
    (flatten -> ransac -> refine -> depth map -> pose table); the one-call rsdsfm_solve_frame_dev is the same launch sequence:
    bit-identical pose, counts, depth map and pose table"""
    import torch

    dev = torch.device("cuda", 0)
    d = rsdsfm.synth.make_config(5, rows=150, cols=260)
    rows, cols, K, gamma = d["rows"], d["cols"], d["K"], d["gamma"]
    stream = torch.cuda.Stream(dev)
    with torch.cuda.stream(stream):
        img = torch.from_numpy(d["flow_img"]).to(dev)
        dm = torch.empty((cols, rows), dtype=torch.float64, device=dev)
        R = torch.empty((rows, 9), dtype=torch.float64, device=dev)
        t = torch.empty((rows, 3), dtype=torch.float64, device=dev)
        with rsdsfm.Solver(0, stream=stream.cuda_stream) as s:
            pipe = rsdsfm.pipeline.FramePipeline(s, torch, dev, rows, cols, K, gamma)
            p = pipe.solve(img, trials=16, tol=0.01, seed=5, flow_index_mode=flow_mode)
            r = s.solve_frame_dev(img.data_ptr(), rows, cols, K, gamma, dm.data_ptr(), R.data_ptr(), t.data_ptr(), trials=16, tol=0.01, seed=5,
                                  flow_index_mode=flow_mode)
            s.synchronize()
        assert p["n"] == r["n"] and p["num_inliers"] == r["num_inliers"] and 0 < r["num_inliers"] < r["n"]
        assert np.array_equal(p["v"], r["v"]) and np.array_equal(p["w"], r["w"]) and p["k"] == r["k"] and p["flipped"] == r["flipped"]
        assert p["refine"]["summary"] == r["refine_summary"]
        assert torch.equal(pipe.depth_map, dm) and torch.equal(pipe.R, R) and torch.equal(pipe.t, t)


def test_frame_solve_starts_over_with_the_standard_functions(rsdsfm):
    """round 0 of the depth solves and the minimal solver run sqrt / reciprocal / division through their in-range cores; a frame with a
    pixel whose flow is 1e160 px -- alpha_k overflows, beta = alpha + 0 x inf = NaN: a NaN pixel -- makes the one-call
    solve start its RANSAC over with the standard functions behind the speculated chain (which leaves at once).  Same results as with the
    standard functions from the start, bit for bit, in the single solve and in the sequence solve; the dense frames around it in the
    sequence are not affected."""
    import torch

    dev = torch.device("cuda", 0)
    d = rsdsfm.synth.make_config(5, rows=64, cols=480)  # 30720 points: 20 full tiles of 1536
    rows, cols, K = d["rows"], d["cols"], d["K"]
    gamma = 0.5
    clean = np.array(d["flow_img"])
    bad = clean.copy()
    bad[17, 301] = (3.0, 1e160)
    imgs = {"clean": torch.from_numpy(clean).to(dev), "bad": torch.from_numpy(bad).to(dev)}
    outs = {}
    for math in (0, 1, 2):  # 2: the default arithmetic of the depth solves (analytic LM trajectory): no restart of either kind, the same bits
        with rsdsfm.Solver(0) as s:
            s.set_ransac_math(math & 1)
            s.set_lm_arithmetic(1 if math < 2 else 0)
            s.set_refine_arithmetic(1)  # (the same refinement behind the three forms of the depth solves: their results are compared bit for bit)
            res = []
            for name in ("clean", "bad", "clean"):
                dm = torch.empty((cols, rows), dtype=torch.float64, device=dev)
                R = torch.empty((rows, 9), dtype=torch.float64, device=dev)
                t = torch.empty((rows, 3), dtype=torch.float64, device=dev)
                r = s.solve_frame_dev(imgs[name].data_ptr(), rows, cols, K, gamma, dm.data_ptr(), R.data_ptr(), t.data_ptr(), trials=20, tol=0.05, seed=11)
                s.synchronize()
                res.append((r["n"], r["num_inliers"], r["best_trial"], r["v"].tobytes(), r["w"].tobytes(), r["k"], r["flipped"], str(r["refine_summary"]),
                            dm.cpu().numpy().tobytes(), R.cpu().numpy().tobytes(), t.cpu().numpy().tobytes()))
            restarts_single = s.ransac_restarts()
            # the same three pairs through the sequence solve (lanes of their own)
            dms = [torch.empty((cols, rows), dtype=torch.float64, device=dev) for _ in range(3)]
            jobs = [dict(d_flow_img=imgs[name].data_ptr(), rows=rows, cols=cols, K=K, gamma=gamma, d_depth_map=dm.data_ptr()) for name, dm in zip(("clean", "bad", "clean"), dms)]
            call = s.prepared_frames_solve(jobs, trials=20, tol=0.05)
            rs = call([11, 11, 11])
            seq = [(int(x.n_points), int(x.num_inliers), int(x.best_trial), bytes(bytearray(np.array(x.v[:]).tobytes())), dm.cpu().numpy().tobytes()) for x, dm in zip(rs, dms)]
            restarts_all = s.ransac_restarts()
            # (the analytic pass takes a NaN pixel like the reference takes it: five invalid steps per trial, no guard, no cores, no restart)
            assert s.lma_restarts()[0] == 0
        outs[math] = (res, seq)
        assert restarts_single == (1 if math == 0 else 0)
        assert restarts_all == (2 if math == 0 else 0)  # (the sequence's lane met the pixel with the cores again)
        for (a, b) in zip(res, seq):
            assert (a[0], a[1], a[2], a[3], a[8]) == b
    assert outs[0] == outs[1] == outs[2]
    assert outs[0][0][0] == outs[0][0][2] and outs[0][0][0] != outs[0][0][1]


def test_one_call_solve_equals_stage_pipeline_on_random_frames(rsdsfm):
    """The one-call solve speculates at three places (a dense flow, the RANSAC's final stage, the refinement ending within its first
    chunk) and enqueues each stage behind the one before it; the stage pipeline (pipeline.FramePipeline over the `_dev` entry points)
    waits for the host in between.  36 random frames on ONE context pair -- sizes, noise-free / DeepFlow-like data, pixels without flow,
    1...50 trials, tolerances that make the inlier set a strict subset, acceleration mode, closed-form depths, refinement on / off,
    both flow index modes, each configuration behind a different one (history) -- must agree bit for bit in every output."""
    import torch

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(20261002)
    stream = torch.cuda.Stream(dev)
    with torch.cuda.stream(stream), rsdsfm.Solver(0, stream=stream.cuda_stream) as s_one, rsdsfm.Solver(0, stream=stream.cuda_stream) as s_stage:
        for case in range(36):
            rows, cols = int(rng.integers(40, 140)), int(rng.integers(60, 230))
            cfg = int(rng.choice([1, 3, 5]))
            d = rsdsfm.synth.make_config(cfg, seed=int(rng.integers(1 << 30)), rows=rows, cols=cols)
            K, gamma = d["K"], d["gamma"]
            flow = np.array(d["flow_img"])
            if rng.random() < 0.4:
                flow[rng.random((rows, cols)) < rng.uniform(0.01, 0.3)] = 0.0
            if not np.all(np.isfinite(flow)):
                continue
            kw = dict(trials=int(rng.choice([1, 3, 5, 20, 50])), tol=float(rng.choice([0.05, 0.01, 0.003])), seed=int(rng.integers(1, 1000)),
                      flow_index_mode=int(rng.integers(2)))
            accel, closed, refine = bool(rng.random() < 0.25), bool(rng.random() < 0.2), bool(rng.random() < 0.85)
            img = torch.from_numpy(flow).to(dev)
            dm = torch.empty((cols, rows), dtype=torch.float64, device=dev)
            R = torch.empty((rows, 9), dtype=torch.float64, device=dev)
            t = torch.empty((rows, 3), dtype=torch.float64, device=dev)
            tag = "case %d: %dx%d cfg %d %s accel %s closed %s refine %s" % (case, rows, cols, cfg, kw, accel, closed, refine)
            pipe = rsdsfm.pipeline.FramePipeline(s_stage, torch, dev, rows, cols, K, gamma)
            try:
                p = pipe.solve(img, use_alpha_k=accel, refine=refine, depth_mode=0 if closed else 1, **kw)
            except rsdsfm.RsdsfmError as e:  # e.g. no real k for a hypothesis in acceleration mode, fewer than 9 points: the one-call solve must fail too
                with pytest.raises(rsdsfm.RsdsfmError):
                    s_one.solve_frame_dev(img.data_ptr(), rows, cols, K, gamma, dm.data_ptr(), R.data_ptr(), t.data_ptr(), use_acceleration_mode=accel,
                                          use_refinement=refine, depth_mode=0 if closed else 1, **kw)
                continue
            r = s_one.solve_frame_dev(img.data_ptr(), rows, cols, K, gamma, dm.data_ptr(), R.data_ptr(), t.data_ptr(), use_acceleration_mode=accel,
                                      use_refinement=refine, depth_mode=0 if closed else 1, **kw)
            s_one.synchronize()
            assert p["n"] == r["n"] and p["num_inliers"] == r["num_inliers"], tag
            assert np.array_equal(p["v"], r["v"], equal_nan=True) and np.array_equal(p["w"], r["w"], equal_nan=True), tag
            assert (p["k"] == r["k"] or (np.isnan(p["k"]) and np.isnan(r["k"]))) and p["flipped"] == r["flipped"], tag
            if refine:
                assert p["refine"]["summary"] == r["refine_summary"] or np.isnan(r["refine_summary"]["final_cost"]), tag
            eq = lambda a, b: torch.equal(a, b) or torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0))
            assert eq(pipe.depth_map, dm) and eq(pipe.R, R) and eq(pipe.t, t), tag
