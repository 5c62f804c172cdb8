"""GPU: RS -> GS back projection, crack interpolation and 8-bit depth image through the C ABI against
the oracle and the committed fixtures.  Everything here is byte / index work: bit-exact, including the float32 world
points (same per-pixel operation chain, no contraction)."""
import numpy as np
import pytest

from conftest import RECTIFY_CASES

pytestmark = pytest.mark.gpu


def _scene(rsdsfm, oracle, rows, cols, seed, k=0.0, cfg=1):
    d = rsdsfm.synth.make_config(cfg, rows=rows, cols=cols, k=k)
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, size=(rows, cols, 3), dtype=np.uint8)
    img[rng.random((rows, cols)) < 0.05] = (2, 3, 1)
    img[rng.random((rows, cols)) < 0.01] = (1, 1, 1)
    depth = np.array(d["truth"]["Z"])
    depth[rng.random((rows, cols)) < 0.07] = 0.0
    v, w = np.array([0.12, 0.10, 0.05]), np.array([0.03, -0.02, 0.06])
    R, t = oracle.pose_table(v, w, k, d["gamma"], rows)
    return d, img, depth, R, t


@pytest.mark.parametrize("case", RECTIFY_CASES)
def test_rectify_matches_golden(golden_rectify, rsdsfm, case):
    g = lambda k: golden_rectify[case + "/" + k]
    K = tuple(g("K"))
    rows, cols = g("depth").shape
    with rsdsfm.Solver(0) as s:
        for mode in (0, 1):
            for q5 in (0, 1):
                gs, c3 = s.back_project(g("image"), g("depth"), g("R"), g("t"), K, mode=mode, q5_mode=q5)
                assert np.array_equal(gs, g("gs_m%d_q%d" % (mode, q5)))
                assert np.allclose(c3, g("c3_m%d" % mode), rtol=2e-6, atol=1e-7)
        for off in (1, 2):
            assert np.array_equal(s.interpolate_cracky(g("gs_m0_q0"), off), g("interp_off%d" % off))
        assert np.array_equal(s.depth_preview(g("inliers"), K, rows, cols), g("preview"))


@pytest.mark.parametrize("rows,cols,k", [(1, 1, 0.0), (7, 5, 0.0), (64, 32, 0.0), (65, 33, 0.4), (131, 257, 0.0), (720, 1280, 0.4)])
def test_back_project_equals_oracle(oracle, rsdsfm, rows, cols, k):
    """ragged tile edges (tiles are 32 x 64), a single pixel, and BASELINE's 1280x720"""
    d, img, depth, R, t = _scene(rsdsfm, oracle, rows, cols, seed=rows * 1000 + cols, k=k)
    K = d["K"]
    with rsdsfm.Solver(0) as s:
        for mode, q5 in ((0, 0), (0, 1), (1, 0)):
            gs, c3 = s.back_project(img, depth, R, t, K, mode=mode, q5_mode=q5)
            gs_o, c3_o = oracle.back_project(img, depth, R, t, *K, mode=mode, q5_mode=q5)
            assert np.array_equal(gs, gs_o), (mode, q5)
            assert np.array_equal(c3.view(np.uint32), c3_o.view(np.uint32)), (mode, q5)
        gs2, none = s.back_project(img, depth, R, t, K, want_coords=False)
        assert none is None and np.array_equal(gs2, oracle.back_project(img, depth, R, t, *K)[0])


def test_back_project_collisions_last_writer_wins(oracle, rsdsfm):
    """a strong forward motion squeezes many source pixels onto few targets: the winner must be the pixel latest in the
    reference's scan for every target, independent of the GPU's execution order (integer atomicMax)"""
    rows, cols = 300, 400
    d, img, depth, _, _ = _scene(rsdsfm, oracle, rows, cols, seed=5)
    K = d["K"]
    R, t = oracle.pose_table(np.array([0.0, 0.0, -3.0]), np.array([0.2, 0.1, 0.3]), 0.0, d["gamma"], rows)
    with rsdsfm.Solver(0) as s:
        outs = [s.back_project(img, np.abs(depth) + 1.0, R, t, K, want_coords=False)[0] for _ in range(3)]
    exp = oracle.back_project(img, np.abs(depth) + 1.0, R, t, *K, want_coords=False)[0]
    covered = (exp.reshape(-1, 3).sum(axis=1) != 0).mean()
    assert covered < 0.9  # many collisions / cracks
    for o in outs:
        assert np.array_equal(o, exp)


def test_claim_maps_survive_epoch_wrap_and_resizing(oracle, rsdsfm):
    """the forward-splat claim maps are persistent per context and carry an 8-bit epoch instead of being cleared per frame:
    600 back projections / depth images on ONE context, alternating between two scenes whose targets differ (so stale claims of
    the previous frame sit exactly where the current frame leaves holes), across two epoch wraps and a change of image size,
    each result equal to the oracle's"""
    rng = np.random.default_rng(3)
    scenes = []
    for rows, cols, vz in ((60, 90, -3.0), (60, 90, 2.5), (75, 70, -2.0)):
        d, img, depth, _, _ = _scene(rsdsfm, oracle, rows, cols, seed=rows + cols + int(10 * vz) + 100)
        K = d["K"]
        R, t = oracle.pose_table(np.array([0.05, -0.02, vz]), np.array([0.2, -0.1, 0.3]), 0.0, d["gamma"], rows)
        depth = np.abs(depth) + 1.0
        exp = oracle.back_project(img, depth, R, t, *K, want_coords=False)[0]
        inl = np.column_stack([rng.uniform(-0.15, 0.15, 3000), rng.uniform(-0.1, 0.1, 3000), rng.normal(2.0, 1.0, 3000)])
        scenes.append((img, depth, R, t, K, exp, inl, oracle.depth_preview(inl, *K, rows, cols), rows, cols))
    assert (scenes[0][5].reshape(-1, 3).sum(axis=1) != 0).mean() < 0.9 and not np.array_equal(scenes[0][5] != 0, scenes[1][5] != 0)
    with rsdsfm.Solver(0) as s:
        for i in range(600):
            img, depth, R, t, K, exp, inl, pv, rows, cols = scenes[(i % 2) if i < 560 else 2]
            assert np.array_equal(s.back_project(img, depth, R, t, K, want_coords=False)[0], exp), i
            if i % 3 == 0:
                assert np.array_equal(s.depth_preview(inl, K, rows, cols), pv), i


def test_claim_map_after_a_frame_beyond_2_pow_24_pixels(oracle, rsdsfm):
    """frames of more than 2^24 pixels use a 1-bit tag and a clear per call; the words they leave behind (0x80000000 | index) would
    beat every epoch tag of a later, smaller frame on the same context -- the next narrow call must start from a cleared map.  And
    the depth image's claim word holds the INLIER index: more inliers than 2^24 on a small image must not spill into the epoch bits."""
    import torch

    dev = torch.device("cuda", 0)
    rows, cols = 4100, 4100  # 16.81 M pixels > 2^24
    assert rows * cols > 1 << 24
    d, img, depth, R, t = _scene(rsdsfm, oracle, 60, 90, seed=5)
    K = d["K"]
    depth = np.abs(depth) + 1.0
    exp_small = oracle.back_project(img, depth, R, t, *K, want_coords=False)[0]
    inl_small = np.column_stack([np.linspace(-0.15, 0.15, 3000), np.linspace(-0.1, 0.1, 3000), np.linspace(1.0, 3.0, 3000)])
    pv_small = oracle.depth_preview(inl_small, *K, 60, 90)
    big_img = torch.full((rows, cols, 3), 7, dtype=torch.uint8, device=dev)
    big_depth = torch.full((cols, rows), 2.0, dtype=torch.float64, device=dev)
    Rb = torch.eye(3, dtype=torch.float64, device=dev).reshape(1, 9).repeat(rows, 1).contiguous()
    tb = torch.zeros((rows, 3), dtype=torch.float64, device=dev)
    gs = torch.zeros((rows, cols, 3), dtype=torch.uint8, device=dev)
    Kb = (3000.0, 3000.0, cols / 2.0, rows / 2.0)
    with rsdsfm.Solver(0) as s:
        assert np.array_equal(s.back_project(img, depth, R, t, K, want_coords=False)[0], exp_small)
        s.back_project_dev(big_img.data_ptr(), big_depth.data_ptr(), Rb.data_ptr(), tb.data_ptr(), Kb, rows, cols, gs.data_ptr())
        s.synchronize()
        assert float((gs.view(-1, 3).sum(dim=1) != 0).double().mean().item()) > 0.9  # identity motion: (almost) every pixel lands on itself
        for _ in range(3):
            assert np.array_equal(s.back_project(img, depth, R, t, K, want_coords=False)[0], exp_small)
        # depth image: 2^24 + 5 inliers on a 60 x 90 image (all but the last 3000 outside the image), then the small list again
        m = (1 << 24) + 5
        far = torch.empty((m, 3), dtype=torch.float64, device=dev)
        far[:, 0], far[:, 1], far[:, 2] = 50.0, 50.0, 1.5
        far[m - 3000:] = torch.from_numpy(inl_small).to(dev)
        out = torch.zeros((60, 90), dtype=torch.uint8, device=dev)
        s.depth_preview_dev(far.data_ptr(), m, K, 60, 90, out.data_ptr())
        s.synchronize()
        far_h = np.empty((3005, 3))  # the oracle on the tail only (+ 5 outside points): same image, the others never touch a pixel
        far_h[:5] = (50.0, 50.0, 1.5)
        far_h[5:] = inl_small
        assert np.array_equal(out.cpu().numpy(), oracle.depth_preview(far_h, *K, 60, 90))
        assert np.array_equal(s.depth_preview(inl_small, K, 60, 90), pv_small)


@pytest.mark.parametrize("rows,cols,off", [(3, 3, 1), (40, 61, 1), (40, 61, 3), (5, 4, 2), (720, 1280, 1)])
def test_interpolate_equals_oracle(oracle, rsdsfm, rows, cols, off):
    rng = np.random.default_rng(rows + cols + off)
    img = rng.integers(0, 256, size=(rows, cols, 3), dtype=np.uint8)
    img[rng.random((rows, cols)) < 0.4] = rng.integers(0, 10, size=3, dtype=np.uint8)  # many black pixels, runs of them
    with rsdsfm.Solver(0) as s:
        assert np.array_equal(s.interpolate_cracky(img, off), oracle.interpolate_cracky(img, off))


def test_preview_equals_oracle_and_device_chain(oracle, rsdsfm):
    """host API on random inliers (collisions, out-of-image points, negative depths), then the device chain
    solve_frame_dev -> depth_preview_dev / back_project_dev / interpolate_cracky_dev on buffers that never leave HBM"""
    import torch

    rng = np.random.default_rng(1)
    K = (300.0, 310.0, 100.0, 75.0)
    rows, cols = 150, 200
    inl = np.column_stack([rng.uniform(-0.4, 0.4, 30000), rng.uniform(-0.3, 0.3, 30000), rng.normal(2.0, 1.5, 30000)])
    with rsdsfm.Solver(0) as s:
        assert np.array_equal(s.depth_preview(inl, K, rows, cols), oracle.depth_preview(inl, *K, rows, cols))
        assert np.array_equal(s.depth_preview(inl[:1], K, rows, cols), oracle.depth_preview(inl[:1], *K, rows, cols))
        assert np.array_equal(s.depth_preview(np.zeros((0, 3)), K, rows, cols), np.zeros((rows, cols), dtype=np.uint8))
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    with torch.cuda.stream(stream):
        d = rsdsfm.synth.make_config(3, rows=144, cols=256)
        rows, cols, K, gamma = d["rows"], d["cols"], d["K"], d["gamma"]
        img = rng.integers(16, 256, size=(rows, cols, 3), dtype=np.uint8)
        t_img = torch.from_numpy(img).to(dev)
        flow = torch.from_numpy(d["flow_img"]).to(dev)
        dm = torch.empty((cols, rows), dtype=torch.float64, device=dev)
        R = torch.empty((rows, 9), dtype=torch.float64, device=dev)
        t = torch.empty((rows, 3), dtype=torch.float64, device=dev)
        gs = torch.empty((rows, cols, 3), dtype=torch.uint8, device=dev)
        fixed = torch.empty_like(gs)
        c3 = torch.empty((rows, cols, 3), dtype=torch.float32, device=dev)
        prev = torch.empty((rows, cols), dtype=torch.uint8, device=dev)
        with rsdsfm.Solver(0, stream=stream.cuda_stream) as s:
            r = s.solve_frame_dev(flow.data_ptr(), rows, cols, K, gamma, dm.data_ptr(), R.data_ptr(), t.data_ptr(), trials=10, tol=0.002, seed=4)
            s.depth_preview_dev(r["d_inliers"], r["num_inliers"], K, rows, cols, prev.data_ptr())
            s.back_project_dev(t_img.data_ptr(), dm.data_ptr(), R.data_ptr(), t.data_ptr(), K, rows, cols, gs.data_ptr(), c3.data_ptr())
            s.interpolate_cracky_dev(gs.data_ptr(), rows, cols, fixed.data_ptr(), offset=1)
            s.synchronize()
            # the same three stages in ONE call (rsdsfm_rectify_frame_dev: claims in one launch, writes in one launch): the same bytes
            gs2, fixed2, c32, prev2 = torch.zeros_like(gs), torch.zeros_like(fixed), torch.zeros_like(c3), torch.zeros_like(prev)
            for rep in range(2):  # (twice: the claim maps' epochs advance)
                s.rectify_frame_dev(r["d_inliers"], r["num_inliers"], t_img.data_ptr(), dm.data_ptr(), R.data_ptr(), t.data_ptr(), K, rows, cols,
                                    prev2.data_ptr(), gs2.data_ptr(), fixed2.data_ptr(), c32.data_ptr(), offset=1)
            s.synchronize()
            assert torch.equal(gs2, gs) and torch.equal(fixed2, fixed) and torch.equal(prev2, prev) and torch.equal(c32.view(torch.int32), c3.view(torch.int32))
            with pytest.raises(rsdsfm.RsdsfmError):
                s.rectify_frame_dev(r["d_inliers"], r["num_inliers"], t_img.data_ptr(), dm.data_ptr(), R.data_ptr(), t.data_ptr(), K, rows, cols,
                                    prev2.data_ptr(), gs2.data_ptr(), gs2.data_ptr(), offset=1)  # gs and fixed alias
            m = r["num_inliers"]
            inl_t = torch.empty(3 * m, dtype=torch.float64, device=dev)
            import ctypes

            hip = ctypes.CDLL("libamdhip64.so")
            assert hip.hipMemcpy(ctypes.c_void_p(inl_t.data_ptr()), ctypes.c_void_p(r["d_inliers"]), ctypes.c_size_t(24 * m), 3) == 0
            inl_h = inl_t.cpu().numpy().reshape(m, 3)
        dm_h = dm.cpu().numpy().T
        gs_o, c3_o = oracle.back_project(img, dm_h, R.cpu().numpy(), t.cpu().numpy(), *K)
        assert np.array_equal(gs.cpu().numpy(), gs_o) and np.array_equal(c3.cpu().numpy().view(np.uint32), c3_o.view(np.uint32))
        assert np.array_equal(fixed.cpu().numpy(), oracle.interpolate_cracky(gs_o, 1))
        assert np.array_equal(prev.cpu().numpy(), oracle.depth_preview(inl_h, *K, rows, cols))
        assert (prev.cpu().numpy() != 0).sum() == (dm_h != 0).sum()


@pytest.mark.parametrize("rows,cols,m", [(33, 70, 5000), (16, 64, 0), (150, 200, 40000), (720, 1280, 921600)])
def test_rectify_frame_one_call_equals_the_oracle(oracle, rsdsfm, rows, cols, m):
    """rsdsfm_rectify_frame_dev on ragged sizes, no inliers, colliding inliers and the full 1280x720 frame: every output equals the oracle's
    (depth image, global-shutter image, float3 world points, interpolated image: bytes / bits)"""
    import torch

    rng = np.random.default_rng(rows * 7 + cols)
    dev = torch.device("cuda", 0)
    K = (0.8 * cols, 0.8 * cols, cols / 2.0 - 0.3, rows / 2.0 + 0.2)
    img = rng.integers(16, 256, size=(rows, cols, 3), dtype=np.uint8)
    img[rng.random((rows, cols)) < 0.02] = 1  # marker pixels (rsframe.cc:816)
    depth = rng.uniform(0.6, 2.5, size=(rows, cols))
    inl = np.column_stack([rng.uniform(-0.7, 0.7, m), rng.uniform(-0.45, 0.45, m), rng.normal(2.0, 1.0, m)]) if m else np.zeros((0, 3))
    with rsdsfm.Solver(0) as s:
        R, t = s.pose_table(np.array([0.3, -0.2, 0.1]), np.array([0.02, 0.03, -0.04]), 0.1, 0.9, rows)
        R = R.reshape(rows, 9)
        tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        d_img, d_dm, d_R, d_t, d_inl = tt(img), tt(depth.T), tt(R), tt(t), tt(inl if m else np.zeros((1, 3)))
        prev = torch.zeros((rows, cols), dtype=torch.uint8, device=dev)
        gs, fixed = torch.zeros((rows, cols, 3), dtype=torch.uint8, device=dev), torch.zeros((rows, cols, 3), dtype=torch.uint8, device=dev)
        c3 = torch.zeros((rows, cols, 3), dtype=torch.float32, device=dev)
        for off in (1, 2):
            s.rectify_frame_dev(d_inl.data_ptr(), m, d_img.data_ptr(), d_dm.data_ptr(), d_R.data_ptr(), d_t.data_ptr(), K, rows, cols, prev.data_ptr(),
                                gs.data_ptr(), fixed.data_ptr(), c3.data_ptr(), offset=off)
            s.synchronize()
            gs_o, c3_o = oracle.back_project(img, depth, R, t, *K)
            assert np.array_equal(gs.cpu().numpy(), gs_o) and np.array_equal(c3.cpu().numpy().view(np.uint32), c3_o.view(np.uint32))
            assert np.array_equal(fixed.cpu().numpy(), oracle.interpolate_cracky(gs_o, off))
            assert np.array_equal(prev.cpu().numpy(), oracle.depth_preview(inl, *K, rows, cols))


def test_rectify_argument_errors(rsdsfm):
    img = np.zeros((4, 4, 3), dtype=np.uint8)
    R, t = np.tile(np.eye(3).reshape(1, 9), (4, 1)), np.zeros((4, 3))
    with rsdsfm.Solver(0) as s:
        with pytest.raises(rsdsfm.RsdsfmError):
            s.back_project(img, np.ones((4, 4)), R, t, (1.0, 1.0, 2.0, 2.0), mode=2)
        with pytest.raises(rsdsfm.RsdsfmError):
            s.back_project(img, np.ones((4, 4)), R, t, (1.0, 1.0, 2.0, 2.0), q5_mode=7)
        with pytest.raises(rsdsfm.RsdsfmError):
            s.interpolate_cracky(img, -1)
        e = np.zeros((0, 5, 3), dtype=np.uint8)
        assert s.interpolate_cracky(e, 1).shape == (0, 5, 3)


# ---------------------------------------------------------------------------------------------------
# ground-truth flow search
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", RECTIFY_CASES)
def test_true_flow_matches_golden(golden_rectify, rsdsfm, case):
    g = lambda k: golden_rectify[case + "/" + k]
    K = tuple(g("K"))
    with rsdsfm.Solver(0) as s:
        for q5 in (0, 1):
            flow, best = s.true_flow(g("world"), g("R2"), g("t2"), K, q5_mode=q5)
            assert np.array_equal(best, g("tf_best_q%d" % q5))
            assert np.allclose(flow, g("tf_flow_q%d" % q5), rtol=1e-12, atol=1e-11)


def _flow_scene(rsdsfm, oracle, rows, cols, seed, rows2=None):
    d = rsdsfm.synth.make_config(1, rows=rows, cols=cols)
    fx, fy, cx, cy = d["K"]
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:rows, 0:cols]
    Z = np.array(d["truth"]["Z"])
    world = np.stack([(xx - cx) / fx, (yy - cy) / fy, np.ones((rows, cols))], axis=2) * Z[:, :, None]
    world[rng.random((rows, cols)) < 0.1] = 0.0
    rows2 = rows if rows2 is None else rows2
    R2, t2 = oracle.pose_table(np.array([0.12, 0.10, 0.05]), np.array([0.03, -0.02, 0.06]), 0.3, d["gamma"], rows2)
    t2 = t2 + np.array([0.04, 0.02, 0.01])
    return d["K"], world, R2, t2


@pytest.mark.parametrize("rows,cols,rows2", [(1, 1, 1), (5, 7, 5), (33, 65, 33), (64, 50, 40), (120, 200, 120), (240, 320, 240)])
def test_true_flow_equals_oracle(oracle, rsdsfm, rows, cols, rows2):
    """winning scanlines and flows BIT-exact (same per-projection operation chain, no contraction), incl. a frame 2 with
    a different number of scanlines and pixels whose best scanline is the first / last one"""
    K, world, R2, t2 = _flow_scene(rsdsfm, oracle, rows, cols, seed=rows + cols, rows2=rows2)
    with rsdsfm.Solver(0) as s:
        for q5 in (0, 1):
            flow, best = s.true_flow(world, R2, t2, K, q5_mode=q5)
            flow_o, best_o = oracle.true_flow(world, R2, t2, *K, q5_mode=q5)
            assert np.array_equal(best, best_o)
            assert np.array_equal(flow.view(np.uint64), flow_o.view(np.uint64))
        f2, none = s.true_flow(world, R2, t2, K, want_best_row=False)
        assert none is None and np.array_equal(f2, oracle.true_flow(world, R2, t2, *K)[0])


def test_true_flow_degenerate_points_and_errors(oracle, rsdsfm):
    K = (50.0, 50.0, 13.0, 10.0)
    rows = 12
    R = np.tile(np.eye(3), (rows, 1, 1))
    t = np.zeros((rows, 3))
    w = np.zeros((2, 3, 3))
    w[0, 0] = [0.1, 0.2, 0.0]       # on the principal plane: every displacement is inf
    w[0, 1] = [0.1, 0.2, -3.0]      # behind the camera
    w[0, 2] = [0.0, 0.0, 0.0]       # void
    w[1, 0] = [1e-200, 0.0, 0.0]    # ||W||^2 underflows to 0: treated as void (Eigen's norm())
    w[1, 1] = [0.0, (4.5 - K[3]) / K[1] * 4.0, 4.0]  # tie between scanlines 4 and 5: 4 wins
    w[1, 2] = [np.nan, 1.0, 2.0]
    with rsdsfm.Solver(0) as s:
        flow, best = s.true_flow(w, R, t, K, q5_mode=1)
        flow_o, best_o = oracle.true_flow(w, R, t, *K, q5_mode=1)
        assert np.array_equal(best, best_o) and best[1, 1] == 4 and best[0, 2] == -1 and best[1, 0] == -1
        assert np.array_equal(np.isnan(flow), np.isnan(flow_o))
        fin = np.isfinite(flow_o)
        assert np.array_equal(flow[fin], flow_o[fin])
        with pytest.raises(rsdsfm.RsdsfmError):
            s.true_flow(w, R[:0], t[:0], K)
        with pytest.raises(rsdsfm.RsdsfmError):
            s.true_flow(w, R, t, K, q5_mode=3)
        with pytest.raises(rsdsfm.RsdsfmError):
            s.set_true_flow_search(3)
        s.set_true_flow_search(2)  # the pruned search on the same degenerate points (12 scanlines: one block)
        flow2, best2 = s.true_flow(w, R, t, K, q5_mode=1)
        assert np.array_equal(best2, best) and np.array_equal(flow2.view(np.uint64), flow.view(np.uint64))


def test_true_flow_pruned_search_equals_exhaustive(oracle, rsdsfm):
    """the interval-pruned search (default, frames with >= 96 scanlines) picks the exhaustive loop's winners and flows bit for bit:
    a smooth pose table (few blocks evaluated), wild ones (rotations / translations jumping from scanline to scanline: nothing can
    be skipped, scanlines far from the pixel's own row win), z intervals containing 0, ties between blocks, non-finite table entries
    and world points, a last block of one scanline"""
    rng = np.random.default_rng(4242)
    rows, cols = 150, 97
    cases = []
    for rows2, kind in ((150, "smooth"), (97, "smooth"), (129, "wild"), (200, "wild_small"), (160, "flat"), (128, "nan")):
        K, world, R2, t2 = _flow_scene(rsdsfm, oracle, rows, cols, seed=rows2, rows2=rows2)
        R2, t2 = np.array(R2).reshape(rows2, 3, 3).copy(), np.array(t2).reshape(rows2, 3).copy()
        if kind == "wild":
            R2 += rng.normal(scale=0.5, size=R2.shape)
            t2 += rng.normal(scale=2.0, size=t2.shape)  # camera-frame z of many points changes sign from scanline to scanline
        elif kind == "wild_small":
            R2 += rng.normal(scale=0.02, size=R2.shape)
            t2 += rng.normal(scale=0.05, size=t2.shape)
        elif kind == "flat":  # identical poses: y(i) is constant, |y - i| ties exactly between neighbouring blocks for half-integer rows
            R2[:] = np.eye(3)
            t2[:] = 0.0
            world[::3, :, 1] = ((np.arange(0, rows, 3)[:, None] * 1.0 + 31.5 - K[3]) / K[1]) * world[::3, :, 2]
        elif kind == "nan":
            R2[40, 1, 1] = np.nan
            t2[77, 2] = np.inf
            world[5, 5] = [np.nan, 1.0, 2.0]
            world[6, 6] = [1e300, -1e300, 1e-300]
        cases.append((K, world, R2, t2))
    with rsdsfm.Solver(0) as s:
        for K, world, R2, t2 in cases:
            for q5 in (0, 1):
                s.set_true_flow_search(2)
                flow, best = s.true_flow(world, R2, t2, K, q5_mode=q5)
                s.set_true_flow_search(1)
                flow_x, best_x = s.true_flow(world, R2, t2, K, q5_mode=q5)
                assert np.array_equal(best, best_x)
                assert np.array_equal(flow.view(np.uint64), flow_x.view(np.uint64))
        for trial in range(24):  # random perturbation levels between "smooth" and "wild", random frame-2 heights
            rows2 = int(rng.integers(96, 320))
            K, world, R2, t2 = _flow_scene(rsdsfm, oracle, 64, 80, seed=1000 + trial, rows2=rows2)
            R2, t2 = np.array(R2).reshape(rows2, 3, 3).copy(), np.array(t2).reshape(rows2, 3).copy()
            scale = 10.0 ** rng.uniform(-5, 0)
            R2 += rng.normal(scale=scale, size=R2.shape)
            t2 += rng.normal(scale=3.0 * scale, size=t2.shape)
            if trial % 4 == 0:  # a drift: the winners sit far from the pixel's own row
                t2[:, 1] += np.linspace(0.0, rng.uniform(-1.0, 1.0), rows2)
            s.set_true_flow_search(2)
            flow, best = s.true_flow(world, R2, t2, K)
            s.set_true_flow_search(1)
            flow_x, best_x = s.true_flow(world, R2, t2, K)
            assert np.array_equal(best, best_x), (trial, scale)
            assert np.array_equal(flow.view(np.uint64), flow_x.view(np.uint64)), (trial, scale)
        K, world, R2, t2 = cases[2]
        with np.errstate(all="ignore"):
            flow_o, best_o = oracle.true_flow(world, R2, t2, *K)
        s.set_true_flow_search(0)
        flow, best = s.true_flow(world, R2, t2, K)
        assert np.array_equal(best, best_o) and np.array_equal(flow.view(np.uint64), flow_o.view(np.uint64))
        assert len(np.unique(best)) > 50  # the winners are all over the frame: the pruning has nothing to hold on to, the result is still exact


# ---------------------------------------------------------------------------------------------------
# accuracy metrics
# ---------------------------------------------------------------------------------------------------
def _close_stats(a, b):
    assert (a["number_outliers"], a["scale_inliers"], a["error_inliers"]) == (b["number_outliers"], b["scale_inliers"], b["error_inliers"])
    for key in ("scale", "mean_error", "sum_error"):  # global sums: summation order only
        assert np.isclose(a[key], b[key], rtol=1e-11, equal_nan=True), key


@pytest.mark.parametrize("case", RECTIFY_CASES)
def test_metrics_match_golden(golden_rectify, rsdsfm, case):
    g = lambda k: golden_rectify[case + "/" + k]
    with rsdsfm.Solver(0) as s:
        st, img = s.reprojection_error(g("est_coords"), g("gt_depth"), g("depth"), g("R_abs"), g("t_abs"), tuple(g("K")), max_norm=10.0)
    ref = g("reproj_stats")
    _close_stats(st, dict(scale=ref[0], mean_error=ref[1], sum_error=ref[2], number_outliers=int(ref[3]), scale_inliers=int(ref[4]), error_inliers=int(ref[5])))
    assert np.array_equal(img, g("error_image"))
    w, v = g("w"), g("v")
    assert np.allclose(rsdsfm.velocity_errors(w * 0.97, v * 1.05 + np.array([0.01, 0, 0]), w, v), g("vel_errors"), rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize("rows,cols", [(1, 1), (17, 15), (64, 48), (250, 333), (720, 1280)])
def test_metrics_equal_oracle(oracle, rsdsfm, rows, cols):
    d, img, depth, R, t = _scene(rsdsfm, oracle, rows, cols, seed=rows + 3 * cols)
    K = d["K"]
    _, c3 = oracle.back_project(img, depth, R, t, *K)
    est = (c3.astype(np.float64) * 1.3).astype(np.float32)
    rng = np.random.default_rng(rows)
    est += (rng.normal(0, 0.02, est.shape) * (rng.random(est.shape) < 0.5)).astype(np.float32)
    est[rng.random((rows, cols)) < 0.02] *= 40.0  # gross outliers (|ratio| > 10)
    gt = np.array(d["truth"]["Z"])
    gt[rng.random((rows, cols)) < 0.03] = 0.0
    Ra, ta = oracle.pose_table(np.array([0.11, 0.10, 0.06]), np.array([0.03, -0.02, 0.05]), 0.0, d["gamma"], rows)
    with rsdsfm.Solver(0) as s:
        st, eimg = s.reprojection_error(est, gt, depth, Ra, ta, K, max_norm=4.0)
        st_n, none = s.reprojection_error(est, gt, depth, Ra, ta, K, want_image=False)
    st_o, eimg_o = oracle.reprojection_error(est, gt, depth, Ra, ta, *K, max_norm=4.0)
    _close_stats(st, st_o)
    _close_stats(st_n, st_o)
    assert none is None
    # the image depends on the scale, whose last bits depend on the summation order: bytes may differ only where the
    # value sits within 1e-9 of a rounding boundary (none in practice)
    diff = eimg != eimg_o
    assert diff.mean() < 1e-5


def test_metrics_degenerate(oracle, rsdsfm):
    rows, cols = 6, 5
    z = np.zeros((rows, cols))
    R, t = np.tile(np.eye(3), (rows, 1, 1)), np.zeros((rows, 3))
    with rsdsfm.Solver(0) as s:
        st, img = s.reprojection_error(np.zeros((rows, cols, 3), dtype=np.float32), z + 1.0, z + 1.0, R, t, (10.0, 10.0, 2.0, 3.0))
        assert st["scale_inliers"] == 0 and np.isnan(st["scale"]) and np.isnan(st["mean_error"]) and st["error_inliers"] == 0
        assert np.all(img == 0)  # NaN errors -> 0 (defined here; undefined in the reference)
        st0, _ = s.reprojection_error(np.zeros((0, 4, 3), dtype=np.float32), np.zeros((0, 4)), np.zeros((0, 4)), R[:0], t[:0], (10.0, 10.0, 2.0, 3.0))
        assert np.isnan(st0["mean_error"])


def test_consumers_at_3840x2160(oracle, rsdsfm):
    """BASELINE configs[3] size: 64-bit indexing, grids beyond one wave of workgroups, int32 owner arrays at 8.3 M pixels --
    back projection / interpolation / depth image bit-exact, metrics to summation order, and the flow search against a
    frame 2 with 48 scanlines (the full 2160-scanline search is 1.8e10 projections: GPU-only, checked through its
    size-independent property below)"""
    rows, cols = 2160, 3840
    d, img, depth, R, t = _scene(rsdsfm, oracle, rows, cols, seed=42, cfg=4)
    K = d["K"]
    with rsdsfm.Solver(0) as s:
        gs, c3 = s.back_project(img, depth, R, t, K)
        gs_o, c3_o = oracle.back_project(img, depth, R, t, *K)
        assert np.array_equal(gs, gs_o) and np.array_equal(c3.view(np.uint32), c3_o.view(np.uint32))
        assert np.array_equal(s.interpolate_cracky(gs, 1), oracle.interpolate_cracky(gs_o, 1))
        yy, xx = np.mgrid[0:rows, 0:cols]
        nz = depth != 0
        inl = np.column_stack([((xx - K[2]) / K[0])[nz], ((yy - K[3]) / K[1])[nz], depth[nz]])
        assert np.array_equal(s.depth_preview(inl, K, rows, cols), oracle.depth_preview(inl, *K, rows, cols))
        Ra, ta = oracle.pose_table(np.array([0.11, 0.10, 0.06]), np.array([0.03, -0.02, 0.05]), 0.0, d["gamma"], rows)
        st, _ = s.reprojection_error(c3, np.array(d["truth"]["Z"]), depth, Ra, ta, K, max_norm=4.0)
        st_o, _ = oracle.reprojection_error(c3_o, np.array(d["truth"]["Z"]), depth, Ra, ta, *K, max_norm=4.0)
        _close_stats(st, st_o)
        world = np.stack([(xx - K[2]) / K[0], (yy - K[3]) / K[1], np.ones((rows, cols))], axis=2) * np.where(nz, depth, 0.0)[:, :, None]
        R2, t2 = oracle.pose_table(np.array([0.12, 0.10, 0.05]), np.array([0.03, -0.02, 0.06]), 0.3, d["gamma"], 48)
        flow, best = s.true_flow(world, R2, t2 + 0.01, K)
        flow_o, best_o = oracle.true_flow(world, R2, t2 + 0.01, *K)
        assert np.array_equal(best, best_o) and np.array_equal(flow.view(np.uint64), flow_o.view(np.uint64))
        # full-size search, size-independent property: a static camera (identity poses for all 2160 scanlines) gives zero
        # flow and every pixel's winner is its own row
        Ri, ti = np.tile(np.eye(3), (rows, 1, 1)), np.zeros((rows, 3))
        sub = world[:, ::16]  # 240 columns x 2160 rows x 2160 scanlines = 1.1e9 projections
        f0, b0 = s.true_flow(sub, Ri, ti, K, q5_mode=1)
        m = b0 >= 0
        # column j of `sub` holds the point seen at image column 16 j: the flow is (16 j - j, 0)
        assert np.abs(f0[:, :, 1][m]).max() < 1e-9 and np.abs(f0[:, :, 0] - 15.0 * np.arange(sub.shape[1])[None, :])[m].max() < 1e-9
        assert np.array_equal(b0[m], np.tile(np.arange(rows)[:, None], (1, sub.shape[1]))[m])
