"""CPU: the oracle's RS -> GS back projection, crack interpolation and 8-bit depth image against the
committed fixtures of the independent numpy transcription (tests/golden/make_golden_rectify.py) and against
size-independent properties."""
import numpy as np
import pytest

from conftest import RECTIFY_CASES


@pytest.mark.parametrize("case", RECTIFY_CASES)
def test_back_project_matches_golden(golden_rectify, oracle, case):
    g = lambda k: golden_rectify[case + "/" + k]
    K = tuple(g("K"))
    for mode in (0, 1):
        for q5 in (0, 1):
            gs, c3 = oracle.back_project(g("image"), g("depth"), g("R"), g("t"), *K, mode=mode, q5_mode=q5)
            assert np.array_equal(gs, g("gs_m%d_q%d" % (mode, q5))), (mode, q5)  # bytes: bit-exact
            # world points: float32 of the same chain up to the summation order of the 4x4 products
            assert np.allclose(c3, g("c3_m%d" % mode), rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("case", RECTIFY_CASES)
def test_interpolate_and_preview_match_golden(golden_rectify, oracle, case):
    g = lambda k: golden_rectify[case + "/" + k]
    for off in (1, 2):
        assert np.array_equal(oracle.interpolate_cracky(g("gs_m0_q0"), off), g("interp_off%d" % off))
    rows, cols = g("depth").shape
    assert np.array_equal(oracle.depth_preview(g("inliers"), *tuple(g("K")), rows, cols), g("preview"))


def test_pose_table_of_fixture_is_the_oracles(golden_rectify, oracle):
    for case in RECTIFY_CASES:
        g = lambda k: golden_rectify[case + "/" + k]
        R, t = oracle.pose_table(g("v"), g("w"), float(g("k")), float(g("gamma")), g("R").shape[0])
        assert np.allclose(R, g("R"), rtol=1e-15, atol=1e-18) and np.allclose(t, g("t"), rtol=1e-15, atol=1e-18)


def test_back_project_properties(oracle, rsdsfm):
    """(1) global-shutter mode with the identity pose is the identity on pixels with depth; (2) zero motion makes the
    rolling-shutter mode the identity too; (3) marker pixels and depth holes of scanline 0 never land; (4) the last
    writer wins: reversing which of two colliding pixels comes later changes the winner"""
    rows, cols = 30, 44
    d = rsdsfm.synth.make_config(1, rows=rows, cols=cols)
    K = d["K"]
    rng = np.random.default_rng(3)
    img = rng.integers(20, 255, size=(rows, cols, 3), dtype=np.uint8)
    img[4, 5] = (1, 1, 1)
    depth = np.array(d["truth"]["Z"])
    depth[0, 7] = 0.0  # scanline 0 has t = 0: 0/0 -> skipped
    R0, t0 = oracle.pose_table(np.zeros(3), np.zeros(3), 0.0, d["gamma"], rows)
    for mode in (0, 1):
        gs, c3 = oracle.back_project(img, depth, R0, t0, *K, mode=mode, q5_mode=1)
        expect = img.copy()
        expect[4, 5] = 0
        expect[0, 7] = 0
        assert np.array_equal(gs, expect)
        assert np.all(c3[4, 5] == 0)
        # world point = z * (normalised pixel, 1)
        assert np.allclose(c3[10, 20], depth[10, 20] * np.array([(20 - K[2]) / K[0], (10 - K[3]) / K[1], 1.0]), rtol=1e-6)
    # Q5: with f_x != f_y the compat mode scales y by f_x
    K2 = (K[0], K[1] * 1.25, K[2], K[3])
    gs_c, _ = oracle.back_project(img, depth, R0, t0, *K2, mode=1, q5_mode=0)
    gs_f, _ = oracle.back_project(img, depth, R0, t0, *K2, mode=1, q5_mode=1)
    assert not np.array_equal(gs_c, gs_f)
    # last writer wins: a pure image-plane shift by one row (v_y chosen so that every scanline maps one row up is not
    # constructible exactly; instead collide two pixels through a depth edge)
    v, w = np.array([0.0, 0.3, 0.0]), np.zeros(3)
    R, t = oracle.pose_table(v, w, 0.0, d["gamma"], rows)
    flat = np.full((rows, cols), 2.0)
    gs, _ = oracle.back_project(img, flat, R, t, *K, mode=0, q5_mode=1)
    # reference semantics by brute force (sequential overwrite)
    exp = np.zeros_like(img)
    fx, fy, cx, cy = K
    for y in range(rows):
        for x in range(cols):
            if tuple(img[y, x]) == (1, 1, 1):
                continue
            pc = 2.0 * np.array([(x - cx) / fx, (y - cy) / fy, 1.0])
            pw = R[y].T @ pc - R[y].T @ t[y]
            gx, gy = pw[0] / pw[2] * fx + cx, pw[1] / pw[2] * fy + cy
            ix, iy = int(np.trunc(gx + 0.5)), int(np.trunc(gy + 0.5))
            if 0 <= ix < cols and 0 <= iy < rows:
                exp[iy, ix] = img[y, x]
    assert np.array_equal(gs, exp)
    assert (gs.reshape(-1, 3).sum(axis=1) == 0).sum() > cols  # the vertical motion leaves uncovered rows (cracks)


def test_interpolate_properties(oracle):
    rng = np.random.default_rng(9)
    img = rng.integers(30, 255, size=(20, 25, 3), dtype=np.uint8)
    assert np.array_equal(oracle.interpolate_cracky(img, 1), img)  # no black pixel: untouched
    img2 = img.copy()
    img2[10, 12] = 0
    img2[0, 3] = 0  # border pixel: never touched
    img2[5, 5] = (9, 9, 8)  # norm 15.03 > 15: not black
    img2[6, 6] = (9, 9, 7)  # norm 14.5: black
    out = oracle.interpolate_cracky(img2, 1)
    nb = img[[9, 11, 10, 10], [12, 12, 11, 13]].astype(np.float64)
    assert np.array_equal(out[10, 12], np.clip(np.rint(0.25 * nb.sum(axis=0)), 0, 255).astype(np.uint8))
    assert np.all(out[0, 3] == 0) and np.array_equal(out[5, 5], img2[5, 5]) and not np.array_equal(out[6, 6], img2[6, 6])
    # ties round to even (cv::saturate_cast): two neighbours 10 and 11 -> 10.5 -> 10; 11 and 12 -> 11.5 -> 12
    t = np.zeros((3, 3, 3), dtype=np.uint8)
    t[0, 1] = (100, 11, 21)
    t[2, 1] = (101, 12, 20)
    o = oracle.interpolate_cracky(t, 1)
    assert tuple(o[1, 1]) == (100, 12, 20)
    # an all-black image stays black; offset larger than the image: nothing to do
    z = np.zeros((6, 6, 3), dtype=np.uint8)
    assert np.array_equal(oracle.interpolate_cracky(z, 1), z) and np.array_equal(oracle.interpolate_cracky(img, 30), img)


def test_preview_properties(oracle):
    K = (100.0, 100.0, 16.0, 12.0)
    rows, cols = 24, 32
    inl = np.array([[0.0, 0.0, 1.0], [0.01, 0.0, 3.0], [0.0, 0.0, 2.0], [5.0, 5.0, 9.0]])  # the last one is outside the image
    out = oracle.depth_preview(inl, *K, rows, cols)
    # z_max counts the out-of-image point too (the reference computes it over all inliers): multiplier = 244 / 8
    assert out[12, 16] == 10 + int((2.0 - 1.0) * 30.5) and out[12, 17] == 10 + int((3.0 - 1.0) * 30.5)
    assert (out != 0).sum() == 2
    # single inlier: z_max - z_min = 1 - 1 -> multiplier inf, (z - z_min) * inf = NaN -> defined as 0 -> value 10
    one = oracle.depth_preview(inl[:1], *K, rows, cols)
    assert one[12, 16] == 10 and (one != 0).sum() == 1
    # all-negative depths: z_max stays at its start value 0 (main.cc:482)
    neg = oracle.depth_preview(np.array([[0.0, 0.0, -2.0], [0.01, 0.0, -1.0]]), *K, rows, cols)
    assert neg[12, 16] == 10 and neg[12, 17] == 10 + int(1.0 * 122.0)
    assert np.array_equal(oracle.depth_preview(np.zeros((0, 3)), *K, rows, cols), np.zeros((rows, cols), dtype=np.uint8))


@pytest.mark.parametrize("case", RECTIFY_CASES)
def test_true_flow_matches_golden(golden_rectify, oracle, case):
    g = lambda k: golden_rectify[case + "/" + k]
    K = tuple(g("K"))
    for q5 in (0, 1):
        flow, best = oracle.true_flow(g("world"), g("R2"), g("t2"), *K, q5_mode=q5)
        assert np.array_equal(best, g("tf_best_q%d" % q5))  # winning scanlines: bit-exact
        assert np.allclose(flow, g("tf_flow_q%d" % q5), rtol=1e-12, atol=1e-11)  # 4x4 matmul summation order only


def test_true_flow_properties(oracle, rsdsfm):
    """(1) a static camera (identity poses) returns zero flow with winner = the pixel's own row; (2) void pixels give
    zero flow and winner -1; (3) a pure image-plane shift is recovered; (4) ties keep the FIRST scanline; (5) a point
    behind / on the camera plane never crashes (non-finite displacements lose every comparison)"""
    rows, cols = 20, 26
    K = (50.0, 50.0, 13.0, 10.0)
    fx, fy, cx, cy = K
    yy, xx = np.mgrid[0:rows, 0:cols]
    Z = 2.0 + 0.1 * xx + 0.05 * yy
    world = np.stack([(xx - cx) / fx, (yy - cy) / fy, np.ones((rows, cols))], axis=2) * Z[:, :, None]
    world[3, 4] = 0.0
    R = np.tile(np.eye(3), (rows, 1, 1))
    t = np.zeros((rows, 3))
    flow, best = oracle.true_flow(world, R, t, *K, q5_mode=1)
    assert np.abs(flow).max() < 1e-12 and best[3, 4] == -1 and np.all(flow[3, 4] == 0)
    exp = np.tile(np.arange(rows)[:, None], (1, cols))
    exp[3, 4] = -1
    assert np.array_equal(best, exp)
    # every scanline has the same pose -> |py - i| is minimised at the row nearest to py; shift by t_y so that py moves
    t2 = t.copy()
    t2[:, 1] = 0.1
    flow2, best2 = oracle.true_flow(world, R, t2, *K, q5_mode=1)
    m = best2 >= 0
    with np.errstate(divide="ignore", invalid="ignore"):
        py = (world[:, :, 1] + 0.1) / world[:, :, 2] * fy + cy
    assert np.array_equal(best2[m], np.argmin(np.abs(py[:, :, None] - np.arange(rows)), axis=2)[m])  # argmin = first minimum
    assert np.allclose(flow2[:, :, 1][m], (py - yy)[m], atol=1e-12) and np.allclose(flow2[:, :, 0][m], 0, atol=1e-12)
    # tie: a point exactly half-way between two scanlines (py = 4.5) -> rows 4 and 5 are equally close, 4 wins
    w1 = np.zeros((1, 1, 3))
    w1[0, 0] = [0.0, (4.5 - cy) / fy * 4.0, 4.0]
    _, b = oracle.true_flow(w1, R, t, *K, q5_mode=1)
    assert b[0, 0] == 4
    # behind the camera / on the principal plane
    w2 = np.zeros((1, 2, 3))
    w2[0, 0] = [0.1, 0.2, 0.0]
    w2[0, 1] = [0.1, 0.2, -3.0]
    f, b = oracle.true_flow(w2, R, t, *K, q5_mode=1)
    assert b[0, 0] == 0 and not np.isfinite(f[0, 0]).all()  # z = 0: inf displacement everywhere, scanline 0 kept
    assert b[0, 1] >= 0 and np.isfinite(f[0, 1]).all()


@pytest.mark.parametrize("case", RECTIFY_CASES)
def test_metrics_match_golden(golden_rectify, oracle, case):
    g = lambda k: golden_rectify[case + "/" + k]
    st, img = oracle.reprojection_error(g("est_coords"), g("gt_depth"), g("depth"), g("R_abs"), g("t_abs"), *tuple(g("K")), max_norm=10.0)
    ref = g("reproj_stats")
    assert (st["number_outliers"], st["scale_inliers"], st["error_inliers"]) == tuple(int(v) for v in ref[3:6])  # counts: exact
    assert np.isclose(st["scale"], ref[0], rtol=1e-12) and np.isclose(st["mean_error"], ref[1], rtol=1e-12) and np.isclose(st["sum_error"], ref[2], rtol=1e-12)
    assert np.array_equal(img, g("error_image"))
    w, v = g("w"), g("v")
    we, ve = oracle.velocity_errors(w * 0.97, v * 1.05 + np.array([0.01, 0, 0]), w, v)
    assert np.allclose([we, ve], g("vel_errors"), rtol=1e-12, atol=1e-15)


def test_metrics_properties(oracle, rsdsfm):
    """a perfect estimate at a different global scale: scale recovered, zero error; identical velocities: zero errors;
    rotation error is first order in the difference; opposite translation: pi"""
    rows, cols = 18, 22
    d = rsdsfm.synth.make_config(1, rows=rows, cols=cols)
    K = d["K"]
    fx, fy, cx, cy = K
    Z = np.array(d["truth"]["Z"])
    R, t = oracle.pose_table(np.array([0.1, 0.05, 0.02]), np.array([0.02, -0.01, 0.03]), 0.0, d["gamma"], rows)
    img = np.full((rows, cols, 3), 100, dtype=np.uint8)
    _, c3 = oracle.back_project(img, Z, R, t, *K)  # world points under the same poses = the "truth" the metric rebuilds
    st, eimg = oracle.reprojection_error((c3.astype(np.float64) * 2.5).astype(np.float32), Z, Z, R, t, *K, max_norm=10.0)
    assert abs(st["scale"] - 2.5) < 1e-5 and st["mean_error"] < 1e-5 and st["number_outliers"] == 0
    assert st["error_inliers"] == rows * cols and int(eimg.max()) == 0
    st2, _ = oracle.reprojection_error(np.zeros((rows, cols, 3), dtype=np.float32), Z, Z, R, t, *K)
    assert st2["scale_inliers"] == 0 and np.isnan(st2["scale"]) and st2["error_inliers"] == 0 and np.isnan(st2["mean_error"])
    # ground-truth depth 0 -> the estimated depth map is used (planeToSpace default argument)
    Zg = Z.copy()
    Zg[5, 6] = 0.0
    st3, _ = oracle.reprojection_error(c3, Zg, Z, R, t, *K)
    assert st3["mean_error"] < 1e-5
    w, v = np.array([0.01, -0.02, 0.03]), np.array([0.3, 0.1, -0.2])
    # the reference composes FIRST-ORDER rotations (I + [w]x): identical velocities leave a second-order residue w_i w_j
    we0, ve0 = oracle.velocity_errors(w, v, w, v)
    assert we0 <= float(w @ w) and ve0 < 1e-7
    we, _ = oracle.velocity_errors(w + np.array([1e-2, 0, 0]), v, w, v)
    assert abs(we - 1e-2) < 2e-3
    assert abs(oracle.velocity_errors(w, -v, w, v)[1] - np.pi) < 1e-7
