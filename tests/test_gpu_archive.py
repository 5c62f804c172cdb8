"""GPU: an example archive in the reference's on-disk layout runs end to end through
evaluate.evaluate_single_run -- the reference's evaluateSingleRun for synthetic data (main.cc:364-560): ground-truth
flow, solve, depth image, back projection, crack interpolation, point cloud and the accuracy metrics -- and every
stage equals the oracle chain on the same files."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _write_archive(rsdsfm, oracle, task, rows=60, cols=96):
    """a rolling-shutter pair under constant motion: world frame = scanline-0 camera of frame 1; scanline i of frame n is
    at 'time' beta = n + gamma * i / rows: R = I + beta [w]x, t = beta v (the reference's setRelativePose convention)"""
    d = rsdsfm.synth.make_config(1, rows=rows, cols=cols)
    K, gamma = d["K"], d["gamma"]
    fx, fy, cx, cy = K
    # |v| ~ 0.5 world units per frame with depths ~ 60: the solver normalises v to unit length, so the estimated structure is
    # 1 / |v| = 2 x the truth -- inside the factor-10 window (with the 1 / gamma of the flow normalisation: 2.4 x) of the reference's scale estimate (camera.cc:643-654)
    v, w = np.array([0.4, 0.3, 0.08]), np.array([0.004, -0.003, 0.006])
    Z = np.array(d["truth"]["Z"]) * 60.0
    yy, xx = np.mgrid[0:rows, 0:cols]
    rng = np.random.default_rng(8)
    frames = []
    for n in range(2):
        beta = n + gamma * np.arange(rows) / rows
        R = np.eye(3)[None] + beta[:, None, None] * np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])[None]
        t = beta[:, None] * v[None]
        pc = np.stack([(xx - cx) / fx, (yy - cy) / fy, np.ones((rows, cols))], axis=2) * Z[:, :, None]
        world = np.einsum("yji,yxj->yxi", R, pc - t[:, None, :])  # R^T (P_cam - t), per scanline
        img = rng.integers(16, 256, (rows, cols, 3), dtype=np.uint8)
        gs = np.roll(img, n + 1, axis=1)  # stand-in for the archive's global-shutter rendering (N_initial_gs.png)
        frames.append(dict(rs_image=img, gs_image=gs, R=R, t=t, world=world))
    rsdsfm.formats.write_example_archive(task, K, gamma, v, w, 0.0, frames)
    return K, gamma, v, w, frames


def test_archive_end_to_end(rsdsfm, oracle, tmp_path):
    task, out_dir = str(tmp_path / "task_1"), str(tmp_path / "results")
    K, gamma, v, w, frames = _write_archive(rsdsfm, oracle, task)
    rows, cols = frames[0]["rs_image"].shape[:2]
    with rsdsfm.Solver(0) as s:
        # selective tolerance: the inlier set is a strict subset, so the reference's rank-indexed flow (quirk Q2, the default)
        # would pair inliers with other pixels' flow; this test checks the physics and uses each inlier's own flow
        r = rsdsfm.evaluate.evaluate_single_run(s, task, out_dir, trials=20, tol=0.002, seed=3, flow_index_mode=1)
    # ---- the oracle chain on the same files ----
    a = rsdsfm.formats.load_example_archive(task)
    f1, f2 = a["frames"]
    flow_o, _ = oracle.true_flow(f1["world"], f2["R"], f2["t"], *K)
    assert np.array_equal(r["flow"], flow_o)
    q, u, qpx, fpx = oracle.flatten(flow_o, *K, gamma)
    al, alk = oracle.get_alpha(fpx, rows, gamma), oracle.get_alpha_k(qpx, fpx, rows, gamma)
    ro = oracle.ransac(q, u, al, alk, False, 20, 0.002, oracle.sample_indices(len(q), 20, 3), depth_mode=1)
    assert r["n"] == len(q) and r["num_inliers"] == ro["num_inliers"]
    refo = oracle.refine(u, ro["inliers"], ro["alpha"], ro["alpha_k"], ro["v"], ro["w"], ro["k"], False, 1, ro["inlier_idx"])
    inl_o, v_o, flipped_o = oracle.canonicalize_sign(refo["inliers"], refo["v"])
    assert r["flipped"] == flipped_o and np.allclose(r["v"], v_o, rtol=1e-6, atol=1e-10) and np.allclose(r["w"], refo["w"], rtol=1e-6, atol=1e-10)
    dm_o, _, _ = oracle.scatter_depth(inl_o, *K, rows, cols)
    assert np.array_equal(r["depth_map"] != 0, dm_o != 0) and np.allclose(r["depth_map"], dm_o, rtol=1e-6)
    # downstream stages are compared on the GPU path's own depth map / pose (bit-exact stages)
    assert np.array_equal(r["depth_est"] != 0, dm_o != 0)
    R_rel, t_rel = oracle.pose_table(r["v"], r["w"], r["k"], gamma, rows)
    gs_o, c3_o = oracle.back_project(f1["rs_image"], r["depth_map"], R_rel, t_rel, *K)
    assert np.array_equal(r["gs_image"], gs_o) and np.array_equal(r["coords"].view(np.uint32), c3_o.view(np.uint32))
    assert np.array_equal(r["backprojection"], oracle.interpolate_cracky(gs_o, 1))
    gt_depth = rsdsfm.evaluate.gt_depth_map(f1["world"], f1["R"], f1["t"])
    R_abs, t_abs = rsdsfm.evaluate.relocate_pose(f1["R"], f1["t"])
    st_o, eimg_o = oracle.reprojection_error(c3_o, gt_depth, r["depth_map"], R_abs, t_abs, *K, max_norm=10.0)
    assert np.isclose(r["mean_reprojection_error"], st_o["mean_error"], rtol=1e-9) and r["reprojection"]["error_inliers"] == st_o["error_inliers"]
    assert st_o["error_inliers"] > 0.9 * r["num_inliers"] and abs(st_o["scale"] * gamma * np.linalg.norm(v) - 1.0) < 0.05  # flow is normalised by gamma (main.cc:423-426)
    assert r["mean_reprojection_error"] < 0.15 * 60.0  # differential model on geometric flow: within ~10 % of the scene depth
    assert (r["error_image"] != eimg_o).mean() < 1e-3
    # ---- the solve recovers the archive's motion (differential model on geometric flow: a few percent) ----
    vt = v / np.linalg.norm(v)
    vv = r["v"] / np.linalg.norm(r["v"])
    # velocities come out in units of the gamma-normalised flow (u = flow * gamma / f, main.cc:423-426): w_est ~ gamma * w
    assert float(vv @ vt) > 0.98 and np.linalg.norm(r["w"] - gamma * w) < 0.2 * np.linalg.norm(gamma * w), (r["v"], r["w"])
    # ---- products on disk, readable by the package's own readers ----
    for name in ("MinimalDepth.png", "rs_image.png", "backprojection.png", "error_image.png", "point_cloud.ply", "errors.csv", "w.csv", "v.csv", "k.csv"):
        assert os.path.exists(os.path.join(out_dir, name)), name
    assert np.array_equal(rsdsfm.formats.read_png(out_dir + "/backprojection.png"), r["backprojection"])
    assert np.array_equal(rsdsfm.formats.read_png(out_dir + "/MinimalDepth.png", grayscale=True), r["depth_est"])
    # the diagnostic images of main.cc:386-394, :533-554 (comparisons against the archive's global-shutter image)
    F = rsdsfm.formats
    rd = lambda name: F.read_png(out_dir + "/" + name)
    gs0, rs0 = f1["gs_image"], f1["rs_image"]
    assert np.array_equal(rd("gs_image.png"), gs0)
    diff = np.abs(r["backprojection"].astype(int) - gs0.astype(int)).astype(np.uint8)
    assert np.array_equal(rd("difference.png"), diff)
    assert np.array_equal(rd("remainder.png"), np.abs(gs0.astype(int) - diff.astype(int)).astype(np.uint8))
    assert np.array_equal(rd("overlay_gs_bp.png"), F.create_overlay_image(gs0, F.shift_channel_bgr(diff, 2, 0.5, 0.5)))
    assert np.array_equal(rd("overlay_gs_rs.png"), F.create_overlay_image(gs0, F.shift_channel_bgr(F.abs_diff(rs0, gs0), 2, 0.5, 0.5)))
    of = rd("optical_flow.png")
    assert of.shape == (rows, cols, 3) and of.max() == 255  # hue = direction, value = magnitude / max magnitude
    pc, col = rsdsfm.formats.read_ply(out_dir + "/point_cloud.ply")
    assert np.array_equal(pc, r["coords"].reshape(-1, 3)) and np.array_equal(col, f1["rs_image"].reshape(-1, 3))
    assert open(out_dir + "/errors.csv").read().startswith("task,error_w,error_v,reproject_error\ntask_1,")


def test_parameter_sweep_over_archives(rsdsfm, oracle, tmp_path):
    """main.cc:148-299: two tasks x four evaluations, spread over two solver contexts (threads); the same sweep on one
    context gives bit-identical numbers, the errors are small, and the reference's result files are written"""
    import torch

    root = tmp_path / "sweep"
    os.makedirs(root)
    tasks = ["task_a", "task_b"]
    for t in tasks:
        _write_archive(rsdsfm, oracle, str(root / t), rows=48 if t == "task_a" else 60, cols=80)
    open(root / "tasks.txt", "w").write("\n".join(tasks) + "\n")
    dev = torch.device("cuda", 0)
    streams = [torch.cuda.Stream(dev) for _ in range(2)]
    solvers = [rsdsfm.Solver(0, stream=st.cuda_stream) for st in streams]
    kw = dict(ransac_trials=12, num_evaluations=4, tol=0.002, base_seed=5, flow_index_mode=1)
    res2 = rsdsfm.evaluate.evaluate_parameter_sweep(solvers, str(root), str(tmp_path / "results2"), **kw)
    res1 = rsdsfm.evaluate.evaluate_parameter_sweep(solvers[0], str(root), str(tmp_path / "results1"), **kw)
    for s in solvers:
        s.close()
    for t in tasks:
        for key in ("w", "v", "k", "error_w_vec", "error_v_vec", "error_reproject_vec", "num_inliers"):
            assert np.array_equal(res1[t][key], res2[t][key]), (t, key)
        assert res1[t]["error_v"] < 0.2 and res1[t]["error_reproject"] < 0.15 * 60.0
        assert len(set(res1[t]["num_inliers"].tolist())) >= 1 and res1[t]["w"].shape == (4, 3)
    out = str(tmp_path / "results2")
    lines = open(out + "/errors.csv").read().strip().split("\n")
    assert lines[0] == "task,error_w,error_v,reproject_error" and [ln.split(",")[0] for ln in lines[1:]] == tasks
    for name in ("w.csv", "v.csv", "k.csv", "reproject_errors.csv", "error_v.csv", "error_w.csv", "depthMaps/0/0.png", "depthMaps/1/3.ply"):
        assert os.path.exists(os.path.join(out, name)), name
    assert len(open(out + "/w.csv").read().strip().split("\n")[0].split(",")) == 12  # 4 evaluations x 3 components per task line


def test_real_world_run_with_an_external_flow(rsdsfm, oracle, tmp_path):
    """evaluate_real_run = the real-world branch of evaluateSingleRun (main.cc:341-361, :675-690) with the flow passed in: frame1.png +
    a flow file -> one device-resident solve -> depth image, rectified frame, point cloud.  The device-resident consumers give the
    bytes of the host-boundary ones, the pose is the one the stage chain finds, the files are what the reference writes"""
    import os

    F = rsdsfm.formats
    d = rsdsfm.synth.make_config(3, rows=120, cols=160)
    rows, cols = d["rows"], d["cols"]
    rng = np.random.default_rng(12)
    image = rng.integers(0, 256, size=(rows, cols, 3), dtype=np.uint8)
    prefix = str(tmp_path) + "/"
    F.write_png(prefix + "frame1.png", image)
    np.save(prefix + "flow.npy", d["flow_img"])
    with rsdsfm.Solver(0) as s:
        out = rsdsfm.evaluate.evaluate_real_run(s, prefix, prefix + "flow.npy", camera=d["K"], gamma=d["gamma"], out_dir=prefix + "out", trials=8,
                                                 tol=0.002, seed=3, flow_index_mode=1)
        assert out["n"] == rows * cols and out["num_inliers"] > 0.1 * rows * cols
        # the same products through the host-boundary entry points
        gs, coords = s.back_project(image, out["depth_map"], out["R"], out["t"], d["K"])
        assert np.array_equal(gs, out["gs_image"]) and np.array_equal(coords.view(np.uint32), out["coords"].view(np.uint32))
        assert np.array_equal(s.interpolate_cracky(gs, 1), out["backprojection"])
        # the stage chain on the same flow and seed finds the same motion
        q, u, a, ak = s.flatten(d["flow_img"], d["K"], d["gamma"])
        rr = s.ransac(q, u, a, ak, False, 8, 0.002, samples=None, seed=3)
        ref = s.non_linear_refinement(u, rr["inliers"], rr["alpha"], rr["alpha_k"], rr["v"], rr["w"], rr["k"], False, flow_index_mode=1,
                                      inlier_idx=rr["inlier_idx"])
        dm = s.depth_map(ref["inliers"], ref["v"], d["K"], rows, cols)
        assert out["num_inliers"] == rr["num_inliers"] and np.allclose(out["w"], ref["w"], rtol=1e-9, atol=1e-12) and np.allclose(out["v"], dm["v"], rtol=1e-9)
        assert np.array_equal(out["depth_est"], s.depth_preview(dm["inliers"], d["K"], rows, cols))
        # a float32 .flo carrier of the same flow: the same pipeline, results to the carrier's precision
        F.write_flo(prefix + "flow.flo", d["flow_img"])
        out2 = rsdsfm.evaluate.evaluate_real_run(s, image, prefix + "flow.flo", camera=d["K"], gamma=d["gamma"], trials=8, tol=0.002, seed=3, flow_index_mode=1)
        assert np.allclose(out2["w"], out["w"], atol=1e-5) and abs(out2["num_inliers"] - out["num_inliers"]) < 0.01 * rows * cols
        with pytest.raises(ValueError):
            rsdsfm.evaluate.evaluate_real_run(s, image[:10], prefix + "flow.npy", camera="galaxy")
    for name in ("MinimalDepth.png", "rs_image.png", "backprojection.png", "point_cloud.ply"):
        assert os.path.getsize(prefix + "out/" + name) > 0
    assert np.array_equal(F.read_png(prefix + "out/backprojection.png"), out["backprojection"])
    pts, cols_ = F.read_ply(prefix + "out/point_cloud.ply")
    assert len(pts) == len(cols_) > 0
