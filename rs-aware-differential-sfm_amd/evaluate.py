"""evaluateSingleRun for a synthetic example archive (reference main.cc:364-560) on the HIP path: ground-truth flow ->
flatten + alpha -> ransac -> nonLinearRefinement -> sign fix + depth map + 8-bit depth image -> setRelativePose ->
backProject -> interpolateCrackyImage -> point cloud, images and (synthetic data) error image / mean reprojection error.

Host-side orchestration over the C-ABI wrappers (Solver); file formats in formats.py.  Image decoding / encoding and the
CSV parsing are host work like in the reference; everything per-pixel runs in the HIP kernels.
"""
import os

import numpy as np

from . import formats


def gt_depth_map(world, R_abs, t_abs):
    """RsFrame::getGroundtruthDepthMap (rsframe.cc:416-436): z of the world point in the camera frame of ITS scanline"""
    z = (R_abs[:, 2, :][:, None, :] * world).sum(axis=2) + t_abs[:, 2][:, None]
    return np.where(np.sqrt((world * world).sum(axis=2)) > 0, z, 0.0)


def relocate_pose(R_abs, t_abs):
    """RsFrame::relocatePose (rsframe.cc:951-967): scanline 0 becomes the origin (scanline 0 itself is left untouched)"""
    R, t = np.array(R_abs, dtype=np.float64), np.array(t_abs, dtype=np.float64)
    inv0 = np.linalg.inv(R[0])
    R[1:] = inv0 @ R[1:]
    t[1:] = t[1:] - t[0]
    return R, t


def evaluate_single_run(solver, task_dir, out_dir, trials=50, tol=0.05, seed=1, use_acceleration_mode=False, use_refinement=True,
                        use_global_shutter_mode=False, flow_threshold=1e-10, write_outputs=True, flow_index_mode=0):
    """evaluateSingleRun (main.cc:302-559) on a synthetic example archive.  flow_index_mode 0 (default) = the reference: the
    refinement reads the flow by inlier RANK (main.cc:457, quirk Q2); 1 = each inlier's own pixel."""
    from . import BACKPROJECT_GS, BACKPROJECT_RS, velocity_errors

    a = formats.load_example_archive(task_dir)
    K, truth = a["K"], a["truth"]
    gamma = truth["gamma"]
    f1, f2 = a["frames"]
    rows, cols = f1["rs_image"].shape[:2]
    # main.cc:383 calculateTrueFlow(1, 2)
    flow, _ = solver.true_flow(f1["world"], f2["R"], f2["t"], K, want_best_row=False)
    # main.cc:398-457
    q, u, alpha, alpha_k = solver.flatten(flow, K, gamma, thr=flow_threshold)
    if use_global_shutter_mode:  # main.cc:441-444
        alpha = alpha * 0.0 + 1.0
    rr = solver.ransac(q, u, alpha, alpha_k, use_acceleration_mode, trials, tol, samples=None, seed=seed)
    res = dict(v=rr["v"], w=rr["w"], k=rr["k"], inliers=rr["inliers"])
    if use_refinement:
        ref = solver.non_linear_refinement(u, rr["inliers"], rr["alpha"], rr["alpha_k"], rr["v"], rr["w"], rr["k"], use_acceleration_mode,
                                           flow_index_mode=flow_index_mode, inlier_idx=rr["inlier_idx"] if flow_index_mode else None)
        res = dict(v=ref["v"], w=ref["w"], k=ref["k"], inliers=ref["inliers"], refine_summary=ref["summary"])
    # main.cc:466-509
    dm = solver.depth_map(res["inliers"], res["v"], K, rows, cols)
    depth_est = solver.depth_preview(dm["inliers"], K, rows, cols)
    # main.cc:515-523
    R_rel, t_rel = solver.pose_table(dm["v"], res["w"], res["k"], gamma, rows)
    gs, coords = solver.back_project(f1["rs_image"], dm["depth_map"], R_rel, t_rel, K, mode=BACKPROJECT_GS if use_global_shutter_mode else BACKPROJECT_RS)
    backprojection = solver.interpolate_cracky(gs, 1)
    # synthetic data only (main.cc:533-556): error image + mean reprojection error against the archive's ground truth
    gt_depth = gt_depth_map(f1["world"], f1["R"], f1["t"])
    R_abs, t_abs = relocate_pose(f1["R"], f1["t"])
    stats, error_image = solver.reprojection_error(coords, gt_depth, dm["depth_map"], R_abs, t_abs, K, max_norm=10.0)
    w_err, v_err = velocity_errors(res["w"], dm["v"], truth["w"], truth["v"])
    out = dict(n=len(q), num_inliers=rr["num_inliers"], v=dm["v"], w=res["w"], k=res["k"], flipped=dm["flipped"], w_error=w_err, v_error=v_err,
               mean_reprojection_error=stats["mean_error"], reprojection=stats, flow=flow, depth_map=dm["depth_map"], depth_est=depth_est,
               gs_image=gs, backprojection=backprojection, coords=coords, error_image=error_image, truth=truth)
    if write_outputs:
        os.makedirs(out_dir, exist_ok=True)
        formats.write_png(out_dir + "/MinimalDepth.png", depth_est)
        formats.write_png(out_dir + "/rs_image.png", f1["rs_image"])
        formats.write_png(out_dir + "/backprojection.png", backprojection)
        formats.write_png(out_dir + "/error_image.png", error_image)
        formats.write_png(out_dir + "/optical_flow.png", formats.flow_to_bgr(flow))  # main.cc:390-392
        if f1.get("gs_image") is not None:  # main.cc:535-554: comparisons against the archive's global-shutter image
            original_gs, original_rs = f1["gs_image"], f1["rs_image"]
            difference = formats.abs_diff(backprojection, original_gs)
            formats.write_png(out_dir + "/gs_image.png", original_gs)
            formats.write_png(out_dir + "/difference.png", difference)
            formats.write_png(out_dir + "/remainder.png", formats.abs_diff(original_gs, difference))
            base = formats.shift_channel_bgr(original_gs, 1, 1, 1)
            formats.write_png(out_dir + "/overlay_gs_rs.png", formats.create_overlay_image(
                base, formats.shift_channel_bgr(formats.abs_diff(original_rs, original_gs), 2, 0.5, 0.5)))
            formats.write_png(out_dir + "/overlay_gs_bp.png", formats.create_overlay_image(base, formats.shift_channel_bgr(difference, 2, 0.5, 0.5)))
        formats.write_ply(out_dir + "/point_cloud.ply", coords, f1["rs_image"])
        formats.write_sweep_results(out_dir, [os.path.basename(task_dir.rstrip("/"))], [[w_err]], [[v_err]], [[stats["mean_error"]]],
                                    w=[res["w"]], v=[dm["v"]], k=[[res["k"]]])
    return out


def evaluate_real_run(solver, data_prefix, flow, camera="galaxy", gamma=0.95, out_dir=None, trials=5, tol=0.05, seed=1,
                      use_acceleration_mode=False, use_refinement=True, use_global_shutter_mode=False, flow_threshold=1e-10,
                      flow_index_mode=0, device=0):
    """The real-world branch of evaluateSingleRun (main.cc:341-361, 364-531; setupCameraReal main.cc:675-690): <data_prefix>frame1.png,
    one of the hard-coded phone calibrations (or a (f_x, f_y, c_x, c_y) tuple), gamma 0.95 -- and the optical flow from frame 1 to
    frame 2, which the reference computes with OpenCV's DeepFlow in-process (out of scope) and which is passed in here: an array, a
    .npy or a Middlebury .flo file (formats.load_flow).  Runs the whole solve in ONE device-resident call, then the consumers
    (8-bit depth image, back projection, crack interpolation, point cloud) and writes what the reference writes.  Defaults as in
    main.cc:304-311 (5 trials, tolerance 0.05, refinement on)."""
    import torch

    from . import BACKPROJECT_GS, BACKPROJECT_RS

    image = formats.read_png(data_prefix + "frame1.png") if isinstance(data_prefix, str) else np.ascontiguousarray(data_prefix, dtype=np.uint8)
    K = formats.CAMERA_INTRINSICS[camera] if isinstance(camera, str) else tuple(float(x) for x in camera)
    flow = formats.load_flow(flow)
    rows, cols = image.shape[:2]
    if flow.shape[:2] != (rows, cols):
        raise ValueError("flow is %dx%d, frame1 is %dx%d" % (flow.shape[0], flow.shape[1], rows, cols))
    dev = torch.device("cuda", device)
    mode = BACKPROJECT_GS if use_global_shutter_mode else BACKPROJECT_RS
    with torch.cuda.device(dev):  # everything stays on the device until the products are copied out
        d_flow, d_img = torch.from_numpy(flow).to(dev), torch.from_numpy(image).to(dev)
        d_map = torch.empty(rows * cols, dtype=torch.float64, device=dev)
        d_R = torch.empty(rows * 9, dtype=torch.float64, device=dev)
        d_t = torch.empty(rows * 3, dtype=torch.float64, device=dev)
        d_depth_est = torch.empty((rows, cols), dtype=torch.uint8, device=dev)
        d_gs, d_back = torch.empty_like(d_img), torch.empty_like(d_img)
        d_coords = torch.empty((rows, cols, 3), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        r = solver.solve_frame_dev(d_flow.data_ptr(), rows, cols, K, gamma, d_map.data_ptr(), d_R.data_ptr(), d_t.data_ptr(), trials=trials, tol=tol,
                                   seed=seed, use_acceleration_mode=use_acceleration_mode, use_refinement=use_refinement,
                                   flow_threshold=flow_threshold, flow_index_mode=flow_index_mode, use_global_shutter_mode=use_global_shutter_mode)
        m = r["num_inliers"]
        # main.cc:480-523 -- 8-bit depth image, back projection, crack interpolation -- in ONE call of two launches
        solver.rectify_frame_dev(r["d_inliers"], m, d_img.data_ptr(), d_map.data_ptr(), d_R.data_ptr(), d_t.data_ptr(), K, rows, cols, d_depth_est.data_ptr(),
                                 d_gs.data_ptr(), d_back.data_ptr(), d_coords=d_coords.data_ptr(), mode=mode, offset=1)
        solver.synchronize()
        depth_map = d_map.cpu().numpy().reshape(cols, rows).T.copy()  # the device map is column-major (Eigen MatrixXd)
        R_rel, t_rel = d_R.cpu().numpy().reshape(rows, 3, 3), d_t.cpu().numpy().reshape(rows, 3)
        depth_est, gs, backprojection, coords = d_depth_est.cpu().numpy(), d_gs.cpu().numpy(), d_back.cpu().numpy(), d_coords.cpu().numpy()
    out = dict(n=r["n"], num_inliers=m, v=r["v"], w=r["w"], k=r["k"], flipped=r["flipped"], refine_summary=r["refine_summary"],
               depth_map=depth_map, depth_est=depth_est, gs_image=gs, backprojection=backprojection, coords=coords, R=R_rel, t=t_rel)
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
        formats.write_png(out_dir + "/MinimalDepth.png", depth_est)
        formats.write_png(out_dir + "/rs_image.png", image)
        formats.write_png(out_dir + "/backprojection.png", backprojection)
        formats.write_ply(out_dir + "/point_cloud.ply", coords, image)
    return out


# ---------------------------------------------------------------------------------------------------
# parameter sweep (reference main.cc:148-299 -> error_measure::evaluateVelocities, errorMeasure.cpp:41-254)
# ---------------------------------------------------------------------------------------------------
def evaluate_velocities(solvers, archive, gamma, ransac_trials=50, num_evaluations=5, constant_acceleration=False, global_shutter=False,
                        optimize_results=True, tol=0.05, flow_threshold=1e-10, base_seed=1, image_path=None, flow_index_mode=0):
    """errorMeasure.cpp:41-254 for one task: ground-truth flow once, then `num_evaluations` independent solves (the reference
    reseeds rand() per trial; here evaluation e uses the sampler seed base_seed + e) with their rotation / translation /
    reprojection errors.  `solvers`: one Solver or a list -- the evaluations are independent and are spread over the
    contexts, one host thread each (sequence-throughput mode).  flow_index_mode 0 (default) = the reference's rank-indexed flow
    (errorMeasure.cpp:152 -> nonlinearRefinement.cc:209-212, quirk Q2), 1 = gathered."""
    from concurrent.futures import ThreadPoolExecutor

    from . import BACKPROJECT_GS, BACKPROJECT_RS, Solver, velocity_errors

    if isinstance(solvers, Solver):
        solvers = [solvers]
    K, truth = archive["K"], archive["truth"]
    f1, f2 = archive["frames"]
    rows, cols = f1["rs_image"].shape[:2]
    s0 = solvers[0]
    flow, _ = s0.true_flow(f1["world"], f2["R"], f2["t"], K, want_best_row=False)
    q, u, alpha, alpha_k = s0.flatten(flow, K, gamma, thr=flow_threshold)
    if global_shutter:  # errorMeasure.cpp:107-112
        alpha = alpha * 0.0 + 1.0
        constant_acceleration = False
    gt_depth = gt_depth_map(f1["world"], f1["R"], f1["t"])
    R_abs, t_abs = relocate_pose(f1["R"], f1["t"])

    def one(e):
        s = solvers[e % len(solvers)]
        rr = s.ransac(q, u, alpha, alpha_k, constant_acceleration, ransac_trials, tol, samples=None, seed=base_seed + e)
        res = dict(v=rr["v"], w=rr["w"], k=rr["k"], inliers=rr["inliers"])
        if optimize_results:
            ref = s.non_linear_refinement(u, rr["inliers"], rr["alpha"], rr["alpha_k"], rr["v"], rr["w"], rr["k"], constant_acceleration,
                                          flow_index_mode=flow_index_mode, inlier_idx=rr["inlier_idx"] if flow_index_mode else None)
            res = dict(v=ref["v"], w=ref["w"], k=ref["k"], inliers=ref["inliers"])
        dm = s.depth_map(res["inliers"], res["v"], K, rows, cols)
        w_err, v_err = velocity_errors(res["w"], dm["v"], truth["w"], truth["v"])
        R_rel, t_rel = s.pose_table(dm["v"], res["w"], res["k"], gamma, rows)
        _, coords = s.back_project(f1["rs_image"], dm["depth_map"], R_rel, t_rel, K, mode=BACKPROJECT_GS if global_shutter else BACKPROJECT_RS)
        stats, _ = s.reprojection_error(coords, gt_depth, dm["depth_map"], R_abs, t_abs, K, want_image=False)
        if image_path:
            formats.write_png("%s%d.png" % (image_path, e), s.depth_preview(dm["inliers"], K, rows, cols))
            formats.write_ply("%s%d.ply" % (image_path, e), coords, f1["rs_image"])
        return dict(w=res["w"], v=dm["v"], k=res["k"], w_error=w_err, v_error=v_err, reproject_error=stats["mean_error"], num_inliers=rr["num_inliers"])

    if len(solvers) == 1:
        runs = [one(e) for e in range(num_evaluations)]
    else:
        with ThreadPoolExecutor(max_workers=len(solvers)) as pool:
            # evaluation e always runs on context e % S: one thread per context at a time
            chunks = [list(range(j, num_evaluations, len(solvers))) for j in range(len(solvers))]
            parts = list(pool.map(lambda ch: [(e, one(e)) for e in ch], chunks))
        runs = [r for _, r in sorted((er for p in parts for er in p), key=lambda er: er[0])]
    col = lambda key: np.array([r[key] for r in runs])
    return dict(w=col("w"), v=col("v"), k=col("k"), error_w_vec=col("w_error"), error_v_vec=col("v_error"), error_reproject_vec=col("reproject_error"),
                error_w=float(col("w_error").mean()), error_v=float(col("v_error").mean()), error_reproject=float(col("reproject_error").mean()),
                num_inliers=col("num_inliers"))


def evaluate_parameter_sweep(solvers, path, result_dir, tasks=None, **kw):
    """main.cc:148-299: every task directory listed in <path>/tasks.txt (or `tasks`) -> errors.csv, w.csv, v.csv, k.csv,
    reproject_errors.csv, error_v.csv, error_w.csv and depthMaps/<task index>/<evaluation>.{png,ply} under result_dir"""
    if tasks is None:
        tasks = [ln.strip() for ln in open(os.path.join(path, "tasks.txt")) if ln.strip()]
    os.makedirs(result_dir, exist_ok=True)
    results = []
    for i, task in enumerate(tasks):
        archive = formats.load_example_archive(os.path.join(path, task))
        image_path = os.path.join(result_dir, "depthMaps", str(i)) + "/"
        os.makedirs(image_path, exist_ok=True)
        results.append(evaluate_velocities(solvers, archive, archive["truth"]["gamma"], image_path=image_path, **kw))
    formats.write_sweep_results(result_dir, tasks, [r["error_w_vec"] for r in results], [r["error_v_vec"] for r in results],
                                [r["error_reproject_vec"] for r in results], w=[r["w"] for r in results], v=[r["v"] for r in results],
                                k=[r["k"] for r in results])
    return dict(zip(tasks, results))
