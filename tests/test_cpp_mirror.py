"""The C++ drop-in mirror of the reference's function boundary (rs-aware-differential-sfm_amd/host/*.h).
CPU: it compiles and links against the C-ABI library.  GPU: tests/cpp/single_run.cpp (the solver part of the
reference's evaluateSingleRun, main.cc:398-522, written against the mirror) reproduces the Python-driven path."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

PKG = os.path.join(ROOT, "rs-aware-differential-sfm_amd")


def _build(tmp_path):
    exe = os.path.join(str(tmp_path), "single_run")
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-Werror", "-o", exe, os.path.join(ROOT, "tests", "cpp", "single_run.cpp"),
           "-L", PKG, "-lrsdsfm_hip", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


def _build_tiled(tmp_path):
    """tests/cpp/tiled_run.cpp: a C++ host of the column-tiled whole solve against the C ABI + the HIP runtime only"""
    exe = os.path.join(str(tmp_path), "tiled_run")
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", exe,
           os.path.join(ROOT, "tests", "cpp", "tiled_run.cpp"), "-L", PKG, "-lrsdsfm_hip", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


def test_rsframe_geometry_members_on_the_host(tmp_path, rsdsfm):
    """tests/cpp/geometry_host.cpp: worldToCameraFrame / cameraToWorldFrame / planeToSpace / spaceToPlane (quirk Q5) / pixel
    rounding / synthetic depth maps of the RsFrame mirror -- host arithmetic, runs without a GPU"""
    rsdsfm.load_library()
    exe = os.path.join(str(tmp_path), "geometry_host")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-Werror", "-o", exe, os.path.join(ROOT, "tests", "cpp", "geometry_host.cpp"),
                           "-L", PKG, "-lrsdsfm_hip", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"])
    p = subprocess.run([exe], capture_output=True, text=True)
    assert p.returncode == 0 and p.stdout.strip() == "ok", p.stderr
    # Camera::testProjection (camera.cc:374-408): one block per pixel with ground truth; the depth-map round trip lands on the pixel in x
    p = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, RSDSFM_TEST_PROJECTION="1"))
    blocks = p.stdout.split("---------------------------------------------")
    assert p.returncode == 0 and len(blocks) == 12 * 16 + 1 and blocks[0].startswith("x, y: 0, 0")
    xs = blocks[17].splitlines()
    px = float([ln for ln in xs if ln.startswith("x, y:")][0].split(":")[1].split(",")[0])
    i2 = [ln for ln in xs if ln.startswith("image coordinates2:")][0].split(":")[1].split()  # plane -> space -> plane
    assert abs(float(i2[0]) - px) < 1e-4  # (printed with 6 significant digits)


def test_image_helpers_on_the_host(tmp_path, rsdsfm, oracle):
    """tests/cpp/image_host.cpp: shiftChannelBGR / createOverlayImage / absDiff / isBlackPixel / isColorfulArea / interpolateAreaColor /
    addFrameReal / reconstructImageFromFlow of the Camera mirror against the package's numpy versions (formats.py) and the oracle's crack interpolation -- two
    independent statements of cv::Vec3b arithmetic (rounded products, saturated sums); host only, runs without a GPU"""
    rsdsfm.load_library()
    exe = os.path.join(str(tmp_path), "image_host")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-Werror", "-o", exe, os.path.join(ROOT, "tests", "cpp", "image_host.cpp"),
                           "-L", PKG, "-lrsdsfm_hip", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"])
    rng = np.random.default_rng(11)
    rows, cols = 37, 53
    a = rng.integers(0, 256, (rows, cols, 3), dtype=np.uint8)
    a[rng.random((rows, cols)) < 0.3] = rng.integers(0, 9, 3, dtype=np.uint8)   # black pixels (norm <= 15): cracks to interpolate
    a[5:9, 7:12] = 0                                                              # and a black area with black neighbours
    b = np.clip(a.astype(np.int16) + rng.integers(-40, 41, a.shape), 0, 255).astype(np.uint8)
    b[rng.random((rows, cols)) < 0.5] = a[rng.random((rows, cols)) < 0.5][0]    # many equal / nearly equal pixels: black differences
    pa, pb, out = [os.path.join(str(tmp_path), n) for n in ("a.raw", "b.raw", "out")]
    a.tofile(pa), b.tofile(pb)
    p = subprocess.run([exe, pa, pb, str(rows), str(cols), out], capture_output=True, text=True)
    assert p.returncode == 0 and p.stdout.strip() == "ok", p.stderr
    rd = lambda ext: np.fromfile(out + ext, dtype=np.uint8).reshape(rows, cols, 3)
    F = rsdsfm.formats
    assert np.array_equal(rd(".shift"), F.shift_channel_bgr(a, 2, 0.5, 0.5))
    ov = F.create_overlay_image(F.shift_channel_bgr(a, 1, 1, 1), F.shift_channel_bgr(F.abs_diff(a, b), 2, 0.5, 0.5))
    assert np.array_equal(rd(".overlay"), ov)
    assert (ov != a).any() and (ov == a).all(axis=2).any()  # both branches taken
    assert np.array_equal(rd(".cracky"), oracle.interpolate_cracky(a, 1))
    vv, uu = np.mgrid[0:rows, 0:cols]
    flow = np.stack([((uu * 7 + vv * 3) % 11 - 5) * 0.5, ((uu * 5 + vv * 2) % 7 - 3) * 0.75], axis=2)
    warp = F.reconstruct_image_from_flow(a, flow)
    assert np.array_equal(rd(".warp"), warp) and (warp[0] == 0).all() and (warp[:, 0] == 0).all() and (warp != 0).any()
    ref = np.zeros_like(a)  # the reference's loop, spelled out
    for u in range(cols):
        for v in range(rows):
            nx, ny = u + int(np.floor(flow[v, u, 0] + 0.5)), v + int(np.floor(flow[v, u, 1] + 0.5))
            if 0 < nx < cols and 0 < ny < rows:
                ref[ny, nx] = a[v, u]
    assert np.array_equal(warp, ref)
    # known answers of the 8-bit arithmetic: truncating gain, round-half-even blend, saturated sum
    px = np.array([[[200, 101, 7]]], dtype=np.uint8)
    assert F.shift_channel_bgr(px, 2, 0.5, 0.5).tolist() == [[[255, 50, 3]]]
    o, sh = np.array([[[30, 40, 0]]], dtype=np.uint8), np.array([[[0, 30, 40]]], dtype=np.uint8)  # equal norms: multiplier 0.5
    assert F.create_overlay_image(o, sh).tolist() == [[[15, 35, 20]]]
    assert F.create_overlay_image(o, (sh // 10)).tolist() == o.tolist()  # shift pixel black (norm 5): original kept
    assert F.abs_diff(np.array([3, 250], dtype=np.uint8), np.array([10, 5], dtype=np.uint8)).tolist() == [7, 245]


def _build_sequence(tmp_path):
    """tests/cpp/sequence_run.cpp: a C++ host of the sequence solve (rsdsfm_solve_frames_dev) against the C ABI + the HIP runtime only"""
    exe = os.path.join(str(tmp_path), "sequence_run")
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", exe,
           os.path.join(ROOT, "tests", "cpp", "sequence_run.cpp"), "-L", PKG, "-lrsdsfm_hip", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


def test_sequence_host_compiles_and_links(tmp_path, rsdsfm):
    rsdsfm.load_library()
    exe = _build_sequence(tmp_path)
    assert subprocess.run([exe], capture_output=True).returncode == 2  # usage, no GPU work


@pytest.mark.gpu
def test_sequence_host_in_cpp_equals_single_solves(tmp_path, rsdsfm):
    """BASELINE configs[4] from a C++ host (the reference is C++): 7 pairs with different data seeds, one of them with dropped pixels,
    through ONE rsdsfm_solve_frames_dev call over 3 lanes; every pair equals rsdsfm_solve_frame_dev on another context bit for bit
    (pose, counts, iteration count, depth map), and the Python binding's single solve agrees"""
    import torch

    exe = _build_sequence(tmp_path)
    B, rows, cols = 7, 120, 200
    imgs = []
    for i in range(B):
        d = rsdsfm.synth.make_config(5 if i % 2 == 0 else 3, rows=rows, cols=cols, seed=0x5EED0200 + i)
        im = d["flow_img"].copy()
        if i == 3:
            im[30:50, 60:90] = 0.0
        imgs.append(im)
    K, gamma = d["K"], d["gamma"]
    raw = os.path.join(str(tmp_path), "flows.bin")
    np.stack(imgs).astype(np.float64).tofile(raw)
    T, tol = 10, 0.01
    out = subprocess.run([exe, raw, str(B), str(rows), str(cols)] + ["%.17g" % x for x in K] + ["%.17g" % gamma, str(T), "%.17g" % tol, "3"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    recs = [json.loads(ln) for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(recs) == B and all(r["equals_single_solve"] for r in recs), recs
    assert recs[3]["n"] == rows * cols - 20 * 30 and recs[0]["n"] == rows * cols
    dev = torch.device("cuda", 0)
    with rsdsfm.Solver(0) as s:
        for i in (0, 3, 6):
            img = torch.from_numpy(imgs[i]).to(dev)
            dm = torch.empty((cols, rows), dtype=torch.float64, device=dev)
            one = s.solve_frame_dev(img.data_ptr(), rows, cols, K, gamma, dm.data_ptr(), trials=T, tol=tol, seed=100 + 7 * i)
            assert one["num_inliers"] == recs[i]["num_inliers"] and one["best_trial"] == recs[i]["best_trial"]
            assert np.array_equal(one["v"], recs[i]["v"]) and np.array_equal(one["w"], recs[i]["w"])


def test_tiled_host_compiles_and_links(tmp_path, rsdsfm):
    rsdsfm.load_library()
    exe = _build_tiled(tmp_path)
    assert subprocess.run([exe], capture_output=True).returncode == 2  # usage, no GPU work


@pytest.mark.gpu
def test_tiled_host_in_cpp_over_rccl_matches_python_solve(tmp_path, rsdsfm):
    """the C++ host process (no Python, no torch: RCCL comes from /opt/rocm through the library's dlopen) runs the column-tiled solve
    as rank 0 of 1 over a 1-rank RCCL communicator; its pose, counts, depth map and pose table equal the one-call solve's"""
    import torch

    exe = _build_tiled(tmp_path)
    d = rsdsfm.synth.make_config(3, rows=120, cols=200)
    rows, cols, K, gamma = d["rows"], d["cols"], d["K"], d["gamma"]
    raw = os.path.join(str(tmp_path), "flow.bin")
    d["flow_img"].astype(np.float64).tofile(raw)
    T, tol, seed = 12, 0.002, 9
    out = subprocess.run([exe, raw, str(rows), str(cols)] + ["%.17g" % x for x in K] + ["%.17g" % gamma, str(T), "%.17g" % tol, str(seed)],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    dev = torch.device("cuda", 0)
    img = torch.from_numpy(d["flow_img"]).to(dev)
    dm = torch.empty((cols, rows), dtype=torch.float64, device=dev)
    R = torch.empty((rows, 9), dtype=torch.float64, device=dev)
    t = torch.empty((rows, 3), dtype=torch.float64, device=dev)
    with rsdsfm.Solver(0) as s:
        one = s.solve_frame_dev(img.data_ptr(), rows, cols, K, gamma, dm.data_ptr(), R.data_ptr(), t.data_ptr(), trials=T, tol=tol, seed=seed)
        s.synchronize()  # (both sides with the zero-initialised default: the reference's rank-indexed flow)
    assert r["world"] == 1 and r["n"] == one["n"] and r["num_inliers"] == one["num_inliers"] and r["best_trial"] == one["best_trial"]
    assert np.array_equal(r["v"], one["v"]) and np.array_equal(r["w"], one["w"]) and r["k"] == one["k"]
    assert r["iterations"] == one["refine_summary"]["num_iterations"] and r["flipped"] == int(one["flipped"])
    dmh = dm.cpu().numpy()
    assert r["depth_nonzero"] == int((dmh != 0).sum()) and np.isclose(r["depth_sum"], dmh.sum(), rtol=1e-12)
    assert np.array_equal(r["last_t"], t.cpu().numpy()[-1])
    # counts, RANSAC, one refinement poll per chunk of 5 exchange slots (one slot per LM iteration, a second one behind every iteration whose
    # speculated Schur sums did not apply), the final header -- never per RANSAC round or LM iteration
    # (+ at most two more when the RANSAC needs a second LM round / the separate scoring pass)
    assert r["host_syncs"] <= 7 + -(-2 * r["iterations"] // 5) and r["collectives"] >= 8


def test_mirror_compiles_and_links(tmp_path, rsdsfm):
    rsdsfm.load_library()
    exe = _build(tmp_path)
    assert os.path.exists(exe)
    # without arguments it prints the usage and exits with 2 (no GPU work)
    assert subprocess.run([exe], capture_output=True).returncode == 2


@pytest.mark.gpu
def test_single_run_matches_python_path(tmp_path, rsdsfm, oracle):
    exe = _build(tmp_path)
    d = rsdsfm.synth.make_config(3, rows=120, cols=200)
    img, K, gamma = d["flow_img"], d["K"], d["gamma"]
    raw = os.path.join(str(tmp_path), "flow.bin")
    img.astype(np.float64).tofile(raw)
    T, tol, seed = 15, 0.002, 4242
    out = subprocess.run([exe, raw, "120", "200"] + ["%.17g" % x for x in K] + ["%.17g" % gamma, str(T), "%.17g" % tol, str(seed)],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    r = json.loads(lines[-1])
    # show_messages = true prints what the reference prints (minimal.cc:286-288, nonlinearRefinement.cc:230-234)
    fin = [ln for ln in lines[:-1] if ln.startswith("Finished ")]
    assert [ln.split()[1] for ln in fin] == [str(i + 1) for i in range(T)]
    best_so_far = [int(ln.rstrip(".").split()[-1]) for ln in fin]
    assert best_so_far == sorted(best_so_far) and best_so_far[-1] == r["ransac_inliers"]
    rep = [ln for ln in lines[:-1] if ln.startswith("Ceres Solver Report: Iterations: ")]
    assert len(rep) == 1 and rep[0].endswith("Termination: CONVERGENCE") and "Initial cost: " in rep[0] and "Final cost: " in rep[0]
    assert any(ln.startswith("Total time for solving optimization: ") for ln in lines[:-1])
    with rsdsfm.Solver(0) as s:
        q, u, a, ak = s.flatten(img, K, gamma)
        rr = s.ransac(q, u, a, ak, False, T, tol, samples=None, seed=seed, depth_mode=1)
        ref = s.non_linear_refinement(u, rr["inliers"], rr["alpha"], rr["alpha_k"], rr["v"], rr["w"], rr["k"], False, flow_index_mode=1,
                                      inlier_idx=rr["inlier_idx"])
        sm = ref["summary"]
        assert rep[0].startswith("Ceres Solver Report: Iterations: %d, Initial cost: %e, Final cost: %e," % (
            sm["num_successful_steps"] + sm["num_unsuccessful_steps"], sm["initial_cost"], sm["final_cost"]))
        dm = s.depth_map(ref["inliers"], ref["v"], K, 120, 200)
        R, t = s.pose_table(dm["v"], ref["w"], ref["k"], gamma, 120)
        rho0 = s.estimate_inverse_depth(q[0], rr["v"], rr["w"], u[0], rr["k"], a[0], ak[0])
    assert r["rho0"] == rho0  # estimateInverseDepth with the reference's (Vector2d, Vector3d, ...) signature
    assert r["n"] == len(q) and r["ransac_inliers"] == rr["num_inliers"]
    assert np.array_equal(r["ransac_w"], rr["w"]) and np.array_equal(r["ransac_v"], rr["v"])
    assert np.array_equal(r["w"], ref["w"]) and np.array_equal(r["v"], dm["v"]) and r["k"] == ref["k"]
    assert r["flipped"] == int(dm["flipped"])
    assert r["ysum"] == int(dm["ys"].astype(np.int64).sum())  # scanline indices: bit-exact
    assert np.isclose(r["zsum"], dm["inliers"][:, 2].sum(), rtol=1e-12)
    assert np.array_equal(r["last_t"], t[-1]) and r["last_R01"] == R[-1][0, 1]
    # 8-bit depth image, back projection and crack interpolation of the mirror's deterministic test image
    yy, xx = np.mgrid[0:120, 0:200]
    rs = np.stack([(40 + 5 * xx + 3 * yy) % 256, (200 + 7 * yy + 254 * xx) % 256, (90 + xx + 99 * ((xx // 4 + yy // 4) % 2)) % 256], axis=2).astype(np.uint8)
    wsum = lambda a: int((a.reshape(-1).astype(np.uint64) * (np.arange(a.size, dtype=np.uint64) % np.uint64(251) + np.uint64(1))).sum())
    gs_o, _ = oracle.back_project(rs, dm["depth_map"], R, t, *K)
    assert r["preview_sum"] == int(oracle.depth_preview(dm["inliers"], *K, 120, 200).astype(np.uint64).sum())
    assert r["gs_sum"] == wsum(gs_o) and r["bp_sum"] == wsum(oracle.interpolate_cracky(gs_o, 1))
    # the C++ mirror's PLY writer (createPointCloud) produces the same text as the package's writer on the oracle's points
    _, c3_ply = oracle.back_project(rs, dm["depth_map"], R, t, *K)
    ply_py = os.path.join(str(tmp_path), "py.ply")
    rsdsfm.formats.write_ply(ply_py, c3_ply, rs)
    assert open(raw + ".ply").read() == open(ply_py).read()
    # ground-truth flow of the estimated structure under the estimated motion (Camera::calculateTrueFlow)
    dmm = dm["depth_map"]
    yy, xx = np.mgrid[0:120, 0:200]
    wpts = np.stack([dmm * ((xx - K[2]) * 1.0 / K[0]), dmm * ((yy - K[3]) * 1.0 / K[1]), dmm], axis=2)
    flow_o, _ = oracle.true_flow(wpts, R, t, *K)
    assert np.isclose(r["tf_sum"], float((flow_o[:, :, 0] * 3.0 + flow_o[:, :, 1]).sum()), rtol=1e-12)
    one, _ = oracle.true_flow(wpts[60:61, 66:67], R, t, *K)
    assert np.array_equal(r["tf_point"], one[0, 0])
    # accuracy metrics through the mirror: ground-truth depth from the unprojection maps and the absolute poses
    # (getGroundtruthDepthMap), relocatePose, then meanReprojectionError / createErrorImage on the GPU
    t_abs = t * 1.02
    _, c3 = oracle.back_project(rs, dmm, R, t, *K)
    gt_depth = np.where(np.sqrt((wpts ** 2).sum(axis=2)) > 0, (R[:, 2, :][:, None, :] * wpts).sum(axis=2) + t_abs[:, 2][:, None], 0.0)
    R_rel, t_rel = R.copy(), t_abs - t_abs[0]  # relocatePose: scanline 0 is the identity here, so only the subtraction acts
    st_o, eimg_o = oracle.reprojection_error(c3, gt_depth, dmm, R_rel, t_rel, *K, max_norm=0.05)
    assert np.isclose(r["mean_reproj"], st_o["mean_error"], rtol=1e-9)
    assert abs(r["err_img_sum"] - int(eimg_o.astype(np.uint64).sum())) <= 255 * 3  # scale differs in the last bits only
    we, ve = oracle.velocity_errors(ref["w"], dm["v"], rr["w"], rr["v"])
    assert np.isclose(r["w_err"], we, rtol=1e-12) and np.isclose(r["v_err"], ve, rtol=1e-9, atol=1e-12)
    # and the oracle agrees with the whole chain (same sampler, same seed)
    ro = oracle.ransac(q, u, a, ak, False, T, tol, oracle.sample_indices(len(q), T, seed), depth_mode=1)
    assert ro["num_inliers"] == r["ransac_inliers"]
