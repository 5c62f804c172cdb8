#!/usr/bin/env python3
"""Generates tests/golden/golden_rectify_v1.npz -- fixtures that pin the oracle's and the HIP path's RS -> GS
back projection, crack interpolation and 8-bit depth image.

The reference ships no golden vectors and cannot be built here, so (like make_golden.py) these come from an
INDEPENDENT numpy transcription run in the build container only: homogeneous 4x4 matrices and numpy matmul for the
frame changes (rsframe.cc:687-736), sequential Python loops for the last-writer-wins splat (rsframe.cc:803-878), the
stencil of camera.cc:694-774 and the depth image of main.cc:480-509.  It imports neither the oracle nor the HIP
library.  Run:  python tests/golden/make_golden_rectify.py
"""
import importlib.util
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def _load_synth():
    spec = importlib.util.spec_from_file_location("synth", os.path.join(ROOT, "rs-aware-differential-sfm_amd", "synth.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


synth = _load_synth()


def skew(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=np.float64)


def pose_table_np(v, w, k, gamma, rows):
    """RsFrame::setRelativePose (rsframe.cc:771-800)"""
    R, t = np.zeros((rows, 3, 3)), np.zeros((rows, 3))
    for i in range(rows):
        b1 = 0.0 if i == 0 else (gamma * i / rows + 0.5 * k * (gamma * gamma * i * i) / (rows * rows)) * (2.0 / (2.0 + k))
        R[i] = np.eye(3) + b1 * skew(w)
        t[i] = b1 * np.asarray(v)
    return R, t


def to_int(x):
    """C++ int(double) on x86-64: truncation; non-finite / out of range -> INT_MIN"""
    if not np.isfinite(x) or x <= -2147483649.0 or x >= 2147483648.0:
        return -(2 ** 31)
    return int(np.trunc(x))


def back_project_np(img, depth, R, t, K, mode, q5_fixed):
    fx, fy, cx, cy = K
    rows, cols = img.shape[:2]
    gs = np.zeros_like(img)
    c3 = np.zeros((rows, cols, 3), dtype=np.float32)
    P0 = np.eye(4)
    P0[:3, :3], P0[:3, 3] = R[0], t[0]
    for y in range(rows):
        s = y if mode == 0 else 0
        Pinv = np.eye(4)
        Pinv[:3, :3] = R[s].T
        Pinv[:3, 3] = -(R[s].T) @ t[s]
        for x in range(cols):
            if tuple(img[y, x]) == (1, 1, 1):
                continue
            z = depth[y, x]
            pc = z * np.array([(x - cx) * 1.0 / fx, (y - cy) * 1.0 / fy, 1.0])
            pw = (Pinv @ np.append(pc, 1.0))[:3]
            pg = (P0 @ np.append(pw, 1.0))[:3]
            with np.errstate(divide="ignore", invalid="ignore"):
                gx = pg[0] / pg[2] * fx + cx
                gy = pg[1] / pg[2] * (fy if q5_fixed else fx) + cy
            c3[y, x] = pw.astype(np.float32)
            ix, iy = to_int(gx + 0.5), to_int(gy + 0.5)
            if 0 <= ix < cols and 0 <= iy < rows:
                gs[iy, ix] = img[y, x]
    return gs, c3


def is_black(p):
    return np.sqrt(float(p[0]) ** 2 + float(p[1]) ** 2 + float(p[2]) ** 2) <= 15


def interpolate_np(img, offset):
    rows, cols = img.shape[:2]
    out = img.copy()
    for r in range(offset, rows - offset):
        for c in range(offset, cols - offset):
            if not is_black(img[r, c]):
                continue
            nb = [img[r - offset, c], img[r + offset, c], img[r, c - offset], img[r, c + offset]]
            good = [p.astype(np.float64) for p in nb if not is_black(p)]
            if good:
                avg = (1 / float(len(good))) * np.sum(good, axis=0)
                out[r, c] = np.clip(np.rint(avg), 0, 255).astype(np.uint8)  # rint = nearest even = cvRound
    return out


def preview_np(inl, K, rows, cols):
    fx, fy, cx, cy = K
    out = np.zeros((rows, cols), dtype=np.uint8)
    z_min, z_max = np.inf, 0.0
    for z in inl[:, 2]:
        z_min, z_max = min(z_min, z), max(z_max, z)
    with np.errstate(divide="ignore"):
        mult = np.float64(244.0) / np.float64(z_max - z_min)
    for x_, y_, z in inl:
        x, y = int(fx * x_ + cx + 0.5), int(fy * y_ + cy + 0.5)
        zi = to_int((z - z_min) * mult)
        zi = 0 if zi == -(2 ** 31) else zi
        if 0 <= x < cols and 0 <= y < rows:
            out[y, x] = (10 + zi) % 256
    return out


def true_flow_np(world, R2, t2, K, q5_fixed):
    """Camera::calculateTrueFlow (camera.cc:209-249) + calculateImageCoordinatesRsFrame (rsframe.cc:740-768)"""
    fx, fy, cx, cy = K
    rows, cols = world.shape[:2]
    rows2 = R2.shape[0]
    P = np.zeros((rows2, 4, 4))
    P[:, :3, :3], P[:, :3, 3], P[:, 3, 3] = R2, t2, 1.0
    flow = np.zeros((rows, cols, 2))
    best = np.full((rows, cols), -1, dtype=np.int32)

    def project(i, W):
        pc = (P[i] @ np.append(W, 1.0))[:3]
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.array([pc[0] / pc[2] * fx + cx, pc[1] / pc[2] * (fy if q5_fixed else fx) + cy])

    for v in range(rows):
        for u in range(cols):
            W = world[v, u]
            f2 = np.array([float(u), float(v)])
            if np.linalg.norm(W) != 0:
                min_diff, b = np.inf, 0
                for i in range(rows2):
                    diff = abs(project(i, W)[1] - float(i))
                    if diff < min_diff:
                        min_diff, b = diff, i
                best[v, u] = b
                pt = project(b, W)
                if np.linalg.norm(pt) != 0:
                    f2 = pt
            flow[v, u] = f2 - np.array([float(u), float(v)])
    return flow, best


def velocity_errors_np(w, v, wt, vt):
    """errorMeasure.cpp:178-186"""
    A, B = np.eye(3) + skew(w), np.eye(3) + skew(wt)
    E = A @ B.T
    return float(np.linalg.norm([E[2, 1], E[0, 2], E[1, 0]])), float(np.arccos(np.dot(v, vt) / (np.linalg.norm(v) * np.linalg.norm(vt))))


def reprojection_np(est, gt_depth, est_depth, R, t, K, max_norm):
    """Camera::meanReprojectionError / createErrorImage (camera.cc:503-691), sequential like the reference"""
    fx, fy, cx, cy = K
    rows, cols = gt_depth.shape
    true = np.zeros((rows, cols, 3), dtype=np.float32)
    for x in range(cols):
        for y in range(rows):
            z = gt_depth[y, x]
            if z == 0:
                z = est_depth[y, x]
            pc = z * np.array([(x - cx) * 1.0 / fx, (y - cy) * 1.0 / fy, 1.0])
            Pinv = np.eye(4)
            Pinv[:3, :3] = R[y].T
            Pinv[:3, 3] = -(R[y].T) @ t[y]
            true[y, x] = (Pinv @ np.append(pc, 1.0))[:3].astype(np.float32)
    s, inl, outl = 0.0, 0, 0
    with np.errstate(divide="ignore", invalid="ignore"):
        for x in range(cols):
            for y in range(rows):
                for c in range(3):
                    ratio = np.float32(est[y, x, c]) / np.float32(true[y, x, c])
                    sc = float(ratio)
                    if abs(ratio) > 10:
                        sc = 0.0
                        outl += 1
                    if sc != 0 and sc == sc:
                        inl += 1
                        s += sc
        scale = s / float(inl) if inl else float("nan")
        se, einl = 0.0, 0
        img = np.zeros((rows, cols), dtype=np.uint8)
        for x in range(cols):
            for y in range(rows):
                e = est[y, x].astype(np.float64) / scale
                tr = true[y, x].astype(np.float64)
                n = float(np.sqrt(((e - tr) ** 2).sum()))
                if not np.isnan(e).any() and not np.isnan(tr).any() and n < 50:
                    se += n
                    einl += 1
                v = to_int(n * 255 / max_norm + 0.5)
                img[y, x] = (0 if v == -(2 ** 31) else v) % 256
    return dict(scale=scale, sum_error=se, mean_error=se / einl if einl else float("nan"), number_outliers=outl, scale_inliers=inl, error_inliers=einl), img


def texture(rows, cols, seed):
    """deterministic BGR test image: smooth colour ramps + a checker, dark (black) patches, a few marker pixels"""
    r = synth.splitmix64(seed, rows * cols).reshape(rows, cols)
    yy, xx = np.mgrid[0:rows, 0:cols]
    img = np.stack([(40 + 5 * xx + 3 * yy) % 256, (200 - 2 * xx + 7 * yy) % 256, (90 + 11 * ((xx // 4 + yy // 4) % 2) * 9 + xx) % 256], axis=2).astype(np.uint8)
    img[(r % np.uint64(13)) == 0] = (3, 2, 4)  # black-ish pixels (norm <= 15)
    img[(r % np.uint64(41)) == 1] = (1, 1, 1)  # marker pixels (skipped by backProject)
    img[5:9, 10:16] = 0
    return img


def main():
    out = {}
    for name, cfg, rows, cols, k in (("k0", 1, 40, 56, 0.0), ("k04", 1, 36, 48, 0.4)):
        d = synth.make_config(cfg, rows=rows, cols=cols, k=k)
        K, gamma, t_ = d["K"], d["gamma"], d["truth"]
        # motion scaled up so that pixels really move by several pixels across the frame
        vv, ww = np.array([0.12, 0.10, 0.05]), np.array([0.03, -0.02, 0.06])
        R, t = pose_table_np(vv, ww, k, gamma, rows)
        depth = np.array(t_["Z"], dtype=np.float64).copy()
        holes = synth.splitmix64(77, rows * cols).reshape(rows, cols) % np.uint64(17) == 0
        depth[holes] = 0.0  # pixels without an inlier
        img = texture(rows, cols, 5)
        g = lambda key: name + "/" + key
        out[g("K")], out[g("gamma")], out[g("v")], out[g("w")], out[g("k")] = np.array(K), gamma, vv, ww, k
        out[g("R")], out[g("t")], out[g("depth")], out[g("image")] = R, t, depth, img
        for mode in (0, 1):
            for q5 in (0, 1):
                gs, c3 = back_project_np(img, depth, R, t, K, mode, bool(q5))
                out[g("gs_m%d_q%d" % (mode, q5))] = gs
                out[g("c3_m%d" % mode)] = c3
        for off in (1, 2):
            out[g("interp_off%d" % off)] = interpolate_np(out[g("gs_m0_q0")], off)
        # inliers for the depth image: every third non-hole pixel in the reference's column-major order, z = depth
        fx, fy, cx, cy = K
        pts = [((x - cx) / fx, (y - cy) / fy, depth[y, x]) for x in range(cols) for y in range(rows) if not holes[y, x]][::3]
        inl = np.array(pts)
        out[g("inliers")] = inl
        out[g("preview")] = preview_np(inl, K, rows, cols)
        # ground-truth flow: world points of frame 1 (scanline-0 camera frame = world), frame 2 = the same rolling
        # shutter motion continued by one frame time
        yy, xx = np.mgrid[0:rows, 0:cols]
        world = np.stack([(xx - cx) / fx, (yy - cy) / fy, np.ones((rows, cols))], axis=2) * np.array(t_["Z"])[:, :, None]
        world[holes] = 0.0
        R2 = R.copy()
        t2 = t + np.array([0.04, 0.02, 0.01])
        out[g("world")], out[g("R2")], out[g("t2")] = world, R2, t2
        for q5 in (0, 1):
            fl, bst = true_flow_np(world, R2, t2, K, bool(q5))
            out[g("tf_flow_q%d" % q5)], out[g("tf_best_q%d" % q5)] = fl, bst
        # accuracy metrics: the back-projected points (estimate, arbitrary scale 1.7) against a ground truth with slightly
        # different depth and absolute poses; a few ground-truth zeros exercise the planeToSpace fallback
        gt_depth = np.array(t_["Z"]) * (1.0 + 0.02 * np.sin(0.3 * xx) * np.cos(0.2 * yy))
        gt_depth[(synth.splitmix64(99, rows * cols).reshape(rows, cols) % np.uint64(29)) == 0] = 0.0
        R_abs, t_abs = pose_table_np(vv * 1.05, ww * 0.97, k, gamma, rows)
        est = (out[g("c3_m0")].astype(np.float64) * 1.7).astype(np.float32)
        st, eimg = reprojection_np(est, gt_depth, depth, R_abs, t_abs, K, 10.0)
        out[g("gt_depth")], out[g("R_abs")], out[g("t_abs")], out[g("est_coords")] = gt_depth, R_abs, t_abs, est
        out[g("reproj_stats")] = np.array([st["scale"], st["mean_error"], st["sum_error"], st["number_outliers"], st["scale_inliers"], st["error_inliers"]])
        out[g("error_image")] = eimg
        we, ve = velocity_errors_np(ww * 0.97, vv * 1.05 + np.array([0.01, 0, 0]), ww, vv)
        out[g("vel_errors")] = np.array([we, ve])
    np.savez_compressed(os.path.join(HERE, "golden_rectify_v1.npz"), **out)
    print("wrote golden_rectify_v1.npz:", {k2: v2.shape for k2, v2 in out.items() if hasattr(v2, "shape")})


if __name__ == "__main__":
    main()
