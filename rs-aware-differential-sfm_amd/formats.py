"""On-disk formats of the reference's example archives and outputs -- host-side I/O only.

  A.csv                         3 x 3 intrinsics                      Camera::loadIntrinsicsFromFile   camera.cc:99-176
  {1,2}_rs_t.csv                rows x 3 scanline positions           RsFrame::setPoses                rsframe.cc:444-553
  {1,2}_rs_r.csv                rows x 9 row-major scanline rotations RsFrame::setPoses
  {1,2}_rs_unproject_{x,y,z}.csv rows x cols world coordinates         RsFrame::setUnprojectionMapRs    rsframe.cc:58-218
  {1,2}_rs.png ...              8-bit frames                          cv::imread / cv::imwrite         main.cc:613-671, 526-556
  v.csv w.csv gamma.csv k.csv   ground-truth motion of a task         main.cc:216-256 (matlab start_generating.m:38-42)
  point_cloud.ply               ascii PLY                             Camera::createPointCloud         camera.cc:423-491
  errors.csv, w.csv, ...        sweep results                         main.cc:179-206, 262-300

Numbers are parsed like the reference's `::atof` (longest valid prefix, 0 on garbage); the line-count checks of the
reference loaders (number of '\\n' == rows) are enforced and reported as ValueError (the reference prints a message and
leaves the frame unset).  PNG: non-interlaced 8-bit grey / RGB / RGBA read (all five filters), grey / RGB written
(filter 0, zlib level 0 like the reference's CV_IMWRITE_PNG_COMPRESSION 0 when `compression=0`).  Colour images are
BGR in memory like cv::Mat.
"""
import re
import struct
import zlib

import numpy as np

_NUM = re.compile(r"^[ \t\n\v\f\r]*([+-]?(?:\d+\.?\d*(?:[eE][+-]?\d+)?|\.\d+(?:[eE][+-]?\d+)?|inf(?:inity)?|nan))", re.I)


def atof(s):
    """C atof: longest valid numeric prefix after leading whitespace, 0.0 if there is none"""
    m = _NUM.match(s)
    return float(m.group(1)) if m else 0.0


def _lines(path):
    txt = open(path, "r").read()
    return txt, txt.count("\n")


def read_matrix_csv(path, rows, cols):
    """rows x cols comma-separated numbers; the file must contain exactly `rows` newline characters"""
    txt, n = _lines(path)
    if n != rows:
        raise ValueError("The number of lines: %d in the file: %s does not conform with the expected %d" % (n, path, rows))
    out = np.zeros((rows, cols))
    for i, line in enumerate(txt.split("\n")[:rows]):
        parts = line.split(",")
        # the reference reads cols-1 comma-terminated fields and the rest of the line as the last one
        for j in range(cols):
            field = parts[j] if j < cols - 1 else ",".join(parts[cols - 1:])
            out[i, j] = atof(field) if j < len(parts) else 0.0
    return out


def load_intrinsics(path):
    """Camera::loadIntrinsicsFromFile (camera.cc:99-176): 3 lines x 3 entries -> (fx, fy, cx, cy) and the matrix"""
    K = read_matrix_csv(path, 3, 3)
    return (K[0, 0], K[1, 1], K[0, 2], K[1, 2]), K


def load_poses(csv_t, csv_r, rows):
    """RsFrame::setPoses (rsframe.cc:444-553): t rows x 3, R rows x 9 row-major -> (R [rows,3,3], t [rows,3])"""
    t = read_matrix_csv(csv_t, rows, 3)
    R = read_matrix_csv(csv_r, rows, 9).reshape(rows, 3, 3)
    return R, t


def load_unprojection(csv_x, csv_y, csv_z, rows, cols):
    """RsFrame::setUnprojectionMapRs (rsframe.cc:58-218) -> (rows, cols, 3) world points"""
    return np.stack([read_matrix_csv(p, rows, cols) for p in (csv_x, csv_y, csv_z)], axis=2)


def load_task_truth(task_dir):
    """main.cc:216-256: first comma-separated fields of v.csv, w.csv (3 each), gamma.csv, k.csv"""
    def first(path, n):
        fields = open(path).read().split(",")
        return [atof(f) for f in fields[:n]]

    return dict(v=np.array(first(task_dir + "/v.csv", 3)), w=np.array(first(task_dir + "/w.csv", 3)),
                gamma=first(task_dir + "/gamma.csv", 1)[0], k=first(task_dir + "/k.csv", 1)[0])


def write_matrix_csv(path, M, fmt="%.17g"):
    M = np.atleast_2d(np.asarray(M, dtype=np.float64))
    with open(path, "w") as f:
        for row in M:
            f.write(",".join(fmt % v for v in row) + "\n")


# ---- PLY (camera.cc:423-491) ---------------------------------------------------------------------
def write_ply(path, coords, image_bgr):
    """ascii PLY exactly as Camera::createPointCloud writes it: float coordinates with 9 significant digits, colours RGB"""
    c = np.ascontiguousarray(coords, dtype=np.float32).reshape(-1, 3)
    col = np.ascontiguousarray(image_bgr, dtype=np.uint8).reshape(-1, 3)
    assert len(c) == len(col)
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment PLY File created by RS aware SfM wrapper\nelement vertex %d\n" % len(c))
        f.write("property float x\nproperty float y\nproperty float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n")
        for p, b in zip(c, col):
            f.write("%.9g %.9g %.9g %d %d %d\n" % (float(p[0]), float(p[1]), float(p[2]), b[2], b[1], b[0]))


def read_ply(path):
    lines = open(path).read().split("\n")
    n = int([ln for ln in lines if ln.startswith("element vertex")][0].split()[2])
    body = lines[lines.index("end_header") + 1: lines.index("end_header") + 1 + n]
    arr = np.array([[float(x) for x in ln.split()] for ln in body]).reshape(n, 6)
    return arr[:, :3].astype(np.float32), arr[:, 3:].astype(np.uint8)[:, ::-1]  # coords, BGR


# ---- PNG (8-bit, non-interlaced) -------------------------------------------------------------------
def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def write_png(path, img, compression=0):
    """img: (rows, cols) grey or (rows, cols, 3) BGR uint8"""
    a = np.ascontiguousarray(img, dtype=np.uint8)
    if a.ndim == 3:
        a = a[:, :, ::-1]  # BGR -> RGB on disk
        ctype, ch = 2, 3
    else:
        ctype, ch = 0, 1
    rows, cols = a.shape[:2]
    raw = b"".join(b"\x00" + a[r].tobytes() for r in range(rows))
    png = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", cols, rows, 8, ctype, 0, 0, 0)) + \
        _chunk(b"IDAT", zlib.compress(raw, compression)) + _chunk(b"IEND", b"")
    open(path, "wb").write(png)
    return ch


def read_png(path, grayscale=False):
    """-> (rows, cols, 3) BGR (cv::imread CV_LOAD_IMAGE_COLOR) or (rows, cols) grey (CV_LOAD_IMAGE_GRAYSCALE)"""
    data = open(path, "rb").read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("%s is not a PNG file" % path)
    pos, idat, hdr = 8, b"", None
    while pos < len(data):
        ln, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + ln]
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif tag == b"IDAT":
            idat += body
        elif tag == b"IEND":
            break
        pos += 12 + ln
    cols, rows, depth, ctype, _, _, interlace = hdr
    if depth != 8 or interlace != 0 or ctype not in (0, 2, 4, 6):
        raise ValueError("unsupported PNG variant (need 8-bit non-interlaced grey / RGB / +alpha)")
    ch = {0: 1, 2: 3, 4: 2, 6: 4}[ctype]
    raw = zlib.decompress(idat)
    stride = cols * ch
    out = np.zeros((rows, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.int32)
    p = 0
    for r in range(rows):
        ft = raw[p]
        line = np.frombuffer(raw, dtype=np.uint8, count=stride, offset=p + 1).astype(np.int32)
        p += 1 + stride
        cur = np.zeros(stride, dtype=np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        else:
            for i in range(stride):
                a = cur[i - ch] if i >= ch else 0
                b = prev[i]
                c = prev[i - ch] if i >= ch else 0
                if ft == 1:
                    pred = a
                elif ft == 3:
                    pred = (a + b) >> 1
                else:  # Paeth
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (line[i] + pred) & 255
        out[r] = cur
        prev = cur
    img = out.reshape(rows, cols, ch)
    if ch in (2, 4):
        img = img[:, :, :ch - 1]  # alpha dropped like cv::imread without IMREAD_UNCHANGED
    if img.shape[2] == 1:
        grey = img[:, :, 0]
        return grey.copy() if grayscale else np.repeat(grey[:, :, None], 3, axis=2)
    bgr = img[:, :, ::-1].copy()
    if grayscale:  # cv::cvtColor BGR2GRAY fixed-point weights (R 4899, G 9617, B 1868, 14-bit)
        return ((bgr[:, :, 2].astype(np.int32) * 4899 + bgr[:, :, 1].astype(np.int32) * 9617 + bgr[:, :, 0].astype(np.int32) * 1868 + 8192) >> 14).astype(np.uint8)
    return bgr


# ---- real-world runs: phone calibrations (camera.cc:179-206) and externally computed optical flow ----------------
CAMERA_INTRINSICS = {  # (f_x, f_y, c_x, c_y) of Camera::setIntrinsics(std::string)
    "iphone": (1505.1283359786307, 1513.7789208311444, 657.81734686405991, 349.91807538147589),
    "galaxy_stabil": (1803.29785922382, 1799.35406531529, 945.304708272490, 544.684292978344),
    "galaxy": (1492.41306997746, 1491.09286590722, 949.571146410704, 554.675409391795),
    "galaxy_old": (3154.53208221173, 3152.28696217577, 1969.87107268891, 1521.27056048818),
    "galaxy_vga": (484.450845764569, 485.345469134313, 313.442094604855, 241.383116350144),
}


def read_flo(path):
    """Middlebury .flo optical flow (magic 202021.25, int32 width, int32 height, float32 (u, v) row-major) -> rows x cols x 2 float64.
    The reference computes DeepFlow inside the process (camera.cc:251-300, out of scope here: OpenCV contrib); a flow computed by any
    external tool enters the solver through this file format or a .npy array."""
    with open(path, "rb") as f:
        magic = np.frombuffer(f.read(4), dtype="<f4")
        if magic.size != 1 or magic[0] != 202021.25:
            raise ValueError("%s: not a .flo file" % path)
        w, h = (int(x) for x in np.frombuffer(f.read(8), dtype="<i4"))
        data = np.frombuffer(f.read(8 * w * h), dtype="<f4")
    if data.size != 2 * w * h:
        raise ValueError("%s: truncated .flo file" % path)
    return data.reshape(h, w, 2).astype(np.float64)


def write_flo(path, flow):
    flow = np.asarray(flow, dtype=np.float32)
    with open(path, "wb") as f:
        f.write(np.array([202021.25], dtype="<f4").tobytes())
        f.write(np.array([flow.shape[1], flow.shape[0]], dtype="<i4").tobytes())
        f.write(np.ascontiguousarray(flow, dtype="<f4").tobytes())


def load_flow(path_or_array):
    """rows x cols x 2 float64 flow (x, y displacement in pixels, as cv::Mat_<cv::Point_<double>>) from an array, a .npy or a .flo file"""
    if isinstance(path_or_array, str):
        flow = read_flo(path_or_array) if path_or_array.endswith(".flo") else np.load(path_or_array)
    else:
        flow = path_or_array
    flow = np.ascontiguousarray(flow, dtype=np.float64)
    if flow.ndim != 3 or flow.shape[2] != 2:
        raise ValueError("flow must be rows x cols x 2")
    return flow


# ---- sweep outputs (main.cc:179-206, 262-300) -----------------------------------------------------
# ---- the diagnostic images evaluateSingleRun writes next to its results (main.cc:386-394, :533-554) ----------------------------
def abs_diff(a, b):
    """`abs(a - b)` of two 8-bit cv::Mat as the reference evaluates it (OpenCV folds the expression into absdiff): |a - b| per byte"""
    return np.abs(a.astype(np.int16) - b.astype(np.int16)).astype(np.uint8)


def _saturate_u8(x):
    """cv::saturate_cast<uchar>(double): round half to even (cvRound), then clamp"""
    return np.clip(np.rint(x), 0, 255).astype(np.uint8)


def shift_channel_bgr(img, shift_blue, shift_green, shift_red):
    """Camera::shiftChannelBGR (camera.cc:777-815): per-channel gain, clamped to [0, 255], truncated"""
    v = img.astype(np.float64) * np.array([shift_blue, shift_green, shift_red], dtype=np.float64)
    return np.clip(v, 0.0, 255.0).astype(np.uint8)


def create_overlay_image(original, shift, black_threshold=15):
    """Camera::createOverlayImage (camera.cc:818-840): where `shift` is not black (pixel norm > 15) the two pixels are blended by
    their norms in cv::Vec3b arithmetic -- each product rounded to 8 bits, the sum saturated --, elsewhere the original pixel"""
    o, sft = original.astype(np.float64), shift.astype(np.float64)
    no, ns = np.sqrt((o * o).sum(axis=2)), np.sqrt((sft * sft).sum(axis=2))
    blend = ns > black_threshold
    with np.errstate(invalid="ignore", divide="ignore"):
        m = np.where(blend, no / (no + ns), 1.0)[:, :, None]
    a = _saturate_u8(m * o).astype(np.int16)
    b = _saturate_u8((1.0 - m) * sft).astype(np.int16)
    out = np.minimum(a + b, 255).astype(np.uint8)
    return np.where(blend[:, :, None], out, original)


def reconstruct_image_from_flow(image, flow):
    """Camera::reconstructImageFromFlow (camera.cc:842-865): the image pushed forward along the flow rounded to whole pixels;
    columns outer / rows inner, the last writer wins, targets with x <= 0 or y <= 0 are dropped (the reference's strict test)"""
    rows, cols = flow.shape[:2]
    out = np.zeros_like(image)
    dx = np.floor(flow[:, :, 0] + 0.5).astype(np.int64)
    dy = np.floor(flow[:, :, 1] + 0.5).astype(np.int64)
    vv, uu = np.mgrid[0:rows, 0:cols]
    nx, ny = uu + dx, vv + dy
    ok = (nx > 0) & (nx < cols) & (ny > 0) & (ny < rows)
    order = np.argsort((uu * rows + vv)[ok], kind="stable")  # the reference's scan order: later sources overwrite earlier ones
    out[ny[ok][order], nx[ok][order]] = image[vv[ok][order], uu[ok][order]]
    return out


def flow_to_bgr(flow):
    """Camera::getImageOpticalFlow (camera.cc:280-309) followed by main.cc:391's scaling to 8 bits: direction as hue, magnitude
    (normalised by its maximum) as value, full saturation.  Visualisation only: OpenCV's cartToPolar / cvtColor work in float with a
    polynomial atan2 (0.3 degrees), so this image is not claimed to be byte-identical to the reference's optical_flow.png."""
    f = np.asarray(flow, dtype=np.float32)
    mag = np.sqrt(f[:, :, 0] ** 2 + f[:, :, 1] ** 2)
    ang = np.degrees(np.arctan2(f[:, :, 1], f[:, :, 0])) % 360.0
    mmax = float(mag.max()) if mag.size else 0.0
    val = mag / mmax if mmax > 0 else mag
    h = ang / 60.0
    i = np.floor(h).astype(np.int32) % 6
    fr = h - np.floor(h)
    p, q_, t = np.zeros_like(val), val * (1.0 - fr), val * fr  # saturation 1: p = 0
    r = np.choose(i, [val, q_, p, p, t, val])
    g = np.choose(i, [t, val, val, q_, p, p])
    b = np.choose(i, [p, p, t, val, val, q_])
    return _saturate_u8(np.stack([b, g, r], axis=2).astype(np.float64) * 255.0)


def write_sweep_results(result_dir, tasks, w_errors, v_errors, reproject_errors, w=None, v=None, k=None):
    """errors.csv with the reference's header plus the per-quantity CSVs (one line per task)"""
    with open(result_dir + "/errors.csv", "w") as f:
        f.write("task,error_w,error_v,reproject_error\n")
        for t, ew, ev, er in zip(tasks, w_errors, v_errors, reproject_errors):
            f.write("%s,%.17g,%.17g,%.17g\n" % (t, np.mean(ew), np.mean(ev), np.mean(er)))
    for name, arr in (("error_w.csv", w_errors), ("error_v.csv", v_errors), ("reproject_errors.csv", reproject_errors), ("w.csv", w), ("v.csv", v), ("k.csv", k)):
        if arr is not None:
            with open(result_dir + "/" + name, "w") as f:
                for row in arr:
                    f.write(",".join("%.17g" % x for x in np.ravel(row)) + "\n")


# ---- example archive in the reference's layout (matlab take_sequence.m:27-94, start_generating.m:38-42) -----------
def write_example_archive(task_dir, K, gamma, v, w, k, frames):
    """frames: two dicts with rs_image (rows, cols, 3 BGR), R (rows, 3, 3), t (rows, 3), world (rows, cols, 3).  Writes
    <task_dir>/{v,w,gamma,k}.csv and <task_dir>/images/{A.csv, N_rs.png, N_rs_t.csv, N_rs_r.csv, N_rs_unproject_{x,y,z}.csv} and, when
    a frame carries gs_image, N_initial_gs.png."""
    import os

    os.makedirs(task_dir + "/images", exist_ok=True)
    write_matrix_csv(task_dir + "/v.csv", np.asarray(v).reshape(1, 3))
    write_matrix_csv(task_dir + "/w.csv", np.asarray(w).reshape(1, 3))
    write_matrix_csv(task_dir + "/gamma.csv", [[gamma]])
    write_matrix_csv(task_dir + "/k.csv", [[k]])
    img = task_dir + "/images/"
    fx, fy, cx, cy = K
    write_matrix_csv(img + "A.csv", [[fx, 0, cx], [0, fy, cy], [0, 0, 1]])
    for n, fr in enumerate(frames, start=1):
        rows = fr["rs_image"].shape[0]
        write_png(img + "%d_rs.png" % n, fr["rs_image"])
        if fr.get("gs_image") is not None:  # main.cc:626 / :646 <n>_initial_gs.png
            write_png(img + "%d_initial_gs.png" % n, fr["gs_image"])
        write_matrix_csv(img + "%d_rs_t.csv" % n, fr["t"])
        write_matrix_csv(img + "%d_rs_r.csv" % n, np.asarray(fr["R"]).reshape(rows, 9))
        for c, ax in enumerate("xyz"):
            write_matrix_csv(img + "%d_rs_unproject_%s.csv" % (n, ax), fr["world"][:, :, c])


def load_example_archive(task_dir):
    """the reading side of setupCameraSynthetic (main.cc:613-671) + the task's ground truth (main.cc:216-256)"""
    img = task_dir + "/images/"
    truth = load_task_truth(task_dir)
    K, Kmat = load_intrinsics(img + "A.csv")
    frames = []
    for n in (1, 2):
        rs = read_png(img + "%d_rs.png" % n)
        rows, cols = rs.shape[:2]
        R, t = load_poses(img + "%d_rs_t.csv" % n, img + "%d_rs_r.csv" % n, rows)
        world = load_unprojection(*[img + "%d_rs_unproject_%s.csv" % (n, ax) for ax in "xyz"], rows, cols)
        import os

        gs = read_png(img + "%d_initial_gs.png" % n) if os.path.exists(img + "%d_initial_gs.png" % n) else None
        frames.append(dict(rs_image=rs, gs_image=gs, R=R, t=t, world=world))
    return dict(K=K, Kmat=Kmat, truth=truth, frames=frames)
