"""Analytic synthetic rolling-shutter flow generator (SURVEY.md section 8 d).

Produces model-consistent dense optical flow for a known scene depth Z(x, y), motion (v, w, k) and
readout ratio gamma, plus optional DeepFlow-like noise / outliers.  This replaces the reference's
MATLAB renderer (matlab_synthetic_data/*.m), which needs a closed-source MEX and mesh.

Data tooling for tests and bench (numpy, host side) -- not part of the solve.
"""
import numpy as np

# intrinsics (f_x, f_y, c_x, c_y).  galaxy / galaxy_vga are the reference's (camera.cc:189-200).
INTRINSICS = {
    "galaxy_vga": (484.450845764569, 485.345469134313, 313.442094604855, 241.383116350144),
    "galaxy": (1492.41306997746, 1491.09286590722, 949.571146410704, 554.675409391795),
    "hd720": (995.0, 994.0, 633.0, 370.0),
    "uhd": (2984.82613995492, 2982.18573181444, 1899.142292821408, 1109.35081878359),
}

# BASELINE.json configs -> (rows, cols, intrinsics, gamma, noise_px, outlier_frac)
CONFIGS = {
    1: dict(rows=480, cols=640, K="galaxy_vga", gamma=0.8, noise_px=0.0, outliers=0.0),
    2: dict(rows=720, cols=1280, K="hd720", gamma=0.8, noise_px=0.0, outliers=0.0),
    3: dict(rows=1080, cols=1920, K="galaxy", gamma=0.95, noise_px=0.3, outliers=0.10),
    4: dict(rows=2160, cols=3840, K="uhd", gamma=0.95, noise_px=0.3, outliers=0.10),
    5: dict(rows=720, cols=1280, K="hd720", gamma=0.8, noise_px=0.3, outliers=0.10),
}

_GAMMA64 = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(seed, n):
    """n 64-bit outputs of splitmix64 started at `seed` (vectorised, identical to the scalar generator)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + _GAMMA64 * np.arange(1, n + 1, dtype=np.uint64)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(seed, n):
    return (splitmix64(seed, n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normal(seed, n):
    u = uniform01(seed, 2 * n)
    u1 = np.maximum(u[:n], 1e-300)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u[n:])


def scene_depth(rows, cols):
    """Z(x, y) = 1 + 0.35 sin(2 pi 1.3 x/cols) cos(2 pi 0.9 y/rows) + 0.25 x/cols  (mean depth ~ 1)."""
    x = np.arange(cols, dtype=np.float64)[None, :]
    y = np.arange(rows, dtype=np.float64)[:, None]
    return 1.0 + 0.35 * np.sin(2 * np.pi * 1.3 * x / cols) * np.cos(2 * np.pi * 0.9 * y / rows) + 0.25 * (x / cols)


def make_flow(rows, cols, K, v, w, k=0.0, gamma=0.8, noise_px=0.0, outliers=0.0, seed=0x5EED0000, depth=None, _model_only=False):
    """Dense model-consistent RS flow image (rows, cols, 2) in pixels, row-major, plus ground truth.

    Model (minimal.cc:257-266):  u = beta (A v rho + B w),  u = flow_px * gamma / f,
    alpha = 1 + gamma flow_y/h, alpha_k = 0.5[(1 + gamma (y + flow_y)/h)^2 - (gamma y/h)^2], h = rows,
    beta = 2 (alpha + k alpha_k) / (2 + k).  alpha depends on the flow itself -> fixed point.
    """
    fx, fy, cx, cy = K
    v = np.asarray(v, dtype=np.float64)
    w = np.asarray(w, dtype=np.float64)
    Z = scene_depth(rows, cols) if depth is None else np.asarray(depth, dtype=np.float64)
    rho = 1.0 / Z
    xi = np.arange(cols, dtype=np.float64)[None, :]
    yj = np.arange(rows, dtype=np.float64)[:, None]
    x = (xi - cx) / fx + 0.0 * yj
    y = (yj - cy) / fy + 0.0 * xi
    m0 = (v[0] - x * v[2]) * rho + (-x * y) * w[0] + (1 + x * x) * w[1] + (-y) * w[2]
    m1 = (v[1] - y * v[2]) * rho + (-(1 + y * y)) * w[0] + (x * y) * w[1] + x * w[2]
    h = float(rows)
    fyy = np.zeros((rows, cols))
    for _ in range(60):
        alpha = 1.0 + gamma * fyy / h
        part1 = gamma * yj / h
        part2 = 1.0 + gamma * (yj + fyy) / h
        alpha_k = 0.5 * (part2 * part2 - part1 * part1)
        beta = 2.0 * (alpha + k * alpha_k) / (2.0 + k)
        new = beta * m1 * fy / gamma
        if np.max(np.abs(new - fyy)) < 1e-15:
            fyy = new
            break
        fyy = new
    alpha = 1.0 + gamma * fyy / h
    part1 = gamma * yj / h
    part2 = 1.0 + gamma * (yj + fyy) / h
    alpha_k = 0.5 * (part2 * part2 - part1 * part1)
    beta = 2.0 * (alpha + k * alpha_k) / (2.0 + k)
    fxx = beta * m0 * fx / gamma
    flow = np.stack([fxx, fyy], axis=-1)
    truth = dict(v=v, w=w, k=float(k), gamma=float(gamma), Z=Z, K=K)
    if _model_only:
        return np.ascontiguousarray(flow), truth
    flow, outlier_mask = add_noise(flow, noise_px, outliers, seed)
    truth["outlier_mask"] = outlier_mask
    return np.ascontiguousarray(flow), truth


def add_noise(flow, noise_px, outliers, seed):
    """DeepFlow-like degradation of a model flow field: Gaussian noise (pixels) and a fraction of uniform +-30 px outliers, both from
    splitmix64(seed) -- the second half of make_flow, callable on its own so that a SEQUENCE of pairs (one noise seed each) shares
    the model field.  Returns (flow, outlier mask)."""
    rows, cols = flow.shape[:2]
    n = rows * cols
    if noise_px > 0.0:
        flow = flow + noise_px * normal(seed + 1, 2 * n).reshape(rows, cols, 2)
    outlier_mask = np.zeros((rows, cols), dtype=bool)
    if outliers > 0.0:
        outlier_mask = uniform01(seed + 2, n).reshape(rows, cols) < outliers
        rnd = (uniform01(seed + 3, 2 * n).reshape(rows, cols, 2) * 2.0 - 1.0) * 30.0
        flow = np.where(outlier_mask[..., None], rnd, flow)
    return flow, outlier_mask


def make_flow_sequence(cfg, seeds):
    """flow images (rows, cols, 2) of BASELINE config `cfg` for several data seeds: the same scene and motion, one noise / outlier
    realisation per seed (each identical to make_config(cfg, seed=s)["flow_img"]); the model field is computed once"""
    c = CONFIGS[cfg]
    v, w, k = default_motion()
    K = INTRINSICS[c["K"]]
    base, truth = make_flow(c["rows"], c["cols"], K, v, w, k, c["gamma"], _model_only=True)
    imgs = [np.ascontiguousarray(add_noise(base, c["noise_px"], c["outliers"], s_)[0]) for s_ in seeds]
    return imgs, dict(rows=c["rows"], cols=c["cols"], K=K, gamma=c["gamma"], truth=truth)


def flatten_numpy(flow_img, K, gamma, thr=1e-10):
    """Host restatement of the caller glue main.cc:398-444 (column-major scan, shrinking variant
    errorMeasure.cpp:96-97) + getAlpha/getAlphaK (minimal.cc:179-197), for building solver inputs.
    Returns q, u (normalised, n x 2), alpha, alpha_k (n), and the column-major pixel index of each point."""
    fx, fy, cx, cy = K
    rows, cols = flow_img.shape[:2]
    f = np.transpose(flow_img, (1, 0, 2)).reshape(-1, 2)  # column-major scan: i over cols outer, j inner
    ii = np.repeat(np.arange(cols, dtype=np.float64), rows)
    jj = np.tile(np.arange(rows, dtype=np.float64), cols)
    keep = (f[:, 0] * f[:, 0] + f[:, 1] * f[:, 1]) > thr
    f, ii, jj = f[keep], ii[keep], jj[keep]
    q = np.stack([(ii - cx) * 1.0 / fx, (jj - cy) * 1.0 / fy], axis=1)
    u = np.stack([f[:, 0] * gamma / fx, f[:, 1] * gamma / fy], axis=1)
    h = float(rows)
    alpha = 1 + gamma * f[:, 1] / h
    part1 = gamma * jj / h
    part2 = 1.0 + gamma * (jj + f[:, 1]) / h
    alpha_k = 0.5 * (part2 * part2 - part1 * part1)
    pix = np.nonzero(keep)[0]
    return (np.ascontiguousarray(q), np.ascontiguousarray(u), np.ascontiguousarray(alpha),
            np.ascontiguousarray(alpha_k), pix)


def default_motion():
    """examples/README.md:20 first synthetic example: v = (0.03, 0.03, 0) * mean depth, w = (0, 0, 0.5 deg), k = 0."""
    return np.array([0.03, 0.03, 0.0]), np.array([0.0, 0.0, np.deg2rad(0.5)]), 0.0


def make_config(cfg, seed=None, v=None, w=None, k=None, rows=None, cols=None):
    """Inputs of one BASELINE.json config: flow image + flattened solver inputs + truth."""
    c = dict(CONFIGS[cfg])
    if rows is not None:
        c["rows"] = rows
    if cols is not None:
        c["cols"] = cols
    dv, dw, dk = default_motion()
    v = dv if v is None else v
    w = dw if w is None else w
    k = dk if k is None else k
    K = INTRINSICS[c["K"]]
    if rows is not None or cols is not None:  # keep the principal point centred for resized instances
        base_r, base_c = CONFIGS[cfg]["rows"], CONFIGS[cfg]["cols"]
        sc = c["cols"] / base_c
        K = (K[0] * sc, K[1] * sc, K[2] * sc, K[3] * (c["rows"] / base_r))
    seed = (0x5EED0000 + cfg) if seed is None else seed
    flow, truth = make_flow(c["rows"], c["cols"], K, v, w, k, c["gamma"], c["noise_px"], c["outliers"], seed)
    q, u, alpha, alpha_k, pix = flatten_numpy(flow, K, c["gamma"])
    return dict(flow_img=flow, q=q, u=u, alpha=alpha, alpha_k=alpha_k, pix=pix, truth=truth, rows=c["rows"],
                cols=c["cols"], K=K, gamma=c["gamma"])
