"""ctypes binding of the CPU ORACLE (test infrastructure -- see oracle/rsdsfm_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
PARITY UNPINNED: see the header of oracle/rsdsfm_oracle.c.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

TERMINATION = {0: "gradient", 1: "parameter", 2: "function", 3: "max_iter", 4: "failure", 5: "min_radius"}


class LmSummary(C.Structure):
    _fields_ = [
        ("num_iterations", C.c_int32),
        ("num_successful_steps", C.c_int32),
        ("num_unsuccessful_steps", C.c_int32),
        ("termination", C.c_int32),
        ("initial_cost", C.c_double),
        ("final_cost", C.c_double),
        ("final_radius", C.c_double),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class RansacOut(C.Structure):
    _fields_ = [
        ("num_inliers", C.c_int64),
        ("best_trial", C.c_int32),
        ("_pad", C.c_int32),
        ("w", C.c_double * 3),
        ("v", C.c_double * 3),
        ("k", C.c_double),
        ("inlier_error", C.c_double),
        ("inlier_idx", C.c_void_p),
        ("inliers", C.c_void_p),
        ("alpha", C.c_void_p),
        ("alpha_k", C.c_void_p),
        ("mask", C.c_void_p),
        ("inv_depth", C.c_void_p),
        ("trial_count", C.c_void_p),
        ("trial_err", C.c_void_p),
        ("trial_vel", C.c_void_p),
        ("trial_steps", C.c_void_p),
    ]


def _cpu_has_fma():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return " fma " in line + " "
    except OSError:
        pass
    return False


# Two builds of the same source (see the header of rsdsfm_oracle.c):
#   "reference" (default): the reference's arithmetic -- no fused multiply-add anywhere (-ffp-contract=off, no fma() calls).
#                          The pinned target of every parity test and the checker of librsdsfm_hip.so.
#   "fused":               -DRSO_FUSED=1, fma() at the places the opt-in librsdsfm_hip_fused.so fuses; checker of that
#                          library only.  -mfma makes the calls single instructions; on a host without FMA3 they go
#                          through libm's exact software fma (same bits, slower), and such a host must not load a
#                          library built with -mfma elsewhere, hence the file-name suffix.
CFLAGS = ["-O2", "-ffp-contract=off", "-fPIC", "-std=c99"]
_FUSED_FLAGS = ["-DRSO_FUSED=1"] + (["-mfma"] if _cpu_has_fma() else [])
_FUSED_SUFFIX = "_fused" if _cpu_has_fma() else "_fused_nofma"
_ARITH = "reference"
_LIBS = {}


def _so_path(arith="reference", omp=False):
    return os.path.join(_HERE, "librsdsfm_oracle%s%s.so" % ("_omp" if omp else "", _FUSED_SUFFIX if arith == "fused" else ""))


def build(force=False, arith="reference", omp=False):
    so = _so_path(arith, omp)
    src = os.path.join(_HERE, "rsdsfm_oracle.c")
    hdr = os.path.join(_HERE, "rsdsfm_oracle.h")
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        flags = CFLAGS + (_FUSED_FLAGS if arith == "fused" else []) + (["-fopenmp"] if omp else [])
        tmp = so + ".tmp%d" % os.getpid()
        subprocess.check_call(["gcc"] + flags + ["-shared", "-o", tmp, src, "-lm"])
        os.replace(tmp, so)
    return so


def set_arithmetic(arith):
    """selects which build of the oracle lib() returns: "reference" (default, unfused) or "fused" """
    global _ARITH
    assert arith in ("reference", "fused")
    _ARITH = arith


class arithmetic:
    """context manager: `with oracle_py.arithmetic("fused"): ...`"""

    def __init__(self, arith):
        self.arith = arith

    def __enter__(self):
        self.prev = _ARITH
        set_arithmetic(self.arith)

    def __exit__(self, *a):
        set_arithmetic(self.prev)


_LIB_OMP = None


def lib_omp():
    """all-cores (OpenMP) build of the same source (reference arithmetic); only bench.py's all-cores CPU baseline uses it"""
    global _LIB_OMP
    if _LIB_OMP is None:
        _LIB_OMP = C.CDLL(build(omp=True))
    return _LIB_OMP


def estimate_inverse_depths_all_cores(q, u, v, w, k, alpha, alpha_k, mode=1):
    q, u, alpha, alpha_k = _f64(q), _f64(u), _f64(alpha), _f64(alpha_k)
    n = q.shape[0]
    rho = np.empty(n)
    sm = LmSummary()
    rc = lib_omp().rso_estimate_inverse_depths(_p(q), _p(u), C.c_int64(n), _v3(v), _v3(w), C.c_double(k), _p(alpha), _p(alpha_k), int(mode), _p(rho), C.byref(sm))
    assert rc == 0
    return rho, sm.as_dict()


class all_cores:
    """context manager: inside, every wrapper below calls the OpenMP build (reference arithmetic; the per-pixel loops of the depth
    solve, the RANSAC scoring and the refinement passes run on the thread team, sums by OpenMP reductions -- a TIMING vehicle for
    bench.py's all-cores CPU baseline, never the parity target)"""

    def __enter__(self):
        global _FORCE
        L = lib_omp()
        L.rso_score.restype = C.c_int64
        L.rso_flatten.restype = C.c_int64
        self.prev, _FORCE = _FORCE, L

    def __exit__(self, *a):
        global _FORCE
        _FORCE = self.prev


_FORCE = None


_REF = None


def lib_ref():
    """librsdsfm_cpu_ref.so: the reference-STRUCTURED single-thread CPU baseline (oracle/rsdsfm_cpu_ref.cpp)"""
    global _REF
    if _REF is None:
        build()  # it links against the oracle library
        so = os.path.join(_HERE, "librsdsfm_cpu_ref.so")
        src = os.path.join(_HERE, "rsdsfm_cpu_ref.cpp")
        deps = [src, os.path.join(_HERE, "rsdsfm_oracle.h")]
        if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
            tmp = so + ".tmp%d" % os.getpid()
            subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-fPIC", "-Wall", "-Wextra", "-std=c++14", "-shared", "-o", tmp, src,
                                   "-L" + _HERE, "-lrsdsfm_oracle", "-Wl,-rpath," + _HERE, "-lm"])
            os.replace(tmp, so)
        _REF = C.CDLL(so)
    return _REF


def estimate_inverse_depths_reference_structured(q, u, v, w, k, alpha, alpha_k):
    q, u, alpha, alpha_k = _f64(q), _f64(u), _f64(alpha), _f64(alpha_k)
    n = q.shape[0]
    rho = np.empty(n)
    sm = LmSummary()
    rc = lib_ref().rsr_estimate_inverse_depths(_p(q), _p(u), C.c_int64(n), _v3(v), _v3(w), C.c_double(k), _p(alpha), _p(alpha_k), _p(rho), C.byref(sm))
    if rc != 0:
        raise RuntimeError("rsr_estimate_inverse_depths failed rc=%d" % rc)
    return rho, sm.as_dict()


def ransac_reference_structured(q, u, alpha, alpha_k, use_alpha_k, iterations, tolerance, samples, k_sign_mode=0):
    """minimal::ransac with one reference-structured depth solve (problem build + solve + teardown) per trial"""
    q, u, alpha, alpha_k = _f64(q), _f64(u), _f64(alpha), _f64(alpha_k)
    n, T = q.shape[0], int(iterations)
    smp = np.ascontiguousarray(samples, dtype=np.int32).reshape(-1)
    bufs = dict(inlier_idx=np.zeros(n, dtype=np.int64), inliers=np.zeros((n, 3)), alpha=np.zeros(n), alpha_k=np.zeros(n), mask=np.zeros(n, dtype=np.uint8),
                inv_depth=np.zeros(n), trial_count=np.zeros(max(T, 1), dtype=np.int64), trial_err=np.zeros(max(T, 1)), trial_steps=np.zeros(max(T, 1), dtype=np.int32))
    out = RansacOut()
    for name, arr in bufs.items():
        setattr(out, name, arr.ctypes.data)
    rc = lib_ref().rsr_ransac(_p(q), _p(u), _p(alpha), _p(alpha_k), C.c_int64(n), int(use_alpha_k), C.c_int32(T), C.c_double(tolerance), _p(smp), int(k_sign_mode), C.byref(out))
    if rc != 0:
        raise RuntimeError("rsr_ransac failed rc=%d" % rc)
    m = int(out.num_inliers)
    return dict(num_inliers=m, best_trial=int(out.best_trial), w=np.array(out.w[:]), v=np.array(out.v[:]), k=float(out.k), inlier_error=float(out.inlier_error),
                inlier_idx=bufs["inlier_idx"][:m].copy(), inliers=bufs["inliers"][:m].copy(), alpha=bufs["alpha"][:m].copy(), alpha_k=bufs["alpha_k"][:m].copy(),
                mask=bufs["mask"], inv_depth=bufs["inv_depth"], trial_count=bufs["trial_count"][:T], trial_err=bufs["trial_err"][:T], trial_steps=bufs["trial_steps"][:T])


def refine_reference_structured(flow, inliers, alpha, alpha_k, v, w, k, const_acceleration=False, flow_index_mode=0, inlier_idx=None):
    flow, inliers, alpha, alpha_k = _f64(flow), _f64(inliers), _f64(alpha), _f64(alpha_k)
    m = inliers.shape[0]
    idx = None if inlier_idx is None else np.ascontiguousarray(inlier_idx, dtype=np.int64)
    out = np.empty((m, 3))
    vo, wo, ko = (C.c_double * 3)(), (C.c_double * 3)(), C.c_double()
    sm = LmSummary()
    rc = lib_ref().rsr_refine(_p(flow), C.c_int64(flow.shape[0]), C.c_int64(m), _p(inliers), _p(alpha), _p(alpha_k), None if idx is None else _p(idx), _v3(v), _v3(w),
                              C.c_double(k), int(const_acceleration), int(flow_index_mode), _p(out), vo, wo, C.byref(ko), C.byref(sm))
    if rc != 0:
        raise RuntimeError("rsr_refine failed rc=%d" % rc)
    return dict(inliers=out, v=np.array(vo[:]), w=np.array(wo[:]), k=ko.value, summary=sm.as_dict())


VARIANTS = {"ftol": 0, "jacobi": 1, "mindiag": 2, "dsq": 3, "radius": 4, "ftol_lt": 5, "svd_sign": 6}


class variant:
    """sensitivity study only: `with oracle.variant(ftol=1): ...` replaces ONE recalled detail of the Ceres 1.14 loop by an alternative
    reading (rso_set_variant, oracle/rsdsfm_oracle.c) for the calls inside; everything is reset on exit"""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        for k, v in self.kw.items():
            assert lib().rso_set_variant(C.c_int(VARIANTS[k]), C.c_int(int(v))) == 0
        return self

    def __exit__(self, *a):
        for k in self.kw:
            lib().rso_set_variant(C.c_int(VARIANTS[k]), C.c_int(0))


def lib():
    if _FORCE is not None:
        return _FORCE
    if _ARITH not in _LIBS:
        # RSO_ORACLE_LIB: an alternative build of the same source (e.g. `make -C oracle asan` run under LD_PRELOAD=libasan.so)
        alt = os.environ.get("RSO_ORACLE_LIB") if _ARITH == "reference" else None
        L = C.CDLL(alt or build(arith=_ARITH))
        L.rso_score.restype = C.c_int64
        L.rso_flatten.restype = C.c_int64
        _LIBS[_ARITH] = L
    return _LIBS[_ARITH]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _v3(a):
    return (C.c_double * 3)(*[float(x) for x in a])


def get_alpha(flow_px, h, gamma):
    flow_px = _f64(flow_px)
    n = flow_px.shape[0]
    out = np.empty(n)
    lib().rso_get_alpha(_p(flow_px), C.c_int64(n), C.c_double(h), C.c_double(gamma), _p(out))
    return out


def get_alpha_k(q_px, flow_px, h, gamma):
    q_px, flow_px = _f64(q_px), _f64(flow_px)
    n = flow_px.shape[0]
    out = np.empty(n)
    lib().rso_get_alpha_k(_p(q_px), _p(flow_px), C.c_int64(n), C.c_double(h), C.c_double(gamma), _p(out))
    return out


def calculate_velocities(q9, u9, alpha9, alpha_k9, use_alpha_k=False, k_sign_mode=0):
    q9, u9, alpha9, alpha_k9 = _f64(q9), _f64(u9), _f64(alpha9), _f64(alpha_k9)
    w = (C.c_double * 3)()
    v = (C.c_double * 3)()
    k = C.c_double()
    rc = lib().rso_calculate_velocities(_p(q9), _p(u9), _p(alpha9), _p(alpha_k9), int(use_alpha_k), int(k_sign_mode), w, v, C.byref(k))
    return np.array(w[:]), np.array(v[:]), k.value, rc


def residual(x, y, ux, uy, alpha, alpha_k, v, w, k, rho):
    r = (C.c_double * 2)()
    d = C.c_double
    lib().rso_residual(d(x), d(y), d(ux), d(uy), d(alpha), d(alpha_k), _v3(v), _v3(w), d(k), d(rho), r)
    return np.array(r[:])


def estimate_inverse_depths(q, u, v, w, k, alpha, alpha_k, mode=1):
    q, u, alpha, alpha_k = _f64(q), _f64(u), _f64(alpha), _f64(alpha_k)
    n = q.shape[0]
    rho = np.empty(n)
    sm = LmSummary()
    rc = lib().rso_estimate_inverse_depths(_p(q), _p(u), C.c_int64(n), _v3(v), _v3(w), C.c_double(k), _p(alpha), _p(alpha_k), int(mode), _p(rho), C.byref(sm))
    assert rc == 0
    return rho, sm.as_dict()


def score(q, u, alpha, alpha_k, v, w, k, rho, tol):
    q, u, alpha, alpha_k, rho = _f64(q), _f64(u), _f64(alpha), _f64(alpha_k), _f64(rho)
    n = q.shape[0]
    mask = np.empty(n, dtype=np.uint8)
    err = C.c_double()
    cnt = lib().rso_score(_p(q), _p(u), _p(alpha), _p(alpha_k), C.c_int64(n), _v3(v), _v3(w), C.c_double(k), _p(rho), C.c_double(tol), _p(mask), C.byref(err))
    return int(cnt), err.value, mask


class LmaStats(C.Structure):
    _fields_ = [
        ("listed_clamped", C.c_int64),
        ("listed_near", C.c_int64),
        ("fallback", C.c_int32),
        ("fallback_reason", C.c_int32),
        ("margin_use_max", C.c_double),
        ("rho_diff_max", C.c_double),
        ("flips_unguarded", C.c_int64),
        ("flips_listed", C.c_int64),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def lma_trial(q, u, alpha, alpha_k, v, w, k, tol=-1.0, study=False):
    """one depth solve on the analytic LM trajectory (mode 2) and, with tol >= 0, its inlier score; study: also the distance to mode 1"""
    q, u, alpha, alpha_k = _f64(q), _f64(u), _f64(alpha), _f64(alpha_k)
    n = q.shape[0]
    rho = np.empty(n)
    mask = np.zeros(n, dtype=np.uint8)
    cnt, err = C.c_int64(), C.c_double()
    sm, st = LmSummary(), LmaStats()
    rc = lib().rso_lma_trial(_p(q), _p(u), _p(alpha), _p(alpha_k), C.c_int64(n), _v3(v), _v3(w), C.c_double(k), C.c_double(tol), _p(rho), _p(mask),
                             C.byref(cnt), C.byref(err), C.byref(sm), C.byref(st), int(bool(study)))
    assert rc == 0
    return dict(rho=rho, mask=mask, count=int(cnt.value), err=err.value, summary=sm.as_dict(), stats=st.as_dict())


def one_lm_step(q1, u1, alpha, alpha_k, v, w, k, radius=1e4):
    """rho of one pixel after the first LM step from rho = 1 (mode 1's arithmetic)"""
    L = lib()
    L.rso_one_lm_step.restype = C.c_double
    d = C.c_double
    return float(L.rso_one_lm_step(d(q1[0]), d(q1[1]), d(u1[0]), d(u1[1]), d(alpha), d(alpha_k), _v3(v), _v3(w), d(k), d(radius)))


def lma_last_stats():
    """totals over the trials of the last ransac(depth_mode=2)"""
    st = LmaStats()
    lib().rso_lma_last_stats(C.byref(st))
    return st.as_dict()


def sample_indices(n, trials, seed):
    out = np.empty(trials * 9, dtype=np.int32)
    lib().rso_sample_indices(C.c_int64(n), C.c_int32(trials), C.c_uint64(seed), _p(out))
    return out.reshape(trials, 9)


def ransac(q, u, alpha, alpha_k, use_alpha_k, iterations, tol, samples, depth_mode=1, k_sign_mode=0):
    q, u, alpha, alpha_k = _f64(q), _f64(u), _f64(alpha), _f64(alpha_k)
    samples = np.ascontiguousarray(samples, dtype=np.int32).reshape(-1)
    n = q.shape[0]
    T = int(iterations)
    bufs = dict(
        inlier_idx=np.zeros(n, dtype=np.int64),
        inliers=np.zeros((n, 3)),
        alpha=np.zeros(n),
        alpha_k=np.zeros(n),
        mask=np.zeros(n, dtype=np.uint8),
        inv_depth=np.zeros(n),
        trial_count=np.zeros(max(T, 1), dtype=np.int64),
        trial_err=np.zeros(max(T, 1)),
        trial_vel=np.zeros((max(T, 1), 7)),
        trial_steps=np.zeros(max(T, 1), dtype=np.int32),
    )
    out = RansacOut()
    for name, arr in bufs.items():
        setattr(out, name, arr.ctypes.data)
    rc = lib().rso_ransac(_p(q), _p(u), _p(alpha), _p(alpha_k), C.c_int64(n), int(use_alpha_k), C.c_int32(T), C.c_double(tol), _p(samples), int(depth_mode), int(k_sign_mode), C.byref(out))
    if rc != 0:
        raise RuntimeError("rso_ransac failed rc=%d" % rc)
    m = int(out.num_inliers)
    return dict(
        num_inliers=m,
        best_trial=int(out.best_trial),
        w=np.array(out.w[:]),
        v=np.array(out.v[:]),
        k=float(out.k),
        inlier_error=float(out.inlier_error),
        inlier_idx=bufs["inlier_idx"][:m].copy(),
        inliers=bufs["inliers"][:m].copy(),
        alpha=bufs["alpha"][:m].copy(),
        alpha_k=bufs["alpha_k"][:m].copy(),
        mask=bufs["mask"],
        inv_depth=bufs["inv_depth"],
        trial_count=bufs["trial_count"][:T],
        trial_err=bufs["trial_err"][:T],
        trial_vel=bufs["trial_vel"][:T],
        trial_steps=bufs["trial_steps"][:T],
    )


def refine(flow, inliers, alpha, alpha_k, v, w, k, const_acceleration=False, flow_index_mode=0, inlier_idx=None, trace_rows=0, mode=1):
    """trace_rows > 0: also return `trace`, the per-iteration record laid out like the product's rsdsfm_get_refine_trace.
    mode 1 = the reference's arithmetic (rso_refine, the pinned target); mode 2 = the product's default arithmetic restated (rso_refine_rf:
    radius-factorised Schur sums; also returns `guard` -- 0 or the guard at which the product would run the solve again iterate by iterate --
    and `resolves`)"""
    flow, inliers, alpha, alpha_k = _f64(flow), _f64(inliers), _f64(alpha), _f64(alpha_k)
    m = inliers.shape[0]
    idx = None if inlier_idx is None else np.ascontiguousarray(inlier_idx, dtype=np.int64)
    out = np.empty((m, 3))
    vo, wo, ko = (C.c_double * 3)(), (C.c_double * 3)(), C.c_double()
    sm = LmSummary()
    trace = np.full((trace_rows, 8), np.nan) if trace_rows > 0 else None
    L = lib()
    if trace is not None:
        L.rso_set_refine_trace.restype = None
        L.rso_set_refine_trace(_p(trace), C.c_int(trace_rows))
    guard, resolves = C.c_int32(0), C.c_int32(0)
    try:
        if mode == 2:
            rc = L.rso_refine_rf(_p(flow), C.c_int64(flow.shape[0]), C.c_int64(m), _p(inliers), _p(alpha), _p(alpha_k), None if idx is None else _p(idx), _v3(v), _v3(w), C.c_double(k), int(const_acceleration), int(flow_index_mode), _p(out), vo, wo, C.byref(ko), C.byref(sm), C.byref(guard), C.byref(resolves))
        else:
            rc = L.rso_refine(_p(flow), C.c_int64(flow.shape[0]), C.c_int64(m), _p(inliers), _p(alpha), _p(alpha_k), None if idx is None else _p(idx), _v3(v), _v3(w), C.c_double(k), int(const_acceleration), int(flow_index_mode), _p(out), vo, wo, C.byref(ko), C.byref(sm))
    finally:
        if trace is not None:
            L.rso_set_refine_trace(None, C.c_int(0))
    if rc != 0:
        raise RuntimeError("rso_refine failed rc=%d" % rc)
    res = dict(inliers=out, v=np.array(vo[:]), w=np.array(wo[:]), k=ko.value, summary=sm.as_dict())
    if mode == 2:
        res["guard"], res["resolves"], res["listed_max"] = guard.value, resolves.value, int(L.rso_refine_rf_listed_max())
    if trace is not None:
        res["trace"] = trace
    return res


def flatten(flow_img, fx, fy, cx, cy, gamma, thr=1e-10):
    flow_img = _f64(flow_img)
    rows, cols = flow_img.shape[:2]
    n = rows * cols
    q, u, qpx, fpx = np.empty((n, 2)), np.empty((n, 2)), np.empty((n, 2)), np.empty((n, 2))
    d = C.c_double
    pos = lib().rso_flatten(_p(flow_img), C.c_int32(rows), C.c_int32(cols), d(fx), d(fy), d(cx), d(cy), d(gamma), d(thr), _p(q), _p(u), _p(qpx), _p(fpx))
    return q[:pos].copy(), u[:pos].copy(), qpx[:pos].copy(), fpx[:pos].copy()


def canonicalize_sign(inliers, v):
    inliers = _f64(inliers).copy()
    vv = _v3(v)
    flipped = lib().rso_canonicalize_sign(_p(inliers), C.c_int64(inliers.shape[0]), vv)
    return inliers, np.array(vv[:]), bool(flipped)


def scatter_depth(inliers, fx, fy, cx, cy, rows, cols):
    inliers = _f64(inliers)
    m = inliers.shape[0]
    dm = np.zeros((cols, rows))  # col-major rows x cols  == C-order (cols, rows)
    xs, ys = np.empty(m, dtype=np.int32), np.empty(m, dtype=np.int32)
    d = C.c_double
    lib().rso_scatter_depth(_p(inliers), C.c_int64(m), d(fx), d(fy), d(cx), d(cy), C.c_int32(rows), C.c_int32(cols), _p(dm), _p(xs), _p(ys))
    return dm.T.copy(), xs, ys


def pose_table(v, w, k, gamma, rows):
    R, t = np.empty((rows, 9)), np.empty((rows, 3))
    lib().rso_pose_table(_v3(v), _v3(w), C.c_double(k), C.c_double(gamma), C.c_int32(rows), _p(R), _p(t))
    return R.reshape(rows, 3, 3), t


def jacobi_svd9(Z):
    Z = _f64(Z)
    sv, V = np.empty(9), np.empty((9, 9))
    lib().rso_jacobi_svd9(_p(Z), _p(sv), _p(V))
    return sv, V


def eigvals_general(A):
    A = _f64(A)
    n = A.shape[0]
    re, im = np.empty(n), np.empty(n)
    rc = lib().rso_eigvals_general(_p(A), int(n), _p(re), _p(im))
    assert rc == 0, rc
    return re + 1j * im


def eig_sym3(S):
    S = _f64(S)
    lam, V = np.empty(3), np.empty((3, 3))
    lib().rso_eig_sym3(_p(S), _p(lam), _p(V))
    return lam, V


def depth_preview(inliers, fx, fy, cx, cy, rows, cols):
    inliers = _f64(inliers)
    out = np.zeros((rows, cols), dtype=np.uint8)
    d = C.c_double
    lib().rso_depth_preview(_p(inliers), C.c_int64(inliers.shape[0]), d(fx), d(fy), d(cx), d(cy), C.c_int32(rows), C.c_int32(cols), _p(out))
    return out


def back_project(image_bgr, depth_map, R, t, fx, fy, cx, cy, mode=0, q5_mode=0, want_coords=True):
    """depth_map: (rows, cols) array (as scatter_depth returns it); R: (rows, 3, 3) / (rows, 9); t: (rows, 3)"""
    img = np.ascontiguousarray(image_bgr, dtype=np.uint8)
    rows, cols = img.shape[:2]
    dm = np.ascontiguousarray(np.asarray(depth_map, dtype=np.float64).T)  # column-major rows x cols
    R, t = _f64(np.asarray(R).reshape(rows, 9)), _f64(t)
    gs = np.zeros_like(img)
    c3 = np.zeros((rows, cols, 3), dtype=np.float32) if want_coords else None
    d = C.c_double
    lib().rso_back_project(_p(img), _p(dm), _p(R), _p(t), d(fx), d(fy), d(cx), d(cy), C.c_int32(rows), C.c_int32(cols), int(mode), int(q5_mode), _p(gs), None if c3 is None else _p(c3))
    return gs, c3


def interpolate_cracky(image_bgr, offset=1):
    img = np.ascontiguousarray(image_bgr, dtype=np.uint8)
    rows, cols = img.shape[:2]
    out = np.zeros_like(img)
    lib().rso_interpolate_cracky(_p(img), C.c_int32(rows), C.c_int32(cols), C.c_int32(offset), _p(out))
    return out


def true_flow(world_xyz, R2, t2, fx, fy, cx, cy, q5_mode=0):
    """world_xyz: (rows, cols, 3) world point per pixel of frame 1 (zeros = void); R2: (rows2, 3, 3) / (rows2, 9); t2: (rows2, 3).
    Returns flow (rows, cols, 2) and the winning scanline (rows, cols) int32 (-1 = void)."""
    w = np.asarray(world_xyz, dtype=np.float64)
    rows, cols = w.shape[:2]
    maps = [np.ascontiguousarray(w[:, :, c].T) for c in range(3)]  # column-major rows x cols
    t2 = _f64(t2)
    rows2 = t2.shape[0]
    R2 = _f64(np.asarray(R2).reshape(rows2, 9))
    flow = np.zeros((rows, cols, 2))
    best = np.zeros((rows, cols), dtype=np.int32)
    d = C.c_double
    lib().rso_true_flow(_p(maps[0]), _p(maps[1]), _p(maps[2]), C.c_int32(rows), C.c_int32(cols), _p(R2), _p(t2), C.c_int32(rows2), d(fx), d(fy), d(cx), d(cy), int(q5_mode), _p(flow), _p(best))
    return flow, best


class ReprojectionStats(C.Structure):
    _fields_ = [("scale", C.c_double), ("mean_error", C.c_double), ("sum_error", C.c_double), ("number_outliers", C.c_int64),
                ("scale_inliers", C.c_int64), ("error_inliers", C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def velocity_errors(w_est, v_est, w_true, v_true):
    we, ve = C.c_double(), C.c_double()
    lib().rso_velocity_errors(_v3(w_est), _v3(v_est), _v3(w_true), _v3(v_true), C.byref(we), C.byref(ve))
    return we.value, ve.value


def reprojection_error(est_coords, gt_depth, est_depth, R_abs, t_abs, fx, fy, cx, cy, max_norm=10.0, want_image=True):
    """est_coords: (rows, cols, 3) float32; depth maps: (rows, cols); R_abs: (rows, 3, 3) / (rows, 9); t_abs: (rows, 3)"""
    est = np.ascontiguousarray(est_coords, dtype=np.float32)
    rows, cols = est.shape[:2]
    gd = np.ascontiguousarray(np.asarray(gt_depth, dtype=np.float64).T)
    ed = np.ascontiguousarray(np.asarray(est_depth, dtype=np.float64).T)
    R, t = _f64(np.asarray(R_abs).reshape(rows, 9)), _f64(t_abs)
    st = ReprojectionStats()
    img = np.zeros((rows, cols), dtype=np.uint8) if want_image else None
    d = C.c_double
    lib().rso_reprojection_error(_p(est), _p(gd), _p(ed), _p(R), _p(t), d(fx), d(fy), d(cx), d(cy), C.c_int32(rows), C.c_int32(cols), d(max_norm), C.byref(st), None if img is None else _p(img))
    return st.as_dict(), img
