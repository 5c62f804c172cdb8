"""rsdsfm -- MI355X-native rolling-shutter differential-SfM solver (Python binding of the C ABI).

The package directory is `rs-aware-differential-sfm_amd/`; import it through the repo-root shim
`import rsdsfm`.  Everything here goes through include/rsdsfm.h (ctypes, plain pointers and sizes): the HIP
library `librsdsfm_hip.so` IS the product.  There is no CPU fallback: if the library is missing or no
gfx950 device is usable, loading / Solver() raises.

Mirrors the reference's function boundary (reference: src/minimal.h:79-161, src/nonlinearRefinement.h:37-113):
    Solver.get_alpha / get_alpha_k / calculate_velocities / ransac
    Solver.estimate_inverse_depths / non_linear_refinement
    Solver.flatten / depth_map / pose_table            (caller glue, rsframe.cc:771-800)
"""
import ctypes as C
import os
import sys

import numpy as np

from . import dist, pipeline, synth  # noqa: F401  (multi-GPU driver; analytic data generator used by tests and bench)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RSDSFM_LIB") or os.path.join(_HERE, "librsdsfm_hip.so")
# opt-in build of the same sources with explicit fused multiply-adds in the per-pixel model (csrc/device_math.hpp)
LIB_PATH_FUSED = os.path.join(_HERE, "librsdsfm_hip_fused.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "rsdsfm.h")

OK = 0
ERR_PENDING = -5
DEPTH_CLOSED_FORM, DEPTH_CERES_LM = 0, 1
K_COMPAT, K_FIXED = 0, 1
FLOW_COMPAT_RANK, FLOW_GATHERED = 0, 1
REFINE_TRACE_COLS = 8  # rsdsfm_get_refine_trace
TRACE_REJECTED, TRACE_ACCEPTED, TRACE_INVALID, TRACE_PARAMETER_TOL, TRACE_FUNCTION_TOL, TRACE_ACCEPTED_GRADIENT_TOL = 0.0, 1.0, 2.0, 3.0, 4.0, 5.0
TERMINATION = {0: "gradient", 1: "parameter", 2: "function", 3: "max_iter", 4: "failure", 5: "min_radius"}

_libs = {}


class RsdsfmError(RuntimeError):
    pass


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


class FrameParams(C.Structure):
    _fields_ = [("ransac_trials", C.c_int32), ("use_acceleration_mode", C.c_int32), ("use_refinement", C.c_int32),
                ("depth_mode", C.c_int32), ("k_sign_mode", C.c_int32), ("flow_index_mode", C.c_int32), ("use_global_shutter_mode", C.c_int32),
                ("struct_bytes", C.c_int32), ("ransac_tol", C.c_double),
                ("flow_threshold", C.c_double), ("seed", C.c_uint64)]


class FrameJob(C.Structure):
    _fields_ = [("d_flow_img", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32), ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double),
                ("cy", C.c_double), ("gamma", C.c_double), ("d_depth_map", C.c_void_p), ("d_R", C.c_void_p), ("d_t", C.c_void_p), ("seed", C.c_uint64)]


class FrameResult(C.Structure):
    _fields_ = [("n_points", C.c_int64), ("num_inliers", C.c_int64), ("best_trial", C.c_int32), ("flipped", C.c_int32),
                ("ransac_w", C.c_double * 3), ("ransac_v", C.c_double * 3), ("ransac_k", C.c_double),
                ("w", C.c_double * 3), ("v", C.c_double * 3), ("k", C.c_double), ("refine_summary", LmSummary),
                ("d_inliers", C.c_void_p), ("d_inlier_idx", C.c_void_p), ("d_scanline", C.c_void_p)]


class TiledInfo(C.Structure):
    _fields_ = [("nranks", C.c_int32), ("rank", C.c_int32), ("col0", C.c_int32), ("slab_cols", C.c_int32), ("shard_points", C.c_int64),
                ("shard_inliers", C.c_int64), ("host_syncs", C.c_int32), ("collectives", C.c_int32), ("ransac_rounds", C.c_int32), ("path_flags", C.c_int32)]


ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
ALL_REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
DIST_ID_BYTES = 128


def dist_unique_id():
    """rank 0: the 128-byte RCCL unique id to share with the other ranks (rsdsfm_dist_unique_id)"""
    buf = C.create_string_buffer(DIST_ID_BYTES)
    rc = load_library().rsdsfm_dist_unique_id(buf)
    if rc != OK:
        raise RsdsfmError("rsdsfm_dist_unique_id failed (%d): RCCL not available?" % rc)
    return bytes(buf.raw)


def tiled_slab_bounds(cols, nranks, rank):
    """(col0, slab_cols, stride_cols) of `rank`'s column slab (rsdsfm_tiled_slab_bounds)"""
    c0, sc, per = C.c_int32(), C.c_int32(), C.c_int32()
    rc = load_library().rsdsfm_tiled_slab_bounds(C.c_int32(cols), C.c_int32(nranks), C.c_int32(rank), C.byref(c0), C.byref(sc), C.byref(per))
    if rc != OK:
        raise RsdsfmError("rsdsfm_tiled_slab_bounds failed (%d)" % rc)
    return c0.value, sc.value, per.value


def tiled_shard_bounds(n, nranks, rank):
    """(i0, count, stride) of `rank`'s contiguous shard of an n-point list (rsdsfm_tiled_shard_bounds)"""
    i0, cnt, per = C.c_int64(), C.c_int64(), C.c_int64()
    rc = load_library().rsdsfm_tiled_shard_bounds(C.c_int64(n), C.c_int32(nranks), C.c_int32(rank), C.byref(i0), C.byref(cnt), C.byref(per))
    if rc != OK:
        raise RsdsfmError("rsdsfm_tiled_shard_bounds failed (%d)" % rc)
    return i0.value, cnt.value, per.value


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


def declared_symbols():
    """Names of every function include/rsdsfm.h declares (used by the CPU symbol-export test)."""
    import re

    txt = open(HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rsdsfm_[a-z0-9_]+)\s*\(", txt)))


def load_library(build_if_missing=True, arith="reference"):
    """Loads librsdsfm_hip.so (arith="reference", the default: the reference's unfused arithmetic) or the opt-in
    librsdsfm_hip_fused.so (arith="fused"), building them with hipcc when absent and hipcc exists.  Raises otherwise."""
    if arith in _libs:
        return _libs[arith]
    if arith not in ("reference", "fused"):
        raise RsdsfmError("arith must be 'reference' or 'fused'")
    path = LIB_PATH if arith == "reference" else LIB_PATH_FUSED
    if not os.path.exists(path):
        if not build_if_missing:
            raise RsdsfmError("HIP extension %s is missing (run: python rs-aware-differential-sfm_amd/build.py)" % path)
        from . import build as _build

        _build.build()
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 (same soname as /opt/rocm's).  If this
    # library were loaded first it would pull in the system runtime and a later `import torch` would find "No HIP GPUs".
    # Importing torch first (when it is installed) makes both share torch's runtime.
    if "torch" not in sys.modules and os.environ.get("RSDSFM_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    lib = C.CDLL(path)
    lib.rsdsfm_version.restype = C.c_char_p
    lib.rsdsfm_last_error.restype = C.c_char_p
    lib.rsdsfm_last_error.argtypes = [C.c_void_p]
    lib.rsdsfm_kernel_name.restype = C.c_char_p
    lib.rsdsfm_kernel_name.argtypes = [C.c_char_p]
    lib.rsdsfm_destroy.restype = None
    lib.rsdsfm_destroy.argtypes = [C.c_void_p]
    if lib.rsdsfm_fused_arithmetic() != (1 if arith == "fused" else 0):
        raise RsdsfmError("%s reports the wrong arithmetic mode (stale build?)" % path)
    _libs[arith] = lib
    return lib


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _v3(a):
    return (C.c_double * 3)(*[float(x) for x in a])


def _dp(ptr):
    """device pointer (int / torch data_ptr) -> c_void_p"""
    return C.c_void_p(int(ptr))


class Solver:
    """One context = one HIP device + one stream (single owner).  `stream`: a raw hipStream_t handle to adopt,
    e.g. torch.cuda.current_stream().cuda_stream, or None for a private stream."""

    def __init__(self, device=0, stream=None, arith="reference"):
        """arith: "reference" (default) = librsdsfm_hip.so, the reference's unfused arithmetic; "fused" = the opt-in
        librsdsfm_hip_fused.so (same ABI, explicit fmas in the per-pixel model)"""
        self.lib = load_library(arith=arith)
        self.arith = arith
        self._ctx = C.c_void_p()
        rc = self.lib.rsdsfm_create(C.byref(self._ctx), int(device), C.c_void_p(stream) if stream else None)
        if rc != OK:
            self._ctx = None
            raise RsdsfmError("rsdsfm_create(device=%d) failed with %d (no usable gfx950 device? there is no CPU fallback)" % (device, rc))

    def close(self):
        if getattr(self, "_ctx", None):
            self.lib.rsdsfm_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc, what):
        if rc != OK:
            msg = self.lib.rsdsfm_last_error(self._ctx)
            raise RsdsfmError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))

    def set_depth_variant(self, variant):
        """0 = register-staged fused LM kernel (default), 1 = LDS-DMA double-buffered variant"""
        self._check(self.lib.rsdsfm_set_depth_variant(self._ctx, int(variant)), "rsdsfm_set_depth_variant")

    def set_profiling(self, on):
        self._check(self.lib.rsdsfm_set_profiling(self._ctx, int(bool(on))), "rsdsfm_set_profiling")

    def profile_last_ms(self, what="ransac_lm_round0"):
        ms = C.c_double()
        self._check(self.lib.rsdsfm_profile_last_ms(self._ctx, what.encode(), C.byref(ms)), "rsdsfm_profile_last_ms")
        return ms.value

    def set_refine_stage(self, mode):
        """where the refinement's single-workgroup stage runs: 0 (default) = automatic, 1 = in the next pass's prologue, 2 = a launch of its own
        (rsdsfm_set_refine_stage); never a result"""
        self._check(self.lib.rsdsfm_set_refine_stage(self._ctx, int(mode)), "rsdsfm_set_refine_stage")

    def set_ransac_speculation(self, k0):
        """LM iterations speculated by round 0 of RANSAC's batched depth solves: 0 (default) = automatic, follows the context's previous
        solve (2 when none of its hypotheses went beyond one accepted step, else 3); 2 or 3 = fixed.  Scheduling only: never a result."""
        self._check(self.lib.rsdsfm_set_ransac_speculation(self._ctx, int(k0)), "rsdsfm_set_ransac_speculation")

    def set_ransac_math(self, mode):
        """round 0 of RANSAC's batched depth solves: 0 (default) = in-range cores of sqrt / reciprocal with a restart on an argument out
        of range, 1 = always the standard functions (rsdsfm_set_ransac_math); never a result"""
        self._check(self.lib.rsdsfm_set_ransac_math(self._ctx, int(mode)), "rsdsfm_set_ransac_math")

    def set_lm_arithmetic(self, mode):
        """arithmetic of the depth solves inside a RANSAC: 0 = analytic LM trajectory with guards (default), 1 = iterate by iterate
        (rsdsfm_set_lm_arithmetic); integer outputs never depend on it"""
        self._check(self.lib.rsdsfm_set_lm_arithmetic(self._ctx, int(mode)), "rsdsfm_set_lm_arithmetic")

    def set_refine_arithmetic(self, mode):
        """the joint refinement's arithmetic on its own: 0 (default) = radius-factorised while set_lm_arithmetic is 0, 1 = iterate by iterate"""
        self._check(self.lib.rsdsfm_set_refine_arithmetic(self._ctx, int(mode)), "rsdsfm_set_refine_arithmetic")

    def refine_restarts(self):
        """(refinements on the radius-factorised path, those a guard sent back to the iterate-by-iterate kernels, reduced systems solved
        again from stored sums, the guard that tripped last) -- rsdsfm_refine_restarts"""
        n, r, sv, g = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int32(0)
        self._check(self.lib.rsdsfm_refine_restarts(self._ctx, C.byref(n), C.byref(r), C.byref(sv), C.byref(g)), "rsdsfm_refine_restarts")
        return dict(runs=n.value, restarts=r.value, resolves=sv.value, last_guard=g.value)

    def lma_restarts(self):
        """(RANSAC runs of this context that started over because a guard of the analytic trajectory tripped, bit set of the last guards)"""
        n, g = C.c_int64(0), C.c_int32(0)
        self._check(self.lib.rsdsfm_lma_restarts(self._ctx, C.byref(n), C.byref(g)), "rsdsfm_lma_restarts")
        return int(n.value), int(g.value)

    def lma_count_only(self):
        """(RANSACs of frame solves that ran the count-only form of the analytic pass, those of them that had to fetch error sums)"""
        a, b = C.c_int64(0), C.c_int64(0)
        self._check(self.lib.rsdsfm_lma_count_only(self._ctx, C.byref(a), C.byref(b)), "rsdsfm_lma_count_only")
        return int(a.value), int(b.value)

    def ransac_restarts(self):
        """RANSAC runs of this context that started over with the standard functions (rsdsfm_ransac_restarts)"""
        n = C.c_int64()
        self._check(self.lib.rsdsfm_ransac_restarts(self._ctx, C.byref(n)), "rsdsfm_ransac_restarts")
        return n.value

    def depth_restarts(self):
        """dense depth solves of this context that started over with the standard functions (rsdsfm_depth_restarts)"""
        n = C.c_int64()
        self._check(self.lib.rsdsfm_depth_restarts(self._ctx, C.byref(n)), "rsdsfm_depth_restarts")
        return int(n.value)

    def synchronize(self):
        self._check(self.lib.rsdsfm_synchronize(self._ctx), "rsdsfm_synchronize")

    # ---- minimal:: ----
    def get_alpha(self, flow_px, h, gamma):
        flow_px = _f64(flow_px)
        n = flow_px.shape[0]
        out = np.empty(n)
        self._check(self.lib.rsdsfm_get_alpha(self._ctx, _p(flow_px), C.c_int64(n), C.c_double(h), C.c_double(gamma), _p(out)), "rsdsfm_get_alpha")
        return out

    def get_alpha_k(self, q_px, flow_px, h, gamma):
        q_px, flow_px = _f64(q_px), _f64(flow_px)
        n = flow_px.shape[0]
        out = np.empty(n)
        self._check(self.lib.rsdsfm_get_alpha_k(self._ctx, _p(q_px), _p(flow_px), C.c_int64(n), C.c_double(h), C.c_double(gamma), _p(out)), "rsdsfm_get_alpha_k")
        return out

    def calculate_velocities(self, q, u, alpha, alpha_k, use_alpha_k=False, k_sign_mode=K_COMPAT):
        """q, u: (T, 9, 2) or (9, 2); alpha, alpha_k: (T, 9) or (9,).  Returns w (T,3), v (T,3), k (T,)."""
        q, u, alpha, alpha_k = _f64(q), _f64(u), _f64(alpha), _f64(alpha_k)
        single = q.ndim == 2
        T = 1 if single else q.shape[0]
        w, v, k = np.empty((T, 3)), np.empty((T, 3)), np.empty(T)
        self._check(self.lib.rsdsfm_calculate_velocities(self._ctx, _p(q), _p(u), _p(alpha), _p(alpha_k), C.c_int32(T), int(use_alpha_k), int(k_sign_mode), _p(w), _p(v), _p(k)), "rsdsfm_calculate_velocities")
        return (w[0], v[0], float(k[0])) if single else (w, v, k)

    def ransac(self, q, u, alpha, alpha_k, use_alpha_k, iterations, tolerance, samples=None, seed=0, depth_mode=DEPTH_CERES_LM, k_sign_mode=K_COMPAT, outputs=None):
        """outputs: None = fresh arrays, copies returned; a dict (empty at first) = caller-owned output arrays kept in it and REUSED by the
        next call with the same dict -- the returned arrays are then views into them (what a caller that streams frames does: no page faults
        of fresh arrays, no copies, inside the call)"""
        q, u, alpha, alpha_k = _f64(q), _f64(u), _f64(alpha), _f64(alpha_k)
        n, T = q.shape[0], int(iterations)
        smp = None if samples is None else np.ascontiguousarray(samples, dtype=np.int32).reshape(-1)
        reuse = outputs is not None
        if reuse and outputs.get("_shape") == (n, T):
            bufs = outputs["_bufs"]
        else:
            bufs = dict(
                inlier_idx=np.zeros(n, dtype=np.int64), inliers=np.zeros((n, 3)), alpha=np.zeros(n), alpha_k=np.zeros(n),
                mask=np.zeros(n, dtype=np.uint8), inv_depth=np.zeros(n), trial_count=np.zeros(max(T, 1), dtype=np.int64),
                trial_err=np.zeros(max(T, 1)), trial_vel=np.zeros((max(T, 1), 7)), trial_steps=np.zeros(max(T, 1), dtype=np.int32),
            )
            if reuse:
                outputs["_shape"], outputs["_bufs"] = (n, T), bufs
        out = RansacOut()
        for name, arr in bufs.items():
            # (outputs["only"]: the arrays the caller wants filled -- e.g. what the reference's RansacValues holds: inliers, alpha, alpha_k (+ indices);
            # the others stay NULL and are neither computed for the host nor copied)
            if not reuse or "only" not in outputs or name in outputs["only"]:
                setattr(out, name, arr.ctypes.data)
        self._check(self.lib.rsdsfm_ransac(self._ctx, _p(q), _p(u), _p(alpha), _p(alpha_k), C.c_int64(n), int(use_alpha_k), C.c_int32(T), C.c_double(tolerance), _p(smp), C.c_uint64(seed), int(depth_mode), int(k_sign_mode), C.byref(out)), "rsdsfm_ransac")
        m = int(out.num_inliers)
        tag, hits = C.c_uint64(0), C.c_int64(0)
        self._check(self.lib.rsdsfm_last_ransac_tag(self._ctx, C.byref(tag), C.byref(hits)), "rsdsfm_last_ransac_tag")
        return dict(
            tag=int(tag.value),  # names the device-resident copy of this result: non_linear_refinement(..., tag=...) starts from it
            num_inliers=m, best_trial=int(out.best_trial), w=np.array(out.w[:]), v=np.array(out.v[:]), k=float(out.k),
            inlier_error=float(out.inlier_error), inlier_idx=bufs["inlier_idx"][:m] if reuse else bufs["inlier_idx"][:m].copy(),
            inliers=bufs["inliers"][:m] if reuse else bufs["inliers"][:m].copy(), alpha=bufs["alpha"][:m] if reuse else bufs["alpha"][:m].copy(),
            alpha_k=bufs["alpha_k"][:m] if reuse else bufs["alpha_k"][:m].copy(), mask=bufs["mask"], inv_depth=bufs["inv_depth"],
            trial_count=bufs["trial_count"][:T], trial_err=bufs["trial_err"][:T], trial_vel=bufs["trial_vel"][:T],
            trial_steps=bufs["trial_steps"][:T],
        )

    # ---- nonlinear_refinement:: ----
    def estimate_inverse_depths(self, q, u, v, w, k, alpha, alpha_k, mode=DEPTH_CERES_LM):
        q, u, alpha, alpha_k = _f64(q), _f64(u), _f64(alpha), _f64(alpha_k)
        n = q.shape[0]
        rho = np.empty(n)
        sm = LmSummary()
        self._check(self.lib.rsdsfm_estimate_inverse_depths(self._ctx, _p(q), _p(u), C.c_int64(n), _v3(v), _v3(w), C.c_double(k), _p(alpha), _p(alpha_k), int(mode), _p(rho), C.byref(sm)), "rsdsfm_estimate_inverse_depths")
        return rho, sm.as_dict()

    def estimate_inverse_depth(self, q, v, w, flow, k, alpha, alpha_k, mode=DEPTH_CERES_LM):
        """single-pixel variant (nonlinearRefinement.cc:55-106)"""
        rho, _ = self.estimate_inverse_depths(np.asarray(q).reshape(1, 2), np.asarray(flow).reshape(1, 2), v, w, k, [alpha], [alpha_k], mode)
        return float(rho[0])

    def refine_cache_hits(self):
        """refinements of this context that started from the device-resident outputs of a RANSAC (rsdsfm_refine_from_ransac)"""
        tag, hits = C.c_uint64(0), C.c_int64(0)
        self._check(self.lib.rsdsfm_last_ransac_tag(self._ctx, C.byref(tag), C.byref(hits)), "rsdsfm_last_ransac_tag")
        return int(hits.value)

    def non_linear_refinement(self, flow, inliers, alpha, alpha_k, v, w, k, const_acceleration=False, flow_index_mode=FLOW_COMPAT_RANK, inlier_idx=None, tag=0, out=None):
        """tag: `tag` of the ransac() result these arrays are the UNMODIFIED outputs of (0: upload everything) -- rsdsfm_refine_from_ransac;
        out: a caller-owned (m, 3) array for the refined inliers (None: a fresh one)"""
        flow, inliers, alpha, alpha_k = _f64(flow), _f64(inliers), _f64(alpha), _f64(alpha_k)
        m = inliers.shape[0]
        idx = None if inlier_idx is None else np.ascontiguousarray(inlier_idx, dtype=np.int64)
        if out is None or out.shape != (m, 3) or out.dtype != np.float64 or not out.flags["C_CONTIGUOUS"]:
            out = np.empty((m, 3))
        vo, wo, ko = (C.c_double * 3)(), (C.c_double * 3)(), C.c_double()
        sm = LmSummary()
        if tag:
            self._check(self.lib.rsdsfm_refine_from_ransac(self._ctx, C.c_uint64(int(tag)), _p(flow), C.c_int64(flow.shape[0]), C.c_int64(m), _p(inliers), _p(alpha), _p(alpha_k), _p(idx), _v3(v), _v3(w), C.c_double(k), int(const_acceleration), int(flow_index_mode), _p(out), vo, wo, C.byref(ko), C.byref(sm)), "rsdsfm_refine_from_ransac")
        else:
            self._check(self.lib.rsdsfm_refine(self._ctx, _p(flow), C.c_int64(flow.shape[0]), C.c_int64(m), _p(inliers), _p(alpha), _p(alpha_k), _p(idx), _v3(v), _v3(w), C.c_double(k), int(const_acceleration), int(flow_index_mode), _p(out), vo, wo, C.byref(ko), C.byref(sm)), "rsdsfm_refine")
        return dict(inliers=out, v=np.array(vo[:]), w=np.array(wo[:]), k=ko.value, summary=sm.as_dict())

    def set_refine_trace(self, rows):
        """rows > 0: record the first `rows` LM iterations of every following refinement (rsdsfm_set_refine_trace); 0 = off"""
        self._check(self.lib.rsdsfm_set_refine_trace(self._ctx, C.c_int32(int(rows))), "rsdsfm_set_refine_trace")
        self._refine_trace_rows = int(rows)

    def get_refine_trace(self, rows=None):
        """(rows, 8) array of the last refinement: iteration, cost, candidate cost, model cost change, relative decrease, radius,
        step norm, outcome (TRACE_*); NaN = not computed / no such iteration"""
        rows = getattr(self, "_refine_trace_rows", 0) if rows is None else int(rows)
        out = np.empty((max(rows, 1), REFINE_TRACE_COLS))
        self._check(self.lib.rsdsfm_get_refine_trace(self._ctx, _p(out), C.c_int32(rows)), "rsdsfm_get_refine_trace")
        return out[:rows]

    # ---- caller glue ----
    def flatten(self, flow_img, K, gamma, thr=1e-10):
        flow_img = _f64(flow_img)
        rows, cols = flow_img.shape[:2]
        n = rows * cols
        q, u, a, ak = np.empty((n, 2)), np.empty((n, 2)), np.empty(n), np.empty(n)
        cnt = C.c_int64()
        d = C.c_double
        self._check(self.lib.rsdsfm_flatten(self._ctx, _p(flow_img), C.c_int32(rows), C.c_int32(cols), d(K[0]), d(K[1]), d(K[2]), d(K[3]), d(gamma), d(thr), _p(q), _p(u), _p(a), _p(ak), C.byref(cnt)), "rsdsfm_flatten")
        m = cnt.value
        return q[:m].copy(), u[:m].copy(), a[:m].copy(), ak[:m].copy()

    def depth_map(self, inliers, v, K, rows, cols):
        inl = _f64(inliers).copy()
        m = inl.shape[0]
        vv = _v3(v)
        dm = np.zeros((cols, rows))
        xs, ys = np.empty(m, dtype=np.int32), np.empty(m, dtype=np.int32)
        flipped = C.c_int()
        d = C.c_double
        self._check(self.lib.rsdsfm_depth_map(self._ctx, _p(inl), C.c_int64(m), vv, d(K[0]), d(K[1]), d(K[2]), d(K[3]), C.c_int32(rows), C.c_int32(cols), _p(dm), _p(xs), _p(ys), C.byref(flipped)), "rsdsfm_depth_map")
        return dict(depth_map=dm.T.copy(), inliers=inl, v=np.array(vv[:]), xs=xs, ys=ys, flipped=bool(flipped.value))

    def pose_table(self, v, w, k, gamma, rows):
        R, t = np.empty((rows, 9)), np.empty((rows, 3))
        self._check(self.lib.rsdsfm_pose_table(self._ctx, _v3(v), _v3(w), C.c_double(k), C.c_double(gamma), C.c_int32(rows), _p(R), _p(t)), "rsdsfm_pose_table")
        return R.reshape(rows, 3, 3), t

    # ---- device API (raw device pointers, asynchronous) ----
    def estimate_inverse_depths_dev(self, d_q, d_u, n, v, w, k, d_alpha, d_alpha_k, d_rho, mode=DEPTH_CERES_LM):
        self._check(self.lib.rsdsfm_estimate_inverse_depths_dev(self._ctx, _dp(d_q), _dp(d_u), C.c_int64(n), _v3(v), _v3(w), C.c_double(k), _dp(d_alpha), _dp(d_alpha_k), int(mode), _dp(d_rho)), "rsdsfm_estimate_inverse_depths_dev")

    def prepared_depth_step(self, d_q, d_u, n, v, w, k, d_alpha, d_alpha_k, d_rho, mode=DEPTH_CERES_LM):
        """Returns a zero-argument callable that enqueues one dense depth solve with pre-marshalled arguments
        (keeps the per-step host cost at one foreign call; used by bench.py)."""
        fn = self.lib.rsdsfm_estimate_inverse_depths_dev
        args = (self._ctx, _dp(d_q), _dp(d_u), C.c_int64(n), _v3(v), _v3(w), C.c_double(k), _dp(d_alpha), _dp(d_alpha_k), C.c_int(int(mode)), _dp(d_rho))
        check = self._check

        def call():
            rc = fn(*args)
            if rc != OK:
                check(rc, "rsdsfm_estimate_inverse_depths_dev")

        return call

    def depth_lm_launch_dev(self, d_q, d_u, n, v, w, k, d_alpha, d_alpha_k, d_rho, launch_id=0):
        self._check(self.lib.rsdsfm_depth_lm_launch_dev(self._ctx, _dp(d_q), _dp(d_u), C.c_int64(n), _v3(v), _v3(w), C.c_double(k), _dp(d_alpha), _dp(d_alpha_k), _dp(d_rho), int(launch_id)), "rsdsfm_depth_lm_launch_dev")

    def flatten_dev(self, d_img, rows, cols, K, gamma, d_q, d_u, d_alpha, d_alpha_k, thr=1e-10):
        cnt = C.c_int64()
        d = C.c_double
        self._check(self.lib.rsdsfm_flatten_dev(self._ctx, _dp(d_img), C.c_int32(rows), C.c_int32(cols), d(K[0]), d(K[1]), d(K[2]), d(K[3]), d(gamma), d(thr), _dp(d_q), _dp(d_u), _dp(d_alpha), _dp(d_alpha_k), C.byref(cnt)), "rsdsfm_flatten_dev")
        return cnt.value

    def ransac_dev(self, d_q, d_u, d_alpha, d_alpha_k, n, use_alpha_k, iterations, tolerance, out_ptrs, samples=None, seed=0, depth_mode=DEPTH_CERES_LM, k_sign_mode=K_COMPAT):
        """out_ptrs: dict of DEVICE pointers (inlier_idx, inliers, alpha, alpha_k, mask, inv_depth; missing = not wanted)."""
        T = int(iterations)
        smp = None if samples is None else np.ascontiguousarray(samples, dtype=np.int32).reshape(-1)
        out = RansacOut()
        for name in ("inlier_idx", "inliers", "alpha", "alpha_k", "mask", "inv_depth"):
            setattr(out, name, int(out_ptrs[name]) if out_ptrs.get(name) else None)
        tc = np.zeros(max(T, 1), dtype=np.int64)
        ts = np.zeros(max(T, 1), dtype=np.int32)
        out.trial_count, out.trial_steps = tc.ctypes.data, ts.ctypes.data
        self._check(self.lib.rsdsfm_ransac_dev(self._ctx, _dp(d_q), _dp(d_u), _dp(d_alpha), _dp(d_alpha_k), C.c_int64(n), int(use_alpha_k), C.c_int32(T), C.c_double(tolerance), _p(smp), C.c_uint64(seed), int(depth_mode), int(k_sign_mode), C.byref(out)), "rsdsfm_ransac_dev")
        return dict(num_inliers=int(out.num_inliers), best_trial=int(out.best_trial), w=np.array(out.w[:]), v=np.array(out.v[:]), k=float(out.k),
                    inlier_error=float(out.inlier_error), trial_count=tc[:T], trial_steps=ts[:T])

    def refine_dev(self, d_flow, n_flow, m, d_inl, d_alpha, d_alpha_k, d_idx, v, w, k, const_acceleration, flow_index_mode, d_inl_out):
        vo, wo, ko = (C.c_double * 3)(), (C.c_double * 3)(), C.c_double()
        sm = LmSummary()
        self._check(self.lib.rsdsfm_refine_dev(self._ctx, _dp(d_flow), C.c_int64(n_flow), C.c_int64(m), _dp(d_inl), _dp(d_alpha), _dp(d_alpha_k), _dp(d_idx) if d_idx else None, _v3(v), _v3(w), C.c_double(k), int(const_acceleration), int(flow_index_mode), _dp(d_inl_out), vo, wo, C.byref(ko), C.byref(sm)), "rsdsfm_refine_dev")
        return dict(v=np.array(vo[:]), w=np.array(wo[:]), k=ko.value, summary=sm.as_dict())

    def depth_map_dev(self, d_inl, m, v, K, rows, cols, d_depth_map, d_xs=None, d_ys=None):
        vv = _v3(v)
        flipped = C.c_int()
        d = C.c_double
        self._check(self.lib.rsdsfm_depth_map_dev(self._ctx, _dp(d_inl), C.c_int64(m), vv, d(K[0]), d(K[1]), d(K[2]), d(K[3]), C.c_int32(rows), C.c_int32(cols), _dp(d_depth_map), _dp(d_xs) if d_xs else None, _dp(d_ys) if d_ys else None, C.byref(flipped)), "rsdsfm_depth_map_dev")
        return np.array(vv[:]), bool(flipped.value)

    def pose_table_dev(self, v, w, k, gamma, rows, d_R, d_t):
        self._check(self.lib.rsdsfm_pose_table_dev(self._ctx, _v3(v), _v3(w), C.c_double(k), C.c_double(gamma), C.c_int32(rows), _dp(d_R), _dp(d_t)), "rsdsfm_pose_table_dev")

    def solve_frame_dev(self, d_flow_img, rows, cols, K, gamma, d_depth_map, d_R=None, d_t=None, trials=50, tol=0.05, seed=1,
                        use_acceleration_mode=False, use_refinement=True, depth_mode=DEPTH_CERES_LM, k_sign_mode=K_COMPAT, flow_threshold=1e-10,
                        flow_index_mode=FLOW_COMPAT_RANK, use_global_shutter_mode=False):
        """the whole solve of one frame pair in ONE C-ABI call (rsdsfm_solve_frame_dev).  Defaults = evaluateSingleRun: the
        refinement reads the flow by inlier RANK (quirk Q2, main.cc:457); pass flow_index_mode=FLOW_GATHERED for the flow of each
        inlier's own pixel."""
        prm = FrameParams(int(trials), int(use_acceleration_mode), int(use_refinement), int(depth_mode), int(k_sign_mode),
                          int(flow_index_mode), int(use_global_shutter_mode), 0, float(tol), float(flow_threshold), int(seed))
        res = FrameResult()
        d = C.c_double
        self._check(self.lib.rsdsfm_solve_frame_dev(self._ctx, _dp(d_flow_img), C.c_int32(rows), C.c_int32(cols), d(K[0]), d(K[1]), d(K[2]), d(K[3]),
                                                    d(gamma), C.byref(prm), _dp(d_depth_map), _dp(d_R) if d_R else None, _dp(d_t) if d_t else None,
                                                    C.byref(res)), "rsdsfm_solve_frame_dev")
        return dict(n=int(res.n_points), num_inliers=int(res.num_inliers), best_trial=int(res.best_trial), flipped=bool(res.flipped),
                    ransac_w=np.array(res.ransac_w[:]), ransac_v=np.array(res.ransac_v[:]), ransac_k=float(res.ransac_k),
                    w=np.array(res.w[:]), v=np.array(res.v[:]), k=float(res.k), refine_summary=res.refine_summary.as_dict(),
                    d_inliers=res.d_inliers, d_inlier_idx=res.d_inlier_idx, d_scanline=res.d_scanline)

    # ---- the column-tiled whole solve driven inside the library (RCCL or caller-provided collectives) ----
    def dist_init(self, nranks, rank, unique_id):
        """collective: creates the RCCL communicator of this context from the shared 128-byte id (rsdsfm_dist_init)"""
        self._check(self.lib.rsdsfm_dist_init(self._ctx, C.c_int32(nranks), C.c_int32(rank), C.c_char_p(bytes(unique_id))), "rsdsfm_dist_init")

    def dist_set_transport(self, nranks, rank, all_gather, all_reduce):
        """caller-provided collectives (Python callables taking (d_send, d_recv, bytes_per_rank, stream) / (d_buf, count, stream) and
        returning 0) instead of RCCL; the ctypes thunks are kept alive on the solver"""
        self._ag = ALL_GATHER_FN(lambda user, s_, r_, b_, st: int(all_gather(s_, r_, b_, st)))
        self._ar = ALL_REDUCE_FN(lambda user, p_, n_, st: int(all_reduce(p_, n_, st)))
        self._check(self.lib.rsdsfm_dist_set_transport(self._ctx, C.c_int32(nranks), C.c_int32(rank), self._ag, self._ar, None), "rsdsfm_dist_set_transport")

    def dist_finalize(self):
        self._check(self.lib.rsdsfm_dist_finalize(self._ctx), "rsdsfm_dist_finalize")

    def solve_frame_tiled_dev(self, d_img_slab, rows, cols, K, gamma, d_depth_map, d_R=None, d_t=None, trials=50, tol=0.05, seed=1,
                              use_acceleration_mode=False, use_refinement=True, depth_mode=DEPTH_CERES_LM, k_sign_mode=K_COMPAT,
                              flow_threshold=1e-10, use_global_shutter_mode=False, flow_index_mode=FLOW_COMPAT_RANK):
        """this rank's part of the column-tiled whole solve (rsdsfm_solve_frame_tiled_dev): d_img_slab = this rank's [rows][slab_cols][2]
        slab, cols = width of the whole image; returns the dict of solve_frame_dev (global counts) plus the slab's info.  Defaults as
        solve_frame_dev: the refinement reads the flow by GLOBAL inlier rank (quirk Q2, main.cc:457) -- the columns a rank needs from
        the slabs in front of it are exchanged inside the call; FLOW_GATHERED = the flow of each inlier's own pixel."""
        prm = FrameParams(int(trials), int(use_acceleration_mode), int(use_refinement), int(depth_mode), int(k_sign_mode),
                          int(flow_index_mode), int(use_global_shutter_mode), 0, float(tol), float(flow_threshold), int(seed))
        res, info = FrameResult(), TiledInfo()
        d = C.c_double
        self._check(self.lib.rsdsfm_solve_frame_tiled_dev(self._ctx, _dp(d_img_slab) if d_img_slab else None, C.c_int32(rows), C.c_int32(cols), d(K[0]), d(K[1]),
                                                          d(K[2]), d(K[3]), d(gamma), C.byref(prm), _dp(d_depth_map), _dp(d_R) if d_R else None,
                                                          _dp(d_t) if d_t else None, C.byref(res), C.byref(info)), "rsdsfm_solve_frame_tiled_dev")
        return dict(n=int(res.n_points), num_inliers=int(res.num_inliers), best_trial=int(res.best_trial), flipped=bool(res.flipped),
                    ransac_w=np.array(res.ransac_w[:]), ransac_v=np.array(res.ransac_v[:]), ransac_k=float(res.ransac_k),
                    w=np.array(res.w[:]), v=np.array(res.v[:]), k=float(res.k), refine_summary=res.refine_summary.as_dict(),
                    d_inliers=res.d_inliers, d_inlier_idx=res.d_inlier_idx, d_scanline=res.d_scanline, flow_index_mode=int(flow_index_mode),
                    info={k2: int(getattr(info, k2)) for k2, _ in TiledInfo._fields_ if k2 != "_pad"})

    def estimate_inverse_depths_tiled_dev(self, d_q_shard, d_u_shard, n_total, v, w, k, d_alpha_shard, d_alpha_k_shard, d_inv_depth,
                                          mode=DEPTH_CERES_LM):
        """this rank's part of the row-tiled dense depth solve (rsdsfm_estimate_inverse_depths_tiled_dev): the shard pointers are
        this rank's tiled_shard_bounds slice; d_inv_depth receives all n_total inverse depths.  Returns (lm summary dict or None, info)"""
        sm, info = LmSummary(), TiledInfo()
        opt = lambda p_: _dp(p_) if p_ else None
        self._check(self.lib.rsdsfm_estimate_inverse_depths_tiled_dev(self._ctx, opt(d_q_shard), opt(d_u_shard), C.c_int64(n_total), _v3(v), _v3(w),
                                                                      C.c_double(k), opt(d_alpha_shard), opt(d_alpha_k_shard), C.c_int(int(mode)),
                                                                      opt(d_inv_depth), C.byref(sm), C.byref(info)),
                    "rsdsfm_estimate_inverse_depths_tiled_dev")
        return (sm.as_dict() if int(mode) == DEPTH_CERES_LM else None,
                {k2: int(getattr(info, k2)) for k2, _ in TiledInfo._fields_ if k2 != "_pad"})

    def prepared_frame_solve(self, d_flow_img, rows, cols, K, gamma, d_depth_map, d_R=None, d_t=None, trials=50, tol=0.05,
                             use_acceleration_mode=False, use_refinement=True, depth_mode=DEPTH_CERES_LM, k_sign_mode=K_COMPAT,
                             flow_threshold=1e-10, flow_index_mode=FLOW_COMPAT_RANK, use_global_shutter_mode=False):
        """Returns call(seed) -> FrameResult for repeated whole solves with pre-marshalled arguments (one foreign call per solve;
        the host-side cost of building the argument structures and the result dict -- ~30 us in Python, during which the GPU
        idles -- is paid once).  The returned ctypes struct is reused by the next call."""
        prm = FrameParams(int(trials), int(use_acceleration_mode), int(use_refinement), int(depth_mode), int(k_sign_mode),
                          int(flow_index_mode), int(use_global_shutter_mode), 0, float(tol), float(flow_threshold), 0)
        res = FrameResult()
        d = C.c_double
        fn = self.lib.rsdsfm_solve_frame_dev
        args = (self._ctx, _dp(d_flow_img), C.c_int32(rows), C.c_int32(cols), d(K[0]), d(K[1]), d(K[2]), d(K[3]), d(gamma), C.byref(prm),
                _dp(d_depth_map), _dp(d_R) if d_R else None, _dp(d_t) if d_t else None, C.byref(res))
        check = self._check

        def call(seed):
            prm.seed = seed
            rc = fn(*args)
            if rc != OK:
                check(rc, "rsdsfm_solve_frame_dev")
            return res

        return call

    def set_sequence_lanes(self, lanes):
        """pairs in flight of solve_frames_dev (rsdsfm_set_sequence_lanes): 1..16, 0 = default (3); scheduling only"""
        self._check(self.lib.rsdsfm_set_sequence_lanes(self._ctx, C.c_int32(int(lanes))), "rsdsfm_set_sequence_lanes")

    def set_frame_side_flatten(self, on):
        """where a dense frame's flatten runs (rsdsfm_set_frame_side_flatten): 3 (default) inside the minimal solver's launch (spare
        workgroups flatten while T waves solve), 2 behind the solver, 1 beside it on a second stream, 0 in front of it; scheduling only"""
        self._check(self.lib.rsdsfm_set_frame_side_flatten(self._ctx, int(on)), "rsdsfm_set_frame_side_flatten")

    def prepared_frames_solve(self, jobs, trials=50, tol=0.05, use_acceleration_mode=False, use_refinement=True, depth_mode=DEPTH_CERES_LM,
                              k_sign_mode=K_COMPAT, flow_threshold=1e-10, flow_index_mode=FLOW_COMPAT_RANK, use_global_shutter_mode=False):
        """A SEQUENCE of frame pairs in ONE C-ABI call (rsdsfm_solve_frames_dev), pipelined inside the library.  jobs: list of dicts with
        d_flow_img, rows, cols, K, gamma, d_depth_map and optionally d_R, d_t (device pointers).  Returns call(seeds) -> list of
        FrameResult (the ctypes array is reused by the next call); seeds: one sampler seed per pair."""
        n = len(jobs)
        arr = (FrameJob * n)()
        for a, j in zip(arr, jobs):
            K = j["K"]
            a.d_flow_img, a.rows, a.cols = int(j["d_flow_img"]), int(j["rows"]), int(j["cols"])
            a.fx, a.fy, a.cx, a.cy, a.gamma = float(K[0]), float(K[1]), float(K[2]), float(K[3]), float(j["gamma"])
            a.d_depth_map, a.d_R, a.d_t = int(j["d_depth_map"]), int(j.get("d_R") or 0) or None, int(j.get("d_t") or 0) or None
        prm = FrameParams(int(trials), int(use_acceleration_mode), int(use_refinement), int(depth_mode), int(k_sign_mode),
                          int(flow_index_mode), int(use_global_shutter_mode), 0, float(tol), float(flow_threshold), 0)
        res = (FrameResult * n)()
        fn, ctx, check = self.lib.rsdsfm_solve_frames_dev, self._ctx, self._check

        def call(seeds):
            for a, sd in zip(arr, seeds):
                a.seed = int(sd)
            rc = fn(ctx, arr, C.c_int32(n), C.byref(prm), res)
            if rc != OK:
                check(rc, "rsdsfm_solve_frames_dev")
            return res

        return call

    def solve_frames_dev(self, jobs, seeds, **kw):
        """rsdsfm_solve_frames_dev once; returns one dict per pair (as solve_frame_dev)"""
        res = self.prepared_frames_solve(jobs, **kw)(seeds)
        return [dict(n=int(r.n_points), num_inliers=int(r.num_inliers), best_trial=int(r.best_trial), flipped=bool(r.flipped),
                     ransac_w=np.array(r.ransac_w[:]), ransac_v=np.array(r.ransac_v[:]), ransac_k=float(r.ransac_k),
                     w=np.array(r.w[:]), v=np.array(r.v[:]), k=float(r.k), refine_summary=r.refine_summary.as_dict(),
                     d_inliers=r.d_inliers, d_inlier_idx=r.d_inlier_idx, d_scanline=r.d_scanline) for r in res]

    def depth_lm_reduce_dev(self, n_shard, d_row):
        self._check(self.lib.rsdsfm_depth_lm_reduce_dev(self._ctx, C.c_int64(n_shard), _dp(d_row)), "rsdsfm_depth_lm_reduce_dev")

    def depth_lm_decide_rows_dev(self, d_rows, nrows, n_total, launch_id):
        self._check(self.lib.rsdsfm_depth_lm_decide_rows_dev(self._ctx, _dp(d_rows), C.c_int32(nrows), C.c_int64(n_total), int(launch_id)), "rsdsfm_depth_lm_decide_rows_dev")

    def depth_lm_state(self):
        status, nxt, sm = C.c_int32(), C.c_int32(), LmSummary()
        self._check(self.lib.rsdsfm_depth_lm_state(self._ctx, C.byref(status), C.byref(nxt), C.byref(sm)), "rsdsfm_depth_lm_state")
        return status.value, nxt.value, sm.as_dict()

    def depth_finish_dev(self, d_q, d_u, n, v, w, k, d_alpha, d_alpha_k, d_rho):
        sm = LmSummary()
        extra = C.c_int32()
        self._check(self.lib.rsdsfm_depth_finish_dev(self._ctx, _dp(d_q), _dp(d_u), C.c_int64(n), _v3(v), _v3(w), C.c_double(k), _dp(d_alpha), _dp(d_alpha_k), _dp(d_rho), C.byref(sm), C.byref(extra)), "rsdsfm_depth_finish_dev")
        return sm.as_dict(), extra.value


# ---------------------------------------------------------------------------------------------------
# stage-level wrappers of the row-tiled whole-frame solve (include/rsdsfm.h "ROW-TILED WHOLE-FRAME"; driver: dist.py)
# ---------------------------------------------------------------------------------------------------
def _np0(ptr):
    return _dp(ptr) if ptr else None


def tile_sizes():
    """(bytes per LM state, bytes of the winner record, doubles per LM row, hypotheses per rows call)"""
    lib = load_library()
    lib.rsdsfm_tile_lm_state_bytes.restype = C.c_size_t
    lib.rsdsfm_tile_best_bytes.restype = C.c_size_t
    return (int(lib.rsdsfm_tile_lm_state_bytes()), int(lib.rsdsfm_tile_best_bytes()), int(lib.rsdsfm_tile_ransac_row_size()),
            int(lib.rsdsfm_tile_ransac_batch()))


def sample_indices(n, iterations, seed):
    """the deterministic sampler of rsdsfm_ransac (host side): [iterations, 9] int32 global indices"""
    lib = load_library()
    out = np.zeros((max(int(iterations), 0), 9), dtype=np.int32)
    rc = lib.rsdsfm_sample_indices(C.c_int64(n), C.c_int32(iterations), C.c_uint64(seed), _p(out))
    if rc != OK:
        raise RsdsfmError("rsdsfm_sample_indices failed (%d): needs 9 <= n < 2^31" % rc)
    return out


class _TileMixin:
    def flatten_slab_dev(self, d_img_slab, rows, slab_cols, col0, K, gamma, d_q, d_u, d_alpha, d_alpha_k, thr=1e-10):
        cnt = C.c_int64()
        d = C.c_double
        self._check(self.lib.rsdsfm_flatten_slab_dev(self._ctx, _np0(d_img_slab), C.c_int32(rows), C.c_int32(slab_cols), C.c_int32(col0), d(K[0]), d(K[1]), d(K[2]), d(K[3]), d(gamma), d(thr), _np0(d_q), _np0(d_u), _np0(d_alpha), _np0(d_alpha_k), C.byref(cnt)), "rsdsfm_flatten_slab_dev")
        return cnt.value

    def minimal9_probe_dev(self, d_q9, d_u9, d_a9, d_ak9, count, use_alpha_k, k_sign_mode, use_cores, d_hyp, d_probe4):
        """the wave-per-hypothesis minimal solver with per-hypothesis {SVD sweeps, rotations, SVD clocks, total clocks} (rsdsfm_minimal9_probe_dev)"""
        self._check(self.lib.rsdsfm_minimal9_probe_dev(self._ctx, _dp(d_q9), _dp(d_u9), _dp(d_a9), _dp(d_ak9), C.c_int32(count), int(use_alpha_k), int(k_sign_mode),
                                                       int(use_cores), _dp(d_hyp), _dp(d_probe4)), "rsdsfm_minimal9_probe_dev")

    def minimal9_dev(self, d_q9, d_u9, d_a9, d_ak9, count, use_alpha_k, k_sign_mode, d_hyp):
        self._check(self.lib.rsdsfm_minimal9_dev(self._ctx, _np0(d_q9), _np0(d_u9), _np0(d_a9), _np0(d_ak9), C.c_int32(count), int(use_alpha_k), int(k_sign_mode), _np0(d_hyp)), "rsdsfm_minimal9_dev")

    def tile_ransac_lm_rows_dev(self, d_q, d_u, d_a, d_ak, n, d_hyp, count, d_states, rnd, tol, d_rows):
        self._check(self.lib.rsdsfm_tile_ransac_lm_rows_dev(self._ctx, _np0(d_q), _np0(d_u), _np0(d_a), _np0(d_ak), C.c_int64(n), _dp(d_hyp), C.c_int32(count), _dp(d_states), C.c_int32(rnd), C.c_double(tol), _dp(d_rows)), "rsdsfm_tile_ransac_lm_rows_dev")

    def tile_ransac_decide_dev(self, d_rows_all, nranks, count, d_states, n_total, rnd, d_flags, d_scored, d_tcount, d_terr):
        self._check(self.lib.rsdsfm_tile_ransac_decide_dev(self._ctx, _dp(d_rows_all), C.c_int32(nranks), C.c_int32(count), _dp(d_states), C.c_int64(n_total), C.c_int32(rnd), _dp(d_flags), _dp(d_scored), _dp(d_tcount), _dp(d_terr)), "rsdsfm_tile_ransac_decide_dev")

    def tile_ransac_score_rows_dev(self, d_q, d_u, d_a, d_ak, n, d_hyp, count, d_states, depth_mode, tol, d_scored, d_rows):
        self._check(self.lib.rsdsfm_tile_ransac_score_rows_dev(self._ctx, _np0(d_q), _np0(d_u), _np0(d_a), _np0(d_ak), C.c_int64(n), _dp(d_hyp), C.c_int32(count), _dp(d_states), int(depth_mode), C.c_double(tol), _np0(d_scored), _dp(d_rows)), "rsdsfm_tile_ransac_score_rows_dev")

    def tile_ransac_score_merge_dev(self, d_rows_all, nranks, count, d_scored, d_tcount, d_terr):
        self._check(self.lib.rsdsfm_tile_ransac_score_merge_dev(self._ctx, _dp(d_rows_all), C.c_int32(nranks), C.c_int32(count), _np0(d_scored), _dp(d_tcount), _dp(d_terr)), "rsdsfm_tile_ransac_score_merge_dev")

    def tile_ransac_pick_dev(self, d_tcount, d_terr, iterations, d_hyp, d_best):
        self._check(self.lib.rsdsfm_tile_ransac_pick_dev(self._ctx, _np0(d_tcount), _np0(d_terr), C.c_int32(iterations), _np0(d_hyp), _dp(d_best)), "rsdsfm_tile_ransac_pick_dev")

    def tile_ransac_final_dev(self, d_q, d_u, d_a, d_ak, n, d_best, d_states, depth_mode, tol, d_rho, d_mask, d_idx, d_inl, d_oa, d_oak):
        out = RansacOut()
        self._check(self.lib.rsdsfm_tile_ransac_final_dev(self._ctx, _np0(d_q), _np0(d_u), _np0(d_a), _np0(d_ak), C.c_int64(n), _dp(d_best), _dp(d_states), int(depth_mode), C.c_double(tol), _np0(d_rho), _np0(d_mask), _np0(d_idx), _np0(d_inl), _np0(d_oa), _np0(d_oak), C.byref(out)), "rsdsfm_tile_ransac_final_dev")
        return dict(shard_inliers=int(out.num_inliers), best_trial=int(out.best_trial), w=np.array(out.w[:]), v=np.array(out.v[:]), k=float(out.k), inlier_error=float(out.inlier_error))

    def tile_ransac_global_inliers(self, d_best):
        self.lib.rsdsfm_tile_ransac_global_inliers.restype = C.c_int64
        r = int(self.lib.rsdsfm_tile_ransac_global_inliers(self._ctx, _dp(d_best)))
        if r < 0:
            raise RsdsfmError("rsdsfm_tile_ransac_global_inliers failed")
        return r

    def tile_refine_begin_dev(self, d_flow, n_flow, m, d_inl, d_alpha, d_alpha_k, d_idx, v, w, k, const_acceleration, flow_index_mode=FLOW_GATHERED):
        self._check(self.lib.rsdsfm_tile_refine_begin_dev(self._ctx, _np0(d_flow), C.c_int64(n_flow), C.c_int64(m), _np0(d_inl), _np0(d_alpha), _np0(d_alpha_k), _np0(d_idx), _v3(v), _v3(w), C.c_double(k), int(const_acceleration), int(flow_index_mode)), "rsdsfm_tile_refine_begin_dev")

    def tile_refine_row_size(self, const_acceleration, stage):
        return int(self.lib.rsdsfm_tile_refine_row_size(int(const_acceleration), C.c_int32(stage)))

    def tile_refine_rows_dev(self, stage, d_row):
        self._check(self.lib.rsdsfm_tile_refine_rows_dev(self._ctx, C.c_int32(stage), _dp(d_row)), "rsdsfm_tile_refine_rows_dev")

    def tile_refine_apply_dev(self, stage, d_rows_all, nranks, m_total):
        self._check(self.lib.rsdsfm_tile_refine_apply_dev(self._ctx, C.c_int32(stage), _dp(d_rows_all), C.c_int32(nranks), C.c_int64(m_total)), "rsdsfm_tile_refine_apply_dev")

    def tile_refine_poll(self):
        vo, wo, ko = (C.c_double * 3)(), (C.c_double * 3)(), C.c_double()
        sm = LmSummary()
        self._check(self.lib.rsdsfm_tile_refine_poll(self._ctx, vo, wo, C.byref(ko), C.byref(sm)), "rsdsfm_tile_refine_poll")
        return dict(v=np.array(vo[:]), w=np.array(wo[:]), k=ko.value, summary=sm.as_dict(), running=sm.termination < 0)

    def tile_refine_finish_dev(self, d_inl_out):
        self._check(self.lib.rsdsfm_tile_refine_finish_dev(self._ctx, _np0(d_inl_out)), "rsdsfm_tile_refine_finish_dev")

    def tile_zsum_dev(self, d_inl, m, d_zsum):
        self._check(self.lib.rsdsfm_tile_zsum_dev(self._ctx, _np0(d_inl), C.c_int64(m), _dp(d_zsum)), "rsdsfm_tile_zsum_dev")

    def tile_depth_map_dev(self, d_inl, m, d_zsums_all, nranks, m_total, v, K, rows, col0, slab_cols, d_depth_slab, d_xs=None, d_ys=None):
        vv = _v3(v)
        flipped = C.c_int()
        d = C.c_double
        self._check(self.lib.rsdsfm_tile_depth_map_dev(self._ctx, _np0(d_inl), C.c_int64(m), _dp(d_zsums_all), C.c_int32(nranks), C.c_int64(m_total), vv, d(K[0]), d(K[1]), d(K[2]), d(K[3]), C.c_int32(rows), C.c_int32(col0), C.c_int32(slab_cols), _np0(d_depth_slab), _np0(d_xs), _np0(d_ys), C.byref(flipped)), "rsdsfm_tile_depth_map_dev")
        return np.array(vv[:]), bool(flipped.value)


for _name, _fn in list(vars(_TileMixin).items()):
    if not _name.startswith("__"):
        setattr(Solver, _name, _fn)


# ---------------------------------------------------------------------------------------------------
# consumers of the solve's output: back projection, crack interpolation, 8-bit depth image
# ---------------------------------------------------------------------------------------------------
BACKPROJECT_RS, BACKPROJECT_GS = 0, 1
Q5_COMPAT, Q5_FIXED = 0, 1


class _RectifyMixin:
    def back_project(self, image_bgr, depth_map, R, t, K, mode=BACKPROJECT_RS, q5_mode=Q5_COMPAT, want_coords=True):
        """RsFrame::backProject / backProjectGs.  depth_map: (rows, cols) array; R: (rows, 3, 3) or (rows, 9); t: (rows, 3)."""
        img = np.ascontiguousarray(image_bgr, dtype=np.uint8)
        rows, cols = img.shape[:2]
        dm = np.ascontiguousarray(np.asarray(depth_map, dtype=np.float64).T)  # column-major rows x cols
        Rr, tt = _f64(np.asarray(R).reshape(rows, 9)), _f64(t)
        gs = np.zeros_like(img)
        c3 = np.zeros((rows, cols, 3), dtype=np.float32) if want_coords else None
        d = C.c_double
        self._check(self.lib.rsdsfm_back_project(self._ctx, _p(img), _p(dm), _p(Rr), _p(tt), d(K[0]), d(K[1]), d(K[2]), d(K[3]), C.c_int32(rows), C.c_int32(cols), int(mode), int(q5_mode), _p(gs), _p(c3)), "rsdsfm_back_project")
        return gs, c3

    def back_project_dev(self, d_img, d_depth_map, d_R, d_t, K, rows, cols, d_gs, d_coords=None, mode=BACKPROJECT_RS, q5_mode=Q5_COMPAT):
        d = C.c_double
        self._check(self.lib.rsdsfm_back_project_dev(self._ctx, _dp(d_img), _dp(d_depth_map), _dp(d_R), _dp(d_t), d(K[0]), d(K[1]), d(K[2]), d(K[3]), C.c_int32(rows), C.c_int32(cols), int(mode), int(q5_mode), _dp(d_gs), _dp(d_coords) if d_coords else None), "rsdsfm_back_project_dev")

    def interpolate_cracky(self, image_bgr, offset=1):
        img = np.ascontiguousarray(image_bgr, dtype=np.uint8)
        rows, cols = img.shape[:2]
        out = np.zeros_like(img)
        self._check(self.lib.rsdsfm_interpolate_cracky(self._ctx, _p(img), C.c_int32(rows), C.c_int32(cols), C.c_int32(offset), _p(out)), "rsdsfm_interpolate_cracky")
        return out

    def interpolate_cracky_dev(self, d_in, rows, cols, d_out, offset=1):
        self._check(self.lib.rsdsfm_interpolate_cracky_dev(self._ctx, _dp(d_in), C.c_int32(rows), C.c_int32(cols), C.c_int32(offset), _dp(d_out)), "rsdsfm_interpolate_cracky_dev")

    def depth_preview(self, inliers, K, rows, cols):
        inl = _f64(inliers).reshape(-1, 3)
        out = np.zeros((rows, cols), dtype=np.uint8)
        d = C.c_double
        self._check(self.lib.rsdsfm_depth_preview(self._ctx, _p(inl), C.c_int64(inl.shape[0]), d(K[0]), d(K[1]), d(K[2]), d(K[3]), C.c_int32(rows), C.c_int32(cols), _p(out)), "rsdsfm_depth_preview")
        return out

    def depth_preview_dev(self, d_inl, m, K, rows, cols, d_out):
        d = C.c_double
        self._check(self.lib.rsdsfm_depth_preview_dev(self._ctx, _np0(d_inl), C.c_int64(m), d(K[0]), d(K[1]), d(K[2]), d(K[3]), C.c_int32(rows), C.c_int32(cols), _dp(d_out)), "rsdsfm_depth_preview_dev")

    def rectify_frame_dev(self, d_inl, m, d_img, d_depth_map, d_R, d_t, K, rows, cols, d_preview, d_gs, d_fixed, d_coords=None, mode=BACKPROJECT_RS,
                          q5_mode=Q5_COMPAT, offset=1):
        """main.cc:480-523 in one call (rsdsfm_rectify_frame_dev): depth image + back projection + crack interpolation, two launches (three when
        the interpolation offset exceeds 2 or cols is not a multiple of 4)"""
        d = C.c_double
        self._check(self.lib.rsdsfm_rectify_frame_dev(self._ctx, _np0(d_inl), C.c_int64(m), _dp(d_img), _dp(d_depth_map), _dp(d_R), _dp(d_t), d(K[0]), d(K[1]),
                                                      d(K[2]), d(K[3]), C.c_int32(rows), C.c_int32(cols), int(mode), int(q5_mode), C.c_int32(offset),
                                                      _dp(d_preview), _dp(d_gs), _np0(d_coords), _dp(d_fixed)), "rsdsfm_rectify_frame_dev")


for _name, _fn in list(vars(_RectifyMixin).items()):
    if not _name.startswith("__"):
        setattr(Solver, _name, _fn)


# ---------------------------------------------------------------------------------------------------
# ground-truth flow between two rolling-shutter frames
# ---------------------------------------------------------------------------------------------------
class _TrueFlowMixin:
    def set_true_flow_search(self, mode):
        """0 / False (default): interval-pruned exact search from 96 scanlines on; 1 / True: every scanline for every pixel; 2: pruned
        at any size (rsdsfm_set_true_flow_search).  Identical results."""
        self._check(self.lib.rsdsfm_set_true_flow_search(self._ctx, int(mode)), "rsdsfm_set_true_flow_search")

    def true_flow(self, world_xyz, R2, t2, K, q5_mode=Q5_COMPAT, want_best_row=True):
        """Camera::calculateTrueFlow.  world_xyz: (rows, cols, 3) world point per pixel of frame 1 (zeros = void);
        R2: (rows2, 3, 3) or (rows2, 9); t2: (rows2, 3).  Returns flow (rows, cols, 2) and the winning scanlines."""
        w = np.asarray(world_xyz, dtype=np.float64)
        rows, cols = w.shape[:2]
        maps = [np.ascontiguousarray(w[:, :, c].T) for c in range(3)]  # column-major rows x cols (Eigen MatrixXd)
        tt = _f64(t2)
        rows2 = tt.shape[0]
        Rr = _f64(np.asarray(R2).reshape(rows2, 9))
        flow = np.zeros((rows, cols, 2))
        best = np.zeros((rows, cols), dtype=np.int32) if want_best_row else None
        d = C.c_double
        self._check(self.lib.rsdsfm_true_flow(self._ctx, _p(maps[0]), _p(maps[1]), _p(maps[2]), C.c_int32(rows), C.c_int32(cols), _p(Rr), _p(tt), C.c_int32(rows2), d(K[0]), d(K[1]), d(K[2]), d(K[3]), int(q5_mode), _p(flow), _p(best)), "rsdsfm_true_flow")
        return flow, best

    def true_flow_dev(self, d_wx, d_wy, d_wz, rows, cols, d_R2, d_t2, rows2, K, d_flow, d_best_row=None, q5_mode=Q5_COMPAT):
        d = C.c_double
        self._check(self.lib.rsdsfm_true_flow_dev(self._ctx, _dp(d_wx), _dp(d_wy), _dp(d_wz), C.c_int32(rows), C.c_int32(cols), _dp(d_R2), _dp(d_t2), C.c_int32(rows2), d(K[0]), d(K[1]), d(K[2]), d(K[3]), int(q5_mode), _dp(d_flow), _dp(d_best_row) if d_best_row else None), "rsdsfm_true_flow_dev")


for _name, _fn in list(vars(_TrueFlowMixin).items()):
    if not _name.startswith("__"):
        setattr(Solver, _name, _fn)


# ---------------------------------------------------------------------------------------------------
# accuracy metrics
# ---------------------------------------------------------------------------------------------------
class ReprojectionStats(C.Structure):
    _fields_ = [("scale", C.c_double), ("mean_error", C.c_double), ("sum_error", C.c_double), ("number_outliers", C.c_int64),
                ("scale_inliers", C.c_int64), ("error_inliers", C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def velocity_errors(w_est, v_est, w_true, v_true):
    """(rotation error, translation angle) of errorMeasure.cpp:178-186 -- host-only scalar math through the C ABI"""
    lib = load_library()
    we, ve = C.c_double(), C.c_double()
    rc = lib.rsdsfm_velocity_errors(_v3(w_est), _v3(v_est), _v3(w_true), _v3(v_true), C.byref(we), C.byref(ve))
    if rc != OK:
        raise RsdsfmError("rsdsfm_velocity_errors failed (%d)" % rc)
    return we.value, ve.value


class _MetricsMixin:
    def reprojection_error(self, est_coords, gt_depth, est_depth, R_abs, t_abs, K, max_norm=10.0, want_image=True):
        """Camera::meanReprojectionError (+ createErrorImage).  est_coords: (rows, cols, 3) float32; depth maps (rows, cols)."""
        est = np.ascontiguousarray(est_coords, dtype=np.float32)
        rows, cols = est.shape[:2]
        gd = np.ascontiguousarray(np.asarray(gt_depth, dtype=np.float64).T)
        ed = np.ascontiguousarray(np.asarray(est_depth, dtype=np.float64).T)
        Rr, tt = _f64(np.asarray(R_abs).reshape(rows, 9)), _f64(t_abs)
        st = ReprojectionStats()
        img = np.zeros((rows, cols), dtype=np.uint8) if want_image else None
        d = C.c_double
        self._check(self.lib.rsdsfm_reprojection_error(self._ctx, _p(est), _p(gd), _p(ed), _p(Rr), _p(tt), d(K[0]), d(K[1]), d(K[2]), d(K[3]), C.c_int32(rows), C.c_int32(cols), d(max_norm), C.byref(st), _p(img)), "rsdsfm_reprojection_error")
        return st.as_dict(), img

    def reprojection_error_dev(self, d_est, d_gt_depth, d_est_depth, d_R, d_t, K, rows, cols, max_norm=10.0, d_error_image=None):
        st = ReprojectionStats()
        d = C.c_double
        self._check(self.lib.rsdsfm_reprojection_error_dev(self._ctx, _dp(d_est), _dp(d_gt_depth), _dp(d_est_depth), _dp(d_R), _dp(d_t), d(K[0]), d(K[1]), d(K[2]), d(K[3]), C.c_int32(rows), C.c_int32(cols), d(max_norm), C.byref(st), _dp(d_error_image) if d_error_image else None), "rsdsfm_reprojection_error_dev")
        return st.as_dict()


for _name, _fn in list(vars(_MetricsMixin).items()):
    if not _name.startswith("__"):
        setattr(Solver, _name, _fn)

from . import evaluate, formats  # noqa: E402,F401  (on-disk formats + archive runner)


# ---------------------------------------------------------------------------------------------------
# batched fast path of the dense depth solve: several independent solves per launch (one context each, one shared stream)
# ---------------------------------------------------------------------------------------------------
def prepared_depth_batch(solvers, problems, launch0_only=False):
    """solvers: list of Solver (<= 8, all created on the SAME stream); problems: list of dicts with device pointers
    d_q, d_u, d_alpha, d_alpha_k, d_rho and n, v, w, k.  Returns a zero-argument callable that enqueues the whole batch with
    pre-marshalled arguments (rsdsfm_estimate_inverse_depths_batch_dev; launch0_only: only the streaming launch, for profiling);
    finish each solve with solvers[i].depth_finish_dev."""
    lib = solvers[0].lib
    cnt = len(solvers)
    assert cnt == len(problems) and 1 <= cnt <= 8
    VP = C.c_void_p * cnt
    ctxs = VP(*[s._ctx for s in solvers])
    ptrs = {k2: VP(*[int(p[k2]) for p in problems]) for k2 in ("d_q", "d_u", "d_alpha", "d_alpha_k", "d_rho")}
    ns = (C.c_int64 * cnt)(*[int(p["n"]) for p in problems])
    v3 = (C.c_double * (3 * cnt))(*[float(x) for p in problems for x in p["v"]])
    w3 = (C.c_double * (3 * cnt))(*[float(x) for p in problems for x in p["w"]])
    ks = (C.c_double * cnt)(*[float(p["k"]) for p in problems])
    fn = lib.rsdsfm_depth_lm_batch_launch_dev if launch0_only else lib.rsdsfm_estimate_inverse_depths_batch_dev
    args = (ctxs, C.c_int32(cnt), ptrs["d_q"], ptrs["d_u"], ns, v3, w3, ks, ptrs["d_alpha"], ptrs["d_alpha_k"], ptrs["d_rho"])
    s0 = solvers[0]

    def call():
        rc = fn(*args)
        if rc != OK:
            s0._check(rc, "rsdsfm_estimate_inverse_depths_batch_dev")

    return call
