"""Builds the HIP libraries (all kernels + the C ABI) for gfx950, in-tree.

    librsdsfm_hip.so        the product.  Default path: analytic LM trajectory (depth solves) + radius-factorised refinement, the library's own
                            arithmetic with fused multiply-adds, guarded to the integers of the reference's arithmetic; the iterate-by-iterate
                            kernels behind rsdsfm_set_lm_arithmetic(1) are the REFERENCE's arithmetic (no fused multiply-add; csrc/device_math.hpp)
    librsdsfm_hip_fused.so  opt-in: the same sources with -DRSDSFM_FUSED=1 (explicit fmas also in the iterate-by-iterate per-pixel model)

Every source is compiled to its own object (in parallel) and the objects are linked; only the translation units whose
arithmetic depends on RSDSFM_FUSED are compiled a second time for the fused library.  Optional RCCL support of the
multi-GPU driver (csrc/dist_host.hip) resolves librccl at run time (dlopen), so the libraries load without it.
"""
import concurrent.futures
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "librsdsfm_hip.so")
SO_FUSED = os.path.join(HERE, "librsdsfm_hip_fused.so")
OBJ_DIR = os.path.join(HERE, "build")
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall", "-Wno-unused-function"]
LDFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-ldl"]
# translation units that include the per-pixel model (device_math.hpp / lm_common.hpp) or carry their own RSDSFM_FUSED switch
FUSED_UNITS = ("depth_kernels", "ransac_kernels", "ransac_lma_kernels", "depth_lma_kernels", "gtflow_kernels", "capi")


def sources():
    return sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")))


def _deps():
    return glob.glob(os.path.join(HERE, "csrc", "*.hpp")) + [os.path.join(HERE, "..", "include", "rsdsfm.h")]


def _stale(target, deps):
    return not os.path.exists(target) or os.path.getmtime(target) < max(os.path.getmtime(d) for d in deps)


def needs_build():
    return _stale(SO, sources() + _deps()) or _stale(SO_FUSED, sources() + _deps())


def _compile(job):
    src, obj, extra, verbose = job
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + CFLAGS + extra + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return obj


def build(force=False, verbose=False, jobs=None):
    """compiles what is stale and links both libraries; returns the path of the reference-arithmetic library"""
    if not force and not needs_build():
        return SO
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdrs = _deps()
    todo, objs, objs_fused = [], [], []
    for src in sources():
        stem = os.path.splitext(os.path.basename(src))[0]
        obj = os.path.join(OBJ_DIR, stem + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            todo.append((src, obj, [], verbose))
        if stem in FUSED_UNITS:
            objf = os.path.join(OBJ_DIR, stem + ".fused.o")
            objs_fused.append(objf)
            if force or _stale(objf, [src] + hdrs):
                todo.append((src, objf, ["-DRSDSFM_FUSED=1"], verbose))
        else:
            objs_fused.append(obj)
    jobs = jobs or max(1, min(len(todo), (os.cpu_count() or 4)))
    if todo:
        with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as ex:
            list(ex.map(_compile, todo))
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    for so, oo in ((SO, objs), (SO_FUSED, objs_fused)):
        tmp = so + ".tmp%d" % os.getpid()
        cmd = [hipcc] + LDFLAGS + oo + ["-o", tmp]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        os.replace(tmp, so)
    return SO


if __name__ == "__main__":
    import sys

    print(build(force="--stale" not in sys.argv, verbose="--stale" not in sys.argv))
