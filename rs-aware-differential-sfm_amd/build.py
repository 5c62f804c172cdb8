"""Builds librsdsfm_hip.so (all HIP kernels + the C ABI) for gfx950, in-tree."""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "librsdsfm_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")))


def needs_build():
    if not os.path.exists(SO):
        return True
    deps = sources() + glob.glob(os.path.join(HERE, "csrc", "*.hpp")) + [os.path.join(HERE, "..", "include", "rsdsfm.h")]
    return os.path.getmtime(SO) < max(os.path.getmtime(d) for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + sources() + ["-o", SO]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force=True, verbose=True))
