"""Ties profiles/counters.json to the kernel sources it was collected from.

The roofline record of bench.py multiplies a LIVE launch duration by instruction / byte counts that a rocprofv3 --pmc pass stored in
profiles/counters.json; those counts are only valid for the kernel as it was compiled then.  pmc_to_json.py stamps the file with the
sha256 of every translation unit and header under csrc/ (and of build.py: the compiler flags); bench.py compares the hashes of the
files a kernel is built from with the current sources and reports `counters_stale: true` (and no `frac`) when they differ."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "rs-aware-differential-sfm_amd")
COMMON = ["device_math.hpp", "lm_common.hpp", "lma_common.hpp", "lma_stages.hpp", "rsdsfm_internal.hpp", "build.py", "rsdsfm.h"]
# kernel name prefix -> the translation unit that defines it (every kernel also depends on COMMON)
UNITS = {"refine_rf": "refine_rf_kernels.hip", "ransac_lma": "ransac_lma_kernels.hip", "depth_lma": "depth_lma_kernels.hip", "ransac_": "ransac_kernels.hip", "depth_lm": "depth_kernels.hip", "depth_closed": "depth_kernels.hip", "minimal9": "minimal9_kernels.hip",
         "refine_": "refine_kernels.hip", "back_project": "rectify_kernels.hip", "rectify_": "rectify_kernels.hip", "interpolate_": "rectify_kernels.hip", "preview_": "rectify_kernels.hip",
         "true_flow": "gtflow_kernels.hip", "pose_bounds": "gtflow_kernels.hip", "reproj_": "metrics_kernels.hip", "flatten_": "glue_kernels.hip",
         "cell_scan": "glue_kernels.hip", "depth_claim": "glue_kernels.hip", "depth_write": "glue_kernels.hip", "zsum_": "glue_kernels.hip",
         "pose_table": "glue_kernels.hip", "alpha_": "glue_kernels.hip"}


def source_hashes():
    files = sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")) + glob.glob(os.path.join(PKG, "csrc", "*.hpp")))
    files += [os.path.join(PKG, "build.py"), os.path.join(ROOT, "include", "rsdsfm.h")]
    return {os.path.basename(f): hashlib.sha256(open(f, "rb").read()).hexdigest()[:16] for f in files}


def files_of(kernel):
    unit = next((u for pre, u in UNITS.items() if kernel.startswith(pre)), None)
    extra = ["refine_common.hpp"] if kernel.startswith("refine_") else []
    return ([unit] if unit else []) + extra + COMMON


def stale_files(kernel, stamped):
    """files the kernel is built from whose hash differs from the stamp (all of them when there is no stamp)"""
    now = source_hashes()
    stamped = stamped or {}
    return [f for f in files_of(kernel) if stamped.get(f) != now.get(f)]
