"""Same-box A/B of two builds of the library on the whole frame solve (1280x720, 50 trials, DeepFlow-like pair): the two packages are
imported side by side, blocks of solves alternate between them, and the median and the mean ms per solve of every block are printed
(the mean carries the solves whose speculation did not hold: one DeepFlow-like pair in ten).
usage (GPU box): python tools/ab_lib.py <package dir A> <package dir B> [blocks] [solves per block]
(package dir = a directory holding __init__.py and a built librsdsfm_hip.so, e.g. rs-aware-differential-sfm_amd and a copy of an older
checkout built with its own build.py under ab_old/); AB_ACCEL=1 in the environment: acceleration mode (k estimated and refined)"""
import importlib.util
import os
import statistics
import sys
import time

import torch


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(path, "__init__.py"), submodule_search_locations=[path])
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def main():
    a, b = load(sys.argv[1], "pkg_a"), load(sys.argv[2], "pkg_b")
    blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    per = int(sys.argv[4]) if len(sys.argv) > 4 else 100
    d = a.synth.make_config(5)
    rows, cols = d["rows"], d["cols"]
    img = torch.from_numpy(d["flow_img"]).cuda()
    dm = torch.empty((cols, rows), dtype=torch.float64, device="cuda")
    R = torch.empty((rows, 9), dtype=torch.float64, device="cuda")
    t = torch.empty((rows, 3), dtype=torch.float64, device="cuda")
    with a.Solver(0) as sa, b.Solver(0) as sb:
        calls = [s.prepared_frame_solve(img.data_ptr(), rows, cols, d["K"], d["gamma"], dm.data_ptr(), R.data_ptr(), t.data_ptr(), trials=50, tol=0.05, use_acceleration_mode=bool(int(os.environ.get("AB_ACCEL", "0")))) for s in (sa, sb)]
        for c in calls:
            for i in range(20):
                c(1 + i)
        med = ([], [])
        mean = ([], [])
        for blk in range(blocks):
            for w, c in enumerate(calls):
                ts = []
                for i in range(per):
                    t0 = time.perf_counter()
                    c(1 + i)
                    ts.append((time.perf_counter() - t0) * 1e3)
                med[w].append(statistics.median(ts))
                mean[w].append(statistics.fmean(ts))
        for w, nm in enumerate(("A", "B")):
            print(nm, sys.argv[1 + w], "medians", " ".join("%.4f" % x for x in med[w]), "| median of medians %.4f ms" % statistics.median(med[w]))
            print(nm, sys.argv[1 + w], "means  ", " ".join("%.4f" % x for x in mean[w]), "| mean of means %.4f ms" % statistics.fmean(mean[w]))


if __name__ == "__main__":
    main()
