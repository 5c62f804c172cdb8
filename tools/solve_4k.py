"""The one-call solve (rsdsfm_solve_frame_dev, one context) on BASELINE configs[3]'s 3840x2160 frame: the untiled single-GPU figure next to
bench.py's tiled_full record.  usage (GPU box): python tools/solve_4k.py"""
import importlib, os, sys, time, statistics
import torch
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("rs-aware-differential-sfm_amd")
d = pkg.synth.make_config(4)
rows, cols = d["rows"], d["cols"]
img = torch.from_numpy(d["flow_img"]).cuda()
dm = torch.empty((cols, rows), dtype=torch.float64, device="cuda")
R = torch.empty((rows, 9), dtype=torch.float64, device="cuda"); t = torch.empty((rows, 3), dtype=torch.float64, device="cuda")
with pkg.Solver(0) as s:
    call = s.prepared_frame_solve(img.data_ptr(), rows, cols, d["K"], d["gamma"], dm.data_ptr(), R.data_ptr(), t.data_ptr(), trials=50, tol=0.05)
    ts = []
    for i in range(40):
        t0 = time.perf_counter(); r = call(1 + i); ts.append((time.perf_counter() - t0) * 1e3)
    print("4K one-call solve: median %.3f ms mean %.3f (restarts %d, inliers %d of %d, iterations %d)" % (statistics.median(ts[5:]), statistics.fmean(ts[5:]), s.ransac_restarts(), r.num_inliers, r.n_points, r.refine_summary.num_iterations))
