"""diagnostic: the bench's sequence leg alone, with a stack dump after 40 s;  python tools/seq_hang_probe.py [refine_arithmetic] [pairs]"""
import faulthandler, os, sys, time
faulthandler.dump_traceback_later(40, exit=True)
sys.path.insert(0, ".")
import numpy as np, torch
import rsdsfm
ra = int(sys.argv[1]) if len(sys.argv) > 1 else 0
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
frames, meta = rsdsfm.synth.make_flow_sequence(5, [0x5EED0005 + 1000 * i for i in range(pairs)])
rows, cols = meta["rows"], meta["cols"]
imgs = [torch.from_numpy(f).to(dev) for f in frames]
dms = [torch.empty((cols, rows), dtype=torch.float64, device=dev) for _ in range(pairs)]
jobs = [dict(d_flow_img=im.data_ptr(), rows=rows, cols=cols, K=meta["K"], gamma=meta["gamma"], d_depth_map=dm.data_ptr(), d_R=None, d_t=None) for im, dm in zip(imgs, dms)]
with rsdsfm.Solver(0) as s:
    s.set_refine_arithmetic(ra)
    if os.environ.get("PROBE_STAGE"):
        s.set_refine_stage(int(os.environ["PROBE_STAGE"]))
    if os.environ.get("PROBE_LANES"):
        s.lib.rsdsfm_set_sequence_lanes(s._ctx, int(os.environ["PROBE_LANES"]))
    if len(sys.argv) > 3:  # the host-pointer boundary first (bench.py's order)
        d = rsdsfm.synth.make_config(5, seed=0x5EED0005)
        r = s.ransac(d["q"], d["u"], d["alpha"], d["alpha_k"], False, 50, 0.05, seed=11)
        o = s.non_linear_refinement(d["u"], r["inliers"], r["alpha"], r["alpha_k"], r["v"], r["w"], r["k"], False, tag=r["tag"])
        print("host boundary ok", r["num_inliers"], o["summary"]["num_iterations"], s.refine_cache_hits(), flush=True)
    call = s.prepared_frames_solve(jobs, trials=50, tol=0.05)
    for p in range(4):
        t0 = time.perf_counter()
        res = call([1 + pairs * p + i for i in range(pairs)])
        print("pass", p, "%.2f ms per pair" % ((time.perf_counter() - t0) / pairs * 1e3), [int(r.refine_summary.num_iterations) for r in res][:8], s.refine_restarts(), flush=True)
print("done")
