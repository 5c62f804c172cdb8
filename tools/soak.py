import sys, time
sys.path.insert(0, '.')
import numpy as np, torch, rsdsfm
dev = torch.device("cuda", 0)
d = rsdsfm.synth.make_config(5, seed=0x5EED0005)
rows, cols = d["rows"], d["cols"]
img = torch.from_numpy(d["flow_img"]).to(dev)
dm = torch.empty((cols, rows), dtype=torch.float64, device=dev)
free0 = torch.cuda.mem_get_info()[0]
res = set()
with rsdsfm.Solver(0) as s:
    call = s.prepared_frame_solve(img.data_ptr(), rows, cols, d["K"], d["gamma"], dm.data_ptr(), trials=50, tol=0.05)
    t0 = time.time()
    ts = []
    for i in range(3000):
        t1 = time.perf_counter()
        r = call(1 + i % 7)
        ts.append(time.perf_counter() - t1)
        res.add((i % 7, r.num_inliers, r.best_trial, tuple(r.v[:]), r.refine_summary.num_iterations))
    s.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
print("3000 solves in %.2f s; per-seed results distinct: %d (expect 7); median %.4f ms, p99 %.4f ms, max %.4f ms; device memory delta %.1f MB" % (
    time.time() - t0, len(res), np.median(ts) * 1e3, np.quantile(ts, 0.99) * 1e3, max(ts) * 1e3, (free0 - free1) / 1e6))

# the sequence solve: 200 calls of 16 pairs (3 lanes) on one context -- per-seed results reproducible, no growth of device memory after
# the first call, lanes released with the context
imgs = [img.clone() for _ in range(4)]
dms = [torch.empty((cols, rows), dtype=torch.float64, device=dev) for _ in range(16)]
free0 = torch.cuda.mem_get_info()[0]
res2 = set()
with rsdsfm.Solver(0) as s:
    jobs = [dict(d_flow_img=imgs[i % 4].data_ptr(), rows=rows, cols=cols, K=d["K"], gamma=d["gamma"], d_depth_map=dms[i].data_ptr()) for i in range(16)]
    call = s.prepared_frames_solve(jobs, trials=50, tol=0.05)
    call([1 + i % 7 for i in range(16)])
    s.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    t0 = time.time()
    for rep in range(200):
        rr = call([1 + (i + rep) % 7 for i in range(16)])
        for i, r in enumerate(rr):
            res2.add(((i + rep) % 7, r.num_inliers, r.best_trial, tuple(r.v[:]), r.refine_summary.num_iterations))
    s.synchronize()
    el = time.time() - t0
    free2 = torch.cuda.mem_get_info()[0]
free3 = torch.cuda.mem_get_info()[0]
print("sequence: 3200 solves in %.2f s (%.4f ms per pair); per-seed results distinct: %d (expect 7), identical to the single solves: %s; device memory: "
      "first call %.1f MB, growth over 200 calls %.1f MB, after closing the context %.1f MB" % (
          el, el / 3200 * 1e3, len(res2), res2 == res, (free0 - free1) / 1e6, (free1 - free2) / 1e6, (free0 - free3) / 1e6))
