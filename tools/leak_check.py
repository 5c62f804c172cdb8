import sys
sys.path.insert(0, '.')
import numpy as np, torch, rsdsfm
dev = torch.device("cuda", 0)
d = rsdsfm.synth.make_config(5, seed=0x5EED0005)
rows, cols = d["rows"], d["cols"]
img = torch.from_numpy(d["flow_img"]).to(dev)
dms = [torch.empty((cols, rows), dtype=torch.float64, device=dev) for _ in range(6)]
torch.cuda.synchronize()
base = torch.cuda.mem_get_info()[0]
for rep in range(6):
    with rsdsfm.Solver(0) as s:
        jobs = [dict(d_flow_img=img.data_ptr(), rows=rows, cols=cols, K=d["K"], gamma=d["gamma"], d_depth_map=dms[i].data_ptr()) for i in range(6)]
        s.solve_frames_dev(jobs, [1, 2, 3, 4, 5, 6], trials=50, tol=0.05)
        s.synchronize()
        inside = torch.cuda.mem_get_info()[0]
    after = torch.cuda.mem_get_info()[0]
    print("rep %d: in use with the context %.1f MB, after closing it %.1f MB" % (rep, (base - inside) / 1e6, (base - after) / 1e6), flush=True)
