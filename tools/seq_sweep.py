"""lanes sweep of rsdsfm_solve_frames_dev on the bench's sequence (32 pairs @1280x720, 32 data seeds); prints Mpix/s per lane count"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import rsdsfm

dev = torch.device("cuda", 0)
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
imgs_h, meta = rsdsfm.synth.make_flow_sequence(5, [0x5EED0005 + 1000 * i for i in range(pairs)])
rows, cols = meta["rows"], meta["cols"]
stream = torch.cuda.Stream(dev)
torch.cuda.set_stream(stream)
imgs = [torch.from_numpy(x).to(dev) for x in imgs_h]
dms = [torch.empty((cols, rows), dtype=torch.float64, device=dev) for _ in range(pairs)]
Rs = [torch.empty((rows, 9), dtype=torch.float64, device=dev) for _ in range(pairs)]
ts = [torch.empty((rows, 3), dtype=torch.float64, device=dev) for _ in range(pairs)]
jobs = [dict(d_flow_img=im.data_ptr(), rows=rows, cols=cols, K=meta["K"], gamma=meta["gamma"], d_depth_map=dm.data_ptr(), d_R=R_.data_ptr(), d_t=t_.data_ptr())
        for im, dm, R_, t_ in zip(imgs, dms, Rs, ts)]
for side in (3, 0):
    for lanes in (1, 2, 3, 4, 6):
        with rsdsfm.Solver(0, stream=stream.cuda_stream) as s:
            s.set_sequence_lanes(lanes)
            s.set_frame_side_flatten(side)
            call = s.prepared_frames_solve(jobs, trials=50, tol=0.05)
            call(list(range(1, pairs + 1)))
            torch.cuda.synchronize()
            best, tot = 1e9, 0.0
            for p in range(5):
                t0 = time.perf_counter()
                call([100 * (p + 1) + i for i in range(pairs)])
                el = time.perf_counter() - t0
                best, tot = min(best, el), tot + el
            print("side_flatten %d lanes %2d: %.3f ms/solve mean, %.3f best -> %.0f Mpix/s" % (side, lanes, tot / 5 / pairs * 1e3, best / pairs * 1e3, rows * cols * pairs * 5 / tot / 1e6), flush=True)
# single solve, all three orders, twice
for side in (3, 0, 3, 0, 3, 0):
    with rsdsfm.Solver(0, stream=stream.cuda_stream) as s:
        s.set_frame_side_flatten(side)
        call = s.prepared_frame_solve(imgs[0].data_ptr(), rows, cols, meta["K"], meta["gamma"], dms[0].data_ptr(), Rs[0].data_ptr(), ts[0].data_ptr(), trials=50, tol=0.05)
        for i in range(5):
            call(1 + i)
        tt = []
        for i in range(60):
            t0 = time.perf_counter()
            call(10 + i)
            tt.append(time.perf_counter() - t0)
        tt.sort()
        print("single solve, side_flatten %d: median %.4f ms, min %.4f" % (side, tt[30] * 1e3, tt[0] * 1e3), flush=True)
