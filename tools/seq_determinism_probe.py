"""diagnostic: is the sequence solve deterministic?  the same 32 pairs / seeds several times; compares iteration counts, v and the depth maps bitwise.
   python tools/seq_determinism_probe.py [refine_arithmetic] [lanes] [repeats]"""
import faulthandler, os, sys, time
faulthandler.dump_traceback_later(80, exit=True)
sys.path.insert(0, ".")
import numpy as np, torch
import rsdsfm
ra = int(sys.argv[1]) if len(sys.argv) > 1 else 0
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
pairs = 32
dev = torch.device("cuda", 0)
frames, meta = rsdsfm.synth.make_flow_sequence(5, [0x5EED0005 + 1000 * i for i in range(pairs)])
rows, cols = meta["rows"], meta["cols"]
imgs = [torch.from_numpy(f).to(dev) for f in frames]
dms = [torch.empty((cols, rows), dtype=torch.float64, device=dev) for _ in range(pairs)]
jobs = [dict(d_flow_img=im.data_ptr(), rows=rows, cols=cols, K=meta["K"], gamma=meta["gamma"], d_depth_map=dm.data_ptr(), d_R=None, d_t=None) for im, dm in zip(imgs, dms)]
with rsdsfm.Solver(0) as s:
    s.set_refine_arithmetic(ra)
    s.lib.rsdsfm_set_sequence_lanes(s._ctx, lanes)
    call = s.prepared_frames_solve(jobs, trials=50, tol=0.05, use_acceleration_mode=bool(int(os.environ.get("PROBE_ACCEL", "0"))))
    ref = None
    for p in range(reps):
        res = call([1001 + i for i in range(pairs)])
        s.synchronize()
        sig = [(int(r.num_inliers), int(r.refine_summary.num_iterations), int(r.refine_summary.termination), bytes(bytearray(np.array(r.v[:]).tobytes())), dm.cpu().numpy().tobytes()) for r, dm in zip(res, dms)]
        if ref is None:
            ref = sig
        diff = [i for i in range(pairs) if sig[i] != ref[i]]
        print("rep", p, "pairs that differ from rep 0:", diff, [(ref[i][1], sig[i][1]) for i in diff][:6], s.refine_restarts(), flush=True)
print("done")
