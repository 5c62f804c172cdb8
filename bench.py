#!/usr/bin/env python3
"""bench.py -- headline benchmark: Mpixels/s of the RS depth(+pose) solve on a synthetic 1280x720 pair.

Contract: python bench.py --gpus N --steps K --warmup W  prints ONE JSON line on rank 0.
  step      = one pass of the hot path over one synthetic frame pair resident in HBM.
  workload  = BASELINE.json configs[1]: 1280x720, dense per-pixel depth solve (Ceres-LM emulation, the
              reference-matching mode), pose (v, w) fixed.  `--workload full` times the whole solve instead.
  N > 1     = one process per GPU (torch.distributed / RCCL), each rank solving its own frame pair
              (sequence-throughput mode, BASELINE configs[4]); no data-path collective -> scaling "weak".
Frame pairs rotate through enough distinct HBM buffers to exceed the 256 MiB Infinity Cache, so the timed
loop streams from HBM, not from L3.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_PIXEL_DEPTH = 56  # read q 16 + u 16 + alpha 8 + alpha_k 8, write rho 8
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(data, v, w, budget_s=12.0):
    """The CPU oracle (a scalar port of the reference path, single thread like Ceres' default num_threads=1)
    timed on this box's host cores on a bounded sample: whole 1280x720 depth solves repeated for ~budget_s."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O

    O.lib()
    q, u, a, ak = data["q"], data["u"], data["alpha"], data["alpha_k"]
    t0 = time.perf_counter()
    reps = 0
    while True:
        O.estimate_inverse_depths(q, u, v, w, 0.0, a, ak, mode=1)
        reps += 1
        el = time.perf_counter() - t0
        if el >= budget_s or reps >= 200:
            break
    pix = data["rows"] * data["cols"] * reps
    return {"value": pix / el / 1e6, "unit": "Mpixels/s", "cores": 1, "kind": "port",
            "sample": "%d x full 1280x720 dense depth solve (oracle rso_estimate_inverse_depths, LM mode), %.1f s" % (reps, el)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="depth", choices=["depth", "depth_closed_form"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--nbuf", type=int, default=7, help="rotating HBM buffer sets (7 x 59 MB > 256 MiB L3)")
    args = ap.parse_args()

    import numpy as np
    import torch  # first: one HIP runtime per process (torch's), shared with librsdsfm_hip.so
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import rsdsfm

    # ---- synthetic input: BASELINE config 2 (1280x720, analytic scene, noise-free), one pair per rank ----
    data = rsdsfm.synth.make_config(2, seed=0x5EED0002 + rank)
    rows, cols = data["rows"], data["cols"]
    n = len(data["q"])
    t = data["truth"]
    v = t["v"] / np.linalg.norm(t["v"])  # unit translation, as the minimal solver returns it (minimal.cc:102-105)
    w = t["w"]
    k = 0.0
    mode = rsdsfm.DEPTH_CERES_LM if args.workload == "depth" else rsdsfm.DEPTH_CLOSED_FORM

    nbuf = args.nbuf  # default 7 x (48+8) B x 921600 = 361 MB > 256 MiB Infinity Cache
    sets = []
    for _ in range(nbuf):
        sets.append(dict(
            q=torch.from_numpy(data["q"]).to(dev), u=torch.from_numpy(data["u"]).to(dev),
            a=torch.from_numpy(data["alpha"]).to(dev), ak=torch.from_numpy(data["alpha_k"]).to(dev),
            rho=torch.empty(n, dtype=torch.float64, device=dev)))
    # a NON-default torch stream: its handle is non-null, so the library adopts it (a null handle would make the
    # context create a private stream) and torch.cuda.Event timings see the kernels
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    solver = rsdsfm.Solver(local_rank, stream=stream.cuda_stream)

    calls = [solver.prepared_depth_step(s["q"].data_ptr(), s["u"].data_ptr(), n, v, w, k, s["a"].data_ptr(),
                                        s["ak"].data_ptr(), s["rho"].data_ptr(), mode=mode) for s in sets]

    def step(i):
        calls[i % nbuf]()

    def barrier():
        if world > 1:
            dist.barrier()

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    # the fast path must have completed the LM state machine on its own (no extra launches needed)
    extra = 0
    if mode == rsdsfm.DEPTH_CERES_LM:
        s = sets[(args.warmup - 1) % nbuf] if args.warmup > 0 else sets[0]
        if args.warmup == 0:
            step(0)
        summary, extra = solver.depth_finish_dev(s["q"].data_ptr(), s["u"].data_ptr(), n, v, w, k, s["a"].data_ptr(),
                                                 s["ak"].data_ptr(), s["rho"].data_ptr())
    else:
        summary = None

    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    barrier()
    el = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())

    # correctness of what was timed: state machine finished inside the fixed launch sequence, result sane
    if mode == rsdsfm.DEPTH_CERES_LM:
        s = sets[(args.steps - 1) % nbuf]
        summary, extra2 = solver.depth_finish_dev(s["q"].data_ptr(), s["u"].data_ptr(), n, v, w, k, s["a"].data_ptr(),
                                                  s["ak"].data_ptr(), s["rho"].data_ptr())
        extra = max(extra, extra2)
    rho = sets[(args.steps - 1) % nbuf]["rho"].cpu().numpy()
    rho_true = (1.0 / t["Z"]).T.reshape(-1) * np.linalg.norm(t["v"])
    max_rel = float(np.max(np.abs(rho - rho_true) / np.abs(rho_true)))

    # ---- dominant-kernel duration, HIP events on the launch stream (torch's current stream) ----
    kern_ms = None
    if rank == 0:
        reps = max(20, min(args.steps, 200))
        e0 = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
        e1 = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
        for i in range(reps):
            s = sets[i % nbuf]
            e0[i].record(stream)
            if mode == rsdsfm.DEPTH_CERES_LM:
                solver.depth_lm_launch_dev(s["q"].data_ptr(), s["u"].data_ptr(), n, v, w, k, s["a"].data_ptr(),
                                           s["ak"].data_ptr(), s["rho"].data_ptr(), launch_id=0)
            else:
                step(i)
            e1[i].record(stream)
        torch.cuda.synchronize()
        ts = sorted(a_.elapsed_time(b_) for a_, b_ in zip(e0, e1))
        kern_ms = float(np.mean(ts))
        kern_ms_median = float(ts[len(ts) // 2])

    if rank == 0:
        pixels_per_step = rows * cols * world
        value = pixels_per_step * args.steps / el / 1e6
        alg_bytes = ALG_BYTES_PER_PIXEL_DEPTH * n
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get(args.workload)
            except Exception:
                traffic = None
        line = {
            "metric": "Mpixels/sec RS depth+pose solve, 1280x720 pair",
            "value": value, "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: synthetic 1280x720 pair, per-pixel depth solve only "
                                   "(%s), pose fixed; one pair per GPU, %d rotating HBM buffer sets" %
                                   ("Ceres-1.14 LM emulation" if mode == 1 else "closed-form GN", nbuf),
                       "rows": rows, "cols": cols, "pixels": n, "depth_mode": int(mode),
                       "launches_per_step": 3 if mode == 1 else 1, "extra_lm_launches": int(extra),
                       "lm_summary": summary, "max_rel_err_vs_truth": max_rel},
            "roofline": {"bound": "hbm", "kernel": "depth_lm_kernel" if mode == 1 else "depth_closed_form_kernel",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "alg_bytes_per_launch": alg_bytes, "avg_launch_ms": kern_ms,
                         "median_launch_ms": kern_ms_median},
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(data, v, w)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    solver.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
