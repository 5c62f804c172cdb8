#!/usr/bin/env python3
"""bench.py -- headline benchmark: Mpixels/s of the WHOLE rolling-shutter depth+pose solve on a synthetic 1280x720 pair.

Contract: python bench.py --gpus N --steps K --warmup W  prints ONE JSON line on rank 0.
  step      = one pass of the hot path over one batch of synthetic input resident in HBM.
  workload  = full (default): BASELINE.json's metric -- the whole solve of one 1280x720 DeepFlow-like frame pair (0.3 px noise,
              10 % outliers): flatten + alpha / alpha_k, RANSAC (50 trials, tol 0.05: 9-point minimal solver, hypothesis-batched
              Ceres-LM depth solves of ALL pixels, scoring, best trial, compaction), joint nonlinear refinement, sign fix + depth
              map, per-scanline pose table (reference main.cc:398-522), ONE C-ABI call per pair (rsdsfm_solve_frame_dev), one
              pair at a time.  `value` = pixels of the K timed pairs / the time of the K steps; `median_ms_per_solve` beside it.
              Sub-records of the line:
                roofline            the dominant kernel ransac_lma_kernel<2, true> (fp64 VALU bound: counted fp64 lane-instructions / its
                                    launch duration, measured here with HIP events, / 39.3e12; counts from profiles/counters.json, which
                                    is stamped with the kernel source hashes -- `counters_stale` when they do not match) and, under
                                    "hbm", SURVEY 8(d)'s whole-solve figure (57 N + 64 M iters bytes / solve time / 8 TB/s)
                full_solve_exact_kernel  the same workload with the RANSAC's depth solves iterate by iterate (rsdsfm_set_lm_arithmetic(1))
                lma_restarts        solves whose analytic pass tripped a guard and started over iterate by iterate (+ which guards)
                regimes             the same solve, driver-timed, with T = 5, a selective tolerance, noise-free flow, acceleration mode, 1920x1080
                full_solve_batched  BASELINE configs[4]: 32 pairs with 32 data seeds through rsdsfm_solve_frames_dev (one context, one host thread)
                tiled_full          north_star's multi-GPU claim: ONE 3840x2160 frame in N column slabs through the native RCCL driver
                                    (scaling "strong"), with the DESIGN section 8 model beside the measurement; runs last, under a watchdog
                cpu_baseline        the oracle's whole solve of the SAME pair, 1 thread (+ all_cores, + reference_structured: per-call
                                    problem build with per-pixel heap objects and dual-number Jacobians)
                depth_only          BASELINE configs[1] (dense depth solve alone, pose fixed, batched sequence mode; HBM roofline)
              depth / depth_closed_form / tiled / tiled_full / rectify / true_flow / metrics : those workloads as the timed one.
  N > 1     = one process per GPU.  `python bench.py --gpus N` starts its N ranks itself (the parent never touches a GPU; --dry-launch
              prints the rank environments); under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` the ranks
              run as launched.  `value` = N independent replicas of the whole solve (BASELINE configs[4], no data-path collective ->
              scaling "weak"); `tiled_full` = the strong-scaling record.  RSDSFM_SHARE_GPU=1 puts all ranks on device 0 with one
              NCCL_HOSTID each (RCCL then connects them over its socket transport: the N-rank path on a one-GPU box).
  --arith fused = the opt-in librsdsfm_hip_fused.so (explicit fmas also in the iterate-by-iterate per-pixel model) instead of the
              default library (analytic LM trajectory + radius-factorised refinement: the library's own arithmetic with fused
              multiply-adds, guarded to the integers of the reference's arithmetic; its iterate-by-iterate kernels are the reference's
              arithmetic operation for operation); the default line carries the fused library's whole-solve time as `full_solve_fused`.
Frame pairs rotate through distinct HBM buffers, so the timed loops stream from HBM, not from the 256 MiB Infinity Cache.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_PIXEL_DEPTH = 56  # read q 16 + u 16 + alpha 8 + alpha_k 8, write rho 8
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec
METRIC = "Mpixels/sec RS depth+pose solve, 1280x720 pair"
METRIC_DEPTH = "Mpixels/sec RS per-pixel depth solve (pose fixed), 1280x720 pair"
FP64_VALU_PEAK = 39.3e12        # fp64 lane-instructions / s: 256 CUs x 4 SIMDs x 64 lanes x 2.4 GHz / 4 cycles per wave-instruction
NOMINAL_CLOCK_MHZ = 2400.0      # the clock that peak is priced at (MI355X_MICROARCH.md); the kernels' running clock is measured (shader_clock_mhz)
# DESIGN section 7 scaling model of the column-tiled 3840x2160 whole solve (ms; calibrated on the N = 1 kernel trace
# profiles/r05_trace_tiled_full.txt and the measured 3.26 ms per solve): kernels whose work is per pixel of the slab (the analytic pixel
# pass 1.49, the refinement's slot passes 0.69, iteration zero 0.18, compaction 0.13, flatten 0.13, final stage 0.10, output pass 0.10,
# depth map 0.13) / replicated or latency-bound stages (minimal9 0.16 + ~55 launches and small copies of 2-8 us) / an ASSUMED 25 us per
# small collective over xGMI (9 per solve of 4 LM iterations) / the depth-map all-gather: every rank receives (N - 1) slabs of 66.4 MB / N
# over min(N - 1, 7) links in parallel at an ASSUMED 48 GB/s per link and direction
TILED_MODEL = {"per_pixel_ms": 2.83, "replicated_ms": 0.43, "collective_latency_ms": 0.025, "collectives": 9,
               "depth_gather_ms": lambda n: (66.4e6 / n * (n - 1) / min(n - 1, 7)) / 48e9 * 1e3}
KIND_PORT = "closed-loop port (oracle C restatement; not reference-structured: no per-pixel residual objects / Ceres problem build)"


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
        if el >= budget_s or reps >= 1000:
            break
    pix = data["rows"] * data["cols"] * reps
    return {"value": pix / el / 1e6, "unit": "Mpixels/s", "cores": 1, "kind": "port",
            "sample": "%d x full 1280x720 dense depth solve (oracle rso_estimate_inverse_depths, LM mode), %.1f s" % (reps, el)}


def cpu_baseline_all_cores(data, v, w, budget_s=6.0):
    """the same oracle source built with OpenMP over the per-pixel loops, on all host cores of this box"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O

    try:
        O.lib_omp()
    except Exception as e:  # no compiler / OpenMP runtime on the box: report it instead of failing the bench
        return {"value": None, "error": repr(e)[:200]}
    import ctypes

    q, u, a, ak = data["q"], data["u"], data["alpha"], data["alpha_k"]
    try:
        gomp = ctypes.CDLL("libgomp.so.1")
    except OSError:
        gomp = None
    ncpu = os.cpu_count() or 1
    best = None
    # the solve streams 56 B/pixel: it stops scaling long before 256 threads, so a short sweep picks the best team size
    for nt in sorted({ncpu, max(1, ncpu // 2), max(1, ncpu // 4), max(1, ncpu // 8), min(ncpu, 16), min(ncpu, 8)}, reverse=True):
        if gomp is not None:
            gomp.omp_set_num_threads(int(nt))
        O.estimate_inverse_depths_all_cores(q, u, v, w, 0.0, a, ak, mode=1)  # thread team warm-up
        t0 = time.perf_counter()
        reps = 0
        while True:
            O.estimate_inverse_depths_all_cores(q, u, v, w, 0.0, a, ak, mode=1)
            reps += 1
            el = time.perf_counter() - t0
            if el >= budget_s / 6 or reps >= 100:
                break
        rate = data["rows"] * data["cols"] * reps / el / 1e6
        if best is None or rate > best["value"]:
            best = {"value": rate, "unit": "Mpixels/s", "cores": int(nt) if gomp is not None else ncpu, "kind": "port (OpenMP over pixels)",
                    "sample": "%d x full 1280x720 dense depth solve, %.1f s; best of a thread-count sweep up to %d host cores" % (reps, el, ncpu)}
        if gomp is None:
            break
    return best


def cpu_baseline_full(rsdsfm, np, rank, trials, tol, budget_s=20.0):
    """The oracle's whole solve (flatten, alpha, RANSAC with the SAME trial count, refinement, sign fix, depth map, pose table) of the
    SAME 1280x720 DeepFlow-like pair the GPU solves, single thread like Ceres' default num_threads = 1; repeated while the budget
    lasts (one solve takes a few seconds), median reported."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O

    O.lib()
    d = rsdsfm.synth.make_config(5, seed=0x5EED0005 + rank)
    rows, cols = d["rows"], d["cols"]
    times = []
    t_all = time.perf_counter()
    while True:
        t0 = time.perf_counter()
        q, u, qpx, fpx = O.flatten(d["flow_img"], *d["K"], d["gamma"])
        a, ak = O.get_alpha(fpx, rows, d["gamma"]), O.get_alpha_k(qpx, fpx, rows, d["gamma"])
        r = O.ransac(q, u, a, ak, False, trials, tol, O.sample_indices(len(q), trials, 1), depth_mode=1)
        ref = O.refine(u, r["inliers"], r["alpha"], r["alpha_k"], r["v"], r["w"], r["k"], False, 1, r["inlier_idx"])
        inl, v, _ = O.canonicalize_sign(ref["inliers"], ref["v"])
        O.scatter_depth(inl, *d["K"], rows, cols)
        O.pose_table(v, ref["w"], ref["k"], d["gamma"], rows)
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_all + times[-1] > budget_s or len(times) >= 20:
            break
    med = sorted(times)[len(times) // 2]
    return {"value": rows * cols / med / 1e6, "unit": "Mpixels/s", "cores": 1, "kind": "port", "kind_detail": KIND_PORT,
            "host_cores": os.cpu_count(), "seconds_per_solve": med,
            "sample": "%d x the whole solve of the same 1280x720 DeepFlow-like pair with the same %d RANSAC trials (oracle chain, 1 thread), "
                      "median of %d; %.1f s in total" % (len(times), trials, len(times), sum(times)), "trials": trials}


def cpu_baseline_full_all_cores(rsdsfm, np, rank, trials, tol, single, budget_s=10.0):
    """the same whole solve on the OpenMP build of the oracle (SURVEY section 8 d: "plus an all-cores variant"): the per-pixel loops
    of the T depth solves, the scoring and the refinement passes run on a thread team; best of a short thread-count sweep"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O

    try:
        O.lib_omp()
        import ctypes

        gomp = ctypes.CDLL("libgomp.so.1")
    except Exception as e:  # no compiler / OpenMP runtime on the box: say so instead of failing the bench
        return {"value": None, "error": repr(e)[:200]}
    d = rsdsfm.synth.make_config(5, seed=0x5EED0005 + rank)
    rows, cols = d["rows"], d["cols"]
    ncpu = os.cpu_count() or 1
    best, t_all = None, time.perf_counter()
    with O.all_cores():
        for nt in sorted({min(ncpu, 128), min(ncpu, 64), min(ncpu, 32), min(ncpu, 16)}, reverse=True):
            gomp.omp_set_num_threads(int(nt))
            times = []
            for rep in range(3):
                t0 = time.perf_counter()
                q, u, qpx, fpx = O.flatten(d["flow_img"], *d["K"], d["gamma"])
                a, ak = O.get_alpha(fpx, rows, d["gamma"]), O.get_alpha_k(qpx, fpx, rows, d["gamma"])
                r = O.ransac(q, u, a, ak, False, trials, tol, O.sample_indices(len(q), trials, 1), depth_mode=1)
                ref = O.refine(u, r["inliers"], r["alpha"], r["alpha_k"], r["v"], r["w"], r["k"], False, 1, r["inlier_idx"])
                inl, v, _ = O.canonicalize_sign(ref["inliers"], ref["v"])
                O.scatter_depth(inl, *d["K"], rows, cols)
                O.pose_table(v, ref["w"], ref["k"], d["gamma"], rows)
                times.append(time.perf_counter() - t0)
                if time.perf_counter() - t_all > budget_s:
                    break
            med = sorted(times)[len(times) // 2]
            if best is None or med < best["seconds_per_solve"]:
                best = {"value": rows * cols / med / 1e6, "unit": "Mpixels/s", "cores": int(nt), "host_cores": ncpu, "seconds_per_solve": med,
                        "kind": "port (OpenMP over pixels)", "num_inliers": int(r["num_inliers"]),
                        "sample": "the whole solve of the same pair, %d trials, oracle built with -fopenmp; best median of a thread-count sweep" % trials}
            if time.perf_counter() - t_all > budget_s:
                break
    if best and single:
        best["speedup_over_1_thread"] = single["seconds_per_solve"] / best["seconds_per_solve"]
    return best


def cpu_baseline_reference_structured(rsdsfm, np, rank, trials, tol, budget_s=25.0):
    """BASELINE.md section 3.1's `cpu_ref`: the oracle's arithmetic in the reference's STRUCTURE -- per estimateInverseDepths call (one per
    RANSAC trial, nonlinearRefinement.cc:130-163) a problem is built from scratch: one heap-allocated residual object and one parameter
    block per pixel, an ordering pass over the N blocks, and every residual is evaluated through Jet<double, 8>-style dual numbers (value +
    8 derivatives) at each LM iteration, as Ceres' AutoDiffCostFunction<RsResidual, 2, 3, 3, 1, 1> does; one thread (Ceres num_threads = 1).
    Same 1280x720 DeepFlow-like pair; as many of the `trials` trials as fit the budget (stated in the record), scaled to the whole solve."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O

    if not hasattr(O, "ransac_reference_structured"):
        return None
    O.lib()
    d = rsdsfm.synth.make_config(5, seed=0x5EED0005 + rank)
    rows, cols = d["rows"], d["cols"]
    q, u, qpx, fpx = O.flatten(d["flow_img"], *d["K"], d["gamma"])
    a, ak = O.get_alpha(fpx, rows, d["gamma"]), O.get_alpha_k(qpx, fpx, rows, d["gamma"])
    samples = O.sample_indices(len(q), trials, 1)
    # one trial first: it prices the rest
    t0 = time.perf_counter()
    O.ransac_reference_structured(q, u, a, ak, False, 1, tol, samples[:1])
    t1 = time.perf_counter() - t0
    T = int(max(1, min(trials, (budget_s - t1) // max(t1, 1e-3))))
    t0 = time.perf_counter()
    r = O.ransac_reference_structured(q, u, a, ak, False, T, tol, samples[:T])
    t_ransac = time.perf_counter() - t0
    t0 = time.perf_counter()
    O.refine_reference_structured(u, r["inliers"], r["alpha"], r["alpha_k"], r["v"], r["w"], r["k"], False, 0, None)
    t_refine = time.perf_counter() - t0
    sec = t_ransac / T * trials + t_refine
    return {"value": rows * cols / sec / 1e6, "unit": "Mpixels/s", "cores": 1, "kind": "port, reference-structured",
            "seconds_per_solve_extrapolated": sec, "seconds_per_trial": t_ransac / T, "seconds_refinement": t_refine, "trials_timed": T, "trials": trials,
            "sample": "%d of the %d RANSAC trials (per trial: problem build with one heap residual object + parameter block per pixel, ordering pass, "
                      "Jet<double,8>-style evaluation of all %d residuals per LM iteration) + the joint refinement built the same way, 1 thread; "
                      "whole solve = trials x per-trial time + refinement" % (T, trials, len(q))}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default per workload: 100 whole solves; 160 chunks of 64 pairs for the depth workloads)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="full", choices=["depth", "depth_closed_form", "full", "tiled", "tiled_full", "rectify", "true_flow", "metrics", "launch_check"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sequence-only", action="store_true", help="full workload: only the sequence solve (rsdsfm_solve_frames_dev), --steps passes of 32 pairs (profiling runs)")
    ap.add_argument("--no-side-records", action="store_true", help="full workload: skip depth_only / full_solve_batched / full_solve_fused (profiling runs)")
    ap.add_argument("--tiled-driver", default="native", choices=["native", "python"], help="tiled / tiled_full: the C++ driver inside the library (default) or the Python driver")
    ap.add_argument("--arith", default="reference", choices=["reference", "fused"], help="library: 'reference' = the default librsdsfm_hip.so (see --help's header), 'fused' = the opt-in fused-fma build")
    ap.add_argument("--nbuf", type=int, default=7, help="rotating HBM buffer sets (7 x 59 MB > 256 MiB L3)")
    ap.add_argument("--streams", type=int, default=2, help="HIP streams per GPU feeding independent batches (sequence-throughput mode, BASELINE configs[4])")
    ap.add_argument("--pairs-per-step", type=int, default=64, help="depth workloads: one step = one chunk of this many consecutive frame pairs of the sequence (default 64)")
    ap.add_argument("--batch", type=int, default=8, help="independent frame pairs per launch of the batched depth fast path (1..8, one solver context each)")
    ap.add_argument("--depth-variant", type=int, default=None, help="rsdsfm_set_depth_variant: 0 register-staged, 1 LDS-DMA, 2 decision fused into launch 0")
    ap.add_argument("--trials", type=int, default=50, help="RANSAC trials of the full solve (report section 5.4 used 50)")
    ap.add_argument("--tol", type=float, default=0.05, help="RANSAC tolerance (reference main.cc:310)")
    ap.add_argument("--dry-launch", action="store_true", help="print the rank environments / command lines the launcher would start (JSON) and exit; starts nothing, touches no GPU")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="launcher: seconds after which the ranks are ended")
    ap.add_argument("--launch-grace", type=float, default=10.0, help="launcher: seconds the other ranks get after one rank failed")
    ap.add_argument("--hang-dump", type=float, default=0.0, help="seconds after which every thread's Python stack is written to stderr (diagnosing a bench that does not come back); 0 = off")
    ap.add_argument("--tiled-timeout", type=float, default=240.0, help="full workload: watchdog of the tiled_full (strong scaling) sub-record; on expiry the line is printed without it")
    args = ap.parse_args(argv)
    dsteps = {"depth": 160, "depth_closed_form": 160, "full": 100, "tiled": 2000, "tiled_full": 100, "rectify": 2000, "true_flow": 500, "metrics": 2000, "launch_check": 3}
    if args.steps is None:
        args.steps = dsteps[args.workload]
    if args.warmup is None:
        args.warmup = max(3, args.steps // 25)
    return args


def rank_environments(gpus, port=None, base_env=None):
    """the environments of the N rank processes the launcher starts (what `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1` would set, minus its agent): rank i on GPU i, rendezvous on 127.0.0.1"""
    if port is None:
        import socket

        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    envs = []
    for r in range(gpus):
        e = dict(os.environ if base_env is None else base_env)
        e.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(gpus), "LOCAL_WORLD_SIZE": str(gpus), "MASTER_ADDR": "127.0.0.1",
                  "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": e.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), "RSDSFM_BENCH_CHILD": "1"})
        envs.append(e)
    return envs


def launch_ranks(args, argv):
    """`python bench.py --gpus N` (N > 1) WITHOUT a launcher: this process -- which has not imported torch and never touches a GPU --
    starts the N ranks as child processes (never an exec), relays rank 0's single JSON line and exits with the worst child exit code.
    --dry-launch prints what would be started instead."""
    import subprocess

    envs = rank_environments(args.gpus)
    child_argv = [a for a in argv if a != "--dry-launch"]
    cmd = [sys.executable, os.path.abspath(__file__)] + child_argv
    if args.dry_launch:
        keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY")
        print(json.dumps({"dry_launch": True, "n_ranks": args.gpus, "cmd": cmd, "ranks": [{k2: e[k2] for k2 in keys} for e in envs],
                          "backend": os.environ.get("RSDSFM_DIST_BACKEND", "nccl"),
                          "equivalent": "python -m torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 --master-port %s bench.py %s"
                                        % (args.gpus, envs[0]["MASTER_PORT"], " ".join(child_argv))}), flush=True)
        return 0
    procs = []
    for r, e in enumerate(envs):  # rank 0's stdout carries the line; the other ranks' output goes to this process' stderr
        procs.append(subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    import threading

    out0 = []
    reader = threading.Thread(target=lambda: out0.extend(procs[0].stdout.read().decode(errors="replace").splitlines()), daemon=True)
    reader.start()
    deadline = time.time() + args.launch_timeout
    codes = [None] * len(procs)
    failed_at = None
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
                if codes[i] not in (None, 0) and failed_at is None:
                    failed_at = time.time()
        now = time.time()
        # one rank died (the others would wait in a collective for ever) or the time is up: end exactly the processes started here
        if (failed_at is not None and now - failed_at > args.launch_grace) or now > deadline:
            for i, p in enumerate(procs):
                if codes[i] is None:
                    p.kill()
                    codes[i] = p.wait()
                    codes[i] = codes[i] if codes[i] != 0 else 1
            break
        time.sleep(0.05)
    reader.join(timeout=10.0)
    line = None
    for ln in out0:
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    worst = max((abs(c) for c in codes), default=0)
    if line is not None:
        print(line, flush=True)
    elif worst == 0:
        worst = 1
    return worst


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.hang_dump > 0:
        import faulthandler

        faulthandler.dump_traceback_later(args.hang_dump, repeat=True, file=sys.stderr)
    # a launched rank has WORLD_SIZE == --gpus in its environment (torch.distributed.run or launch_ranks above); anything else with
    # --gpus N > 1 is the bare command line, which must become N ranks itself -- before torch is imported or a GPU is touched
    if args.dry_launch or (args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) <= 1):
        raise SystemExit(launch_ranks(args, argv))
    run(args)


def run(args):
    import numpy as np
    import torch  # first: one HIP runtime per process (torch's), shared with librsdsfm_hip.so
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.workload == "launch_check":
        # the launcher and the timing protocol without a GPU (tests/test_bench_launch.py, world_size 2 over gloo): rendezvous,
        # barrier-bracketed timed region, MAX over ranks, one JSON line from rank 0; RSDSFM_LAUNCH_CHECK_FAIL_RANK makes one rank die
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("RSDSFM_LAUNCH_CHECK_FAIL_RANK") == str(rank):
            raise SystemExit(3)
        seen = [rank]
        el = 0.0
        if world > 1:
            dist.init_process_group(os.environ.get("RSDSFM_DIST_BACKEND", "gloo"), rank=rank, world_size=world)
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                time.sleep(0.01 * (rank + 1))
            dist.barrier()
            tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
            got = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(got, torch.tensor([rank], dtype=torch.int64))
            seen = [int(g.item()) for g in got]
            if os.environ.get("RSDSFM_LAUNCH_CHECK_HANG_RANK") is not None:
                # the watchdog of the bench's last section (tiled_full): one rank never comes back, the others wait for it in a
                # collective -- rank 0 must still print the line it has, and every rank must leave
                line = {"metric": "launch_check", "value": el, "unit": "s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ranks_seen": seen, "guarded": None}
                _watchdog(args.tiled_timeout, rank, line, "guarded")
                if os.environ["RSDSFM_LAUNCH_CHECK_HANG_RANK"] == str(rank):
                    time.sleep(3600)
                dist.barrier()
                time.sleep(3600)  # (not reached in time: the barrier needs the hanging rank)
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"metric": "launch_check", "value": el, "unit": "s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ranks_seen": seen, "local_rank_of_rank0": local_rank, "master": "%s:%s" % (os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT"))}), flush=True)
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    if os.environ.get("RSDSFM_SHARE_GPU"):
        # development boxes have ONE GPU: several ranks share device 0.  RCCL refuses two ranks of one HOST on one device, but it
        # identifies the host by NCCL_HOSTID when that is set -- with one id per rank it treats the ranks as single-GPU nodes and
        # connects them through its socket transport over the loopback interface.  Not xGMI (the rates mean nothing), but the
        # multi-rank RCCL path for real: torch's process group and the library's own communicator with N ranks.
        local_rank = 0
        os.environ.setdefault("NCCL_HOSTID", "rsdsfm-shared-gpu-rank-%d" % rank)
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_IB_DISABLE", "1")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("RSDSFM_DIST_BACKEND", "nccl")  # "nccl" IS RCCL on ROCm; "gloo" only for smoke tests
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import rsdsfm

    # a NON-default torch stream: its handle is non-null, so the library adopts it (a null handle would make the
    # context create a private stream) and torch.cuda.Event timings see the kernels
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    solver = rsdsfm.Solver(local_rank, stream=stream.cuda_stream, arith=args.arith)
    if args.depth_variant is not None:
        solver.set_depth_variant(args.depth_variant)

    def barrier():
        if world > 1:
            dist.barrier()

    def timed(step, steps, warmup):
        for i in range(warmup):
            step(i)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        torch.cuda.synchronize()
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el

    line = {"metric": METRIC, "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "higher_is_better": True, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "arith": ("default library: analytic LM trajectory (depth solves) + radius-factorised refinement, own arithmetic with fused multiply-adds guarded to the "
                      "integers of the reference's arithmetic; iterate-by-iterate kernels = the reference's arithmetic" if args.arith == "reference" else
                      "opt-in fused library: the default's paths, explicit fused multiply-adds also in the iterate-by-iterate per-pixel model")}

    # =================================================================================================
    def run_depth(workload, steps, warmup, side_records=True, cpu_budget=12.0):
        """BASELINE configs[1]: the dense depth solve alone (pose fixed) in batched sequence-throughput mode; returns the record"""
        rec = {}
        data = rsdsfm.synth.make_config(2, seed=0x5EED0002 + rank)  # 1280x720, analytic scene, noise-free
        rows, cols = data["rows"], data["cols"]
        n = len(data["q"])
        t = data["truth"]
        v = t["v"] / np.linalg.norm(t["v"])  # unit translation, as the minimal solver returns it (minimal.cc:102-105)
        w, k = t["w"], 0.0
        mode = rsdsfm.DEPTH_CERES_LM if workload == "depth" else rsdsfm.DEPTH_CLOSED_FORM
        # Sequence-throughput mode: B independent pairs per launch (batched fast path: grid y = pair, one context per pair owns
        # its LM state and partial sums) on each of S HIP streams; nothing is shared between pairs.  The latency-bound
        # follow-up launch, the launch floor and the ramp / tail of the streaming pass are amortised over B pairs and
        # overlapped across the streams.  (closed-form mode has no batched entry point: B = 1 there.)
        S = max(1, args.streams)
        B = max(1, min(args.batch, 8)) if mode == rsdsfm.DEPTH_CERES_LM else 1
        streams = [stream] + [torch.cuda.Stream(dev) for _ in range(S - 1)]
        G = max(2, -(-max(args.nbuf, 9) // (B * S)))  # groups per stream: >= 9 rotating buffer sets in total (> 256 MiB L3)
        rho_true = (1.0 / t["Z"]).T.reshape(-1) * np.linalg.norm(t["v"])
        extra_solvers, keep = [], []

        def new_solver(st):
            if st is stream and not extra_solvers:
                extra_solvers.append(None)  # the first context on stream 0 is the bench's main solver
                return solver
            sv = rsdsfm.Solver(local_rank, stream=st.cuda_stream, arith=args.arith)
            if args.depth_variant is not None:
                sv.set_depth_variant(args.depth_variant)
            extra_solvers.append(sv)
            return sv

        # Two kinds of frame pairs alternate in every context's sequence, in runs of RUN: the noise-free pair (the emulated Ceres loop
        # ends after 3 accepted steps) and the DeepFlow-like pair of BASELINE configs[4] (0.3 px noise + 10 % outliers: 2 accepted
        # steps).  The context's device-resident predictor (accepted-step count of its previous solve) is therefore WRONG once per
        # run and the solve then pays its apply pass -- a sequence whose statistics change every RUN pairs, not the best case of
        # identical data (reported beside it as `identical_pairs`).
        RUN = 8
        data_b = rsdsfm.synth.make_config(5, seed=0x5EED0005 + rank) if mode == rsdsfm.DEPTH_CERES_LM else None
        assert data_b is None or len(data_b["q"]) == n

        def new_set(dd=None):
            dd = data if dd is None else dd
            return dict(q=torch.from_numpy(dd["q"]).to(dev), u=torch.from_numpy(dd["u"]).to(dev), a=torch.from_numpy(dd["alpha"]).to(dev),
                        ak=torch.from_numpy(dd["alpha_k"]).to(dev), rho=torch.empty(n, dtype=torch.float64, device=dev))

        def ptrs(s):
            return (s["q"].data_ptr(), s["u"].data_ptr(), n, v, w, k, s["a"].data_ptr(), s["ak"].data_ptr(), s["rho"].data_ptr())

        def problem(s):
            return dict(d_q=s["q"].data_ptr(), d_u=s["u"].data_ptr(), d_alpha=s["a"].data_ptr(), d_alpha_k=s["ak"].data_ptr(), d_rho=s["rho"].data_ptr(),
                        n=n, v=v, w=w, k=k)

        groups = []  # (call, solvers, sets, call on the DeepFlow-like pairs, visits), stream-major
        for st in streams:
            for _ in range(G):
                svs = [new_solver(st) for _ in range(B)]
                sets = [new_set() for _ in range(B)]
                call_b = None
                if mode == rsdsfm.DEPTH_CERES_LM:
                    call = rsdsfm.prepared_depth_batch(svs, [problem(x) for x in sets])
                    sets_b = [new_set(data_b) for _ in range(B)]
                    call_b = rsdsfm.prepared_depth_batch(svs, [problem(x) for x in sets_b])
                    keep.append(sets_b)
                else:
                    call = svs[0].prepared_depth_step(*ptrs(sets[0]), mode=mode)
                groups.append((call, svs, sets, call_b, [0]))
        order = [groups[(i % S) * G + (i // S) % G] for i in range(S * G)]  # alternate the streams
        torch.cuda.synchronize()
        P = max(1, args.pairs_per_step)
        npairs, nwarm = steps * P, warmup * P  # one step = one chunk of P consecutive frame pairs of the sequence
        nfull, rem = divmod(npairs, B)
        rem_call = None
        if rem:  # exactly `steps` x P pairs are timed: the last call is a smaller batch over the first contexts of one group
            g = order[nfull % len(order)]
            rem_call = rsdsfm.prepared_depth_batch(g[1][:rem], [problem(x) for x in g[2][:rem]])

        mixed = [mode == rsdsfm.DEPTH_CERES_LM]

        def visit(g):  # one batched call on group g: noise-free or DeepFlow-like pairs, switching every RUN visits of the group
            kind_b = mixed[0] and (g[4][0] // RUN) % 2 == 1
            g[4][0] += 1
            (g[3] if kind_b else g[0])()
            return kind_b

        def run_pairs(i):  # pair i of the timed sequence; a batched call every B pairs
            if i % B == 0:
                j = i // B
                if j < nfull:
                    visit(order[j % len(order)])
                elif rem_call is not None:
                    rem_call()

        def reset_visits():
            for g in groups:
                g[4][0] = 0

        for i in range(0, max(nwarm, len(order) * B), B):
            visit(order[(i // B) % len(order)])
        el_same = None
        if mixed[0]:  # best case first: every pair identical (the predictor is always right)
            mixed[0] = False
            el_same = timed(run_pairs, npairs, 0)
            mixed[0] = True
            reset_visits()
        el = timed(run_pairs, npairs, 0)
        # leave every context with a noise-free pair as its last solve, so that the check below can compare with the analytic truth
        torch.cuda.synchronize()
        for g in order:
            g[0]()
        # correctness of what was timed: on the contexts of the last batches the LM state machine finished inside the fixed
        # launch sequence and the pair matches the analytic truth
        extra, summary, max_rel = 0, None, 0.0
        for j in range(max(0, nfull - len(order)), nfull):
            svs, sets = order[j % len(order)][1], order[j % len(order)][2]
            for sv, x in zip(svs, sets):
                if mode == rsdsfm.DEPTH_CERES_LM:
                    summary, ex = sv.depth_finish_dev(*ptrs(x))
                    extra = max(extra, ex)
                rho = x["rho"].cpu().numpy()
                max_rel = max(max_rel, float(np.max(np.abs(rho - rho_true) / np.abs(rho_true))))
        # the same work with ONE pair at a time (single context, single stream), for reference
        own = [grp for grp in groups[:G]]
        single_calls = [grp[1][0].prepared_depth_step(*ptrs(grp[2][0]), mode=mode) for grp in own]
        n1 = max(20, npairs // 8)
        el1 = timed(lambda i: single_calls[i % len(single_calls)](), n1, 5)

        # dominant-kernel duration.  LM mode: the streaming kernel (launch 0 of the batched solve = `depth_lma_batch_kernel`, B pairs per launch)
        # carries its DISPATCH's own start / stop timestamps when profiling is on for the batch's first context (rsdsfm_set_profiling:
        # hipExtLaunchKernel events -- what rocprofv3 --kernel-trace reports for the kernel, no inter-launch gap in it).  Bursts of BURST
        # ordinary batched calls (streaming launch + decide / apply launch) back to back on stream 0, the other streams idle; the record
        # read is the LAST call's, i.e. a launch under the sustained load the clocks settle on.  Closed-form mode (no batched entry point):
        # HIP events around the burst, divided by BURST (includes the ~1 us gaps: slightly conservative).
        kern_ms = kern_med = None
        if rank == 0:
            BURST, reps = 10, 30
            if mode == rsdsfm.DEPTH_CERES_LM:
                own_g = groups[:G]
                for grp in own_g:
                    grp[1][0].set_profiling(True)
                ts = []
                for i in range(reps):
                    for b_ in range(BURST):
                        last = own_g[(i * BURST + b_) % len(own_g)]
                        last[0]()
                    torch.cuda.synchronize()
                    ts.append(last[1][0].profile_last_ms("depth_lm_batch"))
                for grp in own_g:
                    grp[1][0].set_profiling(False)
                ts = sorted(ts[3:])
            else:
                burst_calls = single_calls
                e0 = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
                e1 = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
                for i in range(reps):
                    e0[i].record(stream)
                    for b_ in range(BURST):
                        burst_calls[(i * BURST + b_) % len(burst_calls)]()
                    e1[i].record(stream)
                torch.cuda.synchronize()
                ts = sorted(a_.elapsed_time(b_) / BURST for a_, b_ in zip(e0, e1))
            kern_ms, kern_med = float(np.mean(ts)), float(ts[len(ts) // 2])

        # the streaming kernel of the batched fast path: the analytic LM trajectory's (depth_lma_kernels.hip; its template argument: pixel
        # pairs per thread and iteration)
        batch_kernel = "depth_lma_batch_kernel<2>"
        if rank == 0:
            alg_bytes = ALG_BYTES_PER_PIXEL_DEPTH * n * B
            achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
            job = ALG_BYTES_PER_PIXEL_DEPTH * n / (el / npairs) / 1e9
            rec.update({
                "value": rows * cols * world * npairs / el / 1e6, "ms_per_step": el / steps * 1e3, "scaling": "weak",
                "config": {"workload": "BASELINE configs[1]: synthetic 1280x720 pairs, per-pixel depth solve only (%s), pose fixed; "
                                       "%d independent pairs per launch on each of %d HIP streams per GPU (one solver context per pair), %d "
                                       "rotating HBM buffer sets" % ("Ceres-1.14 LM emulation" if mode == 1 else "closed-form GN", B, S, B * S * G),
                           "pairs_per_step": P, "pairs_timed": npairs, "ms_per_pair": el / npairs * 1e3, "pairs_per_launch": B, "streams": S,
                           "sequence": ("every context alternates runs of %d noise-free pairs (3 accepted LM steps) and %d DeepFlow-like pairs (2 steps): its "
                                        "iterate predictor is wrong once per run" % (RUN, RUN)) if el_same is not None else "identical pairs",
                           "identical_pairs": None if el_same is None else {"value": rows * cols * world * npairs / el_same / 1e6, "ms_per_pair": el_same / npairs * 1e3,
                                                                             "note": "best case: every pair identical, the predictor is always right"},
                           "one_pair_at_a_time": {"value": rows * cols * world * n1 / el1 / 1e6, "ms_per_pair": el1 / n1 * 1e3},
                           "rows": rows, "cols": cols, "pixels": n, "depth_mode": int(mode),
                           "launches_per_step": (2.0 / B) if mode == 1 else 1, "extra_lm_launches": int(extra), "lm_summary": summary,
                           "max_rel_err_vs_truth": max_rel,
                           # every context of the last batches: LM finished inside the timed launch sequence and the depths match the truth
                           "verified": bool(max_rel < 1e-8 and (mode != 1 or (extra == 0 and summary is not None and summary["termination"] >= 0)))},
                "roofline": {"bound": "hbm", "kernel": ("%s (%d pairs per launch)" % (batch_kernel, B)) if mode == 1 else "depth_closed_form_kernel",
                             "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                             "traffic": _traffic("depth_batch%d" % B if mode == 1 else workload, [batch_kernel] if mode == 1 else None), "alg_bytes_per_launch": alg_bytes,
                             "avg_launch_ms": kern_ms, "median_launch_ms": kern_med,
                             "note": "launch duration = the dispatch's own start / stop timestamps (hipExtLaunchKernel events on the library's launch), last "
                                     "call of bursts of 10 batched solves on one stream, the other streams idle; profiles/: rocprofv3 of `bench.py --streams 1`"},
                # the same algorithmic bytes over the JOB's time per pair: what the HBM system delivers to the whole loop
                "roofline_job": {"bound": "hbm", "achieved": job, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": job / HBM_PEAK_GBS,
                                 "pairs_per_launch": B, "streams": S},
            })
        if rank == 0 and side_records:
            rec["cpu_baseline"] = None if (args.no_cpu_baseline or world > 1) else cpu_baseline(data, v, w, budget_s=cpu_budget)
            rec["cpu_baseline_all_cores"] = None if (args.no_cpu_baseline or world > 1 or cpu_budget < 8) else cpu_baseline_all_cores(data, v, w)
        for sv in extra_solvers:
            if sv is not None:
                sv.close()
        return rec

    # =================================================================================================
    if args.workload in ("depth", "depth_closed_form"):
        rec = run_depth(args.workload, args.steps, args.warmup)
        if rank == 0:
            line["metric"] = METRIC_DEPTH
            line.update(rec)

    # =================================================================================================
    elif args.workload == "full" and args.sequence_only:
        rec = _full_solve_sequence(rsdsfm, solver, torch, dev, np, rank, args, passes=args.steps)
        if rank == 0:
            line.update({"value": rec["value"] * world, "ms_per_step": rec["ms_per_solve_amortised"] * rec["pairs"], "scaling": "weak",
                         "config": {"workload": "BASELINE configs[4]: 32 frame pairs @1280x720 (32 data seeds) per step through rsdsfm_solve_frames_dev, one context, one host thread"},
                         "full_solve_batched": rec, "roofline": None, "cpu_baseline": None})

    elif args.workload == "full":
        full = _full_solve(rsdsfm, solver, torch, dev, np, rank, args, steps=args.steps, warmup=args.warmup, timed=timed)
        roof = _full_roofline(rsdsfm, solver, torch, dev, np, stream, args, full) if rank == 0 else None
        side = not args.no_side_records
        # sequence-throughput mode (BASELINE configs[4]): 32 pairs with 32 data seeds through ONE context of each GPU
        batched = _full_solve_sequence(rsdsfm, solver, torch, dev, np, rank, args) if side else None
        fused = depth_only = regimes = threads8 = exact_kernel = host_boundary = None
        if side and world == 1:
            host_boundary = _host_boundary(rsdsfm, solver, np, rank, args)
            threads8 = _full_solve_batched(rsdsfm, torch, dev, local_rank, rank, args, 8, per_thread=12)  # round 2's way, for comparison
            threads8["note"] = "comparison: 8 host threads x 8 contexts, each calling the single solve (what round 2 reported as full_solve_batched)"  # single-GPU side records (a multi-rank run keeps to what scales: replicas + the tiled frame)
            if args.arith == "reference":
                with rsdsfm.Solver(local_rank, stream=stream.cuda_stream, arith="fused") as sf:
                    fr = _full_solve(rsdsfm, sf, torch, dev, np, rank, args, steps=max(20, args.steps // 2), warmup=3, timed=lambda st, k2, w2: _plain_timed(torch, st, k2, w2))
                fused = {k2: fr[k2] for k2 in ("value", "unit", "ms_per_solve", "median_ms_per_solve", "num_inliers")}
                fused["note"] = "same workload on the opt-in librsdsfm_hip_fused.so (explicit fmas in the per-pixel model); not the headline"
            regimes = _full_solve_regimes(rsdsfm, solver, torch, dev, np, rank, args)
            depth_only = run_depth("depth", 40, 3, side_records=False)
            # the same workload on the iterate-by-iterate kernels (rsdsfm_set_lm_arithmetic(1): the reference's arithmetic operation for
            # operation -- what the analytic pass falls back to when a guard trips, and the headline of rounds 1-4)
            with rsdsfm.Solver(local_rank, stream=stream.cuda_stream) as sx:
                sx.set_lm_arithmetic(1)
                xr = _full_solve(rsdsfm, sx, torch, dev, np, rank, args, steps=max(20, args.steps // 2), warmup=3, timed=lambda st, k2, w2: _plain_timed(torch, st, k2, w2))
            exact_kernel = {k2: xr[k2] for k2 in ("value", "unit", "ms_per_solve", "median_ms_per_solve", "num_inliers")}
            exact_kernel["note"] = "same workload with rsdsfm_set_lm_arithmetic(1): the RANSAC's depth solves iterate by iterate (ransac_lm_kernel); identical integer outputs"
        if rank == 0:
            line.update({"value": full["value"] * world, "ms_per_step": full["ms_per_solve"], "median_ms_per_solve": full["median_ms_per_solve"], "scaling": "weak",
                         "config": {"workload": "BASELINE metric: WHOLE depth+pose solve of a synthetic 1280x720 DeepFlow-like pair (BASELINE configs[4] data: 0.3 px noise, "
                                                "10 %% outliers): flatten + alpha, RANSAC(%d trials, tol %g) = 9-point minimal solver + Ceres-LM depth solve of all pixels per "
                                                "trial + scoring, joint nonlinear refinement, sign fix + depth map, pose table; ONE C-ABI call per pair, one pair at a time, "
                                                "one pair per GPU (N ranks = N independent replicas, no data-path collective)" % (args.trials, args.tol),
                                    "pairs": "the timed steps rotate over %d different pairs (data seeds) and a new sampler seed per step" % full["data_seeds"],
                                    **{k2: full[k2] for k2 in ("rows", "cols", "trials", "tol", "n", "num_inliers", "data_seeds", "num_inliers_min_max",
                                                               "refine_iterations_min_max", "distinct_winners", "refine_summary", "w_err", "v_angle_deg")}},
                         "roofline": roof, "refine_pass": _refine_pass_record(full), "refine_restarts": full.get("refine_restarts"), "full_solve_batched": batched, "full_solve_8_threads": threads8, "full_solve_fused": fused, "full_solve_exact_kernel": exact_kernel, "host_boundary": host_boundary, "regimes": regimes, "depth_only": depth_only,
                         "cpu_baseline": None if (args.no_cpu_baseline or world > 1) else cpu_baseline_full(rsdsfm, np, rank, args.trials, args.tol)})
            # the regimes the headline does not exercise, lifted to the top level of the line: `value_selective` = the same one-call solve
            # at the selective tolerance 0.002 (M < N: compaction and the rank-indexed flow are NOT the identity), `value_sequence` =
            # BASELINE configs[4] (32 pairs / 32 data seeds through rsdsfm_solve_frames_dev), both in the line's unit
            # solves of this context that started over with the standard functions: none on the headline's pairs nor (since round 5: an exact
            # zero is a select inside the cores) on noise-free flow; the acceleration-mode regime's minimal solver (6x6 eigenvalues) meets an
            # operand outside the cores' range once or twice
            line["ransac_restarts"] = solver.ransac_restarts()
            # solves of this context whose analytic pass tripped a guard and started over iterate by iterate, and which guards (bit 7: a tie in
            # count and error sum -- the noise-free regime; the headline's DeepFlow-like pairs trip none)
            lr, lg = solver.lma_restarts()
            line["lma_restarts"] = {"count": lr, "last_guards": lg, "per_solve_headline": 0.0 if full.get("lma_restarts") is None else full["lma_restarts"] / max(1, args.steps + args.warmup)}
            line["value_selective"] = regimes["selective_tol_0.002"]["value"] if regimes else None
            line["value_sequence"] = batched["value"] * world if batched else None
            if line["cpu_baseline"] and side:  # the single-thread figure "plus an all-cores variant"
                line["cpu_baseline"]["all_cores"] = cpu_baseline_full_all_cores(rsdsfm, np, rank, args.trials, args.tol, line["cpu_baseline"])
            if line["cpu_baseline"] and side:  # the reference-STRUCTURED single-thread variant
                line["cpu_baseline"]["reference_structured"] = cpu_baseline_reference_structured(rsdsfm, np, rank, args.trials, args.tol)
        # LAST: the strong-scaling sub-record (north_star: one 3840x2160 frame in N column slabs over the native RCCL driver).  Multi-rank
        # RCCL inside the library cannot be exercised on the 1-GPU development boxes, so the section runs under a watchdog: if it
        # does not come back, rank 0 prints the line it has (tiled_full = the error) and every rank leaves.
        if side:
            line["tiled_full"] = None
            emitted = _watchdog(args.tiled_timeout, rank, line, "tiled_full")
            try:
                barrier()
                rec = _tiled_full_record(rsdsfm, solver, torch, dist, dev, np, world, rank, args, max(20, args.steps // 2), 3, timed)
                if rank == 0:
                    rec.update(_tiled_scaling_model(rec, world))
                    line["tiled_full"] = rec
            except Exception as e:  # a failure here must not take the headline with it
                line["tiled_full"] = {"error": repr(e)[:300]}
            finally:
                emitted.cancel()

    # =================================================================================================
    elif args.workload == "rectify":
        # main.cc:480-523 after the solve: 8-bit depth image, backProject, interpolateCrackyImage; one 1280x720 frame per GPU
        data = rsdsfm.synth.make_config(2, seed=0x5EED0002 + rank)
        rows, cols, K, gamma = data["rows"], data["cols"], data["K"], data["gamma"]
        npix = rows * cols
        t = data["truth"]
        rng = np.random.default_rng(7 + rank)
        img_h = rng.integers(16, 256, size=(rows, cols, 3), dtype=np.uint8)
        depth_h = np.ascontiguousarray(np.array(t["Z"]).T)  # column-major rows x cols, every pixel an inlier
        nbuf = args.nbuf
        R = torch.empty((rows, 9), dtype=torch.float64, device=dev)
        tt = torch.empty((rows, 3), dtype=torch.float64, device=dev)
        solver.pose_table_dev(t["v"] * 3.0, t["w"] * 4.0, 0.0, gamma, rows, R.data_ptr(), tt.data_ptr())
        inl_h = np.column_stack([data["q"], (np.array(t["Z"]).T.reshape(-1))])  # (x, y, z) in the flattened order
        sets = [dict(img=torch.from_numpy(img_h).to(dev), depth=torch.from_numpy(depth_h).to(dev), inl=torch.from_numpy(inl_h).to(dev),
                     gs=torch.empty((rows, cols, 3), dtype=torch.uint8, device=dev), fixed=torch.empty((rows, cols, 3), dtype=torch.uint8, device=dev),
                     c3=torch.empty((rows, cols, 3), dtype=torch.float32, device=dev), prev=torch.empty((rows, cols), dtype=torch.uint8, device=dev))
                for _ in range(nbuf)]

        def bp(s):
            solver.back_project_dev(s["img"].data_ptr(), s["depth"].data_ptr(), R.data_ptr(), tt.data_ptr(), K, rows, cols, s["gs"].data_ptr(), s["c3"].data_ptr())

        def step_separate(i):  # the three entry points one after the other: five launches
            s = sets[i % nbuf]
            solver.depth_preview_dev(s["inl"].data_ptr(), npix, K, rows, cols, s["prev"].data_ptr())
            bp(s)
            solver.interpolate_cracky_dev(s["gs"].data_ptr(), rows, cols, s["fixed"].data_ptr(), offset=1)

        def step(i):  # main.cc:480-523 in one call: two launches (rsdsfm_rectify_frame_dev)
            s = sets[i % nbuf]
            solver.rectify_frame_dev(s["inl"].data_ptr(), npix, s["img"].data_ptr(), s["depth"].data_ptr(), R.data_ptr(), tt.data_ptr(), K, rows, cols,
                                     s["prev"].data_ptr(), s["gs"].data_ptr(), s["fixed"].data_ptr(), s["c3"].data_ptr(), offset=1)

        el_sep = timed(step_separate, args.steps, args.warmup)
        ref_out = [sets[(args.steps - 1) % nbuf][k2].clone() for k2 in ("prev", "gs", "fixed", "c3")]
        el = timed(step, args.steps, args.warmup)
        same_bytes = all(torch.equal(a_, sets[(args.steps - 1) % nbuf][k2]) for a_, k2 in zip(ref_out, ("prev", "gs", "fixed", "c3")))
        kern_ms = None
        if rank == 0:
            BURST, reps = 10, 20
            ts = []
            for i in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for b in range(BURST):
                    bp(sets[(i * BURST + b) % nbuf])
                e1.record(stream)
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / BURST)
            kern_ms = float(np.mean(ts))
            alg = 26 * npix  # read 3 B image + 8 B depth, write 3 B image + 12 B world point
            achieved = alg / (kern_ms * 1e-3) / 1e9
            s = sets[(args.steps - 1) % nbuf]
            covered = float((s["gs"].view(-1, 3).sum(dim=1) != 0).double().mean().item())
            line.update({"value": npix * world * args.steps / el / 1e6, "ms_per_step": el / args.steps * 1e3, "scaling": "weak",
                         "metric": "Mpixels/sec RS->GS rectification (depth image + back projection + crack interpolation), 1280x720 frame",
                         "dtype": "u8/f64",
                         "config": {"workload": "SURVEY 8(f-1): 8-bit depth image + backProject (with float3 world points) + interpolateCrackyImage of "
                                                "a synthetic 1280x720 BGR frame, depth map and pose table resident in HBM; one frame per GPU",
                                    "api": "rsdsfm_rectify_frame_dev (one call, two launches)", "rows": rows, "cols": cols, "gs_coverage": covered,
                                    "three_entry_points_five_launches": {"value": npix * world * args.steps / el_sep / 1e6, "ms_per_step": el_sep / args.steps * 1e3},
                                    "same_bytes_as_the_separate_calls": bool(same_bytes)},
                         "roofline": {"bound": "hbm", "kernel": "back projection alone (rsdsfm_back_project_dev) = back_project_claim_kernel + back_project_write_kernel",
                                      "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                      "traffic": _traffic("rectify"), "alg_bytes_per_launch": alg, "avg_launch_ms": kern_ms},
                         "cpu_baseline": None if (args.no_cpu_baseline or world > 1) else cpu_baseline_rectify(img_h, np.array(t["Z"]), R.cpu().numpy(), tt.cpu().numpy(), K, inl_h)})

    # =================================================================================================
    elif args.workload == "true_flow":
        # Camera::calculateTrueFlow: every pixel of frame 1 against every scanline pose of frame 2; one pair per GPU
        data = rsdsfm.synth.make_config(2, seed=0x5EED0002 + rank)
        rows, cols, K, gamma = data["rows"], data["cols"], data["K"], data["gamma"]
        npix = rows * cols
        t = data["truth"]
        fx, fy, cx, cy = K
        yy, xx = np.mgrid[0:rows, 0:cols]
        Z = np.array(t["Z"])
        wpts = np.stack([(xx - cx) / fx, (yy - cy) / fy, np.ones((rows, cols))], axis=2) * Z[:, :, None]
        maps = [torch.from_numpy(np.ascontiguousarray(wpts[:, :, c2].T)).to(dev) for c2 in range(3)]
        R2 = torch.empty((rows, 9), dtype=torch.float64, device=dev)
        t2 = torch.empty((rows, 3), dtype=torch.float64, device=dev)
        solver.pose_table_dev(t["v"] * 3.0, t["w"] * 4.0, 0.0, gamma, rows, R2.data_ptr(), t2.data_ptr())
        t2 += torch.tensor([0.04, 0.02, 0.01], dtype=torch.float64, device=dev)
        flow = torch.empty((rows, cols, 2), dtype=torch.float64, device=dev)
        best = torch.empty((rows, cols), dtype=torch.int32, device=dev)

        def step(i):
            solver.true_flow_dev(maps[0].data_ptr(), maps[1].data_ptr(), maps[2].data_ptr(), rows, cols, R2.data_ptr(), t2.data_ptr(), rows, K,
                                 flow.data_ptr(), best.data_ptr())

        el = timed(step, args.steps, args.warmup)
        solver.set_true_flow_search(1)  # the exhaustive loop (the reference's own), for comparison
        el_x = timed(step, max(20, args.steps // 5), 3)
        solver.set_true_flow_search(0)
        if rank == 0:
            ms = el / args.steps * 1e3
            ms_x = el_x / max(20, args.steps // 5) * 1e3
            proj = float(npix) * rows
            alg = 44 * npix  # 24 B world point read, 16 B flow + 4 B winner written
            achieved = alg / (ms * 1e-3) / 1e9
            cpu = None
            if not (args.no_cpu_baseline or world > 1):
                sys.path.insert(0, os.path.join(ROOT, "oracle"))
                import oracle_py as O

                sub = 240  # rows of frame 1 searched on the CPU (each pixel against all 720 scanlines)
                R2h, t2h = R2.cpu().numpy(), t2.cpu().numpy()
                t0 = time.perf_counter()
                O.true_flow(wpts[:sub], R2h, t2h, *K)
                ce = time.perf_counter() - t0
                cpu = {"value": sub * cols / ce / 1e6, "unit": "Mpixels/s", "cores": 1, "kind": "port",
                       "sample": "%d of the %d image rows (x %d cols x %d scanline projections each) in %.1f s" % (sub, rows, cols, rows, ce)}
            line.update({"value": npix * world * args.steps / el / 1e6, "ms_per_step": ms, "scaling": "weak",
                         "metric": "Mpixels/sec ground-truth RS flow search, 1280x720 pair",
                         "config": {"workload": "SURVEY 8(f-2): calculateTrueFlow of a synthetic 1280x720 pair: argmin over the 720 scanline poses of "
                                                "frame 2 of |y - scanline| for every pixel; exact interval-pruned search (blocks of 32 scanlines are skipped "
                                                "when their lower bound exceeds the best value so far; same winners as the 6.6e8 exhaustive projections); "
                                                "one pair per GPU",
                                    "rows": rows, "cols": cols, "exhaustive_ms_per_step": ms_x, "exhaustive_projections_per_s": proj / (ms_x * 1e-3),
                                    "speedup_over_exhaustive": ms_x / ms, "void_pixels": int((best < 0).sum().item())},
                         "roofline": {"bound": "hbm", "kernel": "true_flow_pruned_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": achieved / HBM_PEAK_GBS, "traffic": _traffic("true_flow"), "alg_bytes_per_launch": alg, "avg_launch_ms": ms,
                                      "note": "compute-bound: one global + 2-3 block interval bounds (~75 fp64 VALU instructions each), 22 two-subtraction tests and ~40 exact projections (~30 each) per 44 B pixel; see DESIGN"},
                         "cpu_baseline": cpu})

    # =================================================================================================
    elif args.workload == "metrics":
        # meanReprojectionError + createErrorImage of a 1280x720 frame resident in HBM; one frame per GPU
        data = rsdsfm.synth.make_config(2, seed=0x5EED0002 + rank)
        rows, cols, K, gamma = data["rows"], data["cols"], data["K"], data["gamma"]
        npix = rows * cols
        t = data["truth"]
        fx, fy, cx, cy = K
        yy, xx = np.mgrid[0:rows, 0:cols]
        Z = np.array(t["Z"])
        est_h = (np.stack([(xx - cx) / fx, (yy - cy) / fy, np.ones((rows, cols))], axis=2) * Z[:, :, None] * 1.7).astype(np.float32)
        nbuf = args.nbuf
        R = torch.empty((rows, 9), dtype=torch.float64, device=dev)
        tt = torch.empty((rows, 3), dtype=torch.float64, device=dev)
        solver.pose_table_dev(t["v"] * 3.0, t["w"] * 4.0, 0.0, gamma, rows, R.data_ptr(), tt.data_ptr())
        sets = [dict(est=torch.from_numpy(est_h).to(dev), gt=torch.from_numpy(np.ascontiguousarray(Z.T)).to(dev),
                     ed=torch.from_numpy(np.ascontiguousarray(Z.T)).to(dev), img=torch.empty((rows, cols), dtype=torch.uint8, device=dev)) for _ in range(nbuf)]
        res = {}

        def step(i):
            s_ = sets[i % nbuf]
            res["st"] = solver.reprojection_error_dev(s_["est"].data_ptr(), s_["gt"].data_ptr(), s_["ed"].data_ptr(), R.data_ptr(), tt.data_ptr(), K, rows, cols,
                                                      10.0, s_["img"].data_ptr())

        el = timed(step, args.steps, args.warmup)
        if rank == 0:
            ms = el / args.steps * 1e3
            alg = (2 * 28 + 1) * npix  # two streaming passes of 12 B point + 2 x 8 B depth, 1 B image written
            achieved = alg / (ms * 1e-3) / 1e9
            cpu = None
            if not (args.no_cpu_baseline or world > 1):
                sys.path.insert(0, os.path.join(ROOT, "oracle"))
                import oracle_py as O

                Rh, th = R.cpu().numpy(), tt.cpu().numpy()
                t0 = time.perf_counter()
                reps = 0
                while time.perf_counter() - t0 < 6.0:
                    O.reprojection_error(est_h, Z, Z, Rh, th, *K)
                    reps += 1
                ce = time.perf_counter() - t0
                cpu = {"value": npix * reps / ce / 1e6, "unit": "Mpixels/s", "cores": 1, "kind": "port", "sample": "%d whole 1280x720 frames in %.1f s" % (reps, ce)}
            line.update({"value": npix * world * args.steps / el / 1e6, "ms_per_step": ms, "scaling": "weak",
                         "metric": "Mpixels/sec reprojection-error metric (mean error + error image), 1280x720 frame", "dtype": "f32/f64",
                         "config": {"workload": "SURVEY 8(f-4): Camera::meanReprojectionError + createErrorImage of a synthetic 1280x720 frame (2 launches per call, "
                                                "the partial sums land in host-mapped memory; the call returns the statistics: one host synchronisation per frame); one frame per GPU", "rows": rows, "cols": cols, "stats": res["st"]},
                         "roofline": {"bound": "hbm", "kernel": "reproj_scale_kernel + reproj_error_kernel (whole synchronous call: ~23 us of kernels + the host round trip)",
                                      "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                                      "alg_bytes_per_launch": alg, "avg_launch_ms": ms},
                         "cpu_baseline": cpu})

    # =================================================================================================
    elif args.workload == "tiled_full":
        rec = _tiled_full_record(rsdsfm, solver, torch, dist, dev, np, world, rank, args, args.steps, args.warmup, timed)
        if rank == 0:
            rec.update(_tiled_scaling_model(rec, world))
            line.update({"value": rec["value"], "ms_per_step": rec["ms_per_solve"], "scaling": "strong", "metric": rec.pop("metric"),
                         "config": rec.pop("config"), "roofline": None, "cpu_baseline": None, "tiled_full": rec})

    # =================================================================================================
    else:  # tiled
        # BASELINE configs[3] as literally stated: the dense DEPTH solve of a 3840x2160 frame, the flattened point list sharded over the
        # ranks, ONE C-ABI call per rank and step (rsdsfm_estimate_inverse_depths_tiled_dev: LM launches, the all-gather of the per-rank
        # sum rows, the decision and the all-gather of the depth shards, all on the context's stream; one host synchronisation).
        # --tiled-driver python = the round-1 Python driver over the stage entry points (dist.TiledDepthSolve), for comparison.
        data = rsdsfm.synth.make_config(4, seed=0x5EED0004)  # every rank generates the same 3840x2160 frame
        n = len(data["q"])
        t = data["truth"]
        v, w, k = t["v"] / np.linalg.norm(t["v"]), t["w"], 0.0
        i0, cnt, per = rsdsfm.tiled_shard_bounds(n, world, rank)
        tt_ = lambda a: torch.from_numpy(np.ascontiguousarray(a[i0:i0 + cnt])).to(dev)
        q_s, u_s, a_s, ak_s = tt_(data["q"]), tt_(data["u"]), tt_(data["alpha"]), tt_(data["alpha_k"])
        res = {}
        native = args.tiled_driver == "native"
        transport = "python driver + torch.distributed"
        if native:
            transport = _dist_setup(solver, rsdsfm, torch, dist, world, rank, dev)
            full = torch.empty(max(n, 2), dtype=torch.float64, device=dev)

            def step(i):
                res["sm"], res["info"] = solver.estimate_inverse_depths_tiled_dev(q_s.data_ptr(), u_s.data_ptr(), n, v, w, k, a_s.data_ptr(), ak_s.data_ptr(),
                                                                                  full.data_ptr(), mode=rsdsfm.DEPTH_CERES_LM)
                res["rho"] = full[:n]
        else:
            stage = rsdsfm.dist.HipDepthStage(solver, q_s, u_s, a_s, ak_s, v, w, k, torch)
            drv = rsdsfm.dist.TiledDepthSolve([stage], n, per, torch, dist if world > 1 else None)

            def step(i):
                res["rho"], res["sm"] = drv.solve(rsdsfm.DEPTH_CERES_LM)

        el = timed(step, args.steps, args.warmup)
        if rank == 0:
            line.update({"value": data["rows"] * data["cols"] * args.steps / el / 1e6, "ms_per_step": el / args.steps * 1e3, "scaling": "strong",
                         "metric": "Mpixels/sec RS depth solve, 3840x2160 frame row-tiled over the ranks",
                         "config": {"workload": "BASELINE configs[3]: synthetic 3840x2160 frame, point list sharded over %d rank(s), dense depth "
                                                "solve (Ceres-LM emulation), all-gather of the LM sum rows + ONE all-gather of the depth map" % world,
                                    "driver": "native C++ (rsdsfm_estimate_inverse_depths_tiled_dev)" if native else "python (dist.TiledDepthSolve)",
                                    "transport": transport, "rows": data["rows"], "cols": data["cols"], "pixels": n, "lm_summary": res["sm"],
                                    "info": res.get("info"), "gathered": int(res["rho"].shape[0])},
                         "roofline": None, "cpu_baseline": None})
        if native:
            solver.dist_finalize()

    if rank == 0:
        try:  # RCCL prints its version banner through C stdio, which is block-buffered on a pipe: push it out BEFORE the JSON line
            import ctypes

            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        _emit_line(line)
    solver.close()
    if world > 1:
        dist.destroy_process_group()


_EMIT_LOCK = threading.Lock()
_EMITTED = [False]
WATCHDOG_EXIT_CODE = 3


def _emit_line(line):
    """prints THE JSON line of this process exactly once (the main thread at the end of run(), or a watchdog that fired first)"""
    with _EMIT_LOCK:
        if _EMITTED[0]:
            return False
        _EMITTED[0] = True
        print(json.dumps(line), flush=True)
        return True


def _watchdog(seconds, rank, line, key):
    """threading.Timer that ends the process if a section hangs: rank 0 first prints the JSON line it has, with line[key] = the error,
    then EVERY rank exits with WATCHDOG_EXIT_CODE -- a hung section is a failed run for the launcher and the harness, while the line
    (whose headline was measured before the guarded section) is still relayed: launch_ranks prints rank 0's line whatever the exit
    codes.  The timer thread works on a COPY of the line taken under the emit lock, so it never prints a dict the main thread is
    updating, and exactly one line leaves the process."""
    def fire():
        if rank == 0:
            with _EMIT_LOCK:
                if not _EMITTED[0]:
                    _EMITTED[0] = True
                    snap = dict(line)
                    snap[key] = {"error": "no result within %.0f s (watchdog); the process exits with code %d" % (seconds, WATCHDOG_EXIT_CODE)}
                    print(json.dumps(snap, default=repr), flush=True)
        os._exit(WATCHDOG_EXIT_CODE)

    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


def _tiled_scaling_model(rec, world):
    """DESIGN section 8's model of the column-tiled 3840x2160 solve on N ranks, printed next to the measurement so that a SCALE
    run can be judged against it: per-slab kernels shrink with 1/N, the minimal solver and the single-workgroup decide stages are
    replicated, every collective costs a latency (small rows) or bytes / link bandwidth (the depth-map all-gather)"""
    n1 = None
    for committed in ("r05_bench_tiled_full.json", "r05_bench_full.json", "r04_bench_full.json"):  # the committed 1-rank measurement
        try:
            n1 = json.load(open(os.path.join(ROOT, "profiles", committed)))["tiled_full"]["ms_per_solve"]
            break
        except Exception:
            pass
    m = TILED_MODEL
    coll = rec.get("collectives") or m["collectives"]
    pred = m["per_pixel_ms"] / world + m["replicated_ms"] + (0.0 if world == 1 else coll * m["collective_latency_ms"] + m["depth_gather_ms"](world))
    out = {"model": {"predicted_ms_per_solve": pred, "per_pixel_ms_at_n1": m["per_pixel_ms"], "replicated_ms": m["replicated_ms"],
                     "collective_latency_ms": m["collective_latency_ms"], "collectives": coll, "depth_gather_ms": 0.0 if world == 1 else m["depth_gather_ms"](world),
                     "formula": "per_pixel / N + replicated + collectives x latency + depth all-gather (DESIGN section 7)"},
           "n1_reference_ms_per_solve": n1}
    if n1:
        out["speedup_vs_committed_n1"] = n1 / rec["ms_per_solve"]
    return out


def _mmm(xs):
    xs = sorted(x for x in xs if x is not None)
    return [xs[0], xs[len(xs) // 2], xs[-1]] if xs else None


def _tiled_full_record(rsdsfm, solver, torch, dist, dev, np, world, rank, args, steps, warmup, timed):
    """BASELINE configs[3] / north_star "large frames tile across the GPUs": ONE 3840x2160 DeepFlow-like frame split into column slabs
    over the ranks, the WHOLE solve (flatten, RANSAC, refinement, sign fix + depth map) by ONE C-ABI call per rank
    (rsdsfm_solve_frame_tiled_dev: the C++ driver issues the stage kernels and the RCCL collectives on the context's stream); every
    rank holds only its slab.  Strong scaling: the frame is fixed, the slabs shrink with N.  --tiled-driver python = the Python driver
    over the rsdsfm_tile_* stage entry points (dist.TiledFrameSolve), for comparison."""
    data = rsdsfm.synth.make_config(4, seed=0x5EED0004)  # every rank generates the same frame and keeps its slab
    rows, cols = data["rows"], data["cols"]
    c0, sc, per = rsdsfm.tiled_slab_bounds(cols, world, rank)
    slab = torch.from_numpy(np.ascontiguousarray(data["flow_img"][:, c0:c0 + sc, :])).to(dev)
    depth_map = torch.empty(rows * cols, dtype=torch.float64, device=dev)
    res = {}
    native = args.tiled_driver == "native"
    transport = "python driver + torch.distributed"
    if native:
        transport = _dist_setup(solver, rsdsfm, torch, dist, world, rank, dev)

        def step(i):
            res["r"] = solver.solve_frame_tiled_dev(slab.data_ptr(), rows, cols, data["K"], data["gamma"], depth_map.data_ptr(),
                                                    trials=args.trials, tol=args.tol, seed=1 + i)
    else:
        def step(i):
            shard = rsdsfm.dist.HipFrameShard(solver, slab, c0, data["K"], data["gamma"], torch)  # flatten is part of the solve
            drv = rsdsfm.dist.TiledFrameSolve([shard], rows, cols, per, torch, dist if world > 1 else None)
            res["r"] = drv.solve(trials=args.trials, tol=args.tol, seed=1 + i)

    per_step = []

    per_info = []

    def tstep(i):
        t0 = time.perf_counter()
        step(i)
        per_step.append(time.perf_counter() - t0)
        inf = (res["r"].get("info") or {})
        per_info.append((inf.get("collectives"), inf.get("host_syncs"), inf.get("path_flags"), res["r"]["refine_summary"]["num_iterations"]))

    el = timed(tstep, steps, warmup)
    rec = None
    if rank == 0:
        r = res["r"]
        t = data["truth"]
        ts = sorted(per_step[-steps:])
        info = r.get("info") or {}
        rec = {"metric": "Mpixels/sec RS whole solve, 3840x2160 frame column-tiled over the ranks", "scaling": "strong", "n_ranks": world,
               "value": rows * cols * steps / el / 1e6, "unit": "Mpixels/s", "ms_per_solve": el / steps * 1e3, "median_ms_per_solve": ts[len(ts) // 2] * 1e3,
               "steps": steps, "rccl_ranks": info.get("nranks"), "collectives": info.get("collectives"), "host_syncs": info.get("host_syncs"),
               # over the timed solves (the sampler seed changes per solve: the refinement's length does too, and a chunk that follows the
               # previous solve's length then costs a poll and two collectives per extra iteration): [min, median, max]
               "collectives_min_med_max": _mmm([x[0] for x in per_info[-steps:]]), "host_syncs_min_med_max": _mmm([x[1] for x in per_info[-steps:]]),
               "refine_iterations_min_med_max": _mmm([x[3] for x in per_info[-steps:]]),
               "solves_ahead_on_dense_counts": sum(1 for x in per_info[-steps:] if x[2] is not None and (x[2] & 1) and not (x[2] & 2)),
               "config": {"workload": "BASELINE configs[3]: synthetic 3840x2160 DeepFlow-like frame, column slabs over %d rank(s): flatten + RANSAC(%d, tol %g) "
                                      "+ refinement + depth map; all-gathers of the stage sum rows + ONE all-gather of the depth slabs"
                                      % (world, args.trials, args.tol),
                          "driver": "native C++ (rsdsfm_solve_frame_tiled_dev)" if native else "python (dist.TiledFrameSolve)", "transport": transport,
                          "rows": rows, "cols": cols, "n": r["n"], "num_inliers": r["num_inliers"], "info": r.get("info"),
                          "flow_index_mode": r.get("flow_index_mode"),
                          "refine_summary": r["refine_summary"], "w_err": float(np.linalg.norm(r["w"] - t["w"]))}}
    if native:
        solver.dist_finalize()
    return rec


def _dist_setup(solver, rsdsfm, torch, dist, world, rank, dev):
    """give the context its communicator for the native tiled solves; returns a description of the transport"""
    backend = dist.get_backend() if world > 1 else "nccl"
    if backend == "nccl":  # RCCL inside the library: rank 0's unique id travels through torch.distributed (1 rank: a 1-rank communicator)
        ident = torch.zeros(rsdsfm.DIST_ID_BYTES, dtype=torch.uint8, device=dev)
        if rank == 0:
            ident = torch.frombuffer(bytearray(rsdsfm.dist_unique_id()), dtype=torch.uint8).to(dev)
        if world > 1:
            dist.broadcast(ident, 0)
        solver.dist_init(world, rank, bytes(ident.cpu().numpy().tobytes()))
        return "RCCL (ncclAllGather / ncclAllReduce on the context's stream), %d-rank communicator" % world
    # smoke tests only (several ranks sharing one GPU over gloo)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from transports import GlooTransport

    solver.dist_set_transport(world, rank, *GlooTransport(dist, torch).callbacks())
    return "caller-provided collectives over torch.distributed/%s (smoke test)" % backend


def cpu_baseline_rectify(img, depth, R, t, K, inl, budget_s=8.0):
    """the oracle's depth image + back projection + crack interpolation (scalar C, 1 thread) on the same frame"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O

    O.lib()
    rows, cols = img.shape[:2]
    t0 = time.perf_counter()
    reps = 0
    while True:
        O.depth_preview(inl, *K, rows, cols)
        gs, _ = O.back_project(img, depth, R, t, *K)
        O.interpolate_cracky(gs, 1)
        reps += 1
        if time.perf_counter() - t0 > budget_s:
            break
    el = time.perf_counter() - t0
    return {"value": rows * cols * reps / el / 1e6, "unit": "Mpixels/s", "cores": 1, "kind": "port",
            "sample": "%d whole 1280x720 frames (depth image + back projection + interpolation) in %.1f s" % (reps, el)}


def _traffic(workload, kernels=None):
    """HBM bytes per launch of the workload's dominant kernel(s) from the rocprofv3 PMC passes committed under profiles/:
    counters.json (round 2: per-kernel FETCH_SIZE / WRITE_SIZE means in KB, FETCH_SIZE x2 on gfx950 per MI355X_MICROARCH.md) when it
    has the kernels, else the round-1 record in traffic.json; None if absent."""
    kernels = {"rectify": ["back_project_claim_kernel", "back_project_write_kernel"], "true_flow": ["true_flow_pruned_kernel"],
               "depth_batch8": ["depth_lma_batch_kernel<2>"], "depth_closed_form": ["depth_closed_form_kernel"]}.get(workload) if kernels is None else kernels
    if kernels:
        ctr = [_counters(k2) for k2 in kernels]
        if any(c2 and c2.get("stale") for c2 in ctr):
            return None  # the kernel sources changed since the counters were collected
        if all(c2 and "FETCH_SIZE" in c2 and "WRITE_SIZE" in c2 for c2 in ctr):
            return sum((2.0 * c2["FETCH_SIZE"] + c2["WRITE_SIZE"]) * 1024.0 for c2 in ctr)
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tf):
        try:
            return json.load(open(tf)).get(workload)
        except Exception:
            return None
    return None


def _plain_timed(torch, step, steps, warmup):
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def _full_solve(rsdsfm, solver, torch, dev, np, rank, args, steps, warmup, timed, data_seeds=4):
    """whole solve on 1280x720 DeepFlow-like pairs (0.3 px noise, 10 % outliers), flow images resident in HBM.  The timed steps rotate
    over `data_seeds` DIFFERENT pairs (one noise / outlier realisation each: other hypotheses, another winner, another refinement
    length per step) and the sampler seed changes every step as well; the per-pair results of the timed steps are kept for the record."""
    seeds = [0x5EED0005 + rank + 7919 * j for j in range(data_seeds)]
    flows, meta = rsdsfm.synth.make_flow_sequence(5, seeds)
    rows, cols, K, gamma = meta["rows"], meta["cols"], meta["K"], meta["gamma"]
    imgs = [torch.from_numpy(f).to(dev) for f in flows]
    depth_map = torch.empty((cols, rows), dtype=torch.float64, device=dev)  # column-major rows x cols
    R, tt = torch.empty((rows, 9), dtype=torch.float64, device=dev), torch.empty((rows, 3), dtype=torch.float64, device=dev)
    per_step, seen = [], []

    # ONE C-ABI call per frame pair (rsdsfm_solve_frame_dev) with pre-marshalled arguments; it returns after its last result has
    # reached the host
    calls = [solver.prepared_frame_solve(im.data_ptr(), rows, cols, K, gamma, depth_map.data_ptr(), R.data_ptr(), tt.data_ptr(),
                                         trials=args.trials, tol=args.tol) for im in imgs]
    clock = time.perf_counter
    nd = len(calls)

    def step(i):
        t0 = clock()
        r_ = calls[i % nd](1 + i)
        per_step.append(clock() - t0)
        seen.append((int(r_.num_inliers), int(r_.refine_summary.num_iterations), int(r_.best_trial)))  # (after the clock: not in the step's own time)

    lma0 = solver.lma_restarts()[0]
    rf0 = solver.refine_restarts()
    el = timed(step, steps, warmup)
    lma_restarts = solver.lma_restarts()[0] - lma0  # guard restarts of the analytic pass during the warm-up + timed steps
    rf1 = solver.refine_restarts()
    # the joint refinement's default path (radius-factorised Schur sums): solves that ran on it, those a guard sent back to the iterate-by-iterate
    # kernels (rate = restarts / runs), reduced systems solved again from kept sums (rejected / invalid steps: no pass of their own)
    refine_restarts = {"runs": rf1["runs"] - rf0["runs"], "restarts": rf1["restarts"] - rf0["restarts"], "resolves": rf1["resolves"] - rf0["resolves"],
                       "rate": (rf1["restarts"] - rf0["restarts"]) / max(1, rf1["runs"] - rf0["runs"]), "last_guard": rf1["last_guard"]}
    # the last timed solve once more through the dict-building wrapper (same seed: same result), for the record
    r = solver.solve_frame_dev(imgs[(steps - 1) % nd].data_ptr(), rows, cols, K, gamma, depth_map.data_ptr(), R.data_ptr(),
                               tt.data_ptr(), trials=args.trials, tol=args.tol, seed=steps)
    t = meta["truth"]
    vt = t["v"] / np.linalg.norm(t["v"])
    vv = r["v"] / np.linalg.norm(r["v"])
    ts = sorted(per_step[-steps:])
    seen = seen[-steps:]
    return {"value": rows * cols * steps / el / 1e6, "unit": "Mpixels/s", "ms_per_solve": el / steps * 1e3,
            "median_ms_per_solve": ts[len(ts) // 2] * 1e3, "min_ms_per_solve": ts[0] * 1e3,
            "rows": rows, "cols": cols, "trials": args.trials, "tol": args.tol, "n": r["n"], "num_inliers": r["num_inliers"],
            "lma_restarts": lma_restarts, "refine_restarts": refine_restarts, "data_seeds": nd, "num_inliers_min_max": [min(x[0] for x in seen), max(x[0] for x in seen)],
            "refine_iterations_min_max": [min(x[1] for x in seen), max(x[1] for x in seen)], "distinct_winners": len({x[2] for x in seen}),
            "refine_summary": r["refine_summary"], "K": K, "gamma": gamma, "_img": imgs[0],
            "w_err": float(np.linalg.norm(r["w"] - t["w"])), "v_angle_deg": float(np.degrees(np.arccos(min(1.0, abs(float(vv @ vt)))))),
            "stages": "flatten+alpha, minimal9 x %d, RANSAC LM sums/decide/score/pick/compaction, refinement, depth map, pose table" % args.trials}


def _host_boundary(rsdsfm, solver, np, rank, args, reps=3):
    """The reference-shaped HOST-pointer boundary (rsdsfm_ransac + rsdsfm_refine: caller-owned host arrays in, host arrays out, synchronous --
    what minimal::ransac / nonLinearRefinement's own signatures give a caller, main.cc:447-457) on one 1280x720 DeepFlow-like pair: the 44 MB
    of inputs and the outputs cross PCIe inside the timed region (pageable numpy arrays) and the ctypes wrapper's marshalling is in it too.
    Reported beside `value`, never as it: the metric's inputs are resident in HBM."""
    d = rsdsfm.synth.make_config(5, seed=0x5EED0005 + rank)
    q, u, a, ak = d["q"], d["u"], d["alpha"], d["alpha_k"]
    n = len(q)

    def run(reuse, reps):
        ts, r, out = [], None, None
        # (reuse: caller-owned output arrays kept from pair to pair, and only what the reference's RansacValues holds -- minimal.h:57-76: inliers,
        # alpha, alpha_k -- plus the inlier indices; the dense inverse depths, the mask and the per-trial diagnostics are not asked for)
        outputs, ref_out = ({"only": ("inliers", "alpha", "alpha_k", "inlier_idx")}, np.empty((n, 3))) if reuse else (None, None)
        for i in range(reps + 2):
            t0 = time.perf_counter()
            r = solver.ransac(q, u, a, ak, False, args.trials, args.tol, seed=11 + i, outputs=outputs)
            m = r["num_inliers"]
            out = solver.non_linear_refinement(u, r["inliers"], r["alpha"], r["alpha_k"], r["v"], r["w"], r["k"], False, tag=r["tag"] if reuse else 0,
                                               out=ref_out[:m] if reuse else None)
            ts.append(time.perf_counter() - t0)
        ts = sorted(ts[2:])
        return ts[len(ts) // 2], r, out

    hits0 = solver.refine_cache_hits()
    med, r, out = run(True, reps + 2)
    hits = solver.refine_cache_hits() - hits0
    med_fresh, _, _ = run(False, reps)
    m = int(r["num_inliers"])
    # bytes that cross PCIe per pair on the streaming path: q, u, alpha, alpha_k in; inliers, indices, alpha, alpha_k out; the refined inliers out (the refinement starts from the RANSAC's device-resident outputs and the flow it was given: no second upload)
    h2d, d2h = 48 * n, (24 + 8 + 8 + 8) * m + 24 * m
    return {"value": d["rows"] * d["cols"] / med / 1e6, "unit": "Mpixels/s", "ms_per_pair": med * 1e3, "pairs": reps + 2, "num_inliers": m,
            "refine_iterations": int(out["summary"]["num_iterations"]), "pcie_bytes_per_pair": {"h2d": h2d, "d2h": d2h},
            "pcie_gbs_achieved": (h2d + d2h) / med / 1e9, "refine_started_from_resident_ransac_outputs": int(hits),
            "fresh_arrays_every_call": {"value": d["rows"] * d["cols"] / med_fresh / 1e6, "ms_per_pair": med_fresh * 1e3,
                                        "note": "the same two calls with fresh numpy output arrays per call and no tag (everything uploaded twice, page faults of 90 MB of new arrays inside the calls)"},
            "note": "PCIe-inclusive: rsdsfm_ransac + rsdsfm_refine_from_ransac on host arrays the caller owns and reuses from pair to pair (the inputs and "
                    "what the reference's RansacValues holds -- inliers, alpha, alpha_k, + indices -- cross PCIe inside the calls -- uploads through csrc/host_xfer.hip's pinned ring, downloads on the runtime's pageable path; Python marshalling "
                    "included; pcie_gbs_achieved = those bytes / the whole time, compute included); not the metric"}


def _full_solve_regimes(rsdsfm, solver, torch, dev, np, rank, args, solves=24):
    """The headline exercises ONE regime (tol 0.05 keeps every pixel an inlier).  The same one-call solve, driver-timed, on the other
    regimes the reference's own constants and modes give: each = `solves` whole solves after 3 warm-ups on the bench's context, one at
    a time, sampler seed changing per solve; median of the per-solve host times (the call returns after its last result reached the host)."""
    out = {}
    d5 = rsdsfm.synth.make_config(5, seed=0x5EED0005 + rank)
    d2 = rsdsfm.synth.make_config(2, seed=0x5EED0002 + rank)
    d3 = rsdsfm.synth.make_config(3, seed=0x5EED0003 + rank)
    cases = [("trials_5", d5, dict(trials=5, tol=args.tol), "main.cc:304's default trial count, T = 5"),
             ("selective_tol_0.002", d5, dict(trials=args.trials, tol=0.002), "selective tolerance 0.002: the +-30 px outliers are rejected (M < N), compaction and the rank-indexed flow of main.cc:457 are not the identity"),
             ("noise_free", d2, dict(trials=args.trials, tol=args.tol), "noise-free model flow (ground-truth flow of a synthetic example): every hypothesis takes three accepted LM steps"),
             ("acceleration_mode", d5, dict(trials=args.trials, tol=args.tol, use_acceleration_mode=True), "use_acceleration_mode (main.cc:306): k estimated by the minimal solver (6x6 eigenvalues) and refined (7x7 Schur complement)"),
             ("configs2_1920x1080", d3, dict(trials=args.trials, tol=args.tol), "BASELINE configs[2]: 1920x1080 DeepFlow-like pair, full RANSAC + depth + refinement")]
    for name, d, kw, note in cases:
        rows, cols = d["rows"], d["cols"]
        img = torch.from_numpy(d["flow_img"]).to(dev)
        dm = torch.empty((cols, rows), dtype=torch.float64, device=dev)
        R, tt = torch.empty((rows, 9), dtype=torch.float64, device=dev), torch.empty((rows, 3), dtype=torch.float64, device=dev)
        call = solver.prepared_frame_solve(img.data_ptr(), rows, cols, d["K"], d["gamma"], dm.data_ptr(), R.data_ptr(), tt.data_ptr(), **kw)
        ts = []
        for i in range(3 + solves):
            t0 = time.perf_counter()
            r = call(1 + i)
            ts.append(time.perf_counter() - t0)
        ts = sorted(ts[3:])
        med = ts[len(ts) // 2]
        out[name] = {"median_ms_per_solve": med * 1e3, "min_ms_per_solve": ts[0] * 1e3, "value": rows * cols / med / 1e6, "unit": "Mpixels/s", "solves": solves,
                     "rows": rows, "cols": cols, "trials": kw["trials"], "tol": kw["tol"], "n": int(r.n_points), "num_inliers": int(r.num_inliers),
                     "refine_iterations": int(r.refine_summary.num_iterations), "note": note}
        del img, dm
    return out


def _counters(kernel):
    """per-launch PMC counter means of `kernel` from profiles/counters.json (rocprofv3 --pmc passes of `bench.py`, see
    profiles/collect_r03.sh); None when absent.  The file is stamped with the hashes of the kernel sources it was collected from
    (profiles/source_hash.py): when the files this kernel is built from have changed since, {"stale": [files]} is returned instead of
    counts that may no longer describe the kernel."""
    f = os.path.join(ROOT, "profiles", "counters.json")
    try:
        data = json.load(open(f))
    except Exception:
        return None
    ctr = data.get(kernel)
    if ctr is None:
        return None
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    import source_hash

    stale = source_hash.stale_files(kernel.split(":")[0], (data.get("_meta") or {}).get("sources"))
    return {"stale": stale} if stale else ctr


def _refine_pass_record(full):
    """The joint refinement's pass (refine_rf_pass_kernel<6, false, false>: one per LM iteration, csrc/refine_rf_kernels.hip) against the two
    rooflines that could bound it: fp64 lane-instructions (SQ_INSTS_VALU_*_F64 x 64) and HBM bytes (64 B per inlier and iteration algorithmic,
    SURVEY 8 d; FETCH_SIZE x 2 + WRITE_SIZE measured) of one launch from profiles/counters.json, over the launch duration the same collection pass
    stored there (`trace_avg_us`: rocprofv3 --kernel-trace of this command, the launches that ran their loop).  Over the WHOLE launch neither bounds
    it: a third of a launch is the replicated single-workgroup stage in every workgroup's prologue and the row reduction, which move no data.  Its
    LOOP phase (11 of the 20.5 us) is bound by its memory accesses: the same loop without its arithmetic takes 10 us (profiles/
    r06_refine_phases_loads_only.txt, tools/build_rfproxy.sh), 56 B per inlier at ~5 TB/s (DESIGN.md sections 4, 8, 11)."""
    kname = "refine_rf_pass_kernel<6, false, false>"
    ctr = _counters(kname)
    if not ctr:
        return {"kernel": kname, "error": "profiles/counters.json has no entry for it"}
    if ctr.get("stale"):
        return {"kernel": kname, "counters_stale": True, "counters_stale_files": ctr["stale"]}
    us = ctr.get("trace_avg_us")
    # (counters.json holds means over ALL launches of the kernel; 3 of the 8 passes enqueued per solve find the solve over and leave at once, counting
    # next to nothing: per launch that ran its loop the means are divided by the fraction of such launches in the same collection's trace)
    frac_full = ctr.get("trace_full_launch_fraction") or 1.0
    insts = sum(ctr.get(k2, 0.0) for k2 in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64")) * 64.0 / frac_full
    m = int(full["num_inliers"])
    alg = 64.0 * m
    traffic = (2.0 * ctr["FETCH_SIZE"] + ctr["WRITE_SIZE"]) * 1024.0 / frac_full if ("FETCH_SIZE" in ctr and "WRITE_SIZE" in ctr) else None
    rec = {"kernel": kname, "avg_launch_us": us, "fp64_lane_instructions_per_launch": insts or None, "alg_bytes_per_launch": alg, "traffic": traffic, "inliers": m,
           "launches_sampled": ctr.get("launches_sampled"), "full_launch_fraction": frac_full, "counters_stale": False,
           "note": "rooflines over the whole launch; the loop phase alone (about 11 of the 20.5 us; the rest is the replicated stage and the row reduction) is bound "
                   "by its memory accesses -- the loop without its arithmetic takes 10 us: profiles/r06_refine_phases_loads_only.txt"}
    if us:
        rec["fp64"] = {"bound": "fp64-valu", "achieved": insts / (us * 1e-6) / 1e12 if insts else None, "peak": FP64_VALU_PEAK / 1e12, "unit": "T fp64 lane-instructions/s",
                       "frac": insts / (us * 1e-6) / FP64_VALU_PEAK if insts else None}
        rec["hbm"] = {"bound": "hbm", "achieved": alg / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}
    return rec


def _full_roofline(rsdsfm, solver, torch, dev, np, stream, args, full):
    """Roofline record of the whole solve's dominant kernel, ransac_lma_kernel<2, true> (the T depth solves of the RANSAC on the analytic LM
    trajectory: ~26 % of the solve, level with the minimal solver's SVD chain and the refinement's passes), measured LIVE and in situ: the library brackets that launch with HIP events on the stream it runs
    on (rsdsfm_set_profiling) inside 27 ordinary whole solves after 3 warm-ups.  Its bound is
    fp64 VALU issue, not HBM: `achieved` = fp64 lane-instructions of one launch (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 x 64 lanes,
    rocprofv3 PMC pass committed as profiles/counters.json) / the measured duration; `peak` = 39.3e12 / s.  "hbm" = SURVEY 8(d):
    (57 N + 64 M iterations) bytes / the solve's time / 8 TB/s."""
    rows, cols, T = full["rows"], full["cols"], args.trials
    img = full.pop("_img")
    depth_map = torch.empty((cols, rows), dtype=torch.float64, device=dev)
    n = full["n"]
    # in situ: the library brackets the kernel with two HIP events on the context's stream inside ordinary solves
    # (rsdsfm_set_profiling), so the launch has the same neighbours and clocks as in the timed loop
    solver.set_profiling(True)
    ts, mhz = [], []
    for i in range(30):
        solver.solve_frame_dev(img.data_ptr(), rows, cols, full["K"], full["gamma"], depth_map.data_ptr(), trials=T, tol=args.tol, seed=1 + i)
        ts.append(solver.profile_last_ms("ransac_lm_round0"))
        mhz.append(solver.profile_last_ms("ransac_lm_round0_clock_mhz"))
    solver.set_profiling(False)
    ts = sorted(ts[3:])
    kern_ms = float(np.mean(ts))
    # the shader clock the kernel actually ran at (one workgroup in the middle of each bracketed launch stamps s_memtime / s_memrealtime):
    # the peak is priced at the nominal 2.4 GHz, the chip runs this kernel below it
    clock_mhz = float(np.mean(mhz[3:]))
    # round 0: three speculated iterations, the score of the two-step iterate fused (DeepFlow-like data); last template argument: sqrt and
    # the reciprocal through their in-range cores (reference-arithmetic library only)
    # the pixel pass of the RANSAC's depth solves on the analytic LM trajectory (ransac_lma_kernels.hip), the scores of two iterates fused
    kname = "ransac_lma_kernel<2, true>"
    ctr = _counters(kname + (":fused" if args.arith == "fused" else ""))
    insts = achieved = frac = traffic = frac_all = issue_floor_ms = None
    stale = ctr.get("stale") if ctr else ["profiles/counters.json: no entry for " + kname]
    if ctr and not stale:
        insts = 64.0 * sum(ctr.get(k2, 0.0) for k2 in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
        achieved = insts / (kern_ms * 1e-3)
        frac = achieved / FP64_VALU_PEAK
        if ctr.get("SQ_INSTS_VALU"):  # every VALU instruction (incl. v_div_scale / fmas / fixup, compares, selects) at the fp64 issue rate
            frac_all = 64.0 * ctr["SQ_INSTS_VALU"] / (kern_ms * 1e-3) / FP64_VALU_PEAK
            # the kernel's own issue floor at the clock it ran at: measured issue costs per wave-instruction and SIMD (tools/valu_rates.hip,
            # profiles/r05_valu_rates.txt): fp64 add / mul / fma 4 cycles, v_rcp_f64 / v_rsq_f64 16, everything else (compares, selects, integer) priced
            # at the 2 cycles of a 32-bit instruction -- a lower bound of the floor; 1024 SIMDs
            f64 = sum(ctr.get(k2, 0.0) for k2 in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64"))
            tr = ctr.get("SQ_INSTS_VALU_TRANS_F64", 0.0)
            cyc = 4.0 * f64 + 16.0 * tr + 2.0 * max(ctr["SQ_INSTS_VALU"] - f64 - tr, 0.0)
            issue_floor_ms = cyc / 1024.0 / (clock_mhz * 1e3)
        if "FETCH_SIZE" in ctr and "WRITE_SIZE" in ctr:  # KB; FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section)
            traffic = (2.0 * ctr["FETCH_SIZE"] + ctr["WRITE_SIZE"]) * 1024.0
    iters = full["refine_summary"]["num_iterations"]
    hbm_bytes = 57.0 * full["n"] + 64.0 * full["num_inliers"] * iters
    hbm_gbs = hbm_bytes / (full["median_ms_per_solve"] * 1e-3) / 1e9
    return {"bound": "fp64-valu", "kernel": kname, "achieved": None if achieved is None else achieved / 1e12, "peak": FP64_VALU_PEAK / 1e12,
            "unit": "T fp64 lane-instructions/s", "frac": frac, "frac_all_valu_instructions": frac_all, "traffic": traffic,
            "shader_clock_mhz": clock_mhz, "nominal_clock_mhz": NOMINAL_CLOCK_MHZ,
            "frac_at_running_clock": None if frac is None else frac * NOMINAL_CLOCK_MHZ / clock_mhz,
            "frac_all_valu_instructions_at_running_clock": None if frac_all is None else frac_all * NOMINAL_CLOCK_MHZ / clock_mhz,
            "issue_floor_ms_at_running_clock": issue_floor_ms, "frac_of_issue_floor": None if issue_floor_ms is None else issue_floor_ms / kern_ms,
            "fp64_lane_instructions_per_launch": insts, "avg_launch_ms": kern_ms, "median_launch_ms": float(ts[len(ts) // 2]),
            "pixel_hypotheses_per_launch": int(n) * T, "alg_bytes_per_launch": 48 * int(n),
            "share_of_solve": kern_ms / full["median_ms_per_solve"], "counters_stale": bool(stale), "counters_stale_files": stale or None,
            "hbm": {"bound": "hbm", "achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_gbs / HBM_PEAK_GBS,
                    "alg_bytes_per_solve": hbm_bytes, "formula": "57 N + 64 M iterations (SURVEY 8 d), N = %d, M = %d, iterations = %d, over the median solve time" % (full["n"], full["num_inliers"], iters)},
            "note": "the whole solve is bound by fp64 VALU issue and by serial latency chains (9x9 Jacobi SVD), not by HBM: its HBM fraction is reported "
                    "because the metric asks for it; counters: profiles/counters.json (rocprofv3 --pmc of this command)"}


def _full_solve_sequence(rsdsfm, solver, torch, dev, np, rank, args, pairs=32, passes=6):
    """BASELINE configs[4] / SURVEY 8(d) row 5: a sequence of 32 frame pairs @1280x720 with 32 DIFFERENT data seeds (one noise / outlier
    realisation each) through ONE context and ONE host thread: rsdsfm_solve_frames_dev pipelines them inside the library (lanes on
    streams of their own; the minimal solver and the single-workgroup stages of one pair run beside the streaming kernels of another).
    One call = the whole sequence; `passes` timed calls after one warm-up call, sampler seeds changing per call."""
    from concurrent.futures import ThreadPoolExecutor

    seeds_data = [0x5EED0005 + 64 * rank + 1000 * i for i in range(pairs)]
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:  # (numpy releases the GIL: ~0.6 s per frame otherwise)
        chunks = list(ex.map(lambda sd: rsdsfm.synth.make_flow_sequence(5, [sd]), seeds_data))
    meta = chunks[0][1]
    rows, cols = meta["rows"], meta["cols"]
    imgs = [torch.from_numpy(ch[0][0]).to(dev) for ch in chunks]
    dms = [torch.empty((cols, rows), dtype=torch.float64, device=dev) for _ in range(pairs)]
    Rs = [torch.empty((rows, 9), dtype=torch.float64, device=dev) for _ in range(pairs)]
    ts = [torch.empty((rows, 3), dtype=torch.float64, device=dev) for _ in range(pairs)]
    jobs = [dict(d_flow_img=im.data_ptr(), rows=rows, cols=cols, K=meta["K"], gamma=meta["gamma"], d_depth_map=dm.data_ptr(), d_R=R_.data_ptr(), d_t=t_.data_ptr())
            for im, dm, R_, t_ in zip(imgs, dms, Rs, ts)]
    call = solver.prepared_frames_solve(jobs, trials=args.trials, tol=args.tol)
    torch.cuda.synchronize()
    call([1 + i for i in range(pairs)])
    times = []
    for p_ in range(passes):
        t0 = time.perf_counter()
        res = call([1 + pairs * (p_ + 1) + i for i in range(pairs)])
        times.append(time.perf_counter() - t0)
    el = sum(times)
    inl = [int(r.num_inliers) for r in res]
    # one pair of the last pass once more through the single solve: the sequence returns the single solve's results
    chk = solver.solve_frame_dev(imgs[3].data_ptr(), rows, cols, meta["K"], meta["gamma"], dms[3].data_ptr(), trials=args.trials, tol=args.tol, seed=1 + pairs * passes + 3)
    same = bool(chk["num_inliers"] == inl[3] and np.array_equal(chk["v"], np.array(res[3].v[:])) and np.array_equal(chk["w"], np.array(res[3].w[:])))
    return {"value": rows * cols * pairs * passes / el / 1e6, "unit": "Mpixels/s", "pairs": pairs, "data_seeds": pairs, "passes": passes, "solves": pairs * passes,
            "ms_per_solve_amortised": el / (pairs * passes) * 1e3, "best_pass_ms_per_solve": min(times) / pairs * 1e3, "host_threads": 1, "contexts": 1,
            "api": "rsdsfm_solve_frames_dev (pairs pipelined inside the library)", "num_inliers_min_max": [min(inl), max(inl)],
            "pair_equals_single_solve": same}


def _full_solve_batched(rsdsfm, torch, dev, local_rank, rank, args, S, per_thread):
    """S whole solves in flight per GPU: one host thread + solver context + HIP stream each (the one-call frame solve
    synchronises internally, ctypes releases the GIL); DeepFlow-like 1280x720 pairs as in _full_solve"""
    import threading

    d = rsdsfm.synth.make_config(5, seed=0x5EED0005 + rank)
    rows, cols = d["rows"], d["cols"]
    img0 = torch.from_numpy(d["flow_img"]).to(dev)
    torch.cuda.synchronize()
    out, barrier = [None] * S, threading.Barrier(S)

    def worker(j):
        st = torch.cuda.Stream(dev)
        with torch.cuda.stream(st):
            sv = rsdsfm.Solver(local_rank, stream=st.cuda_stream, arith=args.arith)
            img = img0.clone()
            dm = torch.empty((cols, rows), dtype=torch.float64, device=dev)
            for i in range(2):
                sv.solve_frame_dev(img.data_ptr(), rows, cols, d["K"], d["gamma"], dm.data_ptr(), trials=args.trials, tol=args.tol, seed=1 + i)
            sv.synchronize()
            barrier.wait()
            t0 = time.perf_counter()
            for i in range(per_thread):
                r = sv.solve_frame_dev(img.data_ptr(), rows, cols, d["K"], d["gamma"], dm.data_ptr(), trials=args.trials, tol=args.tol, seed=1 + i)
            sv.synchronize()
            out[j] = (time.perf_counter() - t0, r["num_inliers"])
            sv.close()

    ths = [threading.Thread(target=worker, args=(j,)) for j in range(S)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    el = max(o[0] for o in out)
    return {"value": rows * cols * S * per_thread / el / 1e6, "unit": "Mpixels/s", "streams": S, "solves": S * per_thread,
            "ms_per_solve_amortised": el / (S * per_thread) * 1e3, "num_inliers": out[0][1]}


if __name__ == "__main__":
    main()
