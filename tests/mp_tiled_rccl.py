"""Helper of tests/test_gpu_tiled_native.py (not a test): one rank of a 2-process run of the NATIVE column-tiled solve over a REAL
2-rank RCCL communicator on ONE GPU.  RCCL refuses two ranks of one host on one device ("Duplicate GPU detected"), but it identifies a
host by NCCL_HOSTID when that is set: giving every rank its own makes RCCL treat the ranks as two single-GPU nodes and connect them
through its socket transport over the loopback interface.  That is not xGMI -- it proves nothing about bandwidth -- but it runs the
multi-rank code path for real: ncclCommInitRank with 2 ranks from the broadcast unique id, in-place ncclAllGather (ncclChar) and
ncclAllReduce on the context's stream, a second communicator next to torch's, and the driver's rank-ordered protocol on top.
The unique id travels over a gloo group.  Environment per rank (set by the test): NCCL_HOSTID, NCCL_SOCKET_IFNAME=lo, NCCL_IB_DISABLE=1.

RSDSFM_TEST_FRAME_NPY = a (rows, cols, 2) flow image saved by the test (BASELINE configs[3], 3840x2160): the ranks solve THAT frame
(RSDSFM_TEST_FRAME_META: json with K, gamma, trials, tol, seed) instead of the small one, every rank loading only its column slab,
and write what the test compares with the oracle chain: pose, counts, refinement summary, the gathered depth map (rank 0) and every
inlier's scanline index (one file per rank, concatenated in rank order by the test).

RSDSFM_TEST_SEQUENCE=1: instead, a SEQUENCE of frames on the one communicator that walks the driver's paths -- cold, ahead on the dense
counts, a frame with a hole in rank 1's slab (every rank starts over through the counts exchange), cold again, ahead, a frame with a pixel
outside the range of the function cores in rank 1's slab (every rank starts the RANSAC over), ahead -- and records, per solve and per
rank, the results and rsdsfm_tiled_info: every collective of every path has to pair up between the two processes, or the run hangs.

RSDSFM_TEST_FUZZ=N: N random cases of tests/fuzz_tiled.py (same generator, same seed on both ranks: frame size, data kind, tolerance, trials,
flow mode, acceleration mode, a sequence of clean / holed / poisoned frames) over the one RCCL communicator; rank 0 also solves every frame
on a single context and compares (fuzz_tiled.compare); the ranks' results are compared with each other bit for bit."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ["NCCL_HOSTID"] = "rsdsfm-test-host-%d" % rank  # before RCCL is loaded
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    os.environ.setdefault("NCCL_IB_DISABLE", "1")
    import torch
    import torch.distributed as dist

    import rsdsfm

    dist.init_process_group("gloo")
    dev = torch.device("cuda", 0)
    flow_mode = int(os.environ.get("RSDSFM_TEST_FLOW_MODE", "0"))
    big = os.environ.get("RSDSFM_TEST_FRAME_NPY")
    if big:
        meta = json.loads(os.environ["RSDSFM_TEST_FRAME_META"])
        flow_img = np.load(big, mmap_mode="r")
        d = dict(flow_img=flow_img, rows=flow_img.shape[0], cols=flow_img.shape[1], K=tuple(meta["K"]), gamma=meta["gamma"])
        skw = dict(trials=meta["trials"], tol=meta["tol"], seed=meta["seed"])
    else:
        d = rsdsfm.synth.make_config(3, rows=96, cols=250)
        skw = dict(trials=14, tol=0.002, seed=7)
    rows, cols, K, gamma = d["rows"], d["cols"], d["K"], d["gamma"]
    c0, sc, per = rsdsfm.tiled_slab_bounds(cols, world, rank)
    solver = rsdsfm.Solver(0)
    ident = torch.zeros(rsdsfm.DIST_ID_BYTES, dtype=torch.uint8)
    if rank == 0:
        ident = torch.frombuffer(bytearray(rsdsfm.dist_unique_id()), dtype=torch.uint8).clone()
    dist.broadcast(ident, 0)
    out = {"world": world, "init": None}
    try:
        solver.dist_init(world, rank, bytes(ident.numpy().tobytes()))
        out["init"] = "ok"
    except rsdsfm.RsdsfmError as e:  # e.g. a box whose RCCL cannot open the loopback interface: reported, the test skips
        out["init"] = str(e)
    if out["init"] == "ok" and os.environ.get("RSDSFM_TEST_FUZZ"):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import fuzz_tiled

        bad, solves, paths = [], 0, {}
        for c in range(int(os.environ["RSDSFM_TEST_FUZZ"])):
            rows, cols, K, gamma, frames, kinds, kw, tag, _, _ = fuzz_tiled.draw_case(rsdsfm, 20261003, c)
            c0, sc, per = rsdsfm.tiled_slab_bounds(cols, world, rank)
            dm = torch.zeros(cols * rows, dtype=torch.float64, device=dev)
            R = torch.empty(rows * 9, dtype=torch.float64, device=dev)
            t = torch.empty(rows * 3, dtype=torch.float64, device=dev)
            for i, f in enumerate(frames):
                slab = torch.from_numpy(np.ascontiguousarray(f[:, c0:c0 + sc, :])).to(dev)
                try:
                    r = solver.solve_frame_tiled_dev(slab.data_ptr() if sc else 0, rows, cols, K, gamma, dm.data_ptr(), R.data_ptr(), t.data_ptr(), **kw)
                    torch.cuda.synchronize()
                    r["depth_map"] = dm.cpu().numpy()
                    r["R"], r["t"] = R.cpu().numpy().reshape(rows, 9), t.cpu().numpy().reshape(rows, 3)
                    err = None
                except rsdsfm.RsdsfmError as e:  # (both ranks fail together: e.g. no real k for a hypothesis in acceleration mode)
                    r, err = None, str(e)[:80]
                mine = None if r is None else (r["n"], r["num_inliers"], r["best_trial"], r["v"].tobytes(), r["w"].tobytes(), r["k"], str(r["refine_summary"]),
                                               r["depth_map"].tobytes(), r["info"]["path_flags"] & 0xFF, r["info"]["collectives"])
                gathered = [None] * world
                dist.all_gather_object(gathered, (mine, err is None))
                solves += 1
                if any(g != gathered[0] for g in gathered):
                    bad.append("%s frame %d %s: the ranks disagree" % (tag, i, kinds[i]))
                if rank == 0 and r is not None:
                    paths[str(r["info"]["path_flags"] & 0xFF)] = paths.get(str(r["info"]["path_flags"] & 0xFF), 0) + 1
                    try:
                        one = fuzz_tiled.single(rsdsfm, torch, f, rows, cols, K, gamma, kw)
                    except rsdsfm.RsdsfmError as e:
                        bad.append("%s frame %d %s: the single-context solve fails (%s), the tiled one did not" % (tag, i, kinds[i], str(e)[:80]))
                        continue
                    rtol = 1e-5 if one["refine_summary"]["num_iterations"] >= 12 else (1e-6 if kw["use_acceleration_mode"] else 1e-9)
                    what = fuzz_tiled.compare(r, one, rtol)
                    if what and not (one["refine_summary"]["num_iterations"] >= 20 and what.startswith(("refinement", "pose", "depth", "flipped"))):
                        bad.append("%s frame %d %s: %s" % (tag, i, kinds[i], what))
        out.update(fuzz=dict(solves=solves, bad=bad, paths=paths))
        solver.dist_finalize()
        if rank == 0:
            with open(os.environ["RSDSFM_TILED_OUT"], "w") as f:
                json.dump(out, f)
        solver.close()
        dist.barrier()
        dist.destroy_process_group()
        return
    if out["init"] == "ok" and os.environ.get("RSDSFM_TEST_SEQUENCE"):
        d = rsdsfm.synth.make_config(5, rows=64, cols=480)  # (alpha = 1 + gamma f_y / h vanishes for f_y = -128 px: h = 64, gamma = 0.5)
        rows, cols, K, gamma = d["rows"], d["cols"], d["K"], 0.5
        c0, sc, per = rsdsfm.tiled_slab_bounds(cols, world, rank)
        clean = np.array(d["flow_img"])
        holed = clean.copy()
        holed[10:30, 300:340] = 0.0          # columns of rank 1's slab
        bad = clean.copy()
        bad[17, 301] = (3.0, 1e160)         # rank 1's slab
        frames = [clean, clean, holed, clean, clean, bad, clean]
        # 1: the iterate-by-iterate kernels (the ones whose function cores a NaN pixel leaves); 0 (default): the analytic LM trajectory
        solver.set_lm_arithmetic(int(os.environ.get("RSDSFM_TEST_LM_ARITHMETIC", "0")))
        dm = torch.zeros(cols * rows, dtype=torch.float64, device=dev)
        seq = []
        for f in frames:
            slab = torch.from_numpy(np.ascontiguousarray(f[:, c0:c0 + sc, :])).to(dev)
            r = solver.solve_frame_tiled_dev(slab.data_ptr(), rows, cols, K, gamma, dm.data_ptr(), flow_index_mode=flow_mode, trials=20, tol=0.05, seed=11)
            torch.cuda.synchronize()
            dmh = dm.cpu().numpy()
            seq.append(dict(n=r["n"], num_inliers=r["num_inliers"], best_trial=r["best_trial"], v=list(r["v"]), w=list(r["w"]),
                            iterations=r["refine_summary"]["num_iterations"], depth_nonzero=int((dmh != 0).sum()), depth_sum=float(dmh.sum()),
                            path_flags=r["info"]["path_flags"], collectives=r["info"]["collectives"], host_syncs=r["info"]["host_syncs"]))
        gathered = [None] * world
        dist.all_gather_object(gathered, seq)
        out.update(sequence=gathered, restarts=solver.ransac_restarts(), lma_restarts=solver.lma_restarts()[0])
        solver.dist_finalize()
        if rank == 0:
            with open(os.environ["RSDSFM_TILED_OUT"], "w") as f:
                json.dump(out, f)
        solver.close()
        dist.barrier()
        dist.destroy_process_group()
        return
    if out["init"] == "ok":
        slab = torch.from_numpy(np.ascontiguousarray(d["flow_img"][:, c0:c0 + sc, :])).to(dev)  # each rank only holds its slab
        dm = torch.zeros(cols * rows, dtype=torch.float64, device=dev)
        for rep in range(2):  # the communicator is reused across solves
            r = solver.solve_frame_tiled_dev(slab.data_ptr(), rows, cols, K, gamma, dm.data_ptr(), flow_index_mode=flow_mode, **skw)
        torch.cuda.synchronize()
        dmh = dm.cpu().numpy()
        if big:  # what the full-size test compares with the oracle chain
            import ctypes

            hip = ctypes.CDLL("libamdhip64.so")
            m = r["info"]["shard_inliers"]
            ys = torch.empty(max(m, 1), dtype=torch.int32, device=dev)
            if m:
                assert hip.hipMemcpy(ctypes.c_void_p(ys.data_ptr()), ctypes.c_void_p(r["d_scanline"]), ctypes.c_size_t(4 * m), 3) == 0
            base = os.environ["RSDSFM_TILED_OUT"]
            np.save(base + ".ys%d.npy" % rank, ys.cpu().numpy()[:m])
            if rank == 0:
                np.save(base + ".depth.npy", dmh)
        mine = np.concatenate([r["v"], r["w"], [r["k"], r["num_inliers"], r["best_trial"], float((dmh != 0).sum()), dmh.sum()]])
        t = torch.from_numpy(mine).clone()
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        out.update(n=r["n"], num_inliers=r["num_inliers"], best_trial=r["best_trial"], v=list(r["v"]), w=list(r["w"]), k=r["k"],
                   depth_nonzero=int((dmh != 0).sum()), depth_sum=float(dmh.sum()), ranks_agree=bool(all(torch.equal(o, outs[0]) for o in outs)), info=r["info"],
                   iterations=r["refine_summary"]["num_iterations"], refine_summary=r["refine_summary"], flipped=bool(r["flipped"]))
    if out["init"] == "ok" and not big:
        # the row-tiled dense depth solve over the same communicator
        t_ = d["truth"]
        v = t_["v"] / np.linalg.norm(t_["v"])
        n = len(d["alpha"])
        i0, cnt, _ = rsdsfm.tiled_shard_bounds(n, world, rank)
        tt = lambda a: torch.from_numpy(np.ascontiguousarray(a[i0:i0 + cnt])).to(dev)
        q, u, a, ak = tt(d["q"]), tt(d["u"]), tt(d["alpha"]), tt(d["alpha_k"])
        full = torch.zeros(max(n, 2), dtype=torch.float64, device=dev)
        sm, info = solver.estimate_inverse_depths_tiled_dev(q.data_ptr(), u.data_ptr(), n, v, t_["w"], 0.0, a.data_ptr(), ak.data_ptr(), full.data_ptr())
        torch.cuda.synchronize()
        out.update(depth_sum_tiled=float(full[:n].sum().item()), depth_lm=sm, depth_info=info)
    if out["init"] == "ok":
        solver.dist_finalize()
    if rank == 0:
        with open(os.environ["RSDSFM_TILED_OUT"], "w") as f:
            json.dump(out, f)
    solver.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
