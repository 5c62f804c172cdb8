"""Row-tiled multi-GPU dense depth solve (SURVEY section 8 e; BASELINE config 4).

The flattened point arrays shard by contiguous index ranges (image row tiles in the reference's column-major scan
order are column tiles; any contiguous range works because every per-pixel solve is independent given the pose).
One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests):

  closed-form mode   no exchange during the solve; ONE all-gather of the depth shards at the end.
  Ceres-LM mode      the accept / converge decisions of the emulated Ceres trust-region loop are global, so per LM
                     launch every rank reduces its shard to one row of NS sums, the rows are all-gathered (rank
                     order => every rank adds them in the same order and takes the same decisions), then the depth
                     shards are all-gathered once.  Scalar traffic: NS x 8 B per rank per launch (latency-bound).

The driver is written against a small stage interface so that the SAME orchestration code runs on the HIP backend
(product) and, in tests only, on a CPU stand-in.  A process may own several logical shards (used to test row-tiling
invariance on one GPU).
"""
import numpy as np

DEPTH_CLOSED_FORM, DEPTH_CERES_LM = 0, 1


def shard_bounds(n, nshards):
    """Contiguous shards; every shard start is even so that the 8-byte-per-point arrays stay 16-byte aligned."""
    per = -(-n // nshards)
    per += per & 1
    return [(min(n, s * per), min(n, (s + 1) * per)) for s in range(nshards)], per


class HipDepthStage:
    """Stage backend over the C ABI for ONE shard resident on this process' GPU (torch tensors)."""

    def __init__(self, solver, q, u, alpha, alpha_k, v, w, k, torch):
        self.s, self.torch = solver, torch
        self.q, self.u, self.a, self.ak = q, u, alpha, alpha_k
        self.v, self.w, self.k = v, w, k
        self.n = int(alpha.shape[0])
        self.rho = torch.empty(max(self.n, 1), dtype=torch.float64, device=alpha.device)
        self.ns = solver.lib.rsdsfm_depth_lm_sums_row_size()

    def _ptrs(self):
        return (self.q.data_ptr(), self.u.data_ptr(), self.n, self.v, self.w, self.k, self.a.data_ptr(), self.ak.data_ptr(), self.rho.data_ptr())

    def closed_form(self):
        self.s.estimate_inverse_depths_dev(*self._ptrs(), mode=DEPTH_CLOSED_FORM)

    def lm_launch(self, launch_id):
        self.s.depth_lm_launch_dev(*self._ptrs(), launch_id=launch_id)

    def lm_reduce(self):
        row = self.torch.empty(self.ns, dtype=self.torch.float64, device=self.a.device)
        self.s.depth_lm_reduce_dev(self.n, row.data_ptr())
        return row

    def lm_decide_rows(self, rows, n_total, launch_id):
        rows = rows.contiguous()
        self.s.depth_lm_decide_rows_dev(rows.data_ptr(), int(rows.shape[0]), n_total, launch_id)
        self._keep = rows  # keep the buffer alive until the stream has consumed it

    def lm_state(self):
        return self.s.depth_lm_state()

    def result(self):
        return self.rho[: self.n]


class TiledDepthSolve:
    """stages: the logical shards this process owns (in global shard order: rank-major).  dist: torch.distributed
    (initialised) or None for a single process."""

    def __init__(self, stages, n_total, per, torch, dist=None, max_launches=120):
        self.stages, self.n_total, self.per, self.torch, self.dist = stages, n_total, per, torch, dist
        self.world = dist.get_world_size() if dist is not None else 1
        self.max_launches = max_launches

    def _all_gather(self, t):
        """[local, ...] -> [world * local, ...] in rank order"""
        if self.dist is None or self.world == 1:
            return t
        outs = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t.contiguous())
        return self.torch.cat(outs, dim=0)

    def _gather_depth(self):
        torch = self.torch
        parts = []
        for st in self.stages:
            r = st.result()
            pad = torch.zeros(self.per, dtype=r.dtype, device=r.device)
            pad[: r.shape[0]] = r
            parts.append(pad)
        full = self._all_gather(torch.stack(parts)).reshape(-1)  # the one data-path all-gather (8 B x N)
        return full[: self.n_total]

    def solve(self, mode=DEPTH_CERES_LM):
        torch = self.torch
        if mode == DEPTH_CLOSED_FORM:
            for st in self.stages:
                st.closed_form()
            return self._gather_depth(), None
        launch_id, summary = 0, None
        for _ in range(self.max_launches):
            for st in self.stages:
                st.lm_launch(launch_id)  # speculative launch: per-pixel LM trajectory + sums
            rows = torch.stack([st.lm_reduce() for st in self.stages])
            rows = self._all_gather(rows)  # [total shards, NS] in global shard order
            for st in self.stages:
                st.lm_decide_rows(rows, self.n_total, launch_id)
            states = [st.lm_state() for st in self.stages]
            summary = states[0][2]
            running = [s[0] == 0 for s in states]
            # "continue" depends only on the (identical) sums; "written already" (1) vs "needs the apply launch" (2)
            # depends on each context's own iterate predictor and may differ between shards
            assert all(running) or not any(running), "shards disagree on the LM decision"
            if not running[0]:
                for st, (status, next_launch, _) in zip(self.stages, states):
                    if status == 2:
                        st.lm_launch(next_launch)  # apply-only launch
                return self._gather_depth(), summary
            assert len({s[1] for s in states}) == 1
            launch_id = states[0][1]
        raise RuntimeError("LM state machine did not terminate")


# ---------------------------------------------------------------------------------------------------
# row-tiled WHOLE-FRAME solve (flatten -> RANSAC -> refinement -> sign fix + depth map), SURVEY section 8(e)
# ---------------------------------------------------------------------------------------------------
def slab_bounds(cols, nslabs):
    """contiguous column slabs [(c0, c1)] of the image; the reference flattens column-major (main.cc:398-444), so the
    slabs' point lists in slab order concatenate to the reference's point list"""
    per = -(-cols // nslabs)
    return [(min(cols, s * per), min(cols, (s + 1) * per)) for s in range(nslabs)], per


class HipFrameShard:
    """One column slab of the flow image resident on this process' GPU; owns the slab's flattened arrays.
    `solver` must run on torch's current stream (Solver(device, stream=torch.cuda.current_stream().cuda_stream))."""

    def __init__(self, solver, img_slab, col0, K, gamma, torch, thr=1e-10):
        self.s, self.torch, self.K, self.gamma, self.col0 = solver, torch, K, gamma, int(col0)
        self.img = img_slab.contiguous()
        self.rows, self.ncols = int(self.img.shape[0]), int(self.img.shape[1])
        npix = max(self.rows * self.ncols, 1)
        f64 = dict(dtype=torch.float64, device=self.img.device)
        self.q, self.u = torch.empty(2 * npix, **f64), torch.empty(2 * npix, **f64)
        self.a, self.ak = torch.empty(npix, **f64), torch.empty(npix, **f64)
        self.n = solver.flatten_slab_dev(self.img.data_ptr(), self.rows, self.ncols, self.col0, K, gamma, self.q.data_ptr(),
                                         self.u.data_ptr(), self.a.data_ptr(), self.ak.data_ptr(), thr)
        self.m = 0

    def point_ptrs(self):
        return self.q.data_ptr(), self.u.data_ptr(), self.a.data_ptr(), self.ak.data_ptr(), self.n

    def packed_points(self, loc):
        """rows (qx, qy, ux, uy, alpha, alpha_k) of the shard-local indices `loc` (int64 tensor)"""
        torch = self.torch
        q, u = self.q.view(-1, 2), self.u.view(-1, 2)
        return torch.cat([q[loc], u[loc], self.a[loc].unsqueeze(1), self.ak[loc].unsqueeze(1)], dim=1)


class TiledFrameSolve:
    """shards: the HipFrameShard objects this process owns, in global slab order (rank-major).  dist: an initialised
    torch.distributed (backend "nccl" = RCCL) or None.  Every collective is an all-gather / all-reduce of a few
    hundred bytes to a few KB (latency-bound) except the final all-gather of the depth-map slabs."""

    def __init__(self, shards, rows, cols, per_cols, torch, dist=None):
        from . import tile_sizes

        self.shards, self.rows, self.cols, self.per_cols, self.torch, self.dist = shards, rows, cols, per_cols, torch, dist
        self.world = dist.get_world_size() if dist is not None else 1
        self.state_bytes, self.best_bytes, self.nsr, self.batch = tile_sizes()
        self.dev = shards[0].img.device
        self.s0 = shards[0].s  # the context that runs the per-process (replicated) decisions

    # -- collectives ------------------------------------------------------------------------------
    def _all_gather(self, t):
        """[local, ...] -> [world * local, ...] in rank order"""
        if self.dist is None or self.world == 1:
            return t.contiguous()
        outs = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t.contiguous())
        return self.torch.cat(outs, dim=0).contiguous()

    def _all_reduce_sum(self, t):
        if self.dist is not None and self.world > 1:
            self.dist.all_reduce(t)
        return t

    # -- stages -----------------------------------------------------------------------------------
    def _sampled_points(self, samples, offsets):
        """the 9 T sampled points (global indices) packed [9 T, 6]: every shard contributes the rows it owns, zeros
        elsewhere; the sum over shards / ranks is exact (one non-zero term per entry)"""
        torch = self.torch
        smp = torch.from_numpy(samples.reshape(-1).astype(np.int64)).to(self.dev)
        acc = torch.zeros(smp.shape[0], 6, dtype=torch.float64, device=self.dev)
        for sh, off in zip(self.shards, offsets):
            if sh.n == 0:
                continue
            loc = smp - off
            mine = (loc >= 0) & (loc < sh.n)
            pts = sh.packed_points(loc.clamp(0, sh.n - 1))
            acc += torch.where(mine.unsqueeze(1), pts, torch.zeros_like(pts))
        return self._all_reduce_sum(acc)

    def _ransac(self, T, tol, seed, use_alpha_k, depth_mode, k_sign_mode, samples):
        from . import sample_indices

        torch, s0 = self.torch, self.s0
        L = len(self.shards)
        counts = self._all_gather(torch.tensor([sh.n for sh in self.shards], dtype=torch.int64, device=self.dev)).cpu().numpy()
        n_total = int(counts.sum())
        first = (self.dist.get_rank() if self.dist is not None else 0) * L
        offs_all = np.concatenate([[0], np.cumsum(counts)])
        self.offsets = [int(offs_all[first + i]) for i in range(L)]
        self.n_total = n_total
        self.counts_all, self.first_shard = counts.astype(np.int64), first
        if n_total < 9:
            raise ValueError("ransac needs at least 9 points (the reference would compute rand() % 0)")
        if samples is None:
            samples = sample_indices(n_total, T, seed)
        samples = np.ascontiguousarray(samples, dtype=np.int32).reshape(T, 9)
        f64 = dict(dtype=torch.float64, device=self.dev)
        i32 = dict(dtype=torch.int32, device=self.dev)
        Tn = max(T, 1)
        hyp = torch.zeros(Tn * 8, **f64)
        states = torch.zeros(Tn * self.state_bytes, dtype=torch.uint8, device=self.dev)
        scored = torch.zeros(Tn, **i32)
        tcount, terr = torch.zeros(Tn, **f64), torch.zeros(Tn, **f64)
        best = torch.zeros(self.best_bytes, dtype=torch.uint8, device=self.dev)
        rounds = 0
        if T > 0:
            pts = self._sampled_points(samples, self.offsets)
            q9, u9 = pts[:, 0:2].contiguous(), pts[:, 2:4].contiguous()
            a9, ak9 = pts[:, 4].contiguous(), pts[:, 5].contiguous()
            s0.minimal9_dev(q9.data_ptr(), u9.data_ptr(), a9.data_ptr(), ak9.data_ptr(), T, use_alpha_k, k_sign_mode, hyp.data_ptr())
            for b0 in range(0, T, self.batch):
                B = min(self.batch, T - b0)
                hyp_b = hyp.data_ptr() + 8 * 8 * b0
                st_b = states.data_ptr() + self.state_bytes * b0
                sc_b, tc_b, te_b = scored.data_ptr() + 4 * b0, tcount.data_ptr() + 8 * b0, terr.data_ptr() + 8 * b0
                need_score = True
                if depth_mode == DEPTH_CERES_LM:
                    flags = torch.zeros(2, **i32)
                    for rnd in range(4 * 50 + 1):
                        rows = torch.empty(L, B, self.nsr, **f64)
                        for i, sh in enumerate(self.shards):
                            sh.s.tile_ransac_lm_rows_dev(*sh.point_ptrs(), hyp_b, B, st_b, rnd, tol, rows[i].data_ptr())
                        rows_all = self._all_gather(rows)  # [slabs, B, NSR] in global slab order
                        s0.tile_ransac_decide_dev(rows_all.data_ptr(), int(rows_all.shape[0]), B, st_b, n_total, rnd, flags.data_ptr(), sc_b, tc_b, te_b)
                        rounds += 1
                        fl = flags.cpu().numpy()
                        if fl[0] == 0:
                            break
                    else:
                        raise RuntimeError("LM state machines did not terminate")
                    need_score = fl[1] > 0
                if need_score:
                    rows = torch.empty(L, B, 2, **f64)
                    for i, sh in enumerate(self.shards):
                        sh.s.tile_ransac_score_rows_dev(*sh.point_ptrs(), hyp_b, B, st_b, depth_mode, tol, sc_b if depth_mode == DEPTH_CERES_LM else 0, rows[i].data_ptr())
                    rows_all = self._all_gather(rows)
                    s0.tile_ransac_score_merge_dev(rows_all.data_ptr(), int(rows_all.shape[0]), B, sc_b if depth_mode == DEPTH_CERES_LM else 0, tc_b, te_b)
        s0.tile_ransac_pick_dev(tcount.data_ptr(), terr.data_ptr(), T, hyp.data_ptr(), best.data_ptr())
        m_total = s0.tile_ransac_global_inliers(best.data_ptr())
        win = None
        for sh in self.shards:
            n1 = max(sh.n, 1)
            sh.rho = torch.empty(n1, **f64)
            sh.mask = torch.empty(n1, dtype=torch.uint8, device=self.dev)
            sh.idx = torch.empty(n1, dtype=torch.int64, device=self.dev)
            sh.inl = torch.empty(3 * n1, **f64)
            sh.in_a, sh.in_ak = torch.empty(n1, **f64), torch.empty(n1, **f64)
            # the compaction scan rewrites the record's scan total, so every shard works on its own copy
            best_s = best.clone()
            win = sh.s.tile_ransac_final_dev(*sh.point_ptrs()[:4], sh.n, best_s.data_ptr(), states.data_ptr(), depth_mode, tol, sh.rho.data_ptr(),
                                             sh.mask.data_ptr(), sh.idx.data_ptr(), sh.inl.data_ptr(), sh.in_a.data_ptr(), sh.in_ak.data_ptr())
            sh.m = win["shard_inliers"]
        ms = self._all_gather(torch.tensor([sh.m for sh in self.shards], dtype=torch.int64, device=self.dev)).cpu().numpy()
        if int(ms.sum()) != m_total:
            raise RuntimeError("inlier count mismatch between scoring (%d) and compaction (%d)" % (m_total, int(ms.sum())))
        self.m_total, self.m_all = m_total, ms
        return dict(n=n_total, num_inliers=m_total, best_trial=win["best_trial"], ransac_w=win["w"], ransac_v=win["v"], ransac_k=win["k"],
                    inlier_error=win["inlier_error"], trial_count=tcount[:T].cpu().numpy().astype(np.int64), trial_err=terr[:T].cpu().numpy(),
                    ransac_rounds=rounds)

    def _staged(self, stage, size):
        """one refinement stage: shard rows -> all-gather -> identical apply on every shard's state machine"""
        torch = self.torch
        rows = torch.empty(len(self.shards), size, dtype=torch.float64, device=self.dev)
        for i, sh in enumerate(self.shards):
            sh.s.tile_refine_rows_dev(stage, rows[i].data_ptr())
        rows_all = self._all_gather(rows)
        for sh in self.shards:
            sh.s.tile_refine_apply_dev(stage, rows_all.data_ptr(), int(rows_all.shape[0]), self.m_total)
        return rows_all

    def rank_indexed_flow(self):
        """The reference's default flow indexing (quirk Q2: main.cc:457 passes the UN-compacted flow, nonlinearRefinement.cc:209-212 reads
        column i for the i-th inlier): per local shard the columns of the GLOBAL flow list at the shard's global inlier ranks
        [prefix, prefix + m) as an [m, 2] tensor.  A shard never holds more inliers than points, so those columns sit on the shards in
        front of it: every shard contributes the head of its flow list up to the global inlier count (ONE all-gather; all counts are
        known everywhere, so all ranks agree on its size -- and on skipping it when every needed column is local, in which case
        None is returned and column i of the shard's own list is the one)."""
        torch = self.torch
        cnt, ms = self.counts_all, np.asarray(self.m_all, dtype=np.int64)
        po = np.concatenate([[0], np.cumsum(cnt)])[:-1]
        pm = np.concatenate([[0], np.cumsum(ms)])[:-1]
        if not np.any((ms > 0) & (pm != po)):
            return None
        heads_len = np.minimum(cnt, np.maximum(self.m_total - po, 0))
        lmax = max(int(heads_len.max()), 1)
        heads = torch.zeros(len(self.shards), lmax, 2, dtype=torch.float64, device=self.dev)
        for i, sh in enumerate(self.shards):
            h = int(heads_len[self.first_shard + i])
            if h:
                heads[i, :h] = sh.u.view(-1, 2)[:h]
        heads_all = self._all_gather(heads)  # [shards, lmax, 2] in global shard order
        ends = torch.from_numpy(np.cumsum(cnt)).to(self.dev)
        starts = torch.from_numpy(po.astype(np.int64)).to(self.dev)
        out = []
        for i, sh in enumerate(self.shards):
            g = int(pm[self.first_shard + i]) + torch.arange(sh.m, dtype=torch.int64, device=self.dev)
            r = torch.searchsorted(ends, g, right=True)
            out.append(heads_all[r, g - starts[r]].contiguous())
        return out

    def _refine(self, v, w, k, const_acceleration, flow_index_mode=1):
        torch = self.torch
        sizes = [self.s0.tile_refine_row_size(const_acceleration, st) for st in range(3)]
        by_rank = self.rank_indexed_flow() if flow_index_mode == 0 else None
        self._flow_keep = by_rank
        for i, sh in enumerate(self.shards):
            sh.inl_ref = torch.empty(3 * max(sh.m, 1), dtype=torch.float64, device=self.dev)
            if flow_index_mode == 0:  # column i of d_flow belongs to inlier i (the shard's own list when nothing had to be fetched)
                fl, nf = (by_rank[i], sh.m) if by_rank is not None else (sh.u, sh.n)
                sh.s.tile_refine_begin_dev(fl.data_ptr(), nf, sh.m, sh.inl.data_ptr(), sh.in_a.data_ptr(), sh.in_ak.data_ptr(), 0, v, w, k, const_acceleration, 0)
                continue
            sh.s.tile_refine_begin_dev(sh.u.data_ptr(), sh.n, sh.m, sh.inl.data_ptr(), sh.in_a.data_ptr(), sh.in_ak.data_ptr(), sh.idx.data_ptr(), v, w, k, const_acceleration)
        keep = [self._staged(0, sizes[0])]
        st = self.s0.tile_refine_poll()
        it = 0
        while st["running"]:
            if it > 4 * 50 + 16:
                raise RuntimeError("refinement did not terminate")
            keep = [self._staged(1, sizes[1]), self._staged(2, sizes[2])]
            st = self.s0.tile_refine_poll()  # synchronises: the gathered rows above are consumed
            it += 1
        for sh in self.shards:
            sh.s.tile_refine_finish_dev(sh.inl_ref.data_ptr())
        del keep
        return st

    def solve(self, trials=50, tol=0.05, seed=1, use_acceleration_mode=False, use_refinement=True, depth_mode=DEPTH_CERES_LM,
              k_sign_mode=0, samples=None, pose_table=False, flow_index_mode=0):
        """returns the dict of Solver.solve_frame_dev (same keys) plus depth_map (torch, column-major [cols * rows]);
        the shards keep their refined inliers (sh.final, 3 x sh.m) and scanline indices (sh.ys).  flow_index_mode as in
        Solver.solve_frame_dev: 0 (default) = the reference's rank-indexed flow (see rank_indexed_flow), 1 = gathered."""
        torch = self.torch
        res = self._ransac(int(trials), float(tol), int(seed), int(bool(use_acceleration_mode)), depth_mode, k_sign_mode, samples)
        v, w, k = res["ransac_v"], res["ransac_w"], res["ransac_k"]
        if use_refinement:
            st = self._refine(v, w, k, bool(use_acceleration_mode), int(flow_index_mode))
            v, w, k = st["v"], st["w"], st["k"]
            res["refine_summary"] = st["summary"]
            for sh in self.shards:
                sh.final = sh.inl_ref
        else:
            for sh in self.shards:
                sh.final = sh.inl
        zs = torch.empty(len(self.shards), dtype=torch.float64, device=self.dev)
        for i, sh in enumerate(self.shards):
            sh.s.tile_zsum_dev(sh.final.data_ptr(), sh.m, zs[i].data_ptr())
        zs_all = self._all_gather(zs)
        slabs = torch.zeros(len(self.shards), self.per_cols * self.rows, dtype=torch.float64, device=self.dev)
        flipped, v_out = False, v
        for i, sh in enumerate(self.shards):
            sh.ys = torch.empty(max(sh.m, 1), dtype=torch.int32, device=self.dev)
            v_out, flipped = sh.s.tile_depth_map_dev(sh.final.data_ptr(), sh.m, zs_all.data_ptr(), int(zs_all.shape[0]), self.m_total, v, sh.K, self.rows,
                                                     sh.col0, sh.ncols, slabs[i].data_ptr(), None, sh.ys.data_ptr())
        depth = self._all_gather(slabs).reshape(-1)[: self.cols * self.rows]  # the one data-path all-gather
        res.update(v=np.asarray(v_out), w=np.asarray(w), k=float(k), flipped=flipped, depth_map=depth)
        if pose_table:
            R = torch.empty(self.rows * 9, dtype=torch.float64, device=self.dev)
            t = torch.empty(self.rows * 3, dtype=torch.float64, device=self.dev)
            self.s0.pose_table_dev(v_out, w, k, self.shards[0].gamma, self.rows, R.data_ptr(), t.data_ptr())
            res.update(R=R, t=t)
        return res
