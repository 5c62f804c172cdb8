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
