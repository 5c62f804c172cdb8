"""TEST INFRASTRUCTURE (not a product path): a CPU stand-in for the rsdsfm_tile_* stage entry points, built on the
oracle, so that the orchestration code of dist.TiledFrameSolve (counts / offsets, the sampled-point exchange, row
all-gathers, winner selection, per-slab compaction, sign decision, slab assembly) runs under a world_size-2 gloo
group on CPU tensors.  Closed-form depth mode without refinement only (the LM rounds and the refinement stages are
exercised on the GPU, tests/test_gpu_tiled_frame.py)."""
import ctypes as C

import numpy as np


def _arr(ptr, n, dtype=np.float64):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    ct = {np.float64: C.c_double, np.int64: C.c_int64, np.int32: C.c_int32, np.uint8: C.c_uint8}[dtype]
    return np.ctypeslib.as_array((ct * n).from_address(int(ptr)))


class OracleTileSolver:
    BEST_DOUBLES = 11  # trial, count, err, hyp[8]

    def __init__(self, oracle):
        self.o = oracle

    # -- flatten of a slab (main.cc:398-444 with the global column in q.x) ------------------------
    def flatten_slab_dev(self, d_img, rows, ncols, col0, K, gamma, d_q, d_u, d_a, d_ak, thr=1e-10):
        fx, fy, cx, cy = K
        img = _arr(d_img, rows * ncols * 2).reshape(rows, ncols, 2)
        f = img.transpose(1, 0, 2).reshape(-1, 2)  # column-major scan order
        ii = np.repeat(np.arange(ncols), rows)
        jj = np.tile(np.arange(rows), ncols)
        keep = f[:, 0] * f[:, 0] + f[:, 1] * f[:, 1] > thr
        f, ii, jj = f[keep], ii[keep], jj[keep].astype(np.float64)
        n = len(f)
        h = float(rows)
        _arr(d_q, 2 * n).reshape(n, 2)[:] = np.stack([((ii + col0) - cx) * 1.0 / fx, (jj - cy) * 1.0 / fy], axis=1)
        _arr(d_u, 2 * n).reshape(n, 2)[:] = np.stack([f[:, 0] * gamma / fx, f[:, 1] * gamma / fy], axis=1)
        _arr(d_a, n)[:] = 1 + gamma * f[:, 1] / h
        p1 = gamma * jj / h
        p2 = 1.0 + gamma * (jj + f[:, 1]) / h
        _arr(d_ak, n)[:] = 0.5 * (p2 * p2 - p1 * p1)
        return n

    def minimal9_dev(self, d_q9, d_u9, d_a9, d_ak9, count, use_alpha_k, k_sign_mode, d_hyp):
        q9, u9 = _arr(d_q9, count * 18).reshape(count, 9, 2), _arr(d_u9, count * 18).reshape(count, 9, 2)
        a9, ak9 = _arr(d_a9, count * 9).reshape(count, 9), _arr(d_ak9, count * 9).reshape(count, 9)
        hyp = _arr(d_hyp, count * 8).reshape(count, 8)
        for t in range(count):
            w, v, k, rc = self.o.calculate_velocities(q9[t], u9[t], a9[t], ak9[t], use_alpha_k, k_sign_mode)
            hyp[t] = np.concatenate([w, v, [k, float(rc)]])

    def _points(self, d_q, d_u, d_a, d_ak, n):
        return _arr(d_q, 2 * n).reshape(n, 2), _arr(d_u, 2 * n).reshape(n, 2), _arr(d_a, n), _arr(d_ak, n)

    def _score_one(self, pts, h, tol):
        q, u, a, ak = pts
        if len(a) == 0:
            return 0, 0.0, np.zeros(0, dtype=np.uint8), np.zeros(0)
        rho, _ = self.o.estimate_inverse_depths(q, u, h[3:6], h[0:3], h[6], a, ak, mode=0)
        cnt, err, mask = self.o.score(q, u, a, ak, h[3:6], h[0:3], h[6], rho, tol)
        return cnt, err, mask, rho

    def tile_ransac_score_rows_dev(self, d_q, d_u, d_a, d_ak, n, d_hyp, count, d_states, depth_mode, tol, d_scored, d_rows):
        assert depth_mode == 0 and not d_scored
        pts = self._points(d_q, d_u, d_a, d_ak, n)
        hyp = _arr(d_hyp, count * 8).reshape(count, 8)
        rows = _arr(d_rows, count * 2).reshape(count, 2)
        for t in range(count):
            cnt, err, _, _ = self._score_one(pts, hyp[t], tol)
            rows[t] = [cnt, err]

    def tile_ransac_score_merge_dev(self, d_rows_all, nranks, count, d_scored, d_tcount, d_terr):
        rows = _arr(d_rows_all, nranks * count * 2).reshape(nranks, count, 2)
        tc, te = _arr(d_tcount, count), _arr(d_terr, count)
        tc[:], te[:] = 0.0, 0.0
        for r in range(nranks):  # rank order, like the kernel
            tc += rows[r, :, 0]
            te += rows[r, :, 1]

    def tile_ransac_pick_dev(self, d_tcount, d_terr, T, d_hyp, d_best):
        best = _arr(d_best, self.BEST_DOUBLES)
        tc, te, hyp = _arr(d_tcount, T), _arr(d_terr, T), _arr(d_hyp, T * 8).reshape(T, 8)
        bi, bc, be = -1, -1.0, 0.0
        for t in range(T):  # minimal.cc:278-285
            if tc[t] > bc or (tc[t] == bc and te[t] < be):
                bi, bc, be = t, tc[t], te[t]
        best[:] = np.concatenate([[bi, max(bc, 0.0), be], hyp[bi] if bi >= 0 else np.zeros(8)])

    def tile_ransac_global_inliers(self, d_best):
        return int(_arr(d_best, self.BEST_DOUBLES)[1])

    def tile_ransac_final_dev(self, d_q, d_u, d_a, d_ak, n, d_best, d_states, depth_mode, tol, d_rho, d_mask, d_idx, d_inl, d_oa, d_oak):
        best = _arr(d_best, self.BEST_DOUBLES)
        pts = self._points(d_q, d_u, d_a, d_ak, n)
        cnt, _, mask, rho = self._score_one(pts, best[3:], tol)
        _arr(d_rho, n)[:] = rho
        _arr(d_mask, n, np.uint8)[:] = mask
        sel = np.nonzero(mask)[0]
        _arr(d_idx, cnt, np.int64)[:] = sel
        _arr(d_inl, 3 * cnt).reshape(cnt, 3)[:] = np.column_stack([pts[0][sel], 1.0 / rho[sel]]) if cnt else np.zeros((0, 3))
        _arr(d_oa, cnt)[:] = pts[2][sel]
        _arr(d_oak, cnt)[:] = pts[3][sel]
        return dict(shard_inliers=cnt, best_trial=int(best[0]), w=best[3:6].copy(), v=best[6:9].copy(), k=float(best[9]), inlier_error=float(best[2]))

    def tile_zsum_dev(self, d_inl, m, d_zsum):
        _arr(d_zsum, 1)[0] = _arr(d_inl, 3 * m).reshape(m, 3)[:, 2].sum() if m else 0.0

    def tile_depth_map_dev(self, d_inl, m, d_zsums_all, nranks, m_total, v, K, rows, col0, slab_cols, d_slab, d_xs=None, d_ys=None):
        fx, fy, cx, cy = K
        total = 0.0
        for z in _arr(d_zsums_all, nranks):
            total += z
        flip = m_total > 0 and total / m_total < 0  # main.cc:466-478
        inl = _arr(d_inl, 3 * m).reshape(m, 3)
        if flip:
            inl[:, 2] *= -1.0
        slab = _arr(d_slab, rows * slab_cols).reshape(slab_cols, rows)
        slab[:] = 0.0
        xs = (fx * inl[:, 0] + cx + 0.5).astype(np.int32)
        ys = (fy * inl[:, 1] + cy + 0.5).astype(np.int32)
        for i in range(m):  # sequential: the last writer wins (main.cc:499-508)
            if col0 <= xs[i] < col0 + slab_cols and 0 <= ys[i] < rows:
                slab[xs[i] - col0, ys[i]] = inl[i, 2]
        if d_ys:
            _arr(d_ys, m, np.int32)[:] = ys
        vv = np.asarray(v, dtype=np.float64)
        return (vv * -1.0 if flip else vv.copy()), bool(flip)
