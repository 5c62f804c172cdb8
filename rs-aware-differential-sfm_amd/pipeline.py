"""Device-resident end-to-end solve of one frame pair -- the solver part of the reference's evaluateSingleRun()
(main.cc:398-522): flatten + alpha -> RANSAC(T) -> nonlinear refinement -> sign flip + depth map -> per-scanline
pose table, every stage a C-ABI call on buffers that stay in HBM (torch tensors are only the allocator)."""
import numpy as np


class FramePipeline:
    def __init__(self, solver, torch, device, rows, cols, K, gamma):
        self.s, self.torch, self.dev = solver, torch, device
        self.rows, self.cols, self.K, self.gamma = rows, cols, K, gamma
        n = rows * cols
        f64 = dict(dtype=torch.float64, device=device)
        self.q, self.u = torch.empty((n, 2), **f64), torch.empty((n, 2), **f64)
        self.alpha, self.alpha_k = torch.empty(n, **f64), torch.empty(n, **f64)
        self.inl, self.inl_ref = torch.empty((n, 3), **f64), torch.empty((n, 3), **f64)
        self.in_alpha, self.in_alpha_k = torch.empty(n, **f64), torch.empty(n, **f64)
        self.idx = torch.empty(n, dtype=torch.int64, device=device)
        self.mask = torch.empty(n, dtype=torch.uint8, device=device)
        self.rho = torch.empty(n, **f64)
        self.depth_map = torch.empty((cols, rows), **f64)  # column-major rows x cols
        self.ys = torch.empty(n, dtype=torch.int32, device=device)
        self.R, self.t = torch.empty((rows, 9), **f64), torch.empty((rows, 3), **f64)

    def solve(self, flow_img_dev, trials=50, tol=0.05, seed=1, use_alpha_k=False, refine=True, depth_mode=1, samples=None,
              flow_index_mode=0):
        """flow_index_mode 0 = the reference's rank-indexed flow in the refinement (main.cc:457, quirk Q2), 1 = gathered"""
        s = self.s
        n = s.flatten_dev(flow_img_dev.data_ptr(), self.rows, self.cols, self.K, self.gamma, self.q.data_ptr(), self.u.data_ptr(),
                          self.alpha.data_ptr(), self.alpha_k.data_ptr())
        outp = dict(inlier_idx=self.idx.data_ptr(), inliers=self.inl.data_ptr(), alpha=self.in_alpha.data_ptr(),
                    alpha_k=self.in_alpha_k.data_ptr(), mask=self.mask.data_ptr(), inv_depth=self.rho.data_ptr())
        r = s.ransac_dev(self.q.data_ptr(), self.u.data_ptr(), self.alpha.data_ptr(), self.alpha_k.data_ptr(), n, use_alpha_k, trials,
                         tol, outp, samples=samples, seed=seed, depth_mode=depth_mode)
        m = r["num_inliers"]
        v, w, k = r["v"], r["w"], r["k"]
        inl = self.inl
        ref = None
        if refine:
            ref = s.refine_dev(self.u.data_ptr(), n, m, self.inl.data_ptr(), self.in_alpha.data_ptr(), self.in_alpha_k.data_ptr(),
                               self.idx.data_ptr(), v, w, k, use_alpha_k, flow_index_mode, self.inl_ref.data_ptr())
            v, w, k = ref["v"], ref["w"], ref["k"]
            inl = self.inl_ref
        v, flipped = s.depth_map_dev(inl.data_ptr(), m, v, self.K, self.rows, self.cols, self.depth_map.data_ptr(), None, self.ys.data_ptr())
        s.pose_table_dev(v, w, k, self.gamma, self.rows, self.R.data_ptr(), self.t.data_ptr())
        return dict(n=n, num_inliers=m, v=np.asarray(v), w=np.asarray(w), k=k, flipped=flipped, ransac=r, refine=ref, inliers=inl)
