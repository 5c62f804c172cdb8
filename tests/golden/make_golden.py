#!/usr/bin/env python3
"""Generates tests/golden/*.npz -- the fixtures that pin the CPU oracle and the HIP path.

The reference (ThomasZiegler/RS-aware-differential-SfM) ships no golden vectors and cannot be built here
(needs Ceres / Eigen / OpenCV / Boost), so these fixtures come from an INDEPENDENT numpy/scipy transcription
of the same algorithm (numpy.linalg.svd / eigh / eigvals, a dense full-Jacobian Levenberg-Marquardt instead
of the Schur-eliminated one) run in the build container only.  Inputs come from the package's analytic
generator.  Run:  python tests/golden/make_golden.py     (writes next to this file)

The numpy transcription follows /root/reference/src/minimal.cc:36-197, :255-275 and
nonlinearRefinement.cc:32-52, :109-180, :183-252 and the Ceres 1.14 trust-region loop described in
DESIGN.md; it does not import the oracle or the HIP library.
"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def _load_synth():
    spec = importlib.util.spec_from_file_location("synth", os.path.join(ROOT, "rs-aware-differential-sfm_amd", "synth.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


synth = _load_synth()


# ------------------------------------------------------------------------------------------------------
# numpy transcription of minimal::calculateVelocities (minimal.cc:36-177)
# ------------------------------------------------------------------------------------------------------
def Ry(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def Rz(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def calculate_velocities_np(q, u, alpha, alpha_k, use_alpha_k, k_sign_mode=0):
    x, y, ux, uy = q[:, 0], q[:, 1], u[:, 0], u[:, 1]
    Z = np.stack([-uy, ux, uy * x - ux * y, x * x, 2 * x * y, 2 * x, y * y, 2 * y, np.ones(9)], axis=1)
    k = 0.0
    if use_alpha_k:
        a_inv = np.linalg.inv(Z[:3, :3])
        efhj, dg, bc = Z[3:, 3:], Z[3:, :3], Z[:3, 3:]
        P = np.diag(alpha[3:]) @ efhj - dg @ a_inv @ np.diag(alpha[:3]) @ bc
        Pk = np.diag(alpha_k[3:]) @ efhj - dg @ a_inv @ np.diag(alpha_k[:3]) @ bc
        ev = np.linalg.eigvals(P @ np.linalg.inv(Pk))
        k = np.inf
        for e in ev:
            if abs(e.imag) < 1e-5 and abs(e.real) < abs(k):
                k = e.real
        if k_sign_mode == 1:
            k = -k
        beta = (alpha + k * alpha_k) * (2.0 / (2.0 + k))
    else:
        beta = alpha
    Z = Z.copy()
    Z[:, 3:] *= beta[:, None]
    _, sv, Vt = np.linalg.svd(Z)
    e = Vt[8]
    e = e / np.linalg.norm(e[:3])
    v0 = e[:3].copy()
    S = np.array([[e[3], e[4], e[5]], [e[4], e[6], e[7]], [e[5], e[7], e[8]]])
    lam, v1 = np.linalg.eigh(S)
    v1 = v1[:, ::-1].copy()  # swap columns 0 and 2
    sigma = np.array([(2 * lam[2] + lam[1] - lam[0]) / 3, (lam[2] + 2 * lam[1] + lam[0]) / 3, (-lam[2] + lam[1] + 2 * lam[0]) / 3])
    lmb = sigma[0] - sigma[2]
    theta = 0.0 if lmb < 1e-6 else np.arccos(-sigma[1] / lmb)
    V_ = v1 @ Ry((theta - np.pi) / 2).T
    U_ = -V_ @ Ry(theta)
    sig1 = np.diag([1.0, 1.0, 0.0])
    rz = [Rz(np.pi / 2), Rz(-np.pi / 2)]
    vee = lambda M: np.array([M[2, 1], M[0, 2], M[1, 0]])
    cands = [vee(B @ R @ sig1 @ B.T) for B in (V_, U_) for R in rz]
    idx = int(np.argmax([c @ v0 for c in cands]))
    B = U_ if idx < 2 else V_
    w = vee(B @ rz[idx % 2] @ (lmb * sig1) @ B.T)
    return w, v0, k, sv


# ------------------------------------------------------------------------------------------------------
# per-pixel model
# ------------------------------------------------------------------------------------------------------
def beta_of(alpha, alpha_k, k):
    return (2.0 / (2.0 + k)) * (alpha + k * alpha_k)


def residuals_np(q, u, alpha, alpha_k, v, w, k, rho):
    """r = u - predicted (nonlinearRefinement.cc:32-52), shape (n, 2)."""
    x, y = q[:, 0], q[:, 1]
    b = beta_of(alpha, alpha_k, k)
    p0 = -b * (rho * (x * v[2] - v[0]) + x * y * w[0] - (1 + x * x) * w[1] + y * w[2])
    p1 = -b * (rho * (y * v[2] - v[1]) + (1 + y * y) * w[0] - x * y * w[1] - x * w[2])
    return np.stack([u[:, 0] - p0, u[:, 1] - p1], axis=1)


def jac_rho_np(q, alpha, alpha_k, v, k):
    b = beta_of(alpha, alpha_k, k)
    return np.stack([b * (q[:, 0] * v[2] - v[0]), b * (q[:, 1] * v[2] - v[1])], axis=1)


def closed_form_depth_np(q, u, alpha, alpha_k, v, w, k):
    J = jac_rho_np(q, alpha, alpha_k, v, k)
    r1 = residuals_np(q, u, alpha, alpha_k, v, w, k, np.ones(len(q)))
    h = (J * J).sum(1)
    g = (J * r1).sum(1)
    return np.where(h > 0, 1.0 - g / np.where(h > 0, h, 1.0), 1.0)


def score_np(q, u, alpha, alpha_k, v, w, k, rho, tol):
    x, y = q[:, 0], q[:, 1]
    b = (alpha + k * alpha_k) * (2.0 / (2.0 + k))
    e0 = b * ((v[0] - x * v[2]) * rho + (-x * y) * w[0] + (1 + x * x) * w[1] - y * w[2]) - u[:, 0]
    e1 = b * ((v[1] - y * v[2]) * rho - (1 + y * y) * w[0] + (x * y) * w[1] + x * w[2]) - u[:, 1]
    err = np.sqrt(e0 * e0 + e1 * e1)
    mask = err < tol
    return int(mask.sum()), float(err[mask].sum()), mask, err


# ------------------------------------------------------------------------------------------------------
# generic dense Ceres-1.14-style trust-region LM (TrustRegionMinimizer + LevenbergMarquardtStrategy)
# ------------------------------------------------------------------------------------------------------
def ceres_lm_dense(fun, jac, x0, max_iter=50, trace=None):
    """fun(x)->residual vector, jac(x)->dense Jacobian.  Returns x, summary dict."""
    x = x0.copy()
    r = fun(x)
    Jm = jac(x)
    cost = 0.5 * r @ r
    scale = 1.0 / (1.0 + np.sqrt((Jm * Jm).sum(0)))
    g = Jm.T @ r
    gmax = np.abs(g).max() if len(g) else 0.0
    x_norm = np.linalg.norm(x)
    radius, dec = 1e4, 2.0
    it, ns, nu, invalid = 0, 0, 0, 0
    term = None
    init_cost = cost
    if gmax <= 1e-10:
        term = 0
    while term is None:
        if it >= max_iter:
            term = 3
            break
        if radius <= 1e-32:
            term = 5
            break
        it += 1
        Js = Jm * scale
        diag = np.clip((Js * Js).sum(0), 1e-6, 1e32)
        D2 = np.sqrt(diag / radius) ** 2
        A = Js.T @ Js + np.diag(D2)
        try:
            y = np.linalg.solve(A, Js.T @ r)
        except np.linalg.LinAlgError:
            y = None
        if y is not None:
            step = -y
            m = Js @ step
            model_change = -(m @ (r + m / 2.0))
        if y is None or not (model_change > 0):
            nu += 1
            invalid += 1
            if invalid >= 5:
                term = 4
                break
            radius *= 0.5
            continue
        invalid = 0
        cand = x + step * scale
        rc = fun(cand)
        ccost = 0.5 * rc @ rc
        step_norm = np.linalg.norm(x - cand)
        if trace is not None:
            trace.append((it, radius, cost, ccost, model_change, step_norm, x_norm))
        if step_norm <= 1e-8 * (x_norm + 1e-8):
            term = 1
            break
        if abs(cost - ccost) <= 1e-6 * cost:
            term = 2
            break
        rel = (cost - ccost) / model_change
        if rel > 1e-3:
            x = cand
            r = rc
            Jm = jac(x)
            cost = ccost
            g = Jm.T @ r
            gmax = np.abs(g).max()
            x_norm = np.linalg.norm(x)
            f = max(1.0 / 3.0, 1.0 - (2.0 * rel - 1.0) ** 3)
            radius = min(1e16, radius / f)
            dec = 2.0
            ns += 1
            if gmax <= 1e-10:
                term = 0
        else:
            nu += 1
            radius /= dec
            dec *= 2.0
    return x, dict(num_iterations=it, num_successful_steps=ns, num_unsuccessful_steps=nu, termination=term,
                   initial_cost=init_cost, final_cost=cost, final_radius=radius)


def lm_depth_np(q, u, alpha, alpha_k, v, w, k):
    """estimateInverseDepths (nonlinearRefinement.cc:109-180) through the dense LM above, exploiting that the
    normal equations are diagonal (each rho_i is its own block) -- implemented with diagonal algebra."""
    n = len(q)
    J = jac_rho_np(q, alpha, alpha_k, v, k)  # constant
    x = np.ones(n)
    r = residuals_np(q, u, alpha, alpha_k, v, w, k, x)
    cost = 0.5 * (r * r).sum()
    s = 1.0 / (1.0 + np.sqrt((J * J).sum(1)))
    gmax = np.abs((J * r).sum(1)).max()
    x_norm = np.linalg.norm(x)
    radius, dec = 1e4, 2.0
    it = ns = nu = invalid = 0
    term = None
    init_cost = cost
    trace = []
    if gmax <= 1e-10:
        term = 0
    while term is None:
        if it >= 50:
            term = 3
            break
        if radius <= 1e-32:
            term = 5
            break
        it += 1
        Js = J * s[:, None]
        ht = (Js * Js).sum(1)
        D = np.sqrt(np.clip(ht, 1e-6, 1e32) / radius)
        step = -((Js * r).sum(1) / (ht + D * D))
        m = Js * step[:, None]
        model_change = -((m * (r + m / 2.0)).sum())
        if not (model_change > 0):
            nu += 1
            invalid += 1
            if invalid >= 5:
                term = 4
                break
            radius *= 0.5
            continue
        invalid = 0
        cand = x + step * s
        rc = residuals_np(q, u, alpha, alpha_k, v, w, k, cand)
        ccost = 0.5 * (rc * rc).sum()
        step_norm = np.linalg.norm(x - cand)
        trace.append((it, radius, cost, ccost, model_change, step_norm, x_norm))
        if step_norm <= 1e-8 * (x_norm + 1e-8):
            term = 1
            break
        if abs(cost - ccost) <= 1e-6 * cost:
            term = 2
            break
        rel = (cost - ccost) / model_change
        if rel > 1e-3:
            x, r, cost = cand, rc, ccost
            gmax = np.abs((J * r).sum(1)).max()
            x_norm = np.linalg.norm(x)
            radius = min(1e16, radius / max(1.0 / 3.0, 1.0 - (2.0 * rel - 1.0) ** 3))
            dec = 2.0
            ns += 1
            if gmax <= 1e-10:
                term = 0
        else:
            nu += 1
            radius /= dec
            dec *= 2.0
    return x, dict(num_iterations=it, num_successful_steps=ns, num_unsuccessful_steps=nu, termination=term,
                   initial_cost=init_cost, final_cost=cost, final_radius=radius), np.array(trace)


def refine_np(uu, inl, alpha, alpha_k, v, w, k, const_acc):
    """nonLinearRefinement (nonlinearRefinement.cc:183-252) as a DENSE full-Jacobian LM (no Schur)."""
    m = len(inl)
    q = inl[:, :2]
    npar = 7 if const_acc else 6
    x0 = np.concatenate([v, w, [k] if const_acc else [], 1.0 / inl[:, 2]])

    def unpack(x):
        kk = x[6] if const_acc else k
        return x[:3], x[3:6], kk, x[npar:]

    def fun(x):
        vv, ww, kk, rho = unpack(x)
        return residuals_np(q, uu, alpha, alpha_k, vv, ww, kk, rho).reshape(-1)

    def jac(x):
        vv, ww, kk, rho = unpack(x)
        xx, yy = q[:, 0], q[:, 1]
        b = beta_of(alpha, alpha_k, kk)
        a0, a1 = xx * vv[2] - vv[0], yy * vv[2] - vv[1]
        in0 = rho * a0 + xx * yy * ww[0] - (1 + xx * xx) * ww[1] + yy * ww[2]
        in1 = rho * a1 + (1 + yy * yy) * ww[0] - xx * yy * ww[1] - xx * ww[2]
        J = np.zeros((2 * m, npar + m))
        br = b * rho
        J[0::2, 0] = -br
        J[1::2, 1] = -br
        J[0::2, 2] = br * xx
        J[1::2, 2] = br * yy
        J[0::2, 3] = b * xx * yy
        J[1::2, 3] = b * (1 + yy * yy)
        J[0::2, 4] = -b * (1 + xx * xx)
        J[1::2, 4] = -b * xx * yy
        J[0::2, 5] = b * yy
        J[1::2, 5] = -b * xx
        if const_acc:
            db = 2.0 * (2.0 * alpha_k - alpha) / (2.0 + kk) ** 2
            J[0::2, 6] = db * in0
            J[1::2, 6] = db * in1
        idx = np.arange(m)
        J[2 * idx, npar + idx] = b * a0
        J[2 * idx + 1, npar + idx] = b * a1
        return J

    # finite-difference check of the analytic Jacobian (guards the transcription itself)
    xt = x0 * (1 + 1e-3)
    Jn = jac(xt)
    for c in list(range(npar)) + [npar, npar + m - 1]:
        hstep = 1e-7 * max(1.0, abs(xt[c]))
        xp, xm = xt.copy(), xt.copy()
        xp[c] += hstep
        xm[c] -= hstep
        fd = (fun(xp) - fun(xm)) / (2 * hstep)
        assert np.allclose(fd, Jn[:, c], rtol=1e-5, atol=1e-7), ("jacobian column", c)

    trace = []
    x, sm = ceres_lm_dense(fun, jac, x0, trace=trace)
    vv, ww, kk, rho = unpack(x)
    out = inl.copy()
    out[:, 2] = 1.0 / rho
    return dict(v=vv.copy(), w=ww.copy(), k=float(kk), inliers=out, summary=sm, trace=np.array(trace))


def sm_arr(sm):
    return np.array([sm["num_iterations"], sm["num_successful_steps"], sm["num_unsuccessful_steps"], sm["termination"],
                     sm["initial_cost"], sm["final_cost"], sm["final_radius"]], dtype=np.float64)


def main():
    out = {}
    rng = np.random.default_rng(20181212)
    cases = [
        ("clean_k0", dict(noise_px=0.0, outliers=0.0, k=0.0, use_k=False)),
        ("noisy_k0", dict(noise_px=0.1, outliers=0.0, k=0.0, use_k=False)),
        ("deepflow_k0", dict(noise_px=0.3, outliers=0.10, k=0.0, use_k=False)),
        ("clean_k04", dict(noise_px=0.0, outliers=0.0, k=0.4, use_k=True)),
        ("noisy_k04", dict(noise_px=0.05, outliers=0.0, k=0.4, use_k=True)),
    ]
    rows, cols = 48, 64
    K = tuple(np.array(synth.INTRINSICS["galaxy_vga"]) * 0.1)
    v_true = np.array([0.03, 0.02, 0.01])
    w_true = np.array([0.002, -0.003, np.deg2rad(0.5)])
    for name, c in cases:
        flow, truth = synth.make_flow(rows, cols, K, v_true, w_true, c["k"], 0.8, c["noise_px"], c["outliers"], seed=0x5EED0000 + len(out))
        q, u, alpha, alpha_k, pix = synth.flatten_numpy(flow, K, 0.8)
        n = len(q)
        T = 10
        samples = np.stack([rng.choice(n, 9, replace=False) for _ in range(T)]).astype(np.int32)
        W, V, KK, SV = [], [], [], []
        for t in range(T):
            s = samples[t]
            w, v, k, sv = calculate_velocities_np(q[s], u[s], alpha[s], alpha_k[s], c["use_k"])
            W.append(w), V.append(v), KK.append(k), SV.append(sv)
        W, V, KK, SV = map(np.array, (W, V, KK, SV))
        # depth + score for each hypothesis, both depth modes
        rho_cf, rho_lm, lm_sm, counts_cf, counts_lm, errs_cf, errs_lm = [], [], [], [], [], [], []
        traces = []
        for t in range(T):
            r0 = closed_form_depth_np(q, u, alpha, alpha_k, V[t], W[t], KK[t])
            r1, sm, tr = lm_depth_np(q, u, alpha, alpha_k, V[t], W[t], KK[t])
            rho_cf.append(r0), rho_lm.append(r1), lm_sm.append(sm_arr(sm))
            c0, e0, _, _ = score_np(q, u, alpha, alpha_k, V[t], W[t], KK[t], r0, 0.05)
            c1, e1, _, _ = score_np(q, u, alpha, alpha_k, V[t], W[t], KK[t], r1, 0.05)
            counts_cf.append(c0), counts_lm.append(c1), errs_cf.append(e0), errs_lm.append(e1)
            if t < 3:
                traces.append(tr[:6] if len(tr) >= 6 else np.pad(tr, ((0, 6 - len(tr)), (0, 0))))
        best = 0
        for t in range(1, T):
            if counts_lm[t] > counts_lm[best] or (counts_lm[t] == counts_lm[best] and errs_lm[t] < errs_lm[best]):
                best = t
        _, _, mask, _ = score_np(q, u, alpha, alpha_k, V[best], W[best], KK[best], rho_lm[best], 0.05)
        inl = np.stack([q[mask, 0], q[mask, 1], 1.0 / rho_lm[best][mask]], axis=1)
        idx = np.nonzero(mask)[0]
        ref_compat = refine_np(u[: len(idx)], inl, alpha[mask], alpha_k[mask], V[best], W[best], KK[best], c["use_k"])
        ref_gather = refine_np(u[idx], inl, alpha[mask], alpha_k[mask], V[best], W[best], KK[best], c["use_k"])
        pre = name + "/"
        out.update({
            pre + "flow_img": flow, pre + "K": np.array(K), pre + "gamma": np.array(0.8), pre + "k_true": np.array(c["k"]),
            pre + "use_k": np.array(int(c["use_k"])),
            pre + "q": q, pre + "u": u, pre + "alpha": alpha, pre + "alpha_k": alpha_k, pre + "samples": samples,
            pre + "hyp_w": W, pre + "hyp_v": V, pre + "hyp_k": KK, pre + "hyp_sv": SV,
            pre + "rho_cf": np.array(rho_cf[:3]), pre + "rho_lm": np.array(rho_lm[:3]), pre + "lm_summary": np.array(lm_sm),
            pre + "lm_trace": np.array(traces),
            pre + "count_cf": np.array(counts_cf), pre + "count_lm": np.array(counts_lm),
            pre + "err_cf": np.array(errs_cf), pre + "err_lm": np.array(errs_lm), pre + "best": np.array(best),
            pre + "best_mask": mask.astype(np.uint8),
            pre + "ref_compat_v": ref_compat["v"], pre + "ref_compat_w": ref_compat["w"], pre + "ref_compat_k": np.array(ref_compat["k"]),
            pre + "ref_compat_z": ref_compat["inliers"][:, 2], pre + "ref_compat_summary": sm_arr(ref_compat["summary"]),
            pre + "ref_gather_v": ref_gather["v"], pre + "ref_gather_w": ref_gather["w"], pre + "ref_gather_k": np.array(ref_gather["k"]),
            pre + "ref_gather_z": ref_gather["inliers"][:, 2], pre + "ref_gather_summary": sm_arr(ref_gather["summary"]),
            pre + "v_true": v_true, pre + "w_true": w_true, pre + "Z_true": truth["Z"],
        })
        print(name, "n", n, "best", best, "count_lm", counts_lm, "steps", [int(s[1]) for s in lm_sm],
              "term", [int(s[3]) for s in lm_sm], "refine", ref_gather["summary"]["num_iterations"],
              ref_gather["summary"]["termination"])
    # third-party pieces: random matrices with numpy answers
    Zs = rng.standard_normal((6, 9, 9))
    out["linalg/svd_in"] = Zs
    out["linalg/svd_sv"] = np.array([np.linalg.svd(z)[1] for z in Zs])
    out["linalg/svd_vlast"] = np.array([np.linalg.svd(z)[2][8] for z in Zs])
    Gs = rng.standard_normal((8, 6, 6))
    out["linalg/eig_in"] = Gs
    out["linalg/eig_vals_sorted"] = np.array([np.sort_complex(np.linalg.eigvals(g)) for g in Gs])
    Ss = rng.standard_normal((8, 3, 3))
    Ss = Ss + np.transpose(Ss, (0, 2, 1))
    out["linalg/sym_in"] = Ss
    out["linalg/sym_vals"] = np.array([np.linalg.eigvalsh(s) for s in Ss])
    np.savez_compressed(os.path.join(HERE, "golden_v1.npz"), **out)
    print("wrote", os.path.join(HERE, "golden_v1.npz"), os.path.getsize(os.path.join(HERE, "golden_v1.npz")) // 1024, "KiB")


if __name__ == "__main__":
    sys.exit(main())
