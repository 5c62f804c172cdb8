import importlib.util
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_v1.npz"))


@pytest.fixture(scope="session")
def golden_rectify():
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_rectify_v1.npz"))


@pytest.fixture(scope="session")
def oracle():
    import oracle_py

    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def rsdsfm():
    import rsdsfm as pkg

    return pkg


@pytest.fixture(scope="session")
def big_config(rsdsfm):
    """full-size BASELINE configs are expensive to synthesise (the 3840x2160 flow field takes tens of seconds): one copy per
    session, shared by the tests that need it; callers must not modify the arrays"""
    cache = {}

    def get(cfg, seed=None):
        if (cfg, seed) not in cache:
            cache[(cfg, seed)] = rsdsfm.synth.make_config(cfg, seed=seed)
        return cache[(cfg, seed)]

    return get


@pytest.fixture(scope="session")
def oracle_chain(oracle, big_config):
    """the ORACLE's whole solve of a full-size config (flatten, alpha, RANSAC with the library's sampler, refinement on the gathered
    flow, sign fix, depth map), cached per session: the 3840x2160 chain costs ~30 s of one host core and several tests check
    against it"""
    cache = {}

    def get(cfg, T, tol, seed, accel=False, data_seed=None, arith="reference", flow_mode=1):
        """flow_mode 1 = the gathered flow, 0 = the reference's rank-indexed flow (main.cc:457, quirk Q2); the RANSAC part is shared"""
        key = (cfg, T, tol, seed, accel, data_seed, arith, flow_mode)
        if key not in cache:
            d = big_config(cfg, data_seed)
            rows, K, gamma = d["rows"], d["K"], d["gamma"]
            other = cache.get(key[:-1] + (1 - flow_mode,))
            with oracle.arithmetic(arith):
                if other is not None:
                    q, u, a, ak, ro = other["q"], other["u"], other["a"], other["ak"], other["ransac"]
                else:
                    q, u, qpx, fpx = oracle.flatten(d["flow_img"], *K, gamma)
                    a, ak = oracle.get_alpha(fpx, rows, gamma), oracle.get_alpha_k(qpx, fpx, rows, gamma)
                    ro = oracle.ransac(q, u, a, ak, accel, T, tol, oracle.sample_indices(len(q), T, seed), depth_mode=1)
                refo = oracle.refine(u, ro["inliers"], ro["alpha"], ro["alpha_k"], ro["v"], ro["w"], ro["k"], accel, flow_mode,
                                     ro["inlier_idx"] if flow_mode else None)
                inl, v, flipped = oracle.canonicalize_sign(refo["inliers"], refo["v"])
                dm, xs, ys = oracle.scatter_depth(inl, *K, d["rows"], d["cols"])
            cache[key] = dict(q=q, u=u, a=a, ak=ak, ransac=ro, refine=refo, inliers=inl, v=v, flipped=flipped, depth_map=dm, ys=ys)
        return cache[key]

    return get


GOLDEN_CASES = ["clean_k0", "noisy_k0", "deepflow_k0", "clean_k04", "noisy_k04"]
RECTIFY_CASES = ["k0", "k04"]
