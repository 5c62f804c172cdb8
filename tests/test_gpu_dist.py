"""GPU: row-tiling invariance of the HIP path through the multi-GPU driver (dist.py) -- 1 vs 2 vs 5 logical shards
on one device give the same LM decisions and bit-identical depths, and match the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(rsdsfm, torch, d, v, w, k, nshards, mode):
    dev = torch.device("cuda", 0)
    n = len(d["alpha"])
    bounds, per = rsdsfm.dist.shard_bounds(n, nshards)
    # the contexts share torch's (non-default) current stream, so the driver's torch ops and the stage kernels are ordered
    stream = torch.cuda.Stream(dev)
    with torch.cuda.stream(stream):
        solvers = [rsdsfm.Solver(0, stream=stream.cuda_stream) for _ in range(nshards)]
        stages = []
        for s, (i0, i1) in zip(solvers, bounds):
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a[i0:i1])).to(dev)
            stages.append(rsdsfm.dist.HipDepthStage(s, t(d["q"]), t(d["u"]), t(d["alpha"]), t(d["alpha_k"]), v, w, k, torch))
        drv = rsdsfm.dist.TiledDepthSolve(stages, n, per, torch, None)
        rho, sm = drv.solve(mode)
        for s in solvers:
            s.synchronize()
        out = rho.cpu().numpy()
    for s in solvers:
        s.close()
    return out, sm


@pytest.mark.parametrize("cfg,noise", [(1, False), (3, True)])
def test_tiling_invariance_and_oracle(oracle, rsdsfm, cfg, noise):
    import torch

    d = rsdsfm.synth.make_config(cfg, rows=150, cols=200)
    t = d["truth"]
    v = t["v"] / np.linalg.norm(t["v"])
    w, k = t["w"], 0.0
    for mode in (0, 1):
        rho1, sm1 = _run(rsdsfm, torch, d, v, w, k, 1, mode)
        rho_o, sm_o = oracle.estimate_inverse_depths(d["q"], d["u"], v, w, k, d["alpha"], d["alpha_k"], mode=mode)
        assert np.allclose(rho1, rho_o, rtol=1e-9, atol=1e-13)
        for ns in (2, 5):
            rho_s, sm_s = _run(rsdsfm, torch, d, v, w, k, ns, mode)
            assert np.array_equal(rho_s, rho1)
            if mode == 1:
                for key in ("num_iterations", "num_successful_steps", "termination"):
                    assert sm_s[key] == sm1[key] == sm_o[key]
