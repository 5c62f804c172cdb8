"""Step 1 of the pinning kit: the committed synthetic inputs (tests/golden/golden_v1.npz: five 64x48 cases with their injected
sample sets) as one .pin file per case for pin_harness.  Nothing of the reference is read or written here.

    python tools/pin_reference/export_inputs.py [out_dir]      (default: tools/pin_reference/build/inputs)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import pinio  # noqa: E402

CASES = ["clean_k0", "noisy_k0", "deepflow_k0", "clean_k04", "noisy_k04"]
RANSAC_TOL = 0.05  # main.cc:305


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "build", "inputs")
    os.makedirs(out, exist_ok=True)
    g = np.load(os.path.join(ROOT, "tests", "golden", "golden_v1.npz"))
    for case in CASES:
        a = {k: g[case + "/" + k] for k in ("q", "u", "alpha", "alpha_k", "samples")}
        a["use_k"] = np.array([int(g[case + "/use_k"])], dtype=np.int32)
        a["tolerance"] = np.array([RANSAC_TOL])
        a["samples"] = a["samples"].astype(np.int32)
        pinio.write(os.path.join(out, case + ".pin"), a)
        print("wrote", os.path.join(out, case + ".pin"), {k: v.shape for k, v in a.items()})


if __name__ == "__main__":
    main()
