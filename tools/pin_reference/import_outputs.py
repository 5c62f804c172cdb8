"""Step 3 of the pinning kit: the harness' outputs (one .pin per case + versions.txt) packed into tests/golden/reference_v1.npz --
the fixture tests/test_reference_golden.py looks for.  Commit that file: it is DATA produced by the reference (inputs we committed,
outputs of the reference's own functions), not reference source.

    python tools/pin_reference/import_outputs.py [harness_out_dir]    (default: tools/pin_reference/build/outputs)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import pinio  # noqa: E402
from export_inputs import CASES  # noqa: E402


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "build", "outputs")
    out = {}
    for case in CASES:
        for k, v in pinio.read(os.path.join(src, case + ".pin")).items():
            out[case + "/" + k] = v
    vers = open(os.path.join(src, "versions.txt")).read()
    out["_meta/versions"] = np.frombuffer(vers.encode(), dtype=np.uint8)
    dst = os.path.join(ROOT, "tests", "golden", "reference_v1.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, "(%d arrays)" % len(out))
    print(vers)


if __name__ == "__main__":
    main()
