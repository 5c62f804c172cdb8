#!/bin/sh
# The whole kit in one go: inputs -> build -> run the reference -> tests/golden/reference_v1.npz -> the tests that read it.
#   REFERENCE_DIR=/path/to/RS-aware-differential-SfM sh tools/pin_reference/run.sh [extra cmake arguments]
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
: "${REFERENCE_DIR:?set REFERENCE_DIR to a checkout of ThomasZiegler/RS-aware-differential-SfM}"
python3 "$HERE/export_inputs.py" "$HERE/build/inputs"
cmake -S "$HERE" -B "$HERE/build" -DREFERENCE_DIR="$REFERENCE_DIR" "$@"
cmake --build "$HERE/build"
mkdir -p "$HERE/build/outputs"
"$HERE/build/pin_harness" "$HERE/build/inputs" "$HERE/build/outputs" clean_k0 noisy_k0 deepflow_k0 clean_k04 noisy_k04
# the boundary too: the drop-in mirror against REAL Eigen, the reference's call sites unchanged (compile + link only; recorded beside the versions)
if cmake --build "$HERE/build" --target mirror_check mirror_single_run > "$HERE/build/outputs/mirror_check.log" 2>&1 && "$HERE/build/mirror_check" >> "$HERE/build/outputs/mirror_check.log" 2>&1; then
  echo "mirror_check: ok" | tee -a "$HERE/build/outputs/versions.txt"
else
  echo "mirror_check: FAILED (see $HERE/build/outputs/mirror_check.log)" | tee -a "$HERE/build/outputs/versions.txt"
fi
python3 "$HERE/import_outputs.py" "$HERE/build/outputs"
cd "$ROOT" && python3 -m pytest tests/test_reference_golden.py -q -m "not gpu" -rs
echo "on a box with an MI355X:  python3 -m pytest tests/test_reference_golden.py -q -m gpu -rs"
