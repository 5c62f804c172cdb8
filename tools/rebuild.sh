#!/bin/bash
# rebuild what is stale (both libraries); prints the path of the product library
cd "$(dirname "$0")/.." && python rs-aware-differential-sfm_amd/build.py --stale "$@" 2>&1 | grep -v "^/opt/rocm/bin/hipcc" | tail -40
