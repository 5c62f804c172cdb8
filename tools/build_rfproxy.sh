#!/bin/bash
# the TIMING build behind DESIGN section 8's "the pass's loop phase is bound by its memory accesses": refine_rf_kernels.hip with RF_LOADS_ONLY=1
# (loads + the one store of the NP = 6 loop, no arithmetic; results are wrong by construction), linked with the product's other objects
# (run after tools/rebuild.sh) -> rs-aware-differential-sfm_amd/librsdsfm_hip_rfproxy.so (git-ignored; travels with gpurun)
set -e
cd "$(dirname "$0")/../rs-aware-differential-sfm_amd"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DRF_LOADS_ONLY=1 -c csrc/refine_rf_kernels.hip -o build/refine_rf_kernels.proxy.o
objs=$(ls build/*.o | grep -v "\.fused\.o$" | grep -v "/refine_rf_kernels\.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -ldl $objs -o librsdsfm_hip_rfproxy.so
rm -f build/refine_rf_kernels.proxy.o
echo "$PWD/librsdsfm_hip_rfproxy.so"
