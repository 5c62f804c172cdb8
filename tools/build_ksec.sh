#!/bin/bash
# the diagnostic build tools/k_sections.py needs: minimal9_kernels.hip with RSDSFM_K_SECTIONS=1, linked with the product's other objects
# (run after tools/rebuild.sh) -> rs-aware-differential-sfm_amd/librsdsfm_hip_ksec.so (git-ignored; travels with gpurun)
set -e
cd "$(dirname "$0")/../rs-aware-differential-sfm_amd"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DRSDSFM_K_SECTIONS=1 -c csrc/minimal9_kernels.hip -o build/minimal9_kernels.ksec.o
objs=$(ls build/*.o | grep -v "\.fused\.o$" | grep -v "/minimal9_kernels\.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -ldl $objs -o librsdsfm_hip_ksec.so
rm -f build/minimal9_kernels.ksec.o
echo "$PWD/librsdsfm_hip_ksec.so"
