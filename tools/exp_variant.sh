#!/bin/bash
# A variant library with extra compile flags for ONE translation unit: build/var_<name>/liblgr_hip.so
#   bash tools/exp_variant.sh NAME lgr_match.hip "-DLGR_EXP_SWEEP_CHAINMIN"
# On the GPU box: LGR_HIP_LIB=$GRAFT_REPO_ROOT/build/var_NAME/liblgr_hip.so python bench.py ...
set -e
cd "$(dirname "$0")/.."
CSRC=lidar-global-registration_amd/csrc
NAME=$1; TU=$2; FL=$3
d=build/var_$NAME
mkdir -p $d
cp $CSRC/*.o $d/
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-result $FL -c $CSRC/$TU -o $d/${TU%.hip}.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $d/liblgr_hip.so $d/*.o
find $d -name '*.o' -delete
echo "built $d/liblgr_hip.so"
