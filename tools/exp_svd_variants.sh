#!/bin/bash
# Builds variant libraries for the multi-context determinism experiment into build/var_*/liblgr_hip.so:
#   var_div      lgr_svd3 with the round-3 lane-divergent `continue` (no scratch anywhere)
#   var_scratch  the round-4 branch-free lgr_svd3 + six dead scratch stores in normals_finish (what the round-3 build carried)
#   var_both     both
# Run on the GPU box:  LGR_HIP_LIB=build/var_div/liblgr_hip.so python tools/exp_concurrent_stages.py --align-only --threads 3 --rounds 50 --distinct 1
set -e
cd "$(dirname "$0")/.."
CSRC=lidar-global-registration_amd/csrc
for v in div scratch both; do
  case $v in
    div) FL="-DLGR_EXP_SVD_DIVERGENT" ;;
    scratch) FL="-DLGR_EXP_NORMALS_SCRATCH" ;;
    both) FL="-DLGR_EXP_SVD_DIVERGENT -DLGR_EXP_NORMALS_SCRATCH" ;;
  esac
  d=build/var_$v
  mkdir -p $d
  cp $CSRC/*.o $d/
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-result $FL -c $CSRC/lgr_features.hip -o $d/lgr_features.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $d/liblgr_hip.so $d/*.o
  rm $d/*.o
  echo "built $d/liblgr_hip.so"
done
