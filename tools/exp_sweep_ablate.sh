#!/bin/bash
# timing ablations of match_sweep (wrong results, kernel time only): build/var_sw_<name>/liblgr_hip.so
set -e
cd "$(dirname "$0")/.."
CSRC=lidar-global-registration_amd/csrc
for v in nodma nocompute; do
  case $v in
    nodma) FL="-DLGR_EXP_SWEEP_NODMA" ;;
    nocompute) FL="-DLGR_EXP_SWEEP_NOCOMPUTE" ;;
  esac
  d=build/var_sw_$v
  mkdir -p $d
  cp $CSRC/*.o $d/
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-result $FL -c $CSRC/lgr_match.hip -o $d/lgr_match.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $d/liblgr_hip.so $d/*.o
  find $d -name '*.o' -delete
  echo "built $d/liblgr_hip.so"
done
