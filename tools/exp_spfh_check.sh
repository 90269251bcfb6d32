#!/bin/bash
# builds build/var_spfhcheck/liblgr_hip.so: the in-tree library with lgr_features.hip compiled -DLGR_SPFH_CHECK (tools/exp_spfh_check.py)
set -e
cd "$(dirname "$0")/.."
CSRC=lidar-global-registration_amd/csrc
d=build/var_spfhcheck
mkdir -p $d
cp $CSRC/*.o $d/
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-result -DLGR_SPFH_CHECK -c $CSRC/lgr_features.hip -o $d/lgr_features.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $d/liblgr_hip.so $d/*.o
rm $d/*.o
echo "built $d/liblgr_hip.so"
