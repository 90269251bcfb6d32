#!/bin/bash
# Variant libraries of the FPFH weighting kernel into build/var_f<W>_<R>/liblgr_hip.so: each argument "W:R" =
# __launch_bounds__(64, W) (waves per SIMD the register allocation aims at) : groups per round (4, 8 = rounds of 8 then 4, 2 / 82 = a tail of 2).
# On the GPU box:  bash tools/prof_fpfh.sh TAG build/var_f5_8/liblgr_hip.so ...
set -e
cd "$(dirname "$0")/.."
CSRC=lidar-global-registration_amd/csrc
for v in "$@"; do
  W=${v%%:*}; R=${v##*:}
  d=build/var_f${W}_$R
  mkdir -p $d
  cp $CSRC/*.o $d/
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-result -DLGR_EXP_FPFH_WAVES=$W -DLGR_EXP_FPFH_ROUND=$R -c $CSRC/lgr_features.hip -o $d/lgr_features.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $d/liblgr_hip.so $d/*.o
  python3 - <<PY
import sys, tempfile
sys.path.insert(0, "tools")
import isa_hazards as h
with tempfile.TemporaryDirectory() as td:
    co = h.code_object("$d/lgr_features.o", td)
    for k, r in h.resources(co).items():
        if "fpfh_mfma" in k: print("var_f${W}_$R:", r)
PY
  rm $d/*.o
done
