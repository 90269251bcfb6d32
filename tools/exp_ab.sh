#!/bin/bash
# A/B on one box: the bench line's ms per pair / match stage / MFMA passes for the in-tree library and each variant, alternating, ROUNDS times
#   bash tools/exp_ab.sh ROUNDS build/var_x/liblgr_hip.so [...]
R=${GRAFT_REPO_ROOT:-.}
N=$1; shift
for r in $(seq 1 $N); do
  for lib in "" "$@"; do
    if [ -n "$lib" ]; then export LGR_HIP_LIB=$R/$lib; else unset LGR_HIP_LIB; fi
    python bench.py --no-cpu-baseline --no-matcher-extremes --no-stage-rooflines --steps 12 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('${lib:-in-tree}', round(d['ms_per_step'],2), 'match', round(d['stage_ms']['match'],2), 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'fpfh', round(d['stage_ms']['fpfh'],2))"
  done
done
