#!/bin/bash
# seed scan of the matcher stage under values of an environment switch: bash tools/exp_seed_env.sh VAR "v1 v2" N seed [seed ...]
VAR=$1; VALS=$2; shift 2
for v in $VALS; do
  export $VAR=$v
  echo "== $VAR=$v"
  REPS=3 python tools/exp_seed_scan.py "$@" 2>&1 | grep -E "^seed" | cut -c1-200
done
