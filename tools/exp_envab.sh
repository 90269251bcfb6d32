#!/bin/bash
# on-box A/B of an environment switch (experiments): bash tools/exp_envab.sh VAR v1 v2 ...  -> ms per pair, stages, sweep statistics per value
for v in "${@:2}"; do
  export $1=$v
  python bench.py --no-cpu-baseline --no-stage-rooflines --no-matcher-extremes --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1=$v:', round(d['ms_per_step'],2), 'mfma', round(r['kernel_ms'],2), 'tested', r.get('coarse_tiles_tested'), 'skipped', r.get('shell_tiles_skipped'), {k: round(x,2) for k,x in d['stage_ms'].items()}, d['result']['n_correspondences'])"
done
