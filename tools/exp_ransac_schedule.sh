#!/bin/bash
# RANSAC stage under the two schedules (launch chain / resident kernel): the bench pair (lr and cluster) and config 4's stress, alternating
#   bash tools/exp_ransac_schedule.sh [ROUNDS]
N=${1:-2}
for r in $(seq 1 $N); do
  for sch in chain resident; do
    for m in lr cluster; do
      python bench.py --no-cpu-baseline --no-matcher-extremes --no-stage-rooflines --steps 10 --warmup 2 --matching $m --ransac-schedule $sch 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$sch', '$m', 'ms/pair', round(d['ms_per_step'],2), 'ransac', round(d['stage_ms']['ransac'],3), 'iters', d['result']['iterations'], 'inl', d['result']['n_inliers'])"
    done
    if [ $sch = chain ]; then export LGR_RANSAC_SCHEDULE=1; else export LGR_RANSAC_SCHEDULE=2; fi
    python tools/bench_configs.py ransac 2>/dev/null
    unset LGR_RANSAC_SCHEDULE
  done
done
