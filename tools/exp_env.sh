# on-box experiment: bench under different environment settings (one per argument, e.g. "LGR_MATCH_NEAR=32"); BENCH_ARGS adds bench options
for v in "$@"; do
  env $v python bench.py --no-cpu-baseline --steps 3 --warmup 1 $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v:', round(d['ms_per_step'],2), 'kernel', round(d['roofline']['kernel_ms'],2), 'tiles', round(d['roofline']['executed_tile_fraction'],4), {k: round(x,1) for k,x in d['stage_ms'].items()}, d['result']['n_correspondences'])"
done
