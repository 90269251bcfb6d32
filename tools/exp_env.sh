# on-box experiment: bench under different matcher options (one per argument, e.g. "near=48" or "coarse_rejection=0": fields of
# lgr_match_options, include/lgr.h); BENCH_ARGS adds bench options
for v in "$@"; do
  python bench.py --no-cpu-baseline --steps 3 --warmup 1 --match-opt "$v" $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v:', round(d['ms_per_step'],2), 'kernel', round(d['roofline']['kernel_ms'],2), 'tiles', round(d['roofline']['executed_tile_fraction'],4), {k: round(x,1) for k,x in d['stage_ms'].items()}, d['result']['n_correspondences'])"
done
