for b in 0 2 1; do
  LGR_MATCH_BOX=$b python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('box $b', round(d['ms_per_step'],2), 'kernel', round(d['roofline']['kernel_ms'],2), 'tiles', round(d['roofline']['executed_tile_fraction'],4), 'match', round(d['stage_ms']['match'],2), d['result']['n_correspondences'], d['result']['n_inliers'])"
done
