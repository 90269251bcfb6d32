b() { python bench.py "$@" --steps 8 --warmup 2 --no-cpu-baseline --no-stage-rooflines 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']; r=d['roofline']; print('%.2f normals %.2f fpfh %.2f match %.2f kernel_ms %s' % (d['ms_per_step'], s['normals'], s['fpfh'], s['match'], r.get('kernel_ms')))"; }
