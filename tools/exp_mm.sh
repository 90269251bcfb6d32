# on-box experiment: rebuild lgr_match.o with variant flags and time the 1M bench (BENCH=1) and / or the matcher alone (dense 400k)
cd lidar-global-registration_amd/csrc
for v in "$@"; do
  rm -f lgr_match.o; make EXP="$v" > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  echo "== $v"
  (cd ../.. && if [ -z "$NOPROBE" ]; then python tools/probe_match.py 400000 2>&1 | tail -1 | sed 's/stats=.*kernel_ms/kernel_ms/'; fi; if [ -n "$BENCH" ]; then python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench ms', round(d['ms_per_step'],1), 'kernel', round(d['roofline']['kernel_ms'],1), 'tiles', round(d['roofline']['executed_tile_fraction'],4), 'match', round(d['stage_ms']['match'],1))"; fi)
done
rm -f lgr_match.o; make > /dev/null 2>&1
