set -e
cd lidar-global-registration_amd/csrc
for v in "" "-DEXP_STORE" "-DEXP_NOFLUSHTILE" "-DEXP_STORE -DEXP_NOFLUSHTILE"; do
  rm -f lgr_match.o; make EXP="$v" > /dev/null 2>&1
  cd ../..
  LGR_MATCH_PRUNE=0 python bench.py --points 400000 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/x.log 2>&1
  echo "variant [$v] dense: $(tail -1 gpurun_out/x.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms"])')"
  LGR_MATCH_PRUNE=1 python bench.py --points 400000 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/x.log 2>&1
  echo "variant [$v] pruned: $(tail -1 gpurun_out/x.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms"], d["roofline"]["executed_tile_fraction"])')"
  cd lidar-global-registration_amd/csrc
done
