# on-box experiment: grid target (points per cell) of the k-NN stages vs normals / density time
set -e
cd lidar-global-registration_amd/csrc
for t in 4 8 12 16 24; do
  sed -i "s/lgr_grid_build(ctx, WS_GRID_A, S, ns, 0.f, [0-9.]*f, &g)/lgr_grid_build(ctx, WS_GRID_A, S, ns, 0.f, $t.f, \&g)/" lgr_features.hip
  make > /dev/null 2>&1
  cd ../..
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('target $t:', d['ms_per_step'], d['stage_ms']['normals'])"
  cd lidar-global-registration_amd/csrc
done
