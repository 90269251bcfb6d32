# on-box experiment: rebuild lgr_match with different pass schedules and report kernel time / executed fraction
set -e
cd lidar-global-registration_amd/csrc
run() {
  sed -i "s/^constexpr int NEAR_T = [0-9]*;/constexpr int NEAR_T = $1;/" lgr_match.hip
  rm -f lgr_match.o; make EXP="-DLGR_PRUNE_BETAS=$2" > /dev/null 2>&1
  cd ../..
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/x.log 2>&1
  echo "NEAR_T=$1 betas=$2: $(tail -1 gpurun_out/x.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["executed_tile_fraction"], d["stage_ms"]["match"])')"
  cd lidar-global-registration_amd/csrc
}
run 64 "1.0f"
run 32 "0.5f,1.0f"
run 16 "0.4f,0.7f,1.0f"
run 32 "0.6f,1.0f"
run 16 "0.5f,1.0f"
