#!/bin/bash
# The round's evidence on ONE box, final build: bash tools/final_evidence.sh a|b TAG   (two gpurun calls: each fits the 20-minute limit)
#   a: kernel-trace profile of bench steps (+ per-kernel summary, timeline, gaps), the four PMC passes, the default bench line
#   b: the cluster / PCL-arithmetic bench lines, the other BASELINE configs, configs[2]'s job on one GPU
# Everything lands under gpurun_out/<TAG>_*; copy what is to be judged into profiles/.
set -e
PART=$1; TAG=${2:-r5_z}
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
R=$GRAFT_REPO_ROOT
cd $R
if [ "$PART" = a ]; then
  bash tools/prof_step.sh $TAG > /dev/null
  python3 tools/kstats.py gpurun_out/${TAG}_trace 60 > gpurun_out/${TAG}_kernel_stats_summary.txt
  python3 tools/timeline.py gpurun_out/${TAG}_trace > gpurun_out/${TAG}_timeline_last_step.txt
  python3 tools/gaps.py gpurun_out/${TAG}_trace > gpurun_out/${TAG}_gaps_last_step.txt
  cp "$(ls gpurun_out/${TAG}_trace/*/*kernel_stats.csv | head -1)" gpurun_out/${TAG}_kernel_stats.csv
  echo "profile done"
  bash tools/pmc_stages.sh $TAG > gpurun_out/${TAG}_pmc_stages.log 2>&1
  echo "pmc done"
  python3 bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
  echo "bench done"
else
  python3 bench.py --matching cluster --no-cpu-baseline --no-matcher-extremes > gpurun_out/${TAG}_bench_cluster.json 2> /dev/null
  echo "cluster done"
  python3 bench.py --arithmetic pcl --no-cpu-baseline --no-matcher-extremes > gpurun_out/${TAG}_bench_pcl.json 2> /dev/null
  echo "pcl done"
  python3 tools/bench_configs.py ransac gror features5m iss1m plane1m > gpurun_out/${TAG}_other_configs.txt 2> /dev/null
  echo "configs done"
  python3 bench.py --job tests156 > gpurun_out/${TAG}_job_tests156.json 2> /dev/null
  echo "job done"
fi
