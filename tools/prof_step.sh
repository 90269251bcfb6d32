#!/bin/bash
# kernel-trace profile of bench steps on the GPU box: per-kernel statistics (short names) -> gpurun_out/<tag>_kstats.txt
#   bash tools/prof_step.sh TAG [bench.py arguments]
set -e
TAG=${1:-prof}
shift || true
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${TAG}_trace
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stage-rooflines --no-matcher-extremes "$@" > $R/gpurun_out/${TAG}_trace.log 2>&1
cd $R
python3 tools/kstats.py gpurun_out/${TAG}_trace 40 > gpurun_out/${TAG}_kstats.txt
cat gpurun_out/${TAG}_kstats.txt
