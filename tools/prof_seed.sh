#!/bin/bash
# kernel statistics of the matcher stage alone on synthetic pairs of other seeds: bash tools/prof_seed.sh TAG N seed [seed ...]
set -e
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${TAG}_trace
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace -- python3 $R/tools/exp_seed_scan.py "$@" > $R/gpurun_out/${TAG}_trace.log 2>&1
cd $R
python3 tools/kstats.py gpurun_out/${TAG}_trace 30 > gpurun_out/${TAG}_kstats.txt
python3 tools/find_gaps.py gpurun_out/${TAG}_trace 3 > gpurun_out/${TAG}_gaps.txt
cat gpurun_out/${TAG}_gaps.txt
