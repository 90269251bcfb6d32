#!/bin/bash
# per-kernel time of one tools/bench_configs.py configuration (kernel trace): bash tools/exp_config_trace.sh gror
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/kt
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt -- python3 $R/tools/bench_configs.py "$@" > $R/gpurun_out/kt.log 2>&1
cd $R
python3 tools/kstats.py gpurun_out/kt 14
tail -2 gpurun_out/kt.log
