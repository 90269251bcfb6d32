#!/bin/bash
# counters of the kernels whose name contains $1 over an arbitrary command: one rocprofv3 --pmc pass (counters + kernel trace only) per group
#   bash tools/pmc_cmd.sh KERNEL TAG "python3 tools/exp_fpfh_time.py --runs 1" "COUNTER GROUP 1" ["COUNTER GROUP 2" ...]
# (the command's program must come first: no env / bash -c hops under the profiler; relative paths are resolved against the repo root)
set -e
K=$1; TAG=$2; CMD=$3; shift 3
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/${TAG}_pmc$i
  ( cd $R && rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/${TAG}_pmc$i -- $CMD > $R/gpurun_out/${TAG}_pmc$i.log 2>&1 ) || echo "pass $i failed: $grp"
done
cd $R
python3 - "$K" "$TAG" <<'PY'
import csv, glob, collections, sys
K, TAG = sys.argv[1], sys.argv[2]
for d in sorted(glob.glob('gpurun_out/%s_pmc*/' % TAG)):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if K in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in sorted(acc.items()):
            print('%-34s n=%d mean=%.6g' % (k, len(v), sum(v) / len(v)))
PY
