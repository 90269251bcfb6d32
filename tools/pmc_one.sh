#!/bin/bash
# counters of the kernels whose name contains $1 over one 1M bench step: rocprofv3 --pmc passes (counters + kernel trace only)
#   bash tools/pmc_one.sh KERNEL TAG "COUNTER GROUP 1" ["COUNTER GROUP 2" ...]
set -e
K=$1; TAG=$2; shift 2
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/${TAG}_pmc$i
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/${TAG}_pmc$i -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-stage-rooflines --no-matcher-extremes > $R/gpurun_out/${TAG}_pmc$i.log 2>&1 || echo "pass $i failed"
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
            print('%-28s n=%d sum=%.6g' % (k, len(v), sum(v)))
PY
