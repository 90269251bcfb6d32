#!/bin/bash
# PMC counters of one kernel (name substring $1) over a 1M bench step: two rocprofv3 --pmc passes (counters only)
set -e
K=$1
cd /tmp && export TMPDIR=/tmp
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the root of the repo copy there)}"
R=$GRAFT_REPO_ROOT
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1))
  rm -rf "$R"/gpurun_out/pmck_$i
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmck_$i -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmck_$i.log 2>&1 || echo "pass $i failed"
done
cd $R
python3 - "$K" <<'PY'
import csv, glob, collections, sys
K = sys.argv[1]
for d in sorted(glob.glob('gpurun_out/pmck_*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if K in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in acc.items():
            print(k, 'n=%d' % len(v), 'mean=%.5g' % (sum(v) / len(v)))
PY
