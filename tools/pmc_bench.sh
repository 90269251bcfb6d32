#!/bin/bash
# PMC passes over the 1M bench step for the matcher kernel (counters only: --kernel-trace + --pmc, one run per group)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmcb_$i -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmcb_$i.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmcb_*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'match_mfma' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in acc.items():
            print(d, k, 'n=%d' % len(v), 'sum=%.6g' % sum(v), ' '.join('%.4g' % x for x in v))
PY
