#!/bin/bash
# PMC passes for the matcher kernel (counters only: --kernel-trace + --pmc, separate runs per counter group)
set -e
cd /tmp && export TMPDIR=/tmp
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the root of the repo copy there)}"
R=$GRAFT_REPO_ROOT
for grp in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/tools/probe_match.py $1 > $R/gpurun_out/pmc_$tag.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc_*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'match_mfma' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in acc.items():
            print(d, k, 'n=%d' % len(v), 'mean=%.6g' % (sum(v) / len(v)))
PY
