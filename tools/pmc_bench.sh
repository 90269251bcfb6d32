#!/bin/bash
# PMC passes over the 1M bench step for the matcher kernel (counters only: --kernel-trace + --pmc, one run per group).
# Writes gpurun_out/pmc_match_mfma.txt (per-launch values) and gpurun_out/pmc_traffic.json (what bench.py reports as
# roofline.traffic once copied to profiles/).
set -e
cd /tmp && export TMPDIR=/tmp
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the root of the repo copy there)}"
R=$GRAFT_REPO_ROOT
i=0
for grp in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf "$R"/gpurun_out/pmcb_$i
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmcb_$i -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmcb_$i.log 2>&1
  echo "pmc group $i done"
done
cd $R
python3 - <<'PY' | tee gpurun_out/pmc_match_mfma.txt
import csv, glob, collections, json
tot = {}
for d in sorted(glob.glob('gpurun_out/pmcb_*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'match_mfma' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in acc.items():
            tot[k] = v
            print(d, k, 'n=%d' % len(v), 'sum=%.6g' % sum(v), ' '.join('%.4g' % x for x in v))
if 'FETCH_SIZE' in tot and 'WRITE_SIZE' in tot:
    # MI355X_MICROARCH.md (HBM / rocprofv3): FETCH_SIZE and WRITE_SIZE are in KiB-like units of 1024 B here; on gfx950 FETCH_SIZE
    # tallies wide streaming reads at half their size -> doubled
    fetch = 2.0 * 1024.0 * sum(tot['FETCH_SIZE']); write = 1024.0 * sum(tot['WRITE_SIZE'])
    fmt = "f16"
    for line in open('gpurun_out/pmcb_3.log'):
        if line.startswith('{"metric"'):
            of = json.loads(line)["roofline"]["operand_format"]
            fmt = "f16r" if "K = 96" in of else ("f32" if of.startswith("f32") else "f16")
    json.dump({"kernel": "match_mfma (both masked launches of one 1M-pt bench step)", "fetch_bytes_corrected": fetch, "write_bytes": write,
               "traffic_bytes": fetch + write, "launches": len(tot['FETCH_SIZE']), "operand_format": fmt,
               "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes), tools/pmc_bench.sh; FETCH_SIZE x2 (gfx950)"},
              open('gpurun_out/pmc_traffic.json', 'w'), indent=1)
PY
