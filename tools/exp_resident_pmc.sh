#!/bin/bash
# counters of the resident RANSAC kernel against the launch chain's kernels on config 4's stress (same work, same results)
G1="GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
G2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_FLAT"
export LGR_RANSAC_SCHEDULE=2
bash tools/pmc_cmd.sh rs_resident res "python3 tools/bench_configs.py ransac" "$G1" "$G2"
export LGR_RANSAC_SCHEDULE=1
for k in rs_hyp count_list metric_kernel; do
  if [ $k = rs_hyp ]; then bash tools/pmc_cmd.sh $k chn "python3 tools/bench_configs.py ransac" "$G1" "$G2"; else
  python3 - $k <<'PY'
import csv, glob, collections, sys
K = sys.argv[1]
print("---", K)
for d in sorted(glob.glob('gpurun_out/chn_pmc*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if K in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in sorted(acc.items()):
            print('%-34s n=%d mean=%.6g sum=%.6g' % (k, len(v), sum(v) / len(v), sum(v)))
PY
  fi
done
