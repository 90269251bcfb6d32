#!/bin/bash
# Per-stage counter evidence over one 1M bench step: four rocprofv3 passes (counters only: --kernel-trace +
# --pmc, one run per group; FETCH_SIZE and WRITE_SIZE need a pass each), then one summary per kernel of interest under
# gpurun_out/<tag>_pmc_<kernel>.txt and the matcher's HBM-side traffic as gpurun_out/pmc_traffic.json (bench.py's roofline.traffic
# once copied to profiles/).
#   bash tools/pmc_stages.sh [tag] [extra bench.py arguments, e.g. --matching cluster]
set -e
TAG=${1:-r3}
shift || true
cd /tmp && export TMPDIR=/tmp
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the root of the repo copy there)}"
R=$GRAFT_REPO_ROOT
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf "$R"/gpurun_out/pmcs_$i
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmcs_$i -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-stage-rooflines --no-matcher-extremes "$@" > $R/gpurun_out/pmcs_$i.log 2>&1
  echo "pmc group $i done"
done
cd $R
python3 tools/pmc_summary.py "$TAG" gpurun_out/pmcs_1 gpurun_out/pmcs_2 gpurun_out/pmcs_3 gpurun_out/pmcs_4
