#!/bin/bash
# kernel-trace of the FPFH stage alone for each library given (paths relative to the repo root): durations of spfh_tile_kernel / fpfh_mfma_kernel
#   bash tools/prof_fpfh.sh TAG lib1 [lib2 ...]      ("" = the in-tree library)
set -e
TAG=$1; shift
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  name=$(echo "${lib:-intree}" | tr '/.' '__')
  if [ -n "$lib" ]; then export LGR_HIP_LIB=$R/$lib; else unset LGR_HIP_LIB; fi
  rm -rf $R/gpurun_out/${TAG}_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_$name -- python3 $R/tools/exp_fpfh_time.py --runs 5 > $R/gpurun_out/${TAG}_$name.log 2>&1
  echo "== ${lib:-in-tree}"
  python3 $R/tools/kstats.py $R/gpurun_out/${TAG}_$name 8 | grep -i "spfh\|fpfh" || true
done
