cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/kt
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt -- python3 $R/tools/bench_configs.py ransac > $R/gpurun_out/kt.log 2>&1
cd $R
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/kt/**/*kernel_trace.csv',recursive=True)[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    for k in ('metric_kernel','count_list','rs_hyp','rs_cand','rs_replay','inlier_hist','refit_kernel','pack_kernel','compact_pairs','mask_flags','exclusive','scan'):
        if k in n: acc[k].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3); break
for k,v in acc.items(): print(k, len(v), [round(x) for x in v[:12]])
PY
