"""Summarise the last align step of a rocprofv3 kernel trace (gpurun_out/<dir>/**/*kernel_trace.csv): per-kernel totals."""
import csv, glob, collections, re, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
ref = [i for i, r in enumerate(rows) if 'refit_kernel' in r['Kernel_Name']]
step = rows[ref[-2] + 1: ref[-1] + 1]
t0, t1 = int(step[0]['Start_Timestamp']), int(step[-1]['End_Timestamp'])
acc = {}
for r in step:
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).split('(')[0][:60]
    a = acc.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
print('step wall ms %.2f  kernels %d  busy %.2f' % ((t1 - t0) / 1e6, len(step), sum(v[1] for v in acc.values())))
for n, (c, d) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"{n:62s} {c:5d} {d:8.3f}")
