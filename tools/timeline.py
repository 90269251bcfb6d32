"""Compact timeline of the last step of a rocprofv3 kernel trace: python tools/timeline.py gpurun_out/<dir> [min_us]
Kernels shorter than min_us (default 60) are folded into runs "n x name".  Columns: start ms, duration us, stream/queue, name."""
import csv, glob, re, sys
d = sys.argv[1]; min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); m = re.match(r"(?:void )?([A-Za-z0-9_:]+(?:<[^(]{0,24})?)", n); return (m.group(1) if m else n)[:60]
starts = [i for i, r in enumerate(rows) if "voxel_keys" in r["Kernel_Name"]]
seg = rows[starts[-2]:]
t0 = int(seg[0]["Start_Timestamp"])
qs = {}
fold = {}
def flush(q):
    if q in fold and fold[q][2]:
        s, e, n, names, _ = fold[q]
        top = max(set(names), key=names.count)
        print(f"{(s - t0) / 1e6:8.3f}  {(e - s) / 1e3:8.1f}  q{q}  {n} small kernels ({top} ...)  busy {fold[q][4] / 1e3:.1f} us")
    fold.pop(q, None)
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"]); q = qs.setdefault(r["Queue_Id"], len(qs)); n = short(r["Kernel_Name"])
    if (e - s) / 1e3 < min_us:
        if q not in fold: fold[q] = [s, e, 0, [], 0]
        fold[q][1] = e; fold[q][2] += 1; fold[q][3].append(n); fold[q][4] += e - s
        continue
    flush(q)
    print(f"{(s - t0) / 1e6:8.3f}  {(e - s) / 1e3:8.1f}  q{q}  {n}")
for q in list(fold): flush(q)
