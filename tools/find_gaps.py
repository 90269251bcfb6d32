"""Largest idle gaps between consecutive kernels of a rocprofv3 kernel trace: python tools/find_gaps.py gpurun_out/<dir> [min_ms]"""
import csv, glob, re, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); m = re.match(r"(?:void )?([A-Za-z0-9_:]+(?:<[^(]{0,24})?)", n); return (m.group(1) if m else n)[:48]
prev_end, prev = int(rows[0]["End_Timestamp"]), rows[0]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[1:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if (s - prev_end) / 1e6 >= min_ms:
        print(f"at {(s - t0) / 1e6:10.2f} ms: idle {(s - prev_end) / 1e6:7.2f} ms between {short(prev['Kernel_Name'])} and {short(r['Kernel_Name'])}")
    if e > prev_end: prev_end, prev = e, r
