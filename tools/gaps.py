"""Idle time of the GPU inside the last step of a rocprofv3 kernel trace (all streams merged): python tools/gaps.py gpurun_out/<dir> [top_n]"""
import csv, glob, re, sys
d = sys.argv[1]; top = int(sys.argv[2]) if len(sys.argv) > 2 else 20
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); m = re.match(r"(?:void )?([A-Za-z0-9_:]+)", n); return (m.group(1) if m else n)[:44]
starts = [i for i, r in enumerate(rows) if "voxel_keys" in r["Kernel_Name"]]
seg = rows[starts[-2]:]          # the last step = from its first voxel_keys launch (two clouds -> two launches per step)
t0 = int(seg[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in seg)
ivs = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in seg)
busy = 0; cur_s, cur_e = ivs[0][0], ivs[0][1]; gaps = []; last_name = ivs[0][2]
for s, e, n in ivs[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, last_name, n)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    if e >= cur_e: last_name = n
busy += cur_e - cur_s
print(f"last step: wall {(t1 - t0) / 1e6:.2f} ms, GPU busy (union of kernels) {busy / 1e6:.2f} ms, sum of kernel durations {sum(e - s for s, e, _ in ivs) / 1e6:.2f} ms, {len(ivs)} launches, idle {sum(g for g, _, _ in gaps) / 1e6:.2f} ms in {len(gaps)} gaps")
for g, a, b in sorted(gaps, reverse=True)[:top]:
    print(f"{g / 1e3:8.1f} us  after {a:44s} before {b}")
