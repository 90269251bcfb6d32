"""Prints a rocprofv3 kernel_stats.csv with short kernel names (the C++ signatures of templated kernels run to kilobytes).
    python tools/kstats.py gpurun_out/<dir> [top_n]"""
import csv, glob, re, sys
d = sys.argv[1]; top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
f = sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.match(r"(?:void )?([A-Za-z0-9_:]+(?:<[^(]{0,40}>)?)", n)
    s = m.group(1) if m else n[:60]
    if "rocprim" in n:
        k = re.search(r"detail::(\w+)", n[n.find("trampoline_kernel") + 10:] if "trampoline_kernel" in n else n)
        s = "rocprim:" + (re.findall(r"(radix_sort_\w+|merge_sort_\w+|scan_impl|lookback\w+|partition\w+|transform\w+|reduce\w+|histogram\w+|onesweep\w+)", n) or ["?"])[0]
    return s[:70]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
agg = {}
for r in rows:
    k = short(r["Name"]); a = agg.setdefault(k, [0, 0.0]); a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"])
print(f"{f}: total kernel time {tot / 1e6:.2f} ms")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{t / 1e6:9.3f} ms {100 * t / tot:5.1f}%  calls {c:5d}  avg {t / c / 1e3:9.1f} us  {k}")
