"""Fills DESIGN.md's @PLACEHOLDERS@ (or refreshes nothing if there are none) from the committed evidence: profiles/r5_bench_default.json,
profiles/pmc_traffic.json, profiles/r5_job_tests156.json.  python tools/fill_design.py [file]"""
import json, sys
f = sys.argv[1] if len(sys.argv) > 1 else "DESIGN.md"
d = json.loads(open("profiles/r5_bench_default.json").read().strip().splitlines()[-1])
j = json.loads(open("profiles/r5_job_tests156.json").read().strip().splitlines()[-1])
t = json.load(open("profiles/pmc_traffic.json"))
st = d["stage_ms"]; r = d["roofline"]
rep = {"@MS@": "%.1f" % d["ms_per_step"], "@RPS@": "%.1f" % d["value"], "@ST0@": "%.2f" % st["downsample"], "@ST1@": "%.2f" % st["normals"], "@ST2@": "%.2f" % st["fpfh"],
       "@ST3@": "%.1f" % st["match"], "@ST5@": "%.2f" % st["ransac"], "@ACH@": "%.0f" % r["achieved"], "@FRAC@": "%.2f" % r["frac"], "@KMS@": "%.2f" % r["kernel_ms"],
       "@TRAF@": "%.1f" % (t["traffic_bytes"] / 1e9), "@JOB@": "%.1f" % j["value"], "@JOBS@": "%.2f" % j["job"]["sum_alignment_seconds"]}
s = open(f).read()
for k, v in rep.items():
    s = s.replace(k, v)
open(f, "w").write(s)
print({k: v for k, v in rep.items()})
