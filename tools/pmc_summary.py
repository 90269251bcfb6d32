"""Summaries of the counter passes of tools/pmc_stages.sh: per kernel of interest the per-launch mean of every counter, the launch
duration from the kernel trace of the same passes, and the derived figures the roofline discussion uses.

    python tools/pmc_summary.py TAG pass_dir [pass_dir ...]      ->  gpurun_out/TAG_pmc_<kernel>.txt, gpurun_out/pmc_traffic.json
Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave; SQ_VALU_MFMA_BUSY_CYCLES counts
cycles; FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies wide streaming reads at half their size (doubled below)."""
import collections
import csv
import glob
import json
import re
import sys

KERNELS = ["normals_kernel", "normals_wave_kernel", "knn_wave_kernel", "knn_tile_kernel", "spfh_tile_kernel", "fpfh_mfma_kernel", "count_kernel", "count_list_kernel", "match_mfma", "match_sweep", "match_tiles", "knn_kernel", "metric_kernel", "plane_kernel",
           "assign_kernel", "pack16_kernel", "rerank_refilter", "init_tables_sparse_kernel", "voxel_accumulate", "filter_flags"]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.match(r"(?:void )?([A-Za-z0-9_:]+(?:<[^(]{0,40}>)?)", n)
    return (m.group(1) if m else n[:60])[:70]


def main():
    tag, dirs = sys.argv[1], sys.argv[2:]
    cnt = collections.defaultdict(lambda: collections.defaultdict(list))     # kernel -> counter -> per-launch values
    dur = collections.defaultdict(list)                                      # kernel -> launch durations (ns), first pass only
    for di, d in enumerate(dirs):
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                cnt[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if di == 0:
            for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
                for r in csv.DictReader(open(f)):
                    dur[short(r["Kernel_Name"])].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    traffic = None
    for want in KERNELS:
        names = sorted(k for k in cnt if want in k)
        if not names:
            continue
        lines = []
        for k in names:
            c = {n: v for n, v in cnt[k].items()}
            nl = max(len(v) for v in c.values())
            mean = {n: sum(v) / len(v) for n, v in c.items()}
            tot = {n: sum(v) for n, v in c.items()}
            d_us = [x / 1e3 for x in dur.get(k, [])]
            lines.append("%s: %d launches per step, duration under the counter pass %s us (sum %.1f)" % (k, nl, " ".join("%.1f" % x for x in d_us[:8]), sum(d_us)))
            for n in sorted(c):
                lines.append("  %-26s per-launch mean %.6g   sum %.6g" % (n, mean[n], tot[n]))
            g = lambda n: tot.get(n, 0.0)
            if g("SQ_WAVE_CYCLES"):
                lines.append("  derived: waves %.0f; wave-cycle split: active %.1f %%, issue-stalled %.1f %%, parked (waitcnt / barrier) %.1f %%, LDS-issue-stall %.1f %%" % (
                    g("SQ_WAVES"), 100 * g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"), 100 * g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"),
                    100 * g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 100 * g("SQ_WAIT_INST_LDS") / g("SQ_WAVE_CYCLES")))
            if g("SQ_BUSY_CU_CYCLES") and g("SQ_WAVE_CYCLES"):
                # SQ_WAVE_CYCLES counts quad-cycles per wave, SQ_BUSY_CU_CYCLES cycles per CU: waves per SIMD = 4 x wave quad-cycles / (4 SIMDs x
                # busy cycles).  (Calibrated on match_mfma, which is resident with exactly 4 waves per SIMD: ratio 3.99.)
                lines.append("  derived: %.2f waves resident per SIMD while the CU is busy" % (g("SQ_WAVE_CYCLES") / g("SQ_BUSY_CU_CYCLES")))
            if g("SQ_INSTS_VALU") and g("SQ_WAVES"):
                lines.append("  derived: per wave %.0f VALU + %.0f SALU + %.0f LDS + %.0f VMEM-read instructions; LDS bank-conflict cycles / LDS active %.1f %%" % (
                    g("SQ_INSTS_VALU") / g("SQ_WAVES"), g("SQ_INSTS_SALU") / g("SQ_WAVES"), g("SQ_INSTS_LDS") / g("SQ_WAVES"), g("SQ_INSTS_VMEM_RD") / g("SQ_WAVES"),
                    100 * g("SQ_LDS_BANK_CONFLICT") / max(1.0, g("SQ_ACTIVE_INST_LDS") * 4)))
            if g("SQ_INSTS_MFMA"):
                lines.append("  derived: %.2f VALU instructions per MFMA; MFMA-busy cycles %.4g" % ((g("SQ_INSTS_VALU") - g("SQ_INSTS_MFMA")) / g("SQ_INSTS_MFMA") if g("SQ_INSTS_VALU") else float("nan"),
                                                                                             g("SQ_VALU_MFMA_BUSY_CYCLES")))
                lines.append("           (SQ_INSTS_VALU counts the MFMAs too: SQ_INSTS_VALU / SQ_INSTS_MFMA = %.2f)" % (g("SQ_INSTS_VALU") / g("SQ_INSTS_MFMA")))
            if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
                fetch, write = 2.0 * 1024.0 * g("FETCH_SIZE"), 1024.0 * g("WRITE_SIZE")
                t = sum(dur.get(k, [])) * 1e-9
                lines.append("  derived: HBM-side traffic per step: fetch %.4g GB (FETCH_SIZE x 2, gfx950) + write %.4g GB = %.4g GB%s" % (
                    fetch / 1e9, write / 1e9, (fetch + write) / 1e9, (" -> %.0f GB/s over the launches' duration" % ((fetch + write) / t / 1e9)) if t > 0 else ""))
            lines.append("")
        text = "\n".join(lines)
        open("gpurun_out/%s_pmc_%s.txt" % (tag, want), "w").write(text)
        if want in ("normals_wave_kernel", "spfh_tile_kernel", "fpfh_mfma_kernel", "count_list_kernel", "match_mfma", "match_sweep", "match_tiles"):
            print(text)
        if want == "match_mfma":
            names = sorted(k for k in cnt if "match_mfma" in k or "match_sweep" in k or "match_tiles" in k)   # all MFMA passes of a step (round 4: the final pass is two kernels)
            # cross-check of the bench line's issued-FLOP count against the hardware's own instruction count (v_mfma_f32_32x32x16_f16 = 32 768 FLOP)
            n_mfma = sum(sum(cnt[k].get("SQ_INSTS_MFMA", [])) for k in names)
            if n_mfma and want == "match_mfma":
                for d in dirs[:1]:
                    try:
                        for line in open(d.rstrip("/") + ".log"):
                            if line.startswith('{"metric"'):
                                rl = json.loads(line)["roofline"]
                                steps = max(1, len(cnt[names[0]].get("SQ_INSTS_MFMA", [])) // max(1, len(names) and 1))
                                per_step = n_mfma * 32768.0 / steps
                                x = ("cross-check: SQ_INSTS_MFMA x 32768 = %.4g FLOP per step (%d profiled step(s)) vs the bench line's mfma_flop_issued %.4g: ratio %.3f"
                                     % (per_step, steps, rl["mfma_flop_issued"], per_step / rl["mfma_flop_issued"]))
                                print(x)
                                open("gpurun_out/%s_pmc_%s.txt" % (tag, want), "a").write(x + "\n")
                                raise StopIteration
                    except (OSError, StopIteration, KeyError):
                        continue
            fs = sum(sum(cnt[k].get("FETCH_SIZE", [])) for k in names)
            ws = sum(sum(cnt[k].get("WRITE_SIZE", [])) for k in names)
            nl = sum(len(cnt[k].get("FETCH_SIZE", [])) for k in names)
            if fs and ws:
                fmt = "f16"
                try:
                    for line in open(dirs[2].rstrip("/") + ".log"):
                        if line.startswith('{"metric"'):
                            of = json.loads(line)["roofline"]["operand_format"]
                            fmt = "f16r" if "K = 96" in of else ("f32" if of.startswith("f32") else "f16")
                except OSError:
                    pass
                traffic = {"kernel": "match_mfma + match_sweep + match_tiles (all masked MFMA passes of one 1M-pt bench step)", "fetch_bytes_corrected": 2.0 * 1024.0 * fs, "write_bytes": 1024.0 * ws,
                           "traffic_bytes": 2.0 * 1024.0 * fs + 1024.0 * ws, "launches": nl, "operand_format": fmt,
                           "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes), tools/pmc_stages.sh; FETCH_SIZE x2 (gfx950)"}
    if traffic:
        json.dump(traffic, open("gpurun_out/pmc_traffic.json", "w"), indent=1)


if __name__ == "__main__":
    main()
