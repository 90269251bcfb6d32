#!/usr/bin/env python3
"""bench.py -- scan-pair registrations/s on the BASELINE.json workload (synthetic 1M-point pair, 1xMI355X per rank).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one alignPointClouds-equivalent (lgr_align_dev: voxel downsample -> k-NN normals -> FPFH -> brute-force
matching both ways -> match filter -> prerejective RANSAC -> SVD refit) over one synthetic scan pair whose clouds
are already resident in HBM.  Scan pairs shard across ranks (one independent pair per rank, weak scaling); the only
collective is one all-gather of the 96-byte per-pair result record per step (RCCL when N > 1).

Rank 0 prints ONE JSON line with the contract keys plus
  "roofline"     : the dominant kernel (match_mfma, the MFMA distance filter of the matcher; both masked launches of a
                   step): MFMA FLOP issued (224 per computed pair on f16-split operands x M x M x executed tile
                   fraction) / launch duration measured with hipEvents on the launch stream, against the dense f16
                   MFMA peak of MI355X; the algorithmic 69 * Mq * Mt over the same time is given beside it
                   ("effective_tflops_algorithmic"); "traffic" = HBM-side bytes from the committed PMC passes;
  "cpu_baseline" : the CPU oracle ("port": the reference itself needs PCL/OpenCV and cannot be built here) timed on
                   this host on a bounded sample of the same workload, extrapolated linearly where the stage is
                   linear in the sampled dimension (the sample is stated in the object).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "lidar-global-registration_amd"))

MFMA_F32_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
MFMA_F16_PEAK_TFLOPS = 2516.8    # same guide: BF16/FP16 MFMA "~2.5 PF dense" = 16 x the f32-input MFMA rate


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--points", type=int, default=1_000_000, help="points per cloud (BASELINE config 2: 1e6)")
    ap.add_argument("--matching", default="lr", choices=["lr", "cluster", "one_sided"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU baseline sample")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (nccl = RCCL; gloo only to rehearse the N > 1 path on a box with fewer GPUs than ranks)")
    ap.add_argument("--dry-run", action="store_true",
                    help="rank formation + the record all-gather only (no GPU, no HIP library): proves that --gpus N forms N ranks; used by the CPU tests")
    ap.add_argument("--verify", action="store_true",
                    help="after the timed region: check sampled queries of the GPU's 1M x 1M matches against the CPU oracle (parity_sample in the JSON line)")
    ap.add_argument("--verify-queries", type=int, default=4096)
    return ap.parse_args()


def fan_out(args):
    """`python bench.py --gpus N` with N > 1 and no rendezvous in the environment: start the N ranks ourselves, exactly as the
    driver's own command would (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...).
    This parent never imports torch or touches HIP (a process that has initialised the GPU must not exec or fork ranks); it
    relays the children's output (rank 0 prints the one JSON line) and returns their exit code."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def dry_run(args, world, rank):
    """Ranks + the one collective of the path, no GPU: every rank packs a record that only it can produce, the records are
    all-gathered (gloo), and rank 0 checks that it holds one from each of the `--gpus` ranks."""
    import torch
    import torch.distributed as dist
    from lgr_amd import distributed
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    rec = distributed.pack_record(rank, np.eye(4, dtype=np.float32).reshape(16) * (rank + 1), 1, 2**31 - 1 - rank, 1000 + rank, 0.0, 0.0)
    t0 = time.perf_counter()
    for _ in range(max(1, args.steps)):
        allr = distributed.gather_records(torch.from_numpy(rec.view(np.int32).copy())[None], world)
    elapsed = time.perf_counter() - t0
    got = [distributed.unpack_record(r) for r in np.ascontiguousarray(allr.numpy()).view(np.float32)]
    ok = [g["pair_id"] for g in got] == list(range(world)) and all(g["iterations"] == 2**31 - 1 - i for i, g in enumerate(got))
    if rank == 0:
        print(json.dumps({"metric": "scan-pair registrations/sec", "value": None, "unit": "registrations/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / max(1, args.steps),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "none",
                          "dry_run": True, "ranks_seen": [g["pair_id"] for g in got], "records_ok": bool(ok)}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if not ok:
        raise SystemExit(3)


def make_params(capi, pair, matching):
    mid = {"lr": capi.MATCH_LR, "cluster": capi.MATCH_CLUSTER, "one_sided": capi.MATCH_ONE_SIDED}[matching]
    # SURVEY 8d / BASELINE.md config 2 profile (data/tests.yaml values where the yaml sets them)
    return capi.default_params(matching_id=mid, metric_id=capi.METRIC_UNIFORMITY, score_id=capi.SCORE_MSE,
                               feature_radius=0.25, feature_nr_points=352, normal_nr_points=30, bf_block_size=200000,
                               edge_thr_coef=0.95, confidence=0.999, max_iterations=1000000, distance_thr=0.1,
                               vp_src=pair["vp_src"], vp_tgt=pair["vp_tgt"])


def cpu_baseline(pair, gpu_corr, gpu_iterations, args, matching):
    """Oracle timed on the host cores on a bounded sample of the SAME 1M-point pair (see module docstring)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as o
    o.build()
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    o.set_num_threads(cores)
    src, tgt = pair["src"], pair["tgt"]
    n = src.shape[0]
    budget = args.cpu_seconds
    t = {}
    # voxel / radius exactly as the pipeline derives them (include/matching.h:172,230-231)
    r = 0.25
    voxel = float(np.sqrt(np.pi * r * r / 352.0).astype(np.float32))
    t0 = time.time(); ds = o.downsample(src, voxel); t["downsample"] = 2 * (time.time() - t0)
    # normals: a prefix sample of the surface, searched in the full surface (linear in the number of queries)
    ns = min(ds.shape[0], 40000)
    t0 = time.time(); o.normals_knn(ds[:ns], 30, surf=ds, vp=pair["vp_src"]); t["normals"] = 2 * (time.time() - t0) * ds.shape[0] / ns
    dsn = o.normals_knn(ds[: min(ds.shape[0], 60000)], 30, vp=pair["vp_src"])   # oriented normals for the FPFH sample
    # FPFH on a spatially compact sub-scene (surface prefix is in (z,y,x) voxel order, so take a slab of keypoints)
    nk = 20000
    slab = dsn
    kp_sel = src[:nk]
    t0 = time.time(); f_s = o.fpfh(kp_sel, slab, r); dt = time.time() - t0
    # cost model: SPFH is linear in surface points, weighting linear in keypoints; measured together on the slab
    t["fpfh"] = 2 * dt * max(n / nk, ds.shape[0] / slab.shape[0])
    # matching: S sampled queries against the full train set of real FPFH rows (O(M) per query), both directions
    rng = np.random.default_rng(0)
    feat_t = np.tile(f_s[~np.isnan(f_s).any(1)], (n // max(1, (~np.isnan(f_s).any(1)).sum()) + 1, 1))[:n]
    feat_t = (feat_t + rng.normal(0, 0.5, feat_t.shape)).astype(np.float32)
    S = 256
    t0 = time.time(); o.match_bf_subset(feat_t, np.arange(S, dtype=np.int32), feat_t, 200000); dt = time.time() - t0
    S2 = int(max(S, min(8192, S * (0.45 * budget) / max(dt, 1e-3))))
    t0 = time.time(); o.match_bf_subset(feat_t, np.arange(S2, dtype=np.int32), feat_t, 200000); dt = time.time() - t0
    n_dir = 1 if matching == "one_sided" else 2
    t["match"] = n_dir * dt * n / S2
    # filter thresholds (two k=2 density passes over 1M points): prefix sample
    nd = 100000
    t0 = time.time(); o.smoothed_densities(src[:nd], 2); t["filter"] = 2 * (time.time() - t0) * n / nd
    # RANSAC on the correspondences the GPU produced for this pair, Philox schedule, first batches, scaled by iterations
    corr = np.zeros(gpu_corr.shape[0], o.CORR_DTYPE)
    corr["query"] = gpu_corr["index_query"]; corr["match"] = gpu_corr["index_match"]
    corr["distance"] = gpu_corr["distance"]; corr["threshold"] = gpu_corr["threshold"]
    it_sample = 16384
    p = o.default_params(rng_mode=o.RNG_PHILOX, metric_id=o.METRIC_UNIFORMITY, max_iterations=it_sample, batch_size=16384)
    t0 = time.time(); res, _ = o.ransac(src, tgt, corr, p); dt = time.time() - t0
    t["ransac_sample_iters"] = res.iterations
    t["ransac"] = dt * max(1.0, gpu_iterations / max(res.iterations, 1))   # linear in the iterations actually needed
    total = sum(v for k, v in t.items() if k not in ("ransac_sample_iters",))
    return t, total, cores, S2


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(fan_out(args))              # before anything imports torch / initialises HIP in this process
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher formed WORLD_SIZE={world} ranks")
    if args.dry_run:
        return dry_run(args, world, rank)
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path is the product and has no CPU fallback")
    if args.backend == "gloo":
        local = local % torch.cuda.device_count()      # rehearsal: several ranks may share one card
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    torch.cuda.set_device(local)
    from lgr_amd import capi, synthetic, distributed
    ctx = capi.Context(local)

    pair = synthetic.make_pair(args.points, seed=synthetic.SEED + rank)
    params = make_params(capi, pair, args.matching)
    src = torch.from_numpy(pair["src"]).cuda(local)
    tgt = torch.from_numpy(pair["tgt"]).cuda(local)
    coll_dev = f"cuda:{local}" if args.backend == "nccl" else "cpu"
    record = torch.zeros((1, distributed.RECORD_FLOATS), dtype=torch.int32, device=coll_dev)   # 96-byte per-pair record (4-byte words)

    def step():
        res = ctx.align(src, tgt, params)
        rec = distributed.pack_record(rank, res.transformation, res.converged, res.iterations, res.n_inliers, res.time_cs, res.time_te)
        record.copy_(torch.from_numpy(rec.view(np.int32))[None])
        distributed.gather_records(record, world)     # the single collective of the path (RCCL all-gather when N > 1)
        return res

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    res = None
    for _ in range(args.warmup):
        res = step()
    barrier()
    t0 = time.perf_counter()
    kernel_ms, stage_ms, work, coarse = [], [], [], []
    for _ in range(args.steps):
        res = step()
        kernel_ms.append(ctx.match_kernel_ms())
        mstats = ctx.match_stats()
        mstats["refilter_pairs_ab"], mstats["refilter_pairs_ba"] = ctx.match_pairs()
        work.append(ctx.match_work())
        coarse.append(ctx.match_coarse())
        stage_ms.append(list(res.stage_ms)[:7])
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        n_dir = 1 if args.matching == "one_sided" else 2
        m = args.points
        # matching = 69 * Mq * Mt FLOP (2*33 MAC + 3 for norm add / compare); one launch serves both directions
        alg_flop = 69.0 * m * m
        k_ms = float(np.mean(kernel_ms))
        # Roofline of the dominant kernel (both masked MFMA launches of a step).  `achieved` counts the MFMA FLOP really
        # issued: the exact bound-based skipping computes only `executed` of the M x M tiles, and per (query, train) pair
        # the kernel issues 7 x v_mfma_f32_32x32x16_f16 steps on two-term f16 splits of the f32 operands (K = 112:
        # 224 FLOP) or 17 x v_mfma_f32_32x32x2_f32 (K = 34: 68 FLOP); `peak` is the dense MFMA peak of that operand
        # type.  The algorithmic 69 FLOP per pair over the same time is reported beside it.
        executed = float(np.mean(work))
        fmt = ctx.match_format()
        flop_per_pair = {"f16": 224.0, "f16r": 192.0, "f32": 68.0}[fmt]
        peak = MFMA_F32_PEAK_TFLOPS if fmt == "f32" else MFMA_F16_PEAK_TFLOPS
        # tiles the kernel abandons after their first two MFMA steps (coarse rejection, rotated format) issue 64 of the 192 FLOP
        c_tested, c_abandoned = [float(x) for x in np.mean(np.array(coarse), 0)]
        issued = flop_per_pair * m * m * executed - (c_abandoned * 1024.0 * (192.0 - 64.0) if fmt == "f16r" else 0.0)
        achieved = issued / (k_ms * 1e-3) / 1e12
        effective = alg_flop / (k_ms * 1e-3) / 1e12
        # HBM-side bytes of the same kernel (both launches of one step) from the committed PMC passes (tools/pmc_bench.sh:
        # separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950);
        # only reported for the configuration it was collected on
        traffic = None
        try:
            pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if m == 1_000_000 and args.matching == "lr" and fmt == pt.get("operand_format", "f16"):
                traffic = float(pt["traffic_bytes"])
        except Exception:
            traffic = None
        T = res.matrix()
        err = float(np.abs(T.astype(np.float64) - pair["T_gt"]).max())
        out = {
            "metric": "scan-pair registrations/sec", "value": world * args.steps / elapsed, "unit": "registrations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: synthetic 1M-pt pair, random SE(3) + Gaussian noise (5 mm), FPFH r=0.25 m",
                       "points_per_cloud": m, "pairs_per_step": world, "matching": args.matching, "metric_id": "uniformity",
                       "bf_block_size": 200000, "max_iterations": 1000000, "parallelism": f"pairs sharded over {world} GPU(s)"},
            "roofline": {"kernel": "match_mfma<both directions> (all masked passes of one step)", "bound": "mfma", "achieved": achieved, "peak": peak,
                         "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                         "operand_format": {"f16": "f16 two-term splits, f32 accumulate, K = 112 (224 MFMA FLOP/pair)",
                                            "f16r": "f16 two-term splits of 30 Helmert coordinates, f32 accumulate, K = 96 (192 MFMA FLOP/pair)",
                                            "f32": "f32 (68 MFMA FLOP/pair)"}[fmt],
                         "kernel_ms": k_ms, "executed_tile_fraction": executed, "coarse_tiles_tested": c_tested, "coarse_tiles_abandoned": c_abandoned,
                         "mfma_flop_issued": issued, "effective_tflops_algorithmic": effective,
                         "rerank": mstats, "algorithmic_flop_per_launch": alg_flop, "directions_per_launch": n_dir},
            "stage_ms": dict(zip(["downsample", "normals", "fpfh", "match", "filter", "ransac", "refit"],
                                 [float(x) for x in np.mean(np.array(stage_ms), 0)])),
            "result": {"converged": int(res.converged), "iterations": int(res.iterations), "n_correspondences": int(res.n_correspondences),
                       "n_inliers": int(res.n_inliers), "max_abs_err_vs_gt": err},
        }
        if not args.no_cpu_baseline and world == 1:   # the CPU baseline is timed on rank 0 of the 1-GPU run only
            corr = ctx.correspondences(src, tgt, params).cpu().numpy().view(capi.CORR_DTYPE).reshape(-1)
            t, total, cores, S2 = cpu_baseline(pair, corr, int(res.iterations), args, args.matching)
            out["cpu_baseline"] = {
                "value": 1.0 / total, "unit": "registrations/s", "cores": cores, "kind": "port",
                "sample": (f"same 1M-pt pair; downsample full; normals 40k-query sample; FPFH 20k keypoints on a 60k-point slab; "
                           f"matching {S2} sampled queries x 1M train rows x {n_dir} direction(s), scaled by M/S; density filter 100k prefix; "
                           f"RANSAC first {t['ransac_sample_iters']} iterations of the Philox schedule, scaled to the {int(res.iterations)} the run needed"),
                "seconds_per_pair_estimate": total, "stage_seconds": {k: float(v) for k, v in t.items()},
            }
            out["speedup_vs_cpu_baseline"] = out["value"] / (world * out["cpu_baseline"]["value"])
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
