#!/usr/bin/env python3
"""bench.py -- scan-pair registrations/s on the BASELINE.json workload (synthetic 1M-point pair, 1xMI355X per rank).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one alignPointClouds-equivalent (lgr_align_dev: voxel downsample -> k-NN normals -> FPFH -> brute-force
matching both ways -> match filter -> prerejective RANSAC -> SVD refit) over one synthetic scan pair whose clouds
are already resident in HBM.  Scan pairs shard across ranks (one independent pair per rank, weak scaling); the only
collective is one all-gather of the 96-byte per-pair result record per step (RCCL when N > 1).

Rank 0 prints ONE JSON line with the contract keys plus
  "roofline"     : the dominant kernels (the MFMA distance filter of the matcher: match_mfma for pass 0, match_sweep + match_tiles for the
                   final pass -- one hipEvent pair around each pass; both masked passes of a
                   step): MFMA FLOP issued (224 per computed pair on f16-split operands x M x M x executed tile
                   fraction) / launch duration measured with hipEvents on the launch stream, against the dense f16
                   MFMA peak of MI355X; the algorithmic 69 * Mq * Mt over the same time is given beside it
                   ("effective_tflops_algorithmic"); "traffic" = HBM-side bytes from the committed PMC passes;
  "roofline_stages": per stage of the path (downsample, normals, fpfh, match, filter, ransac): SURVEY 8(d)'s algorithmic bytes and
                   FLOP over the stage's time when run alone, as fractions of the HBM and fp32 peaks;
  "cpu_baseline" : the CPU oracle ("port": the reference itself needs PCL/OpenCV and cannot be built here) timed on
                   this host's cores on the same pair: every stage in full except the brute-force matcher, which runs a
                   bounded sample of queries (real FPFH rows) against all train rows and is scaled by M/S;
  "parity_sample": what those oracle runs say about the timed HIP path (outside the timed region): per stage the number of
                   compared values and of bit mismatches (each oracle stage is fed with the HIP output of the stage
                   before), the sampled matcher queries (index and distance bits), and whether the staged chain equals
                   the timed one-call pipeline.  `--verify` forces this leg even with --no-cpu-baseline.
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
    ap.add_argument("--pairs", type=int, default=0,
                    help="with --dry-run: shard this many scan pairs over the ranks through distributed.run_pairs (the job shape of data/tests.yaml: "
                         "156 pairs over 8 ranks = shards of 20 / 19) and check the gathered records on rank 0")
    ap.add_argument("--verify", action="store_true",
                    help="after the timed region: check sampled queries of the GPU's 1M x 1M matches against the CPU oracle (parity_sample in the JSON line)")
    ap.add_argument("--verify-queries", type=int, default=4096)
    ap.add_argument("--no-stage-rooflines", action="store_true", help="skip the stand-alone stage timings behind `roofline_stages` (profiling runs)")
    ap.add_argument("--no-matcher-extremes", action="store_true",
                    help="skip `matcher_extremes` (the dense schedule on the same pair and the production schedule on structureless rows, outside the timed region)")
    ap.add_argument("--force-collective", action="store_true",
                    help="with --gpus 1: still form a (one-rank) process group on --backend and run the path's all-gather / barrier / all-reduce "
                         "through it, i.e. execute the RCCL calls of the N > 1 path on a one-GPU box")
    ap.add_argument("--single-context", action="store_true",
                    help="lgr_ctx_options.helper_contexts = 0: no helper host threads / streams (for hosts with fewer than 3 cores per rank)")
    ap.add_argument("--job", default=None, choices=["tests156"],
                    help="instead of the headline step: BASELINE configs[2] as a JOB -- 156 synthetic pairs with a 1e5 .. 1e6 size mix (the shape of "
                         "data/tests.yaml, src/main.cpp:384-407) through distributed.run_pairs on this rank's GPU; pairs/s, ms per pair by size, and the "
                         "8-rank makespan each sharding policy would have from the measured per-pair times")
    ap.add_argument("--job-pairs", type=int, default=156)
    ap.add_argument("--policy", default="lpt", choices=["round_robin", "lpt"], help="sharding policy of --job when N > 1")
    ap.add_argument("--ransac-schedule", default="default", choices=["default", "chain", "resident"],
                    help="lgr_ctx_options.ransac_schedule: how the RANSAC loop is driven (never changes results; include/lgr.h)")
    ap.add_argument("--arithmetic", default="fast", choices=["fast", "pcl"],
                    help="lgr_ctx_options.arithmetic: fast (default) or PCL's own FPFH weighting order and rounding steps (include/lgr.h)")
    ap.add_argument("--match-opt", action="append", default=[], metavar="FIELD=VALUE",
                    help="lgr_match_options override for ablations / profiles (e.g. coarse_rejection=0); never changes results")
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


def _rendezvous_defaults(world):
    """MASTER_ADDR always; for a one-rank group formed without a launcher (--force-collective) also a free port and the rank variables"""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world == 1:
        import socket
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        os.environ.setdefault("MASTER_PORT", str(port))
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")


def dry_run(args, world, rank):
    """Ranks + the one collective of the path, no GPU: every rank packs a record that only it can produce, the records are
    all-gathered (gloo), and rank 0 checks that it holds one from each of the `--gpus` ranks."""
    import torch
    import torch.distributed as dist
    from lgr_amd import distributed
    use_pg = world > 1 or args.force_collective
    if use_pg:
        _rendezvous_defaults(world)
        dist.init_process_group("gloo")
    rec = distributed.pack_record(rank, np.eye(4, dtype=np.float32).reshape(16) * (rank + 1), 1, 2**31 - 1 - rank, 1000 + rank, 0.0, 0.0)
    t0 = time.perf_counter()
    for _ in range(max(1, args.steps)):
        allr = distributed.gather_records(torch.from_numpy(rec.view(np.int32).copy())[None], world, force=use_pg)
    elapsed = time.perf_counter() - t0
    got = [distributed.unpack_record(r) for r in np.ascontiguousarray(allr.numpy()).view(np.float32)]
    ok = [g["pair_id"] for g in got] == list(range(world)) and all(g["iterations"] == 2**31 - 1 - i for i, g in enumerate(got))
    shards = None
    if args.pairs > 0:
        # the whole job shape: pair p -> rank p mod world, every rank aligns its shard (here: a record only that pair id can produce), ONE
        # all-gather of the padded shards, records back in pair order on every rank
        def fake(pid):
            return distributed.pack_record(pid, np.eye(4, dtype=np.float32).reshape(16) * (pid + 1), pid % 2, 2**31 - 1 - pid, 1000 + pid, 0.0, 0.0)
        allp = distributed.run_pairs(args.pairs, world, rank, fake)
        gotp = [distributed.unpack_record(r) for r in allp]
        ok = ok and [g["pair_id"] for g in gotp] == list(range(args.pairs)) and all(g["iterations"] == 2**31 - 1 - i and g["n_inliers"] == 1000 + i
                                                                                  and g["T"][0, 0] == i + 1 for i, g in enumerate(gotp))
        shards = [len(distributed.shard_pairs(args.pairs, world, r)) for r in range(world)]
    if rank == 0:
        print(json.dumps({"metric": "scan-pair registrations/sec", "value": None, "unit": "registrations/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / max(1, args.steps),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "none",
                          "dry_run": True, "ranks_seen": [g["pair_id"] for g in got], "records_ok": bool(ok), "pairs": args.pairs, "shard_sizes": shards,
                          "collective": {"process_group": bool(use_pg), "backend": "gloo" if use_pg else None}}), flush=True)
    if use_pg:
        dist.destroy_process_group()
    if not ok:
        raise SystemExit(3)


SEED_SPREAD = (566, 567, 568, 569, 570, 571)   # matcher_extremes.seed_spread: scenes of the bench pair's generator (synthetic.SEED first)


def job_sizes(n_pairs, seed=566):
    """points per cloud of the job's pairs: log-uniform in [1e5, 1e6], multiples of 10 000, one draw per pair from default_rng(seed) -- the
    reference's data/tests.yaml lists 156 pairs of WHU-TLS / kizhi / arch / ... scans (clouds not shipped) that its loader voxel-filters
    into this range (src/common.cpp:429-470); src and tgt of a pair get the same size"""
    rng = np.random.default_rng(seed)
    return [int(round(10 ** u / 1e4) * 1e4) for u in rng.uniform(5.0, 6.0, n_pairs)]


def _job_make(args):
    from lgr_amd import synthetic
    pid, n = args
    return pid, synthetic.make_pair(n, seed=synthetic.SEED + pid)


def run_job(args, world, rank, local):
    """BASELINE configs[2] (data/tests.yaml: 156 pairs, src/main.cpp:384-407 loops them sequentially) as a job on this rank's GPU.  Pairs are
    generated by a small pool of CPU processes ahead of the GPU (started BEFORE this process touches HIP), uploaded, aligned through
    distributed.run_pairs (the record's time_cs + time_te is the device-synchronised alignment time, include/analysis.h:68-70); rank 0 prints one
    JSON line: pairs/s over the summed alignment time and over the job's wall (generation and upload included), ms per pair by size class, and
    the makespan each sharding policy would give 8 ranks from the measured per-pair times."""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    n_pairs = args.job_pairs
    sizes = job_sizes(n_pairs)
    sys.path.insert(0, os.path.join(ROOT, "lidar-global-registration_amd"))
    from lgr_amd import distributed
    costs = [distributed.pair_cost(n, n) for n in sizes]
    mine = distributed.rank_order(n_pairs, world, rank, args.policy, costs)   # (the order run_pairs asks for them)
    # the generator processes inherit this environment: one BLAS / OpenMP thread each.  (Left alone every worker's numpy starts a thread per
    # visible core -- six workers, ~100 runnable threads on a 16-core lease -- and the library's host threads, which drive ~40 short
    # synchronisations per alignment, wait for a core: single pairs took two to four times their time, profiles/r5_job_tests156.json's maxima)
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
        os.environ[var] = "1"
    pool = ProcessPoolExecutor(max_workers=max(1, min(4, _usable_cores() // max(1, world) - 3)), mp_context=mp.get_context("spawn"))
    window = 8
    futs = {}
    nxt = 0

    def prefetch():
        nonlocal nxt
        while nxt < len(mine) and len(futs) < window:
            futs[mine[nxt]] = pool.submit(_job_make, (mine[nxt], sizes[mine[nxt]]))
            nxt += 1
    prefetch()
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path is the product and has no CPU fallback")
    use_pg = world > 1
    if use_pg:
        _rendezvous_defaults(world)
        dist.init_process_group("nccl" if args.backend == "nccl" else "gloo", **({"device_id": torch.device("cuda", local)} if args.backend == "nccl" else {}))
    torch.cuda.set_device(local)
    from lgr_amd import capi
    ctx = capi.Context(local)
    ctx.set_options(helper_contexts=0 if args.single_context else 1, arithmetic=capi.ARITH_PCL if args.arithmetic == "pcl" else capi.ARITH_FAST,
                    ransac_schedule={"default": 0, "chain": 1, "resident": 2}[args.ransac_schedule])
    per = {}

    def align_fn(pid):
        _, pair = futs.pop(pid).result()
        prefetch()
        src = torch.from_numpy(pair["src"]).cuda(local)
        tgt = torch.from_numpy(pair["tgt"]).cuda(local)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = ctx.align(src, tgt, make_params(capi, pair, args.matching))
        torch.cuda.synchronize()
        per[pid] = dict(ms=1e3 * (time.perf_counter() - t0), err=float(np.abs(res.matrix().astype(np.float64) - pair["T_gt"]).max()), converged=int(res.converged))
        return distributed.pack_record(pid, res.transformation, res.converged, res.iterations, res.n_inliers, res.time_cs, res.time_te)

    t0 = time.perf_counter()
    allr = distributed.run_pairs(n_pairs, world, rank, align_fn, device=(f"cuda:{local}" if use_pg and args.backend == "nccl" else None), policy=args.policy, costs=costs)
    wall = time.perf_counter() - t0
    pool.shutdown()
    if rank == 0:
        recs = [distributed.unpack_record(r) for r in allr]
        assert [r["pair_id"] for r in recs] == list(range(n_pairs))
        t_pair = [r["time_cs"] + r["time_te"] for r in recs]           # seconds, device-synchronised, every rank's pairs
        classes = {}
        for n, t in zip(sizes, t_pair):
            c = "%dk-%dk" % (100 * (n // 100000), 100 * (n // 100000) + 100) if n < 1000000 else "1000k"
            classes.setdefault(c, []).append(1e3 * t)
        pred8 = {pol: distributed.makespan(t_pair, 8, pol, costs) for pol in ("round_robin", "lpt")}
        pred_cost8 = {pol: distributed.makespan(costs, 8, pol, costs) for pol in ("round_robin", "lpt")}
        out = {"metric": "scan-pair registrations/sec (job)", "value": n_pairs / max(t_pair_sum := sum(t_pair), 1e-9) if world == 1 else n_pairs / wall, "unit": "registrations/s", "n_gpus": world,
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "BASELINE configs[2] stand-in: %d synthetic pairs, 1e5 .. 1e6 points per cloud (log-uniform, seed 566), the configs[1] parameter profile" % n_pairs,
                          "pairs": n_pairs, "matching": args.matching, "policy": args.policy, "arithmetic": args.arithmetic, "parallelism": f"pairs sharded over {world} GPU(s) by {args.policy}"},
               "job": {"sum_alignment_seconds": sum(t_pair), "wall_seconds_incl_generation_and_upload": wall,
                       "registrations_per_s_alignment_only": n_pairs / max(sum(t_pair), 1e-9), "registrations_per_s_wall": n_pairs / wall,
                       "ms_per_pair_by_size": {c: {"pairs": len(v), "mean_ms": float(np.mean(v)), "min_ms": float(np.min(v)), "max_ms": float(np.max(v))} for c, v in sorted(classes.items())},
                       "sizes_min_median_max": [int(min(sizes)), int(np.median(sizes)), int(max(sizes))],
                       "converged": int(sum(r["converged"] for r in recs)),
                       "max_abs_err_vs_gt_local_pairs": max((v["err"] for v in per.values()), default=None),
                       "predicted_8_rank_makespan_seconds_from_measured_times": pred8,
                       "predicted_8_rank_speedup_over_one_rank": {k: sum(t_pair) / v for k, v in pred8.items()},
                       "cost_model_makespan_ratio_rr_over_lpt": pred_cost8["round_robin"] / pred_cost8["lpt"],
                       "cost_model_fit": {"corrcoef_cost_vs_time": float(np.corrcoef(costs, t_pair)[0, 1]) if n_pairs > 2 else None}}}
        print(json.dumps(out), flush=True)
    if use_pg:
        dist.destroy_process_group()


def make_params(capi, pair, matching):
    mid = {"lr": capi.MATCH_LR, "cluster": capi.MATCH_CLUSTER, "one_sided": capi.MATCH_ONE_SIDED}[matching]
    # SURVEY 8d / BASELINE.md config 2 profile (data/tests.yaml values where the yaml sets them)
    return capi.default_params(matching_id=mid, metric_id=capi.METRIC_UNIFORMITY, score_id=capi.SCORE_MSE,
                               feature_radius=0.25, feature_nr_points=352, normal_nr_points=30, bf_block_size=200000,
                               edge_thr_coef=0.95, confidence=0.999, max_iterations=1000000, distance_thr=0.1,
                               vp_src=pair["vp_src"], vp_tgt=pair["vp_tgt"])


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _usable_cores():
    """threads the CPU leg may really use: the affinity mask, capped by the cgroup CPU quota when there is one (a GPU box hands a
    one-GPU job a share of the host's cores; 256 threads on a 16-core quota only oversubscribe)."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            tok = open(path).read().split()
            if path.endswith("cpu.max"):
                if tok[0] != "max":
                    cores = min(cores, max(1, int(round(int(tok[0]) / int(tok[1])))))
            else:
                q = int(tok[0])
                if q > 0:
                    cores = min(cores, max(1, int(round(q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))))
            break
        except Exception:
            continue
    return cores


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def gpu_stages(ctx, capi, pair, src, tgt, params, voxel, radius):
    """the HIP path stage by stage (outside the timed region) so that the CPU leg can check every stage's output."""
    out = {}
    for side, cloud in (("src", src), ("tgt", tgt)):
        surf = ctx.downsample(cloud, voxel).clone()
        nrm = ctx.normals_knn(surf.clone(), 30, vp=pair["vp_" + side])
        feat = ctx.fpfh(cloud, nrm, radius)
        out[side] = dict(surf=surf.cpu().numpy(), nrm=nrm.cpu().numpy(), feat=feat, feat_h=feat.cpu().numpy())
    m = ctx.match_bf2(out["src"]["feat"], out["tgt"]["feat"], params.bf_block_size)
    ctx.sync()
    out["match"] = [x.cpu().numpy() for x in m]
    corr = ctx.filter(params.matching_id, src, tgt, *m, params.distance_thr)
    out["corr"] = corr
    res, mask = ctx.ransac(src, tgt, corr, params)
    out["ransac"] = (res, mask)
    return out


HBM_PEAK_GBS = 8000.0            # same guide: HBM3E 8 TB/s (about 6.3 TB/s achievable with a copy)
FP32_PEAK_TFLOPS = 157.3         # fp32 vector = fp32 matrix rate


def stage_rooflines(ctx, capi, torch, pair, src, tgt, params, n_steps_res):
    """`roofline_stages`: every stage of the path run ALONE on the ctx stream (outside the timed region, second of two runs),
    wall time between stream synchronisations, against SURVEY 8(d)'s algorithmic bytes / FLOP per unit of work:
      downsample 48 M + 48 N bytes; normals 48 N + 16 N; FPFH = SPFH (32 N + 132 N bytes, 90 N K FLOP) + weighting
      (132 N + 16 M + 132 M bytes, 66 M K FLOP), K = feature_nr = 352 neighbours by construction; matching 69 Mq Mt FLOP,
      132 (Mq + Mt) + 8 (Mq + Mt) bytes; RANSAC 150 FLOP per iteration + 30 C FLOP per hypothesis that survives the
      prerejection, 28 C bytes per verification launch; M = points, N = surface voxels (both clouds summed).
    The stage time holds the stage's helper launches too (grid build, sorts); the per-kernel split is in profiles/."""
    r = 0.25
    voxel = float(np.sqrt(np.float32(np.pi * r * r / 352.0)))

    def timed(f):
        f(); ctx.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = f()
        ctx.sync(); torch.cuda.synchronize()
        return out, 1e3 * (time.perf_counter() - t0)

    M = N = 0
    ms = dict(downsample=0.0, normals=0.0, fpfh=0.0)
    feats = []
    for cloud, vp in ((src, pair["vp_src"]), (tgt, pair["vp_tgt"])):
        surf, t = timed(lambda: ctx.downsample(cloud, voxel).clone()); ms["downsample"] += t
        nrm, t = timed(lambda: ctx.normals_knn(surf.clone(), 30, vp=vp)); ms["normals"] += t
        f, t = timed(lambda: ctx.fpfh(cloud, nrm, r)); ms["fpfh"] += t
        feats.append(f)
        M += int(cloud.shape[0]); N += int(surf.shape[0])
    m, t = timed(lambda: ctx.match_bf2(feats[0], feats[1], params.bf_block_size)); ms["match"] = t
    corr, t = timed(lambda: ctx.filter(params.matching_id, src, tgt, *m, params.distance_thr)); ms["filter"] = t
    (rres, _), t = timed(lambda: ctx.ransac(src, tgt, corr, params)); ms["ransac"] = t
    mq, mt, C, K = int(src.shape[0]), int(tgt.shape[0]), int(len(corr)), 352
    H = max(0, int(rres.iterations) - int(rres.num_rejections))
    alg = {
        "downsample": dict(bound="hbm", bytes=48.0 * M + 48.0 * N, flop=0.0),
        "normals": dict(bound="hbm", bytes=48.0 * N + 16.0 * N, flop=0.0),
        "fpfh": dict(bound="hbm", bytes=(32.0 + 132.0) * N + 132.0 * N + (16.0 + 132.0) * M, flop=90.0 * N * K + 66.0 * M * K),
        "match": dict(bound="mfma", bytes=140.0 * (mq + mt), flop=69.0 * mq * mt),
        "filter": dict(bound="hbm", bytes=48.0 * (mq + mt) + 8.0 * (mq + mt) + 16.0 * C, flop=0.0),
        "ransac": dict(bound="valu", bytes=28.0 * C, flop=150.0 * rres.iterations + 30.0 * C * H),
    }
    out = {}
    for k, a in alg.items():
        sec = ms[k] * 1e-3
        gbs, tf = a["bytes"] / sec / 1e9, a["flop"] / sec / 1e12
        out[k] = {"bound": a["bound"], "ms_alone": ms[k], "algorithmic_bytes": a["bytes"], "algorithmic_flop": a["flop"],
                  "achieved_GBps": gbs, "frac_hbm": gbs / HBM_PEAK_GBS, "achieved_TFLOPs": tf, "frac_fp32": tf / FP32_PEAK_TFLOPS}
    out["units"] = {"M_points_both_clouds": M, "N_surface_voxels_both_clouds": N, "K_neighbours": K, "C_correspondences": C,
                    "ransac_iterations": int(rres.iterations), "hypotheses_verified": H}
    out["note"] = ("stage run alone, host wall between stream synchronisations (second of two runs); match: the algorithmic 69 Mq Mt FLOP over the whole "
                   "stage -- frac_fp32 above 1 is the exact tile skipping, the issued-FLOP fraction of the f16 MFMA peak is `roofline.frac`")
    return out


def matcher_extremes(ctx, capi, torch, pair, src, tgt, params):
    """What the headline depends on (outside the timed region).  The matcher's cost is a property of the descriptor
    distribution: its exact bound-based skipping computes ~10 % of the (row block, stage) tiles of THIS scene's FPFH rows.  Two more
    measurements of the matcher stage (lgr_match_bf2_dev alone, the faster of two runs after a warm-up call, host wall between stream synchronisations):
      dense          the same pair's rows with lgr_match_options.prune = 0: every tile computed on the f16 MFMA path, no bounds, no masks
      structureless  the production schedule on rows no bound can separate: M x 33 rows of three 11-bin blocks, bins i.i.d. uniform,
                     each block normalised to sum 100 (the FPFH shape with no structure: the worst case for every bound)
    each with the fraction of tiles it executed and its match_mfma time.  Results are identical in every mode (tests/test_gpu_parity_1m.py);
    only the time moves."""
    r = 0.25
    voxel = float(np.sqrt(np.float32(np.pi * r * r / 352.0)))
    feats = []
    for cloud, vp in ((src, pair["vp_src"]), (tgt, pair["vp_tgt"])):
        surf = ctx.downsample(cloud, voxel).clone()
        feats.append(ctx.fpfh(cloud, ctx.normals_knn(surf.clone(), 30, vp=vp), r))
    ctx.sync(); torch.cuda.synchronize()

    def timed(a, b):
        # one warm-up call, then the faster of two (round 5: about one call in a hundred stalls on the host for ~35 ms on these boxes -- no gap inside
        # the call's kernel timeline, python's collector switched off or not --, and a single sample of a 15 ms stage carried that into the report)
        out = {}
        for it in range(3):
            ctx.sync(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.match_bf2(a, b, params.bf_block_size)
            ctx.sync(); torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if it == 2 and 1e3 * dt >= out["match_stage_ms"]:
                continue
            z, f = ctx.match_lbstats()
            out = {"match_stage_ms": 1e3 * dt, "match_mfma_ms": ctx.match_kernel_ms(), "executed_tile_fraction": ctx.match_work(),
                   "issued_tile_fraction": ctx.match_issued(), "operand_format": ctx.match_format(), "zero_lower_bound_fraction": (z / f) if f > 0 else None, "irregular_rows": list(ctx.match_irregular()[:2])}
        return out

    def features_of(pr):
        fs = []
        for side in ("src", "tgt"):
            cloud = torch.from_numpy(pr[side]).cuda()
            surf = ctx.downsample(cloud, voxel).clone()
            fs.append(ctx.fpfh(cloud, ctx.normals_knn(surf.clone(), 30, vp=pr["vp_" + side]), r))
        ctx.sync(); torch.cuda.synchronize()
        return fs

    res = {"this_pair": timed(feats[0], feats[1])}
    ctx.set_match_options(prune=0)
    try:
        res["dense"] = timed(feats[0], feats[1])
    finally:
        ctx.set_match_options()
    g = torch.Generator(device="cuda"); g.manual_seed(566)
    rows = []
    for n in (int(src.shape[0]), int(tgt.shape[0])):
        x = torch.rand((n, 3, 11), generator=g, device="cuda", dtype=torch.float32) + 1e-3
        rows.append((x * (100.0 / x.sum(2, keepdim=True))).reshape(n, 33).contiguous())
    torch.cuda.synchronize()
    res["structureless"] = timed(rows[0], rows[1])
    ctx.set_match_options(prune=0)
    try:
        res["structureless_dense"] = timed(rows[0], rows[1])      # the SAME rows on the dense schedule: what `structureless` has to be compared with
    finally:
        ctx.set_match_options()
    del rows
    # a third scene family: man-made planes (the shape of the reference's WHU-TLS / kizhi configs, data/tests.yaml): 85 % of the points on 41 rectangles
    from lgr_amd import synthetic
    pp = synthetic.make_planar_pair(int(src.shape[0]), seed=synthetic.SEED)
    pf = features_of(pp)
    res["planar"] = timed(pf[0], pf[1])
    res["planar"]["scene"] = "%d rectangles (ground + 8 buildings), %.0f %% of the points on them, 15 %% clutter blobs, noise 5 mm" % (pp["n_planes"], 100 * pp["plane_frac"])
    # the same generator with other seeds: how much of the headline is this one scene?  (Round 5: at 900 k points the match stage of seeds
    # 566 .. 571 took 14.6 / 17.8 / 21.0 / 24.2 / 12.7 / 41.4 ms before pass 0 took every zero lower bound; the bench pair is seed 566.)
    spread = []
    for sd in SEED_SPREAD:
        pr = synthetic.make_pair(int(src.shape[0]), seed=sd)
        f2 = features_of(pr)
        t = timed(f2[0], f2[1])
        spread.append({"seed": sd, "match_stage_ms": t["match_stage_ms"], "match_mfma_ms": t["match_mfma_ms"], "executed_tile_fraction": t["executed_tile_fraction"]})
        del f2, pr
    res["seed_spread"] = spread
    res["note"] = ("matcher stage alone (lgr_match_bf2_dev, both directions), the faster of two runs after a warm-up call; this_pair = the bench pair's FPFH rows on the production "
                   "schedule, dense = the same rows with prune = 0, structureless = uniform random 11-bin blocks normalised to 100 on the production schedule "
                   "(lgr_match_options.auto_dense: >= 90 % zero lower bounds -> pass 0 computes everything), structureless_dense = the same random rows with "
                   "prune = 0, planar = FPFH rows of a planar-dominated pair on the production schedule; seed_spread = the bench pair's generator with other seeds (the bench pair is the first); irregular_rows = rows per side off the block-sum consensus "
                   "(all-zero FPFH rows of isolated points) that took the exact side scan instead of costing the pair the rotated format")
    return res


def cpu_baseline(pair, g, args, matching, capi):
    """The CPU oracle (the builder's restatement of the reference's algorithm, `kind: "port"` -- the reference itself needs PCL /
    OpenCV and cannot be built here) on the SAME 1M-point pair on this host's cores.  Every stage but the matcher runs IN FULL
    on both clouds (timed); the brute-force matcher runs S sampled queries of the pair's REAL FPFH rows against all 1M train
    rows per direction (train rows reused across 16 queries, bf blocks of 200 000 as include/matching.h:594-634) and is scaled
    by M / S.  The same runs are the parity check of the timed HIP path (`parity_sample`): each oracle stage is fed with the HIP
    path's output of the stage before and compared bit for bit."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as o
    o.build()
    cores = _usable_cores()
    o.set_num_threads(cores)
    n = pair["src"].shape[0]
    n_dir = 1 if matching == "one_sided" else 2
    r = 0.25
    voxel = float(np.sqrt(np.float32(np.pi * r * r / 352.0)))      # include/matching.h:231 (double product, float sqrt)
    t = dict(downsample=0.0, normals=0.0, fpfh=0.0)
    par = {"queries": 0, "mismatches": 0, "stages": {}}

    def check(name, got, want):
        bad = int((_bits(got) != _bits(want)).sum()) if got.shape == want.shape else -1
        par["stages"][name] = {"compared": int(want.size), "mismatches": bad}

    feats = {}
    for side in ("src", "tgt"):
        t0 = time.time(); ds = o.downsample(pair[side], voxel); t["downsample"] += time.time() - t0
        check("downsample_" + side, g[side]["surf"], ds)
        t0 = time.time(); nrm = o.normals_knn(g[side]["surf"], 30, vp=pair["vp_" + side]); t["normals"] += time.time() - t0
        check("normals_" + side, g[side]["nrm"], nrm)
        t0 = time.time(); f = o.fpfh(pair[side], g[side]["nrm"], r); t["fpfh"] += time.time() - t0
        check("fpfh_" + side, g[side]["feat_h"], f)
        feats[side] = g[side]["feat_h"]
    # matcher: sampled queries (real rows) x all train rows, both directions; S grows until the sample costs about half the budget
    rng = np.random.default_rng(566)
    ab_i, ab_d, ba_i, ba_d = g["match"]
    S = 512
    dt = 0.0
    while True:
        mism = 0
        t0 = time.time()
        for q, tr, gi, gd in ((feats["src"], feats["tgt"], ab_i, ab_d), (feats["tgt"], feats["src"], ba_i, ba_d))[:n_dir]:
            sel = np.sort(rng.choice(n, S, replace=False)).astype(np.int32)
            oi, od = o.match_bf_subset(q, sel, tr, 200000)
            ok = oi >= 0
            mism += int((gi[sel] != oi).sum()) + int((_bits(gd[sel])[ok] != _bits(od)[ok]).sum())
        dt = time.time() - t0
        par["queries"] += n_dir * S
        par["mismatches"] += mism
        if dt > 0.3 * args.cpu_seconds or S >= 65536:
            break
        S = int(min(65536, max(2 * S, S * 0.5 * args.cpu_seconds / max(dt, 1e-3))))
    t["match"] = dt * n / S
    # filter + RANSAC in full on the HIP path's match tables / correspondences
    mid = {"lr": o.MATCH_LR, "cluster": o.MATCH_CLUSTER, "one_sided": o.MATCH_ONE_SIDED}[matching]
    t0 = time.time()
    oc = o.filter_matches(mid, pair["src"], pair["tgt"], ab_i, ab_d, ba_i, ba_d, 0.1)
    t["filter"] = time.time() - t0
    gc = g["corr"]
    same = len(oc) == len(gc) and bool((oc["query"] == gc["index_query"]).all() and (oc["match"] == gc["index_match"]).all()
                                       and (_bits(oc["distance"]) == _bits(gc["distance"])).all())
    par["stages"]["filter"] = {"compared": int(len(oc)), "mismatches": 0 if same else -1}
    p = o.default_params(rng_mode=o.RNG_PHILOX, metric_id=o.METRIC_UNIFORMITY, score_id=o.SCORE_MSE, max_iterations=1000000,
                         distance_thr=0.1, edge_thr_coef=0.95, confidence=0.999)
    t0 = time.time(); ores, omask = o.ransac(pair["src"], pair["tgt"], oc, p); t["ransac"] = time.time() - t0
    gres, gmask = g["ransac"]
    same = same and (ores.iterations, ores.n_inliers) == (gres.iterations, gres.n_inliers) and bool((omask == gmask).all()) \
        and bool((_bits(ores.matrix()) == _bits(gres.matrix())).all())
    par["stages"]["ransac"] = {"compared": int(ores.iterations), "mismatches": 0 if same else -1,
                               "max_abs_diff_4x4": float(np.abs(ores.matrix().astype(np.float64) - gres.matrix().astype(np.float64)).max())}
    par["mismatches"] += sum(1 for v in par["stages"].values() if v["mismatches"] != 0)
    total = sum(t.values())
    return t, total, cores, S, par


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
    if args.job:
        return run_job(args, world, rank, local)
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path is the product and has no CPU fallback")
    if args.backend == "gloo":
        local = local % torch.cuda.device_count()      # rehearsal: several ranks may share one card
    use_pg = world > 1 or args.force_collective      # a process group exists: every collective of the path really runs
    if use_pg:
        _rendezvous_defaults(world)                  # (--force-collective without a launcher: a one-rank group of our own)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    torch.cuda.set_device(local)
    from lgr_amd import capi, synthetic, distributed
    ctx = capi.Context(local)
    ctx.set_options(helper_contexts=0 if args.single_context else 1, arithmetic=capi.ARITH_PCL if args.arithmetic == "pcl" else capi.ARITH_FAST,
                    ransac_schedule={"default": 0, "chain": 1, "resident": 2}[args.ransac_schedule])
    if args.match_opt:
        ctx.set_match_options(**{k: int(v) for k, v in (kv.split("=", 1) for kv in args.match_opt)})

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
        gathered = distributed.gather_records(record, world, force=use_pg)     # the single collective of the path (RCCL all-gather when N > 1)
        coll["all_gathers"] += int(use_pg)
        coll["records_seen"] = int(gathered.shape[0])
        return res

    coll = {"all_gathers": 0, "records_seen": 0}

    def barrier():
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    res = None
    for _ in range(args.warmup):
        res = step()
    barrier()
    t0 = time.perf_counter()
    kernel_ms, stage_ms, work, issued_w, coarse, shell = [], [], [], [], [], []
    for _ in range(args.steps):
        res = step()
        kernel_ms.append(ctx.match_kernel_ms())
        mstats = ctx.match_stats()
        mstats["refilter_pairs_ab"], mstats["refilter_pairs_ba"] = ctx.match_pairs()
        work.append(ctx.match_work())
        issued_w.append((ctx.match_issued(), ctx.match_issued_pairs()))
        coarse.append(ctx.match_coarse())
        shell.append(ctx.match_shell())
        stage_ms.append(list(res.stage_ms)[:7])
    barrier()
    elapsed = time.perf_counter() - t0
    if use_pg:
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
        executed = float(np.mean(work))          # every (row block, stage) once: <= 1
        issued_frac, issued_pairs = [float(x) for x in np.mean(np.array(issued_w), 0)]   # the passes summed (a stage straddling two leaves can be computed by two passes): >= executed
        fmt = ctx.match_format()
        flop_per_pair = {"f16": 224.0, "f16r": 192.0, "f32": 68.0}[fmt]
        peak = MFMA_F32_PEAK_TFLOPS if fmt == "f32" else MFMA_F16_PEAK_TFLOPS
        # tiles the kernel abandons after their first two MFMA steps (coarse rejection, rotated format) issue 64 of the 192 FLOP
        c_tested, c_abandoned = [float(x) for x in np.mean(np.array(coarse), 0)]
        # ... and tiles the shell test leaves out of a swept stage issue none
        c_skipped = float(np.mean(shell))
        # (issued_pairs counts the PADDED operands' element pairs -- clusters are padded to whole row blocks / column tiles -- which is what the
        #  kernel multiplies: SQ_INSTS_MFMA x 32768 of the PMC pass agrees with this count, tools/pmc_summary.py)
        issued = flop_per_pair * issued_pairs - ((c_abandoned * (192.0 - 64.0) + c_skipped * 192.0) * 1024.0 if fmt == "f16r" else 0.0)
        # (the tiles the sweep keeps are finished by a kernel of their own, from the first step: their two coarse steps are issued twice)
        issued += (c_tested - c_abandoned) * 64.0 * 1024.0 if fmt == "f16r" else 0.0
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
            "roofline": {"kernel": "the MFMA passes of the matcher: match_mfma (pass 0) + match_sweep + match_tiles (final pass), both directions at once", "bound": "mfma", "achieved": achieved, "peak": peak,
                         "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                         "operand_format": {"f16": "f16 two-term splits, f32 accumulate, K = 112 (224 MFMA FLOP/pair)",
                                            "f16r": "f16 two-term splits of 30 Helmert coordinates, f32 accumulate, K = 96 (192 MFMA FLOP/pair)",
                                            "f32": "f32 (68 MFMA FLOP/pair)"}[fmt],
                         "kernel_ms": k_ms, "executed_tile_fraction": executed, "issued_tile_fraction": issued_frac, "coarse_tiles_tested": c_tested, "coarse_tiles_abandoned": c_abandoned, "shell_tiles_skipped": c_skipped,
                         "mfma_flop_issued": issued, "effective_tflops_algorithmic": effective,
                         "rerank": mstats, "algorithmic_flop_per_launch": alg_flop, "directions_per_launch": n_dir},
            "stage_ms": dict(zip(["downsample", "normals", "fpfh", "match", "filter", "ransac", "refit"],
                                 [float(x) for x in np.mean(np.array(stage_ms), 0)])),
            "collective": {"process_group": bool(use_pg), "backend": args.backend if use_pg else None, "all_gathers_executed": coll["all_gathers"],
                           "records_per_gather": coll["records_seen"], "barrier_and_allreduce_max": bool(use_pg)},
            "host": {"threads_per_rank": ctx.host_threads(), "usable_cores": _usable_cores(), "ranks_on_host": world,
                     "usable_cores_per_rank": _usable_cores() / max(1, world), "helper_contexts": int(not args.single_context)},
            "result": {"converged": int(res.converged), "iterations": int(res.iterations), "n_correspondences": int(res.n_correspondences),
                       "n_inliers": int(res.n_inliers), "max_abs_err_vs_gt": err},
        }
        if world == 1 and not args.no_stage_rooflines:
            out["roofline_stages"] = stage_rooflines(ctx, capi, torch, pair, src, tgt, params, res)
        if world == 1 and not args.no_matcher_extremes:
            out["matcher_extremes"] = matcher_extremes(ctx, capi, torch, pair, src, tgt, params)
        if (not args.no_cpu_baseline or args.verify) and world == 1:   # CPU leg: rank 0 of the 1-GPU run only, outside the timed region
            voxel = float(np.sqrt(np.float32(np.pi * 0.25 * 0.25 / 352.0)))
            g = gpu_stages(ctx, capi, pair, src, tgt, params, voxel, 0.25)
            gres = g["ransac"][0]
            staged_equals_pipeline = bool((_bits(gres.matrix()) == _bits(res.matrix())).all()) and int(gres.iterations) == int(res.iterations)
            t, total, cores, S, par = cpu_baseline(pair, g, args, args.matching, capi)
            par["staged_chain_equals_timed_pipeline"] = staged_equals_pipeline
            if not staged_equals_pipeline:
                par["mismatches"] += 1
            out["parity_sample"] = par
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = {
                    "value": 1.0 / total, "unit": "registrations/s", "cores": cores, "kind": "port", "cpu_model": _cpu_model(),
                    "sample": (f"same 1M-pt pair; downsample, k-NN normals, FPFH, match filter and RANSAC run IN FULL (both clouds); brute-force matching: "
                               f"{S} sampled queries of the pair's real FPFH rows x all 1M train rows x {n_dir} direction(s), scaled by M/S"),
                    "note": "parity oracle in the canonical (non-FMA, SSE-lane) arithmetic order, not a tuned CPU implementation; a reported baseline, not the target",
                    "seconds_per_pair_estimate": total, "stage_seconds": {k: float(v) for k, v in t.items()},
                }
        print(json.dumps(out), flush=True)
    if use_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
