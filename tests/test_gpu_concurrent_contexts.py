"""Several contexts on ONE device, driven by several host threads at once (the reference's host is an OpenMP team,
src/sac_prerejective_omp.cpp:79-91; a host that gives each of its threads an lgr_ctx is what include/lgr.h allows).

Default (lgr_ctx_options.concurrent_contexts = 0): the contexts take turns call by call, so every buffer of every alignment must be
bit-equal to a serial run -- asserted.  concurrent_contexts = 1 (EXPERIMENTAL, include/lgr.h) lets them overlap on the device: every kernel is
deterministic and the contexts share nothing, so the results must STILL be bit-equal -- asserted as well since round 5 (ONE run of three
overlapping contexts, no retry loop; a difference FAILS the suite).  Round 3 saw rounding-level differences in the normals on the boxes it
was given; round 4 could not reproduce them with any build, not even round 3's binary.  Every run prints what
identifies the box (GPU serial, firmware, ROCm), so that a failure arrives with its unit named."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_POINTS = 1_000_000


@pytest.fixture(scope="module")
def job():
    import torch
    from lgr_amd import capi, diagnostics, synthetic
    pair = synthetic.make_pair(N_POINTS, seed=synthetic.SEED)
    src, tgt = torch.from_numpy(pair["src"]).cuda(), torch.from_numpy(pair["tgt"]).cuda()
    params = capi.default_params(matching_id=capi.MATCH_LR, metric_id=capi.METRIC_UNIFORMITY, score_id=capi.SCORE_MSE, feature_radius=0.25,
                                 bf_block_size=200000, max_iterations=1000000, distance_thr=0.1, vp_src=pair["vp_src"], vp_tgt=pair["vp_tgt"])
    torch.cuda.synchronize()
    c0 = capi.Context(0, stream=-1)
    ref = diagnostics.align_snapshot(c0, src, tgt, params)
    again = diagnostics.align_snapshot(c0, src, tgt, params)
    c0.close()
    assert diagnostics.first_difference(ref, again) is None          # the serial run reproduces itself
    assert ref["align"][2] > 100_000
    return dict(src=src, tgt=tgt, params=params, ref=ref)


def run_threads(job, n_threads, rounds, **options):
    from lgr_amd import capi, diagnostics
    bad, lock = [], threading.Lock()

    def worker(w):
        ctx = capi.Context(0, stream=-1)
        try:
            if options:
                ctx.set_options(**options)
            for it in range(rounds):
                d = diagnostics.first_difference(job["ref"], diagnostics.align_snapshot(ctx, job["src"], job["tgt"], job["params"]))
                if d is not None:
                    with lock:
                        bad.append((w, it) + d)
        finally:
            ctx.close()

    th = [threading.Thread(target=worker, args=(w,)) for w in range(n_threads)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return bad


def test_two_contexts_two_threads_take_turns_bit_equal(job):
    """the default: 2 host threads x 2 contexts x 6 alignments of the 1M pair, every pipeline buffer bit-equal to the serial run"""
    assert run_threads(job, 2, 6) == []


def test_three_contexts_overlapping_on_the_device(job, capsys):
    """concurrent_contexts = 1 (experimental): 3 host threads x 3 contexts x 4 alignments overlap on the device, ONE run; every pipeline buffer
    of the 12 alignments bit-equal to the serial run, asserted (round 3's rate on its boxes was one differing alignment in three)"""
    import json
    from lgr_amd import diagnostics
    ident = diagnostics.box_identity()
    with capsys.disabled():
        print("\n[concurrent contexts] box: " + json.dumps(ident))
    bad = run_threads(job, 3, 4, concurrent_contexts=1)
    assert bad == [], "overlapping contexts: %d of 12 alignments differ from the serial run, first (thread, round, buffer, bytes) = %s; box: %s" % (
        len(bad), bad[0], json.dumps(ident))
