"""One context, many calls: buffers of the same size at the same addresses with DIFFERENT contents, and a stand-alone matcher call
right after a pipeline call.  The pipeline prepares the matcher's clustering ahead of the call (keyed by pointer and sizes,
consumed once) and keeps workspaces, events and second / third contexts alive between calls; none of that may leak state from
one call into the next.  Each result is compared with a fresh context's."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _params(capi, **kw):
    base = dict(keypoint_id=capi.KEYPOINT_ANY, metric_id=capi.METRIC_UNIFORMITY, matching_id=0, feature_radius=0.25, distance_thr=0.1,
                bf_block_size=200000, max_iterations=20000)
    base.update(kw)
    return capi.default_params(**base)


def test_same_buffers_new_contents_and_standalone_matcher_after_pipeline(lgr):
    import torch
    from lgr_amd import capi, synthetic
    pairs = [synthetic.make_pair(20000, seed=s) for s in (3, 4)]
    src = torch.empty((20000, 12), dtype=torch.float32, device="cuda")
    tgt = torch.empty((20000, 12), dtype=torch.float32, device="cuda")
    p = _params(capi)
    got = []
    for pair in pairs + pairs[:1]:                      # pair 0 again at the end: after another pair went through the same addresses
        src.copy_(torch.from_numpy(pair["src"]).cuda()); tgt.copy_(torch.from_numpy(pair["tgt"]).cuda())
        res = lgr.align(src, tgt, p)
        got.append((res.matrix().copy(), res.n_correspondences, res.n_inliers, res.iterations))
    for pair, g in zip(pairs + pairs[:1], got):
        fresh = capi.Context(0)
        try:
            res = fresh.align(torch.from_numpy(pair["src"]).cuda(), torch.from_numpy(pair["tgt"]).cuda(), p)
            assert np.array_equal(res.matrix().view(np.uint32), g[0].view(np.uint32))
            assert (res.n_correspondences, res.n_inliers, res.iterations) == g[1:]
        finally:
            fresh.close()
    assert np.array_equal(got[0][0].view(np.uint32), got[2][0].view(np.uint32))

    # a stand-alone two-sided match on rows at the addresses / sizes a pipeline call could have prepared for: the clustering a
    # pipeline call prepared is consumed by that call, nothing of it may be reused here
    rng = np.random.default_rng(9)
    a = torch.from_numpy(np.abs(rng.normal(size=(20000, 33))).astype(np.float32)).cuda()
    b = torch.from_numpy(np.abs(rng.normal(size=(20000, 33))).astype(np.float32)).cuda()
    r1 = [t.cpu().numpy() for t in lgr.match_bf2(a, b)]
    a.copy_(torch.from_numpy(np.abs(rng.normal(size=(20000, 33))).astype(np.float32)).cuda())      # same address, new rows
    r2 = [t.cpu().numpy() for t in lgr.match_bf2(a, b)]
    fresh = capi.Context(0)
    try:
        r2f = [t.cpu().numpy() for t in fresh.match_bf2(a, b)]
    finally:
        fresh.close()
    for x, y in zip(r2, r2f):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    assert not np.array_equal(r1[0], r2[0])


def test_single_context_mode_is_identical_and_starts_no_threads():
    """lgr_ctx_options.helper_contexts = 0: every piece on the caller's stream from the caller's thread -- same
    results bit for bit in every matching mode, host_threads() stays 1; the default mode starts its (at most two) persistent helper
    threads once and keeps that number over repeated calls."""
    import torch
    from lgr_amd import capi, synthetic
    pair = synthetic.make_pair(30000, seed=11)
    src, tgt = torch.from_numpy(pair["src"]).cuda(), torch.from_numpy(pair["tgt"]).cuda()
    multi, single = capi.Context(0), capi.Context(0)
    try:
        assert single.set_options(helper_contexts=0).helper_contexts == 0
        assert multi.host_threads() == 1 and single.host_threads() == 1
        for kw in (dict(matching_id=0), dict(matching_id=2), dict(matching_id=1),
                   dict(matching_id=2, keypoint_id=capi.KEYPOINT_ISS, iss_radius_src=0.08, iss_radius_tgt=0.08, feature_radius=0.0)):
            p = _params(capi, **kw)
            a = multi.align(src, tgt, p)
            b = single.align(src, tgt, p)
            assert np.array_equal(a.matrix().view(np.uint32), b.matrix().view(np.uint32)), kw
            assert (a.n_correspondences, a.n_inliers, a.iterations, a.converged) == (b.n_correspondences, b.n_inliers, b.iterations, b.converged)
            ca = multi.correspondences(src, tgt, p).cpu().numpy()
            cb = single.correspondences(src, tgt, p).cpu().numpy()
            assert np.array_equal(ca, cb)
        assert single.host_threads() == 1
        assert 2 <= multi.host_threads() <= 3
        n = multi.host_threads()
        for _ in range(3):
            multi.align(src, tgt, _params(capi))
        assert multi.host_threads() == n                 # persistent: no thread per call
        # switching an existing context over and back: internal contexts are rebuilt, results unchanged
        want = multi.align(src, tgt, _params(capi)).matrix()
        multi.set_options(helper_contexts=0)
        assert multi.host_threads() == 1
        assert np.array_equal(multi.align(src, tgt, _params(capi)).matrix().view(np.uint32), want.view(np.uint32))
        multi.set_options()
        assert np.array_equal(multi.align(src, tgt, _params(capi)).matrix().view(np.uint32), want.view(np.uint32))
    finally:
        multi.close(); single.close()


def test_match_statistics_belong_to_the_context():
    """lgr_match_last_* (they were thread_local): two contexts driven from ONE host thread keep separate figures."""
    import torch
    from lgr_amd import capi
    rng = np.random.default_rng(1)
    # (rows around 24 well-separated centres, so that the bounds really leave tiles out at this size)
    cen = rng.uniform(0, 40, size=(24, 33))
    a = torch.from_numpy((cen[rng.integers(0, 24, 9000)] + rng.normal(size=(9000, 33))).astype(np.float32)).cuda()
    b = torch.from_numpy((cen[rng.integers(0, 24, 7000)] + rng.normal(size=(7000, 33))).astype(np.float32)).cuda()
    c1, c2 = capi.Context(0), capi.Context(0)
    try:
        c1.set_match_options(prune=1, leaves=4, operand_format=capi.FORMAT_F16)
        c2.set_match_options(prune=0, operand_format=capi.FORMAT_F32)
        r1 = [t.cpu().numpy() for t in c1.match_bf2(a, b)]
        r2 = [t.cpu().numpy() for t in c2.match_bf2(a, b)]
        c1.sync(); c2.sync()
        assert c1.match_format() == "f16" and c2.match_format() == "f32"       # c1's figures survived c2's call on the same thread
        # executed fraction: every (row block, stage) once, so never above 1 (the passes summed -- match_issued -- can be: a stage that
        # straddles two leaves is computed whole by every pass that schedules one of them)
        assert c2.match_work() == 1.0 and c2.match_issued() == 1.0
        assert 0.0 < c1.match_work() < 1.0 and c1.match_issued() >= c1.match_work()
        assert c1.match_stats()["sub_cols"] != 0 and c1.match_stats() != c2.match_stats()
        for x, y in zip(r1, r2):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    finally:
        c1.close(); c2.close()
