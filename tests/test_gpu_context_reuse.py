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
