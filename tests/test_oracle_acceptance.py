"""Pins of the oracle by the reference's own END-TO-END tests (CPU, no GPU):

  * tests/point2plane_distance.cpp:29-96 -- the corner scene with the reference's parameters must pass the reference's three
    assertions when run through the oracle (helpers shared with tests/test_gpu_reference_acceptance.py, where the HIP path
    is held to the same assertions at the same size);
  * tests/flann_bf_matcher.h:70-88 -- brute-force matching == exact nearest-neighbour search on the same features, in both
    directions (there: matchBF vs matchFLANN vs matchLocal; here the independent exact search is a float64 exhaustive argmin
    in numpy, FLANN not being available)."""
import numpy as np

import test_gpu_reference_acceptance as acc


def test_oracle_passes_reference_point2plane_acceptance(oracle):
    src, tgt, vp_src, vp_tgt = acc.corner_scene()
    s = oracle.normals_knn(src, 30, vp=vp_src)
    t = oracle.normals_knn(tgt, 30, vp=vp_tgt)
    res, corr, _ = oracle.align(s, t, acc.reference_params(oracle, vp_src, vp_tgt, rng_mode=oracle.RNG_PHILOX))
    assert res.converged == 1 and len(corr) > 100
    ratio, error, overlap = acc.acceptance(s, t, res.matrix(), oracle.cloud_density(t))
    assert abs(ratio - 1.0) <= 1e-5 and error < 2.0 / 3.0 and overlap < 0.72, (ratio, error, overlap)
    # the reference draws its samples from mt19937(566 + thread): with this container's libstdc++ mapping of that stream (the one
    # oracle/_ref pins, tests/test_oracle_ref.py) the scene passes for 1, 8 and 16 threads as well.  (The libstdc++ <= 10 mapping
    # with 8 threads ends at an overlap rmse of 1.25 on this lattice -- 10 000 iterations over 1 472 degenerate correspondences
    # is RNG-sensitive, which is why the reference fixes its seed; not asserted.)
    for n_threads in (1, 8, 16):
        r2, _, _ = oracle.align(s, t, acc.reference_params(oracle, vp_src, vp_tgt, rng_mode=oracle.RNG_MT19937_LEMIRE, n_threads=n_threads))
        ratio, error, overlap = acc.acceptance(s, t, r2.matrix(), oracle.cloud_density(t))
        assert abs(ratio - 1.0) <= 1e-5 and error < 2.0 / 3.0 and overlap < 0.72, (n_threads, ratio, error, overlap)


def _isclose(a, b, rtol=1e-5, atol=1e-8):       # tests/flann_bf_matcher.h:12-14
    return np.abs(a - b) <= atol + rtol * np.abs(b)


def test_bf_equals_exact_search_both_directions(oracle):
    from lgr_amd import synthetic
    pair = synthetic.make_pair(6000, seed=11)
    r = 0.5 * pair["scale"] if "scale" in pair else 0.5
    feats = []
    for cloud, vp in ((pair["src"], pair["vp_src"]), (pair["tgt"], pair["vp_tgt"])):
        surf = oracle.normals_knn(oracle.downsample(cloud, float(np.sqrt(np.pi * r * r / 352.0))), 30, vp=vp)
        feats.append(oracle.fpfh(cloud, surf, r))
    for q, t in ((feats[0], feats[1]), (feats[1], feats[0])):
        idx, dist = oracle.match_bf(q, t, 1000)                       # several bf blocks: the cross-block merge is in play
        tv = ~np.isnan(t).any(1)
        t64 = t[tv].astype(np.float64)
        cols = np.flatnonzero(tv)
        for lo in range(0, len(q), 500):
            qq = q[lo:lo + 500]
            qv = ~np.isnan(qq).any(1)
            q64 = qq[qv].astype(np.float64)
            d2 = np.maximum((q64 * q64).sum(1)[:, None] - 2.0 * (q64 @ t64.T) + (t64 * t64).sum(1)[None], 0.0)
            exact = cols[d2.argmin(1)]
            got = idx[lo:lo + 500][qv]
            same = got == exact
            # an index may differ from the float64 argmin only on a float32 near-tie (isclose of flann_bf_matcher.h)
            dd = np.sqrt(d2[np.arange(len(got)), np.searchsorted(cols, got)])
            assert (same | _isclose(dd, np.sqrt(d2.min(1)))).all()
            assert same.mean() > 0.995
            assert _isclose(dist[lo:lo + 500][qv], dd, rtol=1e-5).all()
            assert (idx[lo:lo + 500][~qv] == -1).all()               # NaN query rows have no match (include/matching.h:614-616)
