"""BASELINE-size checks (-m gpu): at 1M points the oracle is too slow to be the checker, so these use size-independent
properties of the domain: permutation recovery and symmetry of the matcher, sortedness / idempotence of the voxel
grid, rigid-motion equivariance of the pipeline stages, and ground-truth recovery of the whole path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pair1m():
    from lgr_amd import synthetic
    return synthetic.make_pair(1_000_000, seed=566)


@pytest.fixture
def opts(lgr):
    """lgr_match_options for one test; the context goes back to the defaults afterwards"""
    yield lambda **kw: lgr.set_match_options(**kw)
    lgr.set_match_options()


def test_matcher_1m_permutation_and_mutual_consistency(lgr, opts):
    opts(poison_tables=1)   # never-computed table entries hold 0: nothing may read them
    import torch
    rng = np.random.default_rng(1)
    m = 1_000_000
    x = rng.gamma(0.6, 1.0, (m, 3, 11)) + 1e-3
    a = (100.0 * x / x.sum(2, keepdims=True)).reshape(m, 33).astype(np.float32)
    perm = rng.permutation(m)
    b = (a[perm] + rng.normal(0, 1e-3, (m, 33))).astype(np.float32)
    ab_i, ab_d, ba_i, ba_d = lgr.match_bf2(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), 200000)
    lgr.sync()
    ab = ab_i.cpu().numpy(); ba = ba_i.cpu().numpy()
    inv = np.empty(m, np.int64); inv[perm] = np.arange(m)
    assert (ab == inv).mean() > 0.9999 and (ba == perm).mean() > 0.9999
    # mutual matches report the same distance from both sides (the canonical distance is symmetric bit for bit)
    mutual = ba[ab] == np.arange(m)
    assert mutual.mean() > 0.999
    np.testing.assert_array_equal(ab_d.cpu().numpy()[mutual].view(np.uint32), ba_d.cpu().numpy()[ab[mutual]].view(np.uint32))
    st = lgr.match_stats()
    assert st["dense_ab"] == 0 and st["dense_ba"] == 0 and st["items_ab"] < 1.5 * m


def test_matcher_1m_coarse_rejection_is_invisible(lgr, opts):
    """1M x 1M clustered FPFH-like rows: the final MFMA pass abandons most of its tiles after two K steps (coarse rejection);
    matches and distances are bit-identical to the run without it, in both directions, and the device self-check of the
    filter bound (sampled queries, exact group minima in double) holds with the rejection rule in force."""
    import torch
    opts(poison_tables=1, self_check=1)
    rng = np.random.default_rng(7)
    m = 1_000_000
    c = rng.gamma(0.6, 1.0, (3000, 3, 11)) + 1e-3
    c = 100.0 * c / c.sum(2, keepdims=True)
    def cloud():
        x = np.abs(c[rng.integers(0, len(c), m)] + rng.normal(0, 1.0, (m, 3, 11))) + 1e-3
        return (100.0 * x / x.sum(2, keepdims=True)).reshape(m, 33).astype(np.float32)
    ta, tb = torch.from_numpy(cloud()).cuda(), torch.from_numpy(cloud()).cuda()
    on = [t.cpu().numpy() for t in lgr.match_bf2(ta, tb, 200000)]
    lgr.sync()
    assert lgr.match_format() == "f16r"
    tested, abandoned = lgr.match_coarse()
    r_rows, r_cols = lgr.match_check()
    assert 0.0 <= r_rows <= 1.0 and 0.0 <= r_cols <= 1.0, (r_rows, r_cols)
    assert lgr.match_work() < 0.5 and abandoned > 0.5 * tested > 0, (lgr.match_work(), tested, abandoned)
    opts(poison_tables=1, self_check=1, coarse_rejection=0)
    off = [t.cpu().numpy() for t in lgr.match_bf2(ta, tb, 200000)]
    lgr.sync()
    assert lgr.match_coarse() == (0.0, 0.0)
    for x, y in zip(on, off):
        np.testing.assert_array_equal(x.view(np.uint32), y.view(np.uint32))
    print(f"1M coarse rejection: {abandoned:.0f} of {tested:.0f} final-pass tiles abandoned, work {lgr.match_work():.3f}, "
          f"bound ratios {r_rows:.3g} / {r_cols:.3g}")
    # ... and the same without the shell bound (pass 0 over whole leaves, no stage / tile left out for its shell gap)
    opts(poison_tables=1, self_check=1, shell_bound=0)
    noshell = [t.cpu().numpy() for t in lgr.match_bf2(ta, tb, 200000)]
    lgr.sync()
    assert lgr.match_shell() == 0.0
    r_rows, r_cols = lgr.match_check()
    assert 0.0 <= r_rows <= 1.0 and 0.0 <= r_cols <= 1.0, (r_rows, r_cols)
    for x, y in zip(on, noshell):
        np.testing.assert_array_equal(x.view(np.uint32), y.view(np.uint32))


def test_downsample_1m_sorted_weights_idempotent(lgr, pair1m):
    import torch
    voxel = float(np.float32(np.sqrt(np.pi * 0.25 * 0.25 / 352.0)))
    src = torch.from_numpy(pair1m["src"]).cuda()
    ds = lgr.downsample(src, voxel).cpu().numpy()
    assert 1e5 < ds.shape[0] < 1e6
    assert abs(ds[:, 8].sum() - 1_000_000) < 1e-3 * 1_000_000               # weights = number of merged points
    bound = pair1m["src"][:, :3].min(0) - np.float32(voxel) * np.float32(0.5)
    ijk = np.floor((ds[:, :3] - bound) / np.float32(voxel)).astype(np.int64)
    key = (ijk[:, 2] << 42) | (ijk[:, 1] << 21) | ijk[:, 0]
    assert (np.diff(key) > 0).all()                                           # canonical order, one point per voxel
    # downsampling the downsampled cloud with unit weights on the same grid keeps one point per voxel
    ds1 = ds.copy(); ds1[:, 8] = 1.0
    ds2 = lgr.downsample(torch.from_numpy(ds1).cuda(), voxel).cpu().numpy()
    assert ds2.shape[0] >= 0.99 * ds.shape[0]


def test_align_1m_recovers_ground_truth(lgr, pair1m, opts):
    opts(poison_tables=1)
    import torch
    from lgr_amd import capi
    p = capi.default_params(matching_id=capi.MATCH_LR, bf_block_size=200000, max_iterations=1000000, distance_thr=0.1,
                            vp_src=pair1m["vp_src"], vp_tgt=pair1m["vp_tgt"])
    src = torch.from_numpy(pair1m["src"]).cuda(); tgt = torch.from_numpy(pair1m["tgt"]).cuda()
    res = lgr.align(src, tgt, p)
    T = res.matrix().astype(np.float64)
    assert res.converged == 1 and res.n_correspondences > 10000 and res.n_inliers > 500
    R = T[:3, :3]
    assert np.abs(R @ R.T - np.eye(3)).max() < 1e-5 and np.linalg.det(R) > 0.999
    # ground truth within the noise-limited band (sigma = 5 mm, inlier threshold 0.1 m)
    Rg = pair1m["T_gt"][:3, :3]
    ang = np.degrees(np.arccos(np.clip((np.trace(Rg.T @ R) - 1) / 2, -1, 1)))
    assert ang < 0.5 and np.linalg.norm(T[:3, 3] - pair1m["T_gt"][:3, 3]) < 0.05
    # determinism: a second run is bit-identical (order-independent reductions everywhere)
    res2 = lgr.align(src, tgt, p)
    np.testing.assert_array_equal(res2.matrix().view(np.uint32), res.matrix().view(np.uint32))
    assert res2.n_inliers == res.n_inliers and res2.n_correspondences == res.n_correspondences


def test_corner_scene_parity(lgr, oracle):
    """the reference's synthetic end-to-end scene (tests/point2plane_distance.cpp:31-53): three orthogonal 100x100
    grids, tgt shifted by 1 in-plane, src moved by GT^-1.  Perfect lattices make FPFH massively degenerate (thousands
    of identical descriptors) -- the dense / tie paths of the matcher must still agree with the oracle bit for bit."""
    import torch
    from lgr_amd import capi
    from lgr_amd.synthetic import make_points
    n, shift = 40, 5          # 40x40 per plane keeps the oracle in seconds (the reference uses 100)
    s, t = [], []
    for i in range(n):
        for j in range(n):
            s += [(0 * shift + 2.0 * i, 0 * shift + 2.0 * j, 0.0), (1 * shift + 2.0 * i, 0.0, 1 * shift + 2.0 * j), (0.0, 2 * shift + 2.0 * i, 2 * shift + 2.0 * j)]
            t += [(0 * shift + 2.0 * i + 1.0, 0 * shift + 2.0 * j, 0.0), (1 * shift + 2.0 * i, 0.0, 1 * shift + 2.0 * j + 1.0), (0.0, 2 * shift + 2.0 * i + 1.0, 2 * shift + 2.0 * j)]
    gt = np.array([[0.0803703, -0.996763, -0.00201846, 1.2143], [0.996758, 0.080377, -0.00349969, -6.13404],
                   [0.00365057, -0.00173067, 0.999992, -1.17221], [0, 0, 0, 1]])
    src = np.array(s) ; tgt = np.array(t)
    gi = np.linalg.inv(gt)
    src = src @ gi[:3, :3].T + gi[:3, 3]
    src, tgt = make_points(src), make_points(tgt)
    kw = dict(matching_id=0, bf_block_size=1000, max_iterations=3000, distance_thr=1.0, feature_radius=8.0, feature_nr_points=352)
    p_o = oracle.default_params(rng_mode=oracle.RNG_PHILOX, **kw)
    p_g = capi.default_params(**kw)
    ocorr, _ = oracle.correspondences(src, tgt, p_o)
    corr = lgr.correspondences(torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda(), p_g).cpu().numpy().view(capi.CORR_DTYPE).reshape(-1)
    assert len(corr) == len(ocorr)
    np.testing.assert_array_equal(corr["index_query"], ocorr["query"])
    np.testing.assert_array_equal(corr["index_match"], ocorr["match"])
    np.testing.assert_array_equal(corr["distance"].view(np.uint32), ocorr["distance"].view(np.uint32))
