"""Oracle vs every golden vector the reference's own tests hold for this path, plus published known answers of the
third-party pieces it restates (libstdc++ RNG mapping, Philox).  CPU only."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def test_knnresult_reference_golden(oracle):
    """reference tests/knn_result.cpp:30-51: sorted insertion, capacity 3, later equal goes after earlier."""
    g = json.load(open(os.path.join(HERE, "golden", "knn_result.json")))
    adds = g["adds"]
    idx, dist = oracle.knnresult_run(g["capacity"], [], [])
    assert len(idx) == 0 and len(dist) == 0
    for n, exp in enumerate(g["expected_after_each_add"], 1):
        idx, dist = oracle.knnresult_run(g["capacity"], [a[0] for a in adds[:n]], [a[1] for a in adds[:n]])
        assert idx.tolist() == exp["indices"]
        assert dist.tolist() == exp["distances"]


def test_philox_known_answer(oracle):
    g = json.load(open(os.path.join(HERE, "golden", "philox_kat.json")))
    assert len(g["vectors"]) == 3
    for v in g["vectors"]:
        key = int(v["key"][0], 16) | int(v["key"][1], 16) << 32
        out = oracle.philox_full(key, [int(c, 16) for c in v["counter"]])
        assert [f"{x:08x}" for x in out] == v["out"]
        if "iter" in v:          # the sampler's form: counter = (iteration, 0, 0, 0)
            assert [f"{x:08x}" for x in oracle.philox(v["seed"], v["iter"])] == v["out"]


def test_uniform_rand_int_generator_streams(oracle):
    """include/utils.h:13-26 over std::mt19937(566): libstdc++ >= 11 maps one draw to mt() >> 1 (Lemire path),
    libstdc++ <= 10 redraws until mt() < 2^31.  numpy's MT19937 with legacy seeding is std::mt19937."""
    bg = np.random.MT19937()
    bg._legacy_seeding(566)
    raw = bg.random_raw(64).astype(np.uint64)
    lem = oracle.rng_stream(oracle.RNG_MT19937_LEMIRE, 566, 32)
    np.testing.assert_array_equal(lem.astype(np.uint64), raw[:32] >> 1)
    rej = oracle.rng_stream(oracle.RNG_MT19937_REJECT, 566, 16)
    want = [int(x) for x in raw if x < 2 ** 31][:16]
    assert rej.tolist() == want
    # std::mt19937 default-seed check value from the C++ standard: 10000th output of mt19937(5489) is 4123659995
    bg2 = np.random.MT19937(); bg2._legacy_seeding(5489)
    assert int(bg2.random_raw(10000)[-1]) == 4123659995


def _select3_py(r, n):
    return _select_py(r, n)


def _select_py(r, n):
    s = [0] * len(r)
    for i in range(len(r)):
        s[i] = r[i] % n
        j = 0
        while j < i:
            if s[i] >= s[j]:
                if s[i] < n - 1:
                    s[i] += 1; j += 1; continue
                elif s[j] == 0:
                    s[i] = 1; j += 1; continue
                else:
                    s[i] = 0
            t = s[i]
            for k in range(i, j, -1):
                s[k] = s[k - 1]
            s[j] = t
            break
    return s


def test_select_correspondences_control_flow(oracle):
    """src/sac_prerejective_omp.cpp:33-77 restated twice (C++ oracle / python) incl. the wrap-around branch
    (SURVEY A.8 example: chosen=[3, C-2], x=C-2 -> [3, 0, C-2])."""
    rng = np.random.default_rng(0)
    for n in (3, 4, 5, 10, 1000):
        for _ in range(300):
            r = rng.integers(0, 2 ** 31 - 1, 3).tolist()
            assert oracle.select3(r, n) == _select3_py(r, n)
    C = 100
    assert oracle.select3([3, C - 3, C - 2], C) == [3, 0, C - 2]
    assert oracle.select3([5, 9, 2], 1000) == [2, 5, 10]        # common case: ascending; 9 >= 5 is bumped to 10
    # the same loops for any n_samples (the reference is generic in it; 3 everywhere it ships)
    for ns in (3, 4, 5, 8):
        for n in (ns, ns + 1, 10, 1000):
            for _ in range(200):
                r = rng.integers(0, 2 ** 31 - 1, ns).tolist()
                got = oracle.select_n(r, n)
                assert got == _select_py(r, n)
                if ns == 3:
                    assert got == oracle.select3(r, n)
    # Philox draws: draw j is word j % 4 of the block with counter (iteration, j / 4, 0, 0), top 31 bits
    for it in (0, 7, 123456):
        d = oracle.philox_draws(566, it, 8)
        assert d[:4] == [w >> 1 for w in oracle.philox(566, it)]
        assert d[4:] == [w >> 1 for w in oracle.philox_full(566, [it, 1, 0, 0])]
        assert oracle.philox_draws(566, it, 3) == d[:3]


def test_combination_and_estimate_formulas(oracle):
    """include/utils.h:34-43 and src/metric.cpp:116-122 on a problem with a known support count."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "lidar-global-registration_amd"))
    from lgr_amd import synthetic
    pr = synthetic.make_correspondence_problem(n_pts=3000, c=2000, inlier_frac=0.5, sigma=0.001, thr=0.05, seed=1)
    corr = np.zeros(len(pr["corr"]), oracle.CORR_DTYPE)
    for a, b in (("query", "index_query"), ("match", "index_match"), ("distance", "distance"), ("threshold", "threshold")):
        corr[a] = pr["corr"][b]
    mask, n_inl, rmse, metric = oracle.evaluate(pr["src"], pr["tgt"], corr, pr["T_gt"], oracle.METRIC_CORRESPONDENCES, oracle.SCORE_CONSTANT)
    assert abs(n_inl - 1000) <= 5 and abs(metric - n_inl / 2000) < 1e-6
    est = oracle.estimate_max_iterations(pr["src"], pr["tgt"], corr, pr["T_gt"], 0.999, 3)
    f = np.float32(np.float32(n_inl) / np.float32(2000)) / np.float32(4)
    want = int(min(2 ** 31 - 1, np.log(1 - np.float64(np.float32(0.999))) / np.log(1.0 - float(f) ** 3)))
    assert est == want
