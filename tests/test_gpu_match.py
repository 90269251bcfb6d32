"""GPU parity: brute-force FPFH matcher (lgr_match_bf*_dev through the C ABI) vs the oracle.

Bar: match indices AND distances bit-exact (integer / canonical-order float work), including exact ties inside a bf
block (lowest index wins), exact ties across blocks (later block wins, src/common.cpp:517-529), NaN rows
(include/matching.h:576,614), empty and ragged sizes.  Mirrors the idea of the reference's
tests/flann_bf_matcher.h:70-88 (brute force == exact argmin, both directions).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["auto", "prune_sub1", "prune_sub4", "prune_sub64", "prune_sub4_leafcols_nobox", "prune_sub4_nocoarse",
                                      "prune_sub4_groupscan", "auto_groupscan"])
def matcher_mode(request, lgr):
    """Every test runs on the automatic path and with the bound-based stage skipping forced on (it is only
    automatic for >= 65536 x 65536 inputs) with 1, 4 and 64 leaves per cluster; results must not depend on it.
    The paths are selected through lgr_match_options (include/lgr.h) held by the context."""
    base = {}
    if request.param.endswith("_groupscan"):              # exact rerank by scanning whole candidate groups (no MFMA re-filter)
        base["rerank_refilter"] = 0
    if not request.param.startswith("auto"):
        base["prune"] = 1
        base["near"] = 2                                  # narrow first pass, so that tiles really are skipped at test sizes
        base["leaves"] = int(request.param.replace("prune_sub", "").split("_")[0])
        if request.param.endswith("_leafcols_nobox"):     # the simpler schedule: ball bounds only, whole-leaf column criterion
            base["column_stage"] = 0
            base["box_bounds"] = 0
        if request.param.endswith("_nocoarse"):           # the final pass without the in-kernel coarse rejection
            base["coarse_rejection"] = 0
        # the minimum tables are initialised only where a pass computes; everything else is pre-filled with 0 (the most
        # harmful stale value) to show that no uninitialised entry is ever read
        base["poison_tables"] = 1
    lgr._base_opts = base
    lgr.set_match_options(**base)
    yield request.param
    lgr._base_opts = {}
    lgr.set_match_options()


def opts(lgr, **extra):
    """the mode's options plus overrides"""
    lgr.set_match_options(**{**getattr(lgr, "_base_opts", {}), **extra})


FMT = {"f32": 0, "f16": 1, "f16r": 2}


def fpfh_like(rng, m, spread=1.0):
    """rows shaped like FPFH: three 11-bin blocks, each summing to 100."""
    x = rng.gamma(0.6 * spread, 1.0, (m, 3, 11)).astype(np.float64) + 1e-3
    x = 100.0 * x / x.sum(2, keepdims=True)
    return x.reshape(m, 33).astype(np.float32)


def run_both(lgr, oracle, a, b, block):
    import torch
    ta = torch.from_numpy(a).cuda(); tb = torch.from_numpy(b).cuda()
    ab_i, ab_d, ba_i, ba_d = lgr.match_bf2(ta, tb, block)
    lgr.sync()
    oi, od = oracle.match_bf(a, b, block)
    ri, rd = oracle.match_bf(b, a, block)
    np.testing.assert_array_equal(ab_i.cpu().numpy(), oi)
    np.testing.assert_array_equal(ba_i.cpu().numpy(), ri)
    ok = oi >= 0
    np.testing.assert_array_equal(ab_d.cpu().numpy()[ok].view(np.uint32), od[ok].view(np.uint32))
    ok = ri >= 0
    np.testing.assert_array_equal(ba_d.cpu().numpy()[ok].view(np.uint32), rd[ok].view(np.uint32))
    # single-direction entry point agrees too
    i1, d1 = lgr.match_bf(ta, tb, block)
    lgr.sync()
    np.testing.assert_array_equal(i1.cpu().numpy(), oi)
    return oi, ri


@pytest.mark.parametrize("ma,mb,block", [(1, 1, 10), (5, 300, 100), (257, 255, 64), (1000, 1500, 10000),
                                          (3000, 5000, 1024), (4097, 4100, 200000)])
def test_match_random(lgr, oracle, ma, mb, block):
    rng = np.random.default_rng(ma * 7919 + mb)
    run_both(lgr, oracle, fpfh_like(rng, ma), fpfh_like(rng, mb), block)


def test_match_ties_and_nan(lgr, oracle):
    rng = np.random.default_rng(5)
    a = fpfh_like(rng, 700)
    b = fpfh_like(rng, 900)
    # exact duplicates of train rows inside one block and across blocks (block = 128)
    b[10] = b[200]; b[300] = b[200]; b[301] = b[200]; b[850] = b[200]
    a[3] = b[200]                     # distance exactly 0 to five train rows
    a[4] = b[200]; a[4, 0] += 0.5     # equal non-zero distance to the same five rows
    b[5] = a[77]; b[6] = a[77]        # in-block tie -> lowest index
    a[50, 7] = np.nan                 # NaN query row -> -1
    b[20, :] = np.nan                 # NaN train row never matches
    b[21, 3] = np.inf
    a[60] = a[61]                     # duplicate queries
    oi, ri = run_both(lgr, oracle, a, b, 128)
    assert oi[50] == -1 and ri[20] == -1 and ri[21] == -1
    assert oi[3] == 850 and oi[4] == 850          # later block wins the exact tie
    assert oi[77] == 5                            # lowest index inside a block
    oi2, _ = run_both(lgr, oracle, a, b, 100000)  # one block: lowest index wins
    assert oi2[3] == 10


def test_match_degenerate_all_identical(lgr, oracle):
    """every row identical: all groups are candidates -> exercises the dense fallback path."""
    rng = np.random.default_rng(9)
    row = fpfh_like(rng, 1)
    a = np.repeat(row, 300, 0)
    b = np.repeat(row, 2500, 0)
    b[1234, 5] += 1.0
    oi, ri = run_both(lgr, oracle, a, b, 1000)
    assert (oi == 2000).all()      # last block (2000..2499), lowest index inside it


def test_match_host_entry_and_empty(lgr, oracle):
    rng = np.random.default_rng(11)
    a = fpfh_like(rng, 100); b = fpfh_like(rng, 50)
    i, d = lgr.match_bf_host(a, b, 16)
    oi, od = oracle.match_bf(a, b, 16)
    np.testing.assert_array_equal(i, oi)
    np.testing.assert_array_equal(d.view(np.uint32), od.view(np.uint32))
    i, d = lgr.match_bf_host(a, np.zeros((0, 33), np.float32), 16)
    assert (i == -1).all()


def test_match_large_property(lgr):
    """BASELINE-size-independent property at a larger size: matching a set against a permuted noisy copy of
    itself recovers the permutation (noise far below the nearest-neighbour spacing), in both directions."""
    import torch
    rng = np.random.default_rng(3)
    m = 60000
    a = fpfh_like(rng, m)
    perm = rng.permutation(m)
    b = (a[perm] + rng.normal(0, 1e-3, (m, 33))).astype(np.float32)
    ab_i, ab_d, ba_i, ba_d = lgr.match_bf2(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), 10000)
    lgr.sync()
    inv = np.empty(m, np.int64); inv[perm] = np.arange(m)
    assert (ab_i.cpu().numpy() == inv).mean() > 0.999
    assert (ba_i.cpu().numpy() == perm).mean() > 0.999


def clustered(rng, m, n_modes=40, spread=1.5):
    """FPFH-like rows around a few dozen modes: the structure the leaf bounds exploit."""
    modes = fpfh_like(rng, n_modes)
    x = modes[rng.integers(0, n_modes, m)] + rng.normal(0, spread, (m, 33))
    return np.abs(x).astype(np.float32)


@pytest.mark.parametrize("ma,mb", [(20000, 24000), (9000, 30000)])
def test_match_clustered_parity_and_skipping(lgr, oracle, matcher_mode, ma, mb):
    """Clustered data at a size where whole tiles can be skipped: results stay bit-identical to the oracle, and the
    forced skipping path really leaves tiles out."""
    rng = np.random.default_rng(ma)
    a, b = clustered(rng, ma), clustered(rng, mb)
    a[17] = b[5]; b[9000] = b[5]          # exact ties across bf blocks survive the skipping
    run_both(lgr, oracle, a, b, 7000)
    import torch
    lgr.match_bf2(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), 7000)
    lgr.sync()
    w = lgr.match_work()
    if matcher_mode in ("prune_sub4", "prune_sub64", "prune_sub4_leafcols_nobox", "prune_sub4_nocoarse", "prune_sub4_groupscan"):
        assert w < 1.0, w
    if matcher_mode.startswith("auto"):
        assert w == 1.0


@pytest.mark.parametrize("fmt", ["f16", "f16r", "f32"])
@pytest.mark.parametrize("kind", ["fpfh", "clustered", "tiny", "wide", "duplicates"])
def test_filter_bound_self_check(lgr, oracle, matcher_mode, fmt, kind):
    """The MFMA filter value of every computed (query, group) entry must lie within the proven eps of the exact group
    minimum (computed in double on the device, lgr_match_options.self_check) -- on both f16-split operand formats and on f32, for
    FPFH-like rows, tight clusters, tiny and wide dynamic ranges and exact duplicates; results stay oracle-exact."""
    import torch
    if matcher_mode in ("prune_sub1", "prune_sub4_leafcols_nobox", "prune_sub4_nocoarse", "prune_sub4_groupscan", "auto_groupscan"):
        pytest.skip("same filter code path as prune_sub4 / auto")
    # f16r: the rotated 30-coordinate format forced on ANY data (rows whose blocks do not sum to a constant make its
    # dropped-coordinate term large: the bound must still hold and the result stay exact); f16: forced off
    opts(lgr, self_check=1, operand_format=FMT[fmt])
    rng = np.random.default_rng(77)
    ma, mb = 6000, 9000
    if kind == "fpfh":
        a, b = fpfh_like(rng, ma), fpfh_like(rng, mb)
    elif kind == "clustered":
        a, b = clustered(rng, ma, spread=0.3), clustered(rng, mb, spread=0.3)
    elif kind == "tiny":            # values ~1e-3 around a common offset: small centred norms, f16 subnormal halves
        a = (50.0 + 1e-3 * rng.normal(size=(ma, 33))).astype(np.float32)
        b = (50.0 + 1e-3 * rng.normal(size=(mb, 33))).astype(np.float32)
    elif kind == "wide":            # six decades of magnitudes in one set
        sa = 10.0 ** rng.uniform(-3, 3, (ma, 1)); sb = 10.0 ** rng.uniform(-3, 3, (mb, 1))
        a = (sa * rng.normal(size=(ma, 33))).astype(np.float32)
        b = (sb * rng.normal(size=(mb, 33))).astype(np.float32)
    else:
        base = fpfh_like(rng, 50)
        a = base[rng.integers(0, 50, ma)].copy(); b = base[rng.integers(0, 50, mb)].copy()
        a[::3] += rng.normal(0, 1e-4, (len(a[::3]), 33)).astype(np.float32)
    run_both(lgr, oracle, a, b, 2500)
    lgr.match_bf2(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), 2500)
    lgr.sync()
    assert lgr.match_format() == fmt
    r_rows, r_cols = lgr.match_check()
    assert 0.0 <= r_rows <= 1.0 and 0.0 <= r_cols <= 1.0, (r_rows, r_cols)
    print(f"filter bound ratio [{fmt} {kind} {matcher_mode}]: rows {r_rows:.3g} cols {r_cols:.3g}")


def test_coarse_rejection(lgr, oracle, matcher_mode):
    """The final pass abandons tiles after their first two MFMA steps when a proven bound on the coarse distance exceeds the
    upper bounds of the tile's rows and columns (rotated operand format).  The tiles it abandons must not change anything:
    same matches and distances as the oracle and as the run with the rejection switched off, in both directions and through
    the single-direction entry point; and on clustered FPFH-like rows it really does abandon tiles."""
    import torch
    if matcher_mode not in ("prune_sub4", "prune_sub64"):
        pytest.skip("needs the skipping passes (upper bounds) and the default schedule")
    rng = np.random.default_rng(4242)
    centres = fpfh_like(rng, 40)
    def cloud(m):
        x = centres[rng.integers(0, 40, m)].astype(np.float64).reshape(m, 3, 11)
        x = np.abs(x + rng.normal(0, 1.5, x.shape)) + 1e-3
        return (100.0 * x / x.sum(2, keepdims=True)).reshape(m, 33).astype(np.float32)
    # half tight clusters, half one broad distribution (there the bounds exclude little: the final pass has tiles to test)
    a = np.concatenate([cloud(6000), fpfh_like(rng, 6000)]); b = np.concatenate([fpfh_like(rng, 8000), cloud(7000)])
    opts(lgr, self_check=1, coarse_rejection=2)   # (2: the sweep also when the pass schedules most of the tiles, as it does at this size)
    oi, ri = run_both(lgr, oracle, a, b, 4000)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    on = [t.cpu().numpy() for t in lgr.match_bf2(ta, tb, 4000)]
    lgr.sync()
    assert lgr.match_format() == "f16r"
    tested, abandoned = lgr.match_coarse()
    r_rows, r_cols = lgr.match_check()
    assert 0.0 <= r_rows <= 1.0 and 0.0 <= r_cols <= 1.0, (r_rows, r_cols)
    assert tested > 0 and 0 < abandoned <= tested, (tested, abandoned)
    opts(lgr, self_check=1, coarse_rejection=0)
    off = [t.cpu().numpy() for t in lgr.match_bf2(ta, tb, 4000)]
    lgr.sync()
    assert lgr.match_coarse() == (0.0, 0.0)
    for x, y in zip(on, off):
        np.testing.assert_array_equal(x.view(np.uint32), y.view(np.uint32))
    print(f"coarse rejection [{matcher_mode}]: {abandoned:.0f} of {tested:.0f} tiles abandoned")


def test_final_pass_implementations_agree(lgr, oracle, matcher_mode):
    """The final pass of the rotated format runs as a coarse sweep that lists the tiles it keeps plus a kernel that finishes the listed
    tiles (lgr_match_options.split_sweep = 1, the default), as ONE fused kernel (split_sweep = 0, round 3), or -- when the sweep keeps
    more tiles than its list holds (kept_cap) -- as the sweep followed by the fused kernel over the same schedule.  All three must give the
    oracle's matches and distance bits, in both directions."""
    import torch
    if matcher_mode not in ("prune_sub4", "prune_sub64"):
        pytest.skip("needs the skipping passes (upper bounds) and the default schedule")
    rng = np.random.default_rng(31337)
    centres = fpfh_like(rng, 32)
    def cloud(m):
        x = centres[rng.integers(0, 32, m)].astype(np.float64).reshape(m, 3, 11)
        x = np.abs(x + rng.normal(0, 1.5, x.shape)) + 1e-3
        return (100.0 * x / x.sum(2, keepdims=True)).reshape(m, 33).astype(np.float32)
    a = np.concatenate([cloud(5000), fpfh_like(rng, 7000)]); b = np.concatenate([fpfh_like(rng, 6000), cloud(8000)])
    a[5] = b[77]; a[6] = b[77]                                     # exact ties across the two sets
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    results = {}
    for name, extra in (("split", dict(split_sweep=1)), ("fused", dict(split_sweep=0)), ("overflow", dict(split_sweep=1, kept_cap=64))):
        opts(lgr, self_check=1, coarse_rejection=2, **extra)
        run_both(lgr, oracle, a, b, 4000)                          # both directions against the oracle
        results[name] = [t.cpu().numpy() for t in lgr.match_bf2(ta, tb, 4000)]
        lgr.sync()
        assert lgr.match_format() == "f16r"
        tested, abandoned = lgr.match_coarse()
        assert tested > 0 and 0 < abandoned < tested, (name, tested, abandoned)   # (some tiles are kept: more than the 64 the small list holds)
        assert tested - abandoned > 64 or name != "overflow", (tested, abandoned)
    for name in ("fused", "overflow"):
        for x, y in zip(results["split"], results[name]):
            np.testing.assert_array_equal(x.view(np.uint32), y.view(np.uint32))


def test_shell_bound(lgr, oracle, matcher_mode):
    """Rows and columns lie in radial shells about their cluster centre (they are sorted by that radius inside their leaves), and
    |a - b| >= | |a - c| - |b - c| |.  Pass 0 takes only the stages whose shell overlaps the row block's, the later passes drop stages
    -- and, inside the sweep, single tiles -- whose shell gap exceeds every upper bound (lgr_match_options.shell_bound).  None of
    that may show: same matches and distance bits as the oracle and as the run with the shell bound off, both directions and the
    single-direction entry point, with the device self-check of the filter bound in force; and on clustered rows with a spread of
    radii it really does leave tiles out."""
    import torch
    if matcher_mode not in ("prune_sub4", "prune_sub64"):
        pytest.skip("needs the skipping passes (upper bounds) and the default schedule")
    rng = np.random.default_rng(99)
    centres = fpfh_like(rng, 24)
    def cloud(m):
        # clusters whose members sit at very different distances from their centre: wide radial spread, close neighbours
        x = centres[rng.integers(0, 24, m)].astype(np.float64).reshape(m, 3, 11)
        x = np.abs(x + rng.normal(0, 1.0, x.shape) * rng.uniform(0.2, 6.0, (m, 1, 1))) + 1e-3
        return (100.0 * x / x.sum(2, keepdims=True)).reshape(m, 33).astype(np.float32)
    a = np.concatenate([cloud(9000), fpfh_like(rng, 3000)]); b = np.concatenate([fpfh_like(rng, 4000), cloud(11000)])
    a[17] = b[40]; a[18] = b[40]                                   # exact ties across the two sets
    opts(lgr, self_check=1, coarse_rejection=2)   # (2: the sweep also when the pass schedules most of the tiles)
    run_both(lgr, oracle, a, b, 4000)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    on = [t.cpu().numpy() for t in lgr.match_bf2(ta, tb, 4000)]
    lgr.sync()
    assert lgr.match_format() == "f16r"
    skipped, work_on = lgr.match_shell(), lgr.match_work()
    r_rows, r_cols = lgr.match_check()
    assert 0.0 <= r_rows <= 1.0 and 0.0 <= r_cols <= 1.0, (r_rows, r_cols)
    one = [t.cpu().numpy() for t in lgr.match_bf(ta, tb, 4000)]
    opts(lgr, self_check=1, shell_bound=0, coarse_rejection=2)
    off = [t.cpu().numpy() for t in lgr.match_bf2(ta, tb, 4000)]
    lgr.sync()
    assert lgr.match_shell() == 0.0
    work_off = lgr.match_work()
    r_rows, r_cols = lgr.match_check()
    assert 0.0 <= r_rows <= 1.0 and 0.0 <= r_cols <= 1.0, (r_rows, r_cols)
    for x, y in zip(on, off):
        np.testing.assert_array_equal(x.view(np.uint32), y.view(np.uint32))
    np.testing.assert_array_equal(one[0], on[0]); np.testing.assert_array_equal(one[1].view(np.uint32), on[1].view(np.uint32))
    assert skipped > 0 and work_on <= work_off, (skipped, work_on, work_off)
    print(f"shell bound [{matcher_mode}]: {skipped:.0f} tiles left out inside the sweep, stages computed {work_on:.4f} (off: {work_off:.4f})")


def test_rerank_refilter(lgr, oracle, matcher_mode):
    """The exact rerank re-filters each candidate group with the MFMA operands and takes the exact distance only for the
    (query, train row) pairs under the query's threshold.  Same result as scanning the whole groups and as the oracle, in
    both operand formats; far fewer exact distances than group rows; and when the pair buffer is too small the call falls
    back to the group scan (forced here) without changing anything."""
    import torch
    if matcher_mode not in ("auto", "prune_sub4"):
        pytest.skip("the rerank does not depend on the other schedule knobs")
    rng = np.random.default_rng(99)
    a, b = fpfh_like(rng, 9000), fpfh_like(rng, 11000)
    a[100] = b[7]; b[6000] = b[7]; a[101] = b[7]      # exact ties across bf blocks
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    outs = {}
    for fmt in ("f16r", "f16"):
        opts(lgr, operand_format=FMT[fmt])
        run_both(lgr, oracle, a, b, 3000)
        outs[fmt] = [t.cpu().numpy() for t in lgr.match_bf2(ta, tb, 3000)]
        lgr.sync()
        assert lgr.match_format() == fmt
        st = lgr.match_stats()
        p_ab, p_ba = lgr.match_pairs()
        assert 0 < p_ab < 8 * st["items_ab"] and 0 < p_ba < 8 * st["items_ba"], (p_ab, p_ba, st)
        opts(lgr, operand_format=FMT[fmt], pair_cap=100)          # overflow -> whole-group scan
        small = [t.cpu().numpy() for t in lgr.match_bf2(ta, tb, 3000)]
        lgr.sync()
        assert lgr.match_pairs()[0] > 100
        for x, y in zip(outs[fmt], small):
            np.testing.assert_array_equal(x.view(np.uint32), y.view(np.uint32))
    for x, y in zip(outs["f16r"], outs["f16"]):
        np.testing.assert_array_equal(x.view(np.uint32), y.view(np.uint32))
    print(f"rerank re-filter [{matcher_mode}]: {p_ab} + {p_ba} pairs for {st['items_ab']} + {st['items_ba']} candidate groups")


def test_rotated_format_selection(lgr, oracle, matcher_mode):
    """FPFH-like rows (every 11-bin block sums to 100) take the 30-coordinate operand format on their own; one row with a
    different block sum goes through the exact side scan (round 5) and the rest keeps the format -- with that lane switched off
    the row switches the call back to 33 coordinates; all give the oracle's result."""
    import torch
    if matcher_mode != "auto":
        pytest.skip("format selection does not depend on the skipping mode")
    rng = np.random.default_rng(123)
    a, b = fpfh_like(rng, 3000), fpfh_like(rng, 4000)
    run_both(lgr, oracle, a, b, 1000)
    assert lgr.match_format() == "f16r"
    b[77, 3] += 0.25
    run_both(lgr, oracle, a, b, 1000)
    assert lgr.match_format() == "f16r" and lgr.match_irregular() == (0, 1, 0)
    opts(lgr, irregular_rows=0)
    run_both(lgr, oracle, a, b, 1000)
    assert lgr.match_format() == "f16" and lgr.match_irregular() == (0, 0, 0)


def test_extreme_magnitudes_do_not_break_the_bounds(lgr, oracle, matcher_mode):
    """Rows of magnitude 1e18 beside ordinary ones: squared norms and the sample covariance of the box-bound basis overflow
    float.  Bounds that cannot be evaluated must switch themselves off, never skip a tile: results stay the oracle's."""
    rng = np.random.default_rng(31)
    a, b = fpfh_like(rng, 3000), fpfh_like(rng, 4000)
    a[::97] *= 1e18; b[::89] *= 1e18; b[5] = a[97]; a[11] = b[89 * 3]
    run_both(lgr, oracle, a, b, 1500)


def with_irregular(rng, a, n_zero, n_half=2):
    """a copy of FPFH-like rows with n_zero all-zero rows (what PCL writes for a point whose neighbours carry no weight) and n_half rows
    whose middle block sums to 50, at random positions; returns (rows, positions)"""
    a = a.copy()
    pos = rng.choice(a.shape[0], n_zero + n_half, replace=False)
    a[pos[:n_zero]] = 0.0
    a[pos[n_zero:], 11:22] *= 0.5
    return a, np.sort(pos)


def test_match_irregular_rows(lgr, oracle, matcher_mode):
    """Rows off the consensus of the block sums (round 5: lgr_match_options.irregular_rows) take the exact side scan -- as queries and as
    train rows, with the reference's tie rules (several zero rows on the train side: equal distances inside a bf block and across blocks),
    beside NaN rows -- and the rest keeps the rotated 30-coordinate format.  With the lane switched off the matches are the same."""
    import torch
    rng = np.random.default_rng(2024)
    a, pa = with_irregular(rng, clustered(rng, 9000, spread=0.3), 7)
    b, pb = with_irregular(rng, clustered(rng, 12000, spread=0.3), 11)
    # clustered() rows do not have constant block sums: make them so (the regular rows of both sets share the sums 100, 100, 100)
    def normalise(x, keep):
        y = x.reshape(-1, 3, 11).astype(np.float64)
        s = y.sum(2, keepdims=True)
        y = np.where(s > 0, 100.0 * y / np.where(s > 0, s, 1.0), y).reshape(-1, 33).astype(np.float32)
        y[keep] = x[keep]
        return y
    a, b = normalise(a, pa), normalise(b, pb)
    a[100, 3] = np.nan; b[200, :] = np.nan
    a[300] = b[pb[0]]                      # a zero query row: distance 0 to every zero train row -> tie rules of the side scan
    b[400] = a[301]                        # an ordinary exact duplicate beside it
    run_both(lgr, oracle, a, b, 2500)
    lgr.match_bf2(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), 2500)
    lgr.sync()
    def n_irregular(x):
        fin = np.isfinite(x).all(1)
        return int((fin & (np.abs(x.reshape(-1, 3, 11).astype(np.float64).sum(2) - 100.0) > 1e-3).any(1)).sum())
    assert n_irregular(a) >= len(pa) and n_irregular(b) >= len(pb) - 1
    assert lgr.match_irregular() == (n_irregular(a), n_irregular(b), 0)
    if not matcher_mode.startswith("auto"):
        assert lgr.match_format() == "f16r"
    opts(lgr, irregular_rows=0)
    run_both(lgr, oracle, a, b, 2500)
    lgr.match_bf2(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), 2500)
    lgr.sync()
    assert lgr.match_irregular() == (0, 0, 0)


def test_match_irregular_rows_give_up(lgr, oracle, matcher_mode):
    """More irregular rows than the side scan's list holds (1024 per side), and a side that consists of nothing else: the call is
    rebuilt with every finite row in the operands; no consensus at all (rows of arbitrary floats): the lane stays off.  Same matches."""
    import torch
    if matcher_mode not in ("auto", "prune_sub4"):
        pytest.skip("host-side decision: one automatic and one forced-skipping mode")
    rng = np.random.default_rng(31)
    a = fpfh_like(rng, 2500)
    b = fpfh_like(rng, 230000)
    b[rng.choice(b.shape[0], 1100, replace=False)] = 0.0     # 0.48 % of the rows: a consensus exists, the list overflows
    run_both(lgr, oracle, a, b, 100000)
    assert lgr.match_irregular() == (0, 0, 1)                 # gave up: the rebuilt call has them in the operands
    assert lgr.match_format() != "f16r"
    z = np.zeros((40, 33), np.float32)                        # a query side of zero rows only, among regular train rows
    run_both(lgr, oracle, z, fpfh_like(rng, 3000), 1000)
    x = rng.normal(size=(3000, 33)).astype(np.float32)        # no consensus
    run_both(lgr, oracle, x, rng.normal(size=(2000, 33)).astype(np.float32), 1000)
    assert lgr.match_irregular() == (0, 0, 0)

