"""The stable radix sort under every grid / voxel / placement step (csrc/lgr_sort.hip), through the C ABI: against numpy's stable
sort on the same masked keys -- ragged sizes around the 2048-element workgroup tile, narrow and full bit ranges, heavy duplicates,
all-equal keys, 64-bit keys with three fields and the all-ones key of invalid points."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check32(lgr, torch, keys, begin, end):
    n = len(keys)
    vals = np.arange(n, dtype=np.int32)[::-1].copy()            # values not in index order: stability is about positions
    ko, vo = lgr.sort_pairs_u32(torch.from_numpy(keys.view(np.int32)).cuda(), torch.from_numpy(vals).cuda(), begin, end)
    lgr.sync()
    mask = np.uint32(((1 << (end - begin)) - 1) << begin) if end > begin else np.uint32(0)
    order = np.argsort(keys & mask, kind="stable")
    assert np.array_equal(ko.cpu().numpy().view(np.uint32), keys[order])
    assert np.array_equal(vo.cpu().numpy(), vals[order])


@pytest.mark.parametrize("n", [1, 63, 64, 2047, 2048, 2049, 4097, 100003, 1000000])
def test_sort_u32_sizes(lgr, n):
    import torch
    rng = np.random.default_rng(n)
    _check32(lgr, torch, rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32), 0, 32)


@pytest.mark.parametrize("begin,end", [(0, 1), (0, 8), (0, 9), (0, 21), (5, 28), (3, 3), (24, 32)])
def test_sort_u32_bit_ranges(lgr, begin, end):
    import torch
    rng = np.random.default_rng(begin * 40 + end)
    _check32(lgr, torch, rng.integers(0, 2**32, 70001, dtype=np.uint64).astype(np.uint32), begin, end)


def test_sort_u32_duplicates_and_constant_keys(lgr):
    import torch
    rng = np.random.default_rng(5)
    _check32(lgr, torch, rng.integers(0, 7, 300000, dtype=np.uint64).astype(np.uint32), 0, 32)       # seven distinct keys
    _check32(lgr, torch, np.full(50000, 0xDEADBEEF, np.uint32), 0, 32)                                # one key: the identity
    k = rng.integers(0, 2**20, 200000, dtype=np.uint64).astype(np.uint32)
    k[rng.integers(0, 200000, 5000)] = 0xFFFFFFFF                                                     # invalid rows sort last
    _check32(lgr, torch, k, 0, 32)


def test_sort_u64_voxel_keys(lgr):
    import torch
    rng = np.random.default_rng(9)
    n = 500001
    ix, iy, iz = rng.integers(0, 1500, n), rng.integers(0, 900, n), rng.integers(0, 300, n)
    keys = (iz.astype(np.uint64) << np.uint64(42)) | (iy.astype(np.uint64) << np.uint64(21)) | ix.astype(np.uint64)
    keys[rng.integers(0, n, 1000)] = np.uint64(0xFFFFFFFFFFFFFFFF)
    vals = rng.permutation(n).astype(np.int32)
    ranges = [(0, 11), (21, 10), (42, 10)]                                                            # z one bit wider: invalid keys last
    ko, vo = lgr.sort_pairs_u64(torch.from_numpy(keys.view(np.int64)).cuda(), torch.from_numpy(vals).cuda(), ranges)
    lgr.sync()
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(ko.cpu().numpy().view(np.uint64), keys[order])
    assert np.array_equal(vo.cpu().numpy(), vals[order])
    ko, vo = lgr.sort_pairs_u64(torch.from_numpy(keys.view(np.int64)).cuda(), torch.from_numpy(vals).cuda(), [])   # nothing to sort by: a copy
    lgr.sync()
    assert np.array_equal(ko.cpu().numpy().view(np.uint64), keys) and np.array_equal(vo.cpu().numpy(), vals)
