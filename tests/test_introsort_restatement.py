"""torch.sort's unstable CPU default == libstdc++ std::sort over (key, index) pairs (oracle/introsort.py): the
claim the NM_TIES_REFERENCE mode of the BuFF sampler rests on (R9: the reference's voxel ids are what THIS
algorithm does with ties)."""
import math

import torch

from oracle import introsort as I


def _check(x, descending):
    t = torch.sort(x, descending=descending)
    keys, ix = I.introsort(x.tolist(), I.comp_desc if descending else I.comp_asc)
    assert ix == t.indices.tolist()
    got = torch.tensor(keys, dtype=x.dtype)
    assert torch.equal(torch.nan_to_num(got.float(), nan=-7.0), torch.nan_to_num(t.values.float(), nan=-7.0))


def test_hit_mask_sequences():
    g = torch.Generator().manual_seed(1)
    for n in (1, 2, 15, 16, 17, 33, 192, 500, 1536, 1728, 4096):
        for p in (0.0, 0.01, 0.02, 0.3, 1.0):
            _check((torch.rand(n, generator=g) < p).long(), True)


def test_float_keys_with_ties_and_nans():
    g = torch.Generator().manual_seed(2)
    for n in (17, 192, 1728):
        x = torch.randint(0, 12, (n,), generator=g).float() * 0.1        # slab-entry-like: few distinct values
        _check(x, False)
        _check(x, True)
        x[torch.randperm(n, generator=g)[: n // 10]] = float("nan")
        x[0] = float("inf"); x[-1] = -float("inf")
        _check(x, False)
        _check(x, True)
    _check(torch.rand(1728, generator=g), False)


def test_batched_rows_sort_independently():
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(7, 1728, generator=g) < 0.02).long()
    t = x.sort(-1, descending=True)
    for r in range(7):
        assert I.introsort(x[r].tolist(), I.comp_desc)[1] == t.indices[r].tolist()


def test_depth_limit_heap_fallback_matches_too():
    """An adversarial input (McIlroy's killer, generated against the restatement itself) drives the median-of-3
    quicksort to its depth limit; std::sort then heap-sorts the range -- restated as well."""
    for n in (300, 1728):
        keys = I.quicksort_killer(n)
        stats = {}
        _, ix = I.introsort(keys, I.comp_asc, stats)
        assert stats["heap_fallbacks"] >= 1, "the adversary should reach the depth limit"
        x = torch.tensor(keys)
        assert ix == torch.sort(x).indices.tolist()
        assert math.isfinite(sum(keys))
