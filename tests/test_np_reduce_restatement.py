"""numpy's fp32 sum / mean / std == the chunked pairwise scheme restated in oracle/np_pairwise.py (what nm_np_stats
replays on the GPU so that mesh_nerf's adaptive iso level is the reference's bit for bit)."""
import numpy as np
import pytest

from oracle import np_pairwise as P


@pytest.mark.parametrize("n", [1, 5, 8, 9, 100, 128, 129, 1000, 4097, 8192, 8193, 16384, 16385, 3 * 8192 + 77, 100003])
def test_sum_mean_std_match_numpy(n):
    rng = np.random.default_rng(n)
    a = (rng.standard_normal(n) * 100 + 20).astype(np.float32)
    assert P.np_sum(a) == a.sum()
    mean, std = P.np_mean_std(a)
    assert mean == a.mean() and std == a.std()


def test_strided_view_like_the_reference_density():
    """mesh_nerf.py:73 takes radiance[..., 3] -- a strided view of an (n,n,n,4) array: same flat order, same chunks."""
    rng = np.random.default_rng(0)
    rad = (rng.standard_normal((23, 23, 23, 4)) * 50).astype(np.float32)
    d = rad[..., 3]
    mean, std = P.np_mean_std(d)
    assert mean == d.mean() and std == d.std() and P.np_sum(d) == d.sum()
