"""Restatement of numpy's fp32 `add.reduce` / mean / std (TEST INFRASTRUCTURE, not product code): the reduction
iterator hands the flat array to the inner loop in buffer-sized chunks of 8192 elements whose sums are accumulated
sequentially; inside a chunk FLOAT_pairwise_sum (numpy/core/src/umath/loops_utils.h.src) recurses
n -> (n/2 rounded down to a multiple of 8, rest) down to blocks of <= 128 elements with 8 interleaved accumulators.
Pinned against numpy itself in tests/test_np_reduce_restatement.py; the product's nm_np_stats (np_reduce.hip) is an
independent restatement of the same scheme (the reference picks the marching-cubes level from these statistics,
/root/reference/src/mesh_nerf.py:56-65)."""
import numpy as np

f32 = np.float32
CHUNK = 8192


def pairwise(a):
    n = len(a)
    if n < 8:
        r = f32(0.0)
        for x in a:
            r = f32(r + x)
        return r
    if n <= 128:
        r = [f32(a[j]) for j in range(8)]
        i = 8
        while i < n - (n % 8):
            for j in range(8):
                r[j] = f32(r[j] + a[i + j])
            i += 8
        res = f32(f32(f32(r[0] + r[1]) + f32(r[2] + r[3])) + f32(f32(r[4] + r[5]) + f32(r[6] + r[7])))
        while i < n:
            res = f32(res + a[i])
            i += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return f32(pairwise(a[:n2]) + pairwise(a[n2:]))


def np_sum(a):
    a = np.ascontiguousarray(a, dtype=f32).ravel()
    res = None
    for s in range(0, len(a), CHUNK):
        c = pairwise(a[s:s + CHUNK])
        res = c if res is None else f32(res + c)
    return res


def np_mean_std(a):
    a = np.ascontiguousarray(a, dtype=f32).ravel()
    mean = f32(np_sum(a) / f32(a.size))
    dev = a - mean
    var = f32(np_sum(dev * dev) / f32(a.size))
    return mean, np.sqrt(var)
