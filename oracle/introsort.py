"""Restatement of libstdc++'s std::sort (introsort: bits/stl_algo.h __sort = __introsort_loop +
__final_insertion_sort, threshold 16, heap-sort fallback at depth 2*lg(n)) over (key, index) pairs -- TEST
INFRASTRUCTURE, not product code.

Why it exists: the reference's BuFF sampler calls torch.sort with its unstable default three times
(/root/reference/src/nerf/tree.py:300,306,335); torch's CPU kernel implements that as
`std::sort(composite key/index accessor, KeyValueCompAsc|Desc)` (ATen/native/cpu/SortingKernel.cpp), so the
reference's voxel ids depend on how introsort orders ties.  This file pins that claim
(tests/test_introsort_restatement.py compares it with torch.sort on tie-heavy inputs, NaNs and an adversarial input
that reaches the heap fallback); the product's NM_TIES_REFERENCE kernel (nerfmeshes_amd/csrc/buff_tree.hip) is an
independent restatement of the same algorithm.
"""
import math


def comp_asc(a, b):      # KeyValueCompAsc: NaNs last
    return ((not math.isnan(a)) and math.isnan(b)) or a < b


def comp_desc(a, b):     # KeyValueCompDesc: NaNs first
    return (math.isnan(a) and not math.isnan(b)) or a > b


def introsort(keys, comp, count_compares=None):
    """-> (sorted keys, source indices) exactly as std::sort leaves them."""
    n = len(keys)
    k, ix = list(keys), list(range(n))
    stats = {"heap_fallbacks": 0}

    def c(a, b):
        return comp(a, b)

    def swap(i, j):
        k[i], k[j] = k[j], k[i]
        ix[i], ix[j] = ix[j], ix[i]

    def move_median_to_first(result, a, b, cc):
        if c(k[a], k[b]):
            if c(k[b], k[cc]):
                swap(result, b)
            elif c(k[a], k[cc]):
                swap(result, cc)
            else:
                swap(result, a)
        elif c(k[a], k[cc]):
            swap(result, a)
        elif c(k[b], k[cc]):
            swap(result, cc)
        else:
            swap(result, b)

    def unguarded_partition(first, last, pivot):
        while True:
            while c(k[first], k[pivot]):
                first += 1
            last -= 1
            while c(k[pivot], k[last]):
                last -= 1
            if not first < last:
                return first
            swap(first, last)
            first += 1

    def push_heap(first, hole, top, vk, vi):
        parent = (hole - 1) // 2
        while hole > top and c(k[first + parent], vk):
            k[first + hole], ix[first + hole] = k[first + parent], ix[first + parent]
            hole = parent
            parent = (hole - 1) // 2
        k[first + hole], ix[first + hole] = vk, vi

    def adjust_heap(first, hole, length, vk, vi):
        top, child = hole, hole
        while child < (length - 1) // 2:
            child = 2 * (child + 1)
            if c(k[first + child], k[first + child - 1]):
                child -= 1
            k[first + hole], ix[first + hole] = k[first + child], ix[first + child]
            hole = child
        if length % 2 == 0 and child == (length - 2) // 2:
            child = 2 * (child + 1)
            k[first + hole], ix[first + hole] = k[first + child - 1], ix[first + child - 1]
            hole = child - 1
        push_heap(first, hole, top, vk, vi)

    def heap_sort(first, last):
        stats["heap_fallbacks"] += 1
        length = last - first
        if length >= 2:
            parent = (length - 2) // 2
            while True:
                adjust_heap(first, parent, length, k[first + parent], ix[first + parent])
                if parent == 0:
                    break
                parent -= 1
        while last - first > 1:
            last -= 1
            vk, vi = k[last], ix[last]
            k[last], ix[last] = k[first], ix[first]
            adjust_heap(first, 0, last - first, vk, vi)

    def loop(first, last, depth):
        while last - first > 16:
            if depth == 0:
                heap_sort(first, last)
                return
            depth -= 1
            mid = first + (last - first) // 2
            move_median_to_first(first, first + 1, mid, last - 1)
            cut = unguarded_partition(first + 1, last, first)
            loop(cut, last, depth)
            last = cut

    def unguarded_linear_insert(last):
        vk, vi = k[last], ix[last]
        nxt = last - 1
        while c(vk, k[nxt]):
            k[last], ix[last] = k[nxt], ix[nxt]
            last, nxt = nxt, nxt - 1
        k[last], ix[last] = vk, vi

    def insertion_sort(first, last):
        for i in range(first + 1, last):
            if c(k[i], k[first]):
                vk, vi = k[i], ix[i]
                k[first + 1:i + 1], ix[first + 1:i + 1] = k[first:i], ix[first:i]
                k[first], ix[first] = vk, vi
            else:
                unguarded_linear_insert(i)

    if n:
        loop(0, n, 2 * (n.bit_length() - 1))
        if n > 16:
            insertion_sort(0, 16)
            for i in range(16, n):
                unguarded_linear_insert(i)
        else:
            insertion_sort(0, n)
    if count_compares is not None:
        count_compares.update(stats)
    return k, ix


def quicksort_killer(n):
    """McIlroy's adversary ("A Killer Adversary for Quicksort", 1999) run against `introsort` itself: returns float
    keys on which the median-of-3 partitioning degenerates, so the depth limit is reached and the heap fallback runs."""
    gas = n                      # value of undecided ("gas") items
    val = [gas] * n
    state = {"solid": 0, "candidate": 0}

    class Key:                   # comparisons decide values lazily
        __slots__ = ("i",)

        def __init__(self, i):
            self.i = i

    def comp(a, b):
        x, y = a.i, b.i
        if val[x] == gas and val[y] == gas:
            if x == state["candidate"]:
                val[x] = state["solid"]
            else:
                val[y] = state["solid"]
            state["solid"] += 1
        if val[x] == gas:
            state["candidate"] = x
        elif val[y] == gas:
            state["candidate"] = y
        return val[x] < val[y]

    introsort([Key(i) for i in range(n)], comp)
    return [float(v) for v in val]
