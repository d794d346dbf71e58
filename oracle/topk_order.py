"""Oracle: WHICH of several equal keys `torch.topk(x, k, largest=False)` returns on the CPU.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  The reference's n:m rules call it on the m scores of a group
(/root/reference/lavis/compression/pruners/wanda_pruner.py:326-329, :671-677; dsnot_pruner.py:517-519 with k = 1).  The
arithmetic lives in PyTorch (not under /root/reference): ATen's CPU kernel (aten/src/ATen/native/cpu/TopKImpl.h) fills a
vector of (value, index) pairs and, for k * 64 > size -- always, for m <= 8 --, runs `std::nth_element(begin, begin + k - 1,
end, cmp)` with cmp(x, y) = (!isnan(x) && isnan(y)) || x < y; the k pairs left in front are the answer.  libstdc++'s
nth_element is introselect: median-of-three (first + 1, middle, last - 1 -> first) + unguarded partition while more than 3
elements remain, then an insertion sort (bits/stl_algo.h: __introselect, __move_median_to_first, __unguarded_partition,
__insertion_sort).  Restated here move for move; pinned by tests/test_nm_ties.py against the reference's recorded masks
(tests/golden/nm_ties.npz) and against this container's torch.topk on random tie patterns."""
import math


def _lt(x, y):
    return ((not math.isnan(x[0])) and math.isnan(y[0])) or (x[0] < y[0])


def _median_to_first(q, result, a, b, c):
    if _lt(q[a], q[b]):
        if _lt(q[b], q[c]):
            pick = b
        elif _lt(q[a], q[c]):
            pick = c
        else:
            pick = a
    elif _lt(q[a], q[c]):
        pick = a
    elif _lt(q[b], q[c]):
        pick = c
    else:
        pick = b
    q[result], q[pick] = q[pick], q[result]


def _partition(q, first, last, pivot):
    while True:
        while _lt(q[first], q[pivot]):
            first += 1
        last -= 1
        while _lt(q[pivot], q[last]):
            last -= 1
        if not first < last:
            return first
        q[first], q[last] = q[last], q[first]
        first += 1


def _insertion_sort(q, first, last):
    for i in range(first + 1, last):
        val = q[i]
        if _lt(val, q[first]):
            q[first + 1:i + 1] = q[first:i]
            q[first] = val
        else:
            j = i
            while _lt(val, q[j - 1]):
                q[j] = q[j - 1]
                j -= 1
            q[j] = val


def nth_element(q, nth):
    """In place on a list of (value, index) pairs, as std::nth_element(q.begin(), q.begin() + nth, q.end(), cmp)."""
    first, last = 0, len(q)
    if first == last or nth == last:
        return
    depth = int(math.log2(len(q))) * 2
    while last - first > 3:
        if depth == 0:
            raise NotImplementedError("heap-select fallback: not reachable for the group sizes of the n:m rules (m <= 8)")
        depth -= 1
        mid = first + (last - first) // 2
        _median_to_first(q, first, first + 1, mid, last - 1)
        cut = _partition(q, first + 1, last, first)
        if cut <= nth:
            first = cut
        else:
            last = cut
    _insertion_sort(q, first, last)


def smallest(values, k):
    """Indices (a sorted list) that `torch.topk(values, k, largest=False)` returns on the CPU."""
    if k == 0:
        return []
    q = [(float(v), j) for j, v in enumerate(values)]
    nth_element(q, k - 1)
    return sorted(j for _, j in q[:k])
