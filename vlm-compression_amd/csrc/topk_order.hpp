// WHICH of several equal keys the reference's `torch.topk(x, n, largest=False)` returns on the CPU (the n:m rules:
// wanda_pruner.py:326-329, :671-677; dsnot_pruner.py:517-519 with n = 1).  ATen's CPU kernel runs libstdc++'s
// std::nth_element on (value, index) pairs -- introselect: median of (first + 1, middle, last - 1) moved to first,
// unguarded partition, repeated while more than 3 elements remain, then an insertion sort -- and returns the n pairs it
// leaves in front.  Restated move for move (oracle/topk_order.py is the CPU restatement, pinned against the reference's
// recorded masks in tests/golden/nm_ties.npz); keys are the order-preserving unsigned images of the scores (NaN last).
// The elements live in registers: a run-time index is a chain of selects over the M <= 8 slots (the tie path is rare: a
// group whose n-th and (n+1)-th smallest keys are equal).
#pragma once
#include <cstdint>

namespace vlmc {

template <int M> struct SmallQueue {
    uint32_t key[M], idx[M];
    __device__ __forceinline__ uint32_t k(int i) const {
        uint32_t v = 0;
#pragma unroll
        for (int j = 0; j < M; ++j) v = j == i ? key[j] : v;
        return v;
    }
    __device__ __forceinline__ uint32_t ix(int i) const {
        uint32_t v = 0;
#pragma unroll
        for (int j = 0; j < M; ++j) v = j == i ? idx[j] : v;
        return v;
    }
    __device__ __forceinline__ void set(int i, uint32_t kk, uint32_t ii) {
#pragma unroll
        for (int j = 0; j < M; ++j) {
            key[j] = j == i ? kk : key[j];
            idx[j] = j == i ? ii : idx[j];
        }
    }
    __device__ __forceinline__ void swap(int a, int b) {
        const uint32_t ka = k(a), ia = ix(a), kb = k(b), ib = ix(b);
        set(a, kb, ib);
        set(b, ka, ia);
    }
    __device__ __forceinline__ bool lt(int a, int b) const { return k(a) < k(b); }
};

// bit i of the result: column i is among the n that torch.topk(keys, n, largest=False) returns on the CPU
template <int M> __device__ __forceinline__ uint32_t torch_cpu_smallest(const uint32_t (&keys)[M], int n) {
    if (n <= 0) return 0u;
    SmallQueue<M> q;
#pragma unroll
    for (int j = 0; j < M; ++j) {
        q.key[j] = keys[j];
        q.idx[j] = uint32_t(j);
    }
    int first = 0, last = M;
    const int nth = n - 1;
    while (last - first > 3) {                                            // (the depth limit 2 lg M is not reached for M <= 8)
        const int a = first + 1, b = first + (last - first) / 2, c = last - 1;
        int pick;                                                         // median of a, b, c -> first
        if (q.lt(a, b)) pick = q.lt(b, c) ? b : (q.lt(a, c) ? c : a);
        else pick = q.lt(a, c) ? a : (q.lt(b, c) ? c : b);
        q.swap(first, pick);
        int f = first + 1, l = last;                                      // unguarded partition around the pivot at `first`
        for (;;) {
            while (q.lt(f, first)) ++f;
            --l;
            while (q.lt(first, l)) --l;
            if (!(f < l)) break;
            q.swap(f, l);
            ++f;
        }
        if (f <= nth) first = f;
        else last = f;
    }
    for (int i = first + 1; i < last; ++i) {                              // insertion sort of what is left
        const uint32_t vk = q.k(i), vi = q.ix(i);
        int j = i;
        if (vk < q.k(first)) {
            for (; j > first; --j) q.set(j, q.k(j - 1), q.ix(j - 1));
        } else {
            while (vk < q.k(j - 1)) {
                q.set(j, q.k(j - 1), q.ix(j - 1));
                --j;
            }
        }
        q.set(j, vk, vi);
    }
    uint32_t bits = 0;
#pragma unroll
    for (int j = 0; j < M; ++j) bits |= j < n ? 1u << q.idx[j] : 0u;
    return bits;
}

// index of the entry `torch.topk(v[0 .. m), 1, largest=False)` returns on the CPU, m in {2, 4, 8} (dsnot_pruner.py:517-519)
__device__ __forceinline__ int torch_cpu_argmin(const uint32_t (&v)[8], int m) {
    uint32_t mn = 0xFFFFFFFFu;
    int first = 0, cnt = 0;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        if (a < m) {
            if (v[a] < mn || (a == 0)) {
                mn = v[a];
                first = a;
                cnt = 1;
            } else if (v[a] == mn) {
                ++cnt;
            }
        }
    }
    if (cnt <= 1 || m <= 3) return first;                                 // unique, or an insertion sort: the lowest index
    if (m == 4) {
        const uint32_t k4[4] = {v[0], v[1], v[2], v[3]};
        return __builtin_ctz(torch_cpu_smallest<4>(k4, 1));
    }
    return __builtin_ctz(torch_cpu_smallest<8>(v, 1));
}

}  // namespace vlmc
