/* Oracle (C restatement): Wanda statistics, score and mask selection on the CPU.
 *
 * TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  An independent, torch-free restatement
 * of the same reference lines as oracle/wanda.py, used to cross-check that oracle, to run the
 * larger parity cases in seconds, and (with OpenMP over rows / channels) as a compiled CPU
 * baseline.  Built by oracle/Makefile into oracle/_build/libwanda_oracle.so.
 *
 * Reference: /root/reference/lavis/compression/pruners/wanda_pruner.py
 *   wo_act_sqnorm      :73-81   sqrtf(sequential fmaf chain over tokens) squared
 *   wo_scaler_update   :77-81   s *= float(n/(n+b)); n += b; s += normsq / float(n)
 *   wo_select          :318-341 (row rule), :666-687 (matrix rule), :326-329 (n:m)
 * Built with -ffp-contract=off: every fp32 operation is a single IEEE operation.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { WO_F32 = 0, WO_F16 = 1, WO_BF16 = 2 };

static float half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1Fu, man = h & 0x3FFu, bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { /* subnormal: normalise */
            int e = -1;
            do { man <<= 1; ++e; } while (!(man & 0x400u));
            bits = sign | (uint32_t)(127 - 15 - e) << 23 | (man & 0x3FFu) << 13;
        }
    } else if (exp == 31) {
        bits = sign | 0x7F800000u | man << 13;
    } else {
        bits = sign | (exp + 127 - 15) << 23 | man << 13;
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}
static float load_elem(const void *p, int dtype, int64_t i) {
    if (dtype == WO_F32) return ((const float *)p)[i];
    uint16_t h = ((const uint16_t *)p)[i];
    if (dtype == WO_F16) return half_to_float(h);
    uint32_t b = (uint32_t)h << 16;
    float f;
    memcpy(&f, &b, 4);
    return f;
}
static void store_zero(void *p, int dtype, int64_t i) {
    if (dtype == WO_F32) ((float *)p)[i] = 0.0f; else ((uint16_t *)p)[i] = 0;
}

/* normsq[c] = (sqrtf(sum_t x[t,c]^2))^2, tokens reduced sequentially with one fma each */
void wo_act_sqnorm(const void *x, int dtype, int64_t tokens, int64_t in_f, float *out) {
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < in_f; ++c) {
        float acc = 0.0f;
        for (int64_t t = 0; t < tokens; ++t) {
            float v = load_elem(x, dtype, t * in_f + c);
            acc = fmaf(v, v, acc);
        }
        float r = sqrtf(acc);
        out[c] = r * r;
    }
}

void wo_scaler_update(float *s, int64_t in_f, int64_t n0, const float *normsq, int64_t calls, int64_t batch) {
    for (int64_t c = 0; c < calls; ++c) {
        float f = (float)((double)n0 / (double)(n0 + batch));
        n0 += batch;
        float dn = (float)n0;
        for (int64_t ch = 0; ch < in_f; ++ch) {
            float a = s[ch] * f;
            s[ch] = a + normsq[c * in_f + ch] / dn;
        }
    }
}

/* order-preserving key of a score (>= +0 or NaN); NaN sorts last like torch.sort */
static uint32_t score_key(float sc) {
    uint32_t b;
    memcpy(&b, &sc, 4);
    if (sc != sc) return 0xFFFFFFFFu;
    return b & 0x7FFFFFFFu;
}

typedef struct { uint32_t key; uint32_t idx; } kv_t;

/* stable merge sort of (key, idx) by key: equal keys keep index order */
static void merge_sort(kv_t *a, kv_t *tmp, int64_t n) {
    for (int64_t w = 1; w < n; w *= 2) {
        for (int64_t lo = 0; lo < n; lo += 2 * w) {
            int64_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            int64_t i = lo, j = mid, o = lo;
            while (i < mid && j < hi) tmp[o++] = (a[j].key < a[i].key) ? a[j++] : a[i++];
            while (i < mid) tmp[o++] = a[i++];
            while (j < hi) tmp[o++] = a[j++];
        }
        memcpy(a, tmp, (size_t)n * sizeof(kv_t));
    }
}
static int cmp_u32(const void *a, const void *b) {
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/* mode 0: per-row k smallest (stable); 1: matrix-wide strict threshold at flat rank k; 2: n of m.
 * mask: 1 = keep.  score_sum (optional) receives sum(score) in double.  Returns 0 / -1. */
/* libstdc++'s std::nth_element (bits/stl_algo.h: __introselect) on (key, index) pairs ordered by key: median of
   (first + 1, middle, last - 1) to first, unguarded partition, while more than 3 elements remain; then an insertion sort.
   (The depth limit 2 lg(len) is not reached for the group sizes of the n:m rules, m <= 8.) */
static void swap_kv(kv_t *a, kv_t *b) { kv_t t = *a; *a = *b; *b = t; }
static void nth_element_kv(kv_t *q, int len, int nth) {
    int first = 0, last = len;
    while (last - first > 3) {
        int a = first + 1, b = first + (last - first) / 2, c = last - 1, pick;
        if (q[a].key < q[b].key) pick = q[b].key < q[c].key ? b : (q[a].key < q[c].key ? c : a);
        else pick = q[a].key < q[c].key ? a : (q[b].key < q[c].key ? c : b);
        swap_kv(&q[first], &q[pick]);
        int f = first + 1, l = last;
        for (;;) {
            while (q[f].key < q[first].key) ++f;
            --l;
            while (q[first].key < q[l].key) --l;
            if (!(f < l)) break;
            swap_kv(&q[f], &q[l]);
            ++f;
        }
        if (f <= nth) first = f; else last = f;
    }
    for (int i = first + 1; i < last; ++i) {
        kv_t v = q[i];
        int j = i;
        if (v.key < q[first].key) {
            for (; j > first; --j) q[j] = q[j - 1];
        } else {
            while (v.key < q[j - 1].key) { q[j] = q[j - 1]; --j; }
        }
        q[j] = v;
    }
}

int wo_select(void *W, int dtype, int64_t out_f, int64_t in_f, const float *scaler_row, int mode, int64_t k, int n, int m,
              int apply_zero, uint8_t *mask, double *score_sum) {
    float *sq = (float *)malloc((size_t)in_f * sizeof(float));
    if (!sq) return -1;
    for (int64_t c = 0; c < in_f; ++c) sq[c] = sqrtf(scaler_row[c]);
    double total = 0.0;
    uint32_t thr = 0;
    if (mode == 1) {
        int64_t numel = out_f * in_f;
        uint32_t *all = (uint32_t *)malloc((size_t)numel * sizeof(uint32_t));
        if (!all) { free(sq); return -1; }
        for (int64_t i = 0; i < numel; ++i) all[i] = score_key(fabsf(load_elem(W, dtype, i)) * sq[i % in_f]);
        qsort(all, (size_t)numel, sizeof(uint32_t), cmp_u32);
        thr = all[k];
        free(all);
    }
    int fail = 0;
#pragma omp parallel for schedule(static) reduction(+ : total) reduction(| : fail)
    for (int64_t r = 0; r < out_f; ++r) {
        kv_t *kv = (kv_t *)malloc((size_t)in_f * 2 * sizeof(kv_t));
        if (!kv) { fail = 1; continue; }
        double rs = 0.0;
        for (int64_t c = 0; c < in_f; ++c) {
            float sc = fabsf(load_elem(W, dtype, r * in_f + c)) * sq[c];
            rs += (double)sc;
            kv[c].key = score_key(sc);
            kv[c].idx = (uint32_t)c;
            mask[r * in_f + c] = 1;
        }
        total += rs;
        if (mode == 0) {
            merge_sort(kv, kv + in_f, in_f);
            for (int64_t i = 0; i < k; ++i) mask[r * in_f + kv[i].idx] = 0;
        } else if (mode == 1) {
            /* `score < thr` with a NaN threshold is false everywhere */
            if (thr != 0xFFFFFFFFu)
                for (int64_t c = 0; c < in_f; ++c)
                    if (kv[c].key < thr) mask[r * in_f + c] = 0;
        } else {
            /* torch.topk(group, n, largest=False) on the CPU: std::nth_element(begin, begin + n - 1, end) on (value, index)
               pairs, the n pairs in front are the answer -- equal keys in the order libstdc++'s introselect leaves them
               (oracle/topk_order.py has the provenance) */
            for (int64_t g = 0; g < in_f; g += m) {
                if (n > 0) nth_element_kv(kv + g, m, n - 1);
                for (int i = 0; i < n; ++i) mask[r * in_f + kv[g + i].idx] = 0;
            }
        }
        if (apply_zero)
            for (int64_t c = 0; c < in_f; ++c)
                if (!mask[r * in_f + c]) store_zero(W, dtype, r * in_f + c);
        free(kv);
    }
    free(sq);
    if (score_sum) *score_sum = total;
    return fail ? -1 : 0;
}
