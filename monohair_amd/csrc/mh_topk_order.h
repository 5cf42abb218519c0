// mh_topk_order.h -- the order in which torch.topk (CPU) returns tied values, as a device function.
//
// PMVO.Find_max_conf_from_visible_view (PMVO.py:339-343) calls torch.topk(k=20, dim=0) on a [V,N] tensor.  ATen's CPU
// kernel (aten/src/ATen/native/cpu/SortingKernel.cpp -> TopKImpl.h: topk_impl_loop) fills a vector of (value, index)
// pairs per column and, for k*64 > V, runs   std::nth_element(begin, begin+k-1, end, gt)   followed by
// std::sort(begin, begin+k-1, gt),   with gt(x, y) = (isnan(x) && !isnan(y)) || x > y  on the values only.  Neither
// algorithm is stable, so which of several equal values comes first is decided by libstdc++'s introselect / introsort
// (median-of-three pivot, Hoare partition, insertion sort below 4 resp. 16 elements, heap fall-back when the depth
// limit 2*floor(log2(n)) is used up).  Real captures reach PMVO through 8-bit confidence maps that saturate at 1.0, so
// such ties are the rule, and the choice of base views decides which candidates the search tries.  This header restates
// those algorithms step by step on an array of MhTkE; oracle/topk_oracle.cpp calls the real std:: functions, and the
// tests demand identical index arrays (tie-heavy random columns, adversarial columns that exhaust the depth limit, the
// reference's own rankings in tests/golden/).
#pragma once

#define MH_TK_FN __device__ static inline

struct MhTkE {
    float v;
    int i;
};

MH_TK_FN bool mh_tk_gt(const MhTkE &x, const MhTkE &y) {
    const bool xn = x.v != x.v, yn = y.v != y.v;
    return (xn && !yn) || (x.v > y.v);
}
MH_TK_FN void mh_tk_swap(MhTkE *a, int p, int q) {
    const MhTkE t = a[p];
    a[p] = a[q];
    a[q] = t;
}
MH_TK_FN int mh_tk_lg(int n) {   // floor(log2(n)), n >= 1
    int k = 0;
    while (n > 1) {
        n >>= 1;
        ++k;
    }
    return k;
}

// ---- heap primitives (max-heap with respect to gt: the element that compares "less" than all others is on top)
MH_TK_FN void mh_tk_push_heap(MhTkE *a, int first, int hole, int top, MhTkE value) {
    int parent = (hole - 1) / 2;
    while (hole > top && mh_tk_gt(a[first + parent], value)) {
        a[first + hole] = a[first + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    a[first + hole] = value;
}
MH_TK_FN void mh_tk_adjust_heap(MhTkE *a, int first, int hole, int len, MhTkE value) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (mh_tk_gt(a[first + child], a[first + child - 1])) --child;
        a[first + hole] = a[first + child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        a[first + hole] = a[first + child - 1];
        hole = child - 1;
    }
    mh_tk_push_heap(a, first, hole, top, value);
}
MH_TK_FN void mh_tk_make_heap(MhTkE *a, int first, int last) {
    const int len = last - first;
    if (len < 2) return;
    int parent = (len - 2) / 2;
    for (;;) {
        const MhTkE value = a[first + parent];
        mh_tk_adjust_heap(a, first, parent, len, value);
        if (parent == 0) return;
        --parent;
    }
}
MH_TK_FN void mh_tk_pop_heap(MhTkE *a, int first, int last, int result) {
    const MhTkE value = a[result];
    a[result] = a[first];
    mh_tk_adjust_heap(a, first, 0, last - first, value);
}
MH_TK_FN void mh_tk_heap_select(MhTkE *a, int first, int middle, int last) {
    mh_tk_make_heap(a, first, middle);
    for (int i = middle; i < last; ++i)
        if (mh_tk_gt(a[i], a[first])) mh_tk_pop_heap(a, first, middle, i);
}
MH_TK_FN void mh_tk_sort_heap(MhTkE *a, int first, int last) {
    while (last - first > 1) {
        --last;
        mh_tk_pop_heap(a, first, last, last);
    }
}

// ---- median of three to the front, Hoare partition around it
MH_TK_FN void mh_tk_median_to_first(MhTkE *a, int result, int x, int y, int z) {
    if (mh_tk_gt(a[x], a[y])) {
        if (mh_tk_gt(a[y], a[z])) mh_tk_swap(a, result, y);
        else if (mh_tk_gt(a[x], a[z])) mh_tk_swap(a, result, z);
        else mh_tk_swap(a, result, x);
    } else if (mh_tk_gt(a[x], a[z])) {
        mh_tk_swap(a, result, x);
    } else if (mh_tk_gt(a[y], a[z])) {
        mh_tk_swap(a, result, z);
    } else {
        mh_tk_swap(a, result, y);
    }
}
MH_TK_FN int mh_tk_partition_pivot(MhTkE *a, int first, int last) {
    const int mid = first + (last - first) / 2;
    mh_tk_median_to_first(a, first, first + 1, mid, last - 1);
    int lo = first + 1, hi = last;
    for (;;) {
        while (mh_tk_gt(a[lo], a[first])) ++lo;
        --hi;
        while (mh_tk_gt(a[first], a[hi])) --hi;
        if (!(lo < hi)) return lo;
        mh_tk_swap(a, lo, hi);
        ++lo;
    }
}

// ---- insertion sorts
MH_TK_FN void mh_tk_linear_insert(MhTkE *a, int last) {   // unguarded: something not "greater" than a[last] lies to the left
    const MhTkE val = a[last];
    int next = last - 1;
    while (mh_tk_gt(val, a[next])) {
        a[last] = a[next];
        last = next;
        --next;
    }
    a[last] = val;
}
MH_TK_FN void mh_tk_insertion_sort(MhTkE *a, int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        if (mh_tk_gt(a[i], a[first])) {
            const MhTkE val = a[i];
            for (int j = i; j > first; --j) a[j] = a[j - 1];
            a[first] = val;
        } else {
            mh_tk_linear_insert(a, i);
        }
    }
}

// std::nth_element(a+first, a+nth, a+last, gt)
MH_TK_FN void mh_tk_nth_element(MhTkE *a, int first, int nth, int last) {
    if (first == last || nth == last) return;
    int depth = mh_tk_lg(last - first) * 2;
    while (last - first > 3) {
        if (depth == 0) {
            mh_tk_heap_select(a, first, nth + 1, last);
            mh_tk_swap(a, first, nth);
            return;
        }
        --depth;
        const int cut = mh_tk_partition_pivot(a, first, last);
        if (cut <= nth) first = cut;
        else last = cut;
    }
    mh_tk_insertion_sort(a, first, last);
}

// std::sort(a+first, a+last, gt): introsort loop (explicit stack instead of recursion) + final insertion sort
MH_TK_FN void mh_tk_sort(MhTkE *a, int first, int last) {
    if (first == last) return;
    int sf[24], sl[24], sd[24], sp = 0;   // (depth limit 2*floor(log2(n)) <= 20 for n <= 1024: at most that many pending ranges)
    sf[0] = first, sl[0] = last, sd[0] = mh_tk_lg(last - first) * 2, sp = 1;
    while (sp > 0) {
        --sp;
        int f = sf[sp], l = sl[sp], d = sd[sp];
        while (l - f > 16) {
            if (d == 0) {   // partial_sort(f, l, l): heap select over the whole range, then sort the heap
                mh_tk_heap_select(a, f, l, l);
                mh_tk_sort_heap(a, f, l);
                break;
            }
            --d;
            const int cut = mh_tk_partition_pivot(a, f, l);
            // the library recurses into [cut, l) first and continues with [f, cut): the two ranges are disjoint, so the
            // order in which they are finished does not matter for the result
            sf[sp] = cut, sl[sp] = l, sd[sp] = d, ++sp;
            l = cut;
        }
    }
    if (last - first > 16) {
        mh_tk_insertion_sort(a, first, first + 16);
        for (int i = first + 16; i != last; ++i) mh_tk_linear_insert(a, i);
    } else {
        mh_tk_insertion_sort(a, first, last);
    }
}

// the k first entries of a[0..n) in torch.topk's order (largest=True, sorted=True); n >= k >= 1
MH_TK_FN void mh_tk_topk(MhTkE *a, int n, int k) {
    if ((long long)k * 64 <= n) {   // std::partial_sort(begin, begin+k, end)
        mh_tk_heap_select(a, 0, k, n);
        mh_tk_sort_heap(a, 0, k);
    } else {
        mh_tk_nth_element(a, 0, k - 1, n);
        mh_tk_sort(a, 0, k - 1);
    }
}
