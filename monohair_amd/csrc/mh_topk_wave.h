// mh_topk_wave.h -- the order in which torch.topk (CPU) returns tied values, for ONE WAVE per column.
//
// PMVO.Find_max_conf_from_visible_view (PMVO.py:339-343) calls torch.topk(k=20, dim=0) on a [V,N] tensor.  ATen's CPU
// kernel (aten/src/ATen/native/cpu/SortingKernel.cpp -> TopKImpl.h: topk_impl_loop) fills a vector of (value, index)
// pairs per column and, for k*64 > V, runs   std::nth_element(begin, begin+k-1, end, gt)   followed by
// std::sort(begin, begin+k-1, gt),   with gt(x, y) = (isnan(x) && !isnan(y)) || x > y  on the values only.  Neither
// algorithm is stable, so which of several equal values comes first is decided by libstdc++'s introselect / introsort
// (median-of-three pivot, Hoare partition, insertion sort below 4 resp. 16 elements, heap fall-back when the depth
// limit 2*floor(log2(n)) is used up).  Real captures reach PMVO through 8-bit confidence maps that saturate at 1.0, so
// such ties are the rule, and the choice of base views decides which candidates the search tries.  oracle/topk_oracle.cpp
// calls the real std:: functions, and the tests demand identical index arrays (tie-heavy random columns, adversarial
// columns that exhaust the depth limit, the reference's own rankings in tests/golden/).
//
// This header runs the same libstdc++ selection / sort steps in the same order, but the array lives in the registers of a wave
// (element e = 64 r + lane in register r of lane e & 63) and every step that the library spells as a scan or a shift is
// one or two wave operations:
//   * "while (gt(a[lo], pivot)) ++lo" / "while (gt(pivot, a[hi])) --hi"  -> a ballot of the predicate over the whole
//     array and a count-trailing / count-leading-zeros from lo / hi;
//   * __unguarded_linear_insert / move_backward (shift a run right by one and drop the value in front of it)
//     -> a ballot to find the position, one lane-shift of the run;
//   * element reads, writes and swaps -> v_readlane / one masked move.
// The values are carried as order-preserving integer keys (mh_tk_key: gt(x, y) <=> key(x) > key(y), all NaNs one key above
// +inf, -0 == +0), so every uniform comparison is a scalar-unit compare on the v_readlane results.
// All control flow is wave-uniform (the scalar unit branches), nothing diverges, and a column of 60 values costs
// ~1.5 k wave-instructions instead of ~60 k when 64 columns share a wave, each on its own branch.
// R = registers per lane = ceil(V / 64).
// (A literal one-lane-per-column restatement of the library code, mh_topk_order.h, was the first form and the cross-check
// until round 4; the oracle is the cross-check now.)
#pragma once

__device__ static inline int mh_tk_lg(int n) {   // floor(log2(n)), n >= 1
    int k = 0;
    while (n > 1) {
        n >>= 1;
        ++k;
    }
    return k;
}

__device__ __forceinline__ int mh_tk_key(float x) {
    if (x != x) return 0x7fffffff;
    const int b = __float_as_int(x + 0.0f);          // -0 -> +0
    return b < 0 ? (b ^ 0x7fffffff) : b;
}

struct MhTkK {   // one element in scalar registers
    int k, i;
};

template <int R>
struct MhTkWave {
    int k[R];   // keys
    int i[R];   // view indices
    int lane;

    __device__ __forceinline__ MhTkK get(int e) const {   // e wave-uniform
        int x = k[0], y = i[0];
#pragma unroll
        for (int r = 1; r < R; ++r)
            if ((e >> 6) == r) {
                x = k[r];
                y = i[r];
            }
        MhTkK o;
        o.k = __builtin_amdgcn_readlane(x, e & 63);
        o.i = __builtin_amdgcn_readlane(y, e & 63);
        return o;
    }
    __device__ __forceinline__ void set(int e, MhTkK x) {
        const bool me = lane == (e & 63);
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (R == 1 || (e >> 6) == r) {
                k[r] = me ? x.k : k[r];
                i[r] = me ? x.i : i[r];
            }
    }
    __device__ __forceinline__ void swap(int p, int q) {
        const MhTkK a = get(p), b = get(q);
        set(p, b);
        set(q, a);
    }
    // first e >= lo with !(a[e] > x)   (the library's unguarded forward scan: such an element exists)
    __device__ __forceinline__ int scan_up_not_gt_x(int lo, int x) const {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (r < (lo >> 6)) continue;
            unsigned long long m = __ballot(!(k[r] > x));
            if (r == (lo >> 6)) m &= ~0ull << (lo & 63);
            if (m) return r * 64 + __builtin_ctzll(m);
        }
        return 64 * R;   // not reached
    }
    // last e <= hi with !(x > a[e])    (unguarded backward scan)
    __device__ __forceinline__ int scan_down_x_not_gt(int hi, int x) const {
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const int r = R - 1 - rr;
            if (r > (hi >> 6)) continue;
            unsigned long long m = __ballot(!(x > k[r]));
            if (r == (hi >> 6)) m &= ~0ull >> (63 - (hi & 63));
            if (m) return r * 64 + 63 - __builtin_clzll(m);
        }
        return -1;   // not reached
    }
    // elements (b, t] <- their left neighbours (the element that was at t is overwritten; a[b] keeps its value)
    __device__ __forceinline__ void shift_right(int b, int t) {
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const int r = R - 1 - rr;           // high registers first: lane 0 takes the OLD lane 63 of the register below
            if (r * 64 > t || r * 64 + 63 <= b) continue;
            int ok = k[r], oi = i[r];           // what lane 0 receives
            if (r > 0) {
                ok = __builtin_amdgcn_readlane(k[r - 1], 63);
                oi = __builtin_amdgcn_readlane(i[r - 1], 63);
            }
            const int pk = __builtin_amdgcn_update_dpp(ok, k[r], 0x138, 0xf, 0xf, false);   // wave_shr:1
            const int pi = __builtin_amdgcn_update_dpp(oi, i[r], 0x138, 0xf, 0xf, false);
            const bool in = (unsigned)(r * 64 + lane - b - 1) < (unsigned)(t - b);
            k[r] = in ? pk : k[r];
            i[r] = in ? pi : i[r];
        }
    }

    // ---- the library's building blocks on top of these
    __device__ __forceinline__ void median_to_first(int result, int x, int y, int z) {
        const int a = get(x).k, b = get(y).k, c = get(z).k;
        int pick;
        if (a > b) pick = b > c ? y : (a > c ? z : x);
        else if (a > c) pick = x;
        else if (b > c) pick = z;
        else pick = y;
        swap(result, pick);
    }
    __device__ __forceinline__ int partition_pivot(int first, int last) {
        const int mid = first + (last - first) / 2;
        median_to_first(first, first + 1, mid, last - 1);
        const int pivot = get(first).k;
        int lo = first + 1, hi = last;
        for (;;) {
            lo = scan_up_not_gt_x(lo, pivot);             // while (gt(a[lo], pivot)) ++lo
            --hi;
            hi = scan_down_x_not_gt(hi, pivot);           // while (gt(pivot, a[hi])) --hi
            if (!(lo < hi)) return lo;
            swap(lo, hi);
            ++lo;
        }
    }
    __device__ __forceinline__ void linear_insert(int last) {    // __unguarded_linear_insert
        const MhTkK val = get(last);
        const int p = scan_down_x_not_gt(last - 1, val.k);       // first element to the left that val is not greater than
        if (p + 1 == last) return;
        shift_right(p + 1, last);
        set(p + 1, val);
    }
    __device__ __forceinline__ void insertion_sort(int first, int last) {
        if (first == last) return;
        for (int j = first + 1; j != last; ++j) {
            const MhTkK val = get(j);
            if (val.k > get(first).k) {
                shift_right(first, j);                           // move_backward(first, j, j + 1)
                set(first, val);
            } else {
                linear_insert(j);
            }
        }
    }
    // heap fall-back (depth limit used up): element by element, as the library does
    __device__ void push_heap(int first, int hole, int top, MhTkK value) {
        int parent = (hole - 1) / 2;
        while (hole > top && get(first + parent).k > value.k) {
            set(first + hole, get(first + parent));
            hole = parent;
            parent = (hole - 1) / 2;
        }
        set(first + hole, value);
    }
    __device__ void adjust_heap(int first, int hole, int len, MhTkK value) {
        const int top = hole;
        int child = hole;
        while (child < (len - 1) / 2) {
            child = 2 * (child + 1);
            if (get(first + child).k > get(first + child - 1).k) --child;
            set(first + hole, get(first + child));
            hole = child;
        }
        if ((len & 1) == 0 && child == (len - 2) / 2) {
            child = 2 * (child + 1);
            set(first + hole, get(first + child - 1));
            hole = child - 1;
        }
        push_heap(first, hole, top, value);
    }
    __device__ void heap_select(int first, int middle, int last) {
        const int len = middle - first;
        if (len >= 2) {
            int parent = (len - 2) / 2;
            for (;;) {
                adjust_heap(first, parent, len, get(first + parent));
                if (parent == 0) break;
                --parent;
            }
        }
        for (int j = middle; j < last; ++j)
            if (get(j).k > get(first).k) {
                const MhTkK value = get(j);
                set(j, get(first));
                adjust_heap(first, 0, middle - first, value);
            }
    }
    __device__ void sort_heap(int first, int last) {
        while (last - first > 1) {
            --last;
            const MhTkK value = get(last);
            set(last, get(first));
            adjust_heap(first, 0, last - first, value);
        }
    }
    __device__ void nth_element(int first, int nth, int last) {
        if (first == last || nth == last) return;
        int depth = mh_tk_lg(last - first) * 2;
        while (last - first > 3) {
            if (depth == 0) {
                heap_select(first, nth + 1, last);
                swap(first, nth);
                return;
            }
            --depth;
            const int cut = partition_pivot(first, last);
            if (cut <= nth) first = cut;
            else last = cut;
        }
        insertion_sort(first, last);
    }
    // std::sort; `stack` = 72 ints of wave-private LDS for the pending ranges of the introsort loop
    __device__ void sort(int first, int last, int *stack) {
        if (first == last) return;
        int sp = 0;
        stack[0] = first, stack[1] = last, stack[2] = mh_tk_lg(last - first) * 2, sp = 1;
        while (sp > 0) {
            --sp;
            int f = stack[3 * sp], l = stack[3 * sp + 1], d = stack[3 * sp + 2];
            while (l - f > 16) {
                if (d == 0) {
                    heap_select(f, l, l);
                    sort_heap(f, l);
                    break;
                }
                --d;
                const int cut = partition_pivot(f, l);
                stack[3 * sp] = cut, stack[3 * sp + 1] = l, stack[3 * sp + 2] = d, ++sp;
                l = cut;
            }
        }
        if (last - first > 16) {
            insertion_sort(first, first + 16);
            for (int j = first + 16; j != last; ++j) linear_insert(j);
        } else {
            insertion_sort(first, last);
        }
    }
    __device__ void topk(int n, int kk, int *stack) {
        if ((long long)kk * 64 <= n) {
            heap_select(0, kk, n);
            sort_heap(0, kk);
        } else {
            nth_element(0, kk - 1, n);
            sort(0, kk - 1, stack);
        }
    }
};
