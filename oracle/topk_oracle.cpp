/* oracle/topk_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see pmvo_oracle.c).
 *
 * PMVO.Find_max_conf_from_visible_view (/root/reference/PMVO.py:339-343): C' = vis < 1 ? Conf * max(vis, 0) : Conf,
 * torch.topk(C', 20, dim=0).  ATen's CPU kernel (aten/src/ATen/native/cpu/SortingKernel.cpp -> TopKImpl.h,
 * topk_impl_loop) copies every column into a vector of (value, index) pairs and calls
 *     std::partial_sort(begin, begin + k, end, gt)                                  if k * 64 <= n
 *     std::nth_element(begin, begin + k - 1, end, gt); std::sort(begin, begin + k - 1, gt)     otherwise
 * with gt(x, y) = (isnan(x) && !isnan(y)) || x > y on the values.  This file calls exactly those library functions, so
 * the order among equal values is the library's, as in the reference (probed: identical index arrays to torch.topk on
 * tie-heavy columns and on the reference's own rankings in tests/golden/).  The HIP kernel restates the library's
 * algorithms (monohair_amd/csrc/mh_topk_order.h) and is tested against this file. */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <utility>
#include <vector>

namespace {
using elem_t = std::pair<float, int64_t>;
bool gt(const elem_t &x, const elem_t &y) {
    return ((std::isnan(x.first) && !std::isnan(y.first)) || (x.first > y.first));
}
}   // namespace

/* one column of n values -> the k first (index, value) in torch.topk's order */
extern "C" void orc_topk_column(const float *v, int n, int k, int32_t *out_idx, float *out_val) {
    std::vector<elem_t> q((size_t)n);
    for (int j = 0; j < n; ++j) {
        q[j].first = v[j];
        q[j].second = j;
    }
    if ((long long)k * 64 <= n) {
        std::partial_sort(q.begin(), q.begin() + k, q.end(), gt);
    } else {
        std::nth_element(q.begin(), q.begin() + k - 1, q.end(), gt);
        std::sort(q.begin(), q.begin() + k - 1, gt);
    }
    for (int j = 0; j < k; ++j) {
        out_idx[j] = (int32_t)q[j].second;
        if (out_val) out_val[j] = q[j].first;
    }
}

extern "C" void orc_topk_views(const float *vis, const float *conf, int V, int N, int k, int32_t *out_idx, float *out_val) {
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n) {
        std::vector<float> col((size_t)V);
        for (int v = 0; v < V; ++v) {
            const float vb = vis[(size_t)v * N + n], c = conf[(size_t)v * N + n];
            col[v] = (vb < 1.0f) ? c * fmaxf(vb, 0.0f) : c;
        }
        std::vector<int32_t> idx((size_t)k);
        std::vector<float> val((size_t)k);
        orc_topk_column(col.data(), V, k, idx.data(), val.data());
        for (int r = 0; r < k; ++r) {
            out_idx[(size_t)r * N + n] = idx[r];
            out_val[(size_t)r * N + n] = val[r];
        }
    }
}
