// pmvo_filter.hip -- the per-view visibility / mask / confidence votes (K13), gfx950 only:
//   PMVO.filter_points            (PMVO.py:402-459)  -> surface_index, filter_index
//   PMVO.compute_unvisible_points (PMVO.py:461-480)  -> unvisible_index
//   PMVO.filter_head_points       (PMVO.py:110-137)  -> head_filter (the mask vote only; the two scipy
//                                                      KDTree queries of :98-107 stay on the host)
// One wave per point, lane = view.  Each lane writes its per-view terms to LDS and lane 0 adds them in
// ATen's cascade order (mask values in (0, 0.2] stay fractional, PMVO.py:427, so order can matter) -- in its row_sum
// order for the trailing points of a batch of the reference (MhBatch).
#include "mh_device.h"

#define MH_FILTER_VMAX 512
#define MH_NTERM 8

// rows of a launch whose sums over the views are NOT the cascade: the trailing (len mod block) rows of every batch of the
// reference and batches of one point.  `count` virtual indices i -> global row first_tail + (i / per_batch) * stride + i % per_batch.
struct MhTailRows {
    long long first_tail, stride;
    int per_batch, count;
};

// ---------------------------------------------------------------------------------------------------------------
// mh_filter_rows_kernel (round 6): the same votes with lane = POINT.  mh_filter_kernel's wave is one point and its lanes
// are 60 views -- 60 isolated 20-byte gathers in 60 different images per wave, each moving whole lines: 10.2 GB fetched
// per pass for 0.68 GB of algorithmic bytes (profiles/r06_fullpass_summary.txt).  Here a wave is 64 CONSECUTIVE points
// (neighbours in space: the candidates come in volume raster order) walking the views together: a wave's centre gathers
// and patch rows of one view are neighbouring pixels of ONE image, every lane adds its own per-view terms in ATen's
// cascade order in registers (no LDS, no serial lane-0 sum).  Per (point, view) the arithmetic is mh_filter_kernel's,
// statement for statement.  Rows whose sums take another order (MhTailRows) are skipped here and done by mh_filter_kernel.
// ---------------------------------------------------------------------------------------------------------------
template <int PATCH>
__global__ __launch_bounds__(256) void mh_filter_rows_kernel(MhViews vw, const float *__restrict__ pts, int N, float thr,
                                                             float vis_thr, uint8_t *__restrict__ surface_index,
                                                             uint8_t *__restrict__ filter_index,
                                                             uint8_t *__restrict__ unvisible_index,
                                                             uint8_t *__restrict__ head_filter, MhBatch bt,
                                                             const int32_t *__restrict__ order) {
    constexpr int HP = PATCH / 2;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    // order (optional): the rows in an order that puts spatial neighbours into one wave (sorted by grid cell, mh_grid_build):
    // the candidates' own order sweeps every image once per slab of the volume -- 10 GB of line fetches per pass for isolated
    // 20-byte gathers; a wave of one small cube touches a few lines per view (2.15 -> 0.97 ms for 465 k candidates)
    const int n = order ? order[i] : i;
    const bool one_point = mh_batch_single(bt, n);
    if (bt.block > 0 && (one_point || mh_tail_row(bt, n))) return;   // another summation order: mh_filter_kernel's rows
    const bool single = bt.single_ok && one_point;
    const int V = vw.V, H = vw.H, W = vw.W;
    const float X0 = pts[3 * n], X1 = pts[3 * n + 1], X2 = pts[3 * n + 2];
    const bool want_patch = surface_index || filter_index;
    MhCascV acc[MH_NTERM];
#pragma unroll
    for (int t = 0; t < MH_NTERM; ++t) acc[t] = MhCascV{0.f, 0.f, 0.f};
    for (int v = 0; v < V; ++v) {
        if (v > 0 && (v & 15) == 0) {
#pragma unroll
            for (int t = 0; t < MH_NTERM; ++t) mh_cascv_flush(acc[t], v);
        }
        const float *cam = vw.cams + v * MH_CAM_STRIDE;
        float u, w, z, rowf, colf;
        mh_cam_project_b(cam, X0, X1, X2, u, w, z, single);
        mh_ndc_to_pixel(u, w, (float)H, (float)W, rowf, colf);
        float cr = __builtin_rintf(colf), rr = __builtin_rintf(rowf);
        const bool oob = !(cr <= (float)(W - 1)) || (cr < 0.0f) || !(rr <= (float)(H - 1)) || (rr < 0.0f);
        cr = fminf(fmaxf(cr, 0.0f), (float)(W - 1));
        rr = fminf(fmaxf(rr, 0.0f), (float)(H - 1));
        const int r = (int)rr, c = (int)cr;
        const float4 *__restrict__ rec = vw.rec + (size_t)v * H * W;
        const float4 q = rec[(size_t)r * W + c];
        float m = vw.mask[(size_t)v * H * W + (size_t)r * W + c];
        const float gap = (-z / 2.0f) * 255.0f - q.w;
        float cmax = 0.0f;
        if (want_patch && !(oob || gap > 0.1f)) {
            // all PATCH*PATCH loads are issued before the first use (fully unrolled): the lane is latency bound
            float cv[PATCH * PATCH];
#pragma unroll
            for (int i = -HP; i <= HP; ++i)
#pragma unroll
                for (int j = -HP; j <= HP; ++j) {
                    const int rr2 = min(max(r + i, 0), H - 1), cc2 = min(max(c + j, 0), W - 1);
                    cv[(i + HP) * PATCH + j + HP] = rec[(size_t)rr2 * W + cc2].z;
                }
            cmax = q.z;
#pragma unroll
            for (int t = 0; t < PATCH * PATCH; ++t) cmax = (cv[t] > cmax) ? cv[t] : cmax;
        }
        const float unv = (oob || gap > 0.1f) ? 1.0f : 0.0f;
        const float unv1 = (oob || gap > vis_thr) ? 1.0f : 0.0f;
        const float unv9 = (oob || gap > 0.9f) ? 1.0f : 0.0f;
        const float unvh = (gap >= vis_thr) ? 1.0f : 0.0f;   // filter_head_points: '>=' and no oob override
        const float lowc = (cmax < thr) ? 1.0f : 0.0f;
        m = (m > 0.2f) ? 1.0f : m;
        const float visb = 1.0f - unv, visb1 = 1.0f - unv1, visbh = 1.0f - unvh;
        acc[0].a0 = acc[0].a0 + visb * lowc;
        acc[1].a0 = acc[1].a0 + visb;
        acc[2].a0 = acc[2].a0 + visb * m;
        acc[3].a0 = acc[3].a0 + visb1;
        acc[4].a0 = acc[4].a0 + visb1 * m;
        acc[5].a0 = acc[5].a0 + (1.0f - unv9);
        acc[6].a0 = acc[6].a0 + visbh;
        acc[7].a0 = acc[7].a0 + visbh * m;
    }
    const float s_idx = mh_cascv_done(acc[0]), s_vis = mh_cascv_done(acc[1]), s_vm = mh_cascv_done(acc[2]);
    const float s_vis1 = mh_cascv_done(acc[3]), s_vm1 = mh_cascv_done(acc[4]), s_v9 = mh_cascv_done(acc[5]);
    const float s_vh = mh_cascv_done(acc[6]), s_ih = mh_cascv_done(acc[7]);
    const bool low_conf = s_idx > 4.0f;
    const bool hair = (s_vis - s_vm) < (s_vis * 1.0f / 2.0f);
    const bool hair1 = (s_vis1 - s_vm1) < (s_vis1 * 1.0f / 2.0f);
    const bool surf0 = s_vis > 1.0f;
    const bool filt0 = (s_vis1 > 1.0f) && !surf0;
    if (surface_index) surface_index[n] = (surf0 && !low_conf && hair) ? 1 : 0;
    if (filter_index) filter_index[n] = (filt0 && !low_conf && hair1) ? 1 : 0;
    if (unvisible_index) unvisible_index[n] = (s_v9 > 2.0f) ? 0 : 1;
    if (head_filter) head_filter[n] = ((s_vh - s_ih) < (s_vh * 1.0f / 2.0f)) ? 0 : 1;
}

template <int PATCH>
__global__ __launch_bounds__(256) void mh_filter_kernel(MhViews vw, const float *__restrict__ pts, int N, float thr,
                                                        float vis_thr, uint8_t *__restrict__ surface_index,
                                                        uint8_t *__restrict__ filter_index,
                                                        uint8_t *__restrict__ unvisible_index,
                                                        uint8_t *__restrict__ head_filter, MhBatch bt, MhTailRows tr) {
    constexpr int HP = PATCH / 2;
    extern __shared__ float s_tbuf[];   // [4 waves][MH_NTERM][V]
#define s_t(w, t, v) s_tbuf[((w) * MH_NTERM + (t)) * V + (v)]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int n = blockIdx.x * 4 + wave;
    if (tr.count > 0) {   // only the rows whose sums take another order (round 6: the other rows are mh_filter_rows_kernel's)
        if (n >= tr.count) return;
        const long long row = tr.first_tail + (long long)(n / tr.per_batch) * tr.stride + n % tr.per_batch;
        if (row < bt.row0 || row >= bt.row0 + N) return;
        n = (int)(row - bt.row0);
    }
    if (n >= N) return;
    const int V = vw.V, H = vw.H, W = vw.W;
    const float X0 = pts[3 * n], X1 = pts[3 * n + 1], X2 = pts[3 * n + 2];
    // a batch of ONE point: its [V,1] sums over the views are ATen's inner sums whenever the outer-sum rule is on (sum_block > 0)
    // and -- with the batch rule of the products (reproject_rule 0) -- its projections are single-column products
    const bool one_point = mh_batch_single(bt, n);
    const bool single = bt.single_ok && one_point;
    for (int v = lane; v < V; v += MH_WAVE) {
        const float *cam = vw.cams + v * MH_CAM_STRIDE;
        float u, w, z, rowf, colf;
        mh_cam_project_b(cam, X0, X1, X2, u, w, z, single);
        mh_ndc_to_pixel(u, w, (float)H, (float)W, rowf, colf);
        float cr = __builtin_rintf(colf), rr = __builtin_rintf(rowf);
        const bool oob = !(cr <= (float)(W - 1)) || (cr < 0.0f) || !(rr <= (float)(H - 1)) || (rr < 0.0f);
        cr = fminf(fmaxf(cr, 0.0f), (float)(W - 1));
        rr = fminf(fmaxf(rr, 0.0f), (float)(H - 1));
        const int r = (int)rr, c = (int)cr;
        const float4 *__restrict__ rec = vw.rec + (size_t)v * H * W;
        const float4 q = rec[(size_t)r * W + c];
        float m = vw.mask[(size_t)v * H * W + (size_t)r * W + c];
        const float gap = (-z / 2.0f) * 255.0f - q.w;
        // raw (unclamped) patch confidences, zeroed when the point projects outside (PMVO.py:415-420)
        // (only term 0 = visible * low_confidence uses it, so views that do not see the point skip the patch)
        float cmax = 0.0f;
        if ((surface_index || filter_index) && !(oob || gap > 0.1f)) {
            // all PATCH*PATCH loads are issued before the first use (fully unrolled): the lane is latency bound
            float cv[PATCH * PATCH];
#pragma unroll
            for (int i = -HP; i <= HP; ++i)
#pragma unroll
                for (int j = -HP; j <= HP; ++j) {
                    const int rr2 = min(max(r + i, 0), H - 1), cc2 = min(max(c + j, 0), W - 1);
                    cv[(i + HP) * PATCH + j + HP] = rec[(size_t)rr2 * W + cc2].z;
                }
            cmax = q.z;
#pragma unroll
            for (int t = 0; t < PATCH * PATCH; ++t) cmax = (cv[t] > cmax) ? cv[t] : cmax;
            if (oob) cmax = 0.0f;
        }
        const float unv = (oob || gap > 0.1f) ? 1.0f : 0.0f;
        const float unv1 = (oob || gap > vis_thr) ? 1.0f : 0.0f;
        const float unv9 = (oob || gap > 0.9f) ? 1.0f : 0.0f;
        const float unvh = (gap >= vis_thr) ? 1.0f : 0.0f;   // filter_head_points: '>=' and no oob override
        const float lowc = (cmax < thr) ? 1.0f : 0.0f;
        m = (m > 0.2f) ? 1.0f : m;
        const float visb = 1.0f - unv, visb1 = 1.0f - unv1, visbh = 1.0f - unvh;
        s_t(wave, 0, v) = visb * lowc;    // visible with low confidence
        s_t(wave, 1, v) = visb;           // visibles
        s_t(wave, 2, v) = visb * m;       // visibles * masks
        s_t(wave, 3, v) = visb1;          // visibles1
        s_t(wave, 4, v) = visb1 * m;      // visibles1 * masks
        s_t(wave, 5, v) = 1.0f - unv9;    // compute_unvisible_points
        s_t(wave, 6, v) = visbh;          // filter_head_points visibles
        s_t(wave, 7, v) = visbh * m;      // filter_head_points indexs
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < MH_NTERM) {
        MhCascV a = {0.f, 0.f, 0.f};
        for (int v = 0; v < V; ++v) {
            if (v > 0 && (v & 15) == 0) mh_cascv_flush(a, v);
            a.a0 = a.a0 + s_t(wave, lane, v);
        }
        float sum = mh_cascv_done(a);
        // the trailing (length mod 32) points of a batch of the reference: ATen's row_sum order (mh_device.h: MhBatch)
        if (one_point && bt.block > 0) sum = mh_inner_sum_views(V, [&](int v) { return s_t(wave, lane, v); });
        else if (mh_tail_row(bt, n)) sum = mh_row_sum_views(V, [&](int v) { return s_t(wave, lane, v); });
        s_t(wave, lane, 0) = sum;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        const float s_idx = s_t(wave, 0, 0), s_vis = s_t(wave, 1, 0), s_vm = s_t(wave, 2, 0);
        const float s_vis1 = s_t(wave, 3, 0), s_vm1 = s_t(wave, 4, 0), s_v9 = s_t(wave, 5, 0);
        const float s_vh = s_t(wave, 6, 0), s_ih = s_t(wave, 7, 0);
        const bool low_conf = s_idx > 4.0f;
        const bool hair = (s_vis - s_vm) < (s_vis * 1.0f / 2.0f);
        const bool hair1 = (s_vis1 - s_vm1) < (s_vis1 * 1.0f / 2.0f);
        const bool surf0 = s_vis > 1.0f;
        const bool filt0 = (s_vis1 > 1.0f) && !surf0;
        if (surface_index) surface_index[n] = (surf0 && !low_conf && hair) ? 1 : 0;
        if (filter_index) filter_index[n] = (filt0 && !low_conf && hair1) ? 1 : 0;
        if (unvisible_index) unvisible_index[n] = (s_v9 > 2.0f) ? 0 : 1;
        if (head_filter) head_filter[n] = ((s_vh - s_ih) < (s_vh * 1.0f / 2.0f)) ? 0 : 1;
    }
}

extern "C" int mh_launch_filter_points(MhViews vw, const float *pts, int N, int patch, float thr, float vis_thr,
                                       uint8_t *surface_index, uint8_t *filter_index, uint8_t *unvisible_index,
                                       uint8_t *head_filter, int batch, long long row0, long long total, int sum_block,
                                       int rows_kernel, const int32_t *order, hipStream_t st) {
    if (vw.V > MH_FILTER_VMAX) return -1;
    const MhBatch bt = {row0, total, batch, sum_block, vw.batch_rule};
    const size_t lds = (size_t)4 * MH_NTERM * vw.V * sizeof(float);
    // which rows take the wave-per-point kernel: all of them (rows_kernel == 0, or launches too small to fill lanes with
    // neighbours), or only the rows whose sums are not the cascade -- the trailing (len mod block) rows of every batch the
    // launch touches and batches of one point (MhBatch) -- as up to two groups of virtual indices: the full batches, whose
    // tails have one length, and the last batch of `total`
    MhTailRows groups[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    int ngroups = 0;
    const bool split = rows_kernel && N >= 4096;
    if (split && sum_block > 0) {
        const long long tot = batch > 0 ? total : (long long)N, L = (batch > 0 && batch < tot) ? batch : tot;
        const long long base = batch > 0 ? 0 : row0;                        // batch == 0: these N rows are one batch
        const long long nfull = tot / L, rem = tot - nfull * L;             // full batches, then one of `rem` rows
        const int tfull = (int)(L == 1 ? 1 : L % sum_block);
        if (nfull > 0 && tfull > 0) groups[ngroups++] = {base + L - tfull, L, tfull, (int)(nfull * tfull)};
        const int trem = (int)(rem == 1 ? 1 : rem % sum_block);
        if (rem > 0 && trem > 0) groups[ngroups++] = {base + nfull * L + rem - trem, 1, trem, trem};
        for (int g = 0; g < ngroups; ++g)                                   // (per_batch <= 31, count <= 31 * batches)
            if ((long long)groups[g].count * 1 > (1 << 24)) return -1;
    }
#define MH_F_CASE(PS)                                                                                                   \
    case PS:                                                                                                            \
        if (!split) {                                                                                                   \
            hipLaunchKernelGGL(mh_filter_kernel<PS>, dim3((N + 3) / 4), dim3(256), lds, st, vw, pts, N, thr, vis_thr,      \
                               surface_index, filter_index, unvisible_index, head_filter, bt, MhTailRows{0, 0, 0, 0});  \
        } else {                                                                                                        \
            hipLaunchKernelGGL(mh_filter_rows_kernel<PS>, dim3((N + 255) / 256), dim3(256), 0, st, vw, pts, N, thr,      \
                               vis_thr, surface_index, filter_index, unvisible_index, head_filter, bt, order);         \
            for (int g = 0; g < ngroups; ++g)                                                                           \
                hipLaunchKernelGGL(mh_filter_kernel<PS>, dim3((groups[g].count + 3) / 4), dim3(256), lds, st, vw, pts, N, \
                                   thr, vis_thr, surface_index, filter_index, unvisible_index, head_filter, bt,         \
                                   groups[g]);                                                                          \
        }                                                                                                               \
        break;
    switch (patch) {
        MH_F_CASE(1)
        MH_F_CASE(3)
        MH_F_CASE(5)
        MH_F_CASE(7)
        MH_F_CASE(9)
        MH_F_CASE(11)
        default:
            return -1;
    }
#undef MH_F_CASE
    return (int)hipGetLastError();
}

// forces this translation unit's code object onto the device (HIP loads a fat binary on the first use of one of its kernels:
// 2-20 ms each, which a one-shot pass would pay in the middle of its stages); called from mh_ctx_create
extern "C" int mh_preload_pmvo_filter() {
    hipFuncAttributes a;
    return (int)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&mh_filter_kernel<7>));
}
