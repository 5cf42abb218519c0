// knn.hip -- exact k nearest neighbours on a uniform grid, gfx950 only.  Replaces the two scipy KDTree queries of
// refine (PMVO.py:605,612,660,671: `points_tree.query(sub_points, 100)`) that were the last host stage of the
// smoothing loop.  Distances are evaluated in fp64 from the fp32 coordinates (scipy converts to double too) and
// neighbours come back sorted by (distance, index), i.e. scipy's order whenever distances are distinct.
//
// Points are pre-sorted by cell (x fastest), so a row of cells is ONE contiguous range.  One wave per query: for
// ring = ring0, ring0+1, ... look at every point of the (2 ring + 1)^3 cell block and keep, in LDS (ballot-prefix
// append), the ones within reach = ring * h of the query -- every point that close lies inside the block, so as soon
// as k of them are there the k nearest neighbours are among them: bitonic-sort just those by (d2, index) and stop.
// (Sorting only the points within reach instead of the whole block makes the sort 2-4x smaller.)  Queries that
// overflow the candidate buffer (status 2) or reach the ring limit (status 1) are flagged; the host wrapper re-runs
// just those on a finer / coarser grid (still on the GPU).  qperm (optional): the order in which the waves take the
// queries -- for a self-query the cell-sorted order, so that neighbouring waves read the same cells.  valid (optional,
// one byte per original point index): only those points are neighbours -- the k-NN of a SUBSET on the grid that was built
// for all points (refine queries the points kept by the loss threshold right after querying all of them).
#include "mh_device.h"

#define MH_KNN_CAP 2048        // candidate buffer of the second pass (and the largest k)
#define MH_KNN_CAP_FIRST 512   // ... of the first pass: 12 KB of LDS per two-wave block instead of 48 -> 26 waves per CU
                               // instead of 6 (round 6: a block of 27 cells holds ~11 k points, ~1.7 k of them within reach)
#define MH_KNN_MAXRING 6

struct MhGrid {
    float ox, oy, oz, h;
    int dx, dy, dz;
};

__device__ __forceinline__ bool mh_knn_less(double da, int ia, double db, int ib) {
    return (da < db) || (da == db && ia < ib);
}

__global__ __launch_bounds__(128) void mh_knn_kernel(MhGrid g, const float *__restrict__ pts,
                                                     const int32_t *__restrict__ order,
                                                     const int32_t *__restrict__ cell_start,
                                                     const void *__restrict__ queries, int q64, int Q, int k,
                                                     int ring0, const int32_t *__restrict__ qperm,
                                                     const uint8_t *__restrict__ valid,
                                                     int32_t *__restrict__ out_idx, int32_t *__restrict__ status,
                                                     int cap, int only_status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double *s_d = reinterpret_cast<double *>(smem) + (size_t)wave * cap;
    int *s_i = reinterpret_cast<int *>(smem + 2 * (size_t)cap * sizeof(double)) + (size_t)wave * cap;
    const int qw = blockIdx.x * 2 + wave;
    if (qw >= Q) return;
    const int qi = qperm ? qperm[qw] : qw;
    // second pass: only the queries whose candidates did not fit the first pass's buffer
    if (only_status && status[qi] != only_status) return;
    // queries are float32 or float64 (the reference hands float64 shell points to KDTree.query, PMVO.py:671); distances
    // use the exact coordinates, the cell of the query only has to be near enough (see the reach margin below)
    double dqx, dqy, dqz;
    if (q64) {
        const double *__restrict__ q = reinterpret_cast<const double *>(queries);
        dqx = q[3 * qi], dqy = q[3 * qi + 1], dqz = q[3 * qi + 2];
    } else {
        const float *__restrict__ q = reinterpret_cast<const float *>(queries);
        dqx = (double)q[3 * qi], dqy = (double)q[3 * qi + 1], dqz = (double)q[3 * qi + 2];
    }
    const float qx = (float)dqx, qy = (float)dqy, qz = (float)dqz;
    const int cx = min(max((int)floorf((qx - g.ox) / g.h), 0), g.dx - 1);
    const int cy = min(max((int)floorf((qy - g.oy) / g.h), 0), g.dy - 1);
    const int cz = min(max((int)floorf((qz - g.oz) / g.h), 0), g.dz - 1);
    int st = 1;   // 1: ring limit reached (cells too small for this query), 2: candidate buffer overflow (too large)
    for (int ring = max(ring0, 1); ring <= MH_KNN_MAXRING; ++ring) {
        int cnt = 0;
        bool overflow = false;
        const bool whole_grid = (cx - ring <= 0) && (cy - ring <= 0) && (cz - ring <= 0) && (cx + ring >= g.dx - 1) &&
                                (cy + ring >= g.dy - 1) && (cz + ring >= g.dz - 1);
        // the block reaches at least `ring` cells beyond the query's cell; 1e-3 of a cell absorbs the float rounding of
        // the cell indices of the query and of the points (coordinates / h < 2^10, i.e. errors below 1e-4 cells)
        const double reach = ((double)ring - 1.0e-3) * (double)g.h;
        const double reach2 = whole_grid ? __builtin_inf() : reach * reach;
        const int x0 = max(cx - ring, 0), x1 = min(cx + ring, g.dx - 1);
        for (int z = max(cz - ring, 0); z <= min(cz + ring, g.dz - 1); ++z)
            for (int y = max(cy - ring, 0); y <= min(cy + ring, g.dy - 1); ++y) {
                const int row = (z * g.dy + y) * g.dx;
                const int b = cell_start[row + x0], e = cell_start[row + x1 + 1];
                for (int p0 = b; p0 < e; p0 += MH_WAVE) {
                    const int p = p0 + lane;
                    const bool ok = p < e;
                    double d2 = 0.0;
                    int id = 0;
                    if (ok) {
                        const double ax = (double)pts[3 * p] - dqx, ay = (double)pts[3 * p + 1] - dqy,
                                     az = (double)pts[3 * p + 2] - dqz;
                        d2 = (ax * ax + ay * ay) + az * az;
                        id = order[p];
                    }
                    // valid (optional, per ORIGINAL index): search a subset of the data points on the grid of all of them
                    const bool keep = ok && d2 <= reach2 && (!valid || valid[id]);
                    const unsigned long long m = __ballot(keep);
                    const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
                    if (keep && pos < cap) {
                        s_d[pos] = d2;
                        s_i[pos] = id;
                    }
                    cnt += __popcll(m);
                }
            }
        if (cnt > cap) overflow = true;
        if (overflow) {
            st = 2;
            break;
        }
        if (cnt < k && !whole_grid) continue;
        // bitonic sort of the next power of two >= cnt, padded with +inf
        int n2 = 64;
        while (n2 < cnt) n2 <<= 1;
        for (int i = cnt + lane; i < n2; i += MH_WAVE) {
            s_d[i] = __builtin_inf();
            s_i[i] = 0x7fffffff;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int size = 2; size <= n2; size <<= 1)
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int t = lane; t < (n2 >> 1); t += MH_WAVE) {
                    const int lo = 2 * t - (t & (stride - 1));   // index with the `stride` bit clear
                    const int hi = lo + stride;
                    const bool up = (lo & size) == 0;
                    const double dl = s_d[lo], dh = s_d[hi];
                    const int il = s_i[lo], ih = s_i[hi];
                    const bool swap = up ? mh_knn_less(dh, ih, dl, il) : mh_knn_less(dl, il, dh, ih);
                    if (swap) {
                        s_d[lo] = dh;
                        s_d[hi] = dl;
                        s_i[lo] = ih;
                        s_i[hi] = il;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        // every kept candidate is within reach, so with k of them (or the whole grid) the answer is final
        const int kk = min(k, cnt);
        for (int j = lane; j < k; j += MH_WAVE) out_idx[(size_t)qi * k + j] = (j < kk) ? s_i[j] : -1;
        st = 0;
        break;
    }
    if (lane == 0) status[qi] = st;
}

extern "C" int mh_launch_knn(float ox, float oy, float oz, float h, int dx, int dy, int dz, const float *pts,
                             const int32_t *order, const int32_t *cell_start, const void *queries, int q64, int Q,
                             int k, int ring0, const int32_t *qperm, const uint8_t *valid, int32_t *out_idx,
                             int32_t *status, hipStream_t st) {
    if (k < 1 || k > MH_KNN_CAP) return -1;
    MhGrid g{ox, oy, oz, h, dx, dy, dz};
    // two passes on the same grid: a small candidate buffer (high occupancy) for all queries, then the full buffer for
    // the queries that overflowed it (status 2 after the first pass; the other waves of the second launch exit at once)
    const int cap1 = k <= MH_KNN_CAP_FIRST / 2 ? MH_KNN_CAP_FIRST : MH_KNN_CAP;
    const size_t per = sizeof(double) + sizeof(int);
    hipLaunchKernelGGL(mh_knn_kernel, dim3((Q + 1) / 2), dim3(128), 2 * (size_t)cap1 * per, st, g, pts, order, cell_start,
                       queries, q64, Q, k, ring0, qperm, valid, out_idx, status, cap1, 0);
    if (cap1 < MH_KNN_CAP)
        hipLaunchKernelGGL(mh_knn_kernel, dim3((Q + 1) / 2), dim3(128), 2 * (size_t)MH_KNN_CAP * per, st, g, pts, order,
                           cell_start, queries, q64, Q, k, ring0, qperm, valid, out_idx, status, MH_KNN_CAP, 2);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Distance to the nearest of M reference points, exhaustive, in float64: the `scalp_tree.query(points, k=1)` of
// PMVO.filter_head_points (PMVO.py:100-101; scipy's KDTree works in float64 and returns sqrt of the summed squares).
// One lane per query point, the reference points staged through LDS in tiles; d2 = (dx*dx + dy*dy) + dz*dz, the
// minimum of the squares, then one IEEE sqrt.  (A few thousand scalp vertices: 10^8..10^9 fp64 operations per call.)
// ---------------------------------------------------------------------------------------------------------------
#define MH_ND_TILE 1024
__global__ __launch_bounds__(256) void mh_nearest_dist_kernel(const float *__restrict__ pts, int N,
                                                              const double *__restrict__ ref, int M,
                                                              double *__restrict__ out, double max_dist,
                                                              double z_limit, uint8_t *__restrict__ mask) {
    __shared__ double s_r[MH_ND_TILE * 3];
    const int n = blockIdx.x * 256 + threadIdx.x;
    double px = 0.0, py = 0.0, pz = 0.0;
    if (n < N) {
        px = (double)pts[3 * n];
        py = (double)pts[3 * n + 1];
        pz = (double)pts[3 * n + 2];
    }
    double best = __builtin_inf();
    for (int m0 = 0; m0 < M; m0 += MH_ND_TILE) {
        const int nm = min(MH_ND_TILE, M - m0);
        __syncthreads();
        for (int i = threadIdx.x; i < nm * 3; i += 256) s_r[i] = ref[(size_t)m0 * 3 + i];
        __syncthreads();
        for (int j = 0; j < nm; ++j) {
            const double dx = s_r[3 * j] - px, dy = s_r[3 * j + 1] - py, dz = s_r[3 * j + 2] - pz;
            const double d2 = (dx * dx + dy * dy) + dz * dz;
            best = d2 < best ? d2 : best;
        }
    }
    if (n < N) {
        const double d = __builtin_sqrt(best);
        if (out) out[n] = d;
        // filter_head_points' use of it (PMVO.py:102-106): nearer than max_dist AND below z_limit, both in float64
        if (mask) mask[n] = (d < max_dist && pz < z_limit) ? 1 : 0;
    }
}

extern "C" int mh_launch_nearest_dist(const float *pts, int N, const double *ref, int M, double *out, double max_dist,
                                      double z_limit, uint8_t *mask, hipStream_t st) {
    hipLaunchKernelGGL(mh_nearest_dist_kernel, dim3((N + 255) / 256), dim3(256), 0, st, pts, N, ref, M, out, max_dist,
                       z_limit, mask);
    return (int)hipGetLastError();
}

// forces this translation unit's code object onto the device (HIP loads a fat binary on the first use of one of its kernels:
// 2-20 ms each, which a one-shot pass would pay in the middle of its stages); called from mh_ctx_create
extern "C" int mh_preload_knn() {
    hipFuncAttributes a;
    return (int)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&mh_knn_kernel));
}
