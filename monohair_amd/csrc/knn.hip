// knn.hip -- exact k nearest neighbours on a uniform grid, gfx950 only.  Replaces the two scipy KDTree queries of
// refine (PMVO.py:605,612,660,671: `points_tree.query(sub_points, 100)`) that were the last host stage of the
// smoothing loop.  Distances are evaluated in fp64 from the fp32 coordinates (scipy converts to double too) and
// neighbours come back sorted by (distance, index), i.e. scipy's order whenever distances are distinct.
//
// Points are pre-sorted by cell (x fastest), so a row of cells is ONE contiguous range.  One wave per query: for
// ring = 1, 2, ... gather every point of the (2 ring + 1)^3 cell block into LDS (ballot-prefix append), bitonic-sort
// the candidates by (d2, index) and stop as soon as the k-th distance is <= ring * h -- nothing outside the block can
// be closer than that.  Queries that overflow the candidate buffer (status 2) or reach the ring limit (status 1)
// are flagged; the host wrapper re-runs just those on a finer / coarser grid (still on the GPU).
#include "mh_device.h"

#define MH_KNN_CAP 2048
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
                                                     const float *__restrict__ queries, int Q, int k,
                                                     int32_t *__restrict__ out_idx, int32_t *__restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double *s_d = reinterpret_cast<double *>(smem) + (size_t)wave * MH_KNN_CAP;
    int *s_i = reinterpret_cast<int *>(smem + 2 * MH_KNN_CAP * sizeof(double)) + (size_t)wave * MH_KNN_CAP;
    const int qi = blockIdx.x * 2 + wave;
    if (qi >= Q) return;
    const float qx = queries[3 * qi], qy = queries[3 * qi + 1], qz = queries[3 * qi + 2];
    const int cx = min(max((int)floorf((qx - g.ox) / g.h), 0), g.dx - 1);
    const int cy = min(max((int)floorf((qy - g.oy) / g.h), 0), g.dy - 1);
    const int cz = min(max((int)floorf((qz - g.oz) / g.h), 0), g.dz - 1);
    const double dqx = (double)qx, dqy = (double)qy, dqz = (double)qz;
    int st = 1;   // 1: ring limit reached (cells too small for this query), 2: candidate buffer overflow (too large)
    for (int ring = 1; ring <= MH_KNN_MAXRING; ++ring) {
        int cnt = 0;
        bool overflow = false;
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
                    const unsigned long long m = __ballot(ok);
                    const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
                    if (ok && pos < MH_KNN_CAP) {
                        s_d[pos] = d2;
                        s_i[pos] = id;
                    }
                    cnt += __popcll(m);
                }
            }
        if (cnt > MH_KNN_CAP) overflow = true;
        const bool whole_grid = (cx - ring <= 0) && (cy - ring <= 0) && (cz - ring <= 0) && (cx + ring >= g.dx - 1) &&
                                (cy + ring >= g.dy - 1) && (cz + ring >= g.dz - 1);
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
        const int kk = min(k, cnt);
        const double reach = (double)ring * (double)g.h;
        if (whole_grid || (cnt >= k && s_d[k - 1] <= reach * reach)) {
            for (int j = lane; j < k; j += MH_WAVE) out_idx[(size_t)qi * k + j] = (j < kk) ? s_i[j] : -1;
            st = 0;
            break;
        }
    }
    if (lane == 0) status[qi] = st;
}

extern "C" int mh_launch_knn(float ox, float oy, float oz, float h, int dx, int dy, int dz, const float *pts,
                             const int32_t *order, const int32_t *cell_start, const float *queries, int Q, int k,
                             int32_t *out_idx, int32_t *status, hipStream_t st) {
    if (k < 1 || k > MH_KNN_CAP) return -1;
    MhGrid g{ox, oy, oz, h, dx, dy, dz};
    const size_t lds = 2 * (size_t)MH_KNN_CAP * (sizeof(double) + sizeof(int));
    hipLaunchKernelGGL(mh_knn_kernel, dim3((Q + 1) / 2), dim3(128), lds, st, g, pts, order, cell_start, queries, Q, k,
                       out_idx, status);
    return (int)hipGetLastError();
}
