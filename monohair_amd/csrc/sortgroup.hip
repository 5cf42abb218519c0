// sortgroup.hip -- the two "group by cell" steps of refine, gfx950 only:
//   * the uniform grid of the k-NN search (knn.hip): points sorted by cell, first index of every cell;
//   * the voxel grouping of the volume fit (PMVO.py:705-726 collects the points of a voxel in point order): a
//     stable sort of the voxel keys.
// The sort itself is rocPRIM's LSD radix sort (stable); everything around it is a few map kernels.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "mh_device.h"

__global__ __launch_bounds__(256) void mh_cell_key_kernel(const float *__restrict__ pts, int M, float ox, float oy,
                                                          float oz, float h, int dx, int dy, int dz,
                                                          unsigned int *__restrict__ keys,
                                                          int32_t *__restrict__ vals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    // the cell of a point, exactly as mh_knn_kernel computes the cell of a query
    const int cx = min(max((int)floorf((pts[3 * i] - ox) / h), 0), dx - 1);
    const int cy = min(max((int)floorf((pts[3 * i + 1] - oy) / h), 0), dy - 1);
    const int cz = min(max((int)floorf((pts[3 * i + 2] - oz) / h), 0), dz - 1);
    keys[i] = (unsigned)((cz * dy + cy) * dx + cx);
    vals[i] = i;
}

__global__ __launch_bounds__(256) void mh_gather3_kernel(const float *__restrict__ pts,
                                                         const int32_t *__restrict__ order, int M,
                                                         float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const int s = order[i];
    out[3 * i] = pts[3 * s];
    out[3 * i + 1] = pts[3 * s + 1];
    out[3 * i + 2] = pts[3 * s + 2];
}

// cell_start[c] = first sorted position whose key is >= c  (c = 0 .. ncell, so cell_start[ncell] = M)
__global__ __launch_bounds__(256) void mh_cell_start_kernel(const unsigned int *__restrict__ keys, int M,
                                                            long long ncell, int32_t *__restrict__ start) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c > ncell) return;
    int lo = 0, hi = M;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((long long)keys[mid] < c) lo = mid + 1;
        else hi = mid;
    }
    start[c] = lo;
}

__global__ __launch_bounds__(256) void mh_count_runs_kernel(const unsigned int *__restrict__ keys, int M,
                                                            int32_t *__restrict__ count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool head = i < M && (i == 0 || keys[i] != keys[i - 1]);
    const unsigned long long b = __ballot(head);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(count, __popcll(b));
}

__global__ __launch_bounds__(256) void mh_iota_kernel(int32_t *__restrict__ v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = i;
}

// p2v of the volume fit (PMVO_utils.py:386-404) in float64, as numpy evaluates it: y and z negated, (p - min) / size,
// round half to even, the int32 cast of x86 (NaN and out-of-range -> INT_MIN), clip to the grid -> key (x*gy + y)*gz + z
template <typename T>
__global__ __launch_bounds__(256) void mh_voxel_key_kernel(const T *__restrict__ pts, int n, double mx, double my,
                                                           double mz, double vs, int gx, int gy, int gz,
                                                           unsigned long long *__restrict__ keys) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    auto q = [vs](double p, double m, int g) {
        const double v = rint((p - m) / vs);
        const int k = (v >= -2147483648.0 && v < 2147483648.0) ? (int)v : (int)0x80000000;
        return min(max(k, 0), g - 1);
    };
    const long long x = q((double)pts[3 * i], mx, gx);
    const long long y = q((double)(-pts[3 * i + 1]), my, gy);
    const long long z = q((double)(-pts[3 * i + 2]), mz, gz);
    keys[i] = (unsigned long long)((x * gy + y) * gz + z);
}

// rows of `ori` in sorted order, sign canonicalised as PMVO.py:697-698 does before the fit (y > 0 -> negated)
__global__ __launch_bounds__(256) void mh_gather3_canon_kernel(const float *__restrict__ ori,
                                                               const int32_t *__restrict__ order, int n,
                                                               float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s = order[i];
    const float a = ori[3 * s], b = ori[3 * s + 1], c = ori[3 * s + 2];
    const bool up = b > 0.0f;
    out[3 * i] = up ? a * -1.0f : a;
    out[3 * i + 1] = up ? b * -1.0f : b;
    out[3 * i + 2] = up ? c * -1.0f : c;
}

static size_t align256(size_t x) { return (x + 255) / 256 * 256; }

static size_t radix_temp_u32(unsigned n) {
    size_t b = 0;
    (void)rocprim::radix_sort_pairs(nullptr, b, (const unsigned int *)nullptr, (unsigned int *)nullptr,
                                    (const int32_t *)nullptr, (int32_t *)nullptr, n, 0, 32, (hipStream_t)0);
    return b;
}
static size_t radix_temp_u64(unsigned n) {
    size_t b = 0;
    (void)rocprim::radix_sort_pairs(nullptr, b, (const unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                    (const int32_t *)nullptr, (int32_t *)nullptr, n, 0, 64, (hipStream_t)0);
    return b;
}

extern "C" size_t mh_grid_scratch_bytes_impl(int M) {
    const size_t m = (size_t)(M > 0 ? M : 1);
    return 3 * align256(m * 4) + align256(radix_temp_u32((unsigned)m)) + 256;
}

extern "C" size_t mh_sort_scratch_bytes_impl(int n) {
    const size_t m = (size_t)(n > 0 ? n : 1);
    return align256(m * 4) + align256(radix_temp_u64((unsigned)m)) + 256;
}

// scratch: keys_in | keys_out | vals_in | rocPRIM temp
extern "C" int mh_launch_grid_build(const float *pts, int M, float ox, float oy, float oz, float h, int dx, int dy,
                                    int dz, void *scratch, size_t scratch_bytes, float *pts_sorted, int32_t *order,
                                    int32_t *cell_start, int32_t *n_occupied, hipStream_t st) {
    char *base = (char *)scratch;
    const size_t a = align256((size_t)M * 4);
    unsigned int *kin = (unsigned int *)base, *kout = (unsigned int *)(base + a);
    int32_t *vin = (int32_t *)(base + 2 * a);
    void *tmp = base + 3 * a;
    size_t tb = scratch_bytes - 3 * a;
    const int nb = (M + 255) / 256;
    hipLaunchKernelGGL(mh_cell_key_kernel, dim3(nb), dim3(256), 0, st, pts, M, ox, oy, oz, h, dx, dy, dz, kin, vin);
    const long long ncell = (long long)dx * dy * dz;
    int bits = 1;
    while (bits < 32 && (1ll << bits) < ncell) ++bits;
    hipError_t e = rocprim::radix_sort_pairs(tmp, tb, kin, kout, vin, order, (unsigned)M, 0, bits, st);
    if (e != hipSuccess) return (int)e;
    if (pts_sorted) hipLaunchKernelGGL(mh_gather3_kernel, dim3(nb), dim3(256), 0, st, pts, order, M, pts_sorted);
    if (cell_start)
        hipLaunchKernelGGL(mh_cell_start_kernel, dim3((unsigned)((ncell + 1 + 255) / 256)), dim3(256), 0, st, kout, M,
                           ncell, cell_start);
    if (n_occupied) {
        e = hipMemsetAsync(n_occupied, 0, sizeof(int32_t), st);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(mh_count_runs_kernel, dim3(nb), dim3(256), 0, st, kout, M, n_occupied);
    }
    return (int)hipGetLastError();
}

// stable sort of n 64-bit keys (bits [0, end_bit)) -> keys_out, order (original positions)
extern "C" int mh_launch_sort_keys(const unsigned long long *keys, int n, int end_bit, void *scratch,
                                   size_t scratch_bytes, unsigned long long *keys_out, int32_t *order,
                                   hipStream_t st) {
    char *base = (char *)scratch;
    const size_t a = align256((size_t)n * 4);
    int32_t *vin = (int32_t *)base;
    void *tmp = base + a;
    size_t tb = scratch_bytes - a;
    hipLaunchKernelGGL(mh_iota_kernel, dim3((n + 255) / 256), dim3(256), 0, st, vin, n);
    hipError_t e = rocprim::radix_sort_pairs(tmp, tb, keys, keys_out, vin, order, (unsigned)n, 0, end_bit, st);
    if (e != hipSuccess) return (int)e;
    return (int)hipGetLastError();
}

extern "C" size_t mh_voxel_group_scratch_bytes_impl(int n) {
    const size_t m = (size_t)(n > 0 ? n : 1);
    return align256(m * 8) + mh_sort_scratch_bytes_impl(n);
}

// voxel keys of the points (float64 p2v), stable sort, orientation rows gathered in that order (canonicalised)
extern "C" int mh_launch_voxel_group(const void *pts, int pts_f64, const float *ori, int n, const double *vmin, double vs,
                                     const int32_t *dims, void *scratch, size_t scratch_bytes,
                                     unsigned long long *keys_out, int32_t *order, float *ori_sorted, hipStream_t st) {
    char *base = (char *)scratch;
    const size_t a = align256((size_t)n * 8);
    unsigned long long *kin = (unsigned long long *)base;
    const int nb = (n + 255) / 256;
    if (pts_f64)
        hipLaunchKernelGGL(mh_voxel_key_kernel<double>, dim3(nb), dim3(256), 0, st, (const double *)pts, n, vmin[0],
                           vmin[1], vmin[2], vs, dims[0], dims[1], dims[2], kin);
    else
        hipLaunchKernelGGL(mh_voxel_key_kernel<float>, dim3(nb), dim3(256), 0, st, (const float *)pts, n, vmin[0], vmin[1],
                           vmin[2], vs, dims[0], dims[1], dims[2], kin);
    const long long ncell = (long long)dims[0] * dims[1] * dims[2];
    int bits = 1;
    while (bits < 63 && (1ll << bits) < ncell) ++bits;
    const int rc = mh_launch_sort_keys(kin, n, bits, base + a, scratch_bytes - a, keys_out, order, st);
    if (rc) return rc;
    if (ori && ori_sorted)
        hipLaunchKernelGGL(mh_gather3_canon_kernel, dim3(nb), dim3(256), 0, st, ori, order, n, ori_sorted);
    return (int)hipGetLastError();
}

// forces this translation unit's code object onto the device (HIP loads a fat binary on the first use of one of its kernels:
// 2-20 ms each, which a one-shot pass would pay in the middle of its stages); called from mh_ctx_create
extern "C" int mh_preload_sortgroup() {
    hipFuncAttributes a;
    return (int)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&mh_cell_key_kernel));
}
