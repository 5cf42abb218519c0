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

// ---------------------------------------------------------------------------------------------------------------
// Stable stream compaction on the device: the `points[index]`, `ori[index]`, `centres[keep]` and `np.concatenate` steps
// between the stages of refine (PMVO.py:651-693) and the run boundaries of the sorted voxel keys (:705-715) without a
// host round trip.  Three small launches: per-block counts, one-block exclusive scan (+ an optional base offset read
// from device memory, so that a second selection appends to the first), scatter.  Order is preserved: an element's
// output position is base + the number of selected elements before it.
// ---------------------------------------------------------------------------------------------------------------
#define MH_SEL_ROUNDS 4
#define MH_SEL_ITEMS (256 * MH_SEL_ROUNDS)

struct MhPredFlags {     // f[i] && !(g && g[i]), optionally inverted
    const uint8_t *f, *g;
    int invert;
    __device__ __forceinline__ bool operator()(int i) const {
        const bool v = f[i] != 0 && !(g && g[i] != 0);
        return v != (invert != 0);
    }
};
struct MhPredHeads {     // first element of a run of equal sorted keys
    const unsigned long long *k;
    __device__ __forceinline__ bool operator()(int i) const { return i == 0 || k[i] != k[i - 1]; }
};

template <class P>
__global__ __launch_bounds__(256) void mh_sel_count_kernel(P pred, int n, int32_t *__restrict__ block_cnt) {
    __shared__ int s_c[4];
    int c = 0;
#pragma unroll
    for (int r = 0; r < MH_SEL_ROUNDS; ++r) {
        const int i = blockIdx.x * MH_SEL_ITEMS + r * 256 + threadIdx.x;
        c += __popcll(__ballot(i < n && pred(i)));
    }
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_cnt[blockIdx.x] = (s_c[0] + s_c[1]) + (s_c[2] + s_c[3]);
}

// block_cnt[b] <- base + (number selected in blocks before b); total[0] = base + number selected; optional: tail[total] =
// tail_value (the closing entry of a segment table) and total[1] = 0 (the running maximum a later kernel fills)
__global__ __launch_bounds__(256) void mh_sel_scan_kernel(int32_t *__restrict__ block_cnt, int nb,
                                                          const int32_t *__restrict__ base, int32_t *__restrict__ total,
                                                          int32_t *__restrict__ tail, int tail_value, int clear_second) {
    __shared__ int s_w[4];
    int carry = base ? base[0] : 0;
    for (int b0 = 0; b0 < nb; b0 += 256) {
        const int b = b0 + threadIdx.x;
        const int v = b < nb ? block_cnt[b] : 0;
        int x = v;      // inclusive scan inside the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int y = __shfl_up(x, off);
            if ((int)(threadIdx.x & 63) >= off) x += y;
        }
        if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = x;
        __syncthreads();
        int before = 0;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) before += s_w[w];
        if (b < nb) block_cnt[b] = carry + before + x - v;
        carry += (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        total[0] = carry;
        if (clear_second) total[1] = 0;
        if (tail) tail[carry] = tail_value;
    }
}

template <class P>
__global__ __launch_bounds__(256) void mh_sel_scatter_kernel(P pred, int n, const int32_t *__restrict__ block_off,
                                                             const float *__restrict__ a, const float *__restrict__ b,
                                                             float *__restrict__ a_out, float *__restrict__ b_out,
                                                             int32_t *__restrict__ idx_out,
                                                             const unsigned long long *__restrict__ key_in,
                                                             unsigned long long *__restrict__ key_out) {
    __shared__ int s_c[4];
    int off = block_off[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < MH_SEL_ROUNDS; ++r) {
        const int i = blockIdx.x * MH_SEL_ITEMS + r * 256 + threadIdx.x;
        const bool sel = i < n && pred(i);
        const unsigned long long m = __ballot(sel);
        if (lane == 0) s_c[wave] = __popcll(m);
        __syncthreads();
        int before = 0;
        for (int w = 0; w < wave; ++w) before += s_c[w];
        if (sel) {
            const int pos = off + before + __popcll(m & ((1ull << lane) - 1ull));
            if (a_out) {
                a_out[3 * (size_t)pos] = a[3 * (size_t)i];
                a_out[3 * (size_t)pos + 1] = a[3 * (size_t)i + 1];
                a_out[3 * (size_t)pos + 2] = a[3 * (size_t)i + 2];
            }
            if (b_out) {
                b_out[3 * (size_t)pos] = b[3 * (size_t)i];
                b_out[3 * (size_t)pos + 1] = b[3 * (size_t)i + 1];
                b_out[3 * (size_t)pos + 2] = b[3 * (size_t)i + 2];
            }
            if (idx_out) idx_out[pos] = i;
            if (key_out) key_out[pos] = key_in[i];
        }
        off += (s_c[0] + s_c[1]) + (s_c[2] + s_c[3]);
        __syncthreads();
    }
}

// meta[1] = max over the G = meta[0] segments of seg[g+1] - seg[g]   (meta[1] cleared by the scan kernel)
__global__ __launch_bounds__(256) void mh_seg_max_kernel(const int32_t *__restrict__ seg, int32_t *__restrict__ meta, int cap) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    const int G = meta[0];
    int d = (g < G && g < cap) ? seg[g + 1] - seg[g] : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) d = max(d, __shfl_xor(d, off));
    if ((threadIdx.x & 63) == 0 && d > 0) atomicMax(meta + 1, d);
}

__global__ __launch_bounds__(256) void mh_flag_less_kernel(const float *__restrict__ x, float thr, int n,
                                                           uint8_t *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = x[i] < thr ? 1 : 0;      // NaN: not selected, as `np.where(min_loss < threshold)` (PMVO.py:651)
}

// flag[0] = 1 when the two buffers differ in any 32-bit word (bitwise: NaN == NaN, -0 != +0)
__global__ __launch_bounds__(256) void mh_words_differ_kernel(const uint32_t *__restrict__ a,
                                                              const uint32_t *__restrict__ b, size_t nwords,
                                                              int32_t *__restrict__ flag) {
    bool diff = false;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (size_t)gridDim.x * 256)
        diff |= a[i] != b[i];
    if (__ballot(diff) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

extern "C" int mh_launch_words_differ(const void *a, const void *b, size_t nwords, int32_t *flag, hipStream_t st) {
    hipError_t e = hipMemsetAsync(flag, 0, sizeof(int32_t), st);
    if (e != hipSuccess) return (int)e;
    if (nwords) {
        const unsigned nb = (unsigned)((nwords + 1023) / 1024 > 2048 ? 2048 : (nwords + 1023) / 1024);
        hipLaunchKernelGGL(mh_words_differ_kernel, dim3(nb), dim3(256), 0, st, (const uint32_t *)a, (const uint32_t *)b,
                           nwords, flag);
    }
    return (int)hipGetLastError();
}

// bounding box of M float32 points -> out[6] = {min x,y,z, max x,y,z} (ordered-int atomics on the float bits; out pre-set
// by the launcher).  The grid of the neighbour search needs it; torch.aminmax over dim 0 of an [M,3] tensor takes 170-280 us
// and, in a one-shot process, 20+ ms of code-object loading on its first use.
__device__ __forceinline__ int mh_f2ord(float f) {
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__global__ __launch_bounds__(256) void mh_bbox_kernel(const float *__restrict__ pts, int M, int *__restrict__ acc) {
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < M; i += gridDim.x * 256)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float v = pts[3 * i + k];
            lo[k] = fminf(lo[k], v);
            hi[k] = fmaxf(hi[k], v);
        }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], o));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], o));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(acc + k, mh_f2ord(lo[k]));
            atomicMax(acc + 3 + k, mh_f2ord(hi[k]));
        }
    }
}
__global__ void mh_bbox_init_kernel(int *acc) {
    if (threadIdx.x < 3) acc[threadIdx.x] = 0x7fffffff;
    else if (threadIdx.x < 6) acc[threadIdx.x] = (int)0x80000000;
}
__global__ void mh_bbox_done_kernel(const int *acc, float *out) {
    if (threadIdx.x < 6) {
        const int i = acc[threadIdx.x];
        out[threadIdx.x] = __int_as_float(i >= 0 ? i : i ^ 0x7fffffff);
    }
}
extern "C" int mh_launch_points_bbox(const float *pts, int M, float *out6, hipStream_t st) {
    int *acc = reinterpret_cast<int *>(out6);       // (the six floats double as the accumulators, converted in place at the end)
    hipLaunchKernelGGL(mh_bbox_init_kernel, dim3(1), dim3(64), 0, st, acc);
    const int nb = M > 0 ? ((M + 255) / 256 < 512 ? (M + 255) / 256 : 512) : 0;
    if (nb) hipLaunchKernelGGL(mh_bbox_kernel, dim3(nb), dim3(256), 0, st, pts, M, acc);
    hipLaunchKernelGGL(mh_bbox_done_kernel, dim3(1), dim3(64), 0, st, acc, out6);
    return (int)hipGetLastError();
}

// a small host -> device upload as a KERNEL that reads the page-locked host buffer over the link (mh_upload_pinned)
__global__ __launch_bounds__(256) void mh_copy_words_kernel(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst,
                                                            size_t nwords) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
extern "C" int mh_launch_copy_words(const void *src, void *dst, size_t nwords, hipStream_t st) {
    const unsigned nb = (unsigned)((nwords + 255) / 256 > 1024 ? 1024 : (nwords + 255) / 256);
    hipLaunchKernelGGL(mh_copy_words_kernel, dim3(nb), dim3(256), 0, st, (const uint32_t *)src, (uint32_t *)dst, nwords);
    return (int)hipGetLastError();
}

extern "C" size_t mh_select_scratch_bytes_impl(int n) {
    const size_t nb = ((size_t)(n > 0 ? n : 1) + MH_SEL_ITEMS - 1) / MH_SEL_ITEMS;
    return align256(nb * sizeof(int32_t)) + 256;
}

template <class P>
static int mh_select_run(P pred, int n, const float *a, const float *b, float *a_out, float *b_out, int32_t *idx_out,
                         const unsigned long long *key_in, unsigned long long *key_out, const int32_t *base,
                         int32_t *total, int32_t *tail, int tail_value, int clear_second, void *scratch, hipStream_t st) {
    int32_t *blk = (int32_t *)scratch;
    const int nb = (n + MH_SEL_ITEMS - 1) / MH_SEL_ITEMS;
    if (nb > 0) hipLaunchKernelGGL(mh_sel_count_kernel<P>, dim3(nb), dim3(256), 0, st, pred, n, blk);
    hipLaunchKernelGGL(mh_sel_scan_kernel, dim3(1), dim3(256), 0, st, blk, nb, base, total, tail, tail_value, clear_second);
    if (nb > 0)
        hipLaunchKernelGGL(mh_sel_scatter_kernel<P>, dim3(nb), dim3(256), 0, st, pred, n, blk, a, b, a_out, b_out, idx_out,
                           key_in, key_out);
    return (int)hipGetLastError();
}

extern "C" int mh_launch_select_rows(const uint8_t *flags, const uint8_t *veto, int invert, int n, const float *a,
                                     const float *b, float *a_out, float *b_out, int32_t *idx_out, const int32_t *base,
                                     int32_t *count, void *scratch, hipStream_t st) {
    return mh_select_run(MhPredFlags{flags, veto, invert}, n, a, b, a_out, b_out, idx_out, nullptr, nullptr, base, count,
                         nullptr, 0, 0, scratch, st);
}

// seg_start[0..G) = first positions of the runs of equal keys, seg_start[G] = n, head_keys[0..G) = their keys,
// meta = {G, largest run}
extern "C" int mh_launch_segment_heads(const unsigned long long *keys, int n, int32_t *seg_start,
                                       unsigned long long *head_keys, int32_t *meta, void *scratch, hipStream_t st) {
    const int rc = mh_select_run(MhPredHeads{keys}, n, nullptr, nullptr, nullptr, nullptr, seg_start, keys, head_keys,
                                 nullptr, meta, seg_start, n, 1, scratch, st);
    if (rc) return rc;
    if (n > 0) hipLaunchKernelGGL(mh_seg_max_kernel, dim3((n + 255) / 256), dim3(256), 0, st, seg_start, meta, n);
    return (int)hipGetLastError();
}

extern "C" int mh_launch_flag_less(const float *x, float thr, int n, uint8_t *out, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(mh_flag_less_kernel, dim3((n + 255) / 256), dim3(256), 0, st, x, thr, n, out);
    return (int)hipGetLastError();
}

// forces this translation unit's code object onto the device (HIP loads a fat binary on the first use of one of its kernels:
// 2-20 ms each, which a one-shot pass would pay in the middle of its stages); called from mh_ctx_create
extern "C" int mh_preload_sortgroup() {
    hipFuncAttributes a;
    return (int)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&mh_cell_key_kernel));
}
