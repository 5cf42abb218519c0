// consensus.hip -- orientation consensus (K11/K12): compute_points_similarity
// (Utils/PMVO_utils.py:366-382), the medoid of a group of 3D directions under |cos| similarity:
//     argmax_k mean_j max(cos(o_k,o_j), cos(-o_k,o_j))      (self term included, first max wins)
// used on the 100 nearest neighbours of every point (PMVO.py:626,678) and on the points of every voxel
// (PMVO.py:717-726).  One workgroup per group; unit vectors are staged once in LDS, each lane owns a
// candidate k and walks all j (LDS broadcast reads), then a wave-shuffle + LDS argmax picks the winner.
// cos = (x0*y0 + x1*y1) + x2*y2 on vectors normalised as torch.cosine_similarity does; cos(-o_k,o_j) is
// the exact negation, so the max is |cos|.
// The mean over j is torch.mean(dim=-1) of a contiguous [N,K,K] tensor = ATen's sum over the innermost
// dimension (aten/src/ATen/native/cpu/SumKernel.cpp, its AVX2 build: 8 floats per vector) divided by K, and is
// evaluated in exactly that order (MhInnerSum below; oracle/consensus_oracle.c states the same rule in C and is
// pinned to the reference on 24 group sizes, tests/golden/consensus*.npz):
//   K >= 8: the K/8 full vectors go round-robin to 4 vector accumulators = element j to accumulator j mod 32 while
//           j < 32*(K/32), through multi_row_sum's cascade (level 0 flushed into level 1 every 16 rows of 32 elements,
//           level 1 into level 2 every 256 rows, ...); vectors left over go to accumulator vector 0; the four
//           vectors are added 0+1+2+3; a scalar starting at 0 takes the K%8 tail elements, then the 8 lanes in order;
//   K < 8:  the same scheme on scalars: 4 accumulators, leftover elements into accumulator 0, then 0+1+2+3.
#include "mh_device.h"

#define MH_MEDOID_MAXK 4096

__device__ __forceinline__ bool mh_arg_better(float av, int ai, float bv, int bi) {
    // torch.argmax: NaN is the maximum; first index among equals
    const bool an = av != av, bn = bv != bv;
    if (an || bn) return (an && bn) ? (ai < bi) : an;
    return (av > bv) || (av == bv && ai < bi);
}

// ATen's inner-dimension sum of the K values x(0..K-1), consumed in index order.  BIG adds the cascade levels that
// only groups of 512 and more members reach (96 more registers per lane).
template <bool BIG>
struct MhInnerSum {
    float a0[32];
    float a1[BIG ? 32 : 1], a2[BIG ? 32 : 1], a3[BIG ? 32 : 1];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int a = 0; a < 32; ++a) a0[a] = 0.0f;
        if constexpr (BIG) {
#pragma unroll
            for (int a = 0; a < 32; ++a) a1[a] = a2[a] = a3[a] = 0.0f;
        }
    }
    // row r = elements 32 r .. 32 r + 31 (four vectors of eight), r < K / 32
    template <typename F>
    __device__ __forceinline__ void row(int r, F x) {
#pragma unroll
        for (int a = 0; a < 32; ++a) a0[a] = a0[a] + x(32 * r + a);
        if constexpr (BIG) {
            const int i = r + 1;
            if ((i & 15) == 0) {   // a full level-0 block of 16 rows
#pragma unroll
                for (int a = 0; a < 32; ++a) {
                    a1[a] = a1[a] + a0[a];
                    a0[a] = 0.0f;
                }
                if ((i & 0xF0) == 0) {
#pragma unroll
                    for (int a = 0; a < 32; ++a) {
                        a2[a] = a2[a] + a1[a];
                        a1[a] = 0.0f;
                    }
                    if ((i & 0xF00) == 0) {
#pragma unroll
                        for (int a = 0; a < 32; ++a) {
                            a3[a] = a3[a] + a2[a];
                            a2[a] = 0.0f;
                        }
                    }
                }
            }
        }
    }
    // everything behind the last full row (K >= 8): x is called for j = 32 (K/32) .. K-1
    template <typename F>
    __device__ __forceinline__ float finish(int K, F x) {
        if constexpr (BIG) {
#pragma unroll
            for (int a = 0; a < 32; ++a) a0[a] = ((a0[a] + a1[a]) + a2[a]) + a3[a];
        }
        const int vec_size = K >> 3, R = vec_size >> 2;
        for (int i = R * 4; i < vec_size; ++i) {
#pragma unroll
            for (int l = 0; l < 8; ++l) a0[l] = a0[l] + x(8 * i + l);
        }
        float p[8];
#pragma unroll
        for (int l = 0; l < 8; ++l) p[l] = ((a0[l] + a0[8 + l]) + a0[16 + l]) + a0[24 + l];
        float fin = 0.0f;
        for (int j = vec_size * 8; j < K; ++j) fin = fin + x(j);
#pragma unroll
        for (int l = 0; l < 8; ++l) fin = fin + p[l];
        return fin;
    }
};

// K < 8: ATen's scalar_inner_sum (row_sum with four scalar accumulators)
template <typename F>
__device__ __forceinline__ float mh_inner_sum_small(int K, F x) {
    float q[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const int full = (K >> 2) << 2;
    for (int j = 0; j < full; j += 4) {
#pragma unroll
        for (int a = 0; a < 4; ++a) q[a] = q[a] + x(j + a);
    }
    for (int j = full; j < K; ++j) q[0] = q[0] + x(j);
    return ((q[0] + q[1]) + q[2]) + q[3];
}

// big_pass = 0: groups below 512 members (one cascade level); 1: only the larger ones (all four levels)
template <bool BIG>
__global__ __launch_bounds__(256) void mh_medoid_kernel(const float *__restrict__ ori,
                                                        const int32_t *__restrict__ seg_start, int K_dense,
                                                        float *__restrict__ out, int32_t *__restrict__ out_index,
                                                        const int32_t *__restrict__ index) {
    extern __shared__ __attribute__((aligned(16))) float s_u[];   // [K][3] unit vectors (+ reduction scratch)
    __shared__ float s_bv[4];
    __shared__ int s_bi[4];
    const int g = blockIdx.x, tid = threadIdx.x;
    const int nt = (int)blockDim.x;   // 256, or 128 for the dense groups of up to 128 members (the 100 neighbours of refine)
    const int begin = seg_start ? seg_start[g] : g * K_dense;
    const int K = seg_start ? (seg_start[g + 1] - begin) : K_dense;
    if (K <= 0 || (K >= 512) != BIG) return;   // the other launch takes this group
    const float *__restrict__ o = ori + (size_t)begin * 3;
    // dense groups may be given as K row indices into `ori` (the 100 nearest neighbours of refine, PMVO.py:612-618)
    // instead of a materialised [G,K,3] gather
    const int32_t *__restrict__ gidx = index ? index + (size_t)g * K_dense : nullptr;
    auto row_of = [&](int k) -> const float * { return gidx ? ori + (size_t)gidx[k] * 3 : o + 3 * k; };
    auto unit_of = [&](int k, float &u0, float &u1, float &u2) {
        const float *__restrict__ rw = row_of(k);
        const float x0 = rw[0], x1 = rw[1], x2 = rw[2];
        float s = x0 * x0;
        s = mh_fma(x1, x1, s);
        s = mh_fma(x2, x2, s);
        float nrm = __builtin_sqrtf(s);
        nrm = (nrm < 1e-8f) ? 1e-8f : nrm;
        u0 = x0 / nrm;
        u1 = x1 / nrm;
        u2 = x2 / nrm;
    };
    float bv = 0.0f;
    int bi = 0x7fffffff;
    if (K <= MH_MEDOID_MAXK) {
        for (int k = tid; k < K; k += nt) unit_of(k, s_u[3 * k], s_u[3 * k + 1], s_u[3 * k + 2]);
        __syncthreads();
        for (int k = tid; k < K; k += nt) {
            const float a0 = s_u[3 * k], a1 = s_u[3 * k + 1], a2 = s_u[3 * k + 2];
            auto x = [&](int j) {
                return __builtin_fabsf((a0 * s_u[3 * j] + a1 * s_u[3 * j + 1]) + a2 * s_u[3 * j + 2]);
            };
            float sum;
            if (K < 8) {
                sum = mh_inner_sum_small(K, x);
            } else {
                MhInnerSum<BIG> acc;
                acc.init();
                const int R = K >> 5;
                for (int r = 0; r < R; ++r) acc.row(r, x);
                sum = acc.finish(K, x);
            }
            const float mean = sum / (float)K;
            if (bi == 0x7fffffff || mh_arg_better(mean, k, bv, bi)) {
                bv = mean;
                bi = k;
            }
        }
    } else {
        // a group that does not fit in LDS (never the case for the 2.5 mm voxels of a real capture): the units are
        // staged MH_MEDOID_MAXK (= 128 rows of 32) at a time, every lane carries the accumulators of its candidate
        if constexpr (BIG) {
            for (int kb = 0; kb < K; kb += nt) {
                const int k = kb + tid;
                float a0 = 0.f, a1 = 0.f, a2 = 0.f;
                if (k < K) unit_of(k, a0, a1, a2);
                MhInnerSum<true> acc;
                acc.init();
                float sum = 0.0f;
                for (int j0 = 0; j0 < K; j0 += MH_MEDOID_MAXK) {
                    const int nj = min(MH_MEDOID_MAXK, K - j0);
                    __syncthreads();
                    for (int j = tid; j < nj; j += nt) unit_of(j0 + j, s_u[3 * j], s_u[3 * j + 1], s_u[3 * j + 2]);
                    __syncthreads();
                    auto x = [&](int j) {
                        const int q = j - j0;
                        return __builtin_fabsf((a0 * s_u[3 * q] + a1 * s_u[3 * q + 1]) + a2 * s_u[3 * q + 2]);
                    };
                    if (k < K) {
                        const int r1 = min((j0 + nj) >> 5, K >> 5);
                        for (int r = j0 >> 5; r < r1; ++r) acc.row(r, x);
                        if (j0 + nj == K) sum = acc.finish(K, x);   // the remainder lies in this (last) piece
                    }
                }
                if (k < K) {
                    const float mean = sum / (float)K;
                    if (bi == 0x7fffffff || mh_arg_better(mean, k, bv, bi)) {
                        bv = mean;
                        bi = k;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(bv, off);
        const int oi = __shfl_xor(bi, off);
        if (oi != 0x7fffffff && (bi == 0x7fffffff || mh_arg_better(ov, oi, bv, bi))) {
            bv = ov;
            bi = oi;
        }
    }
    if ((tid & 63) == 0) {
        s_bv[tid >> 6] = bv;
        s_bi[tid >> 6] = bi;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < (nt >> 6); ++w)
            if (s_bi[w] != 0x7fffffff && (bi == 0x7fffffff || mh_arg_better(s_bv[w], s_bi[w], bv, bi))) {
                bv = s_bv[w];
                bi = s_bi[w];
            }
        const float *__restrict__ rw = row_of(bi);
        out[3 * g] = rw[0];
        out[3 * g + 1] = rw[1];
        out[3 * g + 2] = rw[2];
        if (out_index) out_index[g] = bi;
    }
}

// refine's replacement rule (PMVO.py:631-636): ori[n] <- center[n] where max(cos(center,ori), cos(center,-ori))
// < thr, with cos as torch.cosine_similarity evaluates it on [N,3] fp32 tensors (norms = sqrt of an fma chain,
// clamped at 1e-8; products rounded separately, added left to right; the second cosine is the exact negation).
__global__ __launch_bounds__(256) void mh_replace_dissimilar_kernel(const float *__restrict__ center,
                                                                    float *__restrict__ ori, float thr, int N) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float c[3], o[3], cu[3], ou[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        c[k] = center[3 * n + k];
        o[k] = ori[3 * n + k];
    }
    float sc = c[0] * c[0];
    sc = mh_fma(c[1], c[1], sc);
    sc = mh_fma(c[2], c[2], sc);
    float so = o[0] * o[0];
    so = mh_fma(o[1], o[1], so);
    so = mh_fma(o[2], o[2], so);
    float nc = __builtin_sqrtf(sc), no = __builtin_sqrtf(so);
    nc = (nc < 1e-8f) ? 1e-8f : nc;
    no = (no < 1e-8f) ? 1e-8f : no;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        cu[k] = c[k] / nc;
        ou[k] = o[k] / no;
    }
    const float cs = (cu[0] * ou[0] + cu[1] * ou[1]) + cu[2] * ou[2];
    if (__builtin_fabsf(cs) < thr) {
#pragma unroll
        for (int k = 0; k < 3; ++k) ori[3 * n + k] = c[k];
    }
}

extern "C" int mh_launch_replace_dissimilar(const float *center, float *ori, float thr, int N, hipStream_t st) {
    hipLaunchKernelGGL(mh_replace_dissimilar_kernel, dim3((N + 255) / 256), dim3(256), 0, st, center, ori, thr, N);
    return (int)hipGetLastError();
}

extern "C" int mh_launch_medoid_dense(const float *ori, const int32_t *index, int G, int K, float *out, int32_t *out_index,
                                      hipStream_t st) {
    const int kl = K < MH_MEDOID_MAXK ? K : MH_MEDOID_MAXK;
    // (a lane owns a candidate: with 100 neighbours a 256-thread workgroup has two waves that only wait at the barriers)
    if (K < 512)
        hipLaunchKernelGGL(mh_medoid_kernel<false>, dim3(G), dim3(K <= 128 ? 128 : 256), (size_t)kl * 3 * sizeof(float), st,
                           ori, nullptr, K, out, out_index, index);
    else
        hipLaunchKernelGGL(mh_medoid_kernel<true>, dim3(G), dim3(256), (size_t)kl * 3 * sizeof(float), st, ori, nullptr,
                           K, out, out_index, index);
    return (int)hipGetLastError();
}

extern "C" int mh_launch_medoid_segmented(const float *ori, const int32_t *seg_start, int G, int max_group,
                                          float *out, int32_t *out_index, hipStream_t st) {
    // two launches over the same groups: every workgroup looks at its group's size and leaves if the other form
    // owns it (the four-level accumulators cost registers that the common small groups should not pay for)
    const int ks = max_group < 511 ? max_group : 511;
    hipLaunchKernelGGL(mh_medoid_kernel<false>, dim3(G), dim3(256), (size_t)(ks > 0 ? ks : 1) * 3 * sizeof(float), st,
                       ori, seg_start, 0, out, out_index, nullptr);
    if (max_group >= 512) {
        const int kl = max_group < MH_MEDOID_MAXK ? max_group : MH_MEDOID_MAXK;
        hipLaunchKernelGGL(mh_medoid_kernel<true>, dim3(G), dim3(256), (size_t)kl * 3 * sizeof(float), st, ori,
                           seg_start, 0, out, out_index, nullptr);
    }
    return (int)hipGetLastError();
}

// forces this translation unit's code object onto the device (HIP loads a fat binary on the first use of one of its kernels:
// 2-20 ms each, which a one-shot pass would pay in the middle of its stages); called from mh_ctx_create
extern "C" int mh_preload_consensus() {
    hipFuncAttributes a;
    return (int)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&mh_medoid_kernel<false>));
}
