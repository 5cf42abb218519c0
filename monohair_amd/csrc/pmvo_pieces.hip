// pmvo_pieces.hip -- the intermediate methods of the reference's class PMVO as stand-alone kernels, gfx950 only.
// PMVO.forward never calls these here (its search is fused, pmvo_search.hip); they exist so that the class keeps
// the reference's method surface -- project_points (PMVO.py:378-397), get_depth/get_ori/get_conf/get_mask/
// get_ori_patch/get_c_patch (:482-523), compute_visible (:525-529), sample_next_3d_pos (:263-335),
// compute_reproject_ori / compute_points_prj_ori (:219-260), compute_prj_loss (:151-209) -- with the same tensors
// in and out.  Same arithmetic helpers as the fused kernels (mh_device.h), so the same bits.
#include "mh_device.h"

// project_points for ONE camera record: rounded+clamped (row, col), z' = -z/2, out-of-image flag, unrounded pixel
__global__ __launch_bounds__(256) void mh_project_points_kernel(const float *__restrict__ cam,
                                                                const float *__restrict__ pts, int N, int H, int W,
                                                                int32_t *__restrict__ rc, float *__restrict__ zp,
                                                                uint8_t *__restrict__ oobo, float *__restrict__ pixf,
                                                                int batch_rule) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const bool single = batch_rule && N == 1;   // (a batch of one point: mh_cam_project_b)
    float u, w, z, rowf, colf;
    mh_cam_project_b(cam, pts[3 * n], pts[3 * n + 1], pts[3 * n + 2], u, w, z, single);
    mh_ndc_to_pixel(u, w, (float)H, (float)W, rowf, colf);
    float cr = __builtin_rintf(colf), rr = __builtin_rintf(rowf);
    const bool oob = !(cr <= (float)(W - 1)) || (cr < 0.0f) || !(rr <= (float)(H - 1)) || (rr < 0.0f);
    cr = fminf(fmaxf(cr, 0.0f), (float)(W - 1));
    rr = fminf(fmaxf(rr, 0.0f), (float)(H - 1));
    if (rc) {
        rc[2 * n] = (int)rr;
        rc[2 * n + 1] = (int)cr;
    }
    if (zp) zp[n] = -z / 2.0f;
    if (oobo) oobo[n] = oob ? 1 : 0;
    if (pixf) {
        pixf[2 * n] = rowf;
        pixf[2 * n + 1] = colf;
    }
}

// get_*: the records of view `v` at (row, col) [+ the size x size taps, each clamped to the image, row offset outer]
__global__ __launch_bounds__(256) void mh_gather_kernel(MhViews vw, int v, const long long *__restrict__ uv, int N,
                                                        int size, float4 *__restrict__ rec_out,
                                                        float *__restrict__ mask_out) {
    const int P = size * size, hp = size / 2;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)N * P) return;
    const int n = (int)(i / P), p = (int)(i - (long long)n * P);
    const int di = p / size - hp, dj = p - (p / size) * size - hp;
    const int H = vw.H, W = vw.W;
    const int r = min(max((int)uv[2 * n] + di, 0), H - 1), c = min(max((int)uv[2 * n + 1] + dj, 0), W - 1);
    const size_t pix = (size_t)v * H * W + (size_t)r * W + c;
    if (rec_out) rec_out[i] = vw.rec[pix];
    if (mask_out) mask_out[i] = vw.mask[pix];
}

__global__ __launch_bounds__(256) void mh_compute_visible_kernel(const float *__restrict__ depth,
                                                                 const float *__restrict__ z, size_t n,
                                                                 float *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = mh_soft_visible(depth[i], z[i]);
}

// points per base view among base_view[0..N) (the M of mh_group_forms) into rank 0 of copy 0 of a zeroed MhRule::gcnt array
__global__ __launch_bounds__(256) void mh_piece_group_sizes_kernel(const int32_t *__restrict__ base_view, int N, int V,
                                                                   int32_t *__restrict__ gcnt) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int b = base_view[n];
    if (b >= 0 && b < V) atomicAdd(&gcnt[b], 1);
}

// sample_next_3d_pos: out[n, s, :] for the base view of point n (the reference leaves points whose base view index
// matches no camera at zero; indices are assumed valid here)
__global__ __launch_bounds__(256) void mh_sample_next_kernel(MhViews vw, const float *__restrict__ pts,
                                                             const int32_t *__restrict__ base_view,
                                                             const float *__restrict__ ori /*[V,N,2]*/,
                                                             const float *__restrict__ offs, int N, int S,
                                                             float *__restrict__ out, MhRule rule) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)N * S) return;
    const int n = (int)(i / S), s = (int)(i - (long long)n * S);
    const int b = base_view[n];
    float S0 = 0.f, S1 = 0.f, S2 = 0.f;
    if (b >= 0 && b < vw.V) {
        const float2 oc = reinterpret_cast<const float2 *>(ori)[(size_t)b * N + n];
        const float *cam = vw.cams + b * MH_CAM_STRIDE;
        float u, v, z, row, col;
        // (the rounding of the reference's sgemms follows the number of points that share this base view: MhRule)
        const int forms = mh_group_forms(rule, 0, vw.V, b, S);
        if (forms & MH_FORM_GEMV) mh_cam_project_single(cam, pts[3 * n], pts[3 * n + 1], pts[3 * n + 2], u, v, z);
        else mh_cam_project(cam, pts[3 * n], pts[3 * n + 1], pts[3 * n + 2], u, v, z);
        mh_ndc_to_pixel(u, v, (float)vw.H, (float)vw.W, row, col);
        float nx = col + oc.y * 2.0f;
        float ny = row + oc.x * 2.0f;
        nx = nx / (float)vw.W;
        ny = ny / (float)vw.H;
        nx = nx * 2.0f - 1.0f;
        ny = ny * 2.0f - 1.0f;
        nx = -nx;
        mh_cam_unproject(cam, nx, ny, z + offs[s], S0, S1, S2, (forms & MH_FORM_CHAIN) != 0);
    }
    out[3 * i] = S0;
    out[3 * i + 1] = S1;
    out[3 * i + 2] = S2;
}

// compute_reproject_ori: D[v, n, s, :] = pixel(sample[n, s]) - pixel(point[n])   (unrounded (row, col))
__global__ __launch_bounds__(256) void mh_reproject_kernel(MhViews vw, const float *__restrict__ pts,
                                                           const float *__restrict__ samples, int N, int S,
                                                           float *__restrict__ D) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long per_view = (long long)N * S;
    if (i >= per_view * vw.V) return;
    const int v = (int)(i / per_view);
    const long long ns = i - (long long)v * per_view;
    const int n = (int)(ns / S);
    const float *cam = vw.cams + v * MH_CAM_STRIDE;
    float r0, c0, r1, c1;
    // (a batch of one point / one candidate in all: single-column sgemms, mh_cam_project_b)
    mh_pixel_of_b(cam, pts[3 * n], pts[3 * n + 1], pts[3 * n + 2], (float)vw.H, (float)vw.W, r0, c0, vw.batch_rule && N == 1);
    mh_pixel_of_b(cam, samples[3 * ns], samples[3 * ns + 1], samples[3 * ns + 2], (float)vw.H, (float)vw.W, r1, c1,
                  vw.batch_rule && per_view == 1);
    D[2 * i] = r1 - r0;
    D[2 * i + 1] = c1 - c0;
}

// compute_prj_loss: one workgroup per point, one lane per sample (S <= 1024 through a lane loop); taps normalised
// as torch.cosine_similarity does, masked minimum with first-index ties, cascade sums over views, then the
// positive / low-confidence rules and the NaN-propagating first-index minimum over the samples.
__device__ __forceinline__ bool mh_piece_min_better(float al, int ai, float bl, int bi) {
    const bool an = al != al, bn = bl != bl;
    if (an || bn) return (an && bn) ? (ai < bi) : an;
    return (al < bl) || (al == bl && ai < bi);
}

__global__ __launch_bounds__(256) void mh_prj_loss_kernel(const float *__restrict__ D /*[V,N,S,2]*/,
                                                          const float *__restrict__ ori_patch /*[V,N,P,2]*/,
                                                          const float *__restrict__ conf_patch /*[V,N,P]*/,
                                                          const float *__restrict__ vis /*[V,N]*/, int V, int N, int S,
                                                          int P, float thr, float *__restrict__ loss,
                                                          long long *__restrict__ index, uint8_t *__restrict__ hc,
                                                          float *__restrict__ all_loss /*[N,S] or null*/,
                                                          long long tail_col0 /* first trailing column of the [V,N*S] sums */) {
    extern __shared__ float s_buf[];   // [S] losses, then [S] positive flags (as floats)
    __shared__ int s_npos;
    __shared__ float s_bl[4];
    __shared__ int s_bi[4];
    const int n = blockIdx.x, tid = threadIdx.x;
    float *s_loss = s_buf, *s_pos = s_buf + S;
    if (tid == 0) s_npos = 0;
    __syncthreads();
    for (int s = tid; s < S; s += blockDim.x) {
        MhCascV num = {0.f, 0.f, 0.f}, den = {0.f, 0.f, 0.f};
        int cnt = 0;
        // the terms of view v: (min_loss * weight, weight) (PMVO.py:160-196)
        auto term = [&](int v, float &tn, float &td) {
            const size_t vn = (size_t)v * N + n;
            const float2 d = reinterpret_cast<const float2 *>(D)[vn * S + s];
            float dx, dy;
            mh_unit2(d.x, d.y, dx, dy);
            const float *__restrict__ cp = conf_patch + vn * P;
            const float2 *__restrict__ op = reinterpret_cast<const float2 *>(ori_patch) + vn * P;
            float cmax = cp[0];
            for (int p = 1; p < P; ++p) cmax = (cp[p] > cmax) ? cp[p] : cmax;
            const bool high = cmax > thr;
            float ml = 0.f, bc = 0.f;
            for (int p = 0; p < P; ++p) {
                float o0, o1;
                mh_unit2(op[p].x, op[p].y, o0, o1);
                const float cs = o0 * dx + o1 * dy;
                const float l = 1.0f - __builtin_fabsf(cs);
                const float c = cp[p];
                const bool upd = (p == 0) || ((l < ml) && (high ? (c > thr) : true));
                ml = upd ? l : ml;
                bc = upd ? c : bc;
            }
            const float w = (vis[vn] == -1.0f ? 0.0f : 1.0f) * bc;
            tn = ml * w;
            td = w;
        };
        for (int v = 0; v < V; ++v) {
            if (v > 0 && (v & 15) == 0) {
                mh_cascv_flush(num, v);
                mh_cascv_flush(den, v);
            }
            float tn, td;
            term(v, tn, td);
            num.a0 = num.a0 + tn;
            den.a0 = den.a0 + td;
            cnt += (td > 0.0f) ? 1 : 0;
        }
        float dn = mh_cascv_done(den), nm = mh_cascv_done(num);
        if ((long long)n * S + s >= tail_col0) {   // a trailing column: ATen's row_sum order (mh_device.h)
            nm = mh_row_sum_views(V, [&](int v) { float a, b; term(v, a, b); return a; });
            dn = mh_row_sum_views(V, [&](int v) { float a, b; term(v, a, b); return b; });
        }
        const bool pos = (dn / (float)cnt) > thr;
        s_pos[s] = pos ? 1.0f : 0.0f;
        s_loss[s] = nm / dn;
        if (pos) atomicAdd(&s_npos, 1);
    }
    __syncthreads();
    const bool low = s_npos < 5;
    float bl = 0.f;
    int bi = 0x7fffffff;
    for (int s = tid; s < S; s += blockDim.x) {
        float l = s_loss[s];
        if (!low && s_pos[s] == 0.0f) l = 1.0f;
        if (all_loss) all_loss[(size_t)n * S + s] = l;
        if (bi == 0x7fffffff || mh_piece_min_better(l, s, bl, bi)) {
            bl = l;
            bi = s;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ol = __shfl_xor(bl, o);
        const int oi = __shfl_xor(bi, o);
        if (oi != 0x7fffffff && (bi == 0x7fffffff || mh_piece_min_better(ol, oi, bl, bi))) {
            bl = ol;
            bi = oi;
        }
    }
    if ((tid & 63) == 0) {
        s_bl[tid >> 6] = bl;
        s_bi[tid >> 6] = bi;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
            if (s_bi[w] != 0x7fffffff && (bi == 0x7fffffff || mh_piece_min_better(s_bl[w], s_bi[w], bl, bi))) {
                bl = s_bl[w];
                bi = s_bi[w];
            }
        loss[n] = bl;
        if (index) index[n] = bi;
        if (hc) hc[n] = s_pos[bi] != 0.0f ? 1 : 0;
    }
}

extern "C" int mh_launch_project_points(const float *cam, const float *pts, int N, int H, int W, int32_t *rc, float *zp,
                                        uint8_t *oob, float *pixf, int batch_rule, hipStream_t st) {
    hipLaunchKernelGGL(mh_project_points_kernel, dim3((N + 255) / 256), dim3(256), 0, st, cam, pts, N, H, W, rc, zp, oob,
                       pixf, batch_rule);
    return (int)hipGetLastError();
}
extern "C" int mh_launch_gather(MhViews vw, int v, const long long *uv, int N, int size, float4 *rec_out,
                                float *mask_out, hipStream_t st) {
    const long long tot = (long long)N * size * size;
    hipLaunchKernelGGL(mh_gather_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, vw, v, uv, N, size,
                       rec_out, mask_out);
    return (int)hipGetLastError();
}
extern "C" int mh_launch_compute_visible(const float *depth, const float *z, size_t n, float *out, hipStream_t st) {
    hipLaunchKernelGGL(mh_compute_visible_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, depth, z, n, out);
    return (int)hipGetLastError();
}
extern "C" int mh_launch_sample_next(MhViews vw, const float *pts, const int32_t *base_view, const float *ori,
                                     const float *offs, int N, int S, float *out, int rule_mode, int fma_min_cols,
                                     int32_t *gcnt /* MH_GROUP_COPIES * MH_GROUP_RANKS * V ints of work space (rule_mode 0) */,
                                     hipStream_t st) {
    const long long tot = (long long)N * S;
    MhRule rule = {};
    rule.mode = rule_mode;
    rule.fma_min_cols = fma_min_cols;
    if (rule_mode == 0) {
        if (!gcnt) return -1;
        if (hipMemsetAsync(gcnt, 0, sizeof(int32_t) * (size_t)MH_GROUP_COPIES * MH_GROUP_RANKS * vw.V, st) != hipSuccess)
            return -1;
        hipLaunchKernelGGL(mh_piece_group_sizes_kernel, dim3((N + 255) / 256), dim3(256), 0, st, base_view, N, vw.V, gcnt);
        rule.gcnt = gcnt;
    }
    hipLaunchKernelGGL(mh_sample_next_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, vw, pts, base_view,
                       ori, offs, N, S, out, rule);
    return (int)hipGetLastError();
}
extern "C" int mh_launch_reproject(MhViews vw, const float *pts, const float *samples, int N, int S, float *D,
                                   hipStream_t st) {
    const long long tot = (long long)N * S * vw.V;
    hipLaunchKernelGGL(mh_reproject_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, vw, pts, samples, N, S,
                       D);
    return (int)hipGetLastError();
}
extern "C" int mh_launch_prj_loss(const float *D, const float *ori_patch, const float *conf_patch, const float *vis, int V,
                                  int N, int S, int P, float thr, float *loss, long long *index, uint8_t *hc,
                                  float *all_loss, int sum_block, hipStream_t st) {
    if (S < 1 || S > 8192) return -1;
    const long long cols = (long long)N * S;
    hipLaunchKernelGGL(mh_prj_loss_kernel, dim3(N), dim3(256), (size_t)2 * S * sizeof(float), st, D, ori_patch,
                       conf_patch, vis, V, N, S, P, thr, loss, index, hc, all_loss,
                       sum_block > 0 ? cols - cols % sum_block : cols);
    return (int)hipGetLastError();
}

// forces this translation unit's code object onto the device (HIP loads a fat binary on the first use of one of its kernels:
// 2-20 ms each, which a one-shot pass would pay in the middle of its stages); called from mh_ctx_create
extern "C" int mh_preload_pmvo_pieces() {
    hipFuncAttributes a;
    return (int)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&mh_project_points_kernel));
}
