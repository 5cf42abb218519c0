// mh_device.h -- device-side arithmetic shared by the PMVO kernels (gfx950 only).
//
// Every helper evaluates its formula operation by operation in the order the reference's PyTorch-CPU
// ops do (see oracle/pmvo_oracle.c for the probed rules); this file is compiled with -ffp-contract=off
// so the only fused multiply-adds are the explicit __builtin_fmaf calls below.  HIP's default f32
// division and sqrt are correctly rounded (no -ffast-math here), f32 denormals are kept.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MH_CAM_STRIDE 48
#define MH_TOPK 20
#define MH_WAVE 64

struct MhViews {
    int V, H, W;
    const float4 *rec;   // [V][H][W] {ori_row, ori_col, conf, depth}
    const float *mask;   // [V][H][W]
    const float *cams;   // [V][MH_CAM_STRIDE]
    const float4 *tap;   // [V][H][W] {unit ori_row, unit ori_col, clamped conf, 0}: what a patch tap of the search is, per pixel,
                         // made once at upload by the same mh_unit2 / mh_clampf the front end would apply per iteration
                         // (nullptr: not resident -- views uploaded as 8-bit codes use the code tables instead)
    int batch_rule;      // 1: option reproject_rule 0 -- the projections follow the batch (a batch of ONE point projects
                         // through the single-column form, see mh_cam_project_b); 0: one form for everything
};

__device__ __forceinline__ float mh_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// Camera.projection (Utils/Camera_utils.py:38-58): camera_v = pose@[X;1] and uv = proj@camera_v are
// k-ordered fma chains (MKL sgemm); proj rows 0,1 are [fx,0,cx,0] / [0,fy,cy,0], so the chain collapses
// to fma(cx, z, fx*x) and fma(cy, z, fy*y) exactly (the zero terms add +-0).
__device__ __forceinline__ void mh_cam_project(const float *__restrict__ cam, float X0, float X1, float X2,
                                               float &u, float &v, float &z) {
    float c0 = cam[0] * X0;
    c0 = mh_fma(cam[1], X1, c0);
    c0 = mh_fma(cam[2], X2, c0);
    c0 = mh_fma(cam[3], 1.0f, c0);
    float c1 = cam[4] * X0;
    c1 = mh_fma(cam[5], X1, c1);
    c1 = mh_fma(cam[6], X2, c1);
    c1 = mh_fma(cam[7], 1.0f, c1);
    float c2 = cam[8] * X0;
    c2 = mh_fma(cam[9], X1, c2);
    c2 = mh_fma(cam[10], X2, c2);
    c2 = mh_fma(cam[11], 1.0f, c2);
    float q0 = mh_fma(cam[18], c2, cam[16] * c0);
    float q1 = mh_fma(cam[22], c2, cam[21] * c1);
    z = c2;
    u = q0 / c2;
    v = q1 / c2;
}

// ndc -> unrounded pixel (PMVO.py:380-382, Camera_utils.py:67-69)
__device__ __forceinline__ void mh_ndc_to_pixel(float u, float v, float Hf, float Wf, float &row, float &col) {
    col = ((-u + 1.0f) / 2.0f) * Wf;
    row = ((v + 1.0f) / 2.0f) * Hf;
}

__device__ __forceinline__ void mh_pixel_of(const float *__restrict__ cam, float X0, float X1, float X2, float Hf,
                                            float Wf, float &row, float &col) {
    float u, v, z;
    mh_cam_project(cam, X0, X1, X2, u, v, z);
    mh_ndc_to_pixel(u, v, Hf, Wf, row, col);
}

// ---- the reference's batch composition in the arithmetic (oracle/pmvo_oracle.c, the table above cam_unproject) ----
// PMVO.sample_next_3d_pos (PMVO.py:263-335) works, per camera, on the M points of the batch whose base view (at this rank)
// that camera is; its sgemms round by their column count (MKL 2024.2 / AVX-512 / 8 threads, where the goldens come from):
//   Camera.projection, [4,4] x [4,M]:      M == 1: p2 + ((fma(a1,b1, a0*b0)) + p3);   M >= 2: k-ordered fma chain
//   Camera.reprojection, [3,3] x [3,S*M]:  S*M <= 3 or S*M >= fma_min_cols (28445): k-ordered fma chain;
//                                          else (a0*b0 + a2*b2) + a1*b1 with separately rounded products
// and ATen's sum(dim=0) of a [V, C] tensor adds its trailing C mod 32 columns (32 = 4 AVX2 vectors) in another order
// (mh_row_sum_views).
// MhRule carries what a kernel needs to follow the batch; mode 1 / 2 force the mid / chain forms for every point.
#define MH_FORM_GEMV 1
#define MH_FORM_CHAIN 2
// The counts are kept in MH_GROUP_COPIES partial arrays (a workgroup of the counting kernel adds to copy blockIdx % COPIES):
// up to 670 points of a 5000-point chunk share one (rank, base view), and that many atomics on ONE address drain in ~9 us.
#define MH_GROUP_COPIES 16
#define MH_GROUP_RANKS 16   // = MH_MAX_RANKS of the search
struct MhRule {
    const int32_t *gcnt;    // [MH_GROUP_COPIES][MH_GROUP_RANKS][V] partial counts of the points per (rank, base view) of this
                            // batch (nullptr: every group counts as mid-size)
    long long tail_col0;    // first trailing column of the [V, N*S] sums; N*S if none
    int mode;               // 0 follow the group size, 1 mid forms, 2 chain forms
    int fma_min_cols;       // chain-form reprojection from this many columns on
};

__device__ __forceinline__ int mh_group_forms(const MhRule &rule, int rank, int V, int b, int S) {
    if (rule.mode == 1) return 0;
    if (rule.mode == 2) return MH_FORM_CHAIN;
    int M = 2;
    if (rule.gcnt) {
        M = 0;
#pragma unroll
        for (int k = 0; k < MH_GROUP_COPIES; ++k) M += rule.gcnt[(k * MH_GROUP_RANKS + rank) * V + b];
    }
    const long long cols = (long long)M * S;
    return (M == 1 ? MH_FORM_GEMV : 0) | ((cols <= 3 || cols >= rule.fma_min_cols) ? MH_FORM_CHAIN : 0);
}

// Camera.projection of a point that is alone in its sgemm (M == 1)
__device__ __forceinline__ void mh_cam_project_single(const float *__restrict__ cam, float X0, float X1, float X2,
                                                      float &u, float &v, float &z) {
    float c[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float f = mh_fma(cam[4 * r + 1], X1, cam[4 * r] * X0);
        const float p2 = cam[4 * r + 2] * X2;
        const float p3 = cam[4 * r + 3] * 1.0f;
        c[r] = p2 + (f + p3);
    }
    float q[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const float f = mh_fma(cam[16 + 4 * r + 1], c[1], cam[16 + 4 * r] * c[0]);
        const float p2 = cam[16 + 4 * r + 2] * c[2];
        const float p3 = cam[16 + 4 * r + 3] * c[3];
        q[r] = p2 + (f + p3);
    }
    z = c[2];
    u = q[0] / c[2];
    v = q[1] / c[2];
}

// Camera.projection of a point of a batch: single = the point is alone in its sgemm (a base view that owns one point of the
// batch in sample_next_3d_pos; a batch of ONE point in every projection of it, e.g. the last chunk of optimize / refine when
// N mod 5000 == 1 -- tests/golden/pmvo_single.npz)
__device__ __forceinline__ void mh_cam_project_b(const float *__restrict__ cam, float X0, float X1, float X2, float &u,
                                                 float &v, float &z, bool single) {
    if (single) mh_cam_project_single(cam, X0, X1, X2, u, v, z);
    else mh_cam_project(cam, X0, X1, X2, u, v, z);
}
__device__ __forceinline__ void mh_pixel_of_b(const float *__restrict__ cam, float X0, float X1, float X2, float Hf, float Wf,
                                              float &row, float &col, bool single) {
    float u, v, z;
    mh_cam_project_b(cam, X0, X1, X2, u, v, z, single);
    mh_ndc_to_pixel(u, v, Hf, Wf, row, col);
}

// Camera.reprojection(to_world=True) (Camera_utils.py:81-106) in the form the column count of its sgemm selects
__device__ __forceinline__ void mh_cam_unproject(const float *__restrict__ cam, float u, float v, float z,
                                                 float &X0, float &X1, float &X2, bool chain = false) {
    float c0 = (u - cam[18]) / cam[16] * z;
    float c1 = (v - cam[22]) / cam[21] * z;
    float d0 = c0 - cam[3], d1 = c1 - cam[7], d2 = z - cam[11];
    const float *Ri = cam + 32;
    if (chain) {
        X0 = mh_fma(Ri[2], d2, mh_fma(Ri[1], d1, Ri[0] * d0));
        X1 = mh_fma(Ri[5], d2, mh_fma(Ri[4], d1, Ri[3] * d0));
        X2 = mh_fma(Ri[8], d2, mh_fma(Ri[7], d1, Ri[6] * d0));
    } else {
        X0 = (Ri[0] * d0 + Ri[2] * d2) + Ri[1] * d1;
        X1 = (Ri[3] * d0 + Ri[5] * d2) + Ri[4] * d1;
        X2 = (Ri[6] * d0 + Ri[8] * d2) + Ri[7] * d1;
    }
}

// Where the points of a launch sit in the reference's batches, for the [V, N] sums over views (S = 1): they are rows
// row0 .. row0 + N - 1 of `total` points that the reference processes `batch` at a time (PMVO.py:604-606: 5000); the
// trailing (length mod 32) rows of every batch are summed in row_sum order.  block 0: cascade order everywhere.
struct MhBatch {
    long long row0, total;
    int batch, block;
    int single_ok;   // reproject_rule 0: a batch of ONE point projects through the single-column form
};
// length of the batch point n of the launch sits in
__device__ __forceinline__ long long mh_batch_len(const MhBatch &bt, int n, long long &pos) {
    const long long row = bt.row0 + n;
    const long long start = bt.batch > 0 ? row / bt.batch * bt.batch : 0;
    const long long left = bt.total - start;
    pos = row - start;
    return (bt.batch > 0 && bt.batch < left) ? bt.batch : left;
}
__device__ __forceinline__ bool mh_tail_row(const MhBatch &bt, int n) {
    if (bt.block <= 0) return false;
    long long pos;
    const long long len = mh_batch_len(bt, n, pos);
    return pos >= len - len % bt.block;
}
__device__ __forceinline__ bool mh_batch_single(const MhBatch &bt, int n) {
    long long pos;
    return mh_batch_len(bt, n, pos) == 1;
}

// ATen's sum over a CONTIGUOUS innermost dimension of V elements (vectorized_inner_sum, 8 floats per vector; restated for the
// medoid in consensus.hip / oracle/consensus_oracle.c): what sum(dim=0) of a [V, 1] tensor is once the size-1 dimension is
// squeezed -- the sums over views of a batch of ONE point.  term(v) = element v.  Cold path (a one-point batch), V < 4096.
template <typename F>
__device__ __forceinline__ float mh_inner_sum_views(int V, F term) {
    if (V < 8) {   // scalar_inner_sum: row_sum on single floats
        const int L = V >> 2;
        float p[4] = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < L; ++i)
            for (int k = 0; k < 4; ++k) p[k] = p[k] + term(4 * i + k);
        for (int i = L * 4; i < V; ++i) p[0] = p[0] + term(i);
        return ((p[0] + p[1]) + p[2]) + p[3];
    }
    const int vec = V >> 3, R = vec >> 2;   // full vectors; rows of four vectors
    float a0[32], a1[32], a2[32];
    for (int a = 0; a < 32; ++a) a0[a] = a1[a] = a2[a] = 0.0f;
    for (int r = 0; r < R; ++r) {
        for (int a = 0; a < 32; ++a) a0[a] = a0[a] + term(32 * r + a);
        if (((r + 1) & 15) == 0) {
            for (int a = 0; a < 32; ++a) {
                a1[a] = a1[a] + a0[a];
                a0[a] = 0.0f;
            }
            if (((r + 1) & 0xF0) == 0)
                for (int a = 0; a < 32; ++a) {
                    a2[a] = a2[a] + a1[a];
                    a1[a] = 0.0f;
                }
        }
    }
    for (int a = 0; a < 32; ++a) a0[a] = (a0[a] + a1[a]) + a2[a];
    for (int i = R * 4; i < vec; ++i)
        for (int l = 0; l < 8; ++l) a0[l] = a0[l] + term(8 * i + l);
    for (int k = 1; k < 4; ++k)
        for (int l = 0; l < 8; ++l) a0[l] = a0[l] + a0[8 * k + l];
    float fin = 0.0f;
    for (int j = vec * 8; j < V; ++j) fin = fin + term(j);
    for (int l = 0; l < 8; ++l) fin = fin + a0[l];
    return fin;
}

// ATen's row_sum (aten/src/ATen/native/cpu/SumKernel.cpp) of one trailing column of a [V, C] outer sum: rows k, k+4, ...
// into partial k, each partial a multi_row_sum cascade (16 rows per level-0 block) over its V/4 rows; the V mod 4 left-over
// rows into partial 0; then partial 0 += partial 1, 2, 3.  term(v) = row v of the column.  Exact for V < 4096.
template <typename F>
__device__ __forceinline__ float mh_row_sum_views(int V, F term) {
    const int L = V >> 2;
    float part[4];
    for (int k = 0; k < 4; ++k) {
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
        int i = 0;
        while (i + 16 <= L) {
            for (int j = 0; j < 16; ++j, ++i) a0 = a0 + term(4 * i + k);
            a1 = a1 + a0;
            a0 = 0.0f;
            if ((i & 0xF0) == 0) {
                a2 = a2 + a1;
                a1 = 0.0f;
            }
        }
        for (; i < L; ++i) a0 = a0 + term(4 * i + k);
        part[k] = (a0 + a1) + a2;
    }
    for (int i = L * 4; i < V; ++i) part[0] = part[0] + term(i);
    return ((part[0] + part[1]) + part[2]) + part[3];
}

// x / max(|x|, 1e-8) of a 2-vector as torch.cosine_similarity normalises it (norm = sqrt of an fma chain)
__device__ __forceinline__ void mh_unit2(float x0, float x1, float &o0, float &o1) {
    float s = x0 * x0;
    s = mh_fma(x1, x1, s);
    float nrm = __builtin_sqrtf(s);
    nrm = (nrm < 1e-8f) ? 1e-8f : nrm;   // clamp_min keeps NaN (comparison false)
    o0 = x0 / nrm;
    o1 = x1 / nrm;
}

// ---- two items at a time: the same operations, element by element, on 64-bit register pairs, so that the fma
// chains become v_pk_fma_f32 (two single-rounded fp32 FMAs per issue slot) -- bit-identical to the scalar forms.
typedef float mh_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ mh_v2f mh_fma2(mh_v2f a, mh_v2f b, mh_v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ mh_v2f mh_splat(float x) { return mh_v2f{x, x}; }

// n0/d and n1/d for two items (element-wise on register pairs), bit-identical to IEEE-754 division for operands in the normal range: this is the very
// instruction sequence LLVM expands an f32 fdiv to on gfx9 (v_rcp_f32, two Newton steps on the reciprocal,
// quotient, two residual corrections) minus the v_div_scale / v_div_fixup range handling, which is the
// identity unless an operand or the quotient is (nearly) denormal or the exponents differ by >= 96 -- never
// the case for camera-space depths and pixel distances.  Sharing the refined reciprocal saves ~40 % of the
// instructions of two separate divisions.  (tests: the kernels using it stay bit-exact against the CPU oracle,
// which divides with the host's IEEE division.)
__device__ __forceinline__ void mh_div2x2(mh_v2f n0, mh_v2f n1, mh_v2f d, mh_v2f &q0, mh_v2f &q1) {
    mh_v2f r = mh_v2f{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const mh_v2f nd = -d;
    mh_v2f e = mh_fma2(nd, r, mh_splat(1.0f));
    r = mh_fma2(e, r, r);
    mh_v2f a = n0 * r;
    mh_v2f ea = mh_fma2(nd, a, n0);
    a = mh_fma2(ea, r, a);
    ea = mh_fma2(nd, a, n0);
    q0 = mh_fma2(ea, r, a);
    mh_v2f b = n1 * r;
    mh_v2f eb = mh_fma2(nd, b, n1);
    b = mh_fma2(eb, r, b);
    eb = mh_fma2(nd, b, n1);
    q1 = mh_fma2(eb, r, b);
}

// pixel (row, col) of two world points in one view: mh_pixel_of with the two divisions by the common denominator
// sharing one refined reciprocal (mh_div2x2)
__device__ __forceinline__ void mh_pixel_of_fast2(const float *__restrict__ cam, mh_v2f X0, mh_v2f X1, mh_v2f X2,
                                                  float Hf, float Wf, mh_v2f &row, mh_v2f &col) {
    mh_v2f c0 = mh_splat(cam[0]) * X0;
    c0 = mh_fma2(mh_splat(cam[1]), X1, c0);
    c0 = mh_fma2(mh_splat(cam[2]), X2, c0);
    c0 = mh_fma2(mh_splat(cam[3]), mh_splat(1.0f), c0);
    mh_v2f c1 = mh_splat(cam[4]) * X0;
    c1 = mh_fma2(mh_splat(cam[5]), X1, c1);
    c1 = mh_fma2(mh_splat(cam[6]), X2, c1);
    c1 = mh_fma2(mh_splat(cam[7]), mh_splat(1.0f), c1);
    mh_v2f c2 = mh_splat(cam[8]) * X0;
    c2 = mh_fma2(mh_splat(cam[9]), X1, c2);
    c2 = mh_fma2(mh_splat(cam[10]), X2, c2);
    c2 = mh_fma2(mh_splat(cam[11]), mh_splat(1.0f), c2);
    const mh_v2f q0 = mh_fma2(mh_splat(cam[18]), c2, mh_splat(cam[16]) * c0);
    const mh_v2f q1 = mh_fma2(mh_splat(cam[22]), c2, mh_splat(cam[21]) * c1);
    mh_v2f u, v;
    mh_div2x2(q0, q1, c2, u, v);
    // ((1 - u) / 2) * W: halving is exact, so one rounding of (1 - u) * W / 2 either way -- multiply by W/2 directly
    col = (-u + mh_splat(1.0f)) * mh_splat(Wf * 0.5f);
    row = (v + mh_splat(1.0f)) * mh_splat(Hf * 0.5f);
}

// sqrt(x) for a value that is clamped from below at 1e-8 right afterwards: the correctly rounded result for
// x >= 2^-96, something below 1e-8 (which the clamp replaces, exactly as it replaces the true root) for smaller x.
// This is the compiler's own IEEE expansion of an f32 sqrt on gfx9 -- v_sqrt_f32 (1 ulp), the two neighbours, two
// fma residuals, two selects -- minus its input pre-scaling for x < 2^-96 and its class fix-up, neither of which
// can change the clamped value: 9 instructions instead of 16.  NaN and +inf pass through as in the full expansion.
__device__ __forceinline__ float mh_sqrt_before_clamp(float x) {
    float r = __builtin_amdgcn_sqrtf(x);
    const float rd = __uint_as_float(__float_as_uint(r) - 1u), ru = __uint_as_float(__float_as_uint(r) + 1u);
    const float ed = mh_fma(-rd, r, x), eu = mh_fma(-ru, r, x);
    r = (0.0f >= ed) ? rd : r;
    r = (0.0f < eu) ? ru : r;
    return r;
}

__device__ __forceinline__ void mh_unit2_fast2(mh_v2f x0, mh_v2f x1, mh_v2f &o0, mh_v2f &o1) {
    mh_v2f s = x0 * x0;
    s = mh_fma2(x1, x1, s);
    mh_v2f nrm = mh_v2f{mh_sqrt_before_clamp(s.x), mh_sqrt_before_clamp(s.y)};
    nrm.x = (nrm.x < 1e-8f) ? 1e-8f : nrm.x;
    nrm.y = (nrm.y < 1e-8f) ? 1e-8f : nrm.y;
    mh_div2x2(x0, x1, nrm, o0, o1);
}

__device__ __forceinline__ float mh_clampf(float x, float lo, float hi) {
    return x < lo ? lo : (x > hi ? hi : x);
}

// PMVO.compute_visible (PMVO.py:525-529)
__device__ __forceinline__ float mh_soft_visible(float depth, float z255) {
    float d = z255 - depth;
    float vis = (d < 0.1f) ? (1.0f - d / 0.1f) : -1.0f;
    return mh_clampf(vis, -1.0f, 1.0f);
}

// ATen's cascade sum over the leading (view) dimension of a [V, ...] tensor (multi_row_sum, level_power 4): 16 rows
// into level 0, level 0 into level 1 after every full block, level 1 into level 2 every 256 rows; the remainder
// rows stay in level 0 and the levels are added in order at the end.  Exact for V < 4096.
struct MhCascV {
    float a0, a1, a2;
};
// call when v > 0 and v % 16 == 0, before adding row v
__device__ __forceinline__ void mh_cascv_flush(MhCascV &c, int v) {
    c.a1 = c.a1 + c.a0;
    c.a0 = 0.0f;
    if ((v & 0xF0) == 0) {
        c.a2 = c.a2 + c.a1;
        c.a1 = 0.0f;
    }
}
__device__ __forceinline__ float mh_cascv_done(const MhCascV &c) { return (c.a0 + c.a1) + c.a2; }

// two-level form for sums of fewer than 256 rows (the 180 orientations of the Gabor bank)
struct MhCasc {
    float a0, a1;
};
__device__ __forceinline__ void mh_casc_flush(MhCasc &c) {
    c.a1 = c.a1 + c.a0;
    c.a0 = 0.0f;
}

// Launch order of the search (pmvo_search.hip): the work class of a point, 0 = heaviest.  work = (taps of the views that see
// the point) x (item slices its usable base-view ranks need); computed by mh_search_work_kernel, or by the base-view ranking
// kernel of the fused forward, which has the point's ranking in its lanes anyway (MhWorkArgs.cls != nullptr).
#define MH_ORDER_BUCKETS 1024
struct MhWorkArgs {
    const uint8_t *cnt;   // [V,N] tap-list lengths
    int32_t *cls;         // [N] out: work class (nullptr: not wanted)
    int32_t *gcnt;        // [MH_GROUP_COPIES][MH_GROUP_RANKS][V] out, zeroed by the front end: partial counts of the points per
                          // (rank, base view) (MhRule; nullptr: not wanted)
    int P1, nrank, rank_step, S, T;
    int tail_n0;          // points >= tail_n0 hold trailing columns of the batch's sums: class 0 (mh_search3_kernel evaluates
                          // those once more in its epilogue -- such a workgroup should start first, not last)
};
__device__ __forceinline__ int mh_work_class(int nt, int nvalid, int V, int P1, int S, int T) {
    const int maxwork = V * (P1 - 1) * 4;   // taps of all views x 4 slices
    const int work = nt * ((nvalid * S + T - 1) / T);
    const int b = (int)(((long long)work * (MH_ORDER_BUCKETS - 1)) / (maxwork > 0 ? maxwork : 1));
    return MH_ORDER_BUCKETS - 1 - min(max(b, 0), MH_ORDER_BUCKETS - 1);
}
