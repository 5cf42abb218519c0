/*
 * oracle/pmvo_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A CPU restatement, in plain C, of MonoHair's patch-based multi-view optimisation
 * (PMVO) hot path.  It exists to CHECK the HIP kernels in monohair_amd/csrc and to
 * give bench.py its `cpu_baseline` leg; nothing in the product path may call it.
 *
 * Every function cites the reference lines it restates (/root/reference/...).
 * Floating point is written operation by operation in the order the reference's
 * PyTorch-CPU ops evaluate them (probed in the survey container, torch 2.10 + MKL):
 *   - torch.matmul of a 4x4 / 3x3 matrix with a [K,N] matrix  = k-ordered fmaf chain from 0;
 *   - torch.linalg.(vector_)norm over a short last dim          = sqrt of a k-ordered fmaf chain;
 *   - torch.cosine_similarity(x,y)  = sum_k (x_k/max(|x|,eps)) * (y_k/max(|y|,eps)),
 *     products rounded separately, then added left to right (NO fma);
 *   - torch.sum(t, dim=0) on a contiguous [V,...] float tensor   = ATen "cascade sum":
 *     16-row blocks accumulated sequentially, block sums accumulated sequentially,
 *     remainder rows added, then the levels added in order; a third level takes over the
 *     second every 256 rows (ATen multi_row_sum with level_power 4; implemented for V < 4096);
 *   - torch.round = round-half-even; torch.min propagates NaN (first NaN index wins).
 * Compile with -ffp-contract=off so that the compiler adds no fma of its own.
 *
 * Parity status: PINNED -- checked against golden vectors produced by the imported
 * reference itself (tools/gen_golden.py -> tests/golden/, tests/test_oracle_golden.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_CAM_STRIDE 48 /* floats per camera record */
/* camera record: [0..15] pose (world->camera, row-major), [16..31] proj (row-major),
 * [32..40] inverse of pose[:3,:3] (row-major), rest padding. */

typedef struct {
    int V, H, W;
    const float *cams;           /* V * ORC_CAM_STRIDE */
    const float *const *depth;   /* V pointers, [H,W]   (channel 0 of the reference's [H,W,3]) */
    const float *const *ori;     /* V pointers, [H,W,2] */
    const float *const *conf;    /* V pointers, [H,W]   */
    const float *const *mask;    /* V pointers, [H,W]   (channel 0) */
} orc_views;

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

static inline float mm4_elem(const float *a, const float *x, int single);   /* (defined with the table of sgemm forms below) */

/* ---- Camera.projection (Utils/Camera_utils.py:38-58): camera_v = pose@[X;1]; uv = proj@camera_v; uv[:2]/=z */
/* single: the point is ALONE in its sgemm (a [4,4] x [4,1] product rounds in its own way, see the table of forms below):
 * a base view that owns one point of the batch (sample_next_3d_pos), or a batch of one point (every projection of it) */
static inline void cam_project(const float *cam, const float *X, float *u, float *v, float *z, int single) {
    const float *P = cam, *Q = cam + 16;
    const float x[4] = {X[0], X[1], X[2], 1.0f};
    float c[4], q[2];
    for (int r = 0; r < 4; ++r) c[r] = mm4_elem(P + 4 * r, x, single);
    for (int r = 0; r < 2; ++r) q[r] = mm4_elem(Q + 4 * r, c, single);
    *z = c[2];
    *u = q[0] / c[2];
    *v = q[1] / c[2];
}

/* ndc -> unrounded pixel, x (column) then y (row): PMVO.py:380-382 == Camera_utils.py:67-69 */
static inline void ndc_to_pixel(float u, float v, int H, int W, float *col, float *row) {
    *col = ((-u + 1.0f) / 2.0f) * (float)W;
    *row = ((v + 1.0f) / 2.0f) * (float)H;
}

/* PMVO.project_points (PMVO.py:378-397) for one point: rounded+clamped (row,col), z'=-z/2, oob flag */
static inline void project_point(const float *cam, const float *X, int H, int W, int *row, int *col, float *zp,
                                 int *oob, float *rowf, float *colf, int single) {
    float u, v, z, cf, rf;
    cam_project(cam, X, &u, &v, &z, single);
    ndc_to_pixel(u, v, H, W, &cf, &rf);
    float cr = nearbyintf(cf), rr = nearbyintf(rf);
    /* torch: round -> long; compare on the integers. NaN/inf never occur for points in front of a camera. */
    long long ci = (long long)cr, ri = (long long)rr;
    *oob = (ci > W - 1) || (ci < 0) || (ri > H - 1) || (ri < 0);
    if (ci < 0) ci = 0;
    if (ci > W - 1) ci = W - 1;
    if (ri < 0) ri = 0;
    if (ri > H - 1) ri = H - 1;
    *row = (int)ri;
    *col = (int)ci;
    *zp = -z / 2.0f;
    if (rowf) *rowf = rf;
    if (colf) *colf = cf;
}

static int batch_is_single(long long columns);   /* (with the rule, below) */

void orc_project_points(const float *cam, const float *pts, int N, int H, int W, int32_t *rc, float *zp,
                        uint8_t *oob, float *pixf) {
    const int single = batch_is_single(N);
    for (int n = 0; n < N; ++n) {
        int r, c, o;
        float z, rf, cf;
        project_point(cam, pts + 3 * n, H, W, &r, &c, &z, &o, &rf, &cf, single);
        rc[2 * n] = r;
        rc[2 * n + 1] = c;
        zp[n] = z;
        oob[n] = (uint8_t)o;
        if (pixf) {
            pixf[2 * n] = rf;
            pixf[2 * n + 1] = cf;
        }
    }
}

static inline int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }
static inline float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* PMVO.compute_visible (PMVO.py:525-529): d = z'*255 - depth */
static inline float soft_visible(float depth, float z255) {
    float d = z255 - depth;
    float vis = (d < 0.1f) ? (1.0f - d / 0.1f) : -1.0f;
    return clampf(vis, -1.0f, 1.0f);
}

/*
 * PMVO.Compute_Visible_and_Ori (PMVO.py:346-376) with the gathers of PMVO.py:482-523.
 * Outputs (any may be NULL): vis[V,N], ori[V,N,2], conf[V,N] (clamped 1e-6..1), mask[V,N],
 * ori_patch[V,N,P,2], conf_patch[V,N,P] (clamped), pixf[V,N,2] (unrounded row,col; extra).
 * Patch taps: row offset outer, column offset inner, each tap clamped to the image.
 */
void orc_visible_and_ori(const orc_views *vw, const float *pts, int N, int patch, float *vis, float *ori,
                         float *conf, float *mask, float *ori_patch, float *conf_patch, float *pixf) {
    const int V = vw->V, H = vw->H, W = vw->W, P = patch * patch, hp = patch / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int v = 0; v < V; ++v) {
        for (int n = 0; n < N; ++n) {
            const float *cam = vw->cams + (size_t)v * ORC_CAM_STRIDE;
            int r, c, oob;
            float zp, rf, cf;
            project_point(cam, pts + 3 * n, H, W, &r, &c, &zp, &oob, &rf, &cf, batch_is_single(N));
            size_t vn = (size_t)v * N + n;
            size_t pix = (size_t)r * W + c;
            if (vis) {
                float vb = soft_visible(vw->depth[v][pix], zp * 255.0f);
                if (oob) vb = -1.0f;
                vis[vn] = vb;
            }
            if (ori) {
                ori[2 * vn] = vw->ori[v][2 * pix];
                ori[2 * vn + 1] = vw->ori[v][2 * pix + 1];
            }
            if (conf) conf[vn] = clampf(vw->conf[v][pix], 1e-6f, 1.0f);
            if (mask) mask[vn] = vw->mask[v][pix];
            if (pixf) {
                pixf[2 * vn] = rf;
                pixf[2 * vn + 1] = cf;
            }
            if (ori_patch || conf_patch) {
                int t = 0;
                for (int i = -hp; i <= hp; ++i)
                    for (int j = -hp; j <= hp; ++j, ++t) {
                        size_t q = (size_t)clampi(r + i, 0, H - 1) * W + clampi(c + j, 0, W - 1);
                        if (ori_patch) {
                            ori_patch[(vn * P + t) * 2] = vw->ori[v][2 * q];
                            ori_patch[(vn * P + t) * 2 + 1] = vw->ori[v][2 * q + 1];
                        }
                        if (conf_patch) conf_patch[vn * P + t] = clampf(vw->conf[v][q], 1e-6f, 1.0f);
                    }
            }
        }
    }
}

/*
 * PMVO.Find_max_conf_from_visible_view (PMVO.py:339-343): C' = vis<1 ? conf*max(vis,0) : conf; top-k over views.
 * This is the SIMPLE rule (value descending, view index ascending among equal values) that round 1 shipped and that the
 * HIP library still offers as topk_order = 1; the reference's own order among equal values -- torch.topk's, i.e.
 * libstdc++'s nth_element + sort -- is orc_topk_views in topk_oracle.cpp, which forward() below uses.
 * out_idx/out_val are [k,N].
 */
void orc_topk_views(const float *vis, const float *conf, int V, int N, int k, int32_t *out_idx, float *out_val);
void orc_topk_views_by_index(const float *vis, const float *conf, int V, int N, int k, int32_t *out_idx, float *out_val) {
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n) {
        float cv[1024];
        for (int v = 0; v < V; ++v) {
            float vb = vis[(size_t)v * N + n], c = conf[(size_t)v * N + n];
            cv[v] = (vb < 1.0f) ? c * fmaxf(vb, 0.0f) : c;
        }
        for (int v = 0; v < V; ++v) {
            int rank = 0;
            for (int u = 0; u < V; ++u) rank += (cv[u] > cv[v]) || (cv[u] == cv[v] && u < v);
            if (rank < k) {
                out_idx[(size_t)rank * N + n] = v;
                out_val[(size_t)rank * N + n] = cv[v];
            }
        }
    }
}

/* the 90 depth offsets of PMVO.sample_next_3d_pos (PMVO.py:274-278) are torch.arange values that are
 * not start+i*step in fp32; they are passed in (fixture tests/golden/depth_offsets.npy). */

/*
 * ---- how the reference's matmuls round, by batch composition -------------------------------------------------
 * PMVO.sample_next_3d_pos (PMVO.py:263-335) loops over the cameras and, for camera i, works on the M points of the
 * batch whose base view (at this rank) is i: Camera.projection(points[index]) is two [4,4] x [4,M] sgemms and
 * Camera.reprojection(...) one [3,3] x [3,S*M] sgemm (Camera_utils.py:50-53,103).  Which MKL kernel an sgemm lands in
 * -- and so how its fp32 sums are associated -- depends on the number of columns.  Probed column by column in the
 * container the goldens were generated in (torch 2.10.0 CPU, MKL 2024.2, AVX-512, torch.get_num_threads() == 8;
 * tools/probe_mkl_forms.py repeats the probe and prints the table):
 *   [4,4] x [4,M]:   M == 1   p2 + ((fma(a1,b1, a0*b0)) + p3)      (a gemv-like kernel, separately rounded p2, p3)
 *                    M >= 2   fma(a3,b3, fma(a2,b2, fma(a1,b1, a0*b0)))               (k-ordered chain, any M to 1e6)
 *   [3,3] x [3,C]:   C <= 3   fma(a2,b2, fma(a1,b1, a0*b0))
 *                    4 <= C <= 28444        (a0*b0 + a2*b2) + a1*b1, separately rounded products
 *                    C >= 28445             fma(a2,b2, fma(a1,b1, a0*b0))
 * The upper threshold is MKL's switch to its threaded kernel and moves with the thread count (1 thread: never; 2: 21334;
 * 3: 16001; 4: 14223; 5, 7: 28889; 6, 8: 28445 columns); the other boundaries do not.  With S = 90 samples per point the
 * chain form starts at M = 317 points per (rank, base view) -- at the headline shape (60 x 1080p views, 5000-point chunks)
 * a fifth to a half of a chunk's (rank, point) items sit in such groups.
 * g_rule.mode: 0 = follow the group size as above (default: what the reference computes for THIS batch),
 *              1 = the mid-size forms for every point (M >= 2 projection, separately rounded reprojection),
 *              2 = the chain forms for every point.
 */
static struct { int mode; long long fma_min_cols; } g_rule = {0, 28445};

void orc_set_reproject_rule(int mode, long long fma_min_cols) {
    g_rule.mode = mode;
    if (fma_min_cols > 0) g_rule.fma_min_cols = fma_min_cols;
}

void orc_get_reproject_rule(int *mode, long long *fma_min_cols) {
    *mode = g_rule.mode;
    *fma_min_cols = g_rule.fma_min_cols;
}

/* a [4,4] x [4,columns] product of one column: its own form (unless the rule forces one form for everything) */
static int batch_is_single(long long columns) { return g_rule.mode == 0 && columns == 1; }

/* forms for a point whose (rank, base view) group holds M points, S samples each: bit 0 = single-column projection,
 * bit 1 = chain-form reprojection */
#define ORC_FORM_GEMV 1
#define ORC_FORM_CHAIN 2
static inline int group_forms(int M, int S) {
    if (g_rule.mode == 1) return 0;
    if (g_rule.mode == 2) return ORC_FORM_CHAIN;
    const long long cols = (long long)M * S;
    return (M == 1 ? ORC_FORM_GEMV : 0) | ((cols <= 3 || cols >= g_rule.fma_min_cols) ? ORC_FORM_CHAIN : 0);
}

/* one element of a [4,4] x [4,M] sgemm (row a of the matrix, column x) in the form the column count selects */
static inline float mm4_elem(const float *a, const float *x, int single) {
    if (single) {
        float f = fmaf(a[1], x[1], a[0] * x[0]);
        float p2 = a[2] * x[2];
        float p3 = a[3] * x[3];
        return p2 + (f + p3);
    }
    float s = a[0] * x[0];
    s = fmaf(a[1], x[1], s);
    s = fmaf(a[2], x[2], s);
    return fmaf(a[3], x[3], s);
}
/* one element of a [3,3] x [3,C] sgemm */
static inline float mm3_elem(const float *a, const float *d, int chain) {
    float p0 = a[0] * d[0];
    if (chain) return fmaf(a[2], d[2], fmaf(a[1], d[1], p0));
    float p1 = a[1] * d[1];
    float p2 = a[2] * d[2];
    return (p0 + p2) + p1;
}
/* The two products as whole matrices, the form chosen from the column count by the rule in force (g_rule): what
 * tests/test_oracle_forms.py compares with torch.matmul outputs recorded where the goldens were generated
 * (tests/golden/mkl_forms.npz, tools/probe_mkl_forms.py).  B is [K, C] row-major, out [K, C]. */
void orc_mm4(const float *A, const float *B, long long C, float *out) {
    const int single = (g_rule.mode == 0) && C == 1;
    for (long long c = 0; c < C; ++c) {
        const float x[4] = {B[c], B[C + c], B[2 * C + c], B[3 * C + c]};
        for (int r = 0; r < 4; ++r) out[r * C + c] = mm4_elem(A + 4 * r, x, single);
    }
}
void orc_mm3(const float *A, const float *B, long long C, float *out) {
    const int chain = g_rule.mode == 2 || (g_rule.mode == 0 && (C <= 3 || C >= g_rule.fma_min_cols));
    for (long long c = 0; c < C; ++c) {
        const float d[3] = {B[c], B[C + c], B[2 * C + c]};
        for (int r = 0; r < 3; ++r) out[r * C + c] = mm3_elem(A + 3 * r, d, chain);
    }
}

/* Camera.reprojection(..., to_world=True) (Camera_utils.py:81-106); torch.matmul(inv(R) [column-major LAPACK output],
 * (c - t) [transposed view]) in the form the column count selects (table above) */
static inline void cam_unproject(const float *cam, float u, float v, float z, float *X, int chain) {
    const float *P = cam, *Q = cam + 16, *Ri = cam + 32;
    float c0 = (u - Q[2]) / Q[0] * z;
    float c1 = (v - Q[6]) / Q[5] * z;
    float c2 = z;
    const float d[3] = {c0 - P[3], c1 - P[7], c2 - P[11]};
    for (int r = 0; r < 3; ++r) X[r] = mm3_elem(Ri + 3 * r, d, chain);
}

/*
 * PMVO.sample_next_3d_pos (PMVO.py:263-335) for one point whose base view is `cam`:
 * pixel (unrounded) + 2*(Ori_col, Ori_row) -> ndc -> unproject at z + offset[s].
 * ori_c = centre orientation sample (row comp, col comp) of the base view at the point.
 * (The surface_points assignments at PMVO.py:333-334 write into temporaries: surface_points == points.)
 */
static inline void sample_next_point(const float *cam, const float *X, const float *ori_c, int H, int W,
                                     const float *offs, int S, float *out /* S*3 */, int forms) {
    float u, v, z, col, row;
    cam_project(cam, X, &u, &v, &z, (forms & ORC_FORM_GEMV) != 0);
    ndc_to_pixel(u, v, H, W, &col, &row);
    float nx = col + ori_c[1] * 2.0f;
    float ny = row + ori_c[0] * 2.0f;
    nx = nx / (float)W;
    ny = ny / (float)H;
    nx = nx * 2.0f - 1.0f;
    ny = ny * 2.0f - 1.0f;
    nx = -nx;
    for (int s = 0; s < S; ++s) cam_unproject(cam, nx, ny, z + offs[s], out + 3 * s, forms & ORC_FORM_CHAIN);
}

/* points per base view among base_view[0..N): the M of the table above (indices outside [0,V) own nothing) */
static void group_sizes(const int32_t *base_view, int N, int V, int32_t *cnt /* V */) {
    memset(cnt, 0, sizeof(int32_t) * (size_t)V);
    for (int n = 0; n < N; ++n)
        if (base_view[n] >= 0 && base_view[n] < V) cnt[base_view[n]]++;
}

void orc_group_sizes(const int32_t *base_view, int N, int V, int32_t *cnt) { group_sizes(base_view, N, V, cnt); }

void orc_sample_next(const orc_views *vw, const float *pts, int N, const int32_t *base_view /*N*/,
                     const float *ori /*V,N,2*/, const float *offs, int S, float *out /*N,S,3*/) {
    int32_t *cnt = (int32_t *)malloc(sizeof(int32_t) * (size_t)vw->V);
    group_sizes(base_view, N, vw->V, cnt);
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n) {
        int b = base_view[n];
        sample_next_point(vw->cams + (size_t)b * ORC_CAM_STRIDE, pts + 3 * n, ori + ((size_t)b * N + n) * 2, vw->H,
                          vw->W, offs, S, out + (size_t)n * S * 3, group_forms(cnt[b], S));
    }
    free(cnt);
}

/* unrounded pixel (row, col) of a world point: Camera.projection + Camera.uv2pixel (Camera_utils.py:60-71) */
static inline void pixel_of(const float *cam, const float *X, int H, int W, float *row, float *col, int single) {
    float u, v, z;
    cam_project(cam, X, &u, &v, &z, single);
    ndc_to_pixel(u, v, H, W, col, row);
}

/* PMVO.compute_reproject_ori (PMVO.py:219-241): D[v,n,s] = pix(sample) - pix(point), as (row, col) */
void orc_reproject_ori(const orc_views *vw, const float *pts, const float *samples, int N, int S, float *D) {
    const int V = vw->V;
#pragma omp parallel for collapse(2) schedule(static)
    for (int v = 0; v < V; ++v)
        for (int n = 0; n < N; ++n) {
            const float *cam = vw->cams + (size_t)v * ORC_CAM_STRIDE;
            float r0, c0;
            pixel_of(cam, pts + 3 * n, vw->H, vw->W, &r0, &c0, batch_is_single(N));
            for (int s = 0; s < S; ++s) {
                float r1, c1;
                pixel_of(cam, samples + ((size_t)n * S + s) * 3, vw->H, vw->W, &r1, &c1, batch_is_single((long long)N * S));
                float *d = D + (((size_t)v * N + n) * S + s) * 2;
                d[0] = r1 - r0;
                d[1] = c1 - c0;
            }
        }
}

/* x / max(|x|, 1e-8) for a 2-vector, as torch.cosine_similarity normalises (norm = sqrt of an fmaf chain) */
static inline void unit2(const float *x, float *o) {
    float s = x[0] * x[0];
    s = fmaf(x[1], x[1], s);
    float nrm = fmaxf(sqrtf(s), 1e-8f);
    /* fmaxf drops a NaN operand; torch.clamp_min keeps it.  Make NaN norms stay NaN. */
    if (s != s) nrm = s;
    o[0] = x[0] / nrm;
    o[1] = x[1] / nrm;
}

/* 1 - max(cos(O,D), cos(-O,D)) on pre-normalised vectors: products rounded, added, no fma */
static inline float tap_loss(const float *oh, const float *dh) {
    float p0 = oh[0] * dh[0];
    float p1 = oh[1] * dh[1];
    float cs = p0 + p1;
    return 1.0f - fabsf(cs);
}

/* ATen cascade sum over the leading (view) dimension, one column: see file header. */
typedef struct {
    float a0, a1, a2;
} casc;
static inline void casc_step(casc *c, int v, float x) {
    if (v > 0 && (v & 15) == 0) {          /* a full 16-row block: level 0 -> level 1 */
        c->a1 = c->a1 + c->a0;
        c->a0 = 0.0f;
        if ((v & 0xF0) == 0) {             /* 256 rows: level 1 -> level 2 (ATen: (i & (15 << 4)) == 0) */
            c->a2 = c->a2 + c->a1;
            c->a1 = 0.0f;
        }
    }
    c->a0 = c->a0 + x;
}
static inline float casc_done(const casc *c) { return (c->a0 + c->a1) + c->a2; }   /* V < 4096 */

/*
 * ATen's sum(dim=0) of a contiguous [V, C] float tensor (aten/src/ATen/native/cpu/SumKernel.cpp, vectorized_outer_sum):
 * columns are taken 4 vectors (32 floats: this kernel is dispatched at AVX2 width even on AVX-512 hosts) at a time through multi_row_sum -- the cascade above -- and the
 * trailing C mod 32 columns through row_sum: the V rows dealt round-robin to four partial cascades (rows k, k+4, ...
 * into partial k), the V mod 4 left-over rows added to partial 0, then partial 0 += partial 1, 2, 3.  Which columns of
 * compute_prj_loss's [V, N*S] tensors (PMVO.py:198-204) are "trailing" depends on the batch: the last (N*S) mod 32
 * samples of the LAST point(s).  Probed at 1 and 8 threads, up to 27 M elements: only the global tail takes this form.
 * multi_row_sum1 / row_sum1: both for one column, x[i * stride], i < R, any R.
 */
static float multi_row_sum1(const float *x, size_t stride, long R) {
    int clog = 0;
    while ((1L << clog) < R) ++clog;
    int level_power = clog / 4;
    if (level_power < 4) level_power = 4;
    const long level_step = 1L << level_power, level_mask = level_step - 1;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    long i = 0;
    while (i + level_step <= R) {
        for (long j = 0; j < level_step; ++j, ++i) acc[0] = acc[0] + x[(size_t)i * stride];
        for (int j = 1; j < 4; ++j) {
            acc[j] = acc[j] + acc[j - 1];
            acc[j - 1] = 0.0f;
            if ((i & (level_mask << (j * level_power))) != 0) break;
        }
    }
    for (; i < R; ++i) acc[0] = acc[0] + x[(size_t)i * stride];
    for (int j = 1; j < 4; ++j) acc[0] = acc[0] + acc[j];
    return acc[0];
}
static float row_sum1(const float *x, size_t stride, long R) {
    const long L = R / 4;
    float part[4];
    for (int k = 0; k < 4; ++k) part[k] = multi_row_sum1(x + (size_t)k * stride, stride * 4, L);
    for (long i = L * 4; i < R; ++i) part[0] = part[0] + x[(size_t)i * stride];
    for (int k = 1; k < 4; ++k) part[0] = part[0] + part[k];
    return part[0];
}
float orc_aten_inner_sum(const float *x, int K);   /* consensus_oracle.c: ATen's sum over a contiguous innermost dimension */
/* columns per vectorised block of the outer sum (32 = 4 AVX2 vectors: probed, tools/probe_mkl_forms.py); 0 = no tail */
static int g_sum_block = 32;
void orc_set_sum_block(int cols) { g_sum_block = cols; }
int orc_get_sum_block(void) { return g_sum_block; }
/* first "trailing" column of a C-column outer sum */
static inline long long sum_tail_start(long long C) { return g_sum_block > 0 ? C - C % g_sum_block : C; }
/* torch.sum(x, dim=0) of a contiguous [V, C] tensor as restated above (tests/test_oracle_forms.py pins it to recorded outputs) */
void orc_outer_sum(const float *x, int V, long long C, float *out) {
    const long long t0 = sum_tail_start(C);
    for (long long c = 0; c < C; ++c)
        out[c] = c >= t0 ? row_sum1(x + c, (size_t)C, V) : multi_row_sum1(x + c, (size_t)C, V);
}

/*
 * PMVO.compute_prj_loss (PMVO.py:151-209) for ONE point.
 *   D        [V,S,2] view-strided by dstride floats (projected 2D segment directions)
 *   opatch   [V,P,2], cpatch [V,P] view-strided (clamped confidences), vis [V] view-strided
 * Returns min loss, argmin sample and the high-confidence flag of that sample.
 * scratch: S*(2 casc + 1 int) -- caller provides.
 */
static void prj_loss_point(int V, int S, int P, float thr, const float *D, size_t dstride, const float *opatch,
                           size_t ostride, const float *cpatch, size_t cstride, const float *vis, size_t vstride,
 float *out_loss, int *out_idx, int *out_hc, float *loss_s /*S or NULL*/,
                           int tail_from /* samples >= tail_from are trailing columns of the [V,N*S] sums; S = none */) {
    /* tail_from == -1: the tensor is [V, 1] (a batch of ONE candidate): squeezed, its sum over dim 0 is ATen's sum over a
     * contiguous innermost dimension (orc_aten_inner_sum), not an outer sum */
    const int inner = tail_from < 0;
    if (tail_from < 0) tail_from = 0;
    const int ntail = tail_from < S ? S - tail_from : 0;
    float *tnum = ntail ? (float *)malloc(sizeof(float) * (size_t)V * ntail) : NULL;
    float *tden = ntail ? (float *)malloc(sizeof(float) * (size_t)V * ntail) : NULL;
    casc *num = (casc *)calloc((size_t)S, sizeof(casc));
    casc *den = (casc *)calloc((size_t)S, sizeof(casc));
    int *cnt = (int *)calloc((size_t)S, sizeof(int));
    float *oh = (float *)malloc(sizeof(float) * 2 * (size_t)P);
    for (int v = 0; v < V; ++v) {
        const float *op = opatch + (size_t)v * ostride, *cp = cpatch + (size_t)v * cstride;
        float cmax = cp[0];
        for (int p = 1; p < P; ++p) cmax = (cp[p] > cmax) ? cp[p] : cmax;
        const int hc = cmax > thr;
        for (int p = 0; p < P; ++p) unit2(op + 2 * p, oh + 2 * p);
        const float visible = vis[(size_t)v * vstride];
        for (int s = 0; s < S; ++s) {
            float dh[2];
            unit2(D + (size_t)v * dstride + 2 * s, dh);
            float ml = tap_loss(oh, dh), bc = cp[0];
            for (int p = 1; p < P; ++p) {
                float l = tap_loss(oh + 2 * p, dh);
                int idx = l < ml;
                int upd = hc ? (idx && (cp[p] > thr)) : idx;
                if (upd) {
                    ml = l;
                    bc = cp[p];
                }
            }
            float w = (visible == -1.0f ? 0.0f : 1.0f) * bc;
            casc_step(&num[s], v, ml * w);
            casc_step(&den[s], v, w);
            cnt[s] += (w > 0.0f);
            if (s >= tail_from) {
                tnum[(size_t)v * ntail + (s - tail_from)] = ml * w;
                tden[(size_t)v * ntail + (s - tail_from)] = w;
            }
        }
    }
    int npos = 0;
    float best = 0.f;
    int besti = 0, seen_nan = 0;
    unsigned char *pos = (unsigned char *)malloc((size_t)S);
    float *ls = (float *)malloc(sizeof(float) * (size_t)S);
    for (int s = 0; s < S; ++s) {
        float dn = casc_done(&den[s]), nm = casc_done(&num[s]);
        if (inner) {
            dn = orc_aten_inner_sum(tden, V);
            nm = orc_aten_inner_sum(tnum, V);
        } else if (s >= tail_from) {
            dn = row_sum1(tden + (s - tail_from), (size_t)ntail, V);
            nm = row_sum1(tnum + (s - tail_from), (size_t)ntail, V);
        }
        float ratio = dn / (float)cnt[s];
        pos[s] = ratio > thr;
        npos += pos[s];
        ls[s] = nm / dn;
    }
    const int low = npos < 5;
    for (int s = 0; s < S; ++s) {
        float l = ls[s];
        if (!low && !pos[s]) l = 1.0f;
        if (loss_s) loss_s[s] = l;
        if (s == 0) {
            best = l;
            besti = 0;
            seen_nan = (l != l);
        } else if (!seen_nan) {
            if (l != l) {
                best = l;
                besti = s;
                seen_nan = 1;
            } else if (l < best) {
                best = l;
                besti = s;
            }
        }
    }
    *out_loss = best;
    *out_idx = besti;
    *out_hc = pos[besti];
    free(num);
    free(den);
    free(cnt);
    free(oh);
    free(pos);
    free(ls);
    free(tnum);
    free(tden);
}

/* first sample of point n (of N, S samples each) that lies in the trailing columns of a [V, N*S] sum; S if none */
static inline int tail_from_of(int n, int N, int S) {
    if ((long long)N * S == 1 && g_sum_block > 0) return -1;   /* [V,1]: see prj_loss_point */
    const long long t0 = sum_tail_start((long long)N * S) - (long long)n * S;
    return t0 >= S ? S : (t0 < 0 ? 0 : (int)t0);
}

/* compute_prj_loss over N points from materialised tensors: D[V,N,S,2], ori_patch[V,N,P,2], conf_patch[V,N,P], vis[V,N] */
void orc_prj_loss(int V, int N, int S, int P, float thr, const float *D, const float *ori_patch,
                  const float *conf_patch, const float *vis, float *loss, int32_t *idx, uint8_t *hc,
                  float *loss_ns /* [N,S] or NULL */) {
#pragma omp parallel for schedule(dynamic, 4)
    for (int n = 0; n < N; ++n) {
        float l;
        int i, h;
        prj_loss_point(V, S, P, thr, D + (size_t)n * S * 2, (size_t)N * S * 2, ori_patch + (size_t)n * P * 2,
                       (size_t)N * P * 2, conf_patch + (size_t)n * P, (size_t)N * P, vis + n, (size_t)N, &l, &i, &h,
                       loss_ns ? loss_ns + (size_t)n * S : NULL, tail_from_of(n, N, S));
        loss[n] = l;
        idx[n] = i;
        hc[n] = (uint8_t)h;
    }
}

/*
 * PMVO.forward (PMVO.py:39-78) for N points.
 *   base_idx/base_val: [K,N] ranking of base views (K >= 2*nrank-1); if NULL it is computed with orc_topk_views.
 * Outputs: line_ori[N,3], min_loss[N], high_conf[N]; optional best_sample[N,3], best_rank[N], best_s[N].
 */
void orc_forward(const orc_views *vw, const float *pts, int N, int patch, float thr, const float *offs, int S,
                 int nrank, int rank_step, const int32_t *base_idx_in, const float *base_val_in, float *line_ori,
                 float *min_loss, uint8_t *high_conf, float *best_sample, int32_t *best_rank, int32_t *best_s) {
    const int V = vw->V, P = patch * patch, K = 20;
    float *vis = (float *)malloc(sizeof(float) * (size_t)V * N);
    float *ori = (float *)malloc(sizeof(float) * (size_t)V * N * 2);
    float *conf = (float *)malloc(sizeof(float) * (size_t)V * N);
    float *opatch = (float *)malloc(sizeof(float) * (size_t)V * N * P * 2);
    float *cpatch = (float *)malloc(sizeof(float) * (size_t)V * N * P);
    orc_visible_and_ori(vw, pts, N, patch, vis, ori, conf, NULL, opatch, cpatch, NULL);
    int32_t *bidx = NULL;
    float *bval = NULL;
    if (!base_idx_in) {
        bidx = (int32_t *)malloc(sizeof(int32_t) * (size_t)K * N);
        bval = (float *)malloc(sizeof(float) * (size_t)K * N);
        orc_topk_views(vis, conf, V, N, K, bidx, bval);
        base_idx_in = bidx;
        base_val_in = bval;
    }
    /* points per (rank, base view) of THIS batch: they select the rounding of sample_next_3d_pos's sgemms */
    int32_t *gcnt = (int32_t *)malloc(sizeof(int32_t) * (size_t)nrank * V);
    for (int r = 0; r < nrank; ++r) group_sizes(base_idx_in + (size_t)r * rank_step * N, N, V, gcnt + (size_t)r * V);
#pragma omp parallel for schedule(dynamic, 2)
    for (int n = 0; n < N; ++n) {
        float *samples = (float *)malloc(sizeof(float) * (size_t)S * 3);
        float *D = (float *)malloc(sizeof(float) * (size_t)V * S * 2);
        const float *X = pts + 3 * n;
        float ml = 0.f, bs[3] = {0, 0, 0};
        int hcb = 0, br = 0, bsi = 0;
        for (int r = 0; r < nrank; ++r) {
            const int i = r * rank_step;
            const int b = base_idx_in[(size_t)i * N + n];
            sample_next_point(vw->cams + (size_t)b * ORC_CAM_STRIDE, X, ori + ((size_t)b * N + n) * 2, vw->H, vw->W,
                              offs, S, samples, group_forms(gcnt[(size_t)r * V + b], S));
            for (int v = 0; v < V; ++v) {
                const float *cam = vw->cams + (size_t)v * ORC_CAM_STRIDE;
                float r0, c0;
                pixel_of(cam, X, vw->H, vw->W, &r0, &c0, batch_is_single(N));
                for (int s = 0; s < S; ++s) {
                    float r1, c1;
                    pixel_of(cam, samples + 3 * s, vw->H, vw->W, &r1, &c1, batch_is_single((long long)N * S));
                    D[((size_t)v * S + s) * 2] = r1 - r0;
                    D[((size_t)v * S + s) * 2 + 1] = c1 - c0;
                }
            }
            float l;
            int idx, h;
            prj_loss_point(V, S, P, thr, D, (size_t)S * 2, opatch + (size_t)n * P * 2, (size_t)N * P * 2,
                           cpatch + (size_t)n * P, (size_t)N * P, vis + n, (size_t)N, &l, &idx, &h, NULL,
                           tail_from_of(n, N, S));
            int take = (r == 0) || ((l < ml) && (base_val_in[(size_t)i * N + n] > 0.0f));
            if (take) {
                ml = l;
                hcb = h;
                br = r;
                bsi = idx;
                memcpy(bs, samples + 3 * idx, sizeof(bs));
            }
        }
        float d0 = bs[0] - X[0], d1 = bs[1] - X[1], d2 = bs[2] - X[2];
        float s2 = d0 * d0;
        s2 = fmaf(d1, d1, s2);
        s2 = fmaf(d2, d2, s2);
        float nrm = sqrtf(s2);
        line_ori[3 * n] = d0 / nrm;
        line_ori[3 * n + 1] = d1 / nrm;
        line_ori[3 * n + 2] = d2 / nrm;
        min_loss[n] = ml;
        high_conf[n] = (uint8_t)hcb;
        if (best_sample) memcpy(best_sample + 3 * n, bs, sizeof(bs));
        if (best_rank) best_rank[n] = br;
        if (best_s) best_s[n] = bsi;
        free(samples);
        free(D);
    }
    free(vis);
    free(ori);
    free(conf);
    free(opatch);
    free(cpatch);
    free(bidx);
    free(bval);
    free(gcnt);
}

/*
 * PMVO.refine's loss (PMVO.py:86-90): next = p + ori*0.005/4, compute_reproject_ori + compute_prj_loss with
 * one candidate per point (S = 1: `low_conf_index` is always true, so the raw num/den comes back).
 */
void orc_refine_loss(const orc_views *vw, const float *pts, const float *dir, float mul, float dv, int N, int patch,
                     float thr, float *loss, uint8_t *hc) {
    const int V = vw->V, P = patch * patch;
    float *vis = (float *)malloc(sizeof(float) * (size_t)V * N);
    float *opatch = (float *)malloc(sizeof(float) * (size_t)V * N * P * 2);
    float *cpatch = (float *)malloc(sizeof(float) * (size_t)V * N * P);
    orc_visible_and_ori(vw, pts, N, patch, vis, NULL, NULL, NULL, opatch, cpatch, NULL);
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n) {
        float *D = (float *)malloc(sizeof(float) * (size_t)V * 2);
        const float *X = pts + 3 * n;
        float Q[3];
        for (int k = 0; k < 3; ++k) Q[k] = X[k] + dir[3 * n + k] * mul / dv;
        for (int v = 0; v < V; ++v) {
            const float *cam = vw->cams + (size_t)v * ORC_CAM_STRIDE;
            float r0, c0, r1, c1;
            pixel_of(cam, X, vw->H, vw->W, &r0, &c0, batch_is_single(N));
            pixel_of(cam, Q, vw->H, vw->W, &r1, &c1, batch_is_single(N));
            D[2 * v] = r1 - r0;
            D[2 * v + 1] = c1 - c0;
        }
        float l;
        int idx, h;
        prj_loss_point(V, 1, P, thr, D, 2, opatch + (size_t)n * P * 2, (size_t)N * P * 2, cpatch + (size_t)n * P,
                       (size_t)N * P, vis + n, (size_t)N, &l, &idx, &h, NULL, tail_from_of(n, N, 1));
        loss[n] = l;
        if (hc) hc[n] = (uint8_t)h;
        free(D);
    }
    free(vis);
    free(opatch);
    free(cpatch);
}

/*
 * The per-view votes of PMVO.filter_points (PMVO.py:402-459), PMVO.compute_unvisible_points (:461-480) and the
 * mask vote of PMVO.filter_head_points (:110-137).  Sums over views in ATen's order (cascade; row_sum for the trailing points).
 */
void orc_filter_points(const orc_views *vw, const float *pts, int N, int patch, float thr, float vis_thr,
                       uint8_t *surface_index, uint8_t *filter_index, uint8_t *unvisible_index,
                       uint8_t *head_filter) {
    const int V = vw->V, H = vw->H, W = vw->W, hp = patch / 2;
    const long long tail0 = sum_tail_start(N);   /* the N points are one batch: its trailing N mod 32 columns (row_sum1) */
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n) {
        casc t[8];
        memset(t, 0, sizeof(t));
        float *terms = (n >= tail0) ? (float *)malloc(sizeof(float) * 8 * (size_t)V) : NULL;
        for (int v = 0; v < V; ++v) {
            const float *cam = vw->cams + (size_t)v * ORC_CAM_STRIDE;
            int r, c, oob;
            float zp;
            project_point(cam, pts + 3 * n, H, W, &r, &c, &zp, &oob, NULL, NULL, batch_is_single(N));
            const size_t pix = (size_t)r * W + c;
            float m = vw->mask[v][pix];
            const float gap = zp * 255.0f - vw->depth[v][pix];
            float cmax = vw->conf[v][pix]; /* raw confidences (PMVO.py:415-418) */
            for (int i = -hp; i <= hp; ++i)
                for (int j = -hp; j <= hp; ++j) {
                    float cv = vw->conf[v][(size_t)clampi(r + i, 0, H - 1) * W + clampi(c + j, 0, W - 1)];
                    cmax = cv > cmax ? cv : cmax;
                }
            if (oob) cmax = 0.0f;
            const float unv = (oob || gap > 0.1f) ? 1.0f : 0.0f;
            const float unv1 = (oob || gap > vis_thr) ? 1.0f : 0.0f;
            const float unv9 = (oob || gap > 0.9f) ? 1.0f : 0.0f;
            const float unvh = (gap >= vis_thr) ? 1.0f : 0.0f;
            const float lowc = (cmax < thr) ? 1.0f : 0.0f;
            m = (m > 0.2f) ? 1.0f : m;
            const float term[8] = {(1.0f - unv) * lowc, 1.0f - unv,          (1.0f - unv) * m, 1.0f - unv1,
                                   (1.0f - unv1) * m,   1.0f - unv9,         1.0f - unvh,      (1.0f - unvh) * m};
            for (int k = 0; k < 8; ++k) casc_step(&t[k], v, term[k]);
            if (terms)
                for (int k = 0; k < 8; ++k) terms[(size_t)k * V + v] = term[k];
        }
        float s[8];
        for (int k = 0; k < 8; ++k)      /* (a batch of one point: [V,1] sums, ATen's inner-dimension order) */
            s[k] = !terms ? casc_done(&t[k]) : (N == 1 ? orc_aten_inner_sum(terms + (size_t)k * V, V) : row_sum1(terms + (size_t)k * V, 1, V));
        free(terms);
        const int low_conf = s[0] > 4.0f;
        const int hair = (s[1] - s[2]) < (s[1] * 1.0f / 2.0f);
        const int hair1 = (s[3] - s[4]) < (s[3] * 1.0f / 2.0f);
        const int surf0 = s[1] > 1.0f;
        const int filt0 = (s[3] > 1.0f) && !surf0;
        if (surface_index) surface_index[n] = (uint8_t)(surf0 && !low_conf && hair);
        if (filter_index) filter_index[n] = (uint8_t)(filt0 && !low_conf && hair1);
        if (unvisible_index) unvisible_index[n] = (uint8_t)!(s[5] > 2.0f);
        if (head_filter) head_filter[n] = (uint8_t)!((s[6] - s[7]) < (s[6] * 1.0f / 2.0f));
    }
}
