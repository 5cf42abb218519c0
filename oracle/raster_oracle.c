/* raster_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU statement of the depth-map producer
 * specified in monohair_amd/csrc/raster.hip.
 *
 * What it stands in for: Utils/Render_utils.py:310-347 (render_bust_hair_depth) = moderngl draw of the hair and
 * bust meshes with the BustObj shader (:146-188: gl_Position = projection * transform * v, colour = -z_cam / 2),
 * DEPTH_TEST (:221), clear colour 1.0 (:239-240), vertical flip on read-back (:255-258), saved times 255 (:338).
 * The reference rasterises with an OpenGL driver; GL leaves sub-pixel snapping, fill-rule ties and interpolation
 * precision to the implementation, so this file pins the HIP kernel to a written specification, and the tests pin the
 * specification (a) to the analytic depth of a sphere and (b) to images drawn by a real OpenGL implementation (Google
 * SwiftShader: tests/golden/gl_raster.npz, tools/gen_golden_gl.py, tests/gl_checks.py) up to those parts.  Parity
 * with the reference's own driver remains unpinned.
 *
 * Specification: vertices through Camera.projection (Utils/Camera_utils.py:38-58) and PMVO's ndc->pixel map
 * (PMVO.py:380-382), snapped to 1/256 pixel; coverage by exact integer edge functions at the pixel centre with a
 * top-left rule; window-space linear z for the LESS depth test, ties to the earlier primitive; perspective-correct
 * -z_cam = 1 / sum(lambda_i / w_i); value (-z_cam / 2) * 255, background 255. */
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

typedef struct {
    int x, y, ok;
    float zw, iw;
} rvert;

/* sub-pixel grid the window positions are snapped to: 2^-bits pixel, 4 <= bits <= 8 (8 = the shipped 1/256; OpenGL asks
 * for at least 4, which is what Google SwiftShader uses -- the comparison with it, tests/gl_checks.py) */
static float g_snap = 256.0f;
static int g_snap_mul = 1;
void ora_set_subpixel_bits(int bits) {
    if (bits < 4) bits = 4;
    if (bits > 8) bits = 8;
    g_snap = (float)(1 << bits);
    g_snap_mul = 256 >> bits;
}

static void project_vertex(const float *cam, const float *X, int H, int W, rvert *o) {
    const float *P = cam, *Q = cam + 16;
    float c[4], q[4];
    for (int r = 0; r < 4; ++r) {
        float a = P[r * 4 + 0] * X[0];
        a = fmaf(P[r * 4 + 1], X[1], a);
        a = fmaf(P[r * 4 + 2], X[2], a);
        a = fmaf(P[r * 4 + 3], 1.0f, a);
        c[r] = a;
    }
    for (int r = 0; r < 3; ++r) {
        float a = Q[r * 4 + 0] * c[0];
        a = fmaf(Q[r * 4 + 1], c[1], a);
        a = fmaf(Q[r * 4 + 2], c[2], a);
        a = fmaf(Q[r * 4 + 3], c[3], a);
        q[r] = a;
    }
    const float z = c[2], w = -z;
    const float u = q[0] / z, v = q[1] / z;
    const float col = ((-u + 1.0f) / 2.0f) * (float)W;
    const float row = ((v + 1.0f) / 2.0f) * (float)H;
    o->zw = (q[2] / w) * 0.5f + 0.5f;
    o->iw = 1.0f / w;
    o->ok = (w > 0.0f) && (fabsf(col) < 1.0e5f) && (fabsf(row) < 1.0e5f);
    o->x = o->ok ? (int)rintf(col * g_snap) * g_snap_mul : INT_MIN;     /* 1/256 pixel units, on a grid of 2^-bits pixel */
    o->y = o->ok ? (int)rintf(row * g_snap) * g_snap_mul : INT_MIN;
}

static int64_t edge_fn(const rvert *s, const rvert *t, int px, int py) {
    return (int64_t)(t->x - s->x) * (int64_t)(py - s->y) - (int64_t)(t->y - s->y) * (int64_t)(px - s->x);
}
static int owns(const rvert *s, const rvert *t) {
    const int dx = t->x - s->x, dy = t->y - s->y;
    return dy < 0 || (dy == 0 && dx > 0);
}
static int ceil_div256(int a) { return (int)ceil((double)a / 256.0); }
static int floor_div256(int a) { return (int)floor((double)a / 256.0); }

/* out[H,W,channels]; returns the number of covered pixels, -1 on allocation failure */
long ora_render_depth(const float *cam, const float *verts, int Nv, const int32_t *faces, int Nf, int H, int W,
                      float pixel_center, float *out, int channels) {
    const int off = (int)(pixel_center * 256.0f + 0.5f);
    rvert *vt = (rvert *)malloc(sizeof(rvert) * (size_t)(Nv > 0 ? Nv : 1));
    float *zb = (float *)malloc(sizeof(float) * (size_t)H * W);
    if (!vt || !zb) {
        free(vt);
        free(zb);
        return -1;
    }
    for (int i = 0; i < Nv; ++i) project_vertex(cam, verts + 3 * i, H, W, &vt[i]);
    for (size_t i = 0; i < (size_t)H * W; ++i) {
        zb[i] = INFINITY;
        for (int k = 0; k < channels; ++k) out[i * channels + k] = 255.0f;
    }
    long covered = 0;
    for (int f = 0; f < Nf; ++f) {   /* draw order: a later primitive must be strictly nearer to replace */
        const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
        if (i0 < 0 || i0 >= Nv || i1 < 0 || i1 >= Nv || i2 < 0 || i2 >= Nv) continue;
        const rvert *a = &vt[i0], *b = &vt[i1], *c = &vt[i2];
        if (!a->ok || !b->ok || !c->ok) continue;
        int64_t area = edge_fn(a, b, c->x, c->y);
        if (area == 0) continue;
        if (area < 0) {
            const rvert *s = b;
            b = c;
            c = s;
            area = -area;
        }
        int minx = a->x, maxx = a->x, miny = a->y, maxy = a->y;
        if (b->x < minx) minx = b->x;
        if (c->x < minx) minx = c->x;
        if (b->x > maxx) maxx = b->x;
        if (c->x > maxx) maxx = c->x;
        if (b->y < miny) miny = b->y;
        if (c->y < miny) miny = c->y;
        if (b->y > maxy) maxy = b->y;
        if (c->y > maxy) maxy = c->y;
        int c0 = ceil_div256(minx - off), c1 = floor_div256(maxx - off);
        int r0 = ceil_div256(miny - off), r1 = floor_div256(maxy - off);
        if (c0 < 0) c0 = 0;
        if (r0 < 0) r0 = 0;
        if (c1 > W - 1) c1 = W - 1;
        if (r1 > H - 1) r1 = H - 1;
        const float fa = (float)area;
        for (int r = r0; r <= r1; ++r)
            for (int cc = c0; cc <= c1; ++cc) {
                const int px = cc * 256 + off, py = r * 256 + off;
                const int64_t e0 = edge_fn(b, c, px, py), e1 = edge_fn(c, a, px, py), e2 = edge_fn(a, b, px, py);
                if (e0 < 0 || e1 < 0 || e2 < 0) continue;
                if ((e0 == 0 && !owns(b, c)) || (e1 == 0 && !owns(c, a)) || (e2 == 0 && !owns(a, b))) continue;
                const float l0 = (float)e0 / fa, l1 = (float)e1 / fa, l2 = (float)e2 / fa;
                const float zw = (l0 * a->zw + l1 * b->zw) + l2 * c->zw;
                if (!(zw >= 0.0f && zw <= 1.0f)) continue;
                const size_t i = (size_t)r * W + cc;
                if (!(zw < zb[i])) continue;
                if (zb[i] == INFINITY) ++covered;
                zb[i] = zw;
                const float s = (l0 * a->iw + l1 * b->iw) + l2 * c->iw;
                const float val = ((1.0f / s) / 2.0f) * 255.0f;
                for (int k = 0; k < channels; ++k) out[i * channels + k] = val;
            }
    }
    free(vt);
    free(zb);
    return covered;
}

/* ---------------------------------------------------------------------------------------------------------------
 * Strand-segment renderer: CPU statement of the second half of monohair_amd/csrc/raster.hip (mh_render_strands).
 * Stands in for Utils/Render_utils.py:269-307 (render_data) = StrandsObj (:8-127, GL_LINES of width 3, colour options
 * 0..3) drawn over BustObj (:130-203, depth options 0..2) with DEPTH_TEST.  Pinned against a real OpenGL
 * implementation (Google SwiftShader: tests/golden/gl_raster.npz, tools/gen_golden_gl.py) up to what GL leaves to the
 * implementation; the specification is in the header of that section of raster.hip:
 * GL's diamond-exit rule for the fragments of a segment (line_rule 1: the end pixel too, as SwiftShader draws), `width`
 * fragments stacked in the minor direction around the nearest pixel,
 * window z linear in t (LESS, ties to the earlier primitive: mesh, then segments in order), perspective-correct
 * attributes, colours evaluated algebraically.  Every float operation below is written in the kernel's order. */
typedef struct {
    int x, y, ok;
    float zw, iw, tx, ty, depth;
} lvert;

static void cam_uvz(const float *cam, const float *X, float *u, float *v, float *z, float *zc) {
    const float *P = cam, *Q = cam + 16;
    float c[4], q[4];
    for (int r = 0; r < 4; ++r) {
        float a = P[r * 4 + 0] * X[0];
        a = fmaf(P[r * 4 + 1], X[1], a);
        a = fmaf(P[r * 4 + 2], X[2], a);
        a = fmaf(P[r * 4 + 3], 1.0f, a);
        c[r] = a;
    }
    q[0] = fmaf(Q[2], c[2], Q[0] * c[0]);       /* proj rows 0,1 are [fx,0,cx,0] / [0,fy,cy,0]: mh_cam_project */
    q[1] = fmaf(Q[6], c[2], Q[5] * c[1]);
    *z = c[2];
    *u = q[0] / c[2];
    *v = q[1] / c[2];
    *zc = fmaf(Q[11], 1.0f, Q[10] * c[2]);
}

static void project_line_vertex(const float *cam, const float *X, const float *T, int H, int W, lvert *o) {
    float u, v, z, zc;
    cam_uvz(cam, X, &u, &v, &z, &zc);
    const float col = ((-u + 1.0f) / 2.0f) * (float)W;
    const float row = ((v + 1.0f) / 2.0f) * (float)H;
    const float w = -z;
    o->zw = (zc / w) * 0.5f + 0.5f;
    o->iw = 1.0f / w;
    o->ok = (w > 0.0f) && (fabsf(col) < 1.0e5f) && (fabsf(row) < 1.0e5f);
    o->x = o->ok ? (int)rintf(col * g_snap) * g_snap_mul : INT_MIN;     /* 1/256 pixel units, on a grid of 2^-bits pixel */
    o->y = o->ok ? (int)rintf(row * g_snap) * g_snap_mul : INT_MIN;
    float s = T[0] * T[0];
    s = fmaf(T[1], T[1], s);
    s = fmaf(T[2], T[2], s);
    const float nrm = sqrtf(s);
    float Y[3];
    for (int k = 0; k < 3; ++k) Y[k] = X[k] + (nrm > 0.0f ? T[k] / nrm : 0.0f) * 0.01f;
    float u2, v2, z2, zc2;
    cam_uvz(cam, Y, &u2, &v2, &z2, &zc2);
    o->tx = u - u2;
    o->ty = v - v2;
    o->depth = w;
}

static int floor_div_i(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }
/* index of the sample nearest to v (1/256 units, samples at i*256): exact halves to the lower / the upper index */
static int half_down_i(int v) { return -floor_div_i(128 - v, 256); }
static int half_up_i(int v) { return floor_div_i(v + 128, 256); }
static long long floor_div_ll(long long a, long long b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }   /* b > 0 */

/* Does major index i of a segment produce a fragment, and at which minor pixel (*jc)?  (A, ma) = p_a, (B, mb) = p_b in
 * (major, minor) coordinates, 1/256 pixel; xmaj: major = column.  OpenGL 4.6 14.5.1, literally: a fragment for every
 * pixel whose diamond |dx| + |dy| < 1/2 the segment intersects, except the one whose diamond contains p_b (rule 0), with
 * the specification's tie-break -- "shift" the segment by (-e, -e^2) in window coordinates, which in this image's
 * coordinates (column right, row DOWN) is (-e in column, +e^2 in row):
 *   - a sample line is crossed on [lo, hi) of the columns, on (lo, hi] of the rows;
 *   - a coordinate exactly between two pixels belongs to the left column, to the lower row (larger index); where the
 *     segment crosses a COLUMN's sample line exactly between two rows, the row it is heading to decides (the column shift
 *     dominates), a horizontal segment goes to the lower row;
 *   - a point exactly on a diamond's boundary is inside iff it is on the right half (dx > 0).
 * For a segment no steeper than 45 degrees in (major, minor), |dM| + |dm| to a pixel centre is smallest where the segment
 * crosses the column's sample line or, if it does not reach it, at the end point: so a crossed column has exactly one
 * fragment (the nearest pixel) and the column of an end point that stops short has one iff the end point is inside the
 * diamond.  Exact integer arithmetic.  rule 1: the pixel that holds p_b is kept (what Google SwiftShader draws). */
static int in_diamond(int dcol, int drow) {
    const int s = abs(dcol) + abs(drow);
    return s < 128 || (s == 128 && dcol > 0);
}
static int seg_fragment(int xmaj, int A, int B, int ma, int mb, int i, int off, int rule, int *jc) {
    const int m = i * 256 + off;
    const int lo = A < B ? A : B, hi = A < B ? B : A;
    const int crossed = xmaj ? (m >= lo && m < hi) : (m > lo && m <= hi);
    if (crossed) {
        /* minor coordinate on the sample line, exactly: ma + (mb - ma) (m - A) / (B - A); nearest pixel */
        long long N = (long long)(ma - off) * (B - A) + (long long)(mb - ma) * (m - A), D = (long long)256 * (B - A);
        if (D < 0) N = -N, D = -D;
        const int up = xmaj ? ((long long)(mb - ma) * (B - A) >= 0) : 0;     /* tie: see above */
        *jc = up ? (int)floor_div_ll(2 * N + D, 2 * D) : (int)-floor_div_ll(D - 2 * N, 2 * D);
    } else {
        const int at_a = (m < lo || (m == lo)) == (A < B);   /* the end point on this side of the sample line */
        const int eM = at_a ? A : B, em = at_a ? ma : mb;
        const int ie = xmaj ? half_down_i(eM - off) : half_up_i(eM - off);
        if (i != ie) return 0;
        *jc = xmaj ? half_up_i(em - off) : half_down_i(em - off);
        const int dM = eM - m, dm = em - (*jc * 256 + off);
        if (!in_diamond(xmaj ? dM : dm, xmaj ? dm : dM)) return 0;
    }
    if (rule == 0) {
        const int ib = xmaj ? half_down_i(B - off) : half_up_i(B - off);
        const int jb = xmaj ? half_up_i(mb - off) : half_down_i(mb - off);
        const int dM = B - (ib * 256 + off), dm = mb - (jb * 256 + off);
        if (i == ib && *jc == jb && in_diamond(xmaj ? dM : dm, xmaj ? dm : dM)) return 0;
    }
    return 1;
}

/* out[H,W,3]; prim[H,W] (may be NULL) receives the winning primitive per pixel (-1 background, < Nf mesh triangle,
 * else Nf + segment).  Returns the number of pixels owned by strand fragments, -1 on allocation failure. */
long ora_render_strands(const float *cam, const float *verts, int Nv, const int32_t *faces, int Nf, const float *lpts,
                        const float *ltan, int Ns, int H, int W, float pixel_center, int width, int line_rule,
                        int color_option, int depth_option, float clear, float *out, int32_t *prim_out) {
    const int off = (int)(pixel_center * 256.0f + 0.5f);
    const int mesh = (Nv > 0 && Nf > 0);
    const int nf = mesh ? Nf : 0;
    rvert *vt = (rvert *)malloc(sizeof(rvert) * (size_t)(Nv > 0 ? Nv : 1));
    lvert *lv = (lvert *)malloc(sizeof(lvert) * (size_t)(Ns > 0 ? 2 * Ns : 1));
    float *zb = (float *)malloc(sizeof(float) * (size_t)H * W);
    int32_t *pr = (int32_t *)malloc(sizeof(int32_t) * (size_t)H * W);
    if (!vt || !lv || !zb || !pr) {
        free(vt); free(lv); free(zb); free(pr);
        return -1;
    }
    for (size_t i = 0; i < (size_t)H * W; ++i) {
        zb[i] = INFINITY;
        pr[i] = -1;
    }
    if (mesh) {
        for (int i = 0; i < Nv; ++i) project_vertex(cam, verts + 3 * i, H, W, &vt[i]);
        for (int f = 0; f < Nf; ++f) {
            const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
            if (i0 < 0 || i0 >= Nv || i1 < 0 || i1 >= Nv || i2 < 0 || i2 >= Nv) continue;
            const rvert *a = &vt[i0], *b = &vt[i1], *c = &vt[i2];
            if (!a->ok || !b->ok || !c->ok) continue;
            int64_t area = edge_fn(a, b, c->x, c->y);
            if (area == 0) continue;
            if (area < 0) {
                const rvert *s = b;
                b = c;
                c = s;
                area = -area;
            }
            int minx = a->x, maxx = a->x, miny = a->y, maxy = a->y;
            if (b->x < minx) minx = b->x;
            if (c->x < minx) minx = c->x;
            if (b->x > maxx) maxx = b->x;
            if (c->x > maxx) maxx = c->x;
            if (b->y < miny) miny = b->y;
            if (c->y < miny) miny = c->y;
            if (b->y > maxy) maxy = b->y;
            if (c->y > maxy) maxy = c->y;
            int c0 = ceil_div256(minx - off), c1 = floor_div256(maxx - off);
            int r0 = ceil_div256(miny - off), r1 = floor_div256(maxy - off);
            if (c0 < 0) c0 = 0;
            if (r0 < 0) r0 = 0;
            if (c1 > W - 1) c1 = W - 1;
            if (r1 > H - 1) r1 = H - 1;
            const float fa = (float)area;
            for (int r = r0; r <= r1; ++r)
                for (int cc = c0; cc <= c1; ++cc) {
                    const int px = cc * 256 + off, py = r * 256 + off;
                    const int64_t e0 = edge_fn(b, c, px, py), e1 = edge_fn(c, a, px, py), e2 = edge_fn(a, b, px, py);
                    if (e0 < 0 || e1 < 0 || e2 < 0) continue;
                    if ((e0 == 0 && !owns(b, c)) || (e1 == 0 && !owns(c, a)) || (e2 == 0 && !owns(a, b))) continue;
                    const float l0 = (float)e0 / fa, l1 = (float)e1 / fa, l2 = (float)e2 / fa;
                    const float zw = (l0 * a->zw + l1 * b->zw) + l2 * c->zw;
                    if (!(zw >= 0.0f && zw <= 1.0f)) continue;
                    const size_t i = (size_t)r * W + cc;
                    if (!(zw < zb[i])) continue;
                    zb[i] = zw;
                    pr[i] = f;
                    float g = 0.0f;
                    if (depth_option == 0) {
                        const float s = (l0 * a->iw + l1 * b->iw) + l2 * c->iw;
                        g = (1.0f / s) / 2.0f;
                    } else if (depth_option == 2) {
                        g = 1.0f;
                    }
                    out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = g;
                }
        }
    }
    if (Ns > 0 && color_option >= 0) {
        for (int i = 0; i < 2 * Ns; ++i) project_line_vertex(cam, lpts + 3 * i, ltan + 3 * i, H, W, &lv[i]);
        for (int s = 0; s < Ns; ++s) {
            const lvert *a = &lv[2 * s], *b = &lv[2 * s + 1];
            if (!a->ok || !b->ok) continue;
            const int dx = b->x - a->x, dy = b->y - a->y;
            const int xmaj = abs(dx) >= abs(dy);
            const int A = xmaj ? a->x : a->y, B = xmaj ? b->x : b->y;
            /* Wide lines (GL 4.6 14.5.2.2): the segment is offset by (width-1)/2 pixels in the minor direction towards
             * smaller WINDOW coordinates, rasterised as a line of width 1, and every fragment becomes a column of `width`
             * fragments going up from there.  Window y grows upwards and the rows here grow downwards, so for an x-major
             * line the offset is +(width-1)/2 rows and the column goes towards smaller rows; for a y-major line it is
             * -(width-1)/2 columns and the column goes right.  For odd widths the offset is whole pixels -- the symmetric
             * stack around the thin line; for even widths the half-pixel offset changes which pixel the diamond rule
             * picks (pinned against Mesa: tests/golden/gl_mesa.npz, widths 1-3). */
            const int wshift = (width - 1) * 128;
            const int ma = (xmaj ? a->y + wshift : a->x - wshift), mb = (xmaj ? b->y + wshift : b->x - wshift);
            if (A == B) continue;
            const int lo = A < B ? A : B, hi = A < B ? B : A;
            /* OpenGL's diamond-exit rule (GL 4.6 14.5.1) with the specification's tie-breaking shift: see seg_fragment */
            int i0 = half_down_i(lo - off), i1 = floor_div_i(hi - off + 128, 256);
            const int nmaj = xmaj ? W : H, nmin = xmaj ? H : W;
            if (i0 < 0) i0 = 0;
            if (i1 > nmaj - 1) i1 = nmaj - 1;
            for (int i = i0; i <= i1; ++i) {
                int jc;
                if (!seg_fragment(xmaj, A, B, ma, mb, i, off, line_rule, &jc)) continue;
                /* GL 4.6 14.5.1: t = (p_r - p_a) . (p_b - p_a) / |p_b - p_a|^2 with p_r the centre of the fragment -- the
                 * foot of the perpendicular from the pixel centre, not clamped to the segment */
                const int m = i * 256 + off, mn = jc * 256 + off;
                const float t = (float)((long long)(m - A) * (B - A) + (long long)(mn - ma) * (mb - ma)) /
                                (float)((long long)(B - A) * (B - A) + (long long)(mb - ma) * (mb - ma));
                const float zw = a->zw + t * (b->zw - a->zw);
                if (!(zw >= 0.0f && zw <= 1.0f)) continue;
                const int j0 = xmaj ? jc - (width - 1) : jc;       /* the column: rows jc-(width-1) .. jc / columns jc .. jc+width-1 */
                const float wa = (1.0f - t) * a->iw, wb = t * b->iw;
                const float den = wa + wb;
                const float depth = (wa * a->depth + wb * b->depth) / den;
                const float tx = (wa * a->tx + wb * b->tx) / den;
                const float ty = (wa * a->ty + wb * b->ty) / den;
                float c0, c1, c2;
                if (color_option == 0) {
                    c0 = c1 = c2 = depth / 2.0f;
                } else if (color_option == 1) {
                    const float rr = sqrtf(tx * tx + ty * ty);
                    const float cs = rr > 0.0f ? tx / rr : 1.0f, sn = rr > 0.0f ? ty / rr : 0.0f;
                    c0 = (cs + 1.0f) * 0.5f;
                    c1 = (sn + 1.0f) * 0.5f;
                    c2 = 0.0f;
                } else if (color_option == 2) {
                    const float xx = tx * tx, yy = ty * ty, s2 = xx + yy;
                    const float cs = s2 > 0.0f ? (xx - yy) / s2 : 1.0f, sn = s2 > 0.0f ? (2.0f * tx * ty) / s2 : 0.0f;
                    c0 = (cs + 1.0f) * 0.5f;
                    c1 = (sn + 1.0f) * 0.5f;
                    c2 = 0.0f;
                } else {
                    c0 = c1 = c2 = 1.0f;
                }
                for (int k = 0; k < width; ++k) {
                    const int j = j0 + k;
                    if (j < 0 || j >= nmin) continue;
                    const size_t pix = xmaj ? ((size_t)j * W + i) : ((size_t)i * W + j);
                    if (!(zw < zb[pix])) continue;       /* LESS; an equal z keeps the earlier primitive */
                    zb[pix] = zw;
                    pr[pix] = nf + s;
                    out[3 * pix] = c0;
                    out[3 * pix + 1] = c1;
                    out[3 * pix + 2] = c2;
                }
            }
        }
    }
    long owned = 0;
    for (size_t i = 0; i < (size_t)H * W; ++i) {
        if (pr[i] < 0) out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = clear;
        if (pr[i] >= nf) ++owned;
        if (prim_out) prim_out[i] = pr[i];
    }
    free(vt); free(lv); free(zb); free(pr);
    return owned;
}
