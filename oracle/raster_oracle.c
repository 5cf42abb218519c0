/* raster_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU statement of the depth-map producer
 * specified in monohair_amd/csrc/raster.hip.
 *
 * What it stands in for: Utils/Render_utils.py:310-347 (render_bust_hair_depth) = moderngl draw of the hair and
 * bust meshes with the BustObj shader (:146-188: gl_Position = projection * transform * v, colour = -z_cam / 2),
 * DEPTH_TEST (:221), clear colour 1.0 (:239-240), vertical flip on read-back (:255-258), saved times 255 (:338).
 * PARITY UNPINNED against the reference here: the reference rasterises with an OpenGL driver (moderngl / EGL are
 * not installed; GL leaves sub-pixel snapping, fill-rule ties and interpolation precision to the implementation),
 * so this file pins the HIP kernel to a written specification and the tests pin the specification to the analytic
 * depth of a sphere.
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
    o->x = o->ok ? (int)rintf(col * 256.0f) : INT_MIN;
    o->y = o->ok ? (int)rintf(row * 256.0f) : INT_MIN;
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
