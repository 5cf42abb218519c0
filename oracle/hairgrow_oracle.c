/*
 * oracle/hairgrow_oracle.c -- TEST INFRASTRUCTURE ONLY (see pmvo_oracle.c).
 *
 * Strand tracing on the fitted orientation/occupancy volume: HairGrowing.trace (/root/reference/HairGrow.py:59-149),
 * HairGrowing.traceFromScalp (:154-223) and the sequential acceptance through the `flag` volume of
 * GenerateGuideStrandFromScalp (:226-265) / randomlyGenerateSegments (:269-299).
 *
 * Volume: vox[z][y][x] = {ori_x, ori_y, ori_z, occ} with the y,z components already negated as
 * HairGrowing.__init__ does (:55).  Positions are voxel coordinates (x,y,z), fp32; voxel index = truncation
 * toward zero, clamped.  torch.dot on 3-vectors = separately rounded products added left to right;
 * torch.linalg.norm = sqrt of an fma chain.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int W, H, Z;
    const float *vox; /* [Z][H][W][4] */
} orc_volume;

static inline int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }
static inline const float *voxel_at(const orc_volume *v, const float *p) {
    int x = clampi((int)p[0], 0, v->W - 1), y = clampi((int)p[1], 0, v->H - 1), z = clampi((int)p[2], 0, v->Z - 1);
    return v->vox + (((size_t)z * v->H + y) * v->W + x) * 4;
}
static inline float dot3(const float *a, const float *b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
static inline float norm3(const float *a) {
    float s = a[0] * a[0];
    s = fmaf(a[1], a[1], s);
    s = fmaf(a[2], a[2], s);
    return sqrtf(s);
}

/*
 * HairGrowing.trace for one (already jittered) seed, without the flag test: forward <= 256 steps along +Tan, then
 * backward <= 256 steps along -Tan from the seed.  out has room for 513 points; the strand occupies
 * out[first .. first+len).  Returns len (>= 1).
 */
static int trace_one(const orc_volume *v, const float *seed, float thr, float *out /*513*3*/, int *first) {
    float pos[3] = {seed[0], seed[1], seed[2]}, tan[3];
    const float *vx = voxel_at(v, pos);
    memcpy(tan, vx, 12);
    int nf = 0, nb = 0;
    memcpy(out + 256 * 3, pos, 12);
    for (int count = 0;;) {
        if (vx[3] == 0.0f) break;
        float nxt[3] = {pos[0] + tan[0], pos[1] + tan[1], pos[2] + tan[2]};
        const float *nv = voxel_at(v, nxt);
        if (dot3(nv, tan) < thr) break;
        memcpy(pos, nxt, 12);
        memcpy(tan, nv, 12);
        vx = nv;
        ++nf;
        memcpy(out + (256 + nf) * 3, pos, 12);
        if (++count >= 256) break;
    }
    memcpy(pos, seed, 12);
    vx = voxel_at(v, pos);
    memcpy(tan, vx, 12);
    for (int count = 0;;) {
        if (vx[3] == 0.0f) break;
        float nxt[3] = {pos[0] - tan[0], pos[1] - tan[1], pos[2] - tan[2]};
        const float *nv = voxel_at(v, nxt);
        if (dot3(nv, tan) < thr) break;
        memcpy(pos, nxt, 12);
        memcpy(tan, nv, 12);
        vx = nv;
        ++nb;
        memcpy(out + (256 - nb) * 3, pos, 12);
        if (++count >= 256) break;
    }
    *first = 256 - nb;
    return nf + nb + 1;
}

void orc_trace_seeds(const orc_volume *v, const float *seeds, int n, float thr, float *out /*n*513*3*/,
                     int32_t *first, int32_t *len) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < n; ++i) {
        int f;
        len[i] = trace_one(v, seeds + 3 * i, thr, out + (size_t)i * 513 * 3, &f);
        first[i] = f;
    }
}

/*
 * HairGrowing.traceFromScalp for one root: out has room for 257 points.  Returns the number of points, or 0 when
 * the reference returns None (the strand never left the "inner" phase).
 */
static int trace_scalp_one(const orc_volume *v, const float *seed, const float *normal, float thr, float *out) {
    const float d[3] = {0.0f, 1.0f, 0.0f};
    float pos[3] = {seed[0], seed[1], seed[2]};
    float lift = dot3(normal, d) + 1.0f;
    if (!(lift < 1.0f)) lift = 1.0f; /* python min(tensor, 1): keeps the first argument when it compares smaller */
    float nrm[3] = {normal[0] + d[0] * lift, normal[1] + d[1] * lift, normal[2] + d[2] * lift};
    float ln = norm3(nrm);
    float tan[3] = {nrm[0] / ln, nrm[1] / ln, nrm[2] / ln};
    const float *vx = voxel_at(v, pos);
    int n = 1, count = 0, inner = 1;
    memcpy(out, pos, 12);
    for (;;) {
        if (vx[3] == 0.0f && !inner) break;
        float nxt[3] = {pos[0] + tan[0], pos[1] + tan[1], pos[2] + tan[2]};
        const float *nv = voxel_at(v, nxt);
        float nt[3] = {nv[0], nv[1], nv[2]};
        if (norm3(nt) < 0.1f && inner) {
            if (dot3(tan, normal) < 0.85f) {
                memcpy(nt, tan, 12);
            } else {
                float t2[3] = {tan[0] + d[0] * lift, tan[1] + d[1] * lift, tan[2] + d[2] * lift};
                float l2 = norm3(t2);
                nt[0] = t2[0] / l2;
                nt[1] = t2[1] / l2;
                nt[2] = t2[2] / l2;
            }
        } else {
            if (dot3(nt, tan) < thr && !inner) {
                float neg[3] = {-nt[0], -nt[1], -nt[2]};
                if (dot3(neg, tan) < thr && !inner) break;
                memcpy(nt, neg, 12);
            }
            if (dot3(nt, tan) < 0.0f && inner) {
                nt[0] = -nt[0];
                nt[1] = -nt[1];
                nt[2] = -nt[2];
            }
            inner = 0;
        }
        memcpy(pos, nxt, 12);
        memcpy(tan, nt, 12);
        vx = voxel_at(v, pos);
        memcpy(out + n * 3, pos, 12);
        ++n;
        ++count;
        if (count >= 256) break;
        if (count >= 25 && inner) break;
    }
    return inner ? 0 : n;
}

void orc_trace_scalp(const orc_volume *v, const float *seeds, const float *normals, int n, float thr,
                     float *out /*n*257*3*/, int32_t *len) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < n; ++i)
        len[i] = trace_scalp_one(v, seeds + 3 * i, normals + 3 * i, thr, out + (size_t)i * 257 * 3);
}

/*
 * The sequential acceptance of GenerateGuideStrandFromScalp / randomlyGenerateSegments: strands are visited in
 * order; a voxel-seeded strand is dropped when `flag` at its seed voxel is already >= 3 (HairGrow.py:72) or when it
 * has fewer than 5 points (:144); an accepted strand adds 1 to `flag` on every distinct voxel it touches
 * (fancy-index `+= 1`, :260 / :292).  mode 0: increment (voxel seeds); mode 1: set to 1 (scalp strands, :247).
 * flag: [Z][H][W] float, in/out.  accepted[i] out.
 */
void orc_accept_strands(int W, int H, int Z, float *flag, const float *pts, const int32_t *first, const int32_t *len,
                        int stride, const float *seeds, int n, int mode, uint8_t *accepted) {
    int32_t *stamp = (int32_t *)malloc(sizeof(int32_t) * (size_t)W * H * Z);
    memset(stamp, 0xff, sizeof(int32_t) * (size_t)W * H * Z);
    for (int i = 0; i < n; ++i) {
        accepted[i] = 0;
        if (mode == 0) {
            const float *s = seeds + 3 * i;
            int x = clampi((int)s[0], 0, W - 1), y = clampi((int)s[1], 0, H - 1), z = clampi((int)s[2], 0, Z - 1);
            if (flag[((size_t)z * H + y) * W + x] >= 3.0f) continue;
            if (len[i] < 5) continue;
        } else if (len[i] <= 0) {
            continue;
        }
        accepted[i] = 1;
        const float *p = pts + ((size_t)i * stride + first[i]) * 3;
        for (int k = 0; k < len[i]; ++k) {
            int x = clampi((int)p[3 * k], 0, W - 1), y = clampi((int)p[3 * k + 1], 0, H - 1),
                z = clampi((int)p[3 * k + 2], 0, Z - 1);
            size_t q = ((size_t)z * H + y) * W + x;
            if (mode == 1) {
                flag[q] = 1.0f;
            } else if (stamp[q] != i) {
                stamp[q] = i;
                flag[q] += 1.0f;
            }
        }
    }
    free(stamp);
}
