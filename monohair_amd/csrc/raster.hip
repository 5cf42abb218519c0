// raster.hip -- depth-map producer for PMVO, gfx950 only.
//
// Replaces the moderngl/EGL pass of Utils/Render_utils.py:310-347 (render_bust_hair_depth) with its BustObj
// shader (:146-188): triangles of the hair + bust meshes are drawn with a depth test, the colour written is
// depth/2 with depth = -z_camera, the clear colour is 1.0, the image is flipped to a top-left origin and saved
// times 255.  OpenGL leaves sub-pixel snapping and attribute interpolation precision to the implementation, so
// this is a specified rasteriser of its own (oracle/raster_oracle.c restates it).  It is pinned against a real OpenGL
// implementation -- Google SwiftShader, tests/golden/gl_raster.npz, tools/gen_golden_gl.py: on SwiftShader's sub-pixel
// grid (option "raster_subpixel_bits" 4) the images have identical coverage and depth within 2e-3 of 255:
//   * vertex: (u, v, z) = Camera.projection (mh_cam_project), pixel = PMVO's own ndc->pixel map, so a depth
//     map is sampled exactly where PMVO.project_points will look it up; snapped to 1/256 pixel;
//   * coverage: exact int64 edge functions at the pixel centre (+centre offset), top-left fill rule -> every
//     pixel of a shared edge belongs to exactly one triangle, independent of draw order;
//   * depth test: window z (screen-space linear, as GL), LESS, ties -> lowest primitive index (GL draw order),
//     as one 64-bit atomicMin of (z bits << 32 | primitive);
//   * colour: perspective-correct -z_camera = 1 / sum(lambda_i / w_i).
#include "mh_device.h"

struct MhRVert {
    int x, y;      // window position in 1/256 pixel (x = INT_MIN: vertex unusable)
    float zw, iw;  // window depth in [0,1], 1 / w_clip
};

#define MH_R_SUB 256
#define MH_R_BAD INT_MIN

__global__ __launch_bounds__(256) void mh_raster_vertex_kernel(const float *__restrict__ cam,
                                                               const float *__restrict__ verts, int Nv, float Hf,
                                                               float Wf, int snap, MhRVert *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Nv) return;
    float u, v, z, rowf, colf;
    mh_cam_project(cam, verts[3 * i], verts[3 * i + 1], verts[3 * i + 2], u, v, z);
    mh_ndc_to_pixel(u, v, Hf, Wf, rowf, colf);
    MhRVert r;
    const float w = -z;
    const float zc = mh_fma(cam[27], 1.0f, cam[26] * z);
    r.zw = (zc / w) * 0.5f + 0.5f;
    r.iw = 1.0f / w;
    const bool ok = (w > 0.0f) && (__builtin_fabsf(colf) < 1.0e5f) && (__builtin_fabsf(rowf) < 1.0e5f);
    // 1/256 pixel units on a grid of 1/snap pixel (snap = 256 shipped; 16 = the 4 sub-pixel bits of SwiftShader)
    r.x = ok ? (int)__builtin_rintf(colf * (float)snap) * (MH_R_SUB / snap) : MH_R_BAD;
    r.y = ok ? (int)__builtin_rintf(rowf * (float)snap) * (MH_R_SUB / snap) : MH_R_BAD;
    out[i] = r;
}

__device__ __forceinline__ long long mh_edge(int sx, int sy, int tx, int ty, int px, int py) {
    return (long long)(tx - sx) * (long long)(py - sy) - (long long)(ty - sy) * (long long)(px - sx);
}
// fill rule for pixels exactly on the edge s->t of a positively oriented triangle
__device__ __forceinline__ bool mh_owns_edge(int sx, int sy, int tx, int ty) {
    const int dx = tx - sx, dy = ty - sy;
    return (dy < 0) || (dy == 0 && dx > 0);
}
__device__ __forceinline__ int mh_floor_div(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

struct MhRTri {
    int ax, ay, bx, by, cx, cy;
    long long area;
    float za, zb, zc, ia, ib, ic;
    int c0, c1, r0, r1;
};

// load + orient + bound one triangle; false if it cannot produce fragments
__device__ __forceinline__ bool mh_setup_tri(const MhRVert *__restrict__ vt, const int32_t *__restrict__ faces,
                                             int f, int Nv, int H, int W, int off, MhRTri &t) {
    const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
    if ((unsigned)i0 >= (unsigned)Nv || (unsigned)i1 >= (unsigned)Nv || (unsigned)i2 >= (unsigned)Nv) return false;
    MhRVert a = vt[i0], b = vt[i1], c = vt[i2];
    if (a.x == MH_R_BAD || b.x == MH_R_BAD || c.x == MH_R_BAD) return false;
    long long area = mh_edge(a.x, a.y, b.x, b.y, c.x, c.y);
    if (area == 0) return false;
    if (area < 0) {
        MhRVert s = b;
        b = c;
        c = s;
        area = -area;
    }
    t.ax = a.x, t.ay = a.y, t.bx = b.x, t.by = b.y, t.cx = c.x, t.cy = c.y;
    t.area = area;
    t.za = a.zw, t.zb = b.zw, t.zc = c.zw, t.ia = a.iw, t.ib = b.iw, t.ic = c.iw;
    const int minx = min(a.x, min(b.x, c.x)), maxx = max(a.x, max(b.x, c.x));
    const int miny = min(a.y, min(b.y, c.y)), maxy = max(a.y, max(b.y, c.y));
    t.c0 = max(-mh_floor_div(-(minx - off), MH_R_SUB), 0);   // ceil
    t.c1 = min(mh_floor_div(maxx - off, MH_R_SUB), W - 1);
    t.r0 = max(-mh_floor_div(-(miny - off), MH_R_SUB), 0);
    t.r1 = min(mh_floor_div(maxy - off, MH_R_SUB), H - 1);
    return t.c0 <= t.c1 && t.r0 <= t.r1;
}

// coverage + barycentrics of pixel (r, c); false if outside
__device__ __forceinline__ bool mh_cover(const MhRTri &t, int r, int c, int off, float &l0, float &l1, float &l2) {
    const int px = c * MH_R_SUB + off, py = r * MH_R_SUB + off;
    const long long e0 = mh_edge(t.bx, t.by, t.cx, t.cy, px, py);
    const long long e1 = mh_edge(t.cx, t.cy, t.ax, t.ay, px, py);
    const long long e2 = mh_edge(t.ax, t.ay, t.bx, t.by, px, py);
    if (e0 < 0 || e1 < 0 || e2 < 0) return false;
    if (e0 == 0 && !mh_owns_edge(t.bx, t.by, t.cx, t.cy)) return false;
    if (e1 == 0 && !mh_owns_edge(t.cx, t.cy, t.ax, t.ay)) return false;
    if (e2 == 0 && !mh_owns_edge(t.ax, t.ay, t.bx, t.by)) return false;
    const float fa = (float)t.area;
    l0 = (float)e0 / fa;
    l1 = (float)e1 / fa;
    l2 = (float)e2 / fa;
    return true;
}

__device__ __forceinline__ void mh_shade(const MhRTri &t, int f, int r, int c, int W, int off,
                                         unsigned long long *__restrict__ zbuf) {
    float l0, l1, l2;
    if (!mh_cover(t, r, c, off, l0, l1, l2)) return;
    const float zw = (l0 * t.za + l1 * t.zb) + l2 * t.zc;
    if (!(zw >= 0.0f && zw <= 1.0f)) return;
    const unsigned long long key = ((unsigned long long)__float_as_uint(zw) << 32) | (unsigned)f;
    atomicMin(&zbuf[(size_t)r * W + c], key);
}

// Pass 1, one lane per triangle: meshes of this pipeline are mostly triangles of a few pixels, which one lane
// finishes by itself (<= MH_R_SMALL box pixels); bigger ones are queued for the wave-per-triangle pass.
#define MH_R_SMALL 24
__global__ __launch_bounds__(256) void mh_raster_small_kernel(const MhRVert *__restrict__ vt,
                                                              const int32_t *__restrict__ faces, int Nf, int Nv,
                                                              int H, int W, int off,
                                                              unsigned long long *__restrict__ zbuf,
                                                              int32_t *__restrict__ queue,
                                                              unsigned int *__restrict__ qcount) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= Nf) return;
    MhRTri t;
    if (!mh_setup_tri(vt, faces, f, Nv, H, W, off, t)) return;
    const int bw = t.c1 - t.c0 + 1, n = bw * (t.r1 - t.r0 + 1);
    if (n > MH_R_SMALL) {
        queue[atomicAdd(qcount, 1u)] = f;
        return;
    }
    for (int r = t.r0; r <= t.r1; ++r)
        for (int c = t.c0; c <= t.c1; ++c) mh_shade(t, f, r, c, W, off, zbuf);
}

// Pass 2, one wave per queued triangle (persistent waves): the lanes sweep the bounding box, 64 pixels per step.
__global__ __launch_bounds__(256) void mh_raster_large_kernel(const MhRVert *__restrict__ vt,
                                                              const int32_t *__restrict__ faces, int Nv, int H,
                                                              int W, int off,
                                                              unsigned long long *__restrict__ zbuf,
                                                              const int32_t *__restrict__ queue,
                                                              const unsigned int *__restrict__ qcount) {
    const int lane = threadIdx.x & 63, nq = (int)*qcount;
    for (int q = blockIdx.x * 4 + (threadIdx.x >> 6); q < nq; q += gridDim.x * 4) {
        const int f = queue[q];
        MhRTri t;
        mh_setup_tri(vt, faces, f, Nv, H, W, off, t);
        const int bw = t.c1 - t.c0 + 1, n = bw * (t.r1 - t.r0 + 1);
        for (int k = lane; k < n; k += MH_WAVE) mh_shade(t, f, t.r0 + k / bw, t.c0 + k % bw, W, off, zbuf);
    }
}

__global__ __launch_bounds__(256) void mh_raster_resolve_kernel(const MhRVert *__restrict__ vt,
                                                                const int32_t *__restrict__ faces, int Nv, int H,
                                                                int W, int off,
                                                                const unsigned long long *__restrict__ zbuf,
                                                                float *__restrict__ out, int channels) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)H * W) return;
    const unsigned long long key = zbuf[i];
    float val = 255.0f;   // clear colour 1.0, saved times 255 (Render_utils.py:239-251,338)
    if (key != ~0ull) {
        const int r = (int)(i / W), c = (int)(i % W);
        MhRTri t;
        float l0, l1, l2;
        mh_setup_tri(vt, faces, (int)(key & 0xffffffffu), Nv, H, W, off, t);
        mh_cover(t, r, c, off, l0, l1, l2);
        const float s = (l0 * t.ia + l1 * t.ib) + l2 * t.ic;
        const float depth = 1.0f / s;          // perspective-correct -z_camera
        val = (depth / 2.0f) * 255.0f;         // shader: depth / depth_range; saved as depth * 255
    }
    for (int k = 0; k < channels; ++k) out[i * channels + k] = val;
}

extern "C" int mh_launch_render_depth(const float *cam, const float *verts, int Nv, const int32_t *faces, int Nf,
                                      int H, int W, int off, int snap, MhRVert *vt, unsigned long long *zbuf, int32_t *queue,
                                      unsigned int *qcount, float *out, int channels, hipStream_t st) {
    hipError_t e = hipMemsetAsync(zbuf, 0xff, (size_t)H * W * sizeof(unsigned long long), st);
    if (e != hipSuccess) return (int)e;
    if (Nv > 0 && Nf > 0) {
        e = hipMemsetAsync(qcount, 0, sizeof(unsigned int), st);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(mh_raster_vertex_kernel, dim3((Nv + 255) / 256), dim3(256), 0, st, cam, verts, Nv, (float)H,
                           (float)W, snap, vt);
        hipLaunchKernelGGL(mh_raster_small_kernel, dim3((Nf + 255) / 256), dim3(256), 0, st, vt, faces, Nf, Nv, H, W,
                           off, zbuf, queue, qcount);
        const int blocks = (Nf + 3) / 4 < 4096 ? (Nf + 3) / 4 : 4096;
        hipLaunchKernelGGL(mh_raster_large_kernel, dim3(blocks), dim3(256), 0, st, vt, faces, Nv, H, W, off, zbuf,
                           queue, qcount);
    }
    hipLaunchKernelGGL(mh_raster_resolve_kernel, dim3((unsigned)(((size_t)H * W + 255) / 256)), dim3(256), 0, st, vt,
                       faces, Nv, H, W, off, zbuf, out, channels);
    return (int)hipGetLastError();
}

// =============================================================================================
// Strand-segment renderer: the moderngl pass of Utils/Render_utils.py:269-307 (render_data) with
// the StrandsObj line shader (:8-127) drawn over the BustObj mesh (:130-203) -- what infer_inner.py:60-73 feeds to
// DeepMVSHair.  As for the depth maps, OpenGL leaves line rasterisation details to the driver, so this is a
// SPECIFIED rasteriser (oracle/raster_oracle.c restates it in C; pinned against SwiftShader like the depth maps):
//   * line vertices: window position / window z / 1/w exactly as the mesh vertices above; per vertex the shader's
//     `Tangent_2d` = ndc(p + normalize(T) * 0.01) - ndc(p) in GL's clip convention (ndc_gl = -(u, v) of
//     Camera.projection) and `depth` = -z_camera (:57-68);
//   * a segment is x-major if |dx| >= |dy| (1/256 pixel units), else y-major; its fragments follow OpenGL's
//     diamond-exit rule (GL 4.6 14.5.1, mh_seg_fragment below): one fragment in every column whose sample line the
//     segment crosses (the pixel whose sample is nearest), one in the column of an end point that lies inside its pixel's
//     diamond, none for the pixel whose diamond holds the END point; the interpolation parameter of a fragment is GL's
//     t = (p_r - p_a).(p_b - p_a) / |p_b - p_a|^2 with p_r the fragment's centre (mh_seg_t);
//     wide lines (ctx.line_width = 3, :30) follow GL's rule: offset by (width-1)/2 in the minor direction, rasterise thin,
//     replicate each fragment `width` times (mh_setup_seg) -- pinned against a desktop GL, tests/golden/gl_mesa.npz.
//     Option "line_rule" 1 keeps the end pixel (every diamond touched): what Google SwiftShader draws -- with it and
//     "raster_subpixel_bits" 4 this rasteriser draws exactly SwiftShader's line pixels (tests/golden/gl_raster.npz,
//     tools/gen_golden_gl.py);
//   * window z linear in t, depth test LESS against the mesh and the other segments (same 64-bit key buffer; ties to
//     the earlier primitive: mesh before strands, segments in buffer order);
//   * attributes perspective-correct in t: a = ((1-t) a0/w0 + t a1/w1) / ((1-t)/w0 + t/w1);
//   * fragment colours (:80-104): 0 depth/2 grey; 1 ((cos th, sin th, 0) + (1,1,0))/2 with th = atan(T2d.y, T2d.x);
//     2 the same with 2 th; 3 white -- evaluated algebraically (cos th = x/r, cos 2th = (x^2-y^2)/r^2, ...) so that
//     the kernel and its C statement agree to the bit; mesh fragments (:171-183): 0 depth/2, 1 black, 2 white.
// Output: float32 [H,W,3] in the shader's range (the reference multiplies by 255 when it writes the PNGs).
// =============================================================================================
struct MhRLVert {
    int x, y;
    float zw, iw;
    float tx, ty, depth;
    int pad;
};

__global__ __launch_bounds__(256) void mh_raster_linevert_kernel(const float *__restrict__ cam,
                                                                 const float *__restrict__ pts,
                                                                 const float *__restrict__ tans, int Nlv, float Hf,
                                                                 float Wf, int snap, MhRLVert *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Nlv) return;
    const float p0 = pts[3 * i], p1 = pts[3 * i + 1], p2 = pts[3 * i + 2];
    const float t0 = tans[3 * i], t1 = tans[3 * i + 1], t2 = tans[3 * i + 2];
    float u, v, z, rowf, colf;
    mh_cam_project(cam, p0, p1, p2, u, v, z);
    mh_ndc_to_pixel(u, v, Hf, Wf, rowf, colf);
    MhRLVert r;
    const float w = -z;
    const float zc = mh_fma(cam[27], 1.0f, cam[26] * z);
    r.zw = (zc / w) * 0.5f + 0.5f;
    r.iw = 1.0f / w;
    const bool ok = (w > 0.0f) && (__builtin_fabsf(colf) < 1.0e5f) && (__builtin_fabsf(rowf) < 1.0e5f);
    // 1/256 pixel units on a grid of 1/snap pixel (snap = 256 shipped; 16 = the 4 sub-pixel bits of SwiftShader)
    r.x = ok ? (int)__builtin_rintf(colf * (float)snap) * (MH_R_SUB / snap) : MH_R_BAD;
    r.y = ok ? (int)__builtin_rintf(rowf * (float)snap) * (MH_R_SUB / snap) : MH_R_BAD;
    float s = t0 * t0;
    s = mh_fma(t1, t1, s);
    s = mh_fma(t2, t2, s);
    const float nrm = __builtin_sqrtf(s);
    const float n0 = nrm > 0.0f ? t0 / nrm : 0.0f, n1 = nrm > 0.0f ? t1 / nrm : 0.0f, n2 = nrm > 0.0f ? t2 / nrm : 0.0f;
    float u2, v2, z2;
    mh_cam_project(cam, p0 + n0 * 0.01f, p1 + n1 * 0.01f, p2 + n2 * 0.01f, u2, v2, z2);
    r.tx = u - u2;   // ndc_gl = -(u, v)
    r.ty = v - v2;
    r.depth = w;
    r.pad = 0;
    out[i] = r;
}

struct MhRSeg {
    int A, B, ma, mb;   // major / minor coordinates of the two ends (1/256 pixel): a = start (p_a), b = end (p_b)
    int xmaj;
    int i0, i1;         // pixel columns along the major axis that can hold a fragment
};

// Wide lines (GL 4.6 14.5.2.2): the segment is offset by (width-1)/2 pixels in the minor direction towards smaller WINDOW
// coordinates, rasterised as a line of width 1, and every fragment becomes a column of `width` fragments going up from there.
// Window y grows upwards and the rows here grow downwards: an x-major segment moves +(width-1)/2 rows and its column is
// rows jc-(width-1) .. jc; a y-major one moves -(width-1)/2 columns and its column is jc .. jc+width-1.  Odd widths: a whole-
// pixel offset, i.e. the symmetric stack around the thin line; even widths: the half-pixel offset changes which pixel the
// diamond rule picks.  Pinned against Mesa llvmpipe (tests/golden/gl_mesa.npz): width 3 -- the reference's -- within 3
// pixels of ~3 800 per view, width 1 within 1, width 2 within 2.5 % (Mesa draws even widths as a rectangle).
__device__ __forceinline__ bool mh_setup_seg(const MhRLVert &a, const MhRLVert &b, int H, int W, int off, int width,
                                             MhRSeg &g) {
    if (a.x == MH_R_BAD || b.x == MH_R_BAD) return false;
    const int dx = b.x - a.x, dy = b.y - a.y;
    const int wshift = (width - 1) * (MH_R_SUB / 2);
    g.xmaj = (abs(dx) >= abs(dy)) ? 1 : 0;
    g.A = g.xmaj ? a.x : a.y;
    g.B = g.xmaj ? b.x : b.y;
    g.ma = g.xmaj ? a.y + wshift : a.x - wshift;
    g.mb = g.xmaj ? b.y + wshift : b.x - wshift;
    if (g.A == g.B) return false;
    const int lo = min(g.A, g.B), hi = max(g.A, g.B);
    g.i0 = max(-mh_floor_div(MH_R_SUB / 2 - (lo - off), MH_R_SUB), 0);                     // the column that holds the lower end
    g.i1 = min(mh_floor_div(hi - off + MH_R_SUB / 2, MH_R_SUB), (g.xmaj ? W : H) - 1);     // ... the upper end
    return g.i0 <= g.i1;
}
// interpolation parameter of the fragment at (major index i, minor index jc): GL 4.6 14.5.1,
// t = (p_r - p_a) . (p_b - p_a) / |p_b - p_a|^2 with p_r the CENTRE of the fragment -- the foot of the perpendicular from
// the pixel centre, not clamped to the segment (window z and the perspective-correct attributes both use it)
__device__ __forceinline__ float mh_seg_t(const MhRSeg &g, int i, int jc, int off) {
    const int m = i * MH_R_SUB + off, mn = jc * MH_R_SUB + off;
    return (float)((long long)(m - g.A) * (g.B - g.A) + (long long)(mn - g.ma) * (g.mb - g.ma)) /
           (float)((long long)(g.B - g.A) * (g.B - g.A) + (long long)(g.mb - g.ma) * (g.mb - g.ma));
}
// index of the sample nearest to v (1/256 units, samples at i*256): exact halves to the lower / the upper index
__device__ __forceinline__ int mh_half_down(int v) { return -mh_floor_div(MH_R_SUB / 2 - v, MH_R_SUB); }
__device__ __forceinline__ int mh_half_up(int v) { return mh_floor_div(v + MH_R_SUB / 2, MH_R_SUB); }
__device__ __forceinline__ long long mh_floor_div_ll(long long a, long long b) {   // b > 0
    return (a >= 0) ? a / b : -((-a + b - 1) / b);
}
__device__ __forceinline__ bool mh_in_diamond(int dcol, int drow) {
    const int s = abs(dcol) + abs(drow);
    return s < MH_R_SUB / 2 || (s == MH_R_SUB / 2 && dcol > 0);
}
// Does major index i of the segment produce a fragment, and at which minor pixel (jc)?  OpenGL 4.6 14.5.1, literally: a
// fragment for every pixel whose diamond |dx| + |dy| < 1/2 the segment intersects, except the one whose diamond contains
// the END point p_b (rule 0), with the specification's tie-break -- "shift" the segment by (-e, -e^2) in window
// coordinates, which in this image's coordinates (column right, row DOWN) is (-e in column, +e^2 in row):
//   * a sample line is crossed on [lo, hi) of the columns, on (lo, hi] of the rows;
//   * a coordinate exactly between two pixels belongs to the left column, to the lower row (larger index); where the
//     segment crosses a COLUMN's sample line exactly between two rows, the row it is heading to decides (the column
//     shift dominates), a horizontal segment goes to the lower row;
//   * a point exactly on a diamond's boundary is inside iff it is on the right half (dx > 0).
// For a segment no steeper than 45 degrees in (major, minor), |dM| + |dm| to a pixel centre is smallest where the segment
// crosses the column's sample line or, if it does not reach it, at the end point: a crossed column has exactly one
// fragment (the nearest pixel) and the column of an end point that stops short has one iff the end point is inside the
// diamond.  Exact integer arithmetic.  rule 1: the pixel that holds p_b is kept -- what Google SwiftShader draws; with it
// and 4 sub-pixel bits this rasteriser reproduces SwiftShader's lines pixel for pixel (tests/golden/gl_raster.npz).
__device__ __forceinline__ bool mh_seg_fragment(const MhRSeg &g, int i, int off, int rule, int &jc) {
    const int m = i * MH_R_SUB + off;
    const int A = g.A, B = g.B, ma = g.ma, mb = g.mb;
    const int lo = min(A, B), hi = max(A, B);
    const bool xmaj = g.xmaj != 0;
    const bool crossed = xmaj ? (m >= lo && m < hi) : (m > lo && m <= hi);
    if (crossed) {
        // minor coordinate on the sample line, exactly: ma + (mb - ma) (m - A) / (B - A); nearest pixel
        long long N = (long long)(ma - off) * (B - A) + (long long)(mb - ma) * (m - A), D = (long long)MH_R_SUB * (B - A);
        if (D < 0) N = -N, D = -D;
        const bool up = xmaj && ((long long)(mb - ma) * (B - A) >= 0);
        jc = up ? (int)mh_floor_div_ll(2 * N + D, 2 * D) : (int)-mh_floor_div_ll(D - 2 * N, 2 * D);
    } else {
        const bool at_a = (m <= lo) == (A < B);   // the end point on this side of the sample line
        const int eM = at_a ? A : B, em = at_a ? ma : mb;
        const int ie = xmaj ? mh_half_down(eM - off) : mh_half_up(eM - off);
        if (i != ie) return false;
        jc = xmaj ? mh_half_up(em - off) : mh_half_down(em - off);
        const int dM = eM - m, dm = em - (jc * MH_R_SUB + off);
        if (!mh_in_diamond(xmaj ? dM : dm, xmaj ? dm : dM)) return false;
    }
    if (rule == 0) {
        const int ib = xmaj ? mh_half_down(B - off) : mh_half_up(B - off);
        const int jb = xmaj ? mh_half_up(mb - off) : mh_half_down(mb - off);
        const int dM = B - (ib * MH_R_SUB + off), dm = mb - (jb * MH_R_SUB + off);
        if (i == ib && jc == jb && mh_in_diamond(xmaj ? dM : dm, xmaj ? dm : dM)) return false;
    }
    return true;
}

__global__ __launch_bounds__(256) void mh_raster_lines_kernel(const MhRLVert *__restrict__ lv, int Ns, int H, int W,
                                                              int off, int width, int rule, unsigned prim_base,
                                                              unsigned long long *__restrict__ zbuf) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= Ns) return;
    const MhRLVert a = lv[2 * s], b = lv[2 * s + 1];
    MhRSeg g;
    if (!mh_setup_seg(a, b, H, W, off, width, g)) return;
    const int nminor = g.xmaj ? H : W;
    for (int i = g.i0; i <= g.i1; ++i) {
        int jc;
        if (!mh_seg_fragment(g, i, off, rule, jc)) continue;
        const float t = mh_seg_t(g, i, jc, off);
        const float zw = a.zw + t * (b.zw - a.zw);
        if (!(zw >= 0.0f && zw <= 1.0f)) continue;
        const unsigned long long key = ((unsigned long long)__float_as_uint(zw) << 32) | (prim_base + (unsigned)s);
        const int j0 = g.xmaj ? jc - (width - 1) : jc;     // the column of `width` fragments (see mh_setup_seg)
        for (int k = 0; k < width; ++k) {
            const int j = j0 + k;
            if (j < 0 || j >= nminor) continue;
            const size_t pix = g.xmaj ? ((size_t)j * W + i) : ((size_t)i * W + j);
            atomicMin(&zbuf[pix], key);
        }
    }
}

__global__ __launch_bounds__(256) void mh_raster_resolve_color_kernel(
    const MhRVert *__restrict__ vt, const int32_t *__restrict__ faces, int Nv, int Nf, const MhRLVert *__restrict__ lv,
    int H, int W, int off, int width, int rule, int color_option, int depth_option, float clear,
    const unsigned long long *__restrict__ zbuf, float *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)H * W) return;
    const unsigned long long key = zbuf[i];
    float c0 = clear, c1 = clear, c2 = clear;
    if (key != ~0ull) {
        const int r = (int)(i / W), c = (int)(i % W);
        const unsigned prim = (unsigned)(key & 0xffffffffu);
        if (prim < (unsigned)Nf) {
            float g = 0.0f;
            if (depth_option == 0) {
                MhRTri t;
                float l0, l1, l2;
                mh_setup_tri(vt, faces, (int)prim, Nv, H, W, off, t);
                mh_cover(t, r, c, off, l0, l1, l2);
                const float s = (l0 * t.ia + l1 * t.ib) + l2 * t.ic;
                g = (1.0f / s) / 2.0f;
            } else if (depth_option == 2) {
                g = 1.0f;
            }
            c0 = c1 = c2 = g;
        } else {
            const int s = (int)(prim - (unsigned)Nf);
            const MhRLVert a = lv[2 * s], b = lv[2 * s + 1];
            MhRSeg g;
            mh_setup_seg(a, b, H, W, off, width, g);
            // the fragment of the (offset) 1-pixel line in this pixel's column (a wide line replicates it `width` times)
            const int mi = g.xmaj ? c : r;
            int jc = g.xmaj ? r : c;
            mh_seg_fragment(g, mi, off, rule, jc);
            const float t = mh_seg_t(g, mi, jc, off);
            const float wa = (1.0f - t) * a.iw, wb = t * b.iw;
            const float den = wa + wb;
            const float depth = (wa * a.depth + wb * b.depth) / den;
            const float tx = (wa * a.tx + wb * b.tx) / den;
            const float ty = (wa * a.ty + wb * b.ty) / den;
            if (color_option == 0) {
                c0 = c1 = c2 = depth / 2.0f;
            } else if (color_option == 1) {
                const float rr = __builtin_sqrtf(tx * tx + ty * ty);
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
        }
    }
    out[i * 3] = c0;
    out[i * 3 + 1] = c1;
    out[i * 3 + 2] = c2;
}

extern "C" int mh_launch_render_strands(const float *cam, const float *verts, int Nv, const int32_t *faces, int Nf,
                                        const float *lpts, const float *ltan, int Ns, int H, int W, int off, int snap,
                                        int width, int rule, int color_option, int depth_option, float clear, MhRVert *vt,
                                        MhRLVert *lv,
                                        unsigned long long *zbuf, int32_t *queue, unsigned int *qcount, float *out,
                                        hipStream_t st) {
    hipError_t e = hipMemsetAsync(zbuf, 0xff, (size_t)H * W * sizeof(unsigned long long), st);
    if (e != hipSuccess) return (int)e;
    if (Nv > 0 && Nf > 0) {
        e = hipMemsetAsync(qcount, 0, sizeof(unsigned int), st);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(mh_raster_vertex_kernel, dim3((Nv + 255) / 256), dim3(256), 0, st, cam, verts, Nv, (float)H,
                           (float)W, snap, vt);
        hipLaunchKernelGGL(mh_raster_small_kernel, dim3((Nf + 255) / 256), dim3(256), 0, st, vt, faces, Nf, Nv, H, W,
                           off, zbuf, queue, qcount);
        const int blocks = (Nf + 3) / 4 < 4096 ? (Nf + 3) / 4 : 4096;
        hipLaunchKernelGGL(mh_raster_large_kernel, dim3(blocks), dim3(256), 0, st, vt, faces, Nv, H, W, off, zbuf,
                           queue, qcount);
    }
    if (Ns > 0 && color_option >= 0) {
        hipLaunchKernelGGL(mh_raster_linevert_kernel, dim3((2 * Ns + 255) / 256), dim3(256), 0, st, cam, lpts, ltan,
                           2 * Ns, (float)H, (float)W, snap, lv);
        hipLaunchKernelGGL(mh_raster_lines_kernel, dim3((Ns + 255) / 256), dim3(256), 0, st, lv, Ns, H, W, off, width,
                           rule, (unsigned)(Nf > 0 && Nv > 0 ? Nf : 0), zbuf);
    }
    hipLaunchKernelGGL(mh_raster_resolve_color_kernel, dim3((unsigned)(((size_t)H * W + 255) / 256)), dim3(256), 0, st,
                       vt, faces, Nv, (Nf > 0 && Nv > 0) ? Nf : 0, lv, H, W, off, width, rule, color_option, depth_option,
                       clear, zbuf, out);
    return (int)hipGetLastError();
}

// forces this translation unit's code object onto the device (HIP loads a fat binary on the first use of one of its kernels:
// 2-20 ms each, which a one-shot pass would pay in the middle of its stages); called from mh_ctx_create
extern "C" int mh_preload_raster() {
    hipFuncAttributes a;
    return (int)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&mh_raster_vertex_kernel));
}
