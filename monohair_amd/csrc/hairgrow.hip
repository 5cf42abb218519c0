// hairgrow.hip -- strand tracing on the fitted orientation/occupancy volume, gfx950 only.
// Reference: HairGrow.py -- HairGrowing.trace :59-149, HairGrowing.traceFromScalp :154-223.  The reference walks
// every seed in a Python `while` loop with per-element tensor indexing; here one lane walks one seed, every step
// is one dependent 16-B voxel fetch {ori_x, ori_y, ori_z, occ} (latency bound; tens of thousands of seeds in
// flight hide it).  The `flag`-volume gate of GenerateGuideStrandFromScalp / randomlyGenerateSegments is
// sequential over seeds but only decides whether a finished trace is kept, so all seeds are traced in parallel and
// the gate is replayed afterwards on the host (capi.cpp: mh_strands_accept).
// Arithmetic as the reference's torch ops: torch.dot of 3-vectors = separately rounded products added left to
// right, torch.linalg.norm = sqrt of an fma chain, .type(torch.long) = truncation.
#include "mh_device.h"

struct MhVolume {
    int W, H, Z;
    const float4 *vox;   // [Z][H][W]
};

__device__ __forceinline__ float4 mh_voxel_at(const MhVolume &v, float x, float y, float z) {
    const int ix = min(max((int)x, 0), v.W - 1), iy = min(max((int)y, 0), v.H - 1), iz = min(max((int)z, 0), v.Z - 1);
    return v.vox[((size_t)iz * v.H + iy) * v.W + ix];
}
__device__ __forceinline__ float mh_dot3(float a0, float a1, float a2, float b0, float b1, float b2) {
    return (a0 * b0 + a1 * b1) + a2 * b2;
}
__device__ __forceinline__ float mh_norm3(float a0, float a1, float a2) {
    float s = a0 * a0;
    s = mh_fma(a1, a1, s);
    s = mh_fma(a2, a2, s);
    return __builtin_sqrtf(s);
}

// HairGrowing.trace without the flag test: forward <= 256 steps, then backward <= 256 steps from the seed.
// The strand of seed i occupies out[i][first .. first+len), centre slot 256.
__global__ __launch_bounds__(256) void mh_trace_seeds_kernel(MhVolume vol, const float *__restrict__ seeds, int n,
                                                             float thr, float *__restrict__ out,
                                                             int32_t *__restrict__ first, int32_t *__restrict__ len) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float *__restrict__ o = out + (size_t)i * 513 * 3;
    const float s0 = seeds[3 * i], s1 = seeds[3 * i + 1], s2 = seeds[3 * i + 2];
    int nf = 0, nb = 0;
    o[256 * 3] = s0;
    o[256 * 3 + 1] = s1;
    o[256 * 3 + 2] = s2;
#pragma unroll
    for (int dir = 0; dir < 2; ++dir) {
        const float sgn = dir == 0 ? 1.0f : -1.0f;
        float p0 = s0, p1 = s1, p2 = s2;
        float4 vx = mh_voxel_at(vol, p0, p1, p2);
        float t0 = vx.x, t1 = vx.y, t2 = vx.z;
        int steps = 0;
        for (int count = 0;;) {
            if (vx.w == 0.0f) break;
            // seedPos + Tan / seedPos - Tan: x - t == x + (-t) exactly
            const float n0 = p0 + sgn * t0, n1 = p1 + sgn * t1, n2 = p2 + sgn * t2;
            const float4 nv = mh_voxel_at(vol, n0, n1, n2);
            if (mh_dot3(nv.x, nv.y, nv.z, t0, t1, t2) < thr) break;
            p0 = n0;
            p1 = n1;
            p2 = n2;
            t0 = nv.x;
            t1 = nv.y;
            t2 = nv.z;
            vx = nv;
            ++steps;
            float *w = o + (256 + (dir == 0 ? steps : -steps)) * 3;
            w[0] = p0;
            w[1] = p1;
            w[2] = p2;
            if (++count >= 256) break;
        }
        if (dir == 0) nf = steps;
        else nb = steps;
    }
    first[i] = 256 - nb;
    len[i] = nf + nb + 1;
}

// HairGrowing.traceFromScalp: len = 0 when the reference returns None.
__global__ __launch_bounds__(256) void mh_trace_scalp_kernel(MhVolume vol, const float *__restrict__ seeds,
                                                             const float *__restrict__ normals, int n, float thr,
                                                             float *__restrict__ out, int32_t *__restrict__ len) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float *__restrict__ o = out + (size_t)i * 257 * 3;
    float p0 = seeds[3 * i], p1 = seeds[3 * i + 1], p2 = seeds[3 * i + 2];
    const float m0 = normals[3 * i], m1 = normals[3 * i + 1], m2 = normals[3 * i + 2];
    float lift = mh_dot3(m0, m1, m2, 0.0f, 1.0f, 0.0f) + 1.0f;
    if (!(lift < 1.0f)) lift = 1.0f;
    const float a0 = m0 + 0.0f * lift, a1 = m1 + 1.0f * lift, a2 = m2 + 0.0f * lift;
    const float ln = mh_norm3(a0, a1, a2);
    float t0 = a0 / ln, t1 = a1 / ln, t2 = a2 / ln;
    float4 vx = mh_voxel_at(vol, p0, p1, p2);
    int cnt = 1, count = 0;
    bool inner = true;
    o[0] = p0;
    o[1] = p1;
    o[2] = p2;
    for (;;) {
        if (vx.w == 0.0f && !inner) break;
        const float n0 = p0 + t0, n1 = p1 + t1, n2 = p2 + t2;
        const float4 nv = mh_voxel_at(vol, n0, n1, n2);
        float u0 = nv.x, u1 = nv.y, u2 = nv.z;
        if (mh_norm3(u0, u1, u2) < 0.1f && inner) {
            if (mh_dot3(t0, t1, t2, m0, m1, m2) < 0.85f) {
                u0 = t0;
                u1 = t1;
                u2 = t2;
            } else {
                const float b0 = t0 + 0.0f * lift, b1 = t1 + 1.0f * lift, b2 = t2 + 0.0f * lift;
                const float l2 = mh_norm3(b0, b1, b2);
                u0 = b0 / l2;
                u1 = b1 / l2;
                u2 = b2 / l2;
            }
        } else {
            if (mh_dot3(u0, u1, u2, t0, t1, t2) < thr && !inner) {
                if (mh_dot3(-u0, -u1, -u2, t0, t1, t2) < thr) break;
                u0 = -u0;
                u1 = -u1;
                u2 = -u2;
            }
            if (mh_dot3(u0, u1, u2, t0, t1, t2) < 0.0f && inner) {
                u0 = -u0;
                u1 = -u1;
                u2 = -u2;
            }
            inner = false;
        }
        p0 = n0;
        p1 = n1;
        p2 = n2;
        t0 = u0;
        t1 = u1;
        t2 = u2;
        vx = mh_voxel_at(vol, p0, p1, p2);
        o[cnt * 3] = p0;
        o[cnt * 3 + 1] = p1;
        o[cnt * 3 + 2] = p2;
        ++cnt;
        ++count;
        if (count >= 256) break;
        if (count >= 25 && inner) break;
    }
    len[i] = inner ? 0 : cnt;
}

// occ[Z][H][W] + ori[Z][H][W][3] (the readers' layout, PMVO_utils.py:86-113) -> packed voxels with y,z negated
// (HairGrowing.__init__, HairGrow.py:55)
__global__ __launch_bounds__(256) void mh_pack_volume_kernel(const float *__restrict__ occ,
                                                             const float *__restrict__ ori, size_t nvox,
                                                             float4 *__restrict__ vox) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t step = (size_t)gridDim.x * 256;
    for (; i < nvox; i += step) vox[i] = make_float4(ori[3 * i], -ori[3 * i + 1], -ori[3 * i + 2], occ[i]);
}

extern "C" int mh_launch_pack_volume(const float *occ, const float *ori, size_t nvox, float4 *vox, hipStream_t st) {
    const int blocks = (int)((nvox + 255) / 256 < 4096 ? (nvox + 255) / 256 : 4096);
    hipLaunchKernelGGL(mh_pack_volume_kernel, dim3(blocks), dim3(256), 0, st, occ, ori, nvox, vox);
    return (int)hipGetLastError();
}
extern "C" int mh_launch_trace_seeds(const float4 *vox, int W, int H, int Z, const float *seeds, int n, float thr,
                                     float *out, int32_t *first, int32_t *len, hipStream_t st) {
    MhVolume v{W, H, Z, vox};
    hipLaunchKernelGGL(mh_trace_seeds_kernel, dim3((n + 255) / 256), dim3(256), 0, st, v, seeds, n, thr, out, first,
                       len);
    return (int)hipGetLastError();
}
extern "C" int mh_launch_trace_scalp(const float4 *vox, int W, int H, int Z, const float *seeds, const float *normals,
                                     int n, float thr, float *out, int32_t *len, hipStream_t st) {
    MhVolume v{W, H, Z, vox};
    hipLaunchKernelGGL(mh_trace_scalp_kernel, dim3((n + 255) / 256), dim3(256), 0, st, v, seeds, normals, n, thr, out,
                       len);
    return (int)hipGetLastError();
}

// The traces come back in fixed-stride rows (513 / 257 points per seed, mostly empty): pack the points that exist
// one strand after the other before they cross PCIe.  One wave per strand; offs[i] = exclusive prefix sum of len.
__global__ __launch_bounds__(256) void mh_strands_compact_kernel(const float *__restrict__ rows,
                                                                 const int32_t *__restrict__ first,
                                                                 const int32_t *__restrict__ len,
                                                                 const int64_t *__restrict__ offs, int n, int stride,
                                                                 float *__restrict__ packed) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    const int l = len[i];
    if (l <= 0) return;
    const float *__restrict__ src = rows + ((size_t)i * stride + (first ? first[i] : 0)) * 3;
    float *__restrict__ dst = packed + (size_t)offs[i] * 3;
    for (int k = lane; k < 3 * l; k += MH_WAVE) dst[k] = src[k];
}

extern "C" int mh_launch_strands_compact(const float *rows, const int32_t *first, const int32_t *len, const int64_t *offs,
                                         int n, int stride, float *packed, hipStream_t st) {
    hipLaunchKernelGGL(mh_strands_compact_kernel, dim3((n + 3) / 4), dim3(256), 0, st, rows, first, len, offs, n, stride,
                       packed);
    return (int)hipGetLastError();
}

// forces this translation unit's code object onto the device (HIP loads a fat binary on the first use of one of its kernels:
// 2-20 ms each, which a one-shot pass would pay in the middle of its stages); called from mh_ctx_create
extern "C" int mh_preload_hairgrow() {
    hipFuncAttributes a;
    return (int)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&mh_trace_seeds_kernel));
}
