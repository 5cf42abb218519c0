// dog.hip -- the difference-of-Gaussians prefilter in front of the Gabor bank, gfx950 only.
// Reference: preprocess_capture_data/GaborFilter.py:190-192 -> skimage.filters.difference_of_gaussians(img, 0.4, 10)
//   = img_as_float (uint8 codes MULTIPLIED by the float64 constant 1/255), then for each sigma
//     scipy.ndimage.gaussian_filter(mode='nearest', truncate=4.0): correlate1d along axis 0, then along axis 1, float64,
//     and the difference of the two results.
// scipy's correlate1d for a symmetric kernel of radius r (ni_filters.c) evaluates, per output element,
//     tmp = x[l] * w[r];   for j = -r .. -1:  tmp += (x[l+j] + x[l-j]) * w[j+r];
// with separately rounded float64 operations -- this file does exactly that (compiled with -ffp-contract=off), so the result
// equals the host evaluation bit for bit (tests/golden/dog.npz pins it to the real scikit-image).
//
// Two launches replace the ~250 elementwise torch launches of the first device version:
//   mh_dog_vert_kernel : codes -> float64 -> correlate along rows' axis (axis 0) for BOTH sigmas; a workgroup owns a
//                        32-row x 64-column tile, the (32 + 2 r_max) x 64 input rows sit in LDS as float64 (a lane walks a
//                        column: consecutive lanes = consecutive LDS words, conflict-free), weights in LDS (broadcast reads);
//   mh_dog_horz_kernel : correlate along axis 1, subtract, cast -- 2 rows x 128 columns per workgroup, both planes with
//                        their halo in LDS.
// HBM traffic: 1 B/px in, 16 B/px written + read between the passes, 4 (+8) B/px out: ~80 MB per 1080p view.
#include "mh_device.h"

#define MH_DG_MAXR 48                 // radius limit: sigma <= 11.9 at truncate 4 (the reference uses 0.4 and 10: 2 and 40)
#define MH_DG_VT_ROWS 32
#define MH_DG_VT_COLS 64
#define MH_DG_HZ_ROWS 2
#define MH_DG_HZ_COLS 128

struct MhDogWeights {                 // device-resident; w[s][j + r[s]] for j = -r[s] .. 0 (the symmetric half incl. the centre)
    double w[2][MH_DG_MAXR + 1];
    int r[2];
};

template <int KIND>                   // 0: uint8 codes (x 1/255), 1: float64 samples
__global__ __launch_bounds__(256) void mh_dog_vert_kernel(const void *__restrict__ img, int H, int W,
                                                          const MhDogWeights *__restrict__ wt, double *__restrict__ ylo,
                                                          double *__restrict__ yhi) {
    __shared__ double tile[(MH_DG_VT_ROWS + 2 * MH_DG_MAXR) * MH_DG_VT_COLS];     // 64 KB at the radius limit
    __shared__ double sw[2][MH_DG_MAXR + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = wt->r[0], r1 = wt->r[1];
    const int R = r0 > r1 ? r0 : r1;
    const int x0 = blockIdx.x * MH_DG_VT_COLS, y0 = blockIdx.y * MH_DG_VT_ROWS;
    if (tid <= MH_DG_MAXR) {
        sw[0][tid] = wt->w[0][tid];
        sw[1][tid] = wt->w[1][tid];
    }
    const int rows = MH_DG_VT_ROWS + 2 * R;
    int gx = x0 + lane;
    gx = gx < W ? gx : W - 1;                                   // (columns past the edge are computed and not stored)
    for (int ly = wave; ly < rows; ly += 4) {
        int gy = y0 - R + ly;
        gy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);               // mode='nearest'
        double v;
        if (KIND == 0)
            v = (double)((const uint8_t *)img)[(size_t)gy * W + gx] * (1.0 / 255);      // img_as_float: a multiplication
        else
            v = ((const double *)img)[(size_t)gy * W + gx];
        tile[ly * MH_DG_VT_COLS + lane] = v;
    }
    __syncthreads();
    if (x0 + lane >= W) return;
    // a lane produces its 8 rows in two blocks of 4: the 4 outputs of a block share their inputs (per group of 4 taps: 7 + 7
    // rows and 4 weights from LDS instead of 4 x 4 x 3 reads -- the pass is LDS-bound), each output keeps its own accumulator
    // and runs scipy's sequence tmp = x[l]*w[r]; tmp += (x[l+j] + x[l-j]) * w[j+r], j = -r .. -1
    const int rem = r1 & 3;
#pragma unroll 1
    for (int blk = 0; blk < MH_DG_VT_ROWS / 16; ++blk) {
        const int ry = wave * (MH_DG_VT_ROWS / 4) + 4 * blk;    // first row of the block in the tile
        if (y0 + ry >= H) break;
        const double *__restrict__ c = tile + (size_t)(ry + R) * MH_DG_VT_COLS + lane;
        double a0[4], a1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double *__restrict__ cq = c + q * MH_DG_VT_COLS;
            double t = cq[0] * sw[0][r0];
            for (int j = -r0; j < 0; ++j) t = t + (cq[j * MH_DG_VT_COLS] + cq[-j * MH_DG_VT_COLS]) * sw[0][j + r0];
            a0[q] = t;
            double u = cq[0] * sw[1][r1];
            for (int j = -r1; j < -r1 + rem; ++j) u = u + (cq[j * MH_DG_VT_COLS] + cq[-j * MH_DG_VT_COLS]) * sw[1][j + r1];
            a1[q] = u;
        }
#pragma unroll 1
        for (int j0 = -r1 + rem; j0 < 0; j0 += 4) {
            double lo[7], hi[7], w[4];
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                lo[i] = c[(j0 + i) * MH_DG_VT_COLS];            // rows l0 + j0 .. l0 + j0 + 6
                hi[i] = c[(-j0 - 3 + i) * MH_DG_VT_COLS];       // rows l0 - j0 - 3 .. l0 - j0 + 3
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) w[t] = sw[1][j0 + t + r1];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) a1[q] = a1[q] + (lo[q + t] + hi[q - t + 3]) * w[t];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (y0 + ry + q < H) {
                const size_t o = (size_t)(y0 + ry + q) * W + x0 + lane;
                ylo[o] = a0[q];
                yhi[o] = a1[q];
            }
        }
    }
}

__global__ __launch_bounds__(256) void mh_dog_horz_kernel(const double *__restrict__ ylo, const double *__restrict__ yhi,
                                                          int H, int W, const MhDogWeights *__restrict__ wt,
                                                          double *__restrict__ out64, float *__restrict__ out32) {
    __shared__ double tlo[MH_DG_HZ_ROWS][MH_DG_HZ_COLS + 2 * MH_DG_MAXR];
    __shared__ double thi[MH_DG_HZ_ROWS][MH_DG_HZ_COLS + 2 * MH_DG_MAXR];
    __shared__ double sw[2][MH_DG_MAXR + 1];
    const int tid = threadIdx.x;
    const int r0 = wt->r[0], r1 = wt->r[1];
    const int x0 = blockIdx.x * MH_DG_HZ_COLS, y0 = blockIdx.y * MH_DG_HZ_ROWS;
    if (tid <= MH_DG_MAXR) {
        sw[0][tid] = wt->w[0][tid];
        sw[1][tid] = wt->w[1][tid];
    }
    for (int row = 0; row < MH_DG_HZ_ROWS; ++row) {
        int gy = y0 + row;
        gy = gy < H ? gy : H - 1;
        for (int lx = tid; lx < MH_DG_HZ_COLS + 2 * r0; lx += 256) {
            int gx = x0 - r0 + lx;
            gx = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
            tlo[row][lx] = ylo[(size_t)gy * W + gx];
        }
        for (int lx = tid; lx < MH_DG_HZ_COLS + 2 * r1; lx += 256) {
            int gx = x0 - r1 + lx;
            gx = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
            thi[row][lx] = yhi[(size_t)gy * W + gx];
        }
    }
    __syncthreads();
    const int row = tid >> 7, col = tid & 127;
    const int y = y0 + row, x = x0 + col;
    if (y >= H || x >= W) return;
    const double *__restrict__ c0 = &tlo[row][col + r0];
    double a0 = c0[0] * sw[0][r0];
    for (int j = -r0; j < 0; ++j) a0 = a0 + (c0[j] + c0[-j]) * sw[0][j + r0];
    const double *__restrict__ c1 = &thi[row][col + r1];
    double a1 = c1[0] * sw[1][r1];
#pragma unroll 4
    for (int j = -r1; j < 0; ++j) a1 = a1 + (c1[j] + c1[-j]) * sw[1][j + r1];
    const double d = a0 - a1;
    const size_t o = (size_t)y * W + x;
    if (out64) out64[o] = d;
    if (out32) out32[o] = (float)d;
}

// in_kind 0: uint8 [H,W]; 1: float64 [H,W].  scratch: 2 * H * W doubles.  weights: device MhDogWeights.
extern "C" int mh_launch_dog(const void *img, int in_kind, int H, int W, const void *weights, double *scratch, double *out64,
                             float *out32, hipStream_t st) {
    const MhDogWeights *wt = (const MhDogWeights *)weights;
    double *ylo = scratch, *yhi = scratch + (size_t)H * W;
    const dim3 gv((W + MH_DG_VT_COLS - 1) / MH_DG_VT_COLS, (H + MH_DG_VT_ROWS - 1) / MH_DG_VT_ROWS);
    if (in_kind == 0)
        hipLaunchKernelGGL(mh_dog_vert_kernel<0>, gv, dim3(256), 0, st, img, H, W, wt, ylo, yhi);
    else
        hipLaunchKernelGGL(mh_dog_vert_kernel<1>, gv, dim3(256), 0, st, img, H, W, wt, ylo, yhi);
    const dim3 gh((W + MH_DG_HZ_COLS - 1) / MH_DG_HZ_COLS, (H + MH_DG_HZ_ROWS - 1) / MH_DG_HZ_ROWS);
    hipLaunchKernelGGL(mh_dog_horz_kernel, gh, dim3(256), 0, st, ylo, yhi, H, W, wt, out64, out32);
    return (int)hipGetLastError();
}

// forces this translation unit's code object onto the device (HIP loads a fat binary on the first use of one of its kernels:
// 2-20 ms each, which a one-shot pass would pay in the middle of its stages); called from mh_ctx_create
extern "C" int mh_preload_dog() {
    hipFuncAttributes a;
    return (int)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&mh_dog_horz_kernel));
}
