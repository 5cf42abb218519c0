// gabor.hip -- the per-view Gabor orientation/confidence bank (K1+K2), gfx950 only.
// Reference: preprocess_capture_data/GaborFilter.py -- gabor_fn :115-145, filter :29-94, forward :98-113.
//
//   R_k = | sum_{i,j} img[y+i-8, x+j-8] * g_k[i,j] |      180 kernels, 17x17, zero padding
//   M = max_k R_k, b = first argmax;  d_k = circular distance between theta_b and theta_k
//   var = sqrt( sum_k d_k * (R_k-M)^2 );  conf = clamp( (var / max_image(var)) / 0.2, 0, 1 )
//
// Direct form, compute bound (2*180*289 FLOP per pixel against 12 B): every lane owns one pixel and keeps
// ALL 180 accumulators in registers (gfx950: 512 VGPR+AGPR per lane at one wave per SIMD); the image tile
// sits in LDS (one ds_read per tap), the bank is stored tap-major ([289][192]) so the 180 weights of a tap
// are one uniform, contiguous run that the scalar unit streams into SGPRs -- the inner loop is 180
// v_fma_f32 with an SGPR operand per LDS read.  The [1,180,H,W] response stack of the reference (1.49 GB
// at 1080p) never exists; argmax, variance and the image-wide maximum are fused behind the accumulators.
#include "mh_device.h"

#define MH_GB_NK 180
#define MH_GB_KS 17
#define MH_GB_NT (MH_GB_KS * MH_GB_KS)
#define MH_GB_KPAD 192
#define MH_GB_TILE 16
#define MH_GB_LDW (MH_GB_TILE + MH_GB_KS - 1)   // 32

// theta_k exactly as the reference builds it in fp32: ones * pi * k / 180  (GaborFilter.py:44,52)
__host__ __device__ __forceinline__ float mh_theta(float k) { return (3.14159265358979323846f * k) / 180.0f; }

// gabor_fn (GaborFilter.py:115-145), fp32 tensor ops in the reference's order; stored tap-major.
__global__ void mh_gabor_build_kernel(float *__restrict__ bankT) {
    const int k = blockIdx.x, t = threadIdx.x;
    if (t >= MH_GB_NT) return;
    const int i = t / MH_GB_KS, j = t - i * MH_GB_KS;
    const float theta = (float)(3.14159265358979323846 * (double)k / 180.0);   // math.pi*k/180 in double, then fp32
    const float x = (float)(i - 8) - 0.5f, y = (float)(j - 8) - 0.5f;
    const float ct = cosf(theta), st = sinf(theta);
    const float xt = x * ct + y * st;
    const float yt = -x * st + y * ct;
    const float sx = 1.8f, sy = 2.4f, lam = 4.0f;
    const float e = -0.5f * ((xt * xt) / (sx * sx) + (yt * yt) / (sy * sy));
    const float car = (2.0f * 3.14159265358979323846f * xt) / lam + 0.0f;
    bankT[t * MH_GB_KPAD + k] = expf(e) * cosf(car);
}

__global__ __launch_bounds__(256) void mh_gabor_bank_kernel(const float *__restrict__ bankT,
                                                            const float *__restrict__ img, int H, int W,
                                                            int32_t *__restrict__ orient, float *__restrict__ var_out,
                                                            unsigned int *__restrict__ maxbits) {
    __shared__ float tile[MH_GB_LDW * MH_GB_LDW];
    const int tid = threadIdx.x;
    const int ty = tid >> 4, tx = tid & 15;
    const int y0 = blockIdx.y * MH_GB_TILE, x0 = blockIdx.x * MH_GB_TILE;
    for (int q = tid; q < MH_GB_LDW * MH_GB_LDW; q += 256) {
        const int ly = q / MH_GB_LDW, lx = q - ly * MH_GB_LDW;
        const int gy = y0 + ly - 8, gx = x0 + lx - 8;
        tile[q] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? img[(size_t)gy * W + gx] : 0.0f;
    }
    __syncthreads();

    // 180 accumulators as 90 register pairs: v_pk_fma_f32 retires two fp32 FMAs per 4-cycle issue slot
    // (measured: v_fma_f32 3.5 cycles for one) and is, per element, the same single-rounded fma.
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f acc2[MH_GB_NK / 2];
#pragma unroll
    for (int k = 0; k < MH_GB_NK / 2; ++k) acc2[k] = v2f{0.0f, 0.0f};
    for (int i = 0; i < MH_GB_KS; ++i) {
        for (int j = 0; j < MH_GB_KS; ++j) {
            const float x = tile[(ty + i) * MH_GB_LDW + tx + j];
            const v2f x2 = v2f{x, x};
            const v2f *__restrict__ wt = reinterpret_cast<const v2f *>(bankT + (i * MH_GB_KS + j) * MH_GB_KPAD);
#pragma unroll
            for (int k = 0; k < MH_GB_NK / 2; ++k) acc2[k] = __builtin_elementwise_fma(x2, wt[k], acc2[k]);
        }
    }
    float acc[MH_GB_NK];
#pragma unroll
    for (int k = 0; k < MH_GB_NK / 2; ++k) {
        acc[2 * k] = acc2[k].x;
        acc[2 * k + 1] = acc2[k].y;
    }

    // argmax of |response| (first maximum), GaborFilter.py:48-51
    float M = __builtin_fabsf(acc[0]);
    int b = 0;
#pragma unroll
    for (int k = 1; k < MH_GB_NK; ++k) {
        const float r = __builtin_fabsf(acc[k]);
        if (r > M) {
            M = r;
            b = k;
        }
    }
    // variance of the response curve, GaborFilter.py:52-78; sum over k in ATen's cascade order
    const float PI_F = 3.14159265358979323846f;
    const float bh = mh_theta((float)b);
    MhCasc s = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < MH_GB_NK; ++k) {
        if (k > 0 && (k & 15) == 0) mh_casc_flush(s);
        const float t1 = bh - mh_theta((float)k);
        const float d = fminf(__builtin_fabsf(t1), fminf(__builtin_fabsf(t1 - PI_F), __builtin_fabsf(t1 + PI_F)));
        const float rd = __builtin_fabsf(acc[k]) - M;
        s.a0 = s.a0 + (d * rd) * rd;
    }
    const float var = __builtin_sqrtf(s.a0 + s.a1);
    const int y = y0 + ty, x = x0 + tx;
    float vmax = 0.0f;
    if (y < H && x < W) {
        var_out[(size_t)y * W + x] = var;
        orient[(size_t)y * W + x] = (var > 0.0f) ? b : 0;
        vmax = var;
    }
    // image-wide maximum: wave shuffle, then one atomic per wave (non-negative floats order like uints)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    if ((tid & 63) == 0) atomicMax(maxbits, __float_as_uint(vmax));
}

// ---------------------------------------------------------------------------------------------
// mh_gabor_mfma2_kernel (the shipped form): the bank as an im2col contraction  C[pixel, k] = sum_t A[pixel, t] * B[t, k]
// (pixels x 289 taps x 180 orientations) on v_mfma_f32_32x32x2_f32.  The MFMA result is bit for bit a k-ordered fp32 fma
// chain, i.e. exactly the tap-ordered chain of the VALU kernel above (kept as the cross-check), so both kernels and the CPU
// oracle produce identical maps.  The coefficient matrix lives in VGPRs distributed over the lanes -- no broadcast through
// SGPRs, which is what holds the v_pk_fma kernel at ~58 %.
//   workgroup = 4 waves = 8 image rows x 32 columns; wave w owns rows 2w, 2w+1 (two 32-pixel M-tiles) and all six 32-wide
//   orientation N-tiles: 12 accumulators x 16 registers, 2 waves per SIMD.  Per K-step (2 taps): the A fragments are one
//   ds_read each from the LDS image tile (lane l: pixel l&31, tap 2s + (l>>5)), the B fragments two dwordx4 loads of the
//   re-laid-out bank, both requested one step ahead; 12 MFMAs (768 cycles).
//   * the bank is re-laid out once per installation as bankQ[step][lane][8]: the six coefficients a lane needs for one K-step
//     are 32 contiguous bytes (coalesced 2 KB per wave);
//   * the K loop runs in blocks of 17 steps (34 taps = two rows of the 17 x 17 window), fully unrolled, so that every LDS
//     read is `base register + immediate`: lanes 32-63 (tap 2s+1) keep a second base for the one step per block whose
//     odd tap wraps to the next window row.  No address arithmetic is left in the loop.
// (The first MFMA form -- six dword bank loads and computed LDS addresses per step, 2.11 ms per 1080p view -- and the
// two-pixels-per-lane "split" VALU form were removed in round 4; docs/HISTORY.md has their measurements.)
// ---------------------------------------------------------------------------------------------
#define MH_GM_ROWS 8
#define MH_GM_COLS 32
#define MH_GM_LDW (MH_GM_COLS + MH_GB_KS - 1)        // 48
#define MH_GM_LDH (MH_GM_ROWS + MH_GB_KS)            // 25 (one spare row for the zero-weight pad tap)
#define MH_GM_RS 33

typedef float mh_f32x16 __attribute__((ext_vector_type(16)));

#define MH_GM_NSTEP ((MH_GB_NT + 1) / 2)     // 145 K-steps; tap 289 is the zero row

// bankT [290][192] -> bankQ [145][64][8]: lane (c = lane & 31, kk = lane >> 5) of step s holds bankT[2s+kk][n*32 + c], n = 0..5
__global__ void mh_gabor_relayout_kernel(const float *__restrict__ bankT, float *__restrict__ bankQ) {
    const int s = blockIdx.x, lane = threadIdx.x;
    const int c = lane & 31, kk = lane >> 5;
#pragma unroll
    for (int n = 0; n < 8; ++n)
        bankQ[((size_t)s * 64 + lane) * 8 + n] = n < 6 ? bankT[(size_t)(2 * s + kk) * MH_GB_KPAD + n * 32 + c] : 0.0f;
}

__host__ __device__ constexpr int mh_gm_off(int u) { return ((2 * u) / MH_GB_KS) * MH_GM_LDW + (2 * u) % MH_GB_KS; }
__host__ __device__ constexpr bool mh_gm_wrap(int u) { return (2 * u) % MH_GB_KS == MH_GB_KS - 1; }

// NU steps of one 17-step block.  lds_n / lds_w: LDS byte addresses of this lane for the block's first window row (normal
// / wrapping step); bqs: the (wave-uniform) bankQ pointer at the block's first step, voff: this lane's byte offset in a
// step's 2 KB; (a0, a1, b0, b1) hold the operands of the block's first step on entry and those of the next block's first
// step on exit.
// The operand requests of step u+1 are written as inline assembly in front of step u's 12 MFMAs and waited for BEHIND them
// (`s_waitcnt` in a second asm that passes the registers through): left to the compiler, the bank loads were sunk to
// just in front of their first use and every K-step began with an exposed L2 round trip (2.00 ms instead of 1.93).
typedef float mh_f32x4 __attribute__((ext_vector_type(4)));

template <int NU, bool LAST>
__device__ __forceinline__ void mh_gm_block(unsigned lds_n, unsigned lds_w, const float *__restrict__ bqs, unsigned voff,
                                            float &a0, float &a1, mh_f32x4 &b0, mh_f32x4 &b1, mh_f32x16 (&acc)[2][6]) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        mh_f32x4 n0 = b0, n1 = b1;
        float a0n = a0, a1n = a1;
        constexpr bool dummy = false;
        (void)dummy;
        if (!(LAST && u == NU - 1)) {
            const float *__restrict__ bsrc = bqs + (size_t)(u + 1) * 512;
            const unsigned la = mh_gm_wrap(u + 1) ? lds_w : lds_n;
            asm volatile("global_load_dwordx4 %0, %4, %5\n\t"
                         "global_load_dwordx4 %1, %4, %5 offset:16\n\t"
                         "ds_read_b32 %2, %6 offset:%7\n\t"
                         "ds_read_b32 %3, %6 offset:%8"
                         : "=&v"(n0), "=&v"(n1), "=&v"(a0n), "=&v"(a1n)
                         : "v"(voff), "s"(bsrc), "v"(la), "n"(mh_gm_off(u + 1) * 4), "n"((mh_gm_off(u + 1) + MH_GM_LDW) * 4)
                         : "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0.x, acc[0][0], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0.x, acc[1][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0.y, acc[0][1], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0.y, acc[1][1], 0, 0, 0);
        acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0.z, acc[0][2], 0, 0, 0);
        acc[1][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0.z, acc[1][2], 0, 0, 0);
        acc[0][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0.w, acc[0][3], 0, 0, 0);
        acc[1][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0.w, acc[1][3], 0, 0, 0);
        acc[0][4] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1.x, acc[0][4], 0, 0, 0);
        acc[1][4] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1.x, acc[1][4], 0, 0, 0);
        acc[0][5] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1.y, acc[0][5], 0, 0, 0);
        acc[1][5] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1.y, acc[1][5], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (!(LAST && u == NU - 1))
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(n0), "+v"(n1), "+v"(a0n), "+v"(a1n)::"memory");
        b0 = n0, b1 = n1, a0 = a0n, a1 = a1n;
    }
}

__global__ __launch_bounds__(256, 2) void mh_gabor_mfma2_kernel(const float *__restrict__ bankQ,
                                                                const float *__restrict__ img, int H, int W,
                                                                int32_t *__restrict__ orient,
                                                                float *__restrict__ var_out,
                                                                unsigned int *__restrict__ maxbits) {
    __shared__ float tile[MH_GM_LDH * MH_GM_LDW];              // [25][48]
    __shared__ float stage[4][2 * 32 * MH_GM_RS];              // per wave: [M-tile][orientation of the N-tile][33]
    __shared__ float s_theta[MH_GB_KPAD];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int y0 = blockIdx.y * MH_GM_ROWS, x0 = blockIdx.x * MH_GM_COLS;
    for (int q = tid; q < MH_GM_LDH * MH_GM_LDW; q += 256) {
        const int ly = q / MH_GM_LDW, lx = q - ly * MH_GM_LDW;
        const int gy = y0 + ly - 8, gx = x0 + lx - 8;
        tile[q] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? img[(size_t)gy * W + gx] : 0.0f;
    }
    if (tid < MH_GB_KPAD) s_theta[tid] = mh_theta((float)tid);
    __syncthreads();

    mh_f32x16 acc[2][6];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int n = 0; n < 6; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][n][r] = 0.0f;

    const int pix = lane & 31, kk = lane >> 5;
    // lane bases into the LDS tile for window row 0 of M-tile 0: tap 2s (+1 for the upper half of the wave); in the one
    // step per block whose even tap sits in window column 16 the odd tap is column 0 of the next row: +(48 - 16) floats
    const float *__restrict__ an = tile + (2 * wave) * MH_GM_LDW + pix + kk;
    const float *__restrict__ aw = tile + (2 * wave) * MH_GM_LDW + pix + kk * (MH_GM_LDW - (MH_GB_KS - 1));
    unsigned lds_n = (unsigned)(size_t)(__attribute__((address_space(3))) const float *)an;
    unsigned lds_w = (unsigned)(size_t)(__attribute__((address_space(3))) const float *)aw;
    const unsigned voff = lane * 32;
    const mh_f32x4 *__restrict__ bq0 = reinterpret_cast<const mh_f32x4 *>(bankQ) + lane * 2;
    mh_f32x4 b0 = bq0[0], b1 = bq0[1];
    float a0 = an[0], a1 = an[MH_GM_LDW];
    // (the first operands are consumed here so that no compiler-tracked load is pending at the loop header: it would
    // otherwise wait there between every block's first requests and its MFMAs)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(b0), "+v"(b1), "+v"(a0), "+v"(a1)::"memory");
    const float *__restrict__ bqs = bankQ;
    constexpr int NBLK = MH_GM_NSTEP / MH_GB_KS, NREM = MH_GM_NSTEP - NBLK * MH_GB_KS;    // 8 blocks of 17 steps + 9
#pragma unroll 1
    for (int q = 0; q < NBLK; ++q) {
        mh_gm_block<MH_GB_KS, false>(lds_n, lds_w, bqs, voff, a0, a1, b0, b1, acc);
        lds_n += 2 * MH_GM_LDW * 4;
        lds_w += 2 * MH_GM_LDW * 4;
        bqs += MH_GB_KS * 512;
    }
    mh_gm_block<NREM, true>(lds_n, lds_w, bqs, voff, a0, a1, b0, b1, acc);
    // Epilogue.  C layout of the 32x32 MFMA: column (orientation) = lane & 31, row (pixel) = (r&3) + 8*(r>>2) + 4*(lane>>5).
    // Each N-tile (32 orientations x 2 x 32 pixels) is transposed ONCE through LDS so that lane L owns pixel L&31 of
    // M-tile L>>5; the 32 transposed |responses| come back into registers (the tile's accumulators are dead by then), the
    // first-maximum argmax runs on the fly, and when all six tiles are through the lane holds its pixel's 180 responses in
    // registers: the cascade-ordered variance is straight-line code on them with theta_k as literals.  Same operations in
    // the same order as the VALU kernels; half the LDS traffic and none of the per-value LDS reads of the first form (the
    // epilogue's issue time adds to the kernel's: VALU / LDS instructions of one wave are not hidden behind the other
    // wave's MFMAs on the same SIMD -- de-phasing the two workgroups of a CU, priorities and one workgroup per CU all
    // measured the same 1.91 ms, and 1.58 ms with the epilogue removed).
    float *__restrict__ st = stage[wave];
    const int mt = lane >> 5;
    const float PI_F = 3.14159265358979323846f;
    float R[MH_GB_NK];
    float M = 0.0f;
    int b = 0;
#pragma unroll
    for (int n = 0; n < 6; ++n) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
                st[(p * 32 + pix) * MH_GM_RS + row] = __builtin_fabsf(acc[p][n][r]);
            }
        // (16-byte stores of four consecutive rows with a 36-float row stride were measured slower: 1.94 vs 1.76 ms)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int km = (n == 5) ? MH_GB_NK - 160 : 32;
#pragma unroll
        for (int kl = 0; kl < km; ++kl) R[n * 32 + kl] = st[(mt * 32 + kl) * MH_GM_RS + pix];
#pragma unroll
        for (int kl = 0; kl < km; ++kl) {
            const float r = R[n * 32 + kl];
            if ((n | kl) == 0) {
                M = r;
            } else if (r > M) {
                M = r;
                b = n * 32 + kl;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    const float bh = s_theta[b];
    MhCasc sc = {0.f, 0.f};
#pragma unroll
    for (int k2 = 0; k2 < MH_GB_NK; ++k2) {
        if (k2 > 0 && (k2 & 15) == 0) mh_casc_flush(sc);
        const float t1 = bh - mh_theta((float)k2);
        const float d = fminf(__builtin_fabsf(t1), fminf(__builtin_fabsf(t1 - PI_F), __builtin_fabsf(t1 + PI_F)));
        const float rd = R[k2] - M;
        sc.a0 = sc.a0 + (d * rd) * rd;
    }
    const float var = __builtin_sqrtf(sc.a0 + sc.a1);
    const int y = y0 + 2 * wave + mt, x = x0 + pix;
    float vmax = 0.0f;
    if (y < H && x < W) {
        var_out[(size_t)y * W + x] = var;
        orient[(size_t)y * W + x] = (var > 0.0f) ? b : 0;
        vmax = var;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    if (lane == 0) atomicMax(maxbits, __float_as_uint(vmax));
}

// confidence = clamp((var / max_image(var)) / 0.2, 0, 1)  (GaborFilter.py:80-93).  Optionally also the two 8-bit FILE CODES
// the reference hands to PMVO: best_ori/<view> = the orientation index in degrees (cv2.imwrite of
// b*180/pi, GaborFilter.py:209) and conf/<view> = floor(conf*255 + 0.5) (torchvision.utils.save_image, :210) -- fp32 mul, add,
// clamp, truncation, exactly the tensor ops (conf*255+0.5).clamp(0,255).to(uint8).
__global__ __launch_bounds__(256) void mh_gabor_finish_kernel(const float *__restrict__ var,
                                                              const unsigned int *__restrict__ maxbits, size_t npix,
                                                              float *__restrict__ conf, const int32_t *__restrict__ orient,
                                                              uint8_t *__restrict__ k8, uint8_t *__restrict__ c8) {
    const float mx = __uint_as_float(*maxbits);
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < npix; i += step) {
        const float v = var[i] / mx;
        const float c = mh_clampf((v - 0.0f) / 0.2f, 0.0f, 1.0f);
        if (conf) conf[i] = c;
        if (c8) c8[i] = (uint8_t)mh_clampf(c * 255.0f + 0.5f, 0.0f, 255.0f);     // (a NaN confidence -- an all-zero image -- stores 0)
        if (k8) {
            const int32_t k = orient[i];
            k8[i] = (uint8_t)(k < 0 ? 0 : (k > 255 ? 255 : k));
        }
    }
}

extern "C" int mh_launch_gabor_build(float *bankT, hipStream_t st) {
    (void)hipMemsetAsync(bankT, 0, (size_t)(MH_GB_NT + 1) * MH_GB_KPAD * sizeof(float), st);   // + zero pad tap
    hipLaunchKernelGGL(mh_gabor_build_kernel, dim3(MH_GB_NK), dim3(320), 0, st, bankT);
    return (int)hipGetLastError();
}

// maxbits: the image maximum of the variance (one uint, zeroed per launch)
extern "C" size_t mh_gabor_state_bytes() { return 64; }

// bankQ: the re-laid-out copy for mh_gabor_mfma2_kernel, mh_gabor_bankq_bytes() bytes, made by mh_launch_gabor_relayout
extern "C" size_t mh_gabor_bankq_bytes() { return (size_t)MH_GM_NSTEP * 64 * 8 * sizeof(float); }

extern "C" int mh_launch_gabor_relayout(const float *bankT, float *bankQ, hipStream_t st) {
    hipLaunchKernelGGL(mh_gabor_relayout_kernel, dim3(MH_GM_NSTEP), dim3(64), 0, st, bankT, bankQ);
    return (int)hipGetLastError();
}

extern "C" int mh_launch_gabor_bank(const float *bankT, const float *bankQ, const float *img, int H, int W, int32_t *orient,
                                    float *conf, float *var, unsigned int *maxbits, int variant, uint8_t *k8, uint8_t *c8,
                                    hipStream_t st) {
    (void)hipMemsetAsync(maxbits, 0, sizeof(unsigned int), st);
    if (variant == 0) {            // the direct v_pk_fma form (cross-check)
        const dim3 grid((W + MH_GB_TILE - 1) / MH_GB_TILE, (H + MH_GB_TILE - 1) / MH_GB_TILE);
        hipLaunchKernelGGL(mh_gabor_bank_kernel, grid, dim3(256), 0, st, bankT, img, H, W, orient, var, maxbits);
    } else {                       // 3: the FP32-MFMA contraction (shipped default)
        const dim3 grid((W + MH_GM_COLS - 1) / MH_GM_COLS, (H + MH_GM_ROWS - 1) / MH_GM_ROWS);
        hipLaunchKernelGGL(mh_gabor_mfma2_kernel, grid, dim3(256), 0, st, bankQ, img, H, W, orient, var, maxbits);
    }
    const size_t npix = (size_t)H * W;
    const int blocks = (int)((npix + 255) / 256 < 2048 ? (npix + 255) / 256 : 2048);
    hipLaunchKernelGGL(mh_gabor_finish_kernel, dim3(blocks), dim3(256), 0, st, var, maxbits, npix, conf, orient, k8, c8);
    return (int)hipGetLastError();
}

// forces this translation unit's code object onto the device (HIP loads a fat binary on the first use of one of its kernels:
// 2-20 ms each, which a one-shot pass would pay in the middle of its stages); called from mh_ctx_create
extern "C" int mh_preload_gabor() {
    hipFuncAttributes a;
    return (int)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&mh_gabor_build_kernel));
}
