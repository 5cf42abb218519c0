/*
 * oracle/gabor_oracle.c -- TEST INFRASTRUCTURE ONLY (see pmvo_oracle.c).
 *
 * calOrientationGabor.filter / forward with iter = 1
 * (/root/reference/preprocess_capture_data/GaborFilter.py:29-113) given the 180 kernels of gabor_fn (:115-145):
 *   R_k = | sum_{i,j} img[y+i-8, x+j-8] * g_k[i,j] |  (F.conv2d = cross-correlation, zero padding 8),
 *   M = max_k R_k, b = first argmax, d_k = min(|bh-th_k|, |bh-th_k-pi|, |bh-th_k+pi|) in fp32,
 *   var = sqrt( cascade-sum_k (d_k*(R_k-M))*(R_k-M) ), orient = var>0 ? b : 0,
 *   conf = clamp( (var / max_image var) / 0.2, 0, 1 ).
 * The 289-term correlation is accumulated tap by tap (row-major) with fma -- the order the HIP kernel uses, and (probed in
 * round 5: every response of three orientations on three images, bit for bit) the order of the reference's conv2d on the CPU.
 * The orientation index and the sum under the root equal the reference's on every pixel of the goldens.  What differs is the
 * root itself: `variance ** (1 / 2)` (GaborFilter.py:77) is an elementwise op on a contiguous tensor, which ATen hands to
 * MKL's vector math library (vsSqrt, "high accuracy" mode: below 1 ulp, NOT correctly rounded -- torch.sqrt differs from the
 * IEEE root on 0.7 % of random floats and doubles, at any tensor size, contiguous or not).  sqrtf here is the IEEE root, so the
 * confidence differs from the reference's by one float32 ulp on < 0.8 % of the pixels; it equals the reference's on EVERY pixel
 * once torch's own root is applied to this oracle's sums (var_sum_out; tests/test_oracle_more.py).  MKL is closed source: the
 * root is not restated.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define NK 180
#define KS 17

static inline float theta_of(float k) { return (3.14159265358979323846f * k) / 180.0f; }

static float *g_var_sum_out = NULL;   /* optional: the sums under the root (set through orc_gabor_bank_sums) */

void orc_gabor_bank(const float *bank /*[180][17][17]*/, const float *img, int H, int W, int32_t *orient,
                    float *conf, float *var_out) {
    float vmax = 0.0f;
#pragma omp parallel for schedule(static) reduction(max : vmax)
    for (int y = 0; y < H; ++y) {
        float R[NK];
        for (int x = 0; x < W; ++x) {
            for (int k = 0; k < NK; ++k) R[k] = 0.0f;
            for (int i = 0; i < KS; ++i) {
                const int gy = y + i - 8;
                for (int j = 0; j < KS; ++j) {
                    const int gx = x + j - 8;
                    const float v = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? img[(size_t)gy * W + gx] : 0.0f;
                    const float *g = bank + i * KS + j;
                    for (int k = 0; k < NK; ++k) R[k] = fmaf(v, g[(size_t)k * KS * KS], R[k]);
                }
            }
            float M = fabsf(R[0]);
            int b = 0;
            for (int k = 1; k < NK; ++k) {
                float r = fabsf(R[k]);
                if (r > M) {
                    M = r;
                    b = k;
                }
            }
            const float PI_F = 3.14159265358979323846f;
            const float bh = theta_of((float)b);
            float a0 = 0.f, a1 = 0.f;
            for (int k = 0; k < NK; ++k) {
                if (k > 0 && (k & 15) == 0) {
                    a1 = a1 + a0;
                    a0 = 0.f;
                }
                float t1 = bh - theta_of((float)k);
                float d = fminf(fabsf(t1), fminf(fabsf(t1 - PI_F), fabsf(t1 + PI_F)));
                float rd = fabsf(R[k]) - M;
                a0 = a0 + (d * rd) * rd;
            }
            const float var = sqrtf(a0 + a1);
            if (g_var_sum_out) g_var_sum_out[(size_t)y * W + x] = a0 + a1;
            var_out[(size_t)y * W + x] = var;
            orient[(size_t)y * W + x] = (var > 0.0f) ? b : 0;
            if (var > vmax) vmax = var;
        }
    }
    for (size_t i = 0; i < (size_t)H * W; ++i) {
        float v = var_out[i] / vmax;
        v = (v - 0.0f) / 0.2f;
        conf[i] = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
    }
}

/* the same, also returning the sum under the root of every pixel (not thread-safe with respect to other callers) */
void orc_gabor_bank_sums(const float *bank, const float *img, int H, int W, int32_t *orient, float *conf, float *var_out,
                         float *var_sum_out) {
    g_var_sum_out = var_sum_out;
    orc_gabor_bank(bank, img, H, W, orient, conf, var_out);
    g_var_sum_out = NULL;
}
