/*
 * oracle/consensus_oracle.c -- TEST INFRASTRUCTURE ONLY (see pmvo_oracle.c).
 *
 * compute_points_similarity (/root/reference/Utils/PMVO_utils.py:366-382): the medoid of a group of 3D
 * directions under |cos| similarity, argmax_k mean_j max(cos(o_k,o_j), cos(-o_k,o_j)), self term included,
 * first maximum wins (torch.argmax).  cos as torch.cosine_similarity evaluates it: vectors normalised by
 * max(|x|, 1e-8) with |x| = sqrt of an fma chain, products rounded separately and added left to right.
 * The mean is ATen's sum over the innermost (contiguous) dimension divided by K, in ATen's order
 * (aten/src/ATen/native/cpu/SumKernel.cpp, probed against torch 2.10 on this container's CPU, where the sum
 * kernel runs its AVX2 build: 8 floats per vector):
 *   K >= 8: vectorized_inner_sum -- the K/8 full vectors are dealt round-robin to 4 vector accumulators
 *           (row_sum, ilp_factor 4) through the cascade of multi_row_sum (16 "rows" of 4 vectors per level-0
 *           block); vectors left over after the last full group of 4 go to accumulator 0; the 4 accumulators
 *           are added 0+1+2+3; then a scalar starts at 0, takes the K%8 tail elements in order and finally the
 *           8 lanes of the vector accumulator in order;
 *   K < 8:  scalar_inner_sum -- the same row_sum on single floats: 4 accumulators, leftover into accumulator 0,
 *           then 0+1+2+3.
 * tests/test_oracle_more.py demands 100 % agreement with the reference's goldens (consensus.npz).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static void unit3(const float *x, float *o) {
    float s = x[0] * x[0];
    s = fmaf(x[1], x[1], s);
    s = fmaf(x[2], x[2], s);
    float nrm = sqrtf(s);
    if (nrm < 1e-8f) nrm = 1e-8f;
    o[0] = x[0] / nrm;
    o[1] = x[1] / nrm;
    o[2] = x[2] / nrm;
}

/* multi_row_sum over `size` rows of NR floats (row r = x[r*NR .. r*NR+NR)), ATen's 4-level cascade */
#define ORC_MAXNR 32
static void aten_multi_row_sum(const float *x, int size, int NR, float *out) {
    int level_power = 4, lg = 0;
    while ((1 << lg) < size) ++lg;          /* CeilLog2(size) */
    if (lg / 4 > level_power) level_power = lg / 4;
    const int level_step = 1 << level_power, level_mask = level_step - 1;
    float acc[4][ORC_MAXNR];
    for (int j = 0; j < 4; ++j)
        for (int k = 0; k < NR; ++k) acc[j][k] = 0.0f;
    int i = 0;
    while (i + level_step <= size) {
        for (int j = 0; j < level_step; ++j, ++i)
            for (int k = 0; k < NR; ++k) acc[0][k] = acc[0][k] + x[(size_t)i * NR + k];
        for (int j = 1; j < 4; ++j) {
            for (int k = 0; k < NR; ++k) {
                acc[j][k] = acc[j][k] + acc[j - 1][k];
                acc[j - 1][k] = 0.0f;
            }
            if ((i & (level_mask << (j * level_power))) != 0) break;
        }
    }
    for (; i < size; ++i)
        for (int k = 0; k < NR; ++k) acc[0][k] = acc[0][k] + x[(size_t)i * NR + k];
    for (int j = 1; j < 4; ++j)
        for (int k = 0; k < NR; ++k) acc[0][k] = acc[0][k] + acc[j][k];
    for (int k = 0; k < NR; ++k) out[k] = acc[0][k];
}

/* torch.sum of K contiguous floats along the innermost dimension (see the header) */
float orc_aten_inner_sum(const float *x, int K) {
    const int VL = (K >= 8) ? 8 : 1;
    const int vec_size = K / VL, size_ilp = vec_size / 4;
    float ps[ORC_MAXNR];
    aten_multi_row_sum(x, size_ilp, 4 * VL, ps);
    for (int i = size_ilp * 4; i < vec_size; ++i)
        for (int l = 0; l < VL; ++l) ps[l] = ps[l] + x[(size_t)i * VL + l];
    for (int k = 1; k < 4; ++k)
        for (int l = 0; l < VL; ++l) ps[l] = ps[l] + ps[k * VL + l];
    if (VL == 1) return ps[0];
    float fin = 0.0f;
    for (int k = vec_size * VL; k < K; ++k) fin = fin + x[k];
    for (int l = 0; l < VL; ++l) fin = fin + ps[l];
    return fin;
}

/* one group of K directions -> index of the medoid */
static int medoid_one(const float *ori, int K) {
    float *u = (float *)malloc(sizeof(float) * 3 * (size_t)K);
    float *row = (float *)malloc(sizeof(float) * (size_t)K);
    for (int k = 0; k < K; ++k) unit3(ori + 3 * k, u + 3 * k);
    int best = 0;
    float bv = 0.f;
    for (int k = 0; k < K; ++k) {
        for (int j = 0; j < K; ++j) {
            float cs = (u[3 * k] * u[3 * j] + u[3 * k + 1] * u[3 * j + 1]) + u[3 * k + 2] * u[3 * j + 2];
            row[j] = fabsf(cs);
        }
        const float mean = orc_aten_inner_sum(row, K) / (float)K;
        /* torch.argmax: NaN is the maximum, the first occurrence wins */
        if (k == 0) {
            bv = mean;
        } else if (bv == bv && (mean != mean || mean > bv)) {
            bv = mean;
            best = k;
        }
    }
    free(u);
    free(row);
    return best;
}

void orc_medoid_dense(const float *ori, int G, int K, float *out, int32_t *out_index) {
#pragma omp parallel for schedule(static)
    for (int g = 0; g < G; ++g) {
        const float *o = ori + (size_t)g * K * 3;
        int b = medoid_one(o, K);
        out[3 * g] = o[3 * b];
        out[3 * g + 1] = o[3 * b + 1];
        out[3 * g + 2] = o[3 * b + 2];
        if (out_index) out_index[g] = b;
    }
}

void orc_medoid_segmented(const float *ori, const int32_t *seg_start, int G, float *out, int32_t *out_index) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int g = 0; g < G; ++g) {
        const int K = seg_start[g + 1] - seg_start[g];
        if (K <= 0) continue;
        const float *o = ori + (size_t)seg_start[g] * 3;
        int b = medoid_one(o, K);
        out[3 * g] = o[3 * b];
        out[3 * g + 1] = o[3 * b + 1];
        out[3 * g + 2] = o[3 * b + 2];
        if (out_index) out_index[g] = b;
    }
}

/* refine's replacement rule (/root/reference/PMVO.py:631-636):
 *     similar = maximum(cosine_similarity(center, ori), cosine_similarity(center, -ori));  ori[similar < thr] = center
 * on [N,3] float32 tensors.  cosine_similarity(center, -ori) is the exact negation of cosine_similarity(center, ori)
 * (negation commutes with every rounding involved), so the maximum is |cos|.  Returns the number of replaced rows.
 * Pinned by tests/golden/e2e_multichunk.npz through oracle.refine_loop (tests/test_oracle_more.py). */
int orc_replace_dissimilar(const float *center, float *ori, float thr, int N) {
    int replaced = 0;
    for (int n = 0; n < N; ++n) {
        float cu[3], ou[3];
        unit3(center + 3 * n, cu);
        unit3(ori + 3 * n, ou);
        const float cs = (cu[0] * ou[0] + cu[1] * ou[1]) + cu[2] * ou[2];
        if (fabsf(cs) < thr) {
            ori[3 * n] = center[3 * n];
            ori[3 * n + 1] = center[3 * n + 1];
            ori[3 * n + 2] = center[3 * n + 2];
            ++replaced;
        }
    }
    return replaced;
}
