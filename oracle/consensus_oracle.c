/*
 * oracle/consensus_oracle.c -- TEST INFRASTRUCTURE ONLY (see pmvo_oracle.c).
 *
 * compute_points_similarity (/root/reference/Utils/PMVO_utils.py:366-382): the medoid of a group of 3D
 * directions under |cos| similarity, argmax_k mean_j max(cos(o_k,o_j), cos(-o_k,o_j)), self term included,
 * first maximum wins (torch.argmax).  cos as torch.cosine_similarity evaluates it: vectors normalised by
 * max(|x|, 1e-8) with |x| = sqrt of an fma chain, products rounded separately and added left to right.
 * The mean adds the K similarities left to right and divides by K.  (ATen's vectorised inner-dim sum uses a
 * lane-strided order for K >= 16; the argmax is insensitive to it except for exact near-ties at the 1e-7
 * level -- tests/test_consensus.py reports the agreement with the reference's goldens.)
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

/* one group of K directions -> index of the medoid */
static int medoid_one(const float *ori, int K) {
    float *u = (float *)malloc(sizeof(float) * 3 * (size_t)K);
    for (int k = 0; k < K; ++k) unit3(ori + 3 * k, u + 3 * k);
    int best = 0;
    float bv = 0.f;
    for (int k = 0; k < K; ++k) {
        float acc = 0.0f;
        for (int j = 0; j < K; ++j) {
            float cs = (u[3 * k] * u[3 * j] + u[3 * k + 1] * u[3 * j + 1]) + u[3 * k + 2] * u[3 * j + 2];
            acc = acc + fabsf(cs);
        }
        const float mean = acc / (float)K;
        /* torch.argmax: NaN is the maximum, the first occurrence wins */
        if (k == 0) {
            bv = mean;
        } else if (bv == bv && (mean != mean || mean > bv)) {
            bv = mean;
            best = k;
        }
    }
    free(u);
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
