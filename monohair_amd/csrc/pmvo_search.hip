// pmvo_search.hip -- the fused loss search of PMVO.forward (PMVO.py:50-78), gfx950 only.
//
// One workgroup = one candidate 3D point.  Its nrank*S (= 10*90 = 900) candidate segment end-points
// ("items") are spread over the lanes, K items per lane, and live in registers for the whole kernel:
//   sample_next_3d_pos (PMVO.py:263-335)      -> item position X            (once, prologue)
//   compute_reproject_ori (PMVO.py:219-241)    -> unit direction d_hat per view (registers)
//   compute_prj_loss (PMVO.py:151-209)         -> masked min over the taps of the patch, weighted
//                                                 sums over views in ATen's cascade order
//   forward's best-so-far update (PMVO.py:57-70) -> LDS epilogue.
// No [V,N,S] tensor ever exists.  The patch of (view, point) is the same for every lane of the
// workgroup, so the tap list (pmvo_project.hip: mh_prep_taps_kernel) is read with wave-uniform loads (header
// and first tap: scalar loads; the rest: broadcast vector loads, what the backend selects for the ping-pong
// groups) and the inner loop is pure VALU: per (item, tap) 2 mul + add + (1-|x|) + cmp + 2 cndmask in the portable kernel and
// the select body of the shipped one; its key body keeps the running (loss, tap) minimum as one integer key per item and
// needs 2 mul + add + sub + lshl_or + half a min3 (see mh_tap_key).
// Views in which the point is not visible (vis == -1 => weight 0, PMVO.py:212) are skipped: adding
// their exact zeros would not change any sum.
#include "mh_device.h"

#define MH_MAX_ITEMS 1024
#define MH_MAX_RANKS 16

// PMVO.sample_next_3d_pos for one item: pixel(unrounded) + 2*(ori_col, ori_row) -> ndc -> unproject
__device__ __forceinline__ void mh_sample_next(const float *__restrict__ cam, float X0, float X1, float X2,
                                               float ori_r, float ori_c, float Hf, float Wf, float off, float &S0,
                                               float &S1, float &S2, int forms = 0) {
    float u, v, z, row, col;
    if (forms & MH_FORM_GEMV) mh_cam_project_single(cam, X0, X1, X2, u, v, z);
    else mh_cam_project(cam, X0, X1, X2, u, v, z);
    mh_ndc_to_pixel(u, v, Hf, Wf, row, col);
    float nx = col + ori_c * 2.0f;
    float ny = row + ori_r * 2.0f;
    nx = nx / Wf;
    ny = ny / Hf;
    nx = nx * 2.0f - 1.0f;
    ny = ny * 2.0f - 1.0f;
    nx = -nx;
    mh_cam_unproject(cam, nx, ny, z + off, S0, S1, S2, (forms & MH_FORM_CHAIN) != 0);
}

// mh_sample_next split at what the S samples of one base-view rank have in common (the point's pixel in the base view, the
// shifted pixel in NDC, the two quotients of mh_cam_unproject) and what is per sample (depth + offset onwards).  Same
// operations on the same values in the same order, so rank part + item part == mh_sample_next bit for bit; the shipped
// search evaluates the rank part once per (point, rank) instead of once per item (90 times) and keeps it in LDS.
// rec[16] = { A, B, z, t0 | t1, t2, Ri0, Ri1 | Ri2 .. Ri5 | Ri6, Ri7, Ri8, forms }
// forms (mh_group_forms): how the sgemms of this (rank, base view) group round in the reference -- MH_FORM_GEMV here,
// MH_FORM_CHAIN in the item part.
__device__ __forceinline__ void mh_sample_rank(const float *__restrict__ cam, float X0, float X1, float X2, float ori_r,
                                               float ori_c, float Hf, float Wf, float *__restrict__ rec, int forms = 0) {
    float u, v, z, row, col;
    if (forms & MH_FORM_GEMV) mh_cam_project_single(cam, X0, X1, X2, u, v, z);
    else mh_cam_project(cam, X0, X1, X2, u, v, z);
    mh_ndc_to_pixel(u, v, Hf, Wf, row, col);
    float nx = col + ori_c * 2.0f;
    float ny = row + ori_r * 2.0f;
    nx = nx / Wf;
    ny = ny / Hf;
    nx = nx * 2.0f - 1.0f;
    ny = ny * 2.0f - 1.0f;
    nx = -nx;
    rec[0] = (nx - cam[18]) / cam[16];
    rec[1] = (ny - cam[22]) / cam[21];
    rec[2] = z;
    rec[3] = cam[3];
    rec[4] = cam[7];
    rec[5] = cam[11];
#pragma unroll
    for (int i = 0; i < 9; ++i) rec[6 + i] = cam[32 + i];
    rec[15] = __int_as_float(forms);
}

__device__ __forceinline__ void mh_sample_item(const float4 *__restrict__ rec, float off, float &S0, float &S1, float &S2) {
    const float4 a = rec[0], b = rec[1], c = rec[2], d = rec[3];
    const float z = a.z + off;
    const float c0 = a.x * z, c1 = a.y * z;
    const float d0 = c0 - a.w, d1 = c1 - b.x, d2 = z - b.y;
    if (__float_as_int(d.w) & MH_FORM_CHAIN) {   // (prologue / epilogue only: once per item)
        S0 = mh_fma(c.x, d2, mh_fma(b.w, d1, b.z * d0));
        S1 = mh_fma(c.w, d2, mh_fma(c.z, d1, c.y * d0));
        S2 = mh_fma(d.z, d2, mh_fma(d.y, d1, d.x * d0));
    } else {
        S0 = (b.z * d0 + c.x * d2) + b.w * d1;
        S1 = (c.y * d0 + c.w * d2) + c.z * d1;
        S2 = (d.x * d0 + d.z * d2) + d.y * d1;
    }
}

// where the trailing columns of the batch's [V, N*S] sums fall in point n: its first trailing sample (S if none)
__device__ __forceinline__ int mh_tail_from(const MhRule &rule, int n, int S) {
    const long long c0 = (long long)n * S;
    if (c0 + S <= rule.tail_col0) return S;
    return c0 >= rule.tail_col0 ? 0 : (int)(rule.tail_col0 - c0);
}

// The weighted sums over the views of ONE candidate (item position X) of point n in ATen's row_sum order (mh_device.h:
// mh_row_sum_views) -- for the trailing columns of the batch's [V, N*S] sums.  The per-view terms are evaluated as the
// portable kernel evaluates them (same operations as the shipped bodies, bit for bit): tap lists from the scratch records,
// views that do not see the point (list length 0) add +0; cnt = the number of views with a positive weight.  It runs for a few
// dozen items per launch, after the view loops, from the item's rank record in LDS -- nothing of it is live in those loops
// (inside mh_search_slices_lds the same code cost the hot kernel 20 spilled registers).
__device__ __forceinline__ void mh_tail_item_sums(const float *__restrict__ cams, int V, float Hf, float Wf,
                                                  const float4 *__restrict__ taps_n, size_t vstride,
                                                  const uint8_t *__restrict__ vcnt_n, int N, float X0, float X1, float X2,
                                                  float &nm_out, float &dn_out, int &cnt_out) {
    int cnt = 0;
    auto term = [&](int v, float &tn, float &td) {
        tn = td = 0.0f;
        const int ntap = vcnt_n ? (int)vcnt_n[(size_t)v * N] : -1;
        if (ntap == 0) return;
        const float4 *__restrict__ rec = taps_n + (size_t)v * vstride;
        const float4 hdr = rec[0];
        if (hdr.y == -1.0f) return;
        const int nt = ntap > 0 ? ntap : __float_as_int(hdr.x);
        float row, col, dx, dy;
        mh_pixel_of(cams + v * MH_CAM_STRIDE, X0, X1, X2, Hf, Wf, row, col);
        mh_unit2(row - hdr.z, col - hdr.w, dx, dy);
        const float4 t0 = rec[1];
        float ml = 1.0f - __builtin_fabsf(t0.x * dx + t0.y * dy), bc = t0.z;
#pragma unroll 4
        for (int t = 1; t < nt; ++t) {
            const float4 tp = rec[1 + t];
            const float l = 1.0f - __builtin_fabsf(tp.x * dx + tp.y * dy);
            const bool upd = l < ml;
            ml = upd ? l : ml;
            bc = upd ? tp.z : bc;
        }
        tn = ml * bc;
        td = bc;
        cnt += (bc > 0.0f) ? 1 : 0;
    };
    const int L = V >> 2;
    float pn[4][3] = {}, pd[4][3] = {};   // partial k (rows k, k+4, ...): cascade levels 0, 1, 2
    for (int i = 0; i < L; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float tn, td;
            term(4 * i + k, tn, td);
            pn[k][0] = pn[k][0] + tn;
            pd[k][0] = pd[k][0] + td;
        }
        if (((i + 1) & 15) == 0) {   // a full block of 16 rows per partial: level 0 -> 1, every 256 rows level 1 -> 2
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                pn[k][1] = pn[k][1] + pn[k][0];
                pn[k][0] = 0.0f;
                pd[k][1] = pd[k][1] + pd[k][0];
                pd[k][0] = 0.0f;
                if (((i + 1) & 0xF0) == 0) {
                    pn[k][2] = pn[k][2] + pn[k][1];
                    pn[k][1] = 0.0f;
                    pd[k][2] = pd[k][2] + pd[k][1];
                    pd[k][1] = 0.0f;
                }
            }
        }
    }
    float sn[4], sd[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sn[k] = (pn[k][0] + pn[k][1]) + pn[k][2];
        sd[k] = (pd[k][0] + pd[k][1]) + pd[k][2];
    }
    for (int v = L * 4; v < V; ++v) {
        float tn, td;
        term(v, tn, td);
        sn[0] = sn[0] + tn;
        sd[0] = sd[0] + td;
    }
    nm_out = ((sn[0] + sn[1]) + sn[2]) + sn[3];
    dn_out = ((sd[0] + sd[1]) + sd[2]) + sd[3];
    cnt_out = cnt;
}

// torch.min over a row with NaN propagation: NaN beats numbers, first index wins among equals
__device__ __forceinline__ bool mh_min_better(float al, int ai, float bl, int bi) {
    const bool an = al != al, bn = bl != bl;
    if (an || bn) return (an && bn) ? (ai < bi) : an;
    return (al < bl) || (al == bl && ai < bi);
}

// 1 - |x| as ONE instruction (abs is a source modifier); kept out of the SLP vectoriser's reach
__device__ __forceinline__ float mh_one_minus_abs(float x) {
    float r;
    asm("v_sub_f32_e64 %0, 1.0, |%1|" : "=v"(r) : "v"(x));
    return r;
}

// mh_search_kernel -- the PORTABLE form of the fused loss search: plain C++ loops over views and taps, the compiler's
// schedule, tap lists read from the scratch records.  It is the cross-check of the shipped mh_search3_kernel (every test that
// compares the search with the oracle runs both: search_variant 1256) and what runs when no list lengths are available.
// (Rounds 1-2 carried this kernel's hand-shaped FAST forms and mh_search2_kernel as variants; they are gone: three
// generations of one arithmetic contract were two too many to keep in step.)
template <int K, int T>
__global__ __launch_bounds__(T) void mh_search_kernel(MhViews vw, const float *__restrict__ offs, int S, int nrank,
                                                      int rank_step, const float *__restrict__ pts, int N, int P1,
                                                      float thr, const float *__restrict__ ori_c,
                                                      const int32_t *__restrict__ base_idx,
                                                      const float *__restrict__ base_val,
                                                      const float4 *__restrict__ taps, float *__restrict__ line_ori,
                                                      float *__restrict__ min_loss, uint8_t *__restrict__ high_conf,
                                                      float *__restrict__ best_sample, int32_t *__restrict__ best_rank,
                                                      int32_t *__restrict__ best_s, MhRule rule) {
    __shared__ float s_loss[MH_MAX_ITEMS];
    __shared__ uint8_t s_pos[MH_MAX_ITEMS];
    __shared__ float s_rl[MH_MAX_RANKS];
    __shared__ int s_ri[MH_MAX_RANKS];
    __shared__ int s_rh[MH_MAX_RANKS];

    const int n = blockIdx.x, tid = threadIdx.x;
    const int nitems = nrank * S;
    const int V = vw.V;
    const float Hf = (float)vw.H, Wf = (float)vw.W;
    const float P0 = pts[3 * n], P1x = pts[3 * n + 1], P2 = pts[3 * n + 2];

    float X0[K], X1[K], X2[K];
    MhCascV num[K], den[K];
    int cnt[K];
    const int tail_from = mh_tail_from(rule, n, S);   // first trailing sample of the batch's [V, N*S] sums in this point
#pragma unroll
    for (int j = 0; j < K; ++j) {
        int it = j * T + tid;
        it = it < nitems ? it : 0;
        const int r = it / S, s = it - r * S;
        // (ranks with base_val <= 0 are unusable and their indices may be anything: clamped, as in mh_search3_kernel)
        const int b = min(max(base_idx[(size_t)(r * rank_step) * N + n], 0), V - 1);
        const float2 oc = reinterpret_cast<const float2 *>(ori_c)[(size_t)b * N + n];
        mh_sample_next(vw.cams + b * MH_CAM_STRIDE, P0, P1x, P2, oc.x, oc.y, Hf, Wf, offs[s], X0[j], X1[j], X2[j],
                       mh_group_forms(rule, r, V, b, S));
        num[j] = den[j] = MhCascV{0.0f, 0.0f, 0.0f};
        cnt[j] = 0;
    }

    for (int v = 0; v < V; ++v) {
        if (v > 0 && (v & 15) == 0) {
#pragma unroll
            for (int j = 0; j < K; ++j) {
                mh_cascv_flush(num[j], v);
                mh_cascv_flush(den[j], v);
            }
        }
        const float4 *__restrict__ rec = taps + ((size_t)v * N + n) * P1;
        const float4 hdr = rec[0];
        if (hdr.y == -1.0f) continue;   // uniform: point not visible in this view, weight 0
        const int ntap = __float_as_int(hdr.x);
        const float *__restrict__ cam = vw.cams + v * MH_CAM_STRIDE;
        const float4 t0 = rec[1];
        {
            float dx[K], dy[K], ml[K], bc[K];
#pragma unroll
            for (int j = 0; j < K; ++j) {
                float row, col;
                mh_pixel_of(cam, X0[j], X1[j], X2[j], Hf, Wf, row, col);
                mh_unit2(row - hdr.z, col - hdr.w, dx[j], dy[j]);
                const float cs = t0.x * dx[j] + t0.y * dy[j];
                ml[j] = 1.0f - __builtin_fabsf(cs);
                bc[j] = t0.z;
            }
            for (int t = 1; t < ntap; ++t) {
                const float4 tp = rec[1 + t];
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const float cs = tp.x * dx[j] + tp.y * dy[j];
                    const float l = 1.0f - __builtin_fabsf(cs);
                    const bool upd = l < ml[j];
                    ml[j] = upd ? l : ml[j];
                    bc[j] = upd ? tp.z : bc[j];
                }
            }
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const float w = bc[j];   // (vis != -1) * best_conf
                num[j].a0 = num[j].a0 + ml[j] * w;
                den[j].a0 = den[j].a0 + w;
                cnt[j] += (w > 0.0f) ? 1 : 0;
            }
        }
    }

    // ---- per-sample loss and "positive" flag (PMVO.py:198-201)
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const int it = j * T + tid;
        if (it < nitems) {
            float dn = mh_cascv_done(den[j]);
            float nm = mh_cascv_done(num[j]);
            int cn = cnt[j];
            if (it - (it / S) * S >= tail_from)   // a trailing column: the same terms in ATen's row_sum order
                mh_tail_item_sums(vw.cams, V, Hf, Wf, taps + (size_t)n * P1, (size_t)N * P1, nullptr, N, X0[j], X1[j], X2[j],
                                  nm, dn, cn);
            const float ratio = dn / (float)cn;
            s_pos[it] = (ratio > thr) ? 1 : 0;
            s_loss[it] = nm / dn;
        }
    }
    __syncthreads();

    // ---- per rank: low-confidence escape hatch, min / argmin over the S samples (PMVO.py:199-206)
    const int wave = tid >> 6, lane = tid & 63, nwaves = T >> 6;
    for (int r = wave; r < nrank; r += nwaves) {
        int npos = 0;
        for (int s0 = 0; s0 < S; s0 += MH_WAVE) {
            const int s = s0 + lane;
            npos += __popcll(__ballot(s < S && s_pos[r * S + s]));
        }
        const bool low = npos < 5;
        float bl = 0.0f;
        int bi = 0x7fffffff;
        for (int s = lane; s < S; s += MH_WAVE) {
            float l = s_loss[r * S + s];
            if (!low && !s_pos[r * S + s]) l = 1.0f;
            if (bi == 0x7fffffff || mh_min_better(l, s, bl, bi)) {
                bl = l;
                bi = s;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ol = __shfl_xor(bl, o);
            const int oi = __shfl_xor(bi, o);
            if (oi != 0x7fffffff && (bi == 0x7fffffff || mh_min_better(ol, oi, bl, bi))) {
                bl = ol;
                bi = oi;
            }
        }
        if (lane == 0) {
            s_rl[r] = bl;
            s_ri[r] = bi;
            s_rh[r] = s_pos[r * S + bi];
        }
    }
    __syncthreads();

    // ---- best candidate across base-view ranks (PMVO.py:57-70) and the 3D direction (:73-74)
    if (tid == 0) {
        const float Hf = (float)vw.H, Wf = (float)vw.W;
        float ml = s_rl[0];
        int br = 0, bs = s_ri[0], hc = s_rh[0];
        for (int r = 1; r < nrank; ++r) {
            const float l = s_rl[r];
            if ((l < ml) && (base_val[(size_t)(r * rank_step) * N + n] > 0.0f)) {
                ml = l;
                br = r;
                bs = s_ri[r];
                hc = s_rh[r];
            }
        }
        const int b = min(max(base_idx[(size_t)(br * rank_step) * N + n], 0), V - 1);
        const float2 oc = reinterpret_cast<const float2 *>(ori_c)[(size_t)b * N + n];
        float B0, B1, B2;
        mh_sample_next(vw.cams + b * MH_CAM_STRIDE, P0, P1x, P2, oc.x, oc.y, Hf, Wf, offs[bs], B0, B1, B2,
                       mh_group_forms(rule, br, V, b, S));
        const float d0 = B0 - P0, d1 = B1 - P1x, d2 = B2 - P2;
        float s2 = d0 * d0;
        s2 = mh_fma(d1, d1, s2);
        s2 = mh_fma(d2, d2, s2);
        const float nrm = __builtin_sqrtf(s2);
        line_ori[3 * n] = d0 / nrm;
        line_ori[3 * n + 1] = d1 / nrm;
        line_ori[3 * n + 2] = d2 / nrm;
        min_loss[n] = ml;
        high_conf[n] = (uint8_t)hc;
        if (best_sample) {
            best_sample[3 * n] = B0;
            best_sample[3 * n + 1] = B1;
            best_sample[3 * n + 2] = B2;
        }
        if (best_rank) best_rank[n] = br;
        if (best_s) best_s[n] = bs;
    }
}

// ---------------------------------------------------------------------------------------------
// Building blocks of the shipped search (mh_search3_kernel below).  What rounds 1-2 measured about the tap body
// (tools/ubench/valu2.hip, valu3.hip): plain v_mul/v_mul/v_add/v_sub are full-rate VALU operations (~2 cycles per
// wave-instruction) and overlap with the half-rate v_cmp / v_cndmask, the packed v_pk_mul/v_pk_add do not (82 -> 69 cycles
// per tap per 4 items); candidates of base-view ranks that can never be taken are not evaluated (ranks > 0 only replace the
// best-so-far when base_view_conf[rank] > 0, PMVO.py:64, and the ranking is sorted by that value, so the usable ranks are a
// prefix); every wave evaluates only the item slices it has (900 items = 15 wave-slices, not 16); workgroups take the points
// in descending order of work (order[], mh_search_order_kernel), so the tail of the launch is made of cheap points.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float mh_vmul(float a, float b) {
    float r;
    asm("v_mul_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float mh_vadd(float a, float b) {
    float r;
    asm("v_add_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// running minimum over the taps with first-index ties (strict '<', PMVO.py:177): compares of all items first, then the
// selects (gfx950 wants 2 wait states between a VALU write of an SGPR mask and a VALU read of it)
template <int KA>
__device__ __forceinline__ void mh_tap_update(float (&ML)[KA], float (&BC)[KA], const float (&l)[KA], float cf) {
    if constexpr (KA == 4) {
        unsigned long long m0, m1, m2, m3;
        asm("v_cmp_lt_f32_e64 %[m0], %[l0], %[a0]\n\t"
            "v_cmp_lt_f32_e64 %[m1], %[l1], %[a1]\n\t"
            "v_cmp_lt_f32_e64 %[m2], %[l2], %[a2]\n\t"
            "v_cmp_lt_f32_e64 %[m3], %[l3], %[a3]\n\t"
            "v_cndmask_b32_e64 %[a0], %[a0], %[l0], %[m0]\n\t"
            "v_cndmask_b32_e64 %[a1], %[a1], %[l1], %[m1]\n\t"
            "v_cndmask_b32_e64 %[a2], %[a2], %[l2], %[m2]\n\t"
            "v_cndmask_b32_e64 %[a3], %[a3], %[l3], %[m3]\n\t"
            "v_cndmask_b32_e64 %[b0], %[b0], %[cf], %[m0]\n\t"
            "v_cndmask_b32_e64 %[b1], %[b1], %[cf], %[m1]\n\t"
            "v_cndmask_b32_e64 %[b2], %[b2], %[cf], %[m2]\n\t"
            "v_cndmask_b32_e64 %[b3], %[b3], %[cf], %[m3]"
            : [a0] "+v"(ML[0]), [a1] "+v"(ML[1]), [a2] "+v"(ML[2]), [a3] "+v"(ML[3]), [b0] "+v"(BC[0]),
              [b1] "+v"(BC[1]), [b2] "+v"(BC[2]), [b3] "+v"(BC[3]), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2),
              [m3] "=&s"(m3)
            : [l0] "v"(l[0]), [l1] "v"(l[1]), [l2] "v"(l[2]), [l3] "v"(l[3]), [cf] "v"(cf));
    } else if constexpr (KA == 3) {
        unsigned long long m0, m1, m2;
        asm("v_cmp_lt_f32_e64 %[m0], %[l0], %[a0]\n\t"
            "v_cmp_lt_f32_e64 %[m1], %[l1], %[a1]\n\t"
            "v_cmp_lt_f32_e64 %[m2], %[l2], %[a2]\n\t"
            "v_cndmask_b32_e64 %[a0], %[a0], %[l0], %[m0]\n\t"
            "v_cndmask_b32_e64 %[a1], %[a1], %[l1], %[m1]\n\t"
            "v_cndmask_b32_e64 %[a2], %[a2], %[l2], %[m2]\n\t"
            "v_cndmask_b32_e64 %[b0], %[b0], %[cf], %[m0]\n\t"
            "v_cndmask_b32_e64 %[b1], %[b1], %[cf], %[m1]\n\t"
            "v_cndmask_b32_e64 %[b2], %[b2], %[cf], %[m2]"
            : [a0] "+v"(ML[0]), [a1] "+v"(ML[1]), [a2] "+v"(ML[2]), [b0] "+v"(BC[0]), [b1] "+v"(BC[1]),
              [b2] "+v"(BC[2]), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2)
            : [l0] "v"(l[0]), [l1] "v"(l[1]), [l2] "v"(l[2]), [cf] "v"(cf));
    } else if constexpr (KA == 2) {
        unsigned long long m0, m1;
        asm("v_cmp_lt_f32_e64 %[m0], %[l0], %[a0]\n\t"
            "v_cmp_lt_f32_e64 %[m1], %[l1], %[a1]\n\t"
            "s_nop 0\n\t"
            "v_cndmask_b32_e64 %[a0], %[a0], %[l0], %[m0]\n\t"
            "v_cndmask_b32_e64 %[a1], %[a1], %[l1], %[m1]\n\t"
            "v_cndmask_b32_e64 %[b0], %[b0], %[cf], %[m0]\n\t"
            "v_cndmask_b32_e64 %[b1], %[b1], %[cf], %[m1]"
            : [a0] "+v"(ML[0]), [a1] "+v"(ML[1]), [b0] "+v"(BC[0]), [b1] "+v"(BC[1]), [m0] "=&s"(m0), [m1] "=&s"(m1)
            : [l0] "v"(l[0]), [l1] "v"(l[1]), [cf] "v"(cf));
    } else {
#pragma unroll
        for (int j = 0; j < KA; ++j) {
            const bool upd = l[j] < ML[j];
            ML[j] = upd ? l[j] : ML[j];
            BC[j] = upd ? cf : BC[j];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The running minimum as ONE integer key per item (round 4; tools/ubench/valu5.hip: 69 -> 54 cycles per tap per 4 items).
// compute_prj_loss keeps, per (item, view), the lexicographic minimum of (loss_t, t) over the taps of the list -- tap 0
// seeds, a later tap replaces it only on a strictly smaller loss (PMVO.py:173-182) -- and wants the loss and the confidence
// of that tap.  loss_t = fl(1 - x), x = |cs_t|.  With C = 1 + 2^-14 the value t' = fl(C - x) is
//   * exactly loss_t + 2^-14 for 2^-14 < x: for x >= 0.5 both subtractions are exact (x is a multiple of 2^-24 there,
//     and |cs| <= 1 + 7 * 2^-24 for two vectors normalised in fp32, so t' > 0 where the loss itself goes negative); for
//     x < 0.5 both results lie in [0.5, 1), where the grid is 2^-24 and round-to-nearest-even commutes with adding the
//     even multiple 2^10 of the grid;
//   * a positive float in [2^-15, 2), whose bits all start 0b00111: (bits << 5) drops only those constant bits and is
//     monotone in t', which leaves five bits for a tap index.
// A list is walked in groups of 64 taps (one group for every patch up to 8 x 8); the even taps of a group fold into one
// accumulator, the odd taps into a second one, key = (bits(t') << 5) | (place of the tap among them, 0..31), two taps per
// v_min3_u32.  The group's first-index minimum is the odd accumulator's key if it is STRICTLY smaller than the even one's
// (equal loss and place: tap 2i is earlier than 2i + 1; equal loss, smaller place i' < i: 2i' + 1 < 2i), else the even
// one's; its tap = 2 * place + parity.  A later group replaces an earlier one only on a strictly smaller loss.  mul, mul,
// add, sub, lshl_or per evaluation + half a min3: 5.5 instructions instead of 7, two accumulators per item, one compare
// and two selects per (item, view) to merge them.
// What the key cannot state -- a winner with x <= 2^-14 (t' >= 1 lands on the coarser grid of [1, 2)), a NaN -- shows as
// key >= MH_KEY_BAD; a wave that sees one on any of its lanes evaluates that view again with the compare-and-select body
// (mh_tap_update), which is also what runs for one-tap lists and for a NaN seed tap.  Both bodies give the same bits
// wherever the key is valid, so the outputs are those of the select body everywhere.
// ---------------------------------------------------------------------------------------------
#define MH_KEY_C 1.00006103515625f   // 1 + 2^-14
#define MH_KEY_E 6.103515625e-05f    // 2^-14
#define MH_KEY_BAD 0xF0000000u       // key of t' = 1.0 (index 0)
#ifndef MH_KEY_MIN_TAPS
#define MH_KEY_MIN_TAPS 10           // lists up to this length go through the select body directly
#endif
#define MH_KEY_PAD 4                 // lists are padded in LDS to a multiple of this many taps with (0, 0): cs = 0, t' = C
// (wave, view) visits of the key body: [0] all, [1] one-tap / short / NaN-seed lists (select body directly), [2] evaluated
// again with the select body after the key body.  [2] is always counted -- one atomic in a branch the bench scene takes 0 times
// in 95 M -- so that a test can assert that it was inside that branch (tests/test_key_reeval_gpu.py); [0] and [1] sit in the
// hot path and are counted only in the -DMH_KEY_STATS build of tools/exp_key_stats.py.
__device__ unsigned long long mh_key_stats_dev[4];
extern "C" int mh_debug_key_stats(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mh_key_stats_dev), sizeof(unsigned long long) * 4) != hipSuccess) return -1;
    if (reset) {
        const unsigned long long z[4] = {0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(mh_key_stats_dev), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#define MH_KEY_COUNT_ALWAYS(i) do { if ((tid & 63) == 0) atomicAdd(&mh_key_stats_dev[i], 1ull); } while (0)
#ifdef MH_KEY_STATS
#define MH_KEY_COUNT(i) MH_KEY_COUNT_ALWAYS(i)
#else
#define MH_KEY_COUNT(i) do { } while (0)
#endif
__device__ __forceinline__ unsigned mh_tap_key(float cs, int idx) {
    float t;
    unsigned k;
    asm("v_sub_f32_e64 %0, %2, |%1|" : "=v"(t) : "v"(cs), "s"(MH_KEY_C));
    asm("v_lshl_or_b32 %0, %1, 5, %2" : "=v"(k) : "v"(t), "s"(idx));
    return k;
}
__device__ __forceinline__ unsigned mh_min3u(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// Four taps x NI items of the key body as ONE hand-ordered block: per item 16 full-rate (mul, mul, add, sub) + 4 v_lshl_or +
// 2 v_min3 (88 instructions for four items), every result used at least NI - 1 instructions after it is produced.  (Left to
// the compiler as one asm statement per instruction, its hazard recogniser pads every inline-asm result that is read by
// the very next instruction with an s_nop -- it cannot see that no dst_sel is involved: ~9 per block.)  The "memory"
// clobber keeps the LDS reads of the NEXT tap group, issued in front of the block, in front of it.
// (The four bodies below are printed by tools/gen_key_blocks.py.)
// Taps g[0], g[2] (places ib, ib + 1 among the even taps) go into ke, taps g[1], g[3] (the same places among the odd taps)
// into ko; the first NI entries of the caller's arrays are used.  (Padding lists to two taps instead of four, with a
// two-tap block for the tail, was measured: the choice between two asm blocks that update the same registers costs eight
// register copies and a wait for every LDS read per call -- 0.634 instead of 0.582 ms.)
template <int NI, int KN>
__device__ __forceinline__ void mh_key_block4(unsigned (&ke)[KN], unsigned (&ko)[KN], const float2 (&g)[4],
                                              const float (&DX)[KN], const float (&DY)[KN], int ib) {
    static_assert(NI >= 1 && NI <= 4 && NI <= KN, "mh_key_block4: 1..4 items");
    float a[NI], b[NI], c[NI];
    if constexpr (NI == 4) {
        asm volatile(
            "v_mul_f32_e32 %[a0], %[t0x], %[x0]\n\t"
            "v_mul_f32_e32 %[a1], %[t0x], %[x1]\n\t"
            "v_mul_f32_e32 %[a2], %[t0x], %[x2]\n\t"
            "v_mul_f32_e32 %[a3], %[t0x], %[x3]\n\t"
            "v_mul_f32_e32 %[b0], %[t0y], %[y0]\n\t"
            "v_mul_f32_e32 %[b1], %[t0y], %[y1]\n\t"
            "v_mul_f32_e32 %[b2], %[t0y], %[y2]\n\t"
            "v_mul_f32_e32 %[b3], %[t0y], %[y3]\n\t"
            "v_mul_f32_e32 %[c0], %[t2x], %[x0]\n\t"
            "v_mul_f32_e32 %[c1], %[t2x], %[x1]\n\t"
            "v_mul_f32_e32 %[c2], %[t2x], %[x2]\n\t"
            "v_mul_f32_e32 %[c3], %[t2x], %[x3]\n\t"
            "v_add_f32_e32 %[a0], %[a0], %[b0]\n\t"
            "v_add_f32_e32 %[a1], %[a1], %[b1]\n\t"
            "v_add_f32_e32 %[a2], %[a2], %[b2]\n\t"
            "v_add_f32_e32 %[a3], %[a3], %[b3]\n\t"
            "v_mul_f32_e32 %[b0], %[t2y], %[y0]\n\t"
            "v_mul_f32_e32 %[b1], %[t2y], %[y1]\n\t"
            "v_mul_f32_e32 %[b2], %[t2y], %[y2]\n\t"
            "v_mul_f32_e32 %[b3], %[t2y], %[y3]\n\t"
            "v_sub_f32_e64 %[a0], %[cc], |%[a0]|\n\t"
            "v_sub_f32_e64 %[a1], %[cc], |%[a1]|\n\t"
            "v_sub_f32_e64 %[a2], %[cc], |%[a2]|\n\t"
            "v_sub_f32_e64 %[a3], %[cc], |%[a3]|\n\t"
            "v_add_f32_e32 %[c0], %[c0], %[b0]\n\t"
            "v_add_f32_e32 %[c1], %[c1], %[b1]\n\t"
            "v_add_f32_e32 %[c2], %[c2], %[b2]\n\t"
            "v_add_f32_e32 %[c3], %[c3], %[b3]\n\t"
            "v_lshl_or_b32 %[a0], %[a0], 5, %[i0]\n\t"
            "v_lshl_or_b32 %[a1], %[a1], 5, %[i0]\n\t"
            "v_lshl_or_b32 %[a2], %[a2], 5, %[i0]\n\t"
            "v_lshl_or_b32 %[a3], %[a3], 5, %[i0]\n\t"
            "v_sub_f32_e64 %[c0], %[cc], |%[c0]|\n\t"
            "v_sub_f32_e64 %[c1], %[cc], |%[c1]|\n\t"
            "v_sub_f32_e64 %[c2], %[cc], |%[c2]|\n\t"
            "v_sub_f32_e64 %[c3], %[cc], |%[c3]|\n\t"
            "v_lshl_or_b32 %[c0], %[c0], 5, %[i1]\n\t"
            "v_lshl_or_b32 %[c1], %[c1], 5, %[i1]\n\t"
            "v_lshl_or_b32 %[c2], %[c2], 5, %[i1]\n\t"
            "v_lshl_or_b32 %[c3], %[c3], 5, %[i1]\n\t"
            "v_min3_u32 %[e0], %[e0], %[a0], %[c0]\n\t"
            "v_min3_u32 %[e1], %[e1], %[a1], %[c1]\n\t"
            "v_min3_u32 %[e2], %[e2], %[a2], %[c2]\n\t"
            "v_min3_u32 %[e3], %[e3], %[a3], %[c3]\n\t"
            "v_mul_f32_e32 %[a0], %[t1x], %[x0]\n\t"
            "v_mul_f32_e32 %[a1], %[t1x], %[x1]\n\t"
            "v_mul_f32_e32 %[a2], %[t1x], %[x2]\n\t"
            "v_mul_f32_e32 %[a3], %[t1x], %[x3]\n\t"
            "v_mul_f32_e32 %[b0], %[t1y], %[y0]\n\t"
            "v_mul_f32_e32 %[b1], %[t1y], %[y1]\n\t"
            "v_mul_f32_e32 %[b2], %[t1y], %[y2]\n\t"
            "v_mul_f32_e32 %[b3], %[t1y], %[y3]\n\t"
            "v_mul_f32_e32 %[c0], %[t3x], %[x0]\n\t"
            "v_mul_f32_e32 %[c1], %[t3x], %[x1]\n\t"
            "v_mul_f32_e32 %[c2], %[t3x], %[x2]\n\t"
            "v_mul_f32_e32 %[c3], %[t3x], %[x3]\n\t"
            "v_add_f32_e32 %[a0], %[a0], %[b0]\n\t"
            "v_add_f32_e32 %[a1], %[a1], %[b1]\n\t"
            "v_add_f32_e32 %[a2], %[a2], %[b2]\n\t"
            "v_add_f32_e32 %[a3], %[a3], %[b3]\n\t"
            "v_mul_f32_e32 %[b0], %[t3y], %[y0]\n\t"
            "v_mul_f32_e32 %[b1], %[t3y], %[y1]\n\t"
            "v_mul_f32_e32 %[b2], %[t3y], %[y2]\n\t"
            "v_mul_f32_e32 %[b3], %[t3y], %[y3]\n\t"
            "v_sub_f32_e64 %[a0], %[cc], |%[a0]|\n\t"
            "v_sub_f32_e64 %[a1], %[cc], |%[a1]|\n\t"
            "v_sub_f32_e64 %[a2], %[cc], |%[a2]|\n\t"
            "v_sub_f32_e64 %[a3], %[cc], |%[a3]|\n\t"
            "v_add_f32_e32 %[c0], %[c0], %[b0]\n\t"
            "v_add_f32_e32 %[c1], %[c1], %[b1]\n\t"
            "v_add_f32_e32 %[c2], %[c2], %[b2]\n\t"
            "v_add_f32_e32 %[c3], %[c3], %[b3]\n\t"
            "v_lshl_or_b32 %[a0], %[a0], 5, %[i0]\n\t"
            "v_lshl_or_b32 %[a1], %[a1], 5, %[i0]\n\t"
            "v_lshl_or_b32 %[a2], %[a2], 5, %[i0]\n\t"
            "v_lshl_or_b32 %[a3], %[a3], 5, %[i0]\n\t"
            "v_sub_f32_e64 %[c0], %[cc], |%[c0]|\n\t"
            "v_sub_f32_e64 %[c1], %[cc], |%[c1]|\n\t"
            "v_sub_f32_e64 %[c2], %[cc], |%[c2]|\n\t"
            "v_sub_f32_e64 %[c3], %[cc], |%[c3]|\n\t"
            "v_lshl_or_b32 %[c0], %[c0], 5, %[i1]\n\t"
            "v_lshl_or_b32 %[c1], %[c1], 5, %[i1]\n\t"
            "v_lshl_or_b32 %[c2], %[c2], 5, %[i1]\n\t"
            "v_lshl_or_b32 %[c3], %[c3], 5, %[i1]\n\t"
            "v_min3_u32 %[o0], %[o0], %[a0], %[c0]\n\t"
            "v_min3_u32 %[o1], %[o1], %[a1], %[c1]\n\t"
            "v_min3_u32 %[o2], %[o2], %[a2], %[c2]\n\t"
            "v_min3_u32 %[o3], %[o3], %[a3], %[c3]"
            : [e0] "+v"(ke[0]), [e1] "+v"(ke[1]), [e2] "+v"(ke[2]), [e3] "+v"(ke[3]), [o0] "+v"(ko[0]), [o1] "+v"(ko[1]), [o2] "+v"(ko[2]), [o3] "+v"(ko[3]), [a0] "=&v"(a[0]), [a1] "=&v"(a[1]), [a2] "=&v"(a[2]), [a3] "=&v"(a[3]), [b0] "=&v"(b[0]), [b1] "=&v"(b[1]), [b2] "=&v"(b[2]), [b3] "=&v"(b[3]), [c0] "=&v"(c[0]), [c1] "=&v"(c[1]), [c2] "=&v"(c[2]), [c3] "=&v"(c[3])
            : [t0x] "v"(g[0].x), [t0y] "v"(g[0].y), [t1x] "v"(g[1].x), [t1y] "v"(g[1].y), [t2x] "v"(g[2].x), [t2y] "v"(g[2].y), [t3x] "v"(g[3].x), [t3y] "v"(g[3].y), [x0] "v"(DX[0]), [y0] "v"(DY[0]), [x1] "v"(DX[1]), [y1] "v"(DY[1]), [x2] "v"(DX[2]), [y2] "v"(DY[2]), [x3] "v"(DX[3]), [y3] "v"(DY[3]), [cc] "s"(MH_KEY_C), [i0] "s"(ib), [i1] "s"(ib + 1)
            : "memory");
    }
    else if constexpr (NI == 3) {
        asm volatile(
            "v_mul_f32_e32 %[a0], %[t0x], %[x0]\n\t"
            "v_mul_f32_e32 %[a1], %[t0x], %[x1]\n\t"
            "v_mul_f32_e32 %[a2], %[t0x], %[x2]\n\t"
            "v_mul_f32_e32 %[b0], %[t0y], %[y0]\n\t"
            "v_mul_f32_e32 %[b1], %[t0y], %[y1]\n\t"
            "v_mul_f32_e32 %[b2], %[t0y], %[y2]\n\t"
            "v_mul_f32_e32 %[c0], %[t2x], %[x0]\n\t"
            "v_mul_f32_e32 %[c1], %[t2x], %[x1]\n\t"
            "v_mul_f32_e32 %[c2], %[t2x], %[x2]\n\t"
            "v_add_f32_e32 %[a0], %[a0], %[b0]\n\t"
            "v_add_f32_e32 %[a1], %[a1], %[b1]\n\t"
            "v_add_f32_e32 %[a2], %[a2], %[b2]\n\t"
            "v_mul_f32_e32 %[b0], %[t2y], %[y0]\n\t"
            "v_mul_f32_e32 %[b1], %[t2y], %[y1]\n\t"
            "v_mul_f32_e32 %[b2], %[t2y], %[y2]\n\t"
            "v_sub_f32_e64 %[a0], %[cc], |%[a0]|\n\t"
            "v_sub_f32_e64 %[a1], %[cc], |%[a1]|\n\t"
            "v_sub_f32_e64 %[a2], %[cc], |%[a2]|\n\t"
            "v_add_f32_e32 %[c0], %[c0], %[b0]\n\t"
            "v_add_f32_e32 %[c1], %[c1], %[b1]\n\t"
            "v_add_f32_e32 %[c2], %[c2], %[b2]\n\t"
            "v_lshl_or_b32 %[a0], %[a0], 5, %[i0]\n\t"
            "v_lshl_or_b32 %[a1], %[a1], 5, %[i0]\n\t"
            "v_lshl_or_b32 %[a2], %[a2], 5, %[i0]\n\t"
            "v_sub_f32_e64 %[c0], %[cc], |%[c0]|\n\t"
            "v_sub_f32_e64 %[c1], %[cc], |%[c1]|\n\t"
            "v_sub_f32_e64 %[c2], %[cc], |%[c2]|\n\t"
            "v_lshl_or_b32 %[c0], %[c0], 5, %[i1]\n\t"
            "v_lshl_or_b32 %[c1], %[c1], 5, %[i1]\n\t"
            "v_lshl_or_b32 %[c2], %[c2], 5, %[i1]\n\t"
            "v_min3_u32 %[e0], %[e0], %[a0], %[c0]\n\t"
            "v_min3_u32 %[e1], %[e1], %[a1], %[c1]\n\t"
            "v_min3_u32 %[e2], %[e2], %[a2], %[c2]\n\t"
            "v_mul_f32_e32 %[a0], %[t1x], %[x0]\n\t"
            "v_mul_f32_e32 %[a1], %[t1x], %[x1]\n\t"
            "v_mul_f32_e32 %[a2], %[t1x], %[x2]\n\t"
            "v_mul_f32_e32 %[b0], %[t1y], %[y0]\n\t"
            "v_mul_f32_e32 %[b1], %[t1y], %[y1]\n\t"
            "v_mul_f32_e32 %[b2], %[t1y], %[y2]\n\t"
            "v_mul_f32_e32 %[c0], %[t3x], %[x0]\n\t"
            "v_mul_f32_e32 %[c1], %[t3x], %[x1]\n\t"
            "v_mul_f32_e32 %[c2], %[t3x], %[x2]\n\t"
            "v_add_f32_e32 %[a0], %[a0], %[b0]\n\t"
            "v_add_f32_e32 %[a1], %[a1], %[b1]\n\t"
            "v_add_f32_e32 %[a2], %[a2], %[b2]\n\t"
            "v_mul_f32_e32 %[b0], %[t3y], %[y0]\n\t"
            "v_mul_f32_e32 %[b1], %[t3y], %[y1]\n\t"
            "v_mul_f32_e32 %[b2], %[t3y], %[y2]\n\t"
            "v_sub_f32_e64 %[a0], %[cc], |%[a0]|\n\t"
            "v_sub_f32_e64 %[a1], %[cc], |%[a1]|\n\t"
            "v_sub_f32_e64 %[a2], %[cc], |%[a2]|\n\t"
            "v_add_f32_e32 %[c0], %[c0], %[b0]\n\t"
            "v_add_f32_e32 %[c1], %[c1], %[b1]\n\t"
            "v_add_f32_e32 %[c2], %[c2], %[b2]\n\t"
            "v_lshl_or_b32 %[a0], %[a0], 5, %[i0]\n\t"
            "v_lshl_or_b32 %[a1], %[a1], 5, %[i0]\n\t"
            "v_lshl_or_b32 %[a2], %[a2], 5, %[i0]\n\t"
            "v_sub_f32_e64 %[c0], %[cc], |%[c0]|\n\t"
            "v_sub_f32_e64 %[c1], %[cc], |%[c1]|\n\t"
            "v_sub_f32_e64 %[c2], %[cc], |%[c2]|\n\t"
            "v_lshl_or_b32 %[c0], %[c0], 5, %[i1]\n\t"
            "v_lshl_or_b32 %[c1], %[c1], 5, %[i1]\n\t"
            "v_lshl_or_b32 %[c2], %[c2], 5, %[i1]\n\t"
            "v_min3_u32 %[o0], %[o0], %[a0], %[c0]\n\t"
            "v_min3_u32 %[o1], %[o1], %[a1], %[c1]\n\t"
            "v_min3_u32 %[o2], %[o2], %[a2], %[c2]"
            : [e0] "+v"(ke[0]), [e1] "+v"(ke[1]), [e2] "+v"(ke[2]), [o0] "+v"(ko[0]), [o1] "+v"(ko[1]), [o2] "+v"(ko[2]), [a0] "=&v"(a[0]), [a1] "=&v"(a[1]), [a2] "=&v"(a[2]), [b0] "=&v"(b[0]), [b1] "=&v"(b[1]), [b2] "=&v"(b[2]), [c0] "=&v"(c[0]), [c1] "=&v"(c[1]), [c2] "=&v"(c[2])
            : [t0x] "v"(g[0].x), [t0y] "v"(g[0].y), [t1x] "v"(g[1].x), [t1y] "v"(g[1].y), [t2x] "v"(g[2].x), [t2y] "v"(g[2].y), [t3x] "v"(g[3].x), [t3y] "v"(g[3].y), [x0] "v"(DX[0]), [y0] "v"(DY[0]), [x1] "v"(DX[1]), [y1] "v"(DY[1]), [x2] "v"(DX[2]), [y2] "v"(DY[2]), [cc] "s"(MH_KEY_C), [i0] "s"(ib), [i1] "s"(ib + 1)
            : "memory");
    }
    else if constexpr (NI == 2) {
        asm volatile(
            "v_mul_f32_e32 %[a0], %[t0x], %[x0]\n\t"
            "v_mul_f32_e32 %[a1], %[t0x], %[x1]\n\t"
            "v_mul_f32_e32 %[b0], %[t0y], %[y0]\n\t"
            "v_mul_f32_e32 %[b1], %[t0y], %[y1]\n\t"
            "v_mul_f32_e32 %[c0], %[t2x], %[x0]\n\t"
            "v_mul_f32_e32 %[c1], %[t2x], %[x1]\n\t"
            "v_add_f32_e32 %[a0], %[a0], %[b0]\n\t"
            "v_add_f32_e32 %[a1], %[a1], %[b1]\n\t"
            "v_mul_f32_e32 %[b0], %[t2y], %[y0]\n\t"
            "v_mul_f32_e32 %[b1], %[t2y], %[y1]\n\t"
            "v_sub_f32_e64 %[a0], %[cc], |%[a0]|\n\t"
            "v_sub_f32_e64 %[a1], %[cc], |%[a1]|\n\t"
            "v_add_f32_e32 %[c0], %[c0], %[b0]\n\t"
            "v_add_f32_e32 %[c1], %[c1], %[b1]\n\t"
            "v_lshl_or_b32 %[a0], %[a0], 5, %[i0]\n\t"
            "v_lshl_or_b32 %[a1], %[a1], 5, %[i0]\n\t"
            "v_sub_f32_e64 %[c0], %[cc], |%[c0]|\n\t"
            "v_sub_f32_e64 %[c1], %[cc], |%[c1]|\n\t"
            "v_lshl_or_b32 %[c0], %[c0], 5, %[i1]\n\t"
            "v_lshl_or_b32 %[c1], %[c1], 5, %[i1]\n\t"
            "v_min3_u32 %[e0], %[e0], %[a0], %[c0]\n\t"
            "v_min3_u32 %[e1], %[e1], %[a1], %[c1]\n\t"
            "v_mul_f32_e32 %[a0], %[t1x], %[x0]\n\t"
            "v_mul_f32_e32 %[a1], %[t1x], %[x1]\n\t"
            "v_mul_f32_e32 %[b0], %[t1y], %[y0]\n\t"
            "v_mul_f32_e32 %[b1], %[t1y], %[y1]\n\t"
            "v_mul_f32_e32 %[c0], %[t3x], %[x0]\n\t"
            "v_mul_f32_e32 %[c1], %[t3x], %[x1]\n\t"
            "v_add_f32_e32 %[a0], %[a0], %[b0]\n\t"
            "v_add_f32_e32 %[a1], %[a1], %[b1]\n\t"
            "v_mul_f32_e32 %[b0], %[t3y], %[y0]\n\t"
            "v_mul_f32_e32 %[b1], %[t3y], %[y1]\n\t"
            "v_sub_f32_e64 %[a0], %[cc], |%[a0]|\n\t"
            "v_sub_f32_e64 %[a1], %[cc], |%[a1]|\n\t"
            "v_add_f32_e32 %[c0], %[c0], %[b0]\n\t"
            "v_add_f32_e32 %[c1], %[c1], %[b1]\n\t"
            "v_lshl_or_b32 %[a0], %[a0], 5, %[i0]\n\t"
            "v_lshl_or_b32 %[a1], %[a1], 5, %[i0]\n\t"
            "v_sub_f32_e64 %[c0], %[cc], |%[c0]|\n\t"
            "v_sub_f32_e64 %[c1], %[cc], |%[c1]|\n\t"
            "v_lshl_or_b32 %[c0], %[c0], 5, %[i1]\n\t"
            "v_lshl_or_b32 %[c1], %[c1], 5, %[i1]\n\t"
            "v_min3_u32 %[o0], %[o0], %[a0], %[c0]\n\t"
            "v_min3_u32 %[o1], %[o1], %[a1], %[c1]"
            : [e0] "+v"(ke[0]), [e1] "+v"(ke[1]), [o0] "+v"(ko[0]), [o1] "+v"(ko[1]), [a0] "=&v"(a[0]), [a1] "=&v"(a[1]), [b0] "=&v"(b[0]), [b1] "=&v"(b[1]), [c0] "=&v"(c[0]), [c1] "=&v"(c[1])
            : [t0x] "v"(g[0].x), [t0y] "v"(g[0].y), [t1x] "v"(g[1].x), [t1y] "v"(g[1].y), [t2x] "v"(g[2].x), [t2y] "v"(g[2].y), [t3x] "v"(g[3].x), [t3y] "v"(g[3].y), [x0] "v"(DX[0]), [y0] "v"(DY[0]), [x1] "v"(DX[1]), [y1] "v"(DY[1]), [cc] "s"(MH_KEY_C), [i0] "s"(ib), [i1] "s"(ib + 1)
            : "memory");
    }
    else if constexpr (NI == 1) {
        asm volatile(
            "v_mul_f32_e32 %[a0], %[t0x], %[x0]\n\t"
            "v_mul_f32_e32 %[b0], %[t0y], %[y0]\n\t"
            "v_mul_f32_e32 %[c0], %[t2x], %[x0]\n\t"
            "v_add_f32_e32 %[a0], %[a0], %[b0]\n\t"
            "v_mul_f32_e32 %[b0], %[t2y], %[y0]\n\t"
            "v_sub_f32_e64 %[a0], %[cc], |%[a0]|\n\t"
            "v_add_f32_e32 %[c0], %[c0], %[b0]\n\t"
            "v_lshl_or_b32 %[a0], %[a0], 5, %[i0]\n\t"
            "v_sub_f32_e64 %[c0], %[cc], |%[c0]|\n\t"
            "v_lshl_or_b32 %[c0], %[c0], 5, %[i1]\n\t"
            "v_min3_u32 %[e0], %[e0], %[a0], %[c0]\n\t"
            "v_mul_f32_e32 %[a0], %[t1x], %[x0]\n\t"
            "v_mul_f32_e32 %[b0], %[t1y], %[y0]\n\t"
            "v_mul_f32_e32 %[c0], %[t3x], %[x0]\n\t"
            "v_add_f32_e32 %[a0], %[a0], %[b0]\n\t"
            "v_mul_f32_e32 %[b0], %[t3y], %[y0]\n\t"
            "v_sub_f32_e64 %[a0], %[cc], |%[a0]|\n\t"
            "v_add_f32_e32 %[c0], %[c0], %[b0]\n\t"
            "v_lshl_or_b32 %[a0], %[a0], 5, %[i0]\n\t"
            "v_sub_f32_e64 %[c0], %[cc], |%[c0]|\n\t"
            "v_lshl_or_b32 %[c0], %[c0], 5, %[i1]\n\t"
            "v_min3_u32 %[o0], %[o0], %[a0], %[c0]"
            : [e0] "+v"(ke[0]), [o0] "+v"(ko[0]), [a0] "=&v"(a[0]), [b0] "=&v"(b[0]), [c0] "=&v"(c[0])
            : [t0x] "v"(g[0].x), [t0y] "v"(g[0].y), [t1x] "v"(g[1].x), [t1y] "v"(g[1].y), [t2x] "v"(g[2].x), [t2y] "v"(g[2].y), [t3x] "v"(g[3].x), [t3y] "v"(g[3].y), [x0] "v"(DX[0]), [y0] "v"(DY[0]), [cc] "s"(MH_KEY_C), [i0] "s"(ib), [i1] "s"(ib + 1)
            : "memory");
    }
}

// ---------------------------------------------------------------------------------------------
// mh_search3_kernel -- the shipped search: the arithmetic of mh_search_kernel in the same order, laid out for the machine.
// Where the wave-uniform tap records come from:
//   * round 2's first form read them with broadcast vector loads: every one of the 4 waves of the workgroup loaded every tap
//     record of every visible view (20.9 M wave-level loads per launch, 768 B of register return each through the CU's one
//     vector-memory path), and the first two loads of a view (first tap, first group) were waited for at full L2 latency;
//   * here the 256 threads copy the lists of the visible views into LDS once (coalesced 16-B loads, one copy per
//     workgroup instead of four), one barrier, and the tap loop reads them back with same-address ds_read (broadcast, no
//     bank conflict, ~100 cycles of latency that the ping-pong groups cover).  Lists that do not fit (MH_S3_CAP records)
//     go in several batches; the typical point (22 visible views x 46 taps) needs one.
//   * the visible views, their list lengths and LDS offsets come from the compact [V,N] byte array of list lengths: one
//     lane per view, a wave prefix sum, v_readlane -- no per-view header load, no branch on it.
// ---------------------------------------------------------------------------------------------
#define MH_S3_GRP 4      // tap records per ping-pong group
#ifndef MH_S3_WAVES
#define MH_S3_WAVES 5   // waves per SIMD the register allocation aims at (A/B builds: -DMH_S3_WAVES=4|6)
#endif
#ifndef MH_PAIR_TAPS
#define MH_PAIR_TAPS 4    // select-only kernel: two views whose lists hold at most this many taps are evaluated together
#endif
#ifndef MH_S3_WAVES_SELECT
#define MH_S3_WAVES_SELECT MH_S3_WAVES   // the same aim for the select-only kernel (A/B builds: -DMH_S3_WAVES_SELECT=4)
#endif
#define MH_S3_CAP 1280   // float4 records per workgroup (20 KB; 6 workgroups of 25 KB per CU)

// the cascade of mh_device.h (MhCascV) with the third level only where it can be reached
template <bool BIG>
struct MhCascS {
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
    __device__ __forceinline__ void flush(int v) {   // v > 0, v % 16 == 0, before adding row v
        a1 = a1 + a0;
        a0 = 0.0f;
        if constexpr (BIG) {
            if ((v & 0xF0) == 0) {
                a2 = a2 + a1;
                a1 = 0.0f;
            }
        }
    }
    __device__ __forceinline__ float done() const { return BIG ? (a0 + a1) + a2 : (a0 + a1); }
};

template <int KA, int T, bool BIGV, bool KEYS, bool BIGP>
__device__ __forceinline__ void mh_search_slices_lds(const MhViews &vw, const float *__restrict__ offs, int S,
                                                     int n, int N, int P1, float thr,
                                                     const float4 *__restrict__ taps, const uint8_t *__restrict__ vcnt,
                                                     int nact, int tid, float *s_loss, uint8_t *s_pos, float4 *s_taps,
                                                     const float4 *s_rank, int c_first) {
    constexpr int KN = KA > 0 ? KA : 1;   // a wave without items (KA == 0) only helps to stage the lists
    constexpr int KM = KA;
    const int V = vw.V;
    const float Hf = (float)vw.H, Wf = (float)vw.W;
    float X0[KN], X1[KN], X2[KN];
    // BIGV = more than 256 views: only then does the cascade of the weighted sums reach its third level (the level stays
    // +0 otherwise and x + (+0) = x for the non-negative sums: 8 registers less)
    MhCascS<BIGV> num[KN], den[KN];
    int cnt[KN];
    if constexpr (KA > 0) {
        const unsigned inv = (1u << 20) / (unsigned)S + 1u;   // it / S for it < 1024 >= S: (it * inv) >> 20 (error < 2^-10 <= 1/S)
#pragma unroll
        for (int j = 0; j < KA; ++j) {
            int it = j * T + tid;
            it = it < nact ? it : 0;
            const int r = (int)(((unsigned)it * inv) >> 20), s = it - r * S;
            mh_sample_item(s_rank + 4 * r, offs[s], X0[j], X1[j], X2[j]);   // (the rank part: mh_search3_kernel's prologue)
            num[j] = den[j] = MhCascS<BIGV>{};
            cnt[j] = 0;
        }
    }
    int nf = 16;   // next view index at which the cascade of the weighted sums is flushed (every multiple of 16 below V)
    auto flush_upto = [&](int v) {
        while (nf <= v) {
#pragma unroll
            for (int j = 0; j < KN; ++j) {
                num[j].flush(nf);
                den[j].flush(nf);
            }
            nf += 16;
        }
    };
    // one visible view: rec = its list in LDS (header, then ntap taps)
    auto one_view = [&](int v, const float4 *rec, int ntap) {
        const float *__restrict__ cam = vw.cams + v * MH_CAM_STRIDE;
        const float4 hdr = rec[0];
        const float4 t0 = rec[1];
        float DX[KN], DY[KN], ML[KN], BC[KN];
#ifdef MH_EXP_NOTAPS   // timing experiments only (wrong results): tools/exp_search_parts.sh
        ntap = 1;
#endif
#ifdef MH_EXP_NOPROJ
#pragma unroll
        for (int j = 0; j < KN; ++j) {
            DX[j] = X0[j] + hdr.z;
            DY[j] = X1[j] + hdr.w;
        }
#else
#pragma unroll
        for (int jp = 0; jp < KA / 2; ++jp) {
            mh_v2f row, col, dx, dy;
            mh_pixel_of_fast2(cam, mh_v2f{X0[2 * jp], X0[2 * jp + 1]}, mh_v2f{X1[2 * jp], X1[2 * jp + 1]},
                              mh_v2f{X2[2 * jp], X2[2 * jp + 1]}, Hf, Wf, row, col);
            mh_unit2_fast2(row - mh_splat(hdr.z), col - mh_splat(hdr.w), dx, dy);
            DX[2 * jp] = dx.x;
            DX[2 * jp + 1] = dx.y;
            DY[2 * jp] = dy.x;
            DY[2 * jp + 1] = dy.y;
        }
        if constexpr (KA & 1) {
            mh_v2f row, col, dx, dy;
            mh_pixel_of_fast2(cam, mh_splat(X0[KA - 1]), mh_splat(X1[KA - 1]), mh_splat(X2[KA - 1]), Hf, Wf, row, col);
            mh_unit2_fast2(row - mh_splat(hdr.z), col - mh_splat(hdr.w), dx, dy);
            DX[KA - 1] = dx.x;
            DY[KA - 1] = dy.x;
        }
#endif
        constexpr int GRP = MH_S3_GRP;
        // the compare-and-select body: exact everywhere (the one body of rounds 1-3; with KEYS the re-evaluation path)
        auto select_body = [&]() {
#pragma unroll
            for (int j = 0; j < KA; ++j) {
                ML[j] = mh_one_minus_abs(mh_vadd(mh_vmul(t0.x, DX[j]), mh_vmul(t0.y, DY[j])));
                BC[j] = t0.z;
            }
            auto process = [&](const float4 (&g)[GRP], int t) {
#pragma unroll
                for (int u = 0; u < GRP; ++u) {
                    if (t + u < ntap) {   // uniform
                        const float4 tp = g[u];
                        float l[KN];
#pragma unroll
                        for (int j = 0; j < KA; ++j)
                            l[j] = mh_one_minus_abs(mh_vadd(mh_vmul(tp.x, DX[j]), mh_vmul(tp.y, DY[j])));
                        mh_tap_update<KN>(ML, BC, l, tp.z);
                    }
                }
            };
            float4 ga[GRP], gb[GRP];
#pragma unroll
            for (int u = 0; u < GRP; ++u) ga[u] = rec[2 + u];
            for (int t = 1; t < ntap;) {
#pragma unroll
                for (int u = 0; u < GRP; ++u) gb[u] = rec[1 + t + GRP + u];
                process(ga, t);
                t += GRP;
                if (t >= ntap) break;
#pragma unroll
                for (int u = 0; u < GRP; ++u) ga[u] = rec[1 + t + GRP + u];
                process(gb, t);
                t += GRP;
            }
        };
        __builtin_amdgcn_s_setprio(0);   // the tap loop: see mh_search3_kernel
        if constexpr (!KEYS) {
            select_body();
        } else {
            // the key body (see mh_tap_key): taps in groups of 64, the list padded to a multiple of MH_KEY_PAD taps
            // uniform: a short list (the key body's fixed cost per view -- padding, decode -- only pays from about a dozen
            // taps on: lists of 8-bit maps are mostly shorter, lists of continuous maps hardly ever) / a NaN seed tap
            bool again = (ntap <= MH_KEY_MIN_TAPS) || !(t0.x == t0.x && t0.y == t0.y);
            MH_KEY_COUNT(0);
            if (again) MH_KEY_COUNT(1);
            if (!again) {
                const int ntp = (ntap + MH_KEY_PAD - 1) & ~(MH_KEY_PAD - 1);
                // even taps of a 64-tap group into ke, odd taps into ko, the key's index = the tap's place among them
                unsigned ke[KN], ko[KN];
                const unsigned rec1 = (unsigned)(size_t)(const __attribute__((address_space(3))) float4 *)(rec + 1);
                auto process = [&](const float2 (&g)[GRP], int t) {
                    const int ib = (t & 63) >> 1;
                    static_assert(MH_KEY_PAD == 4 && GRP == 4, "the key block takes four taps");
                    if constexpr (KM > 0) mh_key_block4<KM, KN>(ke, ko, g, DX, DY, ib);
                };
                // taps [ta, tb) of one 64-tap group: (tx, ty) of tap i is the first half of record 1 + i
                auto group = [&](int ta, int tb) {
                    const float2 *__restrict__ t2 = reinterpret_cast<const float2 *>(rec + 1);
#pragma unroll
                    for (int j = 0; j < KN; ++j) ke[j] = ko[j] = 0xFFFFFFFFu;
                    float2 ga[GRP], gb[GRP];
#pragma unroll
                    for (int u = 0; u < GRP; ++u) ga[u] = t2[2 * (ta + u)];
                    for (int t = ta; t < tb;) {
#pragma unroll
                        for (int u = 0; u < GRP; ++u) gb[u] = t2[2 * (t + GRP + u)];
                        process(ga, t);
                        t += GRP;
                        if (t >= tb) break;
#pragma unroll
                        for (int u = 0; u < GRP; ++u) ga[u] = t2[2 * (t + GRP + u)];
                        process(gb, t);
                        t += GRP;
                    }
                };
                // the group's winner: the odd key only when it is strictly smaller (equal loss and place: the even tap is the
                // earlier one; equal loss, smaller place: 2i + 1 < 2i'); its tap = 2 * place + parity
                unsigned best[KN], badr[KN];   // winning key, LDS byte address of the winner's record
                auto winner = [&](int ta, bool first) {
#pragma unroll
                    for (int j = 0; j < KM; ++j) {
                        const bool odd = ko[j] < ke[j];
                        const unsigned kb = odd ? ko[j] : ke[j];
                        const unsigned ad = rec1 + 16u * (unsigned)ta + ((kb & 31u) << 5) + (odd ? 16u : 0u);
                        const bool take = first || kb < (best[j] & ~31u);   // a later group only on a strictly smaller loss
                        best[j] = take ? kb : best[j];
                        badr[j] = take ? ad : badr[j];
                    }
                };
                if constexpr (!BIGP) {   // patch <= 8x8: every list is one 64-tap group
                    group(0, ntp);
                    winner(0, true);
                } else {                 // patch 9 and 11: up to two groups (a later group wins only on a strictly smaller loss)
                    group(0, ntp < 64 ? ntp : 64);
                    winner(0, true);
                    for (int ta = 64; ta < ntp; ta += 64) {
                        group(ta, ntp < ta + 64 ? ntp : ta + 64);
                        winner(ta, false);
                    }
                }
                // decode: one compare for "any key past the valid range", loss = t' - 2^-14 from the key's upper bits
                // (v_alignbit puts the constant 0b00111 back in front), confidence of the winning tap from its record
                unsigned worst = best[0];
#pragma unroll
                for (int j = 1; j < KM; ++j) worst = max(worst, best[j]);
                const bool bad = worst >= MH_KEY_BAD;
#pragma unroll
                for (int j = 0; j < KM; ++j) {
                    BC[j] = *reinterpret_cast<const __attribute__((address_space(3))) float *>((size_t)(badr[j] + 8u));
                    ML[j] = __uint_as_float(__builtin_amdgcn_alignbit(7u, best[j], 5u)) - MH_KEY_E;
                }
                again = __ballot(bad) != 0ull;
                if (again) MH_KEY_COUNT_ALWAYS(2);
            }
            if (again) select_body();
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < KA; ++j) {
            const float w = BC[j];   // (vis != -1) * best_conf
            num[j].a0 = num[j].a0 + ML[j] * w;
            den[j].a0 = den[j].a0 + w;
            cnt[j] += (w > 0.0f) ? 1 : 0;
        }
    };
#ifdef MH_EXP_VIEW_PAIRS   // (experiment build only: tools/exp_view_pairs.sh -- measured SLOWER in every form, docs/HISTORY.md round 5)
    // Two views in flight (select-only kernel, short lists -- the regime of 8-bit maps, ~2 taps per list): there the time of a
    // view is the dependent chain of its projection (camera record, rcp, sqrt, two refined divisions: 98 of the kernel's 211 us,
    // 24 for the taps), and five waves per SIMD do not hide it (issue utilisation 0.53).  The projections of TWO visible views
    // are written as one block of straight-line code, so that the scheduler interleaves the two chains; then the two short tap
    // loops, then the accumulation in view order -- the same operations on the same values as two calls of one_view.
    // (The caller pairs views only inside one 16-view block of the cascade: no flush between them.)
    auto two_views = [&](int va, const float4 *reca, int na, int vb2, const float4 *recb, int nb) {
        const float *__restrict__ cama = vw.cams + va * MH_CAM_STRIDE;
        const float *__restrict__ camb = vw.cams + vb2 * MH_CAM_STRIDE;
        const float4 ha = reca[0], hb = recb[0];
        const float4 ta = reca[1], tb = recb[1];
        float DXa[KN], DYa[KN], DXb[KN], DYb[KN];
#pragma unroll
        for (int jp = 0; jp < KA / 2; ++jp) {
            mh_v2f ra, ca, rb, cb, dxa, dya, dxb, dyb;
            const mh_v2f x0 = mh_v2f{X0[2 * jp], X0[2 * jp + 1]}, x1 = mh_v2f{X1[2 * jp], X1[2 * jp + 1]},
                         x2 = mh_v2f{X2[2 * jp], X2[2 * jp + 1]};
            mh_pixel_of_fast2(cama, x0, x1, x2, Hf, Wf, ra, ca);
            mh_pixel_of_fast2(camb, x0, x1, x2, Hf, Wf, rb, cb);
            mh_unit2_fast2(ra - mh_splat(ha.z), ca - mh_splat(ha.w), dxa, dya);
            mh_unit2_fast2(rb - mh_splat(hb.z), cb - mh_splat(hb.w), dxb, dyb);
            DXa[2 * jp] = dxa.x; DXa[2 * jp + 1] = dxa.y; DYa[2 * jp] = dya.x; DYa[2 * jp + 1] = dya.y;
            DXb[2 * jp] = dxb.x; DXb[2 * jp + 1] = dxb.y; DYb[2 * jp] = dyb.x; DYb[2 * jp + 1] = dyb.y;
        }
        if constexpr (KA & 1) {
            mh_v2f ra, ca, rb, cb, dxa, dya, dxb, dyb;
            const mh_v2f x0 = mh_splat(X0[KA - 1]), x1 = mh_splat(X1[KA - 1]), x2 = mh_splat(X2[KA - 1]);
            mh_pixel_of_fast2(cama, x0, x1, x2, Hf, Wf, ra, ca);
            mh_pixel_of_fast2(camb, x0, x1, x2, Hf, Wf, rb, cb);
            mh_unit2_fast2(ra - mh_splat(ha.z), ca - mh_splat(ha.w), dxa, dya);
            mh_unit2_fast2(rb - mh_splat(hb.z), cb - mh_splat(hb.w), dxb, dyb);
            DXa[KA - 1] = dxa.x; DYa[KA - 1] = dya.x;
            DXb[KA - 1] = dxb.x; DYb[KA - 1] = dyb.x;
        }
        float MLa[KN], BCa[KN], MLb[KN], BCb[KN];
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int j = 0; j < KA; ++j) {
            MLa[j] = mh_one_minus_abs(mh_vadd(mh_vmul(ta.x, DXa[j]), mh_vmul(ta.y, DYa[j])));
            BCa[j] = ta.z;
            MLb[j] = mh_one_minus_abs(mh_vadd(mh_vmul(tb.x, DXb[j]), mh_vmul(tb.y, DYb[j])));
            BCb[j] = tb.z;
        }
        for (int t = 1; t < na; ++t) {   // uniform
            const float4 tp = reca[1 + t];
            float l[KN];
#pragma unroll
            for (int j = 0; j < KA; ++j) l[j] = mh_one_minus_abs(mh_vadd(mh_vmul(tp.x, DXa[j]), mh_vmul(tp.y, DYa[j])));
            mh_tap_update<KN>(MLa, BCa, l, tp.z);
        }
        for (int t = 1; t < nb; ++t) {
            const float4 tp = recb[1 + t];
            float l[KN];
#pragma unroll
            for (int j = 0; j < KA; ++j) l[j] = mh_one_minus_abs(mh_vadd(mh_vmul(tp.x, DXb[j]), mh_vmul(tp.y, DYb[j])));
            mh_tap_update<KN>(MLb, BCb, l, tp.z);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < KA; ++j) {
            num[j].a0 = num[j].a0 + MLa[j] * BCa[j];
            den[j].a0 = den[j].a0 + BCa[j];
            cnt[j] += (BCa[j] > 0.0f) ? 1 : 0;
            num[j].a0 = num[j].a0 + MLb[j] * BCb[j];
            den[j].a0 = den[j].a0 + BCb[j];
            cnt[j] += (BCb[j] > 0.0f) ? 1 : 0;
        }
    };
#endif
    const int lane = tid & 63, wave = tid >> 6;
    for (int vb = 0; vb < V; vb += 64) {
        const int vv = vb + lane;
        // list length of view vv (0: the view does not see the point); the first block's was requested in the kernel's
        // prologue, in front of the first barrier
        const int c = vb == 0 ? c_first : ((vv < V) ? (int)vcnt[(size_t)vv * N + n] : 0);
        // records: header + taps (KEYS: the taps of a list that goes through the key body padded to a multiple of
        // MH_KEY_PAD with neutral (0, 0) records)
        const int len = c ? ((KEYS && c > MH_KEY_MIN_TAPS) ? ((c + MH_KEY_PAD - 1) & ~(MH_KEY_PAD - 1)) : c) + 1 : 0;
        int pre = len;                                               // inclusive prefix sum over the lanes
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(pre, o);
            pre += (lane >= o) ? y : 0;
        }
        unsigned long long todo = __ballot(c != 0);
        int base = 0;
        while (todo) {
            // the next lists that fit together (pre is monotone: a prefix of the views left; one list always fits)
            const unsigned long long take = todo & __ballot(pre - base <= MH_S3_CAP);
            {
                unsigned long long m = take;
                int k = 0;
                while (m) {
                    const int b = (int)__builtin_ctzll(m);
                    m &= m - 1;
                    if ((k & (T / 64 - 1)) == wave) {
                        const int L = __builtin_amdgcn_readlane(len, b);
                        const int off = __builtin_amdgcn_readlane(pre, b) - L - base;
                        const float4 *__restrict__ src = taps + ((size_t)(vb + b) * N + n) * P1;
                        if constexpr (KEYS) {
                            const int lr = __builtin_amdgcn_readlane(c, b) + 1;   // records the list really has
                            for (int i = lane; i < L; i += 64)
                                s_taps[off + i] = (i < lr) ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                        } else {
                            for (int i = lane; i < L; i += 64) s_taps[off + i] = src[i];
                        }
                    }
                    ++k;
                }
            }
            __syncthreads();
            if constexpr (KA > 0) {
                unsigned long long m = take;
                while (m) {
                    const int b = (int)__builtin_ctzll(m);
                    m &= m - 1;
                    flush_upto(vb + b);
                    const int L = __builtin_amdgcn_readlane(len, b);
                    const int off = __builtin_amdgcn_readlane(pre, b) - L - base;
#ifdef MH_EXP_VIEW_PAIRS
                    if constexpr (!KEYS) {   // (see two_views)
                        if (m && L <= MH_PAIR_TAPS + 1) {
                            const int b2 = (int)__builtin_ctzll(m);
                            const int L2 = __builtin_amdgcn_readlane(len, b2);
                            if (L2 <= MH_PAIR_TAPS + 1 && vb + b2 < nf) {
                                m &= m - 1;
                                const int off2 = __builtin_amdgcn_readlane(pre, b2) - L2 - base;
                                two_views(vb + b, s_taps + off, L - 1, vb + b2, s_taps + off2, L2 - 1);
                                continue;
                            }
                        }
                    }
#endif
                    one_view(vb + b, s_taps + off, KEYS ? __builtin_amdgcn_readlane(c, b) : L - 1);
                }
            }
            __syncthreads();
            base = __builtin_amdgcn_readlane(pre, 63 - (int)__builtin_clzll(take));
            todo &= ~take;
        }
    }
    if constexpr (KA > 0) {
        flush_upto(V - 1);
#pragma unroll
        for (int j = 0; j < KA; ++j) {
            const int it = j * T + tid;
            if (it < nact) {
                const float dn = den[j].done();
                const float nm = num[j].done();
                const float ratio = dn / (float)cnt[j];
                s_pos[it] = (ratio > thr) ? 1 : 0;
                s_loss[it] = nm / dn;
            }
        }
    }
}

template <int T, bool BIGV, bool KEYS, bool BIGP>
// amdgpu_waves_per_eu(5): the register allocator stops at 96 VGPRs (it takes 109 unconstrained = 4 waves per SIMD); the
// few values it spills are reloaded once per view.  Measured: 4 waves 1305 it/s, 5 waves 1345, 6 waves (80 VGPRs) 1328.
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(KEYS ? MH_S3_WAVES : MH_S3_WAVES_SELECT))) void mh_search3_kernel(MhViews vw, const float *__restrict__ offs, int S, int nrank,
                                                       int rank_step, const float *__restrict__ pts, int N, int P1,
                                                       float thr, const float *__restrict__ ori_c,
                                                       const int32_t *__restrict__ base_idx,
                                                       const float *__restrict__ base_val,
                                                       const float4 *__restrict__ taps,
                                                       const uint8_t *__restrict__ vcnt,
                                                       const int32_t *__restrict__ order, float *__restrict__ line_ori,
                                                       float *__restrict__ min_loss, uint8_t *__restrict__ high_conf,
                                                       float *__restrict__ best_sample, int32_t *__restrict__ best_rank,
                                                       int32_t *__restrict__ best_s, MhRule rule) {
    __shared__ float4 s_taps[MH_S3_CAP + 8];   // (+8: the last prefetch group of a list reads past its end)
    __shared__ float s_loss[MH_MAX_ITEMS];
    __shared__ uint8_t s_pos[MH_MAX_ITEMS];
    __shared__ float s_rl[MH_MAX_RANKS];
    __shared__ int s_ri[MH_MAX_RANKS];
    __shared__ int s_rh[MH_MAX_RANKS];
    __shared__ float4 s_rank[MH_MAX_RANKS * 4];   // mh_sample_rank's record of every base-view rank
    __shared__ float s_bval[MH_MAX_RANKS];        // base_view_conf of the ranks
    __shared__ int s_tail;                        // first trailing sample of this point (mh_tail_from); S = none

    // Wave priority: everything that is not the tap loop -- prologue, staging, the per-view projection, the epilogue -- runs
    // at priority 1, the tap loop at 0.  The tap loops saturate the VALU whatever the arbiter picks; the other phases are
    // chains of dependent long-latency operations (loads, LDS, rcp / sqrt) whose waves should get their instruction in as
    // soon as it is ready, so that they are back in a tap loop sooner: +5 % (the other way round: -8 %).
    __builtin_amdgcn_s_setprio(1);
    const int tid = threadIdx.x;
    const int n = order ? order[blockIdx.x] : (int)blockIdx.x;
    const float P0 = pts[3 * n], P1x = pts[3 * n + 1], P2 = pts[3 * n + 2];
    // what the S samples of a rank share (mh_sample_rank), once per (point, rank): lane r of the first wave -- for every
    // rank, usable or not, so that the rank's confidence, its base view, that view's centre orientation and camera are one
    // chain of loads per lane, all ranks in parallel (a scalar loop over base_view_conf first cost five more round trips
    // before the workgroup's first barrier)
    const int c_first = ((tid & 63) < vw.V) ? (int)vcnt[(size_t)(tid & 63) * N + n] : 0;   // (see mh_search_slices_lds)
    if (tid == 64) s_tail = mh_tail_from(rule, n, S);
    if (tid < nrank) {
        const size_t ro = (size_t)(tid * rank_step) * N + n;
        s_bval[tid] = base_val[ro];
        const int b = min(max(base_idx[ro], 0), vw.V - 1);   // (a caller's unusable ranks may carry any index)
        const float2 oc = reinterpret_cast<const float2 *>(ori_c)[(size_t)b * N + n];
        mh_sample_rank(vw.cams + b * MH_CAM_STRIDE, P0, P1x, P2, oc.x, oc.y, (float)vw.H, (float)vw.W,
                       reinterpret_cast<float *>(s_rank + 4 * tid), mh_group_forms(rule, tid, vw.V, b, S));
    }
    __syncthreads();
    // usable base-view ranks: rank 0 always, rank r > 0 only if base_view_conf[r] > 0 (PMVO.py:57-64); keep every rank up
    // to the last usable one
    int nvalid = 1;
    for (int r = 1; r < nrank; ++r)
        if (s_bval[r] > 0.0f) nvalid = r + 1;
    const int nact = nvalid * S;
    const int wave0 = tid & ~63;   // first item of this wave in slice 0
    int ka = 0;
    for (int j = 0; j < 4; ++j) ka += (j * T + wave0 < nact) ? 1 : 0;
#define MH_S3_ARGS vw, offs, S, n, N, P1, thr, taps, vcnt, nact, tid, s_loss, s_pos, s_taps, s_rank, c_first
    if (ka == 4) mh_search_slices_lds<4, T, BIGV, KEYS, BIGP>(MH_S3_ARGS);
    else if (ka == 3) mh_search_slices_lds<3, T, BIGV, KEYS, BIGP>(MH_S3_ARGS);
    else if (ka == 2) mh_search_slices_lds<2, T, BIGV, KEYS, BIGP>(MH_S3_ARGS);
    else if (ka == 1) mh_search_slices_lds<1, T, BIGV, KEYS, BIGP>(MH_S3_ARGS);
    else mh_search_slices_lds<0, T, BIGV, KEYS, BIGP>(MH_S3_ARGS);
#undef MH_S3_ARGS
    __syncthreads();
    // ---- the trailing columns of the batch's [V, N*S] sums (samples >= s_tail of the batch's last point(s); s_tail == S, i.e.
    // none, in every other workgroup): ATen adds those in its row_sum order -- their losses once more, that way
    if (s_tail < S) {   // uniform
        const int tail_from = s_tail;
        for (int it = tid; it < nact; it += T) {
            const int r = it / S, s = it - r * S;
            if (s < tail_from) continue;
            float X0, X1, X2, nm, dn;
            int cnt;
            mh_sample_item(s_rank + 4 * r, offs[s], X0, X1, X2);
            mh_tail_item_sums(vw.cams, vw.V, (float)vw.H, (float)vw.W, taps + (size_t)n * P1, (size_t)N * P1, vcnt + n, N, X0, X1,
                              X2, nm, dn, cnt);
            s_pos[it] = (dn / (float)cnt > thr) ? 1 : 0;
            s_loss[it] = nm / dn;
        }
        __syncthreads();
    }

    // ---- per rank: low-confidence escape hatch, min / argmin over the S samples (PMVO.py:199-206)
    const int wave = tid >> 6, lane = tid & 63, nwaves = T >> 6;
    for (int r = wave; r < nvalid; r += nwaves) {
        int npos = 0;
        for (int s0 = 0; s0 < S; s0 += MH_WAVE) {
            const int s = s0 + lane;
            npos += __popcll(__ballot(s < S && s_pos[r * S + s]));
        }
        const bool low = npos < 5;
        float bl = 0.0f;
        int bi = 0x7fffffff;
        for (int s = lane; s < S; s += MH_WAVE) {
            float l = s_loss[r * S + s];
            if (!low && !s_pos[r * S + s]) l = 1.0f;
            if (bi == 0x7fffffff || mh_min_better(l, s, bl, bi)) {
                bl = l;
                bi = s;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ol = __shfl_xor(bl, o);
            const int oi = __shfl_xor(bi, o);
            if (oi != 0x7fffffff && (bi == 0x7fffffff || mh_min_better(ol, oi, bl, bi))) {
                bl = ol;
                bi = oi;
            }
        }
        if (lane == 0) {
            s_rl[r] = bl;
            s_ri[r] = bi;
            s_rh[r] = s_pos[r * S + bi];
        }
    }
    __syncthreads();

    // ---- best candidate across base-view ranks (PMVO.py:57-70) and the 3D direction (:73-74)
    if (tid == 0) {
        float ml = s_rl[0];
        int br = 0, bs = s_ri[0], hc = s_rh[0];
        for (int r = 1; r < nvalid; ++r) {
            const float l = s_rl[r];
            if ((l < ml) && (s_bval[r] > 0.0f)) {
                ml = l;
                br = r;
                bs = s_ri[r];
                hc = s_rh[r];
            }
        }
        float B0, B1, B2;
        mh_sample_item(s_rank + 4 * br, offs[bs], B0, B1, B2);
        const float d0 = B0 - P0, d1 = B1 - P1x, d2 = B2 - P2;
        float s2 = d0 * d0;
        s2 = mh_fma(d1, d1, s2);
        s2 = mh_fma(d2, d2, s2);
        const float nrm = __builtin_sqrtf(s2);
        line_ori[3 * n] = d0 / nrm;
        line_ori[3 * n + 1] = d1 / nrm;
        line_ori[3 * n + 2] = d2 / nrm;
        min_loss[n] = ml;
        high_conf[n] = (uint8_t)hc;
        if (best_sample) {
            best_sample[3 * n] = B0;
            best_sample[3 * n + 1] = B1;
            best_sample[3 * n + 2] = B2;
        }
        if (best_rank) best_rank[n] = br;
        if (best_s) best_s[n] = bs;
    }
}

// Launch order of the search: points sorted by descending work = (taps of the views that see the point) x (item slices
// it needs).  mh_search_work_kernel: one lane per point -> work class (0 = heaviest) in order[0..N);
// mh_search_order_kernel: one workgroup -- histogram of the MH_ORDER_BUCKETS classes, exclusive scan, scatter of the
// point indices into order[N..2N).  The order inside a class is whatever the atomics give: it only decides WHEN a
// point is processed, never what is computed for it.
__global__ __launch_bounds__(256) void mh_search_work_kernel(const uint8_t *__restrict__ cnt, int V, int N, int P1,
                                                             const float *__restrict__ base_val, int nrank,
                                                             int rank_step, int S, int T, int32_t *__restrict__ order,
                                                             int tail_n0) {
    // one wave per point: lanes over the views (list lengths, 0 for views that do not see the point) and over the ranks
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + wave;
    if (n >= N) return;
    int nt = 0;
    for (int v = lane; v < V; v += MH_WAVE) nt += cnt[(size_t)v * N + n];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nt += __shfl_xor(nt, o);
    const bool usable = lane > 0 && lane < nrank && base_val[(size_t)(lane * rank_step) * N + n] > 0.0f;
    const unsigned long long m = __ballot(usable);
    const int nvalid = m ? (64 - __builtin_clzll(m)) : 1;   // last usable rank + 1
    if (lane == 0) order[n] = n >= tail_n0 ? 0 : mh_work_class(nt, nvalid, V, P1, S, T);   // (MhWorkArgs::tail_n0)
}

// Points per (rank, base view) of the batch -- the M of mh_group_forms: gcnt[r * V + b] = #{n : base_idx[r * rank_step, n] == b}.
// Workgroup (x, r) counts 256 points of rank r in LDS and adds its non-zero cells to the zeroed global array.  (The fused
// forward does not launch this: its ranking kernel adds the counts itself, mh_topk_wave_kernel.  A first form -- one
// workgroup, all ranks, inside mh_search_order_kernel -- took 146 us at the headline size: up to 670 LDS atomics on one address.)
__global__ __launch_bounds__(256) void mh_group_sizes_kernel(const int32_t *__restrict__ base_idx, int N, int V,
                                                             int rank_step, int32_t *__restrict__ gcnt) {
    extern __shared__ int s_g[];   // V
    const int r = blockIdx.y, tid = threadIdx.x;
    for (int i = tid; i < V; i += 256) s_g[i] = 0;
    __syncthreads();
    const int n = blockIdx.x * 256 + tid;
    if (n < N) {
        const int b = base_idx[(size_t)(r * rank_step) * N + n];
        if (b >= 0 && b < V) atomicAdd(&s_g[b], 1);
    }
    __syncthreads();
    for (int i = tid; i < V; i += 256)
        if (s_g[i]) atomicAdd(&gcnt[(((int)blockIdx.x & (MH_GROUP_COPIES - 1)) * MH_GROUP_RANKS + r) * V + i], s_g[i]);
}

__global__ __launch_bounds__(1024) void mh_search_order_kernel(int N, int32_t *__restrict__ order) {
    __shared__ int s_hist[MH_ORDER_BUCKETS];
    __shared__ int s_part[1024 / 64];
    const int tid = threadIdx.x;
    s_hist[tid] = 0;
    __syncthreads();
    for (int n = tid; n < N; n += 1024) atomicAdd(&s_hist[order[n]], 1);
    __syncthreads();
    const int mine = s_hist[tid];   // lane = class
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int x = __shfl_up(incl, o);
        if ((tid & 63) >= o) incl += x;
    }
    if ((tid & 63) == 63) s_part[tid >> 6] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < (tid >> 6); ++w) base += s_part[w];
    __syncthreads();
    s_hist[tid] = base + incl - mine;   // first position of this class
    __syncthreads();
    for (int n = tid; n < N; n += 1024) order[N + atomicAdd(&s_hist[order[n]], 1)] = n;
}

// ---------------------------------------------------------------------------------------------
// PMVO.refine's loss of ONE given direction per point (PMVO.py:86-90): next = p + dir*mul/div,
// compute_reproject_ori + compute_prj_loss with S = 1 (then `low_conf_index` is always true and the raw
// num/den is returned, PMVO.py:199-204).  One wave per point, lane = view; the per-view terms go through
// LDS so that lane 0 can add them in ATen's cascade order.  Patches are read raw ([V,N,P,..] layout).
// ---------------------------------------------------------------------------------------------
#define MH_REFINE_VMAX 512
__global__ __launch_bounds__(256) void mh_refine_loss_kernel(MhViews vw, const float *__restrict__ pts,
                                                             const float *__restrict__ dir, float mul, float dv,
                                                             int N, int P, float thr, const float *__restrict__ vis,
                                                             const float *__restrict__ ori_patch,
                                                             const float *__restrict__ conf_patch,
                                                             float *__restrict__ loss, uint8_t *__restrict__ hcout,
                                                             MhBatch bt) {
    __shared__ float s_num[4][MH_REFINE_VMAX], s_den[4][MH_REFINE_VMAX];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + wave;
    if (n >= N) return;
    const int V = vw.V;
    const float Hf = (float)vw.H, Wf = (float)vw.W;
    const float P0 = pts[3 * n], P1 = pts[3 * n + 1], P2 = pts[3 * n + 2];
    const float Q0 = P0 + dir[3 * n] * mul / dv, Q1 = P1 + dir[3 * n + 1] * mul / dv,
                Q2 = P2 + dir[3 * n + 2] * mul / dv;
    // a batch of ONE point: its [V,1] sums over the views are ATen's inner sums whenever the outer-sum rule is on (sum_block > 0)
    // and -- with the batch rule of the products (reproject_rule 0) -- its projections are single-column products
    const bool one_point = mh_batch_single(bt, n);
    const bool single = bt.single_ok && one_point;
    for (int v = lane; v < V; v += MH_WAVE) {
        const float *cam = vw.cams + v * MH_CAM_STRIDE;
        float r0, c0, r1, c1, dx, dy;
        mh_pixel_of_b(cam, P0, P1, P2, Hf, Wf, r0, c0, single);
        mh_pixel_of_b(cam, Q0, Q1, Q2, Hf, Wf, r1, c1, single);
        mh_unit2(r1 - r0, c1 - c0, dx, dy);
        const size_t vn = (size_t)v * N + n;
        const float *__restrict__ cp = conf_patch + vn * P;
        const float2 *__restrict__ op = reinterpret_cast<const float2 *>(ori_patch) + vn * P;
        float cmax = cp[0];
        for (int p = 1; p < P; ++p) cmax = (cp[p] > cmax) ? cp[p] : cmax;
        const bool hc = cmax > thr;
        float ml = 0.f, bc = 0.f;
        for (int p = 0; p < P; ++p) {
            float o0, o1;
            const float2 o = op[p];
            mh_unit2(o.x, o.y, o0, o1);
            const float cs = o0 * dx + o1 * dy;
            const float l = 1.0f - __builtin_fabsf(cs);
            const float c = cp[p];
            const bool upd = (p == 0) || ((l < ml) && (hc ? (c > thr) : true));
            ml = upd ? l : ml;
            bc = upd ? c : bc;
        }
        const float w = (vis[vn] == -1.0f ? 0.0f : 1.0f) * bc;
        s_num[wave][v] = ml * w;
        s_den[wave][v] = w;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        MhCascV nm = {0.f, 0.f, 0.f}, dn = {0.f, 0.f, 0.f};
        int cnt = 0;
        for (int v = 0; v < V; ++v) {
            if (v > 0 && (v & 15) == 0) {
                mh_cascv_flush(nm, v);
                mh_cascv_flush(dn, v);
            }
            const float w = s_den[wave][v];
            nm.a0 = nm.a0 + s_num[wave][v];
            dn.a0 = dn.a0 + w;
            cnt += (w > 0.0f) ? 1 : 0;
        }
        float d = mh_cascv_done(dn), m = mh_cascv_done(nm);
        if (one_point && bt.block > 0) {   // [V, 1]: ATen's sum over a contiguous innermost dimension
            m = mh_inner_sum_views(V, [&](int v) { return s_num[wave][v]; });
            d = mh_inner_sum_views(V, [&](int v) { return s_den[wave][v]; });
        } else if (mh_tail_row(bt, n)) {   // a trailing column of the batch's [V, N] sums (ATen's row_sum order)
            m = mh_row_sum_views(V, [&](int v) { return s_num[wave][v]; });
            d = mh_row_sum_views(V, [&](int v) { return s_den[wave][v]; });
        }
        loss[n] = m / d;
        if (hcout) hcout[n] = (d / (float)cnt > thr) ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------------------------
// The same loss straight from the maps (refine's smoothing loop, PMVO.py:602-650, calls PMVO.refine once per 5000-point
// chunk): projection, visibility and the patch of every view that sees the point are evaluated in the kernel, the
// [V,N,P,..] patch tensors (365 MB per chunk at the headline size) are never written.  Per (view, point) the
// operations are those of mh_project_gather_kernel followed by mh_refine_loss_kernel, in the same order, so the
// result is bit-identical to the two-kernel path; views that do not see the point have weight 0 (PMVO.py:212) and
// are skipped, as in mh_search_kernel.
// ---------------------------------------------------------------------------------------------
// Round 6: lane = TAP for the patches.  A wave owns one point.  Phase 1 (lane = view, 64 views at a time): projection,
// depth test, the projected direction of the candidate.  Phase 2: the views that see the point are walked on the ballot
// mask; for each, the wave's lanes gather the P taps of the patch as PATCH contiguous runs (one coalesced request per view
// instead of 2 P per-lane gathers with one address per view -- the round-1..5 form spent 5.1 ms per 288 k points, 4 % of
// HBM), evaluate 1 - |cos| one tap per lane, and find (a) the patch maximum of the confidence and (b) the lexicographic
// minimum of (loss, tap index) over tap 0 and the eligible taps with two shuffle reductions.  That IS the sequential rule
// of compute_prj_loss (PMVO.py:160-190: tap 0 unconditionally, a later tap only if strictly smaller and eligible), NaN
// cases included: a NaN loss never wins a `<`; a NaN at tap 0 stays.  The next view's taps are requested before the
// current view is reduced.  Per-view terms go to LDS and lane 0 adds them in ATen's order, as before.
template <int PATCH>
__global__ __launch_bounds__(256) void mh_refine_loss_maps_kernel(MhViews vw, const float *__restrict__ pts,
                                                                  const float *__restrict__ dir, float mul, float dv,
                                                                  int N, float thr, float *__restrict__ loss,
                                                                  uint8_t *__restrict__ hcout, MhBatch bt) {
    constexpr int P = PATCH * PATCH, HP = PATCH / 2, ROUNDS = (P + MH_WAVE - 1) / MH_WAVE;
    __shared__ float s_num[4][MH_REFINE_VMAX], s_den[4][MH_REFINE_VMAX];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + wave;
    if (n >= N) return;
    const int V = vw.V, H = vw.H, W = vw.W;
    const float Hf = (float)H, Wf = (float)W;
    const float P0 = pts[3 * n], P1 = pts[3 * n + 1], P2 = pts[3 * n + 2];
    const float Q0 = P0 + dir[3 * n] * mul / dv, Q1 = P1 + dir[3 * n + 1] * mul / dv,
                Q2 = P2 + dir[3 * n + 2] * mul / dv;
    // a batch of ONE point: its [V,1] sums over the views are ATen's inner sums whenever the outer-sum rule is on (sum_block > 0)
    // and -- with the batch rule of the products (reproject_rule 0) -- its projections are single-column products
    const bool one_point = mh_batch_single(bt, n);
    const bool single = bt.single_ok && one_point;
    int ti[ROUNDS], tj[ROUNDS];
#pragma unroll
    for (int t = 0; t < ROUNDS; ++t) {
        const int p = min(lane + MH_WAVE * t, P - 1);
        ti[t] = p / PATCH - HP;
        tj[t] = p - (p / PATCH) * PATCH - HP;
    }
    auto rdf = [](float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); };
    for (int v0 = 0; v0 < V; v0 += MH_WAVE) {
        const int v = v0 + lane;
        float visv = -1.0f, dx = 0.0f, dy = 0.0f;
        int r = 0, c = 0;
        if (v < V) {
            const float *cam = vw.cams + v * MH_CAM_STRIDE;
            float u, w, z, r0, c0;
            mh_cam_project_b(cam, P0, P1, P2, u, w, z, single);
            mh_ndc_to_pixel(u, w, Hf, Wf, r0, c0);
            float cr = __builtin_rintf(c0), rr = __builtin_rintf(r0);
            const bool oob = !(cr <= (float)(W - 1)) || (cr < 0.0f) || !(rr <= (float)(H - 1)) || (rr < 0.0f);
            cr = fminf(fmaxf(cr, 0.0f), (float)(W - 1));
            rr = fminf(fmaxf(rr, 0.0f), (float)(H - 1));
            r = (int)rr;
            c = (int)cr;
            const float4 q = vw.rec[(size_t)v * H * W + (size_t)r * W + c];
            visv = oob ? -1.0f : mh_soft_visible(q.w, (-z / 2.0f) * 255.0f);
            if (visv != -1.0f) {
                float r1, c1;
                mh_pixel_of_b(cam, Q0, Q1, Q2, Hf, Wf, r1, c1, single);
                mh_unit2(r1 - r0, c1 - c0, dx, dy);
            } else {
                s_num[wave][v] = 0.0f;
                s_den[wave][v] = 0.0f;
            }
        }
        unsigned long long m = __ballot(visv != -1.0f);
        // taps of one view: {unit ori_row, unit ori_col, clamped conf} per lane and round
        float o0[ROUNDS], o1[ROUNDS], cf[ROUNDS], no0[ROUNDS], no1[ROUNDS], ncf[ROUNDS];
        auto gather = [&](int src, float *a0, float *a1, float *ac) {
            const int rv = __builtin_amdgcn_readlane(r, src), cv = __builtin_amdgcn_readlane(c, src);
            const size_t base = (size_t)(v0 + src) * H * W;
#pragma unroll
            for (int t = 0; t < ROUNDS; ++t) {
                const int r2 = min(max(rv + ti[t], 0), H - 1), c2 = min(max(cv + tj[t], 0), W - 1);
                if (vw.tap) {   // (the plane of ready-made taps, MhViews::tap: the same values, made once at upload)
                    const float4 tq = vw.tap[base + (size_t)r2 * W + c2];
                    a0[t] = tq.x;
                    a1[t] = tq.y;
                    ac[t] = tq.z;
                } else {
                    const float4 tq = vw.rec[base + (size_t)r2 * W + c2];
                    mh_unit2(tq.x, tq.y, a0[t], a1[t]);
                    ac[t] = mh_clampf(tq.z, 1e-6f, 1.0f);
                }
            }
        };
        int src = m ? __builtin_amdgcn_readfirstlane(__builtin_ctzll(m)) : 0;
        if (m) gather(src, no0, no1, ncf);
        while (m) {
            const int cur = src;
            m &= m - 1;
#pragma unroll
            for (int t = 0; t < ROUNDS; ++t) {
                o0[t] = no0[t];
                o1[t] = no1[t];
                cf[t] = ncf[t];
            }
            if (m) {
                src = __builtin_amdgcn_readfirstlane(__builtin_ctzll(m));
                gather(src, no0, no1, ncf);
            }
            const float dxv = rdf(dx, cur), dyv = rdf(dy, cur);
            // (a) cmax as `cmax = (p == 0 || cf > cmax) ? cf : cmax` leaves it: the maximum, NaNs skipped -- unless tap 0 is NaN
            const float cf0 = rdf(cf[0], 0);
            float mx = -__builtin_inff();
#pragma unroll
            for (int t = 0; t < ROUNDS; ++t)
                if (lane + MH_WAVE * t < P && cf[t] == cf[t]) mx = fmaxf(mx, cf[t]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            const float cmax = (cf0 != cf0) ? cf0 : mx;
            const bool hc = cmax > thr;
            // (b) lexicographic minimum of (loss, tap) over tap 0 and the eligible taps
            float bl = __builtin_inff(), bcf = 0.0f;
            int bp = 0x7fffffff;
            float l0 = 0.0f;
#pragma unroll
            for (int t = 0; t < ROUNDS; ++t) {
                const int p = lane + MH_WAVE * t;
                const float cs = o0[t] * dxv + o1[t] * dyv;
                const float l = 1.0f - __builtin_fabsf(cs);
                if (t == 0) l0 = l;
                const bool cand = p < P && (p == 0 || ((hc ? (cf[t] > thr) : true) && l == l));
                if (cand && (bp == 0x7fffffff || l < bl)) {   // (rounds ascend in p: a tie keeps the earlier tap)
                    bl = l;
                    bp = p;
                    bcf = cf[t];
                }
            }
            l0 = rdf(l0, 0);
            const float c0v = cf0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ol = __shfl_xor(bl, o), oc = __shfl_xor(bcf, o);
                const int op = __shfl_xor(bp, o);
                const bool take = op != 0x7fffffff && (bp == 0x7fffffff || ol < bl || (ol == bl && op < bp));
                bl = take ? ol : bl;
                bcf = take ? oc : bcf;
                bp = take ? op : bp;
            }
            // a NaN at tap 0 is never replaced (`l < NaN` is false for every later tap)
            const float ml = (l0 != l0) ? l0 : bl, bc = (l0 != l0) ? c0v : bcf;
            if (lane == 0) {
                s_num[wave][v0 + cur] = ml * bc;
                s_den[wave][v0 + cur] = bc;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        MhCascV nm = {0.f, 0.f, 0.f}, dn = {0.f, 0.f, 0.f};
        int cnt = 0;
        for (int v = 0; v < V; ++v) {
            if (v > 0 && (v & 15) == 0) {
                mh_cascv_flush(nm, v);
                mh_cascv_flush(dn, v);
            }
            const float w = s_den[wave][v];
            nm.a0 = nm.a0 + s_num[wave][v];
            dn.a0 = dn.a0 + w;
            cnt += (w > 0.0f) ? 1 : 0;
        }
        float d = mh_cascv_done(dn), m = mh_cascv_done(nm);
        if (one_point && bt.block > 0) {   // [V, 1]: ATen's sum over a contiguous innermost dimension
            m = mh_inner_sum_views(V, [&](int v) { return s_num[wave][v]; });
            d = mh_inner_sum_views(V, [&](int v) { return s_den[wave][v]; });
        } else if (mh_tail_row(bt, n)) {   // a trailing column of the batch's [V, N] sums (ATen's row_sum order)
            m = mh_row_sum_views(V, [&](int v) { return s_num[wave][v]; });
            d = mh_row_sum_views(V, [&](int v) { return s_den[wave][v]; });
        }
        loss[n] = m / d;
        if (hcout) hcout[n] = (d / (float)cnt > thr) ? 1 : 0;
    }
}

// loss[n] <- -1 where the head filter fires (PMVO.py:91-92), the replacement rule of the smoothing loop on the
// orientations in place (:631-636, as mh_replace_dissimilar_kernel), and loss -1 -> 0.5 (:641-642) into loss_out
__global__ __launch_bounds__(256) void mh_refine_combine_kernel(const float *__restrict__ center,
                                                                const float *__restrict__ loss_u,
                                                                const uint8_t *__restrict__ head,
                                                                const uint8_t *__restrict__ head_top, float thr,
                                                                float *__restrict__ ori, float *__restrict__ loss_out,
                                                                int N) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const bool filt = head[n] && !head_top[n];
    const float ul = filt ? -1.0f : loss_u[n];
    loss_out[n] = (ul == -1.0f) ? 0.5f : ul;
    if (!ori) return;   // (the replacement was applied already: mh_replace_dissimilar in the chain of the smoothing loop)
    float c[3], o[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        c[k] = center[3 * n + k];
        o[k] = ori[3 * n + k];
    }
    float sc = c[0] * c[0];
    sc = mh_fma(c[1], c[1], sc);
    sc = mh_fma(c[2], c[2], sc);
    float so = o[0] * o[0];
    so = mh_fma(o[1], o[1], so);
    so = mh_fma(o[2], o[2], so);
    float nc = __builtin_sqrtf(sc), no = __builtin_sqrtf(so);
    nc = (nc < 1e-8f) ? 1e-8f : nc;
    no = (no < 1e-8f) ? 1e-8f : no;
    const float cs = ((c[0] / nc) * (o[0] / no) + (c[1] / nc) * (o[1] / no)) + (c[2] / nc) * (o[2] / no);
    if (__builtin_fabsf(cs) < thr) {
#pragma unroll
        for (int k = 0; k < 3; ++k) ori[3 * n + k] = c[k];
    }
}

// ---------------------------------------------------------------------------------------------
extern "C" int mh_launch_search(MhViews vw, const float *offs, int S, int nrank, int rank_step, const float *pts,
                                int N, int P1, float thr, const float *ori_c, const int32_t *base_idx,
                                const float *base_val, const float4 *taps, int32_t *order /* 2N ints of work space */,
                                const uint8_t *cnt /* [V,N] list lengths */,
                                float *line_ori, float *min_loss, uint8_t *high_conf, float *best_sample,
                                int32_t *best_rank, int32_t *best_s, int variant, int rule_mode, int fma_min_cols,
                                int sum_block, int32_t *gcnt /* MH_GROUP_COPIES * MH_GROUP_RANKS * V ints of work space */,
                                int groups_ready /* gcnt holds the batch's group sizes already (the fused forward) */,
                                hipStream_t st) {
    const int nitems = nrank * S;
    if (nitems > MH_MAX_ITEMS || nrank > MH_MAX_RANKS || nitems < 1) return -1;
    // the batch in the arithmetic (MhRule, mh_device.h): group sizes per (rank, base view) and the trailing columns of the
    // [V, N*S] sums.  rule_mode 0 needs gcnt (the C API hands it out of its scratch).
    if (sum_block < 0) return -1;
    MhRule rule;
    rule.mode = rule_mode;
    rule.fma_min_cols = fma_min_cols;
    rule.gcnt = (rule_mode == 0) ? gcnt : nullptr;
    if (rule_mode == 0 && !gcnt) return -1;
    const long long cols = (long long)N * S;
    rule.tail_col0 = sum_block ? cols - cols % sum_block : cols;
    // (variant 9 / 10: the launch split for measurements -- 9 runs what precedes the search (group sizes, work classes, launch
    // order) and stops, 10 runs mh_search3_kernel alone on what 9 left in the scratch; bench.py times the two with HIP events)
    const bool select_body = variant >= 100 && variant < 200;
    if (select_body) variant -= 100;
    if (variant == 10) groups_ready = 1;
    if (rule.gcnt && !groups_ready) {
        if (hipMemsetAsync(gcnt, 0, sizeof(int32_t) * (size_t)MH_GROUP_COPIES * MH_GROUP_RANKS * vw.V, st) != hipSuccess) return -1;
        hipLaunchKernelGGL(mh_group_sizes_kernel, dim3((N + 255) / 256, nrank), dim3(256), sizeof(int) * (size_t)vw.V, st,
                           base_idx, N, vw.V, rank_step, gcnt);
    }
    // variant 0 (default): mh_search3_kernel, workgroups in descending order of work; 7: the same in natural order (A/B);
    // 1256: the portable mh_search_kernel (cross-check) -- also what runs when the caller has no list lengths
    // (8: as 0, the work classes are in order[0..N) already -- the fused forward lets the ranking kernel write them)
    // (+100: the compare-and-select tap body of rounds 1-3 instead of the key body -- 100 / 107 are the A/B and cross-check
    // forms of 0 / 7)
    if (variant == 0) variant = cnt ? 6 : 1256;
    if (variant == 6 || variant == 7 || variant == 8 || variant == 9 || variant == 10) {
        if (!cnt) return -1;
        const int32_t *ord = nullptr;
        if (variant != 7 && order && N > 1) {
            if (variant == 6 || variant == 9)
                hipLaunchKernelGGL(mh_search_work_kernel, dim3((N + 3) / 4), dim3(256), 0, st, cnt, vw.V, N, P1, base_val,
                                   nrank, rank_step, S, 256, order, (int)(rule.tail_col0 / S));
            if (variant != 10) hipLaunchKernelGGL(mh_search_order_kernel, dim3(1), dim3(1024), 0, st, N, order);
            ord = order + N;
        }
        if (variant == 9) return (int)hipGetLastError();
#define MH_S3_LAUNCH(BIG, KEYS, BIGP)                                                                                   \
    hipLaunchKernelGGL((mh_search3_kernel<256, BIG, KEYS, BIGP>), dim3(N), dim3(256), 0, st, vw, offs, S, nrank, rank_step, \
                       pts, N, P1, thr, ori_c, base_idx, base_val, taps, cnt, ord, line_ori, min_loss, high_conf,       \
                       best_sample, best_rank, best_s, rule)
        // (the key kernel for lists of up to 64 taps -- every patch up to 8 x 8 -- keeps two key registers per item; the one
        // for longer lists two more)
        const bool bigp = P1 - 1 > 64;
        if (vw.V > 256) {
            if (select_body) MH_S3_LAUNCH(true, false, false);
            else if (bigp) MH_S3_LAUNCH(true, true, true);
            else MH_S3_LAUNCH(true, true, false);
        } else {
            if (select_body) MH_S3_LAUNCH(false, false, false);
            else if (bigp) MH_S3_LAUNCH(false, true, true);
            else MH_S3_LAUNCH(false, true, false);
        }
#undef MH_S3_LAUNCH
    } else if (variant == 1256) {
        hipLaunchKernelGGL((mh_search_kernel<4, 256>), dim3(N), dim3(256), 0, st, vw, offs, S, nrank, rank_step, pts, N, P1,
                           thr, ori_c, base_idx, base_val, taps, line_ori, min_loss, high_conf, best_sample, best_rank,
                           best_s, rule);
    } else {
        return -1;
    }
    return (int)hipGetLastError();
}

extern "C" int mh_launch_refine_loss_maps(MhViews vw, const float *pts, const float *dir, float mul, float dv, int N,
                                         int patch, float thr, float *loss, uint8_t *hc, int batch, long long row0,
                                         long long total, int sum_block, hipStream_t st) {
    if (vw.V > MH_REFINE_VMAX) return -1;
    const MhBatch bt = {row0, total, batch, sum_block, vw.batch_rule};
    const dim3 grid((N + 3) / 4), block(256);
#define MH_RM_CASE(PS)                                                                                               \
    case PS:                                                                                                         \
        hipLaunchKernelGGL(mh_refine_loss_maps_kernel<PS>, grid, block, 0, st, vw, pts, dir, mul, dv, N, thr, loss, hc, bt); \
        break;
    switch (patch) {
        MH_RM_CASE(1)
        MH_RM_CASE(3)
        MH_RM_CASE(5)
        MH_RM_CASE(7)
        MH_RM_CASE(9)
        MH_RM_CASE(11)
        default:
            return -1;
    }
#undef MH_RM_CASE
    return (int)hipGetLastError();
}

extern "C" int mh_launch_refine_combine(const float *center, const float *loss_u, const uint8_t *head,
                                        const uint8_t *head_top, float thr, float *ori, float *loss_out, int N,
                                        hipStream_t st) {
    hipLaunchKernelGGL(mh_refine_combine_kernel, dim3((N + 255) / 256), dim3(256), 0, st, center, loss_u, head,
                       head_top, thr, ori, loss_out, N);
    return (int)hipGetLastError();
}

extern "C" int mh_launch_refine_loss(MhViews vw, const float *pts, const float *dir, float mul, float dv, int N,
                                     int P, float thr, const float *vis, const float *ori_patch,
                                     const float *conf_patch, float *loss, uint8_t *hc, int sum_block, hipStream_t st) {
    if (vw.V > MH_REFINE_VMAX) return -1;
    const MhBatch bt = {0, N, 0, sum_block, vw.batch_rule};   // (the stand-alone method: its N points are one batch of the reference)
    hipLaunchKernelGGL(mh_refine_loss_kernel, dim3((N + 3) / 4), dim3(256), 0, st, vw, pts, dir, mul, dv, N, P, thr,
                       vis, ori_patch, conf_patch, loss, hc, bt);
    return (int)hipGetLastError();
}

// forces this translation unit's code object onto the device (HIP loads a fat binary on the first use of one of its kernels:
// 2-20 ms each, which a one-shot pass would pay in the middle of its stages); called from mh_ctx_create
extern "C" int mh_preload_pmvo_search() {
    hipFuncAttributes a;
    return (int)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&mh_search_order_kernel));
}
