// pmvo_project.hip -- view packing, project-and-gather (K3+K4+K5), base-view ranking (K6) and the
// tap-list preparation for the search kernel.  gfx950 only.
//
// HBM layout: every view is one array of 16-byte pixel records
// {ori_row, ori_col, conf, depth} plus one fp32 mask plane.  A patch tap is ONE aligned dwordx4
// load; the 7 (9) taps of a patch row are 112 (144) contiguous bytes.
#include "mh_device.h"

// ---------------------------------------------------------------------------------------------
// pack one view: reference layout (depth[H,W,Cd], ori[H,W,2], conf[H,W], mask[H,W,Cm]) -> records
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mh_pack_view_kernel(float4 *__restrict__ rec, float *__restrict__ maskp,
                                                           const float *__restrict__ depth, int dstride,
                                                           const float *__restrict__ ori,
                                                           const float *__restrict__ conf,
                                                           const float *__restrict__ mask, int mstride,
                                                           size_t npix, float4 *__restrict__ tapp) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < npix; i += step) {
        const float2 o = reinterpret_cast<const float2 *>(ori)[i];
        const float c = conf[i];
        rec[i] = make_float4(o.x, o.y, c, depth[i * dstride]);
        maskp[i] = mask[i * mstride];
        if (tapp) {   // the pixel as a patch tap (MhViews::tap): normalised and clamped here, once, instead of per iteration
            float o0, o1;
            mh_unit2(o.x, o.y, o0, o1);
            tapp[i] = make_float4(o0, o1, mh_clampf(c, 1e-6f, 1.0f), 0.0f);
        }
    }
}

// the same from the 8-bit files (Utils/PMVO_utils.py:255-313): pixel code -> loader value through a 256-entry LUT
__global__ __launch_bounds__(256) void mh_pack_view_u8_kernel(float4 *__restrict__ rec, float *__restrict__ maskp,
                                                              const float *__restrict__ depth, int dstride,
                                                              const uint8_t *__restrict__ ori,
                                                              const uint8_t *__restrict__ conf,
                                                              const uint8_t *__restrict__ mask,
                                                              const float4 *__restrict__ lut, size_t npix,
                                                              uint16_t *__restrict__ oc, float4 *__restrict__ tapp) {
    __shared__ float4 s_lut[256];
    s_lut[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < npix; i += step) {
        const uint8_t ko = ori[i], kc = conf[i];
        const float4 o = s_lut[ko];
        rec[i] = make_float4(o.x, o.y, s_lut[kc].z, depth[i * dstride]);
        maskp[i] = s_lut[mask[i]].w;
        if (oc) oc[i] = (uint16_t)(ko | (kc << 8));      // the two file codes themselves, kept resident for the tap gathers
        if (tapp) {   // (a context that also holds fp32 views keeps the tap plane complete)
            float o0, o1;
            mh_unit2(o.x, o.y, o0, o1);
            tapp[i] = make_float4(o0, o1, mh_clampf(s_lut[kc].z, 1e-6f, 1.0f), 0.0f);
        }
    }
}

// What a tap of an 8-bit map can be, per pixel code (derived once from the loaders' table): the unit orientation exactly as
// mh_unit2 makes it from the decoded (ori_row, ori_col), the clamped confidence, and the smallest code whose unit
// orientation has the same bits ("canonical" code: duplicate taps are recognised by it).
struct MhCodeTabs {
    float2 unit[256];
    float conf[256];
    uint8_t canon[256];
};
__global__ __launch_bounds__(256) void mh_code_tabs_kernel(const float4 *__restrict__ lut, MhCodeTabs *__restrict__ t) {
    __shared__ float2 s_u[256];
    const int c = threadIdx.x;
    const float4 e = lut[c];
    float o0, o1;
    mh_unit2(e.x, e.y, o0, o1);
    s_u[c] = make_float2(o0, o1);
    __syncthreads();
    int k = c;
    for (int j = c - 1; j >= 0; --j)
        if (__float_as_uint(s_u[j].x) == __float_as_uint(o0) && __float_as_uint(s_u[j].y) == __float_as_uint(o1)) k = j;
    t->unit[c] = make_float2(o0, o1);
    t->conf[c] = mh_clampf(e.z, 1e-6f, 1.0f);
    t->canon[c] = (uint8_t)k;
}

// ---------------------------------------------------------------------------------------------
// PMVO.Compute_Visible_and_Ori (PMVO.py:346-376): one workgroup = one view x 64 consecutive points.
//   phase 1: 64 lanes project their point (PMVO.project_points :378-397), fetch the centre record,
//            write vis / ori / conf / mask / pixf coalesced over n;
//   phase 2: all 256 lanes sweep the 64*P (point, tap) pairs of the tile in output order, so the
//            ori_patch / conf_patch stores are fully coalesced (8 B + 4 B per lane) and the taps of
//            one patch row are neighbouring lanes reading neighbouring 16-B records.
// Workgroups are renumbered so that each XCD (block b runs on XCD b%8) owns a contiguous range of
// views: its private L2 then holds the touched regions of ~V/8 maps instead of all V.
// ---------------------------------------------------------------------------------------------
#define MH_PG_TILE 64

template <int PATCH>
__global__ __launch_bounds__(256) void mh_project_gather_kernel(MhViews vw, const float *__restrict__ pts, int N,
                                                                int tiles, float *__restrict__ vis,
                                                                float *__restrict__ ori, float *__restrict__ conf,
                                                                float *__restrict__ mask,
                                                                float *__restrict__ ori_patch,
                                                                float *__restrict__ conf_patch,
                                                                float *__restrict__ pixf) {
    constexpr int P = PATCH * PATCH, HP = PATCH / 2;
    __shared__ int s_r[MH_PG_TILE], s_c[MH_PG_TILE];
    // the grid is padded to a multiple of 8 workgroups; XCD x (= blockIdx % 8) takes the x-th eighth of the
    // (view, tile) list, i.e. a contiguous range of views
    const int bid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (bid >= vw.V * tiles) return;
    const int v = bid / tiles, tile = bid - v * tiles;
    const int n0 = tile * MH_PG_TILE;
    const int tid = threadIdx.x;
    const int H = vw.H, W = vw.W;
    const float4 *__restrict__ rec = vw.rec + (size_t)v * H * W;

    if (tid < MH_PG_TILE) {
        const int n = n0 + tid;
        if (n < N) {
            const float *cam = vw.cams + v * MH_CAM_STRIDE;
            float u, w, z, rowf, colf;
            mh_cam_project_b(cam, pts[3 * n], pts[3 * n + 1], pts[3 * n + 2], u, w, z, vw.batch_rule && N == 1);
            mh_ndc_to_pixel(u, w, (float)H, (float)W, rowf, colf);
            // torch.round (half to even) -> long; bounds tests on the integers (PMVO.py:383-390)
            float cr = __builtin_rintf(colf), rr = __builtin_rintf(rowf);
            const bool oob = !(cr <= (float)(W - 1)) || (cr < 0.0f) || !(rr <= (float)(H - 1)) || (rr < 0.0f);
            cr = fminf(fmaxf(cr, 0.0f), (float)(W - 1));
            rr = fminf(fmaxf(rr, 0.0f), (float)(H - 1));
            const int r = (int)rr, c = (int)cr;
            s_r[tid] = r;
            s_c[tid] = c;
            const size_t pix = (size_t)r * W + c;
            const float4 q = rec[pix];
            const size_t vn = (size_t)v * N + n;
            if (vis) {
                const float zp = -z / 2.0f;
                float vb = mh_soft_visible(q.w, zp * 255.0f);
                vis[vn] = oob ? -1.0f : vb;
            }
            if (ori) reinterpret_cast<float2 *>(ori)[vn] = make_float2(q.x, q.y);
            if (conf) conf[vn] = mh_clampf(q.z, 1e-6f, 1.0f);
            if (mask) mask[vn] = vw.mask[(size_t)v * H * W + pix];
            if (pixf) reinterpret_cast<float2 *>(pixf)[vn] = make_float2(rowf, colf);
        }
    }
    if (!ori_patch && !conf_patch) return;
    __syncthreads();
    const int npts = min(MH_PG_TILE, N - n0);
    const int cnt = npts * P;
    const size_t obase = ((size_t)v * N + n0) * P;
    // 4 independent 16-B gathers in flight per lane before the first store: the kernel is bound by bytes in
    // flight (Little's law at ~2 us loaded HBM latency), not by issue.
    constexpr int UNR = 4;
    for (int base = 0; base < cnt; base += 256 * UNR) {
        float4 q[UNR];
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            const int idx = base + k * 256 + tid;
            if (idx < cnt) {
                const int nl = idx / P, p = idx - nl * P;
                const int i = p / PATCH - HP, j = p - (p / PATCH) * PATCH - HP;
                const int r = min(max(s_r[nl] + i, 0), H - 1);
                const int c = min(max(s_c[nl] + j, 0), W - 1);
                q[k] = rec[(size_t)r * W + c];
            }
        }
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            const int idx = base + k * 256 + tid;
            if (idx < cnt) {
                // streaming (non-temporal) stores: the 176 MB of patch output would otherwise wash the map lines
                // that neighbouring points re-read out of the XCD's 4 MB L2
                if (ori_patch)
                    __builtin_nontemporal_store(mh_v2f{q[k].x, q[k].y},
                                                reinterpret_cast<mh_v2f *>(ori_patch) + (obase + idx));
                if (conf_patch) __builtin_nontemporal_store(mh_clampf(q[k].z, 1e-6f, 1.0f), conf_patch + obase + idx);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// PMVO.Find_max_conf_from_visible_view (PMVO.py:339-343): one wave per point, rank-by-counting
// (V^2/64 compares per lane, no sort network needed for V <= 1024).
// ---------------------------------------------------------------------------------------------
#define MH_TOPK_VMAX 1024
__global__ __launch_bounds__(256) void mh_topk_kernel(const float *__restrict__ vis, const float *__restrict__ conf,
                                                      int V, int N, int32_t *__restrict__ out_idx,
                                                      float *__restrict__ out_val) {
    __shared__ float s_cv[4][MH_TOPK_VMAX];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + wave;
    if (n >= N) return;   // whole wave leaves together; no block-wide barrier below
    float *cv = s_cv[wave];
    for (int v = lane; v < V; v += MH_WAVE) {
        const float vb = vis[(size_t)v * N + n], c = conf[(size_t)v * N + n];
        cv[v] = (vb < 1.0f) ? c * fmaxf(vb, 0.0f) : c;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    for (int v = lane; v < V; v += MH_WAVE) {
        const float x = cv[v];
        int rank = 0;
        for (int u = 0; u < V; ++u) {
            const float y = cv[u];
            rank += (y > x) || (y == x && u < v);
        }
        if (rank < MH_TOPK) {
            out_idx[(size_t)rank * N + n] = v;
            out_val[(size_t)rank * N + n] = x;
        }
    }
}

#include "mh_topk_wave.h"
// torch.topk's CPU order, one WAVE per point (mh_topk_wave.h): the same steps, each scan / shift of the library one wave
// operation.  The W waves of a workgroup take W consecutive points (their strided column reads share 64-byte sectors);
// small workgroups spread the 5000 waves evenly over the SIMDs, which matters more: 33 us at W = 4, 41 us at W = 16
// (rank-by-counting: 11 us; the literal form: 130 us).
template <int R, int W>
__global__ __launch_bounds__(64 * W) void mh_topk_wave_kernel(const float *__restrict__ vis, const float *__restrict__ conf,
                                                            int V, int N, int32_t *__restrict__ out_idx,
                                                            float *__restrict__ out_val, MhWorkArgs wk) {
    __shared__ int s_stack[W][72];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x * W + wave;
    if (n >= N) return;   // whole wave leaves together; no block-wide barrier below
    MhTkWave<R> a;
    a.lane = lane;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int v = r * 64 + lane;
        int key = (int)0x80000000;    // below every value: padding lanes are never selected
        if (v < V) {
            const float vb = vis[(size_t)v * N + n], c = conf[(size_t)v * N + n];
            key = mh_tk_key((vb < 1.0f) ? c * fmaxf(vb, 0.0f) : c);
        }
        a.k[r] = key;
        a.i[r] = v;
    }
    a.topk(V, MH_TOPK, s_stack[wave]);
    float val = 0.0f;
    if (lane < MH_TOPK) {   // MH_TOPK <= 64: the result sits in register 0; the value is read again (keys fold NaN payloads)
        const int v = a.i[0];
        const float vb = vis[(size_t)v * N + n], c = conf[(size_t)v * N + n];
        val = (vb < 1.0f) ? c * fmaxf(vb, 0.0f) : c;
        out_idx[(size_t)lane * N + n] = v;
        out_val[(size_t)lane * N + n] = val;
    }
    if (wk.cls) {
        // the fused forward: the work class of the point for the search's launch order (what mh_search_work_kernel computes
        // from the same numbers -- this wave has the point's ranking in its lanes already; one launch less per iteration)
        int nt = 0;
        for (int v = lane; v < V; v += MH_WAVE) nt += wk.cnt[(size_t)v * N + n];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nt += __shfl_xor(nt, o);
        const int r = lane / wk.rank_step;
        const bool usable = lane > 0 && lane < MH_TOPK && r * wk.rank_step == lane && r < wk.nrank && val > 0.0f;
        const unsigned long long m = __ballot(usable);
        const int nvalid = m ? (63 - (int)__builtin_clzll(m)) / wk.rank_step + 1 : 1;   // last usable rank + 1
        if (lane == 0) wk.cls[n] = n >= wk.tail_n0 ? 0 : mh_work_class(nt, nvalid, V, wk.P1, wk.S, wk.T);
        // ... and the points per (rank, base view) of the batch (mh_device.h: MhRule): one fire-and-forget atomic per used
        // rank into the array the front end zeroed
        if (wk.gcnt && lane < MH_TOPK && r * wk.rank_step == lane && r < wk.nrank)
            atomicAdd(&wk.gcnt[(((int)blockIdx.x & (MH_GROUP_COPIES - 1)) * MH_GROUP_RANKS + r) * V + a.i[0]], 1);
    }
}

// ---------------------------------------------------------------------------------------------
// Tap lists for the search kernel.  For every (view, point) the per-tap work that does not depend on
// the candidate sample is done ONCE here instead of 900 times in the search:
//   - normalise the tap orientation the way torch.cosine_similarity does (PMVO.py:171),
//   - decide which taps can ever replace the running minimum (PMVO.py:162,177-182): tap 0 always;
//     tap p>=1 iff (patch max conf <= thr) or conf_p > thr,
//   - compact the eligible taps in their original order (strict '<' keeps first-index tie-breaking).
// Record (v,n): float4 header {count(bits), vis, pix_row, pix_col} followed by `count` float4 taps
// {ox_hat, oy_hat, conf, 0}; stride (P+1) float4.  One wave per (v,n), lane = tap.
// ---------------------------------------------------------------------------------------------
#define MH_PREP_PMAX 128
__global__ __launch_bounds__(256) void mh_prep_taps_kernel(const float *__restrict__ ori_patch,
                                                           const float *__restrict__ conf_patch,
                                                           const float *__restrict__ vis,
                                                           const float *__restrict__ pixf, int VN, int P, float thr,
                                                           float4 *__restrict__ taps, uint8_t *__restrict__ cnt) {
    __shared__ float2 s_o[4][MH_PREP_PMAX];
    __shared__ unsigned char s_el[4][MH_PREP_PMAX];
    __shared__ unsigned int s_first[4][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int vn = blockIdx.x * 4 + wave;
    if (vn >= VN) return;
    const float visv = vis[vn];
    float4 *__restrict__ out = taps + (size_t)vn * (P + 1);
    if (visv == -1.0f) {   // the search skips this view for this point: header only
        if (lane == 0) {
            out[0] = make_float4(__int_as_float(0), visv, 0.0f, 0.0f);
            cnt[vn] = 0;
        }
        return;
    }
    const float *__restrict__ cp = conf_patch + (size_t)vn * P;
    const float2 *__restrict__ op = reinterpret_cast<const float2 *>(ori_patch) + (size_t)vn * P;
    float cmax = -1.0f;
    for (int p = lane; p < P; p += MH_WAVE) cmax = fmaxf(cmax, cp[p]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cmax = fmaxf(cmax, __shfl_xor(cmax, o));
    const bool hc = cmax > thr;
    for (int b = lane; b < 256; b += MH_WAVE) s_first[wave][b] = 0xffffffffu;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // pass 1: normalise, eligibility; every eligible tap bids for "first tap with this orientation" in a
    // 256-bucket table keyed by a hash of the unit vector's bits (one wave owns its LDS slice)
    for (int p = lane; p < P; p += MH_WAVE) {
        float o0, o1;
        const float2 o = op[p];
        mh_unit2(o.x, o.y, o0, o1);
        const bool el = (p == 0) || (hc ? (cp[p] > thr) : true);
        s_o[wave][p] = make_float2(o0, o1);
        s_el[wave][p] = el;
        if (el) {
            const unsigned h = ((__float_as_uint(o0) * 0x9E3779B1u) ^ (__float_as_uint(o1) * 0x85EBCA77u)) >> 24;
            atomicMin(&s_first[wave][h], (unsigned)p);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // pass 2: a tap whose unit orientation is bit-identical to an EARLIER eligible tap can never win the strict
    // '<' of PMVO.py:177 (its loss equals that tap's for every candidate), so it is dropped -- exact, and on
    // 8-bit orientation maps (<= 180 distinct angles) it removes most of a patch.  A hash collision only means
    // that a duplicate survives (harmless); nothing distinct is ever dropped because the bits are compared.
    int base = 0;
    for (int p0 = 0; p0 < P; p0 += MH_WAVE) {
        const int p = p0 + lane;
        bool el = false;
        float2 o = make_float2(0.f, 0.f);
        if (p < P) {
            el = s_el[wave][p] != 0;
            o = s_o[wave][p];
            if (el) {
                const unsigned ox = __float_as_uint(o.x), oy = __float_as_uint(o.y);
                const unsigned q = s_first[wave][((ox * 0x9E3779B1u) ^ (oy * 0x85EBCA77u)) >> 24];
                if (q < (unsigned)p) {
                    const float2 e = s_o[wave][q];
                    if (__float_as_uint(e.x) == ox && __float_as_uint(e.y) == oy) el = false;
                }
            }
        }
        const unsigned long long m = __ballot(el);
        const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
        if (el) out[1 + pos] = make_float4(o.x, o.y, cp[p], 0.0f);
        base += __popcll(m);
    }
    if (lane == 0) {
        const float2 pf = reinterpret_cast<const float2 *>(pixf)[vn];
        out[0] = make_float4(__int_as_float(base), visv, pf.x, pf.y);
        cnt[vn] = (uint8_t)base;   // compact copy of the list lengths [V,N]: the search's work estimate reads it coalesced
    }
}

// ---------------------------------------------------------------------------------------------
// Fused front end of PMVO.forward: projection + visibility + centre samples (the part of
// Compute_Visible_and_Ori that the base-view ranking and the sampler need) AND the tap list of the search,
// straight from the packed maps -- the [V,N,P,..] patch tensors are never written or re-read, and for the
// (view, point) pairs whose depth test fails (weight 0 in the loss, ~2/3 of them on a closed surface) the
// patch is not even gathered.  Same arithmetic, same eligibility / duplicate rules and same record layout as
// mh_project_gather_kernel + mh_prep_taps_kernel (the unfused pair, kept as the cross-check).  Two kernels:
// mh_project_taps_codes_kernel for maps uploaded as 8-bit file codes, mh_project_taps2_kernel for fp32 records.
// One workgroup = one view x 64 consecutive points (the tiling and XCD mapping of mh_project_gather_kernel).
// (The first fused form -- lane = point for the whole tile, three dependent round trips per wave -- was removed in
// round 4; docs/HISTORY.md has its measurements.)
// ---------------------------------------------------------------------------------------------

// ---------------------------------------------------------------------------------------------
// The fused front end for maps uploaded as 8-bit FILE CODES (mh_ctx_set_view_u8 -- what every real capture is, SURVEY.md
// App. A.18): same outputs as the fp32 form (mh_project_taps2_kernel below), organised around what bounds it in this regime.  The fp32 form is a
// chain of three dependent memory round trips per wave (points -> centre record -> patch gather of the points that passed
// the depth test) at ~50 % occupancy, 64 % of its wave time parked on them (profiles/r03_8bit_*); its LDS work is 87 %
// bank-conflict cycles (49 lanes bidding for "first tap with this orientation" on one address).  Here
//   * a tap is the two CODES of its pixel, 2 bytes (a 7-tap patch row is 14 bytes; the touched part of a view shrinks 8x in
//     the XCD's L2); the patches of all visible points of a wave are requested back to back: up to 16 patch gathers in
//     flight per wave instead of 3;
//   * one wave owns 16 points of the tile (lane = point for the projection, lane = tap for the patches): no workgroup
//     barrier between the phases;
//   * unit orientation and clamped confidence of a code come from 256-entry LDS tables (no square root / division per tap);
//   * "the first eligible tap with this orientation" is found on the wave's lane masks: take the first remaining eligible
//     lane, ballot the lanes with the same canonical code, keep the first, drop the class -- as many steps as the patch
//     has distinct orientations (2.2 on average on 8-bit maps), no LDS, no atomics, no hash collisions.
// The list of a (view, point) holds the fp32 form's records in the same order, minus the duplicates that form keeps on a
// hash collision; the search result is bit-identical either way (a duplicate can never win the strict '<' of PMVO.py:177).
// ---------------------------------------------------------------------------------------------
template <int PATCH>
__global__ __launch_bounds__(256) void mh_project_taps_codes_kernel(MhViews vw, const float *__restrict__ pts, int N,
                                                                    int tiles, float thr, float *__restrict__ vis,
                                                                    float *__restrict__ ori, float *__restrict__ conf,
                                                                    float *__restrict__ mask, float4 *__restrict__ taps,
                                                                    uint8_t *__restrict__ cnt,
                                                                    const uint16_t *__restrict__ oc_all,
                                                                    const MhCodeTabs *__restrict__ tabs,
                                                                    int32_t *__restrict__ zero, int nzero) {
    constexpr int P = PATCH * PATCH, HP = PATCH / 2, NCH = (P + MH_WAVE - 1) / MH_WAVE, PW = 16;
    __shared__ float2 s_unit[256];
    __shared__ float s_confc[256];
    __shared__ uint8_t s_canon[256];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (blockIdx.x == 0)   // (the group sizes the ranking kernel behind this launch counts: mh_topk_wave_kernel)
        for (int i = tid; i < nzero; i += 256) zero[i] = 0;
    const int bid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);   // XCD-contiguous view ranges
    const int V = vw.V, H = vw.H, W = vw.W;
    if (bid >= V * tiles) return;
    const int v = bid / tiles, tile = bid - v * tiles;
    const int n0 = tile * MH_PG_TILE + wave * PW;            // this wave's 16 points
    // the tables travel while the wave projects and gathers; they are needed only when the first patch is decoded
    const float2 t_unit = tabs->unit[tid];
    const float t_conf = tabs->conf[tid];
    const uint8_t t_canon = tabs->canon[tid];
    const float4 *__restrict__ rec = vw.rec + (size_t)v * H * W;
    const uint16_t *__restrict__ oc = oc_all + (size_t)v * H * W;

    // lane = point: projection (PMVO.project_points, PMVO.py:378-397)
    const int n = n0 + lane;
    const bool mine = lane < PW && n < N;
    int r = 0, c = 0;
    float rowf = 0.0f, colf = 0.0f, z = 0.0f;
    bool oob = true;
    if (mine) {
        const float *cam = vw.cams + v * MH_CAM_STRIDE;
        float u, w;
        mh_cam_project_b(cam, pts[3 * n], pts[3 * n + 1], pts[3 * n + 2], u, w, z, vw.batch_rule && N == 1);
        mh_ndc_to_pixel(u, w, (float)H, (float)W, rowf, colf);
        float cr = __builtin_rintf(colf), rr = __builtin_rintf(rowf);
        oob = !(cr <= (float)(W - 1)) || (cr < 0.0f) || !(rr <= (float)(H - 1)) || (rr < 0.0f);
        cr = fminf(fmaxf(cr, 0.0f), (float)(W - 1));
        rr = fminf(fmaxf(rr, 0.0f), (float)(H - 1));
        r = (int)rr;
        c = (int)cr;
    }
    float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f);
    float mk = 0.0f;
    if (mine) {
        q0 = rec[(size_t)r * W + c];
        if (mask) mk = vw.mask[(size_t)v * H * W + (size_t)r * W + c];
    }
    // per-(view, point) outputs and the depth test (soft visibility, PMVO.py:346-376)
    float visv = -1.0f;
    if (mine) {
        visv = mh_soft_visible(q0.w, (-z / 2.0f) * 255.0f);
        visv = oob ? -1.0f : visv;
        const size_t vn = (size_t)v * N + n;
        vis[vn] = visv;
        reinterpret_cast<float2 *>(ori)[vn] = make_float2(q0.x, q0.y);
        conf[vn] = mh_clampf(q0.z, 1e-6f, 1.0f);
        if (mask) mask[vn] = mk;
        if (visv == -1.0f) {
            taps[vn * (P + 1)] = make_float4(__int_as_float(0), visv, 0.0f, 0.0f);
            cnt[vn] = 0;
        }
    }
    // lane = tap: the patch codes of the wave's VISIBLE points, all requested back to back (up to 16 gathers in flight per
    // wave).  (Requesting the patches of all 16 points before the depth test is known takes a dependent round trip off the
    // critical path but triples the scattered 2-byte loads the CU's address unit has to walk through: measured 51.5 us
    // against 49.5 us for this order.)
    int di[NCH], dj[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int p = ch * MH_WAVE + lane;
        di[ch] = p / PATCH - HP;
        dj[ch] = p - (p / PATCH) * PATCH - HP;
    }
    const int npw = min(PW, N - n0);                          // points of this wave (uniform; <= 0: nothing to do)
    unsigned k[PW][NCH];
#pragma unroll
    for (int j = 0; j < PW; ++j) {
        const int rj = __builtin_amdgcn_readlane(r, j), cj = __builtin_amdgcn_readlane(c, j);
        const float vj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(visv), j));
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            k[j][ch] = 0;
            if (j < npw && vj != -1.0f && ch * MH_WAVE + lane < P)
                k[j][ch] = oc[(size_t)min(max(rj + di[ch], 0), H - 1) * W + min(max(cj + dj[ch], 0), W - 1)];
        }
    }
    s_unit[tid] = t_unit;
    s_confc[tid] = t_conf;
    s_canon[tid] = t_canon;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PW; ++j) {
        const float vj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(visv), j));
        if (j < npw && vj != -1.0f) {                         // (uniform)
        float cc[NCH];
        unsigned kc[NCH];
        bool el[NCH];
        float cmax = -1.0f;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            cc[ch] = s_confc[k[j][ch] >> 8];
            kc[ch] = s_canon[k[j][ch] & 255u];
            if (ch * MH_WAVE + lane < P) cmax = fmaxf(cmax, cc[ch]);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cmax = fmaxf(cmax, __shfl_xor(cmax, o));
        const bool hc = cmax > thr;
        unsigned long long rem[NCH], keep[NCH];
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int p = ch * MH_WAVE + lane;
            el[ch] = (p < P) && ((p == 0) || (hc ? (cc[ch] > thr) : true));
            rem[ch] = __ballot(el[ch]);
            keep[ch] = 0ull;
        }
        // first eligible tap of every distinct (canonical) orientation code, in tap order
        while (true) {
            int c0 = -1;
#pragma unroll
            for (int ch = NCH - 1; ch >= 0; --ch)
                if (rem[ch]) c0 = ch;
            if (c0 < 0) break;
            unsigned long long rsel = rem[0];
            unsigned ksel = kc[0];
#pragma unroll
            for (int ch = 1; ch < NCH; ++ch)
                if (c0 == ch) rsel = rem[ch], ksel = kc[ch];
            const int l0 = __builtin_ctzll(rsel);
            const unsigned code = (unsigned)__builtin_amdgcn_readlane((int)ksel, l0);
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const unsigned long long m = __ballot(el[ch] && kc[ch] == code);
                rem[ch] &= ~m;
                if (ch == c0) keep[ch] |= 1ull << l0;
            }
        }
        const size_t vn = (size_t)v * N + n0 + j;
        float4 *__restrict__ out = taps + vn * (P + 1);
        int base = 0;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const unsigned long long m = keep[ch];
            if ((m >> lane) & 1ull) {
                const float2 o = s_unit[k[j][ch] & 255u];
                out[1 + base + __popcll(m & ((1ull << lane) - 1ull))] = make_float4(o.x, o.y, cc[ch], 0.0f);
            }
            base += __popcll(m);
        }
        if (lane == 0) {
            const float rfj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rowf), j));
            const float cfj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(colf), j));
            out[0] = make_float4(__int_as_float(base), vj, rfj, cfj);
            cnt[vn] = (uint8_t)base;
        }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// mh_project_taps2_kernel -- the fused front end for fp32 records in the organisation of the code form above (round 3): one
// wave owns 16 points of the 64-point tile (lane = point for the projection, lane = tap for the patches), no workgroup
// barrier between projection and patches, and the 16-byte patch gathers of up to INF visible points are in flight at once
// instead of three.  Arithmetic, eligibility, duplicate rule (hash table + bit compare) and record layout are those of
// the unfused pair mh_project_gather_kernel + mh_prep_taps_kernel: the tap lists are identical.
// ---------------------------------------------------------------------------------------------
// PW = points per wave (lane = point for the projection): 16 gives the kernel its best time alone (72 us; more waves = more
// of their dependent chains overlap), 64 the best ITERATION (fewer, longer waves take less slot time from the search that
// runs beside it on the other streams: +1.4 % iterations/s at 90 us alone) -- the metric is iterations/s, 64 is the default
// (round 5; option "taps_tile" 16 / 32 selects the others).
template <int PATCH, int PW>
__global__ __launch_bounds__(256) void mh_project_taps2_kernel(MhViews vw, const float *__restrict__ pts, int N, int tiles,
                                                               float thr, float *__restrict__ vis,
                                                               float *__restrict__ ori, float *__restrict__ conf,
                                                               float *__restrict__ mask, float4 *__restrict__ taps,
                                                               uint8_t *__restrict__ cnt, int32_t *__restrict__ zero,
                                                               int nzero) {
    constexpr int P = PATCH * PATCH, HP = PATCH / 2, NCH = (P + MH_WAVE - 1) / MH_WAVE;
    constexpr int INF = NCH == 1 ? 8 : 4;            // points whose patch gathers are in flight together (32 VGPRs)
    __shared__ float2 s_o[4][MH_PREP_PMAX];
    __shared__ float s_c[4][MH_PREP_PMAX];
    __shared__ unsigned char s_el[4][MH_PREP_PMAX];
    __shared__ unsigned int s_first[4][256];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (blockIdx.x == 0)   // (the group sizes the ranking kernel behind this launch counts: mh_topk_wave_kernel)
        for (int i = tid; i < nzero; i += 256) zero[i] = 0;
    const int bid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);   // XCD-contiguous view ranges
    const int V = vw.V, H = vw.H, W = vw.W;
    if (bid >= V * tiles) return;
    const int v = bid / tiles, tile = bid - v * tiles;
    const int n0 = tile * (4 * PW) + wave * PW;
    const float4 *__restrict__ rec = vw.rec + (size_t)v * H * W;
    // the patch taps: from the plane of ready-made taps when it is resident (uniform), else normalised here per tap
    const bool pre = vw.tap != nullptr;
    const float4 *__restrict__ tsrc = pre ? vw.tap + (size_t)v * H * W : rec;
    const int n = n0 + lane;
    const bool mine = lane < PW && n < N;
    int r = 0, c = 0;
    float rowf = 0.0f, colf = 0.0f, visv = -1.0f;
    if (mine) {
        const float *cam = vw.cams + v * MH_CAM_STRIDE;
        float u, w, z;
        mh_cam_project_b(cam, pts[3 * n], pts[3 * n + 1], pts[3 * n + 2], u, w, z, vw.batch_rule && N == 1);
        mh_ndc_to_pixel(u, w, (float)H, (float)W, rowf, colf);
        float cr = __builtin_rintf(colf), rr = __builtin_rintf(rowf);
        const bool oob = !(cr <= (float)(W - 1)) || (cr < 0.0f) || !(rr <= (float)(H - 1)) || (rr < 0.0f);
        cr = fminf(fmaxf(cr, 0.0f), (float)(W - 1));
        rr = fminf(fmaxf(rr, 0.0f), (float)(H - 1));
        r = (int)rr;
        c = (int)cr;
        const float4 q0 = rec[(size_t)r * W + c];
        visv = mh_soft_visible(q0.w, (-z / 2.0f) * 255.0f);
        visv = oob ? -1.0f : visv;
        const size_t vn = (size_t)v * N + n;
        vis[vn] = visv;
        reinterpret_cast<float2 *>(ori)[vn] = make_float2(q0.x, q0.y);
        conf[vn] = mh_clampf(q0.z, 1e-6f, 1.0f);
        if (mask) mask[vn] = vw.mask[(size_t)v * H * W + (size_t)r * W + c];
        if (visv == -1.0f) {
            taps[vn * (P + 1)] = make_float4(__int_as_float(0), visv, 0.0f, 0.0f);
            cnt[vn] = 0;
        }
    }
    const int npw = min(PW, N - n0);
    if (npw <= 0) return;
    const unsigned long long vmask = __ballot(mine && visv != -1.0f);     // bit j: point j of this wave is visible
    int di[NCH], dj[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int p = ch * MH_WAVE + lane;
        di[ch] = p / PATCH - HP;
        dj[ch] = p - (p / PATCH) * PATCH - HP;
    }
    unsigned long long todo = vmask;
    while (todo) {
        // the next INF visible points: their patches are requested back to back, then processed in turn
        int pj[INF];
        float4 q[INF][NCH];
        unsigned long long t = todo;
#pragma unroll
        for (int k = 0; k < INF; ++k) {
            pj[k] = t ? (int)__builtin_ctzll(t) : -1;
            t &= t - 1;
            if (pj[k] >= 0) {
                const int rj = __builtin_amdgcn_readlane(r, pj[k]), cj = __builtin_amdgcn_readlane(c, pj[k]);
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) {
                    q[k][ch] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (ch * MH_WAVE + lane < P)
                        q[k][ch] = tsrc[(size_t)min(max(rj + di[ch], 0), H - 1) * W + min(max(cj + dj[ch], 0), W - 1)];
                }
            }
        }
        todo = t;
#pragma unroll
        for (int k = 0; k < INF; ++k) {
            if (pj[k] < 0) continue;
            const int j = pj[k];
            const size_t vn = (size_t)v * N + n0 + j;
            float4 *__restrict__ out = taps + vn * (P + 1);
            for (int b = lane; b < 256; b += MH_WAVE) s_first[wave][b] = 0xffffffffu;
            float cmax = -1.0f;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int p = ch * MH_WAVE + lane;
                if (p < P) {
                    float cc = q[k][ch].z, o0 = q[k][ch].x, o1 = q[k][ch].y;
                    if (!pre) {
                        cc = mh_clampf(cc, 1e-6f, 1.0f);
                        mh_unit2(q[k][ch].x, q[k][ch].y, o0, o1);
                    }
                    s_o[wave][p] = make_float2(o0, o1);
                    s_c[wave][p] = cc;
                    cmax = fmaxf(cmax, cc);
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) cmax = fmaxf(cmax, __shfl_xor(cmax, o));
            const bool hc = cmax > thr;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int p = lane; p < P; p += MH_WAVE) {
                const bool el = (p == 0) || (hc ? (s_c[wave][p] > thr) : true);
                s_el[wave][p] = el;
                if (el) {
                    const float2 o = s_o[wave][p];
                    const unsigned h = ((__float_as_uint(o.x) * 0x9E3779B1u) ^ (__float_as_uint(o.y) * 0x85EBCA77u)) >> 24;
                    atomicMin(&s_first[wave][h], (unsigned)p);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            int base = 0;
            for (int p0 = 0; p0 < P; p0 += MH_WAVE) {
                const int p = p0 + lane;
                bool el = false;
                float2 o = make_float2(0.f, 0.f);
                float cc = 0.f;
                if (p < P) {
                    el = s_el[wave][p] != 0;
                    o = s_o[wave][p];
                    cc = s_c[wave][p];
                    if (el) {
                        const unsigned ox = __float_as_uint(o.x), oy = __float_as_uint(o.y);
                        const unsigned qf = s_first[wave][((ox * 0x9E3779B1u) ^ (oy * 0x85EBCA77u)) >> 24];
                        if (qf < (unsigned)p) {
                            const float2 e = s_o[wave][qf];
                            if (__float_as_uint(e.x) == ox && __float_as_uint(e.y) == oy) el = false;
                        }
                    }
                }
                const unsigned long long m = __ballot(el);
                const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
                if (el) out[1 + pos] = make_float4(o.x, o.y, cc, 0.0f);
                base += __popcll(m);
            }
            const float vjj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(visv), j));
            const float rfj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rowf), j));
            const float cfj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(colf), j));
            if (lane == 0) {
                out[0] = make_float4(__int_as_float(base), vjj, rfj, cfj);
                cnt[vn] = (uint8_t)base;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host-side launchers (called from capi.cpp)
// ---------------------------------------------------------------------------------------------
extern "C" int mh_launch_pack_view(float4 *rec, float *maskp, const float *depth, int dstride, const float *ori,
                                   const float *conf, const float *mask, int mstride, size_t npix,
                                   float4 *tapp, hipStream_t st) {
    const int blocks = (int)((npix + 255) / 256 < 4096 ? (npix + 255) / 256 : 4096);
    hipLaunchKernelGGL(mh_pack_view_kernel, dim3(blocks), dim3(256), 0, st, rec, maskp, depth, dstride, ori, conf,
                       mask, mstride, npix, tapp);
    return (int)hipGetLastError();
}

extern "C" int mh_launch_pack_view_u8(float4 *rec, float *maskp, const float *depth, int dstride, const uint8_t *ori,
                                      const uint8_t *conf, const uint8_t *mask, const float4 *lut, size_t npix,
                                      uint16_t *oc, float4 *tapp, hipStream_t st) {
    const int blocks = (int)((npix + 255) / 256 < 4096 ? (npix + 255) / 256 : 4096);
    hipLaunchKernelGGL(mh_pack_view_u8_kernel, dim3(blocks), dim3(256), 0, st, rec, maskp, depth, dstride, ori, conf,
                       mask, lut, npix, oc, tapp);
    return (int)hipGetLastError();
}

extern "C" size_t mh_code_tabs_bytes() { return sizeof(MhCodeTabs); }

extern "C" int mh_launch_code_tabs(const float4 *lut, void *tabs, hipStream_t st) {
    hipLaunchKernelGGL(mh_code_tabs_kernel, dim3(1), dim3(256), 0, st, lut, (MhCodeTabs *)tabs);
    return (int)hipGetLastError();
}

extern "C" int mh_launch_project_gather(MhViews vw, const float *pts, int N, int patch, float *vis, float *ori,
                                        float *conf, float *mask, float *ori_patch, float *conf_patch,
                                        float *pixf, hipStream_t st) {
    const int tiles = (N + MH_PG_TILE - 1) / MH_PG_TILE;
    const dim3 grid((vw.V * tiles + 7) & ~7), block(256);
#define MH_PG_CASE(PS)                                                                                          \
    case PS:                                                                                                    \
        hipLaunchKernelGGL(mh_project_gather_kernel<PS>, grid, block, 0, st, vw, pts, N, tiles, vis, ori, conf,  \
                           mask, ori_patch, conf_patch, pixf);                                                  \
        break;
    switch (patch) {
        MH_PG_CASE(1)
        MH_PG_CASE(3)
        MH_PG_CASE(5)
        MH_PG_CASE(7)
        MH_PG_CASE(9)
        MH_PG_CASE(11)
        default:
            return -1;
    }
#undef MH_PG_CASE
    return (int)hipGetLastError();
}

// wk.cls != nullptr (the fused forward; wave form only): the kernel also writes the work classes of the search's launch order
extern "C" int mh_launch_topk_work(const float *vis, const float *conf, int V, int N, int32_t *out_idx, float *out_val,
                                   int order, const uint8_t *cnt, int32_t *cls, int P1, int nrank, int rank_step, int S,
                                   int32_t *gcnt /* [nrank][V], zeroed: the batch's group sizes (or nullptr) */,
                                   int tail_n0 /* first point with trailing columns of the batch's sums (N: none) */,
                                   hipStream_t st) {
    if (V > MH_TOPK_VMAX || V < MH_TOPK) return -1;
    if (cls && ((order & 255) != 0 || !cnt || rank_step < 1)) return -1;
    if (gcnt && !cls) return -1;
    const MhWorkArgs wk{cnt, cls, gcnt, P1, nrank, rank_step < 1 ? 1 : rank_step, S, 256, tail_n0};
    if ((order & 255) == 1) {   // value descending, view index ascending among equal values (round 1's rule; A/B)
        hipLaunchKernelGGL(mh_topk_kernel, dim3((N + 3) / 4), dim3(256), 0, st, vis, conf, V, N, out_idx, out_val);
    } else {            // torch.topk's CPU order, one wave per point
        const int W = (order >> 8) == 16 ? 16 : ((order >> 8) == 8 ? 8 : 4);   // waves (= points) per workgroup; 4 measured best
        const dim3 grid((N + W - 1) / W), block(64 * W);
#define MH_TKW(RR)                                                                                                    \
    do {                                                                                                              \
        if (W == 4) hipLaunchKernelGGL((mh_topk_wave_kernel<RR, 4>), grid, block, 0, st, vis, conf, V, N, out_idx, out_val, wk); \
        else if (W == 8) hipLaunchKernelGGL((mh_topk_wave_kernel<RR, 8>), grid, block, 0, st, vis, conf, V, N, out_idx, out_val, wk); \
        else hipLaunchKernelGGL((mh_topk_wave_kernel<RR, 16>), grid, block, 0, st, vis, conf, V, N, out_idx, out_val, wk); \
    } while (0)
        if (V <= 64) MH_TKW(1);
        else if (V <= 128) MH_TKW(2);
        else if (V <= 256) MH_TKW(4);
        else if (V <= 512) MH_TKW(8);
        else MH_TKW(16);
#undef MH_TKW
    }
    return (int)hipGetLastError();
}

extern "C" int mh_launch_topk(const float *vis, const float *conf, int V, int N, int32_t *out_idx, float *out_val,
                              int order, hipStream_t st) {
    return mh_launch_topk_work(vis, conf, V, N, out_idx, out_val, order, nullptr, nullptr, 0, 0, 1, 0, nullptr, N, st);
}

extern "C" int mh_launch_prep_taps(const float *ori_patch, const float *conf_patch, const float *vis,
                                   const float *pixf, int VN, int P, float thr, float4 *taps, uint8_t *cnt,
                                   hipStream_t st) {
    if (P > MH_PREP_PMAX) return -1;
    hipLaunchKernelGGL(mh_prep_taps_kernel, dim3((VN + 3) / 4), dim3(256), 0, st, ori_patch, conf_patch, vis, pixf,
                       VN, P, thr, taps, cnt);
    return (int)hipGetLastError();
}

extern "C" int mh_launch_project_taps(MhViews vw, const float *pts, int N, int patch, float thr, float *vis,
                                      float *ori, float *conf, float *mask, float4 *taps, uint8_t *cnt, int tile,
                                      const uint16_t *oc, const void *tabs_v,
                                      int32_t *zero, int nzero /* ints the first workgroup clears (group sizes of the batch) */,
                                      hipStream_t st) {
    const MhCodeTabs *tabs = (const MhCodeTabs *)tabs_v;
    const bool codes = oc && tabs;
    if (patch * patch > MH_PREP_PMAX) return -1;
    // points per wave of the fp32 form: 64 (default; tile 0 / 1 / 64), 32 or 16 (A/B; 16 = the kernel's own best time).  The
    // codes form keeps 16.
    const int pw = codes ? 16 : (tile == 16 || tile == 32 ? tile : 64);
    const int tiles = (N + 4 * pw - 1) / (4 * pw);
    const dim3 grid((vw.V * tiles + 7) & ~7), block(256);
#define MH_PT_CASE(PS)                                                                                             \
    case PS:                                                                                                       \
        if (codes)                                                                                                 \
            hipLaunchKernelGGL((mh_project_taps_codes_kernel<PS>), grid, block, 0, st, vw, pts, N, tiles, thr, vis, ori, \
                               conf, mask, taps, cnt, oc, tabs, zero, nzero);                                      \
        else if (pw == 64)                                                                                         \
            hipLaunchKernelGGL((mh_project_taps2_kernel<PS, 64>), grid, block, 0, st, vw, pts, N, tiles, thr, vis, ori, conf, \
                               mask, taps, cnt, zero, nzero);                                                     \
        else if (pw == 32)                                                                                         \
            hipLaunchKernelGGL((mh_project_taps2_kernel<PS, 32>), grid, block, 0, st, vw, pts, N, tiles, thr, vis, ori, conf, \
                               mask, taps, cnt, zero, nzero);                                                     \
        else                                                                                                       \
            hipLaunchKernelGGL((mh_project_taps2_kernel<PS, 16>), grid, block, 0, st, vw, pts, N, tiles, thr, vis, ori, conf, \
                               mask, taps, cnt, zero, nzero);                                                     \
        break;
    switch (patch) {
        MH_PT_CASE(1)
        MH_PT_CASE(3)
        MH_PT_CASE(5)
        MH_PT_CASE(7)
        MH_PT_CASE(9)
        MH_PT_CASE(11)
        default:
            return -1;
    }
#undef MH_PT_CASE
    return (int)hipGetLastError();
}

// forces this translation unit's code object onto the device (HIP loads a fat binary on the first use of one of its kernels:
// 2-20 ms each, which a one-shot pass would pay in the middle of its stages); called from mh_ctx_create
extern "C" int mh_preload_pmvo_project() {
    hipFuncAttributes a;
    return (int)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&mh_pack_view_kernel));
}
