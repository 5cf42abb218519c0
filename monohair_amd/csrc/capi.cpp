// capi.cpp -- the C ABI of libmhpmvo.so (include/mh_pmvo.h): argument checking, the context that
// owns the packed maps, and the launch sequences.  All arithmetic lives in the .hip kernels.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/mh_pmvo.h"

struct MhViews {   // = MhViews of csrc/mh_device.h
    int V, H, W;
    const float4 *rec;
    const float *mask;
    const float *cams;
    const float4 *tap;
    int batch_rule;
};

#define MH_DG_MAXR 48
struct MhDogWeightsHost {          // = MhDogWeights of csrc/dog.hip
    double w[2][MH_DG_MAXR + 1];
    int r[2];
};

struct mh_ctx {
    int device = 0;
    int V = 0, H = 0, W = 0;
    float4 *rec = nullptr;    // [V][H][W]
    float *mask = nullptr;    // [V][H][W]
    float *cams = nullptr;    // [V][MH_CAM_STRIDE]
    // every pixel as a ready-made patch tap {unit ori, clamped conf} (MhViews::tap; 16 B per pixel more, 2 GB at 60 x 1080p):
    // allocated with the first view that comes in as fp32 planes, used by the fused front end once every view has been
    // written since; contexts of 8-bit code views never allocate it (their front end reads the 2 B codes)
    float4 *tapp = nullptr;
    bool tapp_failed = false;
    int use_tap_plane = 1;             // option "tap_plane": 0 = normalise per iteration (A/B, cross-check)
    // the plane doubles the resident map memory for +1.2 % iterations/s: only below this size (option "tap_plane_max_mb",
    // environment MH_TAP_PLANE_MAX_MB at context creation; 60 x 1080p = 1 991 MB fits, 120 x 4K = 15 925 MB does not) and
    // only while it leaves a quarter of the device's free memory to the scratch buffers of the drivers
    long long tap_plane_max_mb = 4096;
    std::vector<unsigned char> tap_view;   // per view: its slice of tapp is current
    const float4 *tap_ready() const {
        if (!tapp || !use_tap_plane || (int)tap_view.size() != V) return nullptr;
        for (unsigned char c : tap_view)
            if (!c) return nullptr;
        return tapp;
    }
    float *offs = nullptr;    // [S]
    float *gabor = nullptr;   // tap-major Gabor bank [289][192]
    float *gabor_q = nullptr; // the same coefficients in the operand order of mh_gabor_mfma2_kernel [145][64][8]
    unsigned int *gabor_max = nullptr;
    void *dog_w = nullptr;    // device MhDogWeights of the difference-of-Gaussians prefilter (csrc/dog.hip)
    MhDogWeightsHost *dog_w_host = nullptr;   // what dog_w holds
    float4 *lut = nullptr;    // [256] pixel-code table of the 8-bit map files
    // views uploaded as 8-bit file codes keep the codes resident as well (2 B per pixel: orientation | confidence << 8) for
    // the per-iteration tap gathers of mh_forward_prepare; used when EVERY view was uploaded that way with one table
    uint16_t *oc = nullptr;           // [V][H][W]
    bool oc_failed = false;           // the optional allocation of `oc` failed once: stay on the records
    void *code_tabs = nullptr;        // MhCodeTabs (csrc/pmvo_project.hip), derived from lut
    std::vector<unsigned char> code_view;   // per view: uploaded as codes
    float lut_host[1024];             // the table the resident records and code tables were made with
    bool lut_set = false, lut_mixed = false;
    int use_codes = 1;                // option "tap_codes": 0 = always gather the fp32 records (A/B, cross-check)
    bool codes_ready() const {
        if (!oc || !code_tabs || !use_codes || lut_mixed || (int)code_view.size() != V) return false;
        for (unsigned char c : code_view)
            if (!c) return false;
        return true;
    }
    bool views_8bit() const {   // every view came in as 8-bit file codes (whatever "tap_codes" says)
        if ((int)code_view.size() != V || V == 0) return false;
        for (unsigned char c : code_view)
            if (!c) return false;
        return true;
    }
    int S = 0;
    int search_variant = 0;
    int search_body = 0;      // tap body of mh_search3_kernel: 0 = by the maps (see mh_ctx_set_option), 1 = keys, 2 = select
    // The shipped search has two tap bodies with the same results (csrc/pmvo_search.hip): the key body (5.5 instructions per
    // evaluation, a fixed cost per view) and the compare-and-select body (7, none).  Lists of continuous maps hold ~45 taps,
    // lists of 8-bit maps ~2 after the exact duplicate removal: the kernel that carries both bodies runs short lists 5 %
    // slower than the select-only kernel (register allocation), so contexts whose views are all 8-bit codes get that one.
    int search_launch_variant(int v) const {
        if (v != 0 && v != 6 && v != 7 && v != 8 && v != 9 && v != 10) return v;   // (100.. = select body asked for; 1256 = portable kernel)
        const bool select = search_body == 2 || (search_body == 0 && views_8bit());
        return select ? (v == 0 ? 100 : v + 100) : v;
    }
    // The reference's batch composition in the arithmetic (csrc/mh_device.h: MhRule, MhBatch; oracle/pmvo_oracle.c):
    int reproject_rule = 0;   // 0: sample_next_3d_pos's sgemms round by the size of the (rank, base view) group as MKL does in
                              //    the reference; 1: the mid-size forms for every point; 2: the chain forms
    int reproject_fma_min_cols = 28445;   // columns (S x group) from which MKL's threaded sgemm (fma chain) takes over
    int sum_block = 32;       // ATen's outer sum adds the trailing (columns mod 32) of a batch in row_sum order; 0: never
    int topk_order = 0;       // 0: torch.topk's CPU tie order (mh_topk_wave.h); 1: value desc, view asc (round 1's rule)
    int filter_rows = 1;      // lab "filter_rows": 1 = votes of large launches with lane = point (mh_filter_rows_kernel), 0 = wave per point
    int taps_tile = 1;        // points per wave of mh_project_taps2_kernel: 16 / 32 (A/B), anything else = 64 (default)
    int line_rule = 0;        // strand renderer: 0 GL's diamond-exit, 1 every touched diamond (SwiftShader)
    int raster_subpixel_bits = 8;   // both rasterisers: window positions snapped to 2^-bits pixel (SwiftShader: 4)
    int gabor_variant = 3;    // 3: FP32-MFMA im2col contraction (default); 0: direct v_pk_fma form (cross-check).
                              // (1 and 2 named two forms removed in round 4.)
    MhViews views() const { return MhViews{V, H, W, rec, mask, cams, tap_ready(), reproject_rule == 0 ? 1 : 0}; }
};

// launchers implemented in the .hip files
extern "C" {
int mh_launch_pack_view(float4 *, float *, const float *, int, const float *, const float *, const float *, int,
                        size_t, float4 *, hipStream_t);
int mh_launch_pack_view_u8(float4 *, float *, const float *, int, const uint8_t *, const uint8_t *, const uint8_t *,
                           const float4 *, size_t, uint16_t *, float4 *, hipStream_t);
size_t mh_code_tabs_bytes();
int mh_preload_pmvo_project();
int mh_preload_pmvo_search();
int mh_preload_pmvo_filter();
int mh_preload_consensus();
int mh_preload_gabor();
int mh_preload_hairgrow();
int mh_preload_knn();
int mh_preload_raster();
int mh_preload_sortgroup();
int mh_preload_pmvo_pieces();
int mh_preload_dog();

int mh_launch_code_tabs(const float4 *, void *, hipStream_t);
int mh_launch_render_depth(const float *, const float *, int, const int32_t *, int, int, int, int, int, void *,
                           unsigned long long *, int32_t *, unsigned int *, float *, int, hipStream_t);
size_t mh_grid_scratch_bytes_impl(int);
size_t mh_sort_scratch_bytes_impl(int);
int mh_launch_grid_build(const float *, int, float, float, float, float, int, int, int, void *, size_t, float *,
                         int32_t *, int32_t *, int32_t *, hipStream_t);
int mh_launch_sort_keys(const unsigned long long *, int, int, void *, size_t, unsigned long long *, int32_t *,
                        hipStream_t);
size_t mh_voxel_group_scratch_bytes_impl(int);
size_t mh_select_scratch_bytes_impl(int);
int mh_launch_select_rows(const uint8_t *, const uint8_t *, int, int, const float *, const float *, float *, float *,
                          int32_t *, const int32_t *, int32_t *, void *, hipStream_t);
int mh_launch_segment_heads(const unsigned long long *, int, int32_t *, unsigned long long *, int32_t *, void *,
                            hipStream_t);
int mh_launch_flag_less(const float *, float, int, uint8_t *, hipStream_t);
int mh_launch_words_differ(const void *, const void *, size_t, int32_t *, hipStream_t);
int mh_launch_copy_words(const void *, void *, size_t, hipStream_t);
int mh_launch_points_bbox(const float *, int, float *, hipStream_t);
int mh_launch_voxel_group(const void *, int, const float *, int, const double *, double, const int32_t *, void *, size_t,
                          unsigned long long *, int32_t *, float *, hipStream_t);
int mh_launch_render_strands(const float *, const float *, int, const int32_t *, int, const float *, const float *, int,
                             int, int, int, int, int, int, int, int, float, void *, void *, unsigned long long *, int32_t *,
                             unsigned int *, float *, hipStream_t);
int mh_launch_project_points(const float *, const float *, int, int, int, int32_t *, float *, uint8_t *, float *, int,
                             hipStream_t);
int mh_launch_gather(MhViews, int, const long long *, int, int, float4 *, float *, hipStream_t);
int mh_launch_compute_visible(const float *, const float *, size_t, float *, hipStream_t);
int mh_launch_sample_next(MhViews, const float *, const int32_t *, const float *, const float *, int, int, float *, int, int,
                          int32_t *,
                          hipStream_t);
int mh_launch_reproject(MhViews, const float *, const float *, int, int, float *, hipStream_t);
int mh_launch_prj_loss(const float *, const float *, const float *, const float *, int, int, int, int, float, float *,
                       long long *, uint8_t *, float *, int, hipStream_t);
int mh_launch_project_gather(MhViews, const float *, int, int, float *, float *, float *, float *, float *, float *,
                             float *, hipStream_t);
int mh_launch_topk(const float *, const float *, int, int, int32_t *, float *, int, hipStream_t);
int mh_launch_topk_work(const float *, const float *, int, int, int32_t *, float *, int, const uint8_t *, int32_t *, int, int,
                        int, int, int32_t *, int, hipStream_t);
int mh_launch_prep_taps(const float *, const float *, const float *, const float *, int, int, float, float4 *,
                        uint8_t *, hipStream_t);
int mh_launch_project_taps(MhViews, const float *, int, int, float, float *, float *, float *, float *, float4 *,
                           uint8_t *, int, const uint16_t *, const void *, int32_t *, int, hipStream_t);
int mh_launch_search(MhViews, const float *, int, int, int, const float *, int, int, float, const float *,
                     const int32_t *, const float *, const float4 *, int32_t *, const uint8_t *, float *, float *,
                     uint8_t *, float *, int32_t *, int32_t *, int, int, int, int, int32_t *, int, hipStream_t);
int mh_launch_refine_loss(MhViews, const float *, const float *, float, float, int, int, float, const float *,
                          const float *, const float *, float *, uint8_t *, int, hipStream_t);
int mh_launch_filter_points(MhViews, const float *, int, int, float, float, uint8_t *, uint8_t *, uint8_t *,
                            uint8_t *, int, long long, long long, int, int, const int32_t *, hipStream_t);
int mh_launch_medoid_dense(const float *, const int32_t *, int, int, float *, int32_t *, hipStream_t);
int mh_launch_refine_loss_maps(MhViews, const float *, const float *, float, float, int, int, float, float *, uint8_t *,
                               int, long long, long long, int, hipStream_t);
int mh_launch_refine_combine(const float *, const float *, const uint8_t *, const uint8_t *, float, float *, float *,
                             int, hipStream_t);
int mh_launch_medoid_segmented(const float *, const int32_t *, int, int, float *, int32_t *, hipStream_t);
int mh_launch_gabor_bank(const float *, const float *, const float *, int, int, int32_t *, float *, float *, unsigned int *,
                         int, uint8_t *, uint8_t *, hipStream_t);
size_t mh_gabor_state_bytes();
size_t mh_gabor_bankq_bytes();
int mh_launch_gabor_relayout(const float *, float *, hipStream_t);
int mh_launch_dog(const void *, int, int, int, const void *, double *, double *, float *, hipStream_t);
int mh_launch_gabor_build(float *, hipStream_t);
int mh_launch_replace_dissimilar(const float *, float *, float, int, hipStream_t);
int mh_launch_knn(float, float, float, float, int, int, int, const float *, const int32_t *, const int32_t *,
                  const void *, int, int, int, int, const int32_t *, const uint8_t *, int32_t *, int32_t *, hipStream_t);
int mh_launch_nearest_dist(const float *, int, const double *, int, double *, double, double, uint8_t *, hipStream_t);
int mh_launch_pack_volume(const float *, const float *, size_t, float4 *, hipStream_t);
int mh_launch_trace_seeds(const float4 *, int, int, int, const float *, int, float, float *, int32_t *, int32_t *,
                          hipStream_t);
int mh_launch_strands_compact(const float *, const int32_t *, const int32_t *, const int64_t *, int, int, float *,
                              hipStream_t);
int mh_launch_trace_scalp(const float4 *, int, int, int, const float *, const float *, int, float, float *, int32_t *,
                          hipStream_t);
}

static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define MH_HIP(call)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) return fail(MH_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

static int launched(int rc, const char *what) {
    if (rc == -1) return fail(MH_ERR_ARG, "%s: unsupported size/shape", what);
    if (rc != 0) return fail(MH_ERR_HIP, "%s: launch failed: %s", what, hipGetErrorString((hipError_t)rc));
    return MH_OK;
}

extern "C" const char *mh_last_error(void) { return g_err; }
extern "C" int mh_version(void) { return 100; }

extern "C" int mh_ctx_create(int device_id, mh_ctx **out) {
    if (!out) return fail(MH_ERR_ARG, "mh_ctx_create: out is NULL");
    int ndev = 0;
    MH_HIP(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) return fail(MH_ERR_ARG, "mh_ctx_create: no device %d", device_id);
    mh_ctx *c = new (std::nothrow) mh_ctx();
    if (!c) return fail(MH_ERR_NOMEM, "mh_ctx_create: out of host memory");
    c->device = device_id;
    if (const char *e = getenv("MH_TAP_PLANE_MAX_MB")) c->tap_plane_max_mb = atoll(e);
    // the code objects of the library go onto the device now (HIP would load each translation unit's on the first launch
    // of one of its kernels -- in the middle of the first pass's stages); a failure here only means they load lazily
    if (hipSetDevice(device_id) == hipSuccess) {
        int (*const preload[])() = {mh_preload_pmvo_project, mh_preload_pmvo_search, mh_preload_pmvo_filter,
                                    mh_preload_consensus,    mh_preload_gabor,       mh_preload_hairgrow,
                                    mh_preload_knn,          mh_preload_raster,      mh_preload_sortgroup,
                                    mh_preload_pmvo_pieces,  mh_preload_dog};
        for (auto f : preload) (void)f();
        (void)hipGetLastError();
    }
    *out = c;
    return MH_OK;
}

static void free_views(mh_ctx *c) {
    if (c->rec) (void)hipFree(c->rec);
    if (c->mask) (void)hipFree(c->mask);
    if (c->cams) (void)hipFree(c->cams);
    if (c->oc) (void)hipFree(c->oc);
    c->oc = nullptr;
    if (c->tapp) (void)hipFree(c->tapp);
    c->tapp = nullptr;
    c->tapp_failed = false;
    c->tap_view.clear();
    c->code_view.clear();
    c->lut_set = c->lut_mixed = false;
    c->rec = nullptr;
    c->mask = nullptr;
    c->cams = nullptr;
    c->V = c->H = c->W = 0;
}

extern "C" void mh_ctx_destroy(mh_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    free_views(ctx);
    if (ctx->offs) (void)hipFree(ctx->offs);
    if (ctx->gabor) (void)hipFree(ctx->gabor);
    if (ctx->gabor_max) (void)hipFree(ctx->gabor_max);
    if (ctx->gabor_q) (void)hipFree(ctx->gabor_q);
    if (ctx->dog_w) (void)hipFree(ctx->dog_w);
    delete ctx->dog_w_host;
    if (ctx->lut) (void)hipFree(ctx->lut);
    if (ctx->code_tabs) (void)hipFree(ctx->code_tabs);
    delete ctx;
}

extern "C" int mh_ctx_alloc_views(mh_ctx *ctx, int V, int H, int W) {
    if (!ctx || V < 1 || H < 1 || W < 1) return fail(MH_ERR_ARG, "mh_ctx_alloc_views: bad arguments");
    MH_HIP(hipSetDevice(ctx->device));
    free_views(ctx);
    const size_t npix = (size_t)V * H * W;
    if (hipMalloc(&ctx->rec, npix * sizeof(float4)) != hipSuccess ||
        hipMalloc(&ctx->mask, npix * sizeof(float)) != hipSuccess ||
        hipMalloc(&ctx->cams, (size_t)V * MH_CAM_STRIDE * sizeof(float)) != hipSuccess) {
        free_views(ctx);
        return fail(MH_ERR_NOMEM, "mh_ctx_alloc_views: hipMalloc of %zu bytes failed", npix * 20);
    }
    ctx->V = V;
    ctx->H = H;
    ctx->W = W;
    ctx->code_view.assign(V, 0);
    ctx->tap_view.assign(V, 0);
    return MH_OK;
}

extern "C" int mh_ctx_set_view(mh_ctx *ctx, int view, const float *cam_host, const float *depth, int depth_stride,
                               const float *ori, const float *conf, const float *mask, int mask_stride,
                               void *stream) {
    if (!ctx || !ctx->rec) return fail(MH_ERR_STATE, "mh_ctx_set_view: views not allocated");
    if (view < 0 || view >= ctx->V || !cam_host || !depth || !ori || !conf || !mask || depth_stride < 1 ||
        mask_stride < 1)
        return fail(MH_ERR_ARG, "mh_ctx_set_view: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const size_t npix = (size_t)ctx->H * ctx->W;
    // pageable host -> device copy of 192 bytes: synchronous w.r.t. the host buffer, ordered on `st`
    MH_HIP(hipMemcpyAsync(ctx->cams + (size_t)view * MH_CAM_STRIDE, cam_host, MH_CAM_STRIDE * sizeof(float),
                          hipMemcpyHostToDevice, st));
    if ((int)ctx->code_view.size() == ctx->V) ctx->code_view[view] = 0;      // this view has no resident codes (any more)
    // the plane of ready-made taps (MhViews::tap): optional, like the code plane of the 8-bit views
    if (!ctx->tapp && !ctx->tapp_failed && ctx->use_tap_plane) {
        MH_HIP(hipSetDevice(ctx->device));
        const size_t bytes = (size_t)ctx->V * npix * sizeof(float4);
        size_t free_b = 0, total_b = 0;
        const bool fits = bytes <= (size_t)(ctx->tap_plane_max_mb > 0 ? ctx->tap_plane_max_mb : 0) * 1048576ull &&
                          hipMemGetInfo(&free_b, &total_b) == hipSuccess && bytes <= free_b - free_b / 4;
        if (!fits || hipMalloc(&ctx->tapp, bytes) != hipSuccess) {
            ctx->tapp = nullptr;
            ctx->tapp_failed = true;     // not retried per view: the front end normalises per iteration instead
            (void)hipGetLastError();
        }
    }
    const int rc = launched(mh_launch_pack_view(ctx->rec + (size_t)view * npix, ctx->mask + (size_t)view * npix, depth,
                                                depth_stride, ori, conf, mask, mask_stride, npix,
                                                ctx->tapp ? ctx->tapp + (size_t)view * npix : nullptr, st),
                            "mh_ctx_set_view");
    // (the view's slice of the plane counts as current only once its pack launch has been accepted)
    if (ctx->tapp && (int)ctx->tap_view.size() == ctx->V) ctx->tap_view[view] = rc == MH_OK ? 1 : 0;
    return rc;
}

extern "C" int mh_ctx_set_view_u8(mh_ctx *ctx, int view, const float *cam_host, const float *depth, int depth_stride,
                                  const unsigned char *ori_u8, const unsigned char *conf_u8,
                                  const unsigned char *mask_u8, const float *lut_host, void *stream) {
    if (!ctx || !ctx->rec) return fail(MH_ERR_STATE, "mh_ctx_set_view_u8: views not allocated");
    if (view < 0 || view >= ctx->V || !cam_host || !depth || !ori_u8 || !conf_u8 || !mask_u8 || !lut_host ||
        depth_stride < 1)
        return fail(MH_ERR_ARG, "mh_ctx_set_view_u8: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const size_t npix = (size_t)ctx->H * ctx->W;
    MH_HIP(hipSetDevice(ctx->device));
    if (!ctx->lut) MH_HIP(hipMalloc(&ctx->lut, 256 * sizeof(float4)));
    if (!ctx->code_tabs) MH_HIP(hipMalloc(&ctx->code_tabs, mh_code_tabs_bytes()));
    if (!ctx->oc && !ctx->oc_failed) {   // (2 B per pixel next to the 20 B of records; without it the tap gathers use the records)
        if (hipMalloc(&ctx->oc, (size_t)ctx->V * npix * sizeof(uint16_t)) != hipSuccess) {
            ctx->oc = nullptr;
            ctx->oc_failed = true;       // not retried per view
            (void)hipGetLastError();     // the fallback is intended: do not leave the error for the next launch check
        }
    }
    // one table per context: views decoded through DIFFERENT tables cannot share the code tables of the tap gather
    if (ctx->lut_set && memcmp(ctx->lut_host, lut_host, sizeof ctx->lut_host) != 0) ctx->lut_mixed = true;
    if (!ctx->lut_set || ctx->lut_mixed) {
        if (ctx->lut_set) MH_HIP(hipDeviceSynchronize());     // (a pack kernel of another stream may still read the old table)
        memcpy(ctx->lut_host, lut_host, sizeof ctx->lut_host);
        ctx->lut_set = true;
        MH_HIP(hipMemcpyAsync(ctx->lut, ctx->lut_host, 256 * sizeof(float4), hipMemcpyHostToDevice, st));
        if (int rc = launched(mh_launch_code_tabs(ctx->lut, ctx->code_tabs, st), "mh_ctx_set_view_u8(tables)")) return rc;
    }
    MH_HIP(hipMemcpyAsync(ctx->cams + (size_t)view * MH_CAM_STRIDE, cam_host, MH_CAM_STRIDE * sizeof(float),
                          hipMemcpyHostToDevice, st));
    if ((int)ctx->code_view.size() == ctx->V) ctx->code_view[view] = ctx->oc ? 1 : 0;
    const int rc = launched(mh_launch_pack_view_u8(ctx->rec + (size_t)view * npix, ctx->mask + (size_t)view * npix, depth,
                                                   depth_stride, ori_u8, conf_u8, mask_u8, ctx->lut, npix,
                                                   ctx->oc ? ctx->oc + (size_t)view * npix : nullptr,
                                                   ctx->tapp ? ctx->tapp + (size_t)view * npix : nullptr, st),
                            "mh_ctx_set_view_u8");
    // (only when an fp32 view allocated the plane; current only once the pack launch has been accepted)
    if (ctx->tapp && (int)ctx->tap_view.size() == ctx->V) ctx->tap_view[view] = rc == MH_OK ? 1 : 0;
    return rc;
}

static size_t render_vt_bytes(int Nv) { return (((size_t)(Nv > 0 ? Nv : 1) * 16) + 255) / 256 * 256; }

static size_t render_q_bytes(int Nf) { return (((size_t)(Nf > 0 ? Nf : 1) * 4) + 255) / 256 * 256; }

// scratch: [camera | queue counter] 512 B | vertices | z/primitive keys | queue of large triangles
extern "C" size_t mh_render_scratch_bytes(int Nv, int Nf, int H, int W) {
    if (Nv < 0 || Nf < 0 || H < 1 || W < 1) return 0;
    return 512 + render_vt_bytes(Nv) + (size_t)H * W * sizeof(unsigned long long) + render_q_bytes(Nf);
}

extern "C" int mh_render_depth(mh_ctx *ctx, const float *cam_host, const float *verts, int Nv, const int32_t *faces,
                               int Nf, int H, int W, float pixel_center, void *scratch, size_t scratch_bytes,
                               float *out, int channels, void *stream) {
    if (!ctx) return fail(MH_ERR_ARG, "mh_render_depth: no context");
    if (!cam_host || !out || !scratch || H < 1 || W < 1 || Nv < 0 || Nf < 0 || channels < 1 ||
        ((Nv > 0 && Nf > 0) && (!verts || !faces)) || !(pixel_center >= 0.0f && pixel_center < 1.0f))
        return fail(MH_ERR_ARG, "mh_render_depth: bad arguments");
    if (scratch_bytes < mh_render_scratch_bytes(Nv, Nf, H, W))
        return fail(MH_ERR_ARG, "mh_render_depth: scratch too small (%zu < %zu)", scratch_bytes,
                    mh_render_scratch_bytes(Nv, Nf, H, W));
    MH_HIP(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    char *base = (char *)scratch;
    float *cam = (float *)base;
    MH_HIP(hipMemcpyAsync(cam, cam_host, MH_CAM_STRIDE * sizeof(float), hipMemcpyHostToDevice, st));
    unsigned int *qcount = (unsigned int *)(base + 256);
    void *vt = base + 512;
    unsigned long long *zbuf = (unsigned long long *)(base + 512 + render_vt_bytes(Nv));
    int32_t *queue = (int32_t *)((char *)zbuf + (size_t)H * W * sizeof(unsigned long long));
    const int off = (int)(pixel_center * 256.0f + 0.5f);
    return launched(mh_launch_render_depth(cam, verts, Nv, faces, Nf, H, W, off, 1 << ctx->raster_subpixel_bits, vt, zbuf,
                                           queue, qcount, out, channels, st),
                    "mh_render_depth");
}

extern "C" size_t mh_render_strands_scratch_bytes(int Nv, int Nf, int Nseg, int H, int W) {
    if (Nv < 0 || Nf < 0 || Nseg < 0 || H < 1 || W < 1) return 0;
    return mh_render_scratch_bytes(Nv, Nf, H, W) + 64 + (size_t)2 * Nseg * 32;
}

extern "C" int mh_render_strands(mh_ctx *ctx, const float *cam_host, const float *verts, int Nv, const int32_t *faces,
                                 int Nf, const float *line_pts, const float *line_tan, int Nseg, int H, int W,
                                 float pixel_center, int line_width, int color_option, int depth_option, float clear,
                                 void *scratch, size_t scratch_bytes, float *out, void *stream) {
    if (!ctx) return fail(MH_ERR_ARG, "mh_render_strands: no context");
    if (!cam_host || !out || !scratch || H < 1 || W < 1 || Nv < 0 || Nf < 0 || Nseg < 0 || line_width < 1 ||
        line_width > 64 || color_option > 3 || depth_option < 0 || depth_option > 2 ||
        ((Nv > 0 && Nf > 0) && (!verts || !faces)) || (Nseg > 0 && color_option >= 0 && (!line_pts || !line_tan)) ||
        !(pixel_center >= 0.0f && pixel_center < 1.0f))
        return fail(MH_ERR_ARG, "mh_render_strands: bad arguments");
    if (scratch_bytes < mh_render_strands_scratch_bytes(Nv, Nf, Nseg, H, W))
        return fail(MH_ERR_ARG, "mh_render_strands: scratch too small (%zu < %zu)", scratch_bytes,
                    mh_render_strands_scratch_bytes(Nv, Nf, Nseg, H, W));
    MH_HIP(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    char *base = (char *)scratch;
    float *cam = (float *)base;
    MH_HIP(hipMemcpyAsync(cam, cam_host, MH_CAM_STRIDE * sizeof(float), hipMemcpyHostToDevice, st));
    unsigned int *qcount = (unsigned int *)(base + 256);
    void *vt = base + 512;
    unsigned long long *zbuf = (unsigned long long *)(base + 512 + render_vt_bytes(Nv));
    int32_t *queue = (int32_t *)((char *)zbuf + (size_t)H * W * sizeof(unsigned long long));
    char *lv = base + ((mh_render_scratch_bytes(Nv, Nf, H, W) + 63) / 64) * 64;
    const int off = (int)(pixel_center * 256.0f + 0.5f);
    return launched(mh_launch_render_strands(cam, verts, Nv, faces, Nf, line_pts, line_tan, Nseg, H, W, off,
                                             1 << ctx->raster_subpixel_bits, line_width, ctx->line_rule, color_option, depth_option, clear, vt, lv, zbuf, queue, qcount,
                                             out, st),
                    "mh_render_strands");
}

extern "C" int mh_ctx_set_depth_offsets(mh_ctx *ctx, const float *offsets_host, int S) {
    if (!ctx || !offsets_host || S < 1 || S > 256) return fail(MH_ERR_ARG, "mh_ctx_set_depth_offsets: bad arguments");
    MH_HIP(hipSetDevice(ctx->device));
    if (ctx->offs) (void)hipFree(ctx->offs);
    ctx->offs = nullptr;
    MH_HIP(hipMalloc(&ctx->offs, S * sizeof(float)));
    MH_HIP(hipMemcpy(ctx->offs, offsets_host, S * sizeof(float), hipMemcpyHostToDevice));
    ctx->S = S;
    return MH_OK;
}

extern "C" int mh_upload_async(mh_ctx *ctx, const void *host, void *device, size_t bytes, void *stream) {
    if (!ctx || (bytes && (!host || !device))) return fail(MH_ERR_ARG, "mh_upload_async: bad arguments");
    if (!bytes) return MH_OK;
    MH_HIP(hipMemcpyAsync(device, host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return MH_OK;
}

// the same copy for a PAGE-LOCKED source: up to 1 MiB as a kernel that reads the host buffer over the link (see the header)
extern "C" int mh_upload_pinned(mh_ctx *ctx, const void *pinned_host, void *device, size_t bytes, void *stream) {
    if (!ctx || (bytes && (!pinned_host || !device))) return fail(MH_ERR_ARG, "mh_upload_pinned: bad arguments");
    if (!bytes) return MH_OK;
    static const bool by_copy_engine = getenv("MH_UPLOAD_KERNEL") && atoi(getenv("MH_UPLOAD_KERNEL")) == 0;
    if (!by_copy_engine && !(bytes & 3) && !((uintptr_t)pinned_host & 3) && !((uintptr_t)device & 3) && bytes <= (1u << 20)) {
        MH_HIP(hipSetDevice(ctx->device));
        return launched(mh_launch_copy_words(pinned_host, device, bytes / 4, (hipStream_t)stream), "mh_upload_pinned");
    }
    MH_HIP(hipMemcpyAsync(device, pinned_host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return MH_OK;
}

extern "C" int mh_ctx_set_option(mh_ctx *ctx, const char *key, int value) {
    if (!ctx || !key) return fail(MH_ERR_ARG, "mh_ctx_set_option: bad arguments");
    if (!strcmp(key, "topk_order")) {
        if ((value & 255) > 1) return fail(MH_ERR_ARG, "mh_ctx_set_option: topk_order must be 0 (torch.topk's order) or 1");
        ctx->topk_order = value;
        return MH_OK;
    }
    if (!strcmp(key, "reproject_rule")) {
        if (value < 0 || value > 2)
            return fail(MH_ERR_ARG, "mh_ctx_set_option: reproject_rule must be 0 (by group size), 1 (mid forms) or 2 (chain forms)");
        ctx->reproject_rule = value;
        return MH_OK;
    }
    if (!strcmp(key, "reproject_fma_min_cols")) {
        if (value < 1) return fail(MH_ERR_ARG, "mh_ctx_set_option: reproject_fma_min_cols must be >= 1");
        ctx->reproject_fma_min_cols = value;
        return MH_OK;
    }
    if (!strcmp(key, "sum_block")) {
        if (value != 0 && value != 32) return fail(MH_ERR_ARG, "mh_ctx_set_option: sum_block must be 32 (ATen's outer sum) or 0");
        ctx->sum_block = value;
        return MH_OK;
    }
    if (!strcmp(key, "tap_plane_max_mb")) {      // takes effect for planes not yet allocated (before the first fp32 view)
        if (value < 0) return fail(MH_ERR_ARG, "mh_ctx_set_option: tap_plane_max_mb must be >= 0");
        ctx->tap_plane_max_mb = value;
        return MH_OK;
    }
    if (!strcmp(key, "gabor_variant")) {
        if (value != 0 && value != 3) return fail(MH_ERR_ARG, "mh_ctx_set_option: gabor_variant must be 0 (valu) or 3 (mfma2)");
        ctx->gabor_variant = value;
        return MH_OK;
    }
    if (!strcmp(key, "line_rule")) {
        ctx->line_rule = value ? 1 : 0;
        return MH_OK;
    }
    if (!strcmp(key, "raster_subpixel_bits")) {
        if (value < 4 || value > 8) return fail(MH_ERR_ARG, "mh_ctx_set_option: raster_subpixel_bits must be 4..8");
        ctx->raster_subpixel_bits = value;
        return MH_OK;
    }
    for (const char *lab : {"search_variant", "search_body", "tap_plane", "tap_codes", "taps_tile", "filter_rows"})
        if (!strcmp(key, lab))
            return fail(MH_ERR_ARG, "mh_ctx_set_option: %s is a lab switch, not a supported option: mh_ctx_set_lab_option "
                                    "(include/mh_pmvo_lab.h)", key);
    return fail(MH_ERR_ARG, "mh_ctx_set_option: unknown key %s", key);
}

// ---- lab switches (include/mh_pmvo_lab.h): A/B forms and cross-check kernels; same results, not part of the supported surface
extern "C" int mh_ctx_set_lab_option(mh_ctx *ctx, const char *key, int value) {
    if (!ctx || !key) return fail(MH_ERR_ARG, "mh_ctx_set_lab_option: bad arguments");
    if (!strcmp(key, "search_variant")) {
        ctx->search_variant = value;
        return MH_OK;
    }
    if (!strcmp(key, "search_body")) {
        if (value < 0 || value > 2) return fail(MH_ERR_ARG, "mh_ctx_set_lab_option: search_body must be 0 (by the maps), 1 (keys) or 2 (select)");
        ctx->search_body = value;
        return MH_OK;
    }
    if (!strcmp(key, "tap_plane")) {
        ctx->use_tap_plane = value ? 1 : 0;
        return MH_OK;
    }
    if (!strcmp(key, "tap_codes")) {
        ctx->use_codes = value ? 1 : 0;
        return MH_OK;
    }
    if (!strcmp(key, "taps_tile")) {
        ctx->taps_tile = value;
        return MH_OK;
    }
    if (!strcmp(key, "filter_rows")) {
        ctx->filter_rows = value ? 1 : 0;
        return MH_OK;
    }
    return fail(MH_ERR_ARG, "mh_ctx_set_lab_option: unknown key %s", key);
}

extern "C" int mh_project_gather(mh_ctx *ctx, const float *points, int N, int patch, float *vis, float *ori,
                                 float *conf, float *mask, float *ori_patch, float *conf_patch, float *pixf,
                                 void *stream) {
    if (!ctx || !ctx->rec) return fail(MH_ERR_STATE, "mh_project_gather: views not set");
    if (N == 0) return MH_OK;
    if (!points || N < 0 || patch < 1 || !(patch & 1)) return fail(MH_ERR_ARG, "mh_project_gather: bad arguments");
    return launched(mh_launch_project_gather(ctx->views(), points, N, patch, vis, ori, conf, mask, ori_patch,
                                             conf_patch, pixf, (hipStream_t)stream),
                    "mh_project_gather");
}

extern "C" int mh_topk_views(mh_ctx *ctx, const float *vis, const float *conf, int N, int32_t *out_idx,
                             float *out_val, void *stream) {
    if (!ctx || !ctx->rec) return fail(MH_ERR_STATE, "mh_topk_views: views not set");
    if (N == 0) return MH_OK;
    if (!vis || !conf || !out_idx || !out_val || N < 0) return fail(MH_ERR_ARG, "mh_topk_views: bad arguments");
    if (ctx->V < MH_TOPK)
        return fail(MH_ERR_ARG, "mh_topk_views: %d views < %d (the reference's torch.topk raises too, PMVO.py:341)",
                    ctx->V, MH_TOPK);
    if (N == 0) return MH_OK;
    return launched(mh_launch_topk(vis, conf, ctx->V, N, out_idx, out_val, ctx->topk_order, (hipStream_t)stream),
                    "mh_topk_views");
}

static size_t search_order_offset(const mh_ctx *ctx, int N, int patch) {
    return ((size_t)ctx->V * (size_t)N * (size_t)(patch * patch + 1) + 16) * sizeof(float4);
}

static size_t search_count_offset(const mh_ctx *ctx, int N, int patch) {
    return search_order_offset(ctx, N, patch) + 2 * (size_t)N * sizeof(int32_t);
}

// behind the list lengths: the points per (rank, base view) of the batch (16 partial copies x 16 ranks x V ints) --
// csrc/mh_device.h: MhRule
static size_t search_groups_offset(const mh_ctx *ctx, int N, int patch) {
    return (search_count_offset(ctx, N, patch) + (size_t)ctx->V * (size_t)N + 255) & ~(size_t)255;
}

extern "C" size_t mh_search_counts_offset(mh_ctx *ctx, int N, int patch) {
    if (!ctx || N < 0 || patch < 1) return 0;
    return search_count_offset(ctx, N, patch);
}

extern "C" size_t mh_search_scratch_bytes(mh_ctx *ctx, int N, int patch) {
    if (!ctx || N < 0 || patch < 1) return 0;
    // + 16 records of slack: the search kernel prefetches tap records in groups past the end of a list;
    // + 2N ints behind them: the launch order of the search (mh_search_order_kernel) and its staging area
    // + V*N bytes: the list lengths once more, compact, for the work estimate
    // + the group sizes of the batch (search_groups_offset)
    return search_groups_offset(ctx, N, patch) + (size_t)16 * 16 * ctx->V * sizeof(int32_t);
}

extern "C" int mh_search_forward(mh_ctx *ctx, const float *points, int N, int patch, float conf_threshold, int nrank,
                                 int rank_step, const float *vis, const float *ori, const float *pixf,
                                 const float *ori_patch, const float *conf_patch, const int32_t *base_idx,
                                 const float *base_val, void *scratch, size_t scratch_bytes, float *line_ori,
                                 float *min_loss, uint8_t *high_conf, float *best_sample, int32_t *best_rank,
                                 int32_t *best_s, void *stream) {
    if (!ctx || !ctx->rec) return fail(MH_ERR_STATE, "mh_search_forward: views not set");
    if (!ctx->offs) return fail(MH_ERR_STATE, "mh_search_forward: depth offsets not set");
    if (N == 0) return MH_OK;
    if (!points || !vis || !ori || !pixf || !ori_patch || !conf_patch || !base_idx || !base_val || !scratch ||
        !line_ori || !min_loss || !high_conf || N < 0 || nrank < 1 || rank_step < 1 ||
        (nrank - 1) * rank_step >= MH_TOPK)
        return fail(MH_ERR_ARG, "mh_search_forward: bad arguments");
    if (scratch_bytes < mh_search_scratch_bytes(ctx, N, patch))
        return fail(MH_ERR_ARG, "mh_search_forward: scratch too small (%zu < %zu)", scratch_bytes,
                    mh_search_scratch_bytes(ctx, N, patch));
    if (ctx->V >= 4096) return fail(MH_ERR_ARG, "mh_search_forward: V >= 4096 needs a fourth cascade level");
    if (N == 0) return MH_OK;
    hipStream_t st = (hipStream_t)stream;
    const int P = patch * patch;
    int rc = launched(mh_launch_prep_taps(ori_patch, conf_patch, vis, pixf, ctx->V * N, P, conf_threshold,
                                          (float4 *)scratch,
                                          (uint8_t *)scratch + search_count_offset(ctx, N, patch), st),
                      "mh_search_forward(prep)");
    if (rc) return rc;
    return launched(mh_launch_search(ctx->views(), ctx->offs, ctx->S, nrank, rank_step, points, N, P + 1,
                                     conf_threshold, ori, base_idx, base_val, (const float4 *)scratch,
                                     (int32_t *)((char *)scratch + search_order_offset(ctx, N, patch)),
                                     (const uint8_t *)scratch + search_count_offset(ctx, N, patch), line_ori,
                                     min_loss, high_conf, best_sample, best_rank, best_s,
                                     ctx->search_launch_variant(ctx->search_variant), ctx->reproject_rule,
                                     ctx->reproject_fma_min_cols, ctx->sum_block,
                                     (int32_t *)((char *)scratch + search_groups_offset(ctx, N, patch)), 0, st),
                    "mh_search_forward");
}

static int forward_prepare(mh_ctx *ctx, const float *points, int N, int patch, float conf_threshold, float *vis, float *ori,
                           float *conf, float *mask, void *scratch, size_t scratch_bytes, int zero_groups, void *stream) {
    if (!ctx || !ctx->rec) return fail(MH_ERR_STATE, "mh_forward_prepare: views not set");
    if (N == 0) return MH_OK;
    if (!points || !vis || !ori || !conf || !scratch || N < 0 || patch < 1 || !(patch & 1))
        return fail(MH_ERR_ARG, "mh_forward_prepare: bad arguments");
    if (scratch_bytes < mh_search_scratch_bytes(ctx, N, patch))
        return fail(MH_ERR_ARG, "mh_forward_prepare: scratch too small");
    return launched(mh_launch_project_taps(ctx->views(), points, N, patch, conf_threshold, vis, ori, conf, mask,
                                           (float4 *)scratch,
                                           (uint8_t *)scratch + search_count_offset(ctx, N, patch), ctx->taps_tile,
                                           ctx->codes_ready() ? ctx->oc : nullptr, ctx->code_tabs,
                                           (int32_t *)((char *)scratch + search_groups_offset(ctx, N, patch)),
                                           zero_groups ? 16 * 16 * ctx->V : 0, (hipStream_t)stream),
                    "mh_forward_prepare");
}

extern "C" int mh_forward_prepare(mh_ctx *ctx, const float *points, int N, int patch, float conf_threshold,
                                  float *vis, float *ori, float *conf, float *mask, void *scratch,
                                  size_t scratch_bytes, void *stream) {
    return forward_prepare(ctx, points, N, patch, conf_threshold, vis, ori, conf, mask, scratch, scratch_bytes, 0, stream);
}

static int search_prepared(mh_ctx *ctx, const float *points, int N, int patch, float conf_threshold, int nrank,
                           int rank_step, const float *ori, const int32_t *base_idx, const float *base_val,
                           void *scratch, float *line_ori, float *min_loss, uint8_t *high_conf,
                           float *best_sample, int32_t *best_rank, int32_t *best_s, int variant, int groups_ready,
                           void *stream) {
    if (!ctx || !ctx->rec) return fail(MH_ERR_STATE, "mh_search_prepared: views not set");
    if (!ctx->offs) return fail(MH_ERR_STATE, "mh_search_prepared: depth offsets not set");
    if (N == 0) return MH_OK;
    if (!points || !ori || !base_idx || !base_val || !scratch || !line_ori || !min_loss || !high_conf || N < 0 ||
        nrank < 1 || rank_step < 1 || (nrank - 1) * rank_step >= MH_TOPK)
        return fail(MH_ERR_ARG, "mh_search_prepared: bad arguments");
    if (ctx->V >= 4096) return fail(MH_ERR_ARG, "mh_search_prepared: V >= 4096 needs a fourth cascade level");
    return launched(mh_launch_search(ctx->views(), ctx->offs, ctx->S, nrank, rank_step, points, N,
                                     patch * patch + 1, conf_threshold, ori, base_idx, base_val,
                                     (const float4 *)scratch,
                                     (int32_t *)((char *)scratch + search_order_offset(ctx, N, patch)),
                                     (const uint8_t *)scratch + search_count_offset(ctx, N, patch), line_ori,
                                     min_loss, high_conf, best_sample, best_rank, best_s,
                                     ctx->search_launch_variant(variant), ctx->reproject_rule, ctx->reproject_fma_min_cols,
                                     ctx->sum_block, (int32_t *)((char *)scratch + search_groups_offset(ctx, N, patch)),
                                     groups_ready, (hipStream_t)stream),
                    "mh_search_prepared");
}

extern "C" int mh_search_prepared(mh_ctx *ctx, const float *points, int N, int patch, float conf_threshold, int nrank,
                                  int rank_step, const float *ori, const int32_t *base_idx, const float *base_val,
                                  void *scratch, float *line_ori, float *min_loss, uint8_t *high_conf,
                                  float *best_sample, int32_t *best_rank, int32_t *best_s, void *stream) {
    return search_prepared(ctx, points, N, patch, conf_threshold, nrank, rank_step, ori, base_idx, base_val, scratch, line_ori,
                           min_loss, high_conf, best_sample, best_rank, best_s, ctx ? ctx->search_variant : 0, 0, stream);
}

extern "C" int mh_forward(mh_ctx *ctx, const float *points, int N, int patch, float conf_threshold, int nrank, int rank_step,
                          float *vis, float *ori, float *conf, float *mask, void *scratch, size_t scratch_bytes,
                          int32_t *base_idx, float *base_val, float *line_ori, float *min_loss, uint8_t *high_conf,
                          float *best_sample, int32_t *best_rank, int32_t *best_s, void *stream) {
    if (!ctx || !ctx->rec) return fail(MH_ERR_STATE, "mh_forward: views not set");
    // with the default kernels the ranking kernel also writes the work classes of the search's launch order (it has the
    // point's ranking in its lanes): one launch less per iteration than the three separate calls
    const bool fuse_cls = ctx->search_variant == 0 && (ctx->topk_order & 255) == 0 && N > 1 && base_idx && base_val && vis &&
                          conf && nrank >= 1 && rank_step >= 1 && ctx->V >= MH_TOPK && ctx->offs;
    // ... and counts the points per (rank, base view) of the batch (csrc/mh_device.h: MhRule) into the array the front end cleared
    const int fuse_groups = fuse_cls && ctx->reproject_rule == 0 && nrank <= 16;
    if (int rc = forward_prepare(ctx, points, N, patch, conf_threshold, vis, ori, conf, mask, scratch, scratch_bytes, fuse_groups,
                                 stream))
        return rc;
    if (!fuse_cls) {
        if (int rc = mh_topk_views(ctx, vis, conf, N, base_idx, base_val, stream)) return rc;
        return mh_search_prepared(ctx, points, N, patch, conf_threshold, nrank, rank_step, ori, base_idx, base_val, scratch,
                                  line_ori, min_loss, high_conf, best_sample, best_rank, best_s, stream);
    }
    int32_t *order = (int32_t *)((char *)scratch + search_order_offset(ctx, N, patch));
    const uint8_t *cnt = (const uint8_t *)scratch + search_count_offset(ctx, N, patch);
    // (the points that hold the trailing columns of the batch's [V, N*S] sums go first in the search's launch order)
    const long long cols = (long long)N * ctx->S;
    const int tail_n0 = ctx->sum_block > 0 ? (int)((cols - cols % ctx->sum_block) / ctx->S) : N;
    if (int rc = launched(mh_launch_topk_work(vis, conf, ctx->V, N, base_idx, base_val, ctx->topk_order, cnt, order,
                                              patch * patch + 1, nrank, rank_step, ctx->S,
                                              fuse_groups ? (int32_t *)((char *)scratch + search_groups_offset(ctx, N, patch))
                                                          : nullptr,
                                              tail_n0, (hipStream_t)stream),
                          "mh_forward (base-view ranking)"))
        return rc;
    return search_prepared(ctx, points, N, patch, conf_threshold, nrank, rank_step, ori, base_idx, base_val, scratch, line_ori,
                           min_loss, high_conf, best_sample, best_rank, best_s, 8 /* ordered, classes written */, fuse_groups,
                           stream);
}

extern "C" int mh_refine_loss(mh_ctx *ctx, const float *points, const float *dir, float step_mul, float step_div,
                              int N, int patch, float conf_threshold, const float *vis, const float *ori_patch,
                              const float *conf_patch, float *loss, uint8_t *high_conf, void *stream) {
    if (!ctx || !ctx->rec) return fail(MH_ERR_STATE, "mh_refine_loss: views not set");
    if (N == 0) return MH_OK;
    if (!points || !dir || !vis || !ori_patch || !conf_patch || !loss || N < 0)
        return fail(MH_ERR_ARG, "mh_refine_loss: bad arguments");
    if (N == 0) return MH_OK;
    return launched(mh_launch_refine_loss(ctx->views(), points, dir, step_mul, step_div, N, patch * patch,
                                          conf_threshold, vis, ori_patch, conf_patch, loss, high_conf, ctx->sum_block,
                                          (hipStream_t)stream),
                    "mh_refine_loss");
}

static int filter_points_impl(mh_ctx *ctx, const float *points, int N, int patch, float conf_threshold,
                              float visible_threshold, uint8_t *surface_index, uint8_t *filter_index,
                              uint8_t *unvisible_index, uint8_t *head_filter, int batch, long long row0, long long total,
                              const int32_t *order, void *stream, const char *what) {
    if (!ctx || !ctx->rec) return fail(MH_ERR_STATE, "%s: views not set", what);
    if (N == 0) return MH_OK;
    if (!points || N < 0 || patch < 1 || !(patch & 1) || batch < 0 || row0 < 0 || (batch > 0 && total < row0 + N))
        return fail(MH_ERR_ARG, "%s: bad arguments", what);
    if (batch == 0) row0 = 0, total = N;
    return launched(mh_launch_filter_points(ctx->views(), points, N, patch, conf_threshold, visible_threshold,
                                            surface_index, filter_index, unvisible_index, head_filter, batch, row0, total,
                                            ctx->sum_block, ctx->filter_rows, order, (hipStream_t)stream),
                    what);
}

extern "C" int mh_filter_points(mh_ctx *ctx, const float *points, int N, int patch, float conf_threshold,
                                float visible_threshold, uint8_t *surface_index, uint8_t *filter_index,
                                uint8_t *unvisible_index, uint8_t *head_filter, int batch, long long row0,
                                long long total, void *stream) {
    return filter_points_impl(ctx, points, N, patch, conf_threshold, visible_threshold, surface_index, filter_index,
                              unvisible_index, head_filter, batch, row0, total, nullptr, stream, "mh_filter_points");
}

extern "C" int mh_filter_points_ordered(mh_ctx *ctx, const float *points, int N, int patch, float conf_threshold,
                                        float visible_threshold, uint8_t *surface_index, uint8_t *filter_index,
                                        uint8_t *unvisible_index, uint8_t *head_filter, int batch, long long row0,
                                        long long total, const int32_t *order, void *stream) {
    return filter_points_impl(ctx, points, N, patch, conf_threshold, visible_threshold, surface_index, filter_index,
                              unvisible_index, head_filter, batch, row0, total, order, stream, "mh_filter_points_ordered");
}

extern "C" int mh_medoid_dense(mh_ctx *ctx, const float *ori, int G, int K, float *out, int32_t *out_index,
                               void *stream) {
    if (!ctx || !ori || !out || G < 0 || K < 1) return fail(MH_ERR_ARG, "mh_medoid_dense: bad arguments");
    if (G == 0) return MH_OK;
    return launched(mh_launch_medoid_dense(ori, nullptr, G, K, out, out_index, (hipStream_t)stream), "mh_medoid_dense");
}

extern "C" int mh_medoid_indexed(mh_ctx *ctx, const float *ori_rows, const int32_t *index, int G, int K, float *out,
                                 int32_t *out_index, void *stream) {
    if (!ctx || !ori_rows || !index || !out || G < 0 || K < 1) return fail(MH_ERR_ARG, "mh_medoid_indexed: bad arguments");
    if (G == 0) return MH_OK;
    return launched(mh_launch_medoid_dense(ori_rows, index, G, K, out, out_index, (hipStream_t)stream),
                    "mh_medoid_indexed");
}

extern "C" int mh_refine_loss_maps(mh_ctx *ctx, const float *points, const float *dir, float step_mul, float step_div,
                                   int N, int patch, float conf_threshold, float *loss, uint8_t *high_conf,
                                   int batch, long long row0, long long total, void *stream) {
    if (!ctx || !ctx->rec) return fail(MH_ERR_STATE, "mh_refine_loss_maps: views not set");
    if (N == 0) return MH_OK;
    if (!points || !dir || !loss || N < 0 || patch < 1 || !(patch & 1) || batch < 0 || row0 < 0 ||
        (batch > 0 && total < row0 + N))
        return fail(MH_ERR_ARG, "mh_refine_loss_maps: bad arguments");
    if (batch == 0) row0 = 0, total = N;
    if (patch > 11)
        return fail(MH_ERR_ARG, "mh_refine_loss_maps: patch side %d is not built in (odd sides 1..11 are; the reference's "
                                "configurations use 5, 7 and 9) -- use mh_project_gather + mh_refine_loss for larger patches",
                    patch);
    if (ctx->V > 512) return fail(MH_ERR_ARG, "mh_refine_loss_maps: %d views exceed the limit of 512", ctx->V);
    return launched(mh_launch_refine_loss_maps(ctx->views(), points, dir, step_mul, step_div, N, patch, conf_threshold,
                                               loss, high_conf, batch, row0, total, ctx->sum_block, (hipStream_t)stream),
                    "mh_refine_loss_maps");
}

extern "C" int mh_refine_combine(mh_ctx *ctx, const float *center, const float *loss_u, const uint8_t *head_filter,
                                 const uint8_t *head_top, float replace_threshold, float *ori, float *loss_out, int N,
                                 void *stream) {
    if (N == 0) return MH_OK;
    if (!ctx || !center || !loss_u || !head_filter || !head_top || !loss_out || N < 0)   // (ori may be NULL: loss only)
        return fail(MH_ERR_ARG, "mh_refine_combine: bad arguments");
    return launched(mh_launch_refine_combine(center, loss_u, head_filter, head_top, replace_threshold, ori, loss_out, N,
                                             (hipStream_t)stream),
                    "mh_refine_combine");
}

extern "C" int mh_medoid_segmented(mh_ctx *ctx, const float *ori, const int32_t *seg_start, int G, int max_group,
                                   float *out, int32_t *out_index, void *stream) {
    if (!ctx || !ori || !seg_start || !out || G < 0 || max_group < 1)
        return fail(MH_ERR_ARG, "mh_medoid_segmented: bad arguments");
    if (G == 0) return MH_OK;
    return launched(mh_launch_medoid_segmented(ori, seg_start, G, max_group, out, out_index, (hipStream_t)stream),
                    "mh_medoid_segmented");
}

extern "C" int mh_replace_dissimilar(mh_ctx *ctx, const float *center, float *ori, float threshold, int N,
                                     void *stream) {
    if (N == 0) return MH_OK;
    if (!ctx || !center || !ori || N < 0) return fail(MH_ERR_ARG, "mh_replace_dissimilar: bad arguments");
    return launched(mh_launch_replace_dissimilar(center, ori, threshold, N, (hipStream_t)stream),
                    "mh_replace_dissimilar");
}

extern "C" int mh_knn_grid(mh_ctx *ctx, const float *grid_origin_h /*host: ox,oy,oz,h*/, const int32_t *grid_dims /*host*/,
                           const float *pts_sorted, const int32_t *order, const int32_t *cell_start,
                           const void *queries, int query_f64, int Q, int k, int first_ring, const int32_t *query_order,
                           const unsigned char *valid, int32_t *out_idx, int32_t *status, void *stream) {
    if (Q == 0) return MH_OK;
    if (!ctx || !grid_origin_h || !grid_dims || !pts_sorted || !order || !cell_start || !queries || !out_idx ||
        !status || Q < 0)
        return fail(MH_ERR_ARG, "mh_knn_grid: bad arguments");
    return launched(mh_launch_knn(grid_origin_h[0], grid_origin_h[1], grid_origin_h[2], grid_origin_h[3], grid_dims[0],
                                  grid_dims[1], grid_dims[2], pts_sorted, order, cell_start, queries, query_f64 ? 1 : 0, Q,
                                  k, first_ring, query_order, valid, out_idx, status, (hipStream_t)stream),
                    "mh_knn_grid");
}

extern "C" int mh_nearest_distance(mh_ctx *ctx, const float *points, int N, const double *ref_points, int M,
                                   double *out_dist, double max_dist, double z_limit, unsigned char *out_mask,
                                   void *stream) {
    if (N == 0) return MH_OK;
    if (!ctx || !points || !ref_points || (!out_dist && !out_mask) || N < 0 || M < 1)
        return fail(MH_ERR_ARG, "mh_nearest_distance: bad arguments");
    return launched(mh_launch_nearest_dist(points, N, ref_points, M, out_dist, max_dist, z_limit, out_mask,
                                           (hipStream_t)stream),
                    "mh_nearest_distance");
}

extern "C" size_t mh_grid_scratch_bytes(int M) { return M < 0 ? 0 : mh_grid_scratch_bytes_impl(M); }
extern "C" size_t mh_sort_scratch_bytes(int n) { return n < 0 ? 0 : mh_sort_scratch_bytes_impl(n); }

extern "C" int mh_grid_build(mh_ctx *ctx, const float *g, const int32_t *d, const float *points, int M, void *scratch,
                             size_t scratch_bytes, float *pts_sorted, int32_t *order, int32_t *cell_start,
                             int32_t *n_occupied, void *stream) {
    if (M == 0) return MH_OK;
    if (!ctx || !g || !d || !points || !scratch || !order || M < 0 || !(g[3] > 0.0f) || d[0] < 1 || d[1] < 1 ||
        d[2] < 1 || (long long)d[0] * d[1] * d[2] > 0x7fffffffll)
        return fail(MH_ERR_ARG, "mh_grid_build: bad arguments");
    if (scratch_bytes < mh_grid_scratch_bytes_impl(M)) return fail(MH_ERR_ARG, "mh_grid_build: scratch too small");
    MH_HIP(hipSetDevice(ctx->device));
    return launched(mh_launch_grid_build(points, M, g[0], g[1], g[2], g[3], d[0], d[1], d[2], scratch, scratch_bytes,
                                         pts_sorted, order, cell_start, n_occupied, (hipStream_t)stream),
                    "mh_grid_build");
}

extern "C" int mh_sort_keys(mh_ctx *ctx, const unsigned long long *keys, int n, int end_bit, void *scratch,
                            size_t scratch_bytes, unsigned long long *keys_out, int32_t *order, void *stream) {
    if (n == 0) return MH_OK;
    if (!ctx || !keys || !scratch || !keys_out || !order || n < 0 || end_bit < 1 || end_bit > 64)
        return fail(MH_ERR_ARG, "mh_sort_keys: bad arguments");
    if (scratch_bytes < mh_sort_scratch_bytes_impl(n)) return fail(MH_ERR_ARG, "mh_sort_keys: scratch too small");
    MH_HIP(hipSetDevice(ctx->device));
    return launched(mh_launch_sort_keys(keys, n, end_bit, scratch, scratch_bytes, keys_out, order, (hipStream_t)stream),
                    "mh_sort_keys");
}

extern "C" size_t mh_voxel_group_scratch_bytes(int n) { return n < 0 ? 0 : mh_voxel_group_scratch_bytes_impl(n); }

extern "C" int mh_voxel_group(mh_ctx *ctx, const void *points, int points_f64, const float *ori, int n,
                              const double *voxel_min, double voxel_size, const int32_t *dims, void *scratch,
                              size_t scratch_bytes, unsigned long long *keys_sorted, int32_t *order, float *ori_sorted,
                              void *stream) {
    if (n == 0) return MH_OK;
    if (!ctx || !points || !voxel_min || !dims || !scratch || !keys_sorted || !order || n < 0 || !(voxel_size > 0.0) ||
        dims[0] < 1 || dims[1] < 1 || dims[2] < 1 || (ori == nullptr) != (ori_sorted == nullptr))
        return fail(MH_ERR_ARG, "mh_voxel_group: bad arguments");
    if (scratch_bytes < mh_voxel_group_scratch_bytes_impl(n)) return fail(MH_ERR_ARG, "mh_voxel_group: scratch too small");
    MH_HIP(hipSetDevice(ctx->device));
    return launched(mh_launch_voxel_group(points, points_f64, ori, n, voxel_min, voxel_size, dims, scratch, scratch_bytes,
                                          keys_sorted, order, ori_sorted, (hipStream_t)stream),
                    "mh_voxel_group");
}

// ---- device-side selection between the stages of refine (csrc/sortgroup.hip): no host round trip ------------------
extern "C" size_t mh_select_scratch_bytes(int n) { return n < 0 ? 0 : mh_select_scratch_bytes_impl(n); }

extern "C" int mh_select_rows(mh_ctx *ctx, const uint8_t *flags, const uint8_t *veto, int invert, int n, const float *a,
                              const float *b, float *a_out, float *b_out, int32_t *index_out, const int32_t *base,
                              int32_t *count, void *scratch, size_t scratch_bytes, void *stream) {
    if (!ctx || !count || !scratch || n < 0 ||
        (n > 0 && (!flags || (a == nullptr) != (a_out == nullptr) || (b == nullptr) != (b_out == nullptr))))
        return fail(MH_ERR_ARG, "mh_select_rows: bad arguments");
    if (scratch_bytes < mh_select_scratch_bytes_impl(n)) return fail(MH_ERR_ARG, "mh_select_rows: scratch too small");
    MH_HIP(hipSetDevice(ctx->device));
    return launched(mh_launch_select_rows(flags, veto, invert, n, a, b, a_out, b_out, index_out, base, count, scratch,
                                          (hipStream_t)stream),
                    "mh_select_rows");
}

extern "C" int mh_segment_heads(mh_ctx *ctx, const unsigned long long *keys_sorted, int n, int32_t *seg_start,
                                unsigned long long *head_keys, int32_t *meta, void *scratch, size_t scratch_bytes,
                                void *stream) {
    if (!ctx || !seg_start || !meta || !scratch || n < 0 || (n > 0 && !keys_sorted))
        return fail(MH_ERR_ARG, "mh_segment_heads: bad arguments");
    if (scratch_bytes < mh_select_scratch_bytes_impl(n)) return fail(MH_ERR_ARG, "mh_segment_heads: scratch too small");
    MH_HIP(hipSetDevice(ctx->device));
    return launched(mh_launch_segment_heads(keys_sorted, n, seg_start, head_keys, meta, scratch, (hipStream_t)stream),
                    "mh_segment_heads");
}

extern "C" int mh_flag_less(mh_ctx *ctx, const float *x, float threshold, int n, uint8_t *out, void *stream) {
    if (n == 0) return MH_OK;
    if (!ctx || !x || !out || n < 0) return fail(MH_ERR_ARG, "mh_flag_less: bad arguments");
    MH_HIP(hipSetDevice(ctx->device));
    return launched(mh_launch_flag_less(x, threshold, n, out, (hipStream_t)stream), "mh_flag_less");
}

extern "C" int mh_points_bbox(mh_ctx *ctx, const float *points, int M, float *out6, void *stream) {
    if (!ctx || !out6 || M < 0 || (M > 0 && !points)) return fail(MH_ERR_ARG, "mh_points_bbox: bad arguments");
    MH_HIP(hipSetDevice(ctx->device));
    return launched(mh_launch_points_bbox(points, M, out6, (hipStream_t)stream), "mh_points_bbox");
}

extern "C" int mh_buffers_differ(mh_ctx *ctx, const void *a, const void *b, size_t bytes, int32_t *flag, void *stream) {
    if (!ctx || !flag || (bytes && (!a || !b)) || (bytes & 3)) return fail(MH_ERR_ARG, "mh_buffers_differ: bad arguments");
    MH_HIP(hipSetDevice(ctx->device));
    return launched(mh_launch_words_differ(a, b, bytes / 4, flag, (hipStream_t)stream), "mh_buffers_differ");
}

// ---- the intermediate methods of the reference's class, as stand-alone calls (csrc/pmvo_pieces.hip) ---------------
extern "C" int mh_project_points(mh_ctx *ctx, int view, const float *points, int N, int32_t *row_col, float *z_half,
                                 uint8_t *out_of_image, float *pixel_unrounded, void *stream) {
    if (!ctx || !ctx->rec) return fail(MH_ERR_STATE, "mh_project_points: views not set");
    if (N == 0) return MH_OK;
    if (!points || N < 0 || view < 0 || view >= ctx->V) return fail(MH_ERR_ARG, "mh_project_points: bad arguments");
    return launched(mh_launch_project_points(ctx->cams + (size_t)view * MH_CAM_STRIDE, points, N, ctx->H, ctx->W, row_col,
                                             z_half, out_of_image, pixel_unrounded, ctx->reproject_rule == 0 ? 1 : 0,
                                             (hipStream_t)stream),
                    "mh_project_points");
}

extern "C" int mh_gather_pixels(mh_ctx *ctx, int view, const long long *row_col, int N, int size, float *records,
                                float *mask, void *stream) {
    if (!ctx || !ctx->rec) return fail(MH_ERR_STATE, "mh_gather_pixels: views not set");
    if (N == 0) return MH_OK;
    if (!row_col || N < 0 || view < 0 || view >= ctx->V || size < 1 || !(size & 1))
        return fail(MH_ERR_ARG, "mh_gather_pixels: bad arguments");
    return launched(mh_launch_gather(ctx->views(), view, row_col, N, size, (float4 *)records, mask, (hipStream_t)stream),
                    "mh_gather_pixels");
}

extern "C" int mh_compute_visible(mh_ctx *ctx, const float *depth, const float *z, size_t n, float *out, void *stream) {
    if (n == 0) return MH_OK;
    if (!ctx || !depth || !z || !out) return fail(MH_ERR_ARG, "mh_compute_visible: bad arguments");
    return launched(mh_launch_compute_visible(depth, z, n, out, (hipStream_t)stream), "mh_compute_visible");
}

extern "C" int mh_sample_next(mh_ctx *ctx, const float *points, const int32_t *base_view, const float *ori,
                              const float *offsets, int N, int S, float *samples, void *stream) {
    if (!ctx || !ctx->rec) return fail(MH_ERR_STATE, "mh_sample_next: views not set");
    if (N == 0) return MH_OK;
    if (!points || !base_view || !ori || !offsets || !samples || N < 0 || S < 1)
        return fail(MH_ERR_ARG, "mh_sample_next: bad arguments");
    // (the V group sizes of this batch -- stream-ordered work space, so that calls on several streams do not share it)
    int32_t *gcnt = nullptr;
    if (ctx->reproject_rule == 0)
        MH_HIP(hipMallocAsync((void **)&gcnt, sizeof(int32_t) * (size_t)16 * 16 * ctx->V, (hipStream_t)stream));
    const int rc = launched(mh_launch_sample_next(ctx->views(), points, base_view, ori, offsets, N, S, samples,
                                                  ctx->reproject_rule, ctx->reproject_fma_min_cols, gcnt,
                                                  (hipStream_t)stream),
                            "mh_sample_next");
    if (gcnt) (void)hipFreeAsync(gcnt, (hipStream_t)stream);
    return rc;
}

extern "C" int mh_reproject_ori(mh_ctx *ctx, const float *points, const float *samples, int N, int S, float *D,
                                void *stream) {
    if (!ctx || !ctx->rec) return fail(MH_ERR_STATE, "mh_reproject_ori: views not set");
    if (N == 0) return MH_OK;
    if (!points || !samples || !D || N < 0 || S < 1) return fail(MH_ERR_ARG, "mh_reproject_ori: bad arguments");
    return launched(mh_launch_reproject(ctx->views(), points, samples, N, S, D, (hipStream_t)stream), "mh_reproject_ori");
}

extern "C" int mh_prj_loss(mh_ctx *ctx, const float *D, const float *ori_patch, const float *conf_patch,
                           const float *vis, int V, int N, int S, int P, float conf_threshold, float *loss,
                           long long *index, uint8_t *high_conf, float *all_loss, void *stream) {
    if (N == 0) return MH_OK;
    if (!ctx || !D || !ori_patch || !conf_patch || !vis || !loss || V < 1 || V >= 4096 || N < 0 || S < 1 || P < 1)
        return fail(MH_ERR_ARG, "mh_prj_loss: bad arguments");
    return launched(mh_launch_prj_loss(D, ori_patch, conf_patch, vis, V, N, S, P, conf_threshold, loss, index, high_conf,
                                       all_loss, ctx->sum_block, (hipStream_t)stream),
                    "mh_prj_loss");
}

// ---- strand tracing on the fitted volume (HairGrow.py:59-299) ------------------------------------------------
extern "C" int mh_volume_pack(mh_ctx *ctx, const float *occ, const float *ori, int W, int H, int Z, void *vox,
                              void *stream) {
    if (!ctx || !occ || !ori || !vox || W < 1 || H < 1 || Z < 1) return fail(MH_ERR_ARG, "mh_volume_pack: bad arguments");
    return launched(mh_launch_pack_volume(occ, ori, (size_t)W * H * Z, (float4 *)vox, (hipStream_t)stream),
                    "mh_volume_pack");
}

extern "C" int mh_trace_seeds(mh_ctx *ctx, const void *vox, int W, int H, int Z, const float *seeds, int n,
                              float thr_dot, float *out, int32_t *first, int32_t *len, void *stream) {
    if (n == 0) return MH_OK;
    if (!ctx || !vox || !seeds || !out || !first || !len || n < 0) return fail(MH_ERR_ARG, "mh_trace_seeds: bad arguments");
    return launched(mh_launch_trace_seeds((const float4 *)vox, W, H, Z, seeds, n, thr_dot, out, first, len,
                                          (hipStream_t)stream),
                    "mh_trace_seeds");
}

extern "C" int mh_trace_scalp(mh_ctx *ctx, const void *vox, int W, int H, int Z, const float *seeds,
                              const float *normals, int n, float thr_dot, float *out, int32_t *len, void *stream) {
    if (n == 0) return MH_OK;
    if (!ctx || !vox || !seeds || !normals || !out || !len || n < 0)
        return fail(MH_ERR_ARG, "mh_trace_scalp: bad arguments");
    return launched(mh_launch_trace_scalp((const float4 *)vox, W, H, Z, seeds, normals, n, thr_dot, out, len,
                                          (hipStream_t)stream),
                    "mh_trace_scalp");
}

extern "C" int mh_strands_compact(mh_ctx *ctx, const float *rows, const int32_t *first, const int32_t *len,
                                  const long long *offsets, int n, int stride, float *packed, void *stream) {
    if (n == 0) return MH_OK;
    if (!ctx || !rows || !len || !offsets || !packed || n < 0 || stride < 1)
        return fail(MH_ERR_ARG, "mh_strands_compact: bad arguments");
    return launched(mh_launch_strands_compact(rows, first, len, (const int64_t *)offsets, n, stride, packed,
                                              (hipStream_t)stream),
                    "mh_strands_compact");
}

// The sequential `flag` gate (HairGrow.py:72,144,247,260,292), replayed on the HOST over finished traces: all
// pointers are host pointers.  mode 0: voxel seeds (skip if flag[seed voxel] >= 3 or fewer than 5 points; an accepted
// strand adds 1 to every distinct voxel it touches); mode 1: scalp roots (kept when len > 0; their voxels are set to 1).
extern "C" int mh_strands_accept(int W, int H, int Z, float *flag, const float *pts, const int32_t *first,
                                 const int32_t *len, int stride, const float *seeds, int n, int mode,
                                 uint8_t *accepted) {
    if (!flag || !pts || !first || !len || !seeds || !accepted || n < 0 || stride < 0)
        return fail(MH_ERR_ARG, "mh_strands_accept: bad arguments");
    const size_t nvox = (size_t)W * H * Z;
    int32_t *stamp = new (std::nothrow) int32_t[nvox];
    if (!stamp) return fail(MH_ERR_NOMEM, "mh_strands_accept: out of host memory");
    memset(stamp, 0xff, sizeof(int32_t) * nvox);
    auto clampi = [](int x, int hi) { return x < 0 ? 0 : (x > hi ? hi : x); };
    for (int i = 0; i < n; ++i) {
        accepted[i] = 0;
        if (mode == 0) {
            const float *s = seeds + 3 * (size_t)i;
            const size_t q = ((size_t)clampi((int)s[2], Z - 1) * H + clampi((int)s[1], H - 1)) * W + clampi((int)s[0], W - 1);
            if (flag[q] >= 3.0f || len[i] < 5) continue;
        } else if (len[i] <= 0) {
            continue;
        }
        accepted[i] = 1;
        const float *p = pts + ((size_t)i * stride + first[i]) * 3;
        for (int k = 0; k < len[i]; ++k) {
            const size_t q = ((size_t)clampi((int)p[3 * k + 2], Z - 1) * H + clampi((int)p[3 * k + 1], H - 1)) * W +
                             clampi((int)p[3 * k], W - 1);
            if (mode == 1) {
                flag[q] = 1.0f;
            } else if (stamp[q] != i) {
                stamp[q] = i;
                flag[q] += 1.0f;
            }
        }
    }
    delete[] stamp;
    return MH_OK;
}

static int gabor_alloc(mh_ctx *ctx) {
    if (ctx->gabor) return MH_OK;
    MH_HIP(hipSetDevice(ctx->device));
    MH_HIP(hipMalloc(&ctx->gabor, 290 * 192 * sizeof(float)));   // 289 taps + one zero pad tap (MFMA K = 290)
    MH_HIP(hipMalloc(&ctx->gabor_max, mh_gabor_state_bytes()));
    MH_HIP(hipMalloc(&ctx->gabor_q, mh_gabor_bankq_bytes()));
    return MH_OK;
}

extern "C" int mh_gabor_set_bank(mh_ctx *ctx, const float *bank_host) {
    if (!ctx || !bank_host) return fail(MH_ERR_ARG, "mh_gabor_set_bank: bad arguments");
    int rc = gabor_alloc(ctx);
    if (rc) return rc;
    // kernel-major [180][289] -> tap-major [289][192], zero padded
    float *tmp = new (std::nothrow) float[290 * 192]();
    if (!tmp) return fail(MH_ERR_NOMEM, "mh_gabor_set_bank: out of host memory");
    for (int k = 0; k < 180; ++k)
        for (int t = 0; t < 289; ++t) tmp[t * 192 + k] = bank_host[k * 289 + t];
    hipError_t e = hipMemcpy(ctx->gabor, tmp, 290 * 192 * sizeof(float), hipMemcpyHostToDevice);
    delete[] tmp;
    if (e != hipSuccess) return fail(MH_ERR_HIP, "mh_gabor_set_bank: %s", hipGetErrorString(e));
    rc = launched(mh_launch_gabor_relayout(ctx->gabor, ctx->gabor_q, nullptr), "mh_gabor_set_bank(relayout)");
    if (rc) return rc;
    MH_HIP(hipStreamSynchronize(nullptr));      // (installation is rare; later launches may come on any stream)
    return MH_OK;
}

extern "C" int mh_gabor_bank(mh_ctx *ctx, const float *image, int H, int W, int32_t *orient_index, float *conf,
                             float *variance, void *stream) {
    if (!ctx || !image || !orient_index || !conf || !variance || H < 1 || W < 1)
        return fail(MH_ERR_ARG, "mh_gabor_bank: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (!ctx->gabor) {
        int rc = gabor_alloc(ctx);
        if (rc) return rc;
        rc = launched(mh_launch_gabor_build(ctx->gabor, st), "mh_gabor_bank(build)");
        if (rc) return rc;
        rc = launched(mh_launch_gabor_relayout(ctx->gabor, ctx->gabor_q, st), "mh_gabor_bank(relayout)");
        if (rc) return rc;
    }
    return launched(mh_launch_gabor_bank(ctx->gabor, ctx->gabor_q, image, H, W, orient_index, conf, variance, ctx->gabor_max,
                                         ctx->gabor_variant, nullptr, nullptr, st),
                    "mh_gabor_bank");
}

// The DoG weights (two symmetric halves, float64, computed by the caller the way scipy.ndimage does) live in a small device
// struct; it is re-uploaded only when they change (in practice once: the reference always calls (0.4, 10)).
static int dog_weights(mh_ctx *ctx, const double *w_lo, int r_lo, const double *w_hi, int r_hi, hipStream_t st) {
    if (!w_lo || !w_hi || r_lo < 0 || r_hi < 0)
        return fail(MH_ERR_ARG, "mh_dog: weights missing");
    if (r_lo > MH_DG_MAXR || r_hi > MH_DG_MAXR)
        return fail(MH_ERR_ARG, "mh_dog: kernel radius %d exceeds the built-in limit of %d (sigma <= %.1f at truncate 4); the "
                                "reference uses sigma 0.4 and 10 (radius 2 and 40)", r_lo > r_hi ? r_lo : r_hi, MH_DG_MAXR,
                    (MH_DG_MAXR + 0.49) / 4.0);
    MhDogWeightsHost h;
    memset(&h, 0, sizeof h);
    memcpy(h.w[0], w_lo, sizeof(double) * (r_lo + 1));
    memcpy(h.w[1], w_hi, sizeof(double) * (r_hi + 1));
    h.r[0] = r_lo;
    h.r[1] = r_hi;
    MH_HIP(hipSetDevice(ctx->device));
    if (!ctx->dog_w) {
        MH_HIP(hipMalloc(&ctx->dog_w, sizeof(MhDogWeightsHost)));
        ctx->dog_w_host = new MhDogWeightsHost;
        memset(ctx->dog_w_host, 0xff, sizeof(MhDogWeightsHost));
    }
    if (memcmp(ctx->dog_w_host, &h, sizeof h) != 0) {
        // (other streams may still be reading the old weights: wait for the device before replacing them)
        MH_HIP(hipDeviceSynchronize());
        *ctx->dog_w_host = h;
        MH_HIP(hipMemcpy(ctx->dog_w, ctx->dog_w_host, sizeof h, hipMemcpyHostToDevice));
    }
    (void)st;
    return MH_OK;
}

extern "C" size_t mh_dog_scratch_bytes(int H, int W) { return (size_t)2 * H * W * sizeof(double); }

extern "C" int mh_dog(mh_ctx *ctx, const void *image, int in_kind, int H, int W, const double *w_lo, int r_lo,
                      const double *w_hi, int r_hi, void *scratch, double *out64, float *out32, void *stream) {
    if (!ctx || !image || !scratch || (!out64 && !out32) || H < 1 || W < 1 || (in_kind != 0 && in_kind != 1))
        return fail(MH_ERR_ARG, "mh_dog: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (int rc = dog_weights(ctx, w_lo, r_lo, w_hi, r_hi, st)) return rc;
    return launched(mh_launch_dog(image, in_kind, H, W, ctx->dog_w, (double *)scratch, out64, out32, st), "mh_dog");
}

// One view of the Gabor stage, device to device: gray uint8 image -> DoG (float64, cast to float32) -> bank -> confidence
// -> the two 8-bit file codes.  scratch: mh_gabor_view_scratch_bytes(H, W) = two float64 planes + the float32 DoG image +
// the image-maximum slot (in the caller's scratch, so views on different streams do not share it).
extern "C" size_t mh_gabor_view_scratch_bytes(int H, int W) { return (size_t)H * W * (16 + 4) + mh_gabor_state_bytes(); }

extern "C" int mh_gabor_view(mh_ctx *ctx, const uint8_t *gray, int H, int W, const double *w_lo, int r_lo, const double *w_hi,
                             int r_hi, void *scratch, int32_t *orient_index, float *conf, float *variance, uint8_t *k8,
                             uint8_t *c8, void *stream) {
    if (!ctx || !gray || !scratch || !orient_index || !variance || H < 1 || W < 1)
        return fail(MH_ERR_ARG, "mh_gabor_view: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (int rc = dog_weights(ctx, w_lo, r_lo, w_hi, r_hi, st)) return rc;
    if (!ctx->gabor) {
        int rc = gabor_alloc(ctx);
        if (rc) return rc;
        rc = launched(mh_launch_gabor_build(ctx->gabor, st), "mh_gabor_view(build)");
        if (rc) return rc;
        rc = launched(mh_launch_gabor_relayout(ctx->gabor, ctx->gabor_q, st), "mh_gabor_view(relayout)");
        if (rc) return rc;
    }
    char *base = (char *)scratch;
    double *planes = (double *)base;
    float *dog32 = (float *)(base + (size_t)H * W * 16);
    unsigned int *maxbits = (unsigned int *)(base + (size_t)H * W * 20);
    if (int rc = launched(mh_launch_dog(gray, 0, H, W, ctx->dog_w, planes, nullptr, dog32, st), "mh_gabor_view(dog)")) return rc;
    return launched(mh_launch_gabor_bank(ctx->gabor, ctx->gabor_q, dog32, H, W, orient_index, conf, variance, maxbits,
                                         ctx->gabor_variant, k8, c8, st),
                    "mh_gabor_view");
}

// ---------------------------------------------------------------------------------------------
// RCCL through the C ABI: the ONE exchange of the data path -- every rank has fitted a disjoint
// slab of the orientation/occupancy volume, rank `root` ends up with all of it.  librccl is bound at run time
// (dlopen by its soname: in a torch process that is the copy torch already loaded, so both share one RCCL), so the
// single-GPU path does not depend on it.
// ---------------------------------------------------------------------------------------------
#include <dlfcn.h>

namespace {
struct MhNcclId {
    char internal[128];   // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES)
};
typedef void *MhNcclComm;
struct MhRccl {
    void *h = nullptr;
    int (*GetUniqueId)(MhNcclId *) = nullptr;
    int (*CommInitRank)(MhNcclComm *, int, MhNcclId, int) = nullptr;
    int (*CommDestroy)(MhNcclComm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, MhNcclComm, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, MhNcclComm, hipStream_t) = nullptr;
    int (*Reduce)(const void *, void *, size_t, int, int, int, MhNcclComm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
MhRccl g_rccl;
const int MH_NCCL_FLOAT32 = 7, MH_NCCL_SUM = 0;   // rccl.h: ncclFloat32, ncclSum

int rccl_load() {
    if (g_rccl.h) return MH_OK;
    // MH_RCCL_LIB=<path>: bind this library instead (a site's own RCCL build; tests/fake_rccl.cpp -- a stand-in compiled
    // against rccl.h that moves the data between processes sharing ONE GPU, so that the nranks > 1 branches below run on a
    // one-GPU box).  It must be loadable: a wrong path is an error, never a silent fall-through to the system library.
    const char *over = getenv("MH_RCCL_LIB");
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    if (over && *over) {
        h = dlopen(over, RTLD_NOW | RTLD_LOCAL);
        if (!h) return fail(MH_ERR_STATE, "MH_RCCL_LIB=%s cannot be loaded: %s", over, dlerror());
    } else {
        for (const char *n : names)
            if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    }
    if (!h) return fail(MH_ERR_STATE, "librccl.so.1 not found: %s", dlerror());
    MhRccl r;
    r.h = h;
#define MH_SYM(field, name)                                                               \
    *(void **)(&r.field) = dlsym(h, name);                                                \
    if (!r.field) return fail(MH_ERR_STATE, "librccl: symbol %s missing", name)
    MH_SYM(GetUniqueId, "ncclGetUniqueId");
    MH_SYM(CommInitRank, "ncclCommInitRank");
    MH_SYM(CommDestroy, "ncclCommDestroy");
    MH_SYM(GroupStart, "ncclGroupStart");
    MH_SYM(GroupEnd, "ncclGroupEnd");
    MH_SYM(Send, "ncclSend");
    MH_SYM(Recv, "ncclRecv");
    MH_SYM(Reduce, "ncclReduce");
    MH_SYM(GetErrorString, "ncclGetErrorString");
#undef MH_SYM
    g_rccl = r;
    return MH_OK;
}
}   // namespace

#define MH_NCCL(call)                                                                                   \
    do {                                                                                                \
        int e_ = (call);                                                                                \
        if (e_ != 0) return fail(MH_ERR_HIP, "%s: %s", #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(e_) : "?"); \
    } while (0)

// The grouped point-to-point exchange both entry points below issue: every peer sends its slab to the root, the root
// receives each one at its place in the dense volume.  The group is closed on every path (a failed call inside an open
// group would otherwise leave the thread's group depth raised for every later call).
static int slab_exchange(int rank, int nranks, int root, const float *own_slab, float *volume, size_t plane,
                         const int32_t *slab_host, MhNcclComm comm, hipStream_t st) {
    MH_NCCL(g_rccl.GroupStart());
    int e = 0;
    const char *what = "";
    if (rank == root) {
        for (int r = 0; r < nranks && e == 0; ++r) {
            const size_t cnt = (size_t)(slab_host[r + 1] - slab_host[r]) * plane;
            if (r == root || cnt == 0) continue;
            e = g_rccl.Recv(volume + (size_t)slab_host[r] * plane, cnt, MH_NCCL_FLOAT32, r, comm, st);
            what = "ncclRecv";
        }
    } else if (own_slab) {
        const size_t cnt = (size_t)(slab_host[rank + 1] - slab_host[rank]) * plane;
        e = g_rccl.Send(own_slab, cnt, MH_NCCL_FLOAT32, root, comm, st);
        what = "ncclSend";
    }
    const int e2 = g_rccl.GroupEnd();
    if (e == 0 && e2 != 0) {
        e = e2;
        what = "ncclGroupEnd";
    }
    if (e != 0) return fail(MH_ERR_HIP, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?");
    return MH_OK;
}

extern "C" int mh_comm_unique_id(void *id_out_host) {
    if (!id_out_host) return fail(MH_ERR_ARG, "mh_comm_unique_id: NULL");
    if (int rc = rccl_load()) return rc;
    MH_NCCL(g_rccl.GetUniqueId((MhNcclId *)id_out_host));
    return MH_OK;
}

extern "C" int mh_comm_init(mh_ctx *ctx, const void *id_host, int nranks, int rank, void **comm_out) {
    if (!ctx || !id_host || !comm_out || nranks < 1 || rank < 0 || rank >= nranks)
        return fail(MH_ERR_ARG, "mh_comm_init: bad arguments");
    if (int rc = rccl_load()) return rc;
    MH_HIP(hipSetDevice(ctx->device));
    MhNcclId id;
    memcpy(&id, id_host, sizeof(id));
    MhNcclComm c = nullptr;
    MH_NCCL(g_rccl.CommInitRank(&c, nranks, id, rank));
    *comm_out = c;
    return MH_OK;
}

extern "C" int mh_comm_destroy(void *comm) {
    if (!comm) return MH_OK;
    if (int rc = rccl_load()) return rc;
    MH_NCCL(g_rccl.CommDestroy((MhNcclComm)comm));
    return MH_OK;
}

extern "C" int mh_volume_reduce(mh_ctx *ctx, void *comm, int rank, int nranks, int root, float *volume, int X, int Y,
                                int Z, int C, const int32_t *slab_host, int mode, void *stream) {
    if (!ctx || !comm || !volume || !slab_host || nranks < 1 || rank < 0 || rank >= nranks || root < 0 ||
        root >= nranks || X < 1 || Y < 1 || Z < 1 || C < 1 || (mode != 0 && mode != 1))
        return fail(MH_ERR_ARG, "mh_volume_reduce: bad arguments");
    if (slab_host[0] != 0 || slab_host[nranks] != X) return fail(MH_ERR_ARG, "mh_volume_reduce: slabs must cover [0, X)");
    for (int r = 0; r < nranks; ++r)
        if (slab_host[r] > slab_host[r + 1]) return fail(MH_ERR_ARG, "mh_volume_reduce: slabs must be ascending");
    if (int rc = rccl_load()) return rc;
    hipStream_t st = (hipStream_t)stream;
    const size_t plane = (size_t)Y * Z * C;   // floats per x index: a slab is one contiguous block
    if (mode == 1) {   // dense sum into root (x + 0 is exact, so this is the same volume; C*X*Y*Z floats over every link)
        MH_NCCL(g_rccl.Reduce(volume, volume, plane * X, MH_NCCL_FLOAT32, MH_NCCL_SUM, root, (MhNcclComm)comm, st));
        return MH_OK;
    }
    // mode 0: slab ownership is disjoint, so nothing has to be added: every peer sends its own slab straight to the
    // root over its own xGMI link and the root receives it in place -- (nranks-1)/nranks of the volume in total, each
    // link carrying one slab
    const size_t own = (size_t)(slab_host[rank + 1] - slab_host[rank]) * plane;
    return slab_exchange(rank, nranks, root, own ? volume + (size_t)slab_host[rank] * plane : nullptr, volume, plane,
                         slab_host, (MhNcclComm)comm, st);
}

// mh_volume_gather: the slab gather with slab-sized buffers on the peers.  `slab` holds this rank's own x-slab
// ([slab_host[rank+1]-slab_host[rank], Y, Z, C], contiguous); only the root has the dense volume.  The root's own slab is
// copied into place on the stream unless it already lives there (slab == volume + offset).  Same wire traffic as mode 0
// of mh_volume_reduce; a peer allocates 1/nranks of the volume instead of all of it (2.15 GB at 512^3).
extern "C" int mh_volume_gather(mh_ctx *ctx, void *comm, int rank, int nranks, int root, const float *slab, float *volume,
                                int X, int Y, int Z, int C, const int32_t *slab_host, void *stream) {
    if (!ctx || !comm || !slab_host || nranks < 1 || rank < 0 || rank >= nranks || root < 0 || root >= nranks || X < 1 ||
        Y < 1 || Z < 1 || C < 1)
        return fail(MH_ERR_ARG, "mh_volume_gather: bad arguments");
    if (slab_host[0] != 0 || slab_host[nranks] != X) return fail(MH_ERR_ARG, "mh_volume_gather: slabs must cover [0, X)");
    for (int r = 0; r < nranks; ++r)
        if (slab_host[r] > slab_host[r + 1]) return fail(MH_ERR_ARG, "mh_volume_gather: slabs must be ascending");
    const size_t plane = (size_t)Y * Z * C;
    const size_t mine = (size_t)(slab_host[rank + 1] - slab_host[rank]) * plane;
    if (mine && !slab) return fail(MH_ERR_ARG, "mh_volume_gather: rank %d owns %zu floats but slab is NULL", rank, mine);
    if (rank == root && !volume) return fail(MH_ERR_ARG, "mh_volume_gather: the root needs the dense volume");
    if (int rc = rccl_load()) return rc;
    hipStream_t st = (hipStream_t)stream;
    MH_HIP(hipSetDevice(ctx->device));
    if (rank == root) {
        float *dst = volume + (size_t)slab_host[root] * plane;
        if (mine && dst != slab) MH_HIP(hipMemcpyAsync(dst, slab, mine * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    if (nranks == 1) return MH_OK;
    return slab_exchange(rank, nranks, root, mine ? slab : nullptr, volume, plane, slab_host, (MhNcclComm)comm, st);
}

// ---------------------------------------------------------------------------------------------
// Host-side IO of the volume files (PMVO.py:753-764 scipy.io.savemat of the dense float64 arrays): the MAT-v5 payload is
// a zero-filled array of which only the occupied voxels are non-zero, so the file is created sparse and the occupied
// elements are scattered into a shared mapping.  What costs time is the first touch of each 4 KB page (allocation +
// zero fill in the page cache); the scatter is therefore split over threads by DESTINATION range, which keeps every
// page with one thread and preserves "later rows win" for duplicate elements (each thread walks the list in order).
// ---------------------------------------------------------------------------------------------
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <thread>
#include <vector>

extern "C" int mh_mat_write_sparse(const char *path, const void *prefix, size_t prefix_bytes, size_t payload_bytes,
                                   const long long *elem_index, const double *values, size_t n, int threads) {
    if (!path || (!prefix && prefix_bytes) || (payload_bytes & 7) || (n && (!elem_index || !values)))
        return fail(MH_ERR_ARG, "mh_mat_write_sparse: bad arguments");
    const size_t nelem = payload_bytes / 8;
    for (size_t i = 0; i < n; ++i)
        if (elem_index[i] < 0 || (size_t)elem_index[i] >= nelem)
            return fail(MH_ERR_ARG, "mh_mat_write_sparse: element %zu out of range", i);
    const int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return fail(MH_ERR_STATE, "mh_mat_write_sparse: cannot create %s", path);
    const size_t total = prefix_bytes + payload_bytes;
    bool ok = (size_t)write(fd, prefix, prefix_bytes) == prefix_bytes && ftruncate(fd, (off_t)total) == 0;
    if (ok && n) {
        void *m = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (m == MAP_FAILED) {
            ok = false;
        } else {
            char *payload = (char *)m + prefix_bytes;      // (8-byte elements at an 8-byte aligned prefix: MAT v5 pads)
            int T = threads < 1 ? 1 : (threads > 64 ? 64 : threads);
            if (n < 4096) T = 1;
            const size_t span = (nelem + T - 1) / T;
            auto work = [&](int t) {
                const size_t lo = (size_t)t * span, hi = lo + span;
                for (size_t i = 0; i < n; ++i) {
                    const size_t e = (size_t)elem_index[i];
                    if (e >= lo && e < hi) memcpy(payload + e * 8, &values[i], 8);
                }
            };
            if (T == 1) {
                work(0);
            } else {
                std::vector<std::thread> pool;
                for (int t = 0; t < T; ++t) pool.emplace_back(work, t);
                for (auto &th : pool) th.join();
            }
            munmap(m, total);
        }
    }
    close(fd);
    if (!ok) return fail(MH_ERR_STATE, "mh_mat_write_sparse: writing %s failed", path);
    return MH_OK;
}

// The same file in steps, so that the page faults of the zero-filled mapping (4 KB of page cache to clear per touched page:
// 16-19 ms for the two volume files of a pass) can be taken by a background thread while the GPU still works:
//   open  -> creates the file, writes the prefix, maps it;
//   touch -> makes the pages of the given elements resident without changing their contents (reads the element and writes it
//            back: call it BEFORE store, not beside it), each page once -- called early with a superset of the voxels that
//            can become occupied (every candidate point's voxel);
//   store -> the occupied elements, later entries win;   close -> unmap, close.
struct MhMatSparse {
    int fd;
    char *map;
    size_t total, prefix_bytes, nelem;
};

extern "C" int mh_mat_sparse_open(const char *path, const void *prefix, size_t prefix_bytes, size_t payload_bytes,
                                  void **handle) {
    if (!path || !handle || (!prefix && prefix_bytes) || (payload_bytes & 7) || (prefix_bytes & 7))
        return fail(MH_ERR_ARG, "mh_mat_sparse_open: bad arguments");
    *handle = nullptr;
    const int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return fail(MH_ERR_STATE, "mh_mat_sparse_open: cannot create %s", path);
    const size_t total = prefix_bytes + payload_bytes;
    void *m = MAP_FAILED;
    if ((size_t)write(fd, prefix, prefix_bytes) == prefix_bytes && ftruncate(fd, (off_t)total) == 0 && total)
        m = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (m == MAP_FAILED) {
        close(fd);
        return fail(MH_ERR_STATE, "mh_mat_sparse_open: writing %s failed", path);
    }
    *handle = new MhMatSparse{fd, (char *)m, total, prefix_bytes, payload_bytes / 8};
    return MH_OK;
}

extern "C" int mh_mat_sparse_touch(void *handle, const long long *elem_index, size_t n) {
    MhMatSparse *h = (MhMatSparse *)handle;
    if (!h || (n && !elem_index)) return fail(MH_ERR_ARG, "mh_mat_sparse_touch: bad arguments");
    // pages in ASCENDING order, each once (the block allocation of a sparse file is cheaper front to back than in the order
    // the points happen to come in: 3-5 ms instead of 6-11 for the two files of a pass)
    const size_t npage = (h->total >> 12) + 1;
    std::vector<bool> want(npage, false);
    for (size_t i = 0; i < n; ++i) {
        if (elem_index[i] < 0 || (size_t)elem_index[i] >= h->nelem) continue;      // (a hint: out-of-range entries are skipped)
        want[(h->prefix_bytes + (size_t)elem_index[i] * 8) >> 12] = true;
    }
    for (size_t pg = 0; pg < npage; ++pg) {
        if (!want[pg]) continue;
        size_t byte = pg << 12;
        if (byte < h->prefix_bytes) byte = h->prefix_bytes;      // (prefix and total are multiples of 8)
        if (byte + 8 > h->total) continue;
        volatile unsigned long long *q = (volatile unsigned long long *)(h->map + byte);
        *q = *q;      // a WRITE fault (a read would map the shared zero page); the value stays -- touch precedes store
    }
    return MH_OK;
}

extern "C" int mh_mat_sparse_store(void *handle, const long long *elem_index, const double *values, size_t n) {
    MhMatSparse *h = (MhMatSparse *)handle;
    if (!h || (n && (!elem_index || !values))) return fail(MH_ERR_ARG, "mh_mat_sparse_store: bad arguments");
    for (size_t i = 0; i < n; ++i)
        if (elem_index[i] < 0 || (size_t)elem_index[i] >= h->nelem)
            return fail(MH_ERR_ARG, "mh_mat_sparse_store: element %zu out of range", i);
    char *payload = h->map + h->prefix_bytes;
    for (size_t i = 0; i < n; ++i) memcpy(payload + (size_t)elem_index[i] * 8, &values[i], 8);
    return MH_OK;
}

// store straight from the voxel list of the fit: vox [G,3] (x, y, z), element (y + Y*(x + X*z)) [+ c*X*Y*Z for the three
// orientation channels of Ori]; ori == NULL writes 1.0 (Occ).  Later rows win, as the reference's fancy assignments do.
extern "C" int mh_mat_sparse_store_voxels(void *handle, const long long *vox, const void *ori, int ori_is_f64, size_t G, int X,
                                          int Y, int Z) {
    MhMatSparse *h = (MhMatSparse *)handle;
    const size_t plane = (size_t)X * Y * Z;
    if (!h || (G && !vox) || X < 1 || Y < 1 || Z < 1 || h->nelem != plane * (ori ? 3 : 1))
        return fail(MH_ERR_ARG, "mh_mat_sparse_store_voxels: bad arguments");
    for (size_t g = 0; g < G; ++g)
        if (vox[3 * g] < 0 || vox[3 * g] >= X || vox[3 * g + 1] < 0 || vox[3 * g + 1] >= Y || vox[3 * g + 2] < 0 ||
            vox[3 * g + 2] >= Z)
            return fail(MH_ERR_ARG, "mh_mat_sparse_store_voxels: voxel %zu outside the grid", g);
    double *payload = (double *)(h->map + h->prefix_bytes);
    for (size_t g = 0; g < G; ++g) {
        const size_t lin = (size_t)vox[3 * g + 1] + (size_t)Y * ((size_t)vox[3 * g] + (size_t)X * (size_t)vox[3 * g + 2]);
        if (!ori) {
            payload[lin] = 1.0;
        } else {
            for (int c = 0; c < 3; ++c)
                payload[lin + c * plane] = ori_is_f64 ? ((const double *)ori)[3 * g + c] : (double)((const float *)ori)[3 * g + c];
        }
    }
    return MH_OK;
}

extern "C" int mh_mat_sparse_close(void *handle) {
    MhMatSparse *h = (MhMatSparse *)handle;
    if (!h) return MH_OK;
    munmap(h->map, h->total);
    close(h->fd);
    delete h;
    return MH_OK;
}
