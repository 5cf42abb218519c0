/* mh_pmvo_lab.h -- LAB switches of libmhpmvo.so: A/B forms, cross-check kernels and one debug counter that the tests,
 * bench.py and the experiment scripts under tools/ use.  Every setting computes the same results as the default; nothing
 * here is part of the supported surface an integrator binds (include/mh_pmvo.h) and any of it may change or go between
 * rounds.  The reference has no counterpart for any of it. */
#ifndef MH_PMVO_LAB_H
#define MH_PMVO_LAB_H
#include "mh_pmvo.h"
#ifdef __cplusplus
extern "C" {
#endif

/*   "search_variant": 0 = default: mh_search3_kernel (tap lists staged in LDS), points in descending order of work;
 *       7: the same with the points in their natural order (A/B); 100 / 107: 0 / 7 with the compare-and-select tap body
 *       whatever "search_body" says; 1256: the portable mh_search_kernel, the cross-check of the shipped kernel (also
 *       what runs when the caller has no list lengths); 9 / 10 (109 / 110): the launch split bench.py times -- 9 = what
 *       precedes the search in the unfused sequence, 10 = mh_search3_kernel on what 9 left.
 *   "search_body": which tap body mh_search3_kernel runs.  0 (default) = by the maps: contexts whose views were ALL
 *       uploaded with mh_ctx_set_view_u8 (tap lists of ~2 entries after the exact duplicate removal) take the
 *       compare-and-select body, all others (lists of ~45 entries) the key body -- the running minimum as one integer key
 *       per candidate, v_min3_u32 over two taps at a time; 1 = key body, 2 = select body.
 *   "tap_codes": 1 (default) = contexts whose views were ALL uploaded with mh_ctx_set_view_u8 gather a patch tap as the
 *       two resident 8-bit codes of its pixel (mh_project_taps_codes_kernel); 0 = always the decoded records.
 *   "tap_plane": 1 (default) = use the plane of ready-made taps when the context has one (mh_pmvo.h: "tap_plane_max_mb");
 *       0 = normalise per iteration (A/B and cross-check).
 *   "taps_tile": points per wave of the fp32 front end (mh_project_taps2_kernel): 64 (default; any other value) gives the
 *       fastest iteration, 32 / 16 the kernel's own best time.
 *   "filter_rows": 1 (default) = mh_filter_points launches of >= 4096 points vote with lane = point (mh_filter_rows_kernel;
 *       the rows whose sums take another order stay with the wave-per-point kernel), 0 = wave per point for every row. */
int mh_ctx_set_lab_option(mh_ctx *ctx, const char *key, int value);

/* Debug counter of the search's key body (csrc/pmvo_search.hip: mh_tap_key): out[2] = how many (wave, view) visits were
 * evaluated a second time with the compare-and-select body because a key could not state the winner (a best tap with
 * |cos| <= 2^-14, a NaN).  Process-wide, all contexts; reset != 0 clears it after the read.  out[0], out[1], out[3] are
 * only counted in the -DMH_KEY_STATS build (tools/exp_key_stats.py).  It exists so that a test can prove it entered that
 * branch (tests/test_key_reeval_gpu.py).  Synchronises the device. */
int mh_debug_key_stats(unsigned long long *out /* 4 */, int reset);

#ifdef __cplusplus
}
#endif
#endif /* MH_PMVO_LAB_H */
