// Micro-benchmark 2 (gfx950): issue cost of single VALU instructions and of whole candidate tap bodies of
// mh_search_kernel, at 1/2/4/8 waves per SIMD.  Prints cycles per wave-instruction per SIMD using the
// shader clock measured with s_memtime around the loop (so DVFS does not distort the figure).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define ITER 1000
typedef float v2 __attribute__((ext_vector_type(2)));
typedef float v16 __attribute__((ext_vector_type(16)));
typedef float v32 __attribute__((ext_vector_type(32)));

#define PRO                                                                                                   \
    v2 r0 = {a + threadIdx.x, b}, r1 = {b, a}, r2 = {a, a}, r3 = {b, b}, r4 = {a, b + 1}, r5 = {a + 2, b},   \
       r6 = {a, b + 3}, r7 = {a + 4, b};                                                                      \
    v2 s = {a, b}, t = {b, a};                                                                                \
    unsigned long long c0 = __builtin_readcyclecounter();
#define EPI                                                                                                   \
    unsigned long long c1 = __builtin_readcyclecounter();                                                     \
    out[blockIdx.x * 256 + threadIdx.x] =                                                                     \
        r0.x + r1.x + r2.x + r3.x + r4.x + r5.x + r6.x + r7.x + r0.y + r1.y + r2.y + r3.y + r4.y + r5.y + r6.y + r7.y + s.x + t.x; \
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = c1 - c0;

#define KERNEL(NAME, BODY)                                                                  \
    __global__ __launch_bounds__(256) void NAME(float *out, unsigned long long *cyc, float a, float b) { \
        PRO for (int i = 0; i < ITER; ++i) { BODY BODY BODY BODY } EPI }

#define A1(OP, R) asm volatile(OP : "+v"(R.x) : "v"(s.x), "v"(t.x));
#define ALL8_1(OP) A1(OP, r0) A1(OP, r1) A1(OP, r2) A1(OP, r3) A1(OP, r4) A1(OP, r5) A1(OP, r6) A1(OP, r7)
#define A2(OP, R) asm volatile(OP : "+v"(R) : "v"(s), "v"(t));
#define ALL8_2(OP) A2(OP, r0) A2(OP, r1) A2(OP, r2) A2(OP, r3) A2(OP, r4) A2(OP, r5) A2(OP, r6) A2(OP, r7)

KERNEL(k_mul, ALL8_1("v_mul_f32_e32 %0, %0, %1"))
KERNEL(k_mul64, ALL8_1("v_mul_f32_e64 %0, %0, %1"))
KERNEL(k_add, ALL8_1("v_add_f32_e32 %0, %0, %1"))
KERNEL(k_fma, ALL8_1("v_fma_f32 %0, %0, %1, %2"))
KERNEL(k_fmac, ALL8_1("v_fmac_f32_e32 %0, %1, %2"))
KERNEL(k_sub_abs, ALL8_1("v_sub_f32_e64 %0, 1.0, |%0|"))
KERNEL(k_cmp32, ALL8_1("v_cmp_lt_f32_e32 vcc, %0, %1"))
KERNEL(k_cmp64, ALL8_1("v_cmp_lt_f32_e64 s[40:41], %0, %1"))
KERNEL(k_cnd32, ALL8_1("v_cndmask_b32_e32 %0, %0, %1, vcc"))
KERNEL(k_cnd64, ALL8_1("v_cndmask_b32_e64 %0, %0, %1, s[40:41]"))
KERNEL(k_mov, ALL8_1("v_mov_b32 %0, %1"))
KERNEL(k_min, ALL8_1("v_min_f32_e32 %0, %0, %1"))
KERNEL(k_max, ALL8_1("v_max_f32_e32 %0, %0, %1"))
KERNEL(k_min3, ALL8_1("v_min3_f32 %0, %0, %1, %2"))
KERNEL(k_med3, ALL8_1("v_med3_f32 %0, %0, %1, %2"))
KERNEL(k_minu, ALL8_1("v_min_u32_e32 %0, %0, %1"))
KERNEL(k_mini, ALL8_1("v_min_i32_e32 %0, %0, %1"))
KERNEL(k_and, ALL8_1("v_and_b32_e32 %0, %0, %1"))
KERNEL(k_andor, ALL8_1("v_and_or_b32 %0, %0, %1, %2"))
KERNEL(k_bfi, ALL8_1("v_bfi_b32 %0, %0, %1, %2"))
KERNEL(k_ashr, ALL8_1("v_ashrrev_i32_e32 %0, 31, %0"))
KERNEL(k_addu, ALL8_1("v_add_u32_e32 %0, %0, %1"))
KERNEL(k_subu, ALL8_1("v_sub_u32_e32 %0, %0, %1"))
KERNEL(k_lshl, ALL8_1("v_lshlrev_b32_e32 %0, 3, %0"))
KERNEL(k_xor, ALL8_1("v_xor_b32_e32 %0, %0, %1"))
KERNEL(k_perm, ALL8_1("v_perm_b32 %0, %0, %1, %2"))
KERNEL(k_pk_mul, ALL8_2("v_pk_mul_f32 %0, %0, %1"))
KERNEL(k_pk_add, ALL8_2("v_pk_add_f32 %0, %0, %1"))
KERNEL(k_pk_fma, ALL8_2("v_pk_fma_f32 %0, %0, %1, %2"))
KERNEL(k_pk_mov, ALL8_2("v_pk_mov_b32 %0, %0, %1"))
KERNEL(k_rcp, ALL8_1("v_rcp_f32 %0, %0"))
KERNEL(k_rsq, ALL8_1("v_rsq_f32 %0, %0"))
KERNEL(k_sqrt, ALL8_1("v_sqrt_f32 %0, %0"))
KERNEL(k_mul_dpp, ALL8_1("v_mul_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"))
KERNEL(k_min_sdwa, ALL8_1("v_min_f32_sdwa %0, %0, %1 dst_sel:DWORD src0_sel:DWORD src1_sel:DWORD"))
KERNEL(k_max3abs, ALL8_1("v_max3_f32 %0, %0, |%1|, |%2|"))



#define SUBABS(D, X) asm volatile("v_sub_f32_e64 %0, 1.0, |%1|" : "=v"(D) : "v"(X))
#define ADD(D, A, B) asm volatile("v_add_f32_e32 %0, %1, %2" : "=v"(D) : "v"(A), "v"(B))
#define MUL(D, A, B) asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(D) : "v"(A), "v"(B))
#define CMP(M, A, B) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(M) : "v"(A), "v"(B))
#define CND(D, X, M) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(D) : "v"(X), "s"(M))
#define ARITH float a0, a1, a2, a3, b0, b1, b2, b3; float l0, l1, l2, l3;                                       \
      MUL(a0, s.x, r0.x); MUL(a1, s.x, r0.y); MUL(a2, s.x, r2.x); MUL(a3, s.x, r2.y);                         \
      MUL(b0, s.y, r1.x); MUL(b1, s.y, r1.y); MUL(b2, s.y, r3.x); MUL(b3, s.y, r3.y);                         \
      ADD(a0, a0, b0); ADD(a1, a1, b1); ADD(a2, a2, b2); ADD(a3, a3, b3);                                     \
      SUBABS(l0, a0); SUBABS(l1, a1); SUBABS(l2, a2); SUBABS(l3, a3);
// U1: compare into SGPR masks, two selects per item (the shipped update, on scalar arithmetic)
#define BODY_U1 { ARITH unsigned long long m0, m1, m2, m3;                                                      \
      CMP(m0, l0, r4.x); CMP(m1, l1, r4.y); CMP(m2, l2, r5.x); CMP(m3, l3, r5.y);                             \
      CND(r4.x, l0, m0); CND(r4.y, l1, m1); CND(r5.x, l2, m2); CND(r5.y, l3, m3);                             \
      CND(r6.x, t.x, m0); CND(r6.y, t.x, m1); CND(r7.x, t.x, m2); CND(r7.y, t.x, m3); }
// U4: v_cmpx narrows EXEC to the lanes that improve, two plain moves, EXEC restored by the scalar unit
#define UPD_X(ML, BC, L) asm volatile("v_cmpx_lt_f32_e32 vcc, %2, %0\n\tv_mov_b32_e32 %0, %2\n\tv_mov_b32_e32 %1, %3\n\ts_mov_b64 exec, -1" \
                                      : "+v"(ML), "+v"(BC) : "v"(L), "v"(t.x) : "vcc")
#define BODY_U4 { ARITH UPD_X(r4.x, r6.x, l0); UPD_X(r4.y, r6.y, l1); UPD_X(r5.x, r7.x, l2); UPD_X(r5.y, r7.y, l3); }
// U5: v_cmp into an SGPR mask, the scalar unit copies it to EXEC, two plain moves
#define BODY_U5 { ARITH unsigned long long m0, m1, m2, m3;                                                      \
      CMP(m0, l0, r4.x); CMP(m1, l1, r4.y); CMP(m2, l2, r5.x); CMP(m3, l3, r5.y);                             \
      asm volatile("s_mov_b64 exec, %8\n\tv_mov_b32_e32 %0, %12\n\tv_mov_b32_e32 %4, %16\n\t"                 \
                   "s_mov_b64 exec, %9\n\tv_mov_b32_e32 %1, %13\n\tv_mov_b32_e32 %5, %16\n\t"                 \
                   "s_mov_b64 exec, %10\n\tv_mov_b32_e32 %2, %14\n\tv_mov_b32_e32 %6, %16\n\t"                \
                   "s_mov_b64 exec, %11\n\tv_mov_b32_e32 %3, %15\n\tv_mov_b32_e32 %7, %16\n\t"                \
                   "s_mov_b64 exec, -1"                                                                         \
                   : "+v"(r4.x), "+v"(r4.y), "+v"(r5.x), "+v"(r5.y), "+v"(r6.x), "+v"(r6.y), "+v"(r7.x), "+v"(r7.y) \
                   : "s"(m0), "s"(m1), "s"(m2), "s"(m3), "v"(l0), "v"(l1), "v"(l2), "v"(l3), "v"(t.x)); }
// U6: as U4 with one 64-bit move of the {loss, confidence} pair (v_pk_mov_b32 picks lo of src0, lo of src1)
#define UPD_P(MB, L) asm volatile("v_cmpx_lt_f32_e32 vcc, %1, %L0\n\tv_pk_mov_b32 %0, %1, %2 op_sel:[0,0]\n\ts_mov_b64 exec, -1" \
                                      : "+v"(MB) : "v"(L), "v"(t) : "vcc")
// U7: as U4 but all four v_cmpx first is impossible (EXEC), so interleave the scalar restores as early as possible: same as U4
// U8: compare through the integer unit: d = l - ML (as floats, full rate), sign -> mask -> bit select (all full-rate ops)
#define UPD_I(ML, BC, L) { float d; int m; asm volatile("v_sub_f32_e32 %0, %1, %2" : "=v"(d) : "v"(L), "v"(ML));       \
      asm volatile("v_ashrrev_i32_e32 %0, 31, %1" : "=v"(m) : "v"(d));                                         \
      asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(ML) : "v"(m), "v"(L));                                    \
      asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(BC) : "v"(m), "v"(t.x)); }
#define BODY_U8 { ARITH UPD_I(r4.x, r6.x, l0) UPD_I(r4.y, r6.y, l1) UPD_I(r5.x, r7.x, l2) UPD_I(r5.y, r7.y, l3) }
// U9: value by v_min_f32, confidence by cmp + one select
#define BODY_U9 { ARITH unsigned long long m0, m1, m2, m3;                                                      \
      CMP(m0, l0, r4.x); CMP(m1, l1, r4.y); CMP(m2, l2, r5.x); CMP(m3, l3, r5.y);                             \
      asm volatile("v_min_f32_e32 %0, %0, %1" : "+v"(r4.x) : "v"(l0)); asm volatile("v_min_f32_e32 %0, %0, %1" : "+v"(r4.y) : "v"(l1)); \
      asm volatile("v_min_f32_e32 %0, %0, %1" : "+v"(r5.x) : "v"(l2)); asm volatile("v_min_f32_e32 %0, %0, %1" : "+v"(r5.y) : "v"(l3)); \
      CND(r6.x, t.x, m0); CND(r6.y, t.x, m1); CND(r7.x, t.x, m2); CND(r7.y, t.x, m3); }
// UA: arithmetic only (scalar) -- the floor of any update scheme
#define BODY_UA { ARITH r4.x += l0; r4.y += l1; r5.x += l2; r5.y += l3; }

#define BKERNEL(NAME, BODY)                                                                 \
    __global__ __launch_bounds__(256) void NAME(float *out, unsigned long long *cyc, float a, float b) { \
        PRO for (int i = 0; i < ITER; ++i) { BODY BODY BODY BODY BODY BODY BODY BODY } EPI }
BKERNEL(k_u1, BODY_U1)
BKERNEL(k_u4, BODY_U4)
BKERNEL(k_u5, BODY_U5)
BKERNEL(k_u8, BODY_U8)
BKERNEL(k_u9, BODY_U9)
BKERNEL(k_ua, BODY_UA)

struct Res { double cyc_per_inst; double ms; };
template <typename K>
Res run(K kern, const char *name, float *d_out, unsigned long long *d_cyc, int waves_per_simd, double inst_per_iter) {
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, 1.0f, 0.999f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, 1.0f, 0.999f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long cyc;
    hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
    // s_memtime ticks at a constant 100 MHz on gfx9; convert through the wall clock instead: report both
    const double per_simd = (double)waves_per_simd * ITER * inst_per_iter;
    const double ns = ms * 1e6 / per_simd;
    Res r{ns * 2.4, ms};
    printf("%-14s w/simd=%d  %8.3f ms  %6.3f ns/inst/SIMD  = %5.2f cyc@2.4GHz  (memtime: %.2f ticks/inst/SIMD)\n", name, waves_per_simd, ms, ns,
           ns * 2.4, (double)cyc / (ITER * inst_per_iter * waves_per_simd));
    return r;
}


int main() {
    float *d;
    unsigned long long *c;
    hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    hipMalloc(&c, 64);
    printf("--- tap bodies on scalar arithmetic: ms for ITER*8 taps per wave; cycles per tap per 4 items at an assumed 2.4 GHz\n");
    for (int w : {8, 5, 4, 2}) {
        Res r;
        r = run(k_u1, "U1 cmp+2cnd", d, c, w, 8 * 28.0); printf("      -> %.1f cycles per tap per 4 items\n", r.cyc_per_inst * 28);
        r = run(k_u4, "U4 cmpx+2mov", d, c, w, 8 * 28.0); printf("      -> %.1f cycles per tap per 4 items\n", r.cyc_per_inst * 28);
        r = run(k_u5, "U5 cmp,sexec", d, c, w, 8 * 28.0); printf("      -> %.1f cycles per tap per 4 items\n", r.cyc_per_inst * 28);
        r = run(k_u8, "U8 bfi", d, c, w, 8 * 28.0); printf("      -> %.1f cycles per tap per 4 items\n", r.cyc_per_inst * 28);
        r = run(k_u9, "U9 min+cnd", d, c, w, 8 * 28.0); printf("      -> %.1f cycles per tap per 4 items\n", r.cyc_per_inst * 28);
        r = run(k_ua, "UA arith", d, c, w, 8 * 28.0); printf("      -> %.1f cycles per tap per 4 items\n", r.cyc_per_inst * 28);
    }
    return 0;
}
