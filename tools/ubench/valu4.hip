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


#define BODY_V0 { float pA0,pA1,pA2,pA3,qA0,qA1,qA2,qA3,pB0,pB1,pB2,pB3,qB0,qB1,qB2,qB3; unsigned long long m0,m1,m2,m3; asm volatile("v_mul_f32_e32 %[pA0], %[tx], %[dx0]\n\t" \
        "v_mul_f32_e32 %[pA1], %[tx], %[dx1]\n\t" \
        "v_mul_f32_e32 %[pA2], %[tx], %[dx2]\n\t" \
        "v_mul_f32_e32 %[pA3], %[tx], %[dx3]\n\t" \
        "v_mul_f32_e32 %[qA0], %[ty], %[dy0]\n\t" \
        "v_mul_f32_e32 %[qA1], %[ty], %[dy1]\n\t" \
        "v_mul_f32_e32 %[qA2], %[ty], %[dy2]\n\t" \
        "v_mul_f32_e32 %[qA3], %[ty], %[dy3]\n\t" \
        "v_add_f32_e32 %[pA0], %[pA0], %[qA0]\n\t" \
        "v_add_f32_e32 %[pA1], %[pA1], %[qA1]\n\t" \
        "v_add_f32_e32 %[pA2], %[pA2], %[qA2]\n\t" \
        "v_add_f32_e32 %[pA3], %[pA3], %[qA3]\n\t" \
        "v_sub_f32_e64 %[pA0], 1.0, |%[pA0]|\n\t" \
        "v_sub_f32_e64 %[pA1], 1.0, |%[pA1]|\n\t" \
        "v_sub_f32_e64 %[pA2], 1.0, |%[pA2]|\n\t" \
        "v_sub_f32_e64 %[pA3], 1.0, |%[pA3]|\n\t" \
        "v_cmp_lt_f32_e64 %[m0], %[pA0], %[ml0]\n\t" \
        "v_cmp_lt_f32_e64 %[m1], %[pA1], %[ml1]\n\t" \
        "v_cmp_lt_f32_e64 %[m2], %[pA2], %[ml2]\n\t" \
        "v_cmp_lt_f32_e64 %[m3], %[pA3], %[ml3]\n\t" \
        "v_cndmask_b32_e64 %[ml0], %[ml0], %[pA0], %[m0]\n\t" \
        "v_cndmask_b32_e64 %[ml1], %[ml1], %[pA1], %[m1]\n\t" \
        "v_cndmask_b32_e64 %[ml2], %[ml2], %[pA2], %[m2]\n\t" \
        "v_cndmask_b32_e64 %[ml3], %[ml3], %[pA3], %[m3]\n\t" \
        "v_cndmask_b32_e64 %[bc0], %[bc0], %[tc], %[m0]\n\t" \
        "v_cndmask_b32_e64 %[bc1], %[bc1], %[tc], %[m1]\n\t" \
        "v_cndmask_b32_e64 %[bc2], %[bc2], %[tc], %[m2]\n\t" \
        "v_cndmask_b32_e64 %[bc3], %[bc3], %[tc], %[m3]\n\t" \
        "v_mul_f32_e32 %[pB0], %[tx], %[dx0]\n\t" \
        "v_mul_f32_e32 %[pB1], %[tx], %[dx1]\n\t" \
        "v_mul_f32_e32 %[pB2], %[tx], %[dx2]\n\t" \
        "v_mul_f32_e32 %[pB3], %[tx], %[dx3]\n\t" \
        "v_mul_f32_e32 %[qB0], %[ty], %[dy0]\n\t" \
        "v_mul_f32_e32 %[qB1], %[ty], %[dy1]\n\t" \
        "v_mul_f32_e32 %[qB2], %[ty], %[dy2]\n\t" \
        "v_mul_f32_e32 %[qB3], %[ty], %[dy3]\n\t" \
        "v_add_f32_e32 %[pB0], %[pB0], %[qB0]\n\t" \
        "v_add_f32_e32 %[pB1], %[pB1], %[qB1]\n\t" \
        "v_add_f32_e32 %[pB2], %[pB2], %[qB2]\n\t" \
        "v_add_f32_e32 %[pB3], %[pB3], %[qB3]\n\t" \
        "v_sub_f32_e64 %[pB0], 1.0, |%[pB0]|\n\t" \
        "v_sub_f32_e64 %[pB1], 1.0, |%[pB1]|\n\t" \
        "v_sub_f32_e64 %[pB2], 1.0, |%[pB2]|\n\t" \
        "v_sub_f32_e64 %[pB3], 1.0, |%[pB3]|\n\t" \
        "v_cmp_lt_f32_e64 %[m0], %[pB0], %[ml0]\n\t" \
        "v_cmp_lt_f32_e64 %[m1], %[pB1], %[ml1]\n\t" \
        "v_cmp_lt_f32_e64 %[m2], %[pB2], %[ml2]\n\t" \
        "v_cmp_lt_f32_e64 %[m3], %[pB3], %[ml3]\n\t" \
        "v_cndmask_b32_e64 %[ml0], %[ml0], %[pB0], %[m0]\n\t" \
        "v_cndmask_b32_e64 %[ml1], %[ml1], %[pB1], %[m1]\n\t" \
        "v_cndmask_b32_e64 %[ml2], %[ml2], %[pB2], %[m2]\n\t" \
        "v_cndmask_b32_e64 %[ml3], %[ml3], %[pB3], %[m3]\n\t" \
        "v_cndmask_b32_e64 %[bc0], %[bc0], %[tc], %[m0]\n\t" \
        "v_cndmask_b32_e64 %[bc1], %[bc1], %[tc], %[m1]\n\t" \
        "v_cndmask_b32_e64 %[bc2], %[bc2], %[tc], %[m2]\n\t" \
        "v_cndmask_b32_e64 %[bc3], %[bc3], %[tc], %[m3]" : [ml0] "+v"(r4.x), [ml1] "+v"(r4.y), [ml2] "+v"(r5.x), [ml3] "+v"(r5.y), [bc0] "+v"(r6.x), [bc1] "+v"(r6.y), [bc2] "+v"(r7.x), [bc3] "+v"(r7.y), [pA0] "=&v"(pA0), [qA0] "=&v"(qA0), [pA1] "=&v"(pA1), [qA1] "=&v"(qA1), [pA2] "=&v"(pA2), [qA2] "=&v"(qA2), [pA3] "=&v"(pA3), [qA3] "=&v"(qA3), [pB0] "=&v"(pB0), [qB0] "=&v"(qB0), [pB1] "=&v"(pB1), [qB1] "=&v"(qB1), [pB2] "=&v"(pB2), [qB2] "=&v"(qB2), [pB3] "=&v"(pB3), [qB3] "=&v"(qB3), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3) : [dx0] "v"(r0.x), [dx1] "v"(r0.y), [dx2] "v"(r2.x), [dx3] "v"(r2.y), [dy0] "v"(r1.x), [dy1] "v"(r1.y), [dy2] "v"(r3.x), [dy3] "v"(r3.y), [tx] "v"(s.x), [ty] "v"(s.y), [tc] "v"(t.x)); }
#define BODY_V1 { float pA0,pA1,pA2,pA3,qA0,qA1,qA2,qA3,pB0,pB1,pB2,pB3,qB0,qB1,qB2,qB3; unsigned long long m0,m1,m2,m3; asm volatile("v_mul_f32_e32 %[pA0], %[tx], %[dx0]\n\t" \
        "v_mul_f32_e32 %[pA1], %[tx], %[dx1]\n\t" \
        "v_mul_f32_e32 %[pA2], %[tx], %[dx2]\n\t" \
        "v_mul_f32_e32 %[pA3], %[tx], %[dx3]\n\t" \
        "v_mul_f32_e32 %[qA0], %[ty], %[dy0]\n\t" \
        "v_mul_f32_e32 %[qA1], %[ty], %[dy1]\n\t" \
        "v_mul_f32_e32 %[qA2], %[ty], %[dy2]\n\t" \
        "v_mul_f32_e32 %[qA3], %[ty], %[dy3]\n\t" \
        "v_add_f32_e32 %[pA0], %[pA0], %[qA0]\n\t" \
        "v_add_f32_e32 %[pA1], %[pA1], %[qA1]\n\t" \
        "v_add_f32_e32 %[pA2], %[pA2], %[qA2]\n\t" \
        "v_add_f32_e32 %[pA3], %[pA3], %[qA3]\n\t" \
        "v_sub_f32_e64 %[pA0], 1.0, |%[pA0]|\n\t" \
        "v_sub_f32_e64 %[pA1], 1.0, |%[pA1]|\n\t" \
        "v_sub_f32_e64 %[pA2], 1.0, |%[pA2]|\n\t" \
        "v_sub_f32_e64 %[pA3], 1.0, |%[pA3]|\n\t" \
        "v_cmp_lt_f32_e64 %[m0], %[pA0], %[ml0]\n\t" \
        "v_mul_f32_e32 %[pB0], %[tx], %[dx0]\n\t" \
        "v_mul_f32_e32 %[pB1], %[tx], %[dx1]\n\t" \
        "v_cmp_lt_f32_e64 %[m1], %[pA1], %[ml1]\n\t" \
        "v_mul_f32_e32 %[pB2], %[tx], %[dx2]\n\t" \
        "v_cmp_lt_f32_e64 %[m2], %[pA2], %[ml2]\n\t" \
        "v_mul_f32_e32 %[pB3], %[tx], %[dx3]\n\t" \
        "v_cmp_lt_f32_e64 %[m3], %[pA3], %[ml3]\n\t" \
        "v_mul_f32_e32 %[qB0], %[ty], %[dy0]\n\t" \
        "v_mul_f32_e32 %[qB1], %[ty], %[dy1]\n\t" \
        "v_cndmask_b32_e64 %[ml0], %[ml0], %[pA0], %[m0]\n\t" \
        "v_mul_f32_e32 %[qB2], %[ty], %[dy2]\n\t" \
        "v_cndmask_b32_e64 %[ml1], %[ml1], %[pA1], %[m1]\n\t" \
        "v_mul_f32_e32 %[qB3], %[ty], %[dy3]\n\t" \
        "v_cndmask_b32_e64 %[ml2], %[ml2], %[pA2], %[m2]\n\t" \
        "v_add_f32_e32 %[pB0], %[pB0], %[qB0]\n\t" \
        "v_add_f32_e32 %[pB1], %[pB1], %[qB1]\n\t" \
        "v_cndmask_b32_e64 %[ml3], %[ml3], %[pA3], %[m3]\n\t" \
        "v_add_f32_e32 %[pB2], %[pB2], %[qB2]\n\t" \
        "v_cndmask_b32_e64 %[bc0], %[bc0], %[tc], %[m0]\n\t" \
        "v_add_f32_e32 %[pB3], %[pB3], %[qB3]\n\t" \
        "v_cndmask_b32_e64 %[bc1], %[bc1], %[tc], %[m1]\n\t" \
        "v_sub_f32_e64 %[pB0], 1.0, |%[pB0]|\n\t" \
        "v_sub_f32_e64 %[pB1], 1.0, |%[pB1]|\n\t" \
        "v_cndmask_b32_e64 %[bc2], %[bc2], %[tc], %[m2]\n\t" \
        "v_sub_f32_e64 %[pB2], 1.0, |%[pB2]|\n\t" \
        "v_cndmask_b32_e64 %[bc3], %[bc3], %[tc], %[m3]\n\t" \
        "v_sub_f32_e64 %[pB3], 1.0, |%[pB3]|\n\t" \
        "v_cmp_lt_f32_e64 %[m0], %[pB0], %[ml0]\n\t" \
        "v_cmp_lt_f32_e64 %[m1], %[pB1], %[ml1]\n\t" \
        "v_cmp_lt_f32_e64 %[m2], %[pB2], %[ml2]\n\t" \
        "v_cmp_lt_f32_e64 %[m3], %[pB3], %[ml3]\n\t" \
        "v_cndmask_b32_e64 %[ml0], %[ml0], %[pB0], %[m0]\n\t" \
        "v_cndmask_b32_e64 %[ml1], %[ml1], %[pB1], %[m1]\n\t" \
        "v_cndmask_b32_e64 %[ml2], %[ml2], %[pB2], %[m2]\n\t" \
        "v_cndmask_b32_e64 %[ml3], %[ml3], %[pB3], %[m3]\n\t" \
        "v_cndmask_b32_e64 %[bc0], %[bc0], %[tc], %[m0]\n\t" \
        "v_cndmask_b32_e64 %[bc1], %[bc1], %[tc], %[m1]\n\t" \
        "v_cndmask_b32_e64 %[bc2], %[bc2], %[tc], %[m2]\n\t" \
        "v_cndmask_b32_e64 %[bc3], %[bc3], %[tc], %[m3]" : [ml0] "+v"(r4.x), [ml1] "+v"(r4.y), [ml2] "+v"(r5.x), [ml3] "+v"(r5.y), [bc0] "+v"(r6.x), [bc1] "+v"(r6.y), [bc2] "+v"(r7.x), [bc3] "+v"(r7.y), [pA0] "=&v"(pA0), [qA0] "=&v"(qA0), [pA1] "=&v"(pA1), [qA1] "=&v"(qA1), [pA2] "=&v"(pA2), [qA2] "=&v"(qA2), [pA3] "=&v"(pA3), [qA3] "=&v"(qA3), [pB0] "=&v"(pB0), [qB0] "=&v"(qB0), [pB1] "=&v"(pB1), [qB1] "=&v"(qB1), [pB2] "=&v"(pB2), [qB2] "=&v"(qB2), [pB3] "=&v"(pB3), [qB3] "=&v"(qB3), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3) : [dx0] "v"(r0.x), [dx1] "v"(r0.y), [dx2] "v"(r2.x), [dx3] "v"(r2.y), [dy0] "v"(r1.x), [dy1] "v"(r1.y), [dy2] "v"(r3.x), [dy3] "v"(r3.y), [tx] "v"(s.x), [ty] "v"(s.y), [tc] "v"(t.x)); }
#define BODY_V2 { float pA0,pA1,pA2,pA3,qA0,qA1,qA2,qA3,pB0,pB1,pB2,pB3,qB0,qB1,qB2,qB3; unsigned long long m0,m1,m2,m3; asm volatile("v_cmp_lt_f32_e64 %[m0], %[pA0], %[ml0]\n\t" \
        "v_mul_f32_e32 %[pB0], %[tx], %[dx0]\n\t" \
        "v_mul_f32_e32 %[pB1], %[tx], %[dx1]\n\t" \
        "v_cmp_lt_f32_e64 %[m1], %[pA1], %[ml1]\n\t" \
        "v_mul_f32_e32 %[pB2], %[tx], %[dx2]\n\t" \
        "v_cmp_lt_f32_e64 %[m2], %[pA2], %[ml2]\n\t" \
        "v_mul_f32_e32 %[pB3], %[tx], %[dx3]\n\t" \
        "v_cmp_lt_f32_e64 %[m3], %[pA3], %[ml3]\n\t" \
        "v_mul_f32_e32 %[qB0], %[ty], %[dy0]\n\t" \
        "v_mul_f32_e32 %[qB1], %[ty], %[dy1]\n\t" \
        "v_cndmask_b32_e64 %[ml0], %[ml0], %[pA0], %[m0]\n\t" \
        "v_mul_f32_e32 %[qB2], %[ty], %[dy2]\n\t" \
        "v_cndmask_b32_e64 %[ml1], %[ml1], %[pA1], %[m1]\n\t" \
        "v_mul_f32_e32 %[qB3], %[ty], %[dy3]\n\t" \
        "v_cndmask_b32_e64 %[ml2], %[ml2], %[pA2], %[m2]\n\t" \
        "v_add_f32_e32 %[pB0], %[pB0], %[qB0]\n\t" \
        "v_add_f32_e32 %[pB1], %[pB1], %[qB1]\n\t" \
        "v_cndmask_b32_e64 %[ml3], %[ml3], %[pA3], %[m3]\n\t" \
        "v_add_f32_e32 %[pB2], %[pB2], %[qB2]\n\t" \
        "v_cndmask_b32_e64 %[bc0], %[bc0], %[tc], %[m0]\n\t" \
        "v_add_f32_e32 %[pB3], %[pB3], %[qB3]\n\t" \
        "v_cndmask_b32_e64 %[bc1], %[bc1], %[tc], %[m1]\n\t" \
        "v_sub_f32_e64 %[pB0], 1.0, |%[pB0]|\n\t" \
        "v_sub_f32_e64 %[pB1], 1.0, |%[pB1]|\n\t" \
        "v_cndmask_b32_e64 %[bc2], %[bc2], %[tc], %[m2]\n\t" \
        "v_sub_f32_e64 %[pB2], 1.0, |%[pB2]|\n\t" \
        "v_cndmask_b32_e64 %[bc3], %[bc3], %[tc], %[m3]\n\t" \
        "v_sub_f32_e64 %[pB3], 1.0, |%[pB3]|\n\t" \
        "v_cmp_lt_f32_e64 %[m0], %[pB0], %[ml0]\n\t" \
        "v_mul_f32_e32 %[pA0], %[tx], %[dx0]\n\t" \
        "v_mul_f32_e32 %[pA1], %[tx], %[dx1]\n\t" \
        "v_cmp_lt_f32_e64 %[m1], %[pB1], %[ml1]\n\t" \
        "v_mul_f32_e32 %[pA2], %[tx], %[dx2]\n\t" \
        "v_cmp_lt_f32_e64 %[m2], %[pB2], %[ml2]\n\t" \
        "v_mul_f32_e32 %[pA3], %[tx], %[dx3]\n\t" \
        "v_cmp_lt_f32_e64 %[m3], %[pB3], %[ml3]\n\t" \
        "v_mul_f32_e32 %[qA0], %[ty], %[dy0]\n\t" \
        "v_mul_f32_e32 %[qA1], %[ty], %[dy1]\n\t" \
        "v_cndmask_b32_e64 %[ml0], %[ml0], %[pB0], %[m0]\n\t" \
        "v_mul_f32_e32 %[qA2], %[ty], %[dy2]\n\t" \
        "v_cndmask_b32_e64 %[ml1], %[ml1], %[pB1], %[m1]\n\t" \
        "v_mul_f32_e32 %[qA3], %[ty], %[dy3]\n\t" \
        "v_cndmask_b32_e64 %[ml2], %[ml2], %[pB2], %[m2]\n\t" \
        "v_add_f32_e32 %[pA0], %[pA0], %[qA0]\n\t" \
        "v_add_f32_e32 %[pA1], %[pA1], %[qA1]\n\t" \
        "v_cndmask_b32_e64 %[ml3], %[ml3], %[pB3], %[m3]\n\t" \
        "v_add_f32_e32 %[pA2], %[pA2], %[qA2]\n\t" \
        "v_cndmask_b32_e64 %[bc0], %[bc0], %[tc], %[m0]\n\t" \
        "v_add_f32_e32 %[pA3], %[pA3], %[qA3]\n\t" \
        "v_cndmask_b32_e64 %[bc1], %[bc1], %[tc], %[m1]\n\t" \
        "v_sub_f32_e64 %[pA0], 1.0, |%[pA0]|\n\t" \
        "v_sub_f32_e64 %[pA1], 1.0, |%[pA1]|\n\t" \
        "v_cndmask_b32_e64 %[bc2], %[bc2], %[tc], %[m2]\n\t" \
        "v_sub_f32_e64 %[pA2], 1.0, |%[pA2]|\n\t" \
        "v_cndmask_b32_e64 %[bc3], %[bc3], %[tc], %[m3]\n\t" \
        "v_sub_f32_e64 %[pA3], 1.0, |%[pA3]|" : [ml0] "+v"(r4.x), [ml1] "+v"(r4.y), [ml2] "+v"(r5.x), [ml3] "+v"(r5.y), [bc0] "+v"(r6.x), [bc1] "+v"(r6.y), [bc2] "+v"(r7.x), [bc3] "+v"(r7.y), [pA0] "=&v"(pA0), [qA0] "=&v"(qA0), [pA1] "=&v"(pA1), [qA1] "=&v"(qA1), [pA2] "=&v"(pA2), [qA2] "=&v"(qA2), [pA3] "=&v"(pA3), [qA3] "=&v"(qA3), [pB0] "=&v"(pB0), [qB0] "=&v"(qB0), [pB1] "=&v"(pB1), [qB1] "=&v"(qB1), [pB2] "=&v"(pB2), [qB2] "=&v"(qB2), [pB3] "=&v"(pB3), [qB3] "=&v"(qB3), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3) : [dx0] "v"(r0.x), [dx1] "v"(r0.y), [dx2] "v"(r2.x), [dx3] "v"(r2.y), [dy0] "v"(r1.x), [dy1] "v"(r1.y), [dy2] "v"(r3.x), [dy3] "v"(r3.y), [tx] "v"(s.x), [ty] "v"(s.y), [tc] "v"(t.x)); }
#define BODY_V4 { float pA0,pA1,pA2,pA3,qA0,qA1,qA2,qA3,pB0,pB1,pB2,pB3,qB0,qB1,qB2,qB3; unsigned long long m0,m1,m2,m3; asm volatile("v_cmp_lt_f32_e64 %[m0], %[pA0], %[ml0]\n\t" \
        "v_cmp_lt_f32_e64 %[m1], %[pA1], %[ml1]\n\t" \
        "v_cmp_lt_f32_e64 %[m2], %[pA2], %[ml2]\n\t" \
        "v_cmp_lt_f32_e64 %[m3], %[pA3], %[ml3]\n\t" \
        "v_cndmask_b32_e64 %[ml0], %[ml0], %[pA0], %[m0]\n\t" \
        "v_cndmask_b32_e64 %[ml1], %[ml1], %[pA1], %[m1]\n\t" \
        "v_cndmask_b32_e64 %[ml2], %[ml2], %[pA2], %[m2]\n\t" \
        "v_cndmask_b32_e64 %[ml3], %[ml3], %[pA3], %[m3]\n\t" \
        "v_cndmask_b32_e64 %[bc0], %[bc0], %[tc], %[m0]\n\t" \
        "v_cndmask_b32_e64 %[bc1], %[bc1], %[tc], %[m1]\n\t" \
        "v_cndmask_b32_e64 %[bc2], %[bc2], %[tc], %[m2]\n\t" \
        "v_cndmask_b32_e64 %[bc3], %[bc3], %[tc], %[m3]\n\t" \
        "v_mul_f32_e32 %[pA0], %[tx], %[dx0]\n\t" \
        "v_mul_f32_e32 %[pA1], %[tx], %[dx1]\n\t" \
        "v_mul_f32_e32 %[pA2], %[tx], %[dx2]\n\t" \
        "v_mul_f32_e32 %[pA3], %[tx], %[dx3]\n\t" \
        "v_mul_f32_e32 %[qA0], %[ty], %[dy0]\n\t" \
        "v_mul_f32_e32 %[qA1], %[ty], %[dy1]\n\t" \
        "v_mul_f32_e32 %[qA2], %[ty], %[dy2]\n\t" \
        "v_mul_f32_e32 %[qA3], %[ty], %[dy3]\n\t" \
        "v_add_f32_e32 %[pA0], %[pA0], %[qA0]\n\t" \
        "v_add_f32_e32 %[pA1], %[pA1], %[qA1]\n\t" \
        "v_add_f32_e32 %[pA2], %[pA2], %[qA2]\n\t" \
        "v_add_f32_e32 %[pA3], %[pA3], %[qA3]\n\t" \
        "v_sub_f32_e64 %[pA0], 1.0, |%[pA0]|\n\t" \
        "v_sub_f32_e64 %[pA1], 1.0, |%[pA1]|\n\t" \
        "v_sub_f32_e64 %[pA2], 1.0, |%[pA2]|\n\t" \
        "v_sub_f32_e64 %[pA3], 1.0, |%[pA3]|\n\t" \
        "v_cmp_lt_f32_e64 %[m0], %[pB0], %[ml0]\n\t" \
        "v_cmp_lt_f32_e64 %[m1], %[pB1], %[ml1]\n\t" \
        "v_cmp_lt_f32_e64 %[m2], %[pB2], %[ml2]\n\t" \
        "v_cmp_lt_f32_e64 %[m3], %[pB3], %[ml3]\n\t" \
        "v_cndmask_b32_e64 %[ml0], %[ml0], %[pB0], %[m0]\n\t" \
        "v_cndmask_b32_e64 %[ml1], %[ml1], %[pB1], %[m1]\n\t" \
        "v_cndmask_b32_e64 %[ml2], %[ml2], %[pB2], %[m2]\n\t" \
        "v_cndmask_b32_e64 %[ml3], %[ml3], %[pB3], %[m3]\n\t" \
        "v_cndmask_b32_e64 %[bc0], %[bc0], %[tc], %[m0]\n\t" \
        "v_cndmask_b32_e64 %[bc1], %[bc1], %[tc], %[m1]\n\t" \
        "v_cndmask_b32_e64 %[bc2], %[bc2], %[tc], %[m2]\n\t" \
        "v_cndmask_b32_e64 %[bc3], %[bc3], %[tc], %[m3]\n\t" \
        "v_mul_f32_e32 %[pB0], %[tx], %[dx0]\n\t" \
        "v_mul_f32_e32 %[pB1], %[tx], %[dx1]\n\t" \
        "v_mul_f32_e32 %[pB2], %[tx], %[dx2]\n\t" \
        "v_mul_f32_e32 %[pB3], %[tx], %[dx3]\n\t" \
        "v_mul_f32_e32 %[qB0], %[ty], %[dy0]\n\t" \
        "v_mul_f32_e32 %[qB1], %[ty], %[dy1]\n\t" \
        "v_mul_f32_e32 %[qB2], %[ty], %[dy2]\n\t" \
        "v_mul_f32_e32 %[qB3], %[ty], %[dy3]\n\t" \
        "v_add_f32_e32 %[pB0], %[pB0], %[qB0]\n\t" \
        "v_add_f32_e32 %[pB1], %[pB1], %[qB1]\n\t" \
        "v_add_f32_e32 %[pB2], %[pB2], %[qB2]\n\t" \
        "v_add_f32_e32 %[pB3], %[pB3], %[qB3]\n\t" \
        "v_sub_f32_e64 %[pB0], 1.0, |%[pB0]|\n\t" \
        "v_sub_f32_e64 %[pB1], 1.0, |%[pB1]|\n\t" \
        "v_sub_f32_e64 %[pB2], 1.0, |%[pB2]|\n\t" \
        "v_sub_f32_e64 %[pB3], 1.0, |%[pB3]|" : [ml0] "+v"(r4.x), [ml1] "+v"(r4.y), [ml2] "+v"(r5.x), [ml3] "+v"(r5.y), [bc0] "+v"(r6.x), [bc1] "+v"(r6.y), [bc2] "+v"(r7.x), [bc3] "+v"(r7.y), [pA0] "=&v"(pA0), [qA0] "=&v"(qA0), [pA1] "=&v"(pA1), [qA1] "=&v"(qA1), [pA2] "=&v"(pA2), [qA2] "=&v"(qA2), [pA3] "=&v"(pA3), [qA3] "=&v"(qA3), [pB0] "=&v"(pB0), [qB0] "=&v"(qB0), [pB1] "=&v"(pB1), [qB1] "=&v"(qB1), [pB2] "=&v"(pB2), [qB2] "=&v"(qB2), [pB3] "=&v"(pB3), [qB3] "=&v"(qB3), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3) : [dx0] "v"(r0.x), [dx1] "v"(r0.y), [dx2] "v"(r2.x), [dx3] "v"(r2.y), [dy0] "v"(r1.x), [dy1] "v"(r1.y), [dy2] "v"(r3.x), [dy3] "v"(r3.y), [tx] "v"(s.x), [ty] "v"(s.y), [tc] "v"(t.x)); }

#define BKERNEL(NAME, BODY)                                                                 \
    __global__ __launch_bounds__(256) void NAME(float *out, unsigned long long *cyc, float a, float b) { \
        PRO for (int i = 0; i < ITER; ++i) { BODY BODY BODY BODY } EPI }
BKERNEL(k_v0, BODY_V0)
BKERNEL(k_v1, BODY_V1)
BKERNEL(k_v2, BODY_V2)
BKERNEL(k_v4, BODY_V4)
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
    printf("--- blocked vs interleaved issue order of the same 56 instructions (2 taps x 4 items); cycles per tap per 4 items @2.4GHz\n");
    for (int w : {8, 5, 4, 2, 1}) {
        Res r;
        r = run(k_v0, "V0 blocked", d, c, w, 8 * 28.0); printf("      -> %.1f\n", r.cyc_per_inst * 28);
        r = run(k_v1, "V1 pipelined", d, c, w, 8 * 28.0); printf("      -> %.1f\n", r.cyc_per_inst * 28);
        r = run(k_v2, "V2 interleaved", d, c, w, 8 * 28.0); printf("      -> %.1f\n", r.cyc_per_inst * 28);
        r = run(k_v4, "V4 upd-first", d, c, w, 8 * 28.0); printf("      -> %.1f\n", r.cyc_per_inst * 28);
    }
    return 0;
}
