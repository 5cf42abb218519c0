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


// ---- whole tap bodies, 4 items per lane.  One asm volatile per instruction: order is kept, registers are the compiler's.
#define PKMUL_LO(D, T, X) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(D) : "v"(T), "v"(X))
#define PKMUL_HI(D, T, X) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(D) : "v"(T), "v"(X))
#define PKADD(D, A, B) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(D) : "v"(A), "v"(B))
#define SUBABS(D, X) asm volatile("v_sub_f32_e64 %0, 1.0, |%1|" : "=v"(D) : "v"(X))
#define ADD(D, A, B) asm volatile("v_add_f32_e32 %0, %1, %2" : "=v"(D) : "v"(A), "v"(B))
#define MUL(D, A, B) asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(D) : "v"(A), "v"(B))
#define CMP(M, A, B) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(M) : "v"(A), "v"(B))
#define CND(D, X, M) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(D) : "v"(X), "s"(M))
#define MINF(D, X) asm volatile("v_min_f32_e32 %0, %0, %1" : "+v"(D) : "v"(X))

// B0: the shipped body: 4 pk_mul, 2 pk_add, 4 sub|abs|, 4 cmp -> 4 sgpr masks, 8 cndmask  (22 instr / 4 items)
#define BODY_B0 { v2 p0, p1, p2, p3; float l0, l1, l2, l3; unsigned long long m0, m1, m2, m3;                \
      PKMUL_LO(p0, s, r0); PKMUL_HI(p1, s, r1); PKMUL_LO(p2, s, r2); PKMUL_HI(p3, s, r3);                     \
      PKADD(p0, p0, p1); PKADD(p2, p2, p3);                                                                   \
      SUBABS(l0, p0.x); SUBABS(l1, p0.y); SUBABS(l2, p2.x); SUBABS(l3, p2.y);                                 \
      CMP(m0, l0, r4.x); CMP(m1, l1, r4.y); CMP(m2, l2, r5.x); CMP(m3, l3, r5.y);                             \
      CND(r4.x, l0, m0); CND(r4.y, l1, m1); CND(r5.x, l2, m2); CND(r5.y, l3, m3);                             \
      CND(r6.x, t.x, m0); CND(r6.y, t.x, m1); CND(r7.x, t.x, m2); CND(r7.y, t.x, m3); }
// B1: scalar multiplies/adds instead of packed (8 mul, 4 add), rest as B0   (28 instr / 4 items)
#define BODY_B1 { float a0, a1, a2, a3, b0, b1, b2, b3; float l0, l1, l2, l3; unsigned long long m0, m1, m2, m3; \
      MUL(a0, s.x, r0.x); MUL(a1, s.x, r0.y); MUL(a2, s.x, r2.x); MUL(a3, s.x, r2.y);                         \
      MUL(b0, s.y, r1.x); MUL(b1, s.y, r1.y); MUL(b2, s.y, r3.x); MUL(b3, s.y, r3.y);                         \
      ADD(a0, a0, b0); ADD(a1, a1, b1); ADD(a2, a2, b2); ADD(a3, a3, b3);                                     \
      SUBABS(l0, a0); SUBABS(l1, a1); SUBABS(l2, a2); SUBABS(l3, a3);                                         \
      CMP(m0, l0, r4.x); CMP(m1, l1, r4.y); CMP(m2, l2, r5.x); CMP(m3, l3, r5.y);                             \
      CND(r4.x, l0, m0); CND(r4.y, l1, m1); CND(r5.x, l2, m2); CND(r5.y, l3, m3);                             \
      CND(r6.x, t.x, m0); CND(r6.y, t.x, m1); CND(r7.x, t.x, m2); CND(r7.y, t.x, m3); }
// B2: products come from elsewhere (MFMA): per item add, sub|abs|, cmp, 2 cndmask   (20 instr / 4 items)
#define BODY_B2 { float a0, a1, a2, a3; float l0, l1, l2, l3; unsigned long long m0, m1, m2, m3;              \
      ADD(a0, r0.x, r1.x); ADD(a1, r0.y, r1.y); ADD(a2, r2.x, r3.x); ADD(a3, r2.y, r3.y);                     \
      SUBABS(l0, a0); SUBABS(l1, a1); SUBABS(l2, a2); SUBABS(l3, a3);                                         \
      CMP(m0, l0, r4.x); CMP(m1, l1, r4.y); CMP(m2, l2, r5.x); CMP(m3, l3, r5.y);                             \
      CND(r4.x, l0, m0); CND(r4.y, l1, m1); CND(r5.x, l2, m2); CND(r5.y, l3, m3);                             \
      CND(r6.x, t.x, m0); CND(r6.y, t.x, m1); CND(r7.x, t.x, m2); CND(r7.y, t.x, m3); }
// B3: value-only minimum (no argmin): 4 pk_mul, 2 pk_add, 4 sub, 4 v_min   (14 instr / 4 items)
#define BODY_B3 { v2 p0, p1, p2, p3; float l0, l1, l2, l3;                                                    \
      PKMUL_LO(p0, s, r0); PKMUL_HI(p1, s, r1); PKMUL_LO(p2, s, r2); PKMUL_HI(p3, s, r3);                     \
      PKADD(p0, p0, p1); PKADD(p2, p2, p3);                                                                   \
      SUBABS(l0, p0.x); SUBABS(l1, p0.y); SUBABS(l2, p2.x); SUBABS(l3, p2.y);                                 \
      MINF(r4.x, l0); MINF(r4.y, l1); MINF(r5.x, l2); MINF(r5.y, l3); }
// B4: only the compare/select block (4 cmp + 8 cndmask)
#define BODY_B4 { unsigned long long m0, m1, m2, m3;                                                          \
      CMP(m0, r0.x, r4.x); CMP(m1, r0.y, r4.y); CMP(m2, r1.x, r5.x); CMP(m3, r1.y, r5.y);                     \
      CND(r4.x, r0.x, m0); CND(r4.y, r0.y, m1); CND(r5.x, r1.x, m2); CND(r5.y, r1.y, m3);                     \
      CND(r6.x, t.x, m0); CND(r6.y, t.x, m1); CND(r7.x, t.x, m2); CND(r7.y, t.x, m3); }
// B5: only the arithmetic block (4 pk_mul + 2 pk_add + 4 sub)
#define BODY_B5 { v2 p0, p1, p2, p3;                                                                          \
      PKMUL_LO(p0, s, r0); PKMUL_HI(p1, s, r1); PKMUL_LO(p2, s, r2); PKMUL_HI(p3, s, r3);                     \
      PKADD(p0, p0, p1); PKADD(p2, p2, p3);                                                                   \
      SUBABS(r4.x, p0.x); SUBABS(r4.y, p0.y); SUBABS(r5.x, p2.x); SUBABS(r5.y, p2.y); }

#define BKERNEL(NAME, BODY)                                                                 \
    __global__ __launch_bounds__(256) void NAME(float *out, unsigned long long *cyc, float a, float b) { \
        PRO for (int i = 0; i < ITER; ++i) { BODY BODY BODY BODY BODY BODY BODY BODY } EPI }
BKERNEL(k_b0, BODY_B0)
BKERNEL(k_b1, BODY_B1)
BKERNEL(k_b2, BODY_B2)
BKERNEL(k_b3, BODY_B3)
BKERNEL(k_b4, BODY_B4)
BKERNEL(k_b5, BODY_B5)

// ---- MFMA beside VALU: one v_mfma_f32_32x32x1_2b_f32 (64 cycles on the matrix pipe) per NV plain VALU instructions, same wave
template <int NV>
__global__ __launch_bounds__(256) void k_mfma_valu(float *out, unsigned long long *cyc, float a, float b) {
    PRO
    v32 acc0 = {0}, acc1 = {0};
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            asm volatile("v_mfma_f32_32x32x1_2b_f32 %0, %1, %2, 0" : "=v"(acc0) : "v"(s.x), "v"(t.x));
#pragma unroll
            for (int q = 0; q < NV / 8; ++q) { ALL8_1("v_mul_f32_e32 %0, %0, %1") }
            asm volatile("v_mfma_f32_32x32x1_2b_f32 %0, %1, %2, 0" : "=v"(acc1) : "v"(s.y), "v"(t.y));
#pragma unroll
            for (int q = 0; q < NV / 8; ++q) { ALL8_1("v_add_f32_e32 %0, %0, %1") }
        }
    }
    r0.x += acc0[0] + acc1[3] + acc0[17] + acc1[31];
    EPI
}

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
#define R1(K, NAME) for (int w : {8, 2}) run(K, NAME, d, c, w, 32.0);
    R1(k_mul, "v_mul_e32") R1(k_mul64, "v_mul_e64") R1(k_add, "v_add") R1(k_fma, "v_fma") R1(k_fmac, "v_fmac")
    R1(k_sub_abs, "v_sub|abs|") R1(k_cmp32, "v_cmp_e32") R1(k_cmp64, "v_cmp_e64") R1(k_cnd32, "v_cnd_e32") R1(k_cnd64, "v_cnd_e64")
    R1(k_mov, "v_mov") R1(k_min, "v_min_f32") R1(k_max, "v_max_f32") R1(k_min3, "v_min3_f32") R1(k_med3, "v_med3_f32")
    R1(k_minu, "v_min_u32") R1(k_mini, "v_min_i32") R1(k_and, "v_and") R1(k_andor, "v_and_or") R1(k_bfi, "v_bfi") R1(k_ashr, "v_ashr")
    R1(k_addu, "v_add_u32") R1(k_subu, "v_sub_u32") R1(k_lshl, "v_lshl") R1(k_xor, "v_xor") R1(k_perm, "v_perm")
    R1(k_pk_mul, "v_pk_mul") R1(k_pk_add, "v_pk_add") R1(k_pk_fma, "v_pk_fma") R1(k_pk_mov, "v_pk_mov")
    R1(k_rcp, "v_rcp") R1(k_rsq, "v_rsq") R1(k_sqrt, "v_sqrt") R1(k_mul_dpp, "v_mul_dpp") R1(k_min_sdwa, "v_min_sdwa") R1(k_max3abs, "v_max3|abs|")
    printf("--- tap bodies (cycles per INSTRUCTION; multiply by the instruction count for cycles per tap per 4 items)\n");
    for (int w : {8, 5, 4, 2, 1}) {
        Res r;
        r = run(k_b0, "B0 shipped(22)", d, c, w, 8 * 22.0); printf("      -> %.1f cycles per tap per 4 items\n", r.cyc_per_inst * 22);
        r = run(k_b1, "B1 scalar(28)", d, c, w, 8 * 28.0); printf("      -> %.1f cycles per tap per 4 items\n", r.cyc_per_inst * 28);
        r = run(k_b2, "B2 noprod(20)", d, c, w, 8 * 20.0); printf("      -> %.1f cycles per tap per 4 items\n", r.cyc_per_inst * 20);
        r = run(k_b3, "B3 minonly(14)", d, c, w, 8 * 14.0); printf("      -> %.1f cycles per tap per 4 items\n", r.cyc_per_inst * 14);
        r = run(k_b4, "B4 cmpsel(12)", d, c, w, 8 * 12.0); printf("      -> %.1f cycles per tap per 4 items\n", r.cyc_per_inst * 12);
        r = run(k_b5, "B5 arith(10)", d, c, w, 8 * 10.0); printf("      -> %.1f cycles per tap per 4 items\n", r.cyc_per_inst * 10);
    }
    printf("--- MFMA (32x32x1_2b f32, 64 cyc matrix pipe) beside VALU in the same wave: cycles per 2 MFMA + 2*NV VALU\n");
    for (int w : {4, 2, 1}) {
        Res r;
        r = run(k_mfma_valu<8>, "mfma+8 valu", d, c, w, 4.0); printf("      -> %.1f cycles per (MFMA + 8 VALU) x2\n", r.cyc_per_inst);
        r = run(k_mfma_valu<16>, "mfma+16 valu", d, c, w, 4.0); printf("      -> %.1f cycles per (MFMA + 16 VALU) x2\n", r.cyc_per_inst);
        r = run(k_mfma_valu<32>, "mfma+32 valu", d, c, w, 4.0); printf("      -> %.1f cycles per (MFMA + 32 VALU) x2\n", r.cyc_per_inst);
        r = run(k_mfma_valu<64>, "mfma+64 valu", d, c, w, 4.0); printf("      -> %.1f cycles per (MFMA + 64 VALU) x2\n", r.cyc_per_inst);
    }
    return 0;
}
