// Micro-benchmark: issue cost of the VALU instructions the PMVO kernels are made of (gfx950).
// Each kernel runs a long unrolled stream of ONE instruction kind on independent registers;
// reports cycles per wave-instruction per SIMD assuming all 1024 SIMDs busy with 8 waves each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
#define ITER 2000

#define KERNEL(NAME, BODY, NREG)                                                                   \
    __global__ __launch_bounds__(256) void NAME(float *out, float a, float b) {                   \
        typedef float v2 __attribute__((ext_vector_type(2)));                                      \
        v2 r0 = {a + threadIdx.x, b}, r1 = {b, a}, r2 = {a, a}, r3 = {b, b}, r4 = {a, b + 1},      \
           r5 = {a + 2, b}, r6 = {a, b + 3}, r7 = {a + 4, b};                                      \
        v2 s = {a, b};                                                                             \
        for (int i = 0; i < ITER; ++i) {                                                           \
            _Pragma("unroll") for (int k = 0; k < REP / 8; ++k) { BODY }                          \
        }                                                                                          \
        out[blockIdx.x * 256 + threadIdx.x] =                                                      \
            r0.x + r1.x + r2.x + r3.x + r4.x + r5.x + r6.x + r7.x + r0.y + r1.y + r2.y + r3.y;    \
    }

#define A1(OP, R) asm volatile(OP : "+v"(R.x) : "v"(s.x));
#define ALL8_1(OP) A1(OP, r0) A1(OP, r1) A1(OP, r2) A1(OP, r3) A1(OP, r4) A1(OP, r5) A1(OP, r6) A1(OP, r7)
#define A2(OP, R) asm volatile(OP : "+v"(R) : "v"(s));
#define ALL8_2(OP) A2(OP, r0) A2(OP, r1) A2(OP, r2) A2(OP, r3) A2(OP, r4) A2(OP, r5) A2(OP, r6) A2(OP, r7)

KERNEL(k_mul, ALL8_1("v_mul_f32 %0, %0, %1"), 8)
KERNEL(k_add, ALL8_1("v_add_f32 %0, %0, %1"), 8)
KERNEL(k_fma, ALL8_1("v_fma_f32 %0, %0, %1, %1"), 8)
KERNEL(k_sub_abs, ALL8_1("v_sub_f32_e64 %0, 1.0, |%0|"), 8)
KERNEL(k_cmp, ALL8_1("v_cmp_lt_f32_e32 vcc, %0, %1"), 8)
KERNEL(k_cndmask, ALL8_1("v_cndmask_b32_e32 %0, %0, %1, vcc"), 8)
KERNEL(k_mov, ALL8_1("v_mov_b32 %0, %1"), 8)
KERNEL(k_pk_mul, ALL8_2("v_pk_mul_f32 %0, %0, %1"), 8)
KERNEL(k_pk_add, ALL8_2("v_pk_add_f32 %0, %0, %1"), 8)
KERNEL(k_pk_fma, ALL8_2("v_pk_fma_f32 %0, %0, %1, %1"), 8)
KERNEL(k_pk_mov, ALL8_2("v_pk_mov_b32 %0, %0, %1"), 8)
KERNEL(k_mov_b64, ALL8_2("v_mov_b64 %0, %1"), 8)
KERNEL(k_rcp, ALL8_1("v_rcp_f32 %0, %0"), 8)
KERNEL(k_sqrt, ALL8_1("v_sqrt_f32 %0, %0"), 8)
KERNEL(k_min, ALL8_1("v_min_f32 %0, %0, %1"), 8)
KERNEL(k_min3, ALL8_1("v_min3_f32 %0, %0, %1, %1"), 8)

// update sequences: (a) cmp -> sgpr mask + 2 cndmask, (b) cmpx + pk_mov + exec restore, (c) cmp vcc + 2 cndmask e32
#define SEQ_A(R) asm volatile("v_cmp_lt_f32_e64 s[40:41], %1, %0\n v_cndmask_b32_e64 %0, %0, %1, s[40:41]\n v_cndmask_b32_e64 %2, %2, %1, s[40:41]" : "+v"(R.x), "+v"(s.x), "+v"(R.y) : : "s40", "s41");
#define SEQ_B(R) asm volatile("v_cmpx_lt_f32_e32 vcc, %1, %0\n v_pk_mov_b32 %2, %3, %3 op_sel:[0,0]\n s_mov_b64 exec, -1" : "+v"(R.x), "+v"(s.x), "+v"(R) : "v"(s) : "vcc");
#define SEQ_C(R) asm volatile("v_cmp_lt_f32_e32 vcc, %1, %0\n v_cndmask_b32_e32 %0, %0, %1, vcc\n v_cndmask_b32_e32 %2, %2, %1, vcc" : "+v"(R.x), "+v"(s.x), "+v"(R.y) : : "vcc");
#define ALL8(S) S(r0) S(r1) S(r2) S(r3) S(r4) S(r5) S(r6) S(r7)
KERNEL(k_seq_a, ALL8(SEQ_A), 8)
KERNEL(k_seq_b, ALL8(SEQ_B), 8)
KERNEL(k_seq_c, ALL8(SEQ_C), 8)

template <typename K>
double run(K kern, const char *name, float *d_out) {
    const int blocks = 256 * 8;   // 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 1.0f, 0.999f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 1.0f, 0.999f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD = waves_per_simd * ITER * REP ;  blocks*4 waves over 1024 SIMDs
    const double per_simd = (double)blocks * 4 / 1024.0 * ITER * REP;
    const double ns = ms * 1e6 / per_simd;
    printf("%-12s %8.3f ms  %6.3f ns / wave-instr / SIMD  (= %.2f cycles @2.4GHz, %.2f @2.1GHz)\n", name, ms, ns,
           ns * 2.4, ns * 2.1);
    return ns;
}

int main() {
    float *d;
    hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    run(k_mul, "v_mul", d);
    run(k_add, "v_add", d);
    run(k_fma, "v_fma", d);
    run(k_sub_abs, "v_sub|abs|", d);
    run(k_cmp, "v_cmp", d);
    run(k_cndmask, "v_cndmask", d);
    run(k_mov, "v_mov", d);
    run(k_min, "v_min", d);
    run(k_min3, "v_min3", d);
    run(k_pk_mul, "v_pk_mul", d);
    run(k_pk_add, "v_pk_add", d);
    run(k_pk_fma, "v_pk_fma", d);
    run(k_pk_mov, "v_pk_mov", d);
    run(k_mov_b64, "v_mov_b64", d);
    printf("--- update sequences (3 instructions each; cycles are per INSTRUCTION, multiply by 3 per update)\n");
    run(k_seq_a, "cmp+2cnd e64", d);
    run(k_seq_b, "cmpx+pkmov", d);
    run(k_seq_c, "cmp+2cnd vcc", d);
    run(k_rcp, "v_rcp", d);
    run(k_sqrt, "v_sqrt", d);
    return 0;
}
