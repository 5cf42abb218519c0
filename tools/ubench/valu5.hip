// Micro-benchmark 5 (gfx950, round 4): candidate tap bodies of mh_search3_kernel that keep the running minimum as ONE
// integer key per item -- (bits(C - |cs|) << 5) | tap index, v_min3_u32 over two taps at a time -- against the shipped
// compare + two selects (U1).  Prints cycles per tap per 4 items at an assumed 2.4 GHz, like valu3.hip.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu5 tools/ubench/valu5.hip && /tmp/valu5
#include <hip/hip_runtime.h>
#include <cstdio>

#define ITER 1000
typedef float v2 __attribute__((ext_vector_type(2)));

#define PRO                                                                                                   \
    v2 r0 = {a + threadIdx.x, b}, r1 = {b, a}, r2 = {a, a}, r3 = {b, b}, r4 = {a, b + 1}, r5 = {a + 2, b},   \
       r6 = {a, b + 3}, r7 = {a + 4, b};                                                                      \
    v2 s = {a, b}, t = {b, a};                                                                                \
    const float cc = a + 6.1035156e-05f;                                                                      \
    const unsigned iu = (unsigned)blockIdx.x & 31u, iv = iu ^ 1u;                                             \
    unsigned long long c0 = __builtin_readcyclecounter();
#define EPI                                                                                                   \
    unsigned long long c1 = __builtin_readcyclecounter();                                                     \
    out[blockIdx.x * 256 + threadIdx.x] =                                                                     \
        r0.x + r1.x + r2.x + r3.x + r4.x + r5.x + r6.x + r7.x + r0.y + r1.y + r2.y + r3.y + r4.y + r5.y + r6.y + r7.y + s.x + t.x; \
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = c1 - c0;

#define SUBABS(D, X) asm volatile("v_sub_f32_e64 %0, 1.0, |%1|" : "=v"(D) : "v"(X))
#define SUBC(D, X) asm volatile("v_sub_f32_e64 %0, %2, |%1|" : "=v"(D) : "v"(X), "s"(cc))
#define ADD(D, A, B) asm volatile("v_add_f32_e32 %0, %1, %2" : "=v"(D) : "v"(A), "v"(B))
#define MUL(D, A, B) asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(D) : "v"(A), "v"(B))
#define PKMUL(D, A, B) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(D) : "v"(A), "v"(B))
#define CMP(M, A, B) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(M) : "v"(A), "v"(B))
#define CND(D, X, M) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(D) : "v"(X), "s"(M))
#define KEY(D, X, U) asm volatile("v_lshl_or_b32 %0, %1, 5, %2" : "=v"(D) : "v"(X), "s"(U))
#define KEYV(D, X, U) asm volatile("v_lshl_or_b32 %0, %1, 5, %2" : "=v"(D) : "v"(X), "v"(U))
#define MIN3(A, K0, K1) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(A) : "v"(K0), "v"(K1))
#define MIN3F(A, K0, K1) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(A) : "v"(K0), "v"(K1))

// arithmetic of one tap (TP = {tx, ty}) for the four items r0..r3 = {dx, dy}: separately rounded products, sum
#define ARITH_S(TP, l0, l1, l2, l3, SUB) { float a0, a1, a2, a3, b0, b1, b2, b3;                                 \
      MUL(a0, TP.x, r0.x); MUL(a1, TP.x, r1.x); MUL(a2, TP.x, r2.x); MUL(a3, TP.x, r3.x);                     \
      MUL(b0, TP.y, r0.y); MUL(b1, TP.y, r1.y); MUL(b2, TP.y, r2.y); MUL(b3, TP.y, r3.y);                     \
      ADD(a0, a0, b0); ADD(a1, a1, b1); ADD(a2, a2, b2); ADD(a3, a3, b3);                                     \
      SUB(l0, a0); SUB(l1, a1); SUB(l2, a2); SUB(l3, a3); }
#define ARITH_P(TP, l0, l1, l2, l3, SUB) { v2 p0, p1, p2, p3; float a0, a1, a2, a3;                               \
      PKMUL(p0, TP, r0); PKMUL(p1, TP, r1); PKMUL(p2, TP, r2); PKMUL(p3, TP, r3);                             \
      ADD(a0, p0.x, p0.y); ADD(a1, p1.x, p1.y); ADD(a2, p2.x, p2.y); ADD(a3, p3.x, p3.y);                     \
      SUB(l0, a0); SUB(l1, a1); SUB(l2, a2); SUB(l3, a3); }
// half of the products packed
#define ARITH_H(TP, l0, l1, l2, l3, SUB) { v2 p0, p1; float a0, a1, a2, a3, b2, b3;                               \
      PKMUL(p0, TP, r0); PKMUL(p1, TP, r1);                                                                   \
      MUL(a2, TP.x, r2.x); MUL(a3, TP.x, r3.x); MUL(b2, TP.y, r2.y); MUL(b3, TP.y, r3.y);                     \
      ADD(a0, p0.x, p0.y); ADD(a1, p1.x, p1.y); ADD(a2, a2, b2); ADD(a3, a3, b3);                             \
      SUB(l0, a0); SUB(l1, a1); SUB(l2, a2); SUB(l3, a3); }

// U1: the shipped update, one tap
#define BODY_U1 { float l0, l1, l2, l3; ARITH_S(s, l0, l1, l2, l3, SUBABS) unsigned long long m0, m1, m2, m3;  \
      CMP(m0, l0, r4.x); CMP(m1, l1, r4.y); CMP(m2, l2, r5.x); CMP(m3, l3, r5.y);                             \
      CND(r4.x, l0, m0); CND(r4.y, l1, m1); CND(r5.x, l2, m2); CND(r5.y, l3, m3);                             \
      CND(r6.x, t.x, m0); CND(r6.y, t.x, m1); CND(r7.x, t.x, m2); CND(r7.y, t.x, m3); }
#define BODY_U1x2 BODY_U1 BODY_U1

// key bodies, TWO taps (s and t) per body
#define BODY_KEY(AR, KEYOP, U0, U1) { float l0, l1, l2, l3, g0, g1, g2, g3;                                      \
      AR(s, l0, l1, l2, l3, SUBC) AR(t, g0, g1, g2, g3, SUBC)                                                 \
      KEYOP(l0, l0, U0); KEYOP(l1, l1, U0); KEYOP(l2, l2, U0); KEYOP(l3, l3, U0);                             \
      KEYOP(g0, g0, U1); KEYOP(g1, g1, U1); KEYOP(g2, g2, U1); KEYOP(g3, g3, U1);                             \
      MIN3(r4.x, l0, g0); MIN3(r4.y, l1, g1); MIN3(r5.x, l2, g2); MIN3(r5.y, l3, g3); }
#define BODY_N1 BODY_KEY(ARITH_S, KEY, iu, iv)
#define BODY_N2 BODY_KEY(ARITH_P, KEY, iu, iv)
#define BODY_N3 BODY_KEY(ARITH_H, KEY, iu, iv)
#define BODY_N2V BODY_KEY(ARITH_P, KEYV, r6.x, r6.y)
// the floor: float minimum only (no index), two taps
#define BODY_F(AR) { float l0, l1, l2, l3, g0, g1, g2, g3;                                                       \
      AR(s, l0, l1, l2, l3, SUBC) AR(t, g0, g1, g2, g3, SUBC)                                                 \
      MIN3F(r4.x, l0, g0); MIN3F(r4.y, l1, g1); MIN3F(r5.x, l2, g2); MIN3F(r5.y, l3, g3); }
#define BODY_F1 BODY_F(ARITH_S)
#define BODY_F2 BODY_F(ARITH_P)

#define BKERNEL(NAME, BODY)                                                                 \
    __global__ __launch_bounds__(256) void NAME(float *out, unsigned long long *cyc, float a, float b) { \
        PRO for (int i = 0; i < ITER; ++i) { BODY BODY BODY BODY } EPI }
BKERNEL(k_u1, BODY_U1x2)
BKERNEL(k_n1, BODY_N1)
BKERNEL(k_n2, BODY_N2)
BKERNEL(k_n3, BODY_N3)
BKERNEL(k_n2v, BODY_N2V)
BKERNEL(k_f1, BODY_F1)
BKERNEL(k_f2, BODY_F2)

template <typename K>
void run(K kern, const char *name, float *d_out, unsigned long long *d_cyc, int waves_per_simd, int inst_per_tap) {
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
    const double taps = (double)waves_per_simd * ITER * 8;   // taps per SIMD (4 bodies x 2 taps per iteration)
    const double ns = ms * 1e6 / taps;
    printf("%-26s w/simd=%d  %8.3f ms  %6.2f ns = %5.1f cycles@2.4GHz per tap per 4 items  (%d instructions: %.2f cycles each)\n",
           name, waves_per_simd, ms, ns, ns * 2.4, inst_per_tap, ns * 2.4 / inst_per_tap);
}

int main() {
    float *d;
    unsigned long long *c;
    hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    hipMalloc(&c, 64);
    for (int w : {8, 5, 4, 2}) {
        run(k_u1, "U1 cmp+2cnd (shipped)", d, c, w, 28);
        run(k_n1, "N1 scalar+key+min3", d, c, w, 22);
        run(k_n2, "N2 pk_mul+key+min3", d, c, w, 18);
        run(k_n3, "N3 half pk+key+min3", d, c, w, 20);
        run(k_n2v, "N2V pk_mul+key(vgpr)+min3", d, c, w, 18);
        run(k_f1, "F1 scalar+fmin3 (floor)", d, c, w, 18);
        run(k_f2, "F2 pk_mul+fmin3 (floor)", d, c, w, 14);
    }
    return 0;
}
