// Micro-benchmark 6 (gfx950, round 4): the two products of a tap evaluation on the matrix pipe.
//   v_mfma_f32_4x4x1_16b_f32 with C = 0: lane l gives A = tx of tap (l % 4) and B = dx of ITS item; D[r] of lane l is
//   RNE(tx[tap r] * dx[item of lane l]) -- each lane gets the four products of its own item, no cross-lane step.
// (1) numerics: D[r] == v_mul_f32 of the same operands bit for bit (random, signed zeros, subnormal products);
// (2) issue cost of the tap body  2 MFMA per (4 taps, item)  +  add, sub, key per evaluation + min3 per two  against the
//     all-VALU bodies of valu5.hip.
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench/bin/valu6 tools/ubench/valu6.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

typedef float v4 __attribute__((ext_vector_type(4)));
#define ITER 1000

__global__ void k_numerics(const float *a, const float *b, float *d) {
    const int l = threadIdx.x, g = blockIdx.x * 64;
    const v4 z = {0.f, 0.f, 0.f, 0.f};
    const v4 r = __builtin_amdgcn_mfma_f32_4x4x1f32(a[g + l], b[g + l], z, 0, 0, 0);
    for (int i = 0; i < 4; ++i) d[(g + l) * 4 + i] = r[i];
}

#define SUBC(D, X) asm volatile("v_sub_f32_e64 %0, %2, |%1|" : "=v"(D) : "v"(X), "v"(cc))
#define ADD(D, A, B) asm volatile("v_add_f32_e32 %0, %1, %2" : "=v"(D) : "v"(A), "v"(B))
#define KEYI(D, X, U) asm volatile("v_lshl_or_b32 %0, %1, 5, " #U : "=v"(D) : "v"(X))
#define MIN3(A, K0, K1) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(A) : "v"(K0), "v"(K1))

// four taps (rows U0..U0+3) of one item: products P, Q from the matrix pipe -> sum, C - |sum|, key, min3
#define EVAL4(ACC, P, Q, U0, U1, U2, U3) { float s0, s1, s2, s3;                                             \
      ADD(s0, P[0], Q[0]); ADD(s1, P[1], Q[1]); ADD(s2, P[2], Q[2]); ADD(s3, P[3], Q[3]);                 \
      SUBC(s0, s0); SUBC(s1, s1); SUBC(s2, s2); SUBC(s3, s3);                                             \
      KEYI(s0, s0, U0); KEYI(s1, s1, U1); KEYI(s2, s2, U2); KEYI(s3, s3, U3);                             \
      MIN3(ACC, s0, s1); MIN3(ACC, s2, s3); }

__global__ __launch_bounds__(256) void k_m1(float *out, float a, float b) {
    float dx0 = a + threadIdx.x, dx1 = b, dx2 = a * 2, dx3 = b * 3, dy0 = b, dy1 = a, dy2 = b + 1, dy3 = a + 2;
    float tx = a * 0.5f + (threadIdx.x & 3), ty = b * 0.25f, tx2 = tx + 1.0f, ty2 = ty + 1.0f;
    const float cc = a + 6.1035156e-05f;
    unsigned k0 = ~0u, k1 = ~0u, k2 = ~0u, k3 = ~0u;
    const v4 z = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {   // 2 x (8 taps x 4 items)
            const v4 p0 = __builtin_amdgcn_mfma_f32_4x4x1f32(tx, dx0, z, 0, 0, 0);
            const v4 q0 = __builtin_amdgcn_mfma_f32_4x4x1f32(ty, dy0, z, 0, 0, 0);
            const v4 p1 = __builtin_amdgcn_mfma_f32_4x4x1f32(tx, dx1, z, 0, 0, 0);
            const v4 q1 = __builtin_amdgcn_mfma_f32_4x4x1f32(ty, dy1, z, 0, 0, 0);
            const v4 p2 = __builtin_amdgcn_mfma_f32_4x4x1f32(tx, dx2, z, 0, 0, 0);
            const v4 q2 = __builtin_amdgcn_mfma_f32_4x4x1f32(ty, dy2, z, 0, 0, 0);
            const v4 p3 = __builtin_amdgcn_mfma_f32_4x4x1f32(tx, dx3, z, 0, 0, 0);
            const v4 q3 = __builtin_amdgcn_mfma_f32_4x4x1f32(ty, dy3, z, 0, 0, 0);
            EVAL4(k0, p0, q0, 0, 1, 2, 3)
            EVAL4(k1, p1, q1, 0, 1, 2, 3)
            const v4 P0 = __builtin_amdgcn_mfma_f32_4x4x1f32(tx2, dx0, z, 0, 0, 0);
            const v4 Q0 = __builtin_amdgcn_mfma_f32_4x4x1f32(ty2, dy0, z, 0, 0, 0);
            const v4 P1 = __builtin_amdgcn_mfma_f32_4x4x1f32(tx2, dx1, z, 0, 0, 0);
            const v4 Q1 = __builtin_amdgcn_mfma_f32_4x4x1f32(ty2, dy1, z, 0, 0, 0);
            EVAL4(k2, p2, q2, 0, 1, 2, 3)
            EVAL4(k3, p3, q3, 0, 1, 2, 3)
            const v4 P2 = __builtin_amdgcn_mfma_f32_4x4x1f32(tx2, dx2, z, 0, 0, 0);
            const v4 Q2 = __builtin_amdgcn_mfma_f32_4x4x1f32(ty2, dy2, z, 0, 0, 0);
            const v4 P3 = __builtin_amdgcn_mfma_f32_4x4x1f32(tx2, dx3, z, 0, 0, 0);
            const v4 Q3 = __builtin_amdgcn_mfma_f32_4x4x1f32(ty2, dy3, z, 0, 0, 0);
            EVAL4(k0, P0, Q0, 4, 5, 6, 7)
            EVAL4(k1, P1, Q1, 4, 5, 6, 7)
            EVAL4(k2, P2, Q2, 4, 5, 6, 7)
            EVAL4(k3, P3, Q3, 4, 5, 6, 7)
            // keep the operands changing so that nothing is hoisted
            asm volatile("v_xor_b32 %0, %0, %1" : "+v"(tx) : "v"(k0));
            asm volatile("v_xor_b32 %0, %0, %1" : "+v"(ty2) : "v"(k3));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)(k0 ^ k1 ^ k2 ^ k3) + tx + ty2;
}

// the same evaluation count on the VALU only (N1 of valu5.hip) for a same-run comparison
#define MUL(D, A, B) asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(D) : "v"(A), "v"(B))
#define KEYS(D, X, U) asm volatile("v_lshl_or_b32 %0, %1, 5, %2" : "=v"(D) : "v"(X), "s"(U))
__global__ __launch_bounds__(256) void k_n1(float *out, float a, float b) {
    float dx[4] = {a + threadIdx.x, b, a * 2, b * 3}, dy[4] = {b, a, b + 1, a + 2};
    float tx = a * 0.5f, ty = b * 0.25f, tx2 = tx + 1.0f, ty2 = ty + 1.0f;
    const float cc = a + 6.1035156e-05f;
    unsigned k[4] = {~0u, ~0u, ~0u, ~0u};
    const unsigned iu = blockIdx.x & 31u, iv = iu ^ 1u;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int h = 0; h < 8; ++h) {   // 8 x (2 taps x 4 items)
            float l[4], g[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float p, q;
                MUL(p, tx, dx[j]); MUL(q, ty, dy[j]); ADD(p, p, q); SUBC(l[j], p);
                MUL(p, tx2, dx[j]); MUL(q, ty2, dy[j]); ADD(p, p, q); SUBC(g[j], p);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) { KEYS(l[j], l[j], iu); KEYS(g[j], g[j], iv); }
#pragma unroll
            for (int j = 0; j < 4; ++j) MIN3(k[j], l[j], g[j]);
        }
        asm volatile("v_xor_b32 %0, %0, %1" : "+v"(tx) : "v"(k[0]));
        asm volatile("v_xor_b32 %0, %0, %1" : "+v"(ty2) : "v"(k[3]));
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)(k[0] ^ k[1] ^ k[2] ^ k[3]) + tx + ty2;
}

template <typename K>
void run(K kern, const char *name, float *d_out, int waves_per_simd, int inst_per_tap) {
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 1.0f, 0.999f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 1.0f, 0.999f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double taps = (double)waves_per_simd * ITER * 16;   // 16 taps (x 4 items) per iteration per wave
    const double ns = ms * 1e6 / taps;
    printf("%-30s w/simd=%d  %8.3f ms  %6.2f ns = %5.1f cycles@2.4GHz per tap per 4 items  (%d instructions: %.2f cycles each)\n",
           name, waves_per_simd, ms, ns, ns * 2.4, inst_per_tap, ns * 2.4 / inst_per_tap);
}

int main() {
    // ---- numerics
    const int NB = 4096, N = NB * 64;
    std::vector<float> a(N), b(N), d(N * 4);
    srand(1);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (int i = 0; i < N; ++i) {
        a[i] = rnd();
        b[i] = rnd();
        const int m = i % 97;
        if (m == 0) a[i] = 0.0f;
        if (m == 1) a[i] = -0.0f;
        if (m == 2) { a[i] = 1e-20f * rnd(); b[i] = 1e-20f * rnd(); }     // subnormal / underflowing products
        if (m == 3) { a[i] = 1e-19f * rnd(); b[i] = 3e-20f * rnd(); }
        if (m == 4) a[i] = 1e-40f;                                        // subnormal input
        if (m == 5) b[i] = -1e-41f;
        if (m == 6) a[i] = NAN;
        if (m == 7) { a[i] = 1.0f; b[i] = 1.0000001f; }
    }
    float *da, *db, *dd;
    (void)hipMalloc(&da, N * 4);
    (void)hipMalloc(&db, N * 4);
    (void)hipMalloc(&dd, N * 16);
    (void)hipMemcpy(da, a.data(), N * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, b.data(), N * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_numerics, dim3(NB), dim3(64), 0, 0, da, db, dd);
    (void)hipMemcpy(d.data(), dd, N * 16, hipMemcpyDeviceToHost);
    long bad = 0, badzero = 0, n = 0;
    for (int i = 0; i < N; ++i) {
        const int l = i % 64, g = i - l;
        for (int r = 0; r < 4; ++r) {
            const float A = a[g + 4 * (l / 4) + r], B = b[i];
            volatile float want = A * B;   // host RNE product (x86-64 SSE, subnormals kept)
            const float got = d[i * 4 + r];
            float w = want;
            ++n;
            if (std::isnan(w) && std::isnan(got)) continue;
            if (memcmp(&w, &got, 4) != 0) {
                if (w == 0.0f && got == 0.0f) ++badzero;      // only the sign of a zero differs
                else {
                    if (bad < 8) printf("  mismatch lane %d row %d: %a * %a = %a, mfma gave %a\n", l, r, A, B, w, got);
                    ++bad;
                }
            }
        }
    }
    printf("numerics: %ld products, %ld mismatches, %ld zero-sign differences (fma(a,b,+0) turns -0 into +0)\n", n, bad, badzero);

    // ---- issue cost
    float *o;
    (void)hipMalloc(&o, 256 * 8 * 256 * sizeof(float));
    for (int w : {8, 5, 4, 2, 1}) {
        run(k_n1, "N1 all VALU key+min3", o, w, 22);
        run(k_m1, "M1 mfma 4x4x1 products + key", o, w, 16);
    }
    return 0;
}
