// Micro-benchmark 7 (gfx950, round 4): as valu6.hip with v_mfma_f32_16x16x1_4b_f32 (1024 products per instruction, 8 passes):
// does a LONG matrix instruction overlap with the VALU work of the other waves of the SIMD?  Per MFMA pair a lane gets 16
// products pairs = 16 evaluations: add, sub, key each + 8 min3 = 56 VALU instructions (the all-VALU key body: 88).
//   hipcc --offload-arch=gfx950 -O2 -mllvm -amdgpu-mfma-vgpr-form -o tools/ubench/bin/valu7 tools/ubench/valu7.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v16 __attribute__((ext_vector_type(16)));
#define ITER 500

#define SUBC(D, X) asm volatile("v_sub_f32_e64 %0, %2, |%1|" : "=v"(D) : "v"(X), "v"(cc))
#define ADD(D, A, B) asm volatile("v_add_f32_e32 %0, %1, %2" : "=v"(D) : "v"(A), "v"(B))
#define MUL(D, A, B) asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(D) : "v"(A), "v"(B))
#define KEYI(D, X, U) asm volatile("v_lshl_or_b32 %0, %1, 5, " #U : "=v"(D) : "v"(X))
#define KEYS(D, X, U) asm volatile("v_lshl_or_b32 %0, %1, 5, %2" : "=v"(D) : "v"(X), "s"(U))
#define MIN3(A, K0, K1) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(A) : "v"(K0), "v"(K1))

#define EVAL4(ACC, P, Q, O, U0, U1, U2, U3) { float s0, s1, s2, s3;                                        \
      ADD(s0, P[O], Q[O]); ADD(s1, P[O + 1], Q[O + 1]); ADD(s2, P[O + 2], Q[O + 2]); ADD(s3, P[O + 3], Q[O + 3]); \
      SUBC(s0, s0); SUBC(s1, s1); SUBC(s2, s2); SUBC(s3, s3);                                             \
      KEYI(s0, s0, U0); KEYI(s1, s1, U1); KEYI(s2, s2, U2); KEYI(s3, s3, U3);                             \
      MIN3(ACC, s0, s1); MIN3(ACC, s2, s3); }

__global__ __launch_bounds__(256) void k_m2(float *out, float a, float b) {
    float dx0 = a + threadIdx.x, dy0 = b, dx1 = b * 2, dy1 = a * 3;
    float tx = a * 0.5f + (threadIdx.x & 15), ty = b * 0.25f;
    const float cc = a + 6.1035156e-05f;
    unsigned k0 = ~0u, k1 = ~0u, k2 = ~0u, k3 = ~0u;
    v16 z;
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {   // 2 x (two MFMA pairs = 32 evaluations per lane)
            const v16 p = __builtin_amdgcn_mfma_f32_16x16x1f32(tx, dx0, z, 0, 0, 0);
            const v16 q = __builtin_amdgcn_mfma_f32_16x16x1f32(ty, dy0, z, 0, 0, 0);
            const v16 p2 = __builtin_amdgcn_mfma_f32_16x16x1f32(tx, dx1, z, 0, 0, 0);
            const v16 q2 = __builtin_amdgcn_mfma_f32_16x16x1f32(ty, dy1, z, 0, 0, 0);
            EVAL4(k0, p, q, 0, 0, 1, 2, 3)
            EVAL4(k1, p, q, 4, 0, 1, 2, 3)
            EVAL4(k2, p, q, 8, 0, 1, 2, 3)
            EVAL4(k3, p, q, 12, 0, 1, 2, 3)
            EVAL4(k0, p2, q2, 0, 4, 5, 6, 7)
            EVAL4(k1, p2, q2, 4, 4, 5, 6, 7)
            EVAL4(k2, p2, q2, 8, 4, 5, 6, 7)
            EVAL4(k3, p2, q2, 12, 4, 5, 6, 7)
            asm volatile("v_xor_b32 %0, %0, %1" : "+v"(tx) : "v"(k0));
            asm volatile("v_xor_b32 %0, %0, %1" : "+v"(ty) : "v"(k3));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)(k0 ^ k1 ^ k2 ^ k3) + tx + ty;
}

// the same 64 evaluations per lane per iteration on the VALU only
__global__ __launch_bounds__(256) void k_n1(float *out, float a, float b) {
    float dx[4] = {a + threadIdx.x, b, a * 2, b * 3}, dy[4] = {b, a, b + 1, a + 2};
    float tx = a * 0.5f, ty = b * 0.25f, tx2 = tx + 1.0f, ty2 = ty + 1.0f;
    const float cc = a + 6.1035156e-05f;
    unsigned k[4] = {~0u, ~0u, ~0u, ~0u};
    const unsigned iu = blockIdx.x & 31u, iv = iu ^ 1u;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int h = 0; h < 8; ++h) {   // 8 x (2 taps x 4 items) = 64 evaluations
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
void run(K kern, const char *name, float *d_out, int waves_per_simd) {
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
    const double ev = (double)waves_per_simd * ITER * 64;   // evaluations per lane per SIMD
    const double ns = ms * 1e6 / ev;
    printf("%-34s w/simd=%d  %8.3f ms  %6.3f ns = %5.2f cycles@2.4GHz per evaluation (x64 lanes)\n", name, waves_per_simd, ms, ns, ns * 2.4);
}

int main() {
    float *o;
    (void)hipMalloc(&o, 256 * 8 * 256 * sizeof(float));
    for (int w : {8, 5, 4, 2, 1}) {
        run(k_n1, "N1 all VALU (5.5 per evaluation)", o, w);
        run(k_m2, "M2 mfma 16x16x1 products (3.5+)", o, w);
    }
    return 0;
}
