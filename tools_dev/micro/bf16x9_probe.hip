// fp32 GEMM tile through three bf16 planes per operand and 9 v_mfma_f32_32x32x16_bf16 per k-step:
// numerics against a float64 reference next to the native fp32 MFMA, and the issue rate of both.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
#define K 256

__device__ inline void split3(float x, unsigned short& p1, unsigned short& p2, unsigned short& p3) {
    const unsigned u1 = __float_as_uint(x) & 0xFFFF0000u;
    const float r1 = x - __uint_as_float(u1);
    const unsigned u2 = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(u2);
    p1 = u1 >> 16; p2 = u2 >> 16; p3 = __float_as_uint(r2) >> 16;
}

// one wave: C[32x32] = A[32xK] * B[Kx32]
__global__ void gemm_f32(const float* A, const float* B, float* C) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    v16f acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int s = 0; s < K / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + 2 * s + h], B[(2 * s + h) * 32 + r], acc, 0, 0, 0);
    for (int v = 0; v < 16; ++v) C[((v >> 2) * 8 + h * 4 + (v & 3)) * 32 + r] = acc[v];
}
__global__ void gemm_x9(const float* A, const float* B, float* C) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    v16f acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int s = 0; s < K / 16; ++s) {
        union { v8bf v; unsigned short u[8]; } a[3], b[3];
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * s + 8 * h + j;
            split3(A[r * K + k], a[0].u[j], a[1].u[j], a[2].u[j]);
            split3(B[k * 32 + r], b[0].u[j], b[1].u[j], b[2].u[j]);
        }
        for (int t = 4; t >= 0; --t)                       // smallest partial products first
            for (int i = 0; i < 3; ++i) { const int j = t - i; if (j >= 0 && j < 3) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].v, b[j].v, acc, 0, 0, 0); }
    }
    for (int v = 0; v < 16; ++v) C[((v >> 2) * 8 + h * 4 + (v & 3)) * 32 + r] = acc[v];
}
__global__ __launch_bounds__(256) void rate_f32(float* out, int iters) {
    v16f acc[4]; for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    float s = 0; for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) s += acc[t][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void rate_x9(float* out, int iters) {
    v16f acc[4]; for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    v8bf a, b; for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 1e-3f); b[j] = (__bf16)(blockIdx.x * 1e-3f); }
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 9; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
    float s = 0; for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) s += acc[t][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    std::vector<float> A(32 * K), B(K * 32), C1(1024), C2(1024);
    srand(1);
    for (auto& x : A) x = (float)rand() / RAND_MAX * 2 - 1;
    for (auto& x : B) x = ((float)rand() / RAND_MAX * 2 - 1) * 0.1f;
    A[5] = 1e-30f; A[7] = 3e8f; B[9 * 32 + 3] = -7e-12f;
    float *dA, *dB, *dC; (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&dC, 4096 * 256 * 4);
    (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    gemm_f32<<<1, 64>>>(dA, dB, dC); (void)hipMemcpy(C1.data(), dC, 4096, hipMemcpyDeviceToHost);
    gemm_x9<<<1, 64>>>(dA, dB, dC); (void)hipMemcpy(C2.data(), dC, 4096, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0, e12 = 0, nrm = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        double ref = 0, mag = 0; for (int k = 0; k < K; ++k) { ref += (double)A[i * K + k] * B[k * 32 + j]; mag += fabs((double)A[i * K + k] * B[k * 32 + j]); }
        e1 = fmax(e1, fabs(C1[i * 32 + j] - ref) / mag); e2 = fmax(e2, fabs(C2[i * 32 + j] - ref) / mag);
        e12 = fmax(e12, fabs((double)C1[i * 32 + j] - C2[i * 32 + j]) / mag); nrm = fmax(nrm, mag);
    }
    printf("K=%d: max |err| / sum|a*b|: native fp32 MFMA %.3e, bf16x9 %.3e, between them %.3e (2^-24 = %.3e)\n", K, e1, e2, e12, ldexp(1.0, -24));
    hipEvent_t e0, e1v; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1v);
    const int blocks = 2048, iters = 2000;
    rate_f32<<<blocks, 256>>>(dC, 10); rate_x9<<<blocks, 256>>>(dC, 10); (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventRecord(e0); rate_f32<<<blocks, 256>>>(dC, iters); (void)hipEventRecord(e1v); (void)hipEventSynchronize(e1v); (void)hipEventElapsedTime(&ms, e0, e1v);
    double fl = (double)blocks * 4 * iters * 4 * 32.0 * 32 * 16 * 2;      // fp32-equivalent FLOP: K = 16 per outer iteration and accumulator
    printf("fp32 MFMA 32x32x2  : %.2f ms, %.1f TFLOP/s\n", ms, fl / ms / 1e9);
    (void)hipEventRecord(e0); rate_x9<<<blocks, 256>>>(dC, iters); (void)hipEventRecord(e1v); (void)hipEventSynchronize(e1v); (void)hipEventElapsedTime(&ms, e0, e1v);
    printf("9 x bf16 32x32x16  : %.2f ms, %.1f fp32-equivalent TFLOP/s\n", ms, fl / ms / 1e9);
    return 0;
}
