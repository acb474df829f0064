// fp32 MFMA peak probe: pure v_mfma_f32_32x32x2_f32 stream, 4 accumulators per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    v16 acc[4];
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
    float s = 0;
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) s += acc[t][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* d; hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wpb : {1, 2, 4}) {
        int blocks = 256 * wpb * 4, iters = 2000;
        k<<<blocks, 256>>>(d, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<<<blocks, 256>>>(d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double fl = (double)blocks * 4 * iters * 64 * 4096.0;
        printf("blocks/CU %d: %.2f ms  %.1f TF/s\n", wpb, ms, fl / ms / 1e9);
    }
    return 0;
}
