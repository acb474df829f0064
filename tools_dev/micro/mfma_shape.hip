// What limits a bf16 32x32x16 MFMA stream in the shape of the dense conv kernels?
//   mode 0: pure MFMA, NACC accumulators per wave, fragments constant in registers
//   mode 1: + LDS fragment reads per stage (NA A-fragments + NB B-fragments of 16 B / lane, 48-byte rows), no barrier
//   mode 2: + one __syncthreads per stage
// per stage: 6 * MT * NT MFMAs (six partial products), 3*MT + 3*NT ds_read_b128.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16 __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

template <int MT, int NT, int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void k(float* out, int stages) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[48 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 48 * 1024 / 4; i += THREADS) reinterpret_cast<float*>(lds)[i] = 1e-3f * (i & 255);
    __syncthreads();
    v16 acc[MT][NT];
    for (int m = 0; m < MT; ++m) for (int t = 0; t < NT; ++t) for (int i = 0; i < 16; ++i) acc[m][t][i] = 0.f;
    v8bf fa[MT][3], fb[NT][3];
    for (int m = 0; m < MT; ++m) for (int p = 0; p < 3; ++p) for (int i = 0; i < 8; ++i) fa[m][p][i] = (__bf16)(0.001f * (lane + p));
    for (int t = 0; t < NT; ++t) for (int p = 0; p < 3; ++p) for (int i = 0; i < 8; ++i) fb[t][p][i] = (__bf16)(0.002f * (lane + p));
    const int r = lane & 31, h = lane >> 5;
    for (int s = 0; s < stages; ++s) {
        if (MODE >= 1) {
            const unsigned char* Ap = lds + ((wave * 2 + (s % 3)) * 34 + r + (s & 1)) * 48 + h * 16;
            const unsigned char* Bp = lds + 16384 + (s % 3) * 9216 + r * 48 + h * 16;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int p = 0; p < 3; ++p) fa[m][p] = *reinterpret_cast<const v8bf*>(Ap + p * 5000 * 0 + p * 4096 + m * 34 * 48);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int p = 0; p < 3; ++p) fb[t][p] = *reinterpret_cast<const v8bf*>(Bp + p * 3072 + (t & 1) * 32 * 48);
        }
#define MM(PA, PB) _Pragma("unroll") for (int m = 0; m < MT; ++m) _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m][PA], fb[t][PB], acc[m][t], 0, 0, 0);
        MM(0, 2) MM(1, 1) MM(2, 0) MM(0, 1) MM(1, 0) MM(0, 0)
#undef MM
        if (MODE >= 2) __syncthreads();
    }
    float sum = 0;
    for (int m = 0; m < MT; ++m) for (int t = 0; t < NT; ++t) for (int i = 0; i < 16; ++i) sum += acc[m][t][i];
    out[blockIdx.x * THREADS + tid] = sum;
}

template <int MT, int NT, int MODE, int THREADS>
void run(float* d, int wg_per_cu) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * wg_per_cu * 3, stages = 3000;
    k<MT, NT, MODE, THREADS><<<blocks, THREADS>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MT, NT, MODE, THREADS><<<blocks, THREADS>>>(d, stages);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)blocks * (THREADS / 64) * stages * 6.0 * MT * NT * 32768.0;
    printf("MT %d NT %d mode %d threads %d wg/CU(limit by LDS 48K: 3) launched x%d: %.2f ms  %.0f TF/s bf16 = %.2f of 2500\n", MT, NT, MODE, THREADS,
           wg_per_cu, ms, fl / ms / 1e9, fl / ms / 1e9 / 2500.0);
}

int main() {
    float* d; hipMalloc(&d, 8192 * 512 * 4);
    // current forward kernel shape: 2 M x 2 N tiles per wave, 256 threads, 2 WG / CU
    run<2, 2, 0, 256>(d, 2); run<2, 2, 1, 256>(d, 2); run<2, 2, 2, 256>(d, 2);
    run<2, 2, 0, 256>(d, 1); run<2, 2, 1, 256>(d, 1); run<2, 2, 2, 256>(d, 1);
    // wider reuse: 4 M x 2 N, 2 M x 4 N, 4 x 4 (one wave per SIMD)
    run<4, 2, 0, 256>(d, 1); run<4, 2, 1, 256>(d, 1); run<4, 2, 2, 256>(d, 1);
    run<2, 4, 0, 256>(d, 1); run<2, 4, 1, 256>(d, 1); run<2, 4, 2, 256>(d, 1);
    run<4, 2, 1, 256>(d, 2); run<4, 2, 2, 256>(d, 2);
    run<4, 2, 2, 512>(d, 1);
    run<1, 2, 2, 256>(d, 2); run<1, 4, 2, 256>(d, 2);
    return 0;
}
