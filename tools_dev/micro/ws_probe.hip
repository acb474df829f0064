// Producer / consumer split probe: 512 threads, waves 0-3 issue the MFMA stream of one dense-conv stage
// (MT x NT tiles, six partial products, fragments from LDS), waves 4-7 do the staging work of a stage
// (NV dependent-free VALU ops per lane, two 16-byte global loads, four ds_write_b64, two ds_write_b128);
// one __syncthreads per stage. What MFMA rate survives?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16 __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

template <int MT, int NT, int NV>
__global__ __launch_bounds__(512, 1) void k(const float4* __restrict__ g, float* out, int stages) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[128 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 128 * 1024 / 4; i += 512) reinterpret_cast<float*>(lds)[i] = 1e-3f * (i & 255);
    __syncthreads();
    float sum = 0.f;
    if (wave < 4) {
        v16 acc[MT][NT];
        for (int m = 0; m < MT; ++m) for (int t = 0; t < NT; ++t) for (int i = 0; i < 16; ++i) acc[m][t][i] = 0.f;
        v8bf fa[MT][3], fb[NT][3];
        const int r = lane & 31, h = lane >> 5;
        for (int s = 0; s < stages; ++s) {
            const unsigned char* Ap = lds + (s & 1) * 49152 + ((wave * MT + (s % 3)) * 34 + r + (s & 1)) * 48 + h * 16;
            const unsigned char* Bp = lds + 98304 + (s % 3) * 9216 + r * 48 + h * 16;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int p = 0; p < 3; ++p) fa[m][p] = *reinterpret_cast<const v8bf*>(Ap + p * 16320 + m * 34 * 48);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int p = 0; p < 3; ++p) fb[t][p] = *reinterpret_cast<const v8bf*>(Bp + p * 3072 + (t & 1) * 32 * 48);
#define MM(PA, PB) _Pragma("unroll") for (int m = 0; m < MT; ++m) _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m][PA], fb[t][PB], acc[m][t], 0, 0, 0);
            MM(0, 2) MM(1, 1) MM(2, 0) MM(0, 1) MM(1, 0) MM(0, 0)
#undef MM
            __syncthreads();
        }
        for (int m = 0; m < MT; ++m) for (int t = 0; t < NT; ++t) for (int i = 0; i < 16; ++i) sum += acc[m][t][i];
    } else {
        const int pt = tid - 256;
        float4 q0 = make_float4(0, 0, 0, 0), q1 = q0;
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = 1.f + i + lane;
        for (int s = 0; s < stages; ++s) {
            const float4 n0 = g[(s * 512 + pt) & 0xffff], n1 = g[(s * 512 + 256 + pt) & 0xffff];     // next stage's loads
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i & 7] = v[i & 7] * 1.0001f + q0.x;                      // staging arithmetic
            unsigned char* w = lds + ((s + 1) & 1) * 49152 + (pt * 48 + (s % 9) * 12288) % 49000;
            *reinterpret_cast<float2*>(w) = make_float2(v[0], v[1]);
            *reinterpret_cast<float2*>(w + 16320) = make_float2(v[2], v[3]);
            *reinterpret_cast<float2*>(w + 32640) = make_float2(v[4], v[5]);
            *reinterpret_cast<float4*>(lds + 98304 + ((s + 2) % 3) * 9216 + pt * 16) = q0;
            if (pt < 128) *reinterpret_cast<float4*>(lds + 98304 + ((s + 2) % 3) * 9216 + 4096 + pt * 16) = q1;
            q0 = n0; q1 = n1;
            __syncthreads();
        }
        for (int i = 0; i < 8; ++i) sum += v[i];
    }
    out[blockIdx.x * 512 + tid] = sum;
}

template <int MT, int NT, int NV>
void run(const float4* g, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 4, stages = 3000;
    k<MT, NT, NV><<<blocks, 512>>>(g, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MT, NT, NV><<<blocks, 512>>>(g, d, stages);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)blocks * 4 * stages * 6.0 * MT * NT * 32768.0;
    printf("consumer MT %d NT %d, producer %3d VALU/stage: %.2f ms  %.0f TF/s bf16 = %.2f of 2500\n", MT, NT, NV, ms, fl / ms / 1e9,
           fl / ms / 1e9 / 2500.0);
}

int main() {
    float4* g; hipMalloc(&g, 65536 * 16); hipMemset(g, 0, 65536 * 16);
    float* d; hipMalloc(&d, 4096 * 512 * 4);
    run<2, 2, 0>(g, d); run<2, 2, 48>(g, d); run<2, 2, 96>(g, d); run<2, 2, 160>(g, d);
    run<4, 2, 96>(g, d); run<4, 2, 192>(g, d); run<2, 4, 96>(g, d); run<2, 4, 192>(g, d);
    return 0;
}
