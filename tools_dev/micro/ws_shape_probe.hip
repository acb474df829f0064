// Would the consumer waves of dense_conv3x3_ws_kernel deliver more with v_mfma_f32_16x16x32_f16 than with 32x32x16
// (MI355X_MICROARCH.md 'DVFS give-back' item 7: 1.12-1.15 x on random data at equal cycles per FLOP)? The consumer loop of that
// kernel in isolation, both shapes at the SAME output tile per wave (2 image rows x 32 pixels x 128 columns = 128 accumulator
// registers), the same LDS bytes per FLOP, random fp16 operands in LDS, 4 consumer waves (one per SIMD) + 4 waves that only
// meet the stage barrier (the producers' place), one workgroup per CU, launches back to back for > 2 s:
//   shape 0: stage = K 16: 12 ds_read_b128 (48-byte rows) + 24 MFMA 32x32x16 (three partial products), one barrier
//   shape 1: stage = K 32: 24 ds_read_b128 + 96 MFMA 16x16x32, one barrier (lane group g = lane / 16 reads the 8 channels of
//            K-group g: groups 0, 1 from one 16-channel image, 2, 3 from a second one - the layout a two-tap stage would have)
// Printed: sustained median TF/s (useful fp32-equivalent x 3 products), in-kernel clock from s_memtime / s_memrealtime.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float v16 __attribute__((ext_vector_type(16)));
typedef float v4 __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

#define ROWB 48
#define APL 16320          /* 340 halo pixels x 48 */
#define BPL 6144           /* 128 columns x 48 */

__device__ void fill(unsigned char* lds, int bytes, int tid, int nthreads, int seed) {
    for (int i = tid; i < bytes / 4; i += nthreads) {
        uint32_t v = (uint32_t)i * 2654435761u + seed * 40503u; v ^= v >> 15; v *= 2246822519u; v ^= v >> 13;
        const uint32_t lo = (v & 0x83FFu) | ((13u + (v >> 10) % 5u) << 10), hi = ((v >> 16) & 0x83FFu) | ((13u + (v >> 27) % 5u) << 10);
        reinterpret_cast<uint32_t*>(lds)[i] = lo | (hi << 16);
    }
}

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* clk, int stages) {
    __shared__ __attribute__((aligned(16))) unsigned char As[4 * APL];      // two images x two planes
    __shared__ __attribute__((aligned(16))) unsigned char Bs[6 * 2 * BPL];   // six stage buffers x two planes
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    fill(As, sizeof(As), tid, 512, blockIdx.x);
    fill(Bs, sizeof(Bs), tid, 512, blockIdx.x + 977);
    __syncthreads();
    if (wave >= 4) {                       // the producers' place: barriers only
        for (int s = 0; s < stages; ++s) __syncthreads();
        return;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    if (SHAPE == 0) {
        v16 acc[2][4];
        for (int m = 0; m < 2; ++m) for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) acc[m][t][i] = 0.f;
        const int r = lane & 31, h = lane >> 5;
        v8h fa[2][2], fb[4][2];
        for (int s = 0; s < stages; ++s) {
            const int tap = s % 9;
            const unsigned char* Ap = As + (s / 9 & 1) * 2 * APL + ((2 * wave + tap / 3) * 34 + r + tap % 3) * ROWB + h * 16;
            const unsigned char* Bp = Bs + (s % 3) * 2 * BPL + r * ROWB + h * 16;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int p = 0; p < 2; ++p) fa[m][p] = *reinterpret_cast<const v8h*>(Ap + p * APL + m * 34 * ROWB);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int p = 0; p < 2; ++p) fb[t][p] = *reinterpret_cast<const v8h*>(Bp + p * BPL + t * 32 * ROWB);
#define MM(PA, PB) _Pragma("unroll") for (int m = 0; m < 2; ++m) _Pragma("unroll") for (int t = 0; t < 4; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[m][PA], fb[t][PB], acc[m][t], 0, 0, 0);
            MM(0, 1) MM(1, 0) MM(0, 0)
#undef MM
            __syncthreads();
        }
        for (int m = 0; m < 2; ++m) for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) sum += acc[m][t][i];
    } else {
        v4 acc[4][8];                       // 4 pixel groups of 16 (2 rows x 2 halves) x 8 column tiles of 16
        for (int m = 0; m < 4; ++m) for (int t = 0; t < 8; ++t) for (int i = 0; i < 4; ++i) acc[m][t][i] = 0.f;
        const int r = lane & 15, g = lane >> 4;
        v8h fa[4][2], fb[8][2];
        for (int s = 0; s < stages; ++s) {          // a stage = two taps of one 16-channel chunk (K groups 0, 1 | 2, 3)
            const int tap = (2 * s) % 9 + (g >> 1);
            const int tp = tap % 9;
            const unsigned char* Ap = As + (s & 1) * 2 * APL + ((2 * wave + tp / 3) * 34 + r + tp % 3) * ROWB + (g & 1) * 16;
            const unsigned char* Bp = Bs + ((s % 3) * 2 + (g >> 1)) * 2 * BPL + r * ROWB + (g & 1) * 16;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int p = 0; p < 2; ++p) fa[m][p] = *reinterpret_cast<const v8h*>(Ap + p * APL + ((m >> 1) * 34 + (m & 1) * 16) * ROWB);
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int p = 0; p < 2; ++p) fb[t][p] = *reinterpret_cast<const v8h*>(Bp + p * BPL + t * 16 * ROWB);
#define MM(PA, PB) _Pragma("unroll") for (int m = 0; m < 4; ++m) _Pragma("unroll") for (int t = 0; t < 8; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[m][PA], fb[t][PB], acc[m][t], 0, 0, 0);
            MM(0, 1) MM(1, 0) MM(0, 0)
#undef MM
            __syncthreads();
        }
        for (int m = 0; m < 4; ++m) for (int t = 0; t < 8; ++t) for (int i = 0; i < 4; ++i) sum += acc[m][t][i];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + tid] = sum;
    if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE>
void run(float* d, unsigned long long* clk, const char* what) {
    const int blocks = 256, k16_stages = 9 * 400;                 // per launch: 3600 K-16 stages = 1800 K-32 stages
    const int stages = SHAPE == 0 ? k16_stages : k16_stages / 2;
    const double fl = (double)blocks * 4 * k16_stages * 3.0 * 8 * 32768.0;      // 24 MFMAs of 32x32x16 per wave and K-16 stage
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    std::vector<float> ms;
    float total = 0;
    while (total < 2500.f) {            // > 2 s back to back
        hipEventRecord(a);
        k<SHAPE><<<blocks, 512>>>(d, clk, stages);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float t; hipEventElapsedTime(&t, a, b);
        ms.push_back(t); total += t;
    }
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
    std::vector<double> ghz;
    for (int i = 0; i < blocks; ++i) ghz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    std::vector<float> tail(ms.begin() + ms.size() / 2, ms.end());
    std::sort(tail.begin(), tail.end());
    const float med = tail[tail.size() / 2];
    printf("%-28s launches %zu  first %.3f ms  sustained median %.3f ms  %.0f TF/s issued (%.3f of 2500)  in-kernel clock %.2f GHz\n", what, ms.size(),
           ms[0], med, fl / med / 1e9, fl / med / 1e9 / 2500.0, ghz[blocks / 2]);
}

int main() {
    float* d; unsigned long long* clk;
    hipMalloc(&d, 256 * 512 * 4); hipMalloc(&clk, 256 * 2 * 8);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>(d, clk, "32x32x16, K 16 stages");
        run<1>(d, clk, "16x16x32, K 32 stages");
    }
    return 0;
}
