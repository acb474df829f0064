// What does a SUSTAINED stream of v_mfma_f32_32x32x16_f16 deliver on this part? (mfma_shape.hip times one ~5 ms launch; the
// step keeps the matrix pipes loaded for tens of milliseconds, and the PMC runs of rounds 1-3 saw the clock give way as the
// pipes got busier.) The two-plane forward shape: MT x NT accumulator tiles per wave, three partial products per k-step.
//   mode 0: pure MFMA, fragments constant in registers
//   mode 1: + the stage's LDS fragment reads (2*MT + 2*NT ds_read_b128, 48-byte rows)
//   mode 2: + one __syncthreads per stage
// Every configuration runs LAUNCHES back-to-back launches of ~2-4 ms; printed: TFLOP/s of the first launch, of the median of
// the second half, and the ratio of nominal 2500.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float v16 __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

template <int MT, int NT, int MODE, int THREADS, int OCC, int RANDOM>
__global__ __launch_bounds__(THREADS, OCC) void k(float* out, int stages) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[48 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // RANDOM: pseudo-random fp16 pairs (sign, 5 exponent values around 1, 10 random significand bits) - the matrix pipes' power
    // depends on what they multiply; 0: a smooth ramp
    for (int i = tid; i < 48 * 1024 / 4; i += THREADS) {
        if (RANDOM) {
            uint32_t v = (uint32_t)i * 2654435761u + blockIdx.x * 40503u; v ^= v >> 15; v *= 2246822519u; v ^= v >> 13;
            const uint32_t lo = (v & 0x83FFu) | ((13u + (v >> 10) % 5u) << 10), hi = ((v >> 16) & 0x83FFu) | ((13u + (v >> 27) % 5u) << 10);
            reinterpret_cast<uint32_t*>(lds)[i] = lo | (hi << 16);
        } else reinterpret_cast<float*>(lds)[i] = 1e-3f * (i & 255);
    }
    __syncthreads();
    v16 acc[MT][NT];
    for (int m = 0; m < MT; ++m) for (int t = 0; t < NT; ++t) for (int i = 0; i < 16; ++i) acc[m][t][i] = 0.f;
    v8h fa[MT][2], fb[NT][2];
    for (int m = 0; m < MT; ++m) for (int p = 0; p < 2; ++p) for (int i = 0; i < 8; ++i) fa[m][p][i] = (_Float16)(0.001f * (lane + p));
    for (int t = 0; t < NT; ++t) for (int p = 0; p < 2; ++p) for (int i = 0; i < 8; ++i) fb[t][p][i] = (_Float16)(0.002f * (lane + p));
    const int r = lane & 31, h = lane >> 5;
    for (int s = 0; s < stages; ++s) {
        if (MODE >= 1) {
            const unsigned char* Ap = lds + (((wave & 3) * 2 + (s % 3)) * 34 + r + (s & 1)) * 48 + h * 16;
            const unsigned char* Bp = lds + 20480 + (s % 3) * 6144 + r * 48 + h * 16;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int p = 0; p < 2; ++p) fa[m][p] = *reinterpret_cast<const v8h*>(Ap + p * 8192 + (m & 1) * 34 * 48);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int p = 0; p < 2; ++p) fb[t][p] = *reinterpret_cast<const v8h*>(Bp + p * 3072 + (t & 1) * 32 * 48);
        }
#define MM(PA, PB) _Pragma("unroll") for (int m = 0; m < MT; ++m) _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[m][PA], fb[t][PB], acc[m][t], 0, 0, 0);
        MM(0, 1) MM(1, 0) MM(0, 0)
#undef MM
        if (MODE >= 2) __syncthreads();
    }
    float sum = 0;
    for (int m = 0; m < MT; ++m) for (int t = 0; t < NT; ++t) for (int i = 0; i < 16; ++i) sum += acc[m][t][i];
    out[blockIdx.x * THREADS + tid] = sum;
}

template <int MT, int NT, int MODE, int THREADS, int OCC, int RANDOM = 0>
void run(float* d, const char* what) {
    const int LAUNCHES = 60;
    std::vector<hipEvent_t> ev(LAUNCHES + 1);
    for (auto& e : ev) hipEventCreate(&e);
    const int blocks = 256 * OCC * 2, stages = 1500 * 4 / (MT * NT) * 2;
    k<MT, NT, MODE, THREADS, OCC, RANDOM><<<blocks, THREADS>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(ev[0]);
    for (int i = 0; i < LAUNCHES; ++i) { k<MT, NT, MODE, THREADS, OCC, RANDOM><<<blocks, THREADS>>>(d, stages); hipEventRecord(ev[i + 1]); }
    hipDeviceSynchronize();
    const double fl = (double)blocks * (THREADS / 64) * stages * 3.0 * MT * NT * 32768.0;
    std::vector<float> ms(LAUNCHES);
    for (int i = 0; i < LAUNCHES; ++i) hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]);
    const float first = ms[0];
    std::vector<float> tail(ms.begin() + LAUNCHES / 2, ms.end());
    std::sort(tail.begin(), tail.end());
    const float med = tail[tail.size() / 2];
    float total = 0; for (float m : ms) total += m;
    printf("%-34s MT %d NT %d mode %d threads %d wg/CU %d: first %.2f ms %.0f TF/s (%.2f), sustained median %.2f ms %.0f TF/s (%.2f of 2500), %d launches %.0f ms\n",
           what, MT, NT, MODE, THREADS, OCC, first, fl / first / 1e9, fl / first / 1e9 / 2500.0, med, fl / med / 1e9, fl / med / 1e9 / 2500.0, LAUNCHES, total);
    for (auto& e : ev) hipEventDestroy(e);
}

int main() {
    float* d; hipMalloc(&d, 8192 * 1024 * 4);
    run<2, 2, 0, 256, 2>(d, "64-col form, pure");
    run<2, 2, 1, 256, 2>(d, "64-col form, + LDS reads");
    run<2, 2, 2, 256, 2>(d, "64-col form, + barrier");
    run<2, 4, 0, 512, 1>(d, "128-col 16-row form, pure");
    run<2, 4, 1, 512, 1>(d, "128-col 16-row form, + LDS reads");
    run<2, 4, 2, 512, 1>(d, "128-col 16-row form, + barrier");
    run<2, 4, 0, 256, 2>(d, "128-col 8-row form, pure");
    run<2, 4, 2, 256, 2>(d, "128-col 8-row form, + barrier");
    run<4, 4, 0, 256, 1>(d, "4x4 one wave/SIMD, pure");
    run<4, 4, 2, 256, 1>(d, "4x4 one wave/SIMD, + barrier");
    run<2, 2, 0, 256, 1>(d, "2x2 one wave/SIMD, pure");
    run<2, 2, 2, 256, 2, 1>(d, "64-col + barrier, RANDOM data");
    run<2, 4, 2, 512, 1, 1>(d, "128-col 16-row + barrier, RANDOM");
    run<2, 4, 2, 256, 2, 1>(d, "128-col 8-row + barrier, RANDOM");
    run<4, 4, 2, 256, 1, 1>(d, "4x4 + barrier, RANDOM");
    return 0;
}
