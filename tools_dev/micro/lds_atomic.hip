// LDS float accumulate: ds_add_f32 (lane = consecutive address) against a plain read-add-write by the owning wave.
// hipcc --offload-arch=gfx950 -O3 -o lds_atomic lds_atomic.hip && ./lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float win[4][4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4 * 4096; i += 256) (&win[0][0])[i] = 0.f;
    __syncthreads();
    float v = 1.0f + lane;
    unsigned pos = (wave * 977 + blockIdx.x * 31) & 63;
    for (int it = 0; it < iters; ++it) {
        pos = (pos * 1664525u + 1013904223u);
        const int cell = (pos >> 16) & 47;                      // window pixel
        float* p = (MODE == 0 ? &win[0][0] : &win[wave][0]) + cell * 64 + lane;
        if (MODE == 0) {                                        // shared window, atomics (4 corners)
            atomicAdd(p, v); atomicAdd(p + 64, v); atomicAdd(p + 64 * 8, v); atomicAdd(p + 64 * 9, v);
        } else {                                                // wave-private window, read-add-write
            const float a = p[0], b = p[64], c = p[64 * 8], d = p[64 * 9];
            p[0] = a + v; p[64] = b + v; p[64 * 8] = c + v; p[64 * 9] = d + v;
        }
    }
    __syncthreads();
    float s = 0;
    for (int i = tid; i < 4 * 4096; i += 256) s += (&win[0][0])[i];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
void run(float* d, const char* name) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 512, iters = 20000;
    k<MODE><<<blocks, 256>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waveops = (double)blocks * 4 * iters * 4;      // wave-level 64-lane accumulations
    printf("%s: %.2f ms, %.1f G lane-adds/s, %.1f cycles per wave-level add per CU at 2.4 GHz\n", name, ms,
           waveops * 64 / ms / 1e6, ms * 1e-3 * 2.4e9 / (waveops / 256.0));
}

int main() {
    float* d; hipMalloc(&d, 512 * 256 * 4);
    run<0>(d, "ds_add_f32, shared window   ");
    run<1>(d, "read-add-write, private wins");
    return 0;
}
