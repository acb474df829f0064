// LDS read bandwidth per CU with the access pattern of the dense conv kernels: ds_read_b128, 48-byte rows
// (lane (r, h) reads 16 B at row r * 48 + h * 16), 8 waves per CU (2 workgroups of 256).   hipcc --offload-arch=gfx950 -O3 lds_bw.hip -o lds_bw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[48 * 1024];
    for (int i = threadIdx.x; i < 12 * 1024; i += 256) reinterpret_cast<float*>(lds)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const unsigned char* p = lds + wave * 4096 + (MODE == 0 ? r * 48 + h * 16 : lane * 16);
    v4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    for (int i = 0; i < iters; ++i) {
        const unsigned char* q = p + (i & 7) * 1536;
        v4 x0 = *reinterpret_cast<const v4*>(q), x1 = *reinterpret_cast<const v4*>(q + 6144 * 2);
        v4 x2 = *reinterpret_cast<const v4*>(q + 6144 * 4), x3 = *reinterpret_cast<const v4*>(q + 6144 * 5);
        a0 += x0; a1 += x1; a2 += x2; a3 += x3;
        asm volatile("" ::: "memory");
    }
    a0 += a1 + a2 + a3;
    if (a0.x == 123.456f) out[0] = a0.y;
}
int main() {
    float* out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 512;
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters);
            else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)blocks * 256 * iters * 64;
        printf("%s: %.1f TB/s over the chip = %.1f B/clk/CU at 2.4 GHz (256 CUs)\n", mode == 0 ? "48-byte rows" : "contiguous 16 B per lane",
               bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9);
    }
    return 0;
}
