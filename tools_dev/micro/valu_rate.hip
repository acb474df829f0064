// Issue rate of the vector instructions a plane split can be made of (cycles per wave-instruction, one wave per SIMD and four):
// v_sub_f32, v_cvt_f32_f16, v_cvt_pk_f16_f32, v_fma_mix_f32, v_pk_mul_f32, v_pk_add_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <stdint.h>
#define REP16(X) X X X X X X X X X X X X X X X X
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* cyc) {
    float a = threadIdx.x * 0.001f + 1.0f, b = a + 1.0f, c = a + 2.0f, d = a + 3.0f;
    uint32_t u = threadIdx.x * 77u + 0x3c003c00u, v = u + 5u;
    float2 p = make_float2(a, b), q = make_float2(c, d);
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP16(asm volatile("v_sub_f32 %0, %0, %2\n\tv_sub_f32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(c));) }
        if (OP == 1) { REP16(asm volatile("v_cvt_f32_f16 %0, %2\n\tv_cvt_f32_f16 %1, %3" : "+v"(a), "+v"(b) : "v"(u), "v"(v));) }
        if (OP == 2) { REP16(asm volatile("v_cvt_pk_f16_f32 %0, %2, %3\n\tv_cvt_pk_f16_f32 %1, %3, %2" : "+v"(u), "+v"(v) : "v"(a), "v"(b));) }
        if (OP == 3) { REP16(asm volatile("v_fma_mix_f32 %0, %2, -1.0, %0 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %1, %2, -1.0, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a), "+v"(b) : "v"(u));) }
        if (OP == 4) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %2\n\tv_pk_mul_f32 %1, %1, %2" : "+v"(p), "+v"(q) : "v"(p));) }
        if (OP == 5) { REP16(asm volatile("v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %2" : "+v"(p), "+v"(q) : "v"(p));) }
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + p.x + q.y + (float)u + (float)v;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int OP> void run(const char* name, float* d, long long* dc, int threads) {
    const int iters = 2000;
    k<OP><<<256, threads>>>(d, 10, dc); hipDeviceSynchronize();
    k<OP><<<256, threads>>>(d, iters, dc); hipDeviceSynchronize();
    long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("%-18s %d waves/SIMD: %.2f clocks per instruction per wave (s_memtime ticks %.0f per 32)\n", name, threads / 256, (double)c / (iters * 32.0), (double)c / iters);
}
int main() {
    float* d; long long* dc; hipMalloc(&d, 1 << 24); hipMalloc(&dc, 8);
    for (int threads : {256, 1024}) {
        if (threads == 256) { run<0>("v_sub_f32", d, dc, 256); run<1>("v_cvt_f32_f16", d, dc, 256); run<2>("v_cvt_pk_f16_f32", d, dc, 256); run<3>("v_fma_mix_f32", d, dc, 256); run<4>("v_pk_mul_f32", d, dc, 256); run<5>("v_pk_add_f32", d, dc, 256); }
    }
    return 0;
}
