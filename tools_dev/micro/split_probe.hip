// h2_split2 (conv_planes.h): is a 4-instruction form (hi and lo halves straight out of v_fma_mixlo/mixhi_f16, the power-of-two scale
// folded into them) bit-identical to the shipped one (scale multiply, v_cvt_pk_f16_f32, two v_fma_mix_f32, v_cvt_pk_f16_f32)?
// 64 M random floats over the whole fp16-relevant range, scales 2^-20 .. 2^20; prints the number of differing words.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../../gga_amd/csrc/conv_planes.h"

__device__ __forceinline__ void split_new(float a, float b, float s, uint32_t& w0, uint32_t& w1) {
    uint32_t hb = 0, lb = 0;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(hb) : "v"(a), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hb) : "v"(b), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "+v"(lb) : "v"(a), "v"(s), "v"(hb));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lb) : "v"(b), "v"(s), "v"(hb));
    w0 = hb; w1 = lb;
}

__global__ void k(unsigned long long* diff, int rounds) {
    uint32_t x = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    unsigned long long d = 0;
    for (int i = 0; i < rounds; ++i) {
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        const uint32_t ea = 100 + (x & 31), eb = 100 + ((x >> 5) & 31);          // exponents 2^-27 .. 2^4
        const float a = __uint_as_float(((x >> 10) & 0x807FFFFFu) | (ea << 23));
        uint32_t y = x * 2246822519u;
        const float b = __uint_as_float((y & 0x807FFFFFu) | (eb << 23));
        const float s = __uint_as_float((127u + 14u - 4u + ((x >> 27) & 7)) << 23);   // puts the largest values near 2^14 .. 2^21 (overflow cases included)
        uint32_t o0, o1, n0, n1;
        h2_split2(a * s, b * s, o0, o1);
        split_new(a, b, s, n0, n1);
        d += (o0 != n0) + (o1 != n1);
    }
    atomicAdd(diff, d);
}

int main() {
    unsigned long long* d; hipMalloc(&d, 8); hipMemset(d, 0, 8);
    k<<<1024, 256>>>(d, 256);
    unsigned long long h = 0; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("differing words: %llu of %llu\n", h, 2ull * 1024 * 256 * 256);
    return 0;
}
