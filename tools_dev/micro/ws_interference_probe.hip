// What does a producer wave on the same SIMD cost the matrix-instruction wave of dense_conv3x3_ws_kernel? The consumer loop of that
// kernel (32x32x16, 24 products + 12 ds_read_b128 per stage, one barrier) on waves 0-3; waves 4-7 execute, per stage, KV vector
// instructions (the halo split's mix: v_pk_mul_f32, v_cvt_pk_f16_f32, v_fma_mix-like fma), KW ds_write_b64 and KB ds_write_b128,
// then the barrier; KS scalar instructions (s_add_u32) per stage in the producer wave, KC in the consumer wave (after its matrix
// instructions). Printed: shader cycles per stage of consumer wave 0 (s_memtime), 256 workgroups, ~2 s of launches each.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float v16 __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v2f __attribute__((ext_vector_type(2)));
#define ROWB 48
#define APL 16320
#define BPL 6144

template <int KV, int KW, int KB, int KS = 0, int KC = 0>
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* clk, int stages) {
    __shared__ __attribute__((aligned(16))) unsigned char As[4 * APL];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[6 * BPL];
    __shared__ __attribute__((aligned(16))) unsigned char Ws[64 * 1024 - 6 * BPL > 0 ? 16384 : 16384];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < (int)sizeof(As) / 4; i += 512) { uint32_t v = (uint32_t)i * 2654435761u + blockIdx.x; v ^= v >> 15; reinterpret_cast<uint32_t*>(As)[i] = (v & 0x83FF83FFu) | 0x34003400u; }
    for (int i = tid; i < (int)sizeof(Bs) / 4; i += 512) { uint32_t v = (uint32_t)i * 2246822519u + blockIdx.x; v ^= v >> 13; reinterpret_cast<uint32_t*>(Bs)[i] = (v & 0x83FF83FFu) | 0x34003400u; }
    __syncthreads();
    if (wave >= 4) {
        v2f a = {1.0f + lane, 2.0f}, b = {0.5f, 0.25f};
        uint4 w4 = make_uint4(lane, 1, 2, 3);
        for (int s = 0; s < stages; ++s) {
#pragma unroll
            for (int i = 0; i < KV / 4; ++i) {          // 4 vector instructions per round: pk_mul, cvt_pk, fma, cvt_pk
                a *= b;
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                h2 h = {(_Float16)a.x, (_Float16)a.y};
                a.x = __builtin_fmaf((float)h.x, -1.0f, a.x + 1.0f);
                h2 g = {(_Float16)a.y, (_Float16)a.x};
                w4.x ^= __builtin_bit_cast(uint32_t, g);
                asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(w4.x));
            }
            if (KS) { int sa = s; _Pragma("unroll") for (int i = 0; i < KS; ++i) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sa) : : "scc"); if (sa == 0x7fffffff) w4.y ^= 1u; }
#pragma unroll
            for (int i = 0; i < KW; ++i) *reinterpret_cast<uint2*>(Ws + ((tid - 256) * 8 + i * 2048) % 16384) = make_uint2(w4.x, w4.y);
#pragma unroll
            for (int i = 0; i < KB; ++i) *reinterpret_cast<uint4*>(Ws + ((tid - 256) * 16 + i * 4096) % 16384) = w4;
            __syncthreads();
        }
        if (w4.x == 0x12345678u) out[tid] = a.x;
        return;
    }
    v16 acc[2][4];
    for (int m = 0; m < 2; ++m) for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) acc[m][t][i] = 0.f;
    const int r = lane & 31, h = lane >> 5;
    v8h fa[2][2], fb[4][2], ga[2][2], gb[4][2];
    int off = r * ROWB + h * 16;
#define RD(FA, FB, S) { const int tap = (S) % 9; const unsigned char* Ap = As + (((S) / 9) & 1) * 2 * APL + ((2 * wave + tap / 3) * 34 + tap % 3) * ROWB + off; \
        const unsigned char* Bp = Bs + ((S) % 3) * 2 * BPL + off; \
        _Pragma("unroll") for (int m = 0; m < 2; ++m) _Pragma("unroll") for (int p = 0; p < 2; ++p) FA[m][p] = *reinterpret_cast<const v8h*>(Ap + p * APL + m * 34 * ROWB); \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) _Pragma("unroll") for (int p = 0; p < 2; ++p) FB[t][p] = *reinterpret_cast<const v8h*>(Bp + p * BPL + t * 32 * ROWB); }
#define MM(FA, FB, PA, PB) _Pragma("unroll") for (int m = 0; m < 2; ++m) _Pragma("unroll") for (int t = 0; t < 4; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA[m][PA], FB[t][PB], acc[m][t], 0, 0, 0);
#define STAGE(CA, CB, XA, XB, S) { RD(XA, XB, (S) + 1) MM(CA, CB, 0, 1) MM(CA, CB, 1, 0) MM(CA, CB, 0, 0) \
        _Pragma("unroll") for (int g_ = 0; g_ < 12; ++g_) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); } \
        if (KC) { _Pragma("unroll") for (int i_ = 0; i_ < KC; ++i_) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc) : : "scc"); } \
        asm volatile("s_barrier" : "+v"(off)); }
    int sc = blockIdx.x;
    RD(fa, fb, 0)
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < stages; s += 2) { STAGE(fa, fb, ga, gb, s) STAGE(ga, gb, fa, fb, s + 1) }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
    for (int m = 0; m < 2; ++m) for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) sum += acc[m][t][i];
    out[blockIdx.x * 256 + tid] = sum + (sc == 12345 ? 1.f : 0.f);
    if (tid == 0) clk[blockIdx.x] = t1 - t0;
}

template <int KV, int KW, int KB, int KS = 0, int KC = 0>
void run(float* d, unsigned long long* clk) {
    const int blocks = 256, stages = 3600;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float total = 0, last = 0; int n = 0;
    while (total < 1500.f) { hipEventRecord(a); k<KV, KW, KB, KS, KC><<<blocks, 512>>>(d, clk, stages); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&last, a, b); total += last; ++n; }
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), clk, 8 * blocks, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("producer per stage: %2d vector + %d ds_write_b64 + %d ds_write_b128 + %d scalar; consumer + %d scalar -> consumer %.0f cycles per stage (768 = the matrix instructions), %.3f ms per launch\n",
           KV, KW, KB, KS, KC, (double)h[blocks / 2] / stages, last);
}

int main() {
    float* d; unsigned long long* clk;
    hipMalloc(&d, 256 * 512 * 4); hipMalloc(&clk, 256 * 8);
    run<0, 0, 0>(d, clk); run<16, 0, 0>(d, clk); run<32, 0, 0>(d, clk); run<64, 0, 0>(d, clk); run<128, 0, 0>(d, clk);
    run<0, 2, 0>(d, clk); run<0, 4, 0>(d, clk); run<0, 0, 2>(d, clk); run<0, 0, 8>(d, clk);
    run<16, 2, 2>(d, clk); run<32, 4, 2>(d, clk);
    run<0, 0, 0, 16>(d, clk); run<0, 0, 0, 64>(d, clk); run<0, 0, 0, 0, 16>(d, clk); run<0, 0, 0, 0, 64>(d, clk);       // scalar instructions: producer / consumer wave
    return 0;
}
