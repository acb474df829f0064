// If the producers of dense_conv3x3_ws_kernel only COPIED the halo (fp32, no split) and the consumer waves split their own A
// operand between their matrix instructions - would the vector instructions be free there? (ws_interference_probe: a vector
// instruction of the co-resident producer wave costs the matrix wave ~9 cycles.) The consumer loop of that kernel with the A
// fragments read as raw fp32 (pixel rows of 80 bytes: 16 channels + 16 bytes of padding, ds_read_b128 conflict-free) and split
// with the 4-instruction form of conv_planes.h (h2_split2s): per stage MT x 16 vector instructions, placed group by group among the
// 24 matrix instructions. Producers: KB ds_write_b128 per stage and the barrier.
//   MT = 2, NT = 4: 4 raw reads + 8 weight reads + 32 vector + 24 MFMA;   MT = 4, NT = 2: 8 + 4 reads, 64 vector, 24 MFMA
// Printed: shader cycles per stage of consumer wave 0 (768 = the matrix instructions alone).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float v16 __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
#define ROWB 48
#define ROWF 80
#define BPL 6144

__device__ __forceinline__ void split4(float a, float b, float s, uint32_t& w0, uint32_t& w1) {
    uint32_t hb, lb;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hb) : "v"(a), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hb) : "v"(b), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(lb) : "v"(a), "v"(s), "v"(hb));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lb) : "v"(b), "v"(s), "v"(hb));
    w0 = hb; w1 = lb;
}

template <int MT, int NT, int SPLIT, int KB>
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* clk, int stages, float scale) {
    constexpr int HP = (4 * MT + 2) * 34, AIMG = HP * ROWF;
    __shared__ __attribute__((aligned(16))) unsigned char As[2 * AIMG];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[3 * 2 * BPL];
    __shared__ __attribute__((aligned(16))) unsigned char Ws[16384];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < (int)sizeof(As) / 4; i += 512) { uint32_t v = (uint32_t)i * 2654435761u + blockIdx.x; v ^= v >> 15; reinterpret_cast<float*>(As)[i] = (float)(v & 0xFFFF) * (1.0f / 65536.0f) - 0.5f; }
    for (int i = tid; i < (int)sizeof(Bs) / 4; i += 512) { uint32_t v = (uint32_t)i * 2246822519u + blockIdx.x; v ^= v >> 13; reinterpret_cast<uint32_t*>(Bs)[i] = (v & 0x83FF83FFu) | 0x34003400u; }
    __syncthreads();
    if (wave >= 4) {
        uint4 w4 = make_uint4(lane, 1, 2, 3);
        for (int s = 0; s < stages; ++s) {
#pragma unroll
            for (int i = 0; i < KB; ++i) *reinterpret_cast<uint4*>(Ws + ((tid - 256) * 16 + i * 4096) % 16384) = w4;
            __syncthreads();
        }
        return;
    }
    v16 acc[MT][NT];
    for (int m = 0; m < MT; ++m) for (int t = 0; t < NT; ++t) for (int i = 0; i < 16; ++i) acc[m][t][i] = 0.f;
    const int r = lane & 31, h = lane >> 5;
    union Frag { v8h v; uint32_t u[4]; };
    Frag fa[MT][2], ga[MT][2];
    v8h fb[NT][2], gb[NT][2];
    float4 raw[MT][2];
    int offa = r * ROWF + h * 32, offb = r * ROWB + h * 16;
    // a stage, written out group by group (inline-asm vector instructions are invisible to sched_group_barrier): group g of 12 =
    // matrix instructions 2g, 2g + 1 of the CURRENT set, fragment read g of the NEXT stage (the 2 MT raw fp32 reads first, then the
    // 2 NT weight reads), and - from group 4 on, when the first raw reads have landed - PPG pairs of the next set's split
    constexpr int NRA = 2 * MT, PPG = (4 * MT) / 8;
#define STAGE(CA, CB, XA, XB, S) { \
        const int tap_ = ((S) + 1) % 9; \
        const unsigned char* Ap = As + ((((S) + 1) / 9) & 1) * AIMG + ((MT * wave + tap_ / 3) * 34 + tap_ % 3) * ROWF + offa; \
        const unsigned char* Bp = Bs + (((S) + 1) % 3) * 2 * BPL + offb; \
        _Pragma("unroll") for (int g = 0; g < 12; ++g) { \
            _Pragma("unroll") for (int i = 2 * g; i < 2 * g + 2; ++i) { \
                const int pr = i / (MT * NT), m = (i / NT) % MT, t = i % NT, pa = pr == 1 ? 1 : 0, pb = pr == 0 ? 1 : 0; \
                acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(CA[m][pa].v, CB[t][pb], acc[m][t], 0, 0, 0); } \
            if (g < NRA) raw[g / 2][g % 2] = *reinterpret_cast<const float4*>(Ap + (g / 2) * 34 * ROWF + (g % 2) * 16); \
            else XB[(g - NRA) / 2][(g - NRA) % 2] = *reinterpret_cast<const v8h*>(Bp + ((g - NRA) % 2) * BPL + ((g - NRA) / 2) * 32 * ROWB); \
            if (SPLIT && g >= 4) { \
                _Pragma("unroll") for (int j = (g - 4) * PPG; j < (g - 4 + 1) * PPG; ++j) { \
                    const float4 v = raw[j / 4][(j % 4) / 2]; \
                    if (j % 2 == 0) split4(v.x, v.y, scale, XA[j / 4][0].u[j % 4], XA[j / 4][1].u[j % 4]); \
                    else split4(v.z, v.w, scale, XA[j / 4][0].u[j % 4], XA[j / 4][1].u[j % 4]); } } \
            __builtin_amdgcn_sched_barrier(0); } \
        asm volatile("s_barrier" : "+v"(offa), "+v"(offb)); }
#define RD_A(S) { const int tap = (S) % 9; const unsigned char* Ap = As + (((S) / 9) & 1) * AIMG + ((MT * wave + tap / 3) * 34 + tap % 3) * ROWF + offa; \
        _Pragma("unroll") for (int m = 0; m < MT; ++m) { raw[m][0] = *reinterpret_cast<const float4*>(Ap + m * 34 * ROWF); raw[m][1] = *reinterpret_cast<const float4*>(Ap + m * 34 * ROWF + 16); } }
#define RD_B(FB, S) { const unsigned char* Bp = Bs + ((S) % 3) * 2 * BPL + offb; \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) _Pragma("unroll") for (int p = 0; p < 2; ++p) FB[t][p] = *reinterpret_cast<const v8h*>(Bp + p * BPL + t * 32 * ROWB); }
#define SPLIT_A(FA) { _Pragma("unroll") for (int m = 0; m < MT; ++m) { \
        split4(raw[m][0].x, raw[m][0].y, scale, FA[m][0].u[0], FA[m][1].u[0]); split4(raw[m][0].z, raw[m][0].w, scale, FA[m][0].u[1], FA[m][1].u[1]); \
        split4(raw[m][1].x, raw[m][1].y, scale, FA[m][0].u[2], FA[m][1].u[2]); split4(raw[m][1].z, raw[m][1].w, scale, FA[m][0].u[3], FA[m][1].u[3]); } }
    RD_A(0) RD_B(fb, 0) SPLIT_A(fa)
    if (!SPLIT) { for (int m = 0; m < MT; ++m) for (int p = 0; p < 2; ++p) { ga[m][p] = fa[m][p]; } }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < stages; s += 2) { STAGE(fa, fb, ga, gb, s) STAGE(ga, gb, fa, fb, s + 1) }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
    for (int m = 0; m < MT; ++m) for (int t = 0; t < NT; ++t) for (int i = 0; i < 16; ++i) sum += acc[m][t][i];
    if (!SPLIT) for (int m = 0; m < MT; ++m) sum += raw[m][0].x + raw[m][1].w;
    out[blockIdx.x * 256 + tid] = sum;
    if (tid == 0) clk[blockIdx.x] = t1 - t0;
}

template <int MT, int NT, int SPLIT, int KB>
void run(float* d, unsigned long long* clk) {
    const int blocks = 256, stages = 3600;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float total = 0, last = 0; int n = 0;
    while (total < 1500.f) { hipEventRecord(a); k<MT, NT, SPLIT, KB><<<blocks, 512>>>(d, clk, stages, 4.0f); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&last, a, b); total += last; ++n; }
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), clk, 8 * blocks, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("MT %d NT %d: consumer %s (%2d vector per stage), producers %d ds_write_b128 -> %.0f cycles per stage (768 = the matrix instructions), %.3f ms per launch\n",
           MT, NT, SPLIT ? "reads fp32 and splits" : "reads fp32, no split ", SPLIT ? 16 * MT : 0, KB, (double)h[blocks / 2] / stages, last);
}

int main() {
    float* d; unsigned long long* clk;
    hipMalloc(&d, 256 * 512 * 4); hipMalloc(&clk, 256 * 8);
    run<2, 4, 0, 0>(d, clk); run<2, 4, 1, 0>(d, clk); run<2, 4, 1, 3>(d, clk);
    run<4, 2, 0, 0>(d, clk); run<4, 2, 1, 0>(d, clk); run<4, 2, 1, 3>(d, clk);
    return 0;
}
