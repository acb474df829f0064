// Semantics probe for ds_read_b64_tr_b16 as an MFMA 32x32x16 operand loader:
// LDS image [pixel][32 ch] of 16-bit values (64-byte rows); lane l should end up with channel l%32,
// pixels 8*(l/32) .. +7.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
    __shared__ __attribute__((aligned(16))) short img[64 * 32];
    for (int i = threadIdx.x; i < 64 * 32; i += 64) img[i] = (short)((i / 32) * 100 + (i % 32));
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
    const int c0 = 16 * (g & 1), pb = 8 * (g >> 1);
    const short* a0 = img + (pb + q) * 32 + c0 + 4 * p;          // row q of the block, columns 4p..4p+3
    const short* a1 = img + (pb + 4 + q) * 32 + c0 + 4 * p;
    v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)a0);
    v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)a1);
    for (int j = 0; j < 4; ++j) { out[l * 8 + j] = lo[j]; out[l * 8 + 4 + j] = hi[j]; }
}
int main() {
    short* d; hipMalloc(&d, 64 * 8 * 2);
    k<<<1, 64>>>(d);
    short h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) {
        const int want = (8 * (l >> 5) + j) * 100 + (l & 31);
        if (h[l * 8 + j] != want) { if (bad < 8) printf("lane %d j %d: got %d want %d\n", l, j, h[l * 8 + j], want); ++bad; }
    }
    printf("%s (%d mismatches)\n", bad ? "MISMATCH" : "OK: lane l holds channel l%%32, pixels 8*(l/32)..+7", bad);
    return bad != 0;
}
