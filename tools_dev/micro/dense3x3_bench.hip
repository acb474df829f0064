// Stand-alone timing of dense_conv3x3_x9_kernel at [16,64,248,216] -> 64 (1 GiB memset between launches).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../gga_amd/csrc/api.cc"
#include "../../gga_amd/csrc/sparse_conv.hip"
int main() {
    const int B = 16, H = 248, W = 216, C = 64, CO = 64;
    const size_t n = (size_t)B * H * W;
    float *x, *y, *w, *trash; void* wp;
    hipMalloc(&x, n * C * 4); hipMalloc(&y, n * CO * 4); hipMalloc(&w, 9 * C * CO * 4); hipMalloc(&trash, 1ull << 30);
    hipMalloc(&wp, gga_sparse_split_weight_bytes(9, C, CO));
    hipMemset(x, 0x3c, n * C * 4); hipMemset(w, 0x3c, 9 * C * CO * 4);
    gga_sparse_pack_weight_split(w, 9, C, CO, 0, wp, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float tot = 0; const int reps = 10;
    for (int r = 0; r < reps + 2; ++r) {
        float ms;
        hipMemsetAsync(trash, r, 1ull << 30, 0);
        hipEventRecord(e0, 0);
        if (gga_dense_conv3x3(x, wp, B, H, W, C, CO, y, 0)) { printf("%s\n", gga_last_error()); return 1; }
        hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); if (r >= 2) tot += ms;
    }
    const double gf = 2.0 * n * C * CO * 9 / 1e9;
    printf("%.1f us  %.0f TFLOP/s-eq\n", 1e3 * tot / reps, gf / (tot / reps));
    return 0;
}
