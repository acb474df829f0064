// Stand-alone timing of the head output conv kernels at the train-step shape, with a 1 GiB memset
// between launches so nothing is served from L2/MALL.  hipcc -O3 --offload-arch=gfx950 [-DHW_UNROLL=..]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../gga_amd/csrc/api.cc"
#include "../../gga_amd/csrc/headconv.hip"

int main() {
    const int B = 16, H = 248, W = 216;
    const size_t nx = (size_t)B * H * W * 64;
    float *x, *y, *w, *bias, *gw, *gb, *trash;
    void* ws;
    hipMalloc(&x, nx * 4); hipMalloc(&y, (size_t)B * H * W * 4 * 4); hipMalloc(&w, 4 * 64 * 9 * 4); hipMalloc(&bias, 16);
    hipMalloc(&gw, 4 * 64 * 9 * 4); hipMalloc(&gb, 16); hipMalloc(&trash, 1ull << 30);
    const size_t wsb = gga_head_conv3x3_workspace_bytes(4);
    hipMalloc(&ws, wsb);
    hipMemset(x, 0x3c, nx * 4); hipMemset(y, 0x3c, (size_t)B * H * W * 16); hipMemset(w, 0x3c, 4 * 64 * 9 * 4); hipMemset(bias, 0, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int cout = 1; cout <= 3; ++cout) {
        float tf = 0, tw = 0;
        const int reps = 10;
        for (int r = 0; r < reps + 2; ++r) {
            float ms;
            hipMemsetAsync(trash, r, 1ull << 30, 0);
            hipEventRecord(e0, 0);
            if (gga_head_conv3x3_fwd(x, nullptr, w, bias, B, H, W, 64, cout, y, 0)) { printf("fwd: %s\n", gga_last_error()); return 1; }
            hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); if (r >= 2) tf += ms;
            hipMemsetAsync(trash, r + 1, 1ull << 30, 0);
            hipEventRecord(e0, 0);
            if (gga_head_conv3x3_wgrad(x, nullptr, y, B, H, W, 64, cout, gw, gb, ws, wsb, 0)) { printf("wgrad: %s\n", gga_last_error()); return 1; }
            hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); if (r >= 2) tw += ms;
        }
        printf("cout %d: fwd %.1f us, wgrad(+final) %.1f us\n", cout, 1e3 * tf / reps, 1e3 * tw / reps);
    }
    return 0;
}
