// Which lane does `row_ror:4` read?  (The dense policy of the cooperative-quad kernels rotates a state register by 4 s lanes inside each
// 16-lane row; the operator images are built for dst[p] = src[(p - n) mod 16].)  Prints the source lane of the lanes 0 .. 15 for ror:4.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out)
{
    const int lane = threadIdx.x;
    out[lane] = __builtin_amdgcn_update_dpp(0, lane, 0x120 + 4, 0xf, 0xf, false);            // row_ror:4
    out[64 + lane] = __builtin_amdgcn_update_dpp(0, lane, 0x110 + 4, 0xf, 0xf, false);       // row_shr:4
    out[128 + lane] = __builtin_amdgcn_update_dpp(0, lane, 0x100 + 4, 0xf, 0xf, false);      // row_shl:4
}
int main()
{
    int* d;
    int h[192];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int t = 0; t < 3; ++t) {
        printf("%s:", t == 0 ? "row_ror:4" : t == 1 ? "row_shr:4" : "row_shl:4");
        for (int i = 0; i < 20; ++i) printf(" %d", h[64 * t + i]);
        printf("\n");
    }
    return 0;
}
