// v_mfma_f64_4x4x4_4b_f64 on gfx950: operand layouts and rate.
// D_b (4x4) = A_b (4x4) * B_b (4x4) + C_b for four blocks b; one double per lane for A, B, C, D.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ void k_layout(const double* A, const double* B, double* D)
{
    const int l = threadIdx.x;
    D[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, 0, 0, 0);
}
__global__ __launch_bounds__(256, 1) void k_rate(double* out, int iters)
{
    const int l = threadIdx.x & 63;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    double a = 1.0 + 1e-6 * l, b = 1.0 - 1e-6 * l;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 24; ++m) acc[m % 6] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[m % 6], 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + acc[5];
}
int main()
{
    double hA[64], hB[64], hD[64];
    double *dA, *dB, *dD;
    (void)hipMalloc(&dA, 512); (void)hipMalloc(&dB, 512); (void)hipMalloc(&dD, 256 * 256 * 8);
    // unit probes: A has a single 1 at lane la, B a single 1 at lane lb -> which D lanes light up?
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            for (int i = 0; i < 64; ++i) hA[i] = hB[i] = 0.0;
            hA[la] = 1.0; hB[lb] = 1.0;
            (void)hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
            (void)hipMemcpy(hD, dD, 512, hipMemcpyDeviceToHost);
            for (int ld = 0; ld < 64; ++ld)
                if (hD[ld] != 0.0) printf("A lane %2d x B lane %2d -> D lane %2d\n", la, lb, ld);
        }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0; const int iters = 400000;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_rate, dim3(256), dim3(256), 0, 0, dD, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("24 x v_mfma_f64_4x4x4_4b: %.1f cycles per iteration at 2.4 GHz (%.1f per MFMA)\n", ms * 1e-3 * 2.4e9 / iters, ms * 1e-3 * 2.4e9 / iters / 24);
    return 0;
}
