// Probe: cycles per banded product D = C + M x (mm_any of jq_kernels.h) fed from LDS, 1 wave per SIMD.
#include "../juqbox.jl_amd/csrc/jq_kernels.h"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NT, int BW, int MODE>
__global__ __launch_bounds__(256, 1) void k_probe(const double* img, double* out, int reps)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NTILES = band_tiles(NT, BW);
    double* m = (double*)smem;
    for (int i = threadIdx.x; i < NTILES * 64; i += blockDim.x) m[i] = img[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    Arr<NT> A, Ya, Yb, base;
    for (int i = 0; i < NT; ++i) {
        A.t[i] = (d4){1e-3 * lane, 2e-3, 3e-3, 4e-3};
        base.t[i] = (d4){0.5, 0.25, 0.125, 1.0};
    }
    Ya = A;
    const double* M = m + lane;
    long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        if (MODE == 0) {            // dependent chain of products (Horner-like)
            mm_c<NT, BW>(Yb, A, M, Ya);
            mm_c<NT, BW>(Ya, A, M, Yb);
        } else {                    // chain + VALU add between products
            mm_c<NT, BW>(Yb, A, M, Ya);
            a_add(Yb, base);
            mm_c<NT, BW>(Ya, A, M, Yb);
            a_add(Ya, base);
        }
    }
    long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < NT; ++i) s += Ya.t[i][0] + Ya.t[i][1] + Ya.t[i][2] + Ya.t[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0) / (2.0 * reps);
}

int main()
{
    constexpr int NT = 6, BW = 1;
    const int ntiles = band_tiles(NT, BW);
    std::vector<double> img(ntiles * 64, 1e-4);
    double *dimg, *dout;
    CK(hipMalloc(&dimg, img.size() * 8));
    CK(hipMalloc(&dout, 1024 * 256 * 8));
    CK(hipMemcpy(dimg, img.data(), img.size() * 8, hipMemcpyHostToDevice));
    const size_t lds = ntiles * 512;
    CK(hipFuncSetAttribute((const void*)k_probe<NT, BW, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void*)k_probe<NT, BW, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int blocks : {1, 256}) {
        double cyc;
        hipLaunchKernelGGL((k_probe<NT, BW, 0>), dim3(blocks), dim3(256), lds, 0, dimg, dout, 2000);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(&cyc, dout, 8, hipMemcpyDeviceToHost));
        printf("blocks=%d chain          : %.0f cycles per product (%d MFMAs -> ideal %d)\n", blocks, cyc, ntiles, ntiles * 64);
        hipLaunchKernelGGL((k_probe<NT, BW, 1>), dim3(blocks), dim3(256), lds, 0, dimg, dout, 2000);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(&cyc, dout, 8, hipMemcpyDeviceToHost));
        printf("blocks=%d chain + a_add  : %.0f cycles per product\n", blocks, cyc);
    }
    return 0;
}
