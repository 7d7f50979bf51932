// Probe: time per JQ_BW_T4 product (mm_t4 of jq_kernels.h) fed from LDS, 1 wave per SIMD, all CUs busy.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DJQ_BW=8 -mllvm -amdgpu-mfma-vgpr-form=1 -o t4_mm_probe t4_mm_probe.hip
#include "../juqbox.jl_amd/csrc/jq_kernels.h"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NT, int MODE>
__global__ __launch_bounds__(256, 1) void k_probe(const double* img, double* out, int reps, int tmode)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ELEMS = JQ_T4_ELEMS(NT);
    double* m = (double*)smem;
    for (int i = threadIdx.x; i < ELEMS; i += blockDim.x) m[i] = img[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    Arr<NT> A, Ya, Yb, base;
    for (int i = 0; i < NT; ++i) {
        A.t[i] = jq_row((d4){1e-3 * lane, 2e-3, 3e-3, 4e-3});
        base.t[i] = jq_row((d4){0.5, 0.25, 0.125, 1.0});
    }
    Ya = A;
    Yb = A;
    const double* M = m + lane;
    for (int r = 0; r < reps; ++r) {
        if (MODE == 0) {            // in-place Horner recurrence
            mm_c<NT, JQ_BW_T4>(Ya, A, M, Ya);
            mm_c<NT, JQ_BW_T4>(Ya, A, M, Ya);
        } else if (MODE == 1) {     // ping-pong
            mm_c<NT, JQ_BW_T4>(Yb, A, M, Ya);
            mm_c<NT, JQ_BW_T4>(Ya, A, M, Yb);
        } else if (MODE == 2) {     // trace products
            mm_z_bw<NT, JQ_BW_T4>(Yb, M, Ya, tmode);
            mm_z_bw<NT, JQ_BW_T4>(Ya, M, Yb, tmode);
        } else if (MODE == 3) {     // elementwise work between products
            mm_c<NT, JQ_BW_T4>(Ya, A, M, Ya);
            a_add(Ya, base);
            mm_c<NT, JQ_BW_T4>(Ya, A, M, Ya);
            a_add(Ya, base);
        } else if (MODE == 4) {     // product + workgroup barrier
            mm_c<NT, JQ_BW_T4>(Ya, A, M, Ya);
            __builtin_amdgcn_s_barrier();
            mm_c<NT, JQ_BW_T4>(Ya, A, M, Ya);
            __builtin_amdgcn_s_barrier();
        }
    }
    double s = 0;
    for (int i = 0; i < NT; ++i) s += Ya.t[i][0] + Ya.t[i][1] + Ya.t[i][2] + Ya.t[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NT, int MODE>
int run(const char* name, const double* dimg, double* dout, int tmode = 7)
{
    const size_t lds = JQ_T4_ELEMS(NT) * 8;
    CK(hipFuncSetAttribute((const void*)k_probe<NT, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int reps = 4000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_probe<NT, MODE>), dim3(256), dim3(256), lds, 0, dimg, dout, 10, tmode);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_probe<NT, MODE>), dim3(256), dim3(256), lds, 0, dimg, dout, reps, tmode);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("NT=%d %-34s %7.1f ns per product\n", NT, name, ms * 1e6 / (2.0 * reps));
    return 0;
}

int main()
{
    constexpr int NT = 6;
    std::vector<double> img(JQ_T4_ELEMS(NT), 1e-4);
    double *dimg, *dout;
    CK(hipMalloc(&dimg, img.size() * 8));
    CK(hipMalloc(&dout, 1024 * 256 * 8));
    CK(hipMemcpy(dimg, img.data(), img.size() * 8, hipMemcpyHostToDevice));
    run<NT, 0>("in-place chain", dimg, dout);
    run<NT, 1>("ping-pong chain", dimg, dout);
    run<NT, 2>("trace product, full", dimg, dout, 7);
    run<NT, 2>("trace product, DIAG", dimg, dout, 1);
    run<NT, 2>("trace product, RTERMS", dimg, dout, 2);
    run<NT, 2>("trace product, MTERMS", dimg, dout, 4);
    run<NT, 3>("in-place chain + a_add", dimg, dout);
    run<NT, 4>("in-place chain + s_barrier", dimg, dout);
    run<3, 0>("in-place chain", dimg, dout);
    return 0;
}
