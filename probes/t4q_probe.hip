// Probe for a "quad" layout of the JQ_BW_T4 product: one wave = 4 state columns, a register of a state array = a whole
// 16-row block (lane 16 i + 4 b + j  <->  row 16 mt + 4 b + i, column j), so an Ntot = 96 array is 6 registers and ONE
// v_mfma_f64_4x4x4_4b does the four 4x4 diagonal blocks of a 16-row block (its 4 "blocks" are the 4-row groups b).
// Couplings: (i, i+-4) = neighbouring groups = the same register shifted by 4 lanes inside each 16-lane row
// (2 x v_mov_b32_dpp row_shr/row_shl: 64-bit DPP only has row_newbcast), (i, i+-16) = the neighbouring register.
// Measures ns per product (in-place Horner recurrence Y <- A + S Y) for 1, 2, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int CTRL>
__device__ __forceinline__ double row_shift(double x)
{
    union { double d; int i[2]; } a, b;
    a.d = x;
    b.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], CTRL, 0xf, 0xf, true);
    b.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], CTRL, 0xf, 0xf, true);
    return b.d;
}
// image per block: [A tile 64][c_dn 64][c_up 64][c_lo 64][c_hi 64]
template <int NT, bool REGS, bool SPLIT = false>
__global__ __launch_bounds__(256) void k_probe(const double* img, double* out, int reps)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* m = (double*)smem;
    for (int i = threadIdx.x; i < NT * 320; i += blockDim.x) m[i] = img[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const double* M = m + lane;
    double A[NT], Y[NT];
    for (int i = 0; i < NT; ++i) A[i] = 1e-3 * (lane + i), Y[i] = A[i];
    double cr[NT][5];
    if (REGS)
        for (int mt = 0; mt < NT; ++mt)
            for (int k = 0; k < 5; ++k) cr[mt][k] = M[mt * 320 + k * 64];
    for (int r = 0; r < reps; ++r) {
        double xold = 0.0;
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) {
            const double a = REGS ? cr[mt][0] : M[mt * 320], cdn = REGS ? cr[mt][1] : M[mt * 320 + 64], cup = REGS ? cr[mt][2] : M[mt * 320 + 128];
            const double clo = REGS ? cr[mt][3] : M[mt * 320 + 192], chi = REGS ? cr[mt][4] : M[mt * 320 + 256];
            const double x = Y[mt];
            double acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x, A[mt], 0, 0, 0);
            const double xdn = row_shift<0x114>(x);   // row_shr:4  (lane n <- n-4)
            const double xup = row_shift<0x104>(x);   // row_shl:4  (lane n <- n+4)
            if (SPLIT) {                              // the coupling chain runs beside the MFMA, one add at the end
                double z = cdn * xdn;
                z = fma(cup, xup, z);
                if (mt > 0) z = fma(clo, xold, z);
                if (mt + 1 < NT) z = fma(chi, Y[mt + 1], z);
                acc += z;
            } else {
                acc = fma(cdn, xdn, acc);
                acc = fma(cup, xup, acc);
                if (mt > 0) acc = fma(clo, xold, acc);
                if (mt + 1 < NT) acc = fma(chi, Y[mt + 1], acc);
            }
            xold = x;
            Y[mt] = acc;
        }
    }
    double s = 0;
    for (int i = 0; i < NT; ++i) s += Y[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NT, bool REGS, bool SPLIT = false>
int run(const double* dimg, double* dout, int wgs_per_cu)
{
    const size_t lds = NT * 320 * 8;
    const int reps = 20000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_probe<NT, REGS, SPLIT>), dim3(256 * wgs_per_cu), dim3(256), lds, 0, dimg, dout, 10);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_probe<NT, REGS, SPLIT>), dim3(256 * wgs_per_cu), dim3(256), lds, 0, dimg, dout, reps);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("NT=%d %s coefficients in %s, %d wave(s)/SIMD: %7.1f ns per product and wave -> %7.1f ns per 16 columns\n", NT, SPLIT ? "split" : "chain", REGS ? "registers" : "LDS      ",
           wgs_per_cu, ms * 1e6 / reps, ms * 1e6 / reps * 4 / wgs_per_cu);
    return 0;
}
int main()
{
    constexpr int NT = 6;
    std::vector<double> img(NT * 320, 1e-4);
    double *dimg, *dout;
    CK(hipMalloc(&dimg, img.size() * 8));
    CK(hipMalloc(&dout, 256 * 8 * 256 * 8));
    CK(hipMemcpy(dimg, img.data(), img.size() * 8, hipMemcpyHostToDevice));
    for (int w : {1, 2}) {
        run<NT, false>(dimg, dout, w);
        run<NT, true>(dimg, dout, w);
        run<NT, false, true>(dimg, dout, w);
    }
    return 0;
}
