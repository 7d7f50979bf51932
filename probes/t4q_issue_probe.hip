// Issue-cost probe for the quad-layout product (probes/t4q_probe.hip) at 1 .. 4 waves per SIMD: the in-place Horner
// recurrence Y <- A + S Y with parts of the product switched off, to see what each instruction class costs when the
// SIMD has other waves to issue from.  MODE bits: 1 = MFMA, 2 = lane shifts (v_mov_b32_dpp), 4 = (i, i+-4) FMAs,
// 8 = (i, i+-16) FMAs.  Cycles are per product and SIMD (all waves of the SIMD do one product each).
// 16 (round 3) = the middle-subsystem coupling on the matrix pipe INSTEAD of lane shifts + FMAs: v_mfma_f64_4x4x4_4b always
// contracts the index in the lane bits [5:4] and never moves its block index (bits [3:2]), so the (i, i+-4) coupling needs the
// state with the middle index in [5:4]: one MFMA with the state in the A slot and the identity in B transposes it (fast index
// <-> column index ... in the layout (fast, column-block, middle) of DESIGN.md section 9), a second one applies the 4 x 4
// coupling matrix and transposes back.  Per block: 3 MFMAs + 2 slow-coupling FMAs instead of 1 MFMA + 4 v_mov_b32_dpp + 4 FMAs.
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
// UNR (round 3, late): products per loop iteration.  With one product per iteration the loop's taken branch costs the SIMD 30 - 80
// cycles (probes/lone_wave_probe.hip) -- the figures of rounds 1 - 3 (354 cycles per product at four waves per SIMD, "6.7 cycles per
// fp64 FMA") include it.
template <int NT, int MODE, int WPS, int UNR = 1>
__global__ __launch_bounds__(256 * WPS) void k_probe(const double* img, double* out, int reps)
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
    for (int mt = 0; mt < NT; ++mt)
        for (int k = 0; k < 5; ++k) cr[mt][k] = M[mt * 320 + k * 64];
#pragma unroll 1
    for (int r = 0; r < reps; r += UNR) {
#pragma unroll
      for (int un = 0; un < UNR; ++un) {
        double xold = 0.0;
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) {
            const double x = Y[mt];
            double acc = A[mt];
            if (MODE & 1) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(cr[mt][0], x, acc, 0, 0, 0);
            if (MODE & 16) {      // transposition (state as the A operand, identity as B), then the coupling matrix as B
                const double xt = __builtin_amdgcn_mfma_f64_4x4x4f64(x, cr[mt][1], 0.0, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_4x4x4f64(xt, cr[mt][2], acc, 0, 0, 0);
            }
            double xdn = x, xup = x;
            if (MODE & 2) {
                xdn = row_shift<0x114>(x);
                xup = row_shift<0x104>(x);
            }
            if (MODE & 4) {
                acc = fma(cr[mt][1], xdn, acc);
                acc = fma(cr[mt][2], xup, acc);
            } else if (MODE & 2) {
                asm volatile("" :: "v"(xdn), "v"(xup));
            }
            if (MODE & 8) {
                if (mt > 0) acc = fma(cr[mt][3], xold, acc);
                if (mt + 1 < NT) acc = fma(cr[mt][4], Y[mt + 1], acc);
            }
            xold = x;
            Y[mt] = acc;
        }
        asm volatile("" ::: "memory");
      }
    }
    double s = 0;
    for (int i = 0; i < NT; ++i) s += Y[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NT, int MODE, int WPS, int UNR = 1>
int run(const double* dimg, double* dout)
{
    const size_t lds = NT * 320 * 8;
    const int reps = 20000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_probe<NT, MODE, WPS, UNR>), dim3(256), dim3(256 * WPS), lds, 0, dimg, dout, 16);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_probe<NT, MODE, WPS, UNR>), dim3(256), dim3(256 * WPS), lds, 0, dimg, dout, reps);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const int nm = ((MODE & 1) ? NT : 0) + ((MODE & 16) ? 2 * NT : 0), nmov = (MODE & 2) ? 4 * NT : 0, nf = ((MODE & 4) ? 2 * NT : 0) + ((MODE & 8) ? 2 * NT - 2 : 0);
    const double ns = ms * 1e6 / reps;
    if (UNR > 1) printf("[%d products per loop iteration] ", UNR);
    printf("mode %2d (%d MFMA, %2d dpp mov, %2d FMA per product) %d wave(s)/SIMD: %7.1f ns = %6.0f clk per product round; per wave %6.0f clk (issue model 16/4/4: %4d)\n",
           MODE, nm, nmov, nf, WPS, ns, ns * 2.4, ns * 2.4 / WPS, 16 * nm + 4 * nmov + 4 * nf);
    return 0;
}
template <int MODE>
int runw(const double* dimg, double* dout)
{
    if (run<6, MODE, 1>(dimg, dout)) return 1;
    if (run<6, MODE, 2>(dimg, dout)) return 1;
    if (run<6, MODE, 3>(dimg, dout)) return 1;
    if (run<6, MODE, 4>(dimg, dout)) return 1;
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
    runw<15>(dimg, dout);
    runw<1>(dimg, dout);
    runw<2>(dimg, dout);
    runw<12>(dimg, dout);
    runw<13>(dimg, dout);
    runw<3>(dimg, dout);
    runw<14>(dimg, dout);
    runw<25>(dimg, dout);      // 1 + 8 + 16: the product with the middle coupling on two more MFMAs per block (no shifts)
    // the same without the loop branch per product
    run<6, 15, 1, 8>(dimg, dout), run<6, 15, 2, 8>(dimg, dout), run<6, 15, 3, 8>(dimg, dout), run<6, 15, 4, 8>(dimg, dout);
    run<6, 1, 1, 8>(dimg, dout), run<6, 1, 4, 8>(dimg, dout);
    run<6, 2, 1, 8>(dimg, dout), run<6, 2, 4, 8>(dimg, dout);
    run<6, 12, 1, 8>(dimg, dout), run<6, 12, 4, 8>(dimg, dout);
    run<6, 25, 3, 8>(dimg, dout), run<6, 25, 4, 8>(dimg, dout);
    return 0;
}
