#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
template <int K>
__device__ __forceinline__ void fmab(double& y, double m, double x)
{
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(y) : "v"(m), "v"(x), "n"(K));
}
template <int K>
__device__ __forceinline__ void fmabn(double& y, double m, double x)
{
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(y) : "v"(m), "v"(x), "n"(K));
}
extern "C" __device__ long long jq_update_dpp_i64(long long old, long long src, int ctrl, int rm, int bm, bool bc) __asm("llvm.amdgcn.update.dpp.i64");
template <int K>
__device__ __forceinline__ void fmabi(double& y, double m, double x)
{
    y = fma(__builtin_bit_cast(double, jq_update_dpp_i64(0, __builtin_bit_cast(long long, m), 0x150 + K, 0xf, 0xf, true)), x, y);
}
// mode 0: dpp fmac chain; mode 1: plain fma with scalar-ish operand (per-lane copy)
template <int MODE>
__global__ void bench(const double* __restrict__ M, const double* __restrict__ X, double* Y, int iters)
{
    const int lane = threadIdx.x & 63;
    double m = M[lane & 15];
    double x0 = X[lane], x1 = X[64 + lane], x2 = X[128 + lane], x3 = X[192 + lane];
    double y0 = 0, y1 = 0, y2 = 0, y3 = 0;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            fmab<0>(y0, m, x0); fmab<1>(y0, m, x1); fmab<2>(y0, m, x2); fmab<3>(y0, m, x3);
            fmab<4>(y1, m, x0); fmab<5>(y1, m, x1); fmab<6>(y1, m, x2); fmab<7>(y1, m, x3);
            fmab<8>(y2, m, x0); fmab<9>(y2, m, x1); fmab<10>(y2, m, x2); fmab<11>(y2, m, x3);
            fmab<12>(y3, m, x0); fmab<13>(y3, m, x1); fmab<14>(y3, m, x2); fmab<15>(y3, m, x3);
        } else if (MODE == 2) {
            fmabn<0>(y0, m, x0); fmabn<1>(y0, m, x1); fmabn<2>(y0, m, x2); fmabn<3>(y0, m, x3);
            fmabn<4>(y1, m, x0); fmabn<5>(y1, m, x1); fmabn<6>(y1, m, x2); fmabn<7>(y1, m, x3);
            fmabn<8>(y2, m, x0); fmabn<9>(y2, m, x1); fmabn<10>(y2, m, x2); fmabn<11>(y2, m, x3);
            fmabn<12>(y3, m, x0); fmabn<13>(y3, m, x1); fmabn<14>(y3, m, x2); fmabn<15>(y3, m, x3);
        } else if (MODE == 3) {
            fmabi<0>(y0, m, x0); fmabi<1>(y0, m, x1); fmabi<2>(y0, m, x2); fmabi<3>(y0, m, x3);
            fmabi<4>(y1, m, x0); fmabi<5>(y1, m, x1); fmabi<6>(y1, m, x2); fmabi<7>(y1, m, x3);
            fmabi<8>(y2, m, x0); fmabi<9>(y2, m, x1); fmabi<10>(y2, m, x2); fmabi<11>(y2, m, x3);
            fmabi<12>(y3, m, x0); fmabi<13>(y3, m, x1); fmabi<14>(y3, m, x2); fmabi<15>(y3, m, x3);
        } else {
            y0 = fma(m, x0, y0); y0 = fma(m, x1, y0); y0 = fma(m, x2, y0); y0 = fma(m, x3, y0);
            y1 = fma(m, x0, y1); y1 = fma(m, x1, y1); y1 = fma(m, x2, y1); y1 = fma(m, x3, y1);
            y2 = fma(m, x0, y2); y2 = fma(m, x1, y2); y2 = fma(m, x2, y2); y2 = fma(m, x3, y2);
            y3 = fma(m, x0, y3); y3 = fma(m, x1, y3); y3 = fma(m, x2, y3); y3 = fma(m, x3, y3);
        }
        // feed back so the loop is not collapsed; keeps values bounded
        x0 = y3 * 1e-3; x1 = y2 * 1e-3; x2 = y1 * 1e-3; x3 = y0 * 1e-3;
    }
    Y[threadIdx.x + blockIdx.x * blockDim.x] = y0 + y1 + y2 + y3;
}
// hazard test: VALU write of m immediately before the DPP read
__global__ void hazard(const double* __restrict__ M, const double* __restrict__ X, double* Y)
{
    const int lane = threadIdx.x;
    double m = M[lane & 15];
    double x0 = X[lane];
    double y0 = 0;
    asm volatile("v_add_f64 %0, %0, 1.0\n v_fmac_f64_dpp %1, %0, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(m), "+v"(y0) : "v"(x0));
    Y[lane] = y0;
}
int main()
{
    double hM[16], hX[256], hY[256];
    for (int i = 0; i < 16; ++i) hM[i] = 1.0 + i * 0.25;
    for (int i = 0; i < 256; ++i) hX[i] = 0.01 * i - 1.0;
    double *dM, *dX, *dY;
    (void)hipMalloc(&dM, sizeof hM); (void)hipMalloc(&dX, sizeof hX); (void)hipMalloc(&dY, 1 << 24);
    (void)hipMemcpy(dM, hM, sizeof hM, hipMemcpyHostToDevice); (void)hipMemcpy(dX, hX, sizeof hX, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(hazard, dim3(1), dim3(64), 0, 0, dM, dX, dY);
    (void)hipMemcpy(hY, dY, 64 * 8, hipMemcpyDeviceToHost);
    double err = 0;
    for (int l = 0; l < 64; ++l) err = fmax(err, fabs((hM[5] + 1.0) * hX[l] - hY[l]));
    printf("hazard test (no nop) max err %g\n", err);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int mode = 0; mode < 4; ++mode)
        for (int waves = 1; waves <= 4; waves *= 4) {
            const int iters = 100000;
            float ms;
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(bench<0>, dim3(256 * 4), dim3(64 * waves), 0, 0, dM, dX, dY, iters);
                else if (mode == 2) hipLaunchKernelGGL(bench<2>, dim3(256 * 4), dim3(64 * waves), 0, 0, dM, dX, dY, iters);
                else if (mode == 3) hipLaunchKernelGGL(bench<3>, dim3(256 * 4), dim3(64 * waves), 0, 0, dM, dX, dY, iters);
                else hipLaunchKernelGGL(bench<1>, dim3(256 * 4), dim3(64 * waves), 0, 0, dM, dX, dY, iters);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            double fl = 2.0 * 16 * 64 * (double)iters * 256 * 4 * waves;
            printf("mode %d waves/block %d: %.2f ms  %.2f TFLOP/s (FMA only)\n", mode, waves, ms, fl / ms * 1e-9);
        }
    return 0;
}
