// Probe: issue cost of the JQ_BW_T4 group pattern (one v_mfma_f64_4x4x4_4b + <=4 DPP FMAs on its result)
//   mode 0: 16 DPP FMAs, one dependent chain            mode 1: 16 DPP FMAs, 4 independent chains interleaved
//   mode 2: 4 x [MFMA(next); 4 dependent FMAs on cur]   mode 3: 2 x [MFMA, MFMA; 8 FMAs, two chains interleaved]
//   mode 4: 16 plain v_fma_f64 independent               mode 5: 4 MFMA 4x4x4 only (independent)
//   mode 6: mode 2 with s_nop 0 between the FMAs (what hipcc emits between inline-asm statements)
#include <hip/hip_runtime.h>
#include <cstdio>
#define F(y, m, x, K) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf" : "+v"(y) : "v"(m), "v"(x))
#define FN(y, m, x, K) asm volatile("s_nop 0\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf" : "+v"(y) : "v"(m), "v"(x))
template <int MODE>
__global__ void bench(const double* __restrict__ X, double* Y, long long* cyc, int iters)
{
    const int lane = threadIdx.x & 63;
    double m = X[lane] * 1e-3, a = X[64 + lane] * 1e-3;
    double x0 = X[lane], x1 = X[64 + lane], x2 = X[128 + lane], x3 = X[192 + lane];
    double y0 = 0, y1 = 0, y2 = 0, y3 = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            F(y0, m, x0, 0); F(y0, m, x1, 1); F(y0, m, x2, 2); F(y0, m, x3, 3); F(y0, m, x0, 4); F(y0, m, x1, 5); F(y0, m, x2, 6); F(y0, m, x3, 7);
            F(y0, m, x0, 8); F(y0, m, x1, 9); F(y0, m, x2, 10); F(y0, m, x3, 11); F(y0, m, x0, 12); F(y0, m, x1, 13); F(y0, m, x2, 14); F(y0, m, x3, 15);
        } else if (MODE == 1) {
            F(y0, m, x0, 0); F(y1, m, x1, 1); F(y2, m, x2, 2); F(y3, m, x3, 3); F(y0, m, x0, 4); F(y1, m, x1, 5); F(y2, m, x2, 6); F(y3, m, x3, 7);
            F(y0, m, x0, 8); F(y1, m, x1, 9); F(y2, m, x2, 10); F(y3, m, x3, 11); F(y0, m, x0, 12); F(y1, m, x1, 13); F(y2, m, x2, 14); F(y3, m, x3, 15);
        } else if (MODE == 2 || MODE == 6) {
#define G(cur, nxt, xn, K0, K1, K2, K3)                                   \
    nxt = __builtin_amdgcn_mfma_f64_4x4x4f64(a, xn, nxt, 0, 0, 0);        \
    if (MODE == 2) { F(cur, m, x0, K0); F(cur, m, x1, K1); F(cur, m, x2, K2); F(cur, m, x3, K3); } \
    else { F(cur, m, x0, K0); FN(cur, m, x1, K1); FN(cur, m, x2, K2); FN(cur, m, x3, K3); }
            G(y0, y1, x1, 0, 1, 2, 3) G(y1, y2, x2, 4, 5, 6, 7) G(y2, y3, x3, 8, 9, 10, 11) G(y3, y0, x0, 12, 13, 14, 15)
        } else if (MODE == 3) {
            y2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x2, y2, 0, 0, 0);
            y3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x3, y3, 0, 0, 0);
            F(y0, m, x0, 0); F(y1, m, x1, 4); F(y0, m, x1, 1); F(y1, m, x2, 5); F(y0, m, x2, 2); F(y1, m, x3, 6); F(y0, m, x3, 3); F(y1, m, x0, 7);
            y0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x0, y0, 0, 0, 0);
            y1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x1, y1, 0, 0, 0);
            F(y2, m, x0, 8); F(y3, m, x1, 12); F(y2, m, x1, 9); F(y3, m, x2, 13); F(y2, m, x2, 10); F(y3, m, x3, 14); F(y2, m, x3, 11); F(y3, m, x0, 15);
        } else if (MODE == 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(y0) : "v"(m), "v"(x0));
                asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(y1) : "v"(m), "v"(x1));
                asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(y2) : "v"(m), "v"(x2));
                asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(y3) : "v"(m), "v"(x3));
            }
        } else if (MODE == 5) {
            y0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x0, y0, 0, 0, 0);
            y1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x1, y1, 0, 0, 0);
            y2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x2, y2, 0, 0, 0);
            y3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x3, y3, 0, 0, 0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    Y[blockIdx.x * blockDim.x + threadIdx.x] = y0 + y1 + y2 + y3;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE>
void run(const char* name, const double* X, double* Y, long long* cyc, int waves_per_cu)
{
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    bench<MODE><<<256, 64 * waves_per_cu>>>(X, Y, cyc, 10);
    hipEventRecord(e0);
    bench<MODE><<<256, 64 * waves_per_cu>>>(X, Y, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s waves/CU %d: %8.1f ns/iter (%.1f cycles @2.4GHz), cyclecounter %.1f /iter\n", name, waves_per_cu, ms * 1e6 / iters,
           ms * 1e6 / iters * 2.4, (double)c / iters);
}
int main()
{
    double *X, *Y;
    long long* cyc;
    hipMalloc(&X, 256 * 8), hipMalloc(&Y, 256 * 512 * 8), hipMalloc(&cyc, 8);
    double h[256];
    for (int i = 0; i < 256; ++i) h[i] = 1e-3 * (i % 7);
    hipMemcpy(X, h, sizeof h, hipMemcpyHostToDevice);
    for (int w : {4, 8}) {
        run<0>("16 DPP fmac, 1 chain", X, Y, cyc, w);
        run<1>("16 DPP fmac, 4 chains", X, Y, cyc, w);
        run<2>("4 x [mfma4x4(next) + 4 dep fmac]", X, Y, cyc, w);
        run<6>("4 x [mfma4x4(next) + 4 dep fmac + s_nop 0]", X, Y, cyc, w);
        run<3>("2 x [2 mfma4x4 + 8 fmac 2 chains]", X, Y, cyc, w);
        run<4>("16 v_fma_f64, 4 chains", X, Y, cyc, w);
        run<5>("4 mfma4x4 independent", X, Y, cyc, w);
    }
    return 0;
}
