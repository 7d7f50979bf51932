// Probe: f64 MFMA 16x16x4 operand/result layout, C->B chaining identity and issue rate on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// Y(96x16) = M(96x96) * X(96x16), one wave.  M given as A-fragments image: tile (mt,kk) -> 64 doubles, lane l holds M[16*mt + (l&15)][4*kk + (l>>4)]
// X given in "B/C register layout": reg r of tile t, lane l holds X[16*t + 4*r + (l>>4)][l&15]  (kk = 4*t + r)
__global__ void chain_kernel(const double* __restrict__ Mimg, const double* __restrict__ M2img, const double* __restrict__ Xin, double* __restrict__ Yout, int reps)
{
    int l = threadIdx.x;
    double x[24];
    for (int k = 0; k < 24; k++) x[k] = Xin[(4 * k + (l >> 4)) * 16 + (l & 15)];  // row-major X[row][col]
    d4 acc[6];
    for (int rep = 0; rep < reps; rep++) {
        const double* Mi = (rep & 1) ? M2img : Mimg;
#pragma unroll
        for (int mt = 0; mt < 6; mt++) acc[mt] = (d4){0, 0, 0, 0};
#pragma unroll
        for (int kk = 0; kk < 24; kk++) {
#pragma unroll
            for (int mt = 0; mt < 6; mt++) {
                double a = Mi[(mt * 24 + kk) * 64 + l];
                acc[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, x[kk], acc[mt], 0, 0, 0);
            }
        }
        // chain: result registers become next B operands without any movement
#pragma unroll
        for (int mt = 0; mt < 6; mt++) {
            x[4 * mt + 0] = acc[mt][0]; x[4 * mt + 1] = acc[mt][1]; x[4 * mt + 2] = acc[mt][2]; x[4 * mt + 3] = acc[mt][3];
        }
    }
    for (int k = 0; k < 24; k++) Yout[(4 * k + (l >> 4)) * 16 + (l & 15)] = x[k];
}

// throughput: nacc independent accumulators, n MFMAs each wave; all operands in registers
template <int NACC>
__global__ void rate_kernel(double* out, int iters)
{
    double a = threadIdx.x * 0.001, b = 1.0 + threadIdx.x * 0.002;
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = (d4){0, 0, 0, 0};
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0);
}

int main()
{
    const int n = 96;
    std::vector<double> M(n * n), M2(n * n), X(n * 16), Y(n * 16), Yref(n * 16), T(n * 16);
    srand(1);
    for (auto& v : M) v = (rand() / (double)RAND_MAX - 0.5);
    for (auto& v : M2) v = (rand() / (double)RAND_MAX - 0.5);
    for (auto& v : X) v = (rand() / (double)RAND_MAX - 0.5);
    auto img = [&](const std::vector<double>& A) {
        std::vector<double> I(144 * 64);
        for (int mt = 0; mt < 6; mt++) for (int kk = 0; kk < 24; kk++) for (int l = 0; l < 64; l++)
            I[(mt * 24 + kk) * 64 + l] = A[(16 * mt + (l & 15)) * n + 4 * kk + (l >> 4)];  // row-major A[i][k]
        return I;
    };
    auto I1 = img(M), I2 = img(M2);
    double *dI1, *dI2, *dX, *dY;
    CK(hipMalloc(&dI1, I1.size() * 8)); CK(hipMalloc(&dI2, I2.size() * 8)); CK(hipMalloc(&dX, X.size() * 8)); CK(hipMalloc(&dY, Y.size() * 8));
    CK(hipMemcpy(dI1, I1.data(), I1.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dI2, I2.data(), I2.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice));
    for (int reps = 1; reps <= 3; reps++) {
        hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(64), 0, 0, dI1, dI2, dX, dY, reps);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(Y.data(), dY, Y.size() * 8, hipMemcpyDeviceToHost));
        Yref = X;
        for (int rep = 0; rep < reps; rep++) {
            const auto& A = (rep & 1) ? M2 : M;
            for (int i = 0; i < n; i++) for (int c = 0; c < 16; c++) { double s = 0; for (int k = 0; k < n; k++) s += A[i * n + k] * Yref[k * 16 + c]; T[i * 16 + c] = s; }
            Yref = T;
        }
        double err = 0, nrm = 0;
        for (int i = 0; i < n * 16; i++) { err = fmax(err, fabs(Y[i] - Yref[i])); nrm = fmax(nrm, fabs(Yref[i])); }
        printf("chain reps=%d max err %.3e (max |ref| %.3e)\n", reps, err, nrm);
    }
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs %d clock %d kHz LDS/block %zu\n", prop.name, prop.multiProcessorCount, prop.clockRate, prop.sharedMemPerBlock);
    double* dout; CK(hipMalloc(&dout, 1024 * 256 * 8));
    int iters = 20000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](auto kern, int nacc, int blocks, int threads) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, dout, 100);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, dout, iters);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double cyc; CK(hipMemcpy(&cyc, dout, 8, hipMemcpyDeviceToHost));
        double nm = (double)iters * nacc;
        double flops = nm * 2048.0 * (threads / 64) * blocks;
        printf("nacc=%d blocks=%d threads=%d: %.1f clock64 ticks/MFMA (wave0), %.3f ms, %.2f TFLOP/s\n", nacc, blocks, threads, cyc / nm, ms, flops / ms * 1e-9);
    };
    run(rate_kernel<1>, 1, 1, 64);
    run(rate_kernel<2>, 2, 1, 64);
    run(rate_kernel<6>, 6, 1, 64);
    run(rate_kernel<6>, 6, 256, 256);
    run(rate_kernel<6>, 6, 512, 256);
    run(rate_kernel<6>, 6, 256, 512);
    return 0;
}
