// What does a v_fma_f64 cost on gfx950?  Independent chains, several operand forms, 1 .. 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int MODE, int WPS>
__global__ __launch_bounds__(256 * WPS) void k(double* out, int reps, double s0)
{
    const int lane = threadIdx.x;
    double d[16], a[4];
    float f[16];
    for (int i = 0; i < 16; ++i) d[i] = 1e-3 * (lane + i), f[i] = (float)d[i];
    for (int i = 0; i < 4; ++i) a[i] = 1.0 + 1e-9 * (lane + i);
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(d[i]) : "v"(a[i & 3]), "v"(a[(i + 1) & 3]));
            if (MODE == 1) asm volatile("v_fmac_f64 %0, %1, %1" : "+v"(d[i]) : "v"(a[i & 3]));
            if (MODE == 2) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(d[i]) : "s"(s0), "v"(a[i & 3]));
            if (MODE == 3) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(d[i]) : "v"(a[i & 3]), "v"(a[(i + 1) & 3]));
            if (MODE == 4) asm volatile("v_add_f64 %0, %1, %2" : "=v"(d[i]) : "v"(a[i & 3]), "v"(a[(i + 1) & 3]));
            if (MODE == 5) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(f[i]) : "v"(f[(i + 1) & 15]), "v"(f[(i + 2) & 15]));
            if (MODE == 6) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "s"(s0), "v"(a[i & 3]));
            if (MODE == 7) asm volatile("v_fma_f64 %0, %1, 2.0, %0" : "+v"(d[i]) : "v"(a[i & 3]));
            if (MODE == 8) asm volatile("v_mov_b64 %0, %1" : "=v"(d[i]) : "v"(a[i & 3]));
            if (MODE == 9) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(d[i]) : "v"(a[i & 3]), "v"(a[(i + 1) & 3]));
        }
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += d[i] + f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE, int WPS>
int run(double* dout, const char* what)
{
    const int reps = 20000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<MODE, WPS>), dim3(256), dim3(256 * WPS), 0, 0, dout, 10, 1.0000001);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<MODE, WPS>), dim3(256), dim3(256 * WPS), 0, 0, dout, reps, 1.0000001);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s %d wave(s)/SIMD: %5.2f clk per instruction and SIMD (at 2.4 GHz)\n", what, WPS, ms * 1e-3 * 2.4e9 / reps / 16 / WPS);
    return 0;
}
#define RUN(M, W) if (run<M, 1>(dout, W) || run<M, 2>(dout, W) || run<M, 4>(dout, W)) return 1;
int main()
{
    double* dout;
    CK(hipMalloc(&dout, 256 * 8 * 256 * 8));
    RUN(0, "v_fmac_f64 d, a, b")
    RUN(1, "v_fmac_f64 d, a, a")
    RUN(2, "v_fmac_f64 d, s, a")
    RUN(6, "v_fma_f64 d, s, a, d")
    RUN(7, "v_fma_f64 d, a, 2.0, d")
    RUN(3, "v_mul_f64 d, a, b")
    RUN(4, "v_add_f64 d, a, b")
    RUN(5, "v_fmac_f32 d, a, b")
    RUN(8, "v_mov_b64 d, a")
    RUN(9, "v_pk_fma_f32 d, a, b, d")
    return 0;
}
