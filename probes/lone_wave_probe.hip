// What does a wave that is ALONE on its SIMD pay per instruction, and how much of it is the taken branch of the loop?
// 256 workgroups of WPS x 4 waves; the loop body is UNROLL x 16 independent VALU instructions, the loop itself is not unrolled
// (#pragma unroll 1), so one taken s_cbranch per 16 UNROLL instructions.  MODE 0: v_fmac_f32 (VOP2, 4 bytes), 1: v_fmac_f64,
// 2: v_mov_b32_dpp row_shr:4 (8 bytes), 3: a DEPENDENT chain of v_fmac_f64 (one accumulator), 4: two dependent chains interleaved.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int MODE, int UNROLL, int WPS>
__global__ __launch_bounds__(256 * WPS) void k(double* out, int reps)
{
    const int lane = threadIdx.x;
    double d[16], a[4];
    float f[18];
    int iv[16];
    for (int i = 0; i < 16; ++i) d[i] = 1e-3 * (lane + i), iv[i] = lane + i;
    for (int i = 0; i < 18; ++i) f[i] = 1e-3f * (lane + i);
    for (int i = 0; i < 4; ++i) a[i] = 1.0 + 1e-9 * (lane + i);
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(f[i]) : "v"(f[16]), "v"(f[17]));
                if (MODE == 1) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(d[i]) : "v"(a[i & 3]), "v"(a[(i + 1) & 3]));
                if (MODE == 2) asm volatile("v_mov_b32_dpp %0, %1 row_shr:4 row_mask:0xf bank_mask:0xf" : "=v"(iv[i]) : "v"(lane));
                if (MODE == 3) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(d[0]) : "v"(a[i & 3]), "v"(a[(i + 1) & 3]));
                if (MODE == 4) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(d[i & 1]) : "v"(a[i & 3]), "v"(a[(i + 1) & 3]));
                if (MODE == 5) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(d[i & 3]) : "v"(a[i & 3]), "v"(a[(i + 1) & 3]));
                // the row-lane kernels' product instruction: the broadcast operand through DPP (round 6: is the issue cost of the DPP form higher?)
                if (MODE == 6) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(d[0]) : "v"(a[i & 3]), "v"(a[(i + 1) & 3]));
                if (MODE == 7) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(d[i & 1]) : "v"(a[i & 3]), "v"(a[(i + 1) & 3]));
                if (MODE == 8) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(d[i & 3]) : "v"(a[i & 3]), "v"(a[(i + 1) & 3]));
                if (MODE == 9) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(d[i]) : "v"(a[i & 3]), "v"(a[(i + 1) & 3]));
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += d[i] + f[i] + iv[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE, int UNROLL, int WPS>
int run(double* dout, const char* what)
{
    const int reps = 320000 / UNROLL;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<MODE, UNROLL, WPS>), dim3(256), dim3(256 * WPS), 0, 0, dout, 10);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<MODE, UNROLL, WPS>), dim3(256), dim3(256 * WPS), 0, 0, dout, reps);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-40s unroll %2d, %d wave(s)/SIMD: %6.2f clk per instruction and wave (2.4 GHz), %6.2f per instruction and SIMD\n", what, UNROLL, WPS,
           ms * 1e-3 * 2.4e9 / reps / 16 / UNROLL, ms * 1e-3 * 2.4e9 / reps / 16 / UNROLL / WPS);
    return 0;
}
#define RUN(M, W) if (run<M, 1, 1>(dout, W) || run<M, 2, 1>(dout, W) || run<M, 4, 1>(dout, W) || run<M, 16, 1>(dout, W) || run<M, 1, 2>(dout, W) || run<M, 16, 2>(dout, W) || run<M, 16, 4>(dout, W)) return 1;
int main()
{
    double* dout;
    CK(hipMalloc(&dout, 256 * 8 * 256 * 8));
    RUN(0, "v_fmac_f32, independent")
    RUN(1, "v_fmac_f64, independent")
    RUN(2, "v_mov_b32_dpp, independent")
    RUN(3, "v_fmac_f64, ONE dependent chain")
    RUN(4, "v_fmac_f64, two chains interleaved")
    RUN(5, "v_fmac_f64, four chains interleaved")
    RUN(6, "v_fmac_f64_dpp, ONE dependent chain")
    RUN(7, "v_fmac_f64_dpp, two chains interleaved")
    RUN(8, "v_fmac_f64_dpp, four chains interleaved")
    RUN(9, "v_fmac_f64_dpp, independent")
    return 0;
}
