// Does VALU / LDS work of the SAME wave overlap with its own v_mfma_f64_16x16x4_f64 stream on gfx950?
// One wave per SIMD (like the propagators).  Per loop iteration: 24 MFMAs (6 independent accumulators x 4 dependent
// k-steps) with V extra v_fma_f64 and L extra ds_read_b64 distributed between them.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int V, int L, bool MFMA>
__global__ __launch_bounds__(256, 1) void k(double* out, int iters)
{
    __shared__ double lds[4096];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 1.0 + 1e-9 * i;
    __syncthreads();
    d4 acc[6];
    double b[4];
    for (int i = 0; i < 6; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
    for (int i = 0; i < 4; ++i) b[i] = 1.0 + 1e-3 * (lane + i);
    double v[8];
    for (int i = 0; i < 8; ++i) v[i] = 0.5 + 1e-3 * i;
    double a = 1.0 + 1e-6 * lane;
    double ldsum = 0.0;
    const double* lp = lds + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 24; ++m) {
            if (MFMA) acc[m / 4] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[m & 3], acc[m / 4], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < (V + 23 - m) / 24; ++j) v[(m + j) & 7] = __builtin_fma(v[(m + j) & 7], 0.999999, 1e-9);
#pragma unroll
            for (int j = 0; j < (L + 23 - m) / 24; ++j) ldsum += lp[((m * 3 + j) & 63) * 64];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    double s = ldsum;
    for (int i = 0; i < 6; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// 32-bit integer VALU ops and 64-bit moves next to the MFMA stream
template <int W, int KIND, bool MFMA>
__global__ __launch_bounds__(256, 1) void k2(double* out, int iters)
{
    const int lane = threadIdx.x & 63;
    d4 acc[6];
    double b[4];
    for (int i = 0; i < 6; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
    for (int i = 0; i < 4; ++i) b[i] = 1.0 + 1e-3 * (lane + i);
    unsigned u[8];
    double w[8];
    for (int i = 0; i < 8; ++i) { u[i] = lane + i; w[i] = lane * 0.5 + i; }
    double a = 1.0 + 1e-6 * lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 24; ++m) {
            if (MFMA) acc[m / 4] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[m & 3], acc[m / 4], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < (W + 23 - m) / 24; ++j) {
                if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[(m + j) & 7]) : "v"(u[(m + j + 1) & 7]));
                if (KIND == 1) asm volatile("v_mov_b64 %0, %1" : "=v"(w[(m + j) & 7]) : "v"(w[(m + j + 3) & 7]));
                if (KIND == 2) asm volatile("v_accvgpr_write_b32 a[0], %0\n v_accvgpr_read_b32 %0, a[0]" : "+v"(u[(m + j) & 7]) : : "a0");
                if (KIND == 3) asm volatile("v_add_f64 %0, %0, %1" : "+v"(w[(m + j) & 7]) : "v"(w[(m + j + 3) & 7]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    double s = 0;
    for (int i = 0; i < 6; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += u[i] + w[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int W, int KIND, bool MFMA>
static void run2(const char* name, double* d, int iters)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k2<W, KIND, MFMA>), dim3(256), dim3(256), 0, 0, d, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-46s %8.2f ms  %7.0f cycles/iteration\n", name, ms, ms * 1e-3 * 2.4e9 / iters);
}

template <int V, int L, bool MFMA>
static void run(const char* name, double* d, int iters)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<V, L, MFMA>), dim3(256), dim3(256), 0, 0, d, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    // cycles per iteration at 2.4 GHz
    printf("%-46s %8.2f ms  %7.0f cycles/iteration (24 MFMA = 1536)\n", name, ms, ms * 1e-3 * 2.4e9 / iters);
}
int main()
{
    double* d; (void)hipMalloc(&d, 256 * 256 * 8);
    const int it = 200000;
    run<0, 0, true>("24 MFMA", d, it);
    run<48, 0, true>("24 MFMA + 48 v_fma_f64", d, it);
    run<96, 0, true>("24 MFMA + 96 v_fma_f64", d, it);
    run<192, 0, true>("24 MFMA + 192 v_fma_f64", d, it);
    run<0, 36, true>("24 MFMA + 36 ds_read_b64", d, it);
    run<96, 36, true>("24 MFMA + 96 v_fma_f64 + 36 ds_read_b64", d, it);
    run<96, 0, false>("96 v_fma_f64 alone", d, it);
    run<192, 0, false>("192 v_fma_f64 alone", d, it);
    run<0, 36, false>("36 ds_read_b64 alone", d, it);
    run2<96, 0, true>("24 MFMA + 96 v_add_u32", d, it);
    run2<96, 0, false>("96 v_add_u32 alone", d, it);
    run2<96, 1, true>("24 MFMA + 96 v_mov_b64", d, it);
    run2<96, 1, false>("96 v_mov_b64 alone", d, it);
    run2<48, 2, true>("24 MFMA + 48 x (accvgpr write+read)", d, it);
    run2<48, 2, false>("48 x (accvgpr write+read) alone", d, it);
    run2<96, 3, true>("24 MFMA + 96 v_add_f64", d, it);
    return 0;
}
