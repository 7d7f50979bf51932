// Bare-product probe for a columns-in-lanes layout of the 4 x 4 x 6 Kronecker structure (cnot3), the candidate successor of
// the quad layout (probes/t4q_issue_probe.hip): lane = (fast index 0..3) x (16 columns); the middle and the slow subsystem
// index live in REGISTERS (4 x 6 = 24 per vector and 16 columns), so that
//   * the fast coupling + the diagonal is one v_mfma_f64_4x4x4 per register (as in the quad layout),
//   * the middle and slow couplings are v_fma_f64 between register neighbours with SCALAR coefficients (the coefficient does not
//     depend on the lane) -- no v_mov_b32_dpp at all.
// 24 registers per vector are too many for the ~10 live vectors of the backward sweep, so the slow index is split over SPLIT waves
// of one workgroup (24 / SPLIT registers per vector and wave); the couplings across the split travel through LDS: every wave posts
// its boundary registers after a product, one barrier, the neighbours read them at the start of the next product.
// G workgroups (= 16-column slabs) per CU.  In-place Horner recurrence Y <- A + S Y like the other probes.
// Prints cycles per product round and CU, and CU-cycles per column and product (quad layout: 22.8 at 3 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Coef { double m[5]; double s[8]; };

template <int SPLIT, bool EXCH>
__global__ __launch_bounds__(64 * SPLIT) void k_probe(const double* img, double* out, int reps, Coef c)
{
    constexpr int SL = 6 / SPLIT, R = 4 * SL;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* buf = (double*)smem;                                  // [2 parities][SPLIT waves][2 sides][4 mid][64 lanes]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int PAR = SPLIT * 2 * 4 * 64;
    for (int i = threadIdx.x; i < 2 * PAR; i += blockDim.x) buf[i] = 1e-3;
    double A[R], Y[R], cr[R];
    for (int i = 0; i < R; ++i) A[i] = 1e-3 * (lane + i + wave), Y[i] = A[i], cr[i] = img[(wave * R + i) * 64 + lane];
    __syncthreads();
    for (int r = 0; r < reps; ++r) {
        const double* rd = buf + (r & 1) * PAR;
        double* wr = buf + ((r & 1) ^ 1) * PAR;
        double lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
        if (SPLIT > 1 && EXCH) {
            if (wave > 0)
#pragma unroll
                for (int mid = 0; mid < 4; ++mid) lo[mid] = rd[(((wave - 1) * 2 + 1) * 4 + mid) * 64 + lane];
            if (wave < SPLIT - 1)
#pragma unroll
                for (int mid = 0; mid < 4; ++mid) hi[mid] = rd[(((wave + 1) * 2 + 0) * 4 + mid) * 64 + lane];
        }
        double N[R];
#pragma unroll
        for (int sl = 0; sl < SL; ++sl)
#pragma unroll
            for (int mid = 0; mid < 4; ++mid) {
                const int i = sl * 4 + mid;
                double acc = __builtin_amdgcn_mfma_f64_4x4x4f64(cr[i], Y[i], A[i], 0, 0, 0);
                if (mid > 0) acc = fma(c.m[mid], Y[i - 1], acc);
                if (mid < 3) acc = fma(c.m[mid + 1], Y[i + 1], acc);
                const int s = wave * SL + sl;
                if (sl > 0) acc = fma(c.s[s], Y[i - 4], acc);
                else if (SPLIT > 1) acc = fma(c.s[s], lo[mid], acc);         // (coefficient 0 on the outermost wave)
                if (sl < SL - 1) acc = fma(c.s[s + 1], Y[i + 4], acc);
                else if (SPLIT > 1) acc = fma(c.s[s + 1], hi[mid], acc);
                N[i] = acc;
            }
#pragma unroll
        for (int i = 0; i < R; ++i) Y[i] = N[i];
        if (SPLIT > 1 && EXCH) {
            if (wave > 0)
#pragma unroll
                for (int mid = 0; mid < 4; ++mid) wr[((wave * 2 + 0) * 4 + mid) * 64 + lane] = Y[mid];
            if (wave < SPLIT - 1)
#pragma unroll
                for (int mid = 0; mid < 4; ++mid) wr[((wave * 2 + 1) * 4 + mid) * 64 + lane] = Y[(SL - 1) * 4 + mid];
            __syncthreads();
        }
        asm volatile("" ::: "memory");
    }
    double s = 0;
    for (int i = 0; i < R; ++i) s += Y[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SPLIT, bool EXCH>
int run(const double* dimg, double* dout, int G)
{
    // LDS request sized so that exactly G workgroups fit a CU (160 KB): 256 G workgroups are then G per CU
    const size_t lds = (size_t)(160 * 1024 / G) & ~(size_t)1023;
    const int reps = 20000;
    Coef c;
    for (int i = 0; i < 5; ++i) c.m[i] = (i >= 1 && i <= 3) ? 1e-4 * i : 0.0;
    for (int i = 0; i < 8; ++i) c.s[i] = (i >= 1 && i <= 5) ? 2e-4 * i : 0.0;
    CK(hipFuncSetAttribute((const void*)k_probe<SPLIT, EXCH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_probe<SPLIT, EXCH>), dim3(256 * G), dim3(64 * SPLIT), lds, 0, dimg, dout, 10, c);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_probe<SPLIT, EXCH>), dim3(256 * G), dim3(64 * SPLIT), lds, 0, dimg, dout, reps, c);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double clk = ms * 1e6 / reps * 2.4;
    printf("split %d%s, %d slabs/CU (%4.1f waves/SIMD, %2d registers per vector and wave): %7.0f clk per product round and CU = "
           "%5.2f CU-clk per column and product\n", SPLIT, EXCH ? "" : " (no exchange)", G, SPLIT * G / 4.0, 24 / SPLIT, clk, clk / (16.0 * G));
    return 0;
}

// Two-wave split in mirrored coordinates (row t = 0 of either wave is the one next to the split, so both waves run the same code),
// CH independent recurrences per wave (the state and the adjoint chain of the backward sweep): the boundary row of a chain is
// computed and posted first, the rest of that chain and the other chain's product hide the LDS round trip; one barrier per round.
template <int CH>
__global__ __launch_bounds__(128) void k_probe2(const double* img, double* out, int reps, Coef c)
{
    constexpr int SL = 3, R = 4 * SL;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* buf = (double*)smem;                                  // [2 parities][CH][2 waves][4 mid][64 lanes]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int PAR = CH * 2 * 4 * 64;
    for (int i = threadIdx.x; i < 2 * PAR; i += blockDim.x) buf[i] = 1e-3;
    double A[CH][R], Y[CH][R], cr[R];
    for (int i = 0; i < R; ++i) {
        cr[i] = img[(wave * R + i) * 64 + lane];
        for (int ch = 0; ch < CH; ++ch) A[ch][i] = 1e-3 * (lane + i + wave + ch), Y[ch][i] = A[ch][i];
    }
    // mirrored slow coefficients: cs[t] couples rows t-1 and t (t = 1, 2), cs[0] couples the two boundary rows
    const double cs0 = c.s[3], cs1 = wave ? c.s[4] : c.s[2], cs2 = wave ? c.s[5] : c.s[1];
    __syncthreads();
    for (int r = 0; r < reps; ++r) {
        const double* rd = buf + (r & 1) * PAR;
        double* wr = buf + ((r & 1) ^ 1) * PAR;
        double nb[CH][4];
#pragma unroll
        for (int ch = 0; ch < CH; ++ch)
#pragma unroll
            for (int mid = 0; mid < 4; ++mid) nb[ch][mid] = rd[((ch * 2 + (wave ^ 1)) * 4 + mid) * 64 + lane];
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) {
            double N[R];
#pragma unroll
            for (int t = 0; t < SL; ++t) {
#pragma unroll
                for (int mid = 0; mid < 4; ++mid) {
                    const int i = t * 4 + mid;
                    double acc = __builtin_amdgcn_mfma_f64_4x4x4f64(cr[i], Y[ch][i], A[ch][i], 0, 0, 0);
                    if (mid > 0) acc = fma(c.m[mid], Y[ch][i - 1], acc);
                    if (mid < 3) acc = fma(c.m[mid + 1], Y[ch][i + 1], acc);
                    if (t == 0) acc = fma(cs0, nb[ch][mid], acc);
                    if (t == 1) acc = fma(cs1, Y[ch][i - 4], acc);
                    if (t == 2) acc = fma(cs2, Y[ch][i - 4], acc);
                    if (t == 0) acc = fma(cs1, Y[ch][i + 4], acc);
                    if (t == 1) acc = fma(cs2, Y[ch][i + 4], acc);
                    N[i] = acc;
                }
                if (t == 0)
#pragma unroll
                    for (int mid = 0; mid < 4; ++mid) wr[((ch * 2 + wave) * 4 + mid) * 64 + lane] = N[mid];
            }
#pragma unroll
            for (int i = 0; i < R; ++i) Y[ch][i] = N[i];
        }
        __syncthreads();
        asm volatile("" ::: "memory");
    }
    double s = 0;
    for (int ch = 0; ch < CH; ++ch)
        for (int i = 0; i < R; ++i) s += Y[ch][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CH>
int run2(const double* dimg, double* dout, int G)
{
    const size_t lds = (size_t)(160 * 1024 / G) & ~(size_t)1023;
    const int reps = 20000;
    Coef c;
    for (int i = 0; i < 5; ++i) c.m[i] = (i >= 1 && i <= 3) ? 1e-4 * i : 0.0;
    for (int i = 0; i < 8; ++i) c.s[i] = (i >= 1 && i <= 5) ? 2e-4 * i : 0.0;
    CK(hipFuncSetAttribute((const void*)k_probe2<CH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_probe2<CH>), dim3(256 * G), dim3(128), lds, 0, dimg, dout, 10, c);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_probe2<CH>), dim3(256 * G), dim3(128), lds, 0, dimg, dout, reps, c);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double clk = ms * 1e6 / reps * 2.4;
    printf("split 2 mirrored, %d chain(s) per wave, boundary row posted first, %d slabs/CU (%3.1f waves/SIMD): %7.0f clk per round and CU = "
           "%5.2f CU-clk per column and product\n", CH, G, 2 * G / 4.0, clk, clk / (16.0 * G * CH));
    return 0;
}

int main()
{
    std::vector<double> img(24 * 64, 1e-4);
    double *dimg, *dout;
    CK(hipMalloc(&dimg, img.size() * 8));
    CK(hipMalloc(&dout, (size_t)256 * 16 * 192 * 8));
    CK(hipMemcpy(dimg, img.data(), img.size() * 8, hipMemcpyHostToDevice));
    printf("# quad layout for comparison (profiles/r02_issue_probes.txt): 1093 clk per round of 48 columns = 22.8 (3 waves/SIMD), "
           "1416 / 64 = 22.1 (4 waves/SIMD)\n");
    for (int G : {4, 8, 12}) run<1, true>(dimg, dout, G);
    for (int G : {2, 4, 6, 8}) run<2, true>(dimg, dout, G);
    for (int G : {4, 8}) run<3, true>(dimg, dout, G);
    for (int G : {2, 4, 8}) run<6, true>(dimg, dout, G);
    for (int G : {4, 6}) run<2, false>(dimg, dout, G);
    for (int G : {4}) run<3, false>(dimg, dout, G);
    for (int G : {2, 4, 6}) run2<1>(dimg, dout, G);
    for (int G : {2, 4, 6}) run2<2>(dimg, dout, G);
    return 0;
}
