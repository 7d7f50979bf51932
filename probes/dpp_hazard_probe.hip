// Which DPP hazards are real on gfx950 for v_fmac_f64_dpp row_newbcast?  (no interlock => wrong numbers)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ void k(const double* __restrict__ M, const double* __restrict__ X, double* Y)
{
    const int lane = threadIdx.x;
    double m = M[lane & 15], x = X[lane], x2 = X[64 + lane];
    double ya = 0, yb = 0, yc = 0, yd = 0, ye = 0;
    // (a) VALU write of src0 right before: expect (m+1)[5]*x
    double ma = m;
    asm volatile("v_add_f64 %0, %0, 1.0\n v_fmac_f64_dpp %1, %0, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(ma), "+v"(ya) : "v"(x));
    // (b) VALU write of src1 right before: expect m[5]*(x+1)
    double xb = x;
    asm volatile("v_add_f64 %0, %0, 1.0\n v_fmac_f64_dpp %1, %2, %0 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(xb), "+v"(yb) : "v"(m));
    // (c) back-to-back accumulation into the same register: expect m[5]*x + m[6]*x2 + 1
    asm volatile("v_mov_b64 %0, 1.0\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, %1, %3 row_newbcast:6 row_mask:0xf bank_mask:0xf"
                 : "=&v"(yc) : "v"(m), "v"(x), "v"(x2));
    // (d) EXEC write right before (all lanes re-enabled): expect m[5]*x in every lane
    asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 exec, 0xffff\n s_mov_b64 exec, s[20:21]\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf"
                 : "+v"(yd) : "v"(m), "v"(x) : "s20", "s21");
    // (e) VMEM-loaded src0 consumed right after the wait (no VALU in between): control, must be right
    asm volatile("s_nop 4\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(ye) : "v"(m), "v"(x));
    Y[lane] = ya; Y[64 + lane] = yb; Y[128 + lane] = yc; Y[192 + lane] = yd; Y[256 + lane] = ye;
}
int main()
{
    double hM[16], hX[128], hY[320];
    for (int i = 0; i < 16; ++i) hM[i] = 1.0 + i * 0.25;
    for (int i = 0; i < 128; ++i) hX[i] = 0.01 * i - 0.3;
    double *dM, *dX, *dY;
    (void)hipMalloc(&dM, sizeof hM); (void)hipMalloc(&dX, sizeof hX); (void)hipMalloc(&dY, sizeof hY);
    (void)hipMemcpy(dM, hM, sizeof hM, hipMemcpyHostToDevice); (void)hipMemcpy(dX, hX, sizeof hX, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dM, dX, dY);
    (void)hipMemcpy(hY, dY, sizeof hY, hipMemcpyDeviceToHost);
    double e[5] = {0, 0, 0, 0, 0};
    for (int l = 0; l < 64; ++l) {
        e[0] = fmax(e[0], fabs((hM[5] + 1.0) * hX[l] - hY[l]));
        e[1] = fmax(e[1], fabs(hM[5] * (hX[l] + 1.0) - hY[64 + l]));
        e[2] = fmax(e[2], fabs(1.0 + hM[5] * hX[l] + hM[6] * hX[64 + l] - hY[128 + l]));
        e[3] = fmax(e[3], fabs(hM[5] * hX[l] - hY[192 + l]));
        e[4] = fmax(e[4], fabs(hM[5] * hX[l] - hY[256 + l]));
    }
    printf("(a) VALU->src0 (DPP operand) no nop: err %g\n(b) VALU->src1 no nop: err %g\n(c) back-to-back accumulate: err %g\n(d) EXEC write no nop: err %g\n(e) control: err %g\n",
           e[0], e[1], e[2], e[3], e[4]);
    return 0;
}
