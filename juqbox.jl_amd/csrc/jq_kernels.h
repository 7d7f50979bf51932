// jq_kernels.h -- gfx950 (MI355X, CDNA4) device code of the Stormer-Verlet forward / discrete-adjoint
// propagator.  Hand-written HIP; no portability layer.  See DESIGN.md for the derivations.
//
// Execution model
//   * one wave (64 lanes) owns one SLAB = 16 state columns (floor(16/N) ensemble samples x N initial
//     conditions) for the whole time loop; a workgroup = 4 waves = 4 slabs, one wave per SIMD.
//   * every state array (u, v, v05, lambda, ...) lives in registers in the C/D layout of
//     v_mfma_f64_16x16x4_f64:  element [mt][r] of lane l  <->  row 16*mt + 4*r + (l>>4), column l&15.
//     That layout IS the B-operand layout of the next product (k-step kk = 4*mt + r), so chained
//     products  Y = M * X  never move state between lanes or through memory.
//   * the time-dependent operators K(t) = H0 + sum_k p_k(t) Hsym_k,  S(t) = sum_k q_k(t) Hanti_k are
//     shared by every column of every sample; they are pre-assembled per time point as MFMA A-fragment
//     tile images ("tile stream", k_stream), and staged through a ring of LDS slots with direct
//     global->LDS DMA (global_load_lds_dwordx4) up to three operators ahead of the MFMAs.
//   * block-band structure: with 16x16 blocks, block (mt,kb) of an operator is stored / multiplied
//     only when |mt-kb| <= BW (BW = NT-1 is the dense case).  Hamiltonians of coupled oscillators in
//     Kronecker ordering are block-banded (cnot3: block-tridiagonal, 64 of 144 tiles), and the band
//     width is a compile-time parameter, so the MFMA loops stay branch-free and fully scheduled.
//   * the per-sample perturbation  H0_s = H0 + eps_s diag(shift)  of the risk-neutral ensemble
//     (src/ipopt_interface.jl:41-44) is applied in registers after each K product.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

typedef double d4 __attribute__((ext_vector_type(4)));

#define JQ_WAVES 4            // waves (slabs) per workgroup
#define JQ_MAXNC 4            // max coupled controls supported by the trace/carry bookkeeping
#define JQ_NTR 5              // trace scalars per control per backward step
#define JQ_STATE_ARRAYS 4     // U, V, MU, NU
#define JQ_STATE_EXTRA 8      // 64-double rows after the arrays: CARRY[0..3], LEAK, spare
#define JQ_MAXSLOTS 2         // LDS operator slots (double buffer)
#define JQ_MAXSCHED 30        // max operator uses per time step (13 + 3*JQ_MAXNC = 25)

// BW == JQ_BW_OD: block band 1 whose off-diagonal 16x16 blocks are DIAGONAL matrices (an operator of the
// slowest subsystem of a Kronecker-ordered Hilbert space, (c +- c') x I_16: cnot3).  Only the diagonal blocks
// are MFMA tiles; the off-diagonal blocks are 16 coefficients each, applied with 4 v_fma_f64 per block in the
// shadow of the MFMAs (mm_od) -- 24 MFMAs + 40 FMAs per product at Ntot = 96 instead of 64 MFMAs.
#define JQ_BW_OD 9
#define JQ_OD_COEFS(NT) (32 * (NT))   // doubles after the tiles: [mt][dir: below, above][g = lane>>4][r]

// BW == JQ_BW_T4: the JQ_BW_OD structure one level finer -- also inside every 16x16 diagonal block the off-diagonal
// 4x4 blocks are DIAGONAL matrices and couple only neighbouring 4-row groups (a Kronecker-ordered Hilbert space whose
// fastest subsystem has 4 levels and whose operators are sums of single-subsystem terms and diagonal couplings:
// cnot3 = 4 x 4 x 6).  A register of a state array (4 rows x 16 columns, row = lane>>4, column = lane&15) is exactly
// the B and the C/D operand of v_mfma_f64_4x4x4_4b_f64 (four 4x4x4 blocks in 16 cycles, profiles/
// r01_mfma_f64_4x4x4_layout.txt), so the product needs ONE such MFMA per 4-row group for its dense 4x4 diagonal block
// (24 x 16 = 384 cycles at Ntot = 96 instead of 24 x 64) and at most four v_fma_f64 per group for the couplings to the
// groups rho-1, rho+1 (same 16-row block) and rho-4, rho+4 (neighbouring blocks), mm_t4.
#define JQ_BW_T4 8
// BW == JQ_BW_T4Q: the same operators, images and staging as JQ_BW_T4 in the QUAD layout for small batches: one wave carries
// four state columns, a register of a state array is a whole 16-row block (lane 16 i + 4 b + j <-> row 16 mt + 4 b + i,
// column j), an Ntot = 96 array is 6 registers (everything stays in VGPRs), ONE v_mfma_f64_4x4x4_4b does the four 4x4
// diagonal blocks of a 16-row block, the couplings (i, i+-4) read the same register shifted by 4 lanes inside each 16-lane
// row (2 x v_mov_b32_dpp row_shr/row_shl -- 64-bit DPP only has row_newbcast) and the couplings (i, i+-16) the
// neighbouring register.  A product takes 209 ns per wave (probes/t4q_probe.hip) where the cooperative kernels need 656
// (one barrier + LDS exchange per product); per column it is slower than JQ_BW_T4 (590 vs 486 ns per 16 columns), so
// large batches stay there.  The four waves of a workgroup share ONE slab of the JQ_BW_T4 state file (wave q = columns
// 4q .. 4q+3): array file, initial / terminal kernels, column tables and operator images are those of the slab kernels.
#define JQ_BW_T4Q 7
#define JQ_T4_AIDX(rho, k, i) (((rho) >> 2) * 64 + 16 * (k) + 4 * ((rho) & 3) + (i))   // element of B_rho[i][k] in the image: per 16-row block the 64
                                      // doubles in the lane order of the quad-layout MFMA (lane 16 k + 4 b + i; a [b][k][i] order gave those reads
                                      // LDS bank conflicts worth 5 % of an evaluation); the slab kernels read elements 16 k + i of a group's slice
#define JQ_T4_TILE 16                  // doubles per 4-row group in the image: the 4x4 diagonal block (the MFMA's A operand
                                      // repeats it in its four column blocks: lane 16k+4b+i reads element 4k+i, an LDS broadcast)
#define JQ_T4_COEFS(NT) (64 * (NT))   // doubles after the 4NT blocks: per 16-row block [g = row in group][r = group][term], terms: couplings
                                      // to the groups r-1, r+1 (same block), to the blocks mt-1, mt+1.  The slab kernels read one double per
                                      // lane (lane 16 g + p holds position p = JQ_T4_CPOS(r, term) of row g), the quad-layout kernels the
                                      // 32-byte record of their row; neither read has LDS bank conflicts (PMC; a [term pair][g][r][2]
                                      // order was tried for the quad reads: no difference there, 2-way conflicts in the slab reads)
#define JQ_T4_CIDX(g, r, t) ((g) * 16 + (r) * 4 + (t))
#define JQ_T4_CPOS(r, t) (4 * (r) + (t))
#define JQ_T4_ELEMS(NT) (4 * (NT) * JQ_T4_TILE + JQ_T4_COEFS(NT))
// trace-image modes of this variant (a.bw_trace[q]): bit 0 diagonal 4x4 blocks present, bit 1 r+-1 terms, bit 2 mt+-1 terms
#define JQ_T4_DIAG 1
#define JQ_T4_RTERMS 2
#define JQ_T4_MTERMS 4

// Stored tiles of an NT x 4NT tile grid: block (mt,kb) is kept when |mt-kb| <= BW and, for SD ("skip
// diagonal": operators like a3 +- a3' of the slowest subsystem, whose diagonal blocks vanish), mt != kb.
__host__ __device__ constexpr bool block_on(int BW, bool SD, int mt, int kb)
{
    return (BW == JQ_BW_OD || BW == JQ_BW_T4 || BW == JQ_BW_T4Q) ? (mt == kb && !SD) : ((mt - kb <= BW) && (kb - mt <= BW) && !(SD && mt == kb));
}
// number of stored tiles (host + device)
__host__ __device__ constexpr int band_tiles(int NT, int BW, bool SD = false)
{
    int n = 0;
    for (int kb = 0; kb < NT; ++kb)
        for (int mt = 0; mt < NT; ++mt)
            if (block_on(BW, SD, mt, kb)) n += 4;
    return n;
}
// first stored k-block of tile row mt (NT if the row is empty)
__host__ __device__ constexpr int first_kb(int NT, int BW, bool SD, int mt)
{
    for (int kb = 0; kb < NT; ++kb)
        if (block_on(BW, SD, mt, kb)) return kb;
    return NT;
}

// Four doubles WITHOUT the register-tuple constraint of d4 (the 16x16x4 MFMA wants its C/D operand in 8 consecutive
// VGPRs; v_mfma_f64_4x4x4 works on single doubles and the in-place products of JQ_BW_T4 then cost v_mov copies).
struct s4 {
    double e[4];
    __device__ __forceinline__ s4() = default;
    __device__ __forceinline__ s4(d4 v) : e{v[0], v[1], v[2], v[3]} {}
    __device__ __forceinline__ double& operator[](int i) { return e[i]; }
    __device__ __forceinline__ double operator[](int i) const { return e[i]; }
    __device__ __forceinline__ s4& operator+=(const s4& o)
    {
        e[0] += o.e[0], e[1] += o.e[1], e[2] += o.e[2], e[3] += o.e[3];
        return *this;
    }
};
__device__ __forceinline__ s4 operator+(const s4& a, const s4& b) { return s4((d4){a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2], a.e[3] + b.e[3]}); }
__device__ __forceinline__ s4 operator-(const s4& a, const s4& b) { return s4((d4){a.e[0] - b.e[0], a.e[1] - b.e[1], a.e[2] - b.e[2], a.e[3] - b.e[3]}); }
__device__ __forceinline__ s4 operator*(const s4& a, const s4& b) { return s4((d4){a.e[0] * b.e[0], a.e[1] * b.e[1], a.e[2] * b.e[2], a.e[3] * b.e[3]}); }
__device__ __forceinline__ s4 operator*(double c, const s4& b) { return s4((d4){c * b.e[0], c * b.e[1], c * b.e[2], c * b.e[3]}); }
__device__ __forceinline__ s4 operator-(const s4& a) { return s4((d4){-a.e[0], -a.e[1], -a.e[2], -a.e[3]}); }
// One double per 16-row block: the JQ_BW_T4Q ("quad") kernels, whose lanes are (row in group, group, column of a quad).
struct s1 {
    double e[1];
    __device__ __forceinline__ s1() = default;
    __device__ __forceinline__ s1(d4 v) : e{v[0]} {}
    __device__ __forceinline__ explicit s1(double v) : e{v} {}
    __device__ __forceinline__ double& operator[](int i) { return e[0]; }
    __device__ __forceinline__ double operator[](int i) const { return e[0]; }
    __device__ __forceinline__ s1& operator+=(const s1& o)
    {
        e[0] += o.e[0];
        return *this;
    }
};
__device__ __forceinline__ s1 operator+(const s1& a, const s1& b) { return s1(a.e[0] + b.e[0]); }
__device__ __forceinline__ s1 operator-(const s1& a, const s1& b) { return s1(a.e[0] - b.e[0]); }
__device__ __forceinline__ s1 operator*(const s1& a, const s1& b) { return s1(a.e[0] * b.e[0]); }
__device__ __forceinline__ s1 operator*(double c, const s1& b) { return s1(c * b.e[0]); }
__device__ __forceinline__ s1 operator-(const s1& a) { return s1(-a.e[0]); }
__device__ __forceinline__ double row_sum(const d4& p) { return (p[0] + p[1]) + (p[2] + p[3]); }
__device__ __forceinline__ double row_sum(const s4& p) { return (p[0] + p[1]) + (p[2] + p[3]); }
__device__ __forceinline__ double row_sum(const s1& p) { return p[0]; }
// row type of the state arrays of this translation unit (one (NT, BW) instantiation per unit, jq_kernel_inst.hip)
#if defined(JQ_BW) && JQ_BW == 7
typedef s1 jq_row;
#define JQ_RL 1               // doubles per row element of an array
#elif defined(JQ_BW) && JQ_BW == 8 && !defined(JQ_ROW_D4)
typedef s4 jq_row;
#define JQ_RL 4
#else
typedef d4 jq_row;
#define JQ_RL 4
#endif

template <int NT>
struct Arr {
    jq_row t[NT];
};

template <int NT>
__device__ __forceinline__ void a_zero(Arr<NT>& a)
{
#pragma unroll
    for (int i = 0; i < NT; ++i) a.t[i] = (d4){0.0, 0.0, 0.0, 0.0};
}
template <int NT>
__device__ __forceinline__ void a_axpy(Arr<NT>& y, double c, const Arr<NT>& x)  // y += c*x
{
#pragma unroll
    for (int i = 0; i < NT; ++i) y.t[i] += c * x.t[i];
}
template <int NT>
__device__ __forceinline__ void a_add(Arr<NT>& y, const Arr<NT>& x)
{
#pragma unroll
    for (int i = 0; i < NT; ++i) y.t[i] += x.t[i];
}
template <int NT>
__device__ __forceinline__ void a_neg(Arr<NT>& y)
{
#pragma unroll
    for (int i = 0; i < NT; ++i) y.t[i] = -y.t[i];
}
template <int NT>
__device__ __forceinline__ double a_dot(const Arr<NT>& x, const Arr<NT>& y)
{
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
#if JQ_RL == 1 && defined(JQ_EXP_DOTFMA)
        s = fma(x.t[i][0], y.t[i][0], s);
#else
        s += row_sum(x.t[i] * y.t[i]);
#endif
    }
    return s;
}
// y += c * tab .* x   (tab: per-row table in LDS, padded to 16*NT rows, stored [block][g][r] so that the four
// values of a lane are one 32-byte read; g = lane>>4).  Quad layout (JQ_RL == 1): g = this lane's offset in a block of the
// table, 4 * (lane >> 4) + ((lane >> 2) & 3).
template <int NT>
__device__ __forceinline__ void a_axpy_rows(Arr<NT>& y, double c, const double* tab, int g, const Arr<NT>& x)
{
#if JQ_RL == 1
#pragma unroll
    for (int i = 0; i < NT; ++i) y.t[i][0] = fma(c * tab[16 * i + g], x.t[i][0], y.t[i][0]);
#else
    const d4* t4 = (const d4*)(tab + 4 * g);
#pragma unroll
    for (int i = 0; i < NT; ++i) y.t[i] += jq_row(c * t4[4 * i]) * x.t[i];
#endif
}
// y += (+-) tab .* x   (a table that carries its coefficient already)
template <int NT, bool NEG>
__device__ __forceinline__ void a_axpy_rows1(Arr<NT>& y, const double* tab, int g, const Arr<NT>& x)
{
#if JQ_RL == 1
#pragma unroll
    for (int i = 0; i < NT; ++i) y.t[i][0] = fma(NEG ? -tab[16 * i + g] : tab[16 * i + g], x.t[i][0], y.t[i][0]);
#else
    const d4* t4 = (const d4*)(tab + 4 * g);
#pragma unroll
    for (int i = 0; i < NT; ++i) y.t[i] += jq_row(NEG ? -t4[4 * i] : t4[4 * i]) * x.t[i];
#endif
}
// sum_rows tab[row] * x[row]^2
template <int NT>
__device__ __forceinline__ double a_wsq(const double* tab, int g, const Arr<NT>& x)
{
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < JQ_RL; ++r) s += tab[16 * i + (JQ_RL == 1 ? g : 4 * g + r)] * (x.t[i][r] * x.t[i][r]);
    return s;
}

// image <-> registers: array image = [4*NT][64] doubles, element kk*64 + lane  (quad layout: `lane` is this lane's offset
// in a block of the slab image, ((lane >> 2) & 3) * 64 + 16 * (lane >> 4) + first column of the quad + (lane & 3))
template <int NT>
__device__ __forceinline__ void a_load(Arr<NT>& a, const double* __restrict__ img, int lane)
{
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < JQ_RL; ++r) a.t[i][r] = img[(4 * i + r) * 64 + lane];
}
template <int NT>
__device__ __forceinline__ void a_store(const Arr<NT>& a, double* __restrict__ img, int lane)
{
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < JQ_RL; ++r) img[(4 * i + r) * 64 + lane] = a.t[i][r];
}

// D = C + M * x over the stored (band) tiles (D may alias C; D must not alias x).
// M: compact LDS tile image in walk order (kk outer, mt inner, band tiles only), 64 doubles per tile;
// `mat` already carries the lane offset.  Tile (mt,kk) lane l holds M[16*mt + (l&15)][4*kk + (l>>4)].
// The A fragments are fetched JQ_PF tiles ahead of their MFMA through a small register FIFO: a
// v_mfma_f64_16x16x4 occupies the matrix pipe for 64 cycles, an LDS read returns in ~100.
#define JQ_PF 4

// BW == JQ_BW_OD:  D = C + blockdiag(M) x (MFMA)  +  diagonal off-diagonal blocks (VALU).
// Row mt's accumulator input  C[mt] + d_below[mt] .* x[mt-1] + d_above[mt] .* x[mt+1]  is computed right after
// the first MFMA of row mt-1 has been issued (it runs in that MFMA's 64-cycle shadow); its coefficients were
// read from LDS one row earlier.  SD: the operator has no diagonal blocks at all -> no MFMA.
template <int NT, bool ZEROC, bool SD>
__device__ __forceinline__ void mm_od(Arr<NT>& D, const Arr<NT>& C, const double* mat, const Arr<NT>& x)
{
    // Alias-safe: D may be the same array as C and/or x (every x row is copied to a d4 before D's row is
    // written; in SSA form these are renames, not moves) -- the Horner recurrence Y <- A + S Y runs in place.
    constexpr int NTILES = SD ? 0 : 4 * NT;
    const int lane = threadIdx.x & 63;
    const double* cf = mat - lane + NTILES * 64 + (lane >> 4) * 4;
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    d4 ca = *(const d4*)(cf), cb = *(const d4*)(cf + 16);        // row 0: below (unused), above
    if constexpr (SD) {
        d4 x_prev = zero, x_cur = x.t[0];
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) {
            const d4 x_next = (mt + 1 < NT) ? x.t[mt + 1] : zero;
            d4 acc = ZEROC ? zero : C.t[mt];
            if (mt > 0) acc += ca * x_prev;
            if (mt + 1 < NT) acc += cb * x_next;
            if (mt + 1 < NT) {
                ca = *(const d4*)(cf + ((mt + 1) * 2 + 0) * 16);
                cb = *(const d4*)(cf + ((mt + 1) * 2 + 1) * 16);
            }
            D.t[mt] = acc;
            x_prev = x_cur;
            x_cur = x_next;
        }
        return;
    } else {
        // (Interleaving the MFMAs of two rows -- two independent accumulator chains -- was measured to make no
        // difference and costs registers: the matrix pipe is not the bottleneck of this variant, DESIGN.md.)
        double f[JQ_PF];
#pragma unroll
        for (int i = 0; i < JQ_PF; ++i)
            if (i < NTILES) f[i] = mat[i * 64];
        d4 x_cur = x.t[0];
        d4 nxt = ZEROC ? zero : C.t[0];
        if (NT > 1) nxt += cb * x.t[1];
        if (NT > 1) {
            ca = *(const d4*)(cf + 2 * 16);
            cb = *(const d4*)(cf + 3 * 16);
        }
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) {
            const d4 x_next = (mt + 1 < NT) ? x.t[mt + 1] : zero;
            d4 acc = nxt;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int kk = 4 * mt + k;
                const double a = f[kk % JQ_PF];
                if (kk + JQ_PF < NTILES) f[kk % JQ_PF] = mat[(kk + JQ_PF) * 64];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, x_cur[k], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (k == 0 && mt + 1 < NT) {
                    // accumulator input of the next row, in the shadow of the MFMA just issued
                    nxt = ZEROC ? zero : C.t[mt + 1];
                    nxt += ca * x_cur;
                    if (mt + 2 < NT) nxt += cb * x.t[mt + 2];
                    if (mt + 2 < NT) {
                        ca = *(const d4*)(cf + ((mt + 2) * 2 + 0) * 16);
                        cb = *(const d4*)(cf + ((mt + 2) * 2 + 1) * 16);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            D.t[mt] = acc;
            x_cur = x_next;
        }
    }
}

// y += sum_k c[lane K_k of this lane's 16-lane row] * x_k  (v_fmac_f64_dpp: DPP costs nothing extra, probes/dpp_fmac_probe.hip)
// as ONE asm statement (between separate statements hipcc puts an s_nop each) with GUARD wait states in front, where
// nothing can be scheduled in between:
//  2: gfx950 does not interlock a VALU write of the DPP operand c with this read, and the register allocator may reload c
//     (v_accvgpr_read, v_mov) right in front of a group's first FMA;
//  6: y is the result of a v_mfma_f64_4x4x4 -- a software hazard (6 wait states before a VALU read) that the compiler's
//     hazard recognizer does not apply to inline asm.
#define JQ_DPPF(y, c, x, k) "v_fmac_f64_dpp " y ", " c ", " x " row_newbcast:" k " row_mask:0xf bank_mask:0xf\n\t"
// The same for the groups A, B of a pair with their FMA chains interleaved (A0 B0 A1 B1 ...): a dependent v_fma_f64 chain
// issues at 4.7 ns per instruction, two independent ones at 3.4 (probes/t4_group_probe.hip).  NA, NB terms; unused x operands
// are passed as copies of c.
#define JQ_PAIR_OPS : "+v"(yA), "+v"(yB) : "v"(c), "v"(xa0), "v"(xa1), "v"(xa2), "v"(xa3), "v"(xb0), "v"(xb1), "v"(xb2), "v"(xb3), \
                      "n"(GUARD - 1), "n"(KA0), "n"(KA1), "n"(KA2), "n"(KA3), "n"(KB0), "n"(KB1), "n"(KB2), "n"(KB3)
template <int GUARD, int NA, int NB, int KA0, int KA1, int KA2, int KA3, int KB0, int KB1, int KB2, int KB3>
__device__ __forceinline__ void fma_rowbcast_pair(double& yA, double& yB, double c, double xa0, double xa1, double xa2, double xa3,
                                                  double xb0, double xb1, double xb2, double xb3)
{
    static_assert(NA >= 1 && NA <= 4 && NB >= 1 && NB <= 4, "");
    if constexpr (NA == 1 && NB == 1)
        asm("s_nop %11\n\t" JQ_DPPF("%0", "%2", "%3", "%12") JQ_DPPF("%1", "%2", "%7", "%16") JQ_PAIR_OPS);
    else if constexpr (NA == 1 && NB == 2)
        asm("s_nop %11\n\t" JQ_DPPF("%0", "%2", "%3", "%12") JQ_DPPF("%1", "%2", "%7", "%16") JQ_DPPF("%1", "%2", "%8", "%17") JQ_PAIR_OPS);
    else if constexpr (NA == 1 && NB == 3)
        asm("s_nop %11\n\t" JQ_DPPF("%0", "%2", "%3", "%12") JQ_DPPF("%1", "%2", "%7", "%16") JQ_DPPF("%1", "%2", "%8", "%17") JQ_DPPF("%1", "%2", "%9", "%18") JQ_PAIR_OPS);
    else if constexpr (NA == 1 && NB == 4)
        asm("s_nop %11\n\t" JQ_DPPF("%0", "%2", "%3", "%12") JQ_DPPF("%1", "%2", "%7", "%16") JQ_DPPF("%1", "%2", "%8", "%17") JQ_DPPF("%1", "%2", "%9", "%18") JQ_DPPF("%1", "%2", "%10", "%19") JQ_PAIR_OPS);
    else if constexpr (NA == 2 && NB == 1)
        asm("s_nop %11\n\t" JQ_DPPF("%0", "%2", "%3", "%12") JQ_DPPF("%1", "%2", "%7", "%16") JQ_DPPF("%0", "%2", "%4", "%13") JQ_PAIR_OPS);
    else if constexpr (NA == 2 && NB == 2)
        asm("s_nop %11\n\t" JQ_DPPF("%0", "%2", "%3", "%12") JQ_DPPF("%1", "%2", "%7", "%16") JQ_DPPF("%0", "%2", "%4", "%13") JQ_DPPF("%1", "%2", "%8", "%17") JQ_PAIR_OPS);
    else if constexpr (NA == 2 && NB == 3)
        asm("s_nop %11\n\t" JQ_DPPF("%0", "%2", "%3", "%12") JQ_DPPF("%1", "%2", "%7", "%16") JQ_DPPF("%0", "%2", "%4", "%13") JQ_DPPF("%1", "%2", "%8", "%17") JQ_DPPF("%1", "%2", "%9", "%18") JQ_PAIR_OPS);
    else if constexpr (NA == 2 && NB == 4)
        asm("s_nop %11\n\t" JQ_DPPF("%0", "%2", "%3", "%12") JQ_DPPF("%1", "%2", "%7", "%16") JQ_DPPF("%0", "%2", "%4", "%13") JQ_DPPF("%1", "%2", "%8", "%17") JQ_DPPF("%1", "%2", "%9", "%18") JQ_DPPF("%1", "%2", "%10", "%19") JQ_PAIR_OPS);
    else if constexpr (NA == 3 && NB == 1)
        asm("s_nop %11\n\t" JQ_DPPF("%0", "%2", "%3", "%12") JQ_DPPF("%1", "%2", "%7", "%16") JQ_DPPF("%0", "%2", "%4", "%13") JQ_DPPF("%0", "%2", "%5", "%14") JQ_PAIR_OPS);
    else if constexpr (NA == 3 && NB == 2)
        asm("s_nop %11\n\t" JQ_DPPF("%0", "%2", "%3", "%12") JQ_DPPF("%1", "%2", "%7", "%16") JQ_DPPF("%0", "%2", "%4", "%13") JQ_DPPF("%1", "%2", "%8", "%17") JQ_DPPF("%0", "%2", "%5", "%14") JQ_PAIR_OPS);
    else if constexpr (NA == 3 && NB == 3)
        asm("s_nop %11\n\t" JQ_DPPF("%0", "%2", "%3", "%12") JQ_DPPF("%1", "%2", "%7", "%16") JQ_DPPF("%0", "%2", "%4", "%13") JQ_DPPF("%1", "%2", "%8", "%17") JQ_DPPF("%0", "%2", "%5", "%14") JQ_DPPF("%1", "%2", "%9", "%18") JQ_PAIR_OPS);
    else if constexpr (NA == 3 && NB == 4)
        asm("s_nop %11\n\t" JQ_DPPF("%0", "%2", "%3", "%12") JQ_DPPF("%1", "%2", "%7", "%16") JQ_DPPF("%0", "%2", "%4", "%13") JQ_DPPF("%1", "%2", "%8", "%17") JQ_DPPF("%0", "%2", "%5", "%14") JQ_DPPF("%1", "%2", "%9", "%18") JQ_DPPF("%1", "%2", "%10", "%19") JQ_PAIR_OPS);
    else if constexpr (NA == 4 && NB == 1)
        asm("s_nop %11\n\t" JQ_DPPF("%0", "%2", "%3", "%12") JQ_DPPF("%1", "%2", "%7", "%16") JQ_DPPF("%0", "%2", "%4", "%13") JQ_DPPF("%0", "%2", "%5", "%14") JQ_DPPF("%0", "%2", "%6", "%15") JQ_PAIR_OPS);
    else if constexpr (NA == 4 && NB == 2)
        asm("s_nop %11\n\t" JQ_DPPF("%0", "%2", "%3", "%12") JQ_DPPF("%1", "%2", "%7", "%16") JQ_DPPF("%0", "%2", "%4", "%13") JQ_DPPF("%1", "%2", "%8", "%17") JQ_DPPF("%0", "%2", "%5", "%14") JQ_DPPF("%0", "%2", "%6", "%15") JQ_PAIR_OPS);
    else if constexpr (NA == 4 && NB == 3)
        asm("s_nop %11\n\t" JQ_DPPF("%0", "%2", "%3", "%12") JQ_DPPF("%1", "%2", "%7", "%16") JQ_DPPF("%0", "%2", "%4", "%13") JQ_DPPF("%1", "%2", "%8", "%17") JQ_DPPF("%0", "%2", "%5", "%14") JQ_DPPF("%1", "%2", "%9", "%18") JQ_DPPF("%0", "%2", "%6", "%15") JQ_PAIR_OPS);
    else if constexpr (NA == 4 && NB == 4)
        asm("s_nop %11\n\t" JQ_DPPF("%0", "%2", "%3", "%12") JQ_DPPF("%1", "%2", "%7", "%16") JQ_DPPF("%0", "%2", "%4", "%13") JQ_DPPF("%1", "%2", "%8", "%17") JQ_DPPF("%0", "%2", "%5", "%14") JQ_DPPF("%1", "%2", "%9", "%18") JQ_DPPF("%0", "%2", "%6", "%15") JQ_DPPF("%1", "%2", "%10", "%19") JQ_PAIR_OPS);
}

// coupling terms of the 4-row groups R (even) and R+1 of a 16-row block: coefficient lane 4 R + term of this lane's row of c.
// xa / xb: {x of the group below in the block, above in the block, of the block below, of the block above} for A / B.
// LO / HI: the block has a neighbour block below / above.  G: guard of the first FMA.
template <int R, bool RT, bool MTM, bool LO, bool HI, int G>
__device__ __forceinline__ void t4_couple_pair(double& accA, double& accB, double c, const double (&xa)[4], const double (&xb)[4])
{
    constexpr auto on = [](int r, int t) constexpr { return t == 0 ? (RT && r > 0) : t == 1 ? (RT && r < 3) : t == 2 ? (MTM && LO) : (MTM && HI); };
    constexpr auto cnt = [on](int r) constexpr { return on(r, 0) + on(r, 1) + on(r, 2) + on(r, 3); };
    // k-th active term of group r (past the end: term 0, unused)
    constexpr auto act = [on](int r, int k) constexpr {
        for (int t = 0; t < 4; ++t)
            if (on(r, t) && k-- == 0) return t;
        return 0;
    };
    constexpr int NA = cnt(R), NB = cnt(R + 1);
    static_assert((NA == 0) == (NB == 0), "");
    if constexpr (NA > 0) {
#define JQ_XA(k) (k < NA ? xa[act(R, k)] : c)
#define JQ_XB(k) (k < NB ? xb[act(R + 1, k)] : c)
        fma_rowbcast_pair<G, NA, NB, JQ_T4_CPOS(R, act(R, 0)), JQ_T4_CPOS(R, act(R, 1)), JQ_T4_CPOS(R, act(R, 2)), JQ_T4_CPOS(R, act(R, 3)),
                          JQ_T4_CPOS(R + 1, act(R + 1, 0)), JQ_T4_CPOS(R + 1, act(R + 1, 1)), JQ_T4_CPOS(R + 1, act(R + 1, 2)),
                          JQ_T4_CPOS(R + 1, act(R + 1, 3))>(
            accA, accB, c, JQ_XA(0), JQ_XA(1), JQ_XA(2), JQ_XA(3), JQ_XB(0), JQ_XB(1), JQ_XB(2), JQ_XB(3));
#undef JQ_XA
#undef JQ_XB
    }
}
template <int R, bool RT, bool MTM, int G>
__device__ __forceinline__ void t4_couple_pair(double& accA, double& accB, double c, const double (&xa)[4], const double (&xb)[4], bool lo,
                                               bool hi)
{
    if (lo && hi) t4_couple_pair<R, RT, MTM, true, true, G>(accA, accB, c, xa, xb);
    else if (lo) t4_couple_pair<R, RT, MTM, true, false, G>(accA, accB, c, xa, xb);
    else if (hi) t4_couple_pair<R, RT, MTM, false, true, G>(accA, accB, c, xa, xb);
    else t4_couple_pair<R, RT, MTM, false, false, G>(accA, accB, c, xa, xb);
}

// BW == JQ_BW_T4 (see the definition above).  Alias-safe (D may be C and/or x): the old values of the last four
// 4-row groups are kept in a rolling window.  MODE: which parts of the image are non-zero (JQ_T4_* bits).
// The 16 coupling coefficients of a 16-row block and a lane row g = lane>>4 (4 groups x 4 terms) sit in the 16 lanes of
// that row of ONE register (one ds_read_b64 per block; the first version read a d4 per group and lane and was bound by
// the LDS port, 20 of 38 cycles per group and wave); the FMAs pick theirs with a DPP row broadcast (2-wait-state
// hazard on the DPP operand: first FMA of every group guarded, scripts/check_dpp_hazard.py checks the final ISA).
template <int NT, bool ZEROC, int MODE>
__device__ __forceinline__ void mm_t4(Arr<NT>& D, const Arr<NT>& C, const double* mat, const Arr<NT>& x)
{
    constexpr int NR = 4 * NT;
    constexpr bool diag = MODE & JQ_T4_DIAG, rt = MODE & JQ_T4_RTERMS, mtm = MODE & JQ_T4_MTERMS;
    const int lane = threadIdx.x & 63;
    const double* cf = mat + NR * JQ_T4_TILE;     // (lane 16 g + p: position p of row g)
    const double* ma = mat - lane + ((lane >> 4) * 16 + (lane & 3));  // this lane's element (k = lane >> 4, i = lane & 3) of every 4x4 block
    double f[JQ_PF];
    double cq[2];
    if constexpr (diag) {
#pragma unroll
        for (int i = 0; i < JQ_PF; ++i)
            if (i < NR) f[i] = ma[JQ_T4_AIDX(i, 0, 0)];
    }
    if constexpr (rt || mtm) {
        cq[0] = cf[0];
        if (NT > 1) cq[1] = cf[64];
    }
    // MFMA first (three-operand: the C element is read in place, no copy), then the tied DPP FMAs on its result, two
    // groups at a time with their chains interleaved.  The MFMAs run one pair ahead: an MFMA result needs 6 wait states
    // before a VALU read, a software hazard the compiler does not apply to inline asm -- here the FMAs of the pair before,
    // two MFMAs and the guard of the first FMA are in between (first pair: longer guard).  scripts/check_dpp_hazard.py
    // verifies the final ISA.
    double xo[4] = {0.0, 0.0, 0.0, 0.0};
    double pend[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        pend[i] = ZEROC ? 0.0 : C.t[0][i];
        if constexpr (diag) {
            pend[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(f[i % JQ_PF], x.t[0][i], pend[i], 0, 0, 0);
            if (i + JQ_PF < NR) f[i % JQ_PF] = ma[JQ_T4_AIDX(i + JQ_PF, 0, 0)];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
        double c = 0.0;
        if constexpr (rt || mtm) {
            c = cq[mt & 1];
            if (mt + 2 < NT) cq[mt & 1] = cf[(mt + 2) * 64];
        }
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int r = 2 * pr, rho = 4 * mt + r;
            double curA = pend[0], curB = pend[1];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int nx = rho + 2 + i;
                if (nx < NR) {
                    double nxt = ZEROC ? 0.0 : C.t[nx >> 2][nx & 3];
                    if constexpr (diag) {
                        nxt = __builtin_amdgcn_mfma_f64_4x4x4f64(f[nx % JQ_PF], x.t[nx >> 2][nx & 3], nxt, 0, 0, 0);
                        if (nx + JQ_PF < NR) f[nx % JQ_PF] = ma[JQ_T4_AIDX(nx + JQ_PF, 0, 0)];
                    }
                    pend[i] = nxt;
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // both MFMAs of the next pair are issued before this pair's FMAs
            const double xA = x.t[mt][r], xB = x.t[mt][r + 1];
            if constexpr (rt || mtm) {
                const bool hi = mt + 1 < NT;
                const int mu = hi ? mt + 1 : mt;
                // {group below in the block, above in the block, same group of the block below, of the block above}; "below" values
                // are the OLD ones (the product may run in place)
                const double xa[4] = {xo[(rho + 3) & 3], xB, xo[rho & 3], hi ? x.t[mu][r] : 0.0};
                const double xb[4] = {xA, r + 2 < 4 ? x.t[mt][(r + 2) & 3] : 0.0, xo[(rho + 1) & 3], hi ? x.t[mu][r + 1] : 0.0};
                constexpr int G = diag ? 6 : 2;
                if (rho == 0)
                    t4_couple_pair<0, rt, mtm, G>(curA, curB, c, xa, xb, false, hi);
                else if (pr == 0)
                    t4_couple_pair<0, rt, mtm, 2>(curA, curB, c, xa, xb, mt > 0, hi);
                else
                    t4_couple_pair<2, rt, mtm, 2>(curA, curB, c, xa, xb, mt > 0, hi);
            }
            D.t[mt][r] = curA;
            D.t[mt][r + 1] = curB;
            xo[rho & 3] = xA;
            xo[(rho + 1) & 3] = xB;
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch FIFOs in order (hipcc otherwise hoists every load -> spills)
        }
    }
}

// x shifted by 4 lanes inside each row of 16 lanes (CTRL 0x114: lane n <- n - 4, 0x104: lane n <- n + 4; zeros shifted in)
template <int CTRL>
__device__ __forceinline__ double row_shift4(double x)
{
    union {
        double d;
        int i[2];
    } a, b;
    a.d = x;
#ifdef JQ_EXP_NOSHIFT      // timing experiment only (wrong results): what the lane shifts cost
    return x;
#endif
#ifdef JQ_EXP_BPERM        // one shift direction through the LDS crossbar (ds_bpermute_b32) instead of the VALU
    if (CTRL == 0x114) {
        const int addr = (((int)(threadIdx.x & 63) - 4) & 63) * 4;
        b.i[0] = __builtin_amdgcn_ds_bpermute(addr, a.i[0]);
        b.i[1] = __builtin_amdgcn_ds_bpermute(addr, a.i[1]);
        return b.d;
    }
#endif
#ifdef JQ_EXP_BPERM2       // both directions
    {
        const int addr = (((int)(threadIdx.x & 63) + (CTRL == 0x114 ? -4 : 4)) & 63) * 4;
        b.i[0] = __builtin_amdgcn_ds_bpermute(addr, a.i[0]);
        b.i[1] = __builtin_amdgcn_ds_bpermute(addr, a.i[1]);
        return b.d;
    }
#endif
    b.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], CTRL, 0xf, 0xf, true);
    b.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], CTRL, 0xf, 0xf, true);
    return b.d;
}
// BW == JQ_BW_T4Q (see the definition above): D = C + M x on the JQ_BW_T4 image.  Alias-safe (D may be C and/or x).
// This lane's share of an operator: A operand (lane 16 k + 4 b + i holds B_{4 mt + b}[i][k] = element 16 b + 4 k + i of the
// block's 64 doubles) and the coefficients [mt][g = row in group][r = group][term] of its row (four consecutive doubles).
typedef double d2 __attribute__((ext_vector_type(2)));
template <int NT>
struct OpQ {
    double a[NT];
    d4 c[NT];
};
__device__ __forceinline__ const double* t4q_a(const double* mat, int lane)
{
    return mat;      // (lane 16 k + 4 b + i reads element 16 k + 4 b + i of the block: JQ_T4_AIDX)
}
// (this lane's row: g = lane >> 4, r = (lane >> 2) & 3: the record of its four terms)
template <int NT>
__device__ __forceinline__ const d4* t4q_c(const double* mat, int lane)
{
    return (const d4*)(mat - lane + 4 * NT * JQ_T4_TILE + JQ_T4_CIDX(lane >> 4, (lane >> 2) & 3, 0));
}
__device__ __forceinline__ d4 t4q_cload(const d4* cf, int mt) { return cf[mt * 16]; }
// one 16-row block
template <int NT, bool ZEROC, int MODE>
__device__ __forceinline__ void t4q_block(Arr<NT>& D, const Arr<NT>& C, const Arr<NT>& x, int mt, double a, const d4& c, double& xold)
{
    constexpr bool diag = MODE & JQ_T4_DIAG, rt = MODE & JQ_T4_RTERMS, mtm = MODE & JQ_T4_MTERMS;
    const double xc = x.t[mt][0];
    double acc = ZEROC ? 0.0 : C.t[mt][0];
#ifdef JQ_EXP_MFMA_LAST      // experiment: coupling FMAs first, the MFMA last (its result is not read by the VALU for a whole product)
    const double xnext = x.t[mt + 1 < NT ? mt + 1 : mt][0];
    if constexpr (rt) {
        acc = fma(c[0], row_shift4<0x114>(xc), acc);
        acc = fma(c[1], row_shift4<0x104>(xc), acc);
    }
    if constexpr (mtm) {
        if (mt > 0) acc = fma(c[2], xold, acc);
        if (mt + 1 < NT) acc = fma(c[3], xnext, acc);
    }
    if constexpr (diag) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, xc, acc, 0, 0, 0);
#else
    if constexpr (diag) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, xc, acc, 0, 0, 0);
    if constexpr (rt) {
        acc = fma(c[0], row_shift4<0x114>(xc), acc);
        acc = fma(c[1], row_shift4<0x104>(xc), acc);
    }
    if constexpr (mtm) {
        if (mt > 0) acc = fma(c[2], xold, acc);
        if (mt + 1 < NT) acc = fma(c[3], x.t[mt + 1 < NT ? mt + 1 : mt][0], acc);
    }
#endif
    xold = xc;
    D.t[mt][0] = acc;
}
// SH (round 4): the ensemble shift of the sample, K' = K + eps diag(ws), folded into the MFMA's A operand -- lane 16 k + 4 b + i holds
// B[i][k], so the lanes with k == i carry the diagonal and, there, the operand's row equals the row of the lane in the STATE layout:
// a' = a + shd * ws[row], shd = (k == i) ? +-c eps : 0 per lane, ws read from the same row table a_axpy_rows reads.  One FMA per block
// on the operand instead of a multiply and an FMA per block on the result (a wave must hold ONE sample: the UNI kernels).
template <int NT, bool ZEROC, int MODE, bool SH = false>
__device__ __forceinline__ void mm_t4q(Arr<NT>& D, const Arr<NT>& C, const double* mat, const Arr<NT>& x, double shd = 0.0,
                                       const double* wsr = nullptr)
{
    constexpr bool diag = MODE & JQ_T4_DIAG, coef = (MODE & (JQ_T4_RTERMS | JQ_T4_MTERMS)) != 0;
    const int lane = threadIdx.x & 63;
    const double* ma = t4q_a(mat, lane);
    const d4* cf = t4q_c<NT>(mat, lane);
    double a_cur = 0.0;
    d4 c_cur = {0.0, 0.0, 0.0, 0.0};
    if constexpr (diag) a_cur = ma[0];
    if constexpr (diag && SH) a_cur = fma(shd, wsr[0], a_cur);
    if constexpr (coef) c_cur = t4q_cload(cf, 0);
    double xold = 0.0;
#ifdef JQ_EXP_NOPF      // experiment: the operands of THIS block only (no one-block-ahead prefetch: 10 registers less in flight)
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
        double a = 0.0;
        d4 c = {0.0, 0.0, 0.0, 0.0};
        if constexpr (diag) a = ma[mt * 64];
        if constexpr (coef) c = t4q_cload(cf, mt);
        t4q_block<NT, ZEROC, MODE>(D, C, x, mt, a, c, xold);
        __builtin_amdgcn_sched_barrier(0);
    }
    return;
#endif
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
        const double a = a_cur;
        const d4 c = c_cur;
        if (mt + 1 < NT) {
            if constexpr (diag) a_cur = ma[(mt + 1) * 64];
            if constexpr (diag && SH) a_cur = fma(shd, wsr[16 * (mt + 1)], a_cur);
            if constexpr (coef) c_cur = t4q_cload(cf, mt + 1);
        }
        t4q_block<NT, ZEROC, MODE>(D, C, x, mt, a, c, xold);
    }
}
// Two / three products with the SAME right-hand side in one pass over the blocks (round 3): D_k = C_k + M_k x.  The lane shifts
// of x (four v_mov_b32_dpp per block, as expensive as the block's MFMA) are made once for all of them.  x must not alias a D_k.
struct T4qNoHook {
    __device__ __forceinline__ void operator()(int, double, double) const {}
};
// (HOOK: called once per 16-row block with the block's two lane-shifted copies of x -- products with single-subsystem operators
//  that only need those, e.g. the trace products of a control of the middle subsystem, ride along without shifts of their own)
template <int NT, int NP, bool Z0, bool Z1, bool Z2, int SH = 0, typename HOOK = T4qNoHook>      // SH: bit k = operator k is a K image and takes the folded shift
__device__ __forceinline__ void mm_t4q_multi(Arr<NT>& D0, const Arr<NT>& C0, const double* m0, Arr<NT>& D1, const Arr<NT>& C1,
                                             const double* m1, Arr<NT>& D2, const Arr<NT>& C2, const double* m2, const Arr<NT>& x,
                                             double sh0 = 0.0, double sh1 = 0.0, double sh2 = 0.0, const double* wsr = nullptr,
                                             HOOK hook = HOOK())
{
    const int lane = threadIdx.x & 63;
    const double* ma[3] = {t4q_a(m0, lane), t4q_a(m1, lane), t4q_a(NP > 2 ? m2 : m1, lane)};
    const d4* cf[3] = {t4q_c<NT>(m0, lane), t4q_c<NT>(m1, lane), t4q_c<NT>(NP > 2 ? m2 : m1, lane)};
    double xold = 0.0;
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
        // (operands of THIS block only: a one-block-ahead prefetch like mm_t4q's costs 10 NP registers, which the twelve-wave
        // variants do not have; their other two waves per SIMD cover the LDS latency)
        double a[3];
        d4 c[3];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            a[k] = ma[k][mt * 64];
            c[k] = t4q_cload(cf[k], mt);
        }
        if constexpr (SH != 0) {      // (shift folded into the A operands of the K images, see mm_t4q)
            const double w = wsr[16 * mt];
            if constexpr (SH & 1) a[0] = fma(sh0, w, a[0]);
            if constexpr (NP > 1 && (SH & 2)) a[1] = fma(sh1, w, a[1]);
            if constexpr (NP > 2 && (SH & 4)) a[2] = fma(sh2, w, a[2]);
        }
        const double xc = x.t[mt][0], xn = x.t[mt + 1 < NT ? mt + 1 : mt][0];
        const double su = row_shift4<0x114>(xc), sd = row_shift4<0x104>(xc);
        hook(mt, su, sd);
        double acc[3] = {Z0 ? 0.0 : C0.t[mt][0], (NP > 1 && !Z1) ? C1.t[mt][0] : 0.0, (NP > 2 && !Z2) ? C2.t[mt][0] : 0.0};
#ifndef JQ_EXP_MFMA_LAST
#pragma unroll
        for (int k = 0; k < NP; ++k) acc[k] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[k], xc, acc[k], 0, 0, 0);
#endif
#pragma unroll
        for (int k = 0; k < NP; ++k) acc[k] = fma(c[k][0], su, acc[k]);
#pragma unroll
        for (int k = 0; k < NP; ++k) acc[k] = fma(c[k][1], sd, acc[k]);
        if (mt > 0) {
#pragma unroll
            for (int k = 0; k < NP; ++k) acc[k] = fma(c[k][2], xold, acc[k]);
        }
        if (mt + 1 < NT) {
#pragma unroll
            for (int k = 0; k < NP; ++k) acc[k] = fma(c[k][3], xn, acc[k]);
        }
#ifdef JQ_EXP_MFMA_LAST
#pragma unroll
        for (int k = 0; k < NP; ++k) acc[k] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[k], xc, acc[k], 0, 0, 0);
#endif
        xold = xc;
        D0.t[mt][0] = acc[0];
        if constexpr (NP > 1) D1.t[mt][0] = acc[1];
        if constexpr (NP > 2) D2.t[mt][0] = acc[2];
        // (fence per block: the scheduler otherwise hoists the operand reads of all blocks to the top -- 15 NT doubles in flight,
        // 200 spilled registers in the twelve-wave variants)
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <int NT, bool Z0, bool Z1, int SH = 0>
__device__ __forceinline__ void mm_t4q2(Arr<NT>& D0, const Arr<NT>& C0, const double* m0, Arr<NT>& D1, const Arr<NT>& C1,
                                        const double* m1, const Arr<NT>& x, double sh0 = 0.0, double sh1 = 0.0, const double* wsr = nullptr)
{
    mm_t4q_multi<NT, 2, Z0, Z1, true, SH>(D0, C0, m0, D1, C1, m1, D1, C1, m1, x, sh0, sh1, 0.0, wsr);
}
template <int NT, bool Z0, bool Z1, bool Z2, int SH = 0>
__device__ __forceinline__ void mm_t4q3(Arr<NT>& D0, const Arr<NT>& C0, const double* m0, Arr<NT>& D1, const Arr<NT>& C1,
                                        const double* m1, Arr<NT>& D2, const Arr<NT>& C2, const double* m2, const Arr<NT>& x,
                                        double sh0 = 0.0, double sh1 = 0.0, double sh2 = 0.0, const double* wsr = nullptr)
{
    mm_t4q_multi<NT, 3, Z0, Z1, Z2, SH>(D0, C0, m0, D1, C1, m1, D2, C2, m2, x, sh0, sh1, sh2, wsr);
}
// the same with the operator in registers (the m + 1 products of a Horner chain share it: no LDS latency at their heads)
template <int NT>
__device__ __forceinline__ void t4q_load(OpQ<NT>& op, const double* mat)
{
    const int lane = threadIdx.x & 63;
    const double* ma = t4q_a(mat, lane);
    const d4* cf = t4q_c<NT>(mat, lane);
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
        op.a[mt] = ma[mt * 64];
#ifdef JQ_EXP_OPQ_EDGE
        // (round 5 experiment: the first block has no neighbour below and the last one none above -- their coefficients c[2] / c[3]
        //  are never used, but a 32-byte read keeps the register pair of the whole record alive: read the halves that are used)
        const double* cd = (const double*)(cf + mt * 16);
        op.c[mt][0] = cd[0], op.c[mt][1] = cd[1];
        op.c[mt][2] = mt > 0 ? cd[2] : 0.0;
        op.c[mt][3] = mt + 1 < NT ? cd[3] : 0.0;
#else
        op.c[mt] = t4q_cload(cf, mt);
#endif
    }
}
template <int NT>
__device__ __forceinline__ void mm_t4q_regs(Arr<NT>& D, const Arr<NT>& C, const OpQ<NT>& op, const Arr<NT>& x)
{
    double xold = 0.0;
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) t4q_block<NT, false, JQ_T4_DIAG | JQ_T4_RTERMS | JQ_T4_MTERMS>(D, C, x, mt, op.a[mt], op.c[mt], xold);
}

template <int NT, int BW, bool ZEROC, bool SD = false>
__device__ __forceinline__ void mm_band(Arr<NT>& D, const Arr<NT>& C, const double* mat, const Arr<NT>& x);
template <int NT, int BW, bool ZEROC, bool SD = false>
__device__ __forceinline__ void mm_any(Arr<NT>& D, const Arr<NT>& C, const double* mat, const Arr<NT>& x)
{
    if constexpr (BW == JQ_BW_T4)
        mm_t4<NT, ZEROC, JQ_T4_DIAG | JQ_T4_RTERMS | JQ_T4_MTERMS>(D, C, mat, x);
    else if constexpr (BW == JQ_BW_T4Q)
        mm_t4q<NT, ZEROC, JQ_T4_DIAG | JQ_T4_RTERMS | JQ_T4_MTERMS>(D, C, mat, x);
    else if constexpr (BW == JQ_BW_OD)
        mm_od<NT, ZEROC, SD>(D, C, mat, x);
    else
        mm_band<NT, BW, ZEROC, SD>(D, C, mat, x);
}
template <int NT, int BW, bool ZEROC, bool SD>
__device__ __forceinline__ void mm_band(Arr<NT>& D, const Arr<NT>& C, const double* mat, const Arr<NT>& x)
{
    constexpr int NTILES = band_tiles(NT, BW, SD);
    double f[JQ_PF];
#pragma unroll
    for (int i = 0; i < JQ_PF; ++i)
        if (i < NTILES) f[i] = mat[i * 64];
    int idx = 0;
#pragma unroll
    for (int kk = 0; kk < 4 * NT; ++kk) {
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) {
            if (block_on(BW, SD, mt, kk >> 2)) {
                const double a = f[idx % JQ_PF];
                if (idx + JQ_PF < NTILES) f[idx % JQ_PF] = mat[(idx + JQ_PF) * 64];
                const bool first = (kk == 4 * first_kb(NT, BW, SD, mt));
                const d4 zero = {0.0, 0.0, 0.0, 0.0};
                const d4 cin = first ? (ZEROC ? zero : C.t[mt]) : D.t[mt];
                D.t[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, x.t[kk >> 2][kk & 3], cin, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);   // keep the FIFO order: hipcc otherwise hoists every ds_read
                ++idx;
            }
        }
    }
    if (ZEROC) {
        // tile rows without any stored tile (cannot happen for BW >= 0 without SD; with SD only if NT == 1)
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
            if (first_kb(NT, BW, SD, mt) >= NT) D.t[mt] = (d4){0.0, 0.0, 0.0, 0.0};
    }
}
template <int NT, int BW>
__device__ __forceinline__ void mm_c(Arr<NT>& D, const Arr<NT>& C, const double* mat, const Arr<NT>& x)
{
    mm_any<NT, BW, false>(D, C, mat, x);
}
template <int NT, int BW>
__device__ __forceinline__ void mm_z(Arr<NT>& D, const double* mat, const Arr<NT>& x)
{
    mm_any<NT, BW, true>(D, D, mat, x);
}
// D = M * x with a constant (trace) operator stored in its own layout:
//   mode 0: block diagonal (band 0)   mode 1: the kernel's band BW   mode 2: band BW without the diagonal blocks
template <int NT, int BW>
__device__ __forceinline__ void mm_z_bw(Arr<NT>& D, const double* mat, const Arr<NT>& x, int mode)
{
    if constexpr (BW == JQ_BW_T4) {
        // single-subsystem control operators touch exactly one part of the image; anything else takes the full product
        // (always correct: the absent parts are stored as zeros)
        switch (mode) {
        case JQ_T4_DIAG: mm_t4<NT, true, JQ_T4_DIAG>(D, D, mat, x); break;
        case JQ_T4_RTERMS: mm_t4<NT, true, JQ_T4_RTERMS>(D, D, mat, x); break;
        case JQ_T4_MTERMS: mm_t4<NT, true, JQ_T4_MTERMS>(D, D, mat, x); break;
        default: mm_t4<NT, true, JQ_T4_DIAG | JQ_T4_RTERMS | JQ_T4_MTERMS>(D, D, mat, x); break;
        }
    } else if constexpr (BW == JQ_BW_T4Q) {
        switch (mode) {
        case JQ_T4_DIAG: mm_t4q<NT, true, JQ_T4_DIAG>(D, D, mat, x); break;
        case JQ_T4_RTERMS: mm_t4q<NT, true, JQ_T4_RTERMS>(D, D, mat, x); break;
        case JQ_T4_MTERMS: mm_t4q<NT, true, JQ_T4_MTERMS>(D, D, mat, x); break;
        default: mm_t4q<NT, true, JQ_T4_DIAG | JQ_T4_RTERMS | JQ_T4_MTERMS>(D, D, mat, x); break;
        }
    } else {
        if (BW > 0 && mode == 0)
            mm_any<NT, 0, true>(D, D, mat, x);
        else if (BW > 0 && NT > 1 && mode == 2)
            mm_any<NT, BW, true, true>(D, D, mat, x);
        else
            mm_any<NT, BW, true>(D, D, mat, x);
    }
}

// x + (x rotated right by N lanes within each row of 16 lanes), via DPP (no LDS crossbar traffic)
template <int N>
__device__ __forceinline__ double row_ror_add(double x)
{
    union {
        double d;
        int i[2];
    } a, b;
    a.d = x;
    b.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], 0x120 + N, 0xf, 0xf, false);
    b.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], 0x120 + N, 0xf, 0xf, false);
    return x + b.d;
}
__device__ __forceinline__ double lane_bcast(double x, int l)
{
    union {
        double d;
        int i[2];
    } a, b;
    a.d = x;
    b.i[0] = __builtin_amdgcn_readlane(a.i[0], l);
    b.i[1] = __builtin_amdgcn_readlane(a.i[1], l);
    return b.d;
}
// sum over the 64 lanes of the wave (valid in every lane): 4 DPP rotate-adds inside the rows of 16,
// then the four row sums are combined through scalar broadcasts
__device__ __forceinline__ double wave_sum(double x)
{
    x = row_ror_add<8>(x);
    x = row_ror_add<4>(x);
    x = row_ror_add<2>(x);
    x = row_ror_add<1>(x);
    return (lane_bcast(x, 0) + lane_bcast(x, 16)) + (lane_bcast(x, 32) + lane_bcast(x, 48));
}

// sum over the 64 lanes, valid in the lanes 48 .. 63 ONLY (17 VALU instructions instead of wave_sum's 23): the row sums travel down
// the rows with the two DPP row broadcasts (row_bcast:15 into the rows 1, 3, then row_bcast:31 into the rows 2, 3)
__device__ __forceinline__ double wave_sum_hi(double x)
{
    x = row_ror_add<8>(x);
    x = row_ror_add<4>(x);
    x = row_ror_add<2>(x);
    x = row_ror_add<1>(x);
    union {
        double d;
        int i[2];
    } a, b;
    a.d = x;
    b.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], 0x142, 0xa, 0xf, false);
    b.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], 0x142, 0xa, 0xf, false);
    x += b.d;      // rows: S0, S0 + S1, S2, S2 + S3
    a.d = x;
    b.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], 0x143, 0xc, 0xf, false);
    b.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], 0x143, 0xc, 0xf, false);
    return x + b.d;      // row 3: (S2 + S3) + (S0 + S1)
}

// Wave sums of FOUR values at a time (gfx950 row-swap instructions): v_permlane32_swap exchanges the rows 2, 3 of one
// register with the rows 0, 1 of another, so ONE add halves two values at once (rows: a0+a2, a1+a3, b0+b2, b1+b3);
// v_permlane16_swap (odd rows of the first <-> even rows of the second) does the same for the next level and leaves the
// 16 column partials of a, c, b, d in the rows 0, 1, 2, 3; four DPP rotate-adds finish all four.  21 VALU instructions
// for four sums instead of 4 x 23 (wave_sum); the sum of a / c / b / d is valid in every lane of row 0 / 1 / 2 / 3.
typedef unsigned jq_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void row_swap32(double& a, double& b)
{
    union { double d; unsigned u[2]; } x, y;
    x.d = a, y.d = b;
    const jq_u2 lo = __builtin_amdgcn_permlane32_swap(x.u[0], y.u[0], false, false);
    const jq_u2 hi = __builtin_amdgcn_permlane32_swap(x.u[1], y.u[1], false, false);
    x.u[0] = lo[0], y.u[0] = lo[1], x.u[1] = hi[0], y.u[1] = hi[1];
    a = x.d, b = y.d;
}
__device__ __forceinline__ void row_swap16(double& a, double& b)
{
    union { double d; unsigned u[2]; } x, y;
    x.d = a, y.d = b;
    const jq_u2 lo = __builtin_amdgcn_permlane16_swap(x.u[0], y.u[0], false, false);
    const jq_u2 hi = __builtin_amdgcn_permlane16_swap(x.u[1], y.u[1], false, false);
    x.u[0] = lo[0], y.u[0] = lo[1], x.u[1] = hi[0], y.u[1] = hi[1];
    a = x.d, b = y.d;
}
__device__ __forceinline__ double wave_sum4(double a, double b, double c, double d)
{
    row_swap32(a, b);
    double p = a + b;
    row_swap32(c, d);
    double q = c + d;
    row_swap16(p, q);
    double r = p + q;
    r = row_ror_add<8>(r);
    r = row_ror_add<4>(r);
    r = row_ror_add<2>(r);
    return row_ror_add<1>(r);
}
// the same with the sums of r0 .. r3 in the rows 0 .. 3 (lane 16 r holds the r-th sum): the trace scalars of the latency kernels
// (round 3: they reduced every scalar with its own wave_sum -- 23 instructions each at the lone-wave issue rate)
__device__ __forceinline__ double wave_sum4_rows(double r0, double r1, double r2, double r3) { return wave_sum4(r0, r2, r1, r3); }
// two sums at once: the rows 0, 1 hold the wave total of a, the rows 2, 3 that of b (20 VALU instructions; two wave_sum: 46)
__device__ __forceinline__ double wave_sum2(double a, double b)
{
    row_swap32(a, b);
    double p = a + b;      // rows 0, 1: a0+a2, a1+a3;  rows 2, 3: b0+b2, b1+b3
    double q = p;
    row_swap16(p, q);      // p: [p0 p0 p2 p2], q: [p1 p1 p3 p3]
    double r = p + q;
    r = row_ror_add<8>(r);
    r = row_ror_add<4>(r);
    r = row_ror_add<2>(r);
    return row_ror_add<1>(r);
}
// ---------------------------------------------------------------------------------------------
// One entry of the per-step operator schedule: which image the next product group multiplies with.
//   kind 0/1: K / S of the tile stream at time point 2*n + tp of the chunk;  kind 2: constant image #tp
struct SchedEntry {
    int kind;
    int tp;
};

struct PropArgs {
    const double* stream;   // chunk tile stream: time point j -> K at (2j)*stride, S at (2j+1)*stride
    const double* cimg;     // constant trace images [Hsym_0.. | Hanti_0..], `stride` doubles each
    double* state;          // per-slab array file
    const double* colinfo;  // per slab: eps[16], wgt[16]
    double* traces;         // backward: [workgroups][nsteps_chunk][Ncoupled*JQ_NTR] (slab / quad kernels: summed over the workgroup's waves)
    double* hist_r;         // forward history of sample 0 ([Ntot,N,nsteps+1]) or null
    double* hist_i;
    const double* tabs;     // wd[NP] (diag wmat_real, zero padded), ws[NP] (shift weights)
    double* park;           // HBM parking images [nslabs][4*NT*64] (used when park_lds == 0)
    long long stride;       // doubles per operator image slot (multiple of 128 = 1 KiB)
    long long state_stride; // doubles per slab in the array file
    int pieces;             // 1 KiB DMA pieces per operator image
    int nslots;             // LDS ring depth (2..JQ_MAXSLOTS)
    int nsteps_chunk;
    int m;                  // Neumann terms
    int nslabs;
    int Ncoupled;
    int step0;              // global index of the first step of this chunk
    int first_chunk;
    int Ntot, N;
    int parts;              // N > 16: slabs per sample (slab sl = part sl % parts, columns 16 part ..), else 1
    int nsamples, sps;      // evaluations of the batch, samples per slab (cooperative-quad kernels: which column quads are in use)
    int qps;                // ... column quads of a full slab = trace-record rows per slab of those kernels
    int use_shift;
    int forced;             // backward: add the leakage forcing (0: step_no_forcing!)
    int debug;              // profiling experiments only (JQ_DEBUG): 1 skip trace reductions, 2 skip forcing/shift rows, 4 skip parking
    int park_lds;           // 1: the backward kernel parks its dormant array in LDS, 0: in `park`
    int batch;              // 0: one LDS slot per operator use; B > 0: K/S images of B time steps per DMA batch
    int lds_tab_off;        // byte offset of the tables (wd, ws[, carry, park]) in dynamic LDS
    int period;             // operator uses per time step
    int npro;               // operator uses before the first step (backward first chunk: carry products)
    int bw_trace[JQ_MAXNC]; // layout of the trace images per control: 0 block diagonal, 1 band BW, 2 band BW w/o diagonal
    double h;               // signed time step
    double tinv;            // 1/T
    double jacobi_tol2;     // JAC kernels (JACOBI_SOLVER): squared tolerance, max_iter = m
    // operator schedule, 6 bits per entry (kind | tp << 2), 10 entries per 64-bit word: decoded with
    // scalar ALU ops only (a table in memory costs a dependent scalar load in front of every DMA issue)
    unsigned long long sched_bits[3];  // entries 0..29 of the per-step schedule
    unsigned long long pro_bits;       // entries of the prologue (backward first chunk)
    // Full / complex leakage weights (use_custom_forbidden, src/evalobjgrad.jl:214-232) in low-rank form
    //   W = wmat_real + i wmat_imag = sum_{k < wrank} lam_k f_k f_k^H ,  f_k = a_k + i b_k   (jq_update_wmat: eigen-decomposition)
    // wlr (global memory, natural row order): lam[wlam], then per k the rows a_k[wstride], b_k[wstride] (zero padded).
    // wrank == 0: Diagonal weights (the table wd); wrank > 0: wd is all zero and the kernels add the low-rank terms.
    const double* wlr;
    int wrank;
    int wstride;
    int wlr_lds;            // slab / quad kernels: byte offset in dynamic LDS where the workgroup keeps a copy of the table, or -1 (read it from global memory)
    int wlr_sc_lds;         // quad layout: byte offset of the per-wave column scalars of the terms ([wave][JQ_MAX_WRANK][6][4] doubles), or -1
    int jac_wg_lds;         // JAC slab kernels, N > 16 with one workgroup per sample (its <= 4 parts = waves): byte offset of the residual
                            // exchange [2][JQ_WAVES] doubles -- the stopping test then sums the parts like the reference; -1: per part
    int wcplx;              // cooperative-quad kernels (CqW): 0 = real weight matrix, the four slots are a_0 .. a_3; 1 = complex, rank <= 2: a_0, b_0, a_1, b_1
    int wlam;               // lam slots in front of the rows of the low-rank table: max(JQ_MAX_WRANK, wrank) -- ranks beyond JQ_MAX_WRANK (slab, cooperative and
                            // run-time-size kernels only; the row-lane and cooperative-quad kernels see rank <= 16 / <= 4 and the constant)
    // Two- / three-workgroup latency kernels (jq_cq_split_kernels.h): polls of the start-up rendezvous (every workgroup of the launch announces
    // itself and waits for all the others: co-residency is established before anybody depends on it) and of a wait between roles afterwards
    int rdv_polls;
    int wait_polls;
};
#ifndef JQ_MAX_WRANK
#define JQ_MAX_WRANK 16       // largest rank of a full weight matrix the kernels take (include/juqbox_hip.h)
#endif
__host__ __device__ inline void sched_pack(unsigned long long* words, int i, int kind, int tp)
{
    words[i / 10] |= (unsigned long long)((kind & 3) | ((tp & 15) << 2)) << (6 * (i % 10));
}

// ---------------------------------------------------------------------------------------------
// Low-rank leakage weights (PropArgs::wlr) in the slab / quad layouts.  W x for one state column is
//   sum_k lam_k [ a_k (a_k.x) + b_k (b_k.x) ]  (wmat_real x)   and   sum_k lam_k [ b_k (a_k.x) - a_k (b_k.x) ]  (wmat_imag x),
// i.e. two column dot products and two axpys per forbidden state instead of a dense product.  A column's rows sit in the
// lanes l, l ^ 16, l ^ 32, l ^ 48 (slab layout: row 16 i + 4 r + (l >> 4)) or in the 16 lanes with the same l & 3 (quad layout:
// row 16 i + 4 ((l >> 2) & 3) + (l >> 4)); the dots are all-reduced over them, so every lane of a column holds the column's value.
// (the cross-row stages use gfx950's row-swap instructions on two copies of the value -- v_permlane32_swap leaves [lo | lo] and
//  [hi | hi], v_permlane16_swap the even and the odd rows twice -- six VALU instructions per value; __shfl_xor goes through the LDS
//  crossbar, whose latency a lone wave cannot hide: cnot3 with two forbidden states 601 -> 5xx ms per evaluation)
template <bool QUAD>
__device__ __forceinline__ double col_allsum(double x)
{
    if constexpr (QUAD) x = row_ror_add<8>(row_ror_add<4>(x));
    double y = x;
    row_swap32(x, y);      // x = [x_lo, x_lo], y = [x_hi, x_hi] (32-lane halves)
    x += y;
    y = x;
    row_swap16(x, y);      // x: rows 0, 0, 2, 2; y: rows 1, 1, 3, 3
    return x + y;
}
template <int NT, bool QUAD>
struct WLow {
    const double* tab;      // first row of a_0 of this lane (rows 16 i + 4 q further on)
    const double* lamp;     // lam[k]
    int r, stride;
    bool lead;      // one lane per column: adds the column's scalar terms to a per-lane partial sum
    // Column scalars of the terms kept across uses and steps (quad layout, round 5 EXPERIMENT, off by default: host option wlr_sc=1): the dots
    // of a step's vr(t_n), vi05 are needed at two sites of the adjoint step, and the dots with vr(t_n+1) ARE the previous step's dots with
    // vr(t_n) -- five dot pairs per term and step become two.  The rank is a run-time number, so the scalars live in LDS: [term][slot
    // 0 .. 5][column of the quad], written by the lead lanes.  Measured SLOWER (cnot3: 57 -> 70 ms per forbidden state): the LDS round
    // trips sit on the critical path of a wave that is alone on its SIMD, the recomputed dots are independent instructions.
    __attribute__((address_space(3))) double* sc;
    bool has_sc;
    __device__ __forceinline__ void put(int k, int j, double val) const
    {
        if (lead) sc[(k * 6 + j) * 4] = val;
    }
    __device__ __forceinline__ double get(int k, int j) const { return sc[(k * 6 + j) * 4]; }
    // (smem: the workgroup's dynamic LDS.  With a.wlr_lds >= 0 the table is copied there once -- a lone wave cannot hide the latency
    //  of global reads inside every time step; contains a workgroup barrier then: call it from every wave)
    __device__ __forceinline__ void init(const PropArgs& a, int lane_, char* smem)
    {
        r = a.wrank;
        stride = a.wstride;
        const double* base = a.wlr;
        if (r > 0 && a.wlr_lds >= 0) {
            double* l = (double*)(smem + a.wlr_lds);
            for (int i = threadIdx.x; i < a.wlam + 2 * r * stride; i += blockDim.x) l[i] = a.wlr[i];
            __syncthreads();
            base = l;
        }
        lamp = base;
        tab = base + a.wlam + (QUAD ? 4 * ((lane_ >> 2) & 3) + (lane_ >> 4) : (lane_ >> 4));
        lead = QUAD ? lane_ < 4 : lane_ < 16;
        has_sc = QUAD && r > 0 && a.wlr_sc_lds >= 0;
        sc = (__attribute__((address_space(3))) double*)(smem + (has_sc ? a.wlr_sc_lds : 0)) + (size_t)(threadIdx.x >> 6) * (JQ_MAX_WRANK * 24) + (lane_ & 3);
    }
    __device__ __forceinline__ double lam(int k) const { return lamp[k]; }
    // (da, db) = (a_k . x, b_k . x) of this lane's column
    __device__ __forceinline__ void dots(int k, const Arr<NT>& x, double& da, double& db) const
    {
        const double* ta = tab + (size_t)(2 * k) * stride;
        const double* tb = ta + stride;
        double sa = 0.0, sb = 0.0;
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int q = 0; q < JQ_RL; ++q) {
                sa = fma(ta[16 * i + 4 * q], x.t[i][q], sa);
                sb = fma(tb[16 * i + 4 * q], x.t[i][q], sb);
            }
        da = col_allsum<QUAD>(sa);
        db = col_allsum<QUAD>(sb);
    }
    // y += ca a_k + cb b_k
    __device__ __forceinline__ void axpy2(int k, Arr<NT>& y, double ca, double cb) const
    {
        const double* ta = tab + (size_t)(2 * k) * stride;
        const double* tb = ta + stride;
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int q = 0; q < JQ_RL; ++q) y.t[i][q] = fma(ca, ta[16 * i + 4 * q], fma(cb, tb[16 * i + 4 * q], y.t[i][q]));
    }
};

// usaver[:,:,step+1] = vr ; usavei = -vi (src/evalobjgrad.jl:748-752); only sample 0 (slab 0, columns < N)
// (col: state column of this lane; g: its row in a 4-row group -- quad layout: its row in a 16-row block)
template <int NT>
__device__ __forceinline__ void hist_store(const PropArgs& a, int slab, int col, int g, int n, const Arr<NT>& u,
                                           const Arr<NT>& v)
{
    const int scol = a.parts > 1 ? 16 * slab + col : col;      // column of sample 0
    if (slab < a.parts && scol < a.N) {
        const size_t off = (size_t)(a.step0 + n + 1) * a.Ntot * a.N + (size_t)scol * a.Ntot;
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int r = 0; r < JQ_RL; ++r) {
                const int row = 16 * i + (JQ_RL == 1 ? 0 : 4 * r) + g;
                if (row < a.Ntot) {
                    a.hist_r[off + row] = u.t[i][r];
                    a.hist_i[off + row] = -v.t[i][r];
                }
            }
    }
}

// LDS staging of the operator images, fed by global->LDS DMA (global_load_lds_dwordx4).
//
// per-operator mode (large images, batch == 0): operator use #Q lives in slot Q & 1; while the MFMAs of
//   use Q run, the image of use Q+1 streams into the other slot (every wave issues a quarter of the 1 KiB
//   pieces).  One workgroup barrier per operator switch; each wave drains its own DMA (vmcnt(0)) right
//   before the barrier that publishes the image.  (Deeper rings were measured to make no difference: one
//   product takes >= 4096 cycles, the 32 KiB image lands in a fraction of that.)
// batched mode (small images, batch = B > 0): the K/S images of B consecutive time steps (2B+1 time
//   points, contiguous in the tile stream) are fetched with one DMA burst into one of two batch buffers
//   and the constant trace images stay resident, so there is ONE barrier per B steps instead of one per
//   operator use -- for Ntot <= 32 a product is only 4..32 MFMAs and the barrier + DMA latency would
//   otherwise dominate.
// window mode (batch < 0; images small enough that everything below fits the LDS -- the compact JQ_BW_T4 images):
//   a ring of JQ_WIN_TPS = 5 time points (K and S image each) and the constant trace images are RESIDENT: step n works
//   on the time points 2n, 2n+1, 2n+2 while 2n+3 and 2n+4 stream in, every image is fetched once instead of once per
//   use (backward: 4 image fetches per step instead of 13 + 3 Ncoupled), and there is ONE workgroup barrier per time step
//   instead of one per operator use (measured: the per-use wait + barrier + DMA issue was 15% of the cnot3 evaluation).
#define JQ_WIN_TPS 5
// (jq_quad_split_kernels.h -- backward sweep with the two chains of a column quad on two waves, one step apart: the ring also keeps the
//  time points of the previous step; two arrays per step are handed from the state wave to the adjoint wave through global memory)
#define JQ_QS_TPS 7
#define JQ_QS_ARRAYS 2
// WIN: the kernel only ever runs in window mode (the quad-layout kernels): the mode tests are compile-time -- ~ 35 scalar branches
// per backward step less (3 072 cnot3 samples 1 127 -> 1 114 ms); a scheduling barrier stands where each of them was, without it
// hipcc hoists the operand reads of later stages over the whole step (380 / 496 B of scratch, 1.8 x slower).
// TPS: slots of the window ring (time points).  JQ_WIN_TPS = 5 is also the number of time points fetched ahead (the points of a step
// and of the next one); a deeper ring (the split backward kernel, jq_quad_split_kernels.h: 7) keeps the points of the PREVIOUS step
// resident as well -- same fetch schedule, slot = time point % TPS.
template <bool WIN, int TPS = JQ_WIN_TPS>
struct RingT {
    char* smem;
    // copies of the launch parameters the staging needs (kept in SGPRs; taking the address of the kernel
    // argument struct would make hipcc spill all of it to scratch)
    const double* stream;
    const double* cimg;
    unsigned long long sb0, sb1, sb2, pb;
    long long stride;
    int pieces, nsteps_chunk, period, npro, batch, ncoupled, debug;
    int slot_bytes;   // bytes of one slot (per-operator mode) or of one batch buffer (batched mode)
    int Q;            // index of the operator use that comes next
    int Qp;           // per-operator mode: index of the next operator use to prefetch
    int np, ip;       // (step, position) cursor: of Qp (per-operator mode) or of Q (batched mode)
    int wave, lane, nwaves;
    // per-operator mode, prefetch cursor in incremental form (no multiplications by the step number, no schedule word
    // selection, no reloads of kernel arguments in front of every DMA issue -- that code was 10% of the backward step):
    unsigned long long pword;      // schedule entries not yet consumed of the current word (6 bits each)
    const char* pbase;             // stream + images of time point 2 * np (bytes)
    unsigned stride_b;             // bytes per image slot
    // window mode: cursor of the next operator use
    unsigned long long qword;      // its schedule entries not yet consumed
    int iq;                        // its position in the step
    int s0;                        // ring slot of time point 2n of the current step
    unsigned wb0, wb1, wb2;        // byte offsets of the K image of the time points 2n, 2n+1, 2n+2

    __device__ __forceinline__ unsigned lane_off16() const
    {
        // lane byte offset recomputed here (2 VALU ops) so that no long-lived VGPR has to survive
        unsigned lo;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lo));
        return lo * 16u;
    }
    __device__ __forceinline__ void dma(const double* gsrc, char* dst, int pieces) const
    {
        if constexpr (WIN) {
            // scalar base + 32-bit lane offset: no 64-bit per-lane address (hipcc kept its zero-extended half in scratch and
            // re-spilled it in every step: 8 B per lane and step = 25 GB per launch through to HBM).  begin_step / init drain the
            // DMA (vmcnt(0)) in front of the workgroup barrier that publishes the images.
            const unsigned lo16 = lane_off16();
            const unsigned ldst = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)dst;
            for (int p = wave; p < pieces; p += nwaves) {
                const char* sp = (const char*)gsrc + (size_t)p * 1024;
                asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(ldst + (unsigned)p * 1024u), "v"(lo16), "s"(sp) : "memory");
            }
            return;
        }
        const char* src = (const char*)gsrc + lane_off16();
        for (int p = wave; p < pieces; p += nwaves)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)p * 1024),
                                             (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 0);
    }
    // the pieces of an image with their number per wave known at compile time (the cooperative kernels: 2 NB pieces per wave): straight-line
    // code.  The loop above costs a wave a taken branch per piece, 20 .. 120 clk each when the workgroup is alone on its CU -- with the
    // schedule bookkeeping around it that was more than the 8 MFMAs of a product of a 32-level problem (round 6, profiles/r06_midsize_single.txt)
    template <int KPER>
    __device__ __forceinline__ void dma_k(const double* gsrc, char* dst) const
    {
        const char* src = (const char*)gsrc + lane_off16() + (size_t)wave * 1024;
        char* d = dst + wave * 1024;
        const int step = nwaves * 1024;
#pragma unroll
        for (int i = 0; i < KPER; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)i * step),
                                             (__attribute__((address_space(3))) void*)(d + i * step), 16, 0, 0);
    }
    // schedule entry of operator use q (cursor position i within the step); scalar loads, no struct copies
    __device__ __forceinline__ void entry_at(int q, int i, int& kind, int& tp) const
    {
        unsigned long long w;
        int k = i;
        if (q < npro) {
            w = pb;
            k = q;
        } else if (i < 10) {
            w = sb0;
        } else if (i < 20) {
            w = sb1;
            k = i - 10;
        } else {
            w = sb2;
            k = i - 20;
        }
        const unsigned e = (unsigned)(w >> (6 * k)) & 63u;
        kind = (int)(e & 3u);
        tp = (int)(e >> 2);
    }
    __device__ __forceinline__ void advance(int& q, int& n, int& i) const
    {
        if (q >= npro) {
            if (++i == period) {
                i = 0;
                ++n;
            }
        }
        ++q;
    }
    // ---- per-operator mode -------------------------------------------------------------------
    template <int KPER = 0>
    __device__ __forceinline__ void issue_prefetch()
    {
        const unsigned e = (unsigned)pword & 63u, kind = e & 3u, tp = e >> 2;
        // image #tp of the constants, or K (kind 0) / S (kind 1) of time point 2 * np + tp of the chunk
        const char* src = (kind == 2) ? (const char*)cimg + tp * stride_b : pbase + (2 * tp + kind) * stride_b;
        if constexpr (KPER > 0)
            dma_k<KPER>((const double*)src, smem + (size_t)(Qp & 1) * slot_bytes);
        else
            dma((const double*)src, smem + (size_t)(Qp & 1) * slot_bytes, pieces);
        pword >>= 6;
        if (Qp >= npro) {
            ++ip;
            if (ip == 10) pword = sb1;
            if (ip == 20) pword = sb2;
            if (ip == period) {
                ip = 0;
                pword = sb0;
                if (++np < nsteps_chunk) pbase += 4 * (size_t)stride_b;   // (past the end: harmless re-fetch of the last step)
            }
        } else if (Qp + 1 == npro) {
            pword = sb0;
        }
        ++Qp;
    }
    // ---- batched mode ------------------------------------------------------------------------
    __device__ __forceinline__ void issue_batch(int b)
    {
        const int first = b * batch;
        if (first >= nsteps_chunk) return;
        int steps = nsteps_chunk - first;
        if (steps > batch) steps = batch;
        const int npts = 2 * steps + 1;
        dma(stream + (size_t)(4 * first) * stride, smem + (size_t)(b & 1) * slot_bytes, npts * 2 * pieces);
    }
    // call at the top of every time step n (of the chunk)
    __device__ __forceinline__ void set_window()
    {
        const int s1 = s0 + 1 >= TPS ? s0 + 1 - TPS : s0 + 1, s2 = s0 + 2 >= TPS ? s0 + 2 - TPS : s0 + 2;
        wb0 = (unsigned)(s0 * slot_bytes), wb1 = (unsigned)(s1 * slot_bytes), wb2 = (unsigned)(s2 * slot_bytes);
    }
    __device__ __forceinline__ void issue_tp(int j)   // window mode: K and S of time point j of the chunk -> its ring slot
    {
        if (j > 2 * nsteps_chunk) return;
        dma((const double*)((const char*)stream + (size_t)j * 2 * stride_b), smem + (size_t)(j % TPS) * 2 * stride_b, 2 * pieces);
    }
    __device__ __forceinline__ void begin_step(int n)
    {
        if (WIN || batch < 0) {
            if (n > 0) {
                // every wave has finished step n-1 behind this barrier: its time points 2n-2, 2n-1 make room for 2n+3, 2n+4;
                // the images of this step (issued one step ago) have landed
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                issue_tp(2 * n + 3);
                issue_tp(2 * n + 4);
                s0 += 2;
                if (s0 >= TPS) s0 -= TPS;
                set_window();
            }
            return;
        }
        if (batch > 0 && (n % batch) == 0) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issue_batch(n / batch + 1);
        }
    }
    __device__ __forceinline__ void init(char* smem_, const PropArgs& a, int wave_, int lane_, int nwaves_ = JQ_WAVES)
    {
        smem = smem_;
        nwaves = nwaves_;
        stream = a.stream;
        cimg = a.cimg;
        sb0 = a.sched_bits[0];
        sb1 = a.sched_bits[1];
        sb2 = a.sched_bits[2];
        pb = a.pro_bits;
        stride = a.stride;
        pieces = a.pieces;
        nsteps_chunk = a.nsteps_chunk;
        period = a.period;
        npro = a.npro;
        batch = a.batch;
        ncoupled = a.Ncoupled;
        debug = a.debug;
        Q = 0;
        Qp = 0;
        np = 0;
        ip = 0;
        wave = wave_;
        lane = lane_;
        if (WIN || batch < 0) {
            stride_b = (unsigned)(stride * 8);
            slot_bytes = (int)(2 * stride_b);
            // resident constant images behind the ring of time points
            dma(cimg, smem + (size_t)TPS * slot_bytes, 2 * ncoupled * pieces);
            for (int j = 0; j < JQ_WIN_TPS; ++j) issue_tp(j);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            qword = npro > 0 ? pb : sb0;
            iq = 0;
            s0 = 0;
            set_window();
        } else if (batch > 0) {
            slot_bytes = (int)((2 * batch + 1) * 2 * stride * 8);
            // resident constant images behind the two batch buffers
            dma(cimg, smem + 2 * (size_t)slot_bytes, 2 * ncoupled * pieces);
            issue_batch(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // constants are used before the first begin_step
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        } else {
            slot_bytes = (int)(stride * 8);
            stride_b = (unsigned)(stride * 8);
            pbase = (const char*)stream;
            pword = npro > 0 ? pb : sb0;
            // (opaque copies: otherwise hipcc re-reads these kernel arguments from memory in front of every DMA issue)
            asm volatile("" : "+s"(cimg), "+s"(pbase));
            issue_prefetch();
        }
    }
    // LDS image (lane offset applied) of the next operator use  (KPER > 0: per-operator mode with pieces = KPER x nwaves, see dma_k)
    template <int KPER = 0>
    __device__ __forceinline__ const double* next()
    {
        const double* M;
        if (WIN || batch < 0) {
            const unsigned e = (unsigned)qword & 63u, kind = e & 3u, tp = e >> 2;
            unsigned off;
            if (kind == 2) {
                off = (unsigned)(TPS * slot_bytes) + tp * stride_b;
            } else {
                unsigned sl = (unsigned)s0 + tp;
                if (sl >= TPS) sl -= TPS;
                off = sl * (unsigned)slot_bytes + kind * stride_b;
            }
            qword >>= 6;
            if (Q >= npro) {
                ++iq;
                if (iq == 10) qword = sb1;
                if (iq == 20) qword = sb2;
                if (iq == period) {
                    iq = 0;
                    qword = sb0;
                }
            } else if (Q + 1 == npro) {
                qword = sb0;
            }
            ++Q;
            return (const double*)(smem + off) + lane;
        }
        if (batch > 0) {
            int kind, tp;
            entry_at(Q, ip, kind, tp);
            const int n = (Q < npro) ? 0 : np;
            if (kind == 2) {
                M = (const double*)(smem + 2 * (size_t)slot_bytes) + (size_t)tp * stride;
            } else {
                const int b = n / batch, nl = n - b * batch;
                M = (const double*)(smem + (size_t)(b & 1) * slot_bytes) + (size_t)(2 * (2 * nl + tp) + kind) * stride;
            }
            advance(Q, np, ip);
            return M + lane;
        }
        // Publish operator use Q (every wave drains its DMA pieces, then a workgroup barrier), start the
        // fetch of use Q+1 into the slot that use Q-1 just released.
        if (!(debug & 8)) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issue_prefetch<KPER>();
        }
        M = (const double*)(smem + (size_t)(Q & 1) * slot_bytes) + lane;
        ++Q;
        return M;
    }
    // The slab kernels know at every call site which image comes next (the schedule is theirs): K (KIND 0) / S (KIND 1) of
    // time point 2n + TP, or constant image #idx.  In window mode that is two scalar instructions instead of the generic
    // cursor (which was 10 % of a quad-layout step); the other modes ignore the hint.
    template <int KIND, int TP>
    __device__ __forceinline__ const double* next_ks()
    {
        if constexpr (WIN) __builtin_amdgcn_sched_barrier(0);
        if (WIN || batch < 0) return (const double*)(smem + ((TP == 0 ? wb0 : TP == 1 ? wb1 : wb2) + KIND * stride_b)) + lane;
        return next();
    }
    __device__ __forceinline__ const double* next_c(int idx)
    {
        if constexpr (WIN) __builtin_amdgcn_sched_barrier(0);
        if (WIN || batch < 0) return (const double*)(smem + ((unsigned)(TPS * slot_bytes) + (unsigned)idx * stride_b)) + lane;
        return next();
    }
    __device__ __forceinline__ void drain()
    {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
};
typedef RingT<false> Ring;

// ---------------------------------------------------------------------------------------------
// Scaled, signed operator stream.  With c = h/2 the tile stream holds (k_stream)
//     Kp = +c K(t)  at the half time points (odd j),   Kn = -c K(t) at the integer time points (even j),
//     S  =  c S(t)  everywhere,
// so every update of the Stormer-Verlet step is of the form  D = C + M x  and lands directly in the MFMA
// accumulator registers (no axpy passes), and the truncated Neumann series (neumann!,
// src/linear_solvers.jl:81-106) is evaluated in Horner form with the same operator:
//     sum_{j=0..m} S^j A  =  A + S (A + S (A + ... ))                                   (m products).
// The adjoint variable lambda_i is carried negated (nb = -lambda_i), which makes the adjoint step!
// (src/StormerVerlet.jl:255-303) use K05 with '+' and K0, K1 with '-' exactly like the state step.

// out = base_plus_A + sum_{j=1..m} S^j A      (i.e. base + sum_{j=0..m} S^j A with base_plus_A = base + A)
// Ya, Yb are scratch arrays.  out may alias base_plus_A; out must not alias A, Ya, Yb.
// sum over the lane's elements of (x - y)^2
template <int NT>
__device__ __forceinline__ double a_diff2(const Arr<NT>& x, const Arr<NT>& y)
{
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const auto d = x.t[i] - y.t[i];
        s += row_sum(d * d);
    }
    return s;
}

// Sum of a per-lane value over the lanes of ONE SAMPLE of a slab (slab layout: column = lane & 15, four lane rows per column;
// a sample = N consecutive columns): the same value in every lane of the sample, columns added in a fixed order.  N >= 16 (a
// sample wider than the slab): the slab's 16 columns.
__device__ __forceinline__ double sample_sum(double x, int N)
{
    x += __shfl_xor(x, 16);
    x += __shfl_xor(x, 32);      // column totals
    const int lane = threadIdx.x & 63, col = lane & 15;
    const int n = N < 16 ? N : 16, c0 = col - col % n;
    double s = 0.0;
    for (int k = 0; k < n; ++k) {
        const double v = __shfl(x, (lane & 48) | ((c0 + k) & 15));
        s += (c0 + k < 16) ? v : 0.0;      // (the unused tail columns of a ragged slab form a short group of their own)
    }
    return s;
}

// JACOBI_SOLVER (jacobi!, src/linear_solvers.jl:110-153): X_j = A + S X_{j-1}, X_0 = A, stop at the first
// j with ||X_j - X_{j-1}||_F < tol or at j = max_iter.  It is the same fixed-point iteration as the Horner
// form above, plus the convergence test.  The reference solves each evaluation's Ntot x N block on its own, so the test is
// PER SAMPLE here too (round 3; rounds 1-2 tested the whole 16-column slab and agreed with the reference only to O(tol)): the
// residual norm is summed over the lanes of a sample (sample_sum), a sample that has converged keeps its iterate while the wave
// iterates on for the others.  N > 16 (a sample spans several slabs = waves): per 16-column part.
// jac_wg >= 0 (N > 16 with the <= 4 parts of a sample on the waves of ONE workgroup, round 5): the residual norm of the reference is over
// the whole Ntot x N block (src/linear_solvers.jl:121), so the parts' squared norms are added through LDS -- one workgroup barrier per
// iteration, the decision is workgroup-uniform and every wave iterates equally long.  ([2][JQ_WAVES] doubles at byte offset jac_wg of
// the dynamic LDS, double-buffered by the iteration's parity.)
__device__ __forceinline__ double jacobi_wg_sum(double r, int jac_wg, int it)
{
    extern __shared__ __attribute__((aligned(16))) char jq_smem_[];
    double* ex = (double*)(jq_smem_ + jac_wg) + (it & 1) * JQ_WAVES;
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) ex[wave] = r;      // (r is the same in all lanes: the slab's 16 columns, sample_sum)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    double s = 0.0;
    for (int w = 0; w < nw; ++w) s += ex[w];
    return s;
}
template <int NT, int BW>
__device__ __forceinline__ void jacobi_add(Arr<NT>& out, const Arr<NT>& bpa, const Arr<NT>& A, const double* S, int max_iter,
                                           double tol2, Arr<NT>& Ya, Arr<NT>& Yb, int ncol, int jac_wg = -1)
{
    // out = bpa - A + X_j
    if (max_iter <= 0) {
        out = bpa;
        return;
    }
    if (jac_wg >= 0) {      // (one sample per workgroup: the same iteration count in every wave)
        // The exchange buffer is chosen by the parity of the iteration, which starts again at 1 in every call: a call that stopped on an
        // odd iteration is followed by a sum in the SAME buffer, and nothing but the barriers inside jacobi_wg_sum separates the calls
        // of a time step (window staging has no barrier between operator uses).  A fast wave could then overwrite ex[1][wave] while a
        // slow one still adds up the previous call's last sum -- different `done` decisions, barrier counts that no longer match.  One
        // barrier at the entry closes it: every wave has finished reading the previous call's sums before any wave writes a new one.
        __builtin_amdgcn_s_barrier();
        mm_c<NT, BW>(Ya, A, S, A);  // X_1
        bool in_a = true;
        bool done = jacobi_wg_sum(sample_sum(a_diff2(Ya, A), ncol), jac_wg, 1) < tol2;
        for (int j = 2; j <= max_iter && !done; ++j) {
            if (in_a) {
                mm_c<NT, BW>(Yb, A, S, Ya);
                done = jacobi_wg_sum(sample_sum(a_diff2(Yb, Ya), ncol), jac_wg, j) < tol2;
            } else {
                mm_c<NT, BW>(Ya, A, S, Yb);
                done = jacobi_wg_sum(sample_sum(a_diff2(Ya, Yb), ncol), jac_wg, j) < tol2;
            }
            in_a = !in_a;
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) out.t[i] = (bpa.t[i] - A.t[i]) + (in_a ? Ya.t[i] : Yb.t[i]);
        return;
    }
    mm_c<NT, BW>(Ya, A, S, A);  // X_1
    bool in_a = true;
    bool done = sample_sum(a_diff2(Ya, A), ncol) < tol2;      // (per lane: the same for all lanes of a sample)
    for (int j = 2; j <= max_iter && !__all(done); ++j) {
        if (in_a) {
            mm_c<NT, BW>(Yb, A, S, Ya);
            const bool conv = sample_sum(a_diff2(Yb, Ya), ncol) < tol2;
#pragma unroll
            for (int i = 0; i < NT; ++i) Yb.t[i] = done ? Ya.t[i] : Yb.t[i];      // converged samples keep their iterate
            done = done || conv;
        } else {
            mm_c<NT, BW>(Ya, A, S, Yb);
            const bool conv = sample_sum(a_diff2(Ya, Yb), ncol) < tol2;
#pragma unroll
            for (int i = 0; i < NT; ++i) Ya.t[i] = done ? Yb.t[i] : Ya.t[i];
            done = done || conv;
        }
        in_a = !in_a;
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) out.t[i] = (bpa.t[i] - A.t[i]) + (in_a ? Ya.t[i] : Yb.t[i]);
}

// REGOP (quad layout): the operator of the chain's m + 1 products is held in 10 NT registers (OpQ); false: every product reads it from
// LDS again (mm_t4q) -- 60 registers less inside the recurrence at NT = 6
template <int NT, int BW, bool JAC, bool REGOP = true>
__device__ __forceinline__ void horner_add(Arr<NT>& out, const Arr<NT>& bpa, const Arr<NT>& A, const double* S, int m,
                                           Arr<NT>& Ya, Arr<NT>& Yb, double jacobi_tol2, int ncol, int jac_wg = -1)
{
    if (JAC) {
        jacobi_add<NT, BW>(out, bpa, A, S, m, jacobi_tol2, Ya, Yb, ncol, jac_wg);
        return;
    }
    if (m <= 0) {
        out = bpa;
        return;
    }
    int rem = m - 1;  // Horner updates before the final product
    if (rem == 0) {
        mm_c<NT, BW>(out, bpa, S, A);
        return;
    }
#ifdef JQ_EXP_NOOPQ
    if constexpr (false) {
#else
    if constexpr (BW == JQ_BW_T4Q && REGOP) {
#endif
        OpQ<NT> op;
        t4q_load(op, S);
        mm_t4q_regs(Ya, A, op, A);
        // the recurrence two products per loop iteration: a taken branch costs 60 - 80 cycles of the SIMD's issue time even with three
        // waves to choose from (probes/lone_wave_probe.hip; 3 072 cnot3 samples 1 152 -> 1 133 ms, 2 048: 853 -> 827 ms; four per iteration
        // or a fully unrolled switch lose again: code size, 244 B of scratch)
        for (--rem; rem >= 2; rem -= 2) {
            mm_t4q_regs(Ya, A, op, Ya);
            mm_t4q_regs(Ya, A, op, Ya);
        }
        if (rem > 0) mm_t4q_regs(Ya, A, op, Ya);
        mm_t4q_regs(out, bpa, op, Ya);
        return;
    }
    mm_c<NT, BW>(Ya, A, S, A);  // Y1 = A + S A
    --rem;
    if constexpr (BW == JQ_BW_OD || BW == JQ_BW_T4 || BW == JQ_BW_T4Q) {
        // mm_od / mm_t4 are alias-safe: the recurrence runs in place and Yb is never touched (48 registers less)
        for (; rem > 0; --rem) mm_c<NT, BW>(Ya, A, S, Ya);
        mm_c<NT, BW>(out, bpa, S, Ya);
        return;
    }
    while (rem >= 2) {
        mm_c<NT, BW>(Yb, A, S, Ya);
        mm_c<NT, BW>(Ya, A, S, Yb);
        rem -= 2;
    }
    if (rem == 1) {
        mm_c<NT, BW>(Yb, A, S, Ya);
        mm_c<NT, BW>(out, bpa, S, Yb);
    } else {
        mm_c<NT, BW>(out, bpa, S, Ya);
    }
}

// out += sum_{j=1..m} S^j A TERM BY TERM (neumann!, src/linear_solvers.jl:81-106, literally): T_1 = S A, T_{j+1} = S T_j, the terms ping-pong between
// Ya and A's OWN registers (A is destroyed), every term is added to out.  Three arrays where the Horner form needs four (out, A, Ya, Yb) --
// m array additions more per call.  For the widest slab kernels (six tile rows, band / dense tiles): their backward sweep keeps eight
// arrays alive through a recurrence and spills two of them (k_backward<6, 5>: 388 B of scratch per lane in round 5).
#ifndef JQ_BWD_TERMS
#define JQ_BWD_TERMS 1
#endif
constexpr bool jq_bwd_terms(int NT, int BW, bool JAC) { return JQ_BWD_TERMS && !JAC && NT >= 6 && BW != JQ_BW_OD && BW != JQ_BW_T4 && BW != JQ_BW_T4Q; }
template <int NT, int BW>
__device__ __forceinline__ void neumann_terms_add(Arr<NT>& out, Arr<NT>& A, const double* S, int m, Arr<NT>& Ya)
{
    for (int j = 0; j < m; j += 2) {
        mm_z<NT, BW>(Ya, S, A);
        a_add(out, Ya);
        if (j + 1 < m) {
            mm_z<NT, BW>(A, S, Ya);
            a_add(out, A);
        }
    }
}

// LDS / global parking of a dormant state array (one [4*NT][64] image per wave)
template <int NT>
__device__ __forceinline__ void a_park(const Arr<NT>& a, __attribute__((address_space(3))) double* park)
{
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < JQ_RL; ++r) park[(JQ_RL * i + r) * 64] = a.t[i][r];
}
template <int NT>
__device__ __forceinline__ void a_unpark(Arr<NT>& a, const __attribute__((address_space(3))) double* park)
{
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < JQ_RL; ++r) a.t[i][r] = park[(JQ_RL * i + r) * 64];
}
template <int NT>
__device__ __forceinline__ void a_park(const Arr<NT>& a, double* park)
{
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < JQ_RL; ++r) park[(JQ_RL * i + r) * 64] = a.t[i][r];
}
template <int NT>
__device__ __forceinline__ void a_unpark(Arr<NT>& a, const double* park)
{
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < JQ_RL; ++r) a.t[i][r] = park[(JQ_RL * i + r) * 64];
}

// Which sweeps use the fused passes of sv_state (measured with scripts/exp_variants.sh at cnot3 x 3 072 samples, DESIGN.md
// section 6, round 3): forward 337 -> 318 ms -- most of it (-17 ms) from having four stages instead of six (fewer branch fences),
// the shared lane shifts add -2.5 ms; the twelve-wave backward kernel loses with ANY of it in its state step (104 -> 144 .. 540 B
// of scratch: 838 -> 915 .. 1 400 ms) and gains 6 ms from K0 X / K1 X in one pass (no extra registers).
#ifndef JQ_FWD_FUSE
#define JQ_FWD_FUSE 3
#endif
#ifndef JQ_BWD_FUSE
#define JQ_BWD_FUSE 3
#endif
#ifndef JQ_BWD_FUSE3      // ... of the twelve-wave backward kernel (168 registers per wave)
#define JQ_BWD_FUSE3 0
#endif
#ifndef JQ_BWD_ADJ_FUSE   // adjoint step: K0 X and K1 X in one pass
#define JQ_BWD_ADJ_FUSE 1
#endif
#ifndef JQ_BWD3_NOOPQ     // twelve-wave backward kernel: Neumann recurrences of the ADJOINT step with the operator re-read from LDS per product
#define JQ_BWD3_NOOPQ 0
#endif
#ifndef JQ_BWD3_NOOPQ_STATE   // ... and of the state re-integration inside it
#define JQ_BWD3_NOOPQ_STATE 0
#endif
// State (re-)integration, operator uses 0..5 of one Stormer-Verlet step (forward step!,
// src/StormerVerlet.jl:461-504, also used with h<0 by the backward sweep, src/evalobjgrad.jl:879).
//   in : u (preserved), v (CONSUMED: overwritten in place by v05 = v(t+h/2))
//   out: unew = u(t+h), vN = v05 + S05 v05 (the caller finishes v(t+h) = vN + Kp05 unew with use 6)
//   A, Ya, Yb: scratch arrays.          Operator order per step: Kp05 S05 Kn0 S0 Kn1 S1 (Kp05).
// At most 8 arrays are live here (u, v/v05, unew, vN, A, Ya, Yb + one of the caller's).
// FUSE (quad layout only): bit 0 = K05 u with S0 u in one pass, bit 1 = S05 v05 with K0 v05 and K1 v05 in one pass
// FOLD (quad layout, one sample per wave): `ceps` is the per-lane MASKED shift (+c eps on the lanes that hold the diagonal of the MFMA's
// A operand, 0 elsewhere; 0 everywhere without a shift) and every product with a K image folds it into its operand (mm_t4q SH)
template <int NT, int BW, bool JAC, int FUSE = 0, bool FOLD = false, bool REGOP = true, bool TERMS = false, typename RING = RingT<BW == JQ_BW_T4Q>>
__device__ __forceinline__ void sv_state(RING& p, const PropArgs& a, bool active, double ceps, const double* ws, int g,
                                         const Arr<NT>& u, Arr<NT>& v, Arr<NT>& unew, Arr<NT>& vN, Arr<NT>& A, Arr<NT>& Ya,
                                         Arr<NT>& Yb)
{
    if constexpr (BW == JQ_BW_T4Q && !JAC && FUSE != 0) {
        // Quad layout (always window staging: every image of the step is resident, the order of the uses is free): the products
        // that share a right-hand side are made in ONE pass over the blocks so that they share its lane shifts -- K05 u with
        // S0 u, and S05 v05 with K0 v05 and K1 v05: 5 + 2m passes instead of 8 + 2m (the products are the same 8 + 2m).
        // (One `if (active)` block per stage like the generic path below: the branches keep hipcc from hoisting the operand
        // reads of later stages -- one basic block for the whole step spills 200+ registers in the twelve-wave variants.)
        const double* M0 = p.template next_ks<0, 1>();      // Kp05
        const double* M1 = p.template next_ks<1, 0>();      // S0
        if (active) {
            if constexpr (FOLD) {
                static_assert(!FOLD || (FUSE & 3) == 3 || FUSE == 0, "FOLD: fused or generic path");
                mm_t4q2<NT, true, false, 1>(A, A, M0, unew, u, M1, u, ceps, 0.0, ws + g);
            } else {
            if constexpr (FUSE & 1) {
                mm_t4q2<NT, true, false>(A, A, M0, unew, u, M1, u);        // A = c K05 u ;  unew = u + c S0 u
            } else {
                mm_z<NT, BW>(A, M0, u);
                mm_c<NT, BW>(unew, u, M1, u);
            }
            if (a.use_shift) a_axpy_rows(A, ceps, ws, g, u);
            }
        }
        M0 = p.template next_ks<1, 1>();                    // S05
        if (active) {
            mm_c<NT, BW>(A, A, M0, v);                                 // A = c (K05 u + S05 v)
            a_add(v, A);
            horner_add<NT, BW, JAC, REGOP>(v, v, A, M0, a.m, Ya, Yb, a.jacobi_tol2, a.N, a.jac_wg_lds);       // v = v05
        }
        M1 = p.template next_ks<0, 0>();                    // Kn0
        const double* M2 = p.template next_ks<0, 2>();      // Kn1
        if (active) {
            if constexpr (FOLD) {
                mm_t4q3<NT, false, false, true, 6>(vN, v, M0, unew, unew, M1, A, A, M2, v, 0.0, -ceps, -ceps, ws + g);
            } else {
            if constexpr (FUSE & 2) {
                mm_t4q3<NT, false, false, true>(vN, v, M0, unew, unew, M1, A, A, M2, v);     // vN = v05 + S05 v05 ; unew -= c K0 v05 ; A = -c K1 v05
            } else {
                mm_c<NT, BW>(vN, v, M0, v);
                mm_c<NT, BW>(unew, unew, M1, v);
                mm_z<NT, BW>(A, M2, v);
            }
            if (a.use_shift) {
                a_axpy_rows(unew, -ceps, ws, g, v);
                a_axpy_rows(A, -ceps, ws, g, v);
            }
            }
        }
        M0 = p.template next_ks<1, 2>();                    // S1
        if (active) {
            mm_c<NT, BW>(A, A, M0, unew);                              // A = c (S1 (u + c kappa1) - K1 v05)
            a_add(unew, A);
            horner_add<NT, BW, JAC, REGOP>(unew, unew, A, M0, a.m, Ya, Yb, a.jacobi_tol2, a.N, a.jac_wg_lds);
        }
        return;
    }
    constexpr int FULLQ = JQ_T4_DIAG | JQ_T4_RTERMS | JQ_T4_MTERMS;
    // use 0: Kp05 -- A = c K05 u
    const double* M = p.template next_ks<0, 1>();
    if (active) {
        if constexpr (FOLD) {
            mm_t4q<NT, true, FULLQ, true>(A, A, M, u, ceps, ws + g);
        } else {
            mm_z<NT, BW>(A, M, u);
            if (a.use_shift) a_axpy_rows(A, ceps, ws, g, u);
        }
    }
    // use 1: S05 -- A = c (K05 u + S05 v) ; v05 = v + sum_j S^j A (in place) ; vN = v05 + S05 v05
    M = p.template next_ks<1, 1>();
    if (active) {
        mm_c<NT, BW>(A, A, M, v);
        a_add(v, A);
        if constexpr (TERMS) neumann_terms_add<NT, BW>(v, A, M, a.m, Ya);
        else horner_add<NT, BW, JAC, REGOP>(v, v, A, M, a.m, Ya, Yb, a.jacobi_tol2, a.N, a.jac_wg_lds);
        mm_c<NT, BW>(vN, v, M, v);
    }
    // use 2: Kn0 -- unew = u - c K0 v05
    M = p.template next_ks<0, 0>();
    if (active) {
        if constexpr (FOLD) {
            mm_t4q<NT, false, FULLQ, true>(unew, u, M, v, -ceps, ws + g);
        } else {
            mm_c<NT, BW>(unew, u, M, v);
            if (a.use_shift) a_axpy_rows(unew, -ceps, ws, g, v);
        }
    }
    // use 3: S0 -- unew = u + c (S0 u - K0 v05) = u + c kappa1
    M = p.template next_ks<1, 0>();
    if (active) mm_c<NT, BW>(unew, unew, M, u);
    // use 4: Kn1 -- A = -c K1 v05
    M = p.template next_ks<0, 2>();
    if (active) {
        if constexpr (FOLD) {
            mm_t4q<NT, true, FULLQ, true>(A, A, M, v, -ceps, ws + g);
        } else {
            mm_z<NT, BW>(A, M, v);
            if (a.use_shift) a_axpy_rows(A, -ceps, ws, g, v);
        }
    }
    // use 5: S1 -- A = c (S1 (u + c kappa1) - K1 v05) ; unew += sum_j S^j A
    M = p.template next_ks<1, 2>();
    if (active) {
        mm_c<NT, BW>(A, A, M, unew);
        a_add(unew, A);
        if constexpr (TERMS) neumann_terms_add<NT, BW>(unew, A, M, a.m, Ya);
        else horner_add<NT, BW, JAC, REGOP>(unew, unew, A, M, a.m, Ya, Yb, a.jacobi_tol2, a.N, a.jac_wg_lds);
    }
}

// ---------------------------------------------------------------------------------------------
// Forward sweep over one chunk of time steps (src/evalobjgrad.jl:698-753).
// schedule (period 7): Kp05 S05 Kn0 S0 Kn1 S1 Kp05
// WLRT: low-rank full leakage weights (PropArgs::wlr) compiled in -- dedicated instantiations only (quad layout with one slab per
// workgroup; the slab kernels <1, 0> and <6, 5>, which have no cooperative sibling), so every Diagonal fast path is untouched.
// (A run-time test of a.wrank in the band / dense / JQ_BW_OD slab kernels was measured first: the never-taken branches cost
//  their register allocation 17 - 23 % -- cnot3 on <6, 9> 342 -> 402 ms, on <6, 1> 564 -> 692 ms per 4 000 steps.)
template <int BW, bool JAC, bool WLRT>
constexpr bool jq_wlr_on() { return WLRT; }
template <int NT, int BW, int MINW, bool JAC, bool WLRT = false, bool UNI = false>      // (UNI: see k_backward)
__global__ __launch_bounds__((BW == JQ_BW_T4Q) ? 256 * MINW : 256, (BW == JQ_BW_T4Q) ? 1 : MINW) void k_forward(PropArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KT = 4 * NT;
    const int lane_ = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // slab layout: one slab per wave, lane = 16 g + column.  Quad layout (JQ_BW_T4Q): the workgroup's four waves share one
    // slab, wave q carries its columns 4q .. 4q+3; `lane` / `g` are then this lane's offsets in a block of the slab image / of
    // the row tables (a_load, a_axpy_rows) and `col` its state column
    // (quad layout with MINW == 2: workgroups of eight waves = two slabs, two waves per SIMD)
    constexpr bool QUAD = (BW == JQ_BW_T4Q);
    constexpr int NWAVES = QUAD ? JQ_WAVES * MINW : JQ_WAVES;
    const int col = QUAD ? 4 * (wave & 3) + (lane_ & 3) : (lane_ & 15);
    const int lane = QUAD ? ((lane_ >> 2) & 3) * 64 + 16 * (lane_ >> 4) + col : lane_;
    const int g = QUAD ? 4 * (lane_ >> 4) + ((lane_ >> 2) & 3) : lane_ >> 4;
    // (JAC with one workgroup per sample, a.jac_wg_lds >= 0: the workgroup has as many waves as the sample has parts)
    const int wpw = (JAC && !QUAD) ? (int)(blockDim.x >> 6) : JQ_WAVES;
    const int slab = QUAD ? (int)blockIdx.x * (NWAVES / 4) + (wave >> 2) : blockIdx.x * wpw + wave;
    const bool active = slab < a.nslabs;

    double* tab = (double*)(smem + a.lds_tab_off);
    const double* wd = tab;
    const double* ws = tab + 16 * NT;
    for (int i = threadIdx.x; i < 32 * NT; i += blockDim.x) tab[(i & ~15) + 4 * (i & 3) + ((i >> 2) & 3)] = a.tabs[i];   // [block][g][r]

    Arr<NT> ua, va, ub, vb, A, Ya, Yb;
    double leak = 0.0, ceps = 0.0;
    double* st = a.state + (size_t)(active ? slab : 0) * a.state_stride;
    if (active) {
        a_load(ua, st, lane);
        a_load(va, st + KT * 64, lane);
        // per-lane partial of the leak integral: the slab image has one slot per (row in group, column); in the quad layout the
        // lanes of group 0 carry it between chunks
        leak = (!QUAD || ((lane_ >> 2) & 3) == 0) ? st[(JQ_STATE_ARRAYS * KT + JQ_MAXNC) * 64 + (QUAD ? 16 * (lane_ >> 4) + col : lane)] : 0.0;
        ceps = 0.5 * a.h * a.colinfo[(size_t)slab * 32 + col];
        if constexpr (UNI) ceps = (a.use_shift && (lane_ >> 4) == (lane_ & 3)) ? lane_bcast(ceps, 0) : 0.0;      // (masked: folded into the A operands)
    } else {
        a_zero(ua);
        a_zero(va);
    }
    RingT<QUAD> p;
    p.init(smem, a, wave, lane_, QUAD ? NWAVES : wpw);
    constexpr bool WLR = jq_wlr_on<BW, JAC, WLRT>();
    WLow<NT, QUAD> wl;
    if constexpr (WLR) wl.init(a, lane_, smem);

    // one time step: (u, v) -> (unew, vN); v is consumed (becomes v05).  The two array pairs swap
    // roles every step, so the loop body is written for two steps and nothing is ever copied.
#define JQ_FWD_STEP(U, V, UN, VN, NSTEP)                                                                         \
    {                                                                                                            \
        p.begin_step(NSTEP);                                                                                     \
        if (active) leak += a_wsq(wd, g, U); /* trapezoidal part: tr(vr' W vr) at t_n (:700) */                  \
        sv_state<NT, BW, JAC, JQ_FWD_FUSE, UNI>(p, a, active, ceps, ws, g, U, V, UN, VN, A, Ya, Yb);                  \
        /* use 6: Kp05 again -- v(t+h) = v05 + c (K05 u_new + S05 v05) */                                        \
        const double* M6 = p.template next_ks<0, 1>();                                                                             \
        if (active) {                                                                                            \
            if constexpr (UNI) {                                                                                 \
                mm_t4q<NT, false, JQ_T4_DIAG | JQ_T4_RTERMS | JQ_T4_MTERMS, true>(VN, VN, M6, UN, ceps, ws + g);   \
            } else {                                                                                             \
                mm_c<NT, BW>(VN, VN, M6, UN);                                                                    \
                if (a.use_shift) a_axpy_rows(VN, ceps, ws, g, UN);                                               \
            }                                                                                                    \
            /* leak integrand: tr(vr' W vr + 2 vi05' W vi05) after the step (:716, penalf2a :2170-2180) */       \
            leak += a_wsq(wd, g, UN) + 2.0 * a_wsq(wd, g, V);                                                    \
            if constexpr (WLR)                                                                                   \
                if (wl.r > 0) {   /* full weights: tr(vr' Wr vr) at t_n and t_n+1, 2 tr(vi05' Wr vi05), -2 tr(vi05' Wi vr(t_n)) (:700, :716-718) */ \
                    double lk = 0.0;                                                                             \
                    for (int k = 0; k < wl.r; ++k) {                                                             \
                        double p0, q0, p1, q1, rr, ss;                                                           \
                        if (wl.has_sc) p0 = wl.get(k, 0), q0 = wl.get(k, 1);      /* last step's dots with vr(t_n+1) */ \
                        else wl.dots(k, U, p0, q0);                                                              \
                        wl.dots(k, UN, p1, q1);                                                                  \
                        if (wl.has_sc) wl.put(k, 0, p1), wl.put(k, 1, q1);                                       \
                        wl.dots(k, V, rr, ss);                                                                   \
                        lk += wl.lam(k) * ((p0 * p0 + q0 * q0) + (p1 * p1 + q1 * q1) + 2.0 * (rr * rr + ss * ss) - 2.0 * (ss * p0 - rr * q0)); \
                    }                                                                                            \
                    if (wl.lead) leak += lk;                                                                     \
                }                                                                                                \
            if (a.hist_r) hist_store<NT>(a, slab, col, QUAD ? 4 * ((lane_ >> 2) & 3) + (lane_ >> 4) : g, NSTEP, UN, VN); \
        }                                                                                                        \
    }
    if constexpr (WLR)
        if (wl.has_sc && active)      // (the first step's dots with vr(t_0) of the chunk)
            for (int k = 0; k < wl.r; ++k) {
                double p0, q0;
                wl.dots(k, ua, p0, q0);
                wl.put(k, 0, p0), wl.put(k, 1, q0);
            }
    int n = 0;
    for (; n + 1 < a.nsteps_chunk; n += 2) {
        JQ_FWD_STEP(ua, va, ub, vb, n)
        JQ_FWD_STEP(ub, vb, ua, va, n + 1)
    }
    if (n < a.nsteps_chunk) {
        JQ_FWD_STEP(ua, va, ub, vb, n)
        if (active) {
            ua = ub;
            va = vb;
        }
    }
#undef JQ_FWD_STEP
    p.drain();  // land the trailing prefetches before the workgroup exits
    if (active) {
        a_store(ua, st, lane);
        a_store(va, st + KT * 64, lane);
        if constexpr (QUAD) {
            leak = row_ror_add<8>(row_ror_add<4>(leak));   // sum over the four groups of a block (lanes 4 b + j of a row)
            if (((lane_ >> 2) & 3) == 0) st[(JQ_STATE_ARRAYS * KT + JQ_MAXNC) * 64 + 16 * (lane_ >> 4) + col] = leak;
        } else {
            st[(JQ_STATE_ARRAYS * KT + JQ_MAXNC) * 64 + lane] = leak;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Backward sweep over one chunk (src/evalobjgrad.jl:859-921): state re-integration with h<0,
// adjoint step! with forcing (src/StormerVerlet.jl:255-303) or step_no_forcing! (:365-451), and the
// per-step trace scalars of adjoint_grad_calc! (src/evalobjgrad.jl:2567-2619), written to `traces`.
// schedule (period 13 + 3*Ncoupled):
//   Kp05 S05 Kn0 S0 Kn1 S1 Kp05 | S0 | Hanti_0 .. | Kn0 Kn1 S05 Kp05 S1 | Hanti_0 Hsym_0 Hanti_1 Hsym_1 ...
// State file slot NU holds nb = -lambda_i.
// Register budget (512 per lane, 8 per array element pair): at most 9 state-sized arrays are live at
// any point (8 since vr0 dies after the early traces); the one array that is dormant in each phase (lambda_r during the state step, v during the
// adjoint step and the traces) is parked in the wave's LDS (or global) parking image.
// UNI: every wave's columns belong to ONE ensemble sample (quad layout with N a multiple of 4, or N > 16): its shift and weight
// are wave-uniform and live in scalar registers -- four vector registers less in the 168-register three-slab kernel (round 4: with
// the LDS parking pointer 136 -> 108 B of scratch, 60 -> 38 scratch instructions per step, 1 098 -> 1 085 ms; results bit-identical)
// ORD (with UNI, quad layout): control q acts on subsystem q only -- the usual Juqbox set-up, Hsym_ops = [a + a', b + b', c + c'];
// host: a.bw_trace[q] == 1 << q for all q < Ncoupled, 2 <= Ncoupled <= 3.  The trace products are then one part of a product each at
// compile time (no mode dispatch), and Hsym_1 lambda_i_new of the MIDDLE subsystem's control (lane-shift couplings only) rides along
// in the pass that shifts lambda_i_new anyway (K05 lambda_i_new, use 11): backward sweep 755 -> 737 ms (round 4).  Letting Hanti_1 X and
// Hsym_1 X ride along with K0 X / K1 X as well keeps vr(t_n+1) alive through that pass: 464 B of scratch, 1 327 ms -- rejected.
template <int NT, int BW, int MINW, bool JAC, bool WLRT = false, bool UNI = false, bool ORD = false>
__global__ __launch_bounds__((BW == JQ_BW_T4Q) ? 256 * MINW : 256, (BW == JQ_BW_T4Q) ? 1 : MINW) void k_backward(PropArgs a)
{
    static_assert(!ORD || (UNI && BW == JQ_BW_T4Q && !JAC), "ORD: a variant of the UNI quad-layout kernel");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KT = 4 * NT;
    const int lane_ = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // (slab / quad layout: see k_forward)
    constexpr bool QUAD = (BW == JQ_BW_T4Q);
    constexpr int NWAVES = QUAD ? JQ_WAVES * MINW : JQ_WAVES;
    constexpr int NTHREADS = 64 * NWAVES;
    const int col = QUAD ? 4 * (wave & 3) + (lane_ & 3) : (lane_ & 15);
    const int lane = QUAD ? ((lane_ >> 2) & 3) * 64 + 16 * (lane_ >> 4) + col : lane_;
    const int g = QUAD ? 4 * (lane_ >> 4) + ((lane_ >> 2) & 3) : lane_ >> 4;
    const int wpw = (JAC && !QUAD) ? (int)(blockDim.x >> 6) : JQ_WAVES;      // (see k_forward)
    const int slab = QUAD ? (int)blockIdx.x * (NWAVES / 4) + (wave >> 2) : blockIdx.x * wpw + wave;
    const bool active = slab < a.nslabs;
    const int Nc = a.Ncoupled;
    // per-lane trace carries in the array file: like the leak partial of k_forward
    const bool cslot = !QUAD || ((lane_ >> 2) & 3) == 0;
    const int clane = QUAD ? 16 * (lane_ >> 4) + col : lane_;

    double* tab = (double*)(smem + a.lds_tab_off);
    const double* wd = tab;
    const double* ws = tab + 16 * NT;
    double* carry = tab + 32 * NT;  // [Ncoupled][threads of the workgroup]
    // forcing weight c*tinv: c*hr0 = c*tinv*W*vr etc. (:862, :882-888); 0 for step_no_forcing!.  The table wd carries it (round 4:
    // cfw * wd[row] was formed per element and use -- the same rounded product, three times six multiplies per step)
    const double cfw = a.forced ? 0.5 * a.h * a.tinv : 0.0;
    for (int i = threadIdx.x; i < 32 * NT; i += blockDim.x)
        tab[(i & ~15) + 4 * (i & 3) + ((i >> 2) & 3)] = (i < 16 * NT) ? cfw * a.tabs[i] : a.tabs[i];   // [block][g][r]
    double* st = a.state + (size_t)(active ? slab : 0) * a.state_stride;
    // parking image of this wave: in LDS when it fits, else in HBM.  The quad-layout kernels always park in LDS: a 32-bit LDS pointer
    // (ds_read / ds_write) instead of a generic one -- flat_load / flat_store carry a 64-bit address per lane and count on vmcnt AND
    // lgkmcnt (round 4: cnot3 x 3 072 samples 1 098 -> 1 090 ms, fewer spilled registers)
    typedef __attribute__((address_space(3))) double lds_double;
    typedef typename std::conditional<QUAD, lds_double*, double*>::type park_ptr;
    park_ptr P0;
    if constexpr (QUAD)
        P0 = (lds_double*)(carry + Nc * NTHREADS + (size_t)wave * (JQ_RL * NT) * 64 + lane_);
    else
        P0 = a.park_lds ? (carry + Nc * NTHREADS + (size_t)wave * (JQ_RL * NT) * 64 + lane_)
                        : (a.park + (size_t)(active ? slab : 0) * KT * 64 + lane);
    // Per-step trace scalars: every wave leaves its Ncoupled * JQ_NTR wave sums of step n in the LDS record
    // rec[n & 1][wave][8 Ncoupled] (one wave_sum4 group of four slots per control for the early values t1, t3 and one for
    // the late values t2, t4, t5; a group's values a, b, c sit in the rows = slots 0, 2, 1); once the whole workgroup has passed a barrier behind step n, wave 0 adds the waves' records in
    // wave order and writes ONE record per workgroup and step to HBM (a.traces) -- the records of a launch are
    // [workgroups][steps][ntr] instead of [waves][steps][ntr] (12 x less traffic with three slabs per workgroup).
    const int ntr = Nc * JQ_NTR;
    double* rec = carry + Nc * NTHREADS + (a.park_lds ? (size_t)NWAVES * (JQ_RL * NT) * 64 : 0);
    const int rslots = 8 * Nc;
    for (int i = threadIdx.x; i < 2 * NWAVES * rslots; i += blockDim.x) rec[i] = 0.0;   // (inactive waves never write theirs)
    auto flush_traces = [&](int k) {
        if (!QUAD && a.batch > 0) {   // batched staging has no workgroup barrier in every step
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        if (wave == 0 && lane_ < ntr) {
            // trace kk of control q: t1, t3 (kk = 0, 2) are the values a, b of the control's early group, t2, t4, t5
            // (kk = 1, 3, 4) the values a, b, c of its late group
            const int q = lane_ / JQ_NTR, kk = lane_ - q * JQ_NTR;
            const int slot = (kk == 0 ? 0 : kk == 2 ? 2 : 4 * Nc + (kk == 1 ? 0 : kk == 3 ? 2 : 1)) + 4 * q;
            const double* r = rec + (size_t)(k & 1) * NWAVES * rslots + slot;
            double s = r[0];
#pragma unroll
            for (int w = 1; w < NWAVES; ++w) s += r[w * rslots];
            a.traces[((size_t)blockIdx.x * a.nsteps_chunk + k) * ntr + lane_] = s;
        }
    };

    // array roles (register arrays are renamed, never copied, except at the end of a step):
    //   u  : vr before the state step (vr0)      un : vr after it
    //   v  : vi -> vi05 (in place)               vN : vi after the step / scratch Q, G
    //   mu : lambda_r -> X = lambda_r^{1/2} (in place)
    //   nb : -lambda_i (old) -> -(li0 + li)      L  : scratch -> -lambda_i (new)
    //   Ya, Yb: Horner scratch (also trace products)
    // Parking (P0): lambda_r sleeps during the state step, vi(t_n) from use 6 to the end of the step.
    Arr<NT> u, v, mu, nb, un, vN, L, Ya, Yb;
    double ceps = 0.0, wgt = 0.0;
    if (active) {
        a_load(u, st, lane);
        a_load(v, st + KT * 64, lane);
        a_load(mu, st + 2 * KT * 64, lane);
        a_load(nb, st + 3 * KT * 64, lane);
        ceps = 0.5 * a.h * a.colinfo[(size_t)slab * 32 + col];
        wgt = a.colinfo[(size_t)slab * 32 + 16 + col];
        if constexpr (UNI) {
            wgt = lane_bcast(wgt, 0);
            // the sample's shift, folded into the A operands of the K products (mm_t4q SH): nonzero on the lanes that hold the
            // diagonal of the operand (k == i), zero without a shift -- the run-time tests of a.use_shift disappear
            ceps = (a.use_shift && (lane_ >> 4) == (lane_ & 3)) ? lane_bcast(ceps, 0) : 0.0;
        }
        for (int q = 0; q < Nc; ++q) carry[q * NTHREADS + threadIdx.x] = cslot ? st[(JQ_STATE_ARRAYS * KT + q) * 64 + clane] : 0.0;
    } else {
        a_zero(u);
        a_zero(v);
        a_zero(mu);
        a_zero(nb);
    }
    RingT<QUAD> p;
    p.init(smem, a, wave, lane_, QUAD ? NWAVES : wpw);
    // full leakage weights in low-rank form (WLow): forcing hr0 = Wr vr(t_n+1) / T, hi0 = Wr vi05 / T, hr1 = (Wr vr(t_n) + Wi vi05) / T,
    // hi1 = hi0 - Wi vr(t_n) / T (src/evalobjgrad.jl:862, :882-888)
    constexpr bool WLR = jq_wlr_on<BW, JAC, WLRT>();
    WLow<NT, QUAD> wl;
    if constexpr (WLR) wl.init(a, lane_, smem);
    const bool wforce = WLR && a.wrank > 0 && a.forced;
    // (twelve-wave kernel, 168 registers per wave: JQ_BWD3_NOOPQ=1 lets the Neumann recurrences read their operator from LDS per product
    //  instead of holding it in 60 registers -- no spill stores left in the time loop, round 5 experiment)
    constexpr bool BREG = !(QUAD && MINW >= 3 && JQ_BWD3_NOOPQ);

    if (a.first_chunk) {
        // carry_q = tr(vr' Hsym_q lambdai) at t = T: the "vr0/lambdai0" term of the first backward
        // step (:2609); on later steps it is the previous step's tr(vr' Hsym_q lambdai) (:901-902).
        for (int q = 0; q < Nc; ++q) {
            const double* M = p.next_c(q);  // Hsym_q
            if (active) {
                mm_z_bw<NT, BW>(Ya, M, nb, a.bw_trace[q]);
                carry[q * NTHREADS + threadIdx.x] = -a_dot(u, Ya);
            }
        }
    }

    if constexpr (WLR)
        if (wforce && wl.has_sc && active)      // (the first step's dots with vr(t_n+1): the state the chunk starts from)
            for (int k = 0; k < wl.r; ++k) {
                double pu, qu;
                wl.dots(k, u, pu, qu);
                wl.put(k, 0, pu), wl.put(k, 1, qu);
            }
    for (int n = 0; n < a.nsteps_chunk; ++n) {
        p.begin_step(n);
        // ---- state step (lambda_r parked) ------------------------------------------------------
        if (active) a_park(mu, P0);
        // mu's registers serve as the scratch array A of the state step
        // (UNI: the fused stages -- shared lane shifts of u and of v05 -- also in the twelve-wave kernel: round 3 measured them slower there,
        //  104 -> 144 ... 172 B of scratch; with the registers the UNI variant frees they pay: backward sweep 772 -> 757 ms, round 4)
        sv_state<NT, BW, JAC, (MINW >= 3 ? (UNI ? 3 : JQ_BWD_FUSE3) : JQ_BWD_FUSE), UNI, !(QUAD && MINW >= 3 && JQ_BWD3_NOOPQ_STATE), jq_bwd_terms(NT, BW, JAC)>(p, a, active, ceps, ws, g, u, v, un, vN, mu, Ya, Yb);
        // (every wave has passed a workgroup barrier since it finished step n-1: begin_step in window mode, the operator
        // switches of sv_state otherwise)
        if (n > 0) flush_traces(n - 1);
        // use 6: Kp05 -- finish the state step; first adjoint product L = c K05 nb (= -c K05 lambda_i)
        const double* M = p.template next_ks<0, 1>();
        if (active) {
            if constexpr (UNI) {
                constexpr int FULLQ = JQ_T4_DIAG | JQ_T4_RTERMS | JQ_T4_MTERMS;
                mm_t4q<NT, false, FULLQ, true>(vN, vN, M, un, ceps, ws + g);
                mm_t4q<NT, true, FULLQ, true>(L, L, M, nb, ceps, ws + g);
            } else {
                mm_c<NT, BW>(vN, vN, M, un);
                if (a.use_shift) a_axpy_rows(vN, ceps, ws, g, un);
                mm_z<NT, BW>(L, M, nb);
                if (a.use_shift) a_axpy_rows(L, ceps, ws, g, nb);
            }
            a_unpark(mu, P0);
            a_park(vN, P0);     // vi(t_n) sleeps until the end of the step
        }
        // ---- adjoint step ----------------------------------------------------------------------
        // use 7: S0 -- L = c (S0 mu - K05 li + hr0) ; X = mu + sum_j S^j L   (in place: mu becomes X)
        M = p.template next_ks<1, 0>();
        if (active) {
            mm_c<NT, BW>(L, L, M, mu);
            a_axpy_rows1<NT, false>(L, wd, g, u);  // u holds vr before the state step (:862)
            if constexpr (WLR)
                if (wforce)
                    for (int k = 0; k < wl.r; ++k) {
                        double pu, qu;
                        if (wl.has_sc) pu = wl.get(k, 0), qu = wl.get(k, 1);      // (carried: the previous step's dots with vr(t_n))
                        else wl.dots(k, u, pu, qu);
                        const double cl = cfw * wl.lam(k);
                        wl.axpy2(k, L, cl * pu, cl * qu);      // + c hr0
                    }
            a_add(mu, L);
            if constexpr (jq_bwd_terms(NT, BW, JAC)) neumann_terms_add<NT, BW>(mu, L, M, a.m, Ya);      // (L is scratch from here on)
            else horner_add<NT, BW, JAC, BREG>(mu, mu, L, M, a.m, Ya, Yb, a.jacobi_tol2, a.N, a.jac_wg_lds);
        }
        // early traces with X (lets vr0 = u die here): tr1 = tr(vr0' Hanti_q X), tr3 = tr(vr' Hanti_q X)
        double o_p4 = 0.0;      // ORD: the new part of tr4 of control 1, formed in the pass of use 11
        for (int q = 0; q < Nc; ++q) {
            M = p.next_c(Nc + q);  // Hanti_q
            if (active) {
                if constexpr (ORD) {
                    if (q == 0) mm_t4q<NT, true, JQ_T4_DIAG>(Ya, Ya, M, mu);
                    else if (q == 1) mm_t4q<NT, true, JQ_T4_RTERMS>(Ya, Ya, M, mu);
                    else mm_t4q<NT, true, JQ_T4_MTERMS>(Ya, Ya, M, mu);
                } else {
                    mm_z_bw<NT, BW>(Ya, M, mu, a.bw_trace[q]);
                }
                // (UNI: the weight of the wave's one sample multiplies the SUMS)
                const double ts = UNI ? wave_sum4(a_dot(u, Ya), a_dot(un, Ya), 0.0, 0.0) * wgt
                                      : wave_sum4(a_dot(u, Ya) * wgt, a_dot(un, Ya) * wgt, 0.0, 0.0);   // rows 0, 2: t1, t3
                if ((lane_ & 15) == 0) rec[((size_t)(n & 1) * NWAVES + wave) * rslots + 4 * q + (lane_ >> 4)] = ts;
            }
        }
        if constexpr (QUAD && !JAC && (JQ_BWD_ADJ_FUSE & 1)) {
            // uses 8 and 9 in one pass (they share X): L = -c K0 X ; vN(scratch Q) = -c K1 X
            M = p.template next_ks<0, 0>();
            const double* M9 = p.template next_ks<0, 2>();
            if (active) {
                if constexpr (UNI) {
                    mm_t4q2<NT, true, true, 3>(L, L, M, vN, vN, M9, mu, -ceps, -ceps, ws + g);
                } else {
                    mm_t4q2<NT, true, true>(L, L, M, vN, vN, M9, mu);
                    if (a.use_shift) {
                        a_axpy_rows(L, -ceps, ws, g, mu);
                        a_axpy_rows(vN, -ceps, ws, g, mu);
                    }
                }
            }
        } else {
        // use 8: Kn0 -- L = -c K0 X
        M = p.template next_ks<0, 0>();
        if (active) {
            mm_z<NT, BW>(L, M, mu);
            if (a.use_shift) a_axpy_rows(L, -ceps, ws, g, mu);
        }
        // use 9: Kn1 -- vN(scratch Q) = -c K1 X
        M = p.template next_ks<0, 2>();
        if (active) {
            mm_z<NT, BW>(vN, M, mu);
            if (a.use_shift) a_axpy_rows(vN, -ceps, ws, g, mu);
        }
        }
        // use 10: S05 -- L = -c l2 = -c (K0 X + S05 li + hi0) ; Q = -c (S05 (li + c l2) + K1 X + hi1) ;
        //               nb_new = nb + L + sum_j S^j Q          (li_new = li + c (l2 + l1))
        M = p.template next_ks<1, 1>();
        if (active) {
            mm_z<NT, BW>(Ya, M, nb);
            a_axpy_rows1<NT, true>(Ya, wd, g, v);  // v holds vi05;  Ya = c (-S05 li - hi0)
            if constexpr (WLR)
                if (wforce)
                    for (int k = 0; k < wl.r; ++k) {
                        double pn, qn, rr, ss;
                        wl.dots(k, un, pn, qn);
                        wl.dots(k, v, rr, ss);
                        if (wl.has_sc) wl.put(k, 2, pn), wl.put(k, 3, qn), wl.put(k, 4, rr), wl.put(k, 5, ss);      // (again at use 12)
                        const double cl = cfw * wl.lam(k);
                        wl.axpy2(k, Ya, -cl * rr, -cl * ss);     // - c hi0 (goes into L and Q)
                        wl.axpy2(k, vN, -cl * qn, cl * pn);      // Q: - c (hi1 - hi0) = + c Wi vr(t_n) / T
                    }
            a_add(L, Ya);
            a_add(vN, Ya);
            mm_c<NT, BW>(vN, vN, M, L);       // vN = Q
            a_add(L, nb);
            a_add(L, vN);                     // L = nb + L + Q
            if constexpr (jq_bwd_terms(NT, BW, JAC)) neumann_terms_add<NT, BW>(L, vN, M, a.m, Ya);      // (vN = Q is scratch from here on)
            else horner_add<NT, BW, JAC, BREG>(L, L, vN, M, a.m, Ya, Yb, a.jacobi_tol2, a.N, a.jac_wg_lds);  // L = nb_new
            a_add(nb, L);                     // nb = nb_old + nb_new = -(li0 + li)
        }
        // use 11: Kp05 -- vN(scratch G) = X + c K05 nb_new (= lambda_r^{1/2} - c K05 li_new)
        M = p.template next_ks<0, 1>();
        if (active) {
            if constexpr (ORD) {
                // ... and Hsym_1 lambda_i_new (the new part of tr4 of control 1) from the same shifted copies of L = -lambda_i_new
                const d4* cfs = t4q_c<NT>(p.next_c(1), lane_);
                mm_t4q_multi<NT, 1, false, true, true, 1>(vN, mu, M, vN, mu, M, vN, mu, M, L, ceps, 0.0, 0.0, ws + g,
                                                          [&](int mt, double su, double sd) {
                                                              const d4 cs = t4q_cload(cfs, mt);
                                                              o_p4 = fma(-un.t[mt][0], fma(cs[1], sd, cs[0] * su), o_p4);
                                                          });
            } else if constexpr (UNI) {
                mm_t4q<NT, false, JQ_T4_DIAG | JQ_T4_RTERMS | JQ_T4_MTERMS, true>(vN, mu, M, L, ceps, ws + g);
            } else {
                mm_c<NT, BW>(vN, mu, M, L);
                if (a.use_shift) a_axpy_rows(vN, ceps, ws, g, L);
            }
        }
        // use 12: S1 -- lambda_r_new = X + c (S1 X - K05 li_new + hr1)
        M = p.template next_ks<1, 2>();
        if (active) {
            mm_c<NT, BW>(vN, vN, M, mu);
            a_axpy_rows1<NT, false>(vN, wd, g, un);
            if constexpr (WLR)
                if (wforce)
                    for (int k = 0; k < wl.r; ++k) {
                        double pn, qn, rr, ss;
                        if (wl.has_sc) {
                            pn = wl.get(k, 2), qn = wl.get(k, 3), rr = wl.get(k, 4), ss = wl.get(k, 5);
                            wl.put(k, 0, pn), wl.put(k, 1, qn);      // (the next step's dots with vr(t_n+1))
                        } else {
                            wl.dots(k, un, pn, qn);
                            wl.dots(k, v, rr, ss);
                        }
                        const double cl = cfw * wl.lam(k);
                        wl.axpy2(k, vN, cl * (pn - ss), cl * (qn + rr));      // + c hr1
                    }
        }
        // ---- late traces (adjoint_grad_calc!, :2581-2618), per control q, weighted by the sample weight:
        //   tr5 = tr(vi05' Hanti (li0+li))   tr2 = tr(vi05' Hsym X)   tr4 = tr(vr' Hsym li) + tr(vr0' Hsym li0)
        // here: un = vr, v = vi05, mu = X, nb = -(li0+li), L = -li
        for (int q = 0; q < Nc; ++q) {
            double t2 = 0, t4 = 0, t5 = 0;
            const int bwq = a.bw_trace[q];
            M = p.next_c(Nc + q);  // Hanti_q
            if (active) {
                if constexpr (ORD) {
                    if (q == 0) mm_t4q<NT, true, JQ_T4_DIAG>(Ya, Ya, M, nb);
                    else if (q == 1) mm_t4q<NT, true, JQ_T4_RTERMS>(Ya, Ya, M, nb);
                    else mm_t4q<NT, true, JQ_T4_MTERMS>(Ya, Ya, M, nb);
                } else {
                    mm_z_bw<NT, BW>(Ya, M, nb, bwq);
                }
                t5 = -a_dot(v, Ya);
            }
            M = p.next_c(q);  // Hsym_q
            if (active) {
                double p4;
                if constexpr (ORD) {
                    if (q == 1) {
                        mm_t4q<NT, true, JQ_T4_RTERMS>(Ya, Ya, M, mu);
                        t2 = a_dot(v, Ya);
                        p4 = o_p4;
                    } else {
                        if (q == 0) mm_t4q<NT, true, JQ_T4_DIAG>(Ya, Ya, M, mu);
                        else mm_t4q<NT, true, JQ_T4_MTERMS>(Ya, Ya, M, mu);
                        t2 = a_dot(v, Ya);
                        if (q == 0) mm_t4q<NT, true, JQ_T4_DIAG>(Ya, Ya, M, L);
                        else mm_t4q<NT, true, JQ_T4_MTERMS>(Ya, Ya, M, L);
                        p4 = -a_dot(un, Ya);
                    }
                } else {
                    mm_z_bw<NT, BW>(Ya, M, mu, bwq);
                    t2 = a_dot(v, Ya);
                    mm_z_bw<NT, BW>(Ya, M, L, bwq);
                    p4 = -a_dot(un, Ya);
                }
                t4 = p4 + carry[q * NTHREADS + threadIdx.x];
                carry[q * NTHREADS + threadIdx.x] = p4;
                const double ts = UNI ? wave_sum4(t2, t4, t5, 0.0) * wgt : wave_sum4(t2 * wgt, t4 * wgt, t5 * wgt, 0.0);   // rows 0, 2, 1: t2, t4, t5
                if ((lane_ & 15) == 0) rec[((size_t)(n & 1) * NWAVES + wave) * rslots + 4 * (Nc + q) + (lane_ >> 4)] = ts;
            }
        }
        // ---- roles for the next step: u <- un, mu <- vN(new lambda_r), nb <- L, v <- parked vi(t_n)
        if (active) {
            u = un;
            mu = vN;
            nb = L;
            a_unpark(v, P0);
        }
    }
    p.drain();
    flush_traces(a.nsteps_chunk - 1);
    if (active) {
        a_store(u, st, lane);
        a_store(v, st + KT * 64, lane);
        a_store(mu, st + 2 * KT * 64, lane);
        a_store(nb, st + 3 * KT * 64, lane);
        for (int q = 0; q < Nc; ++q) {
            double cv = carry[q * NTHREADS + threadIdx.x];
            if constexpr (QUAD) cv = row_ror_add<8>(row_ror_add<4>(cv));   // only ever used summed over the rows of a column
            if (cslot) st[(JQ_STATE_ARRAYS * KT + q) * 64 + clane] = cv;
        }
    }
}
