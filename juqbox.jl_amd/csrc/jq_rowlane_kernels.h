// jq_rowlane_kernels.h -- latency-optimised propagators for SMALL Hilbert spaces and SMALL batches:
// one lane per (row, column), four state columns per wave.
//
// The lane kernels (jq_lane_kernels.h) give a lane a whole column: NP*NP dependent-ish FMAs per product, which
// is the right shape for throughput (64 columns per wave) but leaves the latency of one evaluation at
// NP*NP*4 cycles per product.  Ensembles of the sizes Juqbox runs (1 ... a few thousand samples) do not fill
// the chip, so what the user waits for is exactly that latency.  Here the product is spread over the lanes:
//
//   lane = 16*c + i  holds element i of column c (c = 0..3: the four 16-lane DPP rows of the wave),
//   an operator row i lives in the lane's registers (M.r[j] = M[i][j]),
//   y_i = c_i + sum_j M[i][j] * x_j   is   NP x  v_fmac_f64_dpp  y, x, M.r[j]  row_newbcast:j
//
// i.e. the DPP operand is now the STATE vector (lane j of the row broadcast to the row) and the matrix
// element is the ordinary per-lane operand: NP FMAs per product instead of NP*NP, a state vector is ONE
// register pair.  The DPP operand is always freshly written by VALU instructions, so the 2-wait-state
// hazard (jq_lane_kernels.h) is systematic here: every product starts with s_nop 1 behind a scheduling
// barrier, and scripts/check_dpp_hazard.py verifies the final ISA.
//
// Throughput per column is 16/NP x 1..1.5 lower than the lane kernels (idle lanes for NP < 16, two accumulator
// chains), so the host uses these kernels only while the batch is small (run_eval, juqbox_hip.hip).
//
// Layouts:  operator images: [16 rows][NPJ] row-major, zero padded (stride = 16*NPJ doubles = a.stride);
//           stream point j -> K at (2j)*stride, S at (2j+1)*stride;   constants a.cimg: [Hsym_q | Hanti_q];
//           state file: [array][wave][64 lanes];   a.colinfo: [eps per column | weight per column], 4 per wave.
#pragma once
#include "jq_lane_kernels.h"

template <int NPJ>
struct RowMat {
    double r[NPJ];
};

template <int NPJ>
__device__ __forceinline__ RowMat<NPJ> row_load(cmat_t img, int row)
{
    RowMat<NPJ> m;
    cmat_t p = img + row * NPJ;
#pragma unroll
    for (int j = 0; j < NPJ; ++j) m.r[j] = p[j];
    return m;
}

// the same from an LDS copy of the image (constant images of the backward sweep when NPJ > 8)
template <int NPJ>
__device__ __forceinline__ RowMat<NPJ> row_load_lds(const double* img, int row)
{
    RowMat<NPJ> m;
    const double* p = img + row;       // [column j][row] (see the copy loops)
#pragma unroll
    for (int j = 0; j < NPJ; ++j) m.r[j] = p[j * 16];
    return m;
}

// y = c + M x : two accumulator chains (even / odd j).
// x was (almost always) just written by VALU instructions: DPP read-after-write hazard, 2 wait states -- `s_nop 1` opens the product.
// The whole product is ONE asm block (round 6): with one asm statement per FMA hipcc put an `s_nop 0` of its own between every pair of
// them (its hazard recognizer cannot see into inline asm) -- four to seven issue slots per product that a wave which is alone on its SIMD
// pays in full (NPJ = 12: 12 FMAs + 5 s_nop; the time step of cnot2 spent 18 % of its issue slots on them).  Same FMAs, same order,
// same accumulators: results bit-identical.  scripts/check_dpp_hazard.py verifies the final ISA as before.
// (The accumulators are early-clobber operands: where c and x are the same value at its last use -- the first term of a two-term Horner
//  recurrence -- the compiler would otherwise give ya and x ONE register, and the second FMA would broadcast the first one's result;
//  the hazard check found it.)
// (BLK = false: one asm statement per FMA behind a scheduling barrier, as before round 6 -- the two-wave backward kernel at NPJ = 16 keeps it:
//  a block wants all 16 operand pairs of a row in VGPRs at once, and that kernel, which holds the operators of BOTH roles, then spills 175
//  registers instead of 54: backward sweep 9.9 -> 12.2 ms per 4 000 steps)
template <int J>
__device__ __forceinline__ void fma_xbcast(double& y, double x, double m)
{
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(y) : "v"(x), "v"(m), "n"(J));
}
template <int NPJ, int... J>
__device__ __forceinline__ void rmv_fold(double& ya, double& yb, double x, const RowMat<NPJ>& M, std::integer_sequence<int, J...>)
{
    (fma_xbcast<J>((J & 1) ? yb : ya, x, M.r[J]), ...);
}
template <int NPJ, bool ZEROC, bool BLK = true>
__device__ __forceinline__ double rmv(double c, const RowMat<NPJ>& M, double x)
{
    if constexpr (!BLK) {
        double ya = ZEROC ? 0.0 : c, yb = 0.0;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 1");
        __builtin_amdgcn_sched_barrier(0);
        rmv_fold<NPJ>(ya, yb, x, M, std::make_integer_sequence<int, NPJ>{});
        return ya + yb;
    }
    static_assert(NPJ == 2 || NPJ == 4 || NPJ == 6 || NPJ == 8 || NPJ == 12 || NPJ == 16, "row lengths the row-lane kernels are instantiated for");
    double ya = ZEROC ? 0.0 : c, yb = 0.0;
    if constexpr (NPJ == 2)
        asm("s_nop 1\n\t"
            "v_fmac_f64_dpp %0, %2, %3 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %2, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf"
            : "+&v"(ya), "+&v"(yb)
            : "v"(x), "v"(M.r[0]), "v"(M.r[1]));
    if constexpr (NPJ == 4)
        asm("s_nop 1\n\t"
            "v_fmac_f64_dpp %0, %2, %3 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %2, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %0, %2, %5 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %2, %6 row_newbcast:3 row_mask:0xf bank_mask:0xf"
            : "+&v"(ya), "+&v"(yb)
            : "v"(x), "v"(M.r[0]), "v"(M.r[1]), "v"(M.r[2]), "v"(M.r[3]));
    if constexpr (NPJ == 6)
        asm("s_nop 1\n\t"
            "v_fmac_f64_dpp %0, %2, %3 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %2, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %0, %2, %5 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %2, %6 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %0, %2, %7 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %2, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf"
            : "+&v"(ya), "+&v"(yb)
            : "v"(x), "v"(M.r[0]), "v"(M.r[1]), "v"(M.r[2]), "v"(M.r[3]), "v"(M.r[4]), "v"(M.r[5]));
    if constexpr (NPJ == 8)
        asm("s_nop 1\n\t"
            "v_fmac_f64_dpp %0, %2, %3 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %2, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %0, %2, %5 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %2, %6 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %0, %2, %7 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %2, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %0, %2, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %2, %10 row_newbcast:7 row_mask:0xf bank_mask:0xf"
            : "+&v"(ya), "+&v"(yb)
            : "v"(x), "v"(M.r[0]), "v"(M.r[1]), "v"(M.r[2]), "v"(M.r[3]), "v"(M.r[4]), "v"(M.r[5]), "v"(M.r[6]), "v"(M.r[7]));
    if constexpr (NPJ == 12)
        asm("s_nop 1\n\t"
            "v_fmac_f64_dpp %0, %2, %3 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %2, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %0, %2, %5 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %2, %6 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %0, %2, %7 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %2, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %0, %2, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %2, %10 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %0, %2, %11 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %2, %12 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %0, %2, %13 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %2, %14 row_newbcast:11 row_mask:0xf bank_mask:0xf"
            : "+&v"(ya), "+&v"(yb)
            : "v"(x), "v"(M.r[0]), "v"(M.r[1]), "v"(M.r[2]), "v"(M.r[3]), "v"(M.r[4]), "v"(M.r[5]), "v"(M.r[6]), "v"(M.r[7]), "v"(M.r[8]), "v"(M.r[9]), "v"(M.r[10]), "v"(M.r[11]));
    if constexpr (NPJ == 16)
        asm("s_nop 1\n\t"
            "v_fmac_f64_dpp %0, %2, %3 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %2, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %0, %2, %5 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %2, %6 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %0, %2, %7 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %2, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %0, %2, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %2, %10 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %0, %2, %11 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %2, %12 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %0, %2, %13 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %2, %14 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %0, %2, %15 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %2, %16 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %0, %2, %17 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %2, %18 row_newbcast:15 row_mask:0xf bank_mask:0xf"
            : "+&v"(ya), "+&v"(yb)
            : "v"(x), "v"(M.r[0]), "v"(M.r[1]), "v"(M.r[2]), "v"(M.r[3]), "v"(M.r[4]), "v"(M.r[5]), "v"(M.r[6]), "v"(M.r[7]), "v"(M.r[8]), "v"(M.r[9]), "v"(M.r[10]), "v"(M.r[11]), "v"(M.r[12]), "v"(M.r[13]), "v"(M.r[14]), "v"(M.r[15]));
    return ya + yb;
}
// bpa + sum_{j=1..m} S^j A  (Horner form)
template <int NPJ, bool BLK = true>
__device__ __forceinline__ double row_horner(double bpa, double A, const RowMat<NPJ>& S, int m)
{
    if (m <= 0) return bpa;
    double Y = A;
    if constexpr (NPJ <= 4) {
        // a product of NPJ <= 4 is shorter than what a taken branch costs a lone wave (probes/lone_wave_probe.hip): the recurrence in
        // blocks of four, two and one products (SWAP-02 12.6 -> 12.2 ms; at NPJ = 6 the same lost 3 %: flux 19.5 -> 20.0 ms)
        int j = m - 1;
        for (; j >= 4; j -= 4) {
            Y = rmv<NPJ, false, BLK>(A, S, Y);
            Y = rmv<NPJ, false, BLK>(A, S, Y);
            Y = rmv<NPJ, false, BLK>(A, S, Y);
            Y = rmv<NPJ, false, BLK>(A, S, Y);
        }
        if (j & 2) {
            Y = rmv<NPJ, false, BLK>(A, S, Y);
            Y = rmv<NPJ, false, BLK>(A, S, Y);
        }
        if (j & 1) Y = rmv<NPJ, false, BLK>(A, S, Y);
    } else {
        for (int j = 1; j < m; ++j) Y = rmv<NPJ, false, BLK>(A, S, Y);
    }
    return rmv<NPJ, false, BLK>(bpa, S, Y);
}

// ... with the number of terms known at compile time (MT = m; MT = 0: the run-time loop above): the same products in the same order,
// no loop branches -- a taken branch costs a lone wave 20 .. 120 clk (probes/lone_wave_probe.hip), and there were 2 (m - 1) + 2 per step
template <int NPJ, bool BLK, int MT>
__device__ __forceinline__ double row_horner_m(double bpa, double A, const RowMat<NPJ>& S, int m)
{
    if constexpr (MT == 0) {
        return row_horner<NPJ, BLK>(bpa, A, S, m);
    } else {
        double Y = A;
#pragma unroll
        for (int j = 1; j < MT; ++j) Y = rmv<NPJ, false, BLK>(A, S, Y);
        return rmv<NPJ, false, BLK>(bpa, S, Y);
    }
}
template <int MT>
struct RlTerms {
    static constexpr int value = MT;
};
// RUN(RlTerms<m>{}) for the m the kernels are specialised for (2 .. 10: rabi has 10), RUN(RlTerms<0>{}) otherwise
#define JQ_RL_DISPATCH_TERMS(m, RUN)             \
    switch (m) {                                 \
    case 2: RUN(RlTerms<2>{}); break;            \
    case 3: RUN(RlTerms<3>{}); break;            \
    case 4: RUN(RlTerms<4>{}); break;            \
    case 5: RUN(RlTerms<5>{}); break;            \
    case 6: RUN(RlTerms<6>{}); break;            \
    case 7: RUN(RlTerms<7>{}); break;            \
    case 8: RUN(RlTerms<8>{}); break;            \
    case 9: RUN(RlTerms<9>{}); break;            \
    case 10: RUN(RlTerms<10>{}); break;          \
    default: RUN(RlTerms<0>{}); break;           \
    }

template <int NPJ>
struct RowOps {
    RowMat<NPJ> Kn0, S0, Kp05, S05, Kn1, S1;
};
template <int NPJ>
__device__ __forceinline__ void rops_load_half(RowOps<NPJ>& o, const PropArgs& a, int n, int row)
{
    cmat_t s = as_const(a.stream) + (size_t)(2 * (2 * n + 1)) * a.stride;
    o.Kp05 = row_load<NPJ>(s, row);
    o.S05 = row_load<NPJ>(s + a.stride, row);
    o.Kn1 = row_load<NPJ>(s + 2 * a.stride, row);
    o.S1 = row_load<NPJ>(s + 3 * a.stride, row);
}
template <int NPJ>
__device__ __forceinline__ void rops_advance(RowOps<NPJ>& o, const RowOps<NPJ>& nxt)
{
    o.Kn0 = o.Kn1;
    o.S0 = o.S1;
    o.Kp05 = nxt.Kp05;
    o.S05 = nxt.S05;
    o.Kn1 = nxt.Kn1;
    o.S1 = nxt.S1;
}

// Round 6: the operator rows of a time step WITHOUT register copies between steps.  A step uses the integer point t_n (Kn0, S0), the
// half point (Kp05, S05) and the integer point t_n+1 (Kn1, S1); the rows of the NEXT half point and of t_n+2 are loaded a whole step
// ahead (a lone wave cannot hide a global load any other way).  With `o` / `nxt` structs that meant 6 NPJ register-pair moves per step
// (rops_advance: 72 v_mov_b64 at NPJ = 12, a sixth of the step's issue slots for a wave that is alone on its SIMD).  Instead: three
// integer-point sets and three half-point sets, the time loop unrolled three times, every step names the sets it reads and the sets its
// loads go to -- after three steps every value is back in the variable it started in (JQ_RL_ROTATE).  Same loads, same products, same
// order: bit-identical results.
template <int NPJ>
struct RowTP {
    RowMat<NPJ> K, S;
};
template <int NPJ>
struct RowOpsV {      // the view a step's code reads its operators through (the member names of RowOps)
    const RowMat<NPJ>&Kn0, &S0, &Kp05, &S05, &Kn1, &S1;
};
// half point of step n -> H, integer point t_n+1 -> I
template <int NPJ>
__device__ __forceinline__ void rops_load_tp(RowTP<NPJ>& H, RowTP<NPJ>& I, const PropArgs& a, int n, int row)
{
    cmat_t s = as_const(a.stream) + (size_t)(2 * (2 * n + 1)) * a.stride;
    H.K = row_load<NPJ>(s, row);
    H.S = row_load<NPJ>(s + a.stride, row);
    I.K = row_load<NPJ>(s + 2 * a.stride, row);
    I.S = row_load<NPJ>(s + 3 * a.stride, row);
    // the loads are ISSUED here, a whole step before their first use: without the barrier the scheduler, which counts registers and not the
    // ~ 800 clk a lone wave waits for L2, sinks them to their uses whenever a step becomes one basic block (measured: forward sweep x 2)
    __builtin_amdgcn_sched_barrier(0);
}
template <int NPJ>
__device__ __forceinline__ void rops_first(RowTP<NPJ>& I0, RowTP<NPJ>& H0, RowTP<NPJ>& I1, const PropArgs& a, int row)
{
    I0.K = row_load<NPJ>(as_const(a.stream), row);
    I0.S = row_load<NPJ>(as_const(a.stream) + a.stride, row);
    rops_load_tp(H0, I1, a, 0, row);
}
// STEP(n, integer point t_n, half point, integer point t_n+1, [load targets:] next half point, integer point t_n+2)
#define JQ_RL_ROTATE(NSTEPS, STEP)                     \
    for (int n_ = 0; n_ < (NSTEPS);) {                 \
        STEP(n_, I0, H0, I1, H1, I2);                  \
        if (++n_ >= (NSTEPS)) break;                   \
        STEP(n_, I1, H1, I2, H2, I0);                  \
        if (++n_ >= (NSTEPS)) break;                   \
        STEP(n_, I2, H2, I0, H0, I1);                  \
        ++n_;                                          \
    }
// ... the same interface WITH the copies (two integer-point sets + one load target, as before round 6): where the third set costs more than
// the moves -- the two-wave backward kernel at NPJ = 16 holds both roles' operators and spilled 245 registers with the rotation (measured:
// backward sweep 9.9 -> 12.2 ms per 4 000 steps; the copy version: 54 spilled registers)
#define JQ_RL_ADVANCE(NSTEPS, STEP)                    \
    for (int n_ = 0; n_ < (NSTEPS); ++n_) {            \
        STEP(n_, I0, H0, I1, H1, I2);                  \
        I0 = I1;                                       \
        H0 = H1;                                       \
        I1 = I2;                                       \
    }

// One Stormer-Verlet state step (forward step!, src/StormerVerlet.jl:461-504), accumulate form; sw = eps*c*ws_i
// (MT: Neumann terms at compile time, row_horner_m.  SHF: the caller passes sw = 0 for "no shift" and the shift FMAs are unconditional --
//  a + 0 x = a: no selects in the step)
template <int NPJ, typename OPS, bool BLK = true, int MT = 0, bool SHF = false>
__device__ __forceinline__ void row_state(const PropArgs& a, const OPS& o, double sw, double u, double v, double& un,
                                          double& v05, double& vnew)
{
    double A = rmv<NPJ, true, BLK>(0.0, o.Kp05, u);
    if (SHF || a.use_shift) A = fma(sw, u, A);
    A = rmv<NPJ, false, BLK>(A, o.S05, v);
    v05 = row_horner_m<NPJ, BLK, MT>(v + A, A, o.S05, a.m);
    const double vN = rmv<NPJ, false, BLK>(v05, o.S05, v05);
    un = rmv<NPJ, false, BLK>(u, o.Kn0, v05);
    if (SHF || a.use_shift) un = fma(-sw, v05, un);
    un = rmv<NPJ, false, BLK>(un, o.S0, u);
    A = rmv<NPJ, true, BLK>(0.0, o.Kn1, v05);
    if (SHF || a.use_shift) A = fma(-sw, v05, A);
    A = rmv<NPJ, false, BLK>(A, o.S1, un);
    un = row_horner_m<NPJ, BLK, MT>(un + A, A, o.S1, a.m);
    vnew = rmv<NPJ, false, BLK>(vN, o.Kp05, un);
    if (SHF || a.use_shift) vnew = fma(sw, un, vnew);
}

// Round 6: the operator rows of the time steps through a RING IN LDS that the wave feeds itself by global->LDS DMA, JQ_RL_AHEAD steps
// ahead.  Measured (scripts/time_small.py with the stream pointer frozen): the K / S images of a sweep (27 MB at cnot1) come from the
// Infinity Cache / HBM at ~ 600 ns per access -- more than a time step of a lone wave (~ 400 ns of instructions) -- and the register
// prefetch of the rotation above reaches one step ahead: every step waited for its rows.  Deeper register prefetch costs 4 NPJ
// registers per step of distance (no room at NPJ = 12, 16); the DMA costs NPJ / 2 instructions per step, no registers, and the rows
// a product needs are read from LDS (~ 100 clk) right in the step that uses them.
//   group g = the four images [K, S](t_g+1/2), [K, S](t_g+1) of the tile stream = NPJ / 2 pieces of 1 KiB; slot = g mod R, R = AHEAD + 2:
//   step n reads groups n - 1 (images 2, 3: the integer point t_n) and n while groups n + 1 .. n + AHEAD are in flight or landed.
//   Group -1 is the first time point of the stream (images 0, 1), placed as images 2, 3 of slot R - 1.
// The DMA instructions are asm statements (the compiler would otherwise wait for ALL of them in front of every LDS read); the wave waits
// with s_waitcnt vmcnt((AHEAD - 1) P) at the top of a step -- its loads return in order -- and with vmcnt(0) before it ends (a DMA
// that lands after the workgroup has gone would write into another workgroup's LDS).
// (NPJ = 12, 16: a step takes longer than the latency and a slot is 6 .. 8 KiB -- two steps ahead)
#define JQ_RL_AHEAD(npj) ((npj) <= 8 ? 3 : 2)
#define JQ_RL_RING_BYTES(npj) ((size_t)(JQ_RL_AHEAD(npj) + 2) * 4 * 128 * (npj))
template <int NPJ, bool ORDERED = true>
struct RlRing {
    static constexpr int D = JQ_RL_AHEAD(NPJ), R = D + 2, P = NPJ / 2, IMG = 16 * NPJ, GROUP_B = 4 * 128 * NPJ;
    static constexpr size_t BYTES = (size_t)R * GROUP_B;
    const double* lds;      // slot 0
    unsigned lds0;          // ... as an LDS address (M0 of the DMA)
    const char* src;        // the stream
    unsigned lo16;          // 16 x lane
    int N;
    int row;
    int sp, sc;             // slots of groups n - 1 and n
    int si;                 // slot the next issue goes to
    int gi;                 // group of the next issue (not clamped)

    __device__ __forceinline__ void dma(unsigned dst, const char* s) const
    {
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(lo16), "s"(s) : "memory");
    }
    __device__ __forceinline__ void issue()
    {
        const int g = min(gi, N - 1);      // (past the end: the last group again -- into a slot nobody reads any more)
        const char* s = src + (size_t)(2 * (2 * g + 1)) * (IMG * 8);
        const unsigned dst = lds0 + (unsigned)si * GROUP_B;
#pragma unroll
        for (int p = 0; p < P; ++p) dma(dst + p * 1024u, s + p * 1024);
        ++gi;
        si = (si == R - 1) ? 0 : si + 1;
    }
    __device__ __forceinline__ void init(const PropArgs& a, double* ring, int lane, int row_)
    {
        lds = ring;
        lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)ring;
        src = (const char*)a.stream;
        lo16 = 16u * lane;
        N = a.nsteps_chunk;
        row = row_;
        // group -1: the two images of t_0 -> images 2, 3 of slot R - 1 (256 NPJ bytes = 16 NPJ lanes of 16 bytes)
        for (int b = 0; b < 256 * NPJ; b += 1024)
            if (b + (int)lo16 < 256 * NPJ) dma(lds0 + (unsigned)(R - 1) * GROUP_B + 2u * 128 * NPJ + b, src + b);
        sp = R - 1;
        sc = 0;
        si = 0;
        gi = 0;
#pragma unroll
        for (int k = 0; k < D; ++k) issue();
    }
    // top of step n: group n has landed; the DMA of group n + AHEAD goes out
    __device__ __forceinline__ void begin_step()
    {
        if constexpr (ORDERED)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * P) : "memory");
        else      // (other memory traffic of the wave -- the state history -- completes out of order with the DMA)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        issue();
    }
    __device__ __forceinline__ void end_step()
    {
        sp = sc;
        sc = (sc == R - 1) ? 0 : sc + 1;
    }
    __device__ __forceinline__ void drain() const { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    __device__ __forceinline__ RowMat<NPJ> rowmat(int slot, int img) const
    {
        typedef double d2_t __attribute__((ext_vector_type(2)));
        const d2_t* p = (const d2_t*)(lds + (size_t)slot * (4 * IMG) + img * IMG + row * NPJ);
        RowMat<NPJ> m;
#pragma unroll
        for (int k = 0; k < NPJ / 2; ++k) {
            const d2_t q = p[k];
            m.r[2 * k] = q[0];
            m.r[2 * k + 1] = q[1];
        }
        return m;
    }
};
// the operators of the current step out of the ring
template <int NPJ>
struct RowOpsL {
    RowMat<NPJ> Kn0, S0, Kp05, S05, Kn1, S1;
    template <bool O>
    __device__ __forceinline__ RowOpsL(const RlRing<NPJ, O>& r)
        : Kn0(r.rowmat(r.sp, 2)), S0(r.rowmat(r.sp, 3)), Kp05(r.rowmat(r.sc, 0)), S05(r.rowmat(r.sc, 1)), Kn1(r.rowmat(r.sc, 2)), S1(r.rowmat(r.sc, 3))
    {
    }
};

// ---------------------------------------------------------------------------------------------
// Low-rank full leakage weights (PropArgs::wlr; jq_kernels.h WLow) in the row-lane layout: the 16 lanes of a DPP row are the rows
// of one column, so a column dot product is one multiply and four DPP rotate-adds (valid in every lane of the row).  The table is
// copied to LDS once (a lone wave would wait ~1 us per global read in every time step): [lam[JQ_MAX_WRANK] | a_k[16], b_k[16] per k].
#define JQ_RL_WTAB (JQ_MAX_WRANK + 2 * JQ_MAX_WRANK * 16)
__device__ __forceinline__ double rl_colsum(double x)
{
    x = row_ror_add<8>(x);
    x = row_ror_add<4>(x);
    x = row_ror_add<2>(x);
    return row_ror_add<1>(x);
}
struct RowW {
    const double* tab;      // LDS
    int r, row;
    __device__ __forceinline__ void init(const PropArgs& a, double* lds, int lane, int nthreads)
    {
        r = a.wrank;
        row = lane & 15;
        tab = lds;
        if (r > 0) {      // (a.wstride == 16: Ntot <= 16 is one tile row)
            for (int i = threadIdx.x; i < JQ_MAX_WRANK + 2 * r * 16; i += nthreads) lds[i] = a.wlr[i];
            __syncthreads();
        }
    }
    __device__ __forceinline__ double lam(int k) const { return tab[k]; }
    __device__ __forceinline__ double a(int k) const { return tab[JQ_MAX_WRANK + 32 * k + row]; }
    __device__ __forceinline__ double b(int k) const { return tab[JQ_MAX_WRANK + 32 * k + 16 + row]; }
};

#define JQ_ROWLANE_ARRAYS 4                                   // U, V, MU, NB
#define JQ_ROWLANE_ROWS (JQ_ROWLANE_ARRAYS + JQ_MAXNC + 1)    // + carry rows + leak row

// Forward sweep of one chunk; a.nslabs = number of waves (4 columns each), grid = a.nslabs, block = 64.
// (WF: low-rank full leakage weights compiled in -- separate instantiations, the Diagonal fast path is untouched)
template <int NPJ, bool WF = false, bool HIST = false>
__global__ __launch_bounds__(64) void k_forward_rowlane(PropArgs a)
{
    const int lane = threadIdx.x;
    const int row = lane & 15;
    const long long w = blockIdx.x, nw = a.nslabs;
    const long long col = 4 * w + (lane >> 4);
    const double wd = a.tabs[row];
    double* st = a.state + w * 64 + lane;
    double u = st[0], v = st[nw * 64];
    double leak = st[(size_t)(JQ_ROWLANE_ARRAYS + JQ_MAXNC) * nw * 64];
    const double sw = a.use_shift ? 0.5 * a.h * a.colinfo[col] * a.tabs[16 + row] : 0.0;
    extern __shared__ double lds_w[];
    RowW wl;
    if constexpr (WF) wl.init(a, lds_w, lane, 64);
    RlRing<NPJ, !HIST> ring;
    ring.init(a, lds_w + (WF ? JQ_RL_WTAB : 0), lane, row);
    auto sweep = [&](auto terms) {
    constexpr int MT = decltype(terms)::value;
    for (int n = 0; n < a.nsteps_chunk; ++n) {
        ring.begin_step();
        const RowOpsL<NPJ> o(ring);
        double un, v05, vnew;
        leak = fma(wd, u * u, leak);
        row_state<NPJ, RowOpsL<NPJ>, true, MT, true>(a, o, sw, u, v, un, v05, vnew);
        if constexpr (WF) {   // full weights: tr(vr' Wr vr) at t_n and t_n+1, 2 tr(vi05' Wr vi05), -2 tr(vi05' Wi vr(t_n)) (:700, :716-718)
            double lk = 0.0;
            for (int k = 0; k < wl.r; ++k) {
                const double ak = wl.a(k), bk = wl.b(k);
                const double p0 = rl_colsum(ak * u), q0 = rl_colsum(bk * u), p1 = rl_colsum(ak * un), q1 = rl_colsum(bk * un);
                const double rr = rl_colsum(ak * v05), ss = rl_colsum(bk * v05);
                lk += wl.lam(k) * ((p0 * p0 + q0 * q0) + (p1 * p1 + q1 * q1) + 2.0 * (rr * rr + ss * ss) - 2.0 * (ss * p0 - rr * q0));
            }
            if (row == 0) leak += lk;
        }
        u = un;
        v = vnew;
        leak = fma(wd, u * u + 2.0 * v05 * v05, leak);
        if constexpr (HIST)
            if (a.hist_r && col < a.N && row < a.Ntot) {
                const size_t off = (size_t)(a.step0 + n + 1) * a.Ntot * a.N + (size_t)col * a.Ntot + row;
                a.hist_r[off] = u;
                a.hist_i[off] = -v;
            }
        ring.end_step();
    }
    };
    JQ_RL_DISPATCH_TERMS(a.m, sweep)
    ring.drain();
    st[0] = u;
    st[nw * 64] = v;
    st[(size_t)(JQ_ROWLANE_ARRAYS + JQ_MAXNC) * nw * 64] = leak;
}

// Backward sweep of one chunk (state re-integration, adjoint step, trace scalars per wave and step).
template <int NPJ, bool WF = false>
__global__ __launch_bounds__(64) void k_backward_rowlane(PropArgs a)
{
    const int lane = threadIdx.x;
    const int row = lane & 15;
    const long long w = blockIdx.x, nw = a.nslabs;
    const long long col = 4 * w + (lane >> 4);
    const int Nc = a.Ncoupled;
    const double wd = a.tabs[row];
    double* st = a.state + w * 64 + lane;
    double u = st[0], v = st[nw * 64], mu = st[2 * nw * 64], nb = st[3 * nw * 64];
    const double sw = 0.5 * a.h * a.colinfo[col] * a.tabs[16 + row];
    const double wgt = a.colinfo[4 * nw + col];
    const double cfw = (a.forced ? 0.5 * a.h * a.tinv : 0.0) * wd;
    double carry[JQ_MAXNC];
    // constant images: register resident for NPJ <= 8; for NPJ = 12, 16 (24..32 registers per image row) they
    // live in LDS (2*Nc*stride doubles, copied once) and a row is read right before its two products
    constexpr bool RESIDENT = (NPJ <= 8);
    extern __shared__ double lds_c[];
    if (!RESIDENT) {
        // transposed copy [image][column j][row]: the 16 lanes of an LDS pass read 16 consecutive doubles (the [row][j] order of the
        // global image gave these reads 4-way bank conflicts at NPJ = 12: 75 % of the LDS cycles)
        for (int i = lane; i < 2 * Nc * (int)a.stride; i += 64) {
            const int im = i / (int)a.stride, e = i - im * (int)a.stride;
            lds_c[im * (int)a.stride + (e % NPJ) * 16 + e / NPJ] = a.cimg[i];
        }
        __syncthreads();
    }
    RowMat<NPJ> Hs[JQ_MAXNC], Ha[JQ_MAXNC];
#pragma unroll
    for (int q = 0; q < JQ_MAXNC; ++q) {
        const int qq = min(q, Nc - 1);
        carry[q] = (q < Nc) ? st[(size_t)(JQ_ROWLANE_ARRAYS + q) * nw * 64] : 0.0;
        if (RESIDENT || a.first_chunk) Hs[q] = row_load<NPJ>(as_const(a.cimg) + (size_t)qq * a.stride, row);
        if (RESIDENT) Ha[q] = row_load<NPJ>(as_const(a.cimg) + (size_t)(Nc + qq) * a.stride, row);
    }
    double* trw = a.traces + ((size_t)w * a.nsteps_chunk) * (Nc * JQ_NTR);

    if (a.first_chunk) {
        // per-lane part of carry_q = tr(vr' Hsym_q lambdai) at t = T (see k_backward); summed with the traces
#pragma unroll
        for (int q = 0; q < JQ_MAXNC; ++q)
            if (q < Nc) carry[q] = -u * rmv<NPJ, true>(0.0, Hs[q], nb);
    }

    // full leakage weights in low-rank form (behind the constant images in LDS): forcing of src/evalobjgrad.jl:862, :882-888
    RowW wl;
    if constexpr (WF) wl.init(a, lds_c + (RESIDENT ? 0 : (size_t)2 * Nc * a.stride), lane, 64);
    const double cf0 = a.forced ? 0.5 * a.h * a.tinv : 0.0;
    RowTP<NPJ> I0, I1, I2, H0, H1, H2;
    rops_first(I0, H0, I1, a, row);
    auto step = [&](int n, const RowTP<NPJ>& Pa, const RowTP<NPJ>& Ph, const RowTP<NPJ>& Pb, RowTP<NPJ>& Lh, RowTP<NPJ>& Li) {
        rops_load_tp(Lh, Li, a, min(n + 1, a.nsteps_chunk - 1), row);   // lands during this step
        const RowOpsV<NPJ> o{Pa.K, Pa.S, Ph.K, Ph.S, Pb.K, Pb.S};
        double un, v05, vnew;
        row_state<NPJ>(a, o, sw, u, v, un, v05, vnew);
        // adjoint step! (src/StormerVerlet.jl:255-303) with nb = -lambda_i, see k_backward
        double R = rmv<NPJ, true>(0.0, o.Kp05, nb);
        if (a.use_shift) R = fma(sw, nb, R);
        R = rmv<NPJ, false>(R, o.S0, mu);
        R = fma(cfw, u, R);
        double wP = 0.0, wQ = 0.0, wG = 0.0;      // low-rank parts of c hi0, c (hi1 - hi0), c hr1
        if constexpr (WF)
            for (int k = 0; k < wl.r; ++k) {
                const double ak = wl.a(k), bk = wl.b(k), cl = cf0 * wl.lam(k);
                const double pu = rl_colsum(ak * u), qu = rl_colsum(bk * u), pn = rl_colsum(ak * un), qn = rl_colsum(bk * un);
                const double rr = rl_colsum(ak * v05), ss = rl_colsum(bk * v05);
                R = fma(cl * pu, ak, fma(cl * qu, bk, R));                    // + c hr0
                wP = fma(cl * rr, ak, fma(cl * ss, bk, wP));                  // c Wr vi05 / T
                wQ = fma(cl * pn, bk, fma(-cl * qn, ak, wQ));                 // c Wi vr(t_n) / T
                wG = fma(cl * (pn - ss), ak, fma(cl * (qn + rr), bk, wG));    // c (Wr vr(t_n) + Wi vi05) / T
            }
        const double X = row_horner<NPJ>(mu + R, R, o.S0, a.m);
        double L = rmv<NPJ, true>(0.0, o.Kn0, X);
        if (a.use_shift) L = fma(-sw, X, L);
        double Qv = rmv<NPJ, true>(0.0, o.Kn1, X);
        if (a.use_shift) Qv = fma(-sw, X, Qv);
        {
            double P = rmv<NPJ, true>(0.0, o.S05, nb);
            P = fma(-cfw, v05, P);
            if constexpr (WF) P -= wP;
            L += P;
            Qv += P;
            if constexpr (WF) Qv += wQ;
        }
        Qv = rmv<NPJ, false>(Qv, o.S05, L);
        const double nbn = row_horner<NPJ>((nb + L) + Qv, Qv, o.S05, a.m);
        const double Bq = nb + nbn;
        double G = rmv<NPJ, false>(X, o.Kp05, nbn);
        if (a.use_shift) G = fma(sw, nbn, G);
        G = rmv<NPJ, false>(G, o.S1, X);
        G = fma(cfw, un, G);
        if constexpr (WF) G += wG;
        // traces (adjoint_grad_calc!, src/evalobjgrad.jl:2581-2618), weighted and summed over the wave
        double t5p[JQ_MAXNC] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < JQ_MAXNC; ++q) {
            if (q < Nc) {
                if (!RESIDENT) {
                    Ha[q] = row_load_lds<NPJ>(lds_c + (size_t)(Nc + q) * a.stride, row);
                    Hs[q] = row_load_lds<NPJ>(lds_c + (size_t)q * a.stride, row);
                }
                const double HaX = rmv<NPJ, true>(0.0, Ha[q], X);
                t5p[q] = -v05 * rmv<NPJ, true>(0.0, Ha[q], Bq) * wgt;
                const double t2 = v05 * rmv<NPJ, true>(0.0, Hs[q], X) * wgt;
                const double p4 = -un * rmv<NPJ, true>(0.0, Hs[q], nbn);
                // t1 .. t4 of the control in ONE reduction: lane 16 r holds the sum of the r-th value
                const double ts = wave_sum4_rows(u * HaX * wgt, t2, un * HaX * wgt, (p4 + carry[q]) * wgt);
                carry[q] = p4;
                if (row == 0) trw[(size_t)n * (Nc * JQ_NTR) + q * JQ_NTR + (lane >> 4)] = ts;
            }
        }
        {   // ... and the t5 of all (at most four) controls in one more
            const double ts = wave_sum4_rows(t5p[0], t5p[1], t5p[2], t5p[3]);
            if (row == 0 && (lane >> 4) < Nc) trw[(size_t)n * (Nc * JQ_NTR) + (lane >> 4) * JQ_NTR + 4] = ts;
        }
        u = un;
        v = vnew;
        mu = G;
        nb = nbn;
    };
    JQ_RL_ROTATE(a.nsteps_chunk, step)
    st[0] = u;
    st[nw * 64] = v;
    st[2 * nw * 64] = mu;
    st[3 * nw * 64] = nb;
#pragma unroll
    for (int q = 0; q < JQ_MAXNC; ++q)
        if (q < Nc) st[(size_t)(JQ_ROWLANE_ARRAYS + q) * nw * 64] = carry[q];
}

// (Round 6: this kernel keeps the `o` / `nxt` operator structs with their copies between steps -- the copy-free rotation of the other
//  row-lane kernels, JQ_RL_ROTATE, was measured here too: at NPJ = 12 it bought 5 % of the backward sweep, at NPJ = 16, where the kernel
//  holds the operators of both roles in 512 registers, it tripled the spilled registers (54 -> 168 .. 245) and cost 24 %.)
// The backward sweep on TWO waves per four columns (round 3): wave 0 re-integrates the state, wave 1 runs the adjoint step and the
// traces of the same time step -- the adjoint step needs the state step only through u(t_{n+1}), v05 and u(t_n) of its own lanes,
// which the state wave leaves in a double-buffered LDS record; ONE workgroup barrier per time step keeps the state wave at most
// one step ahead.  A lone wave issues one instruction every ~11 cycles whatever its kind (probes/dp_rate_probe.hip), so the sweep
// is bound by the instruction count of its longest chain: 18 + 4 Nc products + the trace reductions instead of 36 + 4 Nc.
// Same state file, trace records and results as k_backward_rowlane (the arithmetic of each chain is unchanged: bit-identical).
// Dynamic LDS: [constant images (NPJ > 8) | records 2 x 3 x 64 doubles].
template <int NPJ>
__global__ __launch_bounds__(128) void k_backward_rowlane2(PropArgs a)
{
    constexpr bool BLK = (NPJ <= 12);      // (see rmv)
    const int lane = threadIdx.x & 63;
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // 0: state chain, 1: adjoint chain
    const int row = lane & 15;
    const long long w = blockIdx.x, nw = a.nslabs;
    const long long col = 4 * w + (lane >> 4);
    const int Nc = a.Ncoupled;
    const double wd = a.tabs[row];
    double* st = a.state + w * 64 + lane;
    const double sw = 0.5 * a.h * a.colinfo[col] * a.tabs[16 + row];
    constexpr bool RESIDENT = (NPJ <= 8);
    extern __shared__ double lds_c[];
    double* rec = lds_c + (RESIDENT ? 0 : (size_t)2 * Nc * a.stride) + lane;      // [slot][u, v05, un][64]
    if (!RESIDENT) {
        for (int i = threadIdx.x; i < 2 * Nc * (int)a.stride; i += 128) {
            const int im = i / (int)a.stride, e = i - im * (int)a.stride;
            lds_c[im * (int)a.stride + (e % NPJ) * 16 + e / NPJ] = a.cimg[i];
        }
    }
    __syncthreads();
    RowOps<NPJ> o, nxt;
    o.Kn0 = row_load<NPJ>(as_const(a.stream), row);
    o.S0 = row_load<NPJ>(as_const(a.stream) + a.stride, row);
    rops_load_half(o, a, 0, row);

    if (role == 0) {
        // ---- state chain: one step ahead of the adjoint chain at most
        double u = st[0], v = st[nw * 64];
        for (int n = 0; n < a.nsteps_chunk; ++n) {
            rops_load_half(nxt, a, min(n + 1, a.nsteps_chunk - 1), row);
            double un, v05, vnew;
            row_state<NPJ, RowOps<NPJ>, BLK>(a, o, sw, u, v, un, v05, vnew);
            double* r = rec + (n & 1) * 192;
            r[0] = u;
            r[64] = v05;
            r[128] = un;
            u = un;
            v = vnew;
            rops_advance(o, nxt);
            __syncthreads();      // record n is published (and record n - 1 has been consumed: its slot is the next one written)
        }
        st[0] = u;
        st[nw * 64] = v;
        return;
    }
    // ---- adjoint chain
    double mu = st[2 * nw * 64], nb = st[3 * nw * 64];
    const double wgt = a.colinfo[4 * nw + col];
    const double cfw = (a.forced ? 0.5 * a.h * a.tinv : 0.0) * wd;
    double carry[JQ_MAXNC];
    RowMat<NPJ> Hs[JQ_MAXNC], Ha[JQ_MAXNC];
#pragma unroll
    for (int q = 0; q < JQ_MAXNC; ++q) {
        const int qq = min(q, Nc - 1);
        carry[q] = (q < Nc) ? st[(size_t)(JQ_ROWLANE_ARRAYS + q) * nw * 64] : 0.0;
        if (RESIDENT || a.first_chunk) Hs[q] = row_load<NPJ>(as_const(a.cimg) + (size_t)qq * a.stride, row);
        if (RESIDENT) Ha[q] = row_load<NPJ>(as_const(a.cimg) + (size_t)(Nc + qq) * a.stride, row);
    }
    double* trw = a.traces + ((size_t)w * a.nsteps_chunk) * (Nc * JQ_NTR);
    if (a.first_chunk) {
        const double u0 = st[0];      // (read before the state wave stores: it stores at the end of the chunk only)
#pragma unroll
        for (int q = 0; q < JQ_MAXNC; ++q)
            if (q < Nc) carry[q] = -u0 * rmv<NPJ, true, BLK>(0.0, Hs[q], nb);
    }
    for (int n = 0; n < a.nsteps_chunk; ++n) {
        rops_load_half(nxt, a, min(n + 1, a.nsteps_chunk - 1), row);
        __syncthreads();          // the state wave has published record n
        const double* r = rec + (n & 1) * 192;
        const double u = r[0], v05 = r[64], un = r[128];
        // adjoint step! (src/StormerVerlet.jl:255-303) with nb = -lambda_i, see k_backward
        double R = rmv<NPJ, true, BLK>(0.0, o.Kp05, nb);
        if (a.use_shift) R = fma(sw, nb, R);
        R = rmv<NPJ, false, BLK>(R, o.S0, mu);
        R = fma(cfw, u, R);
        const double X = row_horner<NPJ, BLK>(mu + R, R, o.S0, a.m);
        double L = rmv<NPJ, true, BLK>(0.0, o.Kn0, X);
        if (a.use_shift) L = fma(-sw, X, L);
        double Qv = rmv<NPJ, true, BLK>(0.0, o.Kn1, X);
        if (a.use_shift) Qv = fma(-sw, X, Qv);
        {
            double P = rmv<NPJ, true, BLK>(0.0, o.S05, nb);
            P = fma(-cfw, v05, P);
            L += P;
            Qv += P;
        }
        Qv = rmv<NPJ, false, BLK>(Qv, o.S05, L);
        const double nbn = row_horner<NPJ, BLK>((nb + L) + Qv, Qv, o.S05, a.m);
        const double Bq = nb + nbn;
        double G = rmv<NPJ, false, BLK>(X, o.Kp05, nbn);
        if (a.use_shift) G = fma(sw, nbn, G);
        G = rmv<NPJ, false, BLK>(G, o.S1, X);
        G = fma(cfw, un, G);
        double t5p[JQ_MAXNC] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < JQ_MAXNC; ++q) {
            if (q < Nc) {
                if (!RESIDENT) {
                    Ha[q] = row_load_lds<NPJ>(lds_c + (size_t)(Nc + q) * a.stride, row);
                    Hs[q] = row_load_lds<NPJ>(lds_c + (size_t)q * a.stride, row);
                }
                const double HaX = rmv<NPJ, true, BLK>(0.0, Ha[q], X);
                t5p[q] = -v05 * rmv<NPJ, true, BLK>(0.0, Ha[q], Bq) * wgt;
                const double t2 = v05 * rmv<NPJ, true, BLK>(0.0, Hs[q], X) * wgt;
                const double p4 = -un * rmv<NPJ, true, BLK>(0.0, Hs[q], nbn);
                const double ts = wave_sum4_rows(u * HaX * wgt, t2, un * HaX * wgt, (p4 + carry[q]) * wgt);
                carry[q] = p4;
                if (row == 0) trw[(size_t)n * (Nc * JQ_NTR) + q * JQ_NTR + (lane >> 4)] = ts;
            }
        }
        {
            const double ts = wave_sum4_rows(t5p[0], t5p[1], t5p[2], t5p[3]);
            if (row == 0 && (lane >> 4) < Nc) trw[(size_t)n * (Nc * JQ_NTR) + (lane >> 4) * JQ_NTR + 4] = ts;
        }
        mu = G;
        nb = nbn;
        rops_advance(o, nxt);
    }
    st[2 * nw * 64] = mu;
    st[3 * nw * 64] = nb;
#pragma unroll
    for (int q = 0; q < JQ_MAXNC; ++q)
        if (q < Nc) st[(size_t)(JQ_ROWLANE_ARRAYS + q) * nw * 64] = carry[q];
}

// The backward sweep on THREE waves per four columns (round 6): wave 0 re-integrates the state, wave 1 runs the adjoint step, wave 2
// forms the traces of the gradient (adjoint_grad_calc!, src/evalobjgrad.jl:2581-2618).  In the two-wave kernel the adjoint wave carries
// 10 + 2m products of its own chain AND the 4 Nc trace products with their five wave reductions -- none of which anything waits for.
// A lone wave pays for every instruction it issues, so the sweep runs at the speed of its longest wave: 10 + 2m products now.
// The waves form a pipeline one tick (= one workgroup barrier) apart: at tick k the state wave computes step k, the adjoint wave step
// k - 1, the trace wave step k - 2.  Records in LDS: state wave -> [u(t_n), v05, u(t_n+1)] in THREE slots (read one tick later by the
// adjoint wave and two ticks later by the trace wave), adjoint wave -> [X, nb(t_n+1)] in two slots.  Every wave passes N + 1 barriers.
// The trace wave holds the constant images Hsym_q / Hanti_q in registers at every NPJ (it holds nothing else), so the LDS copy of the
// images is gone.  The arithmetic of each quantity is that of k_backward_rowlane: bit-identical results.
// State and adjoint wave read the operator rows through a ring in LDS each (RlRing: fed by the wave's own DMA, no synchronisation).
// Dynamic LDS: (3 x 3 + 2 x 2) x 64 doubles of records + two rings.
#define JQ_RL3_REC_BYTES ((size_t)(3 * 3 + 2 * 2) * 64 * 8)
#define JQ_RL3_LDS(npj) (JQ_RL3_REC_BYTES + 2 * JQ_RL_RING_BYTES(npj))
template <int NPJ>
__global__ __launch_bounds__(256) void k_backward_rowlane3(PropArgs a)
{
    const int lane = threadIdx.x & 63;
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // 0: state chain, 1: adjoint chain, 2: traces
    const int row = lane & 15;
    const long long w = blockIdx.x, nw = a.nslabs;
    const long long col = 4 * w + (lane >> 4);
    const int Nc = a.Ncoupled;
    const int N = a.nsteps_chunk;
    const double wd = a.tabs[row];
    double* st = a.state + w * 64 + lane;
    const double sw = a.use_shift ? 0.5 * a.h * a.colinfo[col] * a.tabs[16 + row] : 0.0;
    extern __shared__ double lds_c[];
    double* recS = lds_c + lane;                // [3 slots][u, v05, un][64]
    double* recA = lds_c + 3 * 192 + lane;      // [2 slots][X, nbn][64]
    double* rings = lds_c + JQ_RL3_REC_BYTES / 8;

    if (role == 0) {
        // ---- state chain
        double u = st[0], v = st[nw * 64];
        RlRing<NPJ> ring;
        ring.init(a, rings, lane, row);
        auto sweep = [&](auto terms) {
            constexpr int MT = decltype(terms)::value;
            int slot = 0;
            for (int n = 0; n < N; ++n) {
                ring.begin_step();
                const RowOpsL<NPJ> o(ring);
                double un, v05, vnew;
                row_state<NPJ, RowOpsL<NPJ>, true, MT, true>(a, o, sw, u, v, un, v05, vnew);
                double* r = recS + slot * 192;
                r[0] = u;
                r[64] = v05;
                r[128] = un;
                slot = (slot == 2) ? 0 : slot + 1;
                u = un;
                v = vnew;
                ring.end_step();
                __syncthreads();      // tick n: record n is published (its slot was last read two ticks ago)
            }
        };
        JQ_RL_DISPATCH_TERMS(a.m, sweep)
        ring.drain();
        __syncthreads();          // tick N (the trace wave's last step)
        st[0] = u;
        st[nw * 64] = v;
        return;
    }
    if (role == 1) {
        // ---- adjoint chain: adjoint step! (src/StormerVerlet.jl:255-303) with nb = -lambda_i, see k_backward
        double mu = st[2 * nw * 64], nb = st[3 * nw * 64];
        const double cfw = (a.forced ? 0.5 * a.h * a.tinv : 0.0) * wd;
        RlRing<NPJ> ring;
        ring.init(a, rings + JQ_RL_RING_BYTES(NPJ) / 8, lane, row);
        auto sweep = [&](auto terms) {
            constexpr int MT = decltype(terms)::value;
            int slot = 0;
            for (int n = 0; n < N; ++n) {
                ring.begin_step();
                const RowOpsL<NPJ> o(ring);
                __syncthreads();      // tick n: the state wave has published record n
                const double* r = recS + slot * 192;
                slot = (slot == 2) ? 0 : slot + 1;
                const double u = r[0], v05 = r[64], un = r[128];
                double R = rmv<NPJ, true>(0.0, o.Kp05, nb);
                R = fma(sw, nb, R);
                R = rmv<NPJ, false>(R, o.S0, mu);
                R = fma(cfw, u, R);
                const double X = row_horner_m<NPJ, true, MT>(mu + R, R, o.S0, a.m);
                double* ra = recA + (n & 1) * 128;
                ra[0] = X;
                double L = rmv<NPJ, true>(0.0, o.Kn0, X);
                L = fma(-sw, X, L);
                double Qv = rmv<NPJ, true>(0.0, o.Kn1, X);
                Qv = fma(-sw, X, Qv);
                {
                    double P = rmv<NPJ, true>(0.0, o.S05, nb);
                    P = fma(-cfw, v05, P);
                    L += P;
                    Qv += P;
                }
                Qv = rmv<NPJ, false>(Qv, o.S05, L);
                const double nbn = row_horner_m<NPJ, true, MT>((nb + L) + Qv, Qv, o.S05, a.m);
                ra[64] = nbn;
                double G = rmv<NPJ, false>(X, o.Kp05, nbn);
                G = fma(sw, nbn, G);
                G = rmv<NPJ, false>(G, o.S1, X);
                G = fma(cfw, un, G);
                mu = G;
                nb = nbn;
                ring.end_step();
            }
        };
        JQ_RL_DISPATCH_TERMS(a.m, sweep)
        ring.drain();
        __syncthreads();          // tick N: record N - 1 of this wave is published
        st[2 * nw * 64] = mu;
        st[3 * nw * 64] = nb;
        return;
    }
    constexpr bool BLK = true;
    // ---- traces, weighted and summed over the wave (one tick behind the adjoint wave).  With two or more controls the traces are the
    // longest chain of the three (4 Nc products + Nc + 1 wave reductions per step): a launch with 256 threads gives them TWO waves, wave t
    // forming the traces of the controls q = t mod 2 (every trace is formed by one wave exactly as before: bit-identical)
    const int tw = role - 2, tmask = (int)(blockDim.x >> 6) - 3;      // (tmask: 0 = one trace wave, 1 = two)
    const double wgt = a.colinfo[4 * nw + col];
    double nb = st[3 * nw * 64];      // (the adjoint wave stores its final nb after tick N)
    double carry[JQ_MAXNC];
    RowMat<NPJ> Hs[JQ_MAXNC], Ha[JQ_MAXNC];
#pragma unroll
    for (int q = 0; q < JQ_MAXNC; ++q) {
        const int qq = min(q, Nc - 1);
        carry[q] = (q < Nc) ? st[(size_t)(JQ_ROWLANE_ARRAYS + q) * nw * 64] : 0.0;
        Hs[q] = row_load<NPJ>(as_const(a.cimg) + (size_t)qq * a.stride, row);
        Ha[q] = row_load<NPJ>(as_const(a.cimg) + (size_t)(Nc + qq) * a.stride, row);
    }
    double* trw = a.traces + ((size_t)w * N) * (Nc * JQ_NTR);
    if (a.first_chunk) {
        const double u0 = st[0];      // (the state wave stores after tick N)
#pragma unroll
        for (int q = 0; q < JQ_MAXNC; ++q)
            if (q < Nc && (q & tmask) == tw) carry[q] = -u0 * rmv<NPJ, true, BLK>(0.0, Hs[q], nb);
    }
    __syncthreads();              // tick 0
    auto trace_sweep = [&](auto sel) {
    constexpr int SEL = decltype(sel)::value;      // 0: every control, 1: the even ones, 2: the odd ones
    int slot = 0;
    for (int n = 0; n < N; ++n) {
        __syncthreads();          // tick n + 1: the adjoint wave has published record n
        const double* r = recS + slot * 192;
        slot = (slot == 2) ? 0 : slot + 1;
        const double* ra = recA + (n & 1) * 128;
        const double u = r[0], v05 = r[64], un = r[128], X = ra[0], nbn = ra[64];
        const double Bq = nb + nbn;
        double t5p[JQ_MAXNC] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < JQ_MAXNC; ++q) {
            if (q < Nc && (SEL == 0 || (q & 1) == SEL - 1)) {
                const double HaX = rmv<NPJ, true, BLK>(0.0, Ha[q], X);
                t5p[q] = -v05 * rmv<NPJ, true, BLK>(0.0, Ha[q], Bq) * wgt;
                const double t2 = v05 * rmv<NPJ, true, BLK>(0.0, Hs[q], X) * wgt;
                const double p4 = -un * rmv<NPJ, true, BLK>(0.0, Hs[q], nbn);
                const double ts = wave_sum4_rows(u * HaX * wgt, t2, un * HaX * wgt, (p4 + carry[q]) * wgt);
                carry[q] = p4;
                if (row == 0) trw[(size_t)n * (Nc * JQ_NTR) + q * JQ_NTR + (lane >> 4)] = ts;
            }
        }
        {
            const double ts = wave_sum4_rows(t5p[0], t5p[1], t5p[2], t5p[3]);
            if (row == 0 && (lane >> 4) < Nc && (SEL == 0 || ((lane >> 4) & 1) == SEL - 1)) trw[(size_t)n * (Nc * JQ_NTR) + (lane >> 4) * JQ_NTR + 4] = ts;
        }
        nb = nbn;
    }
    };
    if (tmask == 0)
        trace_sweep(RlTerms<0>{});
    else if (tw == 0)
        trace_sweep(RlTerms<1>{});
    else
        trace_sweep(RlTerms<2>{});
#pragma unroll
    for (int q = 0; q < JQ_MAXNC; ++q)
        if (q < Nc && (q & tmask) == tw) st[(size_t)(JQ_ROWLANE_ARRAYS + q) * nw * 64] = carry[q];
}
