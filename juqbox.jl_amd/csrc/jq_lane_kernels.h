// jq_lane_kernels.h -- propagators for SMALL Hilbert spaces (Ntot <= 8): one LANE per state column.
//
// For Ntot <= 16 an MFMA tile is mostly padding (SWAP-02: 4 of 16 rows and 4 of 16 k) and every product is a
// ~600-cycle dependent chain (LDS round trip + 4 dependent 64-cycle MFMAs).  Here each lane owns one column
// (ensemble sample x initial condition): its state vectors are NP doubles in registers and a product
// y = M x is NP*NP v_fmac_f64 and nothing else.
//
// Operand delivery: the operators are uniform across the wave.  An NP x NP image lives in ceil(NP*NP/16)
// VGPR pairs: lane l holds elements (l & 15) + 16 r, identically in each of the four 16-lane rows, and every
// FMA reads "its" element with the DP-ALU DPP control row_newbcast:k (gfx90a+: broadcast lane k of each row
// to the row), which costs nothing -- measured on MI355X (probes/dpp_fmac_probe.hip): 36.0 TFLOP/s for
// v_fmac_f64_dpp vs 41.0 for plain v_fma_f64 at one wave per SIMD, inside a loop that is 20 % other work.
// So there is no LDS, no scalar-cache latency in the product chain and no barrier; the images of the NEXT
// time step (4 new ones: K, S at t+3h/2 and t+2h) are fetched with ordinary per-lane global loads one step
// ahead, and the compiler's own vmcnt bookkeeping covers them.
// (History: a first version fed the FMAs from SGPRs via s_load; SMEM returns out of order, so every 8..16
// doubles cost a full ~150-cycle lgkmcnt(0) round trip and the wave ran 4x below the FMA rate.)
//
// The DPP FMA is inline asm (this clang has no 64-bit update_dpp builtin; the llvm.amdgcn.update.dpp.i64
// intrinsic yields v_mov_b64_dpp + v_fma, twice the VALU work).  gfx950 requires 2 wait states between a
// VALU write of a VGPR and a DPP read of it and does NOT interlock (probe: stale data); the hazard recognizer
// cannot see into inline asm, so (1) every product starts with s_nop 1 and (2) scripts/check_dpp_hazard.py
// verifies the final ISA of every lane kernel at build time (make fails on a violation).
//
// Same math as the MFMA kernels (scaled/signed operator stream, Horner-form Neumann series, negated
// lambda_i, 4 trace products per control), same time-point schedule, same trace/gradient pipeline.
//
// Layouts:  operator images: row-major NP x NP, zero padded to a multiple of 16 doubles (a.stride);
//           stream point j -> K at (2j)*stride, S at (2j+1)*stride;
//           state file: [array][row][column] with the column index fastest.
#pragma once
#include "jq_kernels.h"
#include <utility>

// Operator images are written by earlier kernels (k_stream), never by the propagators: reading them through
// the constant address space lets the loads move freely across the kernels' own global stores.
typedef const __attribute__((address_space(4))) double* cmat_t;
__device__ __forceinline__ cmat_t as_const(const double* p) { return (cmat_t)(unsigned long long)p; }

template <int NP>
struct Vec {
    double e[NP];
};
// operator image in registers: element e of the row-major image sits in r[e / 16], lane (e % 16) of each row
template <int NP>
struct Mat {
    static constexpr int NR = (NP * NP + 15) / 16;
    double r[NR];
};

template <int NP>
__device__ __forceinline__ Mat<NP> mat_load(cmat_t p, int l16)
{
    Mat<NP> m;
#pragma unroll
    for (int k = 0; k < Mat<NP>::NR; ++k) m.r[k] = p[16 * k + l16];
    return m;
}

template <int NP>
__device__ __forceinline__ Vec<NP> v_add(const Vec<NP>& a, const Vec<NP>& b)
{
    Vec<NP> r;
#pragma unroll
    for (int i = 0; i < NP; ++i) r.e[i] = a.e[i] + b.e[i];
    return r;
}
template <int NP>
__device__ __forceinline__ double v_dot(const Vec<NP>& a, const Vec<NP>& b)
{
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NP; ++i) s += a.e[i] * b.e[i];
    return s;
}

// y += m[lane K of the row] * x
template <int K>
__device__ __forceinline__ void fma_bcast(double& y, double m, double x)
{
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(y) : "v"(m), "v"(x), "n"(K));
}
template <int NP, int... E>
__device__ __forceinline__ void mv_fold(Vec<NP>& y, const Mat<NP>& M, const Vec<NP>& x, std::integer_sequence<int, E...>)
{
    (fma_bcast<E % 16>(y.e[E / NP], M.r[E / 16], x.e[E % NP]), ...);
}
// y = c + M x   (y may alias c, not x)
template <int NP, bool ZEROC>
__device__ __forceinline__ Vec<NP> mv(const Vec<NP>& c, const Mat<NP>& M, const Vec<NP>& x)
{
    Vec<NP> y;
#pragma unroll
    for (int i = 0; i < NP; ++i) y.e[i] = ZEROC ? 0.0 : c.e[i];
    // DPP read-after-VALU-write hazard (see header): nothing the register allocator places in front of the
    // product (copies, AGPR reloads of M) may sit closer than 2 wait states to the first FMA
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 1");
    __builtin_amdgcn_sched_barrier(0);   // (the non-volatile FMAs may otherwise move above the s_nop)
    mv_fold<NP>(y, M, x, std::make_integer_sequence<int, NP * NP>{});
    return y;
}
// y += s * tab .* x     (per-lane tables)
template <int NP>
__device__ __forceinline__ void v_axpy_rows(Vec<NP>& y, double s, const Vec<NP>& tab, const Vec<NP>& x)
{
#pragma unroll
    for (int i = 0; i < NP; ++i) y.e[i] = fma(s * tab.e[i], x.e[i], y.e[i]);
}
template <int NP>
__device__ __forceinline__ double v_wsq(const Vec<NP>& tab, const Vec<NP>& x)
{
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NP; ++i) s = fma(tab.e[i], x.e[i] * x.e[i], s);
    return s;
}
template <int NP>
__device__ __forceinline__ Vec<NP> tab_load(const double* t)
{
    Vec<NP> r;
#pragma unroll
    for (int i = 0; i < NP; ++i) r.e[i] = t[i];
    return r;
}
// bpa + sum_{j=1..m} S^j A  (Horner form, see jq_kernels.h)
template <int NP>
__device__ __forceinline__ Vec<NP> lane_horner(const Vec<NP>& bpa, const Vec<NP>& A, const Mat<NP>& S, int m)
{
    if (m <= 0) return bpa;
    Vec<NP> Y = A;
    // two products per loop iteration: a taken branch costs the SIMD 30 - 80 cycles of issue time, as much as a product of NP = 4
    // (probes/lone_wave_probe.hip; SWAP-02 x 8 192 samples 28.3 -> 26.2 ms; four per iteration: 26.8)
    int j = m - 1;
    for (; j >= 2; j -= 2) {
        Y = mv<NP, false>(A, S, Y);
        Y = mv<NP, false>(A, S, Y);
    }
    if (j > 0) Y = mv<NP, false>(A, S, Y);
    return mv<NP, false>(bpa, S, Y);
}

// the six operator images of one time step: K, S at t (0), t + h/2 (05), t + h (1); scaled and signed by
// k_stream (Kp = +cK at half points, Kn = -cK at integer points, S = cS, c = h/2)
template <int NP>
struct LaneOps {
    Mat<NP> Kn0, S0, Kp05, S05, Kn1, S1;
};
template <int NP>
__device__ __forceinline__ void ops_load_half(LaneOps<NP>& o, const PropArgs& a, int n, int l16)
{
    cmat_t s = as_const(a.stream) + (size_t)(2 * (2 * n + 1)) * a.stride;
    o.Kp05 = mat_load<NP>(s, l16);
    o.S05 = mat_load<NP>(s + a.stride, l16);
    o.Kn1 = mat_load<NP>(s + 2 * a.stride, l16);
    o.S1 = mat_load<NP>(s + 3 * a.stride, l16);
}
template <int NP>
__device__ __forceinline__ void ops_load_first(LaneOps<NP>& o, const PropArgs& a, int l16)
{
    cmat_t s = as_const(a.stream);
    o.Kn0 = mat_load<NP>(s, l16);
    o.S0 = mat_load<NP>(s + a.stride, l16);
    ops_load_half(o, a, 0, l16);
}
// roles for the next step: (K,S)(t+h) become (K,S)(t); the four images fetched one step ahead move in
template <int NP>
__device__ __forceinline__ void ops_advance(LaneOps<NP>& o, const LaneOps<NP>& nxt)
{
    o.Kn0 = o.Kn1;
    o.S0 = o.S1;
    o.Kp05 = nxt.Kp05;
    o.S05 = nxt.S05;
    o.Kn1 = nxt.Kn1;
    o.S1 = nxt.S1;
}

// One Stormer-Verlet state step (forward step!, src/StormerVerlet.jl:461-504) in the accumulate form.
// sw = eps * c * ws per lane (reference perturbation of diag(Hconst), src/ipopt_interface.jl:41-44).
template <int NP>
__device__ __forceinline__ void lane_state(const PropArgs& a, const LaneOps<NP>& o, const Vec<NP>& sw, const Vec<NP>& u,
                                           const Vec<NP>& v, Vec<NP>& un, Vec<NP>& v05, Vec<NP>& vnew)
{
    Vec<NP> A = mv<NP, true>(u, o.Kp05, u);
    if (a.use_shift) v_axpy_rows(A, 1.0, sw, u);
    A = mv<NP, false>(A, o.S05, v);
    v05 = lane_horner<NP>(v_add(v, A), A, o.S05, a.m);
    const Vec<NP> vN = mv<NP, false>(v05, o.S05, v05);
    un = mv<NP, false>(u, o.Kn0, v05);
    if (a.use_shift) v_axpy_rows(un, -1.0, sw, v05);
    un = mv<NP, false>(un, o.S0, u);
    A = mv<NP, true>(u, o.Kn1, v05);
    if (a.use_shift) v_axpy_rows(A, -1.0, sw, v05);
    A = mv<NP, false>(A, o.S1, un);
    un = lane_horner<NP>(v_add(un, A), A, o.S1, a.m);
    vnew = mv<NP, false>(vN, o.Kp05, un);
    if (a.use_shift) v_axpy_rows(vnew, 1.0, sw, un);
}

// state file access: [array][row][column]
template <int NP>
__device__ __forceinline__ Vec<NP> lane_load(const double* st, int arr, long long ncols, long long col)
{
    Vec<NP> r;
#pragma unroll
    for (int i = 0; i < NP; ++i) r.e[i] = st[((size_t)arr * NP + i) * ncols + col];
    return r;
}
template <int NP>
__device__ __forceinline__ void lane_store(double* st, int arr, long long ncols, long long col, const Vec<NP>& x)
{
#pragma unroll
    for (int i = 0; i < NP; ++i) st[((size_t)arr * NP + i) * ncols + col] = x.e[i];
}

#define JQ_LANE_ARRAYS 4                         // U, V, MU, NB
#define JQ_LANE_ROWS(NP) (JQ_LANE_ARRAYS * (NP) + JQ_MAXNC + 1)   // + carry rows + leak row

// Forward sweep of one chunk; a.nslabs = padded number of columns (multiple of 64), grid = columns/64.
template <int NP>
__global__ __launch_bounds__(64) void k_forward_lane(PropArgs a)
{
    const long long col = (long long)blockIdx.x * 64 + threadIdx.x;
    const long long ncols = a.nslabs;
    const int l16 = threadIdx.x & 15;
    const Vec<NP> wd = tab_load<NP>(a.tabs);
    Vec<NP> sw = tab_load<NP>(a.tabs + NP);
    Vec<NP> u = lane_load<NP>(a.state, 0, ncols, col), v = lane_load<NP>(a.state, 1, ncols, col);
    double leak = a.state[((size_t)JQ_LANE_ARRAYS * NP + JQ_MAXNC) * ncols + col];
    const double ceps = 0.5 * a.h * a.colinfo[col];
#pragma unroll
    for (int i = 0; i < NP; ++i) sw.e[i] *= ceps;
    LaneOps<NP> o, nxt;
    ops_load_first(o, a, l16);
    for (int n = 0; n < a.nsteps_chunk; ++n) {
        ops_load_half(nxt, a, min(n + 1, a.nsteps_chunk - 1), l16);   // lands during this step
        Vec<NP> un, v05, vnew;
        leak += v_wsq<NP>(wd, u);
        lane_state<NP>(a, o, sw, u, v, un, v05, vnew);
        u = un;
        v = vnew;
        leak += v_wsq<NP>(wd, u) + 2.0 * v_wsq<NP>(wd, v05);
        if (a.hist_r && col < a.N) {
            const size_t off = (size_t)(a.step0 + n + 1) * a.Ntot * a.N + (size_t)col * a.Ntot;
#pragma unroll
            for (int i = 0; i < NP; ++i)
                if (i < a.Ntot) {
                    a.hist_r[off + i] = u.e[i];
                    a.hist_i[off + i] = -v.e[i];
                }
        }
        ops_advance(o, nxt);
    }
    lane_store<NP>(a.state, 0, ncols, col, u);
    lane_store<NP>(a.state, 1, ncols, col, v);
    a.state[((size_t)JQ_LANE_ARRAYS * NP + JQ_MAXNC) * ncols + col] = leak;
}

// Backward sweep of one chunk (state re-integration, adjoint step, trace scalars per wave and step).
// a.cimg: constant images [Hsym_q | Hanti_q], resident in registers for the whole sweep.
template <int NP>
__global__ __launch_bounds__(64) void k_backward_lane(PropArgs a)
{
    const long long col = (long long)blockIdx.x * 64 + threadIdx.x;
    const long long ncols = a.nslabs;
    const int lane = threadIdx.x;
    const int l16 = lane & 15;
    const int Nc = a.Ncoupled;
    const Vec<NP> wd = tab_load<NP>(a.tabs);
    Vec<NP> sw = tab_load<NP>(a.tabs + NP);
    Vec<NP> u = lane_load<NP>(a.state, 0, ncols, col), v = lane_load<NP>(a.state, 1, ncols, col);
    Vec<NP> mu = lane_load<NP>(a.state, 2, ncols, col), nb = lane_load<NP>(a.state, 3, ncols, col);
    const double ceps = 0.5 * a.h * a.colinfo[col];
#pragma unroll
    for (int i = 0; i < NP; ++i) sw.e[i] *= ceps;
    const double wgt = a.colinfo[ncols + col];
    const double cfw = a.forced ? 0.5 * a.h * a.tinv : 0.0;
    double carry[JQ_MAXNC];
    // constant images: resident in registers for NP <= 6; for NP = 8 (32 more VGPR pairs at 4 controls) they
    // are re-fetched every step so that nothing spills to AGPRs
    constexpr bool RESIDENT = (NP <= 6);
    Mat<NP> Hs[JQ_MAXNC], Ha[JQ_MAXNC];
#pragma unroll
    for (int q = 0; q < JQ_MAXNC; ++q) {
        const int qq = min(q, Nc - 1);
        carry[q] = (q < Nc) ? a.state[((size_t)JQ_LANE_ARRAYS * NP + q) * ncols + col] : 0.0;
        if (RESIDENT || a.first_chunk) Hs[q] = mat_load<NP>(as_const(a.cimg) + (size_t)qq * a.stride, l16);
        if (RESIDENT) Ha[q] = mat_load<NP>(as_const(a.cimg) + (size_t)(Nc + qq) * a.stride, l16);
    }
    double* trw = a.traces + ((size_t)blockIdx.x * a.nsteps_chunk) * (Nc * JQ_NTR);

    if (a.first_chunk) {
        // carry_q = tr(vr' Hsym_q lambdai) at t = T (see k_backward)
#pragma unroll
        for (int q = 0; q < JQ_MAXNC; ++q)
            if (q < Nc) carry[q] = -v_dot(u, mv<NP, true>(u, Hs[q], nb));
    }

    LaneOps<NP> o, nxt;
    ops_load_first(o, a, l16);
    for (int n = 0; n < a.nsteps_chunk; ++n) {
        ops_load_half(nxt, a, min(n + 1, a.nsteps_chunk - 1), l16);   // lands during this step
        Vec<NP> un, v05, vnew;
        lane_state<NP>(a, o, sw, u, v, un, v05, vnew);
        // adjoint step! (src/StormerVerlet.jl:255-303) with nb = -lambda_i, see k_backward
        Vec<NP> R = mv<NP, true>(u, o.Kp05, nb);
        if (a.use_shift) v_axpy_rows(R, 1.0, sw, nb);
        R = mv<NP, false>(R, o.S0, mu);
        v_axpy_rows(R, cfw, wd, u);
        const Vec<NP> X = lane_horner<NP>(v_add(mu, R), R, o.S0, a.m);
        Vec<NP> L = mv<NP, true>(u, o.Kn0, X);
        if (a.use_shift) v_axpy_rows(L, -1.0, sw, X);
        Vec<NP> Qv = mv<NP, true>(u, o.Kn1, X);
        if (a.use_shift) v_axpy_rows(Qv, -1.0, sw, X);
        {
            Vec<NP> P = mv<NP, true>(u, o.S05, nb);
            v_axpy_rows(P, -cfw, wd, v05);
            L = v_add(L, P);
            Qv = v_add(Qv, P);
        }
        Qv = mv<NP, false>(Qv, o.S05, L);
        const Vec<NP> nbn = lane_horner<NP>(v_add(v_add(nb, L), Qv), Qv, o.S05, a.m);
        const Vec<NP> Bq = v_add(nb, nbn);
        Vec<NP> G = mv<NP, false>(X, o.Kp05, nbn);
        if (a.use_shift) v_axpy_rows(G, 1.0, sw, nbn);
        G = mv<NP, false>(G, o.S1, X);
        v_axpy_rows(G, cfw, wd, un);
        // traces (adjoint_grad_calc!, src/evalobjgrad.jl:2581-2618), weighted and summed over the wave
        double t5p[JQ_MAXNC] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < JQ_MAXNC; ++q) {
            if (q < Nc) {
                if (!RESIDENT) {
                    Hs[q] = mat_load<NP>(as_const(a.cimg) + (size_t)q * a.stride, l16);
                    Ha[q] = mat_load<NP>(as_const(a.cimg) + (size_t)(Nc + q) * a.stride, l16);
                }
                const Vec<NP> HaX = mv<NP, true>(u, Ha[q], X);
                t5p[q] = -v_dot(v05, mv<NP, true>(u, Ha[q], Bq)) * wgt;
                const double t2 = v_dot(v05, mv<NP, true>(u, Hs[q], X)) * wgt;
                const double p4 = -v_dot(un, mv<NP, true>(u, Hs[q], nbn));
                // t1 .. t4 of the control in ONE reduction: lane 16 r holds the sum of the r-th value
                const double ts = wave_sum4_rows(v_dot(u, HaX) * wgt, t2, v_dot(un, HaX) * wgt, (p4 + carry[q]) * wgt);
                carry[q] = p4;
                if ((lane & 15) == 0) trw[(size_t)n * (Nc * JQ_NTR) + q * JQ_NTR + (lane >> 4)] = ts;
            }
        }
        {   // ... and the t5 of all (at most four) controls in one more
            const double ts = wave_sum4_rows(t5p[0], t5p[1], t5p[2], t5p[3]);
            if ((lane & 15) == 0 && (lane >> 4) < Nc) trw[(size_t)n * (Nc * JQ_NTR) + (lane >> 4) * JQ_NTR + 4] = ts;
        }
        u = un;
        v = vnew;
        mu = G;
        nb = nbn;
        ops_advance(o, nxt);
    }
    lane_store<NP>(a.state, 0, ncols, col, u);
    lane_store<NP>(a.state, 1, ncols, col, v);
    lane_store<NP>(a.state, 2, ncols, col, mu);
    lane_store<NP>(a.state, 3, ncols, col, nb);
#pragma unroll
    for (int q = 0; q < JQ_MAXNC; ++q)
        if (q < Nc) a.state[((size_t)JQ_LANE_ARRAYS * NP + q) * ncols + col] = carry[q];
}

// state file <- (Uinit, 0, 0, 0, 0...).  uinit: [N][NP] (column i of Uinit, zero padded).  thread per column
template <int NP>
__global__ void k_init_state_lane(double* state, long long ncols, const double* __restrict__ uinit, int N, long long ncols_used)
{
    const long long col = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= ncols) return;
    const int ic = (int)(col % N);
    for (int r = 0; r < JQ_LANE_ROWS(NP); ++r) {
        double val = 0.0;
        if (r < NP && col < ncols_used) val = uinit[ic * NP + r];
        state[(size_t)r * ncols + col] = val;
    }
}

// fidelity, leak and adjoint terminal condition per sample (thread per sample; see k_terminal)
template <int NP>
__global__ void k_terminal_lane(double* state, long long ncols, const double* __restrict__ vtr, const double* __restrict__ vti,
                                int N, int nsamples, double leak_scale, double* res)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsamples) return;
    double re = 0.0, im = 0.0, lk = 0.0;
    for (int ic = 0; ic < N; ++ic) {
        const long long col = (long long)s * N + ic;
        for (int r = 0; r < NP; ++r) {
            const double u = state[(size_t)r * ncols + col], v = state[((size_t)NP + r) * ncols + col];
            const double tr = vtr[ic * NP + r], ti = vti[ic * NP + r];
            re += u * tr - v * ti;
            im += u * ti + v * tr;
        }
        lk += state[((size_t)JQ_LANE_ARRAYS * NP + JQ_MAXNC) * ncols + col];
    }
    re /= N;
    im /= N;
    for (int ic = 0; ic < N; ++ic) {
        const long long col = (long long)s * N + ic;
        for (int r = 0; r < NP; ++r) {
            const double tr = vtr[ic * NP + r], ti = vti[ic * NP + r];
            state[((size_t)2 * NP + r) * ncols + col] = (re * tr + im * ti) / N;      // lambda_r
            state[((size_t)3 * NP + r) * ncols + col] = -((im * tr - re * ti) / N);   // nb = -lambda_i
        }
    }
    res[(size_t)s * 4 + 0] = 1.0 - (re * re + im * im);
    res[(size_t)s * 4 + 1] = leak_scale * lk;
    res[(size_t)s * 4 + 2] = re;
    res[(size_t)s * 4 + 3] = im;
}
