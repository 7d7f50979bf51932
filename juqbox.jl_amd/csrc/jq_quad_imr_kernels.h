// jq_quad_imr_kernels.h -- IMPLICIT MIDPOINT propagators (traceobjgrad for Working_Arrays_M, src/evalobjgrad.jl:1042-1481;
// m_step!, src/ImplicitMidpoint.jl:120-227; jacobi_midpoint, src/linear_solvers.jl:156-270) in the QUAD layout of
// jq_kernels.h (JQ_BW_T4Q: operators with the JQ_BW_T4 structure, four state columns per wave, a 16-row block per
// register, the four waves of a workgroup share one slab of the array file).  The algorithm and the exact reproduction
// of the reference's stopping rule are described in jq_rowlane_imr_kernels.h; this file is that algorithm with Arr<NT>
// (NT registers) in place of one double and mm_t4q products in place of the NPJ-FMA row products.  Everything is
// wave-local (no barrier, no LDS exchange inside a step, where the cooperative kernels of jq_coop_imr_kernels.h need
// one per product): the norms of the fixed-point residual are per evaluation, and an evaluation's N columns lie inside
// one quad for N = 1, 2, 4 (the host uses these kernels for those N only).
// Staging: the window mode of Ring (K, S of the midpoint = time point 2n+1 of the step; constant images resident).
#pragma once
#include "jq_kernels.h"

// sum of x over the lanes of MY evaluation (columns j with j / N equal; all rows); every lane gets its evaluation's sum
__device__ __forceinline__ double quad_sample_sum(double x, int N, int j)
{
    if (N == 4) return wave_sum(x);
    x = row_ror_add<8>(row_ror_add<4>(x));      // over the four groups of a block: lanes 4 b + j of a row
    double cs[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) cs[jj] = (lane_bcast(x, jj) + lane_bcast(x, 16 + jj)) + (lane_bcast(x, 32 + jj) + lane_bcast(x, 48 + jj));
    double r = 0.0;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) r += (jj / N == j / N) ? cs[jj] : 0.0;
    return r;
}

template <int NT>
struct QuadImr {
    const double* K;     // images of the step (LDS, lane offset applied)
    const double* S;
    const double* ws;    // shift table (LDS)
    double ceps;         // h/2 * eps of this lane's column
    int g, N, j, max_iter;
    bool use_shift;
    double tol2;

    // q = rhs + [S -K; K S] p   (K, S pre-scaled by h/2; the diagonal shift of K applied row-wise)
    __device__ __forceinline__ void apply(const Arr<NT>& rhs_u, const Arr<NT>& rhs_v, const Arr<NT>& pu, const Arr<NT>& pv, Arr<NT>& qu,
                                          Arr<NT>& qv) const
    {
        constexpr int FULL = JQ_T4_DIAG | JQ_T4_RTERMS | JQ_T4_MTERMS;
        Arr<NT> npv = pv;
        a_neg(npv);
        mm_t4q<NT, false, FULL>(qu, rhs_u, S, pu);
        mm_t4q<NT, false, FULL>(qu, qu, K, npv);
        mm_t4q<NT, false, FULL>(qv, rhs_v, K, pu);
        mm_t4q<NT, false, FULL>(qv, qv, S, pv);
        if (use_shift) {
            a_axpy_rows(qu, ceps, ws, g, npv);
            a_axpy_rows(qv, ceps, ws, g, pu);
        }
    }
    // one implicit-midpoint step of (u, v); (fu, fv): forcing already multiplied by h, added to the right-hand side
    __device__ __forceinline__ void step(Arr<NT>& u, Arr<NT>& v, const Arr<NT>& fu, const Arr<NT>& fv, bool valid) const
    {
        Arr<NT> rhs_u = u, rhs_v = v;
        a_add(rhs_u, fu);
        a_add(rhs_v, fv);
        {
            Arr<NT> tu, tv;
            apply(rhs_u, rhs_v, u, v, tu, tv);       // rhs = (x + f) + B x
            rhs_u = tu;
            rhs_v = tv;
        }
        Arr<NT> cu, cv, nu, nv;
        apply(rhs_u, rhs_v, u, v, cu, cv);           // x_1
        bool done = !valid;
        for (int it = 1; it <= max_iter; ++it) {
            apply(rhs_u, rhs_v, cu, cv, nu, nv);     // x_{it+1};  residual at x_it = x_it - x_{it+1}
            const double ru = quad_sample_sum(done ? 0.0 : a_diff2(cu, nu), N, j), rv = quad_sample_sum(done ? 0.0 : a_diff2(cv, nv), N, j);
            const bool conv = (ru < tol2) && (rv < tol2);
            if (!done && !conv && it < max_iter) {
                cu = nu;
                cv = nv;
            } else {
                done = true;                         // keeps x_it
            }
            if (__ballot(!done) == 0ull) break;
        }
        u = cu;
        v = cv;
    }
};

#define JQ_QUAD_IMR_PROLOGUE                                                                                                  \
    extern __shared__ __attribute__((aligned(16))) char smem[];                                                               \
    constexpr int KT = 4 * NT;                                                                                                \
    const int lane_ = threadIdx.x & 63;                                                                                       \
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);                                                        \
    const int col = 4 * wave + (lane_ & 3);                                                                                   \
    const int lane = ((lane_ >> 2) & 3) * 64 + 16 * (lane_ >> 4) + col;   /* this lane's offset in a block of the slab image */ \
    const int g = 4 * (lane_ >> 4) + ((lane_ >> 2) & 3);                  /* ... of the row tables */                         \
    const int slab = blockIdx.x;                                                                                              \
    double* tab = (double*)(smem + a.lds_tab_off);                                                                            \
    const double* wd = tab;                                                                                                   \
    for (int i = threadIdx.x; i < 32 * NT; i += blockDim.x) tab[(i & ~15) + 4 * (i & 3) + ((i >> 2) & 3)] = a.tabs[i];         \
    double* st = a.state + (size_t)slab * a.state_stride;                                                                     \
    Ring p;                                                                                                                   \
    p.init(smem, a, wave, lane_);      /* (window mode: ends with a barrier, the tables are published too) */                 \
    QuadImr<NT> m;                                                                                                            \
    m.ws = tab + 16 * NT;                                                                                                     \
    m.ceps = 0.5 * a.h * a.colinfo[(size_t)slab * 32 + col];                                                                  \
    m.g = g, m.N = a.N, m.j = lane_ & 3, m.max_iter = a.m, m.use_shift = a.use_shift, m.tol2 = a.jacobi_tol2;                 \
    const bool valid = col < (16 / a.N) * a.N;                                                                                \
    Arr<NT> zero;                                                                                                             \
    a_zero(zero);

// Forward sweep.  a.m = max_iter, a.jacobi_tol2 = tol^2; array file of the slab kernels.
template <int NT>
__global__ __launch_bounds__(256) void k_forward_quad_imr(PropArgs a)
{
    JQ_QUAD_IMR_PROLOGUE
    Arr<NT> u, v;
    a_load(u, st, lane);
    a_load(v, st + KT * 64, lane);
    // per-lane partial of the leak integral: carried by the lanes of group 0 (k_forward, quad layout)
    double leak = (((lane_ >> 2) & 3) == 0) ? st[(JQ_STATE_ARRAYS * KT + JQ_MAXNC) * 64 + 16 * (lane_ >> 4) + col] : 0.0;
    for (int n = 0; n < a.nsteps_chunk; ++n) {
        p.begin_step(n);
        m.K = p.template next_ks<0, 1>();
        m.S = p.template next_ks<1, 1>();
        Arr<NT> su = u, sv = v;
        m.step(u, v, zero, zero, valid);
        a_add(su, u);
        a_add(sv, v);
        leak += a_wsq(wd, g, su) + a_wsq(wd, g, sv);   // penal_m (src/evalobjgrad.jl:1214, :2158-2166)
        if (a.hist_r) hist_store<NT>(a, slab, col, 4 * ((lane_ >> 2) & 3) + (lane_ >> 4), n, u, v);
    }
    p.drain();
    a_store(u, st, lane);
    a_store(v, st + KT * 64, lane);
    leak = row_ror_add<8>(row_ror_add<4>(leak));
    if (((lane_ >> 2) & 3) == 0) st[(JQ_STATE_ARRAYS * KT + JQ_MAXNC) * 64 + 16 * (lane_ >> 4) + col] = leak;
}

// Backward sweep (src/evalobjgrad.jl:1290-1336): state re-integration with h < 0, adjoint m_step! with forcing
// -W (v + v_s) / T and the two gradient scalars of adjoint_grad_calc_m per control (:2660-2702), written in the slots of
// the midpoint weights of k_gradacc (jq_rowlane_imr_kernels.h): tr[3] = -(B + C)/4, tr[4] = (A + D)/4.
template <int NT>
__global__ __launch_bounds__(256) void k_backward_quad_imr(PropArgs a)
{
    JQ_QUAD_IMR_PROLOGUE
    const int Nc = a.Ncoupled;
    Arr<NT> u, v, lr, li;
    a_load(u, st, lane);
    a_load(v, st + KT * 64, lane);
    a_load(lr, st + 2 * KT * 64, lane);
    a_load(li, st + 3 * KT * 64, lane);
    const double wgt = a.colinfo[(size_t)slab * 32 + 16 + col];
    const double cfw = a.forced ? -a.h * a.tinv : 0.0;            // h * (-tinv * W): W applied row-wise
    double* trw = a.traces + ((size_t)(slab * JQ_WAVES + wave) * a.nsteps_chunk) * (Nc * JQ_NTR);
    for (int n = 0; n < a.nsteps_chunk; ++n) {
        p.begin_step(n);
        m.K = p.template next_ks<0, 1>();
        m.S = p.template next_ks<1, 1>();
        Arr<NT> su = u, sv = v, smu = lr, snu = li;
        m.step(u, v, zero, zero, valid);
        a_add(su, u);
        a_add(sv, v);
        Arr<NT> fu, fv;
        a_zero(fu);
        a_zero(fv);
        a_axpy_rows(fu, cfw, wd, g, su);
        a_axpy_rows(fv, cfw, wd, g, sv);
        m.step(lr, li, fu, fv, valid);
        a_add(smu, lr);
        a_add(snu, li);
        for (int q = 0; q < Nc; ++q) {
            const double* Hs = p.next_c(q);
            const double* Ha = p.next_c(Nc + q);
            const int bwq = a.bw_trace[q];
            Arr<NT> Y;
            mm_z_bw<NT, JQ_BW_T4Q>(Y, Hs, sv, bwq);
            const double B = -a_dot(smu, Y);
            mm_z_bw<NT, JQ_BW_T4Q>(Y, Ha, sv, bwq);
            const double D = a_dot(snu, Y);
            mm_z_bw<NT, JQ_BW_T4Q>(Y, Hs, su, bwq);
            const double C = a_dot(snu, Y);
            mm_z_bw<NT, JQ_BW_T4Q>(Y, Ha, su, bwq);
            const double A = a_dot(smu, Y);
            const double P = wave_sum((B + C) * wgt), Q = wave_sum((A + D) * wgt);
            if (lane_ == 0) {
                double* tr = trw + (size_t)n * (Nc * JQ_NTR) + q * JQ_NTR;
                tr[0] = 0.0;
                tr[1] = 0.0;
                tr[2] = 0.0;
                tr[3] = -0.25 * P;
                tr[4] = 0.25 * Q;
            }
        }
    }
    p.drain();
    a_store(u, st, lane);
    a_store(v, st + KT * 64, lane);
    a_store(lr, st + 2 * KT * 64, lane);
    a_store(li, st + 3 * KT * 64, lane);
}
