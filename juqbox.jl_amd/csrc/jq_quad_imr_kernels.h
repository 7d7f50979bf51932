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

// OPREG: the operators of a step live in registers for all its fixed-point iterations (one slab per workgroup, one wave per
// SIMD: 512 registers); otherwise (two slabs per workgroup, two waves per SIMD: 256 registers) every application re-reads
// its A operands and coefficient records from the LDS images.
#ifndef JQ_IMR_CREG      // experiment: coupling-coefficient records of the step's operators in registers (1) or re-read from LDS (0)
#define JQ_IMR_CREG 1
#endif
template <int NT, bool OPREG>
struct QuadImr {
    const double* K;     // images of the step (LDS, lane offset applied)
    const double* S;
    const double* ws;    // shift table (LDS)
    double ceps;         // h/2 * eps of this lane's column
    int g, N, j, max_iter;
    bool use_shift;
    double tol2;
    // the operators of the step in registers (every fixed-point iteration multiplies with the same K, S) and this lane's
    // ensemble-shift coefficients ceps * ws[row] per block (constant over the sweep)
    OpQ<NT> oK, oS;
    double cw[NT];
    // N = 4: the four columns of the quad are ONE evaluation (one ensemble shift eps), so the shift eps * diag(ws) of K can sit on
    // the diagonal of the MFMA's A operand (lane 16 k + 4 b + i holds B[i][k]: the lanes with k == i) instead of costing two FMAs
    // per block and application: cwa = this lane's addend to oK.a (zero off the diagonal)
    double cwa[NT];

    __device__ __forceinline__ bool folded() const { return OPREG && N == 4; }
    __device__ __forceinline__ void load_ops()
    {
        if constexpr (OPREG) {
            t4q_load<NT>(oK, K);
            t4q_load<NT>(oS, S);
            if (N == 4) {
#pragma unroll
                for (int i = 0; i < NT; ++i) oK.a[i] += cwa[i];
            }
        }
    }
    __device__ __forceinline__ void init_shift()
    {
        const int lane = threadIdx.x & 63;
        const bool dg = (lane >> 4) == (lane & 3);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            cw[i] = use_shift ? ceps * ws[16 * i + g] : 0.0;
            cwa[i] = (use_shift && dg) ? ceps * ws[16 * i + 4 * (lane & 3) + ((lane >> 2) & 3)] : 0.0;
        }
    }

    // (qu, qv) = (rhs_u, rhs_v) + [S -K; K S] (pu, pv)   (K, S pre-scaled by h/2; the diagonal ensemble shift of K row-wise).
    // ONE pass over the blocks: the two products with pu (S pu, K pu) share its lane shifts, the two with pv likewise --
    // 48 v_mov_b32_dpp per application instead of 96 (round 2: four independent mm_t4q) -- and -K pv is formed with the
    // FMA's sign modifier and one sign flip per block for the MFMA operand instead of a negated copy of pv.
    // FOLD: the ensemble shift is part of oK.a (folded()): no shift FMAs
    template <bool FOLD = false>
    __device__ __forceinline__ void apply(const Arr<NT>& rhs_u, const Arr<NT>& rhs_v, const Arr<NT>& pu, const Arr<NT>& pv, Arr<NT>& qu,
                                          Arr<NT>& qv) const
    {
        const int lane = threadIdx.x & 63;
        const double *Kp = K, *Sp = S;
        if constexpr (!OPREG || !JQ_IMR_CREG) {
            // the images do not change between the fixed-point iterations: without this the compiler hoists all 60 operand
            // reads out of the iteration loop -- into registers this variant does not have (184 spilled to scratch)
            // (an opaque ZERO offset, not an opaque pointer: the pointers must stay visibly LDS addresses -- ds_read, not flat loads)
            int z = 0;
            asm volatile("" : "+v"(z));
            Kp += z;
            Sp += z;
        }
        const double *maK = t4q_a(Kp, lane), *maS = t4q_a(Sp, lane);
        const d4 *cfK = t4q_c<NT>(Kp, lane), *cfS = t4q_c<NT>(Sp, lane);
        // pass 1: the four MFMAs of every block (two chains of two per block, the chains of all blocks independent of each other);
        // pass 2: lane shifts and coupling FMAs.  In ONE pass per block every MFMA result was consumed by the next instruction:
        // 26 hazard s_nop per application at the lone wave's issue rate (round 3: 321 -> 295 instructions per iteration)
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) {
            const double xu = pu.t[mt][0], xv = pv.t[mt][0];
            const double aS = OPREG ? oS.a[mt] : maS[mt * 64], aK = OPREG ? oK.a[mt] : maK[mt * 64];
            double au = __builtin_amdgcn_mfma_f64_4x4x4f64(aS, xu, rhs_u.t[mt][0], 0, 0, 0);
            double av = __builtin_amdgcn_mfma_f64_4x4x4f64(aK, xu, rhs_v.t[mt][0], 0, 0, 0);
            au = __builtin_amdgcn_mfma_f64_4x4x4f64(aK, -xv, au, 0, 0, 0);
            av = __builtin_amdgcn_mfma_f64_4x4x4f64(aS, xv, av, 0, 0, 0);
            qu.t[mt][0] = au;
            qv.t[mt][0] = av;
        }
        double uold = 0.0, vold = 0.0;
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) {
            const double xu = pu.t[mt][0], xv = pv.t[mt][0];
            const double un = pu.t[mt + 1 < NT ? mt + 1 : mt][0], vn = pv.t[mt + 1 < NT ? mt + 1 : mt][0];
            const double uu = row_shift4<0x114>(xu), ud = row_shift4<0x104>(xu);
            const double vu = row_shift4<0x114>(xv), vd = row_shift4<0x104>(xv);
            const d4 cs = (OPREG && JQ_IMR_CREG) ? oS.c[mt] : t4q_cload(cfS, mt), ck = (OPREG && JQ_IMR_CREG) ? oK.c[mt] : t4q_cload(cfK, mt);
            double au = qu.t[mt][0], av = qv.t[mt][0];
            au = fma(cs[0], uu, au);
            av = fma(ck[0], uu, av);
            au = fma(cs[1], ud, au);
            av = fma(ck[1], ud, av);
            au = fma(-ck[0], vu, au);
            av = fma(cs[0], vu, av);
            au = fma(-ck[1], vd, au);
            av = fma(cs[1], vd, av);
            if (mt > 0) {
                au = fma(cs[2], uold, au);
                av = fma(ck[2], uold, av);
                au = fma(-ck[2], vold, au);
                av = fma(cs[2], vold, av);
            }
            if (mt + 1 < NT) {
                au = fma(cs[3], un, au);
                av = fma(ck[3], un, av);
                au = fma(-ck[3], vn, au);
                av = fma(cs[3], vn, av);
            }
            if constexpr (!FOLD) {
                au = fma(-cw[mt], xv, au);      // (cw = 0 without an ensemble shift: cheaper than a select per block)
                av = fma(cw[mt], xu, av);
            }
            uold = xu;
            vold = xv;
            qu.t[mt][0] = au;
            qv.t[mt][0] = av;
            if constexpr (!OPREG) __builtin_amdgcn_sched_barrier(0);
        }
    }
    // N = 4: the 64 lanes are ONE evaluation -- the stopping decision is wave-uniform, the iterates alternate between two register
    // sets (no copies), both norms come out of one reduction
    template <bool FOLD>
    __device__ __forceinline__ void step4(Arr<NT>& u, Arr<NT>& v, const Arr<NT>& fu, const Arr<NT>& fv) const
    {
        // B x ONCE: rhs = (x + f) + B x and x_1 = rhs + B x (as the cooperative kernels do; round 4: one application of ~ 7.7 per
        // step less)
        Arr<NT> rhs_u = u, rhs_v = v, cu, cv, nu, nv;
        a_add(rhs_u, fu);
        a_add(rhs_v, fv);
        {
            Arr<NT> z;
            a_zero(z);
            apply<FOLD>(z, z, u, v, nu, nv);               // B x
        }
        a_add(rhs_u, nu);
        a_add(rhs_v, nv);
        cu = rhs_u, cv = rhs_v;
        a_add(cu, nu);                                     // x_1
        a_add(cv, nv);
        for (int it = 1;; it += 2) {
            apply<FOLD>(rhs_u, rhs_v, cu, cv, nu, nv);     // x_{it+1};  residual at x_it = x_it - x_{it+1}
            bool conv = __all(wave_sum2(a_diff2(cu, nu), a_diff2(cv, nv)) < tol2);
            if (conv || it >= max_iter) {                  // keeps x_it
                u = cu;
                v = cv;
                return;
            }
            apply<FOLD>(rhs_u, rhs_v, nu, nv, cu, cv);     // x_{it+2}
            conv = __all(wave_sum2(a_diff2(nu, cu), a_diff2(nv, cv)) < tol2);
            if (conv || it + 1 >= max_iter) {              // keeps x_{it+1}
                u = nu;
                v = nv;
                return;
            }
        }
    }
    // one implicit-midpoint step of (u, v); (fu, fv): forcing already multiplied by h, added to the right-hand side
    __device__ __forceinline__ void step(Arr<NT>& u, Arr<NT>& v, const Arr<NT>& fu, const Arr<NT>& fv, bool valid) const
    {
        if (N == 4) {
            step4<OPREG>(u, v, fu, fv);
            return;
        }
        Arr<NT> rhs_u = u, rhs_v = v, cu, cv, nu, nv;
        a_add(rhs_u, fu);
        a_add(rhs_v, fv);
        {
            Arr<NT> z;
            a_zero(z);
            apply(z, z, u, v, nu, nv);               // B x (once, see step4)
        }
        a_add(rhs_u, nu);
        a_add(rhs_v, nv);
        cu = rhs_u, cv = rhs_v;
        a_add(cu, nu);                               // x_1
        a_add(cv, nv);
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
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      /* (SPW slabs per workgroup: 4 SPW waves) */      \
    const int col = 4 * (wave & 3) + (lane_ & 3);                                                                             \
    const int lane = ((lane_ >> 2) & 3) * 64 + 16 * (lane_ >> 4) + col;   /* this lane's offset in a block of the slab image */ \
    const int g = 4 * (lane_ >> 4) + ((lane_ >> 2) & 3);                  /* ... of the row tables */                         \
    const int slab = (int)blockIdx.x * SPW + (wave >> 2);                                                                     \
    const bool active = slab < a.nslabs;                                                                                      \
    double* tab = (double*)(smem + a.lds_tab_off);                                                                            \
    const double* wd = tab;                                                                                                   \
    for (int i = threadIdx.x; i < 32 * NT; i += blockDim.x) tab[(i & ~15) + 4 * (i & 3) + ((i >> 2) & 3)] = a.tabs[i];         \
    double* st = a.state + (size_t)(active ? slab : 0) * a.state_stride;                                                      \
    RingT<true> p;                                                                                                                \
    p.init(smem, a, wave, lane_, 4 * SPW);      /* (window mode: ends with a barrier, the tables are published too) */        \
    QuadImr<NT, SPW == 1> m;                                                                                                  \
    m.ws = tab + 16 * NT;                                                                                                     \
    m.ceps = active ? 0.5 * a.h * a.colinfo[(size_t)slab * 32 + col] : 0.0;                                                   \
    m.g = g, m.N = a.N, m.j = lane_ & 3, m.max_iter = a.m, m.use_shift = a.use_shift, m.tol2 = a.jacobi_tol2;                 \
    m.init_shift();                                                                                                           \
    const bool valid = active && col < (16 / a.N) * a.N;                                                                      \
    Arr<NT> zero;                                                                                                             \
    a_zero(zero);

// Forward sweep.  a.m = max_iter, a.jacobi_tol2 = tol^2; array file of the slab kernels.
template <int NT, int SPW>
__global__ __launch_bounds__(256 * SPW) void k_forward_quad_imr(PropArgs a)
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
        m.load_ops();
        if (!active) continue;      // (a wave without a slab only takes part in the staging)
        Arr<NT> su = u, sv = v;
        m.step(u, v, zero, zero, valid);
        a_add(su, u);
        a_add(sv, v);
        leak += a_wsq(wd, g, su) + a_wsq(wd, g, sv);   // penal_m (src/evalobjgrad.jl:1214, :2158-2166)
        if (a.hist_r) hist_store<NT>(a, slab, col, 4 * ((lane_ >> 2) & 3) + (lane_ >> 4), n, u, v);
    }
    p.drain();
    if (!active) return;
    a_store(u, st, lane);
    a_store(v, st + KT * 64, lane);
    leak = row_ror_add<8>(row_ror_add<4>(leak));
    if (((lane_ >> 2) & 3) == 0) st[(JQ_STATE_ARRAYS * KT + JQ_MAXNC) * 64 + 16 * (lane_ >> 4) + col] = leak;
}

// Backward sweep (src/evalobjgrad.jl:1290-1336): state re-integration with h < 0, adjoint m_step! with forcing
// -W (v + v_s) / T and the two gradient scalars of adjoint_grad_calc_m per control (:2660-2702), written in the slots of
// the midpoint weights of k_gradacc (jq_rowlane_imr_kernels.h): tr[3] = -(B + C)/4, tr[4] = (A + D)/4.
template <int NT, int SPW>
__global__ __launch_bounds__(256 * SPW) void k_backward_quad_imr(PropArgs a)
{
    JQ_QUAD_IMR_PROLOGUE
    const int Nc = a.Ncoupled;
    Arr<NT> u, v, lr, li;
    a_load(u, st, lane);
    a_load(v, st + KT * 64, lane);
    a_load(lr, st + 2 * KT * 64, lane);
    a_load(li, st + 3 * KT * 64, lane);
    const double wgt = active ? a.colinfo[(size_t)slab * 32 + 16 + col] : 0.0;
    const double cfw = a.forced ? -a.h * a.tinv : 0.0;            // h * (-tinv * W): W applied row-wise
    double* trw = a.traces + ((size_t)((active ? slab : 0) * JQ_WAVES + (wave & 3)) * a.nsteps_chunk) * (Nc * JQ_NTR);
    for (int n = 0; n < a.nsteps_chunk; ++n) {
        p.begin_step(n);
        m.K = p.template next_ks<0, 1>();
        m.S = p.template next_ks<1, 1>();
        m.load_ops();
        if (!active) {      // (keep the cursor of the constant images in step with the other waves)
            for (int q = 0; q < Nc; ++q) {
                (void)p.next_c(q);
                (void)p.next_c(Nc + q);
            }
            continue;
        }
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
            const double PQ = wave_sum2((B + C) * wgt, (A + D) * wgt);      // rows 0, 1: P;  rows 2, 3: Q
            double* tr = trw + (size_t)n * (Nc * JQ_NTR) + q * JQ_NTR;
            if (lane_ == 0) {
                tr[0] = 0.0;
                tr[1] = 0.0;
                tr[2] = 0.0;
                tr[3] = -0.25 * PQ;
            }
            if (lane_ == 32) tr[4] = 0.25 * PQ;
        }
    }
    p.drain();
    if (!active) return;
    a_store(u, st, lane);
    a_store(v, st + KT * 64, lane);
    a_store(lr, st + 2 * KT * 64, lane);
    a_store(li, st + 3 * KT * 64, lane);
}
