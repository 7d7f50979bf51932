// jq_quad_split_kernels.h -- the backward sweep of the quad-layout kernels (jq_kernels.h, JQ_BW_T4Q) with the two chains of a column
// quad on TWO waves, one time step apart: mid-size ensembles (one column quad per SIMD: cnot3 x 513 .. 1 024 samples).
//
// k_backward<NT, 7, 1> gives a column quad ONE wave that re-integrates the state (20 products per step at m = 6), runs the adjoint
// step (20) and forms the trace products (12).  With at most one quad per SIMD that wave is alone there and issues an instruction
// every 7.6 cycles against the pipe's 4.6 (dependent fp64 chains).  The coupling between the chains is one-way:
// the adjoint step of time step n reads vr(t_n+1), vi05 and vr(t_n) of the state step -- and vr(t_n+1) is the previous step's
// vr(t_n).  So a workgroup of QW quads runs 2 QW waves:
//   waves 0 .. QW-1      state re-integration of step k in super-step k: the forward step (sv_state + use 6); stores vi05, vr(t_n)
//                        of the step (12 doubles per lane) into the hand-off buffer, slot k & 1
//   waves QW .. 2QW-1    adjoint step + trace products of step k - 1 in super-step k: loads the two arrays from slot (k - 1) & 1
// behind ONE workgroup barrier per super-step (the barrier of the window staging; nsteps + 1 super-steps per chunk).  Wave w and
// wave w + QW land on the same SIMD (QW = 4: two waves per SIMD, each a different chain of the same quad).
//
// Hand-off: global memory (L2), not LDS -- the window ring must keep the time points of step k - 1 next to those of step k (seven
// slots instead of five: 84 KB at cnot3) and 49 KB of double-buffered hand-off do not fit next to it.  Producer and consumer are waves
// of ONE workgroup, i.e. of one CU and one vector L1: plain stores, acknowledged (s_waitcnt vmcnt(0)) in front of the workgroup
// barrier, and plain loads behind it are the workgroup-scope release / acquire of the memory model -- no agent-scope traffic, no
// placement assumption, nothing another workgroup ever reads.  The loads are issued at the top of the adjoint step; their first use
// is eight products later.
//
// Every chain executes the operations of k_backward<NT, 7, 1> in the same order (same functions, same fused passes); the trace sums
// of a workgroup are added over its QW adjoint waves in wave order like there.
#pragma once
#include "jq_kernels.h"

// JQ_QS_TPS = 7 (jq_kernels.h): window ring of the split kernel -- time points 2k-2 .. 2k+2 in use, 2k+3, 2k+4 streaming in
// JQ_QS_ARRAYS = 2: arrays handed over per step -- vi05, vr(t_n)

// doubles of hand-off buffer per column quad (host: allocation)
__host__ __device__ constexpr size_t jq_qs_quad_doubles(int NT) { return (size_t)2 * JQ_QS_ARRAYS * NT * 64; }

// NP (1 or 2) products D_k = C_k + M_k x in ONE pass over the blocks like mm_t4q_multi, with a functor called once per 16-row block with
// the block of x, its two lane-shifted copies and its neighbouring blocks: ride(mt, xc, su, sd, xold, xn).  Partial products with
// single-subsystem operators (the trace products of adjoint_grad_calc!) and their dot products ride along in the pass of a full
// product with the same right-hand side -- no shifts, no passes and no result arrays of their own, and independent instructions in
// a chain that is bound by the latency of its dependent ones.  x must not alias a D_k.
template <int NT, int NP, bool Z0, bool Z1, typename RIDE>
__device__ __forceinline__ void mm_t4q_ride(Arr<NT>& D0, const Arr<NT>& C0, const double* m0, Arr<NT>& D1, const Arr<NT>& C1, const double* m1,
                                            const Arr<NT>& x, RIDE ride)
{
    static_assert(NP == 1 || NP == 2, "");
    const int lane = threadIdx.x & 63;
    const double* ma[2] = {t4q_a(m0, lane), t4q_a(m1, lane)};
    const d4* cf[2] = {t4q_c<NT>(m0, lane), t4q_c<NT>(m1, lane)};
    double xold = 0.0;
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
        double a[2];
        d4 c[2];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            a[k] = ma[k][mt * 64];
            c[k] = t4q_cload(cf[k], mt);
        }
        const double xc = x.t[mt][0], xn = x.t[mt + 1 < NT ? mt + 1 : mt][0];
        const double su = row_shift4<0x114>(xc), sd = row_shift4<0x104>(xc);
        double acc[2] = {Z0 ? 0.0 : C0.t[mt][0], (NP > 1 && !Z1) ? C1.t[mt][0] : 0.0};
#pragma unroll
        for (int k = 0; k < NP; ++k) acc[k] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[k], xc, acc[k], 0, 0, 0);
        ride(mt, xc, su, sd, xold, xn);
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
        xold = xc;
        D0.t[mt][0] = acc[0];
        if constexpr (NP > 1) D1.t[mt][0] = acc[1];
        __builtin_amdgcn_sched_barrier(0);      // (fence per block, see mm_t4q_multi)
    }
}
// The three single-subsystem operators of a set (Hsym_q or Hanti_q, q = 0, 1, 2; control q acts on subsystem q only): this lane's
// operands in the resident constant images -- operator 0: diagonal 4 x 4 blocks (the MFMA's A operand), operator 1: couplings of the
// neighbouring 4-row groups (lane-shift terms), operator 2: couplings of the neighbouring 16-row blocks.  apply() gives block mt of
// y_q = (operator q) x, operation for operation what mm_t4q<.., JQ_T4_DIAG / JQ_T4_RTERMS / JQ_T4_MTERMS> computes.
template <int NT>
struct OrdOps {
    const double* a0;
    const d4 *c1, *c2;
    __device__ __forceinline__ void init(const double* M0, const double* M1, const double* M2, int lane)
    {
        a0 = t4q_a(M0, lane);
        c1 = t4q_c<NT>(M1, lane);
        c2 = t4q_c<NT>(M2, lane);
    }
    __device__ __forceinline__ void apply(int mt, double xc, double su, double sd, double xold, double xn, double& y0, double& y1, double& y2) const
    {
        y0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a0[mt * 64], xc, 0.0, 0, 0, 0);
        const d4 k1 = t4q_cload(c1, mt);
        y1 = fma(k1[1], sd, k1[0] * su);
        const d4 k2 = t4q_cload(c2, mt);
        y2 = 0.0;
        if (mt > 0) y2 = fma(k2[2], xold, y2);
        if (mt + 1 < NT) y2 = fma(k2[3], xn, y2);
    }
};

// grid = ceil(4 nslabs / QW), block = 128 QW threads.  a.park: the hand-off buffer, [quad][parity][array][block][64].
// Dynamic LDS: [ring of JQ_QS_TPS time points | constant trace images | tables wd, ws | trace records 2 x QW x 8 Nc].
// ORD: control q acts on subsystem q only (compile-time trace modes, Hsym_1 lambda_i rides along with K05 lambda_i; see k_backward).
// RIDE (with ORD and exactly three controls): ALL twelve trace products ride along in the passes of the adjoint step that shift the same
// right-hand side (mm_t4q_ride): Hanti_q X with K0 X / K1 X, Hanti_q (-lambda_i) with S05 (-lambda_i), Hsym_q (-lambda_i new) and
// Hanti_q (-lambda_i new) with K05 (-lambda_i new), Hsym_q X with S1 X; tr5 is then the sum of two dot products (with -lambda_i old and new) instead of
// one dot product with their sum -- the only difference in rounding to the other variants (1e-16).
template <int NT, bool ORD, int QW, bool RIDE = false>
__global__ __launch_bounds__(128 * QW, 1) void k_backward_qsplit(PropArgs a)
{
    static_assert(QW == 4 || QW == 2 || QW == 1, "quads per workgroup");
    static_assert(!RIDE || ORD, "RIDE: a variant of the ORD kernel");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KT = 4 * NT;
    constexpr int BW = JQ_BW_T4Q;
    constexpr int NWAVES = 2 * QW;
    const int lane_ = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool adj = wave >= QW;
    const int qw = adj ? wave - QW : wave;
    const int quad = (int)blockIdx.x * QW + qw;      // quad slot of the batch: four per slab
    const int slab = quad >> 2;
    const int col = 4 * (quad & 3) + (lane_ & 3);
    const int lane = ((lane_ >> 2) & 3) * 64 + 16 * (lane_ >> 4) + col;      // this lane's offset in a block of the slab image
    const int g = 4 * (lane_ >> 4) + ((lane_ >> 2) & 3);                      // ... and of the row tables
    const bool active = slab < a.nslabs;
    const int Nc = a.Ncoupled;
    const int nst = a.nsteps_chunk;
    const bool cslot = ((lane_ >> 2) & 3) == 0;
    const int clane = 16 * (lane_ >> 4) + col;

    double* tab = (double*)(smem + a.lds_tab_off);
    const double* wd = tab;
    const double* ws = tab + 16 * NT;
    double* rec = tab + 32 * NT;      // [2][QW][8 Nc]: wave sums of a step, see k_backward
    const int rslots = 8 * Nc;
    const int ntr = Nc * JQ_NTR;
    const double cfw = a.forced ? 0.5 * a.h * a.tinv : 0.0;
    for (int i = threadIdx.x; i < 32 * NT; i += blockDim.x)
        tab[(i & ~15) + 4 * (i & 3) + ((i >> 2) & 3)] = (i < 16 * NT) ? cfw * a.tabs[i] : a.tabs[i];   // [block][g][r]
    for (int i = threadIdx.x; i < 2 * QW * rslots; i += blockDim.x) rec[i] = 0.0;   // (inactive waves never write theirs)
    double* st = a.state + (size_t)(active ? slab : 0) * a.state_stride;
    double* hand = a.park + (size_t)quad * jq_qs_quad_doubles(NT) + lane_;
    const double ceps = active ? 0.5 * a.h * a.colinfo[(size_t)slab * 32 + col] : 0.0;

    RingT<true, JQ_QS_TPS> p;
    p.init(smem, a, wave, lane_, NWAVES);
    // The adjoint wave -- the longer chain, 32 of the 52 products of a step -- goes first in the issue arbitration of the SIMD it shares
    // with the state wave of its quad (QW = 4): it then runs almost as if it were alone there, and the state wave, which has slack
    // until the step's barrier, takes the slots that are left.  One s_setprio: cnot3 x 1 024 samples, backward sweep 280.6 -> 250.0 ms
    // (priority 1 or 3 alike).  (The same in k_backward_cq, whose two sets of waves meet at 5 + 2 m barriers per step: + 2 % -- there the
    //  set that is held back is waited for at the next barrier.)
    if (QW == 4 && adj) __builtin_amdgcn_s_setprio(3);

    if (!adj) {
        // ---- state re-integration: the forward step with h < 0 (src/evalobjgrad.jl:879) ----------------------------------------
        auto flush_traces = [&](int k) {
            if (wave == 0 && lane_ < ntr) {
                const int q = lane_ / JQ_NTR, kk = lane_ - q * JQ_NTR;
                const int slot = (kk == 0 ? 0 : kk == 2 ? 2 : 4 * Nc + (kk == 1 ? 0 : kk == 3 ? 2 : 1)) + 4 * q;
                const double* r = rec + (size_t)(k & 1) * QW * rslots + slot;
                double s = r[0];
#pragma unroll
                for (int w = 1; w < QW; ++w) s += r[w * rslots];
                a.traces[((size_t)blockIdx.x * nst + k) * ntr + lane_] = s;
            }
        };
        Arr<NT> ua, va, ub, vb, A, Ya, Yb;
        if (active) {
            a_load(ua, st, lane);
            a_load(va, st + KT * 64, lane);
        } else {
            a_zero(ua);
            a_zero(va);
        }
#define JQ_QS_STATE_STEP(U, V, UN, VN, K)                                                                               \
    {                                                                                                                   \
        p.begin_step(K);                                                                                                \
        if ((K) >= 2) flush_traces((K) - 2);      /* the adjoint waves finished step K - 2 in super-step K - 1 */        \
        sv_state<NT, BW, false, JQ_BWD_FUSE, false>(p, a, active, ceps, ws, g, U, V, UN, VN, A, Ya, Yb);                \
        const double* M6 = p.template next_ks<0, 1>();                                                                  \
        if (active) {                                                                                                   \
            mm_c<NT, BW>(VN, VN, M6, UN);                                                                               \
            if (a.use_shift) a_axpy_rows(VN, ceps, ws, g, UN);                                                          \
            double* hs = hand + (size_t)((K) & 1) * JQ_QS_ARRAYS * NT * 64;                                             \
            _Pragma("unroll") for (int i = 0; i < NT; ++i)                                                              \
            {                                                                                                           \
                hs[i * 64] = V.t[i][0];      /* vi05 */                                                                 \
                hs[(NT + i) * 64] = UN.t[i][0];      /* vr(t_n) */                                                      \
            }                                                                                                           \
        }                                                                                                               \
    }
        int k = 0;
        for (; k + 1 < nst; k += 2) {
            JQ_QS_STATE_STEP(ua, va, ub, vb, k)
            JQ_QS_STATE_STEP(ub, vb, ua, va, k + 1)
        }
        if (k < nst) {
            JQ_QS_STATE_STEP(ua, va, ub, vb, k)
            if (active) {
                ua = ub;
                va = vb;
            }
        }
#undef JQ_QS_STATE_STEP
        // super-step nst: the adjoint waves run their last step
        p.begin_step(nst);
        if (nst >= 2) flush_traces(nst - 2);
        p.drain();
        flush_traces(nst - 1);
        if (active) {
            a_store(ua, st, lane);
            a_store(va, st + KT * 64, lane);
        }
        return;
    }

    // ---- adjoint step with forcing / step_no_forcing! (src/StormerVerlet.jl:255-451) and the trace scalars of adjoint_grad_calc!
    // (src/evalobjgrad.jl:2567-2619): the adjoint part of k_backward, one step behind the state waves ----------------------------------
    //   u  : vr before the state step            un : vr after it           v : vi05
    //   mu : lambda_r -> X = lambda_r^{1/2}       nb : -lambda_i (old) -> -(li0 + li)      L : -lambda_i (new)
    //   vN : scratch Q, G -> lambda_r (new)       Ya, Yb : Horner scratch, trace products
    Arr<NT> u, v, un, mu, nb, L, vN, Ya, Yb;
    double wgt = 0.0;
    double carry[JQ_MAXNC];
#pragma unroll
    for (int q = 0; q < JQ_MAXNC; ++q) carry[q] = 0.0;
    if (active) {
        a_load(u, st, lane);
        a_load(mu, st + 2 * KT * 64, lane);
        a_load(nb, st + 3 * KT * 64, lane);
        wgt = a.colinfo[(size_t)slab * 32 + 16 + col];
#pragma unroll
        for (int q = 0; q < JQ_MAXNC; ++q)
            if (q < Nc) carry[q] = cslot ? st[(JQ_STATE_ARRAYS * KT + q) * 64 + clane] : 0.0;
    } else {
        a_zero(u);
        a_zero(mu);
        a_zero(nb);
    }
    a_zero(v);
    a_zero(un);
    // (its window lags the state waves' by one step: time point 2 (k - 1) of super-step k)
    p.s0 = JQ_QS_TPS - 2;
    p.set_window();
    if (a.first_chunk) {
        // carry_q = tr(vr' Hsym_q lambdai) at t = T (see k_backward)
#pragma unroll
        for (int q = 0; q < JQ_MAXNC; ++q)
            if (q < Nc) {
                const double* M = p.next_c(q);  // Hsym_q
                if (active) {
                    mm_z_bw<NT, BW>(Ya, M, nb, a.bw_trace[q]);
                    carry[q] = -a_dot(u, Ya);
                }
            }
    }
    p.begin_step(0);      // (super-step 0: the state waves' first step)
    if constexpr (RIDE) {
        OrdOps<NT> hs, ha;      // (the constant images are resident: the operand pointers do not change)
        hs.init(p.next_c(0), p.next_c(1), p.next_c(2), lane_);
        ha.init(p.next_c(3), p.next_c(4), p.next_c(5), lane_);
        for (int k = 1; k <= nst; ++k) {
            const int n = k - 1;
            p.begin_step(k);
            if (active) {
                const double* hsrc = hand + (size_t)(n & 1) * JQ_QS_ARRAYS * NT * 64;
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    v.t[i][0] = hsrc[i * 64];
                    un.t[i][0] = hsrc[(NT + i) * 64];
                }
            }
            double t1[3] = {0.0, 0.0, 0.0}, t3[3] = {0.0, 0.0, 0.0}, t2[3] = {0.0, 0.0, 0.0}, s4[3] = {0.0, 0.0, 0.0}, t5a[3] = {0.0, 0.0, 0.0},
                   t5b[3] = {0.0, 0.0, 0.0};
            // use 6 (adjoint part): L = c K05 nb ; rides (EARLY): Hanti_q nb (the old part of tr5).
            // Measured at cnot3 (backward sweep, ms): two quads per workgroup (the adjoint wave ALONE on its SIMD, bound by the latency
            // of its dependent chain: every independent instruction is free) 210.5 without rides, 198.3 with this ride here, 205.6 with
            // it in use 10; four quads per workgroup (two waves per SIMD: the pipe is full, rides buy nothing) 281.9 / 312.3 / 280.0 --
            // there the ride needs vi05 in the first pass of the step, a moment after it was asked for, and the wave that stalls on the
            // L2 round trip holds up a SIMD that has no idle slots to lose.
#ifndef JQ_QS_EARLY4
#define JQ_QS_EARLY4 0
#endif
            constexpr bool EARLY = (QW <= 2) || JQ_QS_EARLY4;
            const double* M = p.template next_ks<0, 1>();
            if (active) {
                if constexpr (EARLY)
                    mm_t4q_ride<NT, 1, true, true>(L, L, M, L, L, M, nb, [&](int mt, double xc, double su, double sd, double xo, double xn) {
                        double y0, y1, y2;
                        ha.apply(mt, xc, su, sd, xo, xn, y0, y1, y2);
                        t5a[0] = fma(v.t[mt][0], y0, t5a[0]), t5a[1] = fma(v.t[mt][0], y1, t5a[1]), t5a[2] = fma(v.t[mt][0], y2, t5a[2]);
                    });
                else
                    mm_z<NT, BW>(L, M, nb);
                if (a.use_shift) a_axpy_rows(L, ceps, ws, g, nb);
            }
            // use 7: S0 -- L = c (S0 mu - K05 li + hr0) ; X = mu + sum_j S^j L
            M = p.template next_ks<1, 0>();
            if (active) {
                mm_c<NT, BW>(L, L, M, mu);
                a_axpy_rows1<NT, false>(L, wd, g, u);
                a_add(mu, L);
                horner_add<NT, BW, false>(mu, mu, L, M, a.m, Ya, Yb, a.jacobi_tol2, a.N);
            }
            // uses 8 and 9: L = -c K0 X ; vN = -c K1 X ; rides: Hanti_q X (tr1, tr3) -- Hsym_q X rides in use 12 (fewer live values there:
            // with both sets in this pass the four-quad variant spilled 80 - 136 B per lane and was 11 % SLOWER than without any rides)
            M = p.template next_ks<0, 0>();
            const double* M9 = p.template next_ks<0, 2>();
            if (active) {
                mm_t4q_ride<NT, 2, true, true>(L, L, M, vN, vN, M9, mu, [&](int mt, double xc, double su, double sd, double xo, double xn) {
                    double y0, y1, y2;
                    ha.apply(mt, xc, su, sd, xo, xn, y0, y1, y2);
                    t1[0] = fma(u.t[mt][0], y0, t1[0]), t1[1] = fma(u.t[mt][0], y1, t1[1]), t1[2] = fma(u.t[mt][0], y2, t1[2]);
                    t3[0] = fma(un.t[mt][0], y0, t3[0]), t3[1] = fma(un.t[mt][0], y1, t3[1]), t3[2] = fma(un.t[mt][0], y2, t3[2]);
                });
                if (a.use_shift) {
                    a_axpy_rows(L, -ceps, ws, g, mu);
                    a_axpy_rows(vN, -ceps, ws, g, mu);
                }
                // (the early group's wave sums here: tr1, tr3 die before the second Neumann series)
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const double te = wave_sum4(t1[q] * wgt, t3[q] * wgt, 0.0, 0.0);   // rows 0, 2: t1, t3
                    if ((lane_ & 15) == 0) rec[((size_t)(n & 1) * QW + qw) * rslots + 4 * q + (lane_ >> 4)] = te;
                }
            }
            // use 10: S05 -- as in the generic path ; rides of its first product (S05 nb): Hanti_q nb (the old part of tr5)
            M = p.template next_ks<1, 1>();
            if (active) {
                if constexpr (EARLY)
                    mm_z<NT, BW>(Ya, M, nb);
                else
                    mm_t4q_ride<NT, 1, true, true>(Ya, Ya, M, Ya, Ya, M, nb, [&](int mt, double xc, double su, double sd, double xo, double xn) {
                        double y0, y1, y2;
                        ha.apply(mt, xc, su, sd, xo, xn, y0, y1, y2);
                        t5a[0] = fma(v.t[mt][0], y0, t5a[0]), t5a[1] = fma(v.t[mt][0], y1, t5a[1]), t5a[2] = fma(v.t[mt][0], y2, t5a[2]);
                    });
                a_axpy_rows1<NT, true>(Ya, wd, g, v);
                a_add(L, Ya);
                a_add(vN, Ya);
                mm_c<NT, BW>(vN, vN, M, L);
                a_add(L, nb);
                a_add(L, vN);
                horner_add<NT, BW, false>(L, L, vN, M, a.m, Ya, Yb, a.jacobi_tol2, a.N);
                a_add(nb, L);
            }
            // use 11: Kp05 -- vN = X + c K05 nb_new ; rides: Hsym_q L (the new part of tr4), Hanti_q L (the new part of tr5)
            M = p.template next_ks<0, 1>();
            if (active) {
                mm_t4q_ride<NT, 1, false, true>(vN, mu, M, vN, mu, M, L, [&](int mt, double xc, double su, double sd, double xo, double xn) {
                    double y0, y1, y2;
                    hs.apply(mt, xc, su, sd, xo, xn, y0, y1, y2);
                    s4[0] = fma(un.t[mt][0], y0, s4[0]), s4[1] = fma(un.t[mt][0], y1, s4[1]), s4[2] = fma(un.t[mt][0], y2, s4[2]);
                    ha.apply(mt, xc, su, sd, xo, xn, y0, y1, y2);
                    t5b[0] = fma(v.t[mt][0], y0, t5b[0]), t5b[1] = fma(v.t[mt][0], y1, t5b[1]), t5b[2] = fma(v.t[mt][0], y2, t5b[2]);
                });
                if (a.use_shift) a_axpy_rows(vN, ceps, ws, g, L);
            }
            // use 12: S1 -- lambda_r_new = X + c (S1 X - K05 li_new + hr1) ; rides: Hsym_q X (tr2)
            M = p.template next_ks<1, 2>();
            if (active) {
                mm_t4q_ride<NT, 1, false, true>(vN, vN, M, vN, vN, M, mu, [&](int mt, double xc, double su, double sd, double xo, double xn) {
                    double y0, y1, y2;
                    hs.apply(mt, xc, su, sd, xo, xn, y0, y1, y2);
                    t2[0] = fma(v.t[mt][0], y0, t2[0]), t2[1] = fma(v.t[mt][0], y1, t2[1]), t2[2] = fma(v.t[mt][0], y2, t2[2]);
                });
                a_axpy_rows1<NT, false>(vN, wd, g, un);
                // the late group's wave sums (rows 0, 2, 1 = t2, t4, t5)
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const double p4 = -s4[q];
                    const double t4 = p4 + carry[q];
                    carry[q] = p4;
                    const double t5 = -(t5a[q] + t5b[q]);
                    const double tl = wave_sum4(t2[q] * wgt, t4 * wgt, t5 * wgt, 0.0);
                    if ((lane_ & 15) == 0) rec[((size_t)(n & 1) * QW + qw) * rslots + 4 * (3 + q) + (lane_ >> 4)] = tl;
                }
                u = un;
                mu = vN;
                nb = L;
            }
        }
    } else
    for (int k = 1; k <= nst; ++k) {
        const int n = k - 1;      // the step of this super-step
        p.begin_step(k);
        if (active) {
            const double* hs = hand + (size_t)(n & 1) * JQ_QS_ARRAYS * NT * 64;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                v.t[i][0] = hs[i * 64];
                un.t[i][0] = hs[(NT + i) * 64];
            }
        }
        // use 6 (adjoint part): L = c K05 nb (= -c K05 lambda_i)
        const double* M = p.template next_ks<0, 1>();
        if (active) {
            mm_z<NT, BW>(L, M, nb);
            if (a.use_shift) a_axpy_rows(L, ceps, ws, g, nb);
        }
        // use 7: S0 -- L = c (S0 mu - K05 li + hr0) ; X = mu + sum_j S^j L   (in place: mu becomes X)
        M = p.template next_ks<1, 0>();
        if (active) {
            mm_c<NT, BW>(L, L, M, mu);
            a_axpy_rows1<NT, false>(L, wd, g, u);  // u holds vr before the state step (:862)
            a_add(mu, L);
            horner_add<NT, BW, false>(mu, mu, L, M, a.m, Ya, Yb, a.jacobi_tol2, a.N);
        }
        // early traces with X: tr1 = tr(vr0' Hanti_q X), tr3 = tr(vr' Hanti_q X)
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
                const double ts = wave_sum4(a_dot(u, Ya) * wgt, a_dot(un, Ya) * wgt, 0.0, 0.0);   // rows 0, 2: t1, t3
                if ((lane_ & 15) == 0) rec[((size_t)(n & 1) * QW + qw) * rslots + 4 * q + (lane_ >> 4)] = ts;
            }
        }
        // uses 8 and 9 in one pass (they share X): L = -c K0 X ; vN(scratch Q) = -c K1 X
        M = p.template next_ks<0, 0>();
        const double* M9 = p.template next_ks<0, 2>();
        if (active) {
            mm_t4q2<NT, true, true>(L, L, M, vN, vN, M9, mu);
            if (a.use_shift) {
                a_axpy_rows(L, -ceps, ws, g, mu);
                a_axpy_rows(vN, -ceps, ws, g, mu);
            }
        }
        // use 10: S05 -- L = -c l2 = -c (K0 X + S05 li + hi0) ; Q = -c (S05 (li + c l2) + K1 X + hi1) ;
        //               nb_new = nb + L + sum_j S^j Q          (li_new = li + c (l2 + l1))
        M = p.template next_ks<1, 1>();
        if (active) {
            mm_z<NT, BW>(Ya, M, nb);
            a_axpy_rows1<NT, true>(Ya, wd, g, v);  // v holds vi05;  Ya = c (-S05 li - hi0)
            a_add(L, Ya);
            a_add(vN, Ya);
            mm_c<NT, BW>(vN, vN, M, L);       // vN = Q
            a_add(L, nb);
            a_add(L, vN);                     // L = nb + L + Q
            horner_add<NT, BW, false>(L, L, vN, M, a.m, Ya, Yb, a.jacobi_tol2, a.N);  // L = nb_new
            a_add(nb, L);                     // nb = nb_old + nb_new = -(li0 + li)
        }
        // use 11: Kp05 -- vN(scratch G) = X + c K05 nb_new (= lambda_r^{1/2} - c K05 li_new)
        M = p.template next_ks<0, 1>();
        if (active) {
            if constexpr (ORD) {
                // ... and Hsym_1 lambda_i_new (the new part of tr4 of control 1) from the same shifted copies of L = -lambda_i_new
                const d4* cfs = t4q_c<NT>(p.next_c(1), lane_);
                mm_t4q_multi<NT, 1, false, true, true, 0>(vN, mu, M, vN, mu, M, vN, mu, M, L, 0.0, 0.0, 0.0, nullptr,
                                                          [&](int mt, double su, double sd) {
                                                              const d4 cs = t4q_cload(cfs, mt);
                                                              o_p4 = fma(-un.t[mt][0], fma(cs[1], sd, cs[0] * su), o_p4);
                                                          });
            } else {
                mm_c<NT, BW>(vN, mu, M, L);
            }
            if (a.use_shift) a_axpy_rows(vN, ceps, ws, g, L);
        }
        // use 12: S1 -- lambda_r_new = X + c (S1 X - K05 li_new + hr1)
        M = p.template next_ks<1, 2>();
        if (active) {
            mm_c<NT, BW>(vN, vN, M, mu);
            a_axpy_rows1<NT, false>(vN, wd, g, un);
        }
        // ---- late traces (adjoint_grad_calc!, :2581-2618), per control q, weighted by the sample weight:
        //   tr5 = tr(vi05' Hanti (li0+li))   tr2 = tr(vi05' Hsym X)   tr4 = tr(vr' Hsym li) + tr(vr0' Hsym li0)
        // here: un = vr, v = vi05, mu = X, nb = -(li0+li), L = -li
#pragma unroll
        for (int q = 0; q < JQ_MAXNC; ++q)
            if (q < Nc) {
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
                    t4 = p4 + carry[q];
                    carry[q] = p4;
                    const double ts = wave_sum4(t2 * wgt, t4 * wgt, t5 * wgt, 0.0);   // rows 0, 2, 1: t2, t4, t5
                    if ((lane_ & 15) == 0) rec[((size_t)(n & 1) * QW + qw) * rslots + 4 * (Nc + q) + (lane_ >> 4)] = ts;
                }
            }
        // ---- roles for the next step: u <- un, mu <- vN (new lambda_r), nb <- L
        if (active) {
            u = un;
            mu = vN;
            nb = L;
        }
    }
    p.drain();
    if (active) {
        a_store(mu, st + 2 * KT * 64, lane);
        a_store(nb, st + 3 * KT * 64, lane);
#pragma unroll
        for (int q = 0; q < JQ_MAXNC; ++q)
            if (q < Nc) {
                const double cv = row_ror_add<8>(row_ror_add<4>(carry[q]));   // only ever used summed over the rows of a column
                if (cslot) st[(JQ_STATE_ARRAYS * KT + q) * 64 + clane] = cv;
            }
    }
}
