// jq_cq_imr_kernels.h -- IMPLICIT MIDPOINT propagators (traceobjgrad for Working_Arrays_M, src/evalobjgrad.jl:1042-1481;
// m_step!, src/ImplicitMidpoint.jl:120-227; jacobi_midpoint, src/linear_solvers.jl:156-270) as "cooperative quad" kernels: the
// LATENCY path of the JQ_BW_T4 structure for N = 4 (cnot3: one evaluation, small ensembles).
//
// The quad-layout kernels of jq_quad_imr_kernels.h give one wave the four columns of an evaluation and ALL 16-row blocks: every
// fixed-point iteration x <- rhs + B x, B = h/2 [S -K; K S], is four products of NT blocks each on a wave that issues one
// instruction every ~10 cycles (cnot3: ~2 100 cycles per iteration, 0.77 s per evaluation).  Here, as in jq_cq_kernels.h, wave mt
// of a workgroup owns block mt of every array (one register per array), publishes its block of (x_u, x_v) in LDS (channels 0 and
// 1 of the exchange image), passes ONE workgroup barrier per iteration and reads the two neighbouring blocks.  The stopping rule of
// the reference -- |x_it - x_{it+1}|^2 < tol^2 for the u and the v part, summed over the evaluation -- needs a sum over all waves:
// two extra REDUCER waves (one for the u part, one for the v part) read the published blocks of x_{it+1} behind the same barrier,
// keep them for the next round, and leave their halves of the decision in LDS one barrier later, while the block waves go on
// speculatively: when they learn that x_it was the result they have published x_{it+1} and x_{it+2} for nothing.  Every wave sees
// the same decisions.  Publications per implicit step: 3 + (number of iterations).
// Measured at cnot3 (round 4, HISTORY.md): 785 cycles per publication round, 5.7 iterations per step; one evaluation 0.77 s on
// the quad-layout kernels, 0.37 s with the block waves doing the reduction themselves (80 instructions per wave and round), 0.32 s
// with one reducer wave (the block waves still folding their squares), 0.30 s as described (47 instructions per block wave).  The
// round is bound by the two SIMDs that hold two block waves each (six blocks on four SIMDs): 2 x 47 instructions at the ~8 cycles
// per instruction of two waves per SIMD (probes/dp_rate_probe.hip) = 750 cycles.
// The parity of a publication is a run-time value here (the iteration count is data dependent).
//
// Same operator stream (K, S of the midpoint = time point 2n+1 of step n; window staging of WinRing: the DMA of the time points
// 2n+5, 2n+6 is issued behind the first barrier of step n and drained in front of the first barrier of step n+1), state file,
// terminal kernels and trace slots as the quad-layout kernels; one trace record row per wave.
#pragma once
#include "jq_cq_split_kernels.h"

// (DN: the dense policy of CoopQ -- two 16-row blocks without the structure, round 6)
template <int NT, bool DN = false>
struct CqImr {
    typedef typename CoopQ<NT, DN>::Op Op;
    typedef typename CoopQ<NT, DN>::Sh Sh;
    typedef typename CoopQ<NT, DN>::Nb Nb;
    static constexpr int CHS = CoopQ<NT, DN>::CHS, PAR = CoopQ<NT, DN>::PAR;
    CoopQ<NT, DN>* c;
    jq_lds_double* x0;      // front pad block of channel 0 in parity 0, this lane (block w of a channel: + (w + 1) * 64)
    Op K, S;                // my block of the midpoint operators of this step (pre-scaled by h/2)
    double cw;              // h/2 * eps * ws[row] of this lane
    double cwa;             // the same on the diagonal of the MFMA's A operand of K (lanes with k == i, zero elsewhere): the four
                            // columns of the quad are ONE evaluation (N = 4), so the ensemble shift rides on K's 4 x 4 blocks
                            // (fold_shift) instead of costing two FMAs per application
    double tol2;
    int max_iter, par;
    bool use_shift;
    volatile __attribute__((address_space(3))) int* flags;      // decisions of the reducer waves, [2 slots][2 parts]

    struct Acc {
        double au, kv, av, sv;  // q_u = au - kv ; q_v = av + sv  (four independent chains of one MFMA and four FMAs each)
    };
    // the part of  q = rhs + [S -K; K S] p  that needs only my block
    __device__ __forceinline__ Acc own(double ru, double rv, double pu, double pv) const
    {
        const Sh su = c->sh(pu), sv = c->sh(pv);
        Acc r;
        r.au = c->own(ru, S, su);
        r.kv = c->own(0.0, K, sv);
        r.av = c->own(rv, K, su);
        r.sv = c->own(0.0, S, sv);
        // (the diagonal shift of K, src/ipopt_interface.jl:41-44, is part of K.a: fold_shift)
        return r;
    }
    __device__ __forceinline__ void fold_shift()      // (call after loading K; cwa = 0 without an ensemble shift)
    {
        if constexpr (DN) K.ao[0] += cwa;      // (rotation 0 of the own tile: lane 16 k + 4 b + i holds M[4 b + i][4 b + k] -- the same diagonal lanes)
        else K.a += cwa;
    }
    // ... and the (i, i+-16) couplings with the neighbours' blocks of the publication at LDS offset po
    __device__ __forceinline__ void nbr(Acc& r, int po) const
    {
        const Nb nu = c->nbs_at(po), nv = c->nbs_at(po + CHS);
        r.au = c->nbr(r.au, S, nu);
        r.kv = c->nbr(r.kv, K, nv);
        r.av = c->nbr(r.av, K, nu);
        r.sv = c->nbr(r.sv, S, nv);
    }
    __device__ __forceinline__ void post(int po, double xu, double xv) const
    {
        c->xb[po + 64] = xu;
        c->xb[po + CHS + 64] = xv;
    }
    // One implicit-midpoint step of (u, v) on the block waves; (fu, fv): forcing already multiplied by h.  FIRST: the first step of
    // a time step (its first barrier drains the DMA issued one time step ago and is followed by the next DMA, see WinRing).
    // The reducer waves turn the publications j-1 and j into their parts of the decision on x_{j-1} between the barriers j and j+1;
    // the block waves read it behind barrier j+1 -- and have meanwhile computed x_{j+1} and published it.
    template <bool FIRST>
    __device__ __forceinline__ void step(double& u, double& v, double fu, double fv)
    {
        // x = (u, v): B x, rhs = (x + f) + B x, x_1 = rhs + B x
        int po = par * PAR;
        post(po, u, v);
        Acc nx = own(0.0, 0.0, u, v);
        c->template sync<FIRST>();
        if (FIRST) {
            c->ring.issue_next();
            c->ring.issue_next();
        }
        nbr(nx, po);
        par ^= 1;
        const double Bu = nx.au - nx.kv, Bv = (nx.av + nx.sv);
        const double rhs_u = (u + fu) + Bu, rhs_v = (v + fv) + Bv;
        double au = rhs_u + Bu, av = rhs_v + Bv;      // x_1
        po = par * PAR;
        post(po, au, av);
        nx = own(rhs_u, rhs_v, au, av);
        c->sync();
        nbr(nx, po);
        par ^= 1;
        double bu = nx.au - nx.kv, bv = (nx.av + nx.sv);         // x_2
        po = par * PAR;
        post(po, bu, bv);
        nx = own(rhs_u, rhs_v, bu, bv);
        c->sync();
        nbr(nx, po);
        par ^= 1;
        double cu = nx.au - nx.kv, cv = (nx.av + nx.sv);         // x_3
        // (au, av) = x_{j-2}, (bu, bv) = x_{j-1}, (cu, cv) = x_j.  The loop is written for three rounds: the new iterate replaces the
        // oldest one and the three register pairs take turns (a rotating copy cost six moves per round on waves whose SIMD is
        // bound by its issue rate)
        int j = 3;
        auto turn = [&](double xu, double xv, double& ou, double& ov) -> bool {
            const int po_ = par * PAR;
            post(po_, xu, xv);                         // (speculative, like x_{j-1}: dropped if x_{j-2} turns out to be the result)
            Acc t = own(rhs_u, rhs_v, xu, xv);
            c->sync();
            par ^= 1;
            const long long f = *(volatile __attribute__((address_space(3))) long long*)(flags + 2 * (j & 1));      // the decision on x_{j-2}
            ++j;
            nbr(t, po_);                               // (the reads go out together with the flags')
            if (__builtin_amdgcn_readfirstlane((int)f & (int)(f >> 32))) return true;
            ou = t.au - t.kv, ov = t.av + t.sv;
            return false;
        };
        for (;;) {
            if (turn(cu, cv, au, av)) {
                u = au, v = av;
                return;
            }
            if (turn(au, av, bu, bv)) {
                u = bu, v = bv;
                return;
            }
            if (turn(bu, bv, cu, cv)) {
                u = cu, v = cv;
                return;
            }
        }
    }
    // The same step on a reducer wave (part 0: the u part = channel 0 of the exchange image, part 1: the v part): it passes the block
    // waves' barriers; behind barrier j it reads the NT blocks of x_j (every block, in the same order), keeps them for the next
    // round, and has the squared distance |x_{j-1} - x_j|^2 of its part summed over the evaluation (wave_sum): its half of the
    // decision on x_{j-1} -- below tol^2, or the iteration cap of jacobi_midpoint reached -- goes to flags[(j - 1) & 1][part].  Like the
    // block waves it learns the whole decision on x_{j-2} behind barrier j.
    __device__ __forceinline__ void reducer_step(int part)
    {
        c->sync();
        c->sync();      // x_1
        double prev[NT];
        {
            const int po = (par ^ 1) * PAR + part * CHS;
#pragma unroll
            for (int w = 0; w < NT; ++w) prev[w] = x0[po + (w + 1) * 64];
        }
        for (int j = 2;; ++j) {
            const int po = par * PAR + part * CHS;
            c->sync();
            par ^= 1;
            if (j >= 3) {
                const long long f = *(volatile __attribute__((address_space(3))) long long*)(flags + 2 * (j & 1));
                if (__builtin_amdgcn_readfirstlane((int)f & (int)(f >> 32))) return;
            }
            double acc = 0.0, acc1 = 0.0;      // (two chains: the reducer's round trip is on the critical path of the forward sweep)
#pragma unroll
            for (int w = 0; w < NT; ++w) {
                const double cur = x0[po + (w + 1) * 64];
                const double d = prev[w] - cur;
                if (w & 1) acc1 = fma(d, d, acc1);
                else acc = fma(d, d, acc);
                prev[w] = cur;
            }
            acc += acc1;
            // (the sum is valid in the lanes 48 .. 63: they write the decision, the others a dummy word behind the decisions)
            const bool keep = (j - 1 >= max_iter) || (wave_sum_hi(acc) < tol2);      // (a NaN never converges, as in the reference)
            flags[((c->lane >= 48) ? 0 : 4) + 2 * ((j - 1) & 1) + part] = keep ? 1 : 0;
        }
    }
};

#define JQ_CQ_IMR_PROLOGUE                                                                                                       \
    extern __shared__ __attribute__((aligned(16))) char smem[];                                                                  \
    constexpr int KT = 4 * NT;                                                                                                   \
    const int lane_ = s.lane_, wave = s.wave;                                                                                    \
    double* tab = (double*)(smem + a.lds_tab_off);                                                                               \
    for (int i = threadIdx.x; i < 32 * NT; i += blockDim.x) tab[(i & ~15) + 4 * (i & 3) + ((i >> 2) & 3)] = a.tabs[i];            \
    CoopQ<NT, DN> c;                                                                                                             \
    c.setup(tab + 32 * NT, wave, lane_);        /* (reducer waves: chain 1, wave 0 / 1 -- they only read the exchange image) */  \
    c.ring.init(smem, a, wave + NT * s.chain, lane_, NT + 2);      /* (barrier inside: tables, zeroed exchange image) */         \
    c.ring.wave = wave, c.ring.nwaves = NT;     /* (from here on the block waves stage) */                                       \
    const double wdr = tab[16 * wave + s.g], wsr = tab[16 * NT + 16 * wave + s.g];                                               \
    double* st = a.state + (size_t)s.slab * a.state_stride;                                                                      \
    CqImr<NT, DN> m;                                                                                                             \
    m.c = &c;                                                                                                                    \
    m.x0 = (jq_lds_double*)(tab + 32 * NT + lane_);                                                                              \
    m.cw = 0.5 * a.h * a.colinfo[(size_t)s.slab * 32 + s.col] * wsr;                                                             \
    m.cwa = (a.use_shift && (lane_ >> 4) == (lane_ & 3))                                                                         \
                ? 0.5 * a.h * a.colinfo[(size_t)s.slab * 32 + s.col] * tab[16 * NT + 16 * wave + 4 * (lane_ & 3) + ((lane_ >> 2) & 3)] : 0.0; \
    m.tol2 = a.jacobi_tol2, m.max_iter = a.m, m.par = 0, m.use_shift = a.use_shift;                                              \
    m.flags = (volatile __attribute__((address_space(3))) int*)(tab + 32 * NT + 2 * CoopQ<NT>::PAR + NT * 64);

// grid = 4 * nslabs (workgroup = column quad qd of slab blockIdx.x / 4 = one evaluation, N = 4), block = 64 * (NT + 2): NT block
// waves and the two reducer waves
template <int NT, bool DN = false>
__global__ __launch_bounds__(64 * NT + 128) void k_forward_cq_imr(PropArgs a)
{
    const CqSetup<NT> s = cq_setup<NT>(a);
    if (!s.active) return;      // (a quad without columns: the whole workgroup leaves before any barrier)
    JQ_CQ_IMR_PROLOGUE
    double* scratch = tab + 32 * NT + 2 * CoopQ<NT>::PAR;
    if (s.chain) {      // reducer waves
        for (int n = 0; n < a.nsteps_chunk; ++n) m.reducer_step(wave);
        __syncthreads();      // (cq_wg_sum of the block waves)
        __syncthreads();
        return;
    }
    double u = st[s.foff], v = st[(size_t)KT * 64 + s.foff];
    const bool slot0 = wave == 0 && ((lane_ >> 2) & 3) == 0;      // the lanes that carry per-column partials between chunks
    const size_t cslot = 16 * (lane_ >> 4) + s.col;
    double leak = slot0 ? st[(size_t)(JQ_STATE_ARRAYS * KT + JQ_MAXNC) * 64 + cslot] : 0.0;
    for (int n = 0; n < a.nsteps_chunk; ++n) {
        m.K = c.load(c.ring.template ks<0, 1>());
        m.fold_shift();
        m.S = c.load(c.ring.template ks<1, 1>());
        double su = u, sv = v;
        m.template step<true>(u, v, 0.0, 0.0);
        c.ring.advance();
        su += u;
        sv += v;
        leak += wdr * (su * su) + wdr * (sv * sv);      // penal_m (src/evalobjgrad.jl:1214, :2158-2166)
        if (a.hist_r) {
            const int scol = a.parts > 1 ? 16 * s.slab + s.col : s.col;
            const int row = 16 * wave + 4 * ((lane_ >> 2) & 3) + (lane_ >> 4);
            if (s.slab < a.parts && scol < a.N && row < a.Ntot) {
                const size_t off = (size_t)(a.step0 + n + 1) * a.Ntot * a.N + (size_t)scol * a.Ntot + row;
                a.hist_r[off] = u;
                a.hist_i[off] = -v;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    st[s.foff] = u;
    st[(size_t)KT * 64 + s.foff] = v;
    const double tot = cq_wg_sum(leak, scratch, wave, lane_, NT);
    if (slot0) st[(size_t)(JQ_STATE_ARRAYS * KT + JQ_MAXNC) * 64 + cslot] = tot;
}

// Backward sweep (src/evalobjgrad.jl:1290-1336): state re-integration with h < 0, adjoint m_step! with forcing
// -W (v + v_s) / T and the two gradient scalars of adjoint_grad_calc_m per control (:2660-2702) in the slots of the midpoint weights
// of k_gradacc (jq_rowlane_imr_kernels.h): tr[3] = -(B + C)/4, tr[4] = (A + D)/4.  One trace record row per wave (its block's share).
template <int NT, bool DN = false>
__global__ __launch_bounds__(64 * NT + 128) void k_backward_cq_imr(PropArgs a)
{
    typedef typename CoopQ<NT, DN>::Sh Sh;
    typedef typename CoopQ<NT, DN>::Nb Nb;
    const CqSetup<NT> s = cq_setup<NT>(a);
    const int Nc = a.Ncoupled, ntr = Nc * JQ_NTR;
    const size_t trow = ((size_t)s.slab * a.qps + s.qd) * NT;      // first of my workgroup's NT record rows
    if (!s.active) {
        if (s.qd < a.qps)
            for (size_t k = threadIdx.x; k < (size_t)NT * a.nsteps_chunk * ntr; k += blockDim.x) a.traces[trow * a.nsteps_chunk * ntr + k] = 0.0;
        return;
    }
    JQ_CQ_IMR_PROLOGUE
    constexpr int CHS = CoopQ<NT>::CHS, PAR = CoopQ<NT>::PAR;
    if (s.chain) {      // reducer waves: state step, adjoint step, the publication of the trace products
        for (int n = 0; n < a.nsteps_chunk; ++n) {
            m.reducer_step(wave);
            m.reducer_step(wave);
            c.sync();
            m.par ^= 1;
        }
        return;
    }
    double u = st[s.foff], v = st[(size_t)KT * 64 + s.foff];
    double lr = st[(size_t)2 * KT * 64 + s.foff], li = st[(size_t)3 * KT * 64 + s.foff];
    const double wgt = a.colinfo[(size_t)s.slab * 32 + 16 + s.col];
    const double cfw = a.forced ? -a.h * a.tinv * wdr : 0.0;      // h * (-tinv * W): W applied row-wise
    double* trw = a.traces + ((trow + wave) * a.nsteps_chunk) * ntr;
    for (int n = 0; n < a.nsteps_chunk; ++n) {
        m.K = c.load(c.ring.template ks<0, 1>());
        m.fold_shift();
        m.S = c.load(c.ring.template ks<1, 1>());
        double su = u, sv = v, smu = lr, snu = li;
        m.template step<true>(u, v, 0.0, 0.0);
        c.ring.advance();
        su += u;
        sv += v;
        m.template step<false>(lr, li, cfw * su, cfw * sv);
        smu += lr;
        snu += li;
        // trace products with the constant images (Hsym_q: image q, Hanti_q: image Nc + q): one publication of (su, sv)
        const int po = m.par * PAR;
        m.post(po, su, sv);
        const Sh shu = c.sh(su), shv = c.sh(sv);
        c.sync();
        m.par ^= 1;
        const Nb nsu = c.nbs_at(po), nsv = c.nbs_at(po + CHS);
        double* tr = trw + (size_t)n * ntr;
        for (int qp = 0; qp < Nc; qp += 2) {
            double P[2] = {0.0, 0.0}, Q[2] = {0.0, 0.0};
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = qp + j;
                if (q < Nc) {
                    const double* Hs = c.ring.cimg(q);
                    const double* Ha = c.ring.cimg(Nc + q);
                    const double B = -(smu * c.template trace_mm<false>(Hs, q, shv, nsv));
                    const double D = snu * c.template trace_mm<false>(Ha, q, shv, nsv);
                    const double C = snu * c.template trace_mm<false>(Hs, q, shu, nsu);
                    const double A = smu * c.template trace_mm<false>(Ha, q, shu, nsu);
                    P[j] = (B + C) * wgt;
                    Q[j] = (A + D) * wgt;
                }
            }
            // sums over the wave of P[0], P[1], Q[0], Q[1]: valid in the rows 0, 1, 2, 3 (wave_sum4: a, c, b, d)
            const double r = wave_sum4(P[0], Q[0], P[1], Q[1]);
            const int row = lane_ >> 4, q = qp + (row & 1);
            if (q < Nc) {
                const int l = lane_ & 15;
                if (l == 0) tr[q * JQ_NTR + (row < 2 ? 3 : 4)] = row < 2 ? -0.25 * r : 0.25 * r;
                else if (l < 4 && row < 2) tr[q * JQ_NTR + l - 1] = 0.0;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    st[s.foff] = u;
    st[(size_t)KT * 64 + s.foff] = v;
    st[(size_t)2 * KT * 64 + s.foff] = lr;
    st[(size_t)3 * KT * 64 + s.foff] = li;
}

// ---------------------------------------------------------------------------------------------
// Backward sweep with the two chains of a time step on TWO SETS of waves (round 4; the Stormer-Verlet cooperative-quad kernels
// have worked this way since round 2).  k_backward_cq_imr above runs the state step and the adjoint step of a time step one after
// the other on the same NT block waves: 2 x (3 + iterations) publication rounds plus one for the trace products.  But the state
// re-integration does not depend on the adjoint: here set 0 (NT block waves + 2 reducer waves) solves the state step of time step
// k while set 1 solves the adjoint step of time step k - 1 -- a software pipeline of depth one inside a chunk: n + 1 "super-steps"
// for n time steps, set 1 idle in the first, set 0 in the last.  Both sets pass the SAME workgroup barriers: a super-step takes
// max(rounds of the two solves) rounds instead of their sum.  With four waves per SIMD a round is bound by the SIMD's issue rate
// (~4.6 cycles per instruction of any kind), so the round loops are kept as lean as the one-set kernel's: while both sets iterate
// every wave reads both sets' decisions behind the barrier (one more LDS read); once one set has its result its waves only pass
// the barriers and watch the other set's decisions, and the other set runs the one-set loop -- all waves leave at the same barrier.
// The state set hands su = u_old + u_new, sv (the adjoint's forcing and the trace products' vectors) to the adjoint set through
// LDS, double-buffered over the super-steps: no publication round for the trace products any more.
// Staging: the implicit-midpoint kernels read only the MIDPOINT operators (time point 2k+1 of step k), and set 1 needs those of
// step k - 1 in super-step k: the ring of JQ_WIN_TPS slots holds the midpoints of the steps k - 1 .. k + 3 here (the even time
// points are never fetched); behind the first barrier of super-step k the slot of step k - 1 is dead and receives step k + 4.
// LDS behind the tables: [set][2 parities][3 channels][NT + 2 blocks][64] exchange images (channel 2 of image b, parities 0 / 1:
// su, sv of the super-steps with k & 1 == b), then the decisions [set][2 slots][2 parts].
// NT <= 6 (2 (NT + 2) waves <= 16); the results differ from k_backward_cq_imr's in nothing (same operations per chain, same order).
typedef int jq_int4 __attribute__((ext_vector_type(4)));
template <int NT, int SET>
struct CqImr2 : CqImr<NT> {
    typedef CqImr<NT> B;
    typedef typename B::Acc Acc;
    static constexpr int CHS = B::CHS, PAR = B::PAR;
    // decisions of the four reducer waves, [2 slots][2 sets][2 parts]: ONE 16-byte LDS read per round has both sets' (B::flags = my
    // set's pair of slot 0, for the reducers' writes)
    volatile __attribute__((address_space(3))) int* fl;

    // km: my set's decision on its x_{j-2}, ko: the other set's
    __device__ __forceinline__ void decisions(int j, int& km, int& ko) const
    {
        const jq_int4 v = *(volatile __attribute__((address_space(3))) jq_int4*)(fl + 4 * (j & 1));
        const int s0 = v.x & v.y, s1 = v.z & v.w;
        km = __builtin_amdgcn_readfirstlane(SET ? s1 : s0);
        ko = __builtin_amdgcn_readfirstlane(SET ? s0 : s1);
    }
    __device__ __forceinline__ int decided_o(int j) const
    {
        const long long v = *(volatile __attribute__((address_space(3))) long long*)(fl + 4 * (j & 1) + 2 * (1 - SET));
        return __builtin_amdgcn_readfirstlane((int)v & (int)(v >> 32));
    }
    __device__ __forceinline__ int decided_m(int j) const
    {
        const long long v = *(volatile __attribute__((address_space(3))) long long*)(fl + 4 * (j & 1) + 2 * SET);
        return __builtin_amdgcn_readfirstlane((int)v & (int)(v >> 32));
    }
    // a block wave of a set without work in this super-step (set 1 in the first, set 0 in the last): passes the other set's barriers
    template <bool STAGE, typename G>
    __device__ __forceinline__ void idle2(bool oact, G stage)
    {
        this->c->template sync<STAGE>();
        stage();
        this->c->sync();
        this->c->sync();
        if (oact)
            for (int j = 3;; ++j) {
                this->c->sync();
                if (decided_o(j)) break;
            }
    }
    // CqImr::step next to a set that is active in this super-step (oact) or not.  getf(fu, fv): the forcing, available behind the
    // first barrier; stage(): the window staging of this wave, behind the first barrier.
    // With four waves per SIMD a round is bound by the SIMD's issue rate: every instruction of the round loop counts (~ 1.3 % of the
    // sweep each; both sets' decisions in one LDS read: 162 -> 157 ms).  (Measured and rejected: parities and decision slots as
    // compile-time constants -- the step instantiated for both parities of its first publication, the loop written for six rounds:
    // the address arithmetic disappears, but hipcc fills the six exits with copies of the iterate registers: 157 -> 160 ms.)
    template <bool STAGE, typename F, typename G>
    __device__ __forceinline__ void step2(bool oact, double& u, double& v, F getf, G stage)
    {
        int po = this->par * PAR;
        this->post(po, u, v);
        Acc nx = this->own(0.0, 0.0, u, v);
        this->c->template sync<STAGE>();
        stage();
        double fu, fv;
        getf(fu, fv);
        this->nbr(nx, po);
        this->par ^= 1;
        const double Bu = nx.au - nx.kv, Bv = (nx.av + nx.sv);
        const double rhs_u = (u + fu) + Bu, rhs_v = (v + fv) + Bv;
        double au = rhs_u + Bu, av = rhs_v + Bv;      // x_1
        po = this->par * PAR;
        this->post(po, au, av);
        nx = this->own(rhs_u, rhs_v, au, av);
        this->c->sync();
        this->nbr(nx, po);
        this->par ^= 1;
        double bu = nx.au - nx.kv, bv = (nx.av + nx.sv);         // x_2
        po = this->par * PAR;
        this->post(po, bu, bv);
        nx = this->own(rhs_u, rhs_v, bu, bv);
        this->c->sync();
        this->nbr(nx, po);
        this->par ^= 1;
        double cu = nx.au - nx.kv, cv = (nx.av + nx.sv);         // x_3
        // (au, av) = x_{j-2}, (bu, bv) = x_{j-1}, (cu, cv) = x_j; three rounds per loop iteration as in CqImr::step.  Every round
        // reads both sets' decisions (those of a set that has its result, or no work, are stale: doth is true then)
        int j = 3;
        bool doth = !oact;
        auto turn = [&](double xu, double xv, double& ou, double& ov) -> bool {
            const int po_ = this->par * PAR;
            this->post(po_, xu, xv);
            Acc t = this->own(rhs_u, rhs_v, xu, xv);
            this->c->sync();
            this->par ^= 1;
            int km, ko;
            decisions(j, km, ko);                      // on the two x_{j-2}
            ++j;
            if (ko) doth = true;
            this->nbr(t, po_);
            if (km) return true;
            ou = t.au - t.kv, ov = t.av + t.sv;
            return false;
        };
        for (;;) {
            if (turn(cu, cv, au, av)) {
                u = au, v = av;
                break;
            }
            if (turn(au, av, bu, bv)) {
                u = bu, v = bv;
                break;
            }
            if (turn(bu, bv, cu, cv)) {
                u = cu, v = cv;
                break;
            }
        }
        if (!doth)
            for (;; ++j) {                             // the other set still iterates
                this->c->sync();
                if (decided_o(j)) break;
            }
    }
    // CqImr::reducer_step likewise
    __device__ __forceinline__ void reducer_idle2(bool oact)
    {
        this->c->sync();
        this->c->sync();
        this->c->sync();
        if (oact)
            for (int j = 3;; ++j) {
                this->c->sync();
                if (decided_o(j)) break;
            }
    }
    __device__ __forceinline__ void reducer2(bool oact, int part)
    {
        this->c->sync();
        this->c->sync();      // x_1
        double prev[NT];
        {
            const int po = (this->par ^ 1) * PAR + part * CHS;
#pragma unroll
            for (int w = 0; w < NT; ++w) prev[w] = this->x0[po + (w + 1) * 64];
        }
        // behind barrier j: the decision on x_{j-1} from the published x_j
        auto decide = [&](int j, int po) {
            double acc = 0.0, acc1 = 0.0;
#pragma unroll
            for (int w = 0; w < NT; ++w) {
                const double cur = this->x0[po + (w + 1) * 64];
                const double d = prev[w] - cur;
                if (w & 1) acc1 = fma(d, d, acc1);
                else acc = fma(d, d, acc);
                prev[w] = cur;
            }
            acc += acc1;
            // (the sum is valid in the lanes 48 .. 63: they write the decision, the others a dummy word behind the decisions)
            const bool keep = (j - 1 >= this->max_iter) || (wave_sum_hi(acc) < this->tol2);      // (a NaN never converges, as in the reference)
            fl[((this->c->lane >= 48) ? 0 : 8) + 4 * ((j - 1) & 1) + 2 * SET + part] = keep ? 1 : 0;
        };
        {
            const int po = this->par * PAR + part * CHS;
            this->c->sync();
            this->par ^= 1;
            decide(2, po);
        }
        int j = 3;
        bool dme = false, doth = !oact;
        if (!doth)
            for (;; ++j) {
                const int po = this->par * PAR + part * CHS;
                this->c->sync();
                this->par ^= 1;
                int km, ko;
                decisions(j, km, ko);
                if (!km) decide(j, po);
                if (km | ko) {
                    dme = km != 0, doth = ko != 0;
                    ++j;
                    break;
                }
            }
        if (!dme) {
            for (;; ++j) {
                const int po = this->par * PAR + part * CHS;
                this->c->sync();
                this->par ^= 1;
                if (decided_m(j)) break;
                decide(j, po);
            }
        } else if (!doth) {
            for (;; ++j) {
                this->c->sync();
                if (decided_o(j)) break;
            }
        }
    }
};

// grid = 4 * nslabs (workgroup = one evaluation, N = 4), block = 2 * 64 * (NT + 2): set 0 = state chain (NT block waves, 2 reducer
// waves), set 1 = adjoint chain.  Dynamic LDS: staging + tables + 4 PAR doubles + 64 bytes.
template <int NT>
__global__ __launch_bounds__(128 * NT + 256) void k_backward_cq_imr2(PropArgs a)
{
    static_assert(NT <= 6, "2 (NT + 2) waves per workgroup");
    typedef typename CoopQ<NT>::Sh Sh;
    typedef typename CoopQ<NT>::Nb Nb;
    constexpr int CHS = CoopQ<NT>::CHS, PAR = CoopQ<NT>::PAR, KT = 4 * NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    CqSetup<NT> s = cq_setup<NT>(a);
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // waves 0 .. 2 NT - 1: block wave wid % NT of set wid / NT; the last four: reducers (set 0 part 0, 1; set 1 part 0, 1) -- with
    // NT = 6 every SIMD (wave id mod 4) gets three block waves and one reducer wave
    const bool reducer = wid >= 2 * NT;
    const int set = reducer ? (wid - 2 * NT) >> 1 : (wid >= NT ? 1 : 0);
    const int wv = reducer ? NT + ((wid - 2 * NT) & 1) : wid - set * NT;      // wv < NT: block wave wv; NT, NT + 1: reducer of part 0, 1
    const int wave = reducer ? 0 : wv, lane_ = s.lane_;
    s.foff = (size_t)(4 * wave + ((lane_ >> 2) & 3)) * 64 + 16 * (lane_ >> 4) + s.col;
    const int Nc = a.Ncoupled, ntr = Nc * JQ_NTR;
    const size_t trow = ((size_t)s.slab * a.qps + s.qd) * NT;      // first of my workgroup's NT record rows
    if (!s.active) {
        if (s.qd < a.qps)
            for (size_t k = threadIdx.x; k < (size_t)NT * a.nsteps_chunk * ntr; k += blockDim.x) a.traces[trow * a.nsteps_chunk * ntr + k] = 0.0;
        return;
    }
    double* tab = (double*)(smem + a.lds_tab_off);
    for (int i = threadIdx.x; i < 32 * NT; i += blockDim.x) tab[(i & ~15) + 4 * (i & 3) + ((i >> 2) & 3)] = a.tabs[i];
    double* xbuf = tab + 32 * NT;
    for (int i = threadIdx.x; i < 4 * PAR + 8; i += blockDim.x) xbuf[i] = 0.0;      // (the pads stay zero; the decisions too)
    const int n = a.nsteps_chunk;
    CoopQ<NT> c;
    c.mt = wave, c.lane = lane_;
    c.xb = (jq_lds_double*)(xbuf + set * 2 * PAR + wave * 64 + lane_);
    // the ring of midpoint operators (see above): slot i % JQ_WIN_TPS <- time point 2 i + 1; the constant images behind it
    WinRing& r = c.ring;
    r.smem = smem, r.wave = wid, r.lane = lane_, r.nwaves = 2 * NT + 4;
    r.stride_b = (unsigned)(a.stride * 8), r.slot_bytes = 2 * r.stride_b, r.cbase = JQ_WIN_TPS * r.slot_bytes, r.pieces2 = 2 * a.pieces;
    int inext = 0;
    unsigned snext = 0;
    auto fetch = [&] {
        if (inext >= n) return;
        r.dma((const char*)a.stream + (size_t)(2 * inext + 1) * r.slot_bytes, smem + snext, r.pieces2);
        ++inext;
        snext += r.slot_bytes;
        if (snext == r.cbase) snext = 0;
    };
    r.dma((const char*)a.cimg, smem + r.cbase, 2 * a.Ncoupled * a.pieces);
    for (int i = 0; i < JQ_WIN_TPS - 1; ++i) fetch();      // steps 0 .. 3; super-step k fetches step k + 4 into the slot of step k - 1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();      // (tables, zeroed exchange images, the first operators)
    asm volatile("" ::: "memory");
    r.wave = wave, r.nwaves = NT;      // (from here on the block waves of set 0 stage)
    double* st = a.state + (size_t)s.slab * a.state_stride;
    auto init = [&](auto& m) {
        m.c = &c;
        m.x0 = (jq_lds_double*)(xbuf + set * 2 * PAR + lane_);
        m.cw = 0.0;
        m.cwa = (a.use_shift && (lane_ >> 4) == (lane_ & 3))
                    ? 0.5 * a.h * a.colinfo[(size_t)s.slab * 32 + s.col] * tab[16 * NT + 16 * wave + 4 * (lane_ & 3) + ((lane_ >> 2) & 3)] : 0.0;
        m.tol2 = a.jacobi_tol2, m.max_iter = a.m, m.par = 0, m.use_shift = a.use_shift;
        m.fl = (volatile __attribute__((address_space(3))) int*)(xbuf + 4 * PAR);
        m.flags = m.fl + 2 * set;
    };
    if (reducer) {
        if (set == 0) {
            CqImr2<NT, 0> m;
            init(m);
            for (int k = 0; k <= n; ++k) {
                if (k < n) m.reducer2(k >= 1, wv - NT);
                else m.reducer_idle2(true);
            }
        } else {
            CqImr2<NT, 1> m;
            init(m);
            for (int k = 0; k <= n; ++k) {
                if (k >= 1) m.reducer2(k < n, wv - NT);
                else m.reducer_idle2(true);
            }
        }
        return;
    }
    // su, sv of super-step k: channel 2 of image k & 1, parities 0 / 1; my block at + 64 of hb[...], the neighbours' at + 0, + 128
    jq_lds_double* hb = (jq_lds_double*)(xbuf + wave * 64 + lane_);
    unsigned cur = 0;      // byte offset of the ring slot of step k
    if (set == 0) {
        CqImr2<NT, 0> m;
        init(m);
        double u = st[s.foff], v = st[(size_t)KT * 64 + s.foff];
        for (int k = 0; k < n; ++k) {
            m.K = c.load((const double*)(smem + cur) + lane_);
            m.fold_shift();
            m.S = c.load((const double*)(smem + cur + r.stride_b) + lane_);
            cur += r.slot_bytes;
            if (cur == r.cbase) cur = 0;
            const double u0 = u, v0 = v;
            m.template step2<true>(k >= 1, u, v, [](double& fu, double& fv) { fu = 0.0, fv = 0.0; }, fetch);
            const int ho = (k & 1) * 2 * PAR + 2 * CHS + 64;
            hb[ho] = u0 + u;
            hb[ho + PAR] = v0 + v;
        }
        m.template idle2<true>(true, [] {});
        st[s.foff] = u;
        st[(size_t)KT * 64 + s.foff] = v;
        return;
    }
    CqImr2<NT, 1> m;
    init(m);
    double lr = st[(size_t)2 * KT * 64 + s.foff], li = st[(size_t)3 * KT * 64 + s.foff];
    const double wgt = a.colinfo[(size_t)s.slab * 32 + 16 + s.col];
    const double cfw = a.forced ? -a.h * a.tinv * tab[16 * wave + s.g] : 0.0;      // h * (-tinv * W): W applied row-wise
    double* trw = a.traces + ((trow + wave) * a.nsteps_chunk) * ntr;
    m.template idle2<false>(true, [] {});
    for (int k = 1; k <= n; ++k) {
        m.K = c.load((const double*)(smem + cur) + lane_);      // operators of time step k - 1
        m.fold_shift();
        m.S = c.load((const double*)(smem + cur + r.stride_b) + lane_);
        cur += r.slot_bytes;
        if (cur == r.cbase) cur = 0;
        const int ho = ((k + 1) & 1) * 2 * PAR + 2 * CHS;      // su, sv of super-step k - 1
        const double l0 = lr, l1 = li;
        double su = 0.0, sv = 0.0;
        m.template step2<false>(k < n, lr, li, [&](double& fu, double& fv) {
            su = hb[ho + 64], sv = hb[ho + PAR + 64];
            fu = cfw * su, fv = cfw * sv;
        }, [] {});
        const double smu = l0 + lr, snu = l1 + li;
        // trace products with the constant images (Hsym_q: image q, Hanti_q: image Nc + q)
        const Sh shu = c.sh(su), shv = c.sh(sv);
        Nb nsu, nsv;
        nsu.b = hb[ho], nsu.a = hb[ho + 128];
        nsv.b = hb[ho + PAR], nsv.a = hb[ho + PAR + 128];
        double* tr = trw + (size_t)(k - 1) * ntr;
        for (int qp = 0; qp < Nc; qp += 2) {
            double P[2] = {0.0, 0.0}, Q[2] = {0.0, 0.0};
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = qp + j;
                if (q < Nc) {
                    const double* Hs = c.ring.cimg(q);
                    const double* Ha = c.ring.cimg(Nc + q);
                    const double B = -(smu * c.template trace_mm<false>(Hs, q, shv, nsv));
                    const double D = snu * c.template trace_mm<false>(Ha, q, shv, nsv);
                    const double C = snu * c.template trace_mm<false>(Hs, q, shu, nsu);
                    const double A = smu * c.template trace_mm<false>(Ha, q, shu, nsu);
                    P[j] = (B + C) * wgt;
                    Q[j] = (A + D) * wgt;
                }
            }
            // sums over the wave of P[0], P[1], Q[0], Q[1]: valid in the rows 0, 1, 2, 3 (wave_sum4: a, c, b, d)
            const double r4 = wave_sum4(P[0], Q[0], P[1], Q[1]);
            const int row = lane_ >> 4, q = qp + (row & 1);
            if (q < Nc) {
                const int l = lane_ & 15;
                if (l == 0) tr[q * JQ_NTR + (row < 2 ? 3 : 4)] = row < 2 ? -0.25 * r4 : 0.25 * r4;
                else if (l < 4 && row < 2) tr[q * JQ_NTR + l - 1] = 0.0;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    st[(size_t)2 * KT * 64 + s.foff] = lr;
    st[(size_t)3 * KT * 64 + s.foff] = li;
}

// ---------------------------------------------------------------------------------------------
// Backward sweep on THREE workgroups (CUs) per evaluation, as k_backward_cq3 (jq_cq_split_kernels.h, whose hand-off ring, progress
// counters, XCD check and dead-wait fallback this kernel shares): the two sets of k_backward_cq_imr2 on one CU are bound by the
// SIMDs' issue rate (four waves per SIMD); with a CU of its own a chain runs at the forward sweep's pace.
//   role 0  state re-integration: CqImr::step per time step; stores su = u_old + u_new, sv
//   role 1  adjoint step, >= 1 step behind: loads su, sv of its block (forcing -W (v + v_s) / T); stores smu, snu
//   role 2  trace products (adjoint_grad_calc_m): loads su, sv with their neighbouring blocks, smu, snu; one record row per block wave
// Roles 0 and 1: NT block waves + two reducer waves, the code of k_forward_cq_imr (one more barrier per step: behind it the stores
// of the step are acknowledged and the counter is published).  Bit-identical to k_backward_cq_imr / _imr2.
// grid = 24 * ceil(evaluations / 8), block = 64 * (NT + 2); a.park: the hand-off buffer, zeroed by the host before every launch.
template <int NT, bool DN = false>
__global__ __launch_bounds__(64 * NT + 128) void k_backward_cq_imr3(PropArgs a)
{
    typedef typename CoopQ<NT, DN>::Sh Sh;
    typedef typename CoopQ<NT, DN>::Nb Nb;
    const int role = ((int)blockIdx.x >> 3) % 3;
    const int quad = 8 * ((int)blockIdx.x / 24) + ((int)blockIdx.x & 7);
    const CqSetup<NT> s = cq_setup<NT>(a, quad >> 2, quad & 3);
    const int Nc = a.Ncoupled, ntr = Nc * JQ_NTR, nst = a.nsteps_chunk;
    const size_t trow = ((size_t)s.slab * a.qps + s.qd) * NT;      // first of my evaluation's NT record rows
    cq3_arrive(a);      // (start-up rendezvous of the whole grid: jq_cq_split_kernels.h)
    if (s.slab >= a.nslabs) return;
    if (!s.active) {
        if (role == 2 && s.qd < a.qps)
            for (size_t k = threadIdx.x; k < (size_t)NT * nst * ntr; k += blockDim.x) a.traces[trow * nst * ntr + k] = 0.0;
        return;
    }
    Cq3Hand<NT> hd;
    hd.init(a, (size_t)quad, s.lane_);
    {
        extern __shared__ __attribute__((aligned(16))) char smem_rdv[];      // (the window ring at the start of the LDS is not in use yet)
        if (!cq3_rendezvous(a, (int*)smem_rdv)) return;
    }
    // (test hook as in k_backward_cq3: option debug bit 16 / 32 -- the consumer roles / the state role start ~ 5 ms late; results unchanged)
    if ((a.debug & 16) && role != 0)
        for (int i = 0; i < 1500; ++i) __builtin_amdgcn_s_sleep(127);
    if ((a.debug & 32) && role == 0)
        for (int i = 0; i < 1500; ++i) __builtin_amdgcn_s_sleep(127);
    {
        const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));
        if (threadIdx.x == 0) __hip_atomic_store(hd.head + 32 + role, (unsigned long long)xcc + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (role == 2) {
        if (s.chain) return;      // (no reducer waves here)
        extern __shared__ __attribute__((aligned(16))) char smem2[];
        const int lane_ = s.lane_, wave = s.wave;
        CoopQ<NT, DN> c;
        c.mt = wave, c.lane = lane_, c.xb = nullptr;
        WinRing& r = c.ring;      // (only the constant images are staged)
        r.smem = smem2, r.wave = wave, r.lane = lane_, r.nwaves = NT;
        r.stride_b = (unsigned)(a.stride * 8), r.slot_bytes = 2 * r.stride_b, r.cbase = JQ_WIN_TPS * r.slot_bytes, r.pieces2 = 2 * a.pieces;
        r.dma((const char*)a.cimg, smem2 + r.cbase, 2 * a.Ncoupled * a.pieces);
        const double wgt = a.colinfo[(size_t)s.slab * 32 + 16 + s.col];
        double* trw = a.traces + ((trow + wave) * nst) * ntr;
        if (wave == 0) hd.wait(1, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        struct Rec {
            double su, sv, smu, snu;
            Nb nu, nv;
        };
        auto fetch = [&](int n) {
            Rec e;
            e.su = hd.load(n, 0, wave), e.sv = hd.load(n, 1, wave), e.smu = hd.load(n, 2, wave), e.snu = hd.load(n, 3, wave);
            const int wb = wave > 0 ? wave - 1 : wave, wa_ = wave + 1 < NT ? wave + 1 : wave;
            const double ub_ = hd.load(n, 0, wb), ua_ = hd.load(n, 0, wa_), vb_ = hd.load(n, 1, wb), va_ = hd.load(n, 1, wa_);
            const bool lo = wave == 0, hi = wave + 1 == NT;      // (zeros beyond the edge blocks)
            e.nu = c.nb_make(lo ? 0.0 : ub_, hi ? 0.0 : ua_);
            e.nv = c.nb_make(lo ? 0.0 : vb_, hi ? 0.0 : va_);
            return e;
        };
        Rec cur = fetch(0);
        for (int n = 0; n < nst; ++n) {
            if (wave == 0) hd.wait(1, (unsigned long long)(n + 2 < nst ? n + 2 : nst));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (cur has landed)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (wave == 0 && lane_ == 0) hd.publish(2, (unsigned long long)(n + 1));
            Rec nxt = cur;
            if (n + 1 < nst) nxt = fetch(n + 1);
            const Sh shu = c.sh(cur.su), shv = c.sh(cur.sv);
            double* tr = trw + (size_t)n * ntr;
            for (int qp = 0; qp < Nc; qp += 2) {
                double P[2] = {0.0, 0.0}, Q[2] = {0.0, 0.0};
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int q = qp + j;
                    if (q < Nc) {
                        const double* Hs = c.ring.cimg(q);
                        const double* Ha = c.ring.cimg(Nc + q);
                        const double B = -(cur.smu * c.template trace_mm<false>(Hs, q, shv, cur.nv));
                        const double D = cur.snu * c.template trace_mm<false>(Ha, q, shv, cur.nv);
                        const double C = cur.snu * c.template trace_mm<false>(Hs, q, shu, cur.nu);
                        const double A = cur.smu * c.template trace_mm<false>(Ha, q, shu, cur.nu);
                        P[j] = (B + C) * wgt;
                        Q[j] = (A + D) * wgt;
                    }
                }
                const double r4 = wave_sum4(P[0], Q[0], P[1], Q[1]);
                const int row = lane_ >> 4, q = qp + (row & 1);
                if (q < Nc) {
                    const int l = lane_ & 15;
                    if (l == 0) tr[q * JQ_NTR + (row < 2 ? 3 : 4)] = row < 2 ? -0.25 * r4 : 0.25 * r4;
                    else if (l < 4 && row < 2) tr[q * JQ_NTR + l - 1] = 0.0;
                }
            }
            cur = nxt;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            hd.publish(2, (unsigned long long)nst);
            const unsigned long long x0 = __hip_atomic_load(hd.head + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long x1 = __hip_atomic_load(hd.head + 33, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long x2 = __hip_atomic_load(hd.head + 34, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (x0 != x2 || x1 != x2) __hip_atomic_store(hd.gerr, 2ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    JQ_CQ_IMR_PROLOGUE
    (void)wsr;
    if (s.chain) {      // reducer waves: one solve and the publication barrier per time step (+ the start barrier of role 1, the last publication)
        if (role == 1) c.sync();
        for (int n = 0; n < nst; ++n) {
            m.reducer_step(wave);
            c.sync();
        }
        c.sync();
        return;
    }
    // (a step's stores are published one step later -- behind the barrier that follows the NEXT solve: their latency is off the
    //  critical path; the consumers fetch a step ahead anyway)
    if (role == 0) {
        double u = st[s.foff], v = st[(size_t)KT * 64 + s.foff];
        for (int n = 0; n < nst; ++n) {
            if (wave == 0 && n >= JQ_CQ3_SLOTS) hd.wait(2, (unsigned long long)(n - JQ_CQ3_SLOTS + 1));      // (the slot is free)
            m.K = c.load(c.ring.template ks<0, 1>());
            m.fold_shift();
            m.S = c.load(c.ring.template ks<1, 1>());
            const double u0 = u, v0 = v;
            m.template step<true>(u, v, 0.0, 0.0);      // (wave 0 reaches the step's first barrier only when the slot is free)
            c.ring.advance();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the stores of step n - 1: long acknowledged)
            c.sync();
            if (wave == 0 && lane_ == 0) hd.publish(0, (unsigned long long)n);
            hd.store(n, 0, wave, u0 + u);
            hd.store(n, 1, wave, v0 + v);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        c.sync();
        if (wave == 0 && lane_ == 0) hd.publish(0, (unsigned long long)nst);
        st[s.foff] = u;
        st[(size_t)KT * 64 + s.foff] = v;
        return;
    }
    double lr = st[(size_t)2 * KT * 64 + s.foff], li = st[(size_t)3 * KT * 64 + s.foff];
    const double cfw = a.forced ? -a.h * a.tinv * wdr : 0.0;      // h * (-tinv * W): W applied row-wise
    if (wave == 0) hd.wait(0, 1);
    c.sync();
    double hsu = hd.load(0, 0, wave), hsv = hd.load(0, 1, wave);
    for (int n = 0; n < nst; ++n) {
        if (wave == 0) hd.wait(0, (unsigned long long)(n + 2 < nst ? n + 2 : nst));
        m.K = c.load(c.ring.template ks<0, 1>());
        m.fold_shift();
        m.S = c.load(c.ring.template ks<1, 1>());
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (su, sv of this step have landed; my stores of step n - 1 are acknowledged)
        const double su = hsu, sv = hsv, l0 = lr, l1 = li;
        m.template step<true>(lr, li, cfw * su, cfw * sv);
        c.ring.advance();
        c.sync();
        if (wave == 0 && lane_ == 0) hd.publish(1, (unsigned long long)n);      // (everybody's stores of the steps < n: acknowledged in front of the solve)
        if (n + 1 < nst) hsu = hd.load(n + 1, 0, wave), hsv = hd.load(n + 1, 1, wave);      // (role 0 has finished the steps <= n + 1: seen in front of the step's barriers)
        hd.store(n, 2, wave, l0 + lr);
        hd.store(n, 3, wave, l1 + li);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    c.sync();
    if (wave == 0 && lane_ == 0) hd.publish(1, (unsigned long long)nst);
    st[(size_t)2 * KT * 64 + s.foff] = lr;
    st[(size_t)3 * KT * 64 + s.foff] = li;
}
