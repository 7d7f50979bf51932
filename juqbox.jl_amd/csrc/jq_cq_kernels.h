// jq_cq_kernels.h -- "cooperative quad" propagators: the LATENCY path of the JQ_BW_T4 structure (cnot3: 1 .. 256 samples).
//
// The quad-layout kernels of jq_kernels.h give one wave four state columns and ALL 16-row blocks of every array: a single
// evaluation is one wave's chain of 72 dependent products per time step, each of them NT blocks long (cnot3: 6 blocks,
// ~240 ns), and the rest of the GPU idles.  Here the NT blocks of a column quad are split over the NT waves of a workgroup:
// wave `mt` owns block mt of every array (ONE register per array), a product costs one block's work per wave
//     D[mt] = C[mt] + B_mt x[mt] (one v_mfma_f64_4x4x4_4b) + c0 shr4(x[mt]) + c1 shl4(x[mt]) + c2 x[mt-1] + c3 x[mt+1]
// plus the exchange of x with the two neighbouring waves through a double-buffered LDS image (one ds_write, one workgroup
// barrier, two ds_reads that fly while the wave's own block is multiplied).  Because the barrier is what a product costs
// here, the step is regrouped so that every published x serves all the products that need it (K05 u and S0 u; S05 v05,
// K0 v05 and K1 v05; the five products with X, ...), and the backward sweep runs its two chains -- state re-integration and
// adjoint step -- side by side with ONE barrier per pair of publications: 17 barriers per forward step (20 products), 18
// per backward step (52 products) at m = 6 Neumann terms.  Same operators, images, window staging (Ring, batch < 0), state file,
// trace records and reductions as the quad-layout kernels; the regrouping only reorders floating-point additions.
#pragma once
#include "jq_kernels.h"

// Exchange image in LDS: [2 parities][2 channels][NT + 2 blocks][64] doubles -- a zero block in front of and behind the NT
// blocks of a channel, so that the neighbours of the edge blocks need no clamping and every access is ONE base register
// (this wave's block of channel 0 in the current parity) plus a compile-time offset.
typedef __attribute__((address_space(3))) double jq_lds_double;
template <int NT>
struct CoopQ {
    static constexpr int CHS = (NT + 2) * 64;       // doubles per channel
    static constexpr int PAR = 2 * CHS;              // doubles per parity
    Ring ring;
    jq_lds_double* xp;  // my block of channel 0 in the parity that holds the published vectors
    int delta;          // doubles from that parity to the other one (+-PAR)
    int mt;             // my block
    int lane;
    double xown;        // my block of the published x (channel 0)
    double xown1;       // ... of channel 1 (publish2: the backward sweep publishes one vector of each of its two chains per barrier)

    __device__ __forceinline__ void setup(double* xbuf, int wave, int lane_)
    {
        mt = wave, lane = lane_, xown = 0.0, xown1 = 0.0;
        for (int i = threadIdx.x; i < 2 * PAR; i += blockDim.x) xbuf[i] = 0.0;      // (the pads stay zero)
        xp = (jq_lds_double*)(xbuf + (1 + wave) * 64 + lane_);
        delta = PAR;
    }
    __device__ __forceinline__ void flip()
    {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        xp += delta;
        delta = -delta;
    }
    __device__ __forceinline__ void publish(double x)
    {
        xp[delta] = x;
        xown = x;
        flip();
    }
    __device__ __forceinline__ void publish2(double x0, double x1)
    {
        xp[delta] = x0;
        xp[delta + CHS] = x1;
        xown = x0;
        xown1 = x1;
        flip();
    }
    // my block of an operator image (LDS, lane offset applied): A operand of the MFMA + the four coupling coefficients of my row.
    // Loaded BEFORE the publication whose products use it: behind the barrier only the neighbours' x is still in flight.
    struct Op {
        double a;
        d4 c;
    };
    __device__ __forceinline__ Op load(const double* M) const
    {
        Op o;
        o.a = M[mt * 64];
        o.c = t4q_cload(t4q_c<NT>(M, lane), mt);
        return o;
    }
    template <bool ZEROC, int CH = 0>
    __device__ __forceinline__ double mm(double C, const Op& o) const
    {
        const double xo = CH ? xown1 : xown;
        const double xb = xp[CH * CHS - 64];
        const double xa = xp[CH * CHS + 64];
        double acc = ZEROC ? 0.0 : C;
        acc = __builtin_amdgcn_mfma_f64_4x4x4f64(o.a, xo, acc, 0, 0, 0);
        acc = fma(o.c[0], row_shift4<0x114>(xo), acc);
        acc = fma(o.c[1], row_shift4<0x104>(xo), acc);
        acc = fma(o.c[2], xb, acc);      // (the coefficients of a missing neighbour are zero)
        return fma(o.c[3], xa, acc);
    }
    // C + M x  for the published x (M: LDS image with the lane offset applied; MODE: JQ_T4_* parts that are non-zero)
    template <bool ZEROC, int MODE = JQ_T4_DIAG | JQ_T4_RTERMS | JQ_T4_MTERMS, int CH = 0>
    __device__ __forceinline__ double mm(double C, const double* M) const
    {
        constexpr bool diag = MODE & JQ_T4_DIAG, rt = MODE & JQ_T4_RTERMS, mtm = MODE & JQ_T4_MTERMS;
        const double xo = CH ? xown1 : xown;
        double xb = 0.0, xa = 0.0;
        if constexpr (mtm) {      // neighbour blocks first: their LDS latency hides behind this block's own work
            xb = xp[CH * CHS - 64];
            xa = xp[CH * CHS + 64];
        }
        double acc = ZEROC ? 0.0 : C;
        d4 c = {0.0, 0.0, 0.0, 0.0};
        if constexpr (rt || mtm) c = t4q_cload(t4q_c<NT>(M, lane), mt);
        if constexpr (diag) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(M[mt * 64], xo, acc, 0, 0, 0);
        if constexpr (rt) {
            acc = fma(c[0], row_shift4<0x114>(xo), acc);
            acc = fma(c[1], row_shift4<0x104>(xo), acc);
        }
        if constexpr (mtm) {      // (the coefficients of a missing neighbour are zero)
            acc = fma(c[2], xb, acc);
            acc = fma(c[3], xa, acc);
        }
        return acc;
    }
    // trace operators touch one part of the image only (a.bw_trace: JQ_T4_* bits); anything else takes the full product
    template <int CH = 0>
    __device__ __forceinline__ double mm_z_mode(const double* M, int mode) const
    {
        switch (mode) {
        case JQ_T4_DIAG: return mm<true, JQ_T4_DIAG, CH>(0.0, M);
        case JQ_T4_RTERMS: return mm<true, JQ_T4_RTERMS, CH>(0.0, M);
        case JQ_T4_MTERMS: return mm<true, JQ_T4_MTERMS, CH>(0.0, M);
        default: return mm<true, JQ_T4_DIAG | JQ_T4_RTERMS | JQ_T4_MTERMS, CH>(0.0, M);
        }
    }
    // two Horner chains side by side (channel 0: base0 + sum_j S0^j A0, channel 1 likewise): ONE barrier per pair of products
    __device__ __forceinline__ void horner2(double base0, double A0, const Op& S0, double base1, double A1, const Op& S1, int m, double& out0,
                                            double& out1)
    {
        if (m <= 0) {
            out0 = base0, out1 = base1;
            return;
        }
        double Y0 = A0, Y1 = A1;
        for (int j = 1; j < m; ++j) {
            publish2(Y0, Y1);
            Y0 = mm<false, 0>(A0, S0);
            Y1 = mm<false, 1>(A1, S1);
        }
        publish2(Y0, Y1);
        out0 = mm<false, 0>(base0, S0);
        out1 = mm<false, 1>(base1, S1);
    }
    // base + sum_{j=1..m} S^j A  (Horner form, jq_kernels.h): m publications
    __device__ __forceinline__ double horner(double base, double A, const Op& S, int m)
    {
        if (m <= 0) return base;
        double Y = A;
        for (int j = 1; j < m; ++j) {
            publish(Y);
            Y = mm<false>(A, S);
        }
        publish(Y);
        return mm<false>(base, S);
    }
};

// sum of val over the NT waves (wave order) and the four 4-row groups of a block, for the lanes with group 0; valid in wave 0.
// scratch: LDS [NT][64].  Contains workgroup barriers.
template <int NT>
__device__ __forceinline__ double cq_wg_sum(double val, double* scratch, int wave, int lane)
{
    val = row_ror_add<8>(row_ror_add<4>(val));
    __syncthreads();
    scratch[wave * 64 + lane] = val;
    __syncthreads();
    double s = 0.0;
    if (wave == 0)
        for (int w = 0; w < NT; ++w) s += scratch[w * 64 + lane];
    return s;
}

// the six operator blocks of a time step (this wave's share of K, S at the time points 2n, 2n+1, 2n+2 of the chunk)
template <int NT>
struct CqOps {
    typename CoopQ<NT>::Op Kp05, S05, Kn0, S0, Kn1, S1;
};
template <int NT>
__device__ __forceinline__ CqOps<NT> cq_load_ops(CoopQ<NT>& c)
{
    CqOps<NT> o;
    o.Kp05 = c.load(c.ring.template next_ks<0, 1>());
    o.S05 = c.load(c.ring.template next_ks<1, 1>());
    o.Kn0 = c.load(c.ring.template next_ks<0, 0>());
    o.S0 = c.load(c.ring.template next_ks<1, 0>());
    o.Kn1 = c.load(c.ring.template next_ks<0, 2>());
    o.S1 = c.load(c.ring.template next_ks<1, 2>());
    return o;
}

// state step: in u, v; out un = u(t+h), v05, vN = v05 + S05 v05 (the caller adds Kp05 un)
template <int NT>
__device__ __forceinline__ void cq_state(CoopQ<NT>& c, const PropArgs& a, const CqOps<NT>& o, double cw, double u, double v, double& un,
                                         double& v05, double& vN)
{
    // x = u: A = c K05 u ; P = u + c S0 u
    c.publish(u);
    double A = c.template mm<true>(0.0, o.Kp05);
    const double P = c.template mm<false>(u, o.S0);
    if (a.use_shift) A = fma(cw, u, A);
    // x = v: A = c (K05 u + S05 v) ; v05 = v + sum_j S^j A
    c.publish(v);
    A = c.template mm<false>(A, o.S05);
    v05 = c.horner(v + A, A, o.S05, a.m);
    // x = v05: vN = v05 + c S05 v05 ; un = u + c (S0 u - K0 v05) ; A = -c K1 v05
    c.publish(v05);
    vN = c.template mm<false>(v05, o.S05);
    un = c.template mm<false>(P, o.Kn0);
    A = c.template mm<true>(0.0, o.Kn1);
    if (a.use_shift) {
        un = fma(-cw, v05, un);
        A = fma(-cw, v05, A);
    }
    // x = un: A = c (S1 un - K1 v05) ; un += sum_j S^j A
    c.publish(un);
    A = c.template mm<false>(A, o.S1);
    un = c.horner(un + A, A, o.S1, a.m);
}

template <int NT>
struct CqSetup {
    int lane_, wave, qd, slab, col, g;
    bool active;
    size_t foff;        // offset of my element in an array image of the slab file
};
template <int NT>
__device__ __forceinline__ CqSetup<NT> cq_setup(const PropArgs& a)
{
    CqSetup<NT> s;
    s.lane_ = threadIdx.x & 63;
    s.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    s.slab = blockIdx.x >> 2;
    s.qd = blockIdx.x & 3;
    s.col = 4 * s.qd + (s.lane_ & 3);
    s.g = 4 * (s.lane_ >> 4) + ((s.lane_ >> 2) & 3);      // offset in a block of the row tables ([block][row in group][group])
    s.foff = (size_t)(4 * s.wave + ((s.lane_ >> 2) & 3)) * 64 + 16 * (s.lane_ >> 4) + s.col;
    // columns of this slab that carry a state (the slabs are packed with whole samples; N > 16: parts of one sample)
    int used;
    if (a.parts > 1) {
        used = a.N - 16 * (s.slab % a.parts);
        if (used > 16) used = 16;
    } else {
        int ns = a.nsamples - s.slab * a.sps;
        if (ns > a.sps) ns = a.sps;
        used = ns * a.N;
    }
    s.active = 4 * s.qd < used;
    return s;
}

// ---------------------------------------------------------------------------------------------
// grid = 4 * nslabs (workgroup = quad qd of slab blockIdx.x / 4), block = 64 * NT
template <int NT>
__global__ __launch_bounds__(64 * NT) void k_forward_cq(PropArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KT = 4 * NT;
    const CqSetup<NT> s = cq_setup<NT>(a);
    if (!s.active) return;      // (a quad without columns: the whole workgroup leaves before any barrier)
    const int lane_ = s.lane_, wave = s.wave;
    double* tab = (double*)(smem + a.lds_tab_off);
    for (int i = threadIdx.x; i < 32 * NT; i += blockDim.x) tab[(i & ~15) + 4 * (i & 3) + ((i >> 2) & 3)] = a.tabs[i];   // [block][g][r]
    CoopQ<NT> c;
    double* scratch = tab + 32 * NT + 2 * CoopQ<NT>::PAR;
    c.setup(tab + 32 * NT, wave, lane_);
    c.ring.init(smem, a, wave, lane_, NT);      // (window mode: barrier inside)
    const double wdr = tab[16 * wave + s.g], wsr = tab[16 * NT + 16 * wave + s.g];

    double* st = a.state + (size_t)s.slab * a.state_stride;
    double u = st[s.foff], v = st[(size_t)KT * 64 + s.foff];
    const bool slot0 = wave == 0 && ((lane_ >> 2) & 3) == 0;      // the lanes that carry per-column partials between chunks
    const size_t cslot = 16 * (lane_ >> 4) + s.col;
    double leak = slot0 ? st[(size_t)(JQ_STATE_ARRAYS * KT + JQ_MAXNC) * 64 + cslot] : 0.0;
    const double cw = 0.5 * a.h * a.colinfo[(size_t)s.slab * 32 + s.col] * wsr;      // h/2 eps ws[row]

    for (int n = 0; n < a.nsteps_chunk; ++n) {
        c.ring.begin_step(n);
        leak = fma(wdr, u * u, leak);      // trapezoidal part at t_n (src/evalobjgrad.jl:700)
        double un, v05, vN;
        const CqOps<NT> o = cq_load_ops<NT>(c);
        cq_state<NT>(c, a, o, cw, u, v, un, v05, vN);
        // Kp05 again: v(t+h) = v05 + c (K05 u_new + S05 v05)
        c.publish(un);
        v = c.template mm<false>(vN, o.Kp05);
        if (a.use_shift) v = fma(cw, un, v);
        u = un;
        leak += wdr * (u * u) + 2.0 * (wdr * (v05 * v05));      // (:716, penalf2a :2170-2180)
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
    c.ring.drain();
    st[s.foff] = u;
    st[(size_t)KT * 64 + s.foff] = v;
    const double tot = cq_wg_sum<NT>(leak, scratch, wave, lane_);
    if (slot0) st[(size_t)(JQ_STATE_ARRAYS * KT + JQ_MAXNC) * 64 + cslot] = tot;
}

// ---------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(64 * NT) void k_backward_cq(PropArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KT = 4 * NT;
    const CqSetup<NT> s = cq_setup<NT>(a);
    const int Nc = a.Ncoupled;
    // trace records: row slab * qps + qd (qps: quads of a full slab); a quad without columns inside that range (last slab)
    // contributes zeros, the others leave before any barrier
    const size_t trow = (size_t)s.slab * a.qps + s.qd;
    if (!s.active) {
        if (s.qd < a.qps)
            for (int k = threadIdx.x; k < a.nsteps_chunk * Nc * JQ_NTR; k += blockDim.x) a.traces[trow * a.nsteps_chunk * Nc * JQ_NTR + k] = 0.0;
        return;
    }
    const int lane_ = s.lane_, wave = s.wave;
    double* tab = (double*)(smem + a.lds_tab_off);
    for (int i = threadIdx.x; i < 32 * NT; i += blockDim.x) tab[(i & ~15) + 4 * (i & 3) + ((i >> 2) & 3)] = a.tabs[i];   // [block][g][r]
    CoopQ<NT> c;
    double* scratch = tab + 32 * NT + 2 * CoopQ<NT>::PAR;
    // per-step trace records rec[n & 1][wave][8 Nc] (see k_backward)
    const int ntr = Nc * JQ_NTR, rslots = 8 * Nc;
    double* rec = scratch + NT * 64;
    for (int i = threadIdx.x; i < 2 * NT * rslots; i += blockDim.x) rec[i] = 0.0;
    c.setup(tab + 32 * NT, wave, lane_);
    c.ring.init(smem, a, wave, lane_, NT);
    const double wdr = tab[16 * wave + s.g], wsr = tab[16 * NT + 16 * wave + s.g];

    double* st = a.state + (size_t)s.slab * a.state_stride;
    double u = st[s.foff], v = st[(size_t)KT * 64 + s.foff], mu = st[(size_t)2 * KT * 64 + s.foff], nb = st[(size_t)3 * KT * 64 + s.foff];
    const double cw = 0.5 * a.h * a.colinfo[(size_t)s.slab * 32 + s.col] * wsr;
    const double wgt = a.colinfo[(size_t)s.slab * 32 + 16 + s.col];
    const double cfw = (a.forced ? 0.5 * a.h * a.tinv : 0.0) * wdr;      // forcing weight c tinv wd[row]; 0 for step_no_forcing!
    const bool slot0 = wave == 0 && ((lane_ >> 2) & 3) == 0;
    const size_t cslot = 16 * (lane_ >> 4) + s.col;
    double carry[JQ_MAXNC];
#pragma unroll
    for (int q = 0; q < JQ_MAXNC; ++q) carry[q] = (q < Nc && slot0) ? st[(size_t)(JQ_STATE_ARRAYS * KT + q) * 64 + cslot] : 0.0;

    auto flush_traces = [&](int k) {
        if (wave == 0 && lane_ < ntr) {
            const int q = lane_ / JQ_NTR, kk = lane_ - q * JQ_NTR;
            const int slot = (kk == 0 ? 0 : kk == 2 ? 2 : 4 * Nc + (kk == 1 ? 0 : kk == 3 ? 2 : 1)) + 4 * q;
            const double* r = rec + (size_t)(k & 1) * NT * rslots + slot;
            double sum = r[0];
#pragma unroll
            for (int w = 1; w < NT; ++w) sum += r[w * rslots];
            a.traces[(trow * a.nsteps_chunk + k) * ntr + lane_] = sum;
        }
    };

    if (a.first_chunk) {
        // carry_q = tr(vr' Hsym_q lambdai) at t = T (see k_backward)
        c.publish(nb);
#pragma unroll
        for (int q = 0; q < JQ_MAXNC; ++q)
            if (q < Nc) carry[q] = -(u * c.mm_z_mode(c.ring.next_c(q), a.bw_trace[q]));
    }

    // One backward step = the state re-integration (chain 0, channel 0) and the adjoint step (chain 1, channel 1) side by side:
    // the adjoint products depend on the state step of the SAME step only through dot products and forcing terms
    // (un in tr3, tr4 and hr1; v05 in hi0), which are applied once those vectors exist -- so the two chains publish one vector
    // each per barrier: 6 + 2 m = 18 barriers per step (m = 6) instead of 17 + 19.
    for (int n = 0; n < a.nsteps_chunk; ++n) {
        c.ring.begin_step(n);
        if (n > 0) flush_traces(n - 1);      // (every wave has passed the barrier of begin_step since it finished step n-1)
        const CqOps<NT> o = cq_load_ops<NT>(c);
        // (u | nb): A = c K05 u, P = u + c S0 u  |  L = c K05 nb, Tn = c S05 nb
        c.publish2(u, nb);
        double A = c.template mm<true, 0>(0.0, o.Kp05);
        const double P = c.template mm<false, 0>(u, o.S0);
        double L = c.template mm<true, 1>(0.0, o.Kp05);
        const double Tn = c.template mm<true, 1>(0.0, o.S05);
        if (a.use_shift) {
            A = fma(cw, u, A);
            L = fma(cw, nb, L);
        }
        // (v | mu): A = c (K05 u + S05 v)  |  L = c (S0 mu - K05 li + hr0)
        c.publish2(v, mu);
        A = c.template mm<false, 0>(A, o.S05);
        L = c.template mm<false, 1>(L, o.S0);
        L = fma(cfw, u, L);      // u holds vr before the state step (:862)
        // Neumann series: v05 = v + sum_j S05^j A  |  X = mu + sum_j S0^j L
        double v05, X;
        c.horner2(v + A, A, o.S05, mu + L, L, o.S0, a.m, v05, X);
        // (v05 | X): vN = v05 + c S05 v05, un = u + c (S0 u - K0 v05), A = -c K1 v05  |  Hanti_q X (tr1, tr3), Hsym_q X (tr2),
        //            Lk = -c K0 X, Q = -c K1 X, SX = c S1 X
        c.publish2(v05, X);
        double vN = c.template mm<false, 0>(v05, o.S05);
        double un = c.template mm<false, 0>(P, o.Kn0);
        A = c.template mm<true, 0>(0.0, o.Kn1);
        double Lk = c.template mm<true, 1>(0.0, o.Kn0);
        double Q = c.template mm<true, 1>(0.0, o.Kn1);
        const double SX = c.template mm<true, 1>(0.0, o.S1);
        if (a.use_shift) {
            un = fma(-cw, v05, un);
            A = fma(-cw, v05, A);
            Lk = fma(-cw, X, Lk);
            Q = fma(-cw, X, Q);
        }
        double Tq[JQ_MAXNC], t2[JQ_MAXNC];
#pragma unroll
        for (int q = 0; q < JQ_MAXNC; ++q) {
            Tq[q] = t2[q] = 0.0;
            if (q < Nc) {
                Tq[q] = c.template mm_z_mode<1>(c.ring.next_c(Nc + q), a.bw_trace[q]);
                t2[q] = v05 * c.template mm_z_mode<1>(c.ring.next_c(q), a.bw_trace[q]);
            }
        }
        // Lk = -c l2 = -c (K0 X + S05 li + hi0) ; Q = -c (S05 (li + c l2) + K1 X + hi1)
        {
            const double Pn = fma(-cfw, v05, Tn);
            Lk += Pn;
            Q += Pn;
        }
        // (un | Lk): A = c (S1 un - K1 v05)  |  Q += c S05 Lk
        c.publish2(un, Lk);
        A = c.template mm<false, 0>(A, o.S1);
        Q = c.template mm<false, 1>(Q, o.S05);
        // Neumann series: un += sum_j S1^j A  |  nb_new = nb + Lk + sum_j S05^j Q
        double nbn;
        c.horner2(un + A, A, o.S1, (nb + Lk) + Q, Q, o.S05, a.m, un, nbn);
        const double Bq = nb + nbn;      // -(li0 + li)
        // early traces now that vr(t_n) exists: tr1 = tr(vr0' Hanti_q X), tr3 = tr(vr' Hanti_q X)
#pragma unroll
        for (int q = 0; q < JQ_MAXNC; ++q)
            if (q < Nc) {
                const double ts = wave_sum4(u * Tq[q] * wgt, un * Tq[q] * wgt, 0.0, 0.0);      // rows 0, 2: t1, t3
                if ((lane_ & 15) == 0) rec[((size_t)(n & 1) * NT + wave) * rslots + 4 * q + (lane_ >> 4)] = ts;
            }
        // (un | nb_new): vN = v05 + c (K05 un + S05 v05)  |  lambda_r_new = X + c (S1 X - K05 li_new + hr1), Hsym_q li_new (tr4)
        c.publish2(un, nbn);
        vN = c.template mm<false, 0>(vN, o.Kp05);
        double G = c.template mm<false, 1>(X, o.Kp05);
        if (a.use_shift) {
            vN = fma(cw, un, vN);
            G = fma(cw, nbn, G);
        }
        G = (G + SX) + cfw * un;
        double p4[JQ_MAXNC];
#pragma unroll
        for (int q = 0; q < JQ_MAXNC; ++q) p4[q] = (q < Nc) ? -(un * c.template mm_z_mode<1>(c.ring.next_c(q), a.bw_trace[q])) : 0.0;
        // (-(li0 + li)): tr5 = tr(vi05' Hanti (li0+li))
        c.publish(Bq);
#pragma unroll
        for (int q = 0; q < JQ_MAXNC; ++q)
            if (q < Nc) {
                const double t5 = -(v05 * c.template mm_z_mode<0>(c.ring.next_c(Nc + q), a.bw_trace[q]));
                const double t4 = p4[q] + carry[q];
                carry[q] = p4[q];
                const double ts = wave_sum4(t2[q] * wgt, t4 * wgt, t5 * wgt, 0.0);      // rows 0, 2, 1: t2, t4, t5
                if ((lane_ & 15) == 0) rec[((size_t)(n & 1) * NT + wave) * rslots + 4 * (Nc + q) + (lane_ >> 4)] = ts;
            }
        u = un;
        v = vN;
        mu = G;
        nb = nbn;
    }
    c.ring.drain();
    flush_traces(a.nsteps_chunk - 1);
    st[s.foff] = u;
    st[(size_t)KT * 64 + s.foff] = v;
    st[(size_t)2 * KT * 64 + s.foff] = mu;
    st[(size_t)3 * KT * 64 + s.foff] = nb;
#pragma unroll
    for (int q = 0; q < JQ_MAXNC; ++q)
        if (q < Nc) {
            const double tot = cq_wg_sum<NT>(carry[q], scratch, wave, lane_);
            if (slot0) st[(size_t)(JQ_STATE_ARRAYS * KT + q) * 64 + cslot] = tot;
        }
}
