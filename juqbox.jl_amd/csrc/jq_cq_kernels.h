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
// K0 v05 and K1 v05; the five products with X, ...): 17 publications per forward step (20 products) at m = 6 Neumann terms.
// The backward sweep runs its two chains -- state re-integration and adjoint step -- on TWO SETS of NT waves (2 NT waves
// per workgroup): the adjoint products depend on the state step of the SAME time step only through dot products and forcing
// terms (u, v05, un of the wave's own block), which the adjoint wave reads from the state wave's publications; both chains
// publish at the same barriers: 6 + 2 m = 18 per backward step (52 products).  A wave alone on its SIMD issues one VALU
// instruction every ~10 cycles (probes/dp_rate_probe.hip), so the split halves the critical path: each wave executes one
// chain's instructions, and the SIMDs interleave the two sets.
// Same operators, images, window staging (Ring, batch < 0), state file, trace records and reductions as the quad-layout
// kernels; the regrouping only reorders floating-point additions.
#pragma once
#include "jq_kernels.h"

// Exchange image in LDS: [2 parities][2 channels][NT + 2 blocks][64] doubles -- a zero block in front of and behind the NT
// blocks of a channel, so that the neighbours of the edge blocks need no clamping and every access is ONE base register
// (this wave's block of ITS channel in the current parity) plus a compile-time offset.  Channel 0: forward sweep / state
// chain of the backward sweep; channel 1: adjoint chain.
// Window staging (jq_kernels.h, Ring with batch < 0) without the code of the other staging modes: a ring of JQ_WIN_TPS time points
// (K and S image each) and the constant trace images are resident in LDS; step n works on the time points 2n, 2n+1, 2n+2 while
// 2n+3 and 2n+4 stream in (global -> LDS DMA, the pieces of an image pair spread over the waves); ONE barrier per time step.
// These kernels run one wave per SIMD or little more: every instruction of the loop -- scalar ones and taken branches
// included -- is on the critical path, so the cursor is incremental (no multiplications, no modulo, no mode branches).
struct WinRing {
    char* smem;
    const char* gnext;      // global address of the next time point to fetch
    unsigned stride_b;      // bytes per image
    unsigned slot_bytes;    // bytes per time point (K and S image)
    unsigned cbase;         // byte offset of the constant images
    int pieces2;            // 1 KiB pieces of a time point
    int jnext, jlast;       // next time point to fetch, last one of the chunk
    unsigned snext;         // byte offset of its ring slot
    int wave, nwaves, lane;
    unsigned wb0, wb1, wb2; // byte offsets of the time points 2n, 2n+1, 2n+2 of the current step

    __device__ __forceinline__ void dma(const char* gsrc, char* dst, int pieces) const
    {
        unsigned lo;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lo));
        const char* src = gsrc + lo * 16u;
        for (int p = wave; p < pieces; p += nwaves)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)p * 1024),
                                             (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 0);
    }
    __device__ __forceinline__ void issue_next()
    {
        if (jnext > jlast) return;
        dma(gnext, smem + snext, pieces2);
        gnext += slot_bytes;
        ++jnext;
        snext += slot_bytes;
        if (snext == JQ_WIN_TPS * slot_bytes) snext = 0;
    }
    __device__ __forceinline__ void init(char* smem_, const PropArgs& a, int wave_, int lane_, int nwaves_)
    {
        smem = smem_, wave = wave_, lane = lane_, nwaves = nwaves_;
        stride_b = (unsigned)(a.stride * 8);
        slot_bytes = 2 * stride_b;
        cbase = JQ_WIN_TPS * slot_bytes;
        pieces2 = 2 * a.pieces;
        gnext = (const char*)a.stream;
        jnext = 0, jlast = 2 * a.nsteps_chunk, snext = 0;
        dma((const char*)a.cimg, smem + cbase, 2 * a.Ncoupled * a.pieces);
        for (int j = 0; j < JQ_WIN_TPS; ++j) issue_next();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        wb0 = 0, wb1 = slot_bytes, wb2 = 2 * slot_bytes;
    }
    __device__ __forceinline__ void begin_step(int n)
    {
        if (n == 0) return;
        // every wave has finished step n-1 behind this barrier: its time points 2n-2, 2n-1 make room for 2n+3, 2n+4; the images
        // of this step (issued one step ago) have landed
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue_next();
        issue_next();
        wb0 = wb2;
        wb1 = wb0 + slot_bytes;
        if (wb1 == cbase) wb1 = 0;
        wb2 = wb1 + slot_bytes;
        if (wb2 == cbase) wb2 = 0;
    }
    // LDS image (lane offset applied) of K (KIND 0) / S (KIND 1) at time point 2n + TP, of constant image #idx
    template <int KIND, int TP>
    __device__ __forceinline__ const double* next_ks() const
    {
        return (const double*)(smem + ((TP == 0 ? wb0 : TP == 1 ? wb1 : wb2) + KIND * stride_b)) + lane;
    }
    __device__ __forceinline__ const double* next_c(int idx) const { return (const double*)(smem + (cbase + (unsigned)idx * stride_b)) + lane; }
    __device__ __forceinline__ void drain()
    {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
};

typedef __attribute__((address_space(3))) double jq_lds_double;
template <int NT>
struct CoopQ {
    static constexpr int CHS = (NT + 2) * 64;       // doubles per channel
    static constexpr int PAR = 2 * CHS;              // doubles per parity
    WinRing ring;
    jq_lds_double* xp;  // my block of my channel in the parity that holds the published vectors
    int delta;          // doubles from that parity to the other one (+-PAR)
    int mt;             // my block
    int lane;
    double xown;        // my block of the published x

    __device__ __forceinline__ void setup(double* xbuf, int blk, int lane_, int ch = 0)
    {
        mt = blk, lane = lane_, xown = 0.0;
        for (int i = threadIdx.x; i < 2 * PAR; i += blockDim.x) xbuf[i] = 0.0;      // (the pads stay zero)
        xp = (jq_lds_double*)(xbuf + ch * CHS + (1 + blk) * 64 + lane_);
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
    // my block of the vector that the OTHER chain published at the last barrier (OFF = +-CHS: where its channel is)
    template <int OFF>
    __device__ __forceinline__ double other() const
    {
        return xp[OFF];
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
    template <bool ZEROC>
    __device__ __forceinline__ double mm(double C, const Op& o) const
    {
        const double xb = xp[-64];
        const double xa = xp[64];
        double acc = ZEROC ? 0.0 : C;
        acc = __builtin_amdgcn_mfma_f64_4x4x4f64(o.a, xown, acc, 0, 0, 0);
        acc = fma(o.c[0], row_shift4<0x114>(xown), acc);
        acc = fma(o.c[1], row_shift4<0x104>(xown), acc);
        acc = fma(o.c[2], xb, acc);      // (the coefficients of a missing neighbour are zero)
        return fma(o.c[3], xa, acc);
    }
    // C + M x  for the published x (M: LDS image with the lane offset applied; MODE: JQ_T4_* parts that are non-zero)
    template <bool ZEROC, int MODE = JQ_T4_DIAG | JQ_T4_RTERMS | JQ_T4_MTERMS>
    __device__ __forceinline__ double mm(double C, const double* M) const
    {
        constexpr bool diag = MODE & JQ_T4_DIAG, rt = MODE & JQ_T4_RTERMS, mtm = MODE & JQ_T4_MTERMS;
        double xb = 0.0, xa = 0.0;
        if constexpr (mtm) {      // neighbour blocks first: their LDS latency hides behind this block's own work
            xb = xp[-64];
            xa = xp[64];
        }
        double acc = ZEROC ? 0.0 : C;
        d4 c = {0.0, 0.0, 0.0, 0.0};
        if constexpr (rt || mtm) c = t4q_cload(t4q_c<NT>(M, lane), mt);
        if constexpr (diag) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(M[mt * 64], xown, acc, 0, 0, 0);
        if constexpr (rt) {
            acc = fma(c[0], row_shift4<0x114>(xown), acc);
            acc = fma(c[1], row_shift4<0x104>(xown), acc);
        }
        if constexpr (mtm) {      // (the coefficients of a missing neighbour are zero)
            acc = fma(c[2], xb, acc);
            acc = fma(c[3], xa, acc);
        }
        return acc;
    }
    // trace operators touch one part of the image only (a.bw_trace: JQ_T4_* bits); anything else takes the full product
    __device__ __forceinline__ double mm_z_mode(const double* M, int mode) const
    {
#ifdef JQ_CQ_NOTRACE       // timing experiment only (wrong gradients): what the trace products cost
        return xown;
#endif
#ifdef JQ_CQ_FULLTRACE     // branch-free: the absent parts of an image are stored as zeros
        return mm<true>(0.0, M);
#endif
        switch (mode) {
        case JQ_T4_DIAG: return mm<true, JQ_T4_DIAG>(0.0, M);
        case JQ_T4_RTERMS: return mm<true, JQ_T4_RTERMS>(0.0, M);
        case JQ_T4_MTERMS: return mm<true, JQ_T4_MTERMS>(0.0, M);
        default: return mm<true>(0.0, M);
        }
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

// sum of val over the nwaves waves of the workgroup (wave order) and the four 4-row groups of a block, for the lanes with
// group 0; valid in wave 0.  scratch: LDS [nwaves][64].  Contains workgroup barriers.
__device__ __forceinline__ double cq_wg_sum(double val, double* scratch, int wave, int lane, int nwaves)
{
    val = row_ror_add<8>(row_ror_add<4>(val));
    __syncthreads();
    scratch[wave * 64 + lane] = val;
    __syncthreads();
    double s = 0.0;
    if (wave == 0)
        for (int w = 0; w < nwaves; ++w) s += scratch[w * 64 + lane];
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
    int lane_, wave, chain, qd, slab, col, g;      // wave: my block; chain: 0 = forward sweep / state chain, 1 = adjoint chain
    bool active;
    size_t foff;        // offset of my element in an array image of the slab file
};
template <int NT>
__device__ __forceinline__ CqSetup<NT> cq_setup(const PropArgs& a)
{
    CqSetup<NT> s;
    s.lane_ = threadIdx.x & 63;
    s.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    s.chain = s.wave >= NT ? 1 : 0;
    s.wave -= s.chain * NT;
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
    const double tot = cq_wg_sum(leak, scratch, wave, lane_, NT);
    if (slot0) st[(size_t)(JQ_STATE_ARRAYS * KT + JQ_MAXNC) * 64 + cslot] = tot;
}

// ---------------------------------------------------------------------------------------------
// grid = 4 * nslabs, block = 128 * NT: waves 0 .. NT-1 re-integrate the state (channel 0), waves NT .. 2 NT-1 run the adjoint
// step and the traces (channel 1).  Both sets pass the same barriers: begin_step, then 6 + 2 m publications per time step
//   (u | nb)  (v | mu)  m x Neumann  (v05 | X)  (un' | Lk)  m x Neumann  (un | nb_new)  (- | -(li0 + li))
template <int NT>
__global__ __launch_bounds__(128 * NT) void k_backward_cq(PropArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KT = 4 * NT;
    constexpr int CH = CoopQ<NT>::CHS;
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
    const int lane_ = s.lane_, wave = s.wave, wave_all = s.wave + NT * s.chain;
    double* tab = (double*)(smem + a.lds_tab_off);
    for (int i = threadIdx.x; i < 32 * NT; i += blockDim.x) tab[(i & ~15) + 4 * (i & 3) + ((i >> 2) & 3)] = a.tabs[i];   // [block][g][r]
    CoopQ<NT> c;
    double* scratch = tab + 32 * NT + 2 * CoopQ<NT>::PAR;      // [2 NT][64]
    // per-step trace records rec[n & 1][block][8 Nc] of the adjoint waves (see k_backward)
    const int ntr = Nc * JQ_NTR, rslots = 8 * Nc;
    double* rec = scratch + 2 * NT * 64;
    for (int i = threadIdx.x; i < 2 * NT * rslots; i += blockDim.x) rec[i] = 0.0;
    c.setup(tab + 32 * NT, wave, lane_, s.chain);
    c.ring.init(smem, a, wave_all, lane_, 2 * NT);
    const double wdr = tab[16 * wave + s.g], wsr = tab[16 * NT + 16 * wave + s.g];
    double* st = a.state + (size_t)s.slab * a.state_stride;
    const double cw = 0.5 * a.h * a.colinfo[(size_t)s.slab * 32 + s.col] * wsr;      // h/2 eps ws[row]
    const size_t cslot = 16 * (lane_ >> 4) + s.col;
    double carry[JQ_MAXNC];
#pragma unroll
    for (int q = 0; q < JQ_MAXNC; ++q) carry[q] = 0.0;

    if (s.chain == 0) {
        // ---- state re-integration (src/evalobjgrad.jl:879), channel 0 ------------------------------------------------
        double u = st[s.foff], v = st[(size_t)KT * 64 + s.foff];
        if (a.first_chunk) c.publish(u);      // (vr(T) for the carry products of the adjoint waves)
        for (int n = 0; n < a.nsteps_chunk; ++n) {
            c.ring.begin_step(n);
            // (every wave has passed the barrier of begin_step since it finished step n-1)
            if (n > 0 && wave == 0 && lane_ < ntr) {
                const int k = n - 1, q = lane_ / JQ_NTR, kk = lane_ - q * JQ_NTR;
                const int slot = (kk == 0 ? 0 : kk == 2 ? 2 : 4 * Nc + (kk == 1 ? 0 : kk == 3 ? 2 : 1)) + 4 * q;
                const double* r = rec + (size_t)(k & 1) * NT * rslots + slot;
                double sum = r[0];
#pragma unroll
                for (int w = 1; w < NT; ++w) sum += r[w * rslots];
                a.traces[(trow * a.nsteps_chunk + k) * ntr + lane_] = sum;
            }
            double un, v05, vN;
            const CqOps<NT> o = cq_load_ops<NT>(c);
            cq_state<NT>(c, a, o, cw, u, v, un, v05, vN);
            // x = un: v(t_n) = v05 + c (K05 un + S05 v05)
            c.publish(un);
            v = c.template mm<false>(vN, o.Kp05);
            if (a.use_shift) v = fma(cw, un, v);
            u = un;
            c.flip();      // (the adjoint waves' last publication of the step)
        }
        c.ring.drain();
        if (wave == 0 && lane_ < ntr) {
            const int k = a.nsteps_chunk - 1, q = lane_ / JQ_NTR, kk = lane_ - q * JQ_NTR;
            const int slot = (kk == 0 ? 0 : kk == 2 ? 2 : 4 * Nc + (kk == 1 ? 0 : kk == 3 ? 2 : 1)) + 4 * q;
            const double* r = rec + (size_t)(k & 1) * NT * rslots + slot;
            double sum = r[0];
#pragma unroll
            for (int w = 1; w < NT; ++w) sum += r[w * rslots];
            a.traces[(trow * a.nsteps_chunk + k) * ntr + lane_] = sum;
        }
        st[s.foff] = u;
        st[(size_t)KT * 64 + s.foff] = v;
    } else {
        // ---- adjoint step! with forcing (src/StormerVerlet.jl:255-303) and the traces of adjoint_grad_calc!, channel 1 ----
        double mu = st[(size_t)2 * KT * 64 + s.foff], nb = st[(size_t)3 * KT * 64 + s.foff];
        const double wgt = a.colinfo[(size_t)s.slab * 32 + 16 + s.col];
        const double cfw = (a.forced ? 0.5 * a.h * a.tinv : 0.0) * wdr;      // forcing weight c tinv wd[row]; 0 for step_no_forcing!
        const bool slot0 = wave == 0 && ((lane_ >> 2) & 3) == 0;      // the lanes that carry per-column partials between chunks
#pragma unroll
        for (int q = 0; q < JQ_MAXNC; ++q)
            if (q < Nc && slot0) carry[q] = st[(size_t)(JQ_STATE_ARRAYS * KT + q) * 64 + cslot];
        if (a.first_chunk) {
            // carry_q = tr(vr' Hsym_q lambdai) at t = T (see k_backward)
            c.publish(nb);
            const double u0 = c.template other<-CH>();
#pragma unroll
            for (int q = 0; q < JQ_MAXNC; ++q)
                if (q < Nc) carry[q] = -(u0 * c.mm_z_mode(c.ring.next_c(q), a.bw_trace[q]));
        }
        for (int n = 0; n < a.nsteps_chunk; ++n) {
            c.ring.begin_step(n);
            const CqOps<NT> o = cq_load_ops<NT>(c);
            // x = nb (-lambda_i): L = c K05 nb, Tn = c S05 nb (for the second half of the step)
            c.publish(nb);
            const double u = c.template other<-CH>();      // vr before the state step (:862)
            double L = c.template mm<true>(0.0, o.Kp05);
            const double Tn = c.template mm<true>(0.0, o.S05);
            if (a.use_shift) L = fma(cw, nb, L);
            // x = mu: L = c (S0 mu - K05 li + hr0) ; X = mu + sum_j S0^j L
            c.publish(mu);
            L = c.template mm<false>(L, o.S0);
            L = fma(cfw, u, L);
            const double X = c.horner(mu + L, L, o.S0, a.m);
            // x = X: Lk = -c K0 X, Q = -c K1 X, SX = c S1 X, Hanti_q X (tr1, tr3), Hsym_q X (tr2)
            c.publish(X);
            const double v05 = c.template other<-CH>();
            double Lk = c.template mm<true>(0.0, o.Kn0);
            double Q = c.template mm<true>(0.0, o.Kn1);
            const double SX = c.template mm<true>(0.0, o.S1);
            if (a.use_shift) {
                Lk = fma(-cw, X, Lk);
                Q = fma(-cw, X, Q);
            }
            double Tq[JQ_MAXNC], t2[JQ_MAXNC];
#pragma unroll
            for (int q = 0; q < JQ_MAXNC; ++q) {
                Tq[q] = t2[q] = 0.0;
                if (q < Nc) {
                    Tq[q] = c.mm_z_mode(c.ring.next_c(Nc + q), a.bw_trace[q]);
                    t2[q] = v05 * c.mm_z_mode(c.ring.next_c(q), a.bw_trace[q]);
                }
            }
            // Lk = -c l2 = -c (K0 X + S05 li + hi0) ; Q = -c (S05 (li + c l2) + K1 X + hi1)
            {
                const double Pn = fma(-cfw, v05, Tn);
                Lk += Pn;
                Q += Pn;
            }
            // x = Lk: Q += c S05 Lk ; nb_new = nb + Lk + sum_j S05^j Q
            c.publish(Lk);
            Q = c.template mm<false>(Q, o.S05);
            const double nbn = c.horner((nb + Lk) + Q, Q, o.S05, a.m);
            const double Bq = nb + nbn;      // -(li0 + li)
            // x = nb_new: lambda_r_new = X + c (S1 X - K05 li_new + hr1), Hsym_q li_new (tr4); vr(t_n) exists now: tr1, tr3
            c.publish(nbn);
            const double un = c.template other<-CH>();
            double G = c.template mm<false>(X, o.Kp05);
            if (a.use_shift) G = fma(cw, nbn, G);
            G = (G + SX) + cfw * un;
#pragma unroll
            for (int q = 0; q < JQ_MAXNC; ++q)
                if (q < Nc) {
                    const double ts = wave_sum4(u * Tq[q] * wgt, un * Tq[q] * wgt, 0.0, 0.0);      // rows 0, 2: t1 = tr(vr0' Hanti_q X), t3 = tr(vr' Hanti_q X)
                    if ((lane_ & 15) == 0) rec[((size_t)(n & 1) * NT + wave) * rslots + 4 * q + (lane_ >> 4)] = ts;
                }
            double p4[JQ_MAXNC];
#pragma unroll
            for (int q = 0; q < JQ_MAXNC; ++q) p4[q] = (q < Nc) ? -(un * c.mm_z_mode(c.ring.next_c(q), a.bw_trace[q])) : 0.0;
            // x = -(li0 + li): tr5 = tr(vi05' Hanti (li0+li))
            c.publish(Bq);
#pragma unroll
            for (int q = 0; q < JQ_MAXNC; ++q)
                if (q < Nc) {
                    const double t5 = -(v05 * c.mm_z_mode(c.ring.next_c(Nc + q), a.bw_trace[q]));
                    const double t4 = p4[q] + carry[q];
                    carry[q] = p4[q];
                    const double ts = wave_sum4(t2[q] * wgt, t4 * wgt, t5 * wgt, 0.0);      // rows 0, 2, 1: t2, t4, t5
                    if ((lane_ & 15) == 0) rec[((size_t)(n & 1) * NT + wave) * rslots + 4 * (Nc + q) + (lane_ >> 4)] = ts;
                }
            mu = G;
            nb = nbn;
        }
        c.ring.drain();
        st[(size_t)2 * KT * 64 + s.foff] = mu;
        st[(size_t)3 * KT * 64 + s.foff] = nb;
    }
#pragma unroll
    for (int q = 0; q < JQ_MAXNC; ++q)
        if (q < Nc) {
            const double tot = cq_wg_sum(carry[q], scratch, wave_all, lane_, 2 * NT);
            if (wave_all == 0 && ((lane_ >> 2) & 3) == 0) st[(size_t)(JQ_STATE_ARRAYS * KT + q) * 64 + cslot] = tot;
        }
}
