// jq_lane_kernels.h -- propagators for SMALL Hilbert spaces (Ntot <= 12): one LANE per state column.
//
// For Ntot <= 16 an MFMA tile is mostly padding (SWAP-02: 4 of 16 rows and 4 of 16 k) and every product is a
// ~600-cycle dependent chain (LDS round trip + 4 dependent 64-cycle MFMAs).  Here each lane owns one column
// (ensemble sample x initial condition): its state vectors are NP doubles in registers, the operators are
// uniform across the wave and are read with SCALAR loads (s_load through the scalar cache) straight into
// the SGPR operand of v_fma_f64, so a product is NP*NP FMAs and nothing else -- no LDS, no barriers, no
// cross-lane traffic except the per-step trace reduction.  Same math (scaled/signed operator stream,
// Horner-form Neumann series, negated lambda_i, 4 trace products per control), same schedule of time
// points, same trace/gradient pipeline as the MFMA kernels.
//
// Layouts:  operators: plain row-major NP x NP images (zero padded), stream point j -> K at (2j)*NP*NP,
//           S at (2j+1)*NP*NP;  state file: [array][row][column] with the column index fastest.
#pragma once
#include "jq_kernels.h"

// Operator/table pointers in the CONSTANT address space: the images are written by earlier kernels
// (k_stream) and never by the propagators, and a uniform load from address space 4 is always selected
// as s_load (a plain global pointer is not: the kernel also stores to global memory, so the compiler
// cannot prove the operator bytes unclobbered and falls back to per-lane global_load into VGPRs).
typedef const __attribute__((address_space(4))) double* cmat_t;
__device__ __forceinline__ cmat_t as_const(const double* p) { return (cmat_t)(unsigned long long)p; }

// Opaque copy of a table pointer: keeps the compiler from hoisting the (loop invariant) table loads out
// of the time loop, where 2*NP doubles of wd/ws would permanently occupy up to 48 of the ~100 SGPRs.
__device__ __forceinline__ cmat_t launder(cmat_t p)
{
    unsigned long long v = (unsigned long long)p;
    asm volatile("" : "+s"(v));
    return (cmat_t)v;
}

template <int NP>
struct Vec {
    double e[NP];
};

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
// Scalar-load pipeline of one operator image.  The image is read as a flat stream of CH-double chunks
// (s_load_dwordx16 / x8) into SGPR tuples that feed v_fma_f64 directly; chunk c+1 is in flight while the
// FMAs of chunk c run.  Inline asm because hipcc, left alone, hoists every load of a matrix (and of the
// next matrices) to the top of the block and then spills hundreds of SGPRs lane by lane into VGPRs.
// SMEM returns out of order, so the only usable wait is lgkmcnt(0); the wait takes the landed tuple as
// an in/out operand so that no consumer can be scheduled above it.
typedef double sd8 __attribute__((ext_vector_type(8)));
typedef double sd4 __attribute__((ext_vector_type(4)));
template <int CH> struct SChunk;
template <> struct SChunk<8> {
    typedef sd8 type;
    template <int OFF>
    static __device__ __forceinline__ void load(sd8& r, cmat_t p) { asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(r) : "s"(p), "i"(OFF)); }
};
template <> struct SChunk<4> {
    typedef sd4 type;
    template <int OFF>
    static __device__ __forceinline__ void load(sd4& r, cmat_t p) { asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(r) : "s"(p), "i"(OFF)); }
};
template <typename T>
__device__ __forceinline__ void s_landed(T& r)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r));
}

// chunk C of the image: wait for it, issue chunk C+1 (byte offset as an instruction immediate: no pointer
// arithmetic for the compiler to keep alive), run its CH FMAs, recurse with the buffers swapped
template <int NP, int CH, int C, int NCH>
struct MvStep {
    typedef typename SChunk<CH>::type chunk_t;
    static __device__ __forceinline__ void run(Vec<NP>& y, cmat_t M, const Vec<NP>& x, chunk_t& cur, chunk_t& nxt)
    {
        s_landed(cur);
        if constexpr (C + 1 < NCH) SChunk<CH>::template load<(C + 1) * CH * 8>(nxt, M);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int e = C * CH + k;
            y.e[e / NP] = fma(cur[k], x.e[e % NP], y.e[e / NP]);
        }
        __builtin_amdgcn_sched_barrier(0);   // the tuple dies here, its SGPRs are reused by chunk C+2
        if constexpr (C + 1 < NCH) MvStep<NP, CH, C + 1, NCH>::run(y, M, x, nxt, cur);
    }
};

// y = c + M x   (M uniform, row-major NP x NP; y may alias c, not x)
template <int NP, bool ZEROC>
__device__ __forceinline__ Vec<NP> mv(const Vec<NP>& c, cmat_t M, const Vec<NP>& x)
{
    constexpr int CH = (NP * NP % 8 == 0) ? 8 : 4;
    constexpr int NCH = NP * NP / CH;
    static_assert(NP * NP % CH == 0, "NP*NP must be a multiple of 4");
    typedef typename SChunk<CH>::type chunk_t;
    Vec<NP> y;
#pragma unroll
    for (int i = 0; i < NP; ++i) y.e[i] = ZEROC ? 0.0 : c.e[i];
    chunk_t b0, b1;
    SChunk<CH>::template load<0>(b0, M);
    MvStep<NP, CH, 0, NCH>::run(y, M, x, b0, b1);
    // pin the result here: otherwise LLVM sinks the FMAs of a product whose result is only needed in a
    // later basic block below the (volatile) loads and keeps every loaded tuple alive by spilling it
#pragma unroll
    for (int i = 0; i < NP; ++i) asm volatile("" : "+v"(y.e[i]));
    return y;
}
// y += (s * tab) .* x     (tab uniform)
template <int NP>
__device__ __forceinline__ void v_axpy_rows(Vec<NP>& y, double s, cmat_t tab0, const Vec<NP>& x)
{
    cmat_t tab = launder(tab0);
#pragma unroll
    for (int i = 0; i < NP; ++i) y.e[i] = fma(s * tab[i], x.e[i], y.e[i]);
}
template <int NP>
__device__ __forceinline__ double v_wsq(cmat_t tab0, const Vec<NP>& x)
{
    cmat_t tab = launder(tab0);
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NP; ++i) s = fma(tab[i], x.e[i] * x.e[i], s);
    return s;
}
// bpa + sum_{j=1..m} S^j A  (Horner form, see jq_kernels.h)
template <int NP>
__device__ __forceinline__ Vec<NP> lane_horner(const Vec<NP>& bpa, const Vec<NP>& A, cmat_t S, int m)
{
    if (m <= 0) return bpa;
    Vec<NP> Y = A;
    for (int j = 1; j < m; ++j) Y = mv<NP, false>(A, S, Y);
    return mv<NP, false>(bpa, S, Y);
}

struct LaneOps {
    cmat_t Kp05, S05, Kn0, S0, Kn1, S1;
};
__device__ __forceinline__ LaneOps lane_ops(const PropArgs& a, int n, int nn)
{
    LaneOps o;
    cmat_t s = as_const(a.stream);
    o.Kn0 = s + (size_t)(2 * (2 * n)) * nn;
    o.S0 = o.Kn0 + nn;
    o.Kp05 = s + (size_t)(2 * (2 * n + 1)) * nn;
    o.S05 = o.Kp05 + nn;
    o.Kn1 = s + (size_t)(2 * (2 * n + 2)) * nn;
    o.S1 = o.Kn1 + nn;
    return o;
}

// one Stormer-Verlet state step (forward step!, src/StormerVerlet.jl:461-504) in the accumulate form
template <int NP>
__device__ __forceinline__ void lane_state(const PropArgs& a, const LaneOps& o, double ceps, cmat_t ws,
                                           const Vec<NP>& u, const Vec<NP>& v, Vec<NP>& un, Vec<NP>& v05, Vec<NP>& vnew)
{
    Vec<NP> A = mv<NP, true>(u, o.Kp05, u);
    if (a.use_shift) v_axpy_rows(A, ceps, ws, u);
    A = mv<NP, false>(A, o.S05, v);
    v05 = lane_horner<NP>(v_add(v, A), A, o.S05, a.m);
    Vec<NP> vN = mv<NP, false>(v05, o.S05, v05);
    un = mv<NP, false>(u, o.Kn0, v05);
    if (a.use_shift) v_axpy_rows(un, -ceps, ws, v05);
    un = mv<NP, false>(un, o.S0, u);
    A = mv<NP, true>(u, o.Kn1, v05);
    if (a.use_shift) v_axpy_rows(A, -ceps, ws, v05);
    A = mv<NP, false>(A, o.S1, un);
    un = lane_horner<NP>(v_add(un, A), A, o.S1, a.m);
    vnew = mv<NP, false>(vN, o.Kp05, un);
    if (a.use_shift) v_axpy_rows(vnew, ceps, ws, un);
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
    cmat_t wd = as_const(a.tabs);
    cmat_t ws = wd + NP;
    Vec<NP> u = lane_load<NP>(a.state, 0, ncols, col), v = lane_load<NP>(a.state, 1, ncols, col);
    double leak = a.state[((size_t)JQ_LANE_ARRAYS * NP + JQ_MAXNC) * ncols + col];
    const double ceps = 0.5 * a.h * a.colinfo[col];
    for (int n = 0; n < a.nsteps_chunk; ++n) {
        const LaneOps o = lane_ops(a, n, (int)a.stride);
        Vec<NP> un, v05, vnew;
        leak += v_wsq<NP>(wd, u);
        lane_state<NP>(a, o, ceps, ws, u, v, un, v05, vnew);
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
    }
    lane_store<NP>(a.state, 0, ncols, col, u);
    lane_store<NP>(a.state, 1, ncols, col, v);
    a.state[((size_t)JQ_LANE_ARRAYS * NP + JQ_MAXNC) * ncols + col] = leak;
}

// Backward sweep of one chunk (state re-integration, adjoint step, trace scalars per wave and step).
// a.cimg: constant images [Hsym_q | Hanti_q], NP*NP each.
template <int NP>
__global__ __launch_bounds__(64) void k_backward_lane(PropArgs a)
{
    const long long col = (long long)blockIdx.x * 64 + threadIdx.x;
    const long long ncols = a.nslabs;
    const int lane = threadIdx.x;
    const int Nc = a.Ncoupled;
    const int NN = (int)a.stride;   // doubles per operator image (NP*NP padded to 64 B)
    cmat_t cimg = as_const(a.cimg);
    cmat_t wd = as_const(a.tabs);
    cmat_t ws = wd + NP;
    Vec<NP> u = lane_load<NP>(a.state, 0, ncols, col), v = lane_load<NP>(a.state, 1, ncols, col);
    Vec<NP> mu = lane_load<NP>(a.state, 2, ncols, col), nb = lane_load<NP>(a.state, 3, ncols, col);
    const double ceps = 0.5 * a.h * a.colinfo[col];
    const double wgt = a.colinfo[ncols + col];
    const double cfw = a.forced ? 0.5 * a.h * a.tinv : 0.0;
    double carry[JQ_MAXNC];
#pragma unroll
    for (int q = 0; q < JQ_MAXNC; ++q) carry[q] = (q < Nc) ? a.state[((size_t)JQ_LANE_ARRAYS * NP + q) * ncols + col] : 0.0;
    double* trw = a.traces + ((size_t)blockIdx.x * a.nsteps_chunk) * (Nc * JQ_NTR);

    if (a.first_chunk) {
#pragma unroll
        for (int q = 0; q < JQ_MAXNC; ++q)
            if (q < Nc) carry[q] = -v_dot(u, mv<NP, true>(u, cimg + (size_t)q * NN, nb));
    }

    for (int n = 0; n < a.nsteps_chunk; ++n) {
        const LaneOps o = lane_ops(a, n, NN);
        Vec<NP> un, v05, vnew;
        lane_state<NP>(a, o, ceps, ws, u, v, un, v05, vnew);
        // adjoint step! (src/StormerVerlet.jl:255-303) with nb = -lambda_i, see k_backward
        Vec<NP> R = mv<NP, true>(u, o.Kp05, nb);
        if (a.use_shift) v_axpy_rows(R, ceps, ws, nb);
        R = mv<NP, false>(R, o.S0, mu);
        v_axpy_rows(R, cfw, wd, u);
        const Vec<NP> X = lane_horner<NP>(v_add(mu, R), R, o.S0, a.m);
        Vec<NP> L = mv<NP, true>(u, o.Kn0, X);
        if (a.use_shift) v_axpy_rows(L, -ceps, ws, X);
        Vec<NP> Qv = mv<NP, true>(u, o.Kn1, X);
        if (a.use_shift) v_axpy_rows(Qv, -ceps, ws, X);
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
        if (a.use_shift) v_axpy_rows(G, ceps, ws, nbn);
        G = mv<NP, false>(G, o.S1, X);
        v_axpy_rows(G, cfw, wd, un);
        // traces (adjoint_grad_calc!, src/evalobjgrad.jl:2581-2618), weighted and summed over the wave
#pragma unroll
        for (int q = 0; q < JQ_MAXNC; ++q) {
            if (q < Nc) {
                cmat_t Hs = cimg + (size_t)q * NN;
                cmat_t Ha = cimg + (size_t)(Nc + q) * NN;
                const Vec<NP> HaX = mv<NP, true>(u, Ha, X);
                const double t1 = wave_sum(v_dot(u, HaX) * wgt);
                const double t3 = wave_sum(v_dot(un, HaX) * wgt);
                const double t5 = wave_sum(-v_dot(v05, mv<NP, true>(u, Ha, Bq)) * wgt);
                const double t2 = wave_sum(v_dot(v05, mv<NP, true>(u, Hs, X)) * wgt);
                const double p4 = -v_dot(un, mv<NP, true>(u, Hs, nbn));
                const double t4 = wave_sum((p4 + carry[q]) * wgt);
                carry[q] = p4;
                if (lane == 0) {
                    double* tr = trw + (size_t)n * (Nc * JQ_NTR) + q * JQ_NTR;
                    tr[0] = t1;
                    tr[1] = t2;
                    tr[2] = t3;
                    tr[3] = t4;
                    tr[4] = t5;
                }
            }
        }
        u = un;
        v = vnew;
        mu = G;
        nb = nbn;
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
