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
//     tile images ("tile stream", k_stream below), and double-buffered through LDS with direct
//     global->LDS DMA (global_load_lds_dwordx4) one matrix ahead of the MFMAs.
//   * the per-sample perturbation  H0_s = H0 + eps_s diag(shift)  of the risk-neutral ensemble
//     (src/ipopt_interface.jl:41-44) is applied in registers after each K product.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double d4 __attribute__((ext_vector_type(4)));

#define JQ_WAVES 4            // waves (slabs) per workgroup
#define JQ_MAXNC 4            // max coupled controls supported by the trace/carry bookkeeping
#define JQ_NTR 5              // trace scalars per control per backward step
#define JQ_STATE_ARRAYS 4     // U, V, MU, NU
#define JQ_STATE_EXTRA 8      // 64-double rows after the arrays: CARRY[0..3], LEAK, spare

template <int NT>
struct Arr {
    d4 t[NT];
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
        d4 p = x.t[i] * y.t[i];
        s += (p[0] + p[1]) + (p[2] + p[3]);
    }
    return s;
}
// y += c * tab .* x   (tab: per-row table in LDS, padded to 16*NT rows; g = lane>>4)
template <int NT>
__device__ __forceinline__ void a_axpy_rows(Arr<NT>& y, double c, const double* tab, int g, const Arr<NT>& x)
{
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) y.t[i][r] += (c * tab[16 * i + 4 * r + g]) * x.t[i][r];
}
// sum_rows tab[row] * (x^2 * cx + y^2 * cy)
template <int NT>
__device__ __forceinline__ double a_wsq(const double* tab, int g, const Arr<NT>& x)
{
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += tab[16 * i + 4 * r + g] * (x.t[i][r] * x.t[i][r]);
    return s;
}

// image <-> registers: array image = [4*NT][64] doubles, element kk*64 + lane
template <int NT>
__device__ __forceinline__ void a_load(Arr<NT>& a, const double* __restrict__ img, int lane)
{
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) a.t[i][r] = img[(4 * i + r) * 64 + lane];
}
template <int NT>
__device__ __forceinline__ void a_store(const Arr<NT>& a, double* __restrict__ img, int lane)
{
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) img[(4 * i + r) * 64 + lane] = a.t[i][r];
}

// acc += M * x.  M: LDS tile image in walk order (kk outer, mt inner), 64 doubles per tile;
// `mat` already carries the lane offset.  Tile (mt,kk) lane l holds M[16*mt + (l&15)][4*kk + (l>>4)].
template <int NT>
__device__ __forceinline__ void mm(Arr<NT>& acc, const double* mat, const Arr<NT>& x)
{
#pragma unroll
    for (int kk = 0; kk < 4 * NT; ++kk) {
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) {
            double a = mat[(kk * NT + mt) * 64];
            acc.t[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, x.t[kk >> 2][kk & 3], acc.t[mt], 0, 0, 0);
        }
    }
}

__device__ __forceinline__ double wave_sum(double x)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

// ---------------------------------------------------------------------------------------------
// LDS double buffer of operator images fed by global->LDS DMA.
struct Pipe {
    char* smem;
    int slot_bytes;
    int rounds;  // slot_bytes / 4096
    int cur;     // slot holding the matrix that is used next
    int wave, lane;

    __device__ __forceinline__ void dma(const double* src, int slot) const
    {
        char* dst = smem + (size_t)slot * slot_bytes;
        for (int r = 0; r < rounds; ++r) {
            int piece = r * JQ_WAVES + wave;  // 1 KiB per wave-instruction
            const char* s = (const char*)src + (size_t)piece * 1024 + lane * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                             (__attribute__((address_space(3))) void*)(dst + piece * 1024), 16, 0, 0);
        }
    }
    // Barrier: the matrix in slot `cur` has landed and every wave is done with the other slot;
    // start the DMA of `next` into the other slot; return the (lane-offset) LDS image to use now.
    __device__ __forceinline__ const double* use_and_prefetch(const double* next)
    {
        __syncthreads();
        dma(next, cur ^ 1);
        const double* M = (const double*)(smem + (size_t)cur * slot_bytes) + lane;
        cur ^= 1;
        return M;
    }
};

// X = sum_{j=0..m} (c S)^j B   (neumann!, src/linear_solvers.jl:81-106; c = h/2).  B is clobbered.
template <int NT>
__device__ __forceinline__ void neumann(Arr<NT>& X, Arr<NT>& B, const double* S, double c, int m)
{
    X = B;
    double coeff = 1.0;
    for (int j = 0; j < m; ++j) {
        Arr<NT> T;
        a_zero(T);
        mm(T, S, B);
        coeff *= c;
        a_axpy(X, coeff, T);
        B = T;
    }
}

struct PropArgs {
    const double* stream;   // chunk tile stream: time point j -> K at (2j)*mat_elems, S at (2j+1)*mat_elems
    const double* himg;     // constant images [H0 | Hsym_0.. | Hanti_0..], mat_elems each
    double* state;          // per-slab array file
    const double* colinfo;  // per slab: eps[16], wgt[16]
    double* traces;         // backward: [nslabs][nsteps_chunk][Ncoupled*JQ_NTR]
    double* hist_r;         // forward history of sample 0 ([Ntot,N,nsteps+1]) or null
    double* hist_i;
    const double* tabs;     // wd[NP] (diag wmat_real, zero padded), ws[NP] (shift weights)
    long long mat_elems;    // doubles per operator image (multiple of 512)
    int rounds;             // mat_elems*8/4096
    int nsteps_chunk;
    int m;                  // Neumann terms
    int nslabs;
    int Ncoupled;
    int step0;              // global index of the first step of this chunk
    int first_chunk;
    int Ntot, N;
    int use_shift;
    int forced;             // backward: add the leakage forcing (0: step_no_forcing!)
    double h;               // signed time step
    double tinv;            // 1/T
    long long state_stride; // doubles per slab in the array file
};

__device__ __forceinline__ const double* stream_mat(const PropArgs& a, int j, int isS)
{
    return a.stream + (size_t)(2 * j + isS) * a.mat_elems;
}

// State (re-)integration, positions 0..5 of one Stormer-Verlet step (forward step!,
// src/StormerVerlet.jl:461-504, also used with h<0 by the backward sweep, src/evalobjgrad.jl:879):
//   in : u, v at t                 out: u = u(t+h), v unchanged, v05, l1, L2 = S05*v05 (partial l2)
// The caller finishes with position 6:  L2 += K05*u_new ; v += c*(l1 + L2).
// Operator order per step: K05 S05 K0 S0 K1 S1 (K05) -- `after` is prefetched while S1 is in use.
template <int NT>
__device__ __forceinline__ void sv_step_head(Pipe& p, const PropArgs& a, int n, const double* after, bool active,
                                             double eps, const double* ws, int g, Arr<NT>& u, const Arr<NT>& v,
                                             Arr<NT>& v05, Arr<NT>& l1, Arr<NT>& L2)
{
    const double c = 0.5 * a.h;
    const int j0 = 2 * n, j05 = 2 * n + 1, j1 = 2 * n + 2;
    Arr<NT> A;
    // pos 0: K05
    const double* M = p.use_and_prefetch(stream_mat(a, j05, 1));
    if (active) {
        a_zero(A);
        mm(A, M, u);
        if (a.use_shift) a_axpy_rows(A, eps, ws, g, u);
    }
    // pos 1: S05 -- rhs = K05 u + S05 v ; l1 = (I - c S05)^-1 rhs ; v05 = v + c l1 ; L2 = S05 v05
    M = p.use_and_prefetch(stream_mat(a, j0, 0));
    if (active) {
        mm(A, M, v);
        neumann(l1, A, M, c, a.m);
        v05 = v;
        a_axpy(v05, c, l1);
        a_zero(L2);
        mm(L2, M, v05);
    }
    // pos 2: K0 -- A = K0 v05
    M = p.use_and_prefetch(stream_mat(a, j0, 1));
    if (active) {
        a_zero(A);
        mm(A, M, v05);
        if (a.use_shift) a_axpy_rows(A, eps, ws, g, v05);
    }
    // pos 3: S0 -- kappa1 = S0 u - K0 v05 ; u += c kappa1
    M = p.use_and_prefetch(stream_mat(a, j1, 0));
    if (active) {
        a_neg(A);
        mm(A, M, u);
        a_axpy(u, c, A);
    }
    // pos 4: K1 -- A = K1 v05
    M = p.use_and_prefetch(stream_mat(a, j1, 1));
    if (active) {
        a_zero(A);
        mm(A, M, v05);
        if (a.use_shift) a_axpy_rows(A, eps, ws, g, v05);
    }
    // pos 5: S1 -- rhs = S1 (u + c kappa1) - K1 v05 ; kappa2 = (I - c S1)^-1 rhs ; u += c kappa2
    M = p.use_and_prefetch(after);
    if (active) {
        Arr<NT> k2;
        a_neg(A);
        mm(A, M, u);
        neumann(k2, A, M, c, a.m);
        a_axpy(u, c, k2);
    }
}

// ---------------------------------------------------------------------------------------------
// Forward sweep over one chunk of time steps (src/evalobjgrad.jl:698-753).
template <int NT, int MINW>
__global__ __launch_bounds__(256, MINW) void k_forward(PropArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KT = 4 * NT;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4;
    const int slab = blockIdx.x * JQ_WAVES + wave;
    const bool active = slab < a.nslabs;

    Pipe p;
    p.smem = smem;
    p.slot_bytes = (int)(a.mat_elems * 8);
    p.rounds = a.rounds;
    p.cur = 0;
    p.wave = wave;
    p.lane = lane;
    double* tab = (double*)(smem + 2 * (size_t)p.slot_bytes);
    const double* wd = tab;
    const double* ws = tab + 16 * NT;
    for (int i = threadIdx.x; i < 32 * NT; i += blockDim.x) tab[i] = a.tabs[i];

    Arr<NT> u, v;
    double leak = 0.0, eps = 0.0;
    double* st = a.state + (size_t)(active ? slab : 0) * a.state_stride;
    if (active) {
        a_load(u, st, lane);
        a_load(v, st + KT * 64, lane);
        leak = st[(JQ_STATE_ARRAYS * KT + JQ_MAXNC) * 64 + lane];
        eps = a.colinfo[(size_t)slab * 32 + (lane & 15)];
    } else {
        a_zero(u);
        a_zero(v);
    }
    // first operator of the chunk: K05 of step 0
    p.dma(stream_mat(a, 1, 0), 0);

    const double c = 0.5 * a.h;
    for (int n = 0; n < a.nsteps_chunk; ++n) {
        Arr<NT> v05, l1, L2;
        if (active) leak += a_wsq(wd, g, u);  // trapezoidal part: tr(vr' W vr) at t_n (:700)
        sv_step_head<NT>(p, a, n, stream_mat(a, 2 * n + 1, 0), active, eps, ws, g, u, v, v05, l1, L2);
        // pos 6: K05 again -- l2 = K05 u_new + S05 v05 ; v += c (l1 + l2).  Prefetch next step's K05.
        const int nn = (n + 1 < a.nsteps_chunk) ? n + 1 : n;
        const double* M = p.use_and_prefetch(stream_mat(a, 2 * nn + 1, 0));
        if (active) {
            mm(L2, M, u);
            if (a.use_shift) a_axpy_rows(L2, eps, ws, g, u);
            a_add(L2, l1);
            a_axpy(v, c, L2);
            // leak integrand: tr(vr' W vr + 2 vi05' W vi05) after the step (:716, penalf2a :2170-2180)
            leak += a_wsq(wd, g, u) + 2.0 * a_wsq(wd, g, v05);
            if (a.hist_r) {
                // usaver[:,:,step+1] = vr ; usavei = -vi (:748-752); only sample 0 (slab 0, columns < N)
                const int col = lane & 15;
                if (slab == 0 && col < a.N) {
                    const size_t off = (size_t)(a.step0 + n + 1) * a.Ntot * a.N + (size_t)col * a.Ntot;
#pragma unroll
                    for (int i = 0; i < NT; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = 16 * i + 4 * r + g;
                            if (row < a.Ntot) {
                                a.hist_r[off + row] = u.t[i][r];
                                a.hist_i[off + row] = -v.t[i][r];
                            }
                        }
                }
            }
        }
    }
    __syncthreads();  // drain the last (unused) prefetch before the workgroup exits
    if (active) {
        a_store(u, st, lane);
        a_store(v, st + KT * 64, lane);
        st[(JQ_STATE_ARRAYS * KT + JQ_MAXNC) * 64 + lane] = leak;
    }
}

// ---------------------------------------------------------------------------------------------
// Backward sweep over one chunk (src/evalobjgrad.jl:859-921): state re-integration with h<0,
// adjoint step! with forcing (src/StormerVerlet.jl:255-303) or step_no_forcing! (:365-451), and the
// per-step trace scalars of adjoint_grad_calc! (src/evalobjgrad.jl:2567-2619), written to `traces`.
template <int NT, int MINW>
__global__ __launch_bounds__(256, MINW) void k_backward(PropArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KT = 4 * NT;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4;
    const int slab = blockIdx.x * JQ_WAVES + wave;
    const bool active = slab < a.nslabs;
    const int Nc = a.Ncoupled;

    Pipe p;
    p.smem = smem;
    p.slot_bytes = (int)(a.mat_elems * 8);
    p.rounds = a.rounds;
    p.cur = 0;
    p.wave = wave;
    p.lane = lane;
    double* tab = (double*)(smem + 2 * (size_t)p.slot_bytes);
    const double* wd = tab;
    const double* ws = tab + 16 * NT;
    double* carry = tab + 32 * NT;  // [JQ_MAXNC][256]
    for (int i = threadIdx.x; i < 32 * NT; i += blockDim.x) tab[i] = a.tabs[i];

    Arr<NT> u, v, mu, nu;
    double eps = 0.0, wgt = 0.0;
    double* st = a.state + (size_t)(active ? slab : 0) * a.state_stride;
    if (active) {
        a_load(u, st, lane);
        a_load(v, st + KT * 64, lane);
        a_load(mu, st + 2 * KT * 64, lane);
        a_load(nu, st + 3 * KT * 64, lane);
        eps = a.colinfo[(size_t)slab * 32 + (lane & 15)];
        wgt = a.colinfo[(size_t)slab * 32 + 16 + (lane & 15)];
        for (int q = 0; q < Nc; ++q) carry[q * 256 + threadIdx.x] = st[(JQ_STATE_ARRAYS * KT + q) * 64 + lane];
    } else {
        a_zero(u);
        a_zero(v);
        a_zero(mu);
        a_zero(nu);
    }
    const double* Hs0 = a.himg + a.mat_elems;                     // Hsym_q  at Hs0 + q*mat_elems
    const double* Ha0 = a.himg + (size_t)(1 + Nc) * a.mat_elems;  // Hanti_q at Ha0 + q*mat_elems
    const double c = 0.5 * a.h;
    const double fw = a.forced ? a.tinv : 0.0;  // forcing weight: hr0 = tinv*W*vr etc. (:862, :882-888)

    if (a.first_chunk) {
        // carry_q = tr(vr' Hsym_q lambdai) at t = T: the "vr0/lambdai0" term of the first backward
        // step (:2609); on later steps it is the previous step's tr(vr' Hsym_q lambdai) (:901-902).
        p.dma(Hs0, 0);
        for (int q = 0; q < Nc; ++q) {
            const double* nxt = (q + 1 < Nc) ? Hs0 + (size_t)(q + 1) * a.mat_elems : stream_mat(a, 1, 0);
            const double* M = p.use_and_prefetch(nxt);
            if (active) {
                Arr<NT> T;
                a_zero(T);
                mm(T, M, nu);
                carry[q * 256 + threadIdx.x] = a_dot(u, T);
            }
        }
    } else {
        p.dma(stream_mat(a, 1, 0), 0);
    }

    for (int n = 0; n < a.nsteps_chunk; ++n) {
        const int j0 = 2 * n, j05 = 2 * n + 1, j1 = 2 * n + 2;
        Arr<NT> uold, v05, X;
        {
            Arr<NT> l1, L2;
            uold = u;
            sv_step_head<NT>(p, a, n, stream_mat(a, j05, 0), active, eps, ws, g, u, v, v05, l1, L2);
            // pos 6: K05 -- finish the state step; first adjoint product R = K05 nu
            const double* M = p.use_and_prefetch(stream_mat(a, j0, 1));
            Arr<NT> R;
            if (active) {
                mm(L2, M, u);
                if (a.use_shift) a_axpy_rows(L2, eps, ws, g, u);
                a_add(L2, l1);
                a_axpy(v, c, L2);
                a_zero(R);
                mm(R, M, nu);
                if (a.use_shift) a_axpy_rows(R, eps, ws, g, nu);
            }
            // pos 7: S0 -- rhs = S0 mu - K05 nu + hr0 ; kappa2 = (I - c S0)^-1 rhs ; mu += c kappa2 ; X = mu
            M = p.use_and_prefetch(stream_mat(a, j0, 0));
            if (active) {
                Arr<NT> k2;
                a_neg(R);
                mm(R, M, mu);
                a_axpy_rows(R, fw, wd, g, uold);
                neumann(k2, R, M, c, a.m);
                a_axpy(mu, c, k2);
                X = mu;
            }
        }
        {
            Arr<NT> A, Bq;
            // pos 8: K0 -- A = K0 X
            const double* M = p.use_and_prefetch(stream_mat(a, j1, 0));
            if (active) {
                a_zero(A);
                mm(A, M, X);
                if (a.use_shift) a_axpy_rows(A, eps, ws, g, X);
            }
            // pos 9: K1 -- Bq = K1 X
            M = p.use_and_prefetch(stream_mat(a, j05, 1));
            if (active) {
                a_zero(Bq);
                mm(Bq, M, X);
                if (a.use_shift) a_axpy_rows(Bq, eps, ws, g, X);
            }
            // pos 10: S05 -- l2 = K0 X + S05 nu + hi0 ; rhs = S05 (nu + c l2) + K1 X + hi1 ;
            //                l1 = (I - c S05)^-1 rhs ; nu += c (l2 + l1)
            M = p.use_and_prefetch(stream_mat(a, j05, 0));
            if (active) {
                Arr<NT> P, l1;
                a_zero(P);
                mm(P, M, nu);
                a_add(A, P);
                a_axpy_rows(A, fw, wd, g, v05);  // A = l2
                {
                    Arr<NT> Q;
                    a_zero(Q);
                    mm(Q, M, A);
                    a_add(Bq, P);
                    a_axpy(Bq, c, Q);
                }
                a_axpy_rows(Bq, fw, wd, g, v05);  // Bq = rhs
                neumann(l1, Bq, M, c, a.m);
                a_add(A, l1);
                // keep lambdai0 + lambdai for the last trace: P = nu_old + nu_new
                P = nu;
                a_axpy(nu, c, A);
                a_add(P, nu);
                Bq = P;  // Bq now holds (nu_old + nu_new)
            }
            // pos 11: K05 -- R = K05 nu_new
            M = p.use_and_prefetch(stream_mat(a, j1, 1));
            if (active) {
                a_zero(A);
                mm(A, M, nu);
                if (a.use_shift) a_axpy_rows(A, eps, ws, g, nu);
            }
            // pos 12: S1 -- kappa1 = S1 X - K05 nu + hr1 ; mu += c kappa1
            M = p.use_and_prefetch(Ha0);
            if (active) {
                a_neg(A);
                mm(A, M, X);
                a_axpy_rows(A, fw, wd, g, u);
                a_axpy(mu, c, A);
            }
            // traces (adjoint_grad_calc!, :2581-2618), per control q, weighted by the sample weight:
            //   tr1 = tr(vr0' Hanti X)  tr3 = tr(vr' Hanti X)  tr5 = tr(vi05' Hanti (li0+li))
            //   tr2 = tr(vi05' Hsym X)  tr4 = tr(vr' Hsym li) + tr(vr0' Hsym li0)
            for (int q = 0; q < Nc; ++q) {
                double t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;
                M = p.use_and_prefetch(Hs0 + (size_t)q * a.mat_elems);  // now: Hanti_q
                if (active) {
                    Arr<NT> T;
                    a_zero(T);
                    mm(T, M, X);
                    t1 = a_dot(uold, T);
                    t3 = a_dot(u, T);
                    a_zero(T);
                    mm(T, M, Bq);
                    t5 = a_dot(v05, T);
                }
                const int nn = (n + 1 < a.nsteps_chunk) ? n + 1 : n;
                const double* nxt = (q + 1 < Nc) ? Ha0 + (size_t)(q + 1) * a.mat_elems : stream_mat(a, 2 * nn + 1, 0);
                M = p.use_and_prefetch(nxt);  // now: Hsym_q
                if (active) {
                    Arr<NT> T;
                    a_zero(T);
                    mm(T, M, X);
                    t2 = a_dot(v05, T);
                    a_zero(T);
                    mm(T, M, nu);
                    const double p4 = a_dot(u, T);
                    t4 = p4 + carry[q * 256 + threadIdx.x];
                    carry[q * 256 + threadIdx.x] = p4;
                    t1 = wave_sum(t1 * wgt);
                    t2 = wave_sum(t2 * wgt);
                    t3 = wave_sum(t3 * wgt);
                    t4 = wave_sum(t4 * wgt);
                    t5 = wave_sum(t5 * wgt);
                    if (lane == 0) {
                        double* tr = a.traces + ((size_t)slab * a.nsteps_chunk + n) * (Nc * JQ_NTR) + q * JQ_NTR;
                        tr[0] = t1;
                        tr[1] = t2;
                        tr[2] = t3;
                        tr[3] = t4;
                        tr[4] = t5;
                    }
                }
            }
        }
    }
    __syncthreads();
    if (active) {
        a_store(u, st, lane);
        a_store(v, st + KT * 64, lane);
        a_store(mu, st + 2 * KT * 64, lane);
        a_store(nu, st + 3 * KT * 64, lane);
        for (int q = 0; q < Nc; ++q) st[(JQ_STATE_ARRAYS * KT + q) * 64 + lane] = carry[q * 256 + threadIdx.x];
    }
}
