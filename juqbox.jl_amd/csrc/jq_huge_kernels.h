// jq_huge_kernels.h -- Stormer-Verlet propagators for Hilbert spaces beyond 16 tile rows (Ntot > 256): correctness first.
//
// The reference has no size limit (objparams, src/evalobjgrad.jl:152-343); up to ABI 4 this library refused Ntot > 256 because the
// cooperative kernels give every tile row (16 rows of the state) a wave of its own and a workgroup has at most 16 waves.  Here the
// tile-row count NT is a RUN-TIME number: one workgroup of 16 waves per 16-column slab, wave w owns the tile rows w, w + 16, w + 32, ...
// Nothing lives in registers across products: the state arrays and every intermediate vector of a time step are [KT][64] arrays in a
// per-slab work area in global memory (L2-resident: 20 x 2 NT KB), a product D = C + M x is NT x KT dense v_mfma_f64_16x16x4 tiles
// whose A operands come straight from the tile stream (dense window, no structure skipping) and whose B operands are read from x.
// Two workgroup barriers per product (x complete before anybody reads it; everybody done before its owner overwrites it).  The
// arithmetic -- operator schedule, Horner form of the Neumann series, Jacobi stopping rule per sample, leak integrand, forcing, the
// trace products of adjoint_grad_calc!, full leakage weights in low-rank form -- is k_forward_coop / k_backward_coop's
// (jq_coop_kernels.h), statement for statement; only where a value lives differs.  Speed is whatever it is: a 300-level problem is
// far outside what the reference's users run (its own cost grows with Ntot^2 per product too), the point is that it is not refused.
#pragma once
#include "jq_coop_kernels.h"

#define JQ_HUGE_WAVES 16
#define JQ_HUGE_VECS 16      // work-area vectors per slab (below)
enum { HV_UN, HV_V05, HV_VN, HV_A, HV_Y, HV_YB, HV_R, HV_X, HV_L, HV_Q, HV_P, HV_NBN, HV_BQ, HV_G, HV_T, HV_SPARE };

struct Huge {
    int NT, KT, wave, lane, g;
    double* work;               // this slab's work area: vector i at work + i KT 64
    OpCursor cur;
    const double* M;            // current operator image (lane offset applied): tile (mt, kk) at (mt KT + kk) 64
    const double* tabs;         // [wd[16 NT] | ws[16 NT]] in natural row order (global memory)
    double ceps, wgt, tol2;
    int use_shift, m, ncol;
    double* nrm;                // LDS [JQ_HUGE_WAVES][16]
    double* xch;                // LDS [6][JQ_HUGE_WAVES][16] (full weights: column dots)
    __device__ __forceinline__ double* vec(int i) const { return work + (size_t)i * KT * 64; }
    // f(element offset, Hilbert-space row) for every element this lane owns
    template <class F>
    __device__ __forceinline__ void each(F f) const
    {
        for (int mt = wave; mt < NT; mt += JQ_HUGE_WAVES)
#pragma unroll
            for (int r = 0; r < 4; ++r) f((size_t)(4 * mt + r) * 64 + lane, 16 * mt + 4 * r + g);
    }
    __device__ __forceinline__ void next_op() { M = cur.next(); }
    // D = C + M x (C == nullptr: D = M x).  D must not be x; C may be D.  Barriers: x complete on entry, all reads done on exit.
    __device__ __forceinline__ void mm(double* D, const double* C, const double* x) const
    {
        __syncthreads();
        for (int mt = wave; mt < NT; mt += JQ_HUGE_WAVES) {
            d4 acc = {0.0, 0.0, 0.0, 0.0};
            if (C)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = C[(size_t)(4 * mt + r) * 64 + lane];
            const double* Mr = M + (size_t)mt * KT * 64;
            for (int kk = 0; kk < KT; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Mr[(size_t)kk * 64], x[(size_t)kk * 64 + lane], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) D[(size_t)(4 * mt + r) * 64 + lane] = acc[r];
        }
        __syncthreads();
    }
    // D += sign ceps ws .* x  (the ensemble's diagonal perturbation of K; own elements only)
    __device__ __forceinline__ void shift(double* D, const double* x, double sign) const
    {
        if (!use_shift) return;
        const double* ws = tabs + 16 * NT;
        each([&](size_t e, int row) { D[e] += (sign * ceps * ws[row]) * x[e]; });
    }
    // sum over this lane's own elements of wd[row] a[e] b[e]
    __device__ __forceinline__ double wdot(const double* a, const double* b) const
    {
        double s = 0.0;
        each([&](size_t e, int row) { s += tabs[row] * (a[e] * b[e]); });
        return s;
    }
    __device__ __forceinline__ double dot(const double* a, const double* b) const
    {
        double s = 0.0;
        each([&](size_t e, int) { s += a[e] * b[e]; });
        return s;
    }
    // per-lane value -> total over the slab COLUMN of this lane (all rows, all waves), valid in every lane.  Contains barriers.
    template <int D>
    __device__ __forceinline__ void colsum(double (&v)[D])
    {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            v[d] += __shfl_xor(v[d], 16);
            v[d] += __shfl_xor(v[d], 32);
        }
        __syncthreads();
        if (lane < 16)
#pragma unroll
            for (int d = 0; d < D; ++d) xch[(d * JQ_HUGE_WAVES + wave) * 16 + lane] = v[d];
        __syncthreads();
#pragma unroll
        for (int d = 0; d < D; ++d) {
            double s = 0.0;
            for (int w = 0; w < JQ_HUGE_WAVES; ++w) s += xch[(d * JQ_HUGE_WAVES + w) * 16 + (lane & 15)];
            v[d] = s;
        }
    }
    // out = bpa + sum_{j=1..m} S^j A with the current operator (coop_horner): Horner form, or the Jacobi iteration with the reference's
    // per-sample stopping rule when tol2 > 0.  out may be bpa; out, bpa must not be A, Y, Yb.
    __device__ __forceinline__ void horner(double* out, const double* bpa, const double* A)
    {
        if (m <= 0) {
            if (out != bpa) each([&](size_t e, int) { out[e] = bpa[e]; });
            return;
        }
        double *Y = vec(HV_Y), *Yb = vec(HV_YB);
        if (tol2 > 0.0) {
            // X_j = A + S X_{j-1}, X_0 = A; stop at the first j with ||X_j - X_{j-1}||_F^2 < tol2 (per sample) or at j = m
            each([&](size_t e, int) { Y[e] = A[e]; });
            const int col = lane & 15, n = ncol < 16 ? ncol : 16, c0 = col - col % n;
            bool done = false;
            for (int j = 1; j <= m; ++j) {
                mm(Yb, A, Y);
                double e2 = 0.0;
                each([&](size_t e, int) { const double d = Yb[e] - Y[e]; e2 += d * d; });
                e2 += __shfl_xor(e2, 16);
                e2 += __shfl_xor(e2, 32);
                if (lane < 16) nrm[16 * wave + lane] = e2;
                __syncthreads();
                double err2 = 0.0;
                for (int k = 0; k < n; ++k) {
                    double t = 0.0;
                    if (c0 + k < 16)
                        for (int w = 0; w < JQ_HUGE_WAVES; ++w) t += nrm[16 * w + c0 + k];
                    err2 += t;
                }
                if (!done) each([&](size_t e, int) { Y[e] = Yb[e]; });      // a converged sample keeps its iterate
                done = done || err2 < tol2;
                if (__syncthreads_and(done ? 1 : 0)) break;
            }
            each([&](size_t e, int) { out[e] = (bpa[e] - A[e]) + Y[e]; });
            return;
        }
        const double* src = A;
        double* dst = Y;
        for (int j = 1; j < m; ++j) {
            mm(dst, A, src);
            src = dst;
            dst = (dst == Y) ? Yb : Y;
        }
        mm(vec(HV_SPARE), bpa, src);      // (out may be bpa, which is C here: fine; but out must not be the product's x)
        const double* res = vec(HV_SPARE);
        each([&](size_t e, int) { out[e] = res[e]; });
    }
    // the state (re-)integration of one step: uses 0 .. 5 of the cooperative schedule Kp05 S05 Kn0 Kn1 S0 S1 (coop_state)
    //   in: u, v   out: vectors HV_UN, HV_V05, HV_VN (= v05 + S05 v05; the caller adds Kp05 un with use 6)
    __device__ __forceinline__ void state_step(const double* u, const double* v)
    {
        double *un = vec(HV_UN), *v05 = vec(HV_V05), *vN = vec(HV_VN), *A = vec(HV_A);
        next_op();      // use 0: Kp05 -- A = c K05 u
        mm(A, nullptr, u);
        shift(A, u, 1.0);
        next_op();      // use 1: S05 -- A = c (K05 u + S05 v) ; v05 = v + sum_j S^j A ; vN = v05 + S05 v05
        mm(A, A, v);
        each([&](size_t e, int) { v05[e] = v[e] + A[e]; });
        horner(v05, v05, A);
        mm(vN, v05, v05);
        next_op();      // use 2: Kn0 -- un = u - c K0 v05
        mm(un, u, v05);
        shift(un, v05, -1.0);
        next_op();      // use 3: Kn1 -- A = -c K1 v05
        mm(A, nullptr, v05);
        shift(A, v05, -1.0);
        next_op();      // use 4: S0 -- un = u + c (S0 u - K0 v05)
        mm(un, un, u);
        next_op();      // use 5: S1 -- A = c (S1 un - K1 v05) ; un += sum_j S^j A
        mm(A, A, un);
        each([&](size_t e, int) { un[e] += A[e]; });
        horner(un, un, A);
    }
};

// the waves' per-lane values summed in wave order into dst[lane] (+= when accumulate)
__device__ __forceinline__ void huge_wg_sum_store(double val, double* scratch, double* dst, int wave, int lane, bool accumulate)
{
    __syncthreads();
    scratch[wave * 64 + lane] = val;
    __syncthreads();
    if (wave == 0) {
        double s = accumulate ? dst[lane] : 0.0;
        for (int w = 0; w < JQ_HUGE_WAVES; ++w) s += scratch[w * 64 + lane];
        dst[lane] = s;
    }
}

__device__ __forceinline__ void huge_setup(Huge& hg, const PropArgs& a, double* nrm, double* xch)
{
    hg.NT = (a.Ntot + 15) / 16;
    hg.KT = 4 * hg.NT;
    hg.lane = threadIdx.x & 63;
    hg.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    hg.g = hg.lane >> 4;
    hg.work = a.park + (size_t)blockIdx.x * JQ_HUGE_VECS * hg.KT * 64;
    hg.cur.init(nullptr, a, 0, 0, 0);
    hg.M = nullptr;
    hg.tabs = a.tabs;
    hg.use_shift = a.use_shift, hg.m = a.m, hg.ncol = a.N;
    hg.tol2 = a.jacobi_tol2;
    hg.ceps = 0.5 * a.h * a.colinfo[(size_t)blockIdx.x * 32 + (hg.lane & 15)];
    hg.wgt = a.colinfo[(size_t)blockIdx.x * 32 + 16 + (hg.lane & 15)];
    hg.nrm = nrm, hg.xch = xch;
}

// low-rank table rows of term k (a_k: ab = 0, b_k: ab = 1) in natural row order
__device__ __forceinline__ const double* huge_wrow(const PropArgs& a, int k, int ab) { return a.wlr + a.wlam + (size_t)(2 * k + ab) * a.wstride; }
__device__ __forceinline__ double huge_rdot(const Huge& hg, const double* t, const double* x)
{
    double s = 0.0;
    hg.each([&](size_t e, int row) { s += t[row] * x[e]; });
    return s;
}

// Forward sweep: one slab per workgroup of 16 waves (k_forward_coop).
__global__ __launch_bounds__(64 * JQ_HUGE_WAVES) void k_forward_huge(PropArgs a)
{
    __shared__ double scratch[JQ_HUGE_WAVES * 64];
    __shared__ double nrm[JQ_HUGE_WAVES * 16];
    __shared__ double xch[6 * JQ_HUGE_WAVES * 16];
    Huge hg;
    huge_setup(hg, a, nrm, xch);
    const int KT = hg.KT, lane = hg.lane, wave = hg.wave, g = hg.g;
    double* st = a.state + (size_t)blockIdx.x * a.state_stride;
    double *u = st, *v = st + (size_t)KT * 64;
    double *un = hg.vec(HV_UN), *v05 = hg.vec(HV_V05), *vN = hg.vec(HV_VN);
    double leak = 0.0;
    for (int n = 0; n < a.nsteps_chunk; ++n) {
        leak += hg.wdot(u, u);      // trapezoidal part at t_n (src/evalobjgrad.jl:700)
        hg.state_step(u, v);
        hg.next_op();               // use 6: Kp05 -- v(t+h) = v05 + c (K05 u_new + S05 v05)
        hg.mm(vN, vN, un);
        hg.shift(vN, un, 1.0);
        if (a.wrank > 0) {          // full weights (:700, :716-718; k_forward_coop)
            double lk = 0.0;
            for (int k = 0; k < a.wrank; ++k) {
                const double *ak = huge_wrow(a, k, 0), *bk = huge_wrow(a, k, 1);
                double d[6] = {huge_rdot(hg, ak, u), huge_rdot(hg, bk, u), huge_rdot(hg, ak, un), huge_rdot(hg, bk, un), huge_rdot(hg, ak, v05), huge_rdot(hg, bk, v05)};
                hg.colsum<6>(d);
                lk += a.wlr[k] * ((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]) + 2.0 * (d[4] * d[4] + d[5] * d[5]) - 2.0 * (d[5] * d[0] - d[4] * d[1]));
            }
            if (wave == 0 && lane < 16) leak += lk;
        }
        hg.each([&](size_t e, int) { u[e] = un[e]; v[e] = vN[e]; });
        leak += hg.wdot(u, u) + 2.0 * hg.wdot(v05, v05);      // (:716, penalf2a :2170-2180)
        if (a.hist_r) {
            const int col = a.parts > 1 ? 16 * (int)blockIdx.x + (lane & 15) : (lane & 15);      // column of sample 0
            if ((int)blockIdx.x < a.parts && col < a.N) {
                const size_t off = (size_t)(a.step0 + n + 1) * a.Ntot * a.N + (size_t)col * a.Ntot;
                hg.each([&](size_t e, int row) {
                    if (row < a.Ntot) {
                        a.hist_r[off + row] = u[e];
                        a.hist_i[off + row] = -v[e];
                    }
                });
            }
        }
    }
    (void)g;
    huge_wg_sum_store(leak, scratch, &st[(size_t)(JQ_STATE_ARRAYS * KT + JQ_MAXNC) * 64], wave, lane, true);
}

// Backward sweep: one slab per workgroup of 16 waves (k_backward_coop).  Trace scalars per wave: traces[slab 16 + wave][step][Nc JQ_NTR].
__global__ __launch_bounds__(64 * JQ_HUGE_WAVES) void k_backward_huge(PropArgs a)
{
    __shared__ double scratch[JQ_HUGE_WAVES * 64];
    __shared__ double nrm[JQ_HUGE_WAVES * 16];
    __shared__ double xch[6 * JQ_HUGE_WAVES * 16];
    Huge hg;
    huge_setup(hg, a, nrm, xch);
    const int KT = hg.KT, lane = hg.lane, wave = hg.wave, Nc = a.Ncoupled;
    double* st = a.state + (size_t)blockIdx.x * a.state_stride;
    double *u = st, *v = st + (size_t)KT * 64, *mu = st + (size_t)2 * KT * 64, *nb = st + (size_t)3 * KT * 64;
    double *un = hg.vec(HV_UN), *v05 = hg.vec(HV_V05), *vN = hg.vec(HV_VN), *R = hg.vec(HV_R), *X = hg.vec(HV_X), *L = hg.vec(HV_L);
    double *Qv = hg.vec(HV_Q), *P = hg.vec(HV_P), *nbn = hg.vec(HV_NBN), *Bq = hg.vec(HV_BQ), *G = hg.vec(HV_G), *T = hg.vec(HV_T);
    const double wgt = hg.wgt;
    const double cfw = a.forced ? 0.5 * a.h * a.tinv : 0.0;
    double carry[JQ_MAXNC];
    for (int q = 0; q < JQ_MAXNC; ++q) carry[q] = (q < Nc) ? st[(size_t)(JQ_STATE_ARRAYS * KT + q) * 64 + lane] / JQ_HUGE_WAVES : 0.0;
    double* trw = a.traces + ((size_t)((size_t)blockIdx.x * JQ_HUGE_WAVES + wave) * a.nsteps_chunk) * (Nc * JQ_NTR);
    const bool wforce = a.wrank > 0 && a.forced;
    const double* wd = a.tabs;

    if (a.first_chunk) {      // carry_q = tr(vr' Hsym_q lambdai) at t = T
        for (int q = 0; q < JQ_MAXNC; ++q)
            if (q < Nc) {
                hg.next_op();
                hg.mm(T, nullptr, nb);
                carry[q] = -hg.dot(u, T);      // nb = -lambda_i
            }
    }
    for (int n = 0; n < a.nsteps_chunk; ++n) {
        hg.state_step(u, v);
        hg.next_op();      // use 6: Kp05 -- finish the state step; R = c K05 nb
        hg.mm(vN, vN, un);
        hg.shift(vN, un, 1.0);
        hg.mm(R, nullptr, nb);
        hg.shift(R, nb, 1.0);
        hg.next_op();      // use 7: S0 -- R = c (S0 mu - K05 li + hr0) ; X = mu + sum_j S^j R
        hg.mm(R, R, mu);
        hg.each([&](size_t e, int row) { R[e] += (cfw * wd[row]) * u[e]; });
        if (wforce)
            for (int k = 0; k < a.wrank; ++k) {
                const double *ak = huge_wrow(a, k, 0), *bk = huge_wrow(a, k, 1);
                double d[2] = {huge_rdot(hg, ak, u), huge_rdot(hg, bk, u)};
                hg.colsum<2>(d);
                const double cl = cfw * a.wlr[k];
                hg.each([&](size_t e, int row) { R[e] += (cl * d[0]) * ak[row] + (cl * d[1]) * bk[row]; });      // + c hr0
            }
        hg.each([&](size_t e, int) { X[e] = mu[e] + R[e]; });
        hg.horner(X, X, R);
        // early traces with X: t1 = tr(vr0' Hanti_q X), t3 = tr(vr' Hanti_q X)
        for (int q = 0; q < JQ_MAXNC; ++q)
            if (q < Nc) {
                hg.next_op();
                hg.mm(T, nullptr, X);
                const double t1 = wave_sum(hg.dot(u, T) * wgt), t3 = wave_sum(hg.dot(un, T) * wgt);
                if (lane == 0) {
                    trw[(size_t)n * (Nc * JQ_NTR) + q * JQ_NTR + 0] = t1;
                    trw[(size_t)n * (Nc * JQ_NTR) + q * JQ_NTR + 2] = t3;
                }
            }
        hg.next_op();      // use 8: Kn0 -- L = -c K0 X
        hg.mm(L, nullptr, X);
        hg.shift(L, X, -1.0);
        hg.next_op();      // use 9: Kn1 -- Q = -c K1 X
        hg.mm(Qv, nullptr, X);
        hg.shift(Qv, X, -1.0);
        hg.next_op();      // use 10: S05 -- L = -c l2 ; Q = -c (...) ; nb_new = nb + L + sum_j S^j Q
        hg.mm(P, nullptr, nb);
        hg.each([&](size_t e, int row) { P[e] -= (cfw * wd[row]) * v05[e]; });
        if (wforce)
            for (int k = 0; k < a.wrank; ++k) {
                const double *ak = huge_wrow(a, k, 0), *bk = huge_wrow(a, k, 1);
                double d[4] = {huge_rdot(hg, ak, un), huge_rdot(hg, bk, un), huge_rdot(hg, ak, v05), huge_rdot(hg, bk, v05)};
                hg.colsum<4>(d);
                const double cl = cfw * a.wlr[k];
                hg.each([&](size_t e, int row) {
                    P[e] -= (cl * d[2]) * ak[row] + (cl * d[3]) * bk[row];      // - c hi0 (goes into L and Q)
                    Qv[e] += (cl * d[0]) * bk[row] - (cl * d[1]) * ak[row];     // Q: + c Wi vr(t_n) / T
                });
            }
        hg.each([&](size_t e, int) { L[e] += P[e]; Qv[e] += P[e]; });
        hg.mm(Qv, Qv, L);
        hg.each([&](size_t e, int) { nbn[e] = (nb[e] + L[e]) + Qv[e]; });
        hg.horner(nbn, nbn, Qv);
        hg.each([&](size_t e, int) { Bq[e] = nb[e] + nbn[e]; });      // -(li0 + li)
        hg.next_op();      // use 11: Kp05 -- G = X + c K05 nb_new
        hg.mm(G, X, nbn);
        hg.shift(G, nbn, 1.0);
        hg.next_op();      // use 12: S1 -- lambda_r_new = X + c (S1 X - K05 li_new + hr1)
        hg.mm(G, G, X);
        hg.each([&](size_t e, int row) { G[e] += (cfw * wd[row]) * un[e]; });
        if (wforce)
            for (int k = 0; k < a.wrank; ++k) {
                const double *ak = huge_wrow(a, k, 0), *bk = huge_wrow(a, k, 1);
                double d[4] = {huge_rdot(hg, ak, un), huge_rdot(hg, bk, un), huge_rdot(hg, ak, v05), huge_rdot(hg, bk, v05)};
                hg.colsum<4>(d);
                const double cl = cfw * a.wlr[k];
                hg.each([&](size_t e, int row) { G[e] += (cl * (d[0] - d[3])) * ak[row] + (cl * (d[1] + d[2])) * bk[row]; });      // + c hr1
            }
        // late traces: t5 = tr(vi05' Hanti (li0+li)), t2 = tr(vi05' Hsym X), t4 = tr(vr' Hsym li) + carry
        for (int q = 0; q < JQ_MAXNC; ++q)
            if (q < Nc) {
                hg.next_op();      // Hanti_q
                hg.mm(T, nullptr, Bq);
                const double t5 = wave_sum(-hg.dot(v05, T) * wgt);
                hg.next_op();      // Hsym_q
                hg.mm(T, nullptr, X);
                const double t2 = wave_sum(hg.dot(v05, T) * wgt);
                hg.mm(T, nullptr, nbn);
                const double p4 = -hg.dot(un, T);
                const double t4 = wave_sum((p4 + carry[q]) * wgt);
                carry[q] = p4;
                if (lane == 0) {
                    trw[(size_t)n * (Nc * JQ_NTR) + q * JQ_NTR + 4] = t5;
                    trw[(size_t)n * (Nc * JQ_NTR) + q * JQ_NTR + 1] = t2;
                    trw[(size_t)n * (Nc * JQ_NTR) + q * JQ_NTR + 3] = t4;
                }
            }
        hg.each([&](size_t e, int) { u[e] = un[e]; v[e] = vN[e]; mu[e] = G[e]; nb[e] = nbn[e]; });
    }
    for (int q = 0; q < JQ_MAXNC; ++q)
        if (q < Nc) huge_wg_sum_store(carry[q], scratch, &st[(size_t)(JQ_STATE_ARRAYS * KT + q) * 64], wave, lane, false);
}
