// jq_rowlane_imr_kernels.h -- IMPLICIT MIDPOINT propagators (traceobjgrad for Working_Arrays_M,
// src/evalobjgrad.jl:1042-1481; m_step!, src/ImplicitMidpoint.jl:120-227; jacobi_midpoint,
// src/linear_solvers.jl:156-270) in the row-lane mapping of jq_rowlane_kernels.h (Ntot <= 16, small batches).
//
// One step solves  (I - h/2 [S -K; K S]) [u; v] = rhs,  rhs = (I + h/2 [S -K; K S]) [u; v] (+ h * forcing),  with K, S
// at t + h/2, by the reference's fixed-point iteration  x <- rhs + h/2 [S -K; K S] x  started from the old state.
// The stream holds Kp = +cK, S = cS (c = h/2) at the odd time points of the Stormer-Verlet stream generator, so an
// iteration is  u' = (rhs_u + S xu) + K (-xv),  v' = (rhs_v + K xu) + S xv : four NPJ-FMA products.
//
// Convergence: the reference stops after iteration i when BOTH Frobenius norms of the residual at x_i (over the
// Ntot x N block of ONE evaluation) are below tol, at most max_iter iterations, and returns x_i.  The residual at x_i
// is x_i - x_{i+1}, so computing the next iterate doubles as the residual evaluation (4 products per reference
// iteration instead of 8), and the SAME iterate x_i is returned: stopping one iterate later would shift every step by
// O(tol) in the same direction, which accumulates to ~1e-9 over 8 000 steps (measured) -- outside the reference's
// 1e-10 test tolerance.  (A rounding-induced flip of the stopping decision happens only when a residual norm lies
// within ~1e-4 relative of tol and changes that one step by O(tol * rho).)
// The norms are per evaluation, so the columns of a sample must share a wave: this mapping packs
// cpw = N * floor(4/N) columns into a wave (N <= 4; N = 3 leaves one 16-lane row idle) and each sample freezes its
// own lanes when it has converged.
#pragma once
#include "jq_rowlane_kernels.h"

// columns per wave in the sample-aligned packing
__host__ __device__ inline int imr_cols_per_wave(int N) { return N * (4 / N); }

// sum of x over the lanes of MY sample (rows of 16 lanes [N*j, N*j + N) of the wave); every lane gets its sample's sum
__device__ __forceinline__ double sample_sum(double x, int N, int c)
{
    x = row_ror_add<8>(x);
    x = row_ror_add<4>(x);
    x = row_ror_add<2>(x);
    x = row_ror_add<1>(x);
    const double s0 = lane_bcast(x, 0), s1 = lane_bcast(x, 16), s2 = lane_bcast(x, 32), s3 = lane_bcast(x, 48);
    const int j = c / N;
    double r = 0.0;
    r += (0 / N == j) ? s0 : 0.0;
    r += (1 / N == j) ? s1 : 0.0;
    r += (2 / N == j) ? s2 : 0.0;
    r += (3 / N == j) ? s3 : 0.0;
    return r;
}

// x <- the iterate the reference's jacobi_midpoint returns for  x = rhs + [S -K; K S] x  started from (xu, xv);
// (bxu, bxv) = B x_0, which the caller has from forming rhs
// (K, S pre-scaled by h/2; sw: the lane's eps*c*ws shift of diag(K); idle rows: valid == false)
template <int NPJ>
__device__ __forceinline__ void imr_solve(const PropArgs& a, const RowMat<NPJ>& K, const RowMat<NPJ>& S, double sw, double rhs_u,
                                          double rhs_v, double bxu, double bxv, double& xu, double& xv, int c, bool valid)
{
    const double tol2 = a.jacobi_tol2;
    auto apply = [&](double pu, double pv, double& qu, double& qv) {      // q = rhs + B p
        qu = rmv<NPJ, false>(rmv<NPJ, false>(rhs_u, S, pu), K, -pv);
        qv = rmv<NPJ, false>(rmv<NPJ, false>(rhs_v, K, pu), S, pv);
        if (a.use_shift) {
            qu = fma(-sw, pv, qu);
            qv = fma(sw, pu, qv);
        }
    };
    double cu = rhs_u + bxu, cv = rhs_v + bxv;   // x_1 = rhs + B x_0
    if (a.N >= 3) {
        // ONE evaluation per wave (N = 4: all four 16-lane rows, N = 3: three of them, the idle row holds zeros): the stopping
        // decision is wave-uniform, both norms come out of one reduction (wave_sum2: rows 0, 1 hold ru, rows 2, 3 rv; two
        // sample_sum: ~60 instructions against the ~32 of the four products), the iterates alternate between two register pairs
        for (int it = 1;; it += 2) {
            double nu, nv;
            apply(cu, cv, nu, nv);              // x_{it+1};  residual at x_it = x_it - x_{it+1}
            double du = valid ? cu - nu : 0.0, dv = valid ? cv - nv : 0.0;
            if (__all(wave_sum2(du * du, dv * dv) < tol2) || it >= a.m) break;      // keeps x_it = (cu, cv)
            apply(nu, nv, cu, cv);              // x_{it+2}
            du = valid ? nu - cu : 0.0, dv = valid ? nv - cv : 0.0;
            if (__all(wave_sum2(du * du, dv * dv) < tol2) || it + 1 >= a.m) {       // keeps x_{it+1}
                cu = nu;
                cv = nv;
                break;
            }
        }
        xu = cu;
        xv = cv;
        return;
    }
    bool done = !valid;
    for (int it = 1; it <= a.m; ++it) {
        double nu, nv;
        apply(cu, cv, nu, nv);                  // x_{it+1};  residual at x_it = x_it - x_{it+1}
        const double du = cu - nu, dv = cv - nv;
        const double ru = sample_sum(done ? 0.0 : du * du, a.N, c), rv = sample_sum(done ? 0.0 : dv * dv, a.N, c);
        const bool conv = (ru < tol2) && (rv < tol2);          // sqrt(ru) < tol && sqrt(rv) < tol
        if (!done && !conv && it < a.m) {
            cu = nu;
            cv = nv;
        } else {
            done = true;                        // keeps x_it
        }
        if (__ballot(!done) == 0ull) break;
    }
    xu = cu;
    xv = cv;
}
// one implicit-midpoint step of (u, v) with forcing (fu, fv) already multiplied by h
template <int NPJ>
__device__ __forceinline__ void imr_step(const PropArgs& a, const RowMat<NPJ>& K, const RowMat<NPJ>& S, double sw, double& u,
                                         double& v, double fu, double fv, int c, bool valid)
{
    // B x ONCE: rhs = (x + f) + B x, x_1 = rhs + B x (as the cooperative kernels do)
    double bu = rmv<NPJ, false>(rmv<NPJ, false>(0.0, S, u), K, -v);
    double bv = rmv<NPJ, false>(rmv<NPJ, false>(0.0, K, u), S, v);
    if (a.use_shift) {
        bu = fma(-sw, v, bu);
        bv = fma(sw, u, bv);
    }
    imr_solve<NPJ>(a, K, S, sw, (u + fu) + bu, (v + fv) + bv, bu, bv, u, v, c, valid);
}

// Forward sweep.  a.m = max_iter, a.jacobi_tol2 = tol^2; state file as in jq_rowlane_kernels.h, but the four column
// slots of wave w hold columns w*cpw .. w*cpw + cpw - 1 (slots >= cpw idle).
template <int NPJ>
__global__ __launch_bounds__(64) void k_forward_rowlane_imr(PropArgs a)
{
    const int lane = threadIdx.x;
    const int row = lane & 15;
    const long long w = blockIdx.x, nw = a.nslabs;
    const int c = lane >> 4, cpw = imr_cols_per_wave(a.N);
    const bool valid = c < cpw;
    const long long slot = 4 * w + c;
    const long long col = valid ? w * cpw + c : (long long)1 << 40;
    const double wd = a.tabs[row];
    double* st = a.state + w * 64 + lane;
    double u = st[0], v = st[nw * 64];
    double leak = st[(size_t)(JQ_ROWLANE_ARRAYS + JQ_MAXNC) * nw * 64];
    const double sw = 0.5 * a.h * a.colinfo[slot] * a.tabs[16 + row];
    cmat_t s0 = as_const(a.stream);
    RowMat<NPJ> K = row_load<NPJ>(s0 + 2 * a.stride, row), S = row_load<NPJ>(s0 + 3 * a.stride, row);
    for (int n = 0; n < a.nsteps_chunk; ++n) {
        const int nn = min(n + 1, a.nsteps_chunk - 1);
        const RowMat<NPJ> Kn = row_load<NPJ>(s0 + (size_t)(2 * (2 * nn + 1)) * a.stride, row);       // lands during this step
        const RowMat<NPJ> Sn = row_load<NPJ>(s0 + (size_t)(2 * (2 * nn + 1) + 1) * a.stride, row);
        const double us = u, vs = v;
        imr_step<NPJ>(a, K, S, sw, u, v, 0.0, 0.0, c, valid);
        // penal_m(vr_s, vr, wmat) + penal_m(vi_s, vi, wmat)  (src/evalobjgrad.jl:1214, :2158-2166)
        leak = fma(wd, (us + u) * (us + u) + (vs + v) * (vs + v), leak);
        if (a.hist_r && col < a.N && row < a.Ntot) {
            const size_t off = (size_t)(a.step0 + n + 1) * a.Ntot * a.N + (size_t)col * a.Ntot + row;
            a.hist_r[off] = u;
            a.hist_i[off] = -v;
        }
        K = Kn;
        S = Sn;
    }
    st[0] = u;
    st[nw * 64] = v;
    st[(size_t)(JQ_ROWLANE_ARRAYS + JQ_MAXNC) * nw * 64] = leak;
}

// Backward sweep: state re-integration with h < 0, adjoint m_step! with forcing -W (v + v_s) / T
// (src/evalobjgrad.jl:1290-1336) and the two gradient scalars of adjoint_grad_calc_m per control (:2660-2702),
// written in the trace-record slots of the midpoint weights of k_gradacc:  tr[3] = -(B + C)/4,  tr[4] = (A + D)/4
// so that  gradobjfadj = -dt/4 * sum_steps [(B + C) dp/dalpha + (A + D) dq/dalpha]  (:1338) comes out of h * (...).
template <int NPJ>
__global__ __launch_bounds__(64) void k_backward_rowlane_imr(PropArgs a)
{
    const int lane = threadIdx.x;
    const int row = lane & 15;
    const long long w = blockIdx.x, nw = a.nslabs;
    const int c = lane >> 4, cpw = imr_cols_per_wave(a.N);
    const bool valid = c < cpw;
    const long long slot = 4 * w + c;
    const int Nc = a.Ncoupled;
    const double wd = a.tabs[row];
    double* st = a.state + w * 64 + lane;
    double u = st[0], v = st[nw * 64], lr = st[2 * nw * 64], li = st[3 * nw * 64];
    const double sw = 0.5 * a.h * a.colinfo[slot] * a.tabs[16 + row];
    const double wgt = a.colinfo[4 * nw + slot];
    const double cfw = a.forced ? -a.h * a.tinv * wd : 0.0;      // h * (-tinv * W)
    extern __shared__ double lds_c[];
    constexpr bool RESIDENT = (NPJ <= 8);
    if (!RESIDENT) {
        // transposed copy [image][column j][row]: the 16 lanes of an LDS pass read 16 consecutive doubles (the [row][j] order of the
        // global image gave these reads 4-way bank conflicts at NPJ = 12: 75 % of the LDS cycles)
        for (int i = lane; i < 2 * Nc * (int)a.stride; i += 64) {
            const int im = i / (int)a.stride, e = i - im * (int)a.stride;
            lds_c[im * (int)a.stride + (e % NPJ) * 16 + e / NPJ] = a.cimg[i];
        }
        __syncthreads();
    }
    RowMat<NPJ> Hs[JQ_MAXNC], Ha[JQ_MAXNC];
#pragma unroll
    for (int q = 0; q < JQ_MAXNC; ++q) {
        const int qq = min(q, Nc - 1);
        if (RESIDENT) {
            Hs[q] = row_load<NPJ>(as_const(a.cimg) + (size_t)qq * a.stride, row);
            Ha[q] = row_load<NPJ>(as_const(a.cimg) + (size_t)(Nc + qq) * a.stride, row);
        }
    }
    double* trw = a.traces + ((size_t)w * a.nsteps_chunk) * (Nc * JQ_NTR);
    cmat_t s0 = as_const(a.stream);
    RowMat<NPJ> K = row_load<NPJ>(s0 + 2 * a.stride, row), S = row_load<NPJ>(s0 + 3 * a.stride, row);
    for (int n = 0; n < a.nsteps_chunk; ++n) {
        const int nn = min(n + 1, a.nsteps_chunk - 1);
        const RowMat<NPJ> Kn = row_load<NPJ>(s0 + (size_t)(2 * (2 * nn + 1)) * a.stride, row);
        const RowMat<NPJ> Sn = row_load<NPJ>(s0 + (size_t)(2 * (2 * nn + 1) + 1) * a.stride, row);
        const double us = u, vs = v, lrs = lr, lis = li;
        imr_step<NPJ>(a, K, S, sw, u, v, 0.0, 0.0, c, valid);
        imr_step<NPJ>(a, K, S, sw, lr, li, cfw * (u + us), cfw * (v + vs), c, valid);
        const double smu = lr + lrs, sv = v + vs, snu = li + lis, su = u + us;
#pragma unroll
        for (int q = 0; q < JQ_MAXNC; ++q) {
            if (q < Nc) {
                if (!RESIDENT) {
                    Hs[q] = row_load_lds<NPJ>(lds_c + (size_t)q * a.stride, row);
                    Ha[q] = row_load_lds<NPJ>(lds_c + (size_t)(Nc + q) * a.stride, row);
                }
                const double B = -smu * rmv<NPJ, true>(0.0, Hs[q], sv);
                const double C = snu * rmv<NPJ, true>(0.0, Hs[q], su);
                const double A = smu * rmv<NPJ, true>(0.0, Ha[q], su);
                const double D = snu * rmv<NPJ, true>(0.0, Ha[q], sv);
                const double PQ = wave_sum2((B + C) * wgt, (A + D) * wgt);      // rows 0, 1: P;  rows 2, 3: Q
                double* tr = trw + (size_t)n * (Nc * JQ_NTR) + q * JQ_NTR;
                if (lane == 0) {
                    tr[0] = 0.0;
                    tr[1] = 0.0;
                    tr[2] = 0.0;
                    tr[3] = -0.25 * PQ;
                }
                if (lane == 32) tr[4] = 0.25 * PQ;
            }
        }
        K = Kn;
        S = Sn;
    }
    st[0] = u;
    st[nw * 64] = v;
    st[2 * nw * 64] = lr;
    st[3 * nw * 64] = li;
}

// The backward sweep on TWO waves per wave-load of columns (round 3, like k_backward_rowlane2): wave 0 re-integrates the state
// (its own fixed-point solves), wave 1 runs the adjoint m_step! and the traces of the same time step; the adjoint step needs the
// state step only through the step sums su = u + u_s, sv = v + v_s of its own lanes (forcing and traces), left in a double-buffered
// LDS record under ONE workgroup barrier per time step.  Each chain's arithmetic is unchanged.
// Dynamic LDS: [constant images (NPJ > 8) | records 2 x 2 x 64 doubles].
template <int NPJ>
__global__ __launch_bounds__(128) void k_backward_rowlane_imr2(PropArgs a)
{
    const int lane = threadIdx.x & 63;
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // 0: state chain, 1: adjoint chain
    const int row = lane & 15;
    const long long w = blockIdx.x, nw = a.nslabs;
    const int c = lane >> 4, cpw = imr_cols_per_wave(a.N);
    const bool valid = c < cpw;
    const long long slot = 4 * w + c;
    const int Nc = a.Ncoupled;
    const double wd = a.tabs[row];
    double* st = a.state + w * 64 + lane;
    const double sw = 0.5 * a.h * a.colinfo[slot] * a.tabs[16 + row];
    extern __shared__ double lds_c[];
    constexpr bool RESIDENT = (NPJ <= 8);
    double* rec = lds_c + (RESIDENT ? 0 : (size_t)2 * Nc * a.stride) + lane;      // [slot][su, sv][64]
    if (!RESIDENT) {
        for (int i = threadIdx.x; i < 2 * Nc * (int)a.stride; i += 128) {
            const int im = i / (int)a.stride, e = i - im * (int)a.stride;
            lds_c[im * (int)a.stride + (e % NPJ) * 16 + e / NPJ] = a.cimg[i];
        }
    }
    __syncthreads();
    cmat_t s0 = as_const(a.stream);
    RowMat<NPJ> K = row_load<NPJ>(s0 + 2 * a.stride, row), S = row_load<NPJ>(s0 + 3 * a.stride, row);
    if (role == 0) {
        double u = st[0], v = st[nw * 64];
        for (int n = 0; n < a.nsteps_chunk; ++n) {
            const int nn = min(n + 1, a.nsteps_chunk - 1);
            const RowMat<NPJ> Kn = row_load<NPJ>(s0 + (size_t)(2 * (2 * nn + 1)) * a.stride, row);
            const RowMat<NPJ> Sn = row_load<NPJ>(s0 + (size_t)(2 * (2 * nn + 1) + 1) * a.stride, row);
            const double us = u, vs = v;
            imr_step<NPJ>(a, K, S, sw, u, v, 0.0, 0.0, c, valid);
            double* r = rec + (n & 1) * 128;
            r[0] = u + us;
            r[64] = v + vs;
            K = Kn;
            S = Sn;
            __syncthreads();      // record n is published (record n - 1 has been consumed)
        }
        st[0] = u;
        st[nw * 64] = v;
        return;
    }
    double lr = st[2 * nw * 64], li = st[3 * nw * 64];
    const double wgt = a.colinfo[4 * nw + slot];
    const double cfw = a.forced ? -a.h * a.tinv * wd : 0.0;      // h * (-tinv * W)
    RowMat<NPJ> Hs[JQ_MAXNC], Ha[JQ_MAXNC];
#pragma unroll
    for (int q = 0; q < JQ_MAXNC; ++q) {
        const int qq = min(q, Nc - 1);
        if (RESIDENT) {
            Hs[q] = row_load<NPJ>(as_const(a.cimg) + (size_t)qq * a.stride, row);
            Ha[q] = row_load<NPJ>(as_const(a.cimg) + (size_t)(Nc + qq) * a.stride, row);
        }
    }
    double* trw = a.traces + ((size_t)w * a.nsteps_chunk) * (Nc * JQ_NTR);
    for (int n = 0; n < a.nsteps_chunk; ++n) {
        const int nn = min(n + 1, a.nsteps_chunk - 1);
        const RowMat<NPJ> Kn = row_load<NPJ>(s0 + (size_t)(2 * (2 * nn + 1)) * a.stride, row);
        const RowMat<NPJ> Sn = row_load<NPJ>(s0 + (size_t)(2 * (2 * nn + 1) + 1) * a.stride, row);
        __syncthreads();          // the state wave has published record n
        const double* r = rec + (n & 1) * 128;
        const double su = r[0], sv = r[64];
        const double lrs = lr, lis = li;
        imr_step<NPJ>(a, K, S, sw, lr, li, cfw * su, cfw * sv, c, valid);
        const double smu = lr + lrs, snu = li + lis;
#pragma unroll
        for (int q = 0; q < JQ_MAXNC; ++q) {
            if (q < Nc) {
                if (!RESIDENT) {
                    Hs[q] = row_load_lds<NPJ>(lds_c + (size_t)q * a.stride, row);
                    Ha[q] = row_load_lds<NPJ>(lds_c + (size_t)(Nc + q) * a.stride, row);
                }
                const double B = -smu * rmv<NPJ, true>(0.0, Hs[q], sv);
                const double C = snu * rmv<NPJ, true>(0.0, Hs[q], su);
                const double A = smu * rmv<NPJ, true>(0.0, Ha[q], su);
                const double D = snu * rmv<NPJ, true>(0.0, Ha[q], sv);
                const double PQ = wave_sum2((B + C) * wgt, (A + D) * wgt);      // rows 0, 1: P;  rows 2, 3: Q
                double* tr = trw + (size_t)n * (Nc * JQ_NTR) + q * JQ_NTR;
                if (lane == 0) {
                    tr[0] = 0.0;
                    tr[1] = 0.0;
                    tr[2] = 0.0;
                    tr[3] = -0.25 * PQ;
                }
                if (lane == 32) tr[4] = 0.25 * PQ;
            }
        }
        K = Kn;
        S = Sn;
    }
    st[2 * nw * 64] = lr;
    st[3 * nw * 64] = li;
}
