// jq_aux_kernels.h -- the small kernels around the propagators: on-device control evaluation
// (bcarrier2), K(t)/S(t) tile-stream generation, state initialisation, fidelity + adjoint terminal
// condition, trace reduction and gradient assembly (gradbcarrier2! as an 18-entry scatter).
#pragma once
#include "jq_kernels.h"
#include "jq_rowlane_kernels.h"

struct SplineArgs {
    const double* pcof;   // [nCoeff]
    const double* cfreq;  // [Ncoupled x Nfreq] column-major
    int D1, Nfreq, Ncoupled, nCoeff;
    double dtknot;        // T/(D1-2)            (bcparams, src/bsplines.jl:174)
    const double* rfreq;  // uncoupled controls: params.Rfreq [Ncoupled] (Ncoupled then counts them); nullptr otherwise
};

// knot index k (1-based) of bcarrier2 / gradbcarrier2! (src/bsplines.jl:224-225, :335-336)
__device__ __forceinline__ int knot_index(double t, double dtknot, int D1)
{
    int k = (int)ceil(t / dtknot + 2.0);
    k = k < 3 ? 3 : k;
    k = k > D1 ? D1 : k;
    return k;
}

// bcarrier2(t, params, func): src/bsplines.jl:211-304
__device__ double bcarrier2_dev(const SplineArgs& s, double t, int func)
{
    const int osc = func >> 1, q_func = func & 1;
    const double width = 3.0 * s.dtknot;
    const int k = knot_index(t, s.dtknot, s.D1);
    double f = 0.0;
    for (int freq = 0; freq < s.Nfreq; ++freq) {
        const int offset1 = 2 * osc * s.Nfreq * s.D1 + freq * 2 * s.D1;
        const int offset2 = offset1 + s.D1;
        double fbs1 = 0.0, fbs2 = 0.0, tc, tau, b;
        tc = s.dtknot * ((double)k - 1.5);           // tcenter[k]   (:175, :238)
        tau = (t - tc) / width;
        b = 9.0 / 8.0 + 4.5 * tau + 4.5 * tau * tau;
        fbs1 += s.pcof[offset1 + k - 1] * b;
        fbs2 += s.pcof[offset2 + k - 1] * b;
        tc = s.dtknot * ((double)(k - 1) - 1.5);     // tcenter[k-1] (:244)
        tau = (t - tc) / width;
        b = 0.75 - 9.0 * tau * tau;
        fbs1 += s.pcof[offset1 + k - 2] * b;
        fbs2 += s.pcof[offset2 + k - 2] * b;
        tc = s.dtknot * ((double)(k - 2) - 1.5);     // tcenter[k-2] (:250)
        tau = (t - tc) / width;
        b = 9.0 / 8.0 - 4.5 * tau + 4.5 * tau * tau;
        fbs1 += s.pcof[offset1 + k - 3] * b;
        fbs2 += s.pcof[offset2 + k - 3] * b;
        const double om = s.cfreq[osc + freq * s.Ncoupled];
        double sn, cs;
        sincos(om * t, &sn, &cs);
        if (q_func)
            f += fbs1 * sn + fbs2 * cs;   // :258
        else
            f += fbs1 * cs - fbs2 * sn;   // :260
    }
    return f;
}

// time of stream point j of a chunk: t_n (j even) or t_n + h/2 (j odd); t_n is read from the
// host-accumulated table (t = t + h, src/StormerVerlet.jl:502; the backward table starts at exactly T,
// src/evalobjgrad.jl:811) so that knot indices match the reference to the ulp.
__device__ __forceinline__ double stream_time(const double* tt, int n0, int j, double h)
{
    const double t = tt[n0 + (j >> 1)];
    return (j & 1) ? t + 0.5 * h : t;
}

// pq[j][2*Ncoupled] = (p_1, q_1, p_2, q_2, ...)(t_j)   -- KS!'s controlfunc calls (src/evalobjgrad.jl:2366-2367)
__global__ void k_ctrl(SplineArgs s, const double* tt, int n0, int ntp, double h, double* pq)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ntp) return;
    const double t = stream_time(tt, n0, j, h);
    if (s.rfreq) {
        // uncoupled controls (src/evalobjgrad.jl:2373-2387): ft = 2 (p cos(2 pi Rfreq t) - q sin(2 pi Rfreq t)) multiplies
        // Hunc_ops[q], which sits in the symmetric OR the antisymmetric slot of pair q (the other slot is zero)
        for (int q = 0; q < s.Ncoupled; ++q) {
            const double pt = bcarrier2_dev(s, t, 2 * q), qt = bcarrier2_dev(s, t, 2 * q + 1);
            const double ft = 2.0 * (pt * cos(2.0 * M_PI * s.rfreq[q] * t) - qt * sin(2.0 * M_PI * s.rfreq[q] * t));
            pq[(size_t)j * 2 * s.Ncoupled + 2 * q] = ft;
            pq[(size_t)j * 2 * s.Ncoupled + 2 * q + 1] = ft;
        }
        return;
    }
    for (int f = 0; f < 2 * s.Ncoupled; ++f) pq[(size_t)j * 2 * s.Ncoupled + f] = bcarrier2_dev(s, t, f);
}

// KS!: K = Hconst + sum_q p_q Hsym_q ; S = sum_q q_q Hanti_q  (src/evalobjgrad.jl:2354-2370), evaluated
// on the MFMA tile images and stored pre-scaled for the accumulate-in-place step formulation
// (jq_kernels.h): with c = h/2, K -> +c K at the half time points (odd j), -c K at the integer
// time points (even j); S -> c S.   grid = (mat_elems/256, ntp)
__global__ void k_stream(const double* __restrict__ himg, const double* __restrict__ pq, int Ncoupled, long long mat_elems,
                         double c, double* __restrict__ stream)
{
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (e >= mat_elems) return;
    const double* ctl = pq + (size_t)j * 2 * Ncoupled;
    double K = himg[e], S = 0.0;
    for (int q = 0; q < Ncoupled; ++q) {
        K += ctl[2 * q] * himg[(size_t)(1 + q) * mat_elems + e];
        S += ctl[2 * q + 1] * himg[(size_t)(1 + Ncoupled + q) * mat_elems + e];
    }
    stream[(size_t)(2 * j) * mat_elems + e] = ((j & 1) ? c : -c) * K;
    stream[(size_t)(2 * j + 1) * mat_elems + e] = c * S;
}

// state file <- (Uinit, 0, 0, 0, extras 0); grid = nslabs, block = 64  (N > 16: slab sl holds part sl % parts of its sample)
__global__ void k_init_state(double* state, long long stride, const double* __restrict__ uimg, int KT, int parts)
{
    double* st = state + (size_t)blockIdx.x * stride;
    const int lane = threadIdx.x;
    uimg += (size_t)(blockIdx.x % parts) * KT * 64;
    for (int kk = 0; kk < KT; ++kk) {
        st[kk * 64 + lane] = uimg[kk * 64 + lane];
        st[(KT + kk) * 64 + lane] = 0.0;
        st[(2 * KT + kk) * 64 + lane] = 0.0;
        st[(3 * KT + kk) * 64 + lane] = 0.0;
    }
    for (int r = 0; r < JQ_STATE_EXTRA; ++r) st[(JQ_STATE_ARRAYS * KT + r) * 64 + lane] = 0.0;
}

// Fidelity, leak integral and adjoint terminal condition per sample.
//   s = tr(Vtg' V)/N with ur = vr, ui = -vi (tracefidcomplex, src/evalobjgrad.jl:2078-2084)
//   primaryobjf = 1 - |s|^2 (:759) ; secondaryobjf = dt/2 * tinv * sum(leak partials) (:716-718)
//   lambda(T) (init_adjoint!, :2029-2042)
// res[sample][4] = { primaryobjf, secondaryobjf, Re s, Im s }.   grid = nslabs, block = 64
__global__ void k_terminal(double* state, long long stride, const double* __restrict__ vtr_img,
                           const double* __restrict__ vti_img, int KT, int N, int sps, int nsamples, double leak_scale,
                           double* res)
{
    __shared__ double part[3][64];
    double* st = state + (size_t)blockIdx.x * stride;
    const int lane = threadIdx.x;
    const int col = lane & 15;
    const int sl = (col < sps * N) ? col / N : -1;
    double re = 0.0, im = 0.0;
    for (int kk = 0; kk < KT; ++kk) {
        const double u = st[kk * 64 + lane], v = st[(KT + kk) * 64 + lane];
        const double tr = vtr_img[kk * 64 + lane], ti = vti_img[kk * 64 + lane];
        re += u * tr - v * ti;
        im += u * ti + v * tr;
    }
    part[0][lane] = re;
    part[1][lane] = im;
    part[2][lane] = st[(JQ_STATE_ARRAYS * KT + JQ_MAXNC) * 64 + lane];
    __syncthreads();
    double sre = 0.0, sim = 0.0, slk = 0.0;
    for (int l = 0; l < 64; ++l) {
        const int c2 = l & 15;
        const int s2 = (c2 < sps * N) ? c2 / N : -1;
        if (s2 == sl && sl >= 0) {
            sre += part[0][l];
            sim += part[1][l];
            slk += part[2][l];
        }
    }
    sre /= N;
    sim /= N;
    for (int kk = 0; kk < KT; ++kk) {
        const double tr = vtr_img[kk * 64 + lane], ti = vti_img[kk * 64 + lane];
        st[(2 * KT + kk) * 64 + lane] = (sl >= 0) ? (sre * tr + sim * ti) / N : 0.0;  // lambdar
        st[(3 * KT + kk) * 64 + lane] = (sl >= 0) ? -((sim * tr - sre * ti) / N) : 0.0;  // nb = -lambdai
    }
    const int sample = blockIdx.x * sps + sl;
    if (sl >= 0 && (col % N) == 0 && (lane >> 4) == 0 && sample < nsamples) {
        res[(size_t)sample * 4 + 0] = 1.0 - (sre * sre + sim * sim);
        res[(size_t)sample * 4 + 1] = leak_scale * slk;
        res[(size_t)sample * 4 + 2] = sre;
        res[(size_t)sample * 4 + 3] = sim;
    }
}

// k_terminal for N > 16: the columns of a sample are spread over `parts` consecutive slabs (16 columns each); the trace
// fidelity couples them (the only cross-column coupling of an evaluation, src/evalobjgrad.jl:818, :2029-2041).
// grid = nsamples, block = 64; sums over lanes and parts in a fixed order.
// imr != 0: the implicit-midpoint convention of k_terminal_imr (lambda(T) = -2/N (...), the true lambda_i in slot NU)
__global__ void k_terminal_parts(double* state, long long stride, const double* __restrict__ vtr_img,
                                 const double* __restrict__ vti_img, int KT, int N, int parts, double leak_scale, double* res, int imr)
{
    __shared__ double part[3][64];
    const int lane = threadIdx.x;
    double re = 0.0, im = 0.0, lk = 0.0;
    for (int p = 0; p < parts; ++p) {
        const double* st = state + ((size_t)blockIdx.x * parts + p) * stride;
        const double *tr = vtr_img + (size_t)p * KT * 64, *ti = vti_img + (size_t)p * KT * 64;
        for (int kk = 0; kk < KT; ++kk) {           // (columns beyond N hold zeros in state and target images)
            const double u = st[kk * 64 + lane], v = st[(KT + kk) * 64 + lane];
            re += u * tr[kk * 64 + lane] - v * ti[kk * 64 + lane];
            im += u * ti[kk * 64 + lane] + v * tr[kk * 64 + lane];
        }
        lk += st[(JQ_STATE_ARRAYS * KT + JQ_MAXNC) * 64 + lane];
    }
    part[0][lane] = re;
    part[1][lane] = im;
    part[2][lane] = lk;
    __syncthreads();
    double sre = 0.0, sim = 0.0, slk = 0.0;
    for (int l = 0; l < 64; ++l) {
        sre += part[0][l];
        sim += part[1][l];
        slk += part[2][l];
    }
    sre /= N;
    sim /= N;
    for (int p = 0; p < parts; ++p) {
        double* st = state + ((size_t)blockIdx.x * parts + p) * stride;
        const double *tr = vtr_img + (size_t)p * KT * 64, *ti = vti_img + (size_t)p * KT * 64;
        for (int kk = 0; kk < KT; ++kk) {
            if (imr) {
                st[(2 * KT + kk) * 64 + lane] = -2.0 / N * (sre * tr[kk * 64 + lane] + sim * ti[kk * 64 + lane]);    // lambdar
                st[(3 * KT + kk) * 64 + lane] = -2.0 / N * (-sre * ti[kk * 64 + lane] + sim * tr[kk * 64 + lane]);   // lambdai
            } else {
                st[(2 * KT + kk) * 64 + lane] = (sre * tr[kk * 64 + lane] + sim * ti[kk * 64 + lane]) / N;        // lambdar
                st[(3 * KT + kk) * 64 + lane] = -((sim * tr[kk * 64 + lane] - sre * ti[kk * 64 + lane]) / N);   // nb = -lambdai
            }
        }
    }
    if (lane == 0) {
        res[(size_t)blockIdx.x * 4 + 0] = 1.0 - (sre * sre + sim * sim);
        res[(size_t)blockIdx.x * 4 + 1] = leak_scale * slk;
        res[(size_t)blockIdx.x * 4 + 2] = sre;
        res[(size_t)blockIdx.x * 4 + 3] = sim;
    }
}

// Implicit-midpoint variant of k_terminal (src/evalobjgrad.jl:1221-1271): lambda(T) = -2/N (re Vtr + im Vti),
// -2/N (-re Vti + im Vtr) with the true lambda_i in slot NU; secondaryobjf = dt*tinv/4 * sum(penal_m) (leak_scale).
// Fidelity, leak integral and adjoint terminal condition per sample.
//   s = tr(Vtg' V)/N with ur = vr, ui = -vi (tracefidcomplex, src/evalobjgrad.jl:2078-2084)
//   primaryobjf = 1 - |s|^2 (:759) ; secondaryobjf = dt/2 * tinv * sum(leak partials) (:716-718)
//   lambda(T) (init_adjoint!, :2029-2042)
// res[sample][4] = { primaryobjf, secondaryobjf, Re s, Im s }.   grid = nslabs, block = 64
__global__ void k_terminal_imr(double* state, long long stride, const double* __restrict__ vtr_img,
                           const double* __restrict__ vti_img, int KT, int N, int sps, int nsamples, double leak_scale,
                           double* res)
{
    __shared__ double part[3][64];
    double* st = state + (size_t)blockIdx.x * stride;
    const int lane = threadIdx.x;
    const int col = lane & 15;
    const int sl = (col < sps * N) ? col / N : -1;
    double re = 0.0, im = 0.0;
    for (int kk = 0; kk < KT; ++kk) {
        const double u = st[kk * 64 + lane], v = st[(KT + kk) * 64 + lane];
        const double tr = vtr_img[kk * 64 + lane], ti = vti_img[kk * 64 + lane];
        re += u * tr - v * ti;
        im += u * ti + v * tr;
    }
    part[0][lane] = re;
    part[1][lane] = im;
    part[2][lane] = st[(JQ_STATE_ARRAYS * KT + JQ_MAXNC) * 64 + lane];
    __syncthreads();
    double sre = 0.0, sim = 0.0, slk = 0.0;
    for (int l = 0; l < 64; ++l) {
        const int c2 = l & 15;
        const int s2 = (c2 < sps * N) ? c2 / N : -1;
        if (s2 == sl && sl >= 0) {
            sre += part[0][l];
            sim += part[1][l];
            slk += part[2][l];
        }
    }
    sre /= N;
    sim /= N;
    for (int kk = 0; kk < KT; ++kk) {
        const double tr = vtr_img[kk * 64 + lane], ti = vti_img[kk * 64 + lane];
        st[(2 * KT + kk) * 64 + lane] = (sl >= 0) ? -2.0 / N * (sre * tr + sim * ti) : 0.0;   // lambdar
        st[(3 * KT + kk) * 64 + lane] = (sl >= 0) ? -2.0 / N * (-sre * ti + sim * tr) : 0.0;  // lambdai
    }
    const int sample = blockIdx.x * sps + sl;
    if (sl >= 0 && (col % N) == 0 && (lane >> 4) == 0 && sample < nsamples) {
        res[(size_t)sample * 4 + 0] = 1.0 - (sre * sre + sim * sim);
        res[(size_t)sample * 4 + 1] = leak_scale * slk;
        res[(size_t)sample * 4 + 2] = sre;
        res[(size_t)sample * 4 + 3] = sim;
    }
}

// R[m][k] = sum_slab traces[slab][m][k] in slab order (deterministic).  thread per (m,k)
__global__ void k_trace_reduce(const double* __restrict__ traces, int nslabs, int nsteps_chunk, int ntr, double* R)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long tot = (long long)nsteps_chunk * ntr;
    if (i >= tot) return;
    double s = 0.0;
    for (int sl = 0; sl < nslabs; ++sl) s += traces[(size_t)sl * tot + i];
    R[i] = s;
}

// Gradient assembly for one chunk of backward steps: adjoint_grad_calc! + gradbcarrier2!
// (src/evalobjgrad.jl:2581-2618, src/bsplines.jl:321-415) + gradobjfadj += dt*tr_adj (:898).
// One workgroup (JQ_GRADACC_THREADS threads) per coefficient: the threads stride over the steps of the chunk,
// each (step, time point) contributes only when its knot window covers the coefficient; fixed-order tree
// reduction, so the result is deterministic (the summation order differs from the reference's serial
// `gradobjfadj += dt*tr_adj`, a ~1e-16 relative effect).
#define JQ_GRADACC_THREADS 256
__global__ __launch_bounds__(JQ_GRADACC_THREADS) void k_gradacc(SplineArgs s, const double* __restrict__ R,
                                                                const double* __restrict__ tb, int n0, int nsteps_chunk,
                                                                double h, double* grad, int q0, int ng)
{
    // (q0, ng): the control group [q0, q0 + ng) this backward sweep computed the traces of (R rows: [ng][JQ_NTR] per step);
    // the grid covers the coefficients of those controls only
    __shared__ double red[JQ_GRADACC_THREADS];
    const int per_osc = 2 * s.Nfreq * s.D1;
    const int idx = blockIdx.x + q0 * per_osc;
    const int q = idx / per_osc;
    const int rem = idx - q * per_osc;
    const int freq = rem / (2 * s.D1);
    const int rem2 = rem - freq * 2 * s.D1;
    const int part = rem2 / s.D1;          // 0: offset1 block (alpha_1), 1: offset2 block (alpha_2)
    const int kc = rem2 - part * s.D1 + 1; // 1-based coefficient number within the block
    const double om = s.cfreq[q + freq * s.Ncoupled];
    const double width = 3.0 * s.dtknot;
    const double tc = s.dtknot * ((double)kc - 1.5);
    const int ntr = ng * JQ_NTR;
    double acc = 0.0;
    for (int m = threadIdx.x; m < nsteps_chunk; m += JQ_GRADACC_THREADS) {
        const double t0 = tb[n0 + m];
        const double* r = R + (size_t)m * ntr + (q - q0) * JQ_NTR;
        double step_acc = 0.0;
#pragma unroll
        for (int w = 0; w < 3; ++w) {
            const double tau_t = (w == 0) ? t0 : (w == 1 ? t0 + h : t0 + 0.5 * h);
            const int k = knot_index(tau_t, s.dtknot, s.D1);
            const int jj = k - kc;
            if (jj < 0 || jj > 2) continue;
            double P, Q;
            if (w == 0) {
                P = -r[1];
                Q = -r[0];
            } else if (w == 1) {
                P = -r[1];
                Q = -r[2];
            } else {
                P = r[3];
                Q = -r[4];
            }
            if (s.rfreq) {
                // uncoupled control: Hunc_ops[q] sits in ONE slot of pair q (the other image is zero, its traces too) and is
                // applied with ft = 2 (p cos(2 pi Rfreq t) - q sin(2 pi Rfreq t)): grad ft = 2 cos(.) grad p - 2 sin(.) grad q
                const double B = P + Q;
                double sr, cr;
                sincos(2.0 * M_PI * s.rfreq[q] * tau_t, &sr, &cr);
                P = 2.0 * cr * B;
                Q = -2.0 * sr * B;
            }
            const double tau = (tau_t - tc) / width;
            double b;
            if (jj == 0)
                b = 9.0 / 8.0 + 4.5 * tau + 4.5 * tau * tau;
            else if (jj == 1)
                b = 0.75 - 9.0 * tau * tau;
            else
                b = 9.0 / 8.0 - 4.5 * tau + 4.5 * tau * tau;
            double sn, cs;
            sincos(om * tau_t, &sn, &cs);
            step_acc += b * (part == 0 ? (P * cs + Q * sn) : (-P * sn + Q * cs));
        }
        acc += h * step_acc;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int w = JQ_GRADACC_THREADS / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) grad[idx] += red[0];
}

// Packed result of an ensemble evaluation, left on the device for the collective that sums it over GPUs (one all-reduce):
//   out = [ sum_i w_i primaryobjf_i, sum_i w_i secondaryobjf_i, infidelity gradient [ncoeff], leak gradient [ncoeff] ]
// (src/ipopt_interface.jl:48-59; grad = the weighted gradient sums of the forced [0] and unforced [1] backward sweeps:
// objFuncType == 1: infidelgrad = totalgrad, no leak gradient; else leakgrad = totalgrad - infidelgrad, src/evalobjgrad.jl:947).
// One workgroup, fixed summation order (deterministic).  wq == nullptr: unit weights.
__global__ __launch_bounds__(256) void k_pack(const double* __restrict__ res, const double* __restrict__ wq, int nsamples,
                                              const double* __restrict__ grad, int ncoeff, int adjoint, int two_pass,
                                              double* __restrict__ out)
{
    __shared__ double r0[256], r1[256];
    const int t = threadIdx.x;
    double s0 = 0.0, s1 = 0.0;
    for (int i = t; i < nsamples; i += 256) {
        const double w = wq ? wq[i] : 1.0;
        s0 += w * res[(size_t)i * 4 + 0];
        s1 += w * res[(size_t)i * 4 + 1];
    }
    r0[t] = s0;
    r1[t] = s1;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w) {
            r0[t] += r0[t + w];
            r1[t] += r1[t + w];
        }
        __syncthreads();
    }
    if (t == 0) {
        out[0] = r0[0];
        out[1] = r1[0];
    }
    for (int i = t; i < ncoeff; i += 256) {
        const double g0 = adjoint ? grad[i] : 0.0;
        const double g1 = (adjoint && two_pass) ? grad[(size_t)ncoeff + i] : 0.0;
        out[2 + i] = two_pass ? g1 : g0;
        out[2 + (size_t)ncoeff + i] = two_pass ? g0 - g1 : 0.0;
    }
}

// ---------------------------------------------------------------------------------------------
// Consumers of the state history on the device (jq_state_populations).  hist_r/hist_i: [Ntot][N][nsteps+1]
// column-major (row fastest).
// pop[g + ngroups*(q + N*k)] = sum_{rows r: group(r) == g} |psi[r, q, k*every]|^2.   thread per (q, k)
__global__ void k_pop_groups(const double* __restrict__ hr, const double* __restrict__ hi, int Ntot, int N, int every,
                             int nout, const int* __restrict__ group_of_row, int ngroups, double* __restrict__ pop)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)N * nout) return;
    const int q = (int)(i % N);
    const long long k = i / N;
    const size_t src = ((size_t)k * every * N + q) * Ntot;
    double* out = pop + (size_t)ngroups * i;
    for (int g = 0; g < ngroups; ++g) out[g] = 0.0;
    for (int r = 0; r < Ntot; ++r) {
        const int g = group_of_row ? group_of_row[r] : r;
        if (g < 0) continue;
        const double a = hr[src + r], b = hi[src + r];
        out[g] += a * a + b * b;       // rows of a group are summed in increasing row order, like the reference
    }
}

// maxpop[r] = max_{q, step} |psi[r, q, step]|^2.   block per row
__global__ __launch_bounds__(256) void k_pop_max(const double* __restrict__ hr, const double* __restrict__ hi, int Ntot,
                                                 long long ncolsteps, double* __restrict__ maxpop)
{
    __shared__ double red[256];
    const int r = blockIdx.x;
    double m = 0.0;
    for (long long j = threadIdx.x; j < ncolsteps; j += 256) {
        const double a = hr[(size_t)j * Ntot + r], b = hi[(size_t)j * Ntot + r];
        m = fmax(m, a * a + b * b);
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + w]);
        __syncthreads();
    }
    if (threadIdx.x == 0) maxpop[r] = red[0];
}

// ---------------------------------------------------------------------------------------------
// Row-lane kernels (jq_rowlane_kernels.h): initial state and terminal condition in their state-file layout
// state file <- (Uinit, 0, ...).  uinit: [N][16] (column ic of Uinit, zero padded).  grid = waves, block = 64
// cpw: columns per wave (4, or N*floor(4/N) in the sample-aligned packing of the implicit-midpoint kernels)
__global__ void k_init_state_rowlane(double* state, long long nw, const double* __restrict__ uinit, int N, long long ncols_used,
                                     int cpw)
{
    const int lane = threadIdx.x;
    const long long w = blockIdx.x;
    const int c = lane >> 4;
    const long long col = w * cpw + c;
    for (int r = 0; r < JQ_ROWLANE_ROWS; ++r) {
        double val = 0.0;
        if (r == 0 && c < cpw && col < ncols_used) val = uinit[(col % N) * 16 + (lane & 15)];
        state[((size_t)r * nw + w) * 64 + lane] = val;
    }
}

// fidelity, leak and adjoint terminal condition per sample (thread per sample; see k_terminal)
__global__ void k_terminal_rowlane(double* state, long long nw, const double* __restrict__ vtr, const double* __restrict__ vti,
                                   int N, int nsamples, double leak_scale, double* res)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsamples) return;
    double re = 0.0, im = 0.0, lk = 0.0;
    for (int ic = 0; ic < N; ++ic) {
        const long long col = (long long)s * N + ic;
        const size_t base = (size_t)(col >> 2) * 64 + (size_t)(col & 3) * 16;
        for (int r = 0; r < 16; ++r) {
            const double u = state[base + r], v = state[(size_t)nw * 64 + base + r];
            const double tr = vtr[ic * 16 + r], ti = vti[ic * 16 + r];
            re += u * tr - v * ti;
            im += u * ti + v * tr;
            lk += state[(size_t)(JQ_ROWLANE_ARRAYS + JQ_MAXNC) * nw * 64 + base + r];
        }
    }
    re /= N;
    im /= N;
    for (int ic = 0; ic < N; ++ic) {
        const long long col = (long long)s * N + ic;
        const size_t base = (size_t)(col >> 2) * 64 + (size_t)(col & 3) * 16;
        for (int r = 0; r < 16; ++r) {
            const double tr = vtr[ic * 16 + r], ti = vti[ic * 16 + r];
            state[(size_t)2 * nw * 64 + base + r] = (re * tr + im * ti) / N;       // lambda_r
            state[(size_t)3 * nw * 64 + base + r] = -((im * tr - re * ti) / N);    // nb = -lambda_i
        }
    }
    res[(size_t)s * 4 + 0] = 1.0 - (re * re + im * im);
    res[(size_t)s * 4 + 1] = leak_scale * lk;
    res[(size_t)s * 4 + 2] = re;
    res[(size_t)s * 4 + 3] = im;
}

// implicit-midpoint variant (src/evalobjgrad.jl:1221-1271): secondaryobjf = dt * tinv / 4 * sum(penal_m) (leak_scale),
// lambda(T) = -2/N^2 (s1 Vtr + s2 Vti),  -2/N^2 (-s1 Vti + s2 Vtr)  with s1 + i s2 = N s  (true lambda_i, not negated)
__global__ void k_terminal_rowlane_imr(double* state, long long nw, const double* __restrict__ vtr, const double* __restrict__ vti,
                                       int N, int nsamples, double leak_scale, double* res, int cpw)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsamples) return;
    double re = 0.0, im = 0.0, lk = 0.0;
    for (int ic = 0; ic < N; ++ic) {
        const long long col = (long long)s * N + ic;
        const size_t base = (size_t)(col / cpw) * 64 + (size_t)(col % cpw) * 16;
        for (int r = 0; r < 16; ++r) {
            const double u = state[base + r], v = state[(size_t)nw * 64 + base + r];
            const double tr = vtr[ic * 16 + r], ti = vti[ic * 16 + r];
            re += u * tr - v * ti;
            im += u * ti + v * tr;
            lk += state[(size_t)(JQ_ROWLANE_ARRAYS + JQ_MAXNC) * nw * 64 + base + r];
        }
    }
    re /= N;
    im /= N;
    for (int ic = 0; ic < N; ++ic) {
        const long long col = (long long)s * N + ic;
        const size_t base = (size_t)(col / cpw) * 64 + (size_t)(col % cpw) * 16;
        for (int r = 0; r < 16; ++r) {
            const double tr = vtr[ic * 16 + r], ti = vti[ic * 16 + r];
            state[(size_t)2 * nw * 64 + base + r] = -2.0 / N * (re * tr + im * ti);
            state[(size_t)3 * nw * 64 + base + r] = -2.0 / N * (-re * ti + im * tr);
        }
    }
    res[(size_t)s * 4 + 0] = 1.0 - (re * re + im * im);
    res[(size_t)s * 4 + 1] = leak_scale * lk;
    res[(size_t)s * 4 + 2] = re;
    res[(size_t)s * 4 + 3] = im;
}

