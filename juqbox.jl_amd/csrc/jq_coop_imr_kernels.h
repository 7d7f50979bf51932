// jq_coop_imr_kernels.h -- IMPLICIT MIDPOINT propagators for Ntot > 16 (MFMA), small batches: the row-split
// ("cooperative") mapping of jq_coop_kernels.h -- one slab of 16 columns per workgroup of NT waves, wave mt owns tile
// row mt of every array -- applied to traceobjgrad for Working_Arrays_M (src/evalobjgrad.jl:1042-1481; m_step!,
// src/ImplicitMidpoint.jl:120-227; jacobi_midpoint, src/linear_solvers.jl:156-270).  The algorithm and the exact
// reproduction of the reference's stopping rule are described in jq_rowlane_imr_kernels.h.
//
// The row split is what makes this integrator fit: a step keeps rhs, the current and the next iterate (6 arrays) on top
// of the state, the adjoint and their step sums; per wave an array is ONE d4 (8 registers).
//
// LDS: [K(t+h/2) image | S(t+h/2) image | tables wd, ws | two x exchange buffers | norm partials 2 x NT x 64].
// Both operator images of a step stay resident for all products of its fixed-point iterations (they are fetched once
// per step; the two slots are re-used for the (Hsym_q, Hanti_q) pair of every control after the adjoint solve).
// Per-evaluation convergence: every wave writes its rows' partial sums of squares per lane, and after one barrier
// every lane adds up the NT x 4 x N entries of its own evaluation -- all waves reach the same decision.
// Ntot > 96 (NT = 7 .. 16, round 3): the images do not fit the LDS; like the Stormer-Verlet BIG variants (jq_coop_kernels.h) a
// wave reads its own tile row of K(t+h/2), S(t+h/2) -- and of the trace images -- straight from the tile stream in HBM / L2 for
// every product (the ~8 products of a step with the same two images hit L2); LDS holds the tables, the exchange buffers and the
// norm partials only.
#pragma once
#include "jq_coop_kernels.h"

template <int NT, int BW, bool BIG = (NT > 6)>
struct CoopImr {
    Coop<NT, BW, BIG> c;     // exchange buffers + product (c.M is pointed at the wanted image by hand; its ring is unused)
    char* smem;
    const double* stream;
    const double* cimg;
    long long stride;
    int pieces, wave, lane;
    double* normbuf;         // [2][NT][64]
    const double* Kimg;      // my row's tiles of the two resident images (lane offset applied)
    const double* Simg;

    __device__ __forceinline__ void setup(char* smem_, const PropArgs& a, int wave_, int lane_)
    {
        smem = smem_;
        stream = a.stream;
        cimg = a.cimg;
        stride = a.stride;
        pieces = a.pieces;
        wave = wave_;
        lane = lane_;
        double* tab = (double*)(smem + a.lds_tab_off);
        c.xbuf = tab + 32 * NT + lane;
        c.xcur = 0;
        c.mt = wave;
        int k = (BW == JQ_BW_OD) ? wave : wave - BW;
        if (k < 0) k = 0;
        if (k > NT - coop_nb(NT, BW)) k = NT - coop_nb(NT, BW);
        c.kb0 = k;
        c.row_off = wave * coop_row_elems(NT, BW);
        c.xown = (d4){0.0, 0.0, 0.0, 0.0};
        c.M = nullptr;
        normbuf = tab + 32 * NT + 2 * (4 * NT * 64);
        Kimg = (const double*)smem + c.row_off + lane;
        Simg = Kimg + stride;
    }
    // the two images of the next products: `s0`, `s1` in global memory.  LDS variants: fetched into the two slots (all waves take
    // part; ends with a barrier); BIG: the products read them where they are.
    __device__ __forceinline__ void use_images(const double* s0, const double* s1)
    {
        if constexpr (BIG) {
            Kimg = s0 + c.row_off + lane;
            Simg = s1 + c.row_off + lane;
        } else {
            __syncthreads();                               // every wave is done with the previous pair
            unsigned lo;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lo));
            const char* p0 = (const char*)s0 + lo * 16u;
            const char* p1 = (const char*)s1 + lo * 16u;
            for (int p = wave; p < pieces; p += NT) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p0 + (size_t)p * 1024),
                                                 (__attribute__((address_space(3))) void*)(smem + p * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p1 + (size_t)p * 1024),
                                                 (__attribute__((address_space(3))) void*)(smem + stride * 8 + p * 1024), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    // fetch two consecutive images (2*pieces KiB) into the two slots; all waves take part; ends with a barrier
    __device__ __forceinline__ void load_pair(const double* src)
    {
        if constexpr (BIG) {
            use_images(src, src + stride);
            return;
        }
        __syncthreads();                                   // every wave is done with the previous pair
        unsigned lo;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lo));
        const char* s = (const char*)src + lo * 16u;
        for (int p = wave; p < 2 * pieces; p += NT)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + (size_t)p * 1024),
                                             (__attribute__((address_space(3))) void*)(smem + p * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // M * (published x) with the K / S image
    __device__ __forceinline__ d4 mulK()
    {
        c.M = Kimg;
        return c.mm_z();
    }
    __device__ __forceinline__ d4 mulS()
    {
        c.M = Simg;
        return c.mm_z();
    }
    __device__ __forceinline__ void pub(const d4& x)
    {
        c.stage(x);
        c.publish();
    }
    // (S pu - K pv - sw.*pv,  K pu + S pv + sw.*pu): the action of h/2 [S -K; K S] (+ the sample's diagonal shift)
    __device__ __forceinline__ void applyB(const PropArgs& a, const d4& sw, const d4& pu, const d4& pv, d4& bu, d4& bv)
    {
        pub(pu);
        bu = mulS();
        bv = mulK();
        pub(pv);
        bu -= mulK();
        bv += mulS();
        if (a.use_shift) {
            bu -= sw * pv;
            bv += sw * pu;
        }
    }
    // per-evaluation sums of squares (u and v parts) of d over the NT waves; result valid in every lane of every wave
    __device__ __forceinline__ void sample_norms(const d4& du, const d4& dv, bool mask, int N, double& ru, double& rv)
    {
        const d4 pu = du * du, pv = dv * dv;
        normbuf[wave * 64 + lane] = mask ? 0.0 : (pu[0] + pu[1]) + (pu[2] + pu[3]);
        normbuf[(NT + wave) * 64 + lane] = mask ? 0.0 : (pv[0] + pv[1]) + (pv[2] + pv[3]);
        __syncthreads();
        const int c0 = min(((lane & 15) / N) * N, 16 - N);  // first column of my evaluation (idle columns: clamped)
        ru = 0.0;
        rv = 0.0;
        for (int w = 0; w < NT; ++w)
            for (int g = 0; g < 4; ++g)
                for (int cc = 0; cc < N; ++cc) {
                    ru += normbuf[w * 64 + g * 16 + c0 + cc];
                    rv += normbuf[(NT + w) * 64 + g * 16 + c0 + cc];
                }
        __syncthreads();                                   // normbuf may be rewritten
    }
    // one implicit-midpoint step of (u, v); fu, fv: forcing already multiplied by h; valid: my column holds an evaluation
    __device__ __forceinline__ void step(const PropArgs& a, const d4& sw, d4& u, d4& v, const d4& fu, const d4& fv, bool valid)
    {
        d4 bu, bv;
        applyB(a, sw, u, v, bu, bv);
        const d4 rhs_u = (u + fu) + bu, rhs_v = (v + fv) + bv;
        d4 cu = rhs_u + bu, cv = rhs_v + bv;               // x_1 = rhs + B x_0
        bool done = !valid;
        for (int it = 1; it <= a.m; ++it) {
            applyB(a, sw, cu, cv, bu, bv);
            const d4 nu = rhs_u + bu, nv = rhs_v + bv;      // x_{it+1}; residual at x_it = x_it - x_{it+1}
            double ru, rv;
            sample_norms(cu - nu, cv - nv, done, a.N, ru, rv);
            const bool conv = (ru < a.jacobi_tol2) && (rv < a.jacobi_tol2);
            if (!done && !conv && it < a.m) {
                cu = nu;
                cv = nv;
            } else {
                done = true;
            }
            // every wave sees the same per-evaluation sums, so this exit is uniform over the workgroup
            if (__ballot(!done) == 0ull) break;
        }
        u = cu;
        v = cv;
    }
};

// dynamic LDS of the cooperative implicit-midpoint kernels (bytes)
__host__ __device__ inline size_t coop_imr_lds_bytes(int NT, long long stride)
{
    if (NT > 6) stride = 0;      // (images from HBM: no operator slots; the caller passes 0 for the <6, 5, true> variant as well)
    return (size_t)2 * stride * 8 + (size_t)32 * NT * 8 + (size_t)2 * (4 * NT * 64) * 8 + (size_t)2 * NT * 64 * 8;
}

template <int NT, int BW, bool HBM = (NT > 6)>
__global__ __launch_bounds__(64 * NT) void k_forward_coop_imr(PropArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KT = 4 * NT;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4;
    const int slab = blockIdx.x;
    double* tab = (double*)(smem + a.lds_tab_off);
    for (int i = threadIdx.x; i < 32 * NT; i += blockDim.x) tab[(i & ~15) + 4 * (i & 3) + ((i >> 2) & 3)] = a.tabs[i];   // [block][g][r]
    CoopImr<NT, BW, HBM> m;
    m.setup(smem, a, wave, lane);
    __syncthreads();
    const d4 wdr = rows4(tab, wave, g), wsr = rows4(tab + 16 * NT, wave, g);
    double* st = a.state + (size_t)slab * a.state_stride;
    d4 u, v;
    for (int r = 0; r < 4; ++r) {
        u[r] = st[(4 * wave + r) * 64 + lane];
        v[r] = st[(KT + 4 * wave + r) * 64 + lane];
    }
    double leak = 0.0;
    const d4 sw = (0.5 * a.h * a.colinfo[(size_t)slab * 32 + (lane & 15)]) * wsr;
    const bool valid = (lane & 15) < (16 / a.N) * a.N;
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    for (int n = 0; n < a.nsteps_chunk; ++n) {
        m.load_pair(a.stream + (size_t)(2 * (2 * n + 1)) * a.stride);
        const d4 us = u, vs = v;
        m.step(a, sw, u, v, zero, zero, valid);
        leak += dot4(wdr, (us + u) * (us + u) + (vs + v) * (vs + v));   // penal_m (src/evalobjgrad.jl:1214, :2158-2166)
        if (a.hist_r) {
            const int col = lane & 15;
            if (slab == 0 && col < a.N) {
                const size_t off = (size_t)(a.step0 + n + 1) * a.Ntot * a.N + (size_t)col * a.Ntot;
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * wave + 4 * r + g;
                    if (row < a.Ntot) {
                        a.hist_r[off + row] = u[r];
                        a.hist_i[off + row] = -v[r];
                    }
                }
            }
        }
    }
    for (int r = 0; r < 4; ++r) {
        st[(4 * wave + r) * 64 + lane] = u[r];
        st[(KT + 4 * wave + r) * 64 + lane] = v[r];
    }
    wg_sum_store<NT>(leak, m.normbuf, &st[(JQ_STATE_ARRAYS * KT + JQ_MAXNC) * 64], wave, lane, true);
}

// Backward sweep (src/evalobjgrad.jl:1290-1336); trace records per wave as in k_backward_coop, in the slots of the
// midpoint weights of k_gradacc (jq_rowlane_imr_kernels.h): tr[3] = -(B + C)/4, tr[4] = (A + D)/4.
template <int NT, int BW, bool HBM = (NT > 6)>
__global__ __launch_bounds__(64 * NT) void k_backward_coop_imr(PropArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KT = 4 * NT;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4;
    const int slab = blockIdx.x;
    const int Nc = a.Ncoupled;
    double* tab = (double*)(smem + a.lds_tab_off);
    for (int i = threadIdx.x; i < 32 * NT; i += blockDim.x) tab[(i & ~15) + 4 * (i & 3) + ((i >> 2) & 3)] = a.tabs[i];   // [block][g][r]
    CoopImr<NT, BW, HBM> m;
    m.setup(smem, a, wave, lane);
    __syncthreads();
    const d4 wdr = rows4(tab, wave, g), wsr = rows4(tab + 16 * NT, wave, g);
    double* st = a.state + (size_t)slab * a.state_stride;
    d4 u, v, lr, li;
    for (int r = 0; r < 4; ++r) {
        u[r] = st[(4 * wave + r) * 64 + lane];
        v[r] = st[(KT + 4 * wave + r) * 64 + lane];
        lr[r] = st[(2 * KT + 4 * wave + r) * 64 + lane];
        li[r] = st[(3 * KT + 4 * wave + r) * 64 + lane];
    }
    const d4 sw = (0.5 * a.h * a.colinfo[(size_t)slab * 32 + (lane & 15)]) * wsr;
    const double wgt = a.colinfo[(size_t)slab * 32 + 16 + (lane & 15)];
    const d4 cfw = (a.forced ? -a.h * a.tinv : 0.0) * wdr;            // h * (-tinv * W)
    const bool valid = (lane & 15) < (16 / a.N) * a.N;
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    double* trw = a.traces + ((size_t)(slab * NT + wave) * a.nsteps_chunk) * (Nc * JQ_NTR);
    for (int n = 0; n < a.nsteps_chunk; ++n) {
        m.load_pair(a.stream + (size_t)(2 * (2 * n + 1)) * a.stride);
        const d4 us = u, vs = v, lrs = lr, lis = li;
        m.step(a, sw, u, v, zero, zero, valid);
        const d4 su = u + us, sv = v + vs;
        m.step(a, sw, lr, li, cfw * su, cfw * sv, valid);
        const d4 smu = lr + lrs, snu = li + lis;
        for (int q = 0; q < Nc; ++q) {
            // the two slots now take (Hsym_q, Hanti_q): a.cimg = [Hsym_0.. | Hanti_0..] -> two separate fetches
            m.use_images(a.cimg + (size_t)q * a.stride, a.cimg + (size_t)(Nc + q) * a.stride);
            m.pub(sv);
            const double B = -dot4(smu, m.mulK());       // slot 0 holds Hsym_q
            const double D = dot4(snu, m.mulS());        // slot 1 holds Hanti_q
            m.pub(su);
            const double C = dot4(snu, m.mulK());
            const double A = dot4(smu, m.mulS());
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
    for (int r = 0; r < 4; ++r) {
        st[(4 * wave + r) * 64 + lane] = u[r];
        st[(KT + 4 * wave + r) * 64 + lane] = v[r];
        st[(2 * KT + 4 * wave + r) * 64 + lane] = lr[r];
        st[(3 * KT + 4 * wave + r) * 64 + lane] = li[r];
    }
}

// ---------------------------------------------------------------------------------------------
// N > 16 columns per evaluation (round 4).  The fixed-point solver stops on the residual norm of the WHOLE evaluation
// (jacobi_midpoint, src/linear_solvers.jl:156-270: norms over the Ntot x N block), so the `parts` = ceil(N / 16) slabs of an
// evaluation cannot iterate in different workgroups.  Here ONE workgroup of NT waves owns an evaluation and walks over its parts
// inside every phase of a step; what a part needs between phases -- right-hand side, current and next iterate, the step sums for
// the forcing and the traces -- lives in a work area in HBM / L2 (10 arrays per part), the state file is read and written every
// step.  Same products (CoopImr::applyB), same iterates and decisions as the one-slab kernels; a correctness-first path (every
// phase pays the L2 latency per part), for the sizes the reference allows and no example uses.
//   work area per evaluation: [parts][10][KT * 64]: rhs_u, rhs_v, cur_u, cur_v, nxt_u, nxt_v, su, sv, smu, snu
#define JQ_IMRP_ARRAYS 10
template <int NT, int BW, bool HBM>
struct ImrParts {
    CoopImr<NT, BW, HBM> m;
    double* work;      // this evaluation's work area (lane offset applied)
    double* st0;       // state file of part 0 (lane offset applied)
    long long sstride; // doubles between the parts' state files
    int parts, wave, lane;
    static constexpr int KT = 4 * NT;
    __device__ __forceinline__ double* warr(int p, int k) const { return work + ((size_t)p * JQ_IMRP_ARRAYS + k) * (KT * 64) + (size_t)(4 * wave) * 64; }
    __device__ __forceinline__ double* sarr(int p, int k) const { return st0 + (size_t)p * sstride + (size_t)(k * KT + 4 * wave) * 64; }
    static __device__ __forceinline__ d4 ld(const double* a) { return (d4){a[0], a[64], a[128], a[192]}; }
    static __device__ __forceinline__ void sto(double* a, const d4& x) { a[0] = x[0], a[64] = x[1], a[128] = x[2], a[192] = x[3]; }
    // totals of two per-lane partials over the workgroup (wave order), valid in every lane of every wave
    __device__ __forceinline__ void wg_total2(double pu, double pv, double& ru, double& rv)
    {
        const double t = wave_sum2(pu, pv);      // rows 0, 1: total of pu;  rows 2, 3: total of pv
        if (lane == 0) m.normbuf[wave] = t;
        if (lane == 32) m.normbuf[NT + wave] = t;
        __syncthreads();
        ru = 0.0, rv = 0.0;
        for (int w = 0; w < NT; ++w) {
            ru += m.normbuf[w];
            rv += m.normbuf[NT + w];
        }
        __syncthreads();
    }
    // One implicit-midpoint step of all parts of the evaluation: x <- solution of (I - B) x = x + f + B x with f_p = cf * (work
    // array fk_u / fk_v of the part) or 0; in / out: the state-file arrays (ku, kv) of every part.  On return the work arrays su / sv
    // (SUMS = 0) or smu / snu (SUMS = 1) hold new + old.
    template <int SUMS>
    __device__ __forceinline__ void step(const PropArgs& a, const d4& sw, int ku, int kv, bool forced, const d4& cf)
    {
        for (int p = 0; p < parts; ++p) {
            const d4 u = ld(sarr(p, ku)), v = ld(sarr(p, kv));
            d4 fu = {0.0, 0.0, 0.0, 0.0}, fv = fu;
            if (forced) fu = cf * ld(warr(p, 6)), fv = cf * ld(warr(p, 7));
            d4 bu, bv;
            m.applyB(a, sw, u, v, bu, bv);
            const d4 rhs_u = (u + fu) + bu, rhs_v = (v + fv) + bv;
            sto(warr(p, 0), rhs_u), sto(warr(p, 1), rhs_v);
            sto(warr(p, 2), rhs_u + bu), sto(warr(p, 3), rhs_v + bv);      // x_1 = rhs + B x_0
        }
        int cur = 2, nxt = 4;      // work arrays of the current / next iterate
        bool done = false;
        for (int it = 1; it <= a.m && !done; ++it) {
            double pu = 0.0, pv = 0.0;
            for (int p = 0; p < parts; ++p) {
                const d4 cu = ld(warr(p, cur)), cv = ld(warr(p, cur + 1));
                d4 bu, bv;
                m.applyB(a, sw, cu, cv, bu, bv);
                const d4 nu = ld(warr(p, 0)) + bu, nv = ld(warr(p, 1)) + bv;      // x_{it+1}; residual at x_it = x_it - x_{it+1}
                sto(warr(p, nxt), nu), sto(warr(p, nxt + 1), nv);
                const d4 du = cu - nu, dv = cv - nv;
                pu += dot4(du, du);
                pv += dot4(dv, dv);
            }
            double ru, rv;
            wg_total2(pu, pv, ru, rv);
            const bool conv = (ru < a.jacobi_tol2) && (rv < a.jacobi_tol2);
            if (!conv && it < a.m) {
                const int t = cur;
                cur = nxt, nxt = t;
            } else {
                done = true;      // (the same decision in every lane of the workgroup: the totals are)
            }
        }
        for (int p = 0; p < parts; ++p) {
            const d4 xu = ld(warr(p, cur)), xv = ld(warr(p, cur + 1));
            const d4 ou = ld(sarr(p, ku)), ov = ld(sarr(p, kv));
            sto(warr(p, SUMS ? 8 : 6), xu + ou), sto(warr(p, SUMS ? 9 : 7), xv + ov);
            sto(sarr(p, ku), xu), sto(sarr(p, kv), xv);
        }
    }
};

// grid = nsamples (one workgroup per evaluation), block = 64 NT
template <int NT, int BW, bool HBM = (NT > 6)>
__global__ __launch_bounds__(64 * NT) void k_forward_coop_imr_parts(PropArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KT = 4 * NT;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4;
    const int smp = blockIdx.x;
    double* tab = (double*)(smem + a.lds_tab_off);
    for (int i = threadIdx.x; i < 32 * NT; i += blockDim.x) tab[(i & ~15) + 4 * (i & 3) + ((i >> 2) & 3)] = a.tabs[i];   // [block][g][r]
    ImrParts<NT, BW, HBM> ip;
    ip.m.setup(smem, a, wave, lane);
    ip.parts = a.parts, ip.wave = wave, ip.lane = lane, ip.sstride = a.state_stride;
    ip.st0 = a.state + (size_t)smp * a.parts * a.state_stride + lane;
    ip.work = a.park + (size_t)smp * a.parts * JQ_IMRP_ARRAYS * (KT * 64) + lane;
    __syncthreads();
    const d4 wdr = rows4(tab, wave, g), wsr = rows4(tab + 16 * NT, wave, g);
    const d4 sw = (0.5 * a.h * a.colinfo[(size_t)smp * a.parts * 32 + (lane & 15)]) * wsr;      // (one evaluation: the same shift in every part)
    double* leakp = a.state + (size_t)smp * a.parts * a.state_stride + (size_t)(JQ_STATE_ARRAYS * KT + JQ_MAXNC) * 64 + lane;   // part 0's row
    double leak = 0.0;
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    for (int n = 0; n < a.nsteps_chunk; ++n) {
        ip.m.load_pair(a.stream + (size_t)(2 * (2 * n + 1)) * a.stride);
        ip.template step<0>(a, sw, 0, 1, false, zero);
        for (int p = 0; p < a.parts; ++p) {
            const d4 su = ip.ld(ip.warr(p, 6)), sv = ip.ld(ip.warr(p, 7));
            leak += dot4(wdr, su * su + sv * sv);      // penal_m (src/evalobjgrad.jl:1214, :2158-2166)
            if (a.hist_r && smp == 0) {
                const int col = 16 * p + (lane & 15);
                if (col < a.N) {
                    const d4 u = ip.ld(ip.sarr(p, 0)), v = ip.ld(ip.sarr(p, 1));
                    const size_t off = (size_t)(a.step0 + n + 1) * a.Ntot * a.N + (size_t)col * a.Ntot;
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * wave + 4 * r + g;
                        if (row < a.Ntot) {
                            a.hist_r[off + row] = u[r];
                            a.hist_i[off + row] = -v[r];
                        }
                    }
                }
            }
        }
    }
    // the waves' leak partials are combined in wave order into part 0's row of the state file (k_terminal_parts adds the parts' rows)
    wg_sum_store<NT>(leak, ip.m.normbuf, leakp - lane, wave, lane, true);
}

template <int NT, int BW, bool HBM = (NT > 6)>
__global__ __launch_bounds__(64 * NT) void k_backward_coop_imr_parts(PropArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KT = 4 * NT;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4;
    const int smp = blockIdx.x;
    const int Nc = a.Ncoupled;
    double* tab = (double*)(smem + a.lds_tab_off);
    for (int i = threadIdx.x; i < 32 * NT; i += blockDim.x) tab[(i & ~15) + 4 * (i & 3) + ((i >> 2) & 3)] = a.tabs[i];   // [block][g][r]
    ImrParts<NT, BW, HBM> ip;
    ip.m.setup(smem, a, wave, lane);
    ip.parts = a.parts, ip.wave = wave, ip.lane = lane, ip.sstride = a.state_stride;
    ip.st0 = a.state + (size_t)smp * a.parts * a.state_stride + lane;
    ip.work = a.park + (size_t)smp * a.parts * JQ_IMRP_ARRAYS * (KT * 64) + lane;
    __syncthreads();
    const d4 wdr = rows4(tab, wave, g), wsr = rows4(tab + 16 * NT, wave, g);
    const d4 sw = (0.5 * a.h * a.colinfo[(size_t)smp * a.parts * 32 + (lane & 15)]) * wsr;
    const double wgt = a.colinfo[(size_t)smp * a.parts * 32 + 16 + (lane & 15)];
    const d4 cfw = (a.forced ? -a.h * a.tinv : 0.0) * wdr;            // h * (-tinv * W)
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    double* trw = a.traces + ((size_t)(smp * NT + wave) * a.nsteps_chunk) * (Nc * JQ_NTR);
    for (int n = 0; n < a.nsteps_chunk; ++n) {
        ip.m.load_pair(a.stream + (size_t)(2 * (2 * n + 1)) * a.stride);
        ip.template step<0>(a, sw, 0, 1, false, zero);        // state: su, sv = new + old
        ip.template step<1>(a, sw, 2, 3, true, cfw);          // adjoint with the forcing h (-tinv W) (su, sv): smu, snu
        for (int q = 0; q < Nc; ++q) {
            ip.m.use_images(a.cimg + (size_t)q * a.stride, a.cimg + (size_t)(Nc + q) * a.stride);
            double PB = 0.0, PA = 0.0;      // per-lane partials of (B + C) and (A + D) over the parts
            for (int p = 0; p < a.parts; ++p) {
                const d4 su = ip.ld(ip.warr(p, 6)), sv = ip.ld(ip.warr(p, 7)), smu = ip.ld(ip.warr(p, 8)), snu = ip.ld(ip.warr(p, 9));
                ip.m.pub(sv);
                const double B = -dot4(smu, ip.m.mulK());       // slot 0 holds Hsym_q
                const double D = dot4(snu, ip.m.mulS());        // slot 1 holds Hanti_q
                ip.m.pub(su);
                const double C = dot4(snu, ip.m.mulK());
                const double A = dot4(smu, ip.m.mulS());
                PB += B + C;
                PA += A + D;
            }
            const double PQ = wave_sum2(PB * wgt, PA * wgt);      // rows 0, 1: P;  rows 2, 3: Q
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
}
