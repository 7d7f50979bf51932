// jq_coop_kernels.h -- "cooperative" (row-split) variants of the propagators for SMALL batches.
//
// The slab kernels of jq_kernels.h give one wave a whole 16-column slab: maximal MFMA efficiency, but one
// evaluation is a chain of 72 * nsteps dependent products of NT*KT-band MFMAs each, and the time of an
// ensemble evaluation is flat (cnot3: 4.9 s) until every wave of the chip has its own slab.  When there are
// fewer slabs than that, these kernels split ONE slab over the NT waves of a workgroup instead: wave `mt`
// owns tile row mt (16 rows) of every state array (a single d4 = 8 registers per array), computes its
// rows of each product  D[mt] = C[mt] + sum_kb M[mt,kb] x[kb]  and publishes them to an LDS exchange buffer
// from which all waves read the next product's B operands.  A product is then 4*NB MFMAs per wave
// (NB = min(2*BW+1, NT) k-blocks; cnot3: 12 instead of 64) plus one workgroup barrier.
//
// Operator image layout ("row-window"): for tile row mt the NB k-blocks kb0(mt) .. kb0(mt)+NB-1,
// kb0 = min(max(mt-BW, 0), NT-NB), 4 tiles each, rows consecutively.  Blocks inside the window but outside
// the band are simply stored (they are zero).  Same math, schedule bit-fields, state file, tile stream
// generator and reductions as the slab kernels.
#pragma once
#include "jq_kernels.h"
#include <type_traits>

// BW == JQ_BW_OD (jq_kernels.h): the window is the row's own diagonal block (4 MFMA tiles); the two neighbouring
// blocks are diagonal matrices, stored as 2 x 16 coefficients behind the tiles and applied with 8 FMAs.
// (band code 15 = dense for every NT: NT > 16 -- jq_huge_kernels.h -- must not get a 31-block window out of it)
__host__ __device__ constexpr int coop_nb(int NT, int BW) { return (BW == JQ_BW_OD) ? 1 : (BW == 15) ? NT : ((2 * BW + 1 < NT) ? 2 * BW + 1 : NT); }
__host__ __device__ constexpr int coop_row_elems(int NT, int BW) { return 4 * coop_nb(NT, BW) * 64 + (BW == JQ_BW_OD ? 32 : 0); }
__host__ __device__ constexpr int coop_kb0(int NT, int BW, int mt)
{
    if (BW == JQ_BW_OD) return mt;
    const int nb = coop_nb(NT, BW);
    int k = mt - BW;
    if (k < 0) k = 0;
    if (k > NT - nb) k = NT - nb;
    return k;
}
__host__ __device__ constexpr int coop_tiles(int NT, int BW) { return NT * 4 * coop_nb(NT, BW); }
// Operators straight from HBM / L2 into the MFMA's A registers instead of through LDS slots: more than six tile rows -- and, round 6, the
// dense 96 x 96 case (NT = 6, band 5), whose two 72 KiB slots never fitted the LDS next to the exchange buffers: its single evaluations
// and small ensembles ran on the slab kernels, ONE wave per 16 columns doing all six tile rows (259 us per time step; here: six waves)
__host__ __device__ constexpr bool coop_hbm(int NT, int BW) { return NT > 6 || (NT == 6 && BW == 5); }
// 1 KiB pieces of an operator image per wave of the workgroup when they divide evenly (else 0: the generic DMA loop)
__host__ __device__ constexpr int coop_image_pieces(int NT, int BW) { return (NT * coop_row_elems(NT, BW) + 127) / 128; }
__host__ __device__ constexpr int coop_dma_kper(int NT, int BW) { return coop_image_pieces(NT, BW) % NT == 0 ? coop_image_pieces(NT, BW) / NT : 0; }

// D = C + M[mt, window] * x[window]  (ZEROC: C = 0).  Mrow: my row's tiles in LDS (lane offset applied);
// x: exchange buffer [4*NT][64] (lane offset applied), kb0 = first k-block of my window.
// PF: tiles in flight ahead of their MFMA (LDS operands: JQ_PF; operands read straight from HBM / L2 -- the BIG variants
// below -- need a deeper FIFO for the longer latency)
template <int NT, int BW, bool ZEROC, int PF = JQ_PF>
__device__ __forceinline__ d4 cmm(const d4& C, const double* __restrict__ Mrow, const double* x, int kb0)
{
    constexpr int NTL = 4 * coop_nb(NT, BW);
    const double* xs = x + (size_t)kb0 * 4 * 64;
    double fa[PF], fb[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i)
        if (i < NTL) {
            fa[i] = Mrow[i * 64];
            fb[i] = xs[i * 64];
        }
    d4 acc = ZEROC ? (d4){0.0, 0.0, 0.0, 0.0} : C;
#pragma unroll
    for (int i = 0; i < NTL; ++i) {
        const double a = fa[i % PF], b = fb[i % PF];
        if (i + PF < NTL) {
            fa[i % PF] = Mrow[(i + PF) * 64];
            fb[i % PF] = xs[(i + PF) * 64];
        }
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    return acc;
}

// Operator sequencing of the BIG variants (Ntot > 96: NT = 7 .. 16 waves per slab): the images do not go through LDS at all
// -- a wave only ever reads ITS OWN tile row of an operator, so it loads those tiles from the tile stream in HBM (shared by
// all workgroups through L2) straight into the MFMA's A registers.  Same schedule words as Ring, no DMA, no barrier.
struct OpCursor {
    const double* stream;
    const double* cimg;
    unsigned long long sb0, sb1, sb2, pb;
    long long stride;
    int period, npro, Q, np, ip;
    __device__ __forceinline__ void init(char*, const PropArgs& a, int, int, int)
    {
        stream = a.stream, cimg = a.cimg;
        sb0 = a.sched_bits[0], sb1 = a.sched_bits[1], sb2 = a.sched_bits[2], pb = a.pro_bits;
        stride = a.stride, period = a.period, npro = a.npro;
        Q = 0, np = 0, ip = 0;
    }
    __device__ __forceinline__ const double* next()
    {
        unsigned long long w;
        int k = ip;
        if (Q < npro) {
            w = pb;
            k = Q;
        } else if (ip < 10) {
            w = sb0;
        } else if (ip < 20) {
            w = sb1;
            k = ip - 10;
        } else {
            w = sb2;
            k = ip - 20;
        }
        const unsigned e = (unsigned)(w >> (6 * k)) & 63u, kind = e & 3u, tp = e >> 2;
        const double* M = (kind == 2) ? cimg + (size_t)tp * stride : stream + ((size_t)(2 * (2 * np + tp)) + kind) * stride;
        if (Q >= npro) {
            if (++ip == period) {
                ip = 0;
                ++np;
            }
        }
        ++Q;
        return M + (threadIdx.x & 63);
    }
    __device__ __forceinline__ void drain() {}
};

// JQ_BW_OD:  D = C + Mdiag[mt] x[mt] (4 MFMAs, B operand = the wave's own rows, still in registers)
//                 + d_below .* x[mt-1] + d_above .* x[mt+1]  (neighbour rows from the exchange buffer; the
//                 coefficients of a missing neighbour are zero, its index is clamped)
template <int NT, bool ZEROC>
__device__ __forceinline__ d4 cmm_od(const d4& C, const double* Mrow, const double* x, const d4& xown, int mt)
{
    const int lane = threadIdx.x & 63;
    const double* cf = Mrow - lane + 4 * 64 + (lane >> 4) * 4;     // [dir][g][r]
    const int mb = mt > 0 ? mt - 1 : 0, ma = mt + 1 < NT ? mt + 1 : NT - 1;
    const double* xb = x + (size_t)(4 * mb) * 64;
    const double* xa = x + (size_t)(4 * ma) * 64;
    const double a0 = Mrow[0], a1 = Mrow[64], a2 = Mrow[128], a3 = Mrow[192];
    const d4 cb = *(const d4*)(cf), ca = *(const d4*)(cf + 16);
    const d4 vb = {xb[0], xb[64], xb[128], xb[192]}, va = {xa[0], xa[64], xa[128], xa[192]};
    d4 acc = ZEROC ? (d4){0.0, 0.0, 0.0, 0.0} : C;
    acc += cb * vb;
    acc += ca * va;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, xown[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, xown[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, xown[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, xown[3], acc, 0, 0, 0);
    return acc;
}

// per-wave context of a cooperative workgroup (BIG: operators from HBM, see OpCursor)
template <int NT, int BW, bool BIG = coop_hbm(NT, BW)>
struct Coop {
    typename std::conditional<BIG, OpCursor, Ring>::type ring;
    double* xbuf;       // LDS exchange buffers [2][4*NT][64], lane offset applied
    const double* M;    // current operator: my row's tiles (lane offset applied)
    int xcur;           // buffer that holds the published x
    int mt;             // my tile row
    int kb0;
    int row_off;        // doubles from the start of an operator image to my row's tiles
    d4 xown;            // my rows of the staged / published x (JQ_BW_OD: B operand of the diagonal block)
    double* nrm;        // LDS [NT][16]: the waves' column totals of the Jacobi solver's residual norm

    // write my rows of Z into the other exchange buffer (visible after the next barrier)
    __device__ __forceinline__ void stage(const d4& Z)
    {
        double* w = xbuf + (size_t)(xcur ^ 1) * (4 * NT * 64) + (size_t)(4 * mt) * 64;
        xown = Z;
        w[0] = Z[0];
        w[64] = Z[1];
        w[128] = Z[2];
        w[192] = Z[3];
    }
    // publish the staged x for a product with the operator that is already resident
    __device__ __forceinline__ void publish()
    {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my ds_writes are done
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        xcur ^= 1;
    }
    // publish the staged x AND switch to the next operator use (one barrier for both)
    __device__ __forceinline__ void publish_next_op()
    {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (BIG) {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        M = next_image() + row_off;   // (Ring: vmcnt(0) + barrier + prefetch inside)
        xcur ^= 1;
    }
    __device__ __forceinline__ const double* next_image()
    {
        if constexpr (BIG) return ring.next();
        else return ring.template next<coop_dma_kper(NT, BW)>();
    }
    __device__ __forceinline__ void next_op() { M = next_image() + row_off; }
    __device__ __forceinline__ const double* x() const { return xbuf + (size_t)xcur * (4 * NT * 64); }
    __device__ __forceinline__ d4 mm_z() const
    {
        if constexpr (BW == JQ_BW_OD) return cmm_od<NT, true>((d4){0, 0, 0, 0}, M, x(), xown, mt);
        else return cmm<NT, BW, true, BIG ? 12 : JQ_PF>((d4){0, 0, 0, 0}, M, x(), kb0);
    }
    __device__ __forceinline__ d4 mm_c(const d4& C) const
    {
        if constexpr (BW == JQ_BW_OD) return cmm_od<NT, false>(C, M, x(), xown, mt);
        else return cmm<NT, BW, false, BIG ? 12 : JQ_PF>(C, M, x(), kb0);
    }
};

// dst[lane] (+)= sum over the NT waves of val, summed in wave order by wave 0 (deterministic).
// scratch: LDS [NT][64] doubles (lane offset NOT applied).  Contains workgroup barriers.
template <int NT>
__device__ __forceinline__ void wg_sum_store(double val, double* scratch, double* dst, int wave, int lane, bool accumulate)
{
    __syncthreads();
    scratch[wave * 64 + lane] = val;
    __syncthreads();
    if (wave == 0) {
        double s = accumulate ? dst[lane] : 0.0;
        for (int w = 0; w < NT; ++w) s += scratch[w * 64 + lane];
        dst[lane] = s;
    }
}

// rows of this wave/lane: 16*mt + 4*r + g
__device__ __forceinline__ d4 rows4(const double* tab, int mt, int g)
{
    return *(const d4*)(tab + 16 * mt + 4 * g);   // tables are stored [block][g][r]
}
__device__ __forceinline__ double dot4(const d4& a, const d4& b)
{
    const d4 p = a * b;
    return (p[0] + p[1]) + (p[2] + p[3]);
}

// out = bpa + sum_{j=1..m} S^j A with the resident operator S (Horner form, see jq_kernels.h); on entry
// NOTHING needs to be published; on exit the exchange buffer holds an intermediate iterate.
// tol2 > 0: JACOBI_SOLVER instead (jacobi!, src/linear_solvers.jl:110-153; jq_kernels.h jacobi_add): X_j = A + S X_{j-1}, X_0 = A,
// stop at the first j with ||X_j - X_{j-1}||_F^2 < tol2 or at j = m -- PER SAMPLE like the reference (round 3): every wave
// leaves the 16 column totals of its tile row in LDS [tile row][column] (one more barrier per iteration; the publication
// barrier of the next iteration protects their reuse), every lane adds the waves' totals of its sample's columns in a fixed
// order; a converged sample keeps its iterate while the workgroup iterates on for the others (the exit is workgroup-uniform:
// all waves see the same totals).
template <int NT, int BW>
__device__ __forceinline__ d4 coop_horner(Coop<NT, BW>& c, const d4& bpa, const d4& A, int m, double tol2 = 0.0, int ncol = 16)
{
    if (m <= 0) return bpa;
    if (tol2 > 0.0) {
        d4 X = A;
        const int lane = threadIdx.x & 63, col = lane & 15;
        const int n = ncol < 16 ? ncol : 16, c0 = col - col % n;
        bool done = false;
        for (int j = 1; j <= m; ++j) {
            c.stage(X);
            c.publish();
            const d4 Xn = c.mm_c(A);
            const d4 d = Xn - X;
            double e = dot4(d, d);
            e += __shfl_xor(e, 16);
            e += __shfl_xor(e, 32);      // column totals of this tile row
            if (lane < 16) c.nrm[16 * c.mt + lane] = e;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            double err2 = 0.0;
            for (int k = 0; k < n; ++k) {      // my sample's columns, the waves' totals in wave order
                double t = 0.0;
                if (c0 + k < 16)               // (ragged tail of the slab: a short group of unused, zero columns)
                    for (int w = 0; w < NT; ++w) t += c.nrm[16 * w + c0 + k];
                err2 += t;
            }
            X = done ? X : Xn;                 // a converged sample keeps its iterate
            done = done || err2 < tol2;
            // every wave holds all 16 columns in its lanes and reads the same totals: the wave vote is the workgroup's verdict
            if (__all(done)) break;
        }
        return (bpa - A) + X;
    }
    d4 Y = A;
    for (int j = 1; j < m; ++j) {
        c.stage(Y);
        c.publish();
        Y = c.mm_c(A);
    }
    c.stage(Y);
    c.publish();
    return c.mm_c(bpa);
}

// ---------------------------------------------------------------------------------------------
// Low-rank full leakage weights (PropArgs::wlr; jq_kernels.h WLow) in the cooperative layout: a column's rows are spread over the
// g-groups of a wave AND over the NT waves, so a column dot product is a lane partial, two cross-row adds and a sum over the
// waves through LDS: xch[parity][value][wave][column], one workgroup barrier per call (double buffered: the slot written by
// call j is read behind barrier j and written again by call j + 2, behind barrier j + 1).
#define JQ_COOP_WDOTS 6
template <int NT>
struct CoopW {
    const double* tab;      // a_0 rows of this wave / lane (rows 16 mt + 4 r + g: element r at tab[4 r])
    const double* lamp;
    double* xch;
    int r, stride, par, wave, lane;
    __device__ __forceinline__ void init(const PropArgs& a, double* lds, int wave_, int lane_)
    {
        r = a.wrank;
        stride = a.wstride;
        lamp = a.wlr;
        tab = a.wlr + a.wlam + 16 * wave_ + (lane_ >> 4);
        xch = lds;
        par = 0;
        wave = wave_;
        lane = lane_;
    }
    __device__ __forceinline__ double lam(int k) const { return lamp[k]; }
    __device__ __forceinline__ d4 rows(int k, int ab) const
    {
        const double* t = tab + (size_t)(2 * k + ab) * stride;
        return (d4){t[0], t[4], t[8], t[12]};
    }
    // v[d] <- sum over the slab column of this lane (all waves) of the per-lane partials v[d]; valid in every lane
    template <int D>
    __device__ __forceinline__ void colsum(double (&v)[D])
    {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            v[d] += __shfl_xor(v[d], 16);
            v[d] += __shfl_xor(v[d], 32);
        }
        double* x = xch + (size_t)par * (JQ_COOP_WDOTS * NT * 16);
        if (lane < 16)
#pragma unroll
            for (int d = 0; d < D; ++d) x[(d * NT + wave) * 16 + lane] = v[d];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int d = 0; d < D; ++d) {
            double s = 0.0;
            for (int w = 0; w < NT; ++w) s += x[(d * NT + w) * 16 + (lane & 15)];
            v[d] = s;
        }
        par ^= 1;
    }
};

// State (re-)integration: uses 0..5 of a step in the cooperative schedule Kp05 S05 Kn0 Kn1 S0 S1 (Kp05).
//   in: u, v (my rows)   out: un = u(t+h), v05, vN = v05 + S05 v05 (caller adds Kp05 un with use 6)
template <int NT, int BW>
__device__ __forceinline__ void coop_state(Coop<NT, BW>& c, const PropArgs& a, double ceps, const d4& wsr, const d4& u,
                                           const d4& v, d4& un, d4& v05, d4& vN)
{
    // use 0: Kp05 -- A = c K05 u
    c.stage(u);
    c.publish_next_op();
    d4 A = c.mm_z();
    if (a.use_shift) A += (ceps * wsr) * u;
    // use 1: S05 -- A = c (K05 u + S05 v) ; v05 = v + sum_j S^j A ; vN = v05 + S05 v05
    c.stage(v);
    c.publish_next_op();
    A = c.mm_c(A);
    v05 = coop_horner<NT, BW>(c, v + A, A, a.m, a.jacobi_tol2, a.N);
    c.stage(v05);
    c.publish();
    vN = c.mm_c(v05);
    // use 2: Kn0 -- un = u - c K0 v05      (x = v05 stays published)
    c.next_op();
    un = c.mm_c(u);
    if (a.use_shift) un -= (ceps * wsr) * v05;
    // use 3: Kn1 -- A = -c K1 v05
    c.next_op();
    A = c.mm_z();
    if (a.use_shift) A -= (ceps * wsr) * v05;
    // use 4: S0 -- un = u + c (S0 u - K0 v05)
    c.stage(u);
    c.publish_next_op();
    un = c.mm_c(un);
    // use 5: S1 -- A = c (S1 un - K1 v05) ; un += sum_j S^j A
    c.stage(un);
    c.publish_next_op();
    A = c.mm_c(A);
    un = coop_horner<NT, BW>(c, un + A, A, a.m, a.jacobi_tol2, a.N);
}

template <int NT, int BW>
__device__ __forceinline__ void coop_setup(Coop<NT, BW>& c, char* smem, const PropArgs& a, int wave, int lane)
{
    c.ring.init(smem, a, wave, lane, NT);
    c.xbuf = (double*)(smem + a.lds_tab_off) + 32 * NT + lane;
    c.nrm = (double*)(smem + a.lds_tab_off) + 32 * NT + 2 * 4 * NT * 64;      // (behind the exchange buffers)
    c.xcur = 0;
    c.mt = wave;
    c.kb0 = coop_kb0(NT, BW, 0);
    // kb0 and the row offset depend on the (wave-uniform) tile row
    int k = (BW == JQ_BW_OD) ? wave : wave - BW;
    if (k < 0) k = 0;
    if (k > NT - coop_nb(NT, BW)) k = NT - coop_nb(NT, BW);
    c.kb0 = k;
    c.row_off = wave * coop_row_elems(NT, BW);
    c.xown = (d4){0.0, 0.0, 0.0, 0.0};
    c.M = nullptr;
}

// ---------------------------------------------------------------------------------------------
// Forward sweep, one slab per workgroup of NT waves.
template <int NT, int BW>
__global__ __launch_bounds__(64 * NT) void k_forward_coop(PropArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KT = 4 * NT;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4;
    const int slab = blockIdx.x;

    double* tab = (double*)(smem + a.lds_tab_off);
    for (int i = threadIdx.x; i < 32 * NT; i += blockDim.x) tab[(i & ~15) + 4 * (i & 3) + ((i >> 2) & 3)] = a.tabs[i];   // [block][g][r]
    Coop<NT, BW> c;
    coop_setup(c, smem, a, wave, lane);
    __syncthreads();
    const d4 wdr = rows4(tab, wave, g), wsr = rows4(tab + 16 * NT, wave, g);

    double* st = a.state + (size_t)slab * a.state_stride;
    d4 u, v;
    for (int r = 0; r < 4; ++r) {
        u[r] = st[(4 * wave + r) * 64 + lane];
        v[r] = st[(KT + 4 * wave + r) * 64 + lane];
    }
    // leak partials: one value per (wave, lane); the state file row holds 64 per slab, so the waves'
    // partials are combined (in wave order) at the end of the chunk
    double leak = 0.0;
    const double ceps = 0.5 * a.h * a.colinfo[(size_t)slab * 32 + (lane & 15)];
    CoopW<NT> wl;
    wl.init(a, c.nrm + 16 * NT, wave, lane);

    for (int n = 0; n < a.nsteps_chunk; ++n) {
        d4 un, v05, vN;
        leak += dot4(wdr, u * u);  // trapezoidal part at t_n (src/evalobjgrad.jl:700)
        coop_state<NT, BW>(c, a, ceps, wsr, u, v, un, v05, vN);
        // use 6: Kp05 -- v(t+h) = v05 + c (K05 u_new + S05 v05)
        c.stage(un);
        c.publish_next_op();
        v = c.mm_c(vN);
        if (a.use_shift) v += (ceps * wsr) * un;
        if (wl.r > 0) {   // full weights: tr(vr' Wr vr) at t_n and t_n+1, 2 tr(vi05' Wr vi05), -2 tr(vi05' Wi vr(t_n)) (:700, :716-718)
            double lk = 0.0;
            for (int k = 0; k < wl.r; ++k) {
                const d4 ak = wl.rows(k, 0), bk = wl.rows(k, 1);
                double d[6] = {dot4(ak, u), dot4(bk, u), dot4(ak, un), dot4(bk, un), dot4(ak, v05), dot4(bk, v05)};
                wl.template colsum<6>(d);
                lk += wl.lam(k) * ((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]) + 2.0 * (d[4] * d[4] + d[5] * d[5]) -
                                   2.0 * (d[5] * d[0] - d[4] * d[1]));
            }
            if (wave == 0 && lane < 16) leak += lk;
        }
        u = un;
        leak += dot4(wdr, u * u) + 2.0 * dot4(wdr, v05 * v05);  // (:716, penalf2a :2170-2180)
        if (a.hist_r) {
            const int col = a.parts > 1 ? 16 * slab + (lane & 15) : (lane & 15);      // column of sample 0
            if (slab < a.parts && col < a.N) {
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
    c.ring.drain();
    for (int r = 0; r < 4; ++r) {
        st[(4 * wave + r) * 64 + lane] = u[r];
        st[(KT + 4 * wave + r) * 64 + lane] = v[r];
    }
    wg_sum_store<NT>(leak, tab + 32 * NT, &st[(JQ_STATE_ARRAYS * KT + JQ_MAXNC) * 64], wave, lane, true);
}

// ---------------------------------------------------------------------------------------------
// Backward sweep, one slab per workgroup of NT waves.  Cooperative schedule (period 13 + 3*Ncoupled):
//   Kp05 S05 Kn0 Kn1 S0 S1 Kp05 | S0 | Hanti_q.. | Kn0 Kn1 S05 Kp05 S1 | (Hanti_q Hsym_q)..
// Trace scalars are written per wave: traces[(slab*NT + wave)][step][Ncoupled*JQ_NTR] (k_trace_reduce sums).
template <int NT, int BW>
__global__ __launch_bounds__(64 * NT) void k_backward_coop(PropArgs a)
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
    Coop<NT, BW> c;
    coop_setup(c, smem, a, wave, lane);
    __syncthreads();
    const d4 wdr = rows4(tab, wave, g), wsr = rows4(tab + 16 * NT, wave, g);

    double* st = a.state + (size_t)slab * a.state_stride;
    d4 u, v, mu, nb;
    for (int r = 0; r < 4; ++r) {
        u[r] = st[(4 * wave + r) * 64 + lane];
        v[r] = st[(KT + 4 * wave + r) * 64 + lane];
        mu[r] = st[(2 * KT + 4 * wave + r) * 64 + lane];
        nb[r] = st[(3 * KT + 4 * wave + r) * 64 + lane];
    }
    const double ceps = 0.5 * a.h * a.colinfo[(size_t)slab * 32 + (lane & 15)];
    const double wgt = a.colinfo[(size_t)slab * 32 + 16 + (lane & 15)];
    const double cfw = a.forced ? 0.5 * a.h * a.tinv : 0.0;
    // one file row of 64 per control holds the slab total of the trace carry; each wave keeps its own partial
    // in a register (total/NT at load; the waves' partials are summed in wave order at the end of the chunk)
    double carry[JQ_MAXNC];
    for (int q = 0; q < JQ_MAXNC; ++q) carry[q] = (q < Nc) ? st[(JQ_STATE_ARRAYS * KT + q) * 64 + lane] / NT : 0.0;
    double* trw = a.traces + ((size_t)(slab * NT + wave) * a.nsteps_chunk) * (Nc * JQ_NTR);
    // full leakage weights in low-rank form: forcing hr0, hi0, hr1, hi1 of src/evalobjgrad.jl:862, :882-888 (see k_backward)
    CoopW<NT> wl;
    wl.init(a, c.nrm + 16 * NT, wave, lane);
    const bool wforce = a.wrank > 0 && a.forced;

    if (a.first_chunk) {
        // carry_q = tr(vr' Hsym_q lambdai) at t = T (see k_backward)
        c.stage(nb);
        for (int q = 0; q < JQ_MAXNC; ++q) {
            if (q < Nc) {
                if (q == 0)
                    c.publish_next_op();
                else
                    c.next_op();
                carry[q] = -dot4(u, c.mm_z());   // nb = -lambda_i
            }
        }
    }

    for (int n = 0; n < a.nsteps_chunk; ++n) {
        d4 un, v05, vN;
        coop_state<NT, BW>(c, a, ceps, wsr, u, v, un, v05, vN);
        // use 6: Kp05 -- finish the state step; R = c K05 nb
        c.stage(un);
        c.publish_next_op();
        vN = c.mm_c(vN);
        if (a.use_shift) vN += (ceps * wsr) * un;
        c.stage(nb);
        c.publish();
        d4 R = c.mm_z();
        if (a.use_shift) R += (ceps * wsr) * nb;
        // use 7: S0 -- R = c (S0 mu - K05 li + hr0) ; X = mu + sum_j S^j R
        c.stage(mu);
        c.publish_next_op();
        R = c.mm_c(R);
        R += (cfw * wdr) * u;
        if (wforce)
            for (int k = 0; k < wl.r; ++k) {
                const d4 ak = wl.rows(k, 0), bk = wl.rows(k, 1);
                double d[2] = {dot4(ak, u), dot4(bk, u)};
                wl.template colsum<2>(d);
                const double cl = cfw * wl.lam(k);
                R += (cl * d[0]) * ak + (cl * d[1]) * bk;      // + c hr0
            }
        const d4 X = coop_horner<NT, BW>(c, mu + R, R, a.m, a.jacobi_tol2, a.N);
        // early traces with X: tr1 = tr(vr0' Hanti_q X), tr3 = tr(vr' Hanti_q X)
        c.stage(X);
        for (int q = 0; q < JQ_MAXNC; ++q) {
            if (q < Nc) {
                if (q == 0)
                    c.publish_next_op();
                else
                    c.next_op();
                const d4 Tq = c.mm_z();
                const double ts = wave_sum2(dot4(u, Tq) * wgt, dot4(un, Tq) * wgt);      // rows 0, 1: t1;  rows 2, 3: t3
                if ((lane & 31) == 0) trw[(size_t)n * (Nc * JQ_NTR) + q * JQ_NTR + (lane >> 4)] = ts;
            }
        }
        // use 8: Kn0 -- L = -c K0 X ; use 9: Kn1 -- Q = -c K1 X       (x = X stays published)
        c.next_op();
        d4 L = c.mm_z();
        if (a.use_shift) L -= (ceps * wsr) * X;
        c.next_op();
        d4 Qv = c.mm_z();
        if (a.use_shift) Qv -= (ceps * wsr) * X;
        // use 10: S05 -- L = -c l2 ; Q = -c (...) ; nb_new = nb + L + sum_j S^j Q
        c.stage(nb);
        c.publish_next_op();
        {
            d4 P = c.mm_z();
            P -= (cfw * wdr) * v05;
            if (wforce)
                for (int k = 0; k < wl.r; ++k) {
                    const d4 ak = wl.rows(k, 0), bk = wl.rows(k, 1);
                    double d[4] = {dot4(ak, un), dot4(bk, un), dot4(ak, v05), dot4(bk, v05)};
                    wl.template colsum<4>(d);
                    const double cl = cfw * wl.lam(k);
                    P -= (cl * d[2]) * ak + (cl * d[3]) * bk;      // - c hi0 (goes into L and Q)
                    Qv += (cl * d[0]) * bk - (cl * d[1]) * ak;     // Q: - c (hi1 - hi0) = + c Wi vr(t_n) / T
                }
            L += P;
            Qv += P;
        }
        c.stage(L);
        c.publish();
        Qv = c.mm_c(Qv);
        const d4 nbn = coop_horner<NT, BW>(c, (nb + L) + Qv, Qv, a.m, a.jacobi_tol2, a.N);
        const d4 Bq = nb + nbn;   // -(li0 + li)
        // use 11: Kp05 -- G = X + c K05 nb_new
        c.stage(nbn);
        c.publish_next_op();
        d4 G = c.mm_c(X);
        if (a.use_shift) G += (ceps * wsr) * nbn;
        // use 12: S1 -- lambda_r_new = X + c (S1 X - K05 li_new + hr1)
        c.stage(X);
        c.publish_next_op();
        G = c.mm_c(G);
        G += (cfw * wdr) * un;
        if (wforce)
            for (int k = 0; k < wl.r; ++k) {
                const d4 ak = wl.rows(k, 0), bk = wl.rows(k, 1);
                double d[4] = {dot4(ak, un), dot4(bk, un), dot4(ak, v05), dot4(bk, v05)};
                wl.template colsum<4>(d);
                const double cl = cfw * wl.lam(k);
                G += (cl * (d[0] - d[3])) * ak + (cl * (d[1] + d[2])) * bk;      // + c hr1
            }
        // late traces: tr5 = tr(vi05' Hanti (li0+li)), tr2 = tr(vi05' Hsym X), tr4 = tr(vr' Hsym li) + carry
        for (int q = 0; q < JQ_MAXNC; ++q) {
            if (q < Nc) {
                c.stage(Bq);
                c.publish_next_op();   // Hanti_q
                const double t5 = -dot4(v05, c.mm_z()) * wgt;
                c.stage(X);
                c.publish_next_op();   // Hsym_q
                const double t2 = dot4(v05, c.mm_z()) * wgt;
                c.stage(nbn);
                c.publish();
                const double p4 = -dot4(un, c.mm_z());
                // one reduction for the three: row 0: t5, row 1: t2, row 3: t4
                const double ts = wave_sum4_rows(t5, t2, 0.0, (p4 + carry[q]) * wgt);
                carry[q] = p4;
                if ((lane & 15) == 0 && lane != 32)
                    trw[(size_t)n * (Nc * JQ_NTR) + q * JQ_NTR + (lane == 0 ? 4 : lane == 16 ? 1 : 3)] = ts;
            }
        }
        u = un;
        v = vN;
        mu = G;
        nb = nbn;
    }
    c.ring.drain();
    for (int r = 0; r < 4; ++r) {
        st[(4 * wave + r) * 64 + lane] = u[r];
        st[(KT + 4 * wave + r) * 64 + lane] = v[r];
        st[(2 * KT + 4 * wave + r) * 64 + lane] = mu[r];
        st[(3 * KT + 4 * wave + r) * 64 + lane] = nb[r];
    }
    for (int q = 0; q < JQ_MAXNC; ++q)
        if (q < Nc) wg_sum_store<NT>(carry[q], tab + 32 * NT, &st[(JQ_STATE_ARRAYS * KT + q) * 64], wave, lane, false);
}
