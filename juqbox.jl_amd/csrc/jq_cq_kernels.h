// jq_cq_kernels.h -- "cooperative quad" propagators: the LATENCY path of the JQ_BW_T4 structure (cnot3: 1 .. 256 samples).
//
// The quad-layout kernels of jq_kernels.h give one wave four state columns and ALL 16-row blocks of every array: a single
// evaluation is one wave's chain of 72 dependent products per time step, each of them NT blocks long (cnot3: 6 blocks,
// ~240 ns), and the rest of the GPU idles.  Here the NT blocks of a column quad are split over the NT waves of a workgroup:
// wave `mt` owns block mt of every array (ONE register per array), a product costs one block's work per wave
//     D[mt] = C[mt] + B_mt x[mt] (one v_mfma_f64_4x4x4_4b) + c0 shr4(x[mt]) + c1 shl4(x[mt]) + c2 x[mt-1] + c3 x[mt+1]
// plus the exchange of x with the two neighbouring waves through a double-buffered LDS image (one ds_write, one workgroup
// barrier, two ds_reads).  What such a kernel pays for (measured in round 2, HISTORY.md):
//   * a publication interval costs ~230 cycles even when it holds a single product (LDS write -> barrier -> LDS read -> two
//     dependent FMAs), so the step is regrouped around PUBLICATIONS: every published x serves all the products that need it
//     (K05 u and S0 u; S05 v05, K0 v05 and K1 v05; the products with X, ...): 5 + 2 m publications per forward step (20
//     products at m = 6 Neumann terms), and the part of a product that needs only the wave's own block runs under the barrier;
//   * a wave that is (almost) alone on its SIMD issues ONE instruction every 10 .. 12 cycles, whatever its kind
//     (probes/dp_rate_probe.hip): scalar bookkeeping, branches, staging and reductions are as expensive as arithmetic.  The
//     backward sweep therefore runs its two chains -- state re-integration and adjoint step -- on TWO SETS of NT waves (the
//     adjoint products depend on the state step of the SAME time step only through u, v05, un of the wave's own block, read from
//     the state wave's publications), both sets pass the same 5 + 2 m barriers per step (52 products), the trace products are
//     shared between the sets, their reductions handed to one wave per group of four values, and the staging is done by the
//     waves that would otherwise wait (two extra waves in the forward sweep).
// Same operators, images, state file and trace records as the quad-layout kernels; the regrouping only reorders floating-point
// additions.  One cnot3 evaluation: 0.54 s on the quad-layout kernels, 0.20 s here.
#pragma once
#include "jq_kernels.h"

// Window staging (jq_kernels.h, Ring with batch < 0) re-timed for these kernels: a ring of JQ_WIN_TPS = 5 time points (K and S
// image each) and the constant trace images are resident in LDS, the time points stream in by global -> LDS DMA (the 1 KiB
// pieces of an image pair spread over the staging waves).  There is no staging barrier: behind the LAST publication barrier of
// step n every wave has loaded its operator blocks of step n into registers, so the time points 2n, 2n+1 are dead and the DMA
// of 2n+5, 2n+6 (operators of step n+2) is issued into their slots; the issuing waves drain it (vmcnt(0)) in front of the last
// barrier of step n+1, behind which the operators of step n+2 start to be loaded.  The cursor is incremental (no
// multiplications, no modulo, no mode branches).
struct WinRing {
    char* smem;
    const char* gnext;      // global address of the next time point to fetch
    unsigned stride_b;      // bytes per image
    unsigned slot_bytes;    // bytes per time point (K and S image)
    unsigned cbase;         // byte offset of the constant images (= end of the ring)
    int pieces2;            // 1 KiB pieces of a time point
    int jnext, jlast;       // next time point to fetch, last one of the chunk
    unsigned snext;         // byte offset of its ring slot
    int wave, nwaves, lane;
    unsigned wb0, wb1, wb2; // byte offsets of the time points 2n, 2n+1, 2n+2 of the step whose operators are loaded next

    __device__ __forceinline__ void dma(const char* gsrc, char* dst, int pieces) const
    {
        // (inline asm instead of __builtin_amdgcn_global_load_lds: hipcc puts an s_waitcnt vmcnt(0) in front of the next LDS read
        //  that may alias the destination -- here the operator loads at the top of the step, i.e. it waited for the DMA right
        //  after issuing it (~850 cycles per step on the critical path).  The explicit drains in sync<true>() are what orders
        //  the DMA with its readers.)
        unsigned lo;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lo));
        const unsigned lo16 = lo * 16u;      // (scalar base + 32-bit lane offset: no 64-bit per-lane address)
        const unsigned ldst = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)dst;
        for (int p = wave; p < pieces; p += nwaves)
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(ldst + (unsigned)p * 1024u), "v"(lo16), "s"(gsrc + (size_t)p * 1024) : "memory");
    }
    __device__ __forceinline__ void issue_next()
    {
        if (jnext > jlast) return;
        dma(gnext, smem + snext, pieces2);
        gnext += slot_bytes;
        ++jnext;
        snext += slot_bytes;
        if (snext == cbase) snext = 0;
    }
    // time points 0 .. 4 and the constants; wb1, wb2 = time points 1, 2 (step 0); time point 0 is at offset 0
    // (all JQ_WIN_TPS time points are fetched before the first step)
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
    // the window of the next step (call once its operators are about to be loaded)
    __device__ __forceinline__ void advance()
    {
        wb0 = wb2;
        wb1 = wb2 + slot_bytes;
        if (wb1 == cbase) wb1 = 0;
        wb2 = wb1 + slot_bytes;
        if (wb2 == cbase) wb2 = 0;
    }
    // LDS image (lane offset applied) of K (KIND 0) / S (KIND 1) at time point 2n + TP, of constant image #idx
    template <int KIND, int TP>
    __device__ __forceinline__ const double* ks() const
    {
        return (const double*)(smem + ((TP == 0 ? wb0 : TP == 1 ? wb1 : wb2) + KIND * stride_b)) + lane;
    }
    __device__ __forceinline__ const double* cimg(int idx) const { return (const double*)(smem + (cbase + (unsigned)idx * stride_b)) + lane; }
};

#ifdef JQ_CQ_TIMING     // experiment: cycle counter at the marks of one time step, printed by every wave at the end of the kernel
#define JQ_TS_DECL unsigned long long jq_ts[28]; int jq_nts = 0; for (int i_ = 0; i_ < 28; ++i_) jq_ts[i_] = 0;
#define JQ_TS(n) if ((n) == 500 && jq_nts < 28) jq_ts[jq_nts++] = __builtin_readcyclecounter();
#define JQ_TS_PRINT(w) if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { int d_[27]; for (int i_ = 0; i_ < 27; ++i_) d_[i_] = i_ + 1 < jq_nts ? (int)(jq_ts[i_ + 1] - jq_ts[i_]) : -1; printf("wave %2d: %5d %5d %5d %5d %5d %5d %5d %5d %5d %5d %5d %5d %5d %5d %5d %5d %5d %5d %5d %5d %5d %5d %5d %5d total %d\n", (w), d_[0], d_[1], d_[2], d_[3], d_[4], d_[5], d_[6], d_[7], d_[8], d_[9], d_[10], d_[11], d_[12], d_[13], d_[14], d_[15], d_[16], d_[17], d_[18], d_[19], d_[20], d_[21], d_[22], d_[23], (int)(jq_ts[jq_nts - 1] - jq_ts[0])); }
#else
#define JQ_TS_DECL
#define JQ_TS(n)
#define JQ_TS_PRINT(w)
#endif
typedef __attribute__((address_space(3))) double jq_lds_double;
// Two column quads per workgroup (k_forward_cq<.., 2>): every array is a PAIR of doubles per lane, quad A in channel C of the exchange
// image and quad B in channel C + 1; the propagation code below is written once for T = double and T = D2.
struct D2 {
    double a, b;
    __device__ __forceinline__ D2() {}
    __device__ __forceinline__ D2(double x) : a(x), b(x) {}
    __device__ __forceinline__ D2(double x, double y) : a(x), b(y) {}
};
__device__ __forceinline__ D2 operator+(D2 x, D2 y) { return D2(x.a + y.a, x.b + y.b); }
__device__ __forceinline__ D2 operator*(D2 x, D2 y) { return D2(x.a * y.a, x.b * y.b); }
__device__ __forceinline__ D2 operator*(double c, D2 y) { return D2(c * y.a, c * y.b); }
__device__ __forceinline__ D2 operator-(D2 x) { return D2(-x.a, -x.b); }
__device__ __forceinline__ D2 fma(D2 c, D2 x, D2 y) { return D2(fma(c.a, x.a, y.a), fma(c.b, x.b, y.b)); }
__device__ __forceinline__ D2 fma(double c, D2 x, D2 y) { return D2(fma(c, x.a, y.a), fma(c, x.b, y.b)); }
// Exchange image in LDS: [2 parities][3 channels][NT + 2 blocks][64] doubles -- a zero block in front of and behind the NT
// blocks of a channel, so that the neighbours of the edge blocks need no clamping.  Channel 0: forward sweep / state chain of
// the backward sweep; channel 1: adjoint chain; channel 2: its second vector of the last publication of a step.
// Every access is ONE base register -- block mt-1 of channel 0 in parity 0, this lane -- plus a COMPILE-TIME offset: the parity
// P and the channel C of every publication are template arguments.  (A run-time parity cost three instructions per publication
// -- address add, sign flip, neighbour address -- of the ~20 a Neumann publication consists of.)  Publication k of a chunk goes
// to parity k & 1; a time step has an odd number of publications, 5 + 2 m, so the time loop is written for two steps (start
// parity 0, then 1) and the kernels are instantiated for even and odd m (MODD).
//
// A publication has two phases: post(x) writes the wave's block, sync() waits for everybody's.  The products are split into the
// part that needs only the wave's own block (MFMA, lane shifts, (i, i+-4) terms: own()) and the two (i, i+-16) terms with the
// neighbours' blocks (nbr()); own() is written between post() and sync(), and hipcc is left free to place it: it keeps the MFMA
// under the latency of the LDS write and moves the shifts and FMAs BEHIND the barrier, where they run while the neighbours'
// blocks are being read.  (Pinning all of own() in front of the barrier was measured 13 % slower per Neumann publication: an
// instruction in front of the barrier delays everybody's arrival, one behind it hides in the read latency.)
// DN (round 6): the DENSE policy for NT = 2 -- problems of 17 .. 32 levels WITHOUT the 4 x 4 x n structure, whose single evaluations ran
// on the cooperative kernels at 26 us per time step (profiles/r06_midsize_single.txt).  Same layout, publications, chains and trace
// hand-off; only the product differs.  With lane 16 k + 4 b + j <-> (row 4 b + k, column j) a state register IS the B operand of its own
// four diagonal 4 x 4 blocks; rotated by 4 s lanes inside each 16-lane row (row_ror: lane p reads lane p - 4 s) it gives output block b
// the rows of block (b - s) mod 4.  A dense 16 x 16 tile is therefore FOUR v_mfma_f64_4x4x4_4b on the register and its three
// rotations, the A operand of rotation s holding M[16 t + 4 b + i][16 t' + 4 ((b - s) mod 4) + k] on lane 16 k + 4 b + i (host:
// dq_image): own() = the tile (mt, mt), nbr() = the tile (mt, other block) -- at NT = 2 every wave has exactly one neighbour, the
// other one is the zero pad block of the exchange image.  16 clk per MFMA against the 64 of a v_mfma_f64_16x16x4 that N = 4 fills to a quarter.
template <int N>
__device__ __forceinline__ double row_ror64(double x)
{
    union {
        double d;
        int i[2];
    } a, b;
    a.d = x;
    b.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], 0x120 + N, 0xf, 0xf, false);
    b.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], 0x120 + N, 0xf, 0xf, false);
    return b.d;
}
#define JQ_DQ_TILES 8      // 64-lane operand registers per 16-row block of a dense NT = 2 image: [own tile: s = 0 .. 3 | other tile: s = 0 .. 3]
template <int NT, bool DN = false>
struct CoopQ {
    static_assert(!DN || NT == 2, "dense policy: two 16-row blocks");
    static constexpr int CHS = (NT + 2) * 64;       // doubles per channel
    static constexpr int PAR = 3 * CHS;              // doubles per parity
    WinRing ring;
    jq_lds_double* xb;  // block mt-1 (pad block for mt = 0) of channel 0 in parity 0, this lane
    int mt;             // my block
    int lane;

    __device__ __forceinline__ void setup(double* xbuf, int blk, int lane_)
    {
        mt = blk, lane = lane_;
        for (int i = threadIdx.x; i < 2 * PAR; i += blockDim.x) xbuf[i] = 0.0;      // (the pads stay zero)
        xb = (jq_lds_double*)(xbuf + blk * 64 + lane_);
    }
    // my block of channel C, parity P
    template <int P, int C>
    __device__ __forceinline__ void post(double x) { xb[P * PAR + C * CHS + 64] = x; }
    template <int P, int C>
    __device__ __forceinline__ void post(D2 x)
    {
        xb[P * PAR + C * CHS + 64] = x.a;
        xb[P * PAR + (C + 1) * CHS + 64] = x.b;
    }
    template <int P, int C>
    __device__ __forceinline__ double block() const { return xb[P * PAR + C * CHS + 64]; }
    template <bool DMA = false>
    __device__ __forceinline__ void sync()
    {
        if (DMA)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    // my block of an operator image (LDS, lane offset applied): A operand of the MFMA + the four coupling coefficients of my row
    struct OpT4 {
        double a;
        d4 c;
    };
    struct OpDn {
        double ao[4], an[4];      // A operands of the rotations 0 .. 3: my own tile, the other block's tile
    };
    typedef typename std::conditional<DN, OpDn, OpT4>::type Op;
    __device__ __forceinline__ Op load(const double* M) const
    {
        Op o;
        if constexpr (DN) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                o.ao[r] = M[(mt * JQ_DQ_TILES + r) * 64];
                o.an[r] = M[(mt * JQ_DQ_TILES + 4 + r) * 64];
            }
        } else {
            o.a = M[mt * 64];
            o.c = t4q_cload(t4q_c<NT>(M, lane), mt);
        }
        return o;
    }
    // my block of a vector with its two lane shifts (shared by all products with that vector); dense: with its three rotations
    struct ShT4 {
        double x, dn, up;
    };
    struct Rot {
        double x, r1, r2, r3;
    };
    typedef typename std::conditional<DN, Rot, ShT4>::type Sh;
    static __device__ __forceinline__ Rot rot(double x)
    {
        Rot r;
        r.x = x, r.r1 = row_ror64<4>(x), r.r2 = row_ror64<8>(x), r.r3 = row_ror64<12>(x);
        return r;
    }
    __device__ __forceinline__ Sh sh(double x) const
    {
        if constexpr (DN) {
            return rot(x);
        } else {
            Sh s;
            s.x = x, s.dn = row_shift4<0x114>(x), s.up = row_shift4<0x104>(x);
            return s;
        }
    }
    // C + (my block's own part of M x)
    __device__ __forceinline__ double own(double C, const Op& o, const Sh& s) const
    {
        if constexpr (DN) {
            double acc = __builtin_amdgcn_mfma_f64_4x4x4f64(o.ao[0], s.x, C, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_4x4x4f64(o.ao[1], s.r1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_4x4x4f64(o.ao[2], s.r2, acc, 0, 0, 0);
            return __builtin_amdgcn_mfma_f64_4x4x4f64(o.ao[3], s.r3, acc, 0, 0, 0);
        } else {
            double acc = __builtin_amdgcn_mfma_f64_4x4x4f64(o.a, s.x, C, 0, 0, 0);
            acc = fma(o.c[0], s.dn, acc);
            return fma(o.c[1], s.up, acc);
        }
    }
    struct Sh2 {
        Sh a, b;
    };
    __device__ __forceinline__ Sh2 sh(D2 x) const
    {
        Sh2 s;
        s.a = sh(x.a), s.b = sh(x.b);
        return s;
    }
    __device__ __forceinline__ D2 own(D2 C, const Op& o, const Sh2& s) const { return D2(own(C.a, o, s.a), own(C.b, o, s.b)); }
    // the neighbours' blocks of the vector published in channel C, parity P (dense: THE other block -- one of the two is the zero pad --
    // with its rotations)
    struct NbT4 {
        double b, a;
    };
    typedef typename std::conditional<DN, Rot, NbT4>::type Nb;
    struct Nb2 {
        Nb a, b;
    };
    template <int P, int C, typename T = double>
    __device__ __forceinline__ auto nbs() const
    {
        if constexpr (std::is_same<T, D2>::value) {
            Nb2 n;
            n.a = nbs<P, C>(), n.b = nbs<P, C + 1>();
            return n;
        } else if constexpr (DN) {
            const double below = xb[P * PAR + C * CHS], above = xb[P * PAR + C * CHS + 128];
            return rot(below + above);
        } else {
            Nb n;
            n.b = xb[P * PAR + C * CHS], n.a = xb[P * PAR + C * CHS + 128];
            return n;
        }
    }
    // ... from the two values themselves (the split kernels read them out of the state file / the hand-off ring; zeros beyond the edge blocks)
    __device__ __forceinline__ Nb nb_make(double below, double above) const
    {
        if constexpr (DN) {
            return rot(below + above);
        } else {
            Nb n;
            n.b = below, n.a = above;
            return n;
        }
    }
    // ... at a run-time offset (doubles from xb to the block below mine) of the exchange image: the implicit-midpoint kernels
    __device__ __forceinline__ Nb nbs_at(int po) const
    {
        if constexpr (DN) {
            return rot(xb[po] + xb[po + 128]);
        } else {
            Nb n;
            n.b = xb[po], n.a = xb[po + 128];
            return n;
        }
    }
    __device__ __forceinline__ double nbr(double acc, const Op& o, const Nb& n) const
    {
        if constexpr (DN) {
            acc = __builtin_amdgcn_mfma_f64_4x4x4f64(o.an[0], n.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_4x4x4f64(o.an[1], n.r1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_4x4x4f64(o.an[2], n.r2, acc, 0, 0, 0);
            return __builtin_amdgcn_mfma_f64_4x4x4f64(o.an[3], n.r3, acc, 0, 0, 0);
        } else {
            acc = fma(o.c[2], n.b, acc);      // (the coefficients of a missing neighbour are zero)
            return fma(o.c[3], n.a, acc);
        }
    }
    __device__ __forceinline__ D2 nbr(D2 acc, const Op& o, const Nb2& n) const { return D2(nbr(acc.a, o, n.a), nbr(acc.b, o, n.b)); }
    // M x for a constant trace image (my block).  ORD: control q acts on subsystem q only -- q = 0: the 4 x 4 diagonal blocks
    // (MFMA), q = 1: the (i, i+-4) couplings (lane shifts), q = 2: the (i, i+-16) couplings (neighbour blocks) -- so one part
    // of the product and one LDS read suffice; otherwise the whole product (the absent parts of an image are stored as zeros).
    template <bool ORD>
    __device__ __forceinline__ double trace_mm(const double* M, int q, const Sh& s, const Nb& n) const
    {
        if constexpr (ORD && !DN) {
            if (q == 0) return __builtin_amdgcn_mfma_f64_4x4x4f64(M[mt * 64], s.x, 0.0, 0, 0, 0);
            const d4 cc = t4q_cload(t4q_c<NT>(M, lane), mt);
            if (q == 1) return fma(cc[1], s.up, cc[0] * s.dn);
            return fma(cc[3], n.a, cc[2] * n.b);
        } else {
            const Op o = load(M);
            return nbr(own(0.0, o, s), o, n);
        }
    }
    // one publication of a Neumann series: Y <- C + S Y
    template <int P, int C, typename T>
    __device__ __forceinline__ T hstep(T Cv, T Y, const Op& S)
    {
        post<P, C>(Y);
        const T t = own(Cv, S, sh(Y));
        sync();
        return nbr(t, S, nbs<P, C, T>());
    }
    // base + sum_{j=1..m} S^j A  (Horner form, jq_kernels.h): m publications in channel C, the first one in parity PS; m is odd iff
    // MODD.  The inner m - 1 publications run in pairs (static parities); the next publication after the series has parity PS ^ MODD.
    // (pre: issues the loads of the operators that the publication AFTER the series needs, one interval ahead)
    template <int PS, int C, bool MODD, typename T, typename F>
    __device__ __forceinline__ T horner(T base, T A, const Op& S, int m, F pre)
    {
        if (m <= 0) {
            pre();
            return base;
        }
        T Y = A;
        int q = m - 1;      // inner publications; odd iff m is even
        constexpr int PA = MODD ? PS : (PS ^ 1);      // parity of the first publication of the pairs
        if constexpr (!MODD) {
            Y = hstep<PS, C>(A, Y, S);
            --q;
        }
        for (; q > 0; q -= 2) {
            Y = hstep<PA, C>(A, Y, S);
            Y = hstep<PA ^ 1, C>(A, Y, S);
        }
        pre();
        return hstep<PA, C>(base, Y, S);      // (the final publication: parity PS ^ ((m - 1) & 1) = PA)
    }
    template <int PS, int C, bool MODD, typename T>
    __device__ __forceinline__ T horner(T base, T A, const Op& S, int m)
    {
        return horner<PS, C, MODD>(base, A, S, m, [] {});
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

// First three stages of wave_sum4 (jq_kernels.h): the 16 column partials of a, c, b, d end up in the rows 0, 1, 2, 3 of ONE
// register (9 instructions for four values).  The cooperative-quad backward sweep leaves the rest of the reduction -- the sum
// over the NT blocks and the four rotate-adds inside the rows -- to one wave per group of four values, one barrier later.
__device__ __forceinline__ double cq_part4(double a, double b, double c, double d)
{
    row_swap32(a, b);
    double p = a + b;
    row_swap32(c, d);
    double q = c + d;
    row_swap16(p, q);
    return p + q;
}

// Full leakage weights on the cooperative-quad kernels (round 5): W = sum_k lam_k f_k f_k^H, f_k = a_k + i b_k, in FOUR SLOTS -- a real
// weight matrix (wmat_imag = 0, real forbidden states) of rank <= 4: slots a_0 .. a_3; a complex one of rank <= 2: a_0, b_0, a_1, b_1
// (PropArgs::wcplx).  A column's dot products slot . x run over the NT blocks of the quad, i.e. over NT waves: the wave that owns block
// w leaves the 16 column partials of its rows for all four slots in ONE register (row k of the register = slot k) in LDS,
// wpart[vector][w][64], in front of a barrier the step has anyway; whoever needs the dots adds the NT registers behind it.
// Both halves are one v_mfma_f64_4x4x4 in the quad layout (A operand: lane 16 k + 4 b + i holds A_b[i][k]; B operand / result: lane
// 16 i + 4 b + j holds X_b[i][j]):
//   part:  A_b[k][i] = slot_k[4 b + i]           x  my block of x    ->  D_b[k][j] = sum_i slot_k[4 b + i] x[4 b + i][j]; two rotate-adds over b
//   apply: A_b[i][k] = coef_k[my row 4 b + i]     x  d[k][j] (the sum of the NT partial registers, the same for every b)  ->  (W x)[my row][j]
// With (p, q) = (a . x, b . x):  W_r x = sum lam (a p + b q)  (coefficients lam a | lam b),  W_i x = sum lam (b p - a q)  (lam b | -lam a).
// A complex W needs W_i vr(t_n) in the MIDDLE of the adjoint step of step n (hi1, src/evalobjgrad.jl:886-888), which the state chain of
// the SAME workgroup delivers at its end: complex weights run on the two- / three-workgroup backward kernels only (the state role is
// steps ahead there), real ones on the one-workgroup kernel too.
#define JQ_CQ_WRANK 4
struct CqW {
    double akA;       // part(): slot_k[16 blk + 4 b + i] on lane 16 i + 4 b + k
    double* wpart;    // LDS [2][NT][64]
    // row `row` of slot k (k < nslots) in the table lam[JQ_MAX_WRANK] | a_0 | b_0 | a_1 | b_1 ...
    static __device__ __forceinline__ int nslots(const PropArgs& a) { return a.wcplx ? 2 * a.wrank : a.wrank; }
    static __device__ __forceinline__ double slot(const PropArgs& a, int k, int row) { return a.wlr[JQ_MAX_WRANK + (size_t)(a.wcplx ? k : 2 * k) * a.wstride + row]; }
    static __device__ __forceinline__ double lam(const PropArgs& a, int k) { return a.wlr[a.wcplx ? k >> 1 : k]; }
    __device__ __forceinline__ void init(const PropArgs& a, char* smem, int blk, int lane_)
    {
        const int k = lane_ & 3, row = 16 * blk + 4 * ((lane_ >> 2) & 3) + (lane_ >> 4);
        akA = k < nslots(a) ? slot(a, k, row) : 0.0;
        wpart = (double*)(smem + a.wlr_lds);
    }
    // apply()'s A operand for W_r: scale lam slot_k[16 blk + 4 b + i] on lane 16 k + 4 b + i
    static __device__ __forceinline__ double coef(const PropArgs& a, int blk, int lane_, double scale)
    {
        const int k = lane_ >> 4, row = 16 * blk + 4 * ((lane_ >> 2) & 3) + (lane_ & 3);
        return k < nslots(a) ? scale * lam(a, k) * slot(a, k, row) : 0.0;
    }
    // ... for W_i (complex only; 0 for a real W): the slots of a term swapped, the second with a minus sign
    static __device__ __forceinline__ double coef_imag(const PropArgs& a, int blk, int lane_, double scale)
    {
        const int k = lane_ >> 4, row = 16 * blk + 4 * ((lane_ >> 2) & 3) + (lane_ & 3);
        if (!a.wcplx || k >= nslots(a)) return 0.0;
        return (k & 1) ? -(scale * lam(a, k) * slot(a, k - 1, row)) : scale * lam(a, k) * slot(a, k + 1, row);
    }
    // my block's share of the four dots with x, for every column of the quad: register row k = slot k, the same in every 4-row group
    __device__ __forceinline__ double part(double x) const
    {
        return row_ror_add<8>(row_ror_add<4>(__builtin_amdgcn_mfma_f64_4x4x4f64(akA, x, 0.0, 0, 0, 0)));
    }
    template <int NT>
    __device__ __forceinline__ void put(int vec, int blk, int lane_, double x) const
    {
#ifndef JQ_EXP_W_NOPUT      // (timing experiment: wrong results)
        wpart[((size_t)vec * NT + blk) * 64 + lane_] = part(x);
#endif
    }
    // sum_k coef_k[my row] (slot_k . x)[my column] -- behind the barrier that follows the put()s
    template <int NT>
    __device__ __forceinline__ double apply(int vec, int lane_, double cA) const
    {
#ifdef JQ_EXP_W_NOAPPLY     // (timing experiment: wrong results)
        return cA;
#endif
        const double* r = wpart + (size_t)vec * NT * 64 + lane_;
        double d = r[0];
#pragma unroll
        for (int w = 1; w < NT; ++w) d += r[w * 64];
        return __builtin_amdgcn_mfma_f64_4x4x4f64(cA, d, 0.0, 0, 0, 0);
    }
};

// the six operator blocks of a time step (this wave's share of K, S at the time points 2n, 2n+1, 2n+2 of the chunk)
template <int NT, bool DN = false>
struct CqOps {
    typename CoopQ<NT, DN>::Op Kp05, S05, Kn0, S0, Kn1, S1;
};
// step 0: time point 0 sits at the start of the ring
template <int NT, bool DN>
__device__ __forceinline__ CqOps<NT, DN> cq_first_ops(const CoopQ<NT, DN>& c)
{
    CqOps<NT, DN> o;
    o.Kn0 = c.load(c.ring.template ks<0, 0>());
    o.S0 = c.load(c.ring.template ks<1, 0>());
    o.Kp05 = c.load(c.ring.template ks<0, 1>());
    o.S05 = c.load(c.ring.template ks<1, 1>());
    o.Kn1 = c.load(c.ring.template ks<0, 2>());
    o.S1 = c.load(c.ring.template ks<1, 2>());
    return o;
}
// the next step: K0, S0 are this step's K1, S1; the time points 2n+3, 2n+4 have landed (see WinRing)
template <int NT, bool DN>
__device__ __forceinline__ void cq_next_ops(CoopQ<NT, DN>& c, CqOps<NT, DN>& o)
{
    c.ring.advance();
    o.Kn0 = o.Kn1;
    o.S0 = o.S1;
    o.Kp05 = c.load(c.ring.template ks<0, 1>());
    o.S05 = c.load(c.ring.template ks<1, 1>());
    o.Kn1 = c.load(c.ring.template ks<0, 2>());
    o.S1 = c.load(c.ring.template ks<1, 2>());
}

// Parities of the publications of a time step that starts with parity P0 (both sweeps, both chains):
//   I1: P0   I2: P0^1   first Neumann series: from P0   I3: P0^M   I4: P0^M^1   second series: from P0^M   I5: P0      (M = m & 1)
// State step up to its second-to-last publication: in u, v; out un = u(t+h), v05, vN = v05 + S05 v05 (the caller publishes un
// once more and adds Kp05 un).
template <int NT, int P0, bool MODD, typename T, bool DN>
__device__ __forceinline__ void cq_state(CoopQ<NT, DN>& c, const PropArgs& a, const CqOps<NT, DN>& o, T cw, T u, T v, T& un, T& v05, T& vN)
{
    constexpr int M = MODD ? 1 : 0;
    // x = u: A = c K05 u ; P = u + c S0 u
    c.template post<P0, 0>(u);
    T A, P;
    {
        const auto s = c.sh(u);
        A = c.own(T(0.0), o.Kp05, s);
        P = c.own(u, o.S0, s);
        if (a.use_shift) A = fma(cw, u, A);
        c.sync();
        const auto n = c.template nbs<P0, 0, T>();
        A = c.nbr(A, o.Kp05, n);
        P = c.nbr(P, o.S0, n);
    }
    // x = v: A = c (K05 u + S05 v) ; v05 = v + sum_j S^j A
    c.template post<P0 ^ 1, 0>(v);
    A = c.own(A, o.S05, c.sh(v));
    c.sync();
    A = c.nbr(A, o.S05, c.template nbs<P0 ^ 1, 0, T>());
    v05 = c.template horner<P0, 0, MODD>(v + A, A, o.S05, a.m);
    // x = v05: vN = v05 + c S05 v05 ; un = u + c (S0 u - K0 v05) ; A = -c K1 v05
    c.template post<P0 ^ M, 0>(v05);
    {
        const auto s = c.sh(v05);
        vN = c.own(v05, o.S05, s);
        un = c.own(P, o.Kn0, s);
        A = c.own(T(0.0), o.Kn1, s);
        if (a.use_shift) {
            un = fma(-cw, v05, un);
            A = fma(-cw, v05, A);
        }
        c.sync();
        const auto n = c.template nbs<P0 ^ M, 0, T>();
        vN = c.nbr(vN, o.S05, n);
        un = c.nbr(un, o.Kn0, n);
        A = c.nbr(A, o.Kn1, n);
    }
    // x = un: A = c (S1 un - K1 v05) ; un += sum_j S^j A
    c.template post<P0 ^ M ^ 1, 0>(un);
    A = c.own(A, o.S1, c.sh(un));
    c.sync();
    A = c.nbr(A, o.S1, c.template nbs<P0 ^ M ^ 1, 0, T>());
    un = c.template horner<P0 ^ M, 0, MODD>(un + A, A, o.S1, a.m);
}

template <int NT>
struct CqSetup {
    int lane_, wave, chain, qd, slab, col, g;      // wave: my block; chain: 0 = forward sweep / state chain, 1 = adjoint chain
    int used;           // columns of the slab that carry a state
    bool active;
    size_t foff;        // offset of my element in an array image of the slab file
};
template <int NT>
__device__ __forceinline__ CqSetup<NT> cq_setup(const PropArgs& a, int slab, int qd)
{
    CqSetup<NT> s;
    s.lane_ = threadIdx.x & 63;
    s.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    s.chain = s.wave >= NT ? 1 : 0;
    s.wave -= s.chain * NT;
    s.slab = slab;
    s.qd = qd;
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
    s.used = used;
    s.active = 4 * s.qd < used;
    return s;
}
template <int NT>
__device__ __forceinline__ CqSetup<NT> cq_setup(const PropArgs& a)
{
    return cq_setup<NT>(a, (int)blockIdx.x >> 2, (int)blockIdx.x & 3);
}

// ---------------------------------------------------------------------------------------------
// grid = 4 * nslabs (workgroup = quad qd of slab blockIdx.x / 4), block = 64 * (NT + 2); 5 + 2 m barriers per time step.
// The two extra waves only stage: they pass the barriers and, behind the last one of step n, issue the DMA of the time points
// 2n+5, 2n+6 into the slots of 2n, 2n+1 (whose operators everybody has loaded) -- ~25 instructions per step that would
// otherwise sit on the critical path of the six propagating waves (a wave issues one instruction every ~10 cycles).
// NS = 2 (257 .. 512 column quads on 256 CUs; round 4): TWO column quads per workgroup, grid = 2 * nslabs -- workgroup b works on the
// quads 2 (b & 1) and 2 (b & 1) + 1 of slab b / 2, every array a pair of doubles per lane (D2), quad B in channel 1 of the exchange
// image.  The same barriers serve both quads: a publication interval of the forward sweep (two waves per SIMD) is bound by latency,
// not by issue, so twice the work per interval costs ~ 1.5 x -- against 2 x for two rounds of workgroups.  (The backward sweep's
// twelve waves are issue-bound and hold 150 registers: it stays at one quad per workgroup.)
// WLR: full leakage weights in four slots (CqW).  The propagating waves do NOTHING for them: staging wave 0, which otherwise only
// waits, reads the blocks of vi05 and vr(t_n+1) that they publish anyway (exchange image, behind the publication's barrier and before
// the next one lets the image be overwritten), forms the dots a_k . x of the four columns and accumulates
// lam_k [(a_k . vr(t_n))^2 + (a_k . vr(t_n+1))^2 + 2 (a_k . vi05)^2] per slot, complex W: - 2 lam_k [(b_k . vi05)(a_k . vr(t_n)) - (a_k . vi05)
// (b_k . vr(t_n))] on top  (penalf2aTrap, penalf2a, penalf2imag: src/evalobjgrad.jl:700, :716-718, :2170-2233).
// (First version: partial dots by the propagating waves, 34 instructions per step in lock-step: forward sweep 71 -> 91 ms; behind the
//  last barrier instead of in front of it 86 ms; this version 71.)
template <int NT, bool MODD, int NS = 1, bool WLR = false, bool DN = false>
__global__ __launch_bounds__(64 * NT + 128) void k_forward_cq(PropArgs a)
{
    static_assert(!WLR || NS == 1, "full weights: one column quad per workgroup");
    static_assert(!DN || (NS == 1 && !WLR), "dense policy: one column quad per workgroup, Diagonal weights");
    typedef typename std::conditional<NS == 2, D2, double>::type T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KT = 4 * NT;
    const CqSetup<NT> s = NS == 2 ? cq_setup<NT>(a, (int)blockIdx.x >> 1, 2 * ((int)blockIdx.x & 1)) : cq_setup<NT>(a);
    if (!s.active) return;      // (a quad without columns: the whole workgroup leaves before any barrier)
    const bool actB = NS == 2 && 4 * (s.qd + 1) < s.used;      // (the slab's last quad pair may be half empty)
    const int lane_ = s.lane_, wave = s.wave;
    double* tab = (double*)(smem + a.lds_tab_off);
    for (int i = threadIdx.x; i < 32 * NT; i += blockDim.x) tab[(i & ~15) + 4 * (i & 3) + ((i >> 2) & 3)] = a.tabs[i];   // [block][g][r]
    CoopQ<NT, DN> c;
    double* scratch = tab + 32 * NT + 2 * CoopQ<NT, DN>::PAR;
    c.setup(tab + 32 * NT, s.chain ? 0 : wave, lane_);
    c.ring.init(smem, a, wave + NT * s.chain, lane_, NT + 2);      // (barrier inside)
    if (s.chain) {      // staging waves
        c.ring.wave = wave, c.ring.nwaves = 2;
        const int mm = a.m > 0 ? a.m : 0;
        const int nb = 4 + 2 * mm;
        const bool wsum = WLR && wave == 0;
        double wacc = 0.0;
        if constexpr (!WLR) {
            for (int n = 0; n < a.nsteps_chunk; ++n) {
                for (int k = 0; k < nb; ++k) __builtin_amdgcn_s_barrier();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (my pieces of the time points 2n+3, 2n+4)
                __builtin_amdgcn_s_barrier();
                c.ring.issue_next();
                c.ring.issue_next();
            }
        } else {
        double wp0 = 0.0, wlam = 0.0, wlamx = 0.0, wrr = 0.0;
        // a_k as the A operand of v_mfma_f64_4x4x4 (lane 16 i + 4 b + k holds a_k[16 w + 4 b + i], jq_kernels.h mm_t4q): the MFMA of block w
        // adds, for every 4-row group b, the products of the group's four rows with the four columns -- D[k][j] on lane 16 k + 4 b + j
        double akA[NT];
        if (wsum) {
            const bool mine = (lane_ >> 4) < CqW::nslots(a) && ((lane_ >> 2) & 3) == 0;      // (one lane per slot and column accumulates)
            wlam = mine ? CqW::lam(a, lane_ >> 4) : 0.0;
            // complex W: - 2 lam (s p0 - r q0) with (r, s) = (a, b) . vi05, (p0, q0) = (a, b) . vr(t_n): the rows of a term's two slots crossed
            wlamx = (mine && a.wcplx) ? ((lane_ >> 4) & 1 ? 2.0 : -2.0) * wlam : 0.0;
#pragma unroll
            for (int w = 0; w < NT; ++w)
                akA[w] = (lane_ & 3) < CqW::nslots(a) ? CqW::slot(a, lane_ & 3, 16 * w + 4 * ((lane_ >> 2) & 3) + (lane_ >> 4)) : 0.0;
        }
        // The dots with a vector published in channel 0 of parity `par` in three phases, one per barrier interval (the whole of it in one
        // interval made this wave late at the next barrier: forward sweep 71 -> 100 ms): load the NT blocks | NT MFMAs with the a_k
        // operands | add the 4-row groups.  (This wave shares its SIMD with a propagating wave: 24 FMAs + a 9-instruction lane
        // reduction instead of the MFMAs cost 81 ms.)
        double X[NT], acc = 0.0;
        auto wload = [&](int par) {
            const jq_lds_double* x = c.xb + par * CoopQ<NT, DN>::PAR + 64;
#pragma unroll
            for (int w = 0; w < NT; ++w) X[w] = x[w * 64];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (read before this wave arrives at the next barrier)
        };
        auto wfma = [&]() {
            acc = 0.0;
#pragma unroll
            for (int w = 0; w < NT; ++w) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(akA[w], X[w], acc, 0, 0, 0);
        };
        auto wred = [&]() { return row_ror_add<8>(row_ror_add<4>(acc)); };      // (over the 4-row groups: row k = term k, every lane its column)
        // publications of a step: u | v | m x | vi05 | u' | m x | u(t_n+1); publication k of the chunk has parity k & 1, 5 + 2 m per step.
        //   vr(t_n) = the u of step n: load behind barrier 1, multiply-add behind 2, reduce + accumulate step n - 1 behind 3
        //   vi05 of step n:            load behind barrier 3 + m, multiply-add behind 4 + m, reduce behind 5 + m
        //   vr(t_N) after the last step of the chunk: all phases behind its last barrier
        // (This wave must reach every barrier before the propagating waves do -- an interval is ~ 230 cycles and a taken branch costs a
        //  lone wave 20 ... 120 of them: one short piece of straight-line code per interval, none in the interval of the DMA issue.
        //  Everything behind the last barrier: forward sweep 100 ms instead of 71; a loop over the barriers with the phase tests inside: 108;
        //  vr(t_n+1) loaded behind the last barrier, next to the DMA issue: 84.)
        auto bar = [&]() {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        auto waccum = [&]() {
            const double p1 = wred();
            wacc += wlam * ((wp0 * wp0 + p1 * p1) + 2.0 * (wrr * wrr));
            wacc = fma(wlamx, __shfl_xor(wrr, 16) * wp0, wacc);      // (0 for a real W; row k ^ 1 = the other slot of the term)
            wp0 = p1;
        };
        for (int n = 0; n < a.nsteps_chunk; ++n) {
            bar();      // 1: u = vr(t_n) is published
            if (wsum) wload(n & 1);
            bar();      // 2
            if (wsum) wfma();
            if (mm > 0) {
                bar();      // 3
                if (wsum) {
                    if (n > 0) waccum();
                    else wp0 = wred();
                }
                for (int k = 1; k < mm; ++k) bar();      // 4 .. 2 + m
            }
            bar();      // 3 + m: vi05 is published
            if (wsum) {
                if (mm == 0) {
                    if (n > 0) waccum();
                    else wp0 = wred();
                }
                wload((n ^ mm) & 1);
            }
            bar();      // 4 + m
            if (wsum) wfma();
            if (mm > 0) {
                bar();      // 5 + m
                if (wsum) wrr = wred();
                for (int k = 1; k < mm; ++k) bar();      // 6 + m .. 4 + 2 m
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (my pieces of the time points 2n+3, 2n+4)
            bar();      // 5 + 2 m: vr(t_n+1) is published
            c.ring.issue_next();
            c.ring.issue_next();
            if (wsum && mm == 0) wrr = wred();
        }
        if (wsum && a.nsteps_chunk > 0) {
            wload((a.nsteps_chunk - 1) & 1);
            wfma();
            waccum();
        }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int k = 0; k < NS; ++k) {
            __syncthreads();      // (cq_wg_sum of the propagating waves)
            if (wsum) scratch[NT * 64 + lane_] = wacc;
            __syncthreads();
        }
        return;
    }
    const double wdr = tab[16 * wave + s.g], wsr = tab[16 * NT + 16 * wave + s.g];

    double* st = a.state + (size_t)s.slab * a.state_stride;
    const bool slot0 = wave == 0 && ((lane_ >> 2) & 3) == 0;      // the lanes that carry per-column partials between chunks
    const size_t cslot = 16 * (lane_ >> 4) + s.col;
    const size_t lslot = (size_t)(JQ_STATE_ARRAYS * KT + JQ_MAXNC) * 64 + cslot;
    T u, v, leak, cw;
    if constexpr (NS == 2) {      // (quad B: four columns further in every image)
        u = D2(st[s.foff], actB ? st[s.foff + 4] : 0.0);
        v = D2(st[(size_t)KT * 64 + s.foff], actB ? st[(size_t)KT * 64 + s.foff + 4] : 0.0);
        leak = D2(slot0 ? st[lslot] : 0.0, slot0 && actB ? st[lslot + 4] : 0.0);
        cw = D2(0.5 * a.h * a.colinfo[(size_t)s.slab * 32 + s.col] * wsr, actB ? 0.5 * a.h * a.colinfo[(size_t)s.slab * 32 + s.col + 4] * wsr : 0.0);
    } else {
        u = st[s.foff], v = st[(size_t)KT * 64 + s.foff];
        leak = slot0 ? st[lslot] : 0.0;
        cw = 0.5 * a.h * a.colinfo[(size_t)s.slab * 32 + s.col] * wsr;      // h/2 eps ws[row]
    }
    CqOps<NT, DN> o = cq_first_ops<NT>(c);

    auto hist = [&](int n, int colq, double hu, double hv) {
        const int scol = a.parts > 1 ? 16 * s.slab + colq : colq;
        const int row = 16 * wave + 4 * ((lane_ >> 2) & 3) + (lane_ >> 4);
        if (s.slab < a.parts && scol < a.N && row < a.Ntot) {
            const size_t off = (size_t)(a.step0 + n + 1) * a.Ntot * a.N + (size_t)scol * a.Ntot + row;
            a.hist_r[off] = hu;
            a.hist_i[off] = -hv;
        }
    };
    // one time step whose first publication has parity P0 (the next step's has P0 ^ 1: 5 + 2 m publications)
    auto step = [&](auto P0c, int n) {
        constexpr int P0 = decltype(P0c)::value;
        leak = fma(wdr, u * u, leak);      // trapezoidal part at t_n (src/evalobjgrad.jl:700)
        T un, v05, vN;
        cq_state<NT, P0, MODD>(c, a, o, cw, u, v, un, v05, vN);
        // Kp05 again: v(t+h) = v05 + c (K05 u_new + S05 v05)
        c.template post<P0, 0>(un);
        v = c.own(vN, o.Kp05, c.sh(un));
        if (a.use_shift) v = fma(cw, un, v);
        c.sync();
        v = c.nbr(v, o.Kp05, c.template nbs<P0, 0, T>());
        cq_next_ops<NT>(c, o);      // (the staging waves drained the DMA of its time points in front of this barrier)
        u = un;
        leak = leak + (wdr * (u * u) + 2.0 * (wdr * (v05 * v05)));      // (:716, penalf2a :2170-2180)
        if (a.hist_r) {
            if constexpr (NS == 2) {
                hist(n, s.col, u.a, v.a);
                if (actB) hist(n, s.col + 4, u.b, v.b);
            } else {
                hist(n, s.col, u, v);
            }
        }
    };
    int n = 0;
    for (; n + 1 < a.nsteps_chunk; n += 2) {
        step(std::integral_constant<int, 0>{}, n);
        step(std::integral_constant<int, 1>{}, n + 1);
    }
    if (n < a.nsteps_chunk) step(std::integral_constant<int, 0>{}, n);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (NS == 2) {
        st[s.foff] = u.a;
        st[(size_t)KT * 64 + s.foff] = v.a;
        if (actB) {
            st[s.foff + 4] = u.b;
            st[(size_t)KT * 64 + s.foff + 4] = v.b;
        }
        const double totA = cq_wg_sum(leak.a, scratch, wave, lane_, NT);
        const double totB = cq_wg_sum(leak.b, scratch, wave, lane_, NT);
        if (slot0) st[lslot] = totA;
        if (slot0 && actB) st[lslot + 4] = totB;
    } else {
        st[s.foff] = u;
        st[(size_t)KT * 64 + s.foff] = v;
        const double tot = cq_wg_sum(leak, scratch, wave, lane_, WLR ? NT + 1 : NT);      // (WLR: + staging wave 0's low-rank terms)
        if (slot0) st[lslot] = tot;
    }
}

// ---------------------------------------------------------------------------------------------
// grid = 4 * nslabs, block = 128 * NT: waves 0 .. NT-1 re-integrate the state (channel 0), waves NT .. 2 NT-1 run the adjoint
// step (channel 1); the trace products of adjoint_grad_calc! are shared between them (the state waves take the two traces
// with vi05).  Both sets pass the same 5 + 2 m barriers per time step:
//   (u | nb)  (v | mu)  m x Neumann  (v05 | X)  (un' | Lk)  m x Neumann  (un | nb_new, -(li0 + li))
// Staging (WinRing): behind the last barrier of step n the state waves -- which wait for the adjoint waves there -- issue the
// DMA of the time points 2n+5, 2n+6 and drain it in front of the last barrier of step n+1.
// Trace scalars: a wave reduces its per-lane values four at a time to 16 column partials per value (cq_part4) and leaves that
// register in LDS, red[group][block][64]; behind the first barrier of the NEXT step adjoint wave g adds the NT blocks of group g,
// finishes the four sums with rotate-adds inside the rows and writes them to the trace record of the step.
//   group q < Nc (adjoint wave):  rows 0, 1, 2 = t1, t4, t3 of control q
//   group Nc + j (state wave):    rows 0, 1 = t2, t5 of control 2 j, rows 2, 3 = t2, t5 of control 2 j + 1
// WLR: full (real, rank <= JQ_CQ_WRANK) leakage weights: the forcing terms hr0 = W vr(t_n+1) / T, hi0 = hi1 = W vi05 / T,
// hr1 = W vr(t_n) / T (src/evalobjgrad.jl:862, :882-888) from the dots the state waves leave with their publications of vi05 and
// vr(t_n) (CqW); W vr(t_n) is next step's W vr(t_n+1).
template <int NT, bool MODD, bool ORD, bool WLR = false, bool DN = false>
__global__ __launch_bounds__(128 * NT) void k_backward_cq(PropArgs a)
{
    static_assert(!DN || (!WLR && !ORD), "dense policy: Diagonal weights, whole trace products");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KT = 4 * NT;
    constexpr int M = MODD ? 1 : 0;
    typedef typename CoopQ<NT, DN>::Sh Sh;
    typedef typename CoopQ<NT, DN>::Nb Nb;
    typedef typename CoopQ<NT, DN>::Op Op;
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
    CoopQ<NT, DN> c;
    double* scratch = tab + 32 * NT + 2 * CoopQ<NT, DN>::PAR;      // [2 NT][64]: the workgroup sums after the last time step
    const int ntr = Nc * JQ_NTR, ngroups = Nc + (Nc + 1) / 2;
    double* red = scratch;                                      // [ngroups][NT][64]: trace hand-off (dead by then: same LDS)
    c.setup(tab + 32 * NT, wave, lane_);
    c.ring.init(smem, a, wave_all, lane_, 2 * NT);
    c.ring.wave = wave, c.ring.nwaves = NT;      // (from here on the state waves stage)
    const double wdr = tab[16 * wave + s.g], wsr = tab[16 * NT + 16 * wave + s.g];
    double* st = a.state + (size_t)s.slab * a.state_stride;
    const double cw = 0.5 * a.h * a.colinfo[(size_t)s.slab * 32 + s.col] * wsr;      // h/2 eps ws[row]
    const double wgt = a.colinfo[(size_t)s.slab * 32 + 16 + s.col];
    CqW wq;
    if constexpr (WLR) wq.init(a, smem, wave, lane_);
    const size_t cslot = 16 * (lane_ >> 4) + s.col;
    double carry[JQ_MAXNC];
#pragma unroll
    for (int q = 0; q < JQ_MAXNC; ++q) carry[q] = 0.0;
    // my blocks of the constant trace images (Hsym_q: image q, Hanti_q: image Nc + q) are loaded where they are used (trace_mm);
    // ORD: control q acts on subsystem q only (host: a.bw_trace[q] == 1 << q for all q < Ncoupled <= 3), else branch-free full products
    double* redw = red + (size_t)wave * 64 + lane_;      // my block's slot of group 0
    // the trace scalars of step k (adjoint waves; call behind a barrier that follows the step's last hand-off)
    auto finish_traces = [&](int k) {
        for (int g = wave; g < ngroups; g += NT) {
            const double* r = red + (size_t)g * NT * 64 + lane_;
            double sum = r[0];
#pragma unroll
            for (int w = 1; w < NT; ++w) sum += r[w * 64];
            sum = row_ror_add<8>(sum);
            sum = row_ror_add<4>(sum);
            sum = row_ror_add<2>(sum);
            sum = row_ror_add<1>(sum);
            const int row = lane_ >> 4;
            int q, kk;
            if (g < Nc)
                q = g, kk = row == 0 ? 0 : row == 1 ? 3 : row == 2 ? 2 : -1;
            else
                q = 2 * (g - Nc) + (row >> 1), kk = (row & 1) ? 4 : 1;
            if ((lane_ & 15) == 0 && kk >= 0 && q < Nc) a.traces[(trow * a.nsteps_chunk + k) * ntr + q * JQ_NTR + kk] = sum;
        }
    };

    if (s.chain == 0) {
        // ---- state re-integration (src/evalobjgrad.jl:879), channel 0; traces t2 = tr(vi05' Hsym_q X), t5 = tr(vi05' Hanti_q (li0+li))
        double u = st[s.foff], v = st[(size_t)KT * 64 + s.foff];
        Op Kp05 = c.load(c.ring.template ks<0, 1>()), S0 = c.load(c.ring.template ks<1, 0>());
        if (a.first_chunk) {      // (vr(T) for the carry products of the adjoint waves; parity 1: the first step starts with 0)
            c.template post<1, 0>(u);
            c.sync();
        }
        if constexpr (WLR) {      // (the dots with the state the chunk starts from: hr0 of its first step)
            wq.template put<NT>(0, wave, lane_, u);
            c.sync();
        }
        JQ_TS_DECL
        auto step = [&](auto P0c, int n) {
            constexpr int P0 = decltype(P0c)::value;
            JQ_TS(n)
            double un, v05, vN, t2[JQ_MAXNC];
            // x = u: A = c K05 u ; P = u + c S0 u
            c.template post<P0, 0>(u);
            const Op S05 = c.load(c.ring.template ks<1, 1>());
            double A, P;
            {
                const Sh sx = c.sh(u);
                A = c.own(0.0, Kp05, sx);
                P = c.own(u, S0, sx);
                if (a.use_shift) A = fma(cw, u, A);
                JQ_TS(n)
                c.sync();
                JQ_TS(n)
                const Nb nn = c.template nbs<P0, 0>();
                A = c.nbr(A, Kp05, nn);
                P = c.nbr(P, S0, nn);
            }
            // x = v: A = c (K05 u + S05 v) ; v05 = v + sum_j S^j A
            c.template post<P0 ^ 1, 0>(v);
            A = c.own(A, S05, c.sh(v));
            JQ_TS(n)
            c.sync();
            JQ_TS(n)
            A = c.nbr(A, S05, c.template nbs<P0 ^ 1, 0>());
            Op Kn0, Kn1;
            v05 = c.template horner<P0, 0, MODD>(v + A, A, S05, a.m, [&] {
                Kn0 = c.load(c.ring.template ks<0, 0>());
                Kn1 = c.load(c.ring.template ks<0, 2>());
            });
            JQ_TS(n)
            // x = v05: vN = v05 + c S05 v05 ; un = u + c (S0 u - K0 v05) ; A = -c K1 v05
            c.template post<P0 ^ M, 0>(v05);
            if constexpr (WLR) wq.template put<NT>(1, wave, lane_, v05);
            const Op S1 = c.load(c.ring.template ks<1, 2>());
            {
                const Sh sx = c.sh(v05);
                vN = c.own(v05, S05, sx);
                un = c.own(P, Kn0, sx);
                A = c.own(0.0, Kn1, sx);
                if (a.use_shift) {
                    un = fma(-cw, v05, un);
                    A = fma(-cw, v05, A);
                }
                JQ_TS(n)
                c.sync();
                JQ_TS(n)
                const Nb nn = c.template nbs<P0 ^ M, 0>();
                vN = c.nbr(vN, S05, nn);
                un = c.nbr(un, Kn0, nn);
                A = c.nbr(A, Kn1, nn);
            }
            const double v05w = v05 * wgt;
            // x = un: A = c (S1 un - K1 v05)
            c.template post<P0 ^ M ^ 1, 0>(un);
            A = c.own(A, S1, c.sh(un));
            // (under the barrier: the adjoint chain's X of the last publication)
            {
                const Sh sx = c.sh(c.template block<P0 ^ M, 1>());
                const Nb nx = c.template nbs<P0 ^ M, 1>();
#pragma unroll
                for (int q = 0; q < JQ_MAXNC; ++q) {
                    t2[q] = 0.0;
                    if (q < Nc) {
                        t2[q] = v05w * c.template trace_mm<ORD>(c.ring.cimg(q), q, sx, nx);
                    }
                }
            }
            JQ_TS(n)
            c.sync();
            JQ_TS(n)
            A = c.nbr(A, S1, c.template nbs<P0 ^ M ^ 1, 0>());
            un = c.template horner<P0 ^ M, 0, MODD>(un + A, A, S1, a.m, [&] { Kp05 = c.load(c.ring.template ks<0, 1>()); });
            JQ_TS(n)
            // x = un: v(t_n) = v05 + c (K05 un + S05 v05)
            c.template post<P0, 0>(un);
            if constexpr (WLR) wq.template put<NT>(0, wave, lane_, un);
            v = c.own(vN, Kp05, c.sh(un));
            if (a.use_shift) v = fma(cw, un, v);
            JQ_TS(n)
            c.template sync<true>();
            JQ_TS(n)
            v = c.nbr(v, Kp05, c.template nbs<P0, 0>());
            // (the time points of the next step have landed; those of this step are dead)
            c.ring.advance();
            Kp05 = c.load(c.ring.template ks<0, 1>());
            S0 = c.load(c.ring.template ks<1, 0>());
            c.ring.issue_next();
            c.ring.issue_next();
#ifdef JQ_CQ_DMALAT     // experiment: latency of the DMA just issued
            JQ_TS(n)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            JQ_TS(n)
#endif
            u = un;
            // the adjoint chain's -(li0 + li) (channel 2 of the last publication): t5
            {
                const Sh sx = c.sh(c.template block<P0, 2>());
                const Nb nx = c.template nbs<P0, 2>();
                double t5[JQ_MAXNC];
#pragma unroll
                for (int q = 0; q < JQ_MAXNC; ++q) {
                    t5[q] = 0.0;
                    if (q < Nc) {
                        t5[q] = -(v05w * c.template trace_mm<ORD>(c.ring.cimg(Nc + q), q, sx, nx));
                    }
                }
                redw[(size_t)Nc * NT * 64] = cq_part4(t2[0], t2[1], t5[0], t5[1]);
                if (Nc > 2) redw[(size_t)(Nc + 1) * NT * 64] = cq_part4(t2[2], t2[3], t5[2], t5[3]);
            }
        };
        int n = 0;
        for (; n + 1 < a.nsteps_chunk; n += 2) {
            step(std::integral_constant<int, 0>{}, n);
            step(std::integral_constant<int, 1>{}, n + 1);
        }
        if (n < a.nsteps_chunk) step(std::integral_constant<int, 0>{}, n);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();      // (the last hand-off)
        asm volatile("" ::: "memory");
        st[s.foff] = u;
        st[(size_t)KT * 64 + s.foff] = v;
        JQ_TS_PRINT(wave_all)
    } else {
        // ---- adjoint step! with forcing (src/StormerVerlet.jl:255-303), channel 1; traces t1 = tr(vr0' Hanti_q X),
        //      t3 = tr(vr' Hanti_q X), t4 = tr(vr' Hsym_q li) + tr(vr0' Hsym_q li0)
        double mu = st[(size_t)2 * KT * 64 + s.foff], nb = st[(size_t)3 * KT * 64 + s.foff];
        Op Kp05 = c.load(c.ring.template ks<0, 1>()), S05 = c.load(c.ring.template ks<1, 1>());
        const double cfw = (a.forced ? 0.5 * a.h * a.tinv : 0.0) * wdr;      // forcing weight c tinv wd[row]; 0 for step_no_forcing!
        const double wcf = WLR ? CqW::coef(a, wave, lane_, a.forced ? 0.5 * a.h * a.tinv : 0.0) : 0.0;      // full weights: c tinv lam_k a_k[row] as an A operand
        double Wu = 0.0;      // c tinv (W vr(t_n+1))[row, column]
        const bool slot0 = wave == 0 && ((lane_ >> 2) & 3) == 0;      // the lanes that carry per-column partials between chunks
#pragma unroll
        for (int q = 0; q < JQ_MAXNC; ++q)
            if (q < Nc && slot0) carry[q] = st[(size_t)(JQ_STATE_ARRAYS * KT + q) * 64 + cslot];
        if (a.first_chunk) {
            // carry_q = tr(vr' Hsym_q lambdai) at t = T (see k_backward)
            c.template post<1, 1>(nb);
            const Sh sx = c.sh(nb);
            c.sync();
            const double u0 = c.template block<1, 0>();
            const Nb nn = c.template nbs<1, 1>();
#pragma unroll
            for (int q = 0; q < JQ_MAXNC; ++q)
                if (q < Nc) {
                    carry[q] = -(u0 * c.template trace_mm<ORD>(c.ring.cimg(q), q, sx, nn));
                }
        }
        if constexpr (WLR) {
            c.sync();
            Wu = wq.template apply<NT>(0, lane_, wcf);
        }
        JQ_TS_DECL
        auto step = [&](auto P0c, int n) {
            constexpr int P0 = decltype(P0c)::value;
            JQ_TS(n)
            // x = nb (-lambda_i): L = c K05 nb, Tn = c S05 nb (for the second half of the step)
            c.template post<P0, 1>(nb);
            const Op S0 = c.load(c.ring.template ks<1, 0>());
            double L, Tn;
            {
                const Sh sx = c.sh(nb);
                L = c.own(0.0, Kp05, sx);
                Tn = c.own(0.0, S05, sx);
                if (a.use_shift) L = fma(cw, nb, L);
                JQ_TS(n)
                c.sync();
                JQ_TS(n)
                const Nb nn = c.template nbs<P0, 1>();
                L = c.nbr(L, Kp05, nn);
                Tn = c.nbr(Tn, S05, nn);
            }
            const double u = c.template block<P0, 0>();      // vr before the state step (:862)
            if (n > 0) finish_traces(n - 1);
            // x = mu: L = c (S0 mu - K05 li + hr0) ; X = mu + sum_j S0^j L
            c.template post<P0 ^ 1, 1>(mu);
            L = c.own(L, S0, c.sh(mu));
            L = fma(cfw, u, L);
            if constexpr (WLR) L += Wu;
            JQ_TS(n)
            c.sync();
            JQ_TS(n)
            L = c.nbr(L, S0, c.template nbs<P0 ^ 1, 1>());
            Op Kn0, Kn1, S1;
            const double X = c.template horner<P0, 1, MODD>(mu + L, L, S0, a.m, [&] {
                Kn0 = c.load(c.ring.template ks<0, 0>());
                Kn1 = c.load(c.ring.template ks<0, 2>());
                S1 = c.load(c.ring.template ks<1, 2>());
            });
            JQ_TS(n)
            // x = X: Lk = -c K0 X, Q = -c K1 X, SX = c S1 X, Hanti_q X (tr1, tr3)
            c.template post<P0 ^ M, 1>(X);
            double Lk, Q, SX, Tq[JQ_MAXNC];
            {
                const Sh sx = c.sh(X);
                Lk = c.own(0.0, Kn0, sx);
                Q = c.own(0.0, Kn1, sx);
                SX = c.own(0.0, S1, sx);
                if (a.use_shift) {
                    Lk = fma(-cw, X, Lk);
                    Q = fma(-cw, X, Q);
                }
                JQ_TS(n)
                c.sync();
                JQ_TS(n)
                const Nb nn = c.template nbs<P0 ^ M, 1>();
                Lk = c.nbr(Lk, Kn0, nn);
                Q = c.nbr(Q, Kn1, nn);
                SX = c.nbr(SX, S1, nn);
#pragma unroll
                for (int q = 0; q < JQ_MAXNC; ++q) {
                    Tq[q] = 0.0;
                    if (q < Nc) {
                        Tq[q] = c.template trace_mm<ORD>(c.ring.cimg(Nc + q), q, sx, nn);
                    }
                }
            }
            // Lk = -c l2 = -c (K0 X + S05 li + hi0) ; Q = -c (S05 (li + c l2) + K1 X + hi1)
            {
                const double v05 = c.template block<P0 ^ M, 0>();
                double Pn = fma(-cfw, v05, Tn);
                if constexpr (WLR) Pn -= wq.template apply<NT>(1, lane_, wcf);
                Lk += Pn;
                Q += Pn;
            }
            // x = Lk: Q += c S05 Lk ; nb_new = nb + Lk + sum_j S05^j Q
            c.template post<P0 ^ M ^ 1, 1>(Lk);
            Q = c.own(Q, S05, c.sh(Lk));
            JQ_TS(n)
            c.sync();
            JQ_TS(n)
            Q = c.nbr(Q, S05, c.template nbs<P0 ^ M ^ 1, 1>());
            const double nbn = c.template horner<P0 ^ M, 1, MODD>((nb + Lk) + Q, Q, S05, a.m);
            JQ_TS(n)
            const double Bq = nb + nbn;      // -(li0 + li)
            // x = nb_new: lambda_r_new = X + c (S1 X - K05 li_new + hr1), Hsym_q li_new (tr4); vr(t_n) exists behind the barrier
            c.template post<P0, 1>(nbn);
            c.template post<P0, 2>(Bq);      // (channel 2: for the state waves' tr5)
            double G;
            {
                const Sh sx = c.sh(nbn);
                G = c.own(X, Kp05, sx);
                if (a.use_shift) G = fma(cw, nbn, G);
                G += SX;
                JQ_TS(n)
                c.sync();
                JQ_TS(n)
                const Nb nn = c.template nbs<P0, 1>();
                const double un = c.template block<P0, 0>();
                G = c.nbr(G, Kp05, nn);
                G = fma(cfw, un, G);
                if constexpr (WLR) {
                    Wu = wq.template apply<NT>(0, lane_, wcf);      // (hr1 of this step, hr0 of the next)
                    G += Wu;
                }
                // (the time points of the next step have landed)
                c.ring.advance();
                Kp05 = c.load(c.ring.template ks<0, 1>());
                S05 = c.load(c.ring.template ks<1, 1>());
                const double uw = u * wgt, unw = un * wgt;
#pragma unroll
                for (int q = 0; q < JQ_MAXNC; ++q)
                    if (q < Nc) {
                        const double pq = -(un * c.template trace_mm<ORD>(c.ring.cimg(q), q, sx, nn));
                        const double t4 = (pq + carry[q]) * wgt;
                        carry[q] = pq;
                        redw[(size_t)q * NT * 64] = cq_part4(uw * Tq[q], unw * Tq[q], t4, 0.0);      // rows 0, 2, 1: t1, t3, t4
                    }
            }
            mu = G;
            nb = nbn;
        };
        int n = 0;
        for (; n + 1 < a.nsteps_chunk; n += 2) {
            step(std::integral_constant<int, 0>{}, n);
            step(std::integral_constant<int, 1>{}, n + 1);
        }
        if (n < a.nsteps_chunk) step(std::integral_constant<int, 0>{}, n);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();      // (the last hand-off)
        asm volatile("" ::: "memory");
        finish_traces(a.nsteps_chunk - 1);
        st[(size_t)2 * KT * 64 + s.foff] = mu;
        st[(size_t)3 * KT * 64 + s.foff] = nb;
        JQ_TS_PRINT(wave_all)
    }
#pragma unroll
    for (int q = 0; q < JQ_MAXNC; ++q)
        if (q < Nc) {
            const double tot = cq_wg_sum(carry[q], scratch, wave_all, lane_, 2 * NT);
            if (wave_all == 0 && ((lane_ >> 2) & 3) == 0) st[(size_t)(JQ_STATE_ARRAYS * KT + q) * 64 + cslot] = tot;
        }
}
