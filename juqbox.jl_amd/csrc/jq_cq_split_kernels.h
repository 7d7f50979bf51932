// jq_cq_split_kernels.h -- the backward sweep of the cooperative-quad kernels (jq_cq_kernels.h) on THREE workgroups per column quad:
// the latency path of the JQ_BW_T4 structure for single evaluations and small ensembles (3 x quads <= CUs).
//
// k_backward_cq runs the state re-integration and the adjoint step on two sets of NT waves of ONE workgroup: both sets pass the same
// 5 + 2 m barriers per time step, and with three waves per SIMD a publication interval is bound by the SIMD's issue rate (340 .. 405
// cycles per Neumann publication against 232 in the forward sweep, two waves per SIMD).  Measured (round 4, state waves idling at the
// barriers): the adjoint chain alone takes 92 ms of the 119 ms per cnot3 evaluation -- 21 ms of them in the trace products -- and
// the state chain alone is the forward sweep (71 ms).  But the coupling between the chains is ONE-WAY and per block: the adjoint step
// of time step n needs vr(t_n+1), vi05, vr(t_n) of the wave's own 16-row block only (its forcing), and the trace products of
// adjoint_grad_calc! need those three and X, -lambda_i(new), -(li0 + li) with their neighbouring blocks.  So the three jobs run as a
// software pipeline over three workgroups on three CUs:
//   role 0  state re-integration  (the state path of k_backward_cq minus its trace products): stores u, v05, un of every step
//   role 1  adjoint step          (the adjoint path minus its trace products), >= 2 steps behind (it fetches a step ahead): loads u, v05, un;
//                                 stores X, nbn, Bq
//   role 2  trace products        (no publications, no LDS exchange), behind role 1: loads the six arrays with their neighbouring
//                                 blocks, forms the 5 Ncoupled scalars of the step and writes the trace record
// through a ring of JQ_CQ3_SLOTS time steps in global memory, [slot][array][block][64] doubles per quad, with three progress counters
// (steps finished by role 0 / 1 / 2).  Role r + 1 waits for role r's counter; role 0 waits for role 2's before it reuses a slot.
// Everything a role stores in a step is acknowledged by the L2 (s_waitcnt vmcnt(0)) in front of the SECOND barrier of the next step --
// more than a publication interval later, so that the store latency is off the critical path --, the counter is written behind it;
// the consumers read counters and data with agent-scope loads (past their CU's vector cache), one step ahead of their use.  The three workgroups of a quad have block indices 24 i + j, + 8, + 16 (j < 8): workgroups are
// handed to the eight XCDs round-robin, so the three share one XCD and its L2 -- no cache maintenance between them.  Each role checks
// that (XCC_ID register) and that no wait exceeds ~ 1.3 s; otherwise it raises the error word of the quad and every wait of the quad
// is abandoned: the launch ends with garbage, the host falls back to k_backward_cq and disables the split for the handle.
// The arithmetic of every chain and of the trace sums is k_backward_cq's, operation for operation: bit-identical results.
#pragma once
#include "jq_cq_kernels.h"

#define JQ_CQ3_SLOTS 8        // ring depth in time steps
#define JQ_CQ3_ARRAYS 8       // u (vr before the state step), v05, un, X, nbn (-lambda_i new), Bq (-(li0 + li)); full weights: the blocks' partial dots with v05, un (CqW::part)
#define JQ_CQ3_TAIL 64        // doubles behind a quad's ring: full weights, the dots with the state the chunk starts from (written once per launch, see below)
#define JQ_CQ3_HEAD 64        // doubles in front of a quad's ring: [0] steps of role 0, [8] role 1, [16] role 2, [24] error, [32 + r] XCC of role r
// Waiting.  Rounds 4 - 5 HOPED that the workgroups of a quad were resident together and policed it with one long timeout per wait
// (1 000 000 polls ~ 1.3 s: next to a process whose launches hold every CU for 0.15 - 0.37 s a role legitimately waited that long for its
// partners to START).  Round 6 separates the two questions.  (1) Are all workgroups of this launch resident?  Answered once, at the start,
// by a rendezvous of the WHOLE grid (cq3_rendezvous): every workgroup counts itself in; the LAST one to arrive sets the launch's state word
// to GO, a workgroup that has polled a.rdv_polls times (the host passes about one launch duration, 2 .. 100 ms) sets it to ABANDON -- both
// with a compare-and-swap from 0, so the decision is made ONCE and is the same for every workgroup (first version: "count reached
// gridDim.x" against "my time is up" -- workgroups that gave up left, later arrivals completed the count and went on without them:
// the soak next to two load processes reported their quads as "roles on different XCDs", profiles/r06_cq3_soak_load.txt (a)).
// Abandoned: error word 3, every workgroup leaves at once, the host repeats the evaluation on the one-workgroup kernel -- a busy GPU
// costs milliseconds, not a dead wait.  (2) After a passed rendezvous every partner IS resident
// and stays so (workgroups are not preempted), so a wait between roles can only be a short one; a.wait_polls (the host passes ~ 10 x the
// launch's expected duration) is a guard against the impossible, not a scheduling assumption.
__device__ __forceinline__ void cq3_arrive(const PropArgs& a)      // (workgroups without work count too: the grid must be complete)
{
    if (threadIdx.x == 0) __hip_atomic_fetch_add((unsigned long long*)a.park + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// call from every wave of the workgroup that is still alive, after cq3_arrive; flag: one int of LDS nobody uses yet.  True: proceed.
// (a.park: [0] error word of the evaluation, [1] arrival counter, [2] state word of the launch: 0 undecided, 1 GO, 2 ABANDON; the host
//  zeroes [1] and [2] before every launch)
__device__ __forceinline__ bool cq3_rendezvous(const PropArgs& a, int* flag)
{
    unsigned long long* gerr = (unsigned long long*)a.park;
    if (threadIdx.x == 0) {
        unsigned long long st = 0ull;
        if (__hip_atomic_load(gerr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned long long)gridDim.x) {      // the grid is complete: GO, unless somebody gave up first
            unsigned long long expect = 0ull;
            __hip_atomic_compare_exchange_strong(gerr + 2, &expect, 1ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        for (int k = 0; k < a.rdv_polls; ++k) {
            st = __hip_atomic_load(gerr + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (st != 0ull) break;
            if (__hip_atomic_load(gerr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned long long)gridDim.x) {
                unsigned long long expect = 0ull;
                __hip_atomic_compare_exchange_strong(gerr + 2, &expect, 1ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                continue;      // (read the decided state in the next round)
            }
            __builtin_amdgcn_s_sleep(2);
        }
        if (st == 0ull) {      // my time is up: ABANDON -- unless the launch was decided in this very moment
            unsigned long long expect = 0ull;
            st = __hip_atomic_compare_exchange_strong(gerr + 2, &expect, 2ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 2ull : expect;
        }
        if (st != 1ull) __hip_atomic_store(gerr, 3ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = st == 1ull ? 1 : 0;
    }
    __syncthreads();
    const bool ok = *flag != 0;
    __syncthreads();
    return ok;
}

template <int NT>
struct Cq3Hand {
    double* ring;                     // this lane's element of block 0, array 0, slot 0
    unsigned long long *head, *gerr;
    bool dead;                        // (wave-uniform) a wait of this quad timed out, or the workgroups do not share an XCD
    unsigned long long seen, pend;    // the upstream role's counter: last value known / value asked for by the previous wait
    int spin;                         // polls before a wait is declared dead (a.wait_polls)
    static constexpr size_t SLOT = (size_t)JQ_CQ3_ARRAYS * NT * 64;

    __device__ __forceinline__ void init(const PropArgs& a, size_t quad, int lane_)
    {
        double* base = a.park + JQ_CQ3_HEAD + quad * ((size_t)JQ_CQ3_HEAD + JQ_CQ3_SLOTS * SLOT + JQ_CQ3_TAIL);      // (a.park[0]: the error word of the launch)
        head = (unsigned long long*)base;
        gerr = (unsigned long long*)a.park;
        ring = base + JQ_CQ3_HEAD + lane_;
        dead = false;
        seen = 0ull, pend = 0ull;
        spin = a.wait_polls;
    }
    // the quad's tail area (this lane's element)
    __device__ __forceinline__ double* tail() const { return ring + JQ_CQ3_SLOTS * SLOT; }
    __device__ __forceinline__ size_t off(int step, int arr, int blk) const { return ((size_t)(step & (JQ_CQ3_SLOTS - 1)) * JQ_CQ3_ARRAYS + arr) * NT * 64 + (size_t)blk * 64; }
    // (agent scope: a plain store may rest in the CU's vector cache for a while -- its vmcnt acknowledgement does not mean "in the L2")
    __device__ __forceinline__ void store(int step, int arr, int blk, double x) const
    {
        __hip_atomic_store(ring + off(step, arr, blk), x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // (agent scope: the slot was read JQ_CQ3_SLOTS steps ago -- a line of it may still sit in this CU's vector cache)
    __device__ __forceinline__ double load(int step, int arr, int blk) const
    {
        return __hip_atomic_load(ring + off(step, arr, blk), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ void publish(int role, unsigned long long steps) const
    {
        __hip_atomic_store(head + 8 * role, steps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // wait until role `role` has finished `steps` steps (call on ONE wave, always for the same role; the others meet it at the next
    // workgroup barrier).  A counter read costs an L2 round trip (~ 0.5 us) and the waiting wave holds up its whole workgroup, so the
    // read is taken off the critical path: every call folds in the value it asked for one call ago (`pend`, a load that has had a
    // whole time step to land) and asks for the next one; it only polls when that is not enough (first version: a blocking read per
    // step, 16 ms of the 94 ms of the adjoint workgroup at cnot3).
    __device__ __forceinline__ void wait(int role, unsigned long long steps)
    {
        if (dead) return;
        if (pend > seen) seen = pend;
        if (seen < steps) {
            bool ok = false;
            for (int k = 0; k < spin; ++k) {
                seen = __hip_atomic_load(head + 8 * role, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (seen >= steps) {
                    ok = true;
                    break;
                }
                if (__hip_atomic_load(head + 24, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) break;
                __builtin_amdgcn_s_sleep(2);
            }
            if (!ok) {
                __hip_atomic_store(head + 24, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(gerr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                dead = true;
                return;
            }
        }
        pend = __hip_atomic_load(head + 8 * role, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (used by the next call)
    }
};

// grid = 24 * ceil(quads / 8), block = 64 * (NT + 2): workgroup b has role (b / 8) % 3 of quad 8 (b / 24) + b % 8 (slab = quad / 4);
// roles 0 and 1: NT block waves and two staging waves (as k_forward_cq), role 2: NT block waves (the other two leave).
// a.park: the hand-off buffer (zeroed by the host before every launch).  Dynamic LDS as k_backward_cq.
// NR = 2 (round 5; 2 x quads <= CUs, i.e. 81 .. 128 cnot3 samples): TWO workgroups per quad, grid = 16 * ceil(quads / 8) -- role 0 as above,
// role 1 runs the adjoint step AND all trace products (the adjoint path of k_backward_cq with the state waves' share of the traces:
// its publications carry every neighbouring block the trace products need; nothing but u, vi05, vr(t_n) crosses the ring, role 0
// reuses a slot when role 1 has loaded it).  The trace sums are those of k_backward_cq, term for term: bit-identical results.
// WLR (round 5): full leakage weights in four slots -- real of rank <= 4 or complex of rank <= 2 (CqW, jq_cq_kernels.h).  Role 0 -- which has the slack -- forms the
// dots: its block waves leave their partial dots with vi05 and vr(t_n) in LDS with the publications of those vectors (CqW::put), behind
// the barrier wave 0 (vi05) / wave 1 (vr(t_n)) adds the NT registers and stores the sum as block 0 of array 6 / 7 of the step.  Role 1
// fetches the two registers with the step's other operands, one step ahead, and applies each with ONE MFMA.  W vr(t_n+1) of a step is
// W vr(t_n) of the step before; for the first step of a chunk role 0 leaves the dots of the state the chunk starts from in the quad's
// TAIL area.  (First version: where "step -1" would have left them, array 7 of slot 7 -- which role 0 overwrites at step 7, and role 0
// does not wait for anybody before step 8: when the adjoint workgroup starts late, the dots are gone before it reads them.  Idle GPU:
// every test bit-identical; next to two load processes 43 of 240 evaluations differed -- scripts/soak_cq3_load.py with JQ_SOAK_WEIGHTS=1,
// profiles/r05_cq3_soak_load.txt (d).)  The same operations in the same order as the one-workgroup kernel's (k_backward_cq<.., WLR>):
// bit-identical results.  (First version: the partial registers themselves through the ring, 12 more loads per step in the adjoint
// waves: backward sweep 96 ms instead of 81.)
template <int NT, bool MODD, bool ORD, int NR = 3, bool WLR = false, bool DN = false>
__global__ __launch_bounds__(64 * NT + 128) void k_backward_cq3(PropArgs a)
{
    static_assert(NR == 3 || NR == 2, "workgroups per column quad");
    static_assert(!DN || (!WLR && !ORD), "dense policy (jq_cq_kernels.h CoopQ<2, true>): Diagonal weights, whole trace products");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KT = 4 * NT;
    constexpr int M = MODD ? 1 : 0;
    typedef typename CoopQ<NT, DN>::Sh Sh;
    typedef typename CoopQ<NT, DN>::Nb Nb;
    typedef typename CoopQ<NT, DN>::Op Op;
    const int role = ((int)blockIdx.x >> 3) % NR;
    const int quad = 8 * ((int)blockIdx.x / (8 * NR)) + ((int)blockIdx.x & 7);
    const CqSetup<NT> s = cq_setup<NT>(a, quad >> 2, quad & 3);
    const int Nc = a.Ncoupled;
    const size_t trow = (size_t)s.slab * a.qps + s.qd;
    cq3_arrive(a);
    if (s.slab >= a.nslabs) return;
    if (!s.active) {
        if (role == NR - 1 && s.qd < a.qps)
            for (int k = threadIdx.x; k < a.nsteps_chunk * Nc * JQ_NTR; k += blockDim.x) a.traces[trow * a.nsteps_chunk * Nc * JQ_NTR + k] = 0.0;
        return;
    }
    const int lane_ = s.lane_, wave = s.wave;      // (s.chain: the two staging waves)
    double* tab = (double*)(smem + a.lds_tab_off);
    for (int i = threadIdx.x; i < 32 * NT; i += blockDim.x) tab[(i & ~15) + 4 * (i & 3) + ((i >> 2) & 3)] = a.tabs[i];   // [block][g][r]
    if (role == 2 && s.chain) return;      // (the trace workgroup has no staging waves)
    CoopQ<NT, DN> c;
    double* scratch = tab + 32 * NT + 2 * CoopQ<NT, DN>::PAR;      // [NT][64] workgroup sums / [ngroups][NT][64] trace hand-off (role 2)
    Cq3Hand<NT> hd;
    hd.init(a, (size_t)quad, lane_);
    const int nst = a.nsteps_chunk;
    if (!cq3_rendezvous(a, (int*)smem)) return;      // (the window ring at the start of the LDS is not in use yet)
    // TEST HOOK (option debug bit 16 / 32; results unchanged): the consumer roles / the state role start ~ 5 ms late -- on an idle GPU the roles
    // of a quad start together and a hand-off that is only safe then passes every test (round 5: the first version of the full-weights
    // hand-off was one; next to load processes it was not).  With bit 16 role 0 runs ahead as far as the protocol lets it before
    // anybody reads; with bit 32 everybody waits for role 0.
    if ((a.debug & 16) && role != 0)
        for (int i = 0; i < 1500; ++i) __builtin_amdgcn_s_sleep(127);
    if ((a.debug & 32) && role == 0)
        for (int i = 0; i < 1500; ++i) __builtin_amdgcn_s_sleep(127);
    // the three workgroups must share an L2: XCC_ID (hardware register 20, bits 3:0) of every role goes into the header, role 2 compares
    {
        const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));
        if (threadIdx.x == 0) __hip_atomic_store(hd.head + 32 + role, (unsigned long long)xcc + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    double* st = a.state + (size_t)s.slab * a.state_stride;
    const double wgt = a.colinfo[(size_t)s.slab * 32 + 16 + s.col];
    const size_t cslot = 16 * (lane_ >> 4) + s.col;

    if (role == 2) {
        // ---- trace products (adjoint_grad_calc!, src/evalobjgrad.jl:2581-2618), per control q -- the sums of k_backward_cq:
        //   group q < Nc:  rows 0, 1, 2 = t1, t4, t3     group Nc + j:  rows 0, 1 = t2, t5 of control 2 j, rows 2, 3 = of control 2 j + 1
        c.mt = wave, c.lane = lane_;
        c.xb = nullptr;
        WinRing& r = c.ring;      // (only the constant images are staged)
        r.smem = smem, r.wave = wave, r.lane = lane_, r.nwaves = NT;
        r.stride_b = (unsigned)(a.stride * 8), r.slot_bytes = 2 * r.stride_b, r.cbase = JQ_WIN_TPS * r.slot_bytes, r.pieces2 = 2 * a.pieces;
        r.dma((const char*)a.cimg, smem + r.cbase, 2 * a.Ncoupled * a.pieces);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int ntr = Nc * JQ_NTR, ngroups = Nc + (Nc + 1) / 2;
        // hand-off of the column partials, [step parity][ngroups][NT][64] (one barrier per step here: the waves that finish step k - 1
        // read while the others already write step k) and the scratch of cq_wg_sum: in the ring area of the window staging (unused)
        double* red = (double*)smem;
        const size_t redsz = (size_t)ngroups * NT * 64;
        scratch = red + 2 * redsz;
        auto finish_traces = [&](int k) {
            for (int g = wave; g < ngroups; g += NT) {
                const double* rr = red + (size_t)(k & 1) * redsz + (size_t)g * NT * 64 + lane_;
                double sum = rr[0];
#pragma unroll
                for (int w = 1; w < NT; ++w) sum += rr[w * 64];
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
        double carry[JQ_MAXNC];
        const bool slot0 = wave == 0 && ((lane_ >> 2) & 3) == 0;
#pragma unroll
        for (int q = 0; q < JQ_MAXNC; ++q) carry[q] = (q < Nc && slot0) ? st[(size_t)(JQ_STATE_ARRAYS * KT + q) * 64 + cslot] : 0.0;
        if (a.first_chunk) {
            // carry_q = tr(vr' Hsym_q lambdai) at t = T (see k_backward): vr(T) and -lambda_i(T) from the state file
            const double u0 = st[s.foff], nb0 = st[(size_t)3 * KT * 64 + s.foff];
            const Nb nn = c.nb_make(wave > 0 ? st[(size_t)3 * KT * 64 + s.foff - 256] : 0.0, wave + 1 < NT ? st[(size_t)3 * KT * 64 + s.foff + 256] : 0.0);
            const Sh sx = c.sh(nb0);
#pragma unroll
            for (int q = 0; q < JQ_MAXNC; ++q)
                if (q < Nc) carry[q] = -(u0 * c.template trace_mm<ORD>(c.ring.cimg(q), q, sx, nn));
        }
        // the six arrays of a step with the neighbouring blocks of X, nbn, Bq (zeros beyond the edge blocks)
        struct Rec {
            double u, v05, un, X, nbn, Bq;
            Nb nX, nN, nB;
        };
        auto fetch = [&](int n) {
            Rec e;
            e.u = hd.load(n, 0, wave), e.v05 = hd.load(n, 1, wave), e.un = hd.load(n, 2, wave);
            e.X = hd.load(n, 3, wave), e.nbn = hd.load(n, 4, wave), e.Bq = hd.load(n, 5, wave);
            const int wb = wave > 0 ? wave - 1 : wave, wa_ = wave + 1 < NT ? wave + 1 : wave;
            const double xb_ = hd.load(n, 3, wb), xa_ = hd.load(n, 3, wa_), nb_ = hd.load(n, 4, wb), na_ = hd.load(n, 4, wa_);
            const double bb_ = hd.load(n, 5, wb), ba_ = hd.load(n, 5, wa_);
            const bool lo = wave == 0, hi = wave + 1 == NT;      // (zeros beyond the edge blocks)
            e.nX = c.nb_make(lo ? 0.0 : xb_, hi ? 0.0 : xa_);
            e.nN = c.nb_make(lo ? 0.0 : nb_, hi ? 0.0 : na_);
            e.nB = c.nb_make(lo ? 0.0 : bb_, hi ? 0.0 : ba_);
            return e;
        };
        if (wave == 0) hd.wait(1, 1);
        __syncthreads();
        Rec cur = fetch(0);
        for (int n = 0; n < nst; ++n) {
            // (role 1 is asked for one step more than needed: the loads of step n + 1 travel while step n is worked on)
            if (wave == 0) hd.wait(1, (unsigned long long)(n + 2 < nst ? n + 2 : nst));
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // (cur has landed; my record of step n - 1 is written)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (wave == 0 && lane_ == 0) hd.publish(2, (unsigned long long)(n + 1));      // (everybody's loads of the steps <= n have landed)
            if (n > 0) finish_traces(n - 1);
            Rec nxt = cur;
            if (n + 1 < nst) nxt = fetch(n + 1);
            double* redw = red + (size_t)(n & 1) * redsz + (size_t)wave * 64 + lane_;
            const double v05w = cur.v05 * wgt, uw = cur.u * wgt, unw = cur.un * wgt;
            const Sh sX = c.sh(cur.X), sN = c.sh(cur.nbn), sB = c.sh(cur.Bq);
            double t2[JQ_MAXNC], t5[JQ_MAXNC];
#pragma unroll
            for (int q = 0; q < JQ_MAXNC; ++q) {
                t2[q] = 0.0, t5[q] = 0.0;
                if (q < Nc) {
                    t2[q] = v05w * c.template trace_mm<ORD>(c.ring.cimg(q), q, sX, cur.nX);
                    t5[q] = -(v05w * c.template trace_mm<ORD>(c.ring.cimg(Nc + q), q, sB, cur.nB));
                    const double Tq = c.template trace_mm<ORD>(c.ring.cimg(Nc + q), q, sX, cur.nX);
                    const double pq = -(cur.un * c.template trace_mm<ORD>(c.ring.cimg(q), q, sN, cur.nN));
                    const double t4 = (pq + carry[q]) * wgt;
                    carry[q] = pq;
                    redw[(size_t)q * NT * 64] = cq_part4(uw * Tq, unw * Tq, t4, 0.0);      // rows 0, 2, 1: t1, t3, t4
                }
            }
            redw[(size_t)Nc * NT * 64] = cq_part4(t2[0], t2[1], t5[0], t5[1]);
            if (Nc > 2) redw[(size_t)(Nc + 1) * NT * 64] = cq_part4(t2[2], t2[3], t5[2], t5[3]);
            cur = nxt;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (wave == 0 && lane_ == 0) hd.publish(2, (unsigned long long)nst);
        finish_traces(nst - 1);
        // the three roles ran on one XCD?  (headers: XCC + 1 of every role; 0: a role that never started cannot be the case here)
        if (threadIdx.x == 0) {
            const unsigned long long x0 = __hip_atomic_load(hd.head + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long x1 = __hip_atomic_load(hd.head + 33, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long x2 = __hip_atomic_load(hd.head + 34, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (x0 != x2 || x1 != x2) __hip_atomic_store(hd.gerr, 2ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int q = 0; q < JQ_MAXNC; ++q)
            if (q < Nc) {
                const double tot = cq_wg_sum(carry[q], scratch, wave, lane_, NT);
                if (wave == 0 && ((lane_ >> 2) & 3) == 0) st[(size_t)(JQ_STATE_ARRAYS * KT + q) * 64 + cslot] = tot;
            }
        return;
    }

    // ---- roles 0 and 1: a chain of publications as in k_forward_cq (NT block waves, two staging waves) --------------------------------
    c.setup(tab + 32 * NT, s.chain ? 0 : wave, lane_);
    c.ring.init(smem, a, wave + NT * s.chain, lane_, NT + 2);      // (barrier inside)
    if (s.chain) {      // staging waves
        c.ring.wave = wave, c.ring.nwaves = 2;
        const int nb = 4 + 2 * (a.m > 0 ? a.m : 0);
        if (WLR && role == 0) __builtin_amdgcn_s_barrier();      // (the block waves' partial dots with the state the chunk starts from)
        for (int n = 0; n < nst; ++n) {
            for (int k = 0; k < nb; ++k) __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (my pieces of the time points 2n+3, 2n+4)
            __builtin_amdgcn_s_barrier();
            c.ring.issue_next();
            c.ring.issue_next();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();      // (the block waves' last publication)
        if (NR == 2 && role == 1)          // (... and the barriers of their cq_wg_sum of the trace carries)
            for (int q = 0; q < Nc; ++q) {
                __syncthreads();
                __syncthreads();
            }
        return;
    }
    // (the tables are complete behind the barrier of ring.init)
    const double wdr = tab[16 * wave + s.g], wsr = tab[16 * NT + 16 * wave + s.g];
    const double cw = 0.5 * a.h * a.colinfo[(size_t)s.slab * 32 + s.col] * wsr;      // h/2 eps ws[row]
    if (role == 0) {
        // ---- state re-integration (src/evalobjgrad.jl:879): the state path of k_backward_cq; u, v05, un of step n -> ring
        double u = st[s.foff], v = st[(size_t)KT * 64 + s.foff];
        Op Kp05 = c.load(c.ring.template ks<0, 1>()), S0 = c.load(c.ring.template ks<1, 0>());
        CqW wq;
        // the sum of the NT partial registers of vector `vec` (behind the barrier that follows the put()s) -> block 0 of array `arr` of step n
        auto wsum = [&](int n, int arr, int vec) {
            const double* r = wq.wpart + (size_t)vec * NT * 64 + lane_;
            double d = r[0];
#pragma unroll
            for (int w = 1; w < NT; ++w) d += r[w * 64];
            hd.store(n, arr, 0, d);
        };
        if constexpr (WLR) {
            wq.init(a, smem, wave, lane_);
            wq.template put<NT>(0, wave, lane_, u);      // (vr at the start of the chunk)
            c.sync();
            if (wave == (NT > 1 ? 1 : 0)) {
                const double* r = wq.wpart + lane_;
                double d = r[0];
#pragma unroll
                for (int w = 1; w < NT; ++w) d += r[w * 64];
                __hip_atomic_store(hd.tail(), d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        auto step = [&](auto P0c, int n) {
            constexpr int P0 = decltype(P0c)::value;
            // (a slot is reused when the trace workgroup has read it -- it publishes k + 1 once the loads of the steps <= k have landed; NR = 2:
            //  when the adjoint workgroup has -- it publishes k once the loads of the steps <= k have landed in all its waves, so the slot of
            //  step n - 8 would be free at k = n - 8.  But k = 0 is also what the counter holds before that workgroup has started: the first
            //  version waited for n - 8 and, at step 8, overwrote the operands of step 0 of an adjoint workgroup that started late -- never
            //  on an idle GPU, where the roles start together; found with the late-start hook option debug=16, tests/test_gpu_round5.py (8).
            //  One step more, as for NR = 3: the state role is at most 7 steps ahead instead of 8.)
            if (wave == 0 && n >= JQ_CQ3_SLOTS) hd.wait(NR - 1, (unsigned long long)(n - JQ_CQ3_SLOTS + 1));
            double un, v05, vN;
            // x = u: A = c K05 u ; P = u + c S0 u
            c.template post<P0, 0>(u);
            const Op S05 = c.load(c.ring.template ks<1, 1>());
            double A, P;
            {
                const Sh sx = c.sh(u);
                A = c.own(0.0, Kp05, sx);
                P = c.own(u, S0, sx);
                if (a.use_shift) A = fma(cw, u, A);
                c.sync();
                const Nb nn = c.template nbs<P0, 0>();
                A = c.nbr(A, Kp05, nn);
                P = c.nbr(P, S0, nn);
            }
            // x = v: A = c (K05 u + S05 v) ; v05 = v + sum_j S^j A
            c.template post<P0 ^ 1, 0>(v);
            A = c.own(A, S05, c.sh(v));
            // (the stores of step n - 1 were issued more than an interval ago: acknowledged by now, and behind the barrier by every wave)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            c.sync();
            if (wave == 0 && lane_ == 0) hd.publish(0, (unsigned long long)n);
            hd.store(n, 0, wave, u);      // (behind the step's first barrier: wave 0 has seen the slot free)
            A = c.nbr(A, S05, c.template nbs<P0 ^ 1, 0>());
            Op Kn0, Kn1;
            v05 = c.template horner<P0, 0, MODD>(v + A, A, S05, a.m, [&] {
                Kn0 = c.load(c.ring.template ks<0, 0>());
                Kn1 = c.load(c.ring.template ks<0, 2>());
            });
            hd.store(n, 1, wave, v05);
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
                c.sync();
                const Nb nn = c.template nbs<P0 ^ M, 0>();
                vN = c.nbr(vN, S05, nn);
                un = c.nbr(un, Kn0, nn);
                A = c.nbr(A, Kn1, nn);
            }
            if constexpr (WLR)
                if (wave == 0) wsum(n, 6, 1);
            // x = un: A = c (S1 un - K1 v05)
            c.template post<P0 ^ M ^ 1, 0>(un);
            A = c.own(A, S1, c.sh(un));
            c.sync();
            A = c.nbr(A, S1, c.template nbs<P0 ^ M ^ 1, 0>());
            un = c.template horner<P0 ^ M, 0, MODD>(un + A, A, S1, a.m, [&] { Kp05 = c.load(c.ring.template ks<0, 1>()); });
            hd.store(n, 2, wave, un);
            // x = un: v(t_n) = v05 + c (K05 un + S05 v05)
            c.template post<P0, 0>(un);
            if constexpr (WLR) wq.template put<NT>(0, wave, lane_, un);
            v = c.own(vN, Kp05, c.sh(un));
            if (a.use_shift) v = fma(cw, un, v);
            c.sync();      // (the step's stores are published behind the second barrier of the next step: their latency is off the critical path)
            v = c.nbr(v, Kp05, c.template nbs<P0, 0>());
            if constexpr (WLR)
                if (wave == (NT > 1 ? 1 : 0)) wsum(n, 7, 0);
            // (the time points of the next step have landed; those of this step are dead)
            c.ring.advance();
            Kp05 = c.load(c.ring.template ks<0, 1>());
            S0 = c.load(c.ring.template ks<1, 0>());
            u = un;
        };
        int n = 0;
        for (; n + 1 < nst; n += 2) {
            step(std::integral_constant<int, 0>{}, n);
            step(std::integral_constant<int, 1>{}, n + 1);
        }
        if (n < nst) step(std::integral_constant<int, 0>{}, n);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wave == 0 && lane_ == 0) hd.publish(0, (unsigned long long)nst);
        st[s.foff] = u;
        st[(size_t)KT * 64 + s.foff] = v;
        return;
    }
    // ---- adjoint step! with forcing (src/StormerVerlet.jl:255-303): the adjoint path of k_backward_cq; u, v05, un of step n from the
    //      ring (fetched one step ahead), X, nbn, Bq of step n -> ring
    double mu = st[(size_t)2 * KT * 64 + s.foff], nb = st[(size_t)3 * KT * 64 + s.foff];
    Op Kp05 = c.load(c.ring.template ks<0, 1>()), S05 = c.load(c.ring.template ks<1, 1>());
    const double cfw = (a.forced ? 0.5 * a.h * a.tinv : 0.0) * wdr;      // forcing weight c tinv wd[row]; 0 for step_no_forcing!
    double hu = 0.0, hv = 0.0, hn = 0.0;      // u, v05, un of the step (fetched one step ahead)
    // full weights: the blocks' partial dots with v05, un of the step (fetched with them); c tinv W vr(t_n+1), W vi05, W vr(t_n) for my row
    // (W_i: complex weight matrices -- this kernel takes them, the state role being ahead; its coefficients are 0 for a real W)
    double pv = 0.0, pn = 0.0, Wu = 0.0, Wv = 0.0, Wn = 0.0, WvI = 0.0, WnI = 0.0, wcf = 0.0, wcfI = 0.0;
    if constexpr (WLR) {
        wcf = CqW::coef(a, wave, lane_, a.forced ? 0.5 * a.h * a.tinv : 0.0);
        wcfI = CqW::coef_imag(a, wave, lane_, a.forced ? 0.5 * a.h * a.tinv : 0.0);
    }
    auto wfetch = [&](int n) {
        if constexpr (WLR) pv = hd.load(n, 6, 0), pn = hd.load(n, 7, 0);
    };
    auto wapply = [&](double cA, double d) { return __builtin_amdgcn_mfma_f64_4x4x4f64(cA, d, 0.0, 0, 0, 0); };      // (the MFMA of CqW::apply)
    // NR = 2: the trace scalars of k_backward_cq (its adjoint waves' t1, t3, t4 and its state waves' t2, t5), handed over through red
    const int ntr = Nc * JQ_NTR, ngroups = Nc + (Nc + 1) / 2;
    double* red = scratch;                                      // [ngroups][NT][64]
    double* redw = red + (size_t)wave * 64 + lane_;
    double carry[JQ_MAXNC];
#pragma unroll
    for (int q = 0; q < JQ_MAXNC; ++q) carry[q] = 0.0;
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
    if constexpr (NR == 2) {
        const bool slot0 = wave == 0 && ((lane_ >> 2) & 3) == 0;      // the lanes that carry per-column partials between chunks
#pragma unroll
        for (int q = 0; q < JQ_MAXNC; ++q)
            if (q < Nc && slot0) carry[q] = st[(size_t)(JQ_STATE_ARRAYS * KT + q) * 64 + cslot];
        if (a.first_chunk) {
            // carry_q = tr(vr' Hsym_q lambdai) at t = T (see k_backward): vr(T) and -lambda_i(T) with its neighbouring blocks from the state file
            const double u0 = st[s.foff];
            const Nb nn = c.nb_make(wave > 0 ? st[(size_t)3 * KT * 64 + s.foff - 256] : 0.0, wave + 1 < NT ? st[(size_t)3 * KT * 64 + s.foff + 256] : 0.0);
            const Sh sx = c.sh(nb);
#pragma unroll
            for (int q = 0; q < JQ_MAXNC; ++q)
                if (q < Nc) carry[q] = -(u0 * c.template trace_mm<ORD>(c.ring.cimg(q), q, sx, nn));
        }
    }
    auto step = [&](auto P0c, int n) {
        constexpr int P0 = decltype(P0c)::value;
        if (wave == 0) hd.wait(0, (unsigned long long)(n + 2 < nst ? n + 2 : nst));
        // x = nb (-lambda_i): L = c K05 nb, Tn = c S05 nb (for the second half of the step)
        c.template post<P0, 1>(nb);
        const Op S0 = c.load(c.ring.template ks<1, 0>());
        double L, Tn;
        {
            const Sh sx = c.sh(nb);
            L = c.own(0.0, Kp05, sx);
            Tn = c.own(0.0, S05, sx);
            if (a.use_shift) L = fma(cw, nb, L);
            c.sync();      // (behind it everybody knows that role 0 has finished the steps <= n + 1)
            const Nb nn = c.template nbs<P0, 1>();
            L = c.nbr(L, Kp05, nn);
            Tn = c.nbr(Tn, S05, nn);
        }
        if constexpr (NR == 2)
            if (n > 0) finish_traces(n - 1);      // (behind the step's first barrier: everybody's hand-off of step n - 1 is in red)
        if (n == 0) {      // (first step of the chunk: latency exposed once)
            if constexpr (WLR) {
                pn = __hip_atomic_load(hd.tail(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                Wu = wapply(wcf, pn);
            }
            hu = hd.load(0, 0, wave), hv = hd.load(0, 1, wave), hn = hd.load(0, 2, wave);
            wfetch(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (hu, hv, hn of this step have landed; my stores of step n - 1 are acknowledged)
        const double u = hu, v05 = hv, un = hn;      // vr before the state step (:862), vi05, vr after it
        if constexpr (WLR) Wv = wapply(wcf, pv), Wn = wapply(wcf, pn), WvI = wapply(wcfI, pv), WnI = wapply(wcfI, pn);
        // x = mu: L = c (S0 mu - K05 li + hr0) ; X = mu + sum_j S0^j L
        c.template post<P0 ^ 1, 1>(mu);
        L = c.own(L, S0, c.sh(mu));
        L = fma(cfw, u, L);
        if constexpr (WLR) L += Wu;
        c.sync();
        if (wave == 0 && lane_ == 0) hd.publish(1, (unsigned long long)n);      // (everybody's stores of the steps < n are in the L2)
        if (n + 1 < nst) {
            hu = hd.load(n + 1, 0, wave), hv = hd.load(n + 1, 1, wave), hn = hd.load(n + 1, 2, wave);
            wfetch(n + 1);
        }
        L = c.nbr(L, S0, c.template nbs<P0 ^ 1, 1>());
        Op Kn0, Kn1, S1;
        const double X = c.template horner<P0, 1, MODD>(mu + L, L, S0, a.m, [&] {
            Kn0 = c.load(c.ring.template ks<0, 0>());
            Kn1 = c.load(c.ring.template ks<0, 2>());
            S1 = c.load(c.ring.template ks<1, 2>());
        });
        if constexpr (NR == 3) hd.store(n, 3, wave, X);
        // x = X: Lk = -c K0 X, Q = -c K1 X, SX = c S1 X ; NR = 2: Hanti_q X (tr1, tr3), Hsym_q X (tr2)
        c.template post<P0 ^ M, 1>(X);
        double Lk, Q, SX, Tq[JQ_MAXNC], t2[JQ_MAXNC];
        const double v05w = v05 * wgt;
        {
            const Sh sx = c.sh(X);
            Lk = c.own(0.0, Kn0, sx);
            Q = c.own(0.0, Kn1, sx);
            SX = c.own(0.0, S1, sx);
            if (a.use_shift) {
                Lk = fma(-cw, X, Lk);
                Q = fma(-cw, X, Q);
            }
            c.sync();
            const Nb nn = c.template nbs<P0 ^ M, 1>();
            Lk = c.nbr(Lk, Kn0, nn);
            Q = c.nbr(Q, Kn1, nn);
            SX = c.nbr(SX, S1, nn);
#pragma unroll
            for (int q = 0; q < JQ_MAXNC; ++q) {
                Tq[q] = 0.0, t2[q] = 0.0;
                if constexpr (NR == 2)
                    if (q < Nc) {
                        Tq[q] = c.template trace_mm<ORD>(c.ring.cimg(Nc + q), q, sx, nn);
                        t2[q] = v05w * c.template trace_mm<ORD>(c.ring.cimg(q), q, sx, nn);
                    }
            }
        }
        // Lk = -c l2 = -c (K0 X + S05 li + hi0) ; Q = -c (S05 (li + c l2) + K1 X + hi1)
        {
            double Pn = fma(-cfw, v05, Tn);
            if constexpr (WLR) Pn -= Wv;
            Lk += Pn;
            Q += Pn;
            if constexpr (WLR) Q += WnI;      // - c (hi1 - hi0) = + c W_i vr(t_n) / T
        }
        // x = Lk: Q += c S05 Lk ; nb_new = nb + Lk + sum_j S05^j Q
        c.template post<P0 ^ M ^ 1, 1>(Lk);
        Q = c.own(Q, S05, c.sh(Lk));
        c.sync();
        Q = c.nbr(Q, S05, c.template nbs<P0 ^ M ^ 1, 1>());
        const double nbn = c.template horner<P0 ^ M, 1, MODD>((nb + Lk) + Q, Q, S05, a.m);
        const double Bq = nb + nbn;      // -(li0 + li)
        if constexpr (NR == 3) {
            hd.store(n, 4, wave, nbn);
            hd.store(n, 5, wave, Bq);
        }
        // x = nb_new: lambda_r_new = X + c (S1 X - K05 li_new + hr1) ; NR = 2: Hsym_q li_new (tr4), Hanti_q (li0 + li) (tr5)
        c.template post<P0, 1>(nbn);
        if constexpr (NR == 2) c.template post<P0, 2>(Bq);      // (channel 2: the neighbouring blocks of -(li0 + li) for tr5)
        double G;
        {
            const Sh sx = c.sh(nbn);
            G = c.own(X, Kp05, sx);
            if (a.use_shift) G = fma(cw, nbn, G);
            G += SX;
            c.sync();      // (the step's stores are published behind the second barrier of the next step)
            const Nb nn = c.template nbs<P0, 1>();
            G = c.nbr(G, Kp05, nn);
            G = fma(cfw, un, G);
            if constexpr (WLR) {
                G += Wn + WvI;      // + c hr1 = c (W_r vr(t_n) + W_i vi05) / T
                Wu = Wn;            // (c W_r vr(t_n) / T: hr0 of the next step)
            }
            // (the time points of the next step have landed)
            c.ring.advance();
            Kp05 = c.load(c.ring.template ks<0, 1>());
            S05 = c.load(c.ring.template ks<1, 1>());
            if constexpr (NR == 2) {
                const double uw = u * wgt, unw = un * wgt;
                const Sh sb = c.sh(Bq);
                const Nb nx = c.template nbs<P0, 2>();
                double t5[JQ_MAXNC];
#pragma unroll
                for (int q = 0; q < JQ_MAXNC; ++q) {
                    t5[q] = 0.0;
                    if (q < Nc) {
                        const double pq = -(un * c.template trace_mm<ORD>(c.ring.cimg(q), q, sx, nn));
                        const double t4 = (pq + carry[q]) * wgt;
                        carry[q] = pq;
                        redw[(size_t)q * NT * 64] = cq_part4(uw * Tq[q], unw * Tq[q], t4, 0.0);      // rows 0, 2, 1: t1, t3, t4
                        t5[q] = -(v05w * c.template trace_mm<ORD>(c.ring.cimg(Nc + q), q, sb, nx));
                    }
                }
                redw[(size_t)Nc * NT * 64] = cq_part4(t2[0], t2[1], t5[0], t5[1]);
                if (Nc > 2) redw[(size_t)(Nc + 1) * NT * 64] = cq_part4(t2[2], t2[3], t5[2], t5[3]);
            }
        }
        mu = G;
        nb = nbn;
    };
    int n = 0;
    for (; n + 1 < nst; n += 2) {
        step(std::integral_constant<int, 0>{}, n);
        step(std::integral_constant<int, 1>{}, n + 1);
    }
    if (n < nst) step(std::integral_constant<int, 0>{}, n);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (wave == 0 && lane_ == 0) hd.publish(1, (unsigned long long)nst);
    st[(size_t)2 * KT * 64 + s.foff] = mu;
    st[(size_t)3 * KT * 64 + s.foff] = nb;
    if constexpr (NR == 2) {
        finish_traces(nst - 1);
        // the two roles ran on one XCD?
        if (threadIdx.x == 0) {
            const unsigned long long x0 = __hip_atomic_load(hd.head + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long x1 = __hip_atomic_load(hd.head + 33, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (x0 != x1) __hip_atomic_store(hd.gerr, 2ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int q = 0; q < JQ_MAXNC; ++q)
            if (q < Nc) {      // (the staging waves pass these barriers too)
                const double tot = cq_wg_sum(carry[q], scratch, wave, lane_, NT);
                if (wave == 0 && ((lane_ >> 2) & 3) == 0) st[(size_t)(JQ_STATE_ARRAYS * KT + q) * 64 + cslot] = tot;
            }
    }
}
