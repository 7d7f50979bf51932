// jq_host_select.h -- part of the host side of libjuqbox_hip.so (included by juqbox_hip.hip, ONE translation unit; not a stand-alone header):
// the kernel instantiations (compiled in their own translation units) and the tables that pick one.
// ---------------------------------------------------------------------------------------------
typedef void (*prop_kernel_t)(PropArgs);

// The (NT, BW) instantiations are compiled in their own translation units (jq_kernel_inst.hip).
#define JQ_FOR_EACH_INST(X)                                                                       \
    X(1, 0) X(2, 0) X(2, 1) X(3, 0) X(3, 1) X(3, 2) X(4, 0) X(4, 1) X(4, 2) X(4, 3) X(5, 0) X(5, 1) \
    X(5, 2) X(5, 4) X(6, 0) X(6, 1) X(6, 2) X(6, 5) X(2, 9) X(3, 9) X(4, 9) X(5, 9) X(6, 9) X(1, 8) X(2, 8) X(3, 8)       \
    X(4, 8) X(5, 8) X(6, 8) X(7, 8) X(8, 8)
#define JQ_MINW_OF(nt) (((nt) <= JQ_MINW_MAXNT) ? 2 : 1)
#define JQ_DECL(nt, bw)                                                                      \
    extern template __global__ void k_forward<nt, bw, JQ_MINW_OF(nt), false>(PropArgs);      \
    extern template __global__ void k_backward<nt, bw, JQ_MINW_OF(nt), false>(PropArgs);     \
    extern template __global__ void k_forward<nt, bw, JQ_MINW_OF(nt), true>(PropArgs);       \
    extern template __global__ void k_backward<nt, bw, JQ_MINW_OF(nt), true>(PropArgs);
JQ_FOR_EACH_INST(JQ_DECL)
#undef JQ_DECL

// slab kernels with the low-rank full leakage weights compiled in (the two without a cooperative sibling)
extern template __global__ void k_forward<1, 0, JQ_MINW_OF(1), false, true>(PropArgs);
extern template __global__ void k_backward<1, 0, JQ_MINW_OF(1), false, true>(PropArgs);
extern template __global__ void k_forward<6, 5, JQ_MINW_OF(6), false, true>(PropArgs);
extern template __global__ void k_backward<6, 5, JQ_MINW_OF(6), false, true>(PropArgs);
// ... and with the Jacobi solver (ABI 5: full weights are no longer tied to the Neumann solver)
extern template __global__ void k_forward<1, 0, JQ_MINW_OF(1), true, true>(PropArgs);
extern template __global__ void k_backward<1, 0, JQ_MINW_OF(1), true, true>(PropArgs);
extern template __global__ void k_forward<6, 5, JQ_MINW_OF(6), true, true>(PropArgs);
extern template __global__ void k_backward<6, 5, JQ_MINW_OF(6), true, true>(PropArgs);

static int select_kernels(jq_handle* h, prop_kernel_t* fwd, prop_kernel_t* bwd)
{
    const bool jac = (h->solver_id == 2);
    if (h->wrank > 0) {
        if (h->NT == 1 && h->BW == 0) {
            *fwd = jac ? k_forward<1, 0, JQ_MINW_OF(1), true, true> : k_forward<1, 0, JQ_MINW_OF(1), false, true>;
            *bwd = jac ? k_backward<1, 0, JQ_MINW_OF(1), true, true> : k_backward<1, 0, JQ_MINW_OF(1), false, true>;
            return JQ_OK;
        }
        if (h->NT == 6 && h->BW == 5) {
            *fwd = jac ? k_forward<6, 5, JQ_MINW_OF(6), true, true> : k_forward<6, 5, JQ_MINW_OF(6), false, true>;
            *bwd = jac ? k_backward<6, 5, JQ_MINW_OF(6), true, true> : k_backward<6, 5, JQ_MINW_OF(6), false, true>;
            return JQ_OK;
        }
        return fail(h, JQ_EUNSUPPORTED, "full leakage weights (jq_update_wmat): no kernels with the low-rank terms for this plan (row-lane kernels "
                                        "disabled, or cooperative kernels that do not fit the LDS)");
    }
#define JQ_PICK(nt, bw)                                                                                  \
    if (h->NT == nt && h->BW == bw) {                                                                    \
        *fwd = jac ? k_forward<nt, bw, JQ_MINW_OF(nt), true> : k_forward<nt, bw, JQ_MINW_OF(nt), false>; \
        *bwd = jac ? k_backward<nt, bw, JQ_MINW_OF(nt), true> : k_backward<nt, bw, JQ_MINW_OF(nt), false>; \
        return JQ_OK;                                                                                    \
    }
    JQ_FOR_EACH_INST(JQ_PICK)
#undef JQ_PICK
    return fail(h, JQ_EUNSUPPORTED, "unsupported Hilbert dimension / band width");
}

// quad-layout kernels of the JQ_BW_T4 structure (jq_kernels.h JQ_BW_T4Q): small batches, Neumann solver
#define JQ_DECLQ(nt)                                                            \
    extern template __global__ void k_forward<nt, JQ_BW_T4Q, 1, false>(PropArgs);    \
    extern template __global__ void k_backward<nt, JQ_BW_T4Q, 1, false>(PropArgs);   \
    extern template __global__ void k_forward<nt, JQ_BW_T4Q, 2, false>(PropArgs);    \
    extern template __global__ void k_backward<nt, JQ_BW_T4Q, 2, false>(PropArgs);   \
    extern template __global__ void k_forward<nt, JQ_BW_T4Q, 3, false>(PropArgs);    \
    extern template __global__ void k_backward<nt, JQ_BW_T4Q, 3, false>(PropArgs);   \
    extern template __global__ void k_backward<nt, JQ_BW_T4Q, 3, false, false, true>(PropArgs);   \
    extern template __global__ void k_backward<nt, JQ_BW_T4Q, 3, false, false, true, true>(PropArgs);
JQ_DECLQ(1) JQ_DECLQ(2) JQ_DECLQ(3) JQ_DECLQ(4) JQ_DECLQ(5) JQ_DECLQ(6) JQ_DECLQ(7) JQ_DECLQ(8)
#undef JQ_DECLQ
template <int NT, bool MODD, int NS, bool WLR = false, bool DN = false> __global__ void k_forward_cq(PropArgs);    // jq_cq_kernels.h (own translation units); NS: column quads per workgroup; WLR: full (real, low-rank) leakage weights
template <int NT, bool MODD, bool ORD, bool WLR = false, bool DN = false> __global__ void k_backward_cq(PropArgs);
template <int NT, bool MODD, bool ORD, int NR = 3, bool WLR = false, bool DN = false> __global__ void k_backward_cq3(PropArgs);    // jq_cq_split_kernels.h: three (NR = 2: two) workgroups per column quad
#define JQ_DECLCQ(nt)                                                      \
    extern template __global__ void k_forward_cq<nt, false, 1>(PropArgs);  \
    extern template __global__ void k_forward_cq<nt, false, 2>(PropArgs);  \
    extern template __global__ void k_backward_cq<nt, false, false>(PropArgs);    \
    extern template __global__ void k_backward_cq<nt, false, true>(PropArgs);     \
    extern template __global__ void k_forward_cq<nt, true, 1>(PropArgs);   \
    extern template __global__ void k_forward_cq<nt, true, 2>(PropArgs);   \
    extern template __global__ void k_backward_cq<nt, true, false>(PropArgs);     \
    extern template __global__ void k_backward_cq<nt, true, true>(PropArgs);      \
    extern template __global__ void k_forward_cq<nt, false, 1, true>(PropArgs);          \
    extern template __global__ void k_forward_cq<nt, true, 1, true>(PropArgs);           \
    extern template __global__ void k_backward_cq<nt, false, false, true>(PropArgs);     \
    extern template __global__ void k_backward_cq<nt, false, true, true>(PropArgs);      \
    extern template __global__ void k_backward_cq<nt, true, false, true>(PropArgs);      \
    extern template __global__ void k_backward_cq<nt, true, true, true>(PropArgs);       \
    extern template __global__ void k_backward_cq3<nt, false, false, 3, true>(PropArgs);   \
    extern template __global__ void k_backward_cq3<nt, false, true, 3, true>(PropArgs);    \
    extern template __global__ void k_backward_cq3<nt, true, false, 3, true>(PropArgs);    \
    extern template __global__ void k_backward_cq3<nt, true, true, 3, true>(PropArgs);     \
    extern template __global__ void k_backward_cq3<nt, false, false, 2, true>(PropArgs);   \
    extern template __global__ void k_backward_cq3<nt, false, true, 2, true>(PropArgs);    \
    extern template __global__ void k_backward_cq3<nt, true, false, 2, true>(PropArgs);    \
    extern template __global__ void k_backward_cq3<nt, true, true, 2, true>(PropArgs);     \
    extern template __global__ void k_backward_cq3<nt, false, false>(PropArgs);   \
    extern template __global__ void k_backward_cq3<nt, false, true>(PropArgs);    \
    extern template __global__ void k_backward_cq3<nt, true, false>(PropArgs);    \
    extern template __global__ void k_backward_cq3<nt, true, true>(PropArgs);     \
    extern template __global__ void k_backward_cq3<nt, false, false, 2>(PropArgs);   \
    extern template __global__ void k_backward_cq3<nt, false, true, 2>(PropArgs);    \
    extern template __global__ void k_backward_cq3<nt, true, false, 2>(PropArgs);    \
    extern template __global__ void k_backward_cq3<nt, true, true, 2>(PropArgs);
JQ_DECLCQ(1) JQ_DECLCQ(2) JQ_DECLCQ(3) JQ_DECLCQ(4) JQ_DECLCQ(5) JQ_DECLCQ(6) JQ_DECLCQ(7)
#undef JQ_DECLCQ
extern template __global__ void k_forward_cq<2, false, 1, false, true>(PropArgs);      // the dense policy (17 .. 32 levels without the structure)
extern template __global__ void k_forward_cq<2, true, 1, false, true>(PropArgs);
extern template __global__ void k_backward_cq<2, false, false, false, true>(PropArgs);
extern template __global__ void k_backward_cq<2, true, false, false, true>(PropArgs);
extern template __global__ void k_backward_cq3<2, false, false, 3, false, true>(PropArgs);
extern template __global__ void k_backward_cq3<2, true, false, 3, false, true>(PropArgs);
extern template __global__ void k_backward_cq3<2, false, false, 2, false, true>(PropArgs);
extern template __global__ void k_backward_cq3<2, true, false, 2, false, true>(PropArgs);
// (instantiated for even and odd numbers of Neumann terms: the parities of the LDS exchange are compile-time constants)
// fwd2: the forward kernel with two column quads per workgroup (grid = 2 * nslabs); bwd3: the backward sweep on three workgroups per
// quad (k_backward_cq3)
// control q acts on subsystem q only (the usual Juqbox set-up: Hsym_ops = [a + a', b + b', c + c']): its trace products need one part of
// the product each
static bool cq_ord(const jq_handle* h)
{
    bool ord = h->Nc <= 3 && !h->opt.on(O_CQ_GENERIC_TRACES);      // (more than JQ_MAXNC controls: generic traces per control group)
    for (int q = 0; q < h->Nc && ord; ++q) ord = (h->bw_trace[q] == (1 << q));
    return ord;
}
static int select_cq_kernels(jq_handle* h, bool fwd2, int bwd_nr, bool wlr, bool dense, prop_kernel_t* fwd, prop_kernel_t* bwd)      // bwd_nr: workgroups per quad in the backward sweep (0 / 1: one); wlr: full (real, low-rank) leakage weights; dense: no structure, NT = 2
{
    const bool bwd3 = bwd_nr == 3, bwd2 = bwd_nr == 2;
    const bool modd = (h->m > 0 ? h->m : 0) & 1;
    if (dense) {
        if (h->NT != 2 || wlr || fwd2) return fail(h, JQ_EHIP, "internal error: dense cooperative-quad kernels selected for a plan they do not exist for");
        *fwd = modd ? k_forward_cq<2, true, 1, false, true> : k_forward_cq<2, false, 1, false, true>;
        *bwd = bwd3 ? (modd ? k_backward_cq3<2, true, false, 3, false, true> : k_backward_cq3<2, false, false, 3, false, true>)
             : bwd2 ? (modd ? k_backward_cq3<2, true, false, 2, false, true> : k_backward_cq3<2, false, false, 2, false, true>)
                    : (modd ? k_backward_cq<2, true, false, false, true> : k_backward_cq<2, false, false, false, true>);
        return JQ_OK;
    }
    const bool ord = cq_ord(h);
#define JQ_PICKCQ(nt)                                                              \
    if (h->NT == nt && wlr) {                                                      \
        *fwd = modd ? k_forward_cq<nt, true, 1, true> : k_forward_cq<nt, false, 1, true>;                        \
        *bwd = bwd3 ? (modd ? (ord ? k_backward_cq3<nt, true, true, 3, true> : k_backward_cq3<nt, true, false, 3, true>)          \
                            : (ord ? k_backward_cq3<nt, false, true, 3, true> : k_backward_cq3<nt, false, false, 3, true>))       \
             : bwd2 ? (modd ? (ord ? k_backward_cq3<nt, true, true, 2, true> : k_backward_cq3<nt, true, false, 2, true>)          \
                            : (ord ? k_backward_cq3<nt, false, true, 2, true> : k_backward_cq3<nt, false, false, 2, true>))       \
                    : modd ? (ord ? k_backward_cq<nt, true, true, true> : k_backward_cq<nt, true, false, true>)         \
                           : (ord ? k_backward_cq<nt, false, true, true> : k_backward_cq<nt, false, false, true>);      \
        return JQ_OK;                                                              \
    }                                                                              \
    if (h->NT == nt) {                                                             \
        *fwd = fwd2 ? (modd ? k_forward_cq<nt, true, 2> : k_forward_cq<nt, false, 2>) : (modd ? k_forward_cq<nt, true, 1> : k_forward_cq<nt, false, 1>);            \
        *bwd = bwd3 ? (modd ? (ord ? k_backward_cq3<nt, true, true> : k_backward_cq3<nt, true, false>)          \
                            : (ord ? k_backward_cq3<nt, false, true> : k_backward_cq3<nt, false, false>))       \
             : bwd2 ? (modd ? (ord ? k_backward_cq3<nt, true, true, 2> : k_backward_cq3<nt, true, false, 2>)    \
                            : (ord ? k_backward_cq3<nt, false, true, 2> : k_backward_cq3<nt, false, false, 2>)) \
                    : modd ? (ord ? k_backward_cq<nt, true, true> : k_backward_cq<nt, true, false>)          \
                           : (ord ? k_backward_cq<nt, false, true> : k_backward_cq<nt, false, false>);       \
        return JQ_OK;                                                              \
    }
    JQ_PICKCQ(1) JQ_PICKCQ(2) JQ_PICKCQ(3) JQ_PICKCQ(4) JQ_PICKCQ(5) JQ_PICKCQ(6) JQ_PICKCQ(7)
#undef JQ_PICKCQ
    return fail(h, JQ_EUNSUPPORTED, "unsupported Hilbert dimension");
}
template <int NT, int SPW> __global__ void k_forward_quad_imr(PropArgs);      // jq_quad_imr_kernels.h (own translation units)
template <int NT, int SPW> __global__ void k_backward_quad_imr(PropArgs);
#define JQ_DECLQI(nt)                                                         \
    extern template __global__ void k_forward_quad_imr<nt, 1>(PropArgs);      \
    extern template __global__ void k_backward_quad_imr<nt, 1>(PropArgs);
JQ_DECLQI(1) JQ_DECLQI(2) JQ_DECLQI(3) JQ_DECLQI(4) JQ_DECLQI(5) JQ_DECLQI(6) JQ_DECLQI(7) JQ_DECLQI(8)
#undef JQ_DECLQI
static int select_quad_imr_kernels(jq_handle* h, prop_kernel_t* fwd, prop_kernel_t* bwd)
{
#define JQ_PICKQI(nt)                                 \
    if (h->NT == nt) {                                \
        *fwd = k_forward_quad_imr<nt, 1>;             \
        *bwd = k_backward_quad_imr<nt, 1>;            \
        return JQ_OK;                                 \
    }
    JQ_PICKQI(1) JQ_PICKQI(2) JQ_PICKQI(3) JQ_PICKQI(4) JQ_PICKQI(5) JQ_PICKQI(6) JQ_PICKQI(7) JQ_PICKQI(8)
#undef JQ_PICKQI
    return fail(h, JQ_EUNSUPPORTED, "unsupported Hilbert dimension");
}
template <int NT, bool DN = false> __global__ void k_forward_cq_imr(PropArgs);      // jq_cq_imr_kernels.h (own translation units)
template <int NT, bool DN = false> __global__ void k_backward_cq_imr(PropArgs);
template <int NT> __global__ void k_backward_cq_imr2(PropArgs);    // (state and adjoint chain on two sets of waves, NT <= 6)
#define JQ_DECLCI(nt)                                                    \
    extern template __global__ void k_forward_cq_imr<nt>(PropArgs);      \
    extern template __global__ void k_backward_cq_imr<nt>(PropArgs);
JQ_DECLCI(1) JQ_DECLCI(2) JQ_DECLCI(3) JQ_DECLCI(4) JQ_DECLCI(5) JQ_DECLCI(6) JQ_DECLCI(7)
#undef JQ_DECLCI
template <int NT, bool DN = false> __global__ void k_backward_cq_imr3(PropArgs);    // (three workgroups per evaluation, as k_backward_cq3)
#define JQ_DECLCI(nt) extern template __global__ void k_backward_cq_imr3<nt>(PropArgs);
JQ_DECLCI(1) JQ_DECLCI(2) JQ_DECLCI(3) JQ_DECLCI(4) JQ_DECLCI(5) JQ_DECLCI(6) JQ_DECLCI(7)
#undef JQ_DECLCI
#define JQ_DECLCI(nt) extern template __global__ void k_backward_cq_imr2<nt>(PropArgs);
JQ_DECLCI(1) JQ_DECLCI(2) JQ_DECLCI(3) JQ_DECLCI(4) JQ_DECLCI(5) JQ_DECLCI(6)
#undef JQ_DECLCI
// dynamic LDS of k_backward_cq_imr2: staging + tables + two exchange images (one per set of waves) + the decisions
static size_t cq_imr2_lds(const jq_handle* h, size_t lds_stage) { return lds_stage + (size_t)32 * h->NT * 8 + (size_t)12 * (h->NT + 2) * 64 * 8 + 64; }
// two: the backward sweep with the state and the adjoint chain on two sets of waves (NT <= 6, LDS permitting; option imr_cq2=0: the
// one-set kernel of round 3)
extern template __global__ void k_forward_cq_imr<2, true>(PropArgs);      // the dense policy (17 .. 32 levels without the structure)
extern template __global__ void k_backward_cq_imr<2, true>(PropArgs);
extern template __global__ void k_backward_cq_imr3<2, true>(PropArgs);
static int select_cq_imr_kernels(jq_handle* h, bool two, bool three, bool dense, prop_kernel_t* fwd, prop_kernel_t* bwd)
{
    if (dense) {
        if (h->NT != 2 || two) return fail(h, JQ_EHIP, "internal error: dense implicit-midpoint cooperative-quad kernels selected for a plan they do not exist for");
        *fwd = k_forward_cq_imr<2, true>;
        *bwd = three ? k_backward_cq_imr3<2, true> : k_backward_cq_imr<2, true>;
        return JQ_OK;
    }
#define JQ_PICKCI(nt)                            \
    if (h->NT == nt && three) {                  \
        *fwd = k_forward_cq_imr<nt>;             \
        *bwd = k_backward_cq_imr3<nt>;           \
        return JQ_OK;                            \
    }
    JQ_PICKCI(1) JQ_PICKCI(2) JQ_PICKCI(3) JQ_PICKCI(4) JQ_PICKCI(5) JQ_PICKCI(6) JQ_PICKCI(7)
#undef JQ_PICKCI
#define JQ_PICKCI(nt)                            \
    if (h->NT == nt && two) {                    \
        *fwd = k_forward_cq_imr<nt>;             \
        *bwd = k_backward_cq_imr2<nt>;           \
        return JQ_OK;                            \
    }
    JQ_PICKCI(1) JQ_PICKCI(2) JQ_PICKCI(3) JQ_PICKCI(4) JQ_PICKCI(5) JQ_PICKCI(6)
#undef JQ_PICKCI
#define JQ_PICKCI(nt)                            \
    if (h->NT == nt) {                           \
        *fwd = k_forward_cq_imr<nt>;             \
        *bwd = k_backward_cq_imr<nt>;            \
        return JQ_OK;                            \
    }
    JQ_PICKCI(1) JQ_PICKCI(2) JQ_PICKCI(3) JQ_PICKCI(4) JQ_PICKCI(5) JQ_PICKCI(6) JQ_PICKCI(7)
#undef JQ_PICKCI
    return fail(h, JQ_EUNSUPPORTED, "unsupported Hilbert dimension");
}
// (spw: slabs per workgroup = waves per SIMD: workgroups of 4 spw waves)
static int select_quad_kernels(jq_handle* h, int spw, prop_kernel_t* fwd, prop_kernel_t* bwd)
{
    // one ensemble sample per wave (four columns of a slab): N a multiple of 4 (N <= 16 divides the slab into whole samples), or N > 16.
    // Only the twelve-wave BACKWARD kernel has a UNI variant: folding the shift into the MFMA's A operand adds a dependent FMA in front
    // of every MFMA, which three waves per SIMD hide (- 1.2 %) and one or two do not (measured: forward sweep + 1.2 %, one / two slabs
    // per workgroup + 1.6 ... 3.4 %)
    const bool uni = (h->N % 4 == 0 || h->parts > 1) && !h->opt.on(O_NO_UNI);
    // ... and its ORD variant when control q acts on subsystem q only (like the cooperative-quad kernels, select_cq_kernels)
    bool ord = uni && h->Nc >= 2 && h->Nc <= 3 && !h->opt.on(O_NO_ORD);
    for (int q = 0; q < h->Nc && ord; ++q) ord = (h->bw_trace[q] == (1 << q));
#define JQ_PICKQ(nt)                                                                                                                             \
    if (h->NT == nt) {                                                                                                                           \
        *fwd = spw == 3 ? k_forward<nt, JQ_BW_T4Q, 3, false> : spw == 2 ? k_forward<nt, JQ_BW_T4Q, 2, false> : k_forward<nt, JQ_BW_T4Q, 1, false>;     \
        *bwd = spw == 3 ? (ord ? k_backward<nt, JQ_BW_T4Q, 3, false, false, true, true>                                                       \
                                : uni ? k_backward<nt, JQ_BW_T4Q, 3, false, false, true> : k_backward<nt, JQ_BW_T4Q, 3, false>)                 \
                        : spw == 2 ? k_backward<nt, JQ_BW_T4Q, 2, false> : k_backward<nt, JQ_BW_T4Q, 1, false>;  \
        return JQ_OK;                                                                                                                            \
    }
    JQ_PICKQ(1) JQ_PICKQ(2) JQ_PICKQ(3) JQ_PICKQ(4) JQ_PICKQ(5) JQ_PICKQ(6) JQ_PICKQ(7) JQ_PICKQ(8)
#undef JQ_PICKQ
    return fail(h, JQ_EUNSUPPORTED, "unsupported Hilbert dimension");
}

// ... backward sweep with the state and the adjoint chain of a column quad on two waves, one time step apart (jq_quad_split_kernels.h):
// mid-size ensembles -- at most one column quad per SIMD (qw = 4 quads per workgroup: two waves per SIMD) or per two SIMDs (qw = 2)
template <int NT, bool ORD, int QW, bool RIDE = false> __global__ void k_backward_qsplit(PropArgs);
#define JQ_DECLQS(nt)                                                            \
    extern template __global__ void k_backward_qsplit<nt, false, 4>(PropArgs);   \
    extern template __global__ void k_backward_qsplit<nt, true, 4>(PropArgs);    \
    extern template __global__ void k_backward_qsplit<nt, false, 2>(PropArgs);   \
    extern template __global__ void k_backward_qsplit<nt, true, 2>(PropArgs);    \
    extern template __global__ void k_backward_qsplit<nt, true, 4, true>(PropArgs);    \
    extern template __global__ void k_backward_qsplit<nt, true, 2, true>(PropArgs);
JQ_DECLQS(1) JQ_DECLQS(2) JQ_DECLQS(3) JQ_DECLQS(4) JQ_DECLQS(5) JQ_DECLQS(6)
#undef JQ_DECLQS
static size_t qsplit_lds(const jq_handle* h, int qw)      // ring of JQ_QS_TPS time points + constant images, tables, trace records
{
    return (size_t)(2 * JQ_QS_TPS + 2 * h->NcK) * h->mat_elems * 8 + (size_t)32 * h->NT * 8 + (size_t)2 * qw * 8 * h->NcK * 8;
}
static int select_qsplit_kernel(jq_handle* h, int qw, prop_kernel_t* bwd)
{
    // control q acts on subsystem q only (like select_quad_kernels / select_cq_kernels): compile-time trace modes
    bool ord = h->Nc >= 2 && h->Nc <= 3 && !h->opt.on(O_NO_ORD);
    for (int q = 0; q < h->Nc && ord; ++q) ord = (h->bw_trace[q] == (1 << q));
    // ... and with exactly three of them every trace product rides along in a pass of the adjoint step (RIDE; option qs_ride=0: separate passes)
    // -- where the adjoint wave is alone on its SIMD (qw = 2: - 6 %); with two waves per SIMD and the adjoint wave first in the issue
    // arbitration the rides buy nothing (248.9 ms without, 250.0 with): qw = 4 keeps the separate passes (bit-identical to the one-wave
    // kernel); option qs_ride=1 forces the rides there too
    const bool ride_set = h->opt.has(O_QS_RIDE);
    const long long ride_v = h->opt.get(O_QS_RIDE);
    const bool ride = ord && h->Nc == 3 && !(ride_set && ride_v == 0) && (qw == 2 || (ride_set && ride_v == 1));
#define JQ_PICKQS(nt)                                                                                         \
    if (h->NT == nt) {                                                                                        \
        *bwd = qw == 4 ? (ride ? k_backward_qsplit<nt, true, 4, true> : ord ? k_backward_qsplit<nt, true, 4> : k_backward_qsplit<nt, false, 4>)             \
                       : (ride ? k_backward_qsplit<nt, true, 2, true> : ord ? k_backward_qsplit<nt, true, 2> : k_backward_qsplit<nt, false, 2>);            \
        return JQ_OK;                                                                                         \
    }
    JQ_PICKQS(1) JQ_PICKQS(2) JQ_PICKQS(3) JQ_PICKQS(4) JQ_PICKQS(5) JQ_PICKQS(6)
#undef JQ_PICKQS
    return fail(h, JQ_EUNSUPPORTED, "unsupported Hilbert dimension");
}

// ... with the low-rank full leakage weights compiled in (jq_update_wmat; one slab per workgroup)
#define JQ_DECLQW(nt)                                                                  \
    extern template __global__ void k_forward<nt, JQ_BW_T4Q, 1, false, true>(PropArgs);    \
    extern template __global__ void k_backward<nt, JQ_BW_T4Q, 1, false, true>(PropArgs);
JQ_DECLQW(1) JQ_DECLQW(2) JQ_DECLQW(3) JQ_DECLQW(4) JQ_DECLQW(5) JQ_DECLQW(6) JQ_DECLQW(7) JQ_DECLQW(8)
#undef JQ_DECLQW
static int select_quad_w_kernels(jq_handle* h, prop_kernel_t* fwd, prop_kernel_t* bwd)
{
#define JQ_PICKQW(nt)                                          \
    if (h->NT == nt) {                                         \
        *fwd = k_forward<nt, JQ_BW_T4Q, 1, false, true>;       \
        *bwd = k_backward<nt, JQ_BW_T4Q, 1, false, true>;      \
        return JQ_OK;                                          \
    }
    JQ_PICKQW(1) JQ_PICKQW(2) JQ_PICKQW(3) JQ_PICKQW(4) JQ_PICKQW(5) JQ_PICKQW(6) JQ_PICKQW(7) JQ_PICKQW(8)
#undef JQ_PICKQW
    return fail(h, JQ_EUNSUPPORTED, "unsupported Hilbert dimension");
}

#define JQ_DECLC(nt, bw)                                                   \
    extern template __global__ void k_forward_coop<nt, bw>(PropArgs);       \
    extern template __global__ void k_backward_coop<nt, bw>(PropArgs);
#define JQ_FOR_EACH_COOP(X)                                                                               \
    X(2, 0) X(2, 1) X(3, 0) X(3, 1) X(3, 2) X(4, 0) X(4, 1) X(4, 2) X(4, 3) X(5, 0) X(5, 1) X(5, 2) X(5, 4) \
    X(6, 0) X(6, 1) X(6, 2) X(6, 5) X(2, 9) X(3, 9) X(4, 9) X(5, 9) X(6, 9)
JQ_FOR_EACH_COOP(JQ_DECLC)
// Ntot > 96 (NT = 7 .. 16): block band 1, 2 or dense (band code 15 for every NT: a full window); operators read from HBM
// (jq_coop_kernels.h OpCursor)
#define JQ_FOR_EACH_BIG(X)                                                                                   \
    X(7, 1) X(7, 2) X(7, 15) X(8, 1) X(8, 2) X(8, 15) X(9, 1) X(9, 2) X(9, 15) X(10, 1) X(10, 2) X(10, 15)  \
    X(11, 1) X(11, 2) X(11, 15) X(12, 1) X(12, 2) X(12, 15) X(13, 1) X(13, 2) X(13, 15) X(14, 1) X(14, 2)   \
    X(14, 15) X(15, 1) X(15, 2) X(15, 15) X(16, 1) X(16, 2) X(16, 15)
JQ_FOR_EACH_BIG(JQ_DECLC)
#undef JQ_DECLC

static int select_coop_kernels(jq_handle* h, prop_kernel_t* fwd, prop_kernel_t* bwd)
{
    if (h->huge) {
        *fwd = k_forward_huge, *bwd = k_backward_huge;
        return JQ_OK;
    }
#define JQ_PICKC(nt, bw)                      \
    if (h->NT == nt && h->BWc == bw) {        \
        *fwd = k_forward_coop<nt, bw>;        \
        *bwd = k_backward_coop<nt, bw>;       \
        return JQ_OK;                         \
    }
    JQ_FOR_EACH_COOP(JQ_PICKC)
    JQ_FOR_EACH_BIG(JQ_PICKC)
#undef JQ_PICKC
    return fail(h, JQ_EUNSUPPORTED, "no cooperative kernel for this Hilbert dimension / band width");
}

// lane kernels (one lane per column), NP = padded Hilbert dimension
typedef void (*lane_init_t)(double*, long long, const double*, int, long long);
typedef void (*lane_term_t)(double*, long long, const double*, const double*, int, int, double, double*);
#define JQ_FOR_EACH_LANE(X) X(2) X(4) X(6) X(8)
#define JQ_DECLL(np)                                                                                  \
    extern template __global__ void k_forward_lane<np>(PropArgs);                                     \
    extern template __global__ void k_backward_lane<np>(PropArgs);                                    \
    extern template __global__ void k_init_state_lane<np>(double*, long long, const double*, int, long long); \
    extern template __global__ void k_terminal_lane<np>(double*, long long, const double*, const double*, int, int, double, double*);
JQ_FOR_EACH_LANE(JQ_DECLL)
#undef JQ_DECLL

static int select_lane_kernels(jq_handle* h, prop_kernel_t* fwd, prop_kernel_t* bwd, lane_init_t* init, lane_term_t* term)
{
#define JQ_PICKL(np)                     \
    if (h->lane_np == np) {              \
        *fwd = k_forward_lane<np>;       \
        *bwd = k_backward_lane<np>;      \
        *init = k_init_state_lane<np>;   \
        *term = k_terminal_lane<np>;     \
        return JQ_OK;                    \
    }
    JQ_FOR_EACH_LANE(JQ_PICKL)
#undef JQ_PICKL
    return fail(h, JQ_EUNSUPPORTED, "no lane kernel for this Hilbert dimension");
}

// row-lane kernels (one lane per (row, column)), NPJ = padded row length
#define JQ_FOR_EACH_ROWLANE(X) X(2) X(4) X(6) X(8) X(12) X(16)
#define JQ_DECLR(npj)                                                     \
    extern template __global__ void k_forward_rowlane<npj>(PropArgs);     \
    extern template __global__ void k_backward_rowlane<npj>(PropArgs);    \
    extern template __global__ void k_forward_rowlane<npj, false, true>(PropArgs);     \
    extern template __global__ void k_forward_rowlane<npj, true, true>(PropArgs);     \
    extern template __global__ void k_backward_rowlane<npj, true>(PropArgs);    \
    extern template __global__ void k_backward_rowlane2<npj>(PropArgs);   \
    extern template __global__ void k_backward_rowlane3<npj>(PropArgs);
JQ_FOR_EACH_ROWLANE(JQ_DECLR)
#undef JQ_DECLR

static int select_rowlane_kernels(jq_handle* h, int split, bool hist, prop_kernel_t* fwd, prop_kernel_t* bwd)      // split: waves of the backward sweep (1, 2, 3)
{
#define JQ_PICKR(npj)                                                          \
    if (h->rl_npj == npj) {                                                    \
        *fwd = h->wrank > 0 ? k_forward_rowlane<npj, true, true> : hist ? k_forward_rowlane<npj, false, true> : k_forward_rowlane<npj>;     \
        *bwd = h->wrank > 0 ? k_backward_rowlane<npj, true> : split == 3 ? k_backward_rowlane3<npj> : split == 2 ? k_backward_rowlane2<npj> : k_backward_rowlane<npj>;     \
        return JQ_OK;                                                          \
    }
    JQ_FOR_EACH_ROWLANE(JQ_PICKR)
#undef JQ_PICKR
    return fail(h, JQ_EUNSUPPORTED, "no row-lane kernel for this Hilbert dimension");
}

#define JQ_DECLM(npj)                                                        \
    extern template __global__ void k_forward_rowlane_imr<npj>(PropArgs);    \
    extern template __global__ void k_backward_rowlane_imr<npj>(PropArgs);   \
    extern template __global__ void k_backward_rowlane_imr2<npj>(PropArgs);
JQ_FOR_EACH_ROWLANE(JQ_DECLM)
#undef JQ_DECLM

static int select_rowlane_imr_kernels(jq_handle* h, bool split, prop_kernel_t* fwd, prop_kernel_t* bwd)
{
#define JQ_PICKM(npj)                                                                  \
    if (h->rl_npj == npj) {                                                            \
        *fwd = k_forward_rowlane_imr<npj>;                                             \
        *bwd = split ? k_backward_rowlane_imr2<npj> : k_backward_rowlane_imr<npj>;     \
        return JQ_OK;                                                                  \
    }
    JQ_FOR_EACH_ROWLANE(JQ_PICKM)
#undef JQ_PICKM
    return fail(h, JQ_EUNSUPPORTED, "no implicit-midpoint kernel for this Hilbert dimension");
}

template <int NT, int BW, bool HBM> __global__ void k_forward_coop_imr(PropArgs);      // jq_coop_imr_kernels.h
template <int NT, int BW, bool HBM> __global__ void k_backward_coop_imr(PropArgs);
#define JQ_DECLCI(nt, bw)                                                                 \
    extern template __global__ void k_forward_coop_imr<nt, bw, (nt > 6)>(PropArgs);       \
    extern template __global__ void k_backward_coop_imr<nt, bw, (nt > 6)>(PropArgs);
extern template __global__ void k_forward_coop_imr<6, 5, true>(PropArgs);      // (dense 96 x 96: images from HBM / L2)
extern template __global__ void k_backward_coop_imr<6, 5, true>(PropArgs);
JQ_FOR_EACH_COOP(JQ_DECLCI)
JQ_FOR_EACH_BIG(JQ_DECLCI)      // (Ntot > 96: operators read from HBM / L2 per product)
JQ_DECLCI(1, 0)      // (Ntot <= 16 with N > 4: one wave per slab, the evaluation's columns in one wave)
#undef JQ_DECLCI

template <int NT, int BW, bool HBM> __global__ void k_forward_coop_imr_parts(PropArgs);      // N > 16: one workgroup per evaluation
template <int NT, int BW, bool HBM> __global__ void k_backward_coop_imr_parts(PropArgs);
#define JQ_DECLCIP(nt, bw)                                                                      \
    extern template __global__ void k_forward_coop_imr_parts<nt, bw, (nt > 6)>(PropArgs);       \
    extern template __global__ void k_backward_coop_imr_parts<nt, bw, (nt > 6)>(PropArgs);
extern template __global__ void k_forward_coop_imr_parts<6, 5, true>(PropArgs);
extern template __global__ void k_backward_coop_imr_parts<6, 5, true>(PropArgs);
JQ_FOR_EACH_COOP(JQ_DECLCIP)
JQ_FOR_EACH_BIG(JQ_DECLCIP)
#undef JQ_DECLCIP
static int select_coop_imr_parts_kernels(jq_handle* h, bool hbm, prop_kernel_t* fwd, prop_kernel_t* bwd)
{
    if (hbm) {
        *fwd = k_forward_coop_imr_parts<6, 5, true>;
        *bwd = k_backward_coop_imr_parts<6, 5, true>;
        return JQ_OK;
    }
#define JQ_PICKCIP(nt, bw)                                      \
    if (h->NT == nt && h->BWc == bw) {                          \
        *fwd = k_forward_coop_imr_parts<nt, bw, (nt > 6)>;      \
        *bwd = k_backward_coop_imr_parts<nt, bw, (nt > 6)>;     \
        return JQ_OK;                                           \
    }
    JQ_FOR_EACH_COOP(JQ_PICKCIP)
    JQ_FOR_EACH_BIG(JQ_PICKCIP)
#undef JQ_PICKCIP
    return fail(h, JQ_EUNSUPPORTED, "no cooperative implicit-midpoint kernel for this Hilbert dimension / band width");
}

static int select_coop_imr_kernels(jq_handle* h, bool hbm, prop_kernel_t* fwd, prop_kernel_t* bwd)
{
    if (hbm) {
        *fwd = k_forward_coop_imr<6, 5, true>;
        *bwd = k_backward_coop_imr<6, 5, true>;
        return JQ_OK;
    }
#define JQ_PICKCI(nt, bw)                                 \
    if (h->NT == nt && h->BWc == bw) {                    \
        *fwd = k_forward_coop_imr<nt, bw, (nt > 6)>;      \
        *bwd = k_backward_coop_imr<nt, bw, (nt > 6)>;     \
        return JQ_OK;                                     \
    }
    JQ_FOR_EACH_COOP(JQ_PICKCI)
    JQ_FOR_EACH_BIG(JQ_PICKCI)
    JQ_PICKCI(1, 0)
#undef JQ_PICKCI
    return fail(h, JQ_EUNSUPPORTED, "no cooperative implicit-midpoint kernel for this Hilbert dimension / band width");
}

