// juqbox_hip.hip -- C ABI (include/juqbox_hip.h) and host orchestration of the gfx950 propagator.
//
// One evaluation = { forward sweep, terminal condition, backward sweep(s), gradient assembly } over a
// batch of N*nsamples state columns.  Time is processed in chunks: for each chunk the controls are
// evaluated on the device (k_ctrl), the K(t)/S(t) tile stream is generated into a reusable HBM buffer
// (k_stream) and one persistent propagator launch consumes it (k_forward / k_backward).  Everything
// runs in order on the handle's HIP stream; the call returns after one stream synchronisation.
#include "../../include/juqbox_hip.h"
#include "jq_aux_kernels.h"
#include "jq_coop_kernels.h"
#include "jq_kernels.h"
#include "jq_lane_kernels.h"
#include "jq_rowlane_kernels.h"
#include "jq_rowlane_imr_kernels.h"
#include "jq_coop_imr_kernels.h"
#include "jq_huge_kernels.h"
#include "jq_options.h"

#include <rccl/rccl.h>   // types and prototypes only: librccl is loaded at run time by jq_create_multi (load_rccl)

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#ifndef JQ_SRC_HASH
#define JQ_SRC_HASH "unknown"      // (the Makefile passes the SHA-256 prefix of the library's sources)
#endif
#ifndef JQ_CODE_HASH
#define JQ_CODE_HASH "unknown"
#endif
#define JQ_VERSION "gfx950 juqbox_hip 0.4.0 src:" JQ_SRC_HASH " code:" JQ_CODE_HASH
#ifndef JQ_MINW_MAXNT
#define JQ_MINW_MAXNT 2      // tile counts up to which two workgroups share a CU (slab kernels; jq_kernel_inst.hip)
#endif

static thread_local std::string g_create_error;
// JQ_DEBUG_TIMING=1: every propagator launch's HIP-event time on stderr (development aid; changes no result and no kernel choice)
static bool debug_timing()
{
    static const bool on = getenv("JQ_DEBUG_TIMING") != nullptr;
    return on;
}

// build manifest (scripts/make_manifest.py -> build/manifest.c): {"<object tag>": {"vgpr_form": .., "fallback": .., "max_vgprs": ..,
// "max_agprs": .., "max_scratch_bytes": ..}, ...} -- flat entries, quoted by jq_plan_info
extern "C" const char jq_build_manifest[];

struct jq_handle {
    int device = 0;
    JqOptions opt;              // per-handle options (jq_options.h): jq_create_opts / JQ_OPTIONS / jq_set_option
    hipStream_t stream = nullptr;
    // problem
    int Ntot = 0, N = 0, Nc = 0, Nfreq = 0, nsteps = 0, m = 0, objFuncType = 1;
    int integrator = 1;         // 1 Stormer-Verlet, 2 implicit midpoint (JACOBI_SOLVER_M fixed-point solver)
    int imr_max_iter = 100;
    double imr_tol = 1e-12;
    int solver_id = 1;          // 1 NEUMANN_SOLVER, 2 JACOBI_SOLVER
    double solver_tol = 0.0;
    double T = 0.0;
    int NT = 0, KT = 0, NP = 0, sps = 0;
    int parts = 1;              // N > 16: a sample's columns take `parts` = ceil(N / 16) consecutive slabs (sps = 1)
    int BW = 0;                 // block band width the kernels are instantiated for (JQ_BW_OD: see jq_kernels.h)
    int BWc = 0;                // ... of the cooperative kernels (plain band)
    // More than JQ_MAXNC (= 4, the kernels' trace / carry bookkeeping) control Hamiltonians: the propagators see ALL controls
    // in K(t), S(t) (k_ctrl / k_stream are generic), only the gradient traces are per control -- the backward sweep then runs once
    // per GROUP of at most JQ_MAXNC controls (ctrl_groups(); 5 .. 8 controls: two sweeps), each with its own trace images.
    int NcK = 0;                // controls per backward sweep the LDS plan is made for: min(Nc, JQ_MAXNC)
    std::vector<int> bw_trace;  // [Nc]
    long long mat_elems = 0;    // doubles per operator image slot ("stride"): band tiles, padded to 1 KiB
    long long mat_elems_c = 0;  // ... in the row-window layout of the cooperative kernels (0: not available)
    bool coop_ok = false;       // the cooperative Stormer-Verlet kernels fit the LDS (dense 96 x 96: only the implicit-midpoint variant that reads its images from HBM)
    int coop_max_slabs = 256;   // batches with at most this many slabs (= CUs: one workgroup each) use the cooperative kernels
    long long state_stride = 0;
    int nslots = 2;             // LDS ring depth of the forward kernel
    int nslots_bwd = 2;         // ... of the backward kernel (shares LDS with carry + parking images)
    int park_lds = 0;           // backward kernel parks its dormant array in LDS (1) or HBM (0)
    int batch = 0;              // > 0: batched staging (K/S images of `batch` time steps per DMA burst); < 0: window staging
    bool big = false;           // Ntot > 96 (NT = 7 .. 16): only the cooperative kernels with operators read from HBM (jq_coop_kernels.h
                                // OpCursor) exist -- Stormer-Verlet, Neumann solver, any batch size
    bool huge = false;          // Ntot > 256 (more than 16 tile rows): the run-time-size kernels of jq_huge_kernels.h (one workgroup of 16 waves per slab,
                                // every vector of a step in a global work area, dense tiles); a huge handle is also `big`
    bool force_plain = false;   // full leakage weights WITH the Jacobi solver on a 4 x 4 x n plan whose kernels do not combine the two (one tile row,
                                // or seven / eight): the handle is planned without that structure (cooperative / slab kernels that do)
    bool replanned = false;     // jq_update_hconst re-planned this handle (a later drift plans again when it violates the plan or regains a better structure)
    bool in_split = false;      // run_eval is evaluating one part of a split batch
    double* d_pk2 = nullptr;    // packed result of the first part of a split batch
    size_t cap_pk2 = 0;
    int dq_max_quads = 0;       // no structure, 17 .. 32 levels: batches of at most this many column quads on the DENSE cooperative-quad kernels (round 6)
    double *d_himg_dq = nullptr, *d_cimg_dq = nullptr;      // their operator images (jq_host_images.h dq_image), JQ_DQ_ELEMS doubles each
    int cq_max_quads = 0;       // JQ_BW_T4 structure: batches of at most this many column quads (4 columns) run on the cooperative-quad
                                // (latency) kernels, one workgroup of NT waves per quad (0: never)
    int quad_max_slabs = 0;     // JQ_BW_T4 structure: batches of at most this many slabs may use the quad-layout kernels (0: never)
    int num_cu = 256;
    int lane_np = 0;            // > 0: lane kernels available (Ntot <= 12), padded Hilbert dimension NP
    long long lane_stride = 0;  // doubles per plain NP x NP operator image (padded to 64 B)
    int lane_min_cols = 0, lane_max_cols = 0;   // column counts (samples x N) routed to the lane kernels
    int rl_npj = 0;             // > 0: row-lane kernels available (Ntot <= 16), padded row length NPJ
    long long rl_stride = 0;    // doubles per [16][NPJ] operator image
    int rl_max_cols = 0;        // batches of at most this many columns use the row-lane kernels (latency regime)
    std::vector<double> Hconst, Hsym, Hanti, Uinit, Utr, Uti, wd, cfreq;
    std::vector<double> rfreq;  // uncoupled controls (Nunc > 0): params.Rfreq; empty otherwise
    double* d_rfreq = nullptr;
    // Full leakage weights (jq_update_wmat): W = wmat_real + i wmat_imag = sum_{k < wrank} lam_k f_k f_k^H.  wrank > 0: `wd` is all
    // zero, Wr / Wi keep the caller's matrices (re-planning applies them again), wlr is the kernels' table
    // lam[JQ_MAX_WRANK] | a_k[NP], b_k[NP] per k in natural row order (PropArgs::wlr).
    int wrank = 0;
    int wlam = JQ_MAX_WRANK;    // lam slots in front of the rows of the table: max(JQ_MAX_WRANK, wrank)
    std::vector<double> Wr, Wi, wlr;
    double* d_wlr = nullptr;
    bool wlr_real = false;      // every kept eigenvector is real (wmat_imag = 0): the cooperative-quad kernels take rank <= 4 of those
    std::vector<double> tf, tb;
    // device buffers (owned)
    double *d_cimg = nullptr, *d_park = nullptr;
    double* d_cq3 = nullptr;    // hand-off buffer of k_backward_cq3 (jq_cq_split_kernels.h)
    size_t cap_cq3 = 0;
    // Three-workgroup latency kernels (k_backward_cq3 / k_backward_cq_imr3): their workgroups wait for each other, so all of them must be
    // resident at once.  Inside this process that is CHECKED (DevGate below: the split is taken only while no other evaluation runs on
    // the device, and nothing else starts until it is through); what other processes, CU masks or a partitioned device do cannot be
    // seen from here -- a wait that is declared dead raises the error word, the evaluation is repeated without the split (the word is
    // read after the FIRST backward launch), and the handle leaves the split alone for cq3_skip evaluations (4, 8, 16 ... per fault;
    // for good after JQ_CQ3_MAX_FAULTS faults).
    bool cq3_off = false;       // never again on this handle (too many faults)
    int cq3_faults = 0;         // launches that reported a dead wait / workgroups on different XCDs
    int cq3_faults_xcd = 0;     // ... of them: the workgroups of a quad ran on different XCDs (error word 2)
    int cq3_busy = 0;           // launches abandoned at their start-up rendezvous (the GPU was busy: not every workgroup became resident in time)
    int cq3_busy_streak = 0;    // ... in a row (sets the cool-down)
    double cq3_us_per_step = 0.0;   // measured duration of a split backward launch per time step (sizes the in-launch wait guard)
    int cq3_skip = 0;           // evaluations left for which the split is not tried
    std::string cq3_last;       // why the last batch of the latency families did / did not take the split (jq_plan_info)
    double* d_qsplit = nullptr; // hand-off buffer of k_backward_qsplit (jq_quad_split_kernels.h): [quad][parity][2][NT][64]
    size_t cap_qsplit = 0;
    double *d_himg_c = nullptr, *d_cimg_c = nullptr;   // operator images in the cooperative layout
    double *d_himg_l = nullptr, *d_uinit_l = nullptr, *d_vtr_l = nullptr, *d_vti_l = nullptr;   // lane kernels
    double *d_himg_r = nullptr, *d_uinit_r = nullptr, *d_vtr_r = nullptr, *d_vti_r = nullptr;   // row-lane kernels
    double *d_cimg_l = nullptr, *d_cimg_r = nullptr;   // their trace images in control-group order
    double *d_himg = nullptr, *d_uimg = nullptr, *d_vtr = nullptr, *d_vti = nullptr, *d_tabs = nullptr;
    double *d_tf = nullptr, *d_tb = nullptr, *d_cfreq = nullptr, *d_pcof = nullptr;
    double *d_stream = nullptr, *d_pq = nullptr;
    double *d_state = nullptr, *d_state_save = nullptr, *d_colinfo = nullptr, *d_traces = nullptr, *d_R = nullptr;
    double *d_grad = nullptr, *d_res = nullptr;
    double *d_wq = nullptr, *d_pack = nullptr;   // ensemble weights per sample; packed result [2 + 2 nCoeff] (multi-device all-reduce)
    size_t cap_pcof = 0, cap_slabs = 0, cap_traces = 0, cap_grad = 0, cap_res = 0, cap_state = 0, cap_colinfo = 0, cap_wq = 0, cap_pack = 0;
    int chunk_steps = 0;
    // Structure embedding (try_embed): a second handle of the SAME problem with its two fastest Kronecker factors zero-padded
    // to 4 levels each (row i1 + d1 i2 + d1 d2 i3 -> i1 + 4 i2 + 16 i3), under which the operators have the JQ_BW_T4 structure;
    // batches that would otherwise run on the dense / band MFMA kernels go there (quad-layout / JQ_BW_T4 slab kernels).
    jq_handle* emb = nullptr;
    std::vector<int> emb_row;   // user row -> row of the embedded problem
    int emb_mode = 1;           // JQ_EMBED: 0 never, 1 for batches of the MFMA families (default), 2 whenever possible (tests)
    bool is_emb = false;        // this handle IS an embedded twin: no lane / row-lane / cooperative families, no further embedding
    // multi-device handle (jq_create_multi): one single-device handle per GPU and one RCCL communicator each; such a
    // handle owns no device memory itself
    std::vector<jq_handle*> subs;
    std::vector<ncclComm_t> comms;
    bool host_reduce = false;   // option multi_same_device test mode: host-side sum instead of the ncclAllReduce (no communicators)
    bool comm_broken = false;   // an RCCL call failed inside a collective: the communicators are aborted at destroy, calls refuse
    int rccl_checks = 0;        // all-reduces of this handle that were verified against the host-order sum (JQ_RCCL_SELFCHECK)
    std::vector<hipEvent_t> ev;
    std::string err;
    jq_timing timing = {};
};

#define HIPCHK(h, call)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            char buf_[512];                                                                          \
            snprintf(buf_, sizeof buf_, "HIP error '%s' at %s:%d (%s)", hipGetErrorString(e_), __FILE__, __LINE__, #call); \
            (h)->err = buf_;                                                                         \
            return JQ_EHIP;                                                                          \
        }                                                                                            \
    } while (0)

static int fail(jq_handle* h, int code, const char* msg)
{
    h->err = msg;
    return code;
}

// control groups: group g of ctrl_ngroups(Nc) holds the controls [ctrl_gstart(Nc, g), ctrl_gstart(Nc, g + 1)); sizes differ by <= 1
static int ctrl_ngroups(int Nc) { return (Nc + JQ_MAXNC - 1) / JQ_MAXNC; }
static int ctrl_gstart(int Nc, int g)
{
    const int ng = ctrl_ngroups(Nc), base = Nc / ng, rem = Nc % ng;
    return g * base + std::min(g, rem);
}

// restores the caller's current HIP device when a multi-device entry point returns (a Julia / PyTorch caller that was on
// another device must not find itself switched)
struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

#include "jq_host_images.h"      // operator / state images in the kernels' layouts and the structure tests the planner uses
#include "jq_host_create.h"      // device buffers, uploads, jq_create* (planning from the operators' nonzero structure), structure embedding
#include "jq_host_update.h"      // the mutations scripts apply to params after construction: solver / integrator, target, drift (re-planning), leakage weights
#include "jq_host_select.h"      // the kernel instantiations (compiled in their own translation units) and the tables that pick one
#include "jq_host_eval.h"      // run_eval: how a batch is routed to a kernel family and propagated chunk by chunk
extern "C" int jq_traceobjgrad(jq_handle* h, const double* pcof, int32_t ncoeff, int32_t evaladjoint, double* out4,
                               double* totalgrad, double* infidelgrad, double* leakgrad)
{
    if (!h) return JQ_EINVAL;
    if (!pcof || !out4) return fail(h, JQ_EINVAL, "jq_traceobjgrad: NULL pointer");
    if (evaladjoint && (!totalgrad || !infidelgrad || !leakgrad))
        return fail(h, JQ_EINVAL, "jq_traceobjgrad: gradient outputs are required when evaladjoint != 0");
    JQ_ON_FIRST(h, jq_traceobjgrad(s0_, pcof, ncoeff, evaladjoint, out4, totalgrad, infidelgrad, leakgrad))
    EvalOut o;
    int rc = run_eval(h, pcof, ncoeff, 1, nullptr, nullptr, nullptr, evaladjoint != 0, nullptr, nullptr, &o);
    if (rc) return rc;
    const double primary = o.res[0], secondary = o.res[1];
    out4[0] = primary + secondary;  // objfv (src/evalobjgrad.jl:765-766)
    out4[1] = primary;
    out4[2] = secondary;
    out4[3] = primary;              // traceInfidelity == 1 - |s|^2 for pFidType 2 (:792)
    if (evaladjoint) {
        for (int i = 0; i < ncoeff; ++i) totalgrad[i] = o.grad0[i];
        if (h->objFuncType != 1) {
            for (int i = 0; i < ncoeff; ++i) {
                infidelgrad[i] = o.grad1[i];
                leakgrad[i] = o.grad0[i] - o.grad1[i];  // :947
            }
        } else {
            for (int i = 0; i < ncoeff; ++i) {
                infidelgrad[i] = o.grad0[i];  // :951
                leakgrad[i] = 0.0;
            }
        }
    }
    return JQ_OK;
}

extern "C" int jq_state_history(jq_handle* h, const double* pcof, int32_t ncoeff, double* ur, double* ui)
{
    return jq_traceobj_verbose(h, pcof, ncoeff, nullptr, ur, ui);
}

extern "C" int jq_traceobj_verbose(jq_handle* h, const double* pcof, int32_t ncoeff, double* out4, double* ur, double* ui)
{
    if (!h) return JQ_EINVAL;
    if (!pcof || !ur || !ui) return fail(h, JQ_EINVAL, "jq_state_history: NULL pointer");
    JQ_ON_FIRST(h, jq_traceobj_verbose(s0_, pcof, ncoeff, out4, ur, ui))
    HIPCHK(h, hipSetDevice(h->device));
    const size_t len = (size_t)h->Ntot * h->N * (h->nsteps + 1);
    double *d_r = nullptr, *d_i = nullptr;
    HIPCHK(h, hipMalloc((void**)&d_r, len * sizeof(double)));
    if (hipMalloc((void**)&d_i, len * sizeof(double)) != hipSuccess) {
        (void)hipFree(d_r);
        return fail(h, JQ_ENOMEM, "jq_state_history: out of device memory");
    }
    (void)hipMemset(d_r, 0, len * sizeof(double));
    (void)hipMemset(d_i, 0, len * sizeof(double));
    EvalOut o;
    int rc = run_eval(h, pcof, ncoeff, 1, nullptr, nullptr, nullptr, false, d_r, d_i, &o);
    if (rc == JQ_OK) {
        if (hipMemcpy(ur, d_r, len * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(ui, d_i, len * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(h, JQ_EHIP, "jq_state_history: copy back failed");
        // usaver[:,:,1] = Uinit ; usavei[:,:,1] = -vi = 0 (src/evalobjgrad.jl:679-680)
        for (size_t i = 0; i < (size_t)h->Ntot * h->N && rc == JQ_OK; ++i) {
            ur[i] = h->Uinit[i];
            ui[i] = -0.0;
        }
        if (out4 && rc == JQ_OK) {   // the objective of the same forward sweep (src/evalobjgrad.jl:759-792)
            out4[0] = o.res[0] + o.res[1];
            out4[1] = o.res[0];
            out4[2] = o.res[1];
            out4[3] = o.res[0];
        }
    }
    (void)hipFree(d_r);
    (void)hipFree(d_i);
    return rc;
}

extern "C" int jq_state_populations(jq_handle* h, const double* pcof, int32_t ncoeff, const int32_t* group_of_row,
                                    int32_t ngroups, int32_t every, int32_t nout, double* pop, double* maxpop)
{
    if (!h) return JQ_EINVAL;
    if (!pcof) return fail(h, JQ_EINVAL, "jq_state_populations: NULL pointer");
    if (!pop && !maxpop) return fail(h, JQ_EINVAL, "jq_state_populations: pop and maxpop are both NULL");
    JQ_ON_FIRST(h, jq_state_populations(s0_, pcof, ncoeff, group_of_row, ngroups, every, nout, pop, maxpop))
    if (pop) {
        if (every < 1 || nout != h->nsteps / every + 1)
            return fail(h, JQ_EINVAL, "jq_state_populations: need every >= 1 and nout == nsteps/every + 1");
        if (ngroups < 1 || (!group_of_row && ngroups != h->Ntot))
            return fail(h, JQ_EINVAL, "jq_state_populations: ngroups must be Ntot when group_of_row is NULL");
        if (group_of_row)
            for (int r = 0; r < h->Ntot; ++r)
                if (group_of_row[r] >= ngroups) return fail(h, JQ_EINVAL, "jq_state_populations: group index >= ngroups");
    }
    HIPCHK(h, hipSetDevice(h->device));
    const size_t len = (size_t)h->Ntot * h->N * (h->nsteps + 1);
    const size_t npop = pop ? (size_t)ngroups * h->N * nout : 0;
    double *d_r = nullptr, *d_i = nullptr, *d_pop = nullptr, *d_max = nullptr;
    int* d_grp = nullptr;
    int rc = JQ_OK;
    auto cleanup = [&]() {
        (void)hipFree(d_r); (void)hipFree(d_i); (void)hipFree(d_pop); (void)hipFree(d_max); (void)hipFree(d_grp);
    };
    if (hipMalloc((void**)&d_r, len * sizeof(double)) != hipSuccess || hipMalloc((void**)&d_i, len * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&d_pop, std::max<size_t>(npop, 1) * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&d_max, (size_t)h->Ntot * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&d_grp, (size_t)h->Ntot * sizeof(int)) != hipSuccess) {
        cleanup();
        return fail(h, JQ_ENOMEM, "jq_state_populations: out of device memory");
    }
    (void)hipMemset(d_r, 0, len * sizeof(double));
    (void)hipMemset(d_i, 0, len * sizeof(double));
    EvalOut o;
    rc = run_eval(h, pcof, ncoeff, 1, nullptr, nullptr, nullptr, false, d_r, d_i, &o);
    if (rc == JQ_OK) {
        // usaver[:,:,1] = Uinit ; usavei[:,:,1] = 0 (src/evalobjgrad.jl:679-680)
        bool ok = hipMemcpy(d_r, h->Uinit.data(), (size_t)h->Ntot * h->N * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
        if (pop && ok) {
            if (group_of_row) ok = hipMemcpy(d_grp, group_of_row, (size_t)h->Ntot * sizeof(int), hipMemcpyHostToDevice) == hipSuccess;
            const long long nthr = (long long)h->N * nout;
            hipLaunchKernelGGL(k_pop_groups, dim3((unsigned)((nthr + 127) / 128)), dim3(128), 0, h->stream, d_r, d_i, h->Ntot, h->N,
                               every, nout, group_of_row ? d_grp : nullptr, ngroups, d_pop);
        }
        if (maxpop && ok)
            hipLaunchKernelGGL(k_pop_max, dim3(h->Ntot), dim3(256), 0, h->stream, d_r, d_i, h->Ntot,
                               (long long)h->N * (h->nsteps + 1), d_max);
        ok = ok && hipStreamSynchronize(h->stream) == hipSuccess && hipGetLastError() == hipSuccess;
        if (pop && ok) ok = hipMemcpy(pop, d_pop, npop * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess;
        if (maxpop && ok) ok = hipMemcpy(maxpop, d_max, (size_t)h->Ntot * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess;
        if (!ok) rc = fail(h, JQ_EHIP, "jq_state_populations: device reduction or copy failed");
    }
    cleanup();
    return rc;
}

extern "C" int jq_eval_f_g_grad(jq_handle* h, const double* pcof, int32_t ncoeff, const double* nodes, const double* weights,
                                int32_t nquad, const double* shift, int32_t compute_adjoint, double* out2, double* infid_grad,
                                double* leak_grad)
{
    if (!h) return JQ_EINVAL;
    if (!pcof || !nodes || !weights || !out2) return fail(h, JQ_EINVAL, "jq_eval_f_g_grad: NULL pointer");
    if (nquad < 1) return fail(h, JQ_EINVAL, "jq_eval_f_g_grad: nquad must be >= 1");
    if (compute_adjoint && (!infid_grad || !leak_grad))
        return fail(h, JQ_EINVAL, "jq_eval_f_g_grad: gradient outputs are required when compute_adjoint != 0");
    if (!h->subs.empty())
        return multi_eval_f_g_grad(h, pcof, ncoeff, nodes, weights, nquad, shift, compute_adjoint != 0, out2, infid_grad, leak_grad);
    EvalOut o;
    int rc = run_eval(h, pcof, ncoeff, nquad, nodes, weights, shift, compute_adjoint != 0, nullptr, nullptr, &o);
    if (rc) return rc;
    double inf = 0.0, leak = 0.0;
    for (int i = 0; i < nquad; ++i) {  // src/ipopt_interface.jl:58-59
        inf += o.res[(size_t)i * 4 + 0] * weights[i];
        leak += o.res[(size_t)i * 4 + 1] * weights[i];
    }
    out2[0] = inf;
    out2[1] = leak;
    if (compute_adjoint) {
        if (h->objFuncType != 1) {
            for (int i = 0; i < ncoeff; ++i) {
                infid_grad[i] = o.grad1[i];
                leak_grad[i] = o.grad0[i] - o.grad1[i];
            }
        } else {
            for (int i = 0; i < ncoeff; ++i) {
                infid_grad[i] = o.grad0[i];  // "infidelgrad stores the totalgrad" (src/evalobjgrad.jl:949-951)
                leak_grad[i] = 0.0;
            }
        }
    }
    return JQ_OK;
}

extern "C" int jq_eval_f_g_grad_dev(jq_handle* h, const double* pcof, int32_t ncoeff, const double* nodes, const double* weights,
                                    int32_t nquad, const double* shift, int32_t compute_adjoint, void* d_packed)
{
    if (!h) return JQ_EINVAL;
    if (!pcof || !nodes || !weights || !d_packed) return fail(h, JQ_EINVAL, "jq_eval_f_g_grad_dev: NULL pointer");
    if (nquad < 0) return fail(h, JQ_EINVAL, "jq_eval_f_g_grad_dev: nquad must be >= 0");
    if (!h->subs.empty())
        return fail(h, JQ_EINVAL, "jq_eval_f_g_grad_dev: multi-device handles reduce inside jq_eval_f_g_grad; use that entry");
    {   // the packed vector must live on the handle's GPU (a tensor allocated on torch's current device of a process that
        // created the handle on another one would hand k_pack a peer pointer)
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, d_packed) != hipSuccess) {
            (void)hipGetLastError();
            return fail(h, JQ_EINVAL, "jq_eval_f_g_grad_dev: d_packed is not a device pointer");
        }
        if (at.type != hipMemoryTypeDevice || at.device != h->device) {
            char buf[160];
            snprintf(buf, sizeof buf, "jq_eval_f_g_grad_dev: d_packed lives on device %d, the handle on device %d", at.device, h->device);
            return fail(h, JQ_EINVAL, buf);
        }
    }
    if (nquad == 0) {     // a rank without a shard contributes zeros to the all-reduce
        HIPCHK(h, hipSetDevice(h->device));
        HIPCHK(h, hipMemsetAsync(d_packed, 0, (2 + 2 * (size_t)ncoeff) * sizeof(double), h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        h->timing = jq_timing{};
        return JQ_OK;
    }
    EvalOut o;
    return run_eval(h, pcof, ncoeff, nquad, nodes, weights, shift, compute_adjoint != 0, nullptr, nullptr, &o, (double*)d_packed);
}

extern "C" int jq_traceobj_sweep(jq_handle* h, const double* pcof, int32_t ncoeff, const double* nodes, int32_t nquad,
                                 const double* shift, double* out)
{
    if (!h) return JQ_EINVAL;
    if (!pcof || !nodes || !out) return fail(h, JQ_EINVAL, "jq_traceobj_sweep: NULL pointer");
    if (nquad < 1) return fail(h, JQ_EINVAL, "jq_traceobj_sweep: nquad must be >= 1");
    if (!h->subs.empty()) return multi_traceobj_sweep(h, pcof, ncoeff, nodes, nquad, shift, out);
    EvalOut o;
    int rc = run_eval(h, pcof, ncoeff, nquad, nodes, nullptr, shift, false, nullptr, nullptr, &o);
    if (rc) return rc;
    for (int i = 0; i < nquad; ++i) {
        const double primary = o.res[(size_t)i * 4 + 0], secondary = o.res[(size_t)i * 4 + 1];
        out[(size_t)i * 4 + 0] = primary + secondary;
        out[(size_t)i * 4 + 1] = primary;
        out[(size_t)i * 4 + 2] = secondary;
        out[(size_t)i * 4 + 3] = primary;
    }
    return JQ_OK;
}


#include "jq_host_multi.h"      // multi-device handles: one process, N GPUs, one RCCL all-reduce
#include "jq_host_info.h"      // options of a live handle, plan and timing introspection