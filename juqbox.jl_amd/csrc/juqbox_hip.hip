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

// A-fragment tile image of a column-major Ntot x Ntot matrix: only the tiles of the block band
// |mt - kk/4| <= BW are stored, in walk order (kk outer, mt inner); tile (mt,kk) lane l holds
// M[16*mt + (l&15)][4*kk + (l>>4)]; zero padded.
static void tile_image(const double* M, int Ntot, int NT, int BW, double* img, bool SD = false)
{
    if (BW == JQ_BW_T4) {
        // compact image (JQ_T4_ELEMS doubles): per 16-row block the 4x4 diagonal blocks of its four 4-row groups rho = 4 mt + b,
        // element JQ_T4_AIDX(rho, k, i) = M[4 rho + i][4 rho + k] (= lane 16 k + 4 b + i of the quad-layout MFMA's A operand), then the
        // coupling coefficients
        // per block mt: JQ_T4_CIDX(g, r, term: group rho-1, rho+1 (same 16-row block), rho-4, rho+4) <-> row 4 rho + g, rho = 4 mt + r
        const int NR = 4 * NT;
        for (size_t i = 0; i < (size_t)JQ_T4_ELEMS(NT); ++i) img[i] = 0.0;
        if (!SD)
            for (int rho = 0; rho < NR; ++rho)
                for (int k = 0; k < 4; ++k)
                    for (int i = 0; i < 4; ++i) {
                        const int row = 4 * rho + i, col = 4 * rho + k;
                        img[JQ_T4_AIDX(rho, k, i)] = (row < Ntot && col < Ntot) ? M[row + (size_t)Ntot * col] : 0.0;
                    }
        double* cf = img + (size_t)NR * JQ_T4_TILE;
        for (int rho = 0; rho < NR; ++rho)
            for (int g = 0; g < 4; ++g) {
                const int row = 4 * rho + g, r = rho & 3;
                const int nbr[4] = {r > 0 ? row - 4 : -1, r < 3 ? row + 4 : -1, row - 16, row + 16};
                for (int t = 0; t < 4; ++t) {
                    const int col = nbr[t];
                    cf[(rho >> 2) * 64 + JQ_T4_CIDX(g, r, t)] = (col >= 0 && col < Ntot && row < Ntot) ? M[row + (size_t)Ntot * col] : 0.0;
                }
            }
        return;
    }
    const int KT = 4 * NT;
    size_t idx = 0;
    for (int kk = 0; kk < KT; ++kk)
        for (int mt = 0; mt < NT; ++mt) {
            const int kb = kk >> 2;
            if (!block_on(BW, SD, mt, kb)) continue;
            for (int l = 0; l < 64; ++l) {
                const int row = 16 * mt + (l & 15), col = 4 * kk + (l >> 4);
                img[idx * 64 + l] = (row < Ntot && col < Ntot) ? M[row + (size_t)Ntot * col] : 0.0;
            }
            ++idx;
        }
    if (BW == JQ_BW_OD) {
        // diagonals of the first off-diagonal blocks: [mt][dir: block mt-1, block mt+1][g][r] <-> row 16mt + 4r + g
        double* cf = img + idx * 64;
        for (int mt = 0; mt < NT; ++mt)
            for (int dir = 0; dir < 2; ++dir) {
                const int nb = mt + (dir ? 1 : -1);
                for (int g = 0; g < 4; ++g)
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * mt + 4 * r + g, col = 16 * nb + 4 * r + g;
                        cf[((mt * 2 + dir) * 4 + g) * 4 + r] =
                            (nb >= 0 && nb < NT && row < Ntot && col < Ntot) ? M[row + (size_t)Ntot * col] : 0.0;
                    }
            }
    }
}

// JQ_BW_T4 structure: entries outside the 4x4 diagonal blocks only at (i, i +- 4) inside one 16-row block or at (i, i +- 16)
static bool t4_structure(const double* M, int Ntot)
{
    for (int col = 0; col < Ntot; ++col)
        for (int row = 0; row < Ntot; ++row) {
            if (M[row + (size_t)Ntot * col] == 0.0 || row / 4 == col / 4) continue;
            const int d = row - col;
            const bool same16 = (row / 16 == col / 16);
            if (!((same16 && (d == 4 || d == -4)) || d == 16 || d == -16)) return false;
        }
    return true;
}
// parts of the T4 image of M that are non-zero: JQ_T4_DIAG | JQ_T4_RTERMS | JQ_T4_MTERMS
static int t4_mode(const double* M, int Ntot)
{
    int mode = 0;
    for (int col = 0; col < Ntot; ++col)
        for (int row = 0; row < Ntot; ++row) {
            if (M[row + (size_t)Ntot * col] == 0.0) continue;
            if (row / 4 == col / 4) mode |= JQ_T4_DIAG;
            else if (row - col == 4 || col - row == 4) mode |= JQ_T4_RTERMS;
            else mode |= JQ_T4_MTERMS;
        }
    return mode;
}

// true if M is block tridiagonal (16x16 blocks) and every off-diagonal block is a diagonal matrix
static bool offdiag_blocks_diagonal(const double* M, int Ntot)
{
    for (int col = 0; col < Ntot; ++col)
        for (int row = 0; row < Ntot; ++row) {
            if (M[row + (size_t)Ntot * col] == 0.0) continue;
            const int d = row / 16 - col / 16;
            if (d == 0) continue;
            if (std::abs(d) > 1 || row % 16 != col % 16) return false;
        }
    return true;
}

// Row-window layout of the cooperative kernels (jq_coop_kernels.h): for tile row mt the NB k-blocks
// kb0(mt)..kb0(mt)+NB-1, 4 tiles each, rows consecutively.
static void tile_image_coop(const double* M, int Ntot, int NT, int BW, double* img)
{
    const int NB = coop_nb(NT, BW);
    if (BW == JQ_BW_OD) {
        // per tile row: the 4 tiles of the diagonal block, then [dir: block mt-1, mt+1][g][r] <-> row 16mt + 4r + g
        for (int mt = 0; mt < NT; ++mt) {
            double* row = img + (size_t)mt * coop_row_elems(NT, BW);
            for (int r4 = 0; r4 < 4; ++r4)
                for (int l = 0; l < 64; ++l) {
                    const int rr = 16 * mt + (l & 15), col = 4 * (4 * mt + r4) + (l >> 4);
                    row[r4 * 64 + l] = (rr < Ntot && col < Ntot) ? M[rr + (size_t)Ntot * col] : 0.0;
                }
            for (int dir = 0; dir < 2; ++dir) {
                const int nb = mt + (dir ? 1 : -1);
                for (int g = 0; g < 4; ++g)
                    for (int r = 0; r < 4; ++r) {
                        const int rr = 16 * mt + 4 * r + g, col = 16 * nb + 4 * r + g;
                        row[256 + (dir * 4 + g) * 4 + r] = (nb >= 0 && nb < NT && rr < Ntot && col < Ntot) ? M[rr + (size_t)Ntot * col] : 0.0;
                    }
            }
        }
        return;
    }
    size_t idx = 0;
    for (int mt = 0; mt < NT; ++mt) {
        const int kb0 = coop_kb0(NT, BW, mt);
        for (int j = 0; j < NB; ++j)
            for (int r = 0; r < 4; ++r) {
                const int kk = 4 * (kb0 + j) + r;
                for (int l = 0; l < 64; ++l) {
                    const int row = 16 * mt + (l & 15), col = 4 * kk + (l >> 4);
                    img[idx * 64 + l] = (row < Ntot && col < Ntot) ? M[row + (size_t)Ntot * col] : 0.0;
                }
                ++idx;
            }
    }
}

// true if every diagonal 16x16 block of M is zero
static bool diag_blocks_zero(const double* M, int Ntot)
{
    for (int col = 0; col < Ntot; ++col)
        for (int row = 0; row < Ntot; ++row)
            if (row / 16 == col / 16 && M[row + (size_t)Ntot * col] != 0.0) return false;
    return true;
}

// smallest block band width that contains every nonzero of M
static int block_band(const double* M, int Ntot)
{
    int bw = 0;
    for (int col = 0; col < Ntot; ++col)
        for (int row = 0; row < Ntot; ++row)
            if (M[row + (size_t)Ntot * col] != 0.0) bw = std::max(bw, std::abs(row / 16 - col / 16));
    return bw;
}

// register-layout images [parts][KT][64] of an Ntot x N array: N <= 16: one image, the N columns replicated over the sps
// samples of a slab; N > 16: part p holds the columns 16 p .. 16 p + 15
static void slab_image(const double* A, int Ntot, int N, int sps, int KT, double* img, int parts = 1)
{
    for (int p = 0; p < parts; ++p)
        for (int kk = 0; kk < KT; ++kk)
            for (int l = 0; l < 64; ++l) {
                const int row = 4 * kk + (l >> 4), col = l & 15;
                const int scol = parts > 1 ? 16 * p + col : col % N;
                const bool on = parts > 1 ? scol < N : col < sps * N;
                img[((size_t)p * KT + kk) * 64 + l] = (row < Ntot && on) ? A[row + (size_t)Ntot * scol] : 0.0;
            }
}

// plain row-major NP x NP image of a column-major Ntot x Ntot matrix (lane kernels), zero padded
static void plain_image(const double* M, int Ntot, int NP, double* img)
{
    for (int i = 0; i < Ntot; ++i)
        for (int j = 0; j < Ntot; ++j) img[(size_t)i * NP + j] = M[i + (size_t)Ntot * j];
}

// [N][NP] image of an Ntot x N array (lane kernels)
static void column_image(const double* A, int Ntot, int N, int NP, double* img)
{
    for (int c = 0; c < N; ++c)
        for (int r = 0; r < Ntot; ++r) img[(size_t)c * NP + r] = A[r + (size_t)Ntot * c];
}

// [16][NPJ] row-major image of a column-major Ntot x Ntot matrix (row-lane kernels), zero padded
static void rowlane_image(const double* M, int Ntot, int NPJ, double* img)
{
    for (int i = 0; i < Ntot; ++i)
        for (int j = 0; j < Ntot; ++j) img[(size_t)i * NPJ + j] = M[i + (size_t)Ntot * j];
}

template <typename T>
static int dev_alloc(jq_handle* h, T** p, size_t count)
{
    if (*p) {
        (void)hipFree(*p);
        *p = nullptr;
    }
    if (hipMalloc((void**)p, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) {
        *p = nullptr;
        (void)hipGetLastError();      // (clear the sticky error: the handle stays usable for smaller requests)
        char buf[160];
        snprintf(buf, sizeof buf, "out of device memory (%zu bytes requested)", std::max<size_t>(count, 1) * sizeof(T));
        return fail(h, JQ_ENOMEM, buf);
    }
    return JQ_OK;
}
// grow-only buffer with its capacity: the capacity is zeroed BEFORE the old buffer is released, so a failed allocation
// leaves (nullptr, 0) behind and the next call allocates again instead of launching on a stale capacity
template <typename T>
static int dev_grow(jq_handle* h, T** p, size_t* cap, size_t need)
{
    if (need <= *cap && *p) return JQ_OK;
    *cap = 0;
    const int rc = dev_alloc(h, p, need);
    if (rc == JQ_OK) *cap = need;
    return rc;
}

static int upload_operators(jq_handle* h)
{
    const size_t nn = (size_t)h->Ntot * h->Ntot;
    // images the tile stream is generated from: [H0 | Hsym_q | Hanti_q] in the kernels' band layout
    // (Ntot > 96: no slab-kernel images, only the cooperative layout below)
    std::vector<double> img((size_t)(1 + 2 * h->Nc) * h->mat_elems, 0.0);
    if (!h->big) {
        tile_image(h->Hconst.data(), h->Ntot, h->NT, h->BW, img.data());
        for (int q = 0; q < h->Nc; ++q) {
            tile_image(h->Hsym.data() + q * nn, h->Ntot, h->NT, h->BW, img.data() + (size_t)(1 + q) * h->mat_elems);
            tile_image(h->Hanti.data() + q * nn, h->Ntot, h->NT, h->BW, img.data() + (size_t)(1 + h->Nc + q) * h->mat_elems);
        }
    }
    HIPCHK(h, hipMemcpy(h->d_himg, img.data(), img.size() * sizeof(double), hipMemcpyHostToDevice));
    // images of the trace products, per control group g: [Hsym_q, q in g | Hanti_q, q in g] at image offset 2 gstart(g) -- for
    // Nc <= JQ_MAXNC simply [Hsym_q | Hanti_q] -- each pair in its own band (0 or BW)
    auto cslot = [&](int q, bool anti) {      // image index of control q's symmetric / antisymmetric trace image
        int g = 0;
        while (ctrl_gstart(h->Nc, g + 1) <= q) ++g;
        const int gs = ctrl_gstart(h->Nc, g), ng = ctrl_gstart(h->Nc, g + 1) - gs;
        return (size_t)(2 * gs + (anti ? ng : 0) + (q - gs));
    };
    std::vector<double> cimg((size_t)(2 * h->Nc) * h->mat_elems, 0.0);
    for (int q = 0; q < h->Nc && !h->big; ++q) {
        const int bwq = (h->BW == JQ_BW_T4) ? JQ_BW_T4 : (h->bw_trace[q] == 0) ? 0 : h->BW;
        const bool sd = (h->BW == JQ_BW_T4) ? false : (h->bw_trace[q] == 2);
        tile_image(h->Hsym.data() + q * nn, h->Ntot, h->NT, bwq, cimg.data() + cslot(q, false) * h->mat_elems, sd);
        tile_image(h->Hanti.data() + q * nn, h->Ntot, h->NT, bwq, cimg.data() + cslot(q, true) * h->mat_elems, sd);
    }
    HIPCHK(h, hipMemcpy(h->d_cimg, cimg.data(), cimg.size() * sizeof(double), hipMemcpyHostToDevice));
    if (h->mat_elems_c > 0) {
        std::vector<double> ic((size_t)(1 + 2 * h->Nc) * h->mat_elems_c, 0.0);
        tile_image_coop(h->Hconst.data(), h->Ntot, h->NT, h->BWc, ic.data());
        for (int q = 0; q < h->Nc; ++q) {
            tile_image_coop(h->Hsym.data() + q * nn, h->Ntot, h->NT, h->BWc, ic.data() + (size_t)(1 + q) * h->mat_elems_c);
            tile_image_coop(h->Hanti.data() + q * nn, h->Ntot, h->NT, h->BWc, ic.data() + (size_t)(1 + h->Nc + q) * h->mat_elems_c);
        }
        HIPCHK(h, hipMemcpy(h->d_himg_c, ic.data(), ic.size() * sizeof(double), hipMemcpyHostToDevice));
        // trace images: the images 1.. of the same array, in control-group order
        std::vector<double> cc((size_t)2 * h->Nc * h->mat_elems_c);
        for (int q = 0; q < h->Nc; ++q) {
            std::copy_n(ic.data() + (size_t)(1 + q) * h->mat_elems_c, h->mat_elems_c, cc.data() + cslot(q, false) * h->mat_elems_c);
            std::copy_n(ic.data() + (size_t)(1 + h->Nc + q) * h->mat_elems_c, h->mat_elems_c, cc.data() + cslot(q, true) * h->mat_elems_c);
        }
        HIPCHK(h, hipMemcpy(h->d_cimg_c, cc.data(), cc.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    if (h->lane_np > 0) {
        std::vector<double> il((size_t)(1 + 2 * h->Nc) * h->lane_stride, 0.0);
        plain_image(h->Hconst.data(), h->Ntot, h->lane_np, il.data());
        for (int q = 0; q < h->Nc; ++q) {
            plain_image(h->Hsym.data() + q * nn, h->Ntot, h->lane_np, il.data() + (size_t)(1 + q) * h->lane_stride);
            plain_image(h->Hanti.data() + q * nn, h->Ntot, h->lane_np, il.data() + (size_t)(1 + h->Nc + q) * h->lane_stride);
        }
        HIPCHK(h, hipMemcpy(h->d_himg_l, il.data(), il.size() * sizeof(double), hipMemcpyHostToDevice));
        std::vector<double> cl((size_t)2 * h->Nc * h->lane_stride);
        for (int q = 0; q < h->Nc; ++q) {
            std::copy_n(il.data() + (size_t)(1 + q) * h->lane_stride, h->lane_stride, cl.data() + cslot(q, false) * h->lane_stride);
            std::copy_n(il.data() + (size_t)(1 + h->Nc + q) * h->lane_stride, h->lane_stride, cl.data() + cslot(q, true) * h->lane_stride);
        }
        HIPCHK(h, hipMemcpy(h->d_cimg_l, cl.data(), cl.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    if (h->rl_npj > 0) {
        std::vector<double> ir((size_t)(1 + 2 * h->Nc) * h->rl_stride, 0.0);
        rowlane_image(h->Hconst.data(), h->Ntot, h->rl_npj, ir.data());
        for (int q = 0; q < h->Nc; ++q) {
            rowlane_image(h->Hsym.data() + q * nn, h->Ntot, h->rl_npj, ir.data() + (size_t)(1 + q) * h->rl_stride);
            rowlane_image(h->Hanti.data() + q * nn, h->Ntot, h->rl_npj, ir.data() + (size_t)(1 + h->Nc + q) * h->rl_stride);
        }
        HIPCHK(h, hipMemcpy(h->d_himg_r, ir.data(), ir.size() * sizeof(double), hipMemcpyHostToDevice));
        std::vector<double> cr((size_t)2 * h->Nc * h->rl_stride);
        for (int q = 0; q < h->Nc; ++q) {
            std::copy_n(ir.data() + (size_t)(1 + q) * h->rl_stride, h->rl_stride, cr.data() + cslot(q, false) * h->rl_stride);
            std::copy_n(ir.data() + (size_t)(1 + h->Nc + q) * h->rl_stride, h->rl_stride, cr.data() + cslot(q, true) * h->rl_stride);
        }
        HIPCHK(h, hipMemcpy(h->d_cimg_r, cr.data(), cr.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    return JQ_OK;
}

static int upload_targets(jq_handle* h)
{
    std::vector<double> img((size_t)h->parts * h->KT * 64);
    slab_image(h->Utr.data(), h->Ntot, h->N, h->sps, h->KT, img.data(), h->parts);
    HIPCHK(h, hipMemcpy(h->d_vtr, img.data(), img.size() * sizeof(double), hipMemcpyHostToDevice));
    slab_image(h->Uti.data(), h->Ntot, h->N, h->sps, h->KT, img.data(), h->parts);
    HIPCHK(h, hipMemcpy(h->d_vti, img.data(), img.size() * sizeof(double), hipMemcpyHostToDevice));
    if (h->lane_np > 0) {
        std::vector<double> cl((size_t)h->N * h->lane_np, 0.0);
        column_image(h->Utr.data(), h->Ntot, h->N, h->lane_np, cl.data());
        HIPCHK(h, hipMemcpy(h->d_vtr_l, cl.data(), cl.size() * sizeof(double), hipMemcpyHostToDevice));
        std::fill(cl.begin(), cl.end(), 0.0);
        column_image(h->Uti.data(), h->Ntot, h->N, h->lane_np, cl.data());
        HIPCHK(h, hipMemcpy(h->d_vti_l, cl.data(), cl.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    if (h->rl_npj > 0) {
        std::vector<double> cl((size_t)h->N * 16, 0.0);
        column_image(h->Utr.data(), h->Ntot, h->N, 16, cl.data());
        HIPCHK(h, hipMemcpy(h->d_vtr_r, cl.data(), cl.size() * sizeof(double), hipMemcpyHostToDevice));
        std::fill(cl.begin(), cl.end(), 0.0);
        column_image(h->Uti.data(), h->Ntot, h->N, 16, cl.data());
        HIPCHK(h, hipMemcpy(h->d_vti_r, cl.data(), cl.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    return JQ_OK;
}

extern "C" int jq_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int jq_set_device(int device) { return hipSetDevice(device) == hipSuccess ? JQ_OK : JQ_EHIP; }

// experiment builds (scripts/exp_variants.sh) link one object that defines jq_variant_tag: their version string -- and with it the
// build identity bench.py compares with profiles/ -- differs from the production build's although host.o is shared
extern "C" __attribute__((weak)) const char jq_variant_tag[];
extern "C" const char* jq_version(void)
{
    static const std::string v = std::string(JQ_VERSION) + (jq_variant_tag ? std::string("+") + jq_variant_tag : std::string());
    return v.c_str();
}

extern "C" int jq_abi_version(void) { return JQ_ABI_VERSION; }

extern "C" const char* jq_last_error(const jq_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

static void destroy_multi(jq_handle* h);

extern "C" void jq_destroy(jq_handle* h)
{
    if (!h) return;
    if (!h->subs.empty()) {
        destroy_multi(h);
        return;
    }
    (void)hipSetDevice(h->device);
    if (h->emb) jq_destroy(h->emb);
    double** bufs[] = {&h->d_cq3, &h->d_qsplit, &h->d_wlr, &h->d_cimg_l, &h->d_cimg_r, &h->d_rfreq, &h->d_wq, &h->d_pk2, &h->d_pack, &h->d_himg_r, &h->d_uinit_r, &h->d_vtr_r, &h->d_vti_r, &h->d_himg_l, &h->d_uinit_l, &h->d_vtr_l, &h->d_vti_l, &h->d_himg_c, &h->d_cimg_c, &h->d_park, &h->d_cimg, &h->d_himg,  &h->d_uimg,       &h->d_vtr,     &h->d_vti,    &h->d_tabs, &h->d_tf,   &h->d_tb,
                       &h->d_cfreq, &h->d_pcof,       &h->d_stream,  &h->d_pq,     &h->d_state, &h->d_state_save,
                       &h->d_colinfo, &h->d_traces,   &h->d_R,       &h->d_grad,   &h->d_res};
    for (auto b : bufs)
        if (*b) (void)hipFree(*b);
    for (auto e : h->ev) (void)hipEventDestroy(e);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

// LDS bytes of the backward slab / quad kernels (jq_kernels.h k_backward) behind the operator staging: the tables wd, ws,
// the per-thread trace carries [Nc][threads], the parking images (park_doubles per wave; 0: parked in HBM) and the
// double-buffered per-step trace records [2][waves][8 Nc]
static long long bwd_lds_tail(int NT, int Nc, int nwaves, long long park_doubles)
{
    return 32LL * NT * 8 + (long long)Nc * 64 * nwaves * 8 + (long long)nwaves * park_doubles * 8 + 2LL * nwaves * 8 * Nc * 8;
}

// Dense column-major copy of a sparse operator in Julia's SparseMatrixCSC form (jq_csc: 1-based Int64 colptr / rowval); repeated
// entries are summed.  The planner then sees exactly the structure it would see for the dense form of the same operator.
static int csc_to_dense(jq_handle* h, const jq_csc* A, int Ntot, double* out, const char* what)
{
    // (jq_csc carries no nnz field -- like SparseMatrixCSC, whose extent is colptr[n + 1] - 1: the sizes are checked BEFORE colptr is
    //  indexed with them, colptr is checked entry by entry before rowval / nzval are read, and at most 4 Ntot^2 entries are accepted)
    char buf[200];
    if (!A || !A->colptr) {
        snprintf(buf, sizeof buf, "%s: NULL sparse descriptor or colptr", what);
        return fail(h, JQ_EINVAL, buf);
    }
    if (A->m != Ntot || A->n != Ntot) {
        snprintf(buf, sizeof buf, "%s: sparse operator is %lld x %lld, expected %d x %d", what, (long long)A->m, (long long)A->n, Ntot, Ntot);
        return fail(h, JQ_EINVAL, buf);
    }
    if (A->colptr[0] != 1) {
        snprintf(buf, sizeof buf, "%s: colptr[1] must be 1 (1-based SparseMatrixCSC fields)", what);
        return fail(h, JQ_EINVAL, buf);
    }
    for (int j = 0; j < Ntot; ++j)
        if (A->colptr[j + 1] < A->colptr[j] || A->colptr[j + 1] - 1 > (int64_t)Ntot * Ntot * 4) {
            snprintf(buf, sizeof buf, "%s: colptr is not non-decreasing (or names more than 4 Ntot^2 entries)", what);
            return fail(h, JQ_EINVAL, buf);
        }
    if (A->colptr[Ntot] > 1 && (!A->rowval || !A->nzval)) {
        snprintf(buf, sizeof buf, "%s: NULL rowval / nzval array", what);
        return fail(h, JQ_EINVAL, buf);
    }
    std::fill(out, out + (size_t)Ntot * Ntot, 0.0);
    for (int j = 0; j < Ntot; ++j) {
        const int64_t b = A->colptr[j], e = A->colptr[j + 1];
        for (int64_t k = b - 1; k < e - 1; ++k) {
            const int64_t r = A->rowval[k];
            if (r < 1 || r > Ntot) {
                snprintf(buf, sizeof buf, "%s: rowval out of range (1 .. Ntot)", what);
                return fail(h, JQ_EINVAL, buf);
            }
            out[(r - 1) + (size_t)Ntot * j] += A->nzval[k];
        }
    }
    return JQ_OK;
}

static int create_dense(const jq_problem* p, jq_handle* h);

// Sparse storage (jq_problem::Hconst_csc / Hsym_csc / Hanti_csc) is turned into the dense form first; everything else -- planning from
// the nonzero structure, images, kernels -- is one code path.
static int create_impl(const jq_problem* p, jq_handle* h)
{
    if (!p) return fail(h, JQ_EINVAL, "jq_create: problem is NULL");
    const bool sparse = (!p->Hconst && p->Hconst_csc) || (!p->Hsym_ops && p->Hsym_csc) || (!p->Hanti_ops && p->Hanti_csc);
    if (!sparse) return create_dense(p, h);
    if (p->Ntot < 1 || p->Ntot > 16384 || p->Ncoupled < 0 || p->Ncoupled > 4096) return create_dense(p, h);      // (its messages)
    const size_t nn = (size_t)p->Ntot * p->Ntot;
    std::vector<double> H0, Hs, Ha;
    jq_problem q = *p;
    q.Hconst_csc = q.Hsym_csc = q.Hanti_csc = nullptr;
    int rc;
    if (!p->Hconst && p->Hconst_csc) {
        H0.resize(nn);
        if ((rc = csc_to_dense(h, p->Hconst_csc, p->Ntot, H0.data(), "jq_create: Hconst_csc"))) return rc;
        q.Hconst = H0.data();
    }
    if (!p->Hsym_ops && p->Hsym_csc) {
        Hs.resize(nn * std::max(p->Ncoupled, 1));
        for (int k = 0; k < p->Ncoupled; ++k)
            if ((rc = csc_to_dense(h, p->Hsym_csc + k, p->Ntot, Hs.data() + nn * k, "jq_create: Hsym_csc"))) return rc;
        q.Hsym_ops = Hs.data();
    }
    if (!p->Hanti_ops && p->Hanti_csc) {
        Ha.resize(nn * std::max(p->Ncoupled, 1));
        for (int k = 0; k < p->Ncoupled; ++k)
            if ((rc = csc_to_dense(h, p->Hanti_csc + k, p->Ntot, Ha.data() + nn * k, "jq_create: Hanti_csc"))) return rc;
        q.Hanti_ops = Ha.data();
    }
    return create_dense(&q, h);
}

static int create_dense(const jq_problem* p, jq_handle* h)
{
    if (!p) return fail(h, JQ_EINVAL, "jq_create: problem is NULL");
    if (p->Ntot < 1 || p->N < 1 || p->N > p->Ntot) return fail(h, JQ_EINVAL, "jq_create: need 1 <= N <= Ntot");
    if (p->nsteps < 1 || !(p->T > 0.0)) return fail(h, JQ_EINVAL, "jq_create: need nsteps >= 1 and T > 0");
    if (p->Nfreq < 1) return fail(h, JQ_EINVAL, "jq_create: need Nfreq >= 1");
    if (p->neumann_terms < 0) return fail(h, JQ_EINVAL, "jq_create: neumann_terms must be >= 0");
    if (p->Nunc < 0) return fail(h, JQ_EINVAL, "jq_create: Nunc must be >= 0");
    if (p->Nunc > 0 && p->Ncoupled != 0)      // @assert(Ncoupled==0 || Nunc==0), src/evalobjgrad.jl:176
        return fail(h, JQ_EINVAL, "jq_create: coupled and uncoupled controls cannot be combined (Ncoupled == 0 || Nunc == 0)");
    if (!p->Hconst || !p->Uinit || !p->Utarget_r || !p->Utarget_i || !p->wmat_real_diag || !p->Cfreq ||
        (p->Nunc == 0 && (!p->Hsym_ops || !p->Hanti_ops)) || (p->Nunc > 0 && (!p->Hunc_ops || !p->Rfreq)))
        return fail(h, JQ_EINVAL, "jq_create: NULL array in problem description");
    const int nctrl = p->Nunc > 0 ? p->Nunc : p->Ncoupled;     // control pairs the kernels see
    if (nctrl < 1) return fail(h, JQ_EUNSUPPORTED, "jq_create: at least one control Hamiltonian is required");
    // (sanity bounds, not design limits: an Ntot x Ntot fp64 operator set of this size would not fit the device anyway)
    if (nctrl > 4096) return fail(h, JQ_EINVAL, "jq_create: more than 4096 control Hamiltonians");
    if (p->Ntot > 16384) return fail(h, JQ_EINVAL, "jq_create: Ntot > 16384");
    if (p->objFuncType < 1 || p->objFuncType > 3) return fail(h, JQ_EINVAL, "jq_create: objFuncType must be 1, 2 or 3");

    HIPCHK(h, hipGetDevice(&h->device));
    hipDeviceProp_t prop;
    HIPCHK(h, hipGetDeviceProperties(&prop, h->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        char buf[256];
        snprintf(buf, sizeof buf, "jq_create: device arch '%s' is not gfx950 (this library is MI355X-only)", prop.gcnArchName);
        return fail(h, JQ_EUNSUPPORTED, buf);
    }
    HIPCHK(h, hipStreamCreate(&h->stream));

    h->Ntot = p->Ntot; h->N = p->N; h->Nc = nctrl; h->NcK = std::min(nctrl, JQ_MAXNC); h->Nfreq = p->Nfreq; h->nsteps = p->nsteps;
    h->m = p->neumann_terms; h->objFuncType = p->objFuncType; h->T = p->T;
    h->NT = (p->Ntot + 15) / 16;
    h->big = h->NT > 6;
    h->huge = h->NT > 16;
    h->bw_trace.assign(nctrl, 0);
    h->KT = 4 * h->NT;
    h->NP = 16 * h->NT;
    h->parts = p->N > 16 ? (p->N + 15) / 16 : 1;
    h->sps = p->N > 16 ? 1 : 16 / p->N;
    h->state_stride = (long long)(JQ_STATE_ARRAYS * h->KT + JQ_STATE_EXTRA) * 64;
    const size_t nn = (size_t)p->Ntot * p->Ntot, nc = (size_t)p->Ntot * p->N;
    h->Hconst.assign(p->Hconst, p->Hconst + nn);
    if (p->Nunc > 0) {
        // Uncoupled controls (src/evalobjgrad.jl:2373-2387): Hunc_ops[q] takes the symmetric slot of control pair q when it
        // is symmetric (its term goes to K), the antisymmetric slot when it is antisymmetric (-> S); the other slot is
        // zero and k_ctrl feeds both with ft_q(t) = 2 (p cos(2 pi Rfreq t) - q sin(2 pi Rfreq t)).  isSymm: :186-196.
        h->Hsym.assign(nn * nctrl, 0.0);
        h->Hanti.assign(nn * nctrl, 0.0);
        h->rfreq.assign(p->Rfreq, p->Rfreq + nctrl);
        for (int q = 0; q < nctrl; ++q) {
            const double* M = p->Hunc_ops + nn * q;
            bool sym = true;
            double nrm2 = 0.0;
            for (int c = 0; c < p->Ntot; ++c)
                for (int r = 0; r < p->Ntot; ++r) {
                    const double a = M[r + (size_t)p->Ntot * c], b = M[c + (size_t)p->Ntot * r];
                    if (a != b) sym = false;
                    nrm2 += (a + b) * (a + b);
                }
            if (!sym && !(std::sqrt(nrm2) < 1e-15))
                return fail(h, JQ_EINVAL, "jq_create: Uncoupled Hamiltonian is not symmetric or anti-symmetric. This functionality is "
                                          "not currently supported.");
            std::copy(M, M + nn, (sym ? h->Hsym.begin() : h->Hanti.begin()) + nn * q);
        }
    } else {
        h->Hsym.assign(p->Hsym_ops, p->Hsym_ops + nn * p->Ncoupled);
        h->Hanti.assign(p->Hanti_ops, p->Hanti_ops + nn * p->Ncoupled);
    }
    h->Uinit.assign(p->Uinit, p->Uinit + nc);
    h->Utr.assign(p->Utarget_r, p->Utarget_r + nc);
    h->Uti.assign(p->Utarget_i, p->Utarget_i + nc);
    h->wd.assign(p->wmat_real_diag, p->wmat_real_diag + p->Ntot);
    h->cfreq.assign(p->Cfreq, p->Cfreq + (size_t)nctrl * p->Nfreq);

    // 4 x 4 x n structure with n = 7, 8 (Ntot 97 .. 128, e.g. cnot3 with more guard levels): the JQ_BW_T4 slab kernels and the
    // quad-layout kernels are instantiated for it -- such a handle is not "big" (no cooperative kernels, no cooperative-quad ones:
    // their LDS images do not fit).  option t4big=0: treat it like any other Ntot > 96.
    if (h->big && h->NT <= 8 && !h->force_plain) {
        bool t4 = block_band(h->Hconst.data(), h->Ntot) <= 1 && t4_structure(h->Hconst.data(), h->Ntot);
        for (int q = 0; q < h->Nc && t4; ++q)
            t4 = block_band(h->Hsym.data() + q * nn, h->Ntot) <= 1 && block_band(h->Hanti.data() + q * nn, h->Ntot) <= 1 &&
                 t4_structure(h->Hsym.data() + q * nn, h->Ntot) && t4_structure(h->Hanti.data() + q * nn, h->Ntot);
        if (!h->opt.on(O_T4) || !h->opt.on(O_OD) || !h->opt.on(O_T4BIG) || h->opt.on(O_FORCE_DENSE)) t4 = false;
        if (t4) h->big = false;
    }

    // block-band structure (16x16 blocks) of the operators: kernels exist for BW in {0,1,2,NT-1}
    {
        int bw = block_band(h->Hconst.data(), h->Ntot);
        for (int q = 0; q < h->Nc; ++q) {
            const int bq = std::max(block_band(h->Hsym.data() + q * nn, h->Ntot), block_band(h->Hanti.data() + q * nn, h->Ntot));
            h->bw_trace[q] = bq;
            bw = std::max(bw, bq);
        }
        if (h->opt.on(O_FORCE_DENSE)) bw = h->NT - 1;
        h->BW = (bw <= 2 && bw < h->NT - 1) ? bw : h->NT - 1;
        if (h->big && h->BW == 0) h->BW = 1;      // (the big variants are instantiated for block bands 1, 2 and dense)
        // dense at this size = band code 15 (a full window for every NT <= 16): NT - 1 = 7, 8, 9 are the codes of the quad-layout,
        // JQ_BW_T4 and JQ_BW_OD structures -- round 2 instantiated <10, 9> as "dense" and got the JQ_BW_OD product (wrong results
        // for dense operators with Ntot 145 .. 160; found by the round-3 tests)
        if (h->big && h->BW > 2) h->BW = 15;
        if (h->huge) h->BW = 15;      // (the run-time-size kernels know dense windows only)
        h->BWc = h->BW;
        // block tridiagonal with DIAGONAL off-diagonal blocks (operators of the slowest subsystem, cnot3):
        // MFMA only for the diagonal blocks, 16 coefficients per off-diagonal block (option od=0 disables)
        bool od = (!h->big && bw == 1 && h->NT >= 2 && offdiag_blocks_diagonal(h->Hconst.data(), h->Ntot));
        for (int q = 0; q < h->Nc && od; ++q)
            od = offdiag_blocks_diagonal(h->Hsym.data() + q * nn, h->Ntot) && offdiag_blocks_diagonal(h->Hanti.data() + q * nn, h->Ntot);
        if (!h->opt.on(O_OD) || h->opt.on(O_FORCE_DENSE)) od = false;
        if (od) h->BW = h->BWc = JQ_BW_OD;
        // ... and, one level finer, 4x4 diagonal blocks + diagonal couplings of neighbouring 4-row groups: the slab kernels
        // use v_mfma_f64_4x4x4 (JQ_BW_T4; option t4=0 disables); the cooperative kernels stay on the JQ_BW_OD variant
        bool t4 = !h->big && (bw <= 1) && t4_structure(h->Hconst.data(), h->Ntot);
        for (int q = 0; q < h->Nc && t4; ++q)
            t4 = t4_structure(h->Hsym.data() + q * nn, h->Ntot) && t4_structure(h->Hanti.data() + q * nn, h->Ntot);
        if (!h->opt.on(O_T4) || !h->opt.on(O_OD) || h->opt.on(O_FORCE_DENSE) || h->force_plain) t4 = false;
        if (t4) h->BW = JQ_BW_T4;
        // trace image layout per control: 0 block diagonal, 1 band BW, 2 band BW without the diagonal blocks
        for (int q = 0; q < h->Nc && h->BW == JQ_BW_T4; ++q)
            h->bw_trace[q] = t4_mode(h->Hsym.data() + q * nn, h->Ntot) | t4_mode(h->Hanti.data() + q * nn, h->Ntot);
        for (int q = 0; q < h->Nc && h->BW != JQ_BW_T4; ++q) {
            if (h->bw_trace[q] == 0 || h->BW == 0)
                h->bw_trace[q] = (h->BW == 0) ? 1 : 0;
            else
                h->bw_trace[q] = (h->NT > 1 && diag_blocks_zero(h->Hsym.data() + q * nn, h->Ntot) &&
                                  diag_blocks_zero(h->Hanti.data() + q * nn, h->Ntot)) ? 2 : 1;
        }
        h->mat_elems = (((h->BW == JQ_BW_T4 ? (long long)JQ_T4_ELEMS(h->NT)
                                             : 64LL * band_tiles(h->NT, h->BW) + (h->BW == JQ_BW_OD ? JQ_OD_COEFS(h->NT) : 0)) + 127) / 128) * 128;
        const long long slot = h->mat_elems * 8;
        const long long lds_fwd_fixed = (long long)32 * h->NT * 8;
        const long long lds_bwd_fixed = bwd_lds_tail(h->NT, h->NcK, JQ_WAVES, 0);
        const long long park_bytes = (long long)JQ_WAVES * h->KT * 64 * 8;
        if (h->big) {
            h->mat_elems = 128;       // (no slab-kernel images: placeholders)
            // Only the cooperative kernels (band BWc) exist at this size.  BW must not keep a dense band NT - 1 that happens to
            // equal one of the structure codes (NT = 8, 9, 10: 7 = JQ_BW_T4Q, 8 = JQ_BW_T4, 9 = JQ_BW_OD) -- round 2 sent dense
            // problems with Ntot 113 .. 160 to kernel families that do not exist for them (found by the round-3 tests)
            h->BW = -1;
        } else if (2 * slot + lds_fwd_fixed > 163840)
            return fail(h, JQ_EUNSUPPORTED, "jq_create: operator images do not fit the LDS double buffer");
        h->nslots = 2;
        h->nslots_bwd = 2;
        h->park_lds = (2 * slot + lds_bwd_fixed + park_bytes <= 163840) ? 1 : 0;
        // cooperative (row-split) kernels for small batches: NT waves per slab, needs NT >= 2
        // (NT == 1: only the implicit-midpoint kernels are instantiated -- Ntot <= 16 with more than four columns per evaluation)
        h->mat_elems_c = 0;
        if (h->NT > 6) {      // (more than six tile rows: the HBM-operand variants, instantiated for the bands 1, 2 and dense = 15;
            //  also for the 4 x 4 x 7 / 4 x 4 x 8 structures -- round 3: their fallback when the quad-layout kernels do not apply,
            //  e.g. implicit midpoint with N = 3)
            if (h->BWc == 0) h->BWc = 1;
            if (h->BWc > 2) h->BWc = 15;
        }
        if (h->NT >= 2 || h->N > 4) {
            const long long ec = (((long long)h->NT * coop_row_elems(h->NT, h->BWc) + 127) / 128) * 128;
            const long long lds_c = (h->NT > 6 ? 0 : 2 * ec * 8) + lds_fwd_fixed + 2LL * h->KT * 64 * 8 + 16LL * h->NT * 8;      // (operator slots, tables, x exchange, Jacobi column norms)
            h->mat_elems_c = ec;                  // (the images are built whenever the layout exists ...)
            // (... the Stormer-Verlet kernels need two of them in LDS -- or none: NT > 6; the 4 x 4 x 7 / 4 x 4 x 8 structures keep
            //  their JQ_BW_T4 slab kernels as the Stormer-Verlet fallback: the cooperative layout serves their implicit-midpoint path)
            h->coop_ok = lds_c <= 163840 && (h->NT <= 6 || h->big);
            if (h->huge) h->coop_ok = true;      // (static LDS only)
        }
        h->coop_max_slabs = prop.multiProcessorCount;   // one cooperative workgroup per CU = one round
        if (h->opt.has(O_COOP_MAX)) h->coop_max_slabs = (int)h->opt.get(O_COOP_MAX);
        if (h->big) h->coop_max_slabs = 1 << 30;        // the only kernel family at this size
        // Batched staging (K/S images of B time steps per DMA burst, constants resident in LDS) exists for
        // small images but is OFF by default: measured on MI355X it does not help (swap02/cnot2: the
        // ~600-cycle dependent-product latency dominates, not the per-operator barrier) and its 150 KB of
        // LDS allow only one workgroup per CU.  option batch=<B> enables it for experiments.
        h->batch = 0;
        // Window staging (jq_kernels.h Ring, batch < 0): five time points (K and S image each) and the constant trace images
        // resident in LDS, one workgroup barrier per time step.  Used whenever it fits next to the backward kernel's carry
        // and parking images (kernels compiled for two workgroups per CU: in half of the LDS); option window=0 disables it.
        {
            const long long win = (2LL * JQ_WIN_TPS + 2LL * h->NcK) * slot;
            const long long budget = (h->NT <= JQ_MINW_MAXNT) ? 81920 : 163840;
            bool w = !h->big && win + lds_bwd_fixed + park_bytes <= budget;
            if (!h->opt.on(O_WINDOW)) w = false;
            if (w) {
                h->batch = -1;
                h->park_lds = 1;
            }
        }
        // Quad-layout kernels (jq_kernels.h JQ_BW_T4Q) for this structure: workgroups of 4, 8 or 12 waves carry 1, 2 or 3 slabs
        // (1, 2, 3 waves per SIMD; one workgroup per CU because of the LDS).  run_eval picks the variant -- or the slab
        // kernels -- by the number of rounds the batch needs (quad_plan).  option quad=0 disables them, option quad=<n> limits them to
        // batches of at most n slabs.
        // (they always use the window staging and need less LDS next to it than the slab kernels -- a register per 16-row block
        // to park -- so they are also available when the slab kernels have to fall back to the per-operator ring: Ntot > 80, Nc = 4)
        {
            const long long win = (2LL * JQ_WIN_TPS + 2LL * h->NcK) * slot;
            const long long quad_fixed = bwd_lds_tail(h->NT, h->NcK, JQ_WAVES, (long long)h->NT * 64);
            bool w = h->BW == JQ_BW_T4 && win + quad_fixed <= 163840;
            if (!h->opt.on(O_WINDOW)) w = false;
            h->quad_max_slabs = w ? (1 << 30) : 0;
        }
        h->num_cu = prop.multiProcessorCount;
        if (h->opt.has(O_QUAD) && h->quad_max_slabs > 0) h->quad_max_slabs = (int)h->opt.get(O_QUAD);
        // Cooperative-quad kernels (jq_cq_kernels.h): the latency path -- one workgroup of NT waves per column quad while every
        // quad still gets a CU of its own (LDS: the window staging, one workgroup per CU).  NT >= 2 (a single block has no
        // neighbour to split the work with).  option cq=0 disables them, option cq=<n> bounds the number of quads.
        {
            const long long win = (2LL * JQ_WIN_TPS + 2LL * h->NcK) * slot;
            const long long tail = 32LL * h->NT * 8 + 6LL * (h->NT + 2) * 64 * 8 + (long long)std::max(2, h->NcK + (h->NcK + 1) / 2) * h->NT * 64 * 8;      // (run_eval: lds_cq)
            h->cq_max_quads = (h->BW == JQ_BW_T4 && h->NT >= 2 && h->NT <= 7 && h->quad_max_slabs > 0 && win + tail <= 163840) ? 2 * prop.multiProcessorCount : 0;      // (two rounds of them, 2 x 0.20 s at cnot3, still beat one round of the quad-layout kernels, 0.55 s)
            if (h->opt.has(O_CQ) && h->cq_max_quads > 0) h->cq_max_quads = (int)h->opt.get(O_CQ);
        }
        if (h->opt.has(O_BATCH)) {      // (experiment builds only: jq_options.h)
            const int v = (int)h->opt.get(O_BATCH);
            if (v >= 2 && slot <= 8192) {
                const long long fixed = lds_bwd_fixed + park_bytes + 2LL * h->NcK * slot;
                const long long per_buf = (163840 - fixed) / 2;
                long long B = (per_buf / (2 * slot) - 1) / 2;
                if (B > v) B = v;
                if (B >= 2) {
                    h->batch = (int)B;
                    h->park_lds = 1;
                }
            }
        }
    }

    // Lane kernels (jq_lane_kernels.h) for small Hilbert spaces: one lane per state column, operator images in
    // VGPRs read through DPP row broadcasts.  Instantiated for NP in {2,4,6,8}.  option lane=0 disables them,
    // option lane_min / option lane_max bound the column counts (samples x N) they are used for.
    h->lane_np = 0;
    {
        static const int nps[] = {2, 4, 6, 8};
        for (int v : nps)
            if (h->Ntot <= v) {
                h->lane_np = v;
                break;
            }
        if (!h->opt.on(O_LANE)) h->lane_np = 0;
        h->lane_stride = ((long long)h->lane_np * h->lane_np + 15) / 16 * 16;
        h->lane_min_cols = 1;
        h->lane_max_cols = 1 << 30;
        // row-lane kernels (jq_rowlane_kernels.h): same sizes, one lane per (row, column), 4 columns per wave;
        // used while the batch is small enough that the evaluation is bound by the latency of one wave
        // (measured cross-over with the lane kernels, scripts/time_cases.py).  option rowlane_max overrides.
        h->rl_npj = h->Ntot <= 8 ? (h->Ntot + 1) / 2 * 2 : (h->Ntot <= 12 ? 12 : (h->Ntot <= 16 ? 16 : 0));
        if (!h->opt.on(O_LANE)) h->rl_npj = 0;
        if (h->is_emb) h->rl_npj = 0, h->lane_np = 0;      // an embedded twin only serves the JQ_BW_T4 / quad-layout families
        h->rl_stride = 16LL * h->rl_npj;
        // cross-over measured with scripts/time_cases.py: ~2 waves per SIMD against the lane kernels (Ntot <= 8),
        // ~4 against the MFMA slab kernels (Ntot 9..16)
        h->rl_max_cols = 2 * 4 * 4 * prop.multiProcessorCount;      // (round 3: also for NPJ = 12, 16 -- cnot2 x 4 096 samples 94 ms here, 61 ms on the MFMA kernels)
        if (h->opt.has(O_ROWLANE_MAX)) h->rl_max_cols = (int)h->opt.get(O_ROWLANE_MAX);
        if (h->opt.has(O_LANE_MIN)) h->lane_min_cols = (int)h->opt.get(O_LANE_MIN);
        if (h->opt.has(O_LANE_MAX)) h->lane_max_cols = (int)h->opt.get(O_LANE_MAX);
    }

    // time tables, accumulated exactly like the reference: t = t + h (src/StormerVerlet.jl:502);
    // the backward sweep restarts from exactly T with h = -dt (src/evalobjgrad.jl:811-812)
    const double dt = h->T / h->nsteps;
    h->tf.resize(h->nsteps + 1);
    h->tb.resize(h->nsteps + 1);
    double t = 0.0;
    for (int n = 0; n <= h->nsteps; ++n) {
        h->tf[n] = t;
        t = t + dt;
    }
    t = h->T;
    for (int n = 0; n <= h->nsteps; ++n) {
        h->tb[n] = t;
        t = t + (-dt);
    }

    int rc;
    if ((rc = dev_alloc(h, &h->d_himg, (size_t)(1 + 2 * h->Nc) * h->mat_elems))) return rc;
    if ((rc = dev_alloc(h, &h->d_cimg, (size_t)(2 * h->Nc) * h->mat_elems))) return rc;
    if (h->mat_elems_c > 0) {
        if ((rc = dev_alloc(h, &h->d_himg_c, (size_t)(1 + 2 * h->Nc) * h->mat_elems_c))) return rc;
        if ((rc = dev_alloc(h, &h->d_cimg_c, (size_t)(2 * h->Nc) * h->mat_elems_c))) return rc;
    }
    if (h->lane_np > 0) {
        if ((rc = dev_alloc(h, &h->d_himg_l, (size_t)(1 + 2 * h->Nc) * h->lane_stride))) return rc;
        if ((rc = dev_alloc(h, &h->d_cimg_l, (size_t)(2 * h->Nc) * h->lane_stride))) return rc;
        if ((rc = dev_alloc(h, &h->d_uinit_l, (size_t)h->N * h->lane_np))) return rc;
        if ((rc = dev_alloc(h, &h->d_vtr_l, (size_t)h->N * h->lane_np))) return rc;
        if ((rc = dev_alloc(h, &h->d_vti_l, (size_t)h->N * h->lane_np))) return rc;
        std::vector<double> cl((size_t)h->N * h->lane_np, 0.0);
        column_image(h->Uinit.data(), h->Ntot, h->N, h->lane_np, cl.data());
        HIPCHK(h, hipMemcpy(h->d_uinit_l, cl.data(), cl.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    if (h->rl_npj > 0) {
        if ((rc = dev_alloc(h, &h->d_himg_r, (size_t)(1 + 2 * h->Nc) * h->rl_stride))) return rc;
        if ((rc = dev_alloc(h, &h->d_cimg_r, (size_t)(2 * h->Nc) * h->rl_stride))) return rc;
        if ((rc = dev_alloc(h, &h->d_uinit_r, (size_t)h->N * 16))) return rc;
        if ((rc = dev_alloc(h, &h->d_vtr_r, (size_t)h->N * 16))) return rc;
        if ((rc = dev_alloc(h, &h->d_vti_r, (size_t)h->N * 16))) return rc;
        std::vector<double> cl((size_t)h->N * 16, 0.0);
        column_image(h->Uinit.data(), h->Ntot, h->N, 16, cl.data());
        HIPCHK(h, hipMemcpy(h->d_uinit_r, cl.data(), cl.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    if ((rc = dev_alloc(h, &h->d_uimg, (size_t)h->parts * h->KT * 64))) return rc;
    if ((rc = dev_alloc(h, &h->d_vtr, (size_t)h->parts * h->KT * 64))) return rc;
    if ((rc = dev_alloc(h, &h->d_vti, (size_t)h->parts * h->KT * 64))) return rc;
    if ((rc = dev_alloc(h, &h->d_tabs, (size_t)32 * h->NT))) return rc;
    if ((rc = dev_alloc(h, &h->d_tf, (size_t)h->nsteps + 1))) return rc;
    if ((rc = dev_alloc(h, &h->d_tb, (size_t)h->nsteps + 1))) return rc;
    if ((rc = dev_alloc(h, &h->d_cfreq, h->cfreq.size()))) return rc;
    HIPCHK(h, hipMemcpy(h->d_tf, h->tf.data(), h->tf.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->d_tb, h->tb.data(), h->tb.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->d_cfreq, h->cfreq.data(), h->cfreq.size() * sizeof(double), hipMemcpyHostToDevice));
    if (!h->rfreq.empty()) {
        if ((rc = dev_alloc(h, &h->d_rfreq, h->rfreq.size()))) return rc;
        HIPCHK(h, hipMemcpy(h->d_rfreq, h->rfreq.data(), h->rfreq.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    if ((rc = upload_operators(h))) return rc;
    if ((rc = upload_targets(h))) return rc;
    {
        std::vector<double> img((size_t)h->parts * h->KT * 64);
        slab_image(h->Uinit.data(), h->Ntot, h->N, h->sps, h->KT, img.data(), h->parts);
        HIPCHK(h, hipMemcpy(h->d_uimg, img.data(), img.size() * sizeof(double), hipMemcpyHostToDevice));
    }

    // chunking of the time loop: the tile stream of one chunk has (2*cs+1) time points x {K,S}
    size_t budget = (size_t)1 << 30;
    if (h->opt.has(O_STREAM_BYTES) && h->opt.get(O_STREAM_BYTES) > 0) budget = (size_t)h->opt.get(O_STREAM_BYTES);
    // largest operator image of any kernel family the handle may use (slab, cooperative, lane, row-lane)
    const long long img_elems = std::max(std::max(h->mat_elems, h->mat_elems_c), std::max(h->lane_stride, h->rl_stride));
    const size_t per_tp = 2 * (size_t)img_elems * sizeof(double);
    long long cs = ((long long)(budget / per_tp) - 1) / 2;
    cs = std::max<long long>(1, std::min<long long>(cs, h->nsteps));
    if (h->opt.has(O_CHUNK_STEPS) && h->opt.get(O_CHUNK_STEPS) > 0) cs = std::min<long long>(h->opt.get(O_CHUNK_STEPS), h->nsteps);
    // k_ctrl / k_stream put the 2 cs + 1 time points of a chunk into gridDim.y (limit 65535)
    cs = std::min<long long>(cs, 32767);
    h->chunk_steps = (int)cs;
    if ((rc = dev_alloc(h, &h->d_stream, (size_t)(2 * cs + 1) * 2 * img_elems))) return rc;
    if ((rc = dev_alloc(h, &h->d_pq, (size_t)(2 * cs + 1) * 2 * h->Nc))) return rc;
    if ((rc = dev_alloc(h, &h->d_R, (size_t)cs * h->Nc * JQ_NTR))) return rc;
    return JQ_OK;
}


// ---------------------------------------------------------------------------------------------
// Structure embedding.  The JQ_BW_T4 / quad-layout kernels need operators that are sums of 4x4 diagonal blocks and diagonal
// couplings at the strides 4 (inside a 16-row block) and 16 -- a Kronecker-ordered Hilbert space 4 x 4 x n.  A space
// d1 x d2 x d3 with d1, d2 <= 4 (cnot2: 3 x 4) gets there by zero-padding its two fastest factors to 4 levels: rows and
// columns of the padded levels are zero in every operator, in the initial condition, the target and the leakage weights,
// so those levels stay exactly empty and every result (objective, gradients) is unchanged.  The factorisation is found from
// the operators themselves (the C ABI carries no Ne / Ng): the first (d1, d2) with the fewest 16-row blocks under which
// H0, Hsym_q, Hanti_q all pass t4_structure.
static void embed_matrix(const double* M, int Ntot, const std::vector<int>& row, int NtotE, double* out)
{
    std::fill(out, out + (size_t)NtotE * NtotE, 0.0);
    for (int c = 0; c < Ntot; ++c)
        for (int r = 0; r < Ntot; ++r) out[row[r] + (size_t)NtotE * row[c]] = M[r + (size_t)Ntot * c];
}
static void embed_rows(const double* A, int Ntot, int ncol, const std::vector<int>& row, int NtotE, double* out)
{
    std::fill(out, out + (size_t)NtotE * ncol, 0.0);
    for (int c = 0; c < ncol; ++c)
        for (int r = 0; r < Ntot; ++r) out[row[r] + (size_t)NtotE * c] = A[r + (size_t)Ntot * c];
}

static int try_embed(jq_handle* h, const jq_problem* p)
{
    h->emb_mode = (int)h->opt.get(O_EMBED);
    if (h->is_emb || h->emb_mode == 0 || h->BW == JQ_BW_T4 || h->big || h->Ntot > 96) return JQ_OK;
    if (!h->opt.on(O_T4) || !h->opt.on(O_OD) || h->opt.on(O_FORCE_DENSE)) return JQ_OK;
    if (h->force_plain) return JQ_OK;      // (full weights with the Jacobi solver: the twin's 4 x 4 x n kernels do not combine the two either)
    const int Ntot = h->Ntot, Nc = h->Nc;
    const size_t nn = (size_t)Ntot * Ntot;
    int best_d1 = 0, best_d2 = 0, best_d3 = 1 << 30;
    std::vector<int> row(Ntot);
    std::vector<double> E;
    for (int d1 = 1; d1 <= 4; ++d1)
        for (int d2 = 1; d2 <= 4; ++d2) {
            if (Ntot % (d1 * d2) != 0) continue;
            const int d3 = Ntot / (d1 * d2);
            if (d3 > 8 || d3 >= best_d3) continue;      // (the JQ_BW_T4 families are instantiated for n <= 8)
            // n = 7, 8 (quad-layout kernels with one slab per workgroup only, no / fewer cooperative-quad kernels): worth it when
            // the padding at most doubles the space (measured in round 3, HISTORY.md: 3 x 4 x 7 10 x / 3 x faster for one evaluation /
            // 3 072 samples, 3 x 3 x 8 3.3 x / 1.9 x; 2 x 2 x 8 1.7 x faster / 1.5 x SLOWER)
            if (d3 > 6 && 16 * d3 > 2 * Ntot) continue;
            for (int r = 0; r < Ntot; ++r) row[r] = (r % d1) + 4 * ((r / d1) % d2) + 16 * (r / (d1 * d2));
            const int NE = 16 * d3;
            E.assign((size_t)NE * NE, 0.0);
            bool ok = true;
            auto test = [&](const double* M) {
                embed_matrix(M, Ntot, row, NE, E.data());
                return t4_structure(E.data(), NE);
            };
            ok = test(h->Hconst.data());
            for (int q = 0; q < Nc && ok; ++q) ok = test(h->Hsym.data() + q * nn) && test(h->Hanti.data() + q * nn);
            if (ok) best_d1 = d1, best_d2 = d2, best_d3 = d3;
        }
    if (best_d1 == 0) return JQ_OK;
    const int d1 = best_d1, d2 = best_d2, NE = 16 * best_d3;
    h->emb_row.resize(Ntot);
    for (int r = 0; r < Ntot; ++r) h->emb_row[r] = (r % d1) + 4 * ((r / d1) % d2) + 16 * (r / (d1 * d2));
    // the embedded twin of the problem (coupled controls: Hunc problems were turned into pairs by create_impl already)
    std::vector<double> H0((size_t)NE * NE), Hs((size_t)Nc * NE * NE), Ha((size_t)Nc * NE * NE), U0((size_t)NE * h->N),
        Vr((size_t)NE * h->N), Vi((size_t)NE * h->N), wd(NE);
    embed_matrix(h->Hconst.data(), Ntot, h->emb_row, NE, H0.data());
    for (int q = 0; q < Nc; ++q) {
        embed_matrix(h->Hsym.data() + q * nn, Ntot, h->emb_row, NE, Hs.data() + (size_t)q * NE * NE);
        embed_matrix(h->Hanti.data() + q * nn, Ntot, h->emb_row, NE, Ha.data() + (size_t)q * NE * NE);
    }
    embed_rows(h->Uinit.data(), Ntot, h->N, h->emb_row, NE, U0.data());
    embed_rows(h->Utr.data(), Ntot, h->N, h->emb_row, NE, Vr.data());
    embed_rows(h->Uti.data(), Ntot, h->N, h->emb_row, NE, Vi.data());
    embed_rows(h->wd.data(), Ntot, 1, h->emb_row, NE, wd.data());
    jq_problem q = *p;
    q.Ntot = NE;
    q.Ncoupled = Nc;
    q.Nunc = 0;
    q.Hconst = H0.data(); q.Hsym_ops = Hs.data(); q.Hanti_ops = Ha.data(); q.Uinit = U0.data();
    q.Utarget_r = Vr.data(); q.Utarget_i = Vi.data(); q.wmat_real_diag = wd.data(); q.Cfreq = h->cfreq.data();
    q.Hunc_ops = nullptr; q.Rfreq = nullptr;
    q.Hconst_csc = q.Hsym_csc = q.Hanti_csc = nullptr;
    jq_handle* e = new (std::nothrow) jq_handle();
    if (!e) return fail(h, JQ_ENOMEM, "jq_create: out of host memory");
    e->is_emb = true;
    e->opt = h->opt;
    int rc = create_impl(&q, e);
    if (rc == JQ_OK && e->BW != JQ_BW_T4) rc = JQ_EUNSUPPORTED;      // (cannot happen: the structure test above passed)
    if (rc != JQ_OK) {      // the embedding is an optimisation: without it the handle works as before
        jq_destroy(e);
        h->emb_row.clear();
        return JQ_OK;
    }
    e->rfreq = h->rfreq;      // uncoupled controls: the same ft(t) of k_ctrl
    if (!e->rfreq.empty()) {
        if ((rc = dev_alloc(e, &e->d_rfreq, e->rfreq.size()))) { jq_destroy(e); return rc; }
        HIPCHK(h, hipMemcpy(e->d_rfreq, e->rfreq.data(), e->rfreq.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    h->emb = e;
    return JQ_OK;
}

// the options of a new handle: JQ_OPTIONS (the ONE environment variable that reaches the kernel selection; for callers that cannot pass a
// string), then the caller's string
static int parse_create_options(const char* options, JqOptions* opt)
{
    std::string err;
    if (!opt->parse(getenv("JQ_OPTIONS"), &err)) {
        g_create_error = "JQ_OPTIONS: " + err;
        return JQ_EINVAL;
    }
    if (!opt->parse(options, &err)) {
        g_create_error = "jq_create_opts: " + err;
        return JQ_EINVAL;
    }
    return JQ_OK;
}

static int create_with(const jq_problem* problem, const JqOptions& opt, jq_handle** out)
{
    *out = nullptr;
    jq_handle* h = new (std::nothrow) jq_handle();
    if (!h) {
        g_create_error = "jq_create: out of host memory";
        return JQ_ENOMEM;
    }
    h->opt = opt;
    int rc = create_impl(problem, h);
    if (rc == JQ_OK) rc = try_embed(h, problem);
    if (rc != JQ_OK) {
        g_create_error = h->err;
        jq_destroy(h);
        return rc;
    }
    *out = h;
    return JQ_OK;
}

extern "C" int jq_create_opts(const jq_problem* problem, const char* options, jq_handle** out)
{
    if (!out) {
        g_create_error = "jq_create: out is NULL";
        return JQ_EINVAL;
    }
    *out = nullptr;
    JqOptions opt;
    if (int rc = parse_create_options(options, &opt)) return rc;
    return create_with(problem, opt, out);
}

extern "C" int jq_create(const jq_problem* problem, jq_handle** out) { return jq_create_opts(problem, nullptr, out); }

template <typename F>
static int multi_forall(jq_handle* h, F f);
static int multi_eval_f_g_grad(jq_handle* h, const double* pcof, int ncoeff, const double* nodes, const double* weights, int nquad,
                               const double* shift, bool adjoint, double* out2, double* infid_grad, double* leak_grad);
static int multi_traceobj_sweep(jq_handle* h, const double* pcof, int ncoeff, const double* nodes, int nquad, const double* shift,
                                double* out);
// a single evaluation cannot be sharded: multi-device handles run it on their first device
#define JQ_ON_FIRST(h, call)                         \
    if (!(h)->subs.empty()) {                        \
        DeviceGuard guard_;                          \
        jq_handle* s0_ = (h)->subs[0];               \
        const int rc_ = (call);                      \
        if (rc_ != JQ_OK) (h)->err = s0_->err;       \
        (h)->timing = s0_->timing;                   \
        return rc_;                                  \
    }

extern "C" int jq_set_neumann_terms(jq_handle* h, int32_t m)
{
    if (!h) return JQ_EINVAL;
    if (!h->subs.empty()) return multi_forall(h, [&](jq_handle* sub) { return jq_set_neumann_terms(sub, m); });
    if (m < 0) return fail(h, JQ_EINVAL, "jq_set_neumann_terms: m must be >= 0");
    h->m = m;
    if (h->emb) h->emb->m = m;
    return JQ_OK;
}

static int replan(jq_handle* h, const double* Hconst);
// Full leakage weights WITH the Jacobi solver: the cooperative kernels (two or more tile rows) and the slab kernels <1, 0> / <6, 5> combine
// the two.  A 4 x 4 x n plan reaches neither when it has one tile row (its slab kernels are the JQ_BW_T4 ones) or seven / eight (no
// cooperative layout that fits): such a handle is planned again WITHOUT that structure while the combination is in force, and with it
// again afterwards.
static int ensure_wjac_plan(jq_handle* h)
{
    const bool wjac = h->wrank > 0 && h->solver_id == 2;
    const bool need_plain = wjac && (h->force_plain || (h->BW == JQ_BW_T4 && (h->NT == 1 || h->NT > 6)));
    if (need_plain == h->force_plain) return JQ_OK;
    h->force_plain = need_plain;
    const std::vector<double> H0 = h->Hconst;
    const bool was = h->replanned;
    const int rc = replan(h, H0.data());
    if (rc == JQ_OK) h->replanned = was;
    return rc;
}

extern "C" int jq_set_linear_solver(jq_handle* h, int32_t solver_id, int32_t max_iter, double tol)
{
    if (!h) return JQ_EINVAL;
    if (!h->subs.empty()) return multi_forall(h, [&](jq_handle* sub) { return jq_set_linear_solver(sub, solver_id, max_iter, tol); });
    if (max_iter < 0) return fail(h, JQ_EINVAL, "jq_set_linear_solver: max_iter must be >= 0");
    if (solver_id == 2) {
        if (!(tol > 0.0)) return fail(h, JQ_EINVAL, "jq_set_linear_solver: JACOBI_SOLVER needs tol > 0");
    } else if (solver_id != 1) {
        return fail(h, JQ_EUNSUPPORTED, "jq_set_linear_solver: only NEUMANN_SOLVER (1) and JACOBI_SOLVER (2) are implemented");
    }
    h->solver_id = solver_id;
    h->m = max_iter;
    h->solver_tol = tol;
    if (h->emb) h->emb->solver_id = solver_id, h->emb->m = max_iter, h->emb->solver_tol = tol;
    return ensure_wjac_plan(h);
}

extern "C" int jq_set_integrator(jq_handle* h, int32_t integrator_id, int32_t max_iter, double tol)
{
    if (!h) return JQ_EINVAL;
    if (!h->subs.empty()) return multi_forall(h, [&](jq_handle* sub) { return jq_set_integrator(sub, integrator_id, max_iter, tol); });
    if (integrator_id == 1) {
        h->integrator = 1;
        return JQ_OK;
    }
    if (integrator_id != 2) return fail(h, JQ_EUNSUPPORTED, "jq_set_integrator: 1 = Stormer-Verlet, 2 = implicit midpoint");
    if (max_iter < 1 || !(tol > 0.0)) return fail(h, JQ_EINVAL, "jq_set_integrator: implicit midpoint needs max_iter >= 1 and tol > 0");
    if (h->huge)
        return fail(h, JQ_EUNSUPPORTED, "jq_set_integrator: the implicit-midpoint path is implemented up to Ntot = 256 (the Stormer-Verlet path has no size limit)");
    if (h->wrank > 0)
        return fail(h, JQ_EUNSUPPORTED, "jq_set_integrator: the handle carries full leakage weights (jq_update_wmat); the implicit-midpoint "
                                        "path weights with params.wmat (Diagonal): pass it with jq_update_wmat_diag first");
    // (N > 16 columns per evaluation: one workgroup per evaluation walks over its 16-column parts, jq_coop_imr_kernels.h ImrParts)
    if (h->parts > 1 && h->mat_elems_c == 0)
        return fail(h, JQ_EUNSUPPORTED, "jq_set_integrator: no implicit-midpoint kernels for these operators with N > 16 (no cooperative layout)");
    if (!(h->rl_npj > 0 && h->N <= 4) && h->mat_elems_c == 0 && !(h->quad_max_slabs > 0 && (h->N == 1 || h->N == 2 || h->N == 4)))
        return fail(h, JQ_EUNSUPPORTED, "jq_set_integrator: no implicit-midpoint kernels for these operators (the images of a step "
                                        "do not fit the LDS)");
    h->integrator = 2;
    h->imr_max_iter = max_iter;
    h->imr_tol = tol;
    return JQ_OK;
}

extern "C" int jq_update_target(jq_handle* h, const double* Utr, const double* Uti)
{
    if (!h) return JQ_EINVAL;
    if (!Utr || !Uti) return fail(h, JQ_EINVAL, "jq_update_target: NULL pointer");
    if (!h->subs.empty()) return multi_forall(h, [&](jq_handle* sub) { return jq_update_target(sub, Utr, Uti); });
    HIPCHK(h, hipSetDevice(h->device));
    const size_t nc = (size_t)h->Ntot * h->N;
    h->Utr.assign(Utr, Utr + nc);
    h->Uti.assign(Uti, Uti + nc);
    if (h->emb) {
        jq_handle* e = h->emb;
        embed_rows(Utr, h->Ntot, h->N, h->emb_row, e->Ntot, e->Utr.data());
        embed_rows(Uti, h->Ntot, h->N, h->emb_row, e->Ntot, e->Uti.data());
        const int rc = upload_targets(e);
        if (rc != JQ_OK) {
            h->err = e->err;
            return rc;
        }
    }
    return upload_targets(h);
}

// Re-plan a single-device handle for a new drift Hamiltonian: a fresh plan (create_impl + try_embed) from the handle's own copy
// of the problem, the settings applied since jq_create carried over, then swapped into the caller's handle.
static int replan(jq_handle* h, const double* Hconst)
{
    jq_problem q;
    memset(&q, 0, sizeof q);
    q.Ntot = h->Ntot; q.N = h->N; q.Ncoupled = h->Nc; q.Nfreq = h->Nfreq; q.nsteps = h->nsteps; q.neumann_terms = std::max(h->m, 0);
    q.objFuncType = h->objFuncType; q.Nunc = 0; q.T = h->T;      // (uncoupled controls were turned into pairs by create_impl)
    q.Hconst = Hconst; q.Hsym_ops = h->Hsym.data(); q.Hanti_ops = h->Hanti.data(); q.Uinit = h->Uinit.data();
    q.Utarget_r = h->Utr.data(); q.Utarget_i = h->Uti.data(); q.wmat_real_diag = h->wd.data(); q.Cfreq = h->cfreq.data();
    jq_handle* n = new (std::nothrow) jq_handle();
    if (!n) return fail(h, JQ_ENOMEM, "jq_update_hconst: out of host memory");
    n->opt = h->opt;
    n->force_plain = h->force_plain;
    int rc = create_impl(&q, n);
    if (rc == JQ_OK) rc = try_embed(n, &q);
    auto settings = [&](jq_handle* t) {
        t->solver_id = h->solver_id; t->m = h->m; t->solver_tol = h->solver_tol;
    };
    if (rc == JQ_OK) {
        settings(n);
        if (n->emb) settings(n->emb);
        for (jq_handle* t : {n, n->emb}) {
            if (!t || h->rfreq.empty() || rc != JQ_OK) continue;
            t->rfreq = h->rfreq;
            if ((rc = dev_alloc(t, &t->d_rfreq, t->rfreq.size())) == JQ_OK &&
                hipMemcpy(t->d_rfreq, t->rfreq.data(), t->rfreq.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
                rc = fail(n, JQ_EHIP, "jq_update_hconst: upload of the rotation frequencies failed");
        }
    }
    if (rc == JQ_OK && h->integrator == 2) rc = jq_set_integrator(n, 2, h->imr_max_iter, h->imr_tol);
    if (rc == JQ_OK && h->wrank > 0) rc = jq_update_wmat(n, h->Wr.data(), h->Wi.data());      // full leakage weights
    if (rc != JQ_OK) {
        h->err = "jq_update_hconst: re-planning for the new Hconst failed: " + n->err;
        jq_destroy(n);
        return rc;
    }
    n->replanned = true;
    std::swap(*h, *n);
    jq_destroy(n);      // (the old plan and its device memory)
    return JQ_OK;
}

extern "C" int jq_update_hconst(jq_handle* h, const double* Hconst)
{
    if (!h) return JQ_EINVAL;
    if (!Hconst) return fail(h, JQ_EINVAL, "jq_update_hconst: NULL pointer");
    if (!h->subs.empty()) return multi_forall(h, [&](jq_handle* sub) { return jq_update_hconst(sub, Hconst); });
    HIPCHK(h, hipSetDevice(h->device));
    // The kernels, operator images and LDS plan were chosen from the nonzero structure of H0, Hsym_q, Hanti_q at jq_create.  The
    // reference lets scripts mutate params.Hconst arbitrarily: a drift with entries outside that structure (or any new drift
    // after such a re-plan, which may have the structure back) re-plans the handle in place -- same pointer, same settings.
    // (After a re-plan the handle keeps its new, more general plan while the drifts fit it -- a script that mutates Hconst per
    //  iteration, like eval_f_g_grad!'s loop, must not pay a full re-creation per call; it plans again only when a drift violates
    //  the current structure, or when the drift has regained a structure that admits a strictly better kernel family than the
    //  current plan's: 4 x 4 x n when the plan is not JQ_BW_T4, diagonal off-diagonal blocks when it is a plain band.)
    const bool fits = (h->BW == JQ_BW_T4) ? t4_structure(Hconst, h->Ntot)
                      : (h->BW == JQ_BW_OD) ? offdiag_blocks_diagonal(Hconst, h->Ntot) : (h->huge || block_band(Hconst, h->Ntot) <= (h->big ? h->BWc : h->BW));
    bool better = false;
    if (fits && h->replanned && h->BW != JQ_BW_T4 && !h->huge) {
        const size_t nn = (size_t)h->Ntot * h->Ntot;
        int bw = block_band(Hconst, h->Ntot);
        for (int q = 0; q < h->Nc; ++q)
            bw = std::max(bw, std::max(block_band(h->Hsym.data() + q * nn, h->Ntot), block_band(h->Hanti.data() + q * nn, h->Ntot)));
        bool t4 = bw <= 1 && (!h->big || h->NT <= 8) && t4_structure(Hconst, h->Ntot);
        bool od = !h->big && h->BW != JQ_BW_OD && h->NT >= 2 && bw == 1 && offdiag_blocks_diagonal(Hconst, h->Ntot);
        for (int q = 0; q < h->Nc && (t4 || od); ++q) {
            const double *hs = h->Hsym.data() + q * nn, *ha = h->Hanti.data() + q * nn;
            t4 = t4 && t4_structure(hs, h->Ntot) && t4_structure(ha, h->Ntot);
            od = od && offdiag_blocks_diagonal(hs, h->Ntot) && offdiag_blocks_diagonal(ha, h->Ntot);
        }
        // ... or a narrower block band than the plan's (the selection rule of create_dense)
        bool narrower = false;
        if (h->BW != JQ_BW_OD) {
            const int want = h->big ? (bw > 2 ? 15 : std::max(bw, 1)) : ((bw <= 2 && bw < h->NT - 1) ? bw : h->NT - 1);
            narrower = want < (h->big ? h->BWc : h->BW);
        }
        better = t4 || od || narrower;
    }
    if (!fits || better) return replan(h, Hconst);
    h->Hconst.assign(Hconst, Hconst + (size_t)h->Ntot * h->Ntot);
    if (h->emb) {
        jq_handle* e = h->emb;
        embed_matrix(Hconst, h->Ntot, h->emb_row, e->Ntot, e->Hconst.data());
        if (!t4_structure(e->Hconst.data(), e->Ntot) || upload_operators(e) != JQ_OK) {   // the new drift breaks the structure:
            jq_destroy(e);                                                                // work without the embedded twin
            h->emb = nullptr;
        }
    }
    return upload_operators(h);
}

extern "C" int jq_update_hconst_csc(jq_handle* h, const jq_csc* Hconst)
{
    if (!h) return JQ_EINVAL;
    if (!Hconst) return fail(h, JQ_EINVAL, "jq_update_hconst_csc: NULL pointer");
    std::vector<double> H0((size_t)h->Ntot * h->Ntot);
    const int rc = csc_to_dense(h, Hconst, h->Ntot, H0.data(), "jq_update_hconst_csc");
    return rc ? rc : jq_update_hconst(h, H0.data());
}

extern "C" int jq_update_wmat_diag(jq_handle* h, const double* w)
{
    if (!h) return JQ_EINVAL;
    if (!w) return fail(h, JQ_EINVAL, "jq_update_wmat_diag: NULL pointer");
    if (!h->subs.empty()) return multi_forall(h, [&](jq_handle* sub) { return jq_update_wmat_diag(sub, w); });
    h->wd.assign(w, w + h->Ntot);
    h->wrank = 0;      // back to Diagonal weights
    h->Wr.clear(), h->Wi.clear();
    if (h->emb) {
        embed_rows(w, h->Ntot, 1, h->emb_row, h->emb->Ntot, h->emb->wd.data());
        h->emb->wrank = 0;
    }
    return JQ_OK;
}

// Eigen-decomposition of a Hermitian n x n matrix A = Ar + i Ai (column-major) by cyclic complex Jacobi rotations: on return
// lam[k] and the columns V[:, k] = Vr + i Vi with A = V diag(lam) V^H.  n <= 256, called once per jq_update_wmat.
static void hermitian_eig(int n, std::vector<double>& Ar, std::vector<double>& Ai, std::vector<double>& lam, std::vector<double>& Vr,
                          std::vector<double>& Vi)
{
    Vr.assign((size_t)n * n, 0.0);
    Vi.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) Vr[i + (size_t)n * i] = 1.0;
    auto at = [n](std::vector<double>& M, int i, int j) -> double& { return M[i + (size_t)n * j]; };
    double total = 0.0;
    for (size_t i = 0; i < Ar.size(); ++i) total += Ar[i] * Ar[i] + Ai[i] * Ai[i];
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int q = 1; q < n; ++q)
            for (int p = 0; p < q; ++p) off += at(Ar, p, q) * at(Ar, p, q) + at(Ai, p, q) * at(Ai, p, q);
        if (off <= 1e-32 * total) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double xr = at(Ar, p, q), xi = at(Ai, p, q);
                const double g = std::hypot(xr, xi);
                if (g == 0.0 || g * g <= 1e-36 * total) continue;
                // a_pq = g e^{i phi}; with P = diag(1, e^{-i phi}) the 2 x 2 block is P [[a_pp, g], [g, a_qq]] P^H, the real
                // rotation R = [[c, s], [-s, c]] diagonalises the real block: U = P R
                const double er = xr / g, ei = xi / g;      // e^{i phi}
                const double theta = (at(Ar, q, q) - at(Ar, p, p)) / (2.0 * g);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
                // U = [[c, s], [-s e^{-i phi}, c e^{-i phi}]] (rows p, q; columns p, q)
                // columns: M[:, p] <- c M[:, p] - s e^{-i phi} M[:, q] ; M[:, q] <- s M[:, p] + c e^{-i phi} M[:, q]
                auto cols = [&](std::vector<double>& Mr, std::vector<double>& Mi) {
                    for (int i = 0; i < n; ++i) {
                        const double pr = at(Mr, i, p), pi = at(Mi, i, p), qr = at(Mr, i, q), qi = at(Mi, i, q);
                        const double wr = er * qr + ei * qi, wi = er * qi - ei * qr;      // e^{-i phi} M[i, q]
                        at(Mr, i, p) = c * pr - sn * wr;
                        at(Mi, i, p) = c * pi - sn * wi;
                        at(Mr, i, q) = sn * pr + c * wr;
                        at(Mi, i, q) = sn * pi + c * wi;
                    }
                };
                cols(Ar, Ai);
                cols(Vr, Vi);
                // rows (U^H from the left): M[p, :] <- c M[p, :] - s e^{i phi} M[q, :] ; M[q, :] <- s M[p, :] + c e^{i phi} M[q, :]
                for (int j = 0; j < n; ++j) {
                    const double pr = at(Ar, p, j), pi = at(Ai, p, j), qr = at(Ar, q, j), qi = at(Ai, q, j);
                    const double wr = er * qr - ei * qi, wi = er * qi + ei * qr;          // e^{i phi} M[q, j]
                    at(Ar, p, j) = c * pr - sn * wr;
                    at(Ai, p, j) = c * pi - sn * wi;
                    at(Ar, q, j) = sn * pr + c * wr;
                    at(Ai, q, j) = sn * pi + c * wi;
                }
                at(Ar, p, q) = at(Ai, p, q) = at(Ar, q, p) = at(Ai, q, p) = 0.0;
                at(Ai, p, p) = at(Ai, q, q) = 0.0;
            }
    }
    lam.resize(n);
    for (int i = 0; i < n; ++i) lam[i] = at(Ar, i, i);
}

// the kernels' low-rank table of one (sub-)handle from the eigenpairs: lam[JQ_MAX_WRANK] | a_k[NP], b_k[NP] per k
static int upload_wlr(jq_handle* h, const std::vector<int>& keep, const std::vector<double>& lam, const std::vector<double>& Vr,
                      const std::vector<double>& Vi, int n, const std::vector<int>* row_map)
{
    HIPCHK(h, hipSetDevice(h->device));
    const int stride = h->NP;
    const int wlam = std::max<int>(JQ_MAX_WRANK, (int)keep.size());
    const size_t old_size = h->wlr.size();
    h->wlr.assign((size_t)wlam + (size_t)2 * wlam * stride, 0.0);
    for (size_t k = 0; k < keep.size(); ++k) {
        h->wlr[k] = lam[keep[k]];
        for (int i = 0; i < n; ++i) {
            const int row = row_map ? (*row_map)[i] : i;
            h->wlr[wlam + (2 * k) * stride + row] = Vr[i + (size_t)n * keep[k]];
            h->wlr[wlam + (2 * k + 1) * stride + row] = Vi[i + (size_t)n * keep[k]];
        }
    }
    h->wlam = wlam;
    int rc;
    if ((!h->d_wlr || h->wlr.size() > old_size) && (rc = dev_alloc(h, &h->d_wlr, h->wlr.size()))) return rc;
    HIPCHK(h, hipMemcpy(h->d_wlr, h->wlr.data(), h->wlr.size() * sizeof(double), hipMemcpyHostToDevice));
    h->wrank = (int)keep.size();
    h->wlr_real = true;
    for (size_t k = 0; k < keep.size() && h->wlr_real; ++k)
        for (int i = 0; i < n; ++i)
            if (Vi[i + (size_t)n * keep[k]] != 0.0) {
                h->wlr_real = false;
                break;
            }
    std::fill(h->wd.begin(), h->wd.end(), 0.0);
    return JQ_OK;
}

extern "C" int jq_update_wmat(jq_handle* h, const double* Wr, const double* Wi)
{
    if (!h) return JQ_EINVAL;
    if (!Wr) return fail(h, JQ_EINVAL, "jq_update_wmat: NULL pointer");
    if (!h->subs.empty()) return multi_forall(h, [&](jq_handle* sub) { return jq_update_wmat(sub, Wr, Wi); });
    const int n = h->Ntot;
    const size_t nn = (size_t)n * n;
    double wmax = 0.0;
    bool diagonal = true;
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
            const double a = Wr[i + (size_t)n * j], b = Wi ? Wi[i + (size_t)n * j] : 0.0;
            if (!std::isfinite(a) || !std::isfinite(b)) return fail(h, JQ_EINVAL, "jq_update_wmat: non-finite entry");
            wmax = std::max(wmax, std::max(std::fabs(a), std::fabs(b)));
            if (b != 0.0 || (i != j && a != 0.0)) diagonal = false;
        }
    // the same matrices as last time (the Julia binding pushes the weights before every evaluation; a host eigen-decomposition, the
    // reproduction check and a blocking upload cost 9 ms at Ntot = 96, 87 ms at 256 -- per call and per device): nothing to do
    if (!diagonal && h->wrank > 0 && h->Wr.size() == nn && h->Wi.size() == nn && memcmp(h->Wr.data(), Wr, nn * sizeof(double)) == 0) {
        bool same = true;
        if (Wi) same = memcmp(h->Wi.data(), Wi, nn * sizeof(double)) == 0;
        else
            for (size_t i = 0; i < nn && same; ++i) same = (h->Wi[i] == 0.0);
        if (same) return JQ_OK;
    }
    if (diagonal) {      // Diagonal weights written as a full matrix: the fast path
        std::vector<double> d(n);
        for (int i = 0; i < n; ++i) d[i] = Wr[i + (size_t)n * i];
        return jq_update_wmat_diag(h, d.data());
    }
    if (h->integrator == 2)
        return fail(h, JQ_EUNSUPPORTED, "jq_update_wmat: the implicit-midpoint path weights with params.wmat (always Diagonal, "
                                        "src/evalobjgrad.jl:90, :1155): pass it with jq_update_wmat_diag");
    for (int j = 0; j < n; ++j)
        for (int i = 0; i <= j; ++i) {
            const double ds = Wr[i + (size_t)n * j] - Wr[j + (size_t)n * i];
            const double da = Wi ? Wi[i + (size_t)n * j] + Wi[j + (size_t)n * i] : 0.0;
            if (std::fabs(ds) > 1e-12 * wmax || std::fabs(da) > 1e-12 * wmax)
                return fail(h, JQ_EUNSUPPORTED, "jq_update_wmat: wmat_real + i wmat_imag must be Hermitian (wmat_real symmetric, wmat_imag "
                                                "antisymmetric), as objparams builds it from forb_states (src/evalobjgrad.jl:220-231)");
        }
    std::vector<double> Ar(Wr, Wr + nn), Ai(nn, 0.0), lam, Vr, Vi;
    if (Wi) Ai.assign(Wi, Wi + nn);
    for (int j = 0; j < n; ++j)      // exactly Hermitian input for the rotations
        for (int i = 0; i < j; ++i) {
            const double sr = 0.5 * (Ar[i + (size_t)n * j] + Ar[j + (size_t)n * i]), si = 0.5 * (Ai[i + (size_t)n * j] - Ai[j + (size_t)n * i]);
            Ar[i + (size_t)n * j] = Ar[j + (size_t)n * i] = sr;
            Ai[i + (size_t)n * j] = si, Ai[j + (size_t)n * i] = -si;
        }
    for (int i = 0; i < n; ++i) Ai[i + (size_t)n * i] = 0.0;
    hermitian_eig(n, Ar, Ai, lam, Vr, Vi);
    double lmax = 0.0;
    for (double l : lam) lmax = std::max(lmax, std::fabs(l));
    std::vector<int> keep;
    for (int k = 0; k < n; ++k)
        if (std::fabs(lam[k]) > 1e-13 * lmax) keep.push_back(k);
    // (any rank: up to JQ_MAX_WRANK on every kernel family with the low-rank terms, beyond it on the cooperative, slab and run-time-size
    //  kernels -- run_eval routes; a full-rank W costs about two dense products per application)
    {   // the kept terms must reproduce W (guards the decomposition itself)
        double err = 0.0;
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) {
                double sr = 0.0, si = 0.0;
                for (int k : keep) {
                    const double ar = Vr[i + (size_t)n * k], ai = Vi[i + (size_t)n * k], br = Vr[j + (size_t)n * k], bi = Vi[j + (size_t)n * k];
                    sr += lam[k] * (ar * br + ai * bi);      // f_i conj(f_j)
                    si += lam[k] * (ai * br - ar * bi);
                }
                err = std::max(err, std::max(std::fabs(sr - Wr[i + (size_t)n * j]), std::fabs(si - (Wi ? Wi[i + (size_t)n * j] : 0.0))));
            }
        if (err > 1e-11 * wmax) return fail(h, JQ_EHIP, "jq_update_wmat: internal error, the eigen-decomposition does not reproduce W");
    }
    int rc = upload_wlr(h, keep, lam, Vr, Vi, n, nullptr);
    if (rc == JQ_OK && h->emb) {
        rc = upload_wlr(h->emb, keep, lam, Vr, Vi, n, &h->emb_row);
        if (rc != JQ_OK) h->err = h->emb->err;
    }
    if (rc != JQ_OK) {      // nothing half-applied: the early-out above must not take a failed upload for "these weights are in place"
        h->Wr.clear(), h->Wi.clear();
        h->wrank = 0;
        if (h->emb) h->emb->wrank = 0;
        return rc;
    }
    h->Wr.assign(Wr, Wr + nn);
    h->Wi.assign(nn, 0.0);
    if (Wi) h->Wi.assign(Wi, Wi + nn);
    return ensure_wjac_plan(h);
}

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
template <int NT, bool MODD, int NS, bool WLR = false> __global__ void k_forward_cq(PropArgs);    // jq_cq_kernels.h (own translation units); NS: column quads per workgroup; WLR: full (real, low-rank) leakage weights
template <int NT, bool MODD, bool ORD, bool WLR = false> __global__ void k_backward_cq(PropArgs);
template <int NT, bool MODD, bool ORD, int NR = 3, bool WLR = false> __global__ void k_backward_cq3(PropArgs);    // jq_cq_split_kernels.h: three (NR = 2: two) workgroups per column quad
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
static int select_cq_kernels(jq_handle* h, bool fwd2, int bwd_nr, bool wlr, prop_kernel_t* fwd, prop_kernel_t* bwd)      // bwd_nr: workgroups per quad in the backward sweep (0 / 1: one); wlr: full (real, low-rank) leakage weights
{
    const bool bwd3 = bwd_nr == 3, bwd2 = bwd_nr == 2;
    const bool modd = (h->m > 0 ? h->m : 0) & 1;
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
template <int NT> __global__ void k_forward_cq_imr(PropArgs);      // jq_cq_imr_kernels.h (own translation units)
template <int NT> __global__ void k_backward_cq_imr(PropArgs);
template <int NT> __global__ void k_backward_cq_imr2(PropArgs);    // (state and adjoint chain on two sets of waves, NT <= 6)
#define JQ_DECLCI(nt)                                                    \
    extern template __global__ void k_forward_cq_imr<nt>(PropArgs);      \
    extern template __global__ void k_backward_cq_imr<nt>(PropArgs);
JQ_DECLCI(1) JQ_DECLCI(2) JQ_DECLCI(3) JQ_DECLCI(4) JQ_DECLCI(5) JQ_DECLCI(6) JQ_DECLCI(7)
#undef JQ_DECLCI
template <int NT> __global__ void k_backward_cq_imr3(PropArgs);    // (three workgroups per evaluation, as k_backward_cq3)
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
static int select_cq_imr_kernels(jq_handle* h, bool two, bool three, prop_kernel_t* fwd, prop_kernel_t* bwd)
{
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
    extern template __global__ void k_forward_rowlane<npj, true>(PropArgs);     \
    extern template __global__ void k_backward_rowlane<npj, true>(PropArgs);    \
    extern template __global__ void k_backward_rowlane2<npj>(PropArgs);
JQ_FOR_EACH_ROWLANE(JQ_DECLR)
#undef JQ_DECLR

static int select_rowlane_kernels(jq_handle* h, bool split, prop_kernel_t* fwd, prop_kernel_t* bwd)
{
#define JQ_PICKR(npj)                                                          \
    if (h->rl_npj == npj) {                                                    \
        *fwd = h->wrank > 0 ? k_forward_rowlane<npj, true> : k_forward_rowlane<npj>;                                         \
        *bwd = h->wrank > 0 ? k_backward_rowlane<npj, true> : split ? k_backward_rowlane2<npj> : k_backward_rowlane<npj>;     \
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

struct EvalOut {
    std::vector<double> res;    // [nsamples][4] primary, secondary, Re s, Im s
    std::vector<double> grad0;  // forced adjoint (total gradient), weighted sum over samples
    std::vector<double> grad1;  // unforced adjoint (infidelity gradient), only objFuncType != 1
};

// JQ_BW_T4 structure, Stormer-Verlet / Neumann: estimated time of one batch in units of a slab-kernel round (4 #CU slabs), by the
// plan run_eval would choose -- cooperative-quad kernels (<= cq_max_quads column quads: 0.196 s per round of #CU quads against
// 1.917 s at cnot3), quad-layout kernels with 1 / 2 / 3 slabs per workgroup, slab kernels.  (The same figures as in run_eval.)
// Time of one round of the 4 x 4 x n kernel families relative to a round of the slab kernels (4 #CU slabs), measured at cnot3
// (scripts/time_staircase.py, round 3: 0.495 / 0.748 / 1.104 s for #CU / 2 #CU / 3 #CU slabs on the quad-layout kernels with 1 / 2 / 3
// slabs per workgroup, 0.192 s for a round of the cooperative-quad kernels)
// (round 5, same unit of 1.7935 s: 0.378 / 0.741 / 1.038 s -- one slab per workgroup now runs its backward sweep on two waves per column
//  quad, jq_quad_split_kernels.h; 0.192 s for up to #CU column quads on the cooperative-quad kernels, 0.298 s for up to 2 #CU)
static const double T4_REL[4] = {1.0, 0.2108, 0.413, 0.579};
static const double T4_REL_CQ = 0.107;      // <= #CU column quads
static const double T4_REL_CQ2 = 0.166;     // <= 2 #CU: forward sweep with two quads per workgroup, backward sweep k_backward_qsplit<.., 2>
static double t4_plan_cost(const jq_handle* h, long long nsamples)
{
    const long long nslabs = h->parts > 1 ? nsamples * h->parts : (nsamples + h->sps - 1) / h->sps;
    const long long nquads = (nsamples * h->N + 3) / 4;
    if (h->cq_max_quads > 0 && nquads <= h->cq_max_quads) return nquads <= h->num_cu ? T4_REL_CQ : nquads <= 2 * h->num_cu ? T4_REL_CQ2 : T4_REL_CQ * (double)((nquads + h->num_cu - 1) / h->num_cu);
    const double* rel = T4_REL;
    double best = rel[0] * (double)((nslabs + 4 * h->num_cu - 1) / (4 * h->num_cu));
    if (nslabs <= h->quad_max_slabs)
        for (int k = 1; k <= 3; ++k) {
            if ((size_t)(2 * JQ_WIN_TPS + 2 * h->NcK) * h->mat_elems * 8 + (size_t)bwd_lds_tail(h->NT, h->NcK, 4 * k, (long long)h->NT * 64) > 163840) continue;
            if (h->NT <= 2 && nslabs > h->num_cu) continue;
            best = std::min(best, rel[k] * (double)((nslabs + k * h->num_cu - 1) / (k * h->num_cu)));
        }
    return best;
}

// Chunk length of a backward sweep whose per-step trace records have `trace_rows` rows: the tile stream of h->chunk_steps steps fits its
// buffer; the records of a chunk ([trace_rows][cs][NcK JQ_NTR] doubles) are bounded by the option trace_bytes (default 4 GiB), so that
// large ensembles take more, shorter chunks instead of an allocation that grows with batch size x gate length.  ONE function for the
// sweep and for the decision that depends on its first chunk (the split latency kernels need a first chunk longer than their ring).
#define JQ_CQ3_RING 8      // = JQ_CQ3_SLOTS (jq_cq_split_kernels.h, compiled in its own translation units)
static int backward_chunk_steps(const jq_handle* h, size_t trace_rows)
{
    size_t tbudget = (size_t)4 << 30;
    if (h->opt.has(O_TRACE_BYTES) && h->opt.get(O_TRACE_BYTES) > 0) tbudget = (size_t)h->opt.get(O_TRACE_BYTES);
    const long long cst = (long long)(tbudget / (std::max<size_t>(trace_rows, 1) * (size_t)h->NcK * JQ_NTR * sizeof(double)));
    return (int)std::max<long long>(1, std::min<long long>(h->chunk_steps, cst));
}

__global__ void k_add_to(double* __restrict__ y, const double* __restrict__ x, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] += x[i];
}

// The batched evaluation behind every hot-path entry point.
// d_packed != nullptr: the packed ensemble result (k_pack) is also left at this DEVICE address of h's GPU.
#define JQ_ERETRY_INTERNAL (-1000)      // run_eval_impl: k_backward_cq3 gave up (the handle leaves it alone for a while): evaluate again
#define JQ_CQ3_MAX_FAULTS 6
// Evaluations in flight per device, process-wide.  Every outermost run_eval is counted (enter / leave); an evaluation that wants the
// three-workgroup latency kernels asks for the device EXCLUSIVELY (try_exclusive: granted when it is the only one in flight) and new
// evaluations then wait at enter() until it is through (one latency evaluation: ~ 0.15 s at cnot3).  So inside a process a grid whose
// workgroups wait for each other never shares the GPU with another launch of the library -- the co-residency it needs is checked,
// not assumed (two handles in two threads, the sub-handles of a same-device multi handle, ...).
struct DevGate {
    std::mutex m;
    std::condition_variable cv;
    int active = 0;
    bool exclusive = false;
    void enter()
    {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return !exclusive; });
        ++active;
    }
    void leave()
    {
        std::lock_guard<std::mutex> l(m);
        --active;
    }
    bool try_exclusive()      // (the caller is one of the active evaluations)
    {
        std::lock_guard<std::mutex> l(m);
        if (exclusive || active != 1) return false;
        exclusive = true;
        return true;
    }
    void release_exclusive()
    {
        {
            std::lock_guard<std::mutex> l(m);
            exclusive = false;
        }
        cv.notify_all();
    }
};
static DevGate g_gate[64];
static DevGate& dev_gate(int device) { return g_gate[(unsigned)device % 64u]; }
static thread_local int g_eval_depth = 0;      // run_eval calls itself (split batches, the embedded twin): only the outermost call is counted
struct GateHold {      // exclusive use of a device for the rest of a scope
    DevGate* g = nullptr;
    bool acquire(DevGate& gate)
    {
        if (gate.try_exclusive()) g = &gate;
        return g != nullptr;
    }
    ~GateHold()
    {
        if (g) g->release_exclusive();
    }
};
static int run_eval_impl(jq_handle* h, const double* pcof, int ncoeff, int nsamples, const double* eps, const double* wgt,
                         const double* shift, bool adjoint, double* hist_r, double* hist_i, EvalOut* out, double* d_packed);
static int run_eval(jq_handle* h, const double* pcof, int ncoeff, int nsamples, const double* eps, const double* wgt,
                    const double* shift, bool adjoint, double* hist_r, double* hist_i, EvalOut* out, double* d_packed = nullptr)
{
    DevGate& gate = dev_gate(h->device);
    const bool outer = g_eval_depth++ == 0;
    if (outer) gate.enter();
    int rc = run_eval_impl(h, pcof, ncoeff, nsamples, eps, wgt, shift, adjoint, hist_r, hist_i, out, d_packed);
    if (rc == JQ_ERETRY_INTERNAL) rc = run_eval_impl(h, pcof, ncoeff, nsamples, eps, wgt, shift, adjoint, hist_r, hist_i, out, d_packed);
    if (outer) gate.leave();
    --g_eval_depth;
    return rc;
}
static int run_eval_impl(jq_handle* h, const double* pcof, int ncoeff, int nsamples, const double* eps, const double* wgt,
                         const double* shift, bool adjoint, double* hist_r, double* hist_i, EvalOut* out, double* d_packed)
{
    HIPCHK(h, hipSetDevice(h->device));
    // Ensembles that do not fill their last round: the time of a batch is a staircase in its size (every workgroup runs the
    // whole sequential time loop; cnot3: 3 072 samples = one round of the three-slab quad-layout kernels 1.18 s, 3 200 samples =
    // two rounds 2.35 s).  A batch of q full rounds + a remainder is evaluated as two batches when the plan says that is
    // faster -- the remainder on whatever suits ITS size (3 200 samples: 1.18 + 0.20 s on the cooperative-quad kernels).
    // Samples are independent and the results are sums over samples, so only the order of those sums changes.
    // (the cost model is that of the 4 x 4 x n MFMA families: a batch that the row-lane / lane kernels take -- small Hilbert spaces
    //  with that structure, e.g. SWAP-02 -- must not be split: round 2 did, and paid two latency-bound launches for one)
    const long long ncols_split = (long long)nsamples * h->N;
    const bool small_family_batch = (h->rl_npj > 0 && ncols_split <= h->rl_max_cols) ||
                                    (h->lane_np > 0 && ncols_split >= h->lane_min_cols && ncols_split <= h->lane_max_cols);
    if (!h->in_split && !small_family_batch && h->wrank == 0 && h->quad_max_slabs > 0 && h->integrator == 1 && h->solver_id == 1 && !hist_r && eps && nsamples > 1 && !h->opt.on(O_NOSPLIT)) {
        // candidates: the largest number of FULL rounds of the quad-layout kernels with 1, 2 or 3 slabs per workgroup
        long long n_main = 0;
        double best = t4_plan_cost(h, nsamples) - 1e-9;
        for (int k = 1; k <= 3; ++k) {
            const long long per_round = (long long)k * h->num_cu * (h->parts > 1 ? 1 : h->sps) / (h->parts > 1 ? h->parts : 1);      // samples of a full round
            const long long nm = per_round > 0 ? (long long)nsamples / per_round * per_round : 0;
            if (nm <= 0 || nm >= nsamples) continue;
            const double c = t4_plan_cost(h, nm) + t4_plan_cost(h, nsamples - nm);
            if (c < best) best = c, n_main = nm;
        }
        if (n_main > 0) {
            h->in_split = true;
            EvalOut o2;
            const int n1 = (int)n_main, n2 = nsamples - n1;
            int rc = run_eval(h, pcof, ncoeff, n1, eps, wgt, shift, adjoint, nullptr, nullptr, out, d_packed);
            const jq_timing t1 = h->timing;
            const size_t npk = (size_t)2 + 2 * (size_t)ncoeff;
            if (rc == JQ_OK && d_packed) {
                rc = dev_grow(h, &h->d_pk2, &h->cap_pk2, npk);
                if (rc == JQ_OK && hipMemcpyAsync(h->d_pk2, d_packed, npk * sizeof(double), hipMemcpyDeviceToDevice, h->stream) != hipSuccess)
                    rc = fail(h, JQ_EHIP, "hipMemcpyAsync (packed result of the first part of a split batch)");
            }
            if (rc == JQ_OK) rc = run_eval(h, pcof, ncoeff, n2, eps + n1, wgt ? wgt + n1 : nullptr, shift, adjoint, nullptr, nullptr, &o2, d_packed);
            h->in_split = false;
            if (rc != JQ_OK) return rc;
            if (d_packed) {
                hipLaunchKernelGGL(k_add_to, dim3((unsigned)((npk + 255) / 256)), dim3(256), 0, h->stream, d_packed, h->d_pk2, (int)npk);
                HIPCHK(h, hipGetLastError());
                HIPCHK(h, hipStreamSynchronize(h->stream));
            }
            out->res.insert(out->res.end(), o2.res.begin(), o2.res.end());
            for (size_t i = 0; i < out->grad0.size() && i < o2.grad0.size(); ++i) out->grad0[i] += o2.grad0[i];
            for (size_t i = 0; i < out->grad1.size() && i < o2.grad1.size(); ++i) out->grad1[i] += o2.grad1[i];
            // timing: sums; the kernel family / size / band reported are those of the first (larger) part
            h->timing.ms_total += t1.ms_total, h->timing.ms_propagate += t1.ms_propagate, h->timing.ms_generate += t1.ms_generate;
            h->timing.ms_forward += t1.ms_forward, h->timing.ms_backward += t1.ms_backward;
            h->timing.n_forward_launches += t1.n_forward_launches, h->timing.n_backward_launches += t1.n_backward_launches;
            h->timing.mfma_executed += t1.mfma_executed, h->timing.mfma_backward += t1.mfma_backward, h->timing.svts += t1.svts;
            h->timing.kernel_family = t1.kernel_family, h->timing.kernel_size = t1.kernel_size, h->timing.kernel_band = t1.kernel_band;
            h->timing.ms_shard_min = h->timing.ms_shard_max = h->timing.ms_total;
            return JQ_OK;
        }
    }
    const int Nsig = 2 * h->Nc;
    // src/evalobjgrad.jl:604-606
    if (ncoeff % Nsig != 0 || ncoeff < 3 * Nsig) {
        char buf[160];
        snprintf(buf, sizeof buf, "pcof must have an even number of elements >= %d, not %d", 3 * Nsig, ncoeff);
        return fail(h, JQ_EINVAL, buf);
    }
    const int D1 = ncoeff / (Nsig * h->Nfreq);  // :608
    // bcparams: nCoeff = Nfreq*D1*2*Ncoupled must equal length(pcof) (src/bsplines.jl:177-181)
    if (h->Nfreq * D1 * Nsig != ncoeff)
        return fail(h, JQ_EDIM, "DimensionMismatch: Inconsistent number of coefficients and size of parameter vector (nCoeff != length(pcof))");
    if (D1 < 3) return fail(h, JQ_EINVAL, "need at least 3 B-spline coefficients per control function");
    if (nsamples < 1) return fail(h, JQ_EINVAL, "need at least one sample");
    // Structure embedding (try_embed): batches that would run on the dense / band MFMA families go to the embedded twin,
    // whose operators have the JQ_BW_T4 structure (quad-layout / JQ_BW_T4 slab kernels).  State histories stay here (their
    // rows are the user's), the implicit-midpoint path too.
    if (h->emb && !hist_r && h->integrator == 1) {
        const long long nc_used = (long long)nsamples * h->N;
        // (full leakage weights: the row-lane kernels take every batch of an Ntot <= 16 problem -- the lane kernels have no low-rank terms)
        const bool small_family = h->solver_id == 1 && ((h->rl_npj > 0 && (nc_used <= h->rl_max_cols || h->wrank > 0)) ||
                                                        (h->lane_np > 0 && nc_used >= h->lane_min_cols && nc_used <= h->lane_max_cols));
        if (h->emb_mode == 2 || !small_family) {
            jq_handle* e = h->emb;
            std::vector<double> sh(e->Ntot, 0.0);
            for (int i = 0; i < h->Ntot; ++i)   // (default: the reference's 0.01 * 10^(j-2) by the USER's level index, src/ipopt_interface.jl:41-44)
                sh[h->emb_row[i]] = shift ? shift[i] : (i >= 1 ? 0.01 * pow(10.0, (double)(i - 1)) : 0.0);
            const int rc = run_eval(e, pcof, ncoeff, nsamples, eps, wgt, sh.data(), adjoint, nullptr, nullptr, out, d_packed);
            if (rc != JQ_OK) h->err = e->err;
            h->timing = e->timing;
            return rc;
        }
    }
    if (adjoint && !h->rfreq.empty() && h->integrator != 1)
        return fail(h, JQ_EUNSUPPORTED, "uncoupled controls (Hunc_ops): gradients with the Stormer-Verlet integrator only (the reference's "
                                        "implicit-midpoint adjoint has no term for them, src/evalobjgrad.jl:1347)");

    const int nslabs = h->parts > 1 ? nsamples * h->parts : (nsamples + h->sps - 1) / h->sps;
    // small batches: cooperative (row-split) kernels, one workgroup of NT waves per slab; large batches: slab
    // kernels, one wave per slab (jq_coop_kernels.h explains the trade-off)
    // small Hilbert spaces: lane kernels, one lane per column (jq_lane_kernels.h)
    const long long ncols_used = (long long)nsamples * h->N;
    // implicit midpoint: row-lane kernels for Ntot <= 16 with N <= 4 (the columns of an evaluation share one wave for the
    // solver's per-evaluation convergence test), cooperative MFMA kernels (one slab per workgroup) otherwise
    const bool imr = (h->integrator == 2);
    const bool imr_rl = imr && h->rl_npj > 0 && h->N <= 4;
    // JQ_BW_T4 structure with an evaluation's columns inside one quad: quad-layout kernels (jq_quad_imr_kernels.h)
    // (any batch size: one workgroup per slab, rounds of one workgroup per CU)
    const bool imr_quad = imr && !imr_rl && h->quad_max_slabs > 0 && (h->N == 1 || h->N == 2 || h->N == 4);
    const bool imr_coop = imr && !imr_rl && !imr_quad;
    const bool imr_parts = imr_coop && h->parts > 1;      // N > 16: one workgroup per evaluation, its 16-column parts in turn
    // (both images of a step resident in LDS when they fit; dense 96 x 96 operators: the <6, 5> instantiation that reads them from
    //  HBM / L2 per product like the Ntot > 96 variants)
    const bool imr_hbm = imr_coop && h->NT <= 6 && h->mat_elems_c > 0 && coop_imr_lds_bytes(h->NT, h->mat_elems_c) > 163840;
    if (imr_coop && (h->mat_elems_c == 0 || (imr_hbm && !(h->NT == 6 && h->BWc == 5))))
        return fail(h, JQ_EUNSUPPORTED, "implicit midpoint: no kernels for these operators (no cooperative layout / images that do not fit the LDS)");
    const int cpw = imr_rl ? imr_cols_per_wave(h->N) : 4;   // columns per wave of the row-lane kernels
    // Full leakage weights (jq_update_wmat; low-rank terms in the kernels): row-lane kernels for every batch of an Ntot <= 16 problem,
    // quad-layout kernels with one slab per workgroup (their WLRT instantiations) for the 4 x 4 x n structure, else the cooperative
    // kernels (every batch size) and, where those do not exist, the slab kernels <1, 0> / <6, 5>; no lane or JQ_BW_T4 slab kernels;
    // cooperative-quad kernels for REAL weight matrices of rank <= 4 (wfull_cq below).
    const bool wfull = h->wrank > 0;
    if (wfull && imr)
        return fail(h, JQ_EUNSUPPORTED, "full leakage weights (jq_update_wmat): the implicit-midpoint path weights with params.wmat (Diagonal)");
    const bool wjac = wfull && h->solver_id == 2;      // full weights with the Jacobi solver: cooperative kernels, else the slab kernels <1, 0> / <6, 5>
    const bool rl = imr_rl || (!imr && h->rl_npj > 0 && h->solver_id == 1 && (ncols_used <= h->rl_max_cols || wfull));
    const bool lane = !imr && !rl && !wfull && h->lane_np > 0 && h->solver_id == 1 && ncols_used >= h->lane_min_cols && ncols_used <= h->lane_max_cols;
    const long long nwaves_rl = (ncols_used + cpw - 1) / cpw;
    const long long ncols = rl ? 4 * nwaves_rl : (ncols_used + 63) / 64 * 64;      // row-lane: column SLOTS (4 per wave)
    // JQ_BW_T4 structure, small batches: the quad-layout kernels (one workgroup per slab, its four waves carry four columns
    // each; 3 x shorter dependent chain than the cooperative kernels).  option quad=0 disables them.
    // Which kernels for nslabs slabs of this structure?  Time of one round relative to the slab kernels' round of 4 #CU slabs
    // (T4_REL, measured at cnot3): quad layout with 1 / 2 / 3 slabs per workgroup for #CU / 2 #CU /
    // 3 #CU slabs.  Fewest "round units" wins; spw = 0: slab kernels.
    int spw = 0;
    if (!imr && !lane && !rl && h->solver_id == 1 && nslabs <= h->quad_max_slabs) {
        const double* rel = T4_REL;
        auto quad_lds = [&](int k) {    // backward kernel, k slabs per workgroup
            return (size_t)(2 * JQ_WIN_TPS + 2 * h->NcK) * h->mat_elems * 8 + (size_t)bwd_lds_tail(h->NT, h->NcK, 4 * k, (long long)h->NT * 64);
        };
        double best = rel[0] * ((nslabs + 4 * h->num_cu - 1) / (4 * h->num_cu));
        for (int k = 1; k <= 3; ++k) {
            if (quad_lds(k) > 163840) continue;
            const double c = rel[k] * ((nslabs + k * h->num_cu - 1) / (k * h->num_cu));
            if (c < best - 1e-9) {
                best = c;
                spw = k;
            }
        }
        // One or two 16-row blocks (cnot2 embedded: NT = 1): a state array of the slab kernels is only 4 NT registers, nothing
        // spills and two workgroups share a CU -- measured 3.5e9 vs 2.2e9 SVTS/s for cnot2 x 65 536 samples.  The quad
        // layout keeps the latency regime (at most one slab per CU).
        if (h->NT <= 2 && nslabs > h->num_cu) spw = 0;
        if (h->opt.has(O_QUAD8)) {      // experiments / tests: force 4 / 8 / 12 waves (as far as the LDS allows)
            spw = std::max(1, std::min(3, (int)h->opt.get(O_QUAD8) + 1));
            while (spw > 1 && quad_lds(spw) > 163840) --spw;
        }
        if (wfull) spw = 1;      // (the instantiations with the low-rank terms: one slab per workgroup, any number of rounds)
    }
    if (wfull && !wjac && !rl && h->BW == JQ_BW_T4 && spw == 0)
        return fail(h, JQ_EUNSUPPORTED, "full leakage weights (jq_update_wmat): the quad-layout kernels are disabled or do not fit for this "
                                        "4 x 4 x n problem, and the JQ_BW_T4 slab kernels have no low-rank terms");
    // (one slab per workgroup, one wave per SIMD, the operators of a step in registers for all its fixed-point iterations; a
    // two-slab variant that re-reads them from LDS was measured 1.4 x slower, jq_kernel_inst.hip)
    if (imr_quad) spw = 1;
    // latency regime of the JQ_BW_T4 structure: one workgroup of NT waves per column quad (jq_cq_kernels.h)
    const long long nquads_used = (ncols_used + 3) / 4;
    // (full weights, round 5: four slots -- real weight matrices of rank <= 4, complex ones of rank <= 2 -- on the cooperative-quad kernels with
    //  one quad per workgroup, LDS permitting (jq_cq_kernels.h CqW); a complex W only with the backward sweep on two / three workgroups
    //  per quad, see below; option cq_w=0: the quad-layout kernels as before)
    const bool wfull_cq = wfull && (h->wlr_real ? h->wrank <= 4 : h->wrank <= 2) && h->NT <= 7 && h->opt.on(O_CQ_W) && (ncols_used + 3) / 4 <= h->num_cu &&
                          (size_t)(2 * JQ_WIN_TPS + 2 * h->NcK) * h->mat_elems * 8 + (size_t)32 * h->NT * 8 + (size_t)6 * (h->NT + 2) * 64 * 8 +
                                  (size_t)std::max(2, h->NcK + (h->NcK + 1) / 2) * h->NT * 64 * 8 + (size_t)2 * h->NT * 64 * 8 <= 163840;
    bool cq = !imr && !lane && !rl && (!wfull || wfull_cq) && h->solver_id == 1 && h->cq_max_quads > 0 && nquads_used <= h->cq_max_quads &&
              !h->opt.has(O_QUAD8);      // (quad8 asks for a quad-layout variant explicitly)
    const int qps = h->parts > 1 ? 4 : (h->sps * h->N + 3) / 4;      // column quads of a full slab
    // ... and of the implicit-midpoint integrator (jq_cq_imr_kernels.h): N = 4, one workgroup of NT waves per evaluation
    const bool imr_cq = imr_quad && h->N == 4 && h->parts == 1 && h->cq_max_quads > 0 && nquads_used <= h->cq_max_quads &&
                        h->opt.on(O_IMR_CQ);
    // more column quads than CUs: the forward sweep takes two quads per workgroup (one round of workgroups at ~ 1.5 x the time
    // instead of two rounds; option cq_fwd2=0: one quad per workgroup, =1: always two)
    const bool cq_fwd2 = cq && !wfull && (h->opt.has(O_CQ_FWD2) ? h->opt.on(O_CQ_FWD2) : nquads_used > h->num_cu);
    // single evaluations and small ensembles: the backward sweep on three workgroups (CUs) per column quad -- state re-integration,
    // adjoint step, trace products, pipelined through a ring in global memory (jq_cq_split_kernels.h).  All 3 x quads workgroups must
    // be resident at once (groups of 8 quads: 24 workgroups); option cq3=0: the one-workgroup kernel
    // (the kernels address quad q as quad q & 3 of slab q >> 2: every slab has four quad slots, a ragged last slab leaves some idle)
    const long long nq_pad = (4LL * nslabs + 7) / 8 * 8;
    const bool c3_set = h->opt.has(O_CQ3);
    const long long c3_v = h->opt.get(O_CQ3);
    // (not for the sub-handles of the same-device test mode: their launches share the GPU, the workgroups of a quad might not all be resident)
    // Co-residency is checked, not assumed: the split is taken only when this evaluation is the only one of the process on the device
    // (GateHold: others then wait until it is through), when no CU mask is in force (the grid is sized for all CUs the device
    // reports), and not while the handle is cooling down after a fault.
    // (round 5: 2 x quads <= CUs -- 81 .. 128 cnot3 samples -- two workgroups per quad: state re-integration | adjoint step + trace products,
    //  Stormer-Verlet only; option cq3=3: three or none)
    GateHold gate_hold;
    bool cq3 = false;
    int cq_nr = 0;      // workgroups per column quad of the split backward sweep
    if ((cq || imr_cq) && adjoint) {
        const char* why = nullptr;
        cq_nr = 3 * nq_pad <= h->num_cu ? 3 : (cq && 2 * nq_pad <= h->num_cu && !(c3_set && c3_v == 3)) ? 2 : 0;
        // The consumer roles read the state the sweep starts from out of the state file (the carry of the trace products, first chunk
        // only), and the state role writes its end-of-chunk state there when it is through.  It cannot be through before they have
        // started only if it has to WAIT for them -- which it does from step 8 on (the ring has 8 slots): the first chunk must be longer
        // than the ring.  (Shorter first chunks -- tests, problems with a handful of steps -- were a race that the late-start hook
        // option debug=16 exposed in round 5; they take the one-workgroup kernel.)
        // (the SAME function gives the chunk length of the sweep below: backward_chunk_steps; the trace-record rows of these families)
        const long long cs_first = std::min<long long>(backward_chunk_steps(h, (size_t)nslabs * qps * (imr_cq ? h->NT : 1)), h->nsteps);
        if (c3_set && c3_v == 0) why = "not taken: option cq3=0";
        else if (cs_first <= JQ_CQ3_RING) why = "not taken: the first chunk of the sweep is not longer than the hand-off ring (8 steps)";
        else if (cq_nr == 0) why = "not taken: two / three workgroups per column quad exceed the compute units";
        else if (h->cq3_off) why = "not taken: switched off after repeated faults (dead waits between the workgroups of a quad)";
        else if (h->cq3_skip > 0) why = "not taken: cooling down after a fault";
        else if (getenv("HSA_CU_MASK") || getenv("ROC_GLOBAL_CU_MASK")) why = "not taken: a CU mask is set (HSA_CU_MASK / ROC_GLOBAL_CU_MASK)";
        else if (g_eval_depth != 1) why = "not taken: nested evaluation (part of a split batch / embedded twin)";
        else if (!gate_hold.acquire(dev_gate(h->device))) why = "not taken: another evaluation of this process is in flight on the device";
        cq3 = (why == nullptr);
        if (!cq3) cq_nr = 0;
        if (h->cq3_skip > 0) --h->cq3_skip;
        h->cq3_last = cq3 ? (cq_nr == 3 ? "taken: three workgroups per column quad, device held exclusively" : "taken: two workgroups per column quad, device held exclusively") : why;
    }
    // A complex W needs W_i vr(t_n) in the middle of the adjoint step: only the split kernels, whose state role is steps ahead, have it.
    // Without them (more than 128 samples, the gate taken, cooling down, option cq3=0 ...) the evaluation runs on the quad-layout kernels.
    if (cq && wfull && !h->wlr_real && adjoint && !cq3) cq = false;
    const size_t cq3_quad = 64 + (size_t)8 * 8 * h->NT * 64 + 64;      // doubles per quad: JQ_CQ3_HEAD + JQ_CQ3_SLOTS * JQ_CQ3_ARRAYS * NT * 64 + JQ_CQ3_TAIL
    const size_t cq3_need = 64 + (size_t)nq_pad * cq3_quad;
    if (cq3) {
        const int rc0 = dev_grow(h, &h->d_cq3, &h->cap_cq3, cq3_need);
        if (rc0) return rc0;
        HIPCHK(h, hipMemsetAsync(h->d_cq3, 0, 64 * sizeof(double), h->stream));      // (the error word of the evaluation)
    }
    const bool imr_cq3 = imr_cq && cq3 && cq_nr == 3;
    const bool imr_cq2 = imr_cq && !imr_cq3 && h->NT <= 6 && h->opt.on(O_IMR_CQ2) &&
                         cq_imr2_lds(h, (size_t)(2 * JQ_WIN_TPS + 2 * h->NcK) * h->mat_elems * 8) <= 163840;
    if (cq) spw = 0;
    const bool quad = spw > 0;
    const bool quad8 = spw > 1;
    // mid-size ensembles of the 4 x 4 x n structure (at most one column quad per SIMD): the backward sweep with the state and the
    // adjoint chain of a quad on two waves, one time step apart (jq_quad_split_kernels.h; option qsplit=0: the one-wave kernel)
    //   qw = 4: one slab per workgroup, two waves per SIMD (the quad-layout plan with one slab per workgroup);
    //   qw = 2: half a slab per workgroup, one wave per SIMD -- more column quads than CUs on the cooperative-quad plan, whose
    //           backward sweep would take two rounds (the forward sweep stays on k_forward_cq with two quads per workgroup)
    const bool qs_set = h->opt.has(O_QSPLIT);
    const bool qs_on = adjoint && h->NT <= 6 && !(qs_set && h->opt.get(O_QSPLIT) == 0);
    int qs_qw = 0;
    if (qs_on && quad && !imr && spw == 1 && !wfull && qsplit_lds(h, 4) <= 163840) qs_qw = 4;      // (!imr: the implicit-midpoint quad kernels also run with spw = 1)
    // (option qsplit=2: qw = 2 for every batch of the cooperative-quad plan that does not take the three-workgroup kernels -- tests)
    const bool qs_force2 = qs_set && h->opt.get(O_QSPLIT) == 2;
    if (qs_on && cq && !cq3 && !wfull && ((nquads_used > h->num_cu && 2 * nslabs <= h->num_cu) || qs_force2) && qsplit_lds(h, 2) <= 163840) qs_qw = 2;
    const bool qsplit = qs_qw > 0;
    const int qs_blocks = qsplit ? (4 * nslabs + qs_qw - 1) / qs_qw : 0;
    if (qsplit) {
        const int rc0 = dev_grow(h, &h->d_qsplit, &h->cap_qsplit, (size_t)qs_blocks * qs_qw * 2 * JQ_QS_ARRAYS * h->NT * 64);
        if (rc0) return rc0;
    }
    // (full leakage weights: the cooperative kernels sum their column dot products over the waves through an LDS record of
    //  2 x JQ_COOP_WDOTS x NT x 16 doubles behind the Jacobi norms; where that does not fit next to the operator slots the slab kernels serve)
    const size_t coop_w_bytes = wfull ? (size_t)2 * JQ_COOP_WDOTS * h->NT * 16 * 8 : 0;
    const bool coop_w_fits = !wfull || h->NT > 6 ||
                             (size_t)2 * h->mat_elems_c * 8 + (size_t)32 * h->NT * 8 + (size_t)2 * h->KT * 64 * 8 + (size_t)16 * h->NT * 8 + coop_w_bytes <= 163840;
    const bool coop = imr_coop || (!cq && !quad && !lane && !rl && h->NT >= 2 && h->coop_ok && coop_w_fits && (h->solver_id == 1 || h->big || wjac) &&
                                   (nslabs <= h->coop_max_slabs || wfull));
    if (wjac && !coop && h->BW == JQ_BW_T4)      // (jq_update_wmat / jq_set_linear_solver re-plan such handles without the structure: cannot happen)
        return fail(h, JQ_EHIP, "internal error: full leakage weights with the Jacobi solver on a 4 x 4 x n plan without cooperative kernels");      // (Ntot > 96: also the Jacobi solver; full weights: every batch size -- the slab kernels have no low-rank terms)
    // row-lane kernels, Stormer-Verlet: the backward sweep's two chains on two waves (jq_rowlane_kernels.h k_backward_rowlane2);
    // option rl_split=0: one wave (tests: the two variants must agree bit for bit)
    // (both integrators; while the doubled wave count still finds idle issue slots: measured in round 3, HISTORY.md --
    //  NPJ <= 8: up to three waves per SIMD, NPJ = 12, 16 (constant images in LDS, 24 .. 32 operand registers per image row): one)
    bool rl_split = rl && 2 * nwaves_rl <= (long long)(h->rl_npj > 8 ? 4 : 12) * h->num_cu;
    if (!h->opt.on(O_RL_SPLIT)) rl_split = false;
    if (wfull) rl_split = false;      // (the one-wave backward kernel carries the low-rank terms)
    prop_kernel_t kfwd, kbwd;
    lane_init_t klinit = nullptr;
    lane_term_t klterm = nullptr;
    int rc = imr_cq ? select_cq_imr_kernels(h, imr_cq2, imr_cq3, &kfwd, &kbwd)
             : imr_quad ? select_quad_imr_kernels(h, &kfwd, &kbwd)
             : imr_coop ? (imr_parts ? select_coop_imr_parts_kernels(h, imr_hbm, &kfwd, &kbwd) : select_coop_imr_kernels(h, imr_hbm, &kfwd, &kbwd))
             : imr_rl ? select_rowlane_imr_kernels(h, rl_split, &kfwd, &kbwd)
             : rl ? select_rowlane_kernels(h, rl_split, &kfwd, &kbwd)
             : lane ? select_lane_kernels(h, &kfwd, &kbwd, &klinit, &klterm)
                  : cq ? select_cq_kernels(h, cq_fwd2, cq_nr, wfull, &kfwd, &kbwd)
                  : coop ? select_coop_kernels(h, &kfwd, &kbwd) : quad ? (wfull ? select_quad_w_kernels(h, &kfwd, &kbwd) : select_quad_kernels(h, spw, &kfwd, &kbwd)) : select_kernels(h, &kfwd, &kbwd);
    if (rc) return rc;
    if (qsplit && (rc = select_qsplit_kernel(h, qs_qw, &kbwd))) return rc;
    // Jacobi solver with N > 16 on the slab kernels: ONE workgroup per sample when its parts fit one (<= JQ_WAVES = 4 slabs, N <= 64) -- the
    // waves add their parts' residual norms through LDS, so the stopping test is the reference's (norm over the whole Ntot x N block,
    // src/linear_solvers.jl:121) and not a test per 16-column part (round 5; option jac_wg=0: per part).  More parts, or the cooperative
    // kernels (Ntot > 96): per part as before (include/juqbox_hip.h).
    const bool jac_wg = !imr && h->solver_id == 2 && h->parts > 1 && h->parts <= JQ_WAVES && !coop && !cq && !quad && !lane && !rl &&
                        h->opt.on(O_JAC_WG);
    const int nblocks = jac_wg ? nsamples : imr_parts ? nsamples : (cq || imr_cq) ? 4 * nslabs : rl ? (int)nwaves_rl : lane ? (int)(ncols / 64) : quad8 ? (nslabs + spw - 1) / spw : (coop || quad) ? nslabs : (nslabs + JQ_WAVES - 1) / JQ_WAVES;
    const bool huge = coop && h->huge;
    const int nthreads = huge ? 64 * JQ_HUGE_WAVES : jac_wg ? 64 * h->parts : (lane || rl) ? 64 : (coop || cq || imr_cq) ? 64 * h->NT : quad8 ? 256 * spw : 256;
    // per-step trace records: one per wave (cooperative, lane, row-lane, implicit-midpoint kernels) or one per workgroup
    // (slab / quad kernels: summed over the workgroup's waves in LDS)
    const int trace_rows = qsplit ? qs_blocks : imr_parts ? nsamples * h->NT : imr_cq ? nslabs * qps * h->NT : cq ? nslabs * qps : (lane || rl) ? nblocks : huge ? nslabs * JQ_HUGE_WAVES : coop ? nslabs * h->NT : imr_quad ? nslabs * JQ_WAVES : nblocks;
    const long long stride = rl ? h->rl_stride : lane ? h->lane_stride : coop ? h->mat_elems_c : h->mat_elems;
    const double* himg = rl ? h->d_himg_r : lane ? h->d_himg_l : coop ? h->d_himg_c : h->d_himg;
    const size_t state_doubles = rl ? (size_t)JQ_ROWLANE_ROWS * nwaves_rl * 64
                                    : lane ? (size_t)JQ_LANE_ROWS(h->lane_np) * ncols : (size_t)nslabs * h->state_stride;
    const size_t colinfo_doubles = (lane || rl) ? (size_t)2 * ncols : (size_t)nslabs * 32;
    const int ntr = h->NcK * JQ_NTR;      // (trace scalars per step of the LARGEST control group)
    const int ngroups = ctrl_ngroups(h->Nc);
    const bool two_pass = adjoint && h->objFuncType != 1;
    // chunk length: the tile stream of h->chunk_steps steps fits its buffer; the per-step trace records of a backward chunk
    // ([trace_rows][cs][ntr] doubles) are bounded by option trace_bytes (default 4 GiB) so that large ensembles take more,
    // shorter chunks instead of an allocation that grows with batch size x gate length
    const int cs = adjoint ? backward_chunk_steps(h, (size_t)trace_rows) : h->chunk_steps;
    if (cq3 && std::min(cs, h->nsteps) <= JQ_CQ3_RING)      // (the decision above was made for this very chunking)
        return fail(h, JQ_EHIP, "internal error: split latency kernels selected for a first chunk that is not longer than their hand-off ring");

    // ---- capacity ------------------------------------------------------------------------------
    if ((rc = dev_grow(h, &h->d_pcof, &h->cap_pcof, (size_t)ncoeff))) return rc;
    if (state_doubles > h->cap_state || !h->d_state || !h->d_state_save) {
        h->cap_state = 0;
        if ((rc = dev_alloc(h, &h->d_state, state_doubles))) return rc;
        if ((rc = dev_alloc(h, &h->d_state_save, state_doubles))) return rc;
        h->cap_state = state_doubles;
    }
    if ((rc = dev_grow(h, &h->d_colinfo, &h->cap_colinfo, colinfo_doubles))) return rc;
    // (parking images of the slab kernels: one array per slab; implicit midpoint with N > 16: the work area of ImrParts, ten)
    const size_t park_slabs = (size_t)nslabs * (imr_parts ? JQ_IMRP_ARRAYS : huge ? JQ_HUGE_VECS : 1);      // (huge: the work area of a slab)
    if (!lane && !rl && (park_slabs > h->cap_slabs || !h->d_park)) {
        h->cap_slabs = 0;
        if ((rc = dev_alloc(h, &h->d_park, park_slabs * h->KT * 64))) return rc;
        h->cap_slabs = park_slabs;
    }
    if (adjoint && (rc = dev_grow(h, &h->d_traces, &h->cap_traces, (size_t)trace_rows * cs * ntr))) return rc;
    if ((rc = dev_grow(h, &h->d_grad, &h->cap_grad, (size_t)2 * ncoeff))) return rc;
    if ((rc = dev_grow(h, &h->d_res, &h->cap_res, (size_t)nsamples * 4))) return rc;

    // ---- inputs --------------------------------------------------------------------------------
    hipStream_t s = h->stream;
    HIPCHK(h, hipMemcpyAsync(h->d_pcof, pcof, (size_t)ncoeff * sizeof(double), hipMemcpyHostToDevice, s));
    bool use_shift = false;
    std::vector<double> colinfo(colinfo_doubles, 0.0);
    if (lane || rl) {   // [eps per column slot | weight per column slot]
        for (long long c = 0; c < ncols_used; ++c) {
            const int smp = (int)(c / h->N);
            const long long slot = rl ? (c / cpw) * 4 + (c % cpw) : c;
            colinfo[slot] = eps ? eps[smp] : 0.0;
            colinfo[ncols + slot] = wgt ? wgt[smp] : 1.0;
            if (eps && eps[smp] != 0.0) use_shift = true;
        }
    } else {
        for (int sl = 0; sl < nslabs; ++sl)
            for (int c = 0; c < (h->parts > 1 ? 16 : h->sps * h->N); ++c) {
                const int smp = h->parts > 1 ? sl / h->parts : sl * h->sps + c / h->N;
                if (smp < nsamples && (h->parts == 1 || 16 * (sl % h->parts) + c < h->N)) {
                    colinfo[(size_t)sl * 32 + c] = eps ? eps[smp] : 0.0;
                    colinfo[(size_t)sl * 32 + 16 + c] = wgt ? wgt[smp] : 1.0;
                    if (eps && eps[smp] != 0.0) use_shift = true;
                }
            }
    }
    HIPCHK(h, hipMemcpyAsync(h->d_colinfo, colinfo.data(), colinfo.size() * sizeof(double), hipMemcpyHostToDevice, s));
    std::vector<double> tabs((size_t)32 * h->NT, 0.0);
    const size_t ws_off = rl ? 16 : lane ? (size_t)h->lane_np : (size_t)16 * h->NT;   // tables: [wd | ws]
    for (int i = 0; i < h->Ntot; ++i) {
        tabs[i] = h->wd[i];
        // reference perturbation: Hconst[j,j] += ep*0.01*10^(j-2), j = 2..Ntot (src/ipopt_interface.jl:41-44)
        tabs[ws_off + i] = shift ? shift[i] : (i >= 1 ? 0.01 * pow(10.0, (double)(i - 1)) : 0.0);
    }
    HIPCHK(h, hipMemcpyAsync(h->d_tabs, tabs.data(), tabs.size() * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemsetAsync(h->d_grad, 0, (size_t)2 * ncoeff * sizeof(double), s));

    SplineArgs sp;
    sp.pcof = h->d_pcof; sp.cfreq = h->d_cfreq; sp.D1 = D1; sp.Nfreq = h->Nfreq; sp.Ncoupled = h->Nc; sp.nCoeff = ncoeff;
    sp.dtknot = h->T / (D1 - 2);
    sp.rfreq = h->rfreq.empty() ? nullptr : h->d_rfreq;

    const double dt = h->T / h->nsteps;
    PropArgs a;
    memset(&a, 0, sizeof a);
    const double* cimg_base = rl ? h->d_cimg_r : lane ? h->d_cimg_l : coop ? h->d_cimg_c : h->d_cimg;      // (control-group order)
    a.stream = h->d_stream; a.cimg = cimg_base; a.state = h->d_state; a.colinfo = h->d_colinfo;
    a.traces = h->d_traces;
    a.tabs = h->d_tabs; a.stride = stride; a.pieces = (int)(stride * 8 / 1024); a.nslots = h->nslots; a.m = h->m;
    a.nslabs = rl ? (int)nwaves_rl : lane ? (int)ncols : nslabs; a.Ncoupled = ctrl_gstart(h->Nc, 1) /* first control group */; a.Ntot = h->Ntot; a.N = h->N; a.use_shift = use_shift ? 1 : 0;
    a.tinv = 1.0 / h->T; a.state_stride = h->state_stride; a.parts = h->parts; a.nsamples = nsamples; a.sps = h->sps; a.qps = qps;
    a.wlr = h->d_wlr; a.wrank = h->wrank; a.wlam = h->wlam; a.wstride = h->NP; a.wlr_lds = -1; a.wlr_sc_lds = -1; a.jac_wg_lds = -1; a.wcplx = (h->wrank > 0 && !h->wlr_real) ? 1 : 0;
    // JACOBI_SOLVER: the kernels iterate on c-scaled right-hand sides (A = c rhs, c = h / 2: DESIGN.md section 3), so their
    // residual norm is |c| times the reference's ||X_j - X_{j-1}|| (src/linear_solvers.jl:121): the threshold is scaled alike
    a.jacobi_tol2 = (h->solver_id == 2) ? (h->solver_tol * 0.5 * dt) * (h->solver_tol * 0.5 * dt) : 0.0;
    if (imr) {   // fixed-point solver of the implicit-midpoint step: iteration cap and per-lane threshold (jq_rowlane_imr_kernels.h)
        a.m = h->imr_max_iter;
        a.jacobi_tol2 = h->imr_tol * h->imr_tol;
    }
    for (int q = 0; q < JQ_MAXNC; ++q) a.bw_trace[q] = q < h->Nc ? h->bw_trace[q] : 0;      // (first control group; the backward sweeps set their own)
    // dynamic LDS layout: [operator staging | tables wd, ws | (backward: carry, parking images)]
    // cooperative kernels: [two operator slots | tables wd, ws | two x exchange buffers]
    const int batch = coop ? 0 : (quad || cq) ? -1 : h->batch;
    const size_t lds_stage = (coop && (h->NT > 6 || imr_hbm)) ? 0      // operators are read from HBM, no LDS staging
                             : batch > 0   ? (size_t)2 * (2 * batch + 1) * 2 * stride * 8 + (size_t)2 * h->NcK * stride * 8
                             : batch < 0 ? (size_t)(2 * JQ_WIN_TPS + 2 * h->NcK) * stride * 8
                                         : (size_t)2 * stride * 8;
    const size_t lds_cq = lds_stage + (size_t)32 * h->NT * 8 + (size_t)6 * (h->NT + 2) * 64 * 8 + (size_t)std::max(2, h->NcK + (h->NcK + 1) / 2) * h->NT * 64 * 8;      // tables, x exchange, trace hand-off / wg-sum scratch (one region)
    const size_t lds_fwd = huge ? 0 : rl ? (wfull ? (size_t)JQ_RL_WTAB * 8 : 0) : lane ? 0 : (cq || imr_cq) ? lds_cq : imr_coop ? coop_imr_lds_bytes(h->NT, imr_hbm ? 0 : stride)
                                           : lds_stage + (size_t)32 * h->NT * 8 + (coop ? (size_t)2 * h->KT * 64 * 8 + (size_t)16 * h->NT * 8 + coop_w_bytes : 0);      // (+ the Jacobi solver's column norms [NT][16], the low-rank weights' dot exchange)
    const size_t lds_bwd = huge ? 0 : qsplit ? qsplit_lds(h, qs_qw) : rl ? (h->rl_npj > 8 ? (size_t)2 * h->NcK * h->rl_stride * 8 : 0) + (rl_split ? (size_t)2 * 3 * 64 * 8 : 0) /* records: 3 values per lane and slot, implicit midpoint 2 */ + (wfull ? (size_t)JQ_RL_WTAB * 8 : 0) /* low-rank weight table */ : lane ? 0 : imr_cq2 ? cq_imr2_lds(h, lds_stage) : (coop || cq || imr_cq) ? lds_fwd
                                : imr_quad ? lds_fwd + (size_t)JQ_MAXNC * nthreads * 8 + (size_t)(nthreads / 64) * h->NT * 64 * 8
                                : quad ? lds_stage + (size_t)bwd_lds_tail(h->NT, h->NcK, nthreads / 64, (long long)h->NT * 64)   // (a 16-row block per register)
                                : lds_stage + (size_t)bwd_lds_tail(h->NT, h->NcK, JQ_WAVES, h->park_lds ? (long long)h->KT * 64 : 0);
    // full leakage weights on the slab / quad kernels: a copy of the low-rank table behind everything else in LDS when it fits
    const size_t wlr_bytes = (wfull && cq) ? (size_t)2 * h->NT * 64 * 8      // (cooperative quad: the partial dots of two vectors, CqW)
                             : (wfull && !coop && !rl && !lane) ? ((size_t)h->wlam + (size_t)2 * h->wrank * h->NP) * 8 : 0;
    const int wlr_lds_fwd = (wlr_bytes && lds_fwd + wlr_bytes <= 163840) ? (int)lds_fwd : -1;
    const int wlr_lds_bwd = (wlr_bytes && lds_bwd + wlr_bytes <= 163840) ? (int)lds_bwd : -1;
    // (the cooperative-quad kernels have no table in global memory to fall back to: wfull_cq above admitted them only when this fits)
    if (wfull && cq && (wlr_lds_fwd < 0 || wlr_lds_bwd < 0)) return fail(h, JQ_EHIP, "internal error: no LDS left for the partial dots of the full leakage weights");
    // ... and, quad layout, the per-wave column scalars of the terms behind it (jq_kernels.h WLow::sc).  OFF unless option wlr_sc=1: measured
    // SLOWER than recomputing the dots (round 5, cnot3: 57 -> 70 ms per forbidden state -- an LDS round trip on the critical path of a
    // wave that is alone on its SIMD costs more than the ~ 64 independent instructions of a dot pair; profiles/r05_exp_variants.txt (3))
    const size_t wsc_bytes = (wlr_bytes && quad && h->opt.get(O_WLR_SC) == 1) ? (size_t)(nthreads / 64) * JQ_MAX_WRANK * 24 * 8 : 0;
    const size_t wsc_off_fwd = lds_fwd + (wlr_lds_fwd >= 0 ? wlr_bytes : 0), wsc_off_bwd = lds_bwd + (wlr_lds_bwd >= 0 ? wlr_bytes : 0);
    const int wsc_lds_fwd = (wsc_bytes && wsc_off_fwd + wsc_bytes <= 163840) ? (int)wsc_off_fwd : -1;
    const int wsc_lds_bwd = (wsc_bytes && wsc_off_bwd + wsc_bytes <= 163840) ? (int)wsc_off_bwd : -1;
    const size_t jac_bytes = jac_wg ? (size_t)2 * JQ_WAVES * 8 : 0;      // (residual exchange of the workgroup-wide Jacobi test, behind everything else)
    if (jac_wg && std::max(lds_fwd, lds_bwd) + jac_bytes > 163840) return fail(h, JQ_EHIP, "internal error: no LDS left for the Jacobi residual exchange");
    a.batch = batch; a.lds_tab_off = (int)lds_stage;
    a.park = h->d_park; a.park_lds = quad ? 1 : h->park_lds;
    if (cq) a.nslots = 0;
    a.debug = (int)h->opt.get(O_DEBUG);
    if (!lane && !rl) {
        HIPCHK(h, hipFuncSetAttribute((const void*)kfwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_fwd + (wlr_lds_fwd >= 0 ? wlr_bytes : 0) + (wsc_lds_fwd >= 0 ? wsc_bytes : 0) + jac_bytes)));
        HIPCHK(h, hipFuncSetAttribute((const void*)kbwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_bwd + (wlr_lds_bwd >= 0 ? wlr_bytes : 0) + (wsc_lds_bwd >= 0 ? wsc_bytes : 0) + jac_bytes)));
    }

    // events: [0]=start [1]=end, then pairs around every propagator launch
    const int nchunks = (h->nsteps + cs - 1) / cs;
    const size_t nev = 2 + 2 * (size_t)nchunks * (1 + (adjoint ? (two_pass ? 2 : 1) * ngroups : 0));
    while (h->ev.size() < nev) {
        hipEvent_t e;
        HIPCHK(h, hipEventCreate(&e));
        h->ev.push_back(e);
    }
    size_t evi = 2;
    HIPCHK(h, hipEventRecord(h->ev[0], s));

    if (rl)
        hipLaunchKernelGGL(k_init_state_rowlane, dim3((unsigned)nwaves_rl), dim3(64), 0, s, h->d_state, nwaves_rl, h->d_uinit_r, h->N,
                           ncols_used, cpw);
    else if (lane)
        hipLaunchKernelGGL(klinit, dim3((unsigned)(ncols / 64)), dim3(64), 0, s, h->d_state, ncols, h->d_uinit_l, h->N, ncols_used);
    else
        hipLaunchKernelGGL(k_init_state, dim3(nslabs), dim3(64), 0, s, h->d_state, h->state_stride, h->d_uimg, h->KT, h->parts);

    long long mfma = 0, mfma_fwd = 0;
    const long long tiles = (lane || rl) ? 0 : coop ? coop_tiles(h->NT, h->BWc) : band_tiles(h->NT, h->BW);
    std::vector<long long> ttiles(h->Nc, 0);
    for (int q = 0; q < h->Nc && !lane && !rl; ++q)
        ttiles[q] = coop ? coop_tiles(h->NT, h->BWc)
                            : (h->BW == JQ_BW_T4) ? ((h->bw_trace[q] & JQ_T4_DIAG) ? 4 * h->NT : 0)
                                                  : band_tiles(h->NT, h->bw_trace[q] == 0 ? 0 : h->BW, h->bw_trace[q] == 2);
    // ---- forward sweep -------------------------------------------------------------------------
    for (int n0 = 0; n0 < h->nsteps; n0 += cs) {
        const int nc = std::min(cs, h->nsteps - n0);
        const int ntp = 2 * nc + 1;
        hipLaunchKernelGGL(k_ctrl, dim3((ntp + 127) / 128), dim3(128), 0, s, sp, h->d_tf, n0, ntp, dt, h->d_pq);
        hipLaunchKernelGGL(k_stream, dim3((unsigned)((stride + 255) / 256), ntp), dim3(256), 0, s, himg, h->d_pq,
                           h->Nc, stride, 0.5 * dt, h->d_stream);
        a.nsteps_chunk = nc; a.step0 = n0; a.first_chunk = (n0 == 0); a.h = dt; a.forced = 1;
        a.hist_r = hist_r; a.hist_i = hist_i;
        a.wlr_lds = wlr_lds_fwd;
        a.wlr_sc_lds = wsc_lds_fwd;
        a.jac_wg_lds = jac_wg ? (int)lds_fwd : -1;
        a.period = 7; a.npro = 0; a.nslots = h->nslots;
        {   // slab kernels: Kp05 S05 Kn0 S0 Kn1 S1 Kp05 ; cooperative kernels: Kp05 S05 Kn0 Kn1 S0 S1 Kp05
            // {kind (0 K, 1 S, 2 constant image), time point offset / image index}
            const int kinds_s[7] = {0, 1, 0, 1, 0, 1, 0}, tps_s[7] = {1, 1, 0, 0, 2, 2, 1};
            const int kinds_c[7] = {0, 1, 0, 0, 1, 1, 0}, tps_c[7] = {1, 1, 0, 2, 0, 2, 1};
            const int* kinds = coop ? kinds_c : kinds_s;
            const int* tps = coop ? tps_c : tps_s;
            a.sched_bits[0] = a.sched_bits[1] = a.sched_bits[2] = a.pro_bits = 0;
            for (int i = 0; i < 7; ++i) sched_pack(a.sched_bits, i, kinds[i], tps[i]);
        }
        HIPCHK(h, hipEventRecord(h->ev[evi++], s));
        hipLaunchKernelGGL(kfwd, dim3(cq_fwd2 ? nblocks / 2 : nblocks), dim3(cq ? nthreads + 128 : imr_cq ? nthreads + 128 : nthreads), lds_fwd + (wlr_lds_fwd >= 0 ? wlr_bytes : 0) + (wsc_lds_fwd >= 0 ? wsc_bytes : 0) + jac_bytes, s, a);      // (cooperative quad: two staging waves)
        HIPCHK(h, hipEventRecord(h->ev[evi++], s));
        mfma += (long long)nslabs * nc * (8 + 2 * h->m) * tiles;
    }
    HIPCHK(h, hipGetLastError());
    mfma_fwd = mfma;
    const double leak_scale = imr ? 0.25 * dt * (1.0 / h->T) : 0.5 * dt * (1.0 / h->T);
    if (imr_parts)
        hipLaunchKernelGGL(k_terminal_parts, dim3(nsamples), dim3(64), 0, s, h->d_state, h->state_stride, h->d_vtr, h->d_vti, h->KT,
                           h->N, h->parts, leak_scale, h->d_res, 1);
    else if (imr_coop || imr_quad)
        hipLaunchKernelGGL(k_terminal_imr, dim3(nslabs), dim3(64), 0, s, h->d_state, h->state_stride, h->d_vtr, h->d_vti, h->KT,
                           h->N, h->sps, nsamples, leak_scale, h->d_res);
    else if (imr)
        hipLaunchKernelGGL(k_terminal_rowlane_imr, dim3((nsamples + 63) / 64), dim3(64), 0, s, h->d_state, nwaves_rl, h->d_vtr_r,
                           h->d_vti_r, h->N, nsamples, leak_scale, h->d_res, cpw);
    else if (rl)
        hipLaunchKernelGGL(k_terminal_rowlane, dim3((nsamples + 63) / 64), dim3(64), 0, s, h->d_state, nwaves_rl, h->d_vtr_r,
                           h->d_vti_r, h->N, nsamples, leak_scale, h->d_res);
    else if (lane)
        hipLaunchKernelGGL(klterm, dim3((nsamples + 63) / 64), dim3(64), 0, s, h->d_state, ncols, h->d_vtr_l, h->d_vti_l, h->N,
                           nsamples, leak_scale, h->d_res);
    else if (h->parts > 1)
        hipLaunchKernelGGL(k_terminal_parts, dim3(nsamples), dim3(64), 0, s, h->d_state, h->state_stride, h->d_vtr, h->d_vti, h->KT,
                           h->N, h->parts, leak_scale, h->d_res, 0);
    else
        hipLaunchKernelGGL(k_terminal, dim3(nslabs), dim3(64), 0, s, h->d_state, h->state_stride, h->d_vtr, h->d_vti, h->KT,
                           h->N, h->sps, nsamples, leak_scale, h->d_res);

    // ---- backward sweep(s) ---------------------------------------------------------------------
    // one sweep per (control group, forcing): the forced adjoint gives the total gradient, the unforced one (objFuncType != 1)
    // the infidelity gradient; every sweep restarts from the state the forward sweep and the terminal kernel left behind
    unsigned long long cq3_fault = 0;
    if (adjoint) {
        const int nsweeps = (two_pass ? 2 : 1) * ngroups;
        if (nsweeps > 1)
            HIPCHK(h, hipMemcpyAsync(h->d_state_save, h->d_state, state_doubles * sizeof(double),
                                     hipMemcpyDeviceToDevice, s));
        for (int sweep = 0; sweep < nsweeps; ++sweep) {
            const int pass = sweep / ngroups, grp = sweep % ngroups;
            const int q0 = ctrl_gstart(h->Nc, grp), ng = ctrl_gstart(h->Nc, grp + 1) - q0;
            const int ntr_g = ng * JQ_NTR;
            if (sweep > 0)
                HIPCHK(h, hipMemcpyAsync(h->d_state, h->d_state_save, state_doubles * sizeof(double),
                                         hipMemcpyDeviceToDevice, s));
            a.Ncoupled = ng;
            a.cimg = cimg_base + (size_t)2 * q0 * stride;
            long long trace_tiles = 0;
            for (int q = 0; q < JQ_MAXNC; ++q) {
                a.bw_trace[q] = q < ng ? h->bw_trace[q0 + q] : 0;
                if (q < ng) trace_tiles += ttiles[q0 + q];
            }
            for (int n0 = 0; n0 < h->nsteps; n0 += cs) {
                const int nc = std::min(cs, h->nsteps - n0);
                const int ntp = 2 * nc + 1;
                hipLaunchKernelGGL(k_ctrl, dim3((ntp + 127) / 128), dim3(128), 0, s, sp, h->d_tb, n0, ntp, -dt, h->d_pq);
                hipLaunchKernelGGL(k_stream, dim3((unsigned)((stride + 255) / 256), ntp), dim3(256), 0, s, himg,
                                   h->d_pq, h->Nc, stride, -0.5 * dt, h->d_stream);
                a.nsteps_chunk = nc; a.step0 = n0; a.first_chunk = (n0 == 0); a.h = -dt; a.forced = (pass == 0);
                a.hist_r = nullptr; a.hist_i = nullptr;
                a.wlr_lds = wlr_lds_bwd;
                a.wlr_sc_lds = wsc_lds_bwd;
                a.jac_wg_lds = jac_wg ? (int)lds_bwd : -1;
                a.period = 13 + 3 * ng; a.npro = (n0 == 0) ? ng : 0; a.nslots = h->nslots_bwd;
                {   // Kp05 S05 Kn0 S0 Kn1 S1 Kp05 | S0 | Hanti_q.. | Kn0 Kn1 S05 Kp05 S1 | (Hanti_q Hsym_q)..
                    const int kinds_s[8] = {0, 1, 0, 1, 0, 1, 0, 1}, tps_s[8] = {1, 1, 0, 0, 2, 2, 1, 0};
                    const int kinds_c[8] = {0, 1, 0, 0, 1, 1, 0, 1}, tps_c[8] = {1, 1, 0, 2, 0, 2, 1, 0};
                    const int* kinds = coop ? kinds_c : kinds_s;
                    const int* tps = coop ? tps_c : tps_s;
                    const int kinds2[5] = {0, 0, 1, 0, 1}, tps2[5] = {0, 2, 1, 1, 2};
                    a.sched_bits[0] = a.sched_bits[1] = a.sched_bits[2] = a.pro_bits = 0;
                    int k = 0;
                    for (int i = 0; i < 8; ++i) sched_pack(a.sched_bits, k++, kinds[i], tps[i]);
                    for (int q = 0; q < ng; ++q) sched_pack(a.sched_bits, k++, 2, ng + q);   // early traces: Hanti_q
                    for (int i = 0; i < 5; ++i) sched_pack(a.sched_bits, k++, kinds2[i], tps2[i]);
                    for (int q = 0; q < ng; ++q) {
                        sched_pack(a.sched_bits, k++, 2, ng + q);                               // late traces: Hanti_q
                        sched_pack(a.sched_bits, k++, 2, q);                                    //              Hsym_q
                        sched_pack(&a.pro_bits, q, 2, q);          // first chunk: carry products with Hsym_q
                    }
                }
                if (cq3) {      // (progress counters of the launch: the 64-double header in front of every quad's ring -- the ring itself is written
                                // before it is read; the error word in front of everything survives until the end of the evaluation, the
                                // arrival counter of the start-up rendezvous behind it is per launch)
                    HIPCHK(h, hipMemset2DAsync(h->d_cq3 + 64, cq3_quad * sizeof(double), 0, 64 * sizeof(double), (size_t)nq_pad, s));
                    HIPCHK(h, hipMemsetAsync(h->d_cq3 + 1, 0, 2 * sizeof(double), s));      // (arrival counter, state word of the launch)
                    a.park = h->d_cq3;
                    // rendezvous: about ONE launch duration (2 .. 100 ms; option cq3_rdv_us overrides) in polls of ~ 1.3 us -- an abandoned launch
                    // then costs at most what the launch itself would have; waits after a passed rendezvous: ~ 10 x the launch's expected
                    // duration, at least 50 ms (measured on this handle; before the first launch: 25 us per step, four times the slowest size measured)
                    const double us_step = h->cq3_us_per_step > 0.0 ? h->cq3_us_per_step : 25.0;
                    const double rdv_us = h->opt.has(O_CQ3_RDV_US) ? (double)h->opt.get(O_CQ3_RDV_US) : std::min(100.0e3, std::max(2.0e3, us_step * nc));
                    a.rdv_polls = (int)std::min<double>(2.0e9, std::max(16.0, rdv_us / 1.3));
                    a.wait_polls = (int)std::min<double>(2.0e9, (h->opt.has(O_CQ3_WAIT_MS) ? 1.0e3 * (double)h->opt.get(O_CQ3_WAIT_MS) : std::max(50.0e3, 10.0 * us_step * nc)) / 1.3);
                }
                HIPCHK(h, hipEventRecord(h->ev[evi++], s));
                if (qsplit) {      // (two waves per column quad; its window ring is deeper than the forward kernel's)
                    a.park = h->d_qsplit;
                    a.lds_tab_off = (int)((size_t)(2 * JQ_QS_TPS + 2 * h->NcK) * stride * 8);
                    a.batch = -1;
                }
                if (qsplit)
                    hipLaunchKernelGGL(kbwd, dim3((unsigned)qs_blocks), dim3(128 * qs_qw), lds_bwd, s, a);
                else if (cq3)
                    hipLaunchKernelGGL(kbwd, dim3((unsigned)(cq_nr * nq_pad)), dim3(nthreads + 128), lds_bwd + (wlr_lds_bwd >= 0 ? wlr_bytes : 0), s, a);      // (three / two workgroups per quad: NT block waves + two staging waves each)
                else
                hipLaunchKernelGGL(kbwd, dim3(nblocks), dim3((cq || rl_split) ? 2 * nthreads : imr_cq2 ? 2 * (nthreads + 128) : imr_cq ? nthreads + 128 : nthreads), lds_bwd + (wlr_lds_bwd >= 0 ? wlr_bytes : 0) + (wsc_lds_bwd >= 0 ? wsc_bytes : 0) + jac_bytes, s, a);      // (cooperative quad: state and adjoint chain on separate waves)
                HIPCHK(h, hipEventRecord(h->ev[evi++], s));
                if (cq3 && sweep == 0 && n0 == 0) {
                    // the first launch of the split says whether its workgroups were resident together: read the error word now instead
                    // of running every other chunk and sweep (each dead wait costs ~ 1.3 s) before the evaluation is repeated anyway
                    unsigned long long e1 = 0;
                    HIPCHK(h, hipMemcpyAsync(&e1, h->d_cq3, sizeof(e1), hipMemcpyDeviceToHost, s));
                    HIPCHK(h, hipStreamSynchronize(s));
                    if (h->opt.on(O_CQ3_FAULT)) e1 = (unsigned long long)h->opt.get(O_CQ3_FAULT);      // (test hook: as if a wait had been abandoned (1) / the rendezvous had failed (3))
                    if (e1) {
                        cq3_fault = e1;
                        break;
                    }
                }
                hipLaunchKernelGGL(k_trace_reduce, dim3((unsigned)(((long long)nc * ntr_g + 255) / 256)), dim3(256), 0, s,
                                   h->d_traces, trace_rows, nc, ntr_g, h->d_R);
                // gradbcarrier2! as a scatter: one workgroup per coefficient of the group's controls
                hipLaunchKernelGGL(k_gradacc, dim3(ng * 2 * h->Nfreq * D1), dim3(JQ_GRADACC_THREADS), 0, s, sp, h->d_R, h->d_tb, n0, nc, -dt,
                                   h->d_grad + (size_t)pass * ncoeff, q0, ng);
                mfma += (long long)nslabs * nc * (2 * (8 + 2 * h->m) * tiles + 4 * trace_tiles);
                if (n0 == 0) mfma += (long long)nslabs * trace_tiles;
            }
            if (cq3_fault) break;
        }
    }
    HIPCHK(h, hipGetLastError());
    if (d_packed) {
        if (wgt) {
            if ((rc = dev_grow(h, &h->d_wq, &h->cap_wq, (size_t)nsamples))) return rc;
            HIPCHK(h, hipMemcpyAsync(h->d_wq, wgt, (size_t)nsamples * sizeof(double), hipMemcpyHostToDevice, s));
        }
        hipLaunchKernelGGL(k_pack, dim3(1), dim3(256), 0, s, h->d_res, wgt ? h->d_wq : nullptr, nsamples, h->d_grad, ncoeff,
                           adjoint ? 1 : 0, two_pass ? 1 : 0, d_packed);
        HIPCHK(h, hipGetLastError());
    }
    HIPCHK(h, hipEventRecord(h->ev[1], s));

    // ---- outputs -------------------------------------------------------------------------------
    out->res.resize((size_t)nsamples * 4);
    HIPCHK(h, hipMemcpyAsync(out->res.data(), h->d_res, out->res.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    if (adjoint) {
        out->grad0.resize(ncoeff);
        HIPCHK(h, hipMemcpyAsync(out->grad0.data(), h->d_grad, (size_t)ncoeff * sizeof(double), hipMemcpyDeviceToHost, s));
        if (two_pass) {
            out->grad1.resize(ncoeff);
            HIPCHK(h, hipMemcpyAsync(out->grad1.data(), h->d_grad + ncoeff, (size_t)ncoeff * sizeof(double),
                                     hipMemcpyDeviceToHost, s));
        }
    }
    unsigned long long cq3_err = cq3_fault;
    if (cq3 && !cq3_fault) HIPCHK(h, hipMemcpyAsync(&cq3_err, h->d_cq3, sizeof(cq3_err), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    if (cq3 && debug_timing()) {      // development aid: progress counters, error word and XCC ids (+ 1) of the first quads
        std::vector<unsigned long long> hw((size_t)64 + 2 * cq3_quad);
        HIPCHK(h, hipMemcpy(hw.data(), h->d_cq3, hw.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int qd = 0; qd < 2; ++qd) {
            const unsigned long long* q = hw.data() + 64 + (size_t)qd * cq3_quad;
            fprintf(stderr, "jq cq3 quad %d: steps %llu %llu %llu, error %llu (launch %llu), xcc %llu %llu %llu\n", qd, q[0], q[8], q[16], q[24], hw[0], q[32], q[33], q[34]);
        }
    }
    if (cq3 && h->opt.on(O_CQ3_FAULT)) cq3_err = (unsigned long long)h->opt.get(O_CQ3_FAULT);
    if (cq3_err == 3) {
        // the launch was abandoned at its start-up rendezvous: not every workgroup became resident within cq3_rdv_us -- another process
        // holds the compute units.  Nothing is wrong with the handle: repeat on the one-workgroup kernel (milliseconds lost), stay off
        // the split for a few evaluations (2, 4, ... 64 while it keeps happening), never for good.
        ++h->cq3_busy;
        h->cq3_busy_streak = std::min(h->cq3_busy_streak + 1, 5);
        h->cq3_skip = 2 << h->cq3_busy_streak;      // (the repeat below counts as one)
        if (debug_timing()) fprintf(stderr, "jq: split latency kernel abandoned at its start-up rendezvous (GPU busy) -- evaluated again on one workgroup per quad\n");
        return JQ_ERETRY_INTERNAL;
    }
    if (cq3_err) {      // a wait between the three workgroups of a quad was abandoned (1), or they ran on different XCDs (2): the results are void
        ++h->cq3_faults;
        if (cq3_err == 2) ++h->cq3_faults_xcd;
        h->cq3_skip = 2 << std::min(h->cq3_faults, 10);      // (4, 8, 16, ... evaluations; the repeat below counts as one)
        if (h->cq3_faults >= JQ_CQ3_MAX_FAULTS) h->cq3_off = true;
        if (debug_timing()) fprintf(stderr, "jq: k_backward_cq3 reported %llu -- evaluated again with k_backward_cq\n", cq3_err);
        return JQ_ERETRY_INTERNAL;
    }

    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
    h->timing.ms_total = ms;
    double fwd = 0.0, bwd = 0.0;
    const size_t nfwd = (size_t)nchunks;
    const bool show = debug_timing();      // development aid: every propagator launch on stderr
    for (size_t i = 2, k = 0; i + 1 < evi; i += 2, ++k) {
        HIPCHK(h, hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]));
        if (show) fprintf(stderr, "jq launch %zu (%s): %.3f ms\n", k, k < (size_t)nchunks ? "forward" : "backward", ms);
        if (k < nfwd)
            fwd += ms;
        else
            bwd += ms;
    }
    if (cq3 && bwd > 0.0) {
        h->cq3_us_per_step = 1.0e3 * bwd / ((double)h->nsteps * (two_pass ? 2 : 1) * ngroups);
        h->cq3_busy_streak = 0;
    }
    h->timing.ms_forward = fwd;
    h->timing.ms_backward = bwd;
    h->timing.ms_propagate = fwd + bwd;
    h->timing.ms_generate = h->timing.ms_total - (fwd + bwd);
    h->timing.n_forward_launches = (long long)nfwd;
    h->timing.n_backward_launches = (long long)((evi - 2) / 2 - nfwd);
    // (JQ_BW_T4: one v_mfma_f64_4x4x4_4b is 512 FLOP, a quarter of the 16x16x4 instruction this counter is quoted in)
    h->timing.mfma_executed = imr ? 0 : (!coop && !lane && !rl && h->BW == JQ_BW_T4) ? mfma / 4 : mfma;   // (the iteration counts of the implicit-midpoint solver are data dependent)
    h->timing.mfma_backward = h->timing.mfma_executed == 0 ? 0 : (h->timing.mfma_executed == mfma ? mfma - mfma_fwd : (mfma - mfma_fwd) / 4);
    h->timing.svts = (long long)nsamples * h->N * h->nsteps;
    h->timing.kernel_family = imr_cq ? 9 : cq ? 8 : imr_quad ? 7 : imr_coop ? 5 : imr ? 4 : rl ? 3 : lane ? 2 : coop ? 1 : quad ? 6 : 0;
    h->timing.kernel_size = rl ? h->rl_npj : lane ? h->lane_np : h->NT;
    h->timing.kernel_band = (rl || lane) ? 0 : coop ? h->BWc : (quad || cq) ? JQ_BW_T4Q : h->BW;
    h->timing.kernel_variant = cq3 ? cq_nr : qsplit ? 20 + qs_qw : (rl && rl_split) ? 32 : 0;      // (workgroups per column quad in the backward sweep of the cooperative-quad kernels)
    h->timing.ms_allreduce = 0.0;
    h->timing.ms_shard_min = h->timing.ms_shard_max = h->timing.ms_total;
    return JQ_OK;
}

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


// ---------------------------------------------------------------------------------------------
// Multi-device handle: ONE process (the single-threaded Julia caller of src/ipopt_interface.jl:38-65) drives ndev GPUs.
// The quadrature nodes of eval_f_g_grad! are block-partitioned over the devices (jq_shard_bounds), every device evaluates
// its shard concurrently (one host thread per device, each on its device's own stream) and the packed results
// [infidelity, leak, grad_infid(nCoeff), grad_leak(nCoeff)] are summed with ONE ncclAllReduce (RCCL over xGMI).
// librccl is loaded at run time (only multi-device callers need it): the copy that belongs to the HIP runtime in use (load_rccl).
struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;      // (optional)
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;      // (optional)
};
static RcclApi g_rccl;

static int load_rccl(std::string* err)
{
    if (g_rccl.lib) return JQ_OK;
    // RCCL must sit on the SAME HIP / HSA runtime as this library.  A process may carry two ROCm copies -- PyTorch ships
    // libamdhip64, libhsa-runtime64 and librccl side by side, and `import torch` maps them without initialising them -- and an
    // RCCL on the other copy finds an uninitialised HSA runtime ("no ROCm-capable device is detected").  So the librccl NEXT TO
    // the HIP runtime this library is bound to comes first (whether or not it is mapped already), then any librccl that is
    // mapped, then the loader's search path.
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* lib = nullptr;
    // JQ_RCCL_LIB=<path>: load exactly this file (deployments with RCCL elsewhere; the tests point it at a missing file to
    // check that a failing load is an error code, not a crash)
    const char* forced = getenv("JQ_RCCL_LIB");
    if (forced && *forced) {
        lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        if (!lib) {
            const char* e = dlerror();      // (ONE call: dlerror() clears the pending message)
            *err = std::string("jq_create_multi: cannot load librccl from JQ_RCCL_LIB (") + (e ? e : "?") + ")";
            return JQ_EUNSUPPORTED;
        }
    }
    if (!lib) {
        Dl_info di;
        if (dladdr((void*)&hipGetDeviceCount, &di) && di.dli_fname) {
            std::string dir(di.dli_fname);
            const size_t sl = dir.rfind('/');
            if (sl != std::string::npos) {
                dir.resize(sl + 1);
                for (const char* n : {"librccl.so.1", "librccl.so"})
                    if ((lib = dlopen((dir + n).c_str(), RTLD_NOW | RTLD_LOCAL))) break;
            }
        }
    }
    for (const char* n : names) {
        if (lib) break;
        lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    }
    for (const char* n : names) {
        if (lib) break;
        lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    }
    if (!lib) {
        const char* e = dlerror();      // (ONE call: dlerror() clears the pending message, a second call returns NULL)
        *err = std::string("jq_create_multi: cannot load librccl (") + (e ? e : "?") + ")";
        return JQ_EUNSUPPORTED;
    }
    RcclApi a;
    a.lib = lib;
    a.CommInitAll = (decltype(a.CommInitAll))dlsym(lib, "ncclCommInitAll");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(lib, "ncclCommDestroy");
    a.CommAbort = (decltype(a.CommAbort))dlsym(lib, "ncclCommAbort");
    a.AllReduce = (decltype(a.AllReduce))dlsym(lib, "ncclAllReduce");
    a.GroupStart = (decltype(a.GroupStart))dlsym(lib, "ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))dlsym(lib, "ncclGroupEnd");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(lib, "ncclGetErrorString");
    a.CommCount = (decltype(a.CommCount))dlsym(lib, "ncclCommCount");
    if (!a.CommInitAll || !a.CommDestroy || !a.AllReduce || !a.GroupStart || !a.GroupEnd || !a.GetErrorString) {
        *err = "jq_create_multi: librccl lacks a required symbol";
        return JQ_EUNSUPPORTED;
    }
    g_rccl = a;
    return JQ_OK;
}

#define NCCLCHK(h, call)                                                                                      \
    do {                                                                                                      \
        ncclResult_t r_ = (call);                                                                             \
        if (r_ != ncclSuccess) {                                                                              \
            char buf_[512];                                                                                   \
            snprintf(buf_, sizeof buf_, "RCCL error '%s' at %s:%d (%s)", g_rccl.GetErrorString(r_), __FILE__, __LINE__, #call); \
            (h)->err = buf_;                                                                                  \
            return JQ_EHIP;                                                                                   \
        }                                                                                                     \
    } while (0)

extern "C" int jq_shard_bounds(int32_t nquad, int32_t rank, int32_t world, int32_t* lo, int32_t* hi)
{
    if (!lo || !hi || nquad < 0 || world < 1 || rank < 0 || rank >= world) return JQ_EINVAL;
    const int base = nquad / world, rem = nquad % world;
    *lo = rank * base + std::min(rank, rem);
    *hi = *lo + base + (rank < rem ? 1 : 0);
    return JQ_OK;
}

extern "C" int jq_num_devices(const jq_handle* h) { return !h ? 0 : h->subs.empty() ? 1 : (int)h->subs.size(); }

extern "C" int jq_handle_device(const jq_handle* h) { return h ? h->device : -1; }

extern "C" int jq_num_compute_units(const jq_handle* h) { return !h ? 0 : h->subs.empty() ? h->num_cu : h->subs[0]->num_cu; }

static void destroy_multi(jq_handle* h)
{
    int prev = 0;
    const bool have_prev = hipGetDevice(&prev) == hipSuccess;
    for (size_t d = 0; d < h->comms.size(); ++d)
        if (h->comms[d] && g_rccl.CommDestroy) {
            (void)hipSetDevice(h->subs[d]->device);
            if (h->comm_broken && g_rccl.CommAbort) (void)g_rccl.CommAbort(h->comms[d]);
            else (void)g_rccl.CommDestroy(h->comms[d]);
        }
    for (jq_handle* sub : h->subs) jq_destroy(sub);
    if (have_prev) (void)hipSetDevice(prev);
    delete h;
}

extern "C" int jq_create_multi(const jq_problem* problem, const int32_t* devices, int32_t ndev, jq_handle** out)
{
    return jq_create_multi_opts(problem, devices, ndev, nullptr, out);
}

extern "C" int jq_create_multi_opts(const jq_problem* problem, const int32_t* devices, int32_t ndev, const char* options, jq_handle** out)
{
    if (!out) {
        g_create_error = "jq_create_multi: out is NULL";
        return JQ_EINVAL;
    }
    *out = nullptr;
    JqOptions opt;
    if (int rc0 = parse_create_options(options, &opt)) return rc0;
    int avail = 0;
    if (hipGetDeviceCount(&avail) != hipSuccess) avail = 0;
    // option multi_same_device=1 (TEST MODE, tests/test_gpu_round3.py): the `ndev` sub-handles may share physical GPUs (device id
    // taken modulo the visible count, ndev <= 16) -- own streams, own host threads, the same sharding and packing code -- and the
    // ONE step that needs distinct devices, the ncclAllReduce, is replaced by a host-side sum of the devices' packed vectors in
    // device order.  This is how the ndev > 1 code runs on a one-GPU box; it is not a production path (no speed-up).
    const bool same_dev = opt.on(O_MULTI_SAME_DEVICE);
    if (ndev < 1 || (same_dev ? (avail < 1 || ndev > 16) : ndev > avail)) {
        char buf[160];
        snprintf(buf, sizeof buf, "jq_create_multi: ndev = %d but %d HIP device(s) are visible", ndev, avail);
        g_create_error = buf;
        return JQ_EINVAL;
    }
    std::vector<int> devs(ndev);
    for (int d = 0; d < ndev; ++d) {
        devs[d] = devices ? devices[d] : d;
        if (same_dev && devs[d] >= 0) devs[d] %= avail;
        if (devs[d] < 0 || devs[d] >= avail || (!same_dev && std::count(devs.begin(), devs.begin() + d, devs[d]))) {
            g_create_error = "jq_create_multi: device ids must be distinct and < jq_device_count()";
            return JQ_EINVAL;
        }
    }
    DeviceGuard guard;
    jq_handle* h = new (std::nothrow) jq_handle();
    if (!h) {
        g_create_error = "jq_create_multi: out of host memory";
        return JQ_ENOMEM;
    }
    h->host_reduce = same_dev;
    h->opt = opt;
    int rc = JQ_OK;
    for (int d = 0; d < ndev && rc == JQ_OK; ++d) {
        if (hipSetDevice(devs[d]) != hipSuccess) {
            g_create_error = "jq_create_multi: hipSetDevice failed";
            rc = JQ_EHIP;
            break;
        }
        jq_handle* sub = nullptr;
        rc = create_with(problem, opt, &sub);      // (sets g_create_error on failure)
        if (rc == JQ_OK) h->subs.push_back(sub);
    }
    if (rc == JQ_OK && !h->host_reduce) {
        std::string err;
        rc = load_rccl(&err);
        if (rc != JQ_OK) g_create_error = err;
    }
    if (rc == JQ_OK && !h->host_reduce) {
        h->comms.assign(ndev, nullptr);
        const ncclResult_t r = g_rccl.CommInitAll(h->comms.data(), ndev, devs.data());
        if (r != ncclSuccess) {
            g_create_error = std::string("jq_create_multi: ncclCommInitAll failed: ") + g_rccl.GetErrorString(r);
            h->comms.clear();
            rc = JQ_EHIP;
        }
    }
    if (rc != JQ_OK) {
        if (h->subs.empty()) delete h; else destroy_multi(h);
        return rc;
    }
    const jq_handle* s0 = h->subs[0];
    h->device = s0->device;
    h->Ntot = s0->Ntot; h->N = s0->N; h->Nc = s0->Nc; h->Nfreq = s0->Nfreq; h->nsteps = s0->nsteps; h->objFuncType = s0->objFuncType;
    h->T = s0->T;
    *out = h;
    return JQ_OK;
}

// apply f to every device handle; the first failure is reported on the multi handle
template <typename F>
static int multi_forall(jq_handle* h, F f)
{
    for (jq_handle* sub : h->subs) {
        const int rc = f(sub);
        if (rc != JQ_OK) {
            h->err = sub->err;
            return rc;
        }
    }
    return JQ_OK;
}

// timing of a multi-device call: the slowest device's times, work summed over the devices
static void multi_timing(jq_handle* h, double ms_allreduce)
{
    jq_timing t = {};
    bool first = true;
    double smin = 0.0, smax = 0.0;
    for (const jq_handle* sub : h->subs) {
        const jq_timing& u = sub->timing;
        if (u.svts == 0) continue;     // device without a shard in the last call
        smin = first ? u.ms_total : std::min(smin, u.ms_total);
        smax = first ? u.ms_total : std::max(smax, u.ms_total);
        if (first || u.ms_total > t.ms_total) {
            const long long mf = t.mfma_executed, mb = t.mfma_backward, sv = t.svts;
            t = u;
            t.mfma_executed = mf;
            t.mfma_backward = mb;
            t.svts = sv;
        }
        t.mfma_executed += u.mfma_executed;
        t.mfma_backward += u.mfma_backward;
        t.svts += u.svts;
        first = false;
    }
    t.ms_allreduce = ms_allreduce;
    t.ms_shard_min = smin;
    t.ms_shard_max = smax;
    h->timing = t;
}

// The comparison of the all-reduce self-check: `got` (what the collective returned) against `expect` (the sum of the devices' packed
// vectors in device order).  Two summation orders differ by rounding errors that scale with the PARTIAL sums, not with the total --
// near a converged risk-neutral optimum the devices' partial gradients (~ 1e-3) cancel to a total of ~ 1e-5 -- so the bound is
// 1e-13 x sum over the devices of their largest entry (round 4 scaled by the total's largest entry: a spurious failure waiting for a
// restart from an optimised pcof).  A non-finite result is reported as such, not as a mismatch.  Returns an empty string when fine.
static std::string allreduce_check(const std::vector<double>& expect, const std::vector<double>& got, double partial_scale, int nd)
{
    char buf[320];
    double worst = 0.0;
    for (size_t i = 0; i < expect.size(); ++i) {
        if (!std::isfinite(got[i]) || !std::isfinite(expect[i])) {
            snprintf(buf, sizeof buf, "non-finite entry in the ensemble result (entry %zu: all-reduce %g, host-order sum of the %d devices' packed "
                                      "vectors %g): an evaluation diverged or produced NaN -- not a fault of the collective", i, got[i], nd, expect[i]);
            return buf;
        }
        worst = std::max(worst, std::fabs(got[i] - expect[i]));
    }
    if (!(worst <= 1e-13 * partial_scale)) {
        snprintf(buf, sizeof buf, "RCCL all-reduce self-check failed: result differs from the host-order sum of the %d devices' packed "
                                  "vectors by %.3e (sum of the devices' largest entries %.3e); option rccl_selfcheck=0 disables the check", nd, worst, partial_scale);
        return buf;
    }
    return std::string();
}

static int multi_eval_f_g_grad(jq_handle* h, const double* pcof, int ncoeff, const double* nodes, const double* weights, int nquad,
                               const double* shift, bool adjoint, double* out2, double* infid_grad, double* leak_grad)
{
    if (h->comm_broken)
        return fail(h, JQ_EHIP, "jq_eval_f_g_grad: an earlier RCCL failure left the communicators of this handle unusable; destroy it");
    DeviceGuard guard;
    const int nd = (int)h->subs.size();
    const size_t npk = 2 + 2 * (size_t)ncoeff;
    std::vector<int> rcs(nd, JQ_OK);
    std::vector<std::vector<double>> hostpk(h->host_reduce ? nd : 0);
    std::vector<std::thread> th;
    for (int d = 0; d < nd; ++d)
        th.emplace_back([&, d]() {
            jq_handle* sub = h->subs[d];
            int lo = 0, hi = 0;
            jq_shard_bounds(nquad, d, nd, &lo, &hi);
            sub->timing = jq_timing{};
            auto body = [&]() -> int {
                HIPCHK(sub, hipSetDevice(sub->device));
                if (int rc = dev_grow(sub, &sub->d_pack, &sub->cap_pack, npk)) return rc;
                if (hi > lo) {
                    EvalOut o;
                    if (int rc = run_eval(sub, pcof, ncoeff, hi - lo, nodes + lo, weights + lo, shift, adjoint, nullptr, nullptr, &o, sub->d_pack)) return rc;
                } else {
                    HIPCHK(sub, hipMemsetAsync(sub->d_pack, 0, npk * sizeof(double), sub->stream));   // no shard: contributes zeros
                    HIPCHK(sub, hipStreamSynchronize(sub->stream));
                }
                if (h->host_reduce) {      // (test mode: the packed vector goes to the host instead of into an all-reduce)
                    hostpk[d].resize(npk);
                    HIPCHK(sub, hipMemcpyAsync(hostpk[d].data(), sub->d_pack, npk * sizeof(double), hipMemcpyDeviceToHost, sub->stream));
                    HIPCHK(sub, hipStreamSynchronize(sub->stream));
                }
                return JQ_OK;
            };
            rcs[d] = body();
        });
    for (auto& t : th) t.join();
    for (int d = 0; d < nd; ++d)
        if (rcs[d] != JQ_OK) {
            h->err = h->subs[d]->err;
            return rcs[d];
        }
    std::vector<double> packed(npk, 0.0);
    // Self-check of the collective (the first 8-GPU run verifies itself): on the FIRST all-reduce of a handle the devices' packed
    // vectors are also copied to the host before the collective and their sum in device order is compared with what RCCL returns
    // (1e-13 relative to the largest entry: the ring order differs from the device order in the last bits only).
    // option rccl_selfcheck=0 switches it off, =2 checks every call.
    const int selfcheck = (int)h->opt.get(O_RCCL_SELFCHECK);
    // (option rccl_selfcheck=3 in the same-device test mode, where no collective runs: the comparison itself is exercised -- the host-order
    //  sum against the sum in REVERSE device order, i.e. two legitimate summation orders -- so that its tolerance has run somewhere)
    const bool check_now = (!h->host_reduce && (selfcheck >= 2 || (selfcheck == 1 && h->rccl_checks == 0))) || (h->host_reduce && selfcheck == 3);
    std::vector<double> expect;
    double partial_scale = 0.0;
    if (check_now) {
        expect.assign(npk, 0.0);
        std::vector<double> tmp(npk);
        for (int d = 0; d < nd; ++d) {
            jq_handle* sub = h->subs[d];
            const double* src = tmp.data();
            if (h->host_reduce) {
                src = hostpk[nd - 1 - d].data();      // (reverse order)
            } else {
                HIPCHK(h, hipSetDevice(sub->device));
                HIPCHK(h, hipMemcpyAsync(tmp.data(), sub->d_pack, npk * sizeof(double), hipMemcpyDeviceToHost, sub->stream));
                HIPCHK(h, hipStreamSynchronize(sub->stream));
            }
            double mx = 0.0;
            for (size_t i = 0; i < npk; ++i) {
                expect[i] += src[i];
                if (std::isfinite(src[i])) mx = std::max(mx, std::fabs(src[i]));
            }
            partial_scale += mx;
        }
    }
    const auto t0 = std::chrono::steady_clock::now();
    if (h->host_reduce) {
        for (int d = 0; d < nd; ++d)      // fixed order: device 0, 1, ...
            for (size_t i = 0; i < npk; ++i) packed[i] += hostpk[d][i];
    } else {
        // ONE all-reduce (sum, fp64) of the packed vector over the devices.  Errors inside the group are collected: the group
        // is ALWAYS closed (an open group would make the next collective on these communicators hang), then the first error
        // is reported and the communicators are marked unusable.
        std::string first_err;
        auto note = [&](const char* what, const char* msg) {
            if (first_err.empty()) first_err = std::string(what) + ": " + msg;
        };
        ncclResult_t r = g_rccl.GroupStart();
        if (r != ncclSuccess) {
            h->comm_broken = true;
            h->err = std::string("RCCL error in ncclGroupStart: ") + g_rccl.GetErrorString(r);
            return JQ_EHIP;
        }
        for (int d = 0; d < nd; ++d) {
            jq_handle* sub = h->subs[d];
            const hipError_t e = hipSetDevice(sub->device);
            if (e != hipSuccess) {
                note("hipSetDevice", hipGetErrorString(e));
                continue;
            }
            r = g_rccl.AllReduce(sub->d_pack, sub->d_pack, npk, ncclDouble, ncclSum, h->comms[d], sub->stream);
            if (r != ncclSuccess) note("ncclAllReduce", g_rccl.GetErrorString(r));
        }
        r = g_rccl.GroupEnd();
        if (r != ncclSuccess) note("ncclGroupEnd", g_rccl.GetErrorString(r));
        if (!first_err.empty()) {
            h->comm_broken = true;
            h->err = "RCCL all-reduce failed (" + first_err + ")";
            return JQ_EHIP;
        }
        for (int d = nd - 1; d >= 0; --d) {
            jq_handle* sub = h->subs[d];
            HIPCHK(h, hipSetDevice(sub->device));
            if (d == 0) HIPCHK(h, hipMemcpyAsync(packed.data(), sub->d_pack, npk * sizeof(double), hipMemcpyDeviceToHost, sub->stream));
            HIPCHK(h, hipStreamSynchronize(sub->stream));
        }
    }
    const double ms_ar = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (check_now) {
        const std::string bad = allreduce_check(expect, packed, partial_scale, nd);
        if (!bad.empty()) return fail(h, JQ_EHIP, bad.c_str());
        ++h->rccl_checks;
    }
    out2[0] = packed[0];
    out2[1] = packed[1];
    if (adjoint)
        for (int i = 0; i < ncoeff; ++i) {
            infid_grad[i] = packed[2 + i];
            leak_grad[i] = packed[2 + (size_t)ncoeff + i];
        }
    multi_timing(h, ms_ar);
    return JQ_OK;
}

static int multi_traceobj_sweep(jq_handle* h, const double* pcof, int ncoeff, const double* nodes, int nquad, const double* shift,
                                double* out)
{
    DeviceGuard guard;
    const int nd = (int)h->subs.size();
    std::vector<int> rcs(nd, JQ_OK);
    std::vector<std::thread> th;
    for (int d = 0; d < nd; ++d)
        th.emplace_back([&, d]() {
            jq_handle* sub = h->subs[d];
            int lo = 0, hi = 0;
            jq_shard_bounds(nquad, d, nd, &lo, &hi);
            sub->timing = jq_timing{};
            if (hi > lo) rcs[d] = jq_traceobj_sweep(sub, pcof, ncoeff, nodes + lo, hi - lo, shift, out + (size_t)4 * lo);
        });
    for (auto& t : th) t.join();
    for (int d = 0; d < nd; ++d)
        if (rcs[d] != JQ_OK) {
            h->err = h->subs[d]->err;
            return rcs[d];
        }
    multi_timing(h, 0.0);
    return JQ_OK;
}

extern "C" int jq_set_option(jq_handle* h, const char* name, int64_t value)
{
    if (!h) return JQ_EINVAL;
    if (!name) return fail(h, JQ_EINVAL, "jq_set_option: NULL name");
    const int o = JqOptions::find(name, strlen(name));
    if (o < 0) return fail(h, JQ_EINVAL, (std::string("jq_set_option: unknown option '") + name + "'").c_str());
    const long long v = (value == JQ_OPTION_DEFAULT) ? JQ_OPT_UNSET : (long long)value;
    if (!h->subs.empty()) {
        std::string err;
        if (!h->opt.set(o, v, &err)) return fail(h, JQ_EUNSUPPORTED, ("jq_set_option: " + err).c_str());
        if (o == O_MULTI_SAME_DEVICE) return fail(h, JQ_EINVAL, "jq_set_option: multi_same_device is an option of jq_create_multi_opts");
        return multi_forall(h, [&](jq_handle* sub) { return jq_set_option(sub, name, value); });
    }
    const long long old = h->opt.v[o];
    if (old == v) return JQ_OK;
    std::string err;
    if (!h->opt.set(o, v, &err)) return fail(h, JQ_EUNSUPPORTED, ("jq_set_option: " + err).c_str());
    if (g_jq_opt[o].flags & JQ_OPT_PLAN) {      // shapes the plan: plan again from the handle's own copy of the problem
        HIPCHK(h, hipSetDevice(h->device));
        const std::vector<double> H0 = h->Hconst;
        const int rc = replan(h, H0.data());
        if (rc != JQ_OK) {
            h->opt.v[o] = old;
            return rc;
        }
        h->replanned = false;      // (an option change is not a drift outside the planned structure)
        return JQ_OK;
    }
    if (h->emb) h->emb->opt = h->opt;
    return JQ_OK;
}

extern "C" int jq_get_option(const jq_handle* h, const char* name, int64_t* value)
{
    if (!h || !name || !value) return JQ_EINVAL;
    const int o = JqOptions::find(name, strlen(name));
    if (o < 0) return JQ_EINVAL;
    const long long v = h->opt.get(o);
    *value = (v == JQ_OPT_UNSET) ? JQ_OPTION_DEFAULT : (int64_t)v;
    return JQ_OK;
}

// ranks of the RCCL communicator behind a multi-device handle (ncclCommCount of its first communicator): what the first real
// multi-GPU run prints to show that RCCL saw every device.  0: no communicator (single-device handle, same-device test mode).
extern "C" int jq_rccl_world_size(const jq_handle* h)
{
    if (!h || h->comms.empty() || !h->comms[0] || !g_rccl.CommCount) return 0;
    int n = 0;
    if (g_rccl.CommCount(h->comms[0], &n) != ncclSuccess) return -1;
    return n;
}

extern "C" int jq_plan_info(const jq_handle* hh, char* buf, int32_t buflen)
{
    if (!hh || (!buf && buflen > 0) || buflen < 0) return JQ_EINVAL;
    const jq_handle* h = hh->subs.empty() ? hh : hh->subs[0];
    std::string o = "{";
    auto kv = [&](const char* k, const std::string& v, bool quote = false) {
        if (o.size() > 1) o += ", ";
        o += std::string("\"") + k + "\": " + (quote ? "\"" + v + "\"" : v);
    };
    auto num = [](long long v) { return std::to_string(v); };
    kv("devices", num(hh->subs.empty() ? 1 : (long long)hh->subs.size()));
    kv("Ntot", num(h->Ntot));
    kv("N", num(h->N));
    kv("controls", num(h->Nc));
    kv("control_groups", num(ctrl_ngroups(h->Nc)));
    kv("tile_rows", num(h->NT));
    kv("compute_units", num(h->num_cu));
    const char* structure = h->big ? (h->BWc == 15 ? "dense" : "band") : h->BW == JQ_BW_T4 ? "t4" : h->BW == JQ_BW_OD ? "od" : h->BW == h->NT - 1 ? "dense" : "band";
    kv("structure", structure, true);
    kv("block_band", num(h->big ? h->BWc : h->BW));
    kv("embedded_twin_Ntot", num(h->emb ? h->emb->Ntot : 0));
    kv("integrator", h->integrator == 2 ? "implicit_midpoint" : "stormer_verlet", true);
    kv("linear_solver", h->solver_id == 2 ? "jacobi" : "neumann", true);
    kv("neumann_terms_or_max_iter", num(h->integrator == 2 ? h->imr_max_iter : h->m));
    kv("chunk_steps", num(h->chunk_steps));
    kv("replanned", h->replanned ? "true" : "false");
    // kernel families in the order run_eval considers them for a Stormer-Verlet / Neumann batch (the embedded twin, if any, serves
    // the batches beyond the row-lane / lane range with ITS plan)
    std::string fam = "[";
    auto add = [&](int id, const char* name, const char* unit, long long mx) {
        if (fam.size() > 1) fam += ", ";
        fam += std::string("{\"family\": ") + std::to_string(id) + ", \"name\": \"" + name + "\", \"max_" + unit + "\": " + std::to_string(mx) + "}";
    };
    const jq_handle* t = h->emb ? h->emb : h;
    if (h->rl_npj > 0) add(3, "row-lane (VALU, lane per (row, column); backward sweep on two waves)", "columns", h->rl_max_cols);
    if (h->lane_np > 0) add(2, "lane (VALU, lane per column)", "columns", h->lane_max_cols);
    if (t->cq_max_quads > 0) add(8, "cooperative quad (one 16-row block per wave)", "quads", t->cq_max_quads);
    if (t->quad_max_slabs > 0) add(6, "quad layout (four columns per wave; 1 / 2 / 3 slabs per workgroup by round count)", "slabs", t->quad_max_slabs);
    if (t->coop_ok && t->NT >= 2) add(1, "cooperative (tile row per wave)", "slabs", t->coop_max_slabs);
    if (!t->big) add(0, "slab (wave per 16-column slab)", "slabs", 1LL << 30);
    fam += "]";
    kv("families", fam);
    {   // the objects this handle's kernels come from, as the build manifest records them (register form, registers, scratch)
        const std::string man(jq_build_manifest);
        std::vector<std::string> tags;
        auto tag = [&](const char* prefix, int a, int b) {
            char buf[32];
            if (b >= 0) snprintf(buf, sizeof buf, "%s_%d_%d", prefix, a, b);
            else snprintf(buf, sizeof buf, "%s_%d", prefix, a);
            tags.push_back(buf);
        };
        for (const jq_handle* x : {h, (const jq_handle*)h->emb}) {
            if (!x) continue;
            if (x->BW == JQ_BW_T4) {
                for (const char* pre : {"k", "s", "p", "u", "w", "q", "v"}) tag(pre, x->NT, JQ_BW_T4Q);
                tag("k", x->NT, JQ_BW_T4);
            } else if (!x->big) {
                tag("k", x->NT, x->BW);
                tag("j", x->NT, x->BW);
            }
            if (x->mat_elems_c > 0) tag("c", x->NT, x->BWc), tag("i", x->NT, x->BWc);
            if (x->rl_npj > 0) tag("r", x->rl_npj, -1), tag("m", x->rl_npj, -1);
            if (x->lane_np > 0) tag("l", x->lane_np, -1);
        }
        std::string objs = "{";
        for (const std::string& t : tags) {
            const std::string key = "\"" + t + "\": {";
            const size_t at = man.find(key);
            if (at == std::string::npos) continue;
            const size_t end = man.find('}', at);
            if (end == std::string::npos) continue;
            if (objs.size() > 1) objs += ", ";
            objs += man.substr(at, end - at + 1);
        }
        objs += "}";
        std::string hipcc = "null";      // the compiler the kernel objects came from (build manifest)
        {
            const size_t at = man.find("\"hipcc\": {");
            const size_t end = at == std::string::npos ? at : man.find('}', at);
            if (end != std::string::npos) hipcc = man.substr(at + 9, end - at - 8);
        }
        kv("build", std::string("{\"manifest\": ") + (man.size() > 2 ? "true" : "false") + ", \"hipcc\": " + hipcc + ", \"objects\": " + objs + "}");
    }
    kv("full_weight_rank", num(h->wrank));
    kv("options", hh->opt.str(), true);      // the options that are set (jq_create_opts / JQ_OPTIONS / jq_set_option); "" = all defaults
    kv("rccl_selfchecks", num(hh->rccl_checks));      // all-reduces of a multi-device handle verified against the host-order sum
    {   // the three-workgroup latency kernels: what the last batch of the cooperative-quad families decided, and why
        const jq_handle* t2 = h->emb ? h->emb : h;
        const std::string d = t2->cq3_last.empty() ? "no batch of the cooperative-quad families yet" : t2->cq3_last;
        kv("latency_split", std::string("{\"last_decision\": \"") + d + "\", \"faults\": " + num(t2->cq3_faults) + ", \"faults_xcd\": " + num(t2->cq3_faults_xcd) + ", \"abandoned_at_rendezvous\": " + num(t2->cq3_busy) + ", \"cooling_down\": " + num(t2->cq3_skip) +
                                ", \"off\": " + (t2->cq3_off ? "true" : "false") + "}");
    }
    o += "}";
    if (buflen > 0) {
        const size_t n = std::min(o.size(), (size_t)buflen - 1);
        memcpy(buf, o.data(), n);
        buf[n] = 0;
    }
    return (int)o.size();
}

extern "C" int jq_last_timing(const jq_handle* h, jq_timing* t)
{
    if (!h || !t) return JQ_EINVAL;
    *t = h->timing;
    return JQ_OK;
}
