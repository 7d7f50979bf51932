/*
 * juqbox_hip.h -- C ABI of libjuqbox_hip.so: the MI355X (gfx950) replacement for Juqbox.jl's
 * Stormer-Verlet `traceobjgrad` hot path.
 *
 * The reference has NO foreign-function boundary on this path (it is 100% Julia); the seam this
 * ABI plugs into is the multiple-dispatch method
 *     traceobjgrad(pcof0, params::objparams, wa::Working_Arrays, verbose, evaladjoint)
 * (src/evalobjgrad.jl:504) and its ensemble caller eval_f_g_grad! (src/ipopt_interface.jl:24-70).
 * A new working-array type whose traceobjgrad method `ccall`s the entry points below is the
 * drop-in (INTEGRATION.md shows the Julia shim).  File:line citations are relative to the
 * reference repository root.
 *
 * Conventions
 *  - all matrices are Float64, COLUMN-MAJOR (Julia Array layout); indices in comments are 1-based
 *    when they quote the reference.
 *  - every pointer argument is a HOST pointer owned by the caller and only read/written during the
 *    call; the library copies what it keeps.  The handle owns all device memory.
 *  - every function returns 0 on success, a negative JQ_E* code otherwise; jq_last_error() gives
 *    the message.  No C++ exception crosses this boundary.
 *  - one handle = one in-flight call (not re-entrant), like the reference's single-threaded use.
 *  - the handle is bound to the HIP device that was current when jq_create() ran
 *    (jq_set_device() selects it; one process per GPU under torch.distributed / RCCL), or to the
 *    devices given to jq_create_multi() (one process, several GPUs, RCCL inside the library).
 */
#ifndef JUQBOX_HIP_H
#define JUQBOX_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version of this header: the layout of jq_problem / jq_timing and the set of entry points.  jq_abi_version() returns the
 * value the LIBRARY was built with; every binding (juqbox.jl_amd/_lib.py, julia/hip_backend.jl, examples/c_abi_demo.c) compares
 * the two when it loads the library, so a caller compiled against an older header fails loudly instead of having
 * jq_last_timing write past its struct.  History: 1 = round 1; 2 = jq_timing.mfma_backward, jq_problem.Hunc_ops / Rfreq;
 * 3 = jq_timing.ms_allreduce / ms_shard_min / ms_shard_max, jq_abi_version(), up to JQ_MAX_CONTROLS control Hamiltonians;
 * 4 = jq_problem.Hconst_csc / Hsym_csc / Hanti_csc (sparse operator storage), jq_csc, jq_update_hconst_csc, jq_update_wmat (full /
 *     complex leakage weights);
 * 5 = per-handle options instead of environment variables (jq_create_opts, jq_create_multi_opts, jq_set_option, jq_get_option),
 *     jq_timing.reserved renamed kernel_variant, jq_rccl_world_size; no size limits on Ntot, the number of control Hamiltonians or
 *     the rank of a full weight matrix; full leakage weights with the Jacobi solver. */
#define JQ_ABI_VERSION 5

#define JQ_MAX_CONTROLS 16 /* control Hamiltonians the fast kernels hold in registers; more: cooperative kernels (no limit)    */
#define JQ_MAX_WRANK 16    /* rank of a full leakage-weight matrix every kernel family takes; beyond: slab / cooperative kernels */

#define JQ_OK 0
#define JQ_EINVAL -1      /* bad argument (the reference's @assert / error(...) sites)            */
#define JQ_EDIM -2        /* DimensionMismatch: nCoeff != length(pcof) (src/bsplines.jl:178-181)   */
#define JQ_EUNSUPPORTED -3/* valid for the reference but outside what the HIP path implements     */
#define JQ_EHIP -4        /* HIP runtime failure                                                   */
#define JQ_ENOMEM -5

typedef struct jq_handle jq_handle;

/*
 * A sparse operator exactly as Julia's SparseMatrixCSC{Float64,Int64} stores it (the reference keeps Hconst, Hsym_ops, Hanti_ops
 * like this when objparams(...; use_sparse=true), src/evalobjgrad.jl:249-262, and multiplies them through KS!/KS_alloc,
 * :2392-2426, :3072-3105): m x n, column j holds the entries nzval[colptr[j]-1 .. colptr[j+1]-2] in the rows
 * rowval[...] -- colptr and rowval are 1-BASED Int64, like the fields of the Julia struct, so the binding passes the three
 * vectors without copying or shifting them.  Repeated (row, column) entries are summed.
 */
typedef struct jq_csc {
    int64_t m, n;            /* size; must be Ntot x Ntot                                           */
    const int64_t *colptr;   /* [n + 1], colptr[0] == 1, colptr[n] == nnz + 1                       */
    const int64_t *rowval;   /* [nnz], 1 <= rowval <= m                                             */
    const double *nzval;     /* [nnz]                                                               */
} jq_csc;

/*
 * Problem description = the objparams fields the device needs (src/evalobjgrad.jl:53-148;
 * SURVEY.md section 8 row a13).  Hard-wired in the reference and therefore not passed:
 * use_bcarrier=true (:208), pFidType=2 (:164), sv_type=1 (:314), order=2/stages=1 (:507,:644).
 */
typedef struct jq_problem {
    int32_t Ntot;          /* prod(Ne+Ng): Hilbert dimension incl. guard levels                   */
    int32_t N;             /* prod(Ne): number of initial-condition columns                       */
    int32_t Ncoupled;      /* number of (Hsym_ops[k], Hanti_ops[k]) control pairs, <= JQ_MAX_CONTROLS (with more than four the
                              backward sweep runs once per group of four controls)                   */
    int32_t Nfreq;         /* carrier frequencies per control = size(Cfreq,2)                     */
    int32_t nsteps;        /* params.nsteps                                                        */
    int32_t neumann_terms; /* params.linear_solver.max_iter (NEUMANN_SOLVER, linear_solvers.jl:37) */
    int32_t objFuncType;   /* 1: infidelity+leak; 2/3: second, unforced adjoint gives infidelgrad  */
    int32_t Nunc;          /* number of uncoupled controls Hunc_ops (0, or > 0 with Ncoupled == 0: src/evalobjgrad.jl:176) */
    double T;              /* gate duration                                                        */
    const double *Hconst;    /* [Ntot x Ntot]                                                      */
    const double *Hsym_ops;  /* [Ncoupled][Ntot x Ntot]                                            */
    const double *Hanti_ops; /* [Ncoupled][Ntot x Ntot]                                            */
    const double *Uinit;     /* [Ntot x N]                                                         */
    const double *Utarget_r; /* [Ntot x N]  real(Utarget) (rotating frame)                         */
    const double *Utarget_i; /* [Ntot x N]  imag(Utarget)                                          */
    const double *wmat_real_diag; /* [Ntot] diag(params.wmat_real); Diagonal weights only          */
    const double *Cfreq;     /* [(Ncoupled + Nunc) x Nfreq] carrier (angular) frequencies          */
    /* Uncoupled controls (the lab-frame evaluation of a pulse, e.g. examples/cnot2-lab.jl): KS! adds
     * 2 (p_q(t) cos(2 pi Rfreq[q] t) - q_q(t) sin(2 pi Rfreq[q] t)) * Hunc_ops[q] to K when Hunc_ops[q] is symmetric,
     * to S when it is antisymmetric (src/evalobjgrad.jl:2373-2387; anything else is the reference's ArgumentError,
     * :186-196 -> JQ_EINVAL).  Parity-unpinned in the reference (no golden): checked against the CPU oracle.
     * Gradients (Stormer-Verlet integrator): the exact gradient of the discrete objective, the traces of Hunc_ops[q] weighted
     * with grad ft = 2 cos(.) grad p_q - 2 sin(.) grad q_q.  The reference's own code for this (adjoint_grad_calc!, :2620-2656)
     * is not a usable reference -- it differentiates control functions of an older numbering (func = 2 Ncoupled - 1 + q, no
     * rotation factor), i.e. not what KS! applies, and throws for objFuncType != 1 (gradSize, :801) -- so this gradient is
     * pinned by finite differences of the objective (tests/test_uncoupled.py).  Implicit midpoint: JQ_EUNSUPPORTED (the
     * reference's adjoint_grad_calc_m has no term for uncoupled controls). */
    const double *Hunc_ops;  /* [Nunc][Ntot x Ntot], NULL when Nunc == 0                           */
    const double *Rfreq;     /* [Nunc] rotation frequencies params.Rfreq, NULL when Nunc == 0      */
    /* Sparse storage of the coupled operators (use_sparse = true): when Hconst == NULL, Hconst_csc is read instead; likewise
     * Hsym_csc / Hanti_csc (arrays of Ncoupled descriptors) when Hsym_ops / Hanti_ops == NULL.  The library plans its kernels
     * from the nonzero structure either way, so the dense and the CSC form of the same operators give bit-identical results.
     * All three NULL (with the dense pointers set): a dense problem, as before. */
    const jq_csc *Hconst_csc;
    const jq_csc *Hsym_csc;  /* [Ncoupled]                                                         */
    const jq_csc *Hanti_csc; /* [Ncoupled]                                                         */
} jq_problem;

/* Timing of the last evaluation, measured with HIP events on the library's stream. */
typedef struct jq_timing {
    double ms_total;        /* whole call on the device (generate + propagate + reduce)            */
    double ms_propagate;    /* sum over launches of the forward+backward propagator kernels        */
    double ms_generate;     /* everything else: control evaluation, K(t)/S(t) tile stream, reductions*/
    double ms_forward;      /* sum over k_forward launches                                         */
    double ms_backward;     /* sum over k_backward launches                                        */
    int64_t n_forward_launches;
    int64_t n_backward_launches;
    int64_t mfma_executed;  /* v_mfma_f64_16x16x4 instructions issued by the propagators (all waves)*/
    int64_t svts;           /* state-vector-time-steps processed (columns x nsteps), SURVEY 8(d)   */
    int32_t kernel_family;  /* propagators used: 0 slab (MFMA, wave per slab), 1 cooperative (MFMA, row split),
                               2 lane (VALU, lane per column), 3 row-lane (VALU, lane per (row, column)),
                               4 row-lane, implicit midpoint, 5 cooperative, implicit midpoint,
                               6 quad layout (MFMA 4x4x4, four columns per wave; JQ_BW_T4), 7 the same, implicit midpoint,
                               8 cooperative quad (one 16-row block per wave: single evaluations / small ensembles),
                               9 the same, implicit midpoint (N = 4)                                              */
    int32_t kernel_size;    /* template size parameter: NT (16-row tiles) for 0/1, NP for 2, NPJ for 3       */
    int32_t kernel_band;    /* block band of the MFMA families (9 = JQ_BW_OD: diagonal off-diagonal blocks,
                               8 = JQ_BW_T4: 4x4 diagonal blocks + diagonal couplings, 7 = the same, quad layout,
                               10 = dense blocks on the cooperative-quad kernels: 17 .. 32 levels without the structure) */
    int32_t kernel_variant; /* variant of the backward sweep.  Family 8: workgroups (CUs) per column quad -- 3: state re-integration, adjoint
                               step and trace products pipelined over three workgroups (single evaluations, <= 80 cnot3 samples);
                               2: state re-integration | adjoint step + trace products (81 .. 128 samples); 22: more column quads than
                               CUs, backward sweep on k_backward_qsplit with two quads per workgroup.  Family 6: 24 = one slab per
                               workgroup with the state and the adjoint chain of a quad on two waves (k_backward_qsplit).  Family 3: 32 = the backward
                               sweep's state and adjoint chain on two waves (k_backward_rowlane2), 33 = state chain, adjoint chain and traces on
                               three or four waves (k_backward_rowlane3).  Else 0 */
    int64_t mfma_backward;  /* the part of mfma_executed issued by the k_backward launches                  */
    double ms_allreduce;    /* multi-device handles: host wall time of the ONE all-reduce (group start .. result on the host);
                               0 for single-device handles (their caller runs the collective)                  */
    double ms_shard_min;    /* multi-device handles: smallest / largest ms_total over the devices that had a shard */
    double ms_shard_max;    /* (single-device handles: both = ms_total)                                        */
} jq_timing;

/* ---- lifetime --------------------------------------------------------------------------------*/
int jq_device_count(void);
int jq_set_device(int device);
/* Replaces: objparams(...) constructor + Working_Arrays(params, nCoeff) (src/evalobjgrad.jl:152-343,
 * :405-440): validates sizes, uploads the operators as MFMA A-fragment tile images. */
int jq_create(const jq_problem *problem, jq_handle **out);
/*
 * The same with OPTIONS for the new handle: "name=value,name=value" (integers; ',', ';' or blanks separate; NULL or "" = none).
 * Options replace the JQ_* environment variables of ABI <= 4: they belong to ONE handle, are parsed once, and nothing in the
 * environment changes kernel selection behind the caller's back any more.  The defaults are the measured optimum; options exist for
 * tests (kernel variants that must agree bit for bit), bisection and experiments -- a production caller passes none.  The table of
 * names is in INTEGRATION.md section 4 (generated from juqbox.jl_amd/csrc/jq_options.h).  The ONE environment variable that still
 * reaches kernel selection is JQ_OPTIONS (same syntax, applied in front of `options`): for callers that cannot pass a string, e.g. an
 * unmodified script.  Errors: JQ_EINVAL for an unknown name or a malformed string (message: jq_last_error(NULL)); options that exist
 * in experiment builds only (-DJQ_EXPERIMENTS) are refused by a release library.
 */
int jq_create_opts(const jq_problem *problem, const char *options, jq_handle **out);
/*
 * Change one option of a handle (multi-device handles: of every device).  Options that shape the plan (structure, kernel families,
 * chunking: marked "plan" in the table) re-plan the handle in place -- same pointer, settings kept, like jq_update_hconst with a drift
 * outside the planned structure; the others take effect with the next evaluation.  value == JQ_OPTION_DEFAULT: back to "not set".
 * Errors: JQ_EINVAL unknown name; JQ_EUNSUPPORTED an experiment-only option in a release library.
 */
#define JQ_OPTION_DEFAULT INT64_MIN
int jq_set_option(jq_handle *h, const char *name, int64_t value);
/* the value in force (the default when the option is not set; JQ_OPTION_DEFAULT for an unset option without a default) */
int jq_get_option(const jq_handle *h, const char *name, int64_t *value);
void jq_destroy(jq_handle *h);
/* Message of the last failing call on `h` (h == NULL: last jq_create failure of this thread). */
const char *jq_last_error(const jq_handle *h);

/*
 * Multi-device handle: ONE process drives `ndev` GPUs of a node -- what the single-threaded Julia caller of
 * eval_f_g_grad! (src/ipopt_interface.jl:38-65: a serial loop over the quadrature nodes) needs to use 8 GPUs without
 * MPI.  devices: `ndev` distinct HIP device ids, or NULL for 0..ndev-1.  Every entry point below accepts such a handle:
 *   jq_eval_f_g_grad   block-partitions the nquad nodes over the devices (jq_shard_bounds), evaluates the shards
 *                      concurrently and sums the packed results with ONE ncclAllReduce (RCCL over xGMI, sum, fp64);
 *   jq_traceobj_sweep  partitions the nodes the same way (independent outputs: no collective);
 *   jq_traceobjgrad / jq_state_history / jq_state_populations (ONE evaluation: not shardable) run on the first device;
 *   jq_set_* / jq_update_* apply to every device.
 * librccl.so is loaded at run time by this call (JQ_EUNSUPPORTED if it cannot be found); single-device users never
 * need it.  Errors: JQ_EINVAL if ndev < 1, ndev > jq_device_count() or a device id repeats.
 * The environment variable JQ_RCCL_LIB=<path> makes this call load exactly that librccl.
 * TEST MODE: with the option multi_same_device=1 (jq_create_multi_opts, or JQ_OPTIONS) the `ndev` (<= 16) sub-handles may share physical GPUs (ids modulo
 * the visible count) and the all-reduce is replaced by a host-side sum in device order -- host threads, streams, sharding and
 * packing are the production code.  It exists so that the ndev > 1 paths run on a one-GPU box (tests/test_gpu_round3.py).
 */
int jq_create_multi(const jq_problem *problem, const int32_t *devices, int32_t ndev, jq_handle **out);
/* ... with options (jq_create_opts) for every device's handle; multi_same_device=1 selects the TEST MODE below */
int jq_create_multi_opts(const jq_problem *problem, const int32_t *devices, int32_t ndev, const char *options, jq_handle **out);
/* ranks of the RCCL communicator behind a multi-device handle (ncclCommCount): ndev when RCCL saw every device; 0 for single-device
 * handles and in the same-device test mode (no communicator) */
int jq_rccl_world_size(const jq_handle *h);
/* number of GPUs behind a handle (1 for jq_create handles) */
int jq_num_devices(const jq_handle *h);
/* compute units of the handle's GPU (256 on MI355X): the granularity of the batch-size staircase -- one round of the throughput
 * kernels is 3 slabs (= 3 * (16 / N) samples for N <= 16) per compute unit */
int jq_num_compute_units(const jq_handle *h);
/* HIP device id a handle is bound to (the first device of a multi-device handle; -1 for NULL): device pointers handed to
 * jq_eval_f_g_grad_dev must live on THIS device */
int jq_handle_device(const jq_handle *h);
/*
 * The contiguous block partition of `nquad` ensemble samples over `world` shards (devices of a multi-device handle, or
 * the ranks of a torch.distributed / MPI job with one process per GPU): shard `rank` owns samples [*lo, *hi); the first
 * nquad % world shards get one sample more.  Pure host arithmetic (no GPU needed).
 */
int jq_shard_bounds(int32_t nquad, int32_t rank, int32_t world, int32_t *lo, int32_t *hi);

/* ---- mutations scripts apply to `params` after construction ----------------------------------*/
/* params.linear_solver.max_iter = m (estimate_Neumann!, src/evalobjgrad.jl:2922-2925) */
int jq_set_neumann_terms(jq_handle *h, int32_t m);
/* params.linear_solver = lsolver_object(solver=..., max_iter=..., tol=...) (src/linear_solvers.jl:28-78):
 * solver_id 1 = NEUMANN_SOLVER (neumann!, :81-106; tol ignored), 2 = JACOBI_SOLVER (jacobi!, :110-153; `tol` is
 * the already nrhs-scaled tolerance, :40).  Other ids: JQ_EUNSUPPORTED.
 * JACOBI_SOLVER convergence is tested per evaluation like the reference's norm(T - X) over the Ntot x N block -- for N <= 64 with
 * Ntot <= 96 (round 5: N > 16 columns take ceil(N / 16) slabs; up to four of them are the waves of ONE workgroup, which adds their
 * residual norms before it decides).  With N > 64, or with Ntot > 96 (cooperative kernels: a slab per workgroup), every 16-column
 * part is still tested on its own: parts may stop at different iterations, and the result then differs from the reference's by
 * O(tol) instead of agreeing to rounding (tests/test_gpu_round4.py holds such a case to c * tol). */
int jq_set_linear_solver(jq_handle *h, int32_t solver_id, int32_t max_iter, double tol);
/* params.Integrator_id (src/evalobjgrad.jl:100, constants Stormer_Verlet = 1, Implicit_Midpoint = 2) selects which
 * traceobjgrad method runs: 1 = the Stormer-Verlet path (default), 2 = the implicit-midpoint path
 * (traceobjgrad for Working_Arrays_M, src/evalobjgrad.jl:1042-1481) with the fixed-point solver
 * lsolver_object(solver=JACOBI_SOLVER_M, max_iter, tol) (src/linear_solvers.jl:52-55, :156-270).  For integrator 2 the
 * leakage weights are params.wmat (pass them with jq_update_wmat_diag).  Kernels: row-lane (Ntot <= 16, N <= 4), cooperative MFMA
 * (any Ntot and batch size -- run-time-size kernels beyond 256 levels; both images of a step resident in LDS when they fit, else -- dense 96 x 96, Ntot > 96 -- read from
 * HBM / L2 per product), quad-layout and cooperative-quad kernels for the 4 x 4 x n structure with N = 1, 2, 4.  N > 16 columns per
 * evaluation (round 4): the solver's per-evaluation stopping rule needs an evaluation's columns in one workgroup, so ONE cooperative
 * workgroup per evaluation walks over its 16-column parts (a correctness-first path: the parts' iterates live in HBM / L2). */
int jq_set_integrator(jq_handle *h, int32_t integrator_id, int32_t max_iter, double tol);
/* change_target!(params, new_Utarget) (src/evalobjgrad.jl:1492) */
int jq_update_target(jq_handle *h, const double *Utarget_r, const double *Utarget_i);
/* params.Hconst is mutated freely (src/ipopt_interface.jl:41-44, run_all.jl:13-15).  Kernels, operator images and LDS plan are
 * chosen from the operators' nonzero structure at jq_create; a new drift with entries outside that structure re-plans the handle
 * in place (same pointer, settings kept) -- slower kernels may result, never an error for a valid Hconst. */
int jq_update_hconst(jq_handle *h, const double *Hconst);
/* the same with the new drift in sparse storage (use_sparse = true problems) */
int jq_update_hconst_csc(jq_handle *h, const jq_csc *Hconst);
/* params.wmat_real = orig_wmatsetup(Ne,Ng) (e.g. test/cases/cnot3-setup.jl:253); returns the handle to Diagonal weights */
int jq_update_wmat_diag(jq_handle *h, const double *wmat_real_diag);
/*
 * Full leakage weights: params.wmat_real / params.wmat_imag as Ntot x Ntot matrices, what objparams(...;
 * use_custom_forbidden=true, forb_states, forb_weights) builds (src/evalobjgrad.jl:214-232: W = sum_k w_k f_k f_k^H).  They enter
 * the objective through penalf2aTrap / penalf2a (full versions, :2183-2223) and penalf2imag (:2226-2228) at :700, :716-718 and the
 * adjoint forcing through the products at :862, :882-888.  wmat_imag may be NULL (= 0).
 * The device exploits the structure: W = wmat_real + i wmat_imag must be Hermitian (wmat_real symmetric, wmat_imag
 * antisymmetric -- what the constructor produces) and is eigen-decomposed on the host into its rank terms
 * lam_k f_k f_k^H; W x then costs two column dot products and two axpys per term instead of dense products.  A diagonal
 * wmat_real with wmat_imag == 0 is recognised and takes the Diagonal fast path (as jq_update_wmat_diag).
 * Stormer-Verlet integrator (the implicit-midpoint path of the reference reads params.wmat, which is always Diagonal, :90, :1155), with
 * the Neumann or the Jacobi solver; a rank above JQ_MAX_WRANK or the Jacobi solver route the handle to the slab / cooperative /
 * run-time-size kernels (ABI 5; before: refused).  Errors: JQ_EUNSUPPORTED for a non-Hermitian W or a handle set to the
 * implicit-midpoint integrator; evaluations on kernel families without the low-rank terms are refused, never
 * silently evaluated with other weights.  Parity-unpinned in the reference (no test or golden uses the branch): checked against
 * the CPU oracle, whose gradient is checked by finite differences (tests/test_dense_wmat.py).
 * Latency path (4 x 4 x n structure, at most one column quad per compute unit; round 5): a W that fits FOUR SLOTS -- real (wmat_imag
 * = 0, i.e. real forbidden states) of rank <= 4, or complex of rank <= 2 -- runs on the cooperative-quad kernels (kernel_family 8): one
 * cnot3 evaluation 0.161 s (Diagonal weights: 0.153 s) with the backward sweep on three workgroups per quad, real W on one workgroup
 * (more than 128 samples, or the split not available) 0.204 s.  A complex W needs the split kernels; without them, and any W of higher
 * rank, on the quad-layout kernels (family 6: 0.55 s + ~ 0.1 s per further forbidden state).
 */
int jq_update_wmat(jq_handle *h, const double *wmat_real, const double *wmat_imag);

/* ---- the hot path ----------------------------------------------------------------------------*/
/*
 * Replaces traceobjgrad(pcof0, params, wa, false, evaladjoint) (src/evalobjgrad.jl:504-1038).
 * out4 = { objfv, primaryobjf, secondaryobjf, traceInfidelity } (return tuple :1033/:1035; objfv
 * excludes the Tikhonov term, which the Ipopt callbacks add: src/ipopt_interface.jl:96-98).
 * totalgrad / infidelgrad / leakgrad: caller-allocated length ncoeff, written when evaladjoint != 0
 * (leakgrad is zero-filled and infidelgrad == totalgrad when objFuncType == 1, where the reference
 * returns an empty leakgrad, :948-952).  May be NULL when evaladjoint == 0.
 * Errors: JQ_EINVAL for the reference's error at :604-606, JQ_EDIM for bcparams' DimensionMismatch.
 */
int jq_traceobjgrad(jq_handle *h, const double *pcof, int32_t ncoeff, int32_t evaladjoint, double *out4,
                    double *totalgrad, double *infidelgrad, double *leakgrad);

/*
 * Replaces the verbose branch's state history (src/evalobjgrad.jl:677-680, :748-752, returned at
 * :1031 as usaver + im*usavei): ur/ui are [Ntot x N x (nsteps+1)], ui = -vi.
 */
int jq_state_history(jq_handle *h, const double *pcof, int32_t ncoeff, double *ur, double *ui);
/*
 * The whole verbose, evaladjoint = false call traceobjgrad(pcof0, params, wa, true, false) (return tuple
 * src/evalobjgrad.jl:1031) from ONE forward sweep: the history as above plus out4 = { objfv, primaryobjf,
 * secondaryobjf, traceInfidelity } of the same sweep (out4 may be NULL).
 */
int jq_traceobj_verbose(jq_handle *h, const double *pcof, int32_t ncoeff, double *out4, double *ur, double *ui);

/*
 * Device-side consumers of the state history, so that the [Ntot x N x (nsteps+1)] arrays (199 MB at cnot3)
 * never leave the GPU.  Replaces what the reference derives from usaver/usavei on the host:
 *   pop    : [ngroups x N x nout] (column-major) level populations  sum_{rows r in group g} |psi[r,q,k*every]|^2
 *            for the sampled steps k*every, k = 0..nout-1, nout = nsteps/every + 1.  With group_of_row == NULL
 *            (ngroups == Ntot) these are the curves of plotunitary / plotspecified (src/plotstatectrl.jl);
 *            with group_of_row[r] = third-subsystem index they are marginalize3 (src/plotstatectrl.jl:405-423).
 *            Rows with group_of_row[r] < 0 are skipped.  May be NULL (then ngroups/every/nout are ignored).
 *   maxpop : [Ntot] max over columns and ALL time steps of |psi[r,q,:]|^2 -- the forbidden-level maxima of the
 *            verbose branch (src/evalobjgrad.jl:1004-1018).  May be NULL.
 * Errors: JQ_EINVAL for NULL/size errors (nout must equal nsteps/every + 1, group indices < ngroups).
 */
int jq_state_populations(jq_handle *h, const double *pcof, int32_t ncoeff, const int32_t *group_of_row, int32_t ngroups,
                         int32_t every, int32_t nout, double *pop, double *maxpop);

/*
 * Replaces eval_f_g_grad!(pcof, params, wa, nodes, weights, compute_adjoint)
 * (src/ipopt_interface.jl:24-70): for each quadrature node ep_i the drift Hamiltonian is
 * Hconst + ep_i*diag(shift) and the four weighted sums are accumulated.  All nquad evaluations run
 * concurrently as one batch of N*nquad state columns.
 *   shift  : [Ntot] per-level factor; NULL selects the reference's 0.01*10^(j-2), j=2..Ntot (:41-44)
 *   out2   : { last_infidelity, last_leak } (:58-59)
 *   infid_grad / leak_grad : [ncoeff] weighted sums of infidelgrad / leakgrad (:48-53); leak_grad is
 *            zero-filled for objFuncType == 1.  Ignored (may be NULL) when compute_adjoint == 0.
 * Multi-GPU: a jq_create_multi handle shards the nodes over its devices and all-reduces inside this call; with one
 * process per GPU each rank passes its shard of (nodes, weights) (jq_shard_bounds) to jq_eval_f_g_grad_dev and the
 * caller sums the packed vector over the ranks with ONE all-reduce (RCCL), see juqbox.jl_amd/ipopt_interface.py.
 */
int jq_eval_f_g_grad(jq_handle *h, const double *pcof, int32_t ncoeff, const double *nodes, const double *weights,
                     int32_t nquad, const double *shift, int32_t compute_adjoint, double *out2, double *infid_grad,
                     double *leak_grad);

/*
 * The same evaluation with the result left ON THE DEVICE for a caller that runs its own collective (one process per GPU:
 * torch.distributed / RCCL, MPI.jl): d_packed is a DEVICE pointer on the handle's GPU to 2 + 2*ncoeff doubles
 *   { sum_i w_i*infidelity_i, sum_i w_i*leak_i, infid_grad[ncoeff], leak_grad[ncoeff] }
 * -- exactly the vector that ONE all-reduce (sum) over the ranks turns into the results of eval_f_g_grad!.  The call
 * returns after the handle's stream has been synchronised.  nquad == 0 (a rank without a shard) writes zeros.
 * Single-device handles only.
 */
int jq_eval_f_g_grad_dev(jq_handle *h, const double *pcof, int32_t ncoeff, const double *nodes, const double *weights,
                         int32_t nquad, const double *shift, int32_t compute_adjoint, void *d_packed);

/*
 * Robustness sweep (ep_plot, examples/Risk_Neutral/run_all.jl:6-32): nquad independent
 * traceobjgrad evaluations with Hconst + ep_i*diag(shift), forward sweep only.
 * out: [4 x nquad] column-major = { objfv, primaryobjf, secondaryobjf, traceInfidelity } per node.
 */
int jq_traceobj_sweep(jq_handle *h, const double *pcof, int32_t ncoeff, const double *nodes, int32_t nquad,
                      const double *shift, double *out);

/* ---- introspection ---------------------------------------------------------------------------*/
/*
 * The plan of a handle as a JSON object (NUL-terminated, at most buflen - 1 characters; returns the length the full text needs,
 * negative JQ_E* on error): Hilbert dimension and tile rows, the structure the operators were found to have ("t4" = 4 x 4 x n
 * Kronecker structure, "od" = diagonal off-diagonal blocks, "band" / "dense"), the embedded twin if any, the control groups, the
 * integrator / solver settings and -- the part a caller sizing an ensemble needs -- the batch-size thresholds of the kernel
 * families: "families" lists, in the order run_eval tries them, {family, name, max_columns / max_slabs / max_quads} for the
 * current settings.  jq_last_timing() reports which one actually ran.
 */
int jq_plan_info(const jq_handle *h, char *buf, int32_t buflen);

/* ---- measurement -----------------------------------------------------------------------------*/
int jq_last_timing(const jq_handle *h, jq_timing *t);
/* Library build info: "gfx950 juqbox_hip <version> src:<12 hex digits> code:<12 hex digits>" -- src: the SHA-256 prefix of the library's
 * sources (the .hip and .h files under csrc/ and include/juqbox_hip.h) AND of the flags they were built with, so measurements recorded
 * for one build (profiles/) cannot be paired with another build by accident (bench.py compares it); code: of the sources alone (the
 * shipped library and its default-register-form twin of `make check-forms-lib` agree in it). */
const char *jq_version(void);
/* JQ_ABI_VERSION of the header the library was built with */
int jq_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* JUQBOX_HIP_H */
