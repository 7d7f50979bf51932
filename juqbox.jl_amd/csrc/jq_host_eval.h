// jq_host_eval.h -- part of the host side of libjuqbox_hip.so (included by juqbox_hip.hip, ONE translation unit; not a stand-alone header):
// run_eval: how a batch is routed to a kernel family and propagated chunk by chunk.
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
#define JQ_RL_ROOM 12          // waves per compute unit the two-wave implicit-midpoint row-lane kernel may ask for (NPJ <= 8)
#define JQ_RL_ROOM_WIDE 4      // ... NPJ = 12, 16
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
    // (round 6: two 16-row blocks WITHOUT the structure, N = 4: the dense policy of the cooperative-quad kernels -- routed like the
    //  JQ_BW_T4 plans' latency path: slab state file, implicit-midpoint terminal kernel, one workgroup per evaluation)
    const bool imr_dq = imr && !imr_rl && h->quad_max_slabs == 0 && h->dq_max_quads > 0 && h->N == 4 && h->parts == 1 && h->wrank == 0 &&
                        (ncols_used + 3) / 4 <= h->dq_max_quads && h->opt.on(O_IMR_CQ);
    const bool imr_quad = imr && !imr_rl && ((h->quad_max_slabs > 0 && (h->N == 1 || h->N == 2 || h->N == 4)) || imr_dq);
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
    // ... and their DENSE policy (round 6): 17 .. 32 levels without the structure, Neumann solver, Diagonal weights (no two-quad forward
    // variant)
    const bool cq_dn = !cq && !imr && !lane && !rl && !wfull && h->solver_id == 1 && h->dq_max_quads > 0 && nquads_used <= h->dq_max_quads;
    if (cq_dn) cq = true;
    const int qps = h->parts > 1 ? 4 : (h->sps * h->N + 3) / 4;      // column quads of a full slab
    // ... and of the implicit-midpoint integrator (jq_cq_imr_kernels.h): N = 4, one workgroup of NT waves per evaluation
    const bool imr_cq = imr_dq || (imr_quad && h->N == 4 && h->parts == 1 && h->cq_max_quads > 0 && nquads_used <= h->cq_max_quads &&
                                   h->opt.on(O_IMR_CQ));
    // more column quads than CUs: the forward sweep takes two quads per workgroup (one round of workgroups at ~ 1.5 x the time
    // instead of two rounds; option cq_fwd2=0: one quad per workgroup, =1: always two)
    const bool cq_fwd2 = cq && !cq_dn && !wfull && (h->opt.has(O_CQ_FWD2) ? h->opt.on(O_CQ_FWD2) : nquads_used > h->num_cu);
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
    const bool imr_cq2 = imr_cq && !imr_dq && !imr_cq3 && h->NT <= 6 && h->opt.on(O_IMR_CQ2) &&
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
    if (qs_on && cq && !cq_dn && !cq3 && !wfull && ((nquads_used > h->num_cu && 2 * nslabs <= h->num_cu) || qs_force2) && qsplit_lds(h, 2) <= 163840) qs_qw = 2;
    const bool qsplit = qs_qw > 0;
    const int qs_blocks = qsplit ? (4 * nslabs + qs_qw - 1) / qs_qw : 0;
    if (qsplit) {
        const int rc0 = dev_grow(h, &h->d_qsplit, &h->cap_qsplit, (size_t)qs_blocks * qs_qw * 2 * JQ_QS_ARRAYS * h->NT * 64);
        if (rc0) return rc0;
    }
    // (full leakage weights: the cooperative kernels sum their column dot products over the waves through an LDS record of
    //  2 x JQ_COOP_WDOTS x NT x 16 doubles behind the Jacobi norms; where that does not fit next to the operator slots the slab kernels serve)
    const size_t coop_w_bytes = wfull ? (size_t)2 * JQ_COOP_WDOTS * h->NT * 16 * 8 : 0;
    const bool coop_w_fits = !wfull || coop_hbm(h->NT, h->BWc) ||
                             (size_t)2 * h->mat_elems_c * 8 + (size_t)32 * h->NT * 8 + (size_t)2 * h->KT * 64 * 8 + (size_t)16 * h->NT * 8 + coop_w_bytes <= 163840;
    const bool coop = imr_coop || (!cq && !quad && !lane && !rl && h->NT >= 2 && h->coop_ok && coop_w_fits && (h->solver_id == 1 || h->big || wjac) &&
                                   (nslabs <= h->coop_max_slabs || (wfull && !(h->NT == 6 && h->BW == 5))));      // (dense 96 x 96: the slab kernels <6, 5> carry the low-rank terms too -- large batches stay there)
    if (wjac && !coop && h->BW == JQ_BW_T4)      // (jq_update_wmat / jq_set_linear_solver re-plan such handles without the structure: cannot happen)
        return fail(h, JQ_EHIP, "internal error: full leakage weights with the Jacobi solver on a 4 x 4 x n plan without cooperative kernels");      // (Ntot > 96: also the Jacobi solver; full weights: every batch size -- the slab kernels have no low-rank terms)
    // row-lane kernels, Stormer-Verlet: the backward sweep's two chains on two waves (jq_rowlane_kernels.h k_backward_rowlane2);
    // option rl_split=0: one wave (tests: the two variants must agree bit for bit)
    // (both integrators; while the doubled wave count still finds idle issue slots: measured in round 3, HISTORY.md --
    //  NPJ <= 8: up to three waves per SIMD, NPJ = 12, 16 (constant images in LDS, 24 .. 32 operand registers per image row): one)
    // ... Stormer-Verlet since round 6: three or four waves (state | adjoint | traces, two trace waves for two or more controls:
    // k_backward_rowlane3) at every batch size the row-lane kernels serve -- measured over the ensemble size (scripts/time_rl_split.py,
    // profiles/r06_rowlane3.txt): they beat one wave from 1 to 2 048 samples at NPJ = 4, 6 and 12; two waves never beat three.
    // Option rl_split: 0 = one wave, 1 = by these rules, 2 / 3 = two / three waves at every batch size (tests, measurements)
    const int rl_want = (int)h->opt.get(O_RL_SPLIT);
    const bool rl_sv = rl && !imr_rl;
    bool rl_split = rl && (rl_want >= 2 || rl_sv || 2 * nwaves_rl <= (long long)(h->rl_npj > 8 ? JQ_RL_ROOM_WIDE : JQ_RL_ROOM) * h->num_cu);
    if (rl_want == 0) rl_split = false;
    if (wfull) rl_split = false;      // (the one-wave backward kernel carries the low-rank terms)
    const bool rl_split3 = rl_split && rl_sv && rl_want != 2;
    prop_kernel_t kfwd, kbwd;
    lane_init_t klinit = nullptr;
    lane_term_t klterm = nullptr;
    int rc = imr_cq ? select_cq_imr_kernels(h, imr_cq2, imr_cq3, imr_dq, &kfwd, &kbwd)
             : imr_quad ? select_quad_imr_kernels(h, &kfwd, &kbwd)
             : imr_coop ? (imr_parts ? select_coop_imr_parts_kernels(h, imr_hbm, &kfwd, &kbwd) : select_coop_imr_kernels(h, imr_hbm, &kfwd, &kbwd))
             : imr_rl ? select_rowlane_imr_kernels(h, rl_split, &kfwd, &kbwd)
             : rl ? select_rowlane_kernels(h, rl_split3 ? 3 : rl_split ? 2 : 1, hist_r != nullptr, &kfwd, &kbwd)
             : lane ? select_lane_kernels(h, &kfwd, &kbwd, &klinit, &klterm)
                  : cq ? select_cq_kernels(h, cq_fwd2, cq_nr, wfull, cq_dn, &kfwd, &kbwd)
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
    const long long stride = (cq_dn || imr_dq) ? (long long)JQ_DQ_ELEMS : rl ? h->rl_stride : lane ? h->lane_stride : coop ? h->mat_elems_c : h->mat_elems;
    const double* himg = (cq_dn || imr_dq) ? h->d_himg_dq : rl ? h->d_himg_r : lane ? h->d_himg_l : coop ? h->d_himg_c : h->d_himg;
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
    const double* cimg_base = (cq_dn || imr_dq) ? h->d_cimg_dq : rl ? h->d_cimg_r : lane ? h->d_cimg_l : coop ? h->d_cimg_c : h->d_cimg;      // (control-group order)
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
    const int batch = coop ? 0 : (quad || cq || imr_dq) ? -1 : h->batch;
    const size_t lds_stage = (coop && (h->NT > 6 || imr_hbm || (!imr_coop && coop_hbm(h->NT, h->BWc)))) ? 0      // operators are read from HBM, no LDS staging
                             : batch > 0   ? (size_t)2 * (2 * batch + 1) * 2 * stride * 8 + (size_t)2 * h->NcK * stride * 8
                             : batch < 0 ? (size_t)(2 * JQ_WIN_TPS + 2 * h->NcK) * stride * 8
                                         : (size_t)2 * stride * 8;
    const size_t lds_cq = lds_stage + (size_t)32 * h->NT * 8 + (size_t)6 * (h->NT + 2) * 64 * 8 + (size_t)std::max(2, h->NcK + (h->NcK + 1) / 2) * h->NT * 64 * 8;      // tables, x exchange, trace hand-off / wg-sum scratch (one region)
    const size_t lds_fwd = huge ? 0 : (rl && !imr_rl) ? (wfull ? (size_t)JQ_RL_WTAB * 8 : 0) + JQ_RL_RING_BYTES(h->rl_npj) /* operator ring of the forward sweep */ : rl ? 0 : lane ? 0 : (cq || imr_cq) ? lds_cq : imr_coop ? coop_imr_lds_bytes(h->NT, imr_hbm ? 0 : stride)
                                           : lds_stage + (size_t)32 * h->NT * 8 + (coop ? (size_t)2 * h->KT * 64 * 8 + (size_t)16 * h->NT * 8 + coop_w_bytes : 0);      // (+ the Jacobi solver's column norms [NT][16], the low-rank weights' dot exchange)
    const size_t lds_bwd = huge ? 0 : qsplit ? qsplit_lds(h, qs_qw) : rl_split3 ? JQ_RL3_LDS(h->rl_npj) : rl ? (h->rl_npj > 8 ? (size_t)2 * h->NcK * h->rl_stride * 8 : 0) + (rl_split ? (size_t)2 * 3 * 64 * 8 : 0) /* records: 3 values per lane and slot, implicit midpoint 2 */ + (wfull ? (size_t)JQ_RL_WTAB * 8 : 0) /* low-rank weight table */ : lane ? 0 : imr_cq2 ? cq_imr2_lds(h, lds_stage) : (coop || cq || imr_cq) ? lds_fwd
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
                hipLaunchKernelGGL(kbwd, dim3(nblocks), dim3(rl_split3 ? (a.Ncoupled >= 2 ? 4 : 3) * nthreads /* state | adjoint | traces (two waves for two or more controls) */ : (cq || rl_split) ? 2 * nthreads : imr_cq2 ? 2 * (nthreads + 128) : imr_cq ? nthreads + 128 : nthreads), lds_bwd + (wlr_lds_bwd >= 0 ? wlr_bytes : 0) + (wsc_lds_bwd >= 0 ? wsc_bytes : 0) + jac_bytes, s, a);      // (cooperative quad: state and adjoint chain on separate waves)
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
    h->timing.kernel_band = (rl || lane) ? 0 : coop ? h->BWc : (cq_dn || imr_dq) ? 10 /* dense blocks on the cooperative-quad kernels */ : (quad || cq) ? JQ_BW_T4Q : h->BW;
    h->timing.kernel_variant = cq3 ? cq_nr : qsplit ? 20 + qs_qw : (rl && rl_split3) ? 33 : (rl && rl_split) ? 32 : 0;      // (workgroups per column quad in the backward sweep of the cooperative-quad kernels)
    h->timing.ms_allreduce = 0.0;
    h->timing.ms_shard_min = h->timing.ms_shard_max = h->timing.ms_total;
    return JQ_OK;
}

