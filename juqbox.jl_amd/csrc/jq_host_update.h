// jq_host_update.h -- part of the host side of libjuqbox_hip.so (included by juqbox_hip.hip, ONE translation unit; not a stand-alone header):
// the mutations scripts apply to params after construction: solver / integrator, target, drift (re-planning), leakage weights.
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

