// jq_host_create.h -- part of the host side of libjuqbox_hip.so (included by juqbox_hip.hip, ONE translation unit; not a stand-alone header):
// device buffers, uploads, jq_create* (planning from the operators' nonzero structure), structure embedding.
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
    if (h->dq_max_quads > 0) {
        std::vector<double> id((size_t)(1 + 2 * h->Nc) * JQ_DQ_ELEMS, 0.0);
        dq_image(h->Hconst.data(), h->Ntot, id.data());
        for (int q = 0; q < h->Nc; ++q) {
            dq_image(h->Hsym.data() + q * nn, h->Ntot, id.data() + (size_t)(1 + q) * JQ_DQ_ELEMS);
            dq_image(h->Hanti.data() + q * nn, h->Ntot, id.data() + (size_t)(1 + h->Nc + q) * JQ_DQ_ELEMS);
        }
        HIPCHK(h, hipMemcpy(h->d_himg_dq, id.data(), id.size() * sizeof(double), hipMemcpyHostToDevice));
        std::vector<double> cd((size_t)2 * h->Nc * JQ_DQ_ELEMS);
        for (int q = 0; q < h->Nc; ++q) {
            std::copy_n(id.data() + (size_t)(1 + q) * JQ_DQ_ELEMS, JQ_DQ_ELEMS, cd.data() + cslot(q, false) * JQ_DQ_ELEMS);
            std::copy_n(id.data() + (size_t)(1 + h->Nc + q) * JQ_DQ_ELEMS, JQ_DQ_ELEMS, cd.data() + cslot(q, true) * JQ_DQ_ELEMS);
        }
        HIPCHK(h, hipMemcpy(h->d_cimg_dq, cd.data(), cd.size() * sizeof(double), hipMemcpyHostToDevice));
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
    double** bufs[] = {&h->d_himg_dq, &h->d_cimg_dq, &h->d_cq3, &h->d_qsplit, &h->d_wlr, &h->d_cimg_l, &h->d_cimg_r, &h->d_rfreq, &h->d_wq, &h->d_pk2, &h->d_pack, &h->d_himg_r, &h->d_uinit_r, &h->d_vtr_r, &h->d_vti_r, &h->d_himg_l, &h->d_uinit_l, &h->d_vtr_l, &h->d_vti_l, &h->d_himg_c, &h->d_cimg_c, &h->d_park, &h->d_cimg, &h->d_himg,  &h->d_uimg,       &h->d_vtr,     &h->d_vti,    &h->d_tabs, &h->d_tf,   &h->d_tb,
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
            const long long lds_c = (coop_hbm(h->NT, h->BWc) ? 0 : 2 * ec * 8) + lds_fwd_fixed + 2LL * h->KT * 64 * 8 + 16LL * h->NT * 8;      // (operator slots, tables, x exchange, Jacobi column norms)
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
        // ... with the DENSE policy (round 6, jq_cq_kernels.h CoopQ<2, true>): two 16-row blocks WITHOUT the structure (17 .. 32 levels: two
        // five-level subsystems, a drift in its eigenbasis, ...), whose small batches otherwise take the cooperative kernels.  Images of
        // JQ_DQ_ELEMS doubles in the window staging; Neumann solver, Diagonal weights.  option dq=0 disables them.
        {
            const long long win = (2LL * JQ_WIN_TPS + 2LL * h->NcK) * JQ_DQ_ELEMS * 8;
            const long long tail = 32LL * h->NT * 8 + 6LL * (h->NT + 2) * 64 * 8 + (long long)std::max(2, h->NcK + (h->NcK + 1) / 2) * h->NT * 64 * 8;      // (run_eval: lds_cq)
            h->dq_max_quads = (h->NT == 2 && h->BW != JQ_BW_T4 && !h->big && !h->huge && !h->is_emb && win + tail <= 163840 && h->opt.on(O_DQ)) ? 3 * prop.multiProcessorCount : 0;      // (three rounds of them, 3 x 16 ms per 2 000 steps, still beat one round of the cooperative kernels, 52 ms: profiles/r06_midsize_single.txt (g))
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
    if (h->dq_max_quads > 0) {
        if ((rc = dev_alloc(h, &h->d_himg_dq, (size_t)(1 + 2 * h->Nc) * JQ_DQ_ELEMS))) return rc;
        if ((rc = dev_alloc(h, &h->d_cimg_dq, (size_t)(2 * h->Nc) * JQ_DQ_ELEMS))) return rc;
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
    const long long img_elems = std::max(std::max(std::max(h->mat_elems, h->mat_elems_c), std::max(h->lane_stride, h->rl_stride)), h->dq_max_quads > 0 ? (long long)JQ_DQ_ELEMS : 0LL);
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

