// jq_host_multi.h -- part of the host side of libjuqbox_hip.so (included by juqbox_hip.hip, ONE translation unit; not a stand-alone header):
// multi-device handles: one process, N GPUs, one RCCL all-reduce.
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

