// jq_host_info.h -- part of the host side of libjuqbox_hip.so (included by juqbox_hip.hip, ONE translation unit; not a stand-alone header):
// options of a live handle, plan and timing introspection.
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
    if (h->rl_npj > 0) add(3, "row-lane (VALU, lane per (row, column); backward sweep on three or four waves, implicit midpoint: two)", "columns", h->rl_max_cols);
    if (h->lane_np > 0) add(2, "lane (VALU, lane per column)", "columns", h->lane_max_cols);
    if (t->cq_max_quads > 0) add(8, "cooperative quad (one 16-row block per wave)", "quads", t->cq_max_quads);
    if (t->dq_max_quads > 0) add(8, "cooperative quad, dense blocks (17 .. 32 levels without the 4 x 4 x n structure; Neumann, Diagonal weights)", "quads", t->dq_max_quads);
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

