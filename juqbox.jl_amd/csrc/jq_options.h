// jq_options.h -- the per-handle options of libjuqbox_hip.so (host side only).
//
// Up to ABI 4 the library read 37 JQ_* environment variables, some at jq_create and some at every evaluation: process-wide state
// that changed kernel selection behind the caller's back.  Since ABI 5 every knob is an OPTION OF A HANDLE: parsed once into this
// struct -- from the string given to jq_create_opts / jq_create_multi_opts, in front of it the ONE environment variable JQ_OPTIONS
// (same syntax; for callers that cannot reach the handle, e.g. an unmodified Julia script) -- and changed later with
// jq_set_option.  Options that shape the plan (structure, kernel families, chunking) re-plan the handle when they change.
// The defaults are the measured optimum; the options exist for tests (kernel variants that must agree bit for bit), bisection and
// experiments.  Syntax: "name=value,name=value" (separators ',', ';' or blanks; values are integers; unknown names are an error).
#pragma once
#include <climits>
#include <cstdlib>
#include <cstring>
#include <string>

enum JqOpt {
    // ---- plan time ------------------------------------------------------------------------------------------------------
    O_T4, O_OD, O_T4BIG, O_FORCE_DENSE, O_WINDOW, O_LANE, O_LANE_MIN, O_LANE_MAX, O_ROWLANE_MAX, O_COOP_MAX, O_QUAD, O_CQ, O_EMBED,
    O_STREAM_BYTES, O_CHUNK_STEPS, O_BATCH,
    // ---- per evaluation -------------------------------------------------------------------------------------------------
    O_DQ, O_NOSPLIT, O_QUAD8, O_CQ_W, O_IMR_CQ, O_IMR_CQ2, O_CQ_FWD2, O_CQ3, O_QSPLIT, O_RL_SPLIT, O_JAC_WG, O_TRACE_BYTES, O_NO_UNI, O_NO_ORD,
    O_QS_RIDE, O_CQ_GENERIC_TRACES, O_WLR_SC, O_RCCL_SELFCHECK, O_MULTI_SAME_DEVICE, O_CQ3_RDV_US, O_CQ3_WAIT_MS,
    // ---- test hooks (results unchanged) ---------------------------------------------------------------------------------
    O_DEBUG, O_CQ3_FAULT,
    O_COUNT
};

#define JQ_OPT_UNSET LLONG_MIN
#define JQ_OPT_PLAN 1u      // shapes the plan: jq_set_option re-plans the handle
#define JQ_OPT_HOOK 2u      // test hook
#define JQ_OPT_EXP 4u       // experiment only: refused unless the library was built with -DJQ_EXPERIMENTS

struct JqOptDesc {
    const char* name;
    long long dflt;         // JQ_OPT_UNSET: "not set" (the library's own choice)
    unsigned flags;
    const char* doc;
};

// (INTEGRATION.md section 4 is generated from this table: tests/test_abi.py compares the two)
static const JqOptDesc g_jq_opt[O_COUNT] = {
    {"t4", 1, JQ_OPT_PLAN, "0: no 4 x 4 x n variant (4x4 diagonal blocks on v_mfma_f64_4x4x4); falls back to diagonal off-diagonal blocks"},
    {"od", 1, JQ_OPT_PLAN, "0: no diagonal-off-diagonal-block and no 4 x 4 x n variant (plain block-band tiles)"},
    {"t4big", 1, JQ_OPT_PLAN, "0: 4 x 4 x 7 / 4 x 4 x 8 structures (Ntot 97 .. 128) on the general Ntot > 96 cooperative kernels"},
    {"force_dense", 0, JQ_OPT_PLAN, "1: no structure exploitation at all (all 16x16 tiles)"},
    {"window", 1, JQ_OPT_PLAN, "0: per-operator LDS ring instead of the window staging"},
    {"lane", 1, JQ_OPT_PLAN, "0: no lane / row-lane kernels (Ntot <= 16 on the MFMA kernels)"},
    {"lane_min", JQ_OPT_UNSET, JQ_OPT_PLAN, "smallest column count routed to the lane kernels"},
    {"lane_max", JQ_OPT_UNSET, JQ_OPT_PLAN, "largest column count routed to the lane kernels"},
    {"rowlane_max", JQ_OPT_UNSET, JQ_OPT_PLAN, "largest column count routed to the row-lane kernels (0: never)"},
    {"coop_max", JQ_OPT_UNSET, JQ_OPT_PLAN, "cooperative kernels for batches of at most n slabs (default #CU; 0: never)"},
    {"quad", JQ_OPT_UNSET, JQ_OPT_PLAN, "quad-layout kernels for batches of at most n slabs (0: never; default: whenever the round count favours them)"},
    {"cq", JQ_OPT_UNSET, JQ_OPT_PLAN, "cooperative-quad (latency) kernels for batches of at most n column quads (0: never; default 2 #CU)"},
    {"embed", 1, JQ_OPT_PLAN, "structure embedding (d1 x d2 x d3 zero-padded to 4 x 4 x n): 0 off, 1 for batches of the MFMA families, 2 for every batch"},
    {"stream_bytes", JQ_OPT_UNSET, JQ_OPT_PLAN, "bytes of the operator tile stream of one chunk (default 1 GiB)"},
    {"chunk_steps", JQ_OPT_UNSET, JQ_OPT_PLAN, "time steps per chunk (default: what the tile stream holds)"},
    {"batch", JQ_OPT_UNSET, JQ_OPT_PLAN | JQ_OPT_EXP, "batched staging of B time steps per DMA burst (measured: no gain)"},
    {"dq", 1, JQ_OPT_PLAN, "0: no dense cooperative-quad kernels (17 .. 32 levels without the 4 x 4 x n structure: small batches on the cooperative kernels)"},
    {"nosplit", 0, 0, "1: evaluate an ensemble as ONE batch even when full rounds + a remainder on other kernels would be faster"},
    {"quad8", JQ_OPT_UNSET, 0, "force 1 / 2 / 3 slabs per quad-layout workgroup (value 0 / 1 / 2)"},
    {"cq_w", 1, 0, "0: full leakage weights that fit four slots on the quad-layout kernels instead of the cooperative-quad ones"},
    {"imr_cq", 1, 0, "0: implicit midpoint, N = 4, on the quad-layout kernels instead of the cooperative-quad ones"},
    {"imr_cq2", 1, 0, "0: implicit midpoint, cooperative quad: the one-set backward kernel (bit-identical)"},
    {"cq_fwd2", JQ_OPT_UNSET, 0, "cooperative-quad forward sweep with one (0) / two (1) column quads per workgroup (default: two beyond #CU quads; bit-identical)"},
    {"cq3", JQ_OPT_UNSET, 0, "cooperative-quad backward sweep: 0 the one-workgroup kernel, 3 three workgroups per quad or none (default: three / two / one by batch size; bit-identical)"},
    {"qsplit", JQ_OPT_UNSET, 0, "quad-layout backward sweep on two waves per quad: 0 off (one-wave kernel, bit-identical), 2 two quads per workgroup for every batch of the cooperative-quad plan"},
    {"rl_split", 1, 0, "row-lane backward sweep: 0 = one wave, 1 = the library's rule (Stormer-Verlet: three or four waves -- state | adjoint | traces; implicit midpoint: two while idle SIMDs remain), 2 / 3 = two / three at every batch size (all bit-identical)"},
    {"jac_wg", 1, 0, "0: Jacobi solver, N > 16: convergence per 16-column part instead of per sample"},
    {"trace_bytes", JQ_OPT_UNSET, 0, "bytes of the per-step trace records of one backward chunk (default 4 GiB): smaller = more, shorter chunks"},
    {"no_uni", 0, 0, "1: three-slab quad-layout backward sweep on the generic kernel (tests compare)"},
    {"no_ord", 0, 0, "1: no single-subsystem-control specialisation of the quad-layout backward kernels (tests compare)"},
    {"qs_ride", JQ_OPT_UNSET, 0, "split quad-layout backward kernel: trace products as separate passes (0) / riding along also at four quads per workgroup (1)"},
    {"cq_generic_traces", 0, 0, "1: cooperative-quad backward sweep with full trace products even when every control acts on one subsystem"},
    {"wlr_sc", 0, JQ_OPT_EXP, "1: full weights, quad layout: carry the column dots of a step in LDS (measured slower)"},
    {"rccl_selfcheck", 1, 0, "multi-device handles: 0 never, 1 on the first all-reduce, 2 on every all-reduce compare RCCL's result with the host-order sum (3: test mode)"},
    {"multi_same_device", 0, 0, "TEST MODE of jq_create_multi: 1 = up to 16 sub-handles may share physical GPUs, host-side sum in place of the all-reduce"},
    {"cq3_rdv_us", JQ_OPT_UNSET, 0, "two- / three-workgroup latency kernels: microseconds the workgroups of a launch wait for each other at its start before the launch is abandoned and the evaluation repeated on one workgroup per quad (default: one launch duration, 2 .. 100 ms)"},
    {"cq3_wait_ms", JQ_OPT_UNSET, 0, "two- / three-workgroup latency kernels: milliseconds a wait between roles may take after a passed rendezvous before the launch is declared dead (default: 10 x the launch's expected duration, at least 50 ms)"},
    {"debug", 0, JQ_OPT_HOOK, "bit 16 / 32: the consumer roles / the state role of the split latency kernels start ~ 5 ms late (results unchanged); bits 1, 2, 4, 8: profiling experiments with WRONG results (experiment builds only)"},
    {"cq3_fault", 0, JQ_OPT_HOOK, "1: the split latency kernels report a dead wait, 3: a failed start-up rendezvous (exercises fall-back and cool-down)"},
};

struct JqOptions {
    long long v[O_COUNT];
    JqOptions()
    {
        for (int i = 0; i < O_COUNT; ++i) v[i] = JQ_OPT_UNSET;
    }
    bool has(int o) const { return v[o] != JQ_OPT_UNSET; }
    long long get(int o) const { return v[o] != JQ_OPT_UNSET ? v[o] : g_jq_opt[o].dflt; }      // (JQ_OPT_UNSET for options without a default)
    bool on(int o) const { return get(o) != 0; }
    static int find(const char* name, size_t len)
    {
        for (int i = 0; i < O_COUNT; ++i)
            if (strlen(g_jq_opt[i].name) == len && strncmp(g_jq_opt[i].name, name, len) == 0) return i;
        return -1;
    }
    // sets option `o`; false (with a message) when the value is not allowed in this build
    bool set(int o, long long value, std::string* err)
    {
#ifndef JQ_EXPERIMENTS
        if ((g_jq_opt[o].flags & JQ_OPT_EXP) && value != JQ_OPT_UNSET && value != g_jq_opt[o].dflt) {
            *err = std::string("option '") + g_jq_opt[o].name + "' exists in experiment builds only (-DJQ_EXPERIMENTS)";
            return false;
        }
        if (o == O_DEBUG && value != JQ_OPT_UNSET && (value & ~48LL)) {
            *err = "option 'debug': only the test hooks 16 / 32 exist in release builds (the other bits change results)";
            return false;
        }
#endif
        v[o] = value;
        return true;
    }
    // "name=value,name=value" (also ';' and blanks as separators); nullptr / "" is fine
    bool parse(const char* s, std::string* err)
    {
        if (!s) return true;
        while (*s) {
            while (*s == ',' || *s == ';' || *s == ' ' || *s == '\t' || *s == '\n') ++s;
            if (!*s) break;
            const char* e = s;
            while (*e && *e != '=' && *e != ',' && *e != ';' && *e != ' ') ++e;
            if (*e != '=') {
                *err = std::string("options: '") + std::string(s, e - s) + "' is not of the form name=value";
                return false;
            }
            const int o = find(s, (size_t)(e - s));
            if (o < 0) {
                *err = std::string("options: unknown option '") + std::string(s, e - s) + "'";
                return false;
            }
            char* end = nullptr;
            const long long val = strtoll(e + 1, &end, 0);
            if (end == e + 1) {
                *err = std::string("options: '") + g_jq_opt[o].name + "' needs an integer value";
                return false;
            }
            if (!set(o, val, err)) return false;
            s = end;
        }
        return true;
    }
    // the options that differ from "not set", as "name=value,..." (jq_plan_info)
    std::string str() const
    {
        std::string r;
        for (int i = 0; i < O_COUNT; ++i)
            if (has(i)) r += (r.empty() ? "" : ",") + std::string(g_jq_opt[i].name) + "=" + std::to_string(v[i]);
        return r;
    }
};
