# hip_backend.jl -- Julia binding of libjuqbox_hip.so (C ABI: include/juqbox_hip.h), the MI355X drop-in for
# Juqbox.jl's Stormer-Verlet traceobjgrad path.
#
# Usage (inside module Juqbox, after evalobjgrad.jl / ipopt_interface.jl):  include("hip_backend.jl")
#   wa = Working_Arrays_HIP(params, nCoeff)                      # one GPU (the current HIP device)
#   wa = Working_Arrays_HIP(params, nCoeff; devices = 8)         # ONE Julia process, 8 GPUs: the risk-neutral ensemble of
#                                                                # eval_f_g_grad! is sharded over them, one RCCL all-reduce
# Everything above stays as it is: the reference dispatches traceobjgrad on the type of `wa`
# (src/evalobjgrad.jl:504 vs :1042; every caller passes `wa` untyped, src/ipopt_interface.jl:24,47,55,77,104,124,153,267),
# so setup_ipopt_problem / run_optimizer / plot_results run unchanged on top of the methods below.
#
# tests/test_julia_shim.py checks every struct layout and ccall signature of this file against include/juqbox_hip.h
# (Julia itself is not available in the build image).
using LinearAlgebra
using SparseArrays

const libjq = get(ENV, "JUQBOX_HIP_LIB", "libjuqbox_hip")      # on LD_LIBRARY_PATH, or an absolute path

struct JQCsc                              # == jq_csc: the fields of a SparseMatrixCSC{Float64,Int64}, 1-based, passed as they are
    m::Int64
    n::Int64
    colptr::Ptr{Int64}
    rowval::Ptr{Int64}
    nzval::Ptr{Float64}
end
JQCsc(A::SparseMatrixCSC{Float64,Int64}) = JQCsc(size(A, 1), size(A, 2), pointer(A.colptr), pointer(A.rowval), pointer(A.nzval))

struct JQProblem                          # == jq_problem
    Ntot::Int32
    N::Int32
    Ncoupled::Int32
    Nfreq::Int32
    nsteps::Int32
    neumann_terms::Int32
    objFuncType::Int32
    Nunc::Int32
    T::Float64
    Hconst::Ptr{Float64}
    Hsym_ops::Ptr{Float64}
    Hanti_ops::Ptr{Float64}
    Uinit::Ptr{Float64}
    Utarget_r::Ptr{Float64}
    Utarget_i::Ptr{Float64}
    wmat_real_diag::Ptr{Float64}
    Cfreq::Ptr{Float64}
    Hunc_ops::Ptr{Float64}
    Rfreq::Ptr{Float64}
    Hconst_csc::Ptr{JQCsc}                # use_sparse = true: read when the dense pointer above is NULL
    Hsym_csc::Ptr{JQCsc}
    Hanti_csc::Ptr{JQCsc}
end

struct JQTiming                           # == jq_timing
    ms_total::Float64
    ms_propagate::Float64
    ms_generate::Float64
    ms_forward::Float64
    ms_backward::Float64
    n_forward_launches::Int64
    n_backward_launches::Int64
    mfma_executed::Int64
    svts::Int64
    kernel_family::Int32
    kernel_size::Int32
    kernel_band::Int32
    kernel_variant::Int32
    mfma_backward::Int64
    ms_allreduce::Float64
    ms_shard_min::Float64
    ms_shard_max::Float64
end

# the struct layouts above are those of JQ_ABI_VERSION 5 of include/juqbox_hip.h: refuse a library built for another one
const JQ_ABI_VERSION = 5
function jq_check_abi()
    v = ccall((:jq_abi_version, libjq), Cint, ())
    v == JQ_ABI_VERSION || error("libjuqbox_hip has ABI version $v, hip_backend.jl was written for $JQ_ABI_VERSION")
    return nothing
end

abstract type AbstractWorkingArraysHIP end

# Working_Arrays_HIP: the Stormer-Verlet method of traceobjgrad (src/evalobjgrad.jl:504);
# Working_Arrays_M_HIP: the implicit-midpoint method (:1042), same handle with jq_set_integrator(2, ...)
mutable struct Working_Arrays_HIP <: AbstractWorkingArraysHIP
    handle::Ptr{Cvoid}
    gr::Vector{Float64}                   # eval_grad_f_par writes Tikhonov's gradient through wa.gr (ipopt_interface.jl:139-141)
end
mutable struct Working_Arrays_M_HIP <: AbstractWorkingArraysHIP
    handle::Ptr{Cvoid}
    gr::Vector{Float64}
end

jq_create_error() = unsafe_string(ccall((:jq_last_error, libjq), Cstring, (Ptr{Cvoid},), C_NULL))
jqcheck(wa, rc) = rc == 0 || error(unsafe_string(ccall((:jq_last_error, libjq), Cstring, (Ptr{Cvoid},), wa.handle)))

# Leakage weights.  The Stormer-Verlet path reads params.wmat_real / params.wmat_imag (src/evalobjgrad.jl:629-630): Diagonal by
# default, FULL matrices with use_custom_forbidden (:214-232) -- those go to jq_update_wmat; the diagonal alone would silently
# change the objective and the gradient.  The implicit-midpoint path reads params.wmat (:1155), a Diagonal by its field type (:90).
full_weights(params) = !(params.wmat_real isa Diagonal) || !iszero(params.wmat_imag)
function push_weights!(wa::Working_Arrays_HIP, params)
    if full_weights(params)
        params.wmat_imag isa Diagonal && !iszero(params.wmat_imag) &&
            error("hip_backend: a non-zero Diagonal wmat_imag is not a Hermitian weight (the reference ignores it in the objective, " *
                  "src/evalobjgrad.jl:2231-2233, but not in the forcing): refusing instead of evaluating something else")
        Wr = Matrix{Float64}(params.wmat_real)
        Wi = Matrix{Float64}(params.wmat_imag)
        jqcheck(wa, ccall((:jq_update_wmat, libjq), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), wa.handle, Wr, Wi))
    else
        wd = Vector{Float64}(diag(params.wmat_real))
        jqcheck(wa, ccall((:jq_update_wmat_diag, libjq), Cint, (Ptr{Cvoid}, Ptr{Float64}), wa.handle, wd))
    end
end
function push_weights!(wa::Working_Arrays_M_HIP, params)
    params.wmat isa Diagonal || error("hip_backend: the implicit-midpoint path needs Diagonal params.wmat")
    wd = Vector{Float64}(diag(params.wmat))
    jqcheck(wa, ccall((:jq_update_wmat_diag, libjq), Cint, (Ptr{Cvoid}, Ptr{Float64}), wa.handle, wd))
end

# options: "name=value,name=value" for jq_create_opts (INTEGRATION.md section 4) -- per handle, parsed once; "" = none (production)
function jq_new_handle(params, devices, options::AbstractString = "")
    jq_check_abi()
    Ntot = params.N + params.Nguard
    # use_sparse = true (src/evalobjgrad.jl:249-262): Hconst, Hsym_ops, Hanti_ops are SparseMatrixCSC{Float64,Int64}; their
    # colptr / rowval / nzval go to the library as they are (jq_csc), nothing is densified on this side
    sparse = params.Hconst isa SparseMatrixCSC{Float64,Int64} && params.Nunc == 0 &&
             all(h -> h isa SparseMatrixCSC{Float64,Int64}, params.Hsym_ops) && all(h -> h isa SparseMatrixCSC{Float64,Int64}, params.Hanti_ops)
    Hc   = sparse ? zeros(1, 1) : Matrix{Float64}(params.Hconst)                          # dense, column-major
    Hs   = (!sparse && params.Ncoupled > 0) ? reduce(hcat, [vec(Matrix{Float64}(h)) for h in params.Hsym_ops]) : zeros(1, 1)
    Ha   = (!sparse && params.Ncoupled > 0) ? reduce(hcat, [vec(Matrix{Float64}(h)) for h in params.Hanti_ops]) : zeros(1, 1)
    c0   = sparse ? [JQCsc(params.Hconst)] : JQCsc[]
    cs   = sparse ? [JQCsc(h) for h in params.Hsym_ops] : JQCsc[]
    ca   = sparse ? [JQCsc(h) for h in params.Hanti_ops] : JQCsc[]
    Hu   = params.Nunc > 0 ? reduce(hcat, [vec(Matrix{Float64}(h)) for h in params.Hunc_ops]) : zeros(1, 1)   # lab-frame evaluation
    Rf   = Vector{Float64}(params.Rfreq)
    wd   = zeros(Ntot)                                             # the weights follow the creation: push_weights!
    Cf   = Matrix{Float64}(params.Cfreq[1:params.Ncoupled+params.Nunc, :])
    Ui   = Matrix{Float64}(params.Uinit)
    h    = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve Hc Hs Ha Hu Rf wd Cf Ui c0 cs ca params begin
        prob = JQProblem(Ntot, params.N, params.Ncoupled, params.Nfreq, params.nsteps,
                         params.linear_solver.max_iter, params.objFuncType, params.Nunc, params.T,
                         sparse ? Ptr{Float64}(C_NULL) : pointer(Hc), sparse ? Ptr{Float64}(C_NULL) : pointer(Hs),
                         sparse ? Ptr{Float64}(C_NULL) : pointer(Ha), pointer(Ui),
                         pointer(params.Utarget_r), pointer(params.Utarget_i), pointer(wd), pointer(Cf),
                         params.Nunc > 0 ? pointer(Hu) : Ptr{Float64}(C_NULL), params.Nunc > 0 ? pointer(Rf) : Ptr{Float64}(C_NULL),
                         sparse ? pointer(c0) : Ptr{JQCsc}(C_NULL), sparse ? pointer(cs) : Ptr{JQCsc}(C_NULL),
                         sparse ? pointer(ca) : Ptr{JQCsc}(C_NULL))
        if devices === nothing
            rc = isempty(options) ? ccall((:jq_create, libjq), Cint, (Ref{JQProblem}, Ref{Ptr{Cvoid}}), prob, h) :
                                    ccall((:jq_create_opts, libjq), Cint, (Ref{JQProblem}, Cstring, Ref{Ptr{Cvoid}}), prob, options, h)
        else
            devs = devices isa Integer ? collect(Int32, 0:devices-1) : collect(Int32, devices)
            rc = isempty(options) ? ccall((:jq_create_multi, libjq), Cint, (Ref{JQProblem}, Ptr{Int32}, Int32, Ref{Ptr{Cvoid}}),
                                          prob, devs, length(devs), h) :
                                    ccall((:jq_create_multi_opts, libjq), Cint, (Ref{JQProblem}, Ptr{Int32}, Int32, Cstring, Ref{Ptr{Cvoid}}),
                                          prob, devs, length(devs), options, h)
        end
    end
    rc == 0 || error(jq_create_error())
    return h[]
end

function Working_Arrays_HIP(params::objparams, nCoeff::Int64; devices = nothing, options::AbstractString = "")
    @assert params.linear_solver.solver_id in (NEUMANN_SOLVER, JACOBI_SOLVER)
    wa = Working_Arrays_HIP(jq_new_handle(params, devices, options), zeros(nCoeff))
    finalizer(w -> ccall((:jq_destroy, libjq), Cvoid, (Ptr{Cvoid},), w.handle), wa)
    return wa
end
function Working_Arrays_M_HIP(params::objparams, nCoeff::Int64; devices = nothing, options::AbstractString = "")
    @assert params.linear_solver.solver_id == JACOBI_SOLVER_M
    wa = Working_Arrays_M_HIP(jq_new_handle(params, devices, options), zeros(nCoeff))
    finalizer(w -> ccall((:jq_destroy, libjq), Cvoid, (Ptr{Cvoid},), w.handle), wa)
    return wa
end

num_devices(wa::AbstractWorkingArraysHIP) = ccall((:jq_num_devices, libjq), Cint, (Ptr{Cvoid},), wa.handle)
# ranks of the RCCL communicator behind a multi-device handle (ncclCommCount; 0: single device)
rccl_world_size(wa::AbstractWorkingArraysHIP) = ccall((:jq_rccl_world_size, libjq), Cint, (Ptr{Cvoid},), wa.handle)
# options of a live handle (INTEGRATION.md section 4); value = nothing: back to "not set"
function set_option!(wa::AbstractWorkingArraysHIP, name::AbstractString, value)
    v = value === nothing ? typemin(Int64) : Int64(value)      # JQ_OPTION_DEFAULT
    jqcheck(wa, ccall((:jq_set_option, libjq), Cint, (Ptr{Cvoid}, Cstring, Int64), wa.handle, name, v))
end
function get_option(wa::AbstractWorkingArraysHIP, name::AbstractString)
    v = Ref{Int64}(0)
    ccall((:jq_get_option, libjq), Cint, (Ptr{Cvoid}, Cstring, Ref{Int64}), wa.handle, name, v) == 0 || error("hip_backend: unknown option $name")
    return v[] == typemin(Int64) ? nothing : v[]
end
num_compute_units(wa::AbstractWorkingArraysHIP) = ccall((:jq_num_compute_units, libjq), Cint, (Ptr{Cvoid},), wa.handle)
function plan_info(wa::AbstractWorkingArraysHIP)      # JSON text: structure, control groups, batch-size thresholds of the kernel families
    n = ccall((:jq_plan_info, libjq), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Int32), wa.handle, C_NULL, 0)
    n >= 0 || error("jq_plan_info failed")
    buf = Vector{UInt8}(undef, n + 1)
    ccall((:jq_plan_info, libjq), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Int32), wa.handle, buf, n + 1)
    return String(buf[1:n])
end
handle_device(wa::AbstractWorkingArraysHIP) = ccall((:jq_handle_device, libjq), Cint, (Ptr{Cvoid},), wa.handle)

# What the reference's solver WOULD use.  lsolver_object's `solve` closure captures max_iter (and tol) when it is constructed
# (src/linear_solvers.jl:36-57): a script that later assigns params.linear_solver.max_iter without calling
# recreate_linear_solver_closure! (:68-78; estimate_Neumann! does call it, src/evalobjgrad.jl:2924-2925) still solves with the OLD
# values on the CPU path.  The captured variables are fields of the closure object (a Core.Box when the constructor reassigned them:
# tol *= sqrt(nrhs)); after recreate_linear_solver_closure! the closure captures the object itself (field :lsolver) and the struct's
# fields count.  Reading them here makes both back ends agree for such scripts too (round 4 read the struct's fields every call).
function solver_in_effect(ls)
    f = ls.solve
    unbox(x) = x isa Core.Box ? x.contents : x
    mi = hasfield(typeof(f), :max_iter) ? unbox(getfield(f, :max_iter)) : ls.max_iter
    tl = hasfield(typeof(f), :tol) ? unbox(getfield(f, :tol)) : ls.tol
    return Int32(mi), Float64(tl)
end

# params is mutated freely by scripts (Hconst inside eval_f_g_grad!, wmat_real, max_iter, targets): push before each call.
# Order: Diagonal weights go in BEFORE the solver / integrator and full weights AFTER it -- full weights exist with the Neumann solver
# only, so a script that switches (full weights, Neumann) <-> (Diagonal, Jacobi) in one step is valid in either direction
# (jq_update_wmat returns at once when the matrices are the ones it has: no eigen-decomposition per call).
function sync!(wa::AbstractWorkingArraysHIP, params::objparams)
    ls = params.linear_solver            # solver_id 1 = NEUMANN_SOLVER, 2 = JACOBI_SOLVER (src/linear_solvers.jl:5-8)
    max_iter, tol = solver_in_effect(ls)
    fullw = wa isa Working_Arrays_HIP && full_weights(params)
    fullw || push_weights!(wa, params)
    if wa isa Working_Arrays_M_HIP
        jqcheck(wa, ccall((:jq_set_integrator, libjq), Cint, (Ptr{Cvoid}, Int32, Int32, Float64),
                          wa.handle, 2, max_iter, tol))
    else
        jqcheck(wa, ccall((:jq_set_linear_solver, libjq), Cint, (Ptr{Cvoid}, Int32, Int32, Float64),
                          wa.handle, ls.solver_id, max_iter, tol))
    end
    fullw && push_weights!(wa, params)
    if params.Hconst isa SparseMatrixCSC{Float64,Int64}
        Hsp = params.Hconst
        GC.@preserve Hsp begin
            jqcheck(wa, ccall((:jq_update_hconst_csc, libjq), Cint, (Ptr{Cvoid}, Ref{JQCsc}), wa.handle, JQCsc(Hsp)))
        end
    else
        Hc = Matrix{Float64}(params.Hconst)
        jqcheck(wa, ccall((:jq_update_hconst, libjq), Cint, (Ptr{Cvoid}, Ptr{Float64}), wa.handle, Hc))
    end
    jqcheck(wa, ccall((:jq_update_target, libjq), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}),
                      wa.handle, params.Utarget_r, params.Utarget_i))
end

# the new method: same signature and return tuples as src/evalobjgrad.jl:504 / :1027-1036
function traceobjgrad(pcof0::Array{Float64,1}, params::objparams, wa::AbstractWorkingArraysHIP,
                      verbose::Bool = false, evaladjoint::Bool = true)
    sync!(wa, params)
    n = length(pcof0)
    out4 = zeros(4)
    if verbose                                              # plot_results path (plot-results.jl:44): ONE forward sweep
        evaladjoint && error("verbose && evaladjoint (forward-gradient self check) is not accelerated")
        Ntot = params.N + params.Nguard
        ur = zeros(Ntot, params.N, params.nsteps + 1)
        ui = similar(ur)
        jqcheck(wa, ccall((:jq_traceobj_verbose, libjq), Cint,
                          (Ptr{Cvoid}, Ptr{Float64}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                          wa.handle, pcof0, n, out4, ur, ui))
        return out4[1], ur + 1im * ui, 1.0 - out4[4]
    end
    tg = zeros(n); ig = zeros(n); lg = zeros(n)
    jqcheck(wa, ccall((:jq_traceobjgrad, libjq), Cint,
                      (Ptr{Cvoid}, Ptr{Float64}, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                      wa.handle, pcof0, n, evaladjoint ? 1 : 0, out4, tg, ig, lg))
    evaladjoint || return out4[1], out4[2], out4[3]
    return out4[1], tg, out4[2], out4[3], out4[4], ig, (params.objFuncType == 1 ? zeros(0) : lg)
end

# the state history alone (usaver, usavei of src/evalobjgrad.jl:677-680, :748-752)
function state_history(pcof::Vector{Float64}, params::objparams, wa::AbstractWorkingArraysHIP)
    sync!(wa, params)
    Ntot = params.N + params.Nguard
    ur = zeros(Ntot, params.N, params.nsteps + 1)
    ui = similar(ur)
    jqcheck(wa, ccall((:jq_state_history, libjq), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int32, Ptr{Float64}, Ptr{Float64}),
                      wa.handle, pcof, length(pcof), ur, ui))
    return ur, ui
end

# Replaces the serial loop of eval_f_g_grad! (src/ipopt_interface.jl:24-70): all quadrature nodes run concurrently on the
# GPU(s); with a multi-device handle they are sharded over the GPUs and summed with one RCCL all-reduce inside the call.
# shift = nothing: the reference's perturbation Hconst[j,j] += 0.01*ep*10^(j-2) (:41-44).
function eval_f_g_grad!(pcof::Vector{Float64}, params::objparams, wa::AbstractWorkingArraysHIP,
                        nodes::AbstractArray = [0.0], weights::AbstractArray = [1.0], compute_adjoint::Bool = true;
                        shift = nothing)
    sync!(wa, params)
    n = length(pcof)
    out2 = zeros(2); ig = zeros(n); lg = zeros(n)
    nd = collect(Float64, nodes); wt = collect(Float64, weights)
    sh = shift === nothing ? C_NULL : pointer(shift)
    GC.@preserve shift begin
        jqcheck(wa, ccall((:jq_eval_f_g_grad, libjq), Cint,
            (Ptr{Cvoid}, Ptr{Float64}, Int32, Ptr{Float64}, Ptr{Float64}, Int32, Ptr{Float64}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
            wa.handle, pcof, n, nd, wt, length(nd), sh, compute_adjoint ? 1 : 0, out2, ig, lg))
    end
    params.last_pcof .= pcof                                   # :23-27, :48-59, :67-68
    params.last_infidelity, params.last_leak = out2
    params.last_infidelity_grad .= compute_adjoint ? ig : 0.0
    length(params.last_leak_grad) > 0 && (params.last_leak_grad .= compute_adjoint ? lg : 0.0)
    params.lastTraceInfidelity = params.last_infidelity
    params.lastLeakIntegral = params.last_leak
end

# One process per GPU (MPI.jl): each rank evaluates its shard and leaves the packed sums on its device; the caller
# all-reduces `d_packed` (2 + 2 nCoeff doubles on the GPU, e.g. a ROCArray) over the ranks.
function shard_bounds(nquad::Integer, rank::Integer, world::Integer)
    lo = Ref{Int32}(0); hi = Ref{Int32}(0)
    rc = ccall((:jq_shard_bounds, libjq), Cint, (Int32, Int32, Int32, Ref{Int32}, Ref{Int32}), nquad, rank, world, lo, hi)
    rc == 0 || error("jq_shard_bounds: invalid arguments")
    return Int(lo[]) + 1, Int(hi[])                             # 1-based inclusive range lo:hi
end
function eval_f_g_grad_dev!(pcof::Vector{Float64}, params::objparams, wa::AbstractWorkingArraysHIP, nodes::Vector{Float64},
                            weights::Vector{Float64}, compute_adjoint::Bool, d_packed::Ptr{Cvoid}; shift = nothing)
    sync!(wa, params)
    sh = shift === nothing ? C_NULL : pointer(shift)
    GC.@preserve shift begin
        jqcheck(wa, ccall((:jq_eval_f_g_grad_dev, libjq), Cint,
            (Ptr{Cvoid}, Ptr{Float64}, Int32, Ptr{Float64}, Ptr{Float64}, Int32, Ptr{Float64}, Int32, Ptr{Cvoid}),
            wa.handle, pcof, length(pcof), nodes, weights, length(nodes), sh, compute_adjoint ? 1 : 0, d_packed))
    end
end

# ep_plot's serial robustness sweep (examples/Risk_Neutral/run_all.jl:6-32) as one call: 4 x nquad matrix
# (objfv, primaryobjf, secondaryobjf, traceInfidelity) per perturbation
function traceobj_sweep(pcof::Vector{Float64}, params::objparams, wa::AbstractWorkingArraysHIP, ep_vals::AbstractArray;
                        shift = nothing)
    sync!(wa, params)
    ep = collect(Float64, ep_vals)
    out = zeros(4, length(ep))
    sh = shift === nothing ? C_NULL : pointer(shift)
    GC.@preserve shift begin
        jqcheck(wa, ccall((:jq_traceobj_sweep, libjq), Cint,
            (Ptr{Cvoid}, Ptr{Float64}, Int32, Ptr{Float64}, Int32, Ptr{Float64}, Ptr{Float64}),
            wa.handle, pcof, length(pcof), ep, length(ep), sh, out))
    end
    return out
end

# plot_results / the verbose report need the history only through reductions: they stay on the device
# (199 MB at cnot3 never cross PCIe).  marginalize3: src/plotstatectrl.jl:405-423
function marginalize3(params::objparams, pcof::Vector{Float64}, wa::AbstractWorkingArraysHIP; every::Int = 1)
    Ntot = params.N + params.Nguard
    nout = div(params.nsteps, every) + 1
    grp  = Int32[div(k - 1, params.Nt[1] * params.Nt[2]) for k in 1:Ntot]
    pop  = zeros(params.Nt[3], params.N, nout)
    sync!(wa, params)
    jqcheck(wa, ccall((:jq_state_populations, libjq), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Int32, Ptr{Int32}, Int32, Int32, Int32, Ptr{Float64}, Ptr{Float64}),
        wa.handle, pcof, length(pcof), grp, params.Nt[3], every, nout, pop, C_NULL))
    return pop
end

# estimate_Neumann! (src/evalobjgrad.jl:2922-2925) mutates params.linear_solver.max_iter; sync! picks it up.  The direct
# setter, for callers that bypass params:
set_neumann_terms!(wa::AbstractWorkingArraysHIP, m::Integer) =
    jqcheck(wa, ccall((:jq_set_neumann_terms, libjq), Cint, (Ptr{Cvoid}, Int32), wa.handle, m))

function last_timing(wa::AbstractWorkingArraysHIP)
    t = Ref{JQTiming}()
    jqcheck(wa, ccall((:jq_last_timing, libjq), Cint, (Ptr{Cvoid}, Ref{JQTiming}), wa.handle, t))
    return t[]
end

device_count() = ccall((:jq_device_count, libjq), Cint, ())
set_device(d::Integer) = ccall((:jq_set_device, libjq), Cint, (Cint,), d)
version() = unsafe_string(ccall((:jq_version, libjq), Cstring, ()))
