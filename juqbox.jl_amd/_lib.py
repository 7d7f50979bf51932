"""ctypes binding of libjuqbox_hip.so (C ABI: include/juqbox_hip.h).

The library is the product: there is NO CPU fallback.  If the shared object is missing or a symbol
is absent, importing/using the hot path raises immediately (build with
`make -C juqbox.jl_amd/csrc` or `python -c "import __graft_entry__ as g; g.build()"`).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# JQ_LIB: another build of the SAME library (kernel experiments, scripts/exp_variants.sh); never a different backend
LIB_PATH = os.environ.get("JQ_LIB") or os.path.join(_HERE, "libjuqbox_hip.so")

c_dp = ctypes.POINTER(ctypes.c_double)
c_i32 = ctypes.c_int32

JQ_OK, JQ_EINVAL, JQ_EDIM, JQ_EUNSUPPORTED, JQ_EHIP, JQ_ENOMEM = 0, -1, -2, -3, -4, -5
JQ_OPTION_DEFAULT = -2 ** 63      # jq_set_option: back to "not set"


class jq_csc(ctypes.Structure):
    """Julia's SparseMatrixCSC{Float64,Int64}: m, n, colptr, rowval (1-based Int64), nzval"""
    _fields_ = [("m", ctypes.c_int64), ("n", ctypes.c_int64), ("colptr", ctypes.POINTER(ctypes.c_int64)),
                ("rowval", ctypes.POINTER(ctypes.c_int64)), ("nzval", c_dp)]


class jq_problem(ctypes.Structure):
    _fields_ = [("Ntot", c_i32), ("N", c_i32), ("Ncoupled", c_i32), ("Nfreq", c_i32), ("nsteps", c_i32),
                ("neumann_terms", c_i32), ("objFuncType", c_i32), ("Nunc", c_i32), ("T", ctypes.c_double),
                ("Hconst", c_dp), ("Hsym_ops", c_dp), ("Hanti_ops", c_dp), ("Uinit", c_dp), ("Utarget_r", c_dp),
                ("Utarget_i", c_dp), ("wmat_real_diag", c_dp), ("Cfreq", c_dp), ("Hunc_ops", c_dp), ("Rfreq", c_dp),
                ("Hconst_csc", ctypes.POINTER(jq_csc)), ("Hsym_csc", ctypes.POINTER(jq_csc)), ("Hanti_csc", ctypes.POINTER(jq_csc))]


class jq_timing(ctypes.Structure):
    _fields_ = [("ms_total", ctypes.c_double), ("ms_propagate", ctypes.c_double), ("ms_generate", ctypes.c_double),
                ("ms_forward", ctypes.c_double), ("ms_backward", ctypes.c_double),
                ("n_forward_launches", ctypes.c_int64), ("n_backward_launches", ctypes.c_int64), ("mfma_executed", ctypes.c_int64), ("svts", ctypes.c_int64),
                ("kernel_family", ctypes.c_int32), ("kernel_size", ctypes.c_int32), ("kernel_band", ctypes.c_int32),
                ("kernel_variant", ctypes.c_int32), ("mfma_backward", ctypes.c_int64),
                ("ms_allreduce", ctypes.c_double), ("ms_shard_min", ctypes.c_double), ("ms_shard_max", ctypes.c_double)]


JQ_ABI_VERSION = 5      # the struct layouts above (include/juqbox_hip.h JQ_ABI_VERSION); load() refuses any other library


# every symbol include/juqbox_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "jq_device_count": (ctypes.c_int, []),
    "jq_set_device": (ctypes.c_int, [ctypes.c_int]),
    "jq_create": (ctypes.c_int, [ctypes.POINTER(jq_problem), ctypes.POINTER(ctypes.c_void_p)]),
    "jq_create_opts": (ctypes.c_int, [ctypes.POINTER(jq_problem), ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]),
    "jq_set_option": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64]),
    "jq_get_option": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64)]),
    "jq_destroy": (None, [ctypes.c_void_p]),
    "jq_last_error": (ctypes.c_char_p, [ctypes.c_void_p]),
    "jq_set_neumann_terms": (ctypes.c_int, [ctypes.c_void_p, c_i32]),
    "jq_set_linear_solver": (ctypes.c_int, [ctypes.c_void_p, c_i32, c_i32, ctypes.c_double]),
    "jq_set_integrator": (ctypes.c_int, [ctypes.c_void_p, c_i32, c_i32, ctypes.c_double]),
    "jq_update_target": (ctypes.c_int, [ctypes.c_void_p, c_dp, c_dp]),
    "jq_update_hconst": (ctypes.c_int, [ctypes.c_void_p, c_dp]),
    "jq_update_hconst_csc": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(jq_csc)]),
    "jq_update_wmat_diag": (ctypes.c_int, [ctypes.c_void_p, c_dp]),
    "jq_update_wmat": (ctypes.c_int, [ctypes.c_void_p, c_dp, c_dp]),
    "jq_traceobjgrad": (ctypes.c_int, [ctypes.c_void_p, c_dp, c_i32, c_i32, c_dp, c_dp, c_dp, c_dp]),
    "jq_state_history": (ctypes.c_int, [ctypes.c_void_p, c_dp, c_i32, c_dp, c_dp]),
    "jq_traceobj_verbose": (ctypes.c_int, [ctypes.c_void_p, c_dp, c_i32, c_dp, c_dp, c_dp]),
    "jq_state_populations": (ctypes.c_int, [ctypes.c_void_p, c_dp, c_i32, ctypes.POINTER(ctypes.c_int32), c_i32, c_i32, c_i32,
                                            c_dp, c_dp]),
    "jq_eval_f_g_grad": (ctypes.c_int, [ctypes.c_void_p, c_dp, c_i32, c_dp, c_dp, c_i32, c_dp, c_i32, c_dp, c_dp, c_dp]),
    "jq_eval_f_g_grad_dev": (ctypes.c_int, [ctypes.c_void_p, c_dp, c_i32, c_dp, c_dp, c_i32, c_dp, c_i32, ctypes.c_void_p]),
    "jq_create_multi": (ctypes.c_int, [ctypes.POINTER(jq_problem), ctypes.POINTER(ctypes.c_int32), c_i32,
                                       ctypes.POINTER(ctypes.c_void_p)]),
    "jq_create_multi_opts": (ctypes.c_int, [ctypes.POINTER(jq_problem), ctypes.POINTER(ctypes.c_int32), c_i32, ctypes.c_char_p,
                                            ctypes.POINTER(ctypes.c_void_p)]),
    "jq_rccl_world_size": (ctypes.c_int, [ctypes.c_void_p]),
    "jq_num_devices": (ctypes.c_int, [ctypes.c_void_p]),
    "jq_handle_device": (ctypes.c_int, [ctypes.c_void_p]),
    "jq_num_compute_units": (ctypes.c_int, [ctypes.c_void_p]),
    "jq_shard_bounds": (ctypes.c_int, [c_i32, c_i32, c_i32, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    "jq_traceobj_sweep": (ctypes.c_int, [ctypes.c_void_p, c_dp, c_i32, c_dp, c_i32, c_dp, c_dp]),
    "jq_plan_info": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, c_i32]),
    "jq_last_timing": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(jq_timing)]),
    "jq_version": (ctypes.c_char_p, []),
    "jq_abi_version": (ctypes.c_int, []),
}

_lib = None


class JuqboxHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libjuqbox_hip error %d: %s" % (code, msg))
        self.code = code


def load():
    """Load the C-ABI library (once) and bind every declared symbol.  Raises if anything is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libjuqbox_hip.so not built: %s is missing (run `make -C juqbox.jl_amd/csrc`); "
                              "there is no CPU fallback for the hot path" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)          # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if L.jq_abi_version() != JQ_ABI_VERSION:
            raise ImportError("libjuqbox_hip.so has ABI version %d, this binding is written for %d (rebuild: make -C "
                              "juqbox.jl_amd/csrc)" % (L.jq_abi_version(), JQ_ABI_VERSION))
        _lib = L
    return _lib


def check(rc, handle=None):
    if rc != JQ_OK:
        msg = load().jq_last_error(handle)
        raise JuqboxHipError(rc, msg.decode() if msg else "?")
