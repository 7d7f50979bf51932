"""MI355X-native drop-in for Juqbox.jl's Stormer-Verlet `traceobjgrad` hot path (host-side Python
mirror of the reference's objparams / traceobjgrad / Ipopt-callback surface over a C-ABI library of
hand-written gfx950 HIP kernels).  See DESIGN.md / INTEGRATION.md."""
from . import cases, pcof_io, plotstatectrl, setup_utils  # noqa: F401
from .plotstatectrl import (forbidden_level_maxima, identify_forbidden_levels, identify_guard_levels,  # noqa: F401
                            marginalize3, marginalize3_device, specify_level3, state_populations)
from .pcof_io import read_dat, read_jld2, read_pcof, save_dat, save_pcof  # noqa: F401
from .evalobjgrad import Working_Arrays_HIP, Working_Arrays_M_HIP, options, traceobjgrad  # noqa: F401
from .ipopt_interface import (eval_f_g_grad, eval_f_par, eval_g_par, eval_grad_f_par,  # noqa: F401
                              eval_jac_g_par, intermediate_par, run_optimizer, setup_ipopt_problem,
                              traceobj_sweep)
from .objparams import (JACOBI_SOLVER, JACOBI_SOLVER_M, NEUMANN_SOLVER, Implicit_Midpoint,  # noqa: F401
                        Stormer_Verlet, lsolver_object, objparams)
