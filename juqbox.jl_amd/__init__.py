"""MI355X-native drop-in for Juqbox.jl's Stormer-Verlet `traceobjgrad` hot path (host-side Python
mirror of the reference's objparams / traceobjgrad / Ipopt-callback surface over a C-ABI library of
hand-written gfx950 HIP kernels).  See DESIGN.md / INTEGRATION.md."""
from . import cases, setup_utils  # noqa: F401
from .objparams import (JACOBI_SOLVER, NEUMANN_SOLVER, Stormer_Verlet, lsolver_object,  # noqa: F401
                        objparams)
