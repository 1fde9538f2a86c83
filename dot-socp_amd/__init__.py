"""dot-socp_amd: MI355X-native (gfx950, HIP) implementation of the inPALM/ADMM SOCP iteration
loop of chlhnu/DOT-SOCP behind the reference's own entry points.

Import as `import dotsocp_amd` (alias module at the repository root; the directory name
`dot-socp_amd` is not a Python identifier).  The compute path is lib/libdotsocp.so -- there is
no CPU fallback; see include/dotsocp.h for the C ABI and INTEGRATION.md for the MATLAB binding.
"""
from . import capi  # noqa: F401
from .examples import (ensure_barrier_validity, gene_barrier_of_circle_pillar, gene_barrier_of_love_heart,  # noqa: F401
                       gene_weight_circle, gene_weight_circleInv, get_example_1d, get_example_2d,
                       get_weight_by_barrier)
from .mexops import (mexBFd, mexBFd1d, mexBFdConj, mexBFdConj1d, mexProjSoc, mirt_dctn, mirt_idctn,  # noqa: F401
                     oper_poisson, oper_poisson3dim)
from .model import (InitialScaling, ModelHandle, VarHandle, check_massConservation, initialize,  # noqa: F401
                    initialize_slab, recover_q, recover_RhoE, recoverOrgVar)
from .solvers import (InPALMContext, solver_dotsocp1d, solver_dotsocp2d, solver_socp_accADMM,  # noqa: F401
                      solver_socp_inPALM, solver_socp_PALM, solver_wdotsocp2d, solver_wsocp_accADMM, solver_wsocp_inPALM)
