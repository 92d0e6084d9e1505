"""acados_amd -- MI355X-native batched OCP-QP backend behind acados' ocp_qp plugin surface.

The package holds only what the hot path needs: the HIP kernels + C-ABI (csrc/, headers in
include/acados_amd/) and the host-side mirror of the reference's QP interface.  All compute
goes through libacados_amd_qp.so (hipcc, gfx950); there is no CPU fallback.
"""
from .ocp_qp import AcadosOcpQp, AcadosOcpQpDims
from .ocp_qp_options import AcadosOcpQpOptions
from .ocp_qp_solver import AcadosOcpQpSolver, AcadosOcpQpBatchSolver
from .gpu_batch import OcpQpGpuBatch
from .ocp_qp_condensing import AcadosOcpQpCondensing

__all__ = ["AcadosOcpQp", "AcadosOcpQpDims", "AcadosOcpQpOptions", "AcadosOcpQpSolver",
           "AcadosOcpQpBatchSolver", "OcpQpGpuBatch", "AcadosOcpQpCondensing"]
