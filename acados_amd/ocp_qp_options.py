"""Options of the OCP-QP solver: same attributes, defaults and validation as the reference's
`AcadosOcpQpOptions` (interfaces/acados_template/acados_template/acados_ocp_options.py:2372-2500)."""


class AcadosOcpQpOptions:
    SOLVERS = ("PARTIAL_CONDENSING_GPU_IPM", "PARTIAL_CONDENSING_HPIPM", "FULL_CONDENSING_GPU_IPM")

    def __init__(self):
        self.qp_solver = "PARTIAL_CONDENSING_GPU_IPM"
        self.tol_stat = 1e-6
        self.tol_eq = 1e-6
        self.tol_ineq = 1e-6
        self.tol_comp = 1e-6
        self.iter_max = 50
        self.cond_N = None
        self.cond_block_size = None
        self.warm_start = 0
        self.cond_ric_alg = 1
        self.ric_alg = 1
        self.mu0 = None
        self.t0_init = 2
        self.print_level = 0
        self.hpipm_mode = "BALANCE"

    def make_consistent(self, N: int):
        if self.qp_solver not in self.SOLVERS:
            raise ValueError(f"qp_solver {self.qp_solver} is not provided by acados_amd; possible values: {self.SOLVERS}")
        if self.qp_solver.startswith("FULL_CONDENSING"):
            if self.cond_N not in (None, 1):
                raise ValueError("full condensing condenses to one block: cond_N must be 1 (or unset)")
            self.cond_N = 1
        if self.cond_N is None:
            self.cond_N = N
        if self.cond_block_size is not None and sum(self.cond_block_size) != N:
            raise ValueError("cond_block_size must sum to N")
        if self.warm_start not in (0, 1, 2, 3):
            raise ValueError("warm_start must be 0, 1, 2 or 3")
        if self.ric_alg not in (0, 1):
            raise ValueError(f"Invalid ric_alg value. ric_alg must be in [0, 1], got {self.ric_alg}.")
        if self.ric_alg == 0:
            raise ValueError("ric_alg = 0 (classical Riccati for an indefinite full-space Hessian) is not available in acados_amd: "
                             "every kernel family carries the Cholesky factor of P (ric_alg = 1, the acados default)")
        if self.hpipm_mode not in ("BALANCE", "SPEED_ABS", "SPEED", "ROBUST"):
            raise ValueError("invalid hpipm_mode")
