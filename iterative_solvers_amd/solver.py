"""Host-side mirror of the reference's driver, harness and Solver interface, bound to the gfx950
library through the C ABI (``capi.py``).  Names, argument meaning and bookkeeping follow the
reference so a user (and the parity tests) can read one against the other:

* ``read_mtx``                 — io_funcs.hpp:125-289 (symmetric expansion, stable row sort)
* ``preprocessing``            — main.cpp:96-329 (scaling, CRS, D, b, x0, r0, stopping value)
* ``Solver``                   — solver.hpp (iterate / exchange_arrays / restart_gmres / save_x_star)
* ``solve``                    — solver_harness.hpp:14-150 (the do..while loop + history rules)
* ``LoopParams`` / ``Flags``   — structs.hpp:17-69, literals of main.cpp:550-571

Only one-time host setup is done in numpy here (text parsing, scaling, L/U split); every
per-iteration operation runs on the GPU.  Nothing in this module touches ``oracle/``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

from . import capi

SOLVER_KIND = {"jacobi": capi.JACOBI, "gauss-seidel": capi.GAUSS_SEIDEL,
               "conjugate-gradient": capi.CG, "gmres": capi.GMRES}
CLI_SOLVER = {"-j": "jacobi", "-gs": "gauss-seidel", "-cg": "conjugate-gradient", "-gm": "gmres"}


@dataclass
class LoopParams:                      # structs.hpp:29-69, defaults = main.cpp:561-571 + config.mk
    iter_count: int = 0
    residual_count: int = 0
    residual_check_len: int = 1
    max_iters: int = 5000
    stopping_criteria: float = 0.0
    tol: float = 1e-14
    init_b: float = 1.0
    init_x: float = 1.0
    gmres_restart_len: int = 100

    def compute_stopping_criteria(self, solver_type: str, r0_inf: float, r0_two: float) -> float:
        """tol·‖r0‖₂ for GMRES, tol·‖r0‖∞ otherwise (structs.hpp:41-68)."""
        return self.tol * (r0_two if solver_type == "gmres" else r0_inf)


@dataclass
class Flags:                           # structs.hpp:17-26
    print_iters: bool = False
    print_summary: bool = True
    print_residuals: bool = True
    convergence_flag: bool = False


# --------------------------------------------------------------------------- input
def read_mtx(path: str):
    """Matrix Market coordinate reader with the reference's semantics; returns row-sorted COO."""
    with open(path, "r") as f:
        banner = f.readline().split()
        if len(banner) < 5 or banner[0] != "%%MatrixMarket":          # mmio.h:20
            raise ValueError("mm_read_banner: not a Matrix Market file")
        obj, fmt, fld, sym = (s.lower() for s in banner[1:5])
        if obj != "matrix" or fmt != "coordinate" or fld not in ("real", "integer", "pattern") \
                or sym not in ("general", "symmetric"):
            raise ValueError("The matrix market file provided is not supported.")   # io_funcs.hpp:155-174
        line = f.readline()
        while line.startswith("%"):
            line = f.readline()
        M, N, nz = (int(t) for t in line.split())
        if M != N:
            raise ValueError("Matrix not square. Currently only square matrices are supported")
        data = np.loadtxt(f, ndmin=2) if nz else np.zeros((0, 3))
    I = data[:, 0].astype(np.int64) - 1
    J = data[:, 1].astype(np.int64) - 1
    V = np.full(nz, 0.01) if fld == "pattern" else data[:, 2].astype(np.float64)   # mmio.h:195-203
    if sym == "symmetric":                                             # io_funcs.hpp:193-247
        off = I != J
        reps = 1 + off.astype(np.int64)
        pos = np.cumsum(reps) - reps
        tot = int(reps.sum())
        I2 = np.empty(tot, np.int64); J2 = np.empty(tot, np.int64); V2 = np.empty(tot, np.float64)
        I2[pos] = I; J2[pos] = J; V2[pos] = V
        I2[pos[off] + 1] = J[off]; J2[pos[off] + 1] = I[off]; V2[pos[off] + 1] = V[off]
        I, J, V = I2, J2, V2
    order = np.argsort(I, kind="stable")                               # io_funcs.hpp:12-19,258
    return M, I[order], J[order], V[order]


def scale_coo(n, I, J, V, scale_type: str):
    """Equilibration of preprocessing() (main.cpp:117-132); returns (V_scaled, row_scale)."""
    rs = np.ones(n); cs = np.ones(n)
    if scale_type == "max":            # shipped copy-paste bug: both passes write the row vector
        np.maximum.at(rs, I, np.abs(V))
        np.maximum.at(rs, J, np.abs(V))
    elif scale_type == "diag":
        d = I == J
        s = np.sqrt(np.abs(V[d]))
        rs[I[d]] = s; cs[I[d]] = s      # later duplicates win, as in the serial sweep
    elif scale_type != "none":
        raise ValueError("Please choose an available matrix scaling type: max, diag, none")
    V = V / rs[I]
    V = V / cs[J]
    return V, rs


def coo_to_crs(n, I, J, V):
    """convert_to_crs (utility_funcs.hpp:326-377) for a row-sorted COO."""
    row_ptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(I, minlength=n), out=row_ptr[1:])
    return row_ptr.astype(np.int32), J.astype(np.int32), V.astype(np.float64)


def extract_diag(n, row_ptr, col, val):
    """extract_diag (utility_funcs.hpp:157-174); rows without a diagonal keep 0."""
    D = np.zeros(n)
    rows = np.repeat(np.arange(n), np.diff(row_ptr))
    d = rows == col
    D[rows[d]] = val[d]
    return D


def split_L_U(n, row_ptr, col, val):
    """split_L_U + convert_to_crs (utility_funcs.hpp:240-324): strict lower / strict upper."""
    rows = np.repeat(np.arange(n), np.diff(row_ptr))
    out = []
    for mask in (col < rows, col > rows):
        rp = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(np.bincount(rows[mask], minlength=n), out=rp[1:])
        out.append((rp.astype(np.int32), col[mask].astype(np.int32), val[mask].astype(np.float64)))
    return out[0], out[1]


# --------------------------------------------------------------------------- Solver
class Solver:
    """Device-resident counterpart of ``Solver<VT>`` (solver.hpp:15-974)."""

    def __init__(self, solver_type: str, preconditioner_type: str = "", *, device: int = -1,
                 spmv_format: int = capi.FMT_CSR_EXACT, jacobi_mode: int = capi.JACOBI_SHIPPED,
                 gs_order: int = capi.GS_LEVEL):
        if solver_type not in SOLVER_KIND:
            raise ValueError("Please choose an available solver type: -j -gs -gm -cg")
        self.solver_type = solver_type
        self.preconditioner_type = preconditioner_type
        self.spmv_format = spmv_format
        self.jacobi_mode = jacobi_mode
        self.gs_order = gs_order
        self.handle = capi.Handle(device)
        self.n = 0
        self.x_star = None
        self.gmres_restart_length = 100
        self.perm = None               # multicolour ordering (new -> old) when gs_order == GS_MULTICOLOR
        self.n_colors = 0

    # allocate_cpu_solver_structs + convert_to_crs + extract_diag (main.cpp:206-229,268)
    def upload(self, row_ptr, col, val, b, x0):
        import time
        t = [time.perf_counter()]
        self.setup_times = {}

        def lap(name):
            t.append(time.perf_counter()); self.setup_times[name] = t[-1] - t[-2]

        n = self.n = len(row_ptr) - 1
        if self.solver_type == "gauss-seidel" and self.gs_order == capi.GS_MULTICOLOR:
            # multicolour sweep = the reference's natural-order sweep on P A P^T with rows ordered
            # colour by colour (include/its_hip.h, "Gauss-Seidel ordering helpers")
            self.perm, self.n_colors = capi.multicolor_order(row_ptr, col)
            lap("multicolor_order")
            row_ptr, col, val = capi.permute_crs(row_ptr, col, val, self.perm)
            b = np.asarray(b)[self.perm]; x0 = np.asarray(x0)[self.perm]
            lap("permute_crs")
        self.handle.upload_crs(row_ptr, col, val, capi.MAT_A, self.spmv_format)
        lap("upload_A")
        need_L = self.solver_type == "gauss-seidel" or (self.solver_type == "gmres" and self.preconditioner_type == "gauss-seidel")
        if need_L:                                                     # gauss_seidel.hpp:249-630
            (Lp, Lc, Lv), (Up, Uc, Uv) = capi.split_lu(row_ptr, col, val)      # OpenMP host pass of the library
            lap("split_L_U")
            self.handle.upload_crs(Lp, Lc, Lv, capi.MAT_L, capi.FMT_CSR_EXACT)
            self.handle.upload_crs(Up, Uc, Uv, capi.MAT_U, capi.FMT_CSR_EXACT)
            lap("upload_L_U")
        D = capi.extract_diag(row_ptr, col, val)
        lap("extract_diag")
        self.handle.upload_vectors(D, b, x0)
        lap("upload_vectors")

    def init_residual(self):                                           # solver.hpp:838-855
        return self.handle.init_residual()

    def init_structs(self):                                            # solver.hpp:381-401
        pre = {"": capi.PRECOND_NONE, "jacobi": capi.PRECOND_JACOBI, "gauss-seidel": capi.PRECOND_GS}[self.preconditioner_type]
        self.handle.solver_setup(SOLVER_KIND[self.solver_type], self.gmres_restart_length, pre,
                                 self.jacobi_mode, self.gs_order)

    def iterate(self, iter_count: int) -> float:                       # solver.hpp:403-550 (+ record, + exchange_arrays)
        return self.handle.iterate(iter_count)

    def exchange_arrays(self):                                         # solver.hpp:772-836: done on the device inside iterate
        pass

    def restart_gmres(self, iter_count: int):                          # solver.hpp:552-656
        self.handle.gmres_restart(iter_count)

    def save_x_star(self, iter_count: int):                            # solver.hpp:658-726
        x = self.handle.download_x(self.n, iter_count)
        if self.perm is not None:
            out = np.empty_like(x); out[self.perm] = x; x = out
        self.x_star = x
        return self.x_star


@dataclass
class Args:                             # argType<VT> (structs.hpp:71-119), the parts the loop uses
    solver_type: str
    loop_params: LoopParams = field(default_factory=LoopParams)
    flags: Flags = field(default_factory=Flags)
    preconditioner_type: str = ""
    scale_type: str = "diag"            # structs.hpp:96
    normed_residuals: np.ndarray | None = None
    n: int = 0


def preprocessing(args: Args, solver: Solver, n, I, J, V):
    """main.cpp:96-329 from a row-sorted COO."""
    V, row_scale = scale_coo(n, I, J, V, args.scale_type)
    lp = args.loop_params
    args.normed_residuals = np.zeros(lp.max_iters // lp.residual_check_len + 1)   # structs.hpp:103-108
    row_ptr, col, val = coo_to_crs(n, I, J, V)
    b = np.full(n, lp.init_b) / (row_scale * row_scale)                # main.cpp:271-272, utility_funcs.hpp:518-530
    x0 = np.full(n, lp.init_x) / (row_scale * row_scale)               # main.cpp:275-276
    return preprocessing_crs(args, solver, row_ptr, col, val, b, x0)


def preprocessing_crs(args: Args, solver: Solver, row_ptr, col, val, b, x0):
    """The tail of preprocessing() for callers that already hold CRS + scaled vectors."""
    lp = args.loop_params
    args.n = len(row_ptr) - 1
    if args.normed_residuals is None:
        args.normed_residuals = np.zeros(lp.max_iters // lp.residual_check_len + 1)
    solver.gmres_restart_length = lp.gmres_restart_len
    solver.upload(row_ptr, col, val, b, x0)
    r0_inf, r0_two = solver.init_residual()                            # main.cpp:307
    solver.init_structs()                                              # main.cpp:317
    lp.stopping_criteria = lp.compute_stopping_criteria(args.solver_type, r0_inf, r0_two)   # main.cpp:319
    return r0_inf, r0_two


def _record(args: Args, residual_norm: float):
    """History bookkeeping of record_residual_norm (utility_funcs.hpp:488-502)."""
    lp = args.loop_params
    args.normed_residuals[lp.residual_count] = residual_norm
    if args.flags.convergence_flag:
        pass
    elif lp.iter_count == lp.max_iters:
        pass
    elif residual_norm < lp.stopping_criteria:
        pass
    else:
        lp.residual_count += 1


def solve(args: Args, solver: Solver):
    """solve_cpu (solver_harness.hpp:14-150) with the device Solver."""
    lp = args.loop_params
    residual_norm = 0.0
    while True:
        residual_norm = solver.iterate(lp.iter_count)                  # :43-50 (+ :57-84 value)
        if lp.iter_count % lp.residual_check_len == 0:
            _record(args, residual_norm)
        solver.exchange_arrays()                                       # :90
        if (args.solver_type == "gmres" and residual_norm > lp.stopping_criteria
                and lp.iter_count < lp.max_iters
                and (lp.iter_count + 1) % solver.gmres_restart_length == 0):
            solver.restart_gmres(lp.iter_count)                        # :92-104
        lp.iter_count += 1
        if not (residual_norm > lp.stopping_criteria and lp.iter_count < lp.max_iters
                and not math.isinf(residual_norm)):                    # :114-116
            break
    args.flags.convergence_flag = (residual_norm <= lp.stopping_criteria) and not math.isinf(residual_norm)
    solver.save_x_star(lp.iter_count)                                  # :120
    _record(args, solver.handle.final_residual())                      # :122-149
    return residual_norm


def summary_lines(args: Args):
    """The text of summary_output (io_funcs.hpp:291-332), one string per line."""
    lp = args.loop_params
    out = [f"||A*x_{i * lp.residual_check_len} - b||_infty = {args.normed_residuals[i]:.16g}"
           for i in range(lp.residual_count)]
    out.append("")
    if args.flags.convergence_flag:
        out.append(f"{args.solver_type} solver converged in: {lp.iter_count} iterations.")
    else:
        out.append(f"{args.solver_type} solver did not converge after {lp.max_iters} iterations.")
    out.append(f"With the stopping criteria \"tol * || b-A*x_0 ||_infty\" is: {lp.stopping_criteria:.16g}")
    out.append("The residual of the final iteration is: ||A*x_star - b||_infty = "
               f"{args.normed_residuals[lp.residual_count]:.16e}.")
    return out


def run_solver(mtx_path: str, solver_flag: str, *, scale="diag", precond="", max_iters=5000, tol=1e-14,
               restart_len=100, spmv_format=capi.FMT_CSR_EXACT, jacobi_mode=capi.JACOBI_SHIPPED,
               gs_order=capi.GS_LEVEL, device=-1):
    """run_solver (main.cpp:331-520): read, preprocess, solve; returns (args, solver)."""
    solver_type = CLI_SOLVER[solver_flag]
    pre = {"": "", "j": "jacobi", "gs": "gauss-seidel"}[precond]
    args = Args(solver_type, LoopParams(max_iters=max_iters, tol=tol, gmres_restart_len=restart_len),
                Flags(), pre, scale)
    solver = Solver(solver_type, pre, device=device, spmv_format=spmv_format, jacobi_mode=jacobi_mode,
                    gs_order=gs_order)
    n, I, J, V = read_mtx(mtx_path)
    preprocessing(args, solver, n, I, J, V)
    solve(args, solver)
    return args, solver
