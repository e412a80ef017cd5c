"""ctypes view of oracle/libits_oracle.so — the CPU restatement of the reference.

TEST INFRASTRUCTURE: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
only.  Never import this from iterative_solvers_amd/.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(_ROOT, "oracle", "libits_oracle.so")

JACOBI, GAUSS_SEIDEL, CG, GMRES = 0, 1, 2, 3
SCALE = {"none": 0, "diag": 1, "max": 2}
PRECOND = {"": 0, "j": 1, "gs": 2}


class Params(C.Structure):
    _fields_ = [("solver", C.c_int), ("jacobi_mode", C.c_int), ("precond", C.c_int), ("max_iters", C.c_int),
                ("restart_len", C.c_int), ("tol", C.c_double), ("fixed_iters", C.c_int)]


class Result(C.Structure):
    _fields_ = [("iters", C.c_int), ("residual_count", C.c_int), ("converged", C.c_int), ("stopping", C.c_double),
                ("final_residual", C.c_double), ("r0_inf", C.c_double), ("r0_two", C.c_double),
                ("solve_seconds", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB)
        L.orc_dot.restype = C.c_double
        L.orc_nrm2.restype = C.c_double
        L.orc_norm_inf.restype = C.c_double
        L.orc_num_threads.restype = C.c_int
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def read_mtx(path):
    L = lib()
    n, nc, nnz = C.c_int(), C.c_int(), C.c_long()
    I, J, V = C.c_void_p(), C.c_void_p(), C.c_void_p()
    rc = L.orc_read_mtx(path.encode(), C.byref(n), C.byref(nc), C.byref(nnz), C.byref(I), C.byref(J), C.byref(V))
    if rc != 0:
        raise RuntimeError(f"orc_read_mtx({path}) = {rc}")
    k = nnz.value
    Ia = np.ctypeslib.as_array(C.cast(I, C.POINTER(C.c_int)), (max(k, 1),))[:k].copy()
    Ja = np.ctypeslib.as_array(C.cast(J, C.POINTER(C.c_int)), (max(k, 1),))[:k].copy()
    Va = np.ctypeslib.as_array(C.cast(V, C.POINTER(C.c_double)), (max(k, 1),))[:k].copy()
    for p in (I, J, V):
        L.orc_free(p)
    return n.value, Ia, Ja, Va


def scale_coo(n, I, J, V, kind):
    V = _f64(V).copy(); I = _i32(I); J = _i32(J)
    rs = np.empty(n)
    lib().orc_scale_coo(C.c_int(n), C.c_long(len(V)), _p(I), _p(J), _p(V), C.c_int(SCALE[kind]), _p(rs))
    return V, rs


def coo_to_crs(n, I, J, V):
    I = _i32(I); J = _i32(J); V = _f64(V)
    rp = np.empty(n + 1, np.int32); col = np.empty(len(V), np.int32); val = np.empty(len(V))
    rc = lib().orc_coo_to_crs(C.c_int(n), C.c_long(len(V)), _p(I), _p(J), _p(V), _p(rp), _p(col), _p(val))
    assert rc == 0
    return rp, col, val


def extract_diag(rp, col, val):
    n = len(rp) - 1
    D = np.zeros(n)
    lib().orc_extract_diag(C.c_int(n), _p(_i32(rp)), _p(_i32(col)), _p(_f64(val)), _p(D))
    return D


def split_LU(rp, col, val):
    n = len(rp) - 1
    rp = _i32(rp); col = _i32(col); val = _f64(val)
    Lp = np.empty(n + 1, np.int32); Up = np.empty(n + 1, np.int32)
    lib().orc_split_LU(C.c_int(n), _p(rp), _p(col), _p(val), _p(Lp), None, None, _p(Up), None, None)
    Lc = np.empty(Lp[n], np.int32); Lv = np.empty(Lp[n]); Uc = np.empty(Up[n], np.int32); Uv = np.empty(Up[n])
    lib().orc_split_LU(C.c_int(n), _p(rp), _p(col), _p(val), _p(Lp), _p(Lc), _p(Lv), _p(Up), _p(Uc), _p(Uv))
    return (Lp, Lc, Lv), (Up, Uc, Uv)


def spmv(rp, col, val, x):
    n = len(rp) - 1
    y = np.empty(n)
    lib().orc_spmv(C.c_int(n), _p(_i32(rp)), _p(_i32(col)), _p(_f64(val)), _p(_f64(x)), _p(y))
    return y


def dot(a, b):
    a = _f64(a); b = _f64(b)
    return lib().orc_dot(_p(a), _p(b), C.c_long(len(a)))


def nrm2(a):
    a = _f64(a)
    return lib().orc_nrm2(_p(a), C.c_long(len(a)))


def norm_inf(a):
    a = _f64(a)
    return lib().orc_norm_inf(_p(a), C.c_long(len(a)))


def sum_vectors(a, b, s):
    a = _f64(a); b = _f64(b); out = np.empty_like(a)
    lib().orc_sum_vectors(_p(out), _p(a), _p(b), C.c_long(len(a)), C.c_double(s))
    return out


def subtract_vectors(a, b, s):
    a = _f64(a); b = _f64(b); out = np.empty_like(a)
    lib().orc_subtract_vectors(_p(out), _p(a), _p(b), C.c_long(len(a)), C.c_double(s))
    return out


def scale(v, s):
    v = _f64(v); out = np.empty_like(v)
    lib().orc_scale(_p(out), _p(v), C.c_double(s), C.c_long(len(v)))
    return out


def spltsv(Lp, Lc, Lv, D, rhs, x=None):
    n = len(Lp) - 1
    x = np.zeros(n) if x is None else _f64(x).copy()
    lib().orc_spltsv(C.c_int(n), _p(_i32(Lp)), _p(_i32(Lc)), _p(_f64(Lv)), _p(x), _p(_f64(D)), _p(_f64(rhs)))
    return x


def solve(rp, col, val, b, x0, solver, *, jacobi_mode=0, precond=0, max_iters=5000, restart_len=100, tol=1e-14,
          fixed_iters=0):
    """Whole solve; returns dict(history=<printed lines' values>, iters, converged, stopping, final, x_star, ...)."""
    n = len(rp) - 1
    rp = _i32(rp); col = _i32(col); val = _f64(val); b = _f64(b); x0 = _f64(x0)
    p = Params(solver, jacobi_mode, precond, max_iters, restart_len, tol, fixed_iters)
    hist = np.zeros(max(max_iters, fixed_iters) + 2)
    xs = np.empty(n)
    r = Result()
    lib().orc_solve(C.c_int(n), _p(rp), _p(col), _p(val), _p(b), _p(x0), C.byref(p), _p(hist), _p(xs), C.byref(r))
    return dict(history=hist[:r.residual_count].copy(), iters=r.iters, converged=bool(r.converged),
                stopping=r.stopping, final=r.final_residual, x_star=xs, r0_inf=r.r0_inf, r0_two=r.r0_two,
                seconds=r.solve_seconds, all_values=hist[:r.residual_count + 1].copy())


def write_mtx(path, rp, col, val, n_cols=None):
    """Matrix Market text of a CRS matrix, same text as iterative_solvers_amd.matgen.write_mtx, ten times as fast"""
    n = len(rp) - 1
    rc = lib().orc_write_mtx(str(path).encode(), C.c_int(n), C.c_int(n if n_cols is None else n_cols),
                             _p(_i32(rp)), _p(_i32(col)), _p(_f64(val)))
    if rc != 0:
        raise OSError(f"orc_write_mtx({path}) failed: {rc}")


def num_threads():
    return lib().orc_num_threads()


def set_threads(t: int):
    """1 thread reproduces the compiled reference at OMP_NUM_THREADS=1 bit for bit."""
    lib().orc_set_threads(C.c_int(t))
