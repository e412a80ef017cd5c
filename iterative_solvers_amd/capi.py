"""ctypes binding of ``libits_hip.so`` (C ABI declared in ``include/its_hip.h``).

This is the only way Python reaches the gfx950 kernels; there is no CPU fallback: if the shared
library is missing, importing :func:`lib` raises (build it with ``python -c 'import
__graft_entry__ as g; g.build()'`` or ``make -C iterative_solvers_amd/csrc``).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libits_hip.so")

# enums of include/its_hip.h
JACOBI, GAUSS_SEIDEL, CG, GMRES = 0, 1, 2, 3
JACOBI_SHIPPED, JACOBI_REF = 0, 1
PRECOND_NONE, PRECOND_JACOBI, PRECOND_GS = 0, 1, 2
FMT_CSR_EXACT, FMT_CSR_VECTOR, FMT_SELL, FMT_MIXED = 0, 1, 2, 3
AP_DP_SP, AP_DP_HP, AP_SP_HP, AP_DP_SP_HP = 0, 1, 2, 3          # the reference's AP_VALUE_TYPE (config.mk:57)
MAT_A, MAT_L, MAT_U = 0, 1, 2
GS_LEVEL, GS_MULTICOLOR = 0, 1

_c_int_p = C.POINTER(C.c_int)
_c_i64_p = C.POINTER(C.c_int64)
_c_dbl_p = C.POINTER(C.c_double)

# name -> (restype, argtypes); every symbol include/its_hip.h declares
SIGNATURES = {
    "its_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "its_device_count": (C.c_int, []),
    "its_destroy": (C.c_int, [C.c_void_p]),
    "its_last_error": (C.c_char_p, [C.c_void_p]),
    "its_version": (C.c_char_p, []),
    "its_upload_crs": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "its_set_mixed_threshold": (C.c_int, [C.c_void_p, C.c_double]),
    "its_mixed_info": (C.c_int, [C.c_void_p, C.c_int, _c_i64_p, _c_i64_p]),
    "its_set_mixed_mode": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double]),
    "its_mixed_tiers": (C.c_int, [C.c_void_p, C.c_int, _c_i64_p, _c_i64_p, _c_int_p]),
    "its_upload_vectors": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "its_init_residual": (C.c_int, [C.c_void_p, _c_dbl_p, _c_dbl_p]),
    "its_solver_setup": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "its_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "its_iterate": (C.c_int, [C.c_void_p, C.c_int, _c_dbl_p]),
    "its_iterate_n": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "its_gmres_restart": (C.c_int, [C.c_void_p, C.c_int]),
    "its_download_x": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "its_final_residual": (C.c_int, [C.c_void_p, _c_dbl_p]),
    "its_spmv": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "its_dot": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, _c_dbl_p]),
    "its_dot_noalloc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "its_nrm2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, _c_dbl_p]),
    "its_norm_inf": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, _c_dbl_p]),
    "its_axpby": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double]),
    "its_scale": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int64]),
    "its_spltsv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "its_dev_alloc": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t]),
    "its_dev_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "its_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "its_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "its_synchronize": (C.c_int, [C.c_void_p]),
    "its_spmv_timed": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "its_last_region_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "its_last_spmv_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "its_spmv_kernel_info": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_int, _c_i64_p]),
    "its_plan_info": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _c_int_p, _c_dbl_p, _c_int_p]),
    "its_gmres_info": (C.c_int, [C.c_void_p, _c_int_p, _c_int_p]),
    "its_gs_info": (C.c_int, [C.c_void_p, _c_int_p, C.POINTER(C.c_int64), C.c_char_p, C.c_int]),
    "its_sweep_info": (C.c_int, [C.c_void_p, _c_int_p, _c_int_p, _c_int_p, _c_int_p]),
    "its_ell_info": (C.c_int, [C.c_void_p, C.c_int, _c_int_p, _c_int_p, C.POINTER(C.c_int64), _c_int_p]),
    "its_extract_diag": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "its_split_lu": (C.c_int, [C.c_int] + [C.c_void_p] * 9),
    "its_multicolor_order": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, _c_int_p]),
    "its_permute_crs": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "its_comm_unique_id": (C.c_int, [C.c_void_p]),
    "its_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "its_comm_init_host": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "its_dist_set_colors": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "its_dist_setup": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}

_lib = None


def lib() -> C.CDLL:
    """Load libits_hip.so once; raise loudly if it is absent (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
                "Build it with `make -C iterative_solvers_amd/csrc`.")
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class ItsError(RuntimeError):
    pass


def multicolor_order(row_ptr, col):
    """Host-only: (perm_new2old, n_colors) of the greedy colouring (its_multicolor_order)."""
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32); col = np.ascontiguousarray(col, dtype=np.int32)
    n = len(row_ptr) - 1
    perm = np.empty(n, dtype=np.int32); nc = C.c_int()
    rc = lib().its_multicolor_order(n, row_ptr.ctypes.data, col.ctypes.data, perm.ctypes.data, C.byref(nc))
    if rc != 0:
        raise ItsError(f"its_multicolor_order failed ({rc})")
    return perm, nc.value


def extract_diag(row_ptr, col, val):
    """Host-only, OpenMP: extract_diag (utility_funcs.hpp:157-174)."""
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32); col = np.ascontiguousarray(col, dtype=np.int32)
    val = np.ascontiguousarray(val, dtype=np.float64)
    n = len(row_ptr) - 1
    D = np.empty(n)
    rc = lib().its_extract_diag(n, row_ptr.ctypes.data, col.ctypes.data, val.ctypes.data, D.ctypes.data)
    if rc != 0:
        raise ItsError(f"its_extract_diag failed ({rc})")
    return D


def split_lu(row_ptr, col, val):
    """Host-only, OpenMP: strictly lower / strictly upper part in CRS (split_L_U, utility_funcs.hpp:240-324)."""
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32); col = np.ascontiguousarray(col, dtype=np.int32)
    val = np.ascontiguousarray(val, dtype=np.float64)
    n = len(row_ptr) - 1; nnz = len(col)
    Lp = np.empty(n + 1, np.int32); Up = np.empty(n + 1, np.int32)
    Lc = np.empty(nnz, np.int32); Uc = np.empty(nnz, np.int32); Lv = np.empty(nnz); Uv = np.empty(nnz)
    rc = lib().its_split_lu(n, row_ptr.ctypes.data, col.ctypes.data, val.ctypes.data, Lp.ctypes.data, Lc.ctypes.data,
                            Lv.ctypes.data, Up.ctypes.data, Uc.ctypes.data, Uv.ctypes.data)
    if rc != 0:
        raise ItsError(f"its_split_lu failed ({rc})")
    return (Lp, Lc[:Lp[-1]], Lv[:Lp[-1]]), (Up, Uc[:Up[-1]], Uv[:Up[-1]])


def permute_crs(row_ptr, col, val, perm_new2old):
    """Host-only: P A P^T (its_permute_crs)."""
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32); col = np.ascontiguousarray(col, dtype=np.int32)
    val = np.ascontiguousarray(val, dtype=np.float64); perm = np.ascontiguousarray(perm_new2old, dtype=np.int32)
    n = len(row_ptr) - 1
    rp = np.empty(n + 1, np.int32); c = np.empty(len(col), np.int32); v = np.empty(len(col))
    rc = lib().its_permute_crs(n, row_ptr.ctypes.data, col.ctypes.data, val.ctypes.data, perm.ctypes.data,
                               rp.ctypes.data, c.ctypes.data, v.ctypes.data)
    if rc != 0:
        raise ItsError(f"its_permute_crs failed ({rc})")
    return rp, c, v


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def device_count() -> int:
    """HIP devices visible to this process (0: none)."""
    return int(lib().its_device_count())


class Handle:
    """Thin owner of one ``its_handle`` (one GPU, one stream)."""

    def __init__(self, device: int = -1):
        self.L = lib()
        self.h = C.c_void_p()
        rc = self.L.its_create(C.byref(self.h), device)
        if rc != 0:
            raise ItsError(f"its_create failed ({rc}): is a HIP device visible?")
        self._keep = []

    def check(self, rc: int, what: str):
        if rc != 0:
            msg = self.L.its_last_error(self.h)
            raise ItsError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def close(self):
        if self.h:
            self.L.its_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- uploads
    def upload_crs(self, row_ptr, col, val, which=MAT_A, fmt=FMT_CSR_EXACT, n_cols=None):
        row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float64)
        n = len(row_ptr) - 1
        self.check(self.L.its_upload_crs(self.h, which, n, n if n_cols is None else n_cols, len(col),
                                         _ptr(row_ptr), _ptr(col), _ptr(val), fmt), "its_upload_crs")

    def set_mixed_threshold(self, thr: float):
        self.check(self.L.its_set_mixed_threshold(self.h, float(thr)), "its_set_mixed_threshold")

    def set_mixed_mode(self, ap_value_type: int, threshold_1: float, threshold_2: float = 0.0):
        self.check(self.L.its_set_mixed_mode(self.h, int(ap_value_type), float(threshold_1), float(threshold_2)), "its_set_mixed_mode")

    def mixed_tiers(self, which=MAT_A):
        """entries stored as fp64 / fp32 / fp16, bytes of matrix values a product streams, tiered-ELL form or CRS split"""
        nz = (C.c_int64 * 3)(); vb = C.c_int64(); ef = C.c_int()
        self.check(self.L.its_mixed_tiers(self.h, which, nz, C.byref(vb), C.byref(ef)), "its_mixed_tiers")
        return {"nnz": list(nz), "value_bytes": vb.value, "ell_form": bool(ef.value)}

    def mixed_info(self, which=MAT_A):
        a, b = C.c_int64(), C.c_int64()
        self.check(self.L.its_mixed_info(self.h, which, C.byref(a), C.byref(b)), "its_mixed_info")
        return a.value, b.value

    def upload_vectors(self, D, b, x0):
        D = np.ascontiguousarray(D, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        self.check(self.L.its_upload_vectors(self.h, _ptr(D), _ptr(b), _ptr(x0), len(b)), "its_upload_vectors")

    def init_residual(self):
        a, b = C.c_double(), C.c_double()
        self.check(self.L.its_init_residual(self.h, C.byref(a), C.byref(b)), "its_init_residual")
        return a.value, b.value

    def solver_setup(self, kind, restart_len=100, precond=PRECOND_NONE, jacobi_mode=JACOBI_SHIPPED, gs_order=GS_LEVEL):
        self.check(self.L.its_solver_setup(self.h, kind, restart_len, precond, jacobi_mode, gs_order), "its_solver_setup")

    def set_option(self, name: str, value: int):
        self.check(self.L.its_set_option(self.h, name.encode(), value), "its_set_option")

    def iterate(self, iter_count: int) -> float:
        r = C.c_double()
        self.check(self.L.its_iterate(self.h, iter_count, C.byref(r)), "its_iterate")
        return r.value

    def iterate_n(self, iter_count0: int, n_trips: int) -> np.ndarray:
        out = np.empty(n_trips, dtype=np.float64)
        self.check(self.L.its_iterate_n(self.h, iter_count0, n_trips, _ptr(out)), "its_iterate_n")
        return out

    def gmres_restart(self, iter_count: int):
        self.check(self.L.its_gmres_restart(self.h, iter_count), "its_gmres_restart")

    def download_x(self, n: int, iter_count: int) -> np.ndarray:
        x = np.empty(n, dtype=np.float64)
        self.check(self.L.its_download_x(self.h, _ptr(x), n, iter_count), "its_download_x")
        return x

    def final_residual(self) -> float:
        r = C.c_double()
        self.check(self.L.its_final_residual(self.h, C.byref(r)), "its_final_residual")
        return r.value

    # ---- kernels alone
    def spmv_host(self, x, n_rows, which=MAT_A, fmt=FMT_CSR_EXACT) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty(n_rows, dtype=np.float64)
        self.check(self.L.its_spmv(self.h, which, _ptr(x), _ptr(y), 0, fmt), "its_spmv")
        return y

    def spmv_dev(self, x_ptr: int, y_ptr: int, which=MAT_A, fmt=FMT_CSR_EXACT):
        self.check(self.L.its_spmv(self.h, which, C.c_void_p(x_ptr), C.c_void_p(y_ptr), 1, fmt), "its_spmv")

    def spmv_timed(self, x_ptr: int, y_ptr: int, fmt=FMT_CSR_EXACT, reps=20, which=MAT_A, with_dot=False) -> float:
        ms = C.c_float()
        self.check(self.L.its_spmv_timed(self.h, which, C.c_void_p(x_ptr), C.c_void_p(y_ptr), fmt, int(with_dot), reps, C.byref(ms)), "its_spmv_timed")
        return ms.value

    def last_region_ms(self) -> float:
        ms = C.c_float()
        self.check(self.L.its_last_region_ms(self.h, C.byref(ms)), "its_last_region_ms")
        return ms.value

    def last_spmv_ms(self) -> float:
        ms = C.c_float()
        self.check(self.L.its_last_spmv_ms(self.h, C.byref(ms)), "its_last_spmv_ms")
        return ms.value

    def spmv_kernel_info(self, which=MAT_A, with_dot=False):
        """(kernel family name, bytes one launch streams) of the exact SpMV of this matrix."""
        buf = C.create_string_buffer(64); sb = C.c_int64()
        self.check(self.L.its_spmv_kernel_info(self.h, which, int(with_dot), buf, 64, C.byref(sb)), "its_spmv_kernel_info")
        return buf.value.decode(), sb.value

    def plan_info(self, which=MAT_A, fmt=FMT_CSR_EXACT):
        lpr, occ, nb = C.c_int(), C.c_double(), C.c_int()
        self.check(self.L.its_plan_info(self.h, which, fmt, C.byref(lpr), C.byref(occ), C.byref(nb)), "its_plan_info")
        return {"lanes_per_row": lpr.value, "sell_occupancy": occ.value, "n_row_blocks": nb.value}

    def ell_info(self, which=MAT_A):
        """sliced-ELL image of the matrix: kind (0 none, 1 nibble dictionary, 2 byte dictionary, 3 explicit columns), slots
        per row, index words, and for kind 3 the slices that use 8- / 16- / 32-bit column deltas."""
        k, w, iw = C.c_int(), C.c_int(), C.c_int64()
        hist = (C.c_int * 3)()
        self.check(self.L.its_ell_info(self.h, which, C.byref(k), C.byref(w), C.byref(iw), hist), "its_ell_info")
        return {"kind": k.value, "width": w.value, "index_words": iw.value, "slices_by_delta_width": list(hist)}

    def gmres_info(self):
        g, f = C.c_int(), C.c_int()
        self.check(self.L.its_gmres_info(self.h, C.byref(g), C.byref(f)), "its_gmres_info")
        return {"persist_grid": g.value, "fallbacks": f.value}

    def gs_info(self):
        """Gauss-Seidel trips: is the "gs_prefix" form active (the residual pass walks D + U only), its entries, its kernel."""
        r, nz = C.c_int(), C.c_int64()
        buf = C.create_string_buffer(64)
        self.check(self.L.its_gs_info(self.h, C.byref(r), C.byref(nz), buf, 64), "its_gs_info")
        return {"prefix_ready": bool(r.value), "nnz_du": nz.value, "kernel": buf.value.decode()}

    def sweep_info(self):
        """line-pipelined natural-order sweep (csrc/its_lines.hip): workgroups (0 = level schedule), line length, rows per chunk"""
        w, a, c, nl = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self.check(self.L.its_sweep_info(self.h, C.byref(w), C.byref(a), C.byref(c), C.byref(nl)), "its_sweep_info")
        return {"workgroups": w.value, "line_length": a.value, "rows_per_chunk": c.value, "n_levels": nl.value}

    # ---- device memory
    def dev_alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self.check(self.L.its_dev_alloc(self.h, C.byref(p), nbytes), "its_dev_alloc")
        return p.value

    def dev_free(self, p: int):
        self.check(self.L.its_dev_free(self.h, C.c_void_p(p)), "its_dev_free")

    def to_dev(self, a: np.ndarray) -> int:
        a = np.ascontiguousarray(a)
        p = self.dev_alloc(max(a.nbytes, 16))
        if a.nbytes:
            self.check(self.L.its_memcpy_h2d(self.h, C.c_void_p(p), _ptr(a), a.nbytes), "its_memcpy_h2d")
        return p

    def from_dev(self, p: int, n: int, dtype=np.float64) -> np.ndarray:
        out = np.empty(n, dtype=dtype)
        if out.nbytes:
            self.check(self.L.its_memcpy_d2h(self.h, _ptr(out), C.c_void_p(p), out.nbytes), "its_memcpy_d2h")
        return out

    def dot(self, a_ptr, b_ptr, n) -> float:
        r = C.c_double()
        self.check(self.L.its_dot(self.h, C.c_void_p(a_ptr), C.c_void_p(b_ptr), n, C.byref(r)), "its_dot")
        return r.value

    def nrm2(self, v_ptr, n) -> float:
        r = C.c_double()
        self.check(self.L.its_nrm2(self.h, C.c_void_p(v_ptr), n, C.byref(r)), "its_nrm2")
        return r.value

    def norm_inf(self, v_ptr, n) -> float:
        r = C.c_double()
        self.check(self.L.its_norm_inf(self.h, C.c_void_p(v_ptr), n, C.byref(r)), "its_norm_inf")
        return r.value

    def axpby(self, out_ptr, a_ptr, b_ptr, n, s):
        self.check(self.L.its_axpby(self.h, C.c_void_p(out_ptr), C.c_void_p(a_ptr), C.c_void_p(b_ptr), n, s), "its_axpby")

    def scale(self, out_ptr, v_ptr, s, n):
        self.check(self.L.its_scale(self.h, C.c_void_p(out_ptr), C.c_void_p(v_ptr), s, n), "its_scale")

    def spltsv(self, x_ptr, rhs_ptr):
        self.check(self.L.its_spltsv(self.h, C.c_void_p(x_ptr), C.c_void_p(rhs_ptr)), "its_spltsv")

    def synchronize(self):
        self.check(self.L.its_synchronize(self.h), "its_synchronize")
