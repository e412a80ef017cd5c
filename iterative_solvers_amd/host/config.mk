# config.mk — compile-time knobs of the host driver, same role and names as the reference's
# config.mk (reference config.mk:1-67); the values become -D macros (reference Makefile:4).
# [hip]                      the only compiler branch of this build (reference: gcc/icc/icx/nvcc)
COMPILER = hip
# [int]
MAX_ITERS = 5000
# [float]
TOL = 1e-14
# [int]
GMRES_RESTART_LEN = 100
# [double]                   fp64 is the working precision of the gfx950 kernels
WORKING_PRECISION = double
# [crs, vcsr, sell]          SpMV format of the solver loops: crs = bit-exact scalar-CSR stream
#                            kernel, vcsr = sub-wavefront vector-CSR, sell = SELL-64-sigma
SPMV_FORMAT = crs
# [shipped, ref]             Jacobi sweep: shipped = bug-compatible with methods/jacobi.hpp:191,
#                            ref = jacobi_iteration_ref_cpu (correct)
JACOBI_MODE = shipped
# [level, multicolor, auto]  Gauss-Seidel sweep order.  level = the reference's natural order (methods/gauss_seidel.hpp:227,
#                            kernels.hpp:451-473), the iteration every golden output was recorded with.  multicolor = the
#                            same sweep on P A P^T, rows colour by colour: a different (documented) iteration, 8x faster
#                            per trip on a large 3-D stencil, whose natural order is one launch per grid hyperplane.
#                            auto = level unless the natural order has more than 600 dependency levels on a matrix
#                            of a million rows or more (then multicolor; its_solve says so on stderr).
GS_ORDER = level
# [dense, fast]              GMRES least squares on the host: dense = the reference's (m+1)^2
#                            products (its stopping behaviour at TOL=1e-14 exactly), fast = O(m) rotations
GMRES_GIVENS = dense
# [int]                      HIP device
DEVICE = 0
