// its_oracle_cli — drives the CPU restatement end to end with the reference's CLI
// (io_funcs.hpp:22-123) and prints the reference's summary text (io_funcs.hpp:291-332), so its
// stdout can be diffed against oracle/_ref.  TEST INFRASTRUCTURE ONLY.
#include "its_oracle.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <string>
#include <vector>

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s file.mtx -j|-gs|-cg|-gm [-p j|gs] [-scale max|diag|none] "
                            "[--jacobi-ref] [--max-iters N] [--tol T] [--restart M] [--dump-x F]\n", argv[0]); return 2; }
    orc_params p{}; p.max_iters = 5000; p.tol = 1e-14; p.restart_len = 100;
    std::string st = argv[2], name;
    if (st == "-j") { p.solver = ORC_JACOBI; name = "jacobi"; }
    else if (st == "-gs") { p.solver = ORC_GAUSS_SEIDEL; name = "gauss-seidel"; }
    else if (st == "-cg") { p.solver = ORC_CG; name = "conjugate-gradient"; }
    else if (st == "-gm") { p.solver = ORC_GMRES; name = "gmres"; }
    else { fprintf(stderr, "unknown solver\n"); return 2; }
    int scale = ORC_SCALE_DIAG;                                     // structs.hpp:96
    const char *dump = nullptr;
    for (int i = 3; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "-p" && i + 1 < argc) { std::string v = argv[++i]; p.precond = v == "gs" ? ORC_PRECOND_GS : ORC_PRECOND_JACOBI; }
        else if (a == "-scale" && i + 1 < argc) { std::string v = argv[++i]; scale = v == "max" ? ORC_SCALE_MAX : v == "none" ? ORC_SCALE_NONE : ORC_SCALE_DIAG; }
        else if (a == "--jacobi-ref") p.jacobi_mode = ORC_JACOBI_REF;
        else if (a == "--max-iters" && i + 1 < argc) p.max_iters = atoi(argv[++i]);
        else if (a == "--tol" && i + 1 < argc) p.tol = atof(argv[++i]);
        else if (a == "--restart" && i + 1 < argc) p.restart_len = atoi(argv[++i]);
        else if (a == "--dump-x" && i + 1 < argc) dump = argv[++i];
    }
    int n, nc; long nnz; int *I, *J; double *V;
    if (orc_read_mtx(argv[1], &n, &nc, &nnz, &I, &J, &V) != 0) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
    std::vector<double> rs(n), b(n), x0(n), val(nnz), xs(n), hist(p.max_iters + 2);
    std::vector<int> rp(n + 1), col(nnz);
    orc_scale_coo(n, nnz, I, J, V, scale, rs.data());
    orc_coo_to_crs(n, nnz, I, J, V, rp.data(), col.data(), val.data());
    // generate_vector(1.0) + scale_vector (main.cpp:270-276, utility_funcs.hpp:518-530)
    for (int i = 0; i < n; ++i) { b[i] = 1.0 / (rs[i] * rs[i]); x0[i] = 1.0 / (rs[i] * rs[i]); }
    orc_result r{};
    orc_solve(n, rp.data(), col.data(), val.data(), b.data(), x0.data(), &p, hist.data(), xs.data(), &r);
    for (int i = 0; i < r.residual_count; ++i)
        std::cout << "||A*x_" << i << " - b||_infty = " << std::setprecision(16) << hist[i] << std::endl;
    if (r.converged) std::cout << "\n" << name << " solver converged in: " << r.iters << " iterations." << std::endl;
    else std::cout << "\n" << name << " solver did not converge after " << p.max_iters << " iterations." << std::endl;
    std::cout << "With the stopping criteria \"tol * || b-A*x_0 ||_infty\" is: " << r.stopping << std::endl;
    std::cout << "The residual of the final iteration is: ||A*x_star - b||_infty = "
              << std::scientific << hist[r.residual_count] << ".\n";
    if (dump) { FILE *f = fopen(dump, "wb"); fwrite(xs.data(), 8, n, f); fclose(f); }
    orc_free(I); orc_free(J); orc_free(V);
    return 0;
}
