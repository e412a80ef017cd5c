// ref_xstar_driver — validation driver (TEST INFRASTRUCTURE ONLY, built into oracle/_ref/ by `make ref`).
// The reference prints its residual history but never its solution.  This driver includes the reference's own
// translation unit from where it lies (-I$(REF)) with its `main` renamed, runs the same sequence run_solver<double>
// does (main.cpp:331-520: assign_cli_inputs, read_mtx, preprocessing, solve, postprocessing — all the reference's
// own functions, called unchanged) and writes Solver::x_star (solver.hpp:23, filled by save_x_star,
// solver_harness.hpp:120) as raw little-endian doubles to the file named by ITS_XSTAR_OUT before the reference
// frees it.  stdout is the reference's own.  Nothing of the reference is copied: only its API is called.
#define main reference_main_not_used
#include "main.cpp"
#undef main
#include <cstdio>
#include <cstdlib>

int main(int argc, char *argv[])
{
    bogus_init_pin();
    Flags flags{false, true, true, false, false, false, false};                       // main.cpp:550-558
    LoopParams loop_params{0, 0, 1, MAX_ITERS, 0.0, TOL, 1.0, 1.0, GMRES_RESTART_LEN}; // main.cpp:561-571
    Timers *timers = new Timers;
    timeval *t0 = new timeval, *t1 = new timeval;
    Stopwatch *total_wtime = new Stopwatch(t0, t1);
    std::string matrix_file_name;
    SparseMtxFormat<double> *sparse_mat = new SparseMtxFormat<double>;
    COOMtxData<double> *coo_mat = new COOMtxData<double>;
    argType<double> *args = new argType<double>;
    args->timers = timers;
    args->timers->total_wtime = total_wtime;
    args->timers->total_wtime->start_stopwatch();
    assign_cli_inputs<double>(args, argc, argv, &matrix_file_name);
    Solver<double> *solver = new Solver<double>(args->solver_type, args->preconditioner_type);
    solver->gmres_args = new gmresArgs<double>;
    solver->cg_args = new cgArgs<double>;
    std::cout << "Reading Matrix" << std::endl;
    read_mtx(matrix_file_name, coo_mat);
    args->loop_params = &loop_params;
    args->flags = &flags;
    args->matrix_file_name = &matrix_file_name;
    args->sparse_mat = sparse_mat;
    solver->gmres_args->restart_length = loop_params.gmres_restart_len;
    args->coo_mat = coo_mat;
    args->sparse_mat->crs_mat = new CRSMtxData<double>;
    args->sparse_mat->crs_L = new CRSMtxData<double>;
    args->sparse_mat->crs_U = new CRSMtxData<double>;
    preprocessing<double>(args, solver);
    solve<double>(args, solver);
    if (const char *out = getenv("ITS_XSTAR_OUT")) {
        FILE *f = fopen(out, "wb");
        if (!f || fwrite(solver->x_star, sizeof(double), (size_t)args->vec_size, f) != (size_t)args->vec_size) { perror("ITS_XSTAR_OUT"); return 2; }
        fclose(f);
    }
    postprocessing<double>(args, solver);
    return 0;
}
