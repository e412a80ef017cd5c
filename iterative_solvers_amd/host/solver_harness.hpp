// solver_harness.hpp — the do..while loop and history bookkeeping of the reference's solve_cpu
// (solver_harness.hpp:14-150, utility_funcs.hpp:488-502), driving the device Solver.
#pragma once
#include <cmath>
#include <iostream>
#include "solver.hpp"

template <typename VT>
void record_residual_norm(argType *args, VT residual_norm)
{
    LoopParams &lp = args->loop_params;
    args->normed_residuals[lp.residual_count] = residual_norm;
    if (args->flags.convergence_flag) {}
    else if (lp.iter_count == lp.max_iters) {}
    else if (residual_norm < lp.stopping_criteria) {}
    else ++lp.residual_count;
}

template <typename VT>
void solve(argType *args, Solver<VT> *solver)
{
    std::cout << "Entering Solver Harness" << std::endl;
    LoopParams &lp = args->loop_params;
    VT residual_norm = 0;
    args->timers.solver_harness_wtime.start_stopwatch();
    do {
        args->timers.solver_wtime.start_stopwatch();
        solver->iterate(lp.iter_count, &residual_norm);
        args->timers.solver_wtime.end_stopwatch();
        if (lp.iter_count % lp.residual_check_len == 0) record_residual_norm<VT>(args, residual_norm);
        solver->exchange_arrays(args->vec_size);
        if (args->solver_type == "gmres" && residual_norm > lp.stopping_criteria && lp.iter_count < lp.max_iters &&
            (lp.iter_count + 1) % solver->gmres_restart_length == 0)
            solver->restart_gmres(lp.iter_count);
        ++lp.iter_count;
    } while (residual_norm > lp.stopping_criteria && lp.iter_count < lp.max_iters && !std::isinf((double)residual_norm));
    args->flags.convergence_flag = (residual_norm <= lp.stopping_criteria) && !std::isinf((double)residual_norm);
    solver->save_x_star(lp.iter_count);
    VT fin = residual_norm;
    solver->final_residual(&fin);
    record_residual_norm<VT>(args, fin);
    args->timers.solver_harness_wtime.end_stopwatch();
}
