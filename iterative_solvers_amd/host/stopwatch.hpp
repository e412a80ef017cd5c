// stopwatch.hpp — accumulating wall-clock timers with the reference's names (stopwatch.hpp:17-64).
// The device work behind Solver::iterate is synchronous at the C ABI (the residual scalar is on
// the host when it returns), so host timers around it measure real GPU time.
#pragma once
#include <chrono>

class Stopwatch {
    using clk = std::chrono::steady_clock;
    clk::time_point t0_;
    double total_ = 0.0;
public:
    void start_stopwatch() { t0_ = clk::now(); }
    void end_stopwatch() { total_ += std::chrono::duration<double>(clk::now() - t0_).count(); }
    double get_wtime() const { return total_; }
};

struct Timers {
    Stopwatch total_wtime, preprocessing_wtime, solver_harness_wtime, solver_wtime;
};
