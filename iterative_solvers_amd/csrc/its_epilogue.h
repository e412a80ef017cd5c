// its_epilogue.h — what every exact SpMV kernel shares: the per-entry chain step and the per-row epilogue
// (device code, included by its_kernels.hip and its_ell.hip).
#pragma once
#include "its_kernels.h"

namespace its {

// the sweep reads its right-hand side and the diagonal once per trip, for the last time: non-temporal (multicolour GS
// trip at 256^3 0.551 -> 0.546 ms)
constexpr bool SWEEP_NT = true;

// --------------------------------------------------------------------------------------
// What the epilogues ask of the per-row walk, shared by every exact kernel below.
//   chain_step   one entry of the row, in CRS order: `acc` is the epilogue's own chain (all entries; all but the
//                diagonal for the Jacobi forms, jacobi.hpp:76-85; entries right of the diagonal for EPI_GS_FUSED),
//                `acc2` the full chain of the fused forms (the recorded residual of the same x)
//   row_finish   the value to store and the row's contribution to the partial sums
// e0..e2 are the epilogue's per-row operands (epi_operands), loaded where the kernel's pipeline wants them.
// --------------------------------------------------------------------------------------
template <int EPI> struct EpiKind {
    static constexpr bool skip_diag = EPI == EPI_JACOBI_REF || EPI == EPI_JACOBI_FUSED;
    static constexpr bool upper_only = EPI == EPI_GS_FUSED;
    static constexpr bool fused = EPI == EPI_JACOBI_FUSED || EPI == EPI_GS_FUSED;
    // writes r^2 / max|r| partials.  EPI_JACOBI_SHIPPED too: its chain is the full row, i.e. the chain of the residual of
    // the vector it multiplies, so the pass that sweeps x also records ||b - A x|| when it is given part_sum
    static constexpr bool norm = EPI == EPI_RESIDUAL || fused || EPI == EPI_JACOBI_SHIPPED;
    static constexpr bool needs_col = skip_diag || upper_only;
};

template <int EPI>
__device__ __forceinline__ void chain_step(int c, int row, double v, double x, double &acc, double &acc2, double &diag)
{
    // An entry the chain leaves out enters it as fma(0, 0, acc) instead of a select on acc: acc starts at +0.0
    // and round-to-nearest never makes it -0.0, so acc + (+0.0) returns acc bit for bit, and the selects sit on
    // the operands, off the dependent chain (a select per entry on acc itself made the EPI_GS_FUSED walk of a
    // long row 1.4x slower).
    if (EpiKind<EPI>::fused) acc2 = fma(v, x, acc2);
    if (EpiKind<EPI>::skip_diag) {
        const bool dg = c == row;
        diag = dg ? v : diag;
        acc = fma(dg ? 0.0 : v, dg ? 0.0 : x, acc);
    } else if (EpiKind<EPI>::upper_only) {
        const bool up = c > row;
        acc = fma(up ? v : 0.0, up ? x : 0.0, acc);
    } else {
        acc = fma(v, x, acc);
    }
}

// where the full chain of a fused Gauss-Seidel pass starts (SpmvArgs::acc0) and where a sweep leaves its chain sum
// (SpmvArgs::acc_out); both compile away for the other epilogues
template <int EPI>
__device__ __forceinline__ double chain2_start(const SpmvArgs &a, int row)
{
    if (EPI == EPI_GS_FUSED && a.acc0 && !(a.y2 && row >= a.y2_lo && row < a.y2_hi)) return a.acc0[row];
    return 0.0;
}
template <int EPI>
__device__ __forceinline__ void sweep_note(const SpmvArgs &a, int row, double acc)
{
    if (EPI == EPI_SWEEP && a.acc_out) a.acc_out[row] = acc;
}

template <int EPI, bool DOT>
__device__ __forceinline__ void epi_operands(const SpmvArgs &a, int row, double &e0, double &e1, double &e2)
{
    if (EPI == EPI_STORE && DOT) e0 = a.dotv[row];
    if (EPI == EPI_RESIDUAL || EPI == EPI_JACOBI_REF || EPI == EPI_GS_RHS || EpiKind<EPI>::fused)
        e0 = a.b[row];
    if (EPI == EPI_JACOBI_SHIPPED) { e0 = a.D[row]; e1 = a.x[row]; e2 = a.b[row]; }
    if (EPI == EPI_SWEEP) { e0 = SWEEP_NT ? __builtin_nontemporal_load(a.b + row) : a.b[row]; e1 = SWEEP_NT ? __builtin_nontemporal_load(a.D + row) : a.D[row]; }
}

template <int EPI, bool DOT>
__device__ __forceinline__ double row_finish(double acc, double acc2, double diag, double e0, double e1, double e2,
                                             double &lane_sum, double &lane_max)
{
    double outv;
    if (EPI == EPI_STORE) {
        outv = acc;
        if (DOT) lane_sum = fma(acc, e0, lane_sum);
    } else if (EPI == EPI_RESIDUAL) {
        outv = e0 - acc;
        lane_sum = fma(outv, outv, lane_sum);
        lane_max = fmax(lane_max, fabs(outv));
    } else if (EPI == EPI_JACOBI_SHIPPED) {
        const double adj = fma(-e0, acc, e1);
        outv = (e2 - adj) / e0;
        const double r = e2 - acc;                  // b - A x of the swept vector, as EPI_RESIDUAL
        lane_sum = fma(r, r, lane_sum);
        lane_max = fmax(lane_max, fabs(r));
    } else if (EPI == EPI_JACOBI_REF || EPI == EPI_JACOBI_FUSED) {
        outv = (e0 - acc) / diag;
    } else if (EPI == EPI_SWEEP) {
        outv = (e0 - acc) / e1;
    } else {                                        // EPI_GS_RHS, EPI_GS_FUSED
        outv = fma(-1.0, acc, e0);
    }
    if (EpiKind<EPI>::fused) {
        const double r = e0 - acc2;
        lane_sum = fma(r, r, lane_sum);
        lane_max = fmax(lane_max, fabs(r));
    }
    return outv;
}

}  // namespace its
