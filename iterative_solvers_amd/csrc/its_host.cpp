// its_host.cpp — host-only helpers of libits_hip.so (no device code, callable without a GPU):
// multicolour ordering for the Gauss-Seidel sweep and symmetric permutation of a CRS matrix.
//
// Why a permutation and not a different kernel: the reference's Gauss-Seidel is a natural-order
// forward substitution (kernels.hpp:451-473).  The library executes it level by level (rows of
// one dependency level in parallel, identical arithmetic per row).  On P A P^T with rows ordered
// colour by colour, the dependency levels ARE the colours, so the same level-scheduled sweep is
// the multicolour sweep — and the CPU reference run on the same permuted matrix gives bit-
// comparable iterates (SURVEY.md §7 "GS ordering").
#include "../../include/its_hip.h"

#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

extern "C" int its_multicolor_order(int n, const int *row_ptr, const int *col, int *perm_new2old, int *n_colors)
{
    if (n < 0 || !row_ptr || (n > 0 && !perm_new2old)) return ITS_EINVAL;
    const int64_t nnz = n ? row_ptr[n] : 0;
    if (nnz > 0 && !col) return ITS_EINVAL;
    // structure of A^T (rows that reference vertex j): needed because a_ij != 0 or a_ji != 0 both
    // forbid i and j in one colour
    std::vector<int64_t> tp((size_t)n + 1, 0);
    for (int64_t k = 0; k < nnz; ++k) {
        if (col[k] < 0 || col[k] >= n) return ITS_EINVAL;
        ++tp[(size_t)col[k] + 1];
    }
    for (int i = 0; i < n; ++i) tp[i + 1] += tp[i];
    std::vector<int> tcol((size_t)nnz);
    {
        std::vector<int64_t> fill(tp.begin(), tp.end() - 1);
        for (int i = 0; i < n; ++i)
            for (int k = row_ptr[i]; k < row_ptr[i + 1]; ++k) tcol[(size_t)fill[col[k]]++] = i;
    }
    std::vector<int> colour((size_t)n, -1), mark;     // mark[c] == i  <=>  colour c is taken around vertex i
    int nc = 0;
    for (int i = 0; i < n; ++i) {
        auto visit = [&](int j) {
            if (j != i && colour[j] >= 0) {
                if ((int)mark.size() <= colour[j]) mark.resize((size_t)colour[j] + 1, -1);
                mark[colour[j]] = i;
            }
        };
        for (int k = row_ptr[i]; k < row_ptr[i + 1]; ++k) visit(col[k]);
        for (int64_t k = tp[i]; k < tp[i + 1]; ++k) visit(tcol[(size_t)k]);
        int c = 0;
        while (c < (int)mark.size() && mark[c] == i) ++c;
        colour[i] = c;
        nc = std::max(nc, c + 1);
    }
    // stable counting sort by colour: ascending original index inside a colour
    std::vector<int> start((size_t)nc + 1, 0);
    for (int i = 0; i < n; ++i) ++start[(size_t)colour[i] + 1];
    for (int c = 0; c < nc; ++c) start[c + 1] += start[c];
    for (int i = 0; i < n; ++i) perm_new2old[start[colour[i]]++] = i;
    if (n_colors) *n_colors = nc;
    return ITS_OK;
}

extern "C" int its_permute_crs(int n, const int *row_ptr, const int *col, const double *val,
                               const int *perm_new2old, int *row_ptr_out, int *col_out, double *val_out)
{
    if (n < 0 || !row_ptr || !perm_new2old || !row_ptr_out) return ITS_EINVAL;
    std::vector<int> old2new((size_t)n, -1);
    for (int i = 0; i < n; ++i) {
        const int o = perm_new2old[i];
        if (o < 0 || o >= n || old2new[o] != -1) return ITS_EINVAL;      // not a permutation
        old2new[o] = i;
    }
    row_ptr_out[0] = 0;
    std::vector<std::pair<int, double>> row;
    for (int i = 0; i < n; ++i) {
        const int o = perm_new2old[i];
        row.clear();
        for (int k = row_ptr[o]; k < row_ptr[o + 1]; ++k) row.emplace_back(old2new[col[k]], val[k]);
        std::stable_sort(row.begin(), row.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
        int w = row_ptr_out[i];
        for (const auto &e : row) { col_out[w] = e.first; val_out[w] = e.second; ++w; }
        row_ptr_out[i + 1] = w;
    }
    return ITS_OK;
}
