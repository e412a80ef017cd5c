// utility_funcs.hpp — one-time host preprocessing with the reference's semantics
// (utility_funcs.hpp:157-174, 240-377, 518-530, 760-820; main.cpp:117-132).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include "structs.hpp"

// equilibration; `max` keeps the shipped behaviour (both passes fill the ROW vector, main.cpp:121-124)
inline void scale_matrix(COOMtxData &m, const std::string &scale_type, std::vector<double> &row_scale)
{
    std::vector<double> col_scale(m.n_cols, 1.0);
    row_scale.assign(m.n_cols, 1.0);
    if (scale_type == "max") {
        for (long k = 0; k < m.nnz; ++k) row_scale[m.I[k]] = std::max(row_scale[m.I[k]], std::fabs(m.values[k]));
        for (long k = 0; k < m.nnz; ++k) row_scale[m.J[k]] = std::max(row_scale[m.J[k]], std::fabs(m.values[k]));
    } else if (scale_type == "diag") {
        for (long k = 0; k < m.nnz; ++k)
            if (m.I[k] == m.J[k]) row_scale[m.I[k]] = col_scale[m.I[k]] = std::sqrt(std::fabs(m.values[k]));
    }
    for (long k = 0; k < m.nnz; ++k) m.values[k] = m.values[k] / row_scale[m.I[k]];
    for (long k = 0; k < m.nnz; ++k) m.values[k] = m.values[k] / col_scale[m.J[k]];
}

inline void convert_to_crs(const COOMtxData &m, CRSMtxData &c)
{
    c.n_rows = (int)m.n_rows; c.n_cols = (int)m.n_cols; c.nnz = (int)m.nnz;
    c.col.assign(m.J.begin(), m.J.end());
    c.val.assign(m.values.begin(), m.values.end());
    c.row_ptr.assign(c.n_rows + 1, 0);
    for (long k = 0; k < m.nnz; ++k) ++c.row_ptr[m.I[k] + 1];
    for (int i = 0; i < c.n_rows; ++i) c.row_ptr[i + 1] += c.row_ptr[i];
    if (c.row_ptr[c.n_rows] != c.nnz) { printf("ERROR: converting to CRS.\n"); exit(1); }
}

inline void extract_diag(const COOMtxData &m, std::vector<double> &D)
{
    D.assign(m.n_cols, 0.0);
    for (long k = 0; k < m.nnz; ++k) if (m.I[k] == m.J[k]) D[m.I[k]] = m.values[k];
}

inline void split_L_U(const COOMtxData &m, COOMtxData &L, COOMtxData &U)
{
    L = COOMtxData(); U = COOMtxData();
    L.n_rows = U.n_rows = m.n_rows; L.n_cols = U.n_cols = m.n_cols;
    for (long k = 0; k < m.nnz; ++k) {
        COOMtxData *t = m.J[k] < m.I[k] ? &L : (m.J[k] > m.I[k] ? &U : nullptr);
        if (!t) continue;
        t->I.push_back(m.I[k]); t->J.push_back(m.J[k]); t->values.push_back(m.values[k]); ++t->nnz;
    }
}

inline void generate_vector(std::vector<double> &v, int size, double initial_val) { v.assign(size, initial_val); }

inline void scale_vector(std::vector<double> &v, const std::vector<double> &s)
{
    for (size_t i = 0; i < v.size(); ++i) v[i] = v[i] / (s[i] * s[i]);
}
