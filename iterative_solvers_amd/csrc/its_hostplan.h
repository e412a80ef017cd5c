// its_hostplan.h — host-side set-up passes of libits_hip.so that run OpenMP-parallel (its_host.cpp, compiled by
// g++ -fopenmp): everything an upload has to derive from the caller's CRS arrays on the host (keep
// construction off the serial host path).  Results are independent of the thread count.
#pragma once
#include <stdint.h>

namespace its_host {

// monotone row_ptr and columns inside [0, n_cols): 0 ok, 1 row_ptr not monotone, 2 column out of range
int validate_crs(int n_rows, int n_cols, int64_t nnz, const int *rp, const int *col);
// column-offset dictionary: cidx[k] = index of col[k] - row(k) in dict (first-occurrence order), n_off = entries used;
// returns false (n_off = 0) when there are more than 256 distinct offsets
bool offset_dict(int n_rows, const int *rp, const int *col, unsigned char *cidx, int *dict256, int *n_off);
// the (at most max_out <= 16) most frequent values of col - row among the entries of <= 4096 evenly spaced sample rows,
// each carried by at least 2 % of the sampled entries; returns how many (0: no offset stands out).  Deterministic.
int frequent_offsets(int n_rows, const int *rp, const int *col, int *out, int max_out);
// value dictionary over the fp64 bit patterns, same rules
bool value_dict(int64_t nnz, const double *val, unsigned char *vidx, double *dict256, int *n_vals);
// longest row of every block of a stream plan (blk[nb + 1] row cuts)
void block_longest(int nb, const int *blk, const int *rp, int *longest);
// longest row of the matrix; count, column sum and value hash of the entries right of the diagonal
void row_stats(int n_rows, const int *rp, const int *col, const double *val, bool upper, int *max_row_len,
               int64_t *n_upper, int64_t *upper_col_sum, uint64_t *upper_val_xor);

// largest column below `limit` of every row (-1: none): what the sweep schedule and the strictly-lower check need
void row_maxcol(int n_rows, const int *rp, const int *col, int limit, int *maxcol);

// split by magnitude (ITS_FMT_MIXED): |val| >= thr -> (rp_d, col_d, val_d), else -> (rp_s, col_s, val_s as float);
// row pointers n_rows + 1, the other outputs sized nnz; entry order inside a row kept
int64_t sell_order(int n, int n_pad, const int *rp, int sigma, int C, int *order, int *clen);
void split_by_magnitude(int n_rows, const int *rp, const int *col, const double *val, double thr,
                        int *rp_d, int *col_d, double *val_d, int *rp_s, int *col_s, float *val_s);

// split by storage tier (its_ap.h: mode, AP_THRESHOLD_1, AP_THRESHOLD_2): fp64 -> (rp_d, col_d, val_d), fp32 -> (rp_s, ...),
// fp16 -> (rp_h, col_h, val_h as binary16 bits of the float-rounded value); same sizing and order rules as above
void split_by_tiers(int n_rows, const int *rp, const int *col, const double *val, int mode, double t1, double t2,
                    int *rp_d, int *col_d, double *val_d, int *rp_s, int *col_s, float *val_s,
                    int *rp_h, int *col_h, uint16_t *val_h);

// rows of A from the diagonal on (Gauss-Seidel, option "gs_prefix"): row i = (i, diag[i]) followed by row i of the strictly
// upper part; rp sized n + 1, col / val sized nnz_U + n
void assemble_diag_upper(int n, const int *u_rp, const int *u_col, const double *u_val, const double *diag,
                         int *rp, int *col, double *val);

}  // namespace its_host
