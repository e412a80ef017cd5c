// mtx_fast.hpp — a faster way to the SAME row-sorted COO that read_mtx (io_funcs.hpp; reference
// io_funcs.hpp:125-289 + mmio.h:195-226) produces.  SURVEY.md §8 f2: for 10^8-nonzero inputs the
// fscanf loop and the comparison sort take minutes while the solve takes a fraction of a second.
//   * the file is read in one piece and its body parsed by all host threads (strtol/strtod: the same
//     correctly rounded conversions fscanf("%d %d %lg") performs), chunks cut at line ends and glued
//     back in file order;
//   * the stable sort by row is a counting sort (O(nnz));
//   * ITS_MTX_CACHE=1 keeps the result next to the matrix as <file>.itsbin (raw I, J, values with the
//     source's size and mtime in the header) and later runs load that instead.
// Anything unusual (entries spanning lines, stray tokens, a short body) makes the caller fall back
// to the fscanf reader, which then reports the reference's own errors.
#pragma once
#include <sys/stat.h>
#include <algorithm>
#include <charconv>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "structs.hpp"

namespace mtx_fast {

struct CacheHeader {
    char magic[8];                 // "ITSBIN1\0"
    int64_t src_size, src_mtime, n_rows, n_cols, nnz;
};

inline bool stat_file(const std::string &f, int64_t &size, int64_t &mtime)
{
    struct stat st;
    if (stat(f.c_str(), &st) != 0) return false;
    size = (int64_t)st.st_size; mtime = (int64_t)st.st_mtime;
    return true;
}

inline bool load_cache(const std::string &file, COOMtxData *coo)
{
    int64_t size, mtime;
    if (!stat_file(file, size, mtime)) return false;
    FILE *f = fopen((file + ".itsbin").c_str(), "rb");
    if (!f) return false;
    CacheHeader h;
    bool ok = fread(&h, sizeof h, 1, f) == 1 && !memcmp(h.magic, "ITSBIN1", 8) && h.src_size == size &&
              h.src_mtime == mtime && h.nnz >= 0 && h.n_rows >= 0;
    if (ok) {
        coo->n_rows = (long)h.n_rows; coo->n_cols = (long)h.n_cols; coo->nnz = (long)h.nnz;
        coo->I.resize(h.nnz); coo->J.resize(h.nnz); coo->values.resize(h.nnz);
        ok = fread(coo->I.data(), sizeof(int), h.nnz, f) == (size_t)h.nnz &&
             fread(coo->J.data(), sizeof(int), h.nnz, f) == (size_t)h.nnz &&
             fread(coo->values.data(), sizeof(double), h.nnz, f) == (size_t)h.nnz;
    }
    fclose(f);
    return ok;
}

inline void store_cache(const std::string &file, const COOMtxData &coo)
{
    CacheHeader h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, "ITSBIN1", 8);
    if (!stat_file(file, h.src_size, h.src_mtime)) return;
    h.n_rows = coo.n_rows; h.n_cols = coo.n_cols; h.nnz = coo.nnz;
    const std::string tmp = file + ".itsbin.tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return;                                         // read-only directory: no cache, no error
    const size_t n = (size_t)coo.nnz;
    const bool ok = fwrite(&h, sizeof h, 1, f) == 1 && fwrite(coo.I.data(), sizeof(int), n, f) == n &&
                    fwrite(coo.J.data(), sizeof(int), n, f) == n && fwrite(coo.values.data(), sizeof(double), n, f) == n;
    fclose(f);
    if (ok) rename(tmp.c_str(), (file + ".itsbin").c_str()); else remove(tmp.c_str());
}

// decimal integer with optional sign at p (no leading blanks); advances p
inline bool parse_int(const char *&p, const char *e, long &out)
{
    bool neg = false;
    if (p < e && (*p == '-' || *p == '+')) { neg = *p == '-'; ++p; }
    if (p >= e || *p < '0' || *p > '9') return false;
    long v = 0;
    int digits = 0;
    while (p < e && *p >= '0' && *p <= '9') { v = v * 10 + (*p - '0'); ++p; if (++digits > 10) return false; }
    out = neg ? -v : v;
    return true;
}

// Parses the coordinate body [p, e) (one entry per line).  Returns false on anything unexpected.
inline bool parse_chunk(const char *p, const char *e, bool pattern, std::vector<int> &ri, std::vector<int> &ci,
                        std::vector<double> &vv)
{
    while (p < e) {
        while (p < e && (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n')) ++p;
        if (p >= e) break;
        long r, c;
        if (!parse_int(p, e, r)) return false;
        while (p < e && (*p == ' ' || *p == '\t')) ++p;
        if (!parse_int(p, e, c)) return false;
        double v = 0.01;                                    // mmio.h:195-203: pattern entries
        if (!pattern) {
            while (p < e && (*p == ' ' || *p == '\t')) ++p;
            const char *s0 = (p < e && *p == '+') ? p + 1 : p;
            const char *tok = s0;
            while (tok < e && *tok != ' ' && *tok != '\t' && *tok != '\r' && *tok != '\n') ++tok;
            // correctly rounded like strtod / fscanf("%lg"); anything from_chars does not take whole
            // (hex floats, nan(...), locale forms) goes to strtod itself
            const auto res = std::from_chars(s0, tok, v, std::chars_format::general);
            if (res.ec != std::errc() || res.ptr != tok) {
                char *q;
                v = strtod(p, &q);
                if (q == p) return false;
                p = q;
            } else {
                p = tok;
            }
        }
        while (p < e && (*p == ' ' || *p == '\t' || *p == '\r')) ++p;
        if (p < e && *p != '\n') return false;              // more tokens on the line: not ours to guess
        ri.push_back((int)r - 1); ci.push_back((int)c - 1); vv.push_back(v);
    }
    return true;
}

// true: *coo holds what read_mtx would produce.  false: caller must use read_mtx (which also owns all
// error messages).
inline bool read(const std::string &file, COOMtxData *coo)
{
    const bool use_cache = getenv("ITS_MTX_CACHE") && atoi(getenv("ITS_MTX_CACHE")) != 0;
    const bool verbose = getenv("ITS_MTX_VERBOSE") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now();
    auto lap = [&](const char *what) { if (verbose) { const double t1 = now(); fprintf(stderr, "mtx_fast: %-12s %.3f s\n", what, t1 - t0); t0 = t1; } };
    if (use_cache && load_cache(file, coo)) { lap("cache load"); return true; }
    FILE *f = fopen(file.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (size <= 0) { fclose(f); return false; }
    std::vector<char> buf((size_t)size + 1);
    const bool got = fread(buf.data(), 1, (size_t)size, f) == (size_t)size;
    fclose(f);
    if (!got) return false;
    buf[size] = '\0';                                       // strtol/strtod never run off the end
    lap("file read");
    const char *p = buf.data(), *end = buf.data() + size;
    // banner
    char banner[64], obj[64], fmt[64], field[64], symm[64];
    if (sscanf(p, "%63s %63s %63s %63s %63s", banner, obj, fmt, field, symm) != 5 || strcmp(banner, "%%MatrixMarket") != 0) return false;
    auto lower = [](char *s) { for (; *s; ++s) *s = (char)tolower(*s); };
    lower(obj); lower(fmt); lower(field); lower(symm);
    const bool pattern = !strcmp(field, "pattern"), symmetric = !strcmp(symm, "symmetric");
    if (strcmp(obj, "matrix") || strcmp(fmt, "coordinate") || !(pattern || !strcmp(field, "real") || !strcmp(field, "integer")) ||
        !(symmetric || !strcmp(symm, "general")))
        return false;
    auto next_line = [&](const char *s) { while (s < end && *s != '\n') ++s; return s < end ? s + 1 : end; };
    p = next_line(p);
    while (p < end && *p == '%') p = next_line(p);
    int M = 0, N = 0, nz = 0;
    if (p >= end || sscanf(p, "%d %d %d", &M, &N, &nz) != 3 || M != N || nz < 0) return false;
    p = next_line(p);
    // body: cut at line ends, parse in parallel, glue in file order
    int T = (int)std::thread::hardware_concurrency();
    if (const char *e = getenv("ITS_MTX_THREADS")) T = atoi(e);
    T = std::max(1, std::min(T, 64));
    if (end - p < (1 << 20)) T = 1;
    std::vector<const char *> cut(T + 1);
    cut[0] = p; cut[T] = end;
    for (int k = 1; k < T; ++k) {
        const char *c = p + (end - p) / T * k;
        cut[k] = std::max(cut[k - 1], next_line(c));
    }
    std::vector<std::vector<int>> ri(T), ci(T);
    std::vector<std::vector<double>> vv(T);
    std::vector<char> okv(T, 0);
    {
        std::vector<std::thread> th;
        for (int k = 0; k < T; ++k)
            th.emplace_back([&, k] {
                // thread-local vectors: the shared vector headers sit in one cache line and every
                // push_back writes its header
                std::vector<int> r_, c_; std::vector<double> v_;
                const size_t guess = (size_t)(cut[k + 1] - cut[k]) / 12 + 16;
                r_.reserve(guess); c_.reserve(guess); v_.reserve(guess);
                const bool ok = parse_chunk(cut[k], cut[k + 1], pattern, r_, c_, v_);
                ri[k].swap(r_); ci[k].swap(c_); vv[k].swap(v_);
                okv[k] = ok;
            });
        for (auto &t : th) t.join();
    }
    lap("parse");
    int64_t total = 0;
    for (int k = 0; k < T; ++k) { if (!okv[k]) return false; total += (int64_t)ri[k].size(); }
    if (total != nz) return false;                          // short or long body: let read_mtx complain
    // rows, with the mirrored entry right behind its original (read_mtx), then a stable counting sort by row
    std::vector<int64_t> cnt((size_t)M + 1, 0);
    for (int k = 0; k < T; ++k)
        for (size_t i = 0; i < ri[k].size(); ++i) {
            const int r = ri[k][i], c = ci[k][i];
            if (r < 0 || r >= M || c < 0 || c >= N) return false;
            ++cnt[(size_t)r + 1];
            if (symmetric && r != c) ++cnt[(size_t)c + 1];
        }
    for (int i = 0; i < M; ++i) cnt[(size_t)i + 1] += cnt[i];
    const int64_t nnz = cnt[M];
    if (nnz > INT32_MAX) return false;
    coo->n_rows = M; coo->n_cols = N; coo->nnz = (long)nnz;
    coo->I.resize(nnz); coo->J.resize(nnz); coo->values.resize(nnz);
    for (int k = 0; k < T; ++k)
        for (size_t i = 0; i < ri[k].size(); ++i) {
            const int r = ri[k][i], c = ci[k][i];
            const double v = vv[k][i];
            int64_t at = cnt[r]++;
            coo->I[at] = r; coo->J[at] = c; coo->values[at] = v;
            if (symmetric && r != c) { at = cnt[c]++; coo->I[at] = c; coo->J[at] = r; coo->values[at] = v; }
        }
    lap("sort by row");
    if (use_cache) { store_cache(file, *coo); lap("cache store"); }
    return true;
}

// order-sensitive checksum of a COO (tests: fast reader / cache / fscanf reader must agree)
inline uint64_t checksum(const COOMtxData &coo)
{
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void *d, size_t n) { const unsigned char *b = (const unsigned char *)d; for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } };
    const int64_t hdr[3] = {coo.n_rows, coo.n_cols, coo.nnz};
    mix(hdr, sizeof hdr);
    mix(coo.I.data(), coo.I.size() * sizeof(int));
    mix(coo.J.data(), coo.J.size() * sizeof(int));
    mix(coo.values.data(), coo.values.size() * sizeof(double));
    return h;
}

}  // namespace mtx_fast
