// C++ convenience layer over the C ABI for host programs (bin/mf): RAII handles and status ->
// exception, i.e. the error behaviour of the reference's CHECK_CUDA (util.h:27-34).
#pragma once

#include <sys/stat.h>

#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/cu2rec_amd.h"

namespace cu2rec {

inline void check(int status) {
    if (status != CU2REC_OK) throw std::runtime_error(cu2rec_last_error());
}

struct HostCsr {  // createSparseMatrix's host arrays (util.cu:152-166)
    std::vector<int> indptr, indices;
    std::vector<float> data;
    int rows = 0, cols = 0, nnz = 0;
    float global_bias = 0.f;
};

// readCSV + createSparseMatrix, mf.cu:43-44.  With CU2REC_RATINGS_CACHE=1 in the environment a binary copy
// "<path>.cu2rec" is written after the first parse and loaded instead of the text while it is newer than it.
inline HostCsr load_ratings(const std::string &path) {
    cu2rec_ratings *r = nullptr;
    const char *want_cache = std::getenv("CU2REC_RATINGS_CACHE");
    const std::string cache = path + ".cu2rec";
    bool from_cache = false;
    if (want_cache && *want_cache == '1') {
        struct stat text_st, cache_st;
        if (stat(path.c_str(), &text_st) == 0 && stat(cache.c_str(), &cache_st) == 0 && cache_st.st_mtime >= text_st.st_mtime)
            from_cache = cu2rec_ratings_load_binary(cache.c_str(), &r) == CU2REC_OK;
    }
    if (!from_cache) {
        check(cu2rec_ratings_read_csv(path.c_str(), &r));
        if (want_cache && *want_cache == '1') (void)cu2rec_ratings_save_binary(r, cache.c_str());  // best effort
    }
    HostCsr m;
    try {
        check(cu2rec_ratings_info(r, &m.nnz, &m.rows, &m.cols, &m.global_bias));
        m.indptr.resize(static_cast<size_t>(m.rows) + 1);
        m.indices.resize(m.nnz);
        m.data.resize(m.nnz);
        check(cu2rec_csr_build(r, m.rows, m.indptr.data(), m.indices.data(), m.data.data()));
    } catch (...) {
        cu2rec_ratings_free(r);
        throw;
    }
    cu2rec_ratings_free(r);
    return m;
}

struct CsrHandle {
    cu2rec_csr *h = nullptr;
    explicit CsrHandle(const HostCsr &m) {
        check(cu2rec_csr_create(m.rows, m.cols, m.nnz, m.indptr.data(), m.indices.data(), m.data.data(), &h));
    }
    CsrHandle(const CsrHandle &) = delete;
    CsrHandle &operator=(const CsrHandle &) = delete;
    ~CsrHandle() { cu2rec_csr_destroy(h); }
};

struct ModelHandle {
    cu2rec_model *h = nullptr;
    ModelHandle(int rows, int cols, int f, float global_bias, const float *P = nullptr, const float *Q = nullptr,
                const float *ub = nullptr, const float *ib = nullptr) {
        check(cu2rec_model_create(rows, cols, f, P, Q, ub, ib, global_bias, &h));
    }
    ModelHandle(const ModelHandle &) = delete;
    ModelHandle &operator=(const ModelHandle &) = delete;
    ~ModelHandle() { cu2rec_model_destroy(h); }
};

}  // namespace cu2rec
