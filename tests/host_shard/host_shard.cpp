// TEST INFRASTRUCTURE (never part of libcu2rec_amd.so): the product's user-sharded driver -- the template of
// cu2rec_amd/csrc/shard_driver.hpp: exchange cadence, wire exchange and merge weights, global loss reduction, train() over all
// ranks -- instantiated over HOST memory, with the CPU oracle (oracle/cu2rec_oracle.c) as the engine that runs the SGD and loss
// passes.  tests/test_parallel_cpu.py drives it with two gloo ranks on a box without a GPU, so the world-2 CPU tests run the
// product's C++ exchange code itself.  Built by the test: g++ -shared ... -> build/test/libcu2rec_shard_host.so.
//
// Users are addressed by their GLOBAL id on every rank here: a rank's model holds full-size P / user_bias arrays and a CSR whose
// rows outside its user range are empty (the way tests/exchange_reference.py runs the oracle), so user_offset is 0.
#include <chrono>
#include <cstring>
#include <memory>
#include <vector>

#include "../../cu2rec_amd/csrc/shard_driver.hpp"

extern "C" {
#include "../../oracle/cu2rec_oracle.h"
}

namespace cu2rec {

thread_local std::string g_last_error;
void set_last_error(const std::string &msg) { g_last_error = msg; }

struct HostModel {
    int rows, cols, f;
    float *P, *Q, *user_bias, *item_bias;  // caller-owned, dense (ldq == f)
    float global_bias;
};

struct HostCsr {
    const int *indptr, *indices;
    const float *data;
    int rows, nnz, max_item, users_with_ratings;
};

struct HostComm {
    int rank = 0, nranks = 1;
    cu2rec_allreduce_fn fn = nullptr;
    void *ctx = nullptr;
    bool collective() const { return false; }
    void allreduce(void *buf, size_t count, bool is_double, void *stream) const {
        if (count == 0 || nranks == 1) return;
        require(fn != nullptr, "host_shard: no all-reduce behind a multi-rank job");
        if (fn(ctx, buf, count, is_double ? 1 : 0, stream) != 0) fail(CU2REC_EHIP, "host_shard: the caller's all-reduce failed");
    }
    void wait(void *) {}
};

template <class T>
struct HostBuffer {
    T *ptr = nullptr;
    size_t count = 0;
    std::unique_ptr<T[]> store;
    void allocate(size_t n) {
        store.reset(n ? new T[n]() : nullptr);
        ptr = store.get();
        count = n;
    }
    void upload(const T *host, size_t n) { std::memcpy(ptr, host, n * sizeof(T)); }
    void download(T *host, size_t n) const { std::memcpy(host, ptr, n * sizeof(T)); }
};

struct HostClock {
    std::chrono::steady_clock::time_point t0, t1;
    void start(void *) { t0 = std::chrono::steady_clock::now(); }
    void stop(void *) { t1 = std::chrono::steady_clock::now(); }
    float elapsed_ms() { return std::chrono::duration<float, std::milli>(t1 - t0).count(); }
    bool done() { return true; }
    void drain() {}
};

struct HostBackend {
    using Model = HostModel;
    using Csr = HostCsr;
    using Comm = HostComm;
    using Stream = void *;
    template <class T>
    using Buffer = HostBuffer<T>;
    using Clock = HostClock;
    static int dot_order;
    static void require_ready() {}
    static int rows(const Model &m) { return m.rows; }
    static int cols(const Model &m) { return m.cols; }
    static int n_factors(const Model &m) { return m.f; }
    static int ldq(const Model &m) { return m.f; }
    static float *Q(Model &m) { return m.Q; }
    static float *item_bias(Model &m) { return m.item_bias; }
    static int csr_rows(const Csr &c) { return c.rows; }
    static int csr_nnz(const Csr &c) { return c.nnz; }
    static int csr_max_item(const Csr &c) { return c.max_item; }
    static int csr_users_with_ratings(const Csr &c) { return c.users_with_ratings; }
    static void csr_structure(const Csr &c, std::vector<int> &indptr, std::vector<int> &indices) {
        indptr.assign(c.indptr, c.indptr + c.rows + 1);
        indices.assign(c.indices, c.indices + c.indptr[c.rows]);  // (row ranges are positions in the FULL arrays here)
        if (indices.empty()) indices.push_back(0);
    }
    static void copy(float *dst, const float *src, size_t n) { std::memcpy(dst, src, n * sizeof(float)); }
    static void to_backend(void *dst, const void *src, size_t bytes, Stream) { std::memcpy(dst, src, bytes); }
    static void to_host(void *dst, const void *src, size_t bytes, Stream) { std::memcpy(dst, src, bytes); }
    // the arithmetic of items_wire_pack_kernel / items_wire_apply_kernel (cu2rec_amd/csrc/kernels.hip), element by element
    static void wire_pack(const float *Q, const float *ib, const float *Q_base, const float *ib_base, const float *weight, int n_cols, int f,
                          int ldq, float *wire, Stream) {
        const size_t nq = static_cast<size_t>(n_cols) * f;
        for (size_t i = 0; i < nq + n_cols; ++i) {
            if (i < nq) {
                const size_t y = i / f, c = i - y * f, at = y * ldq + c;
                const float d = Q[at] - Q_base[at];
                wire[i] = weight ? weight[y] * d : d;
            } else {
                const size_t y = i - nq;
                const float d = ib[y] - ib_base[y];
                wire[i] = weight ? weight[y] * d : d;
            }
        }
    }
    static void wire_apply(float *Q, float *ib, float *Q_base, float *ib_base, int n_cols, int f, int ldq, const float *wire, float scale,
                           Stream) {
        const size_t nq = static_cast<size_t>(n_cols) * f;
        for (size_t i = 0; i < nq + n_cols; ++i) {
            if (i < nq) {
                const size_t y = i / f, c = i - y * f, at = y * ldq + c;
                const float v = Q_base[at] + scale * wire[i];
                Q[at] = v;
                Q_base[at] = v;
            } else {
                const size_t y = i - nq;
                const float v = ib_base[y] + scale * wire[i];
                ib[y] = v;
                ib_base[y] = v;
            }
        }
    }
    static void sgd(Model &m, const Csr &train, const cu2rec_hyper &h, uint64_t seed, uint64_t iter0, int n, int, int update_items, Stream, int) {
        const orc_hyper oh{h.learning_rate, h.P_reg, h.Q_reg, h.user_bias_reg, h.item_bias_reg};
        orc_sgd_iterations(train.indptr, train.indices, train.data, train.rows, m.P, m.Q, m.user_bias, m.item_bias, m.global_bias, &oh, m.f, seed,
                           iter0, n, dot_order, update_items);
    }
    static void loss(Model &m, const Csr &ratings, double *sum_abs, double *sum_sq, Stream) {
        float mae = 0.f, rmse = 0.f;
        if (ratings.nnz == 0) {
            *sum_abs = *sum_sq = 0.0;
            return;
        }
        orc_loss(ratings.indptr, ratings.indices, ratings.data, ratings.rows, ratings.nnz, m.P, m.Q, m.user_bias, m.item_bias, m.global_bias, m.f,
                 dot_order, ORC_ACC_F64, nullptr, sum_abs, sum_sq, &mae, &rmse);
    }
};
int HostBackend::dot_order = ORC_DOT_TREE16;

}  // namespace cu2rec

using namespace cu2rec;

struct host_job {
    HostComm comm;
    HostModel model;
    HostCsr train;
    std::unique_ptr<ShardDriver<HostBackend>> drv;
};

static HostCsr make_csr(const int *indptr, const int *indices, const float *data, int rows, int nnz) {
    HostCsr c{indptr, indices, data, rows, nnz, -1, 0};  // nnz: the rank's own ratings, [indptr[0], indptr[rows]) of the full arrays
    for (int k = indptr[0]; k < indptr[rows]; ++k) c.max_item = std::max(c.max_item, indices[k]);
    for (int u = 0; u < rows; ++u) c.users_with_ratings += indptr[u + 1] > indptr[u];
    return c;
}

extern "C" {

const char *host_shard_last_error(void) { return g_last_error.c_str(); }

// model arrays and CSR arrays stay the caller's (numpy); options = {sync_every, merge}
int host_shard_create(cu2rec_allreduce_fn fn, void *ctx, int rank, int nranks, int rows, int cols, int f, float *P, float *Q, float *user_bias,
                      float *item_bias, float global_bias, const int *indptr, const int *indices, const float *data, int nnz, int sync_every,
                      int merge, host_job **out) {
    return guarded([&] {
        require(out != nullptr, "out is null");
        std::unique_ptr<host_job> j(new host_job);
        j->comm.rank = rank;
        j->comm.nranks = nranks;
        j->comm.fn = fn;
        j->comm.ctx = ctx;
        j->model = HostModel{rows, cols, f, P, Q, user_bias, item_bias, global_bias};
        j->train = make_csr(indptr, indices, data, rows, nnz);
        j->drv.reset(new ShardDriver<HostBackend>(j->comm, j->model, j->train, 0, cu2rec_shard_options{sync_every, merge}));
        *out = j.release();
    });
}

void host_shard_destroy(host_job *j) { delete j; }

int host_shard_run(host_job *j, const cu2rec_hyper *h, uint64_t seed, uint64_t iter0, int n_iters, int update_items) {
    return guarded([&] { j->drv->run(*h, seed, iter0, n_iters, CU2REC_SGD_ORDERED, update_items, nullptr); });
}

int host_shard_exchange(host_job *j) {
    return guarded([&] {
        if (j->drv->since_sync > 0) j->drv->exchange(nullptr);
    });
}

int host_shard_loss(host_job *j, const int *indptr, const int *indices, const float *data, int rows, int nnz, double *sum_abs, double *sum_sq,
                    double *n_total, float *mae, float *rmse) {
    return guarded([&] {
        const HostCsr c = make_csr(indptr, indices, data, rows, nnz);
        j->drv->loss(c, sum_abs, sum_sq, n_total, mae, rmse, nullptr);
    });
}

int host_shard_info(const host_job *j, int *sync_every, int *exchanges, double *users_total, double *nnz_total) {
    return guarded([&] {
        *sync_every = j->drv->sync_every;
        *exchanges = j->drv->exchanges;
        *users_total = j->drv->users_total;
        *nnz_total = j->drv->nnz_total;
    });
}

int host_shard_exchange_stats(host_job *j, int *timed, double *seconds, double *max_seconds) {
    return guarded([&] { j->drv->exchange_stats(timed, seconds, max_seconds); });
}

int host_shard_train(host_job *j, const int *te_indptr, const int *te_indices, const float *te_data, int te_rows, int te_nnz, cu2rec_config *cfg,
                     int verbose, float *losses, cu2rec_train_stats *stats) {
    return guarded([&] {
        const HostCsr test = make_csr(te_indptr, te_indices, te_data, te_rows, te_nnz);
        shard_train<HostBackend>(*j->drv, test, *cfg, CU2REC_SGD_ORDERED, verbose != 0, losses, stats);
    });
}

}  // extern "C"
