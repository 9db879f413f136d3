// User-sharded training over several GPUs, one process per GPU (the reference is single device,
// the interface extended is train(), training.h:12-15).  Host code is C++ calling HIP and RCCL directly.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <vector>

#include "device.hpp"
#include "shard_driver.hpp"
#include "train_schedule.hpp"

namespace cu2rec {

// A communicator: RCCL (ncclComm_t, created here or handed in by the caller), or a caller-supplied sum-all-reduce
// (tests: gloo through the host, several ranks sharing one GPU).  world 1 needs neither.
struct Comm {
    int rank = 0, nranks = 1;
    void *nccl = nullptr;  // ncclComm_t
    bool owns_nccl = false;
    bool broken = false;   // a collective failed or timed out: the communicator is aborted, not destroyed
    cu2rec_allreduce_fn fn = nullptr;
    void *ctx = nullptr;
    ~Comm();
    bool collective() const { return nccl != nullptr; }  // a real collective is attached (even at one rank: tests)
    // in place, sum over ranks, ordered on `stream` (the callback form may synchronise it)
    void allreduce(void *device_buf, size_t count, bool is_double, hipStream_t stream) const;
    // hipStreamSynchronize that polls RCCL's asynchronous error state and gives up after CU2REC_COMM_TIMEOUT_S (default 600 s):
    // aborts the communicator and throws instead of hanging on a peer that died
    void wait(hipStream_t stream);
};

void comm_unique_id(void *id_out128);
Comm *comm_create_rccl(const void *id128, int rank, int nranks);
Comm *comm_adopt_rccl(void *nccl_comm, int rank, int nranks);
Comm *comm_from_callback(cu2rec_allreduce_fn fn, void *ctx, int rank, int nranks);
void comm_info(const Comm &c, cu2rec_comm_info_t &out);

// The backend of the driver template (shard_driver.hpp) in the product: device memory, the HIP kernels, RCCL.
struct HipBackend {
    using Model = DeviceModel;
    using Csr = DeviceCsr;
    using Comm = cu2rec::Comm;
    using Stream = hipStream_t;
    template <class T>
    using Buffer = DeviceBuffer<T>;
    using Clock = HipClock;
    static void require_ready() { require_device(); }
    static int rows(const Model &m) { return m.rows; }
    static int cols(const Model &m) { return m.cols; }
    static int n_factors(const Model &m) { return m.n_factors; }
    static int ldq(const Model &m) { return m.ldq; }
    static float *Q(Model &m) { return m.Q.ptr; }
    static float *item_bias(Model &m) { return m.item_bias.ptr; }
    static int csr_rows(const Csr &c) { return c.rows; }
    static int csr_nnz(const Csr &c) { return c.nnz; }
    static int csr_max_item(const Csr &c) { return c.max_item; }
    static int csr_users_with_ratings(const Csr &c) { return c.users_with_ratings; }
    static void csr_structure(const Csr &c, std::vector<int> &indptr, std::vector<int> &indices);
    static void copy(float *dst, const float *src, size_t n) {
        if (n) CU2REC_HIP(hipMemcpy(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice));
    }
    static void to_backend(void *dst, const void *src, size_t bytes, Stream s) { CU2REC_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s)); }
    static void to_host(void *dst, const void *src, size_t bytes, Stream s) { CU2REC_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s)); }
    static void wire_pack(const float *Q, const float *ib, const float *Q_base, const float *ib_base, const float *weight, int n_cols, int f,
                          int ldq, float *wire, Stream s);
    static void wire_apply(float *Q, float *ib, float *Q_base, float *ib_base, int n_cols, int f, int ldq, const float *wire, float scale,
                           Stream s);
    static void sgd(Model &m, const Csr &train, const cu2rec_hyper &h, uint64_t seed, uint64_t iter0, int n, int mode, int update_items,
                    Stream s, int user_offset) {
        m.sgd(train, h, seed, iter0, n, mode, update_items, s, false, user_offset);
    }
    static void loss(Model &m, const Csr &ratings, double *sum_abs, double *sum_sq, Stream s) { m.loss(ratings, sum_abs, sum_sq, nullptr, nullptr, s); }
};

// One rank's share of a sharded run (cu2rec_shard_job): the driver template over the HIP backend
using ShardJob = ShardDriver<HipBackend>;

void train_sharded(ShardJob &job, const DeviceCsr &test, cu2rec_config &cfg, int mode, bool verbose, float *losses,
                   cu2rec_train_stats *stats);

}  // namespace cu2rec

struct cu2rec_comm {
    cu2rec::Comm *impl;
};
struct cu2rec_shard_job {
    cu2rec::ShardJob impl;
    template <class... A>
    explicit cu2rec_shard_job(A &&...a) : impl(std::forward<A>(a)...) {}
};
