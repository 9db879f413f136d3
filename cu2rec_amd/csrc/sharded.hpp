// User-sharded training over several GPUs, one process per GPU (the reference is single device,
// the interface extended is train(), training.h:12-15).  Host code is C++ calling HIP and RCCL directly.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <vector>

#include "device.hpp"

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

// One rank's share of a sharded run: its users' CSR slice and model slice (P, user_bias for the local users; Q and
// item_bias replicated), the snapshot the item deltas are taken against, the wire buffer.
struct ShardJob {
    ShardJob(Comm &comm, DeviceModel &model, const DeviceCsr &train, int user_offset, const cu2rec_shard_options &opt);
    // n_iters iterations on the local shard, an exchange every sync_every iterations (the cadence runs across calls)
    void run(const cu2rec_hyper &h, uint64_t seed, uint64_t iter0, int n_iters, int mode, int update_items, hipStream_t stream);
    void exchange(hipStream_t stream);
    // global MAE / RMSE over all shards' slices of `ratings`
    void loss(const DeviceCsr &ratings, double *sum_abs, double *sum_sq, double *n_total, float *mae, float *rmse,
              hipStream_t stream);

    Comm &comm;
    DeviceModel &model;
    const DeviceCsr &train;
    int user_offset;
    int sync_every;     // iterations between exchanges
    int merge;          // CU2REC_MERGE_*
    int since_sync = 0;
    int exchanges = 0;
    double users_total = 0, nnz_total = 0;
    DeviceBuffer<float> Q_base, ib_base, wire, weight;
    DeviceBuffer<double> sums;  // 3 doubles for the loss reduction
    float scale() const;
};

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
