// User-sharded training, one process per GPU: the C++ driver behind cu2rec_train_sharded and bin/mf -g N.
// Each rank owns a contiguous range of users (their CSR slice, their rows of P and user_bias) and a replica of the item
// side.  Every sync_every iterations the replicas are reconciled: wire = [Q - Q_base | item_bias - ib_base] (row padding
// stripped: I * (f + 1) floats, 10.8 MB for the ML-20M shape at f = 100), ONE ncclAllReduce(sum) over xGMI, then
// Q = Q_base + scale * wire, which is also the new snapshot.  Sampler draws are keyed by the GLOBAL user id, so a shard
// draws what the unsharded run draws.  The loss is an all-reduce of three doubles.  RCCL is resolved at run time
// (dlopen: a copy PyTorch has already loaded is reused, otherwise /opt/rocm's), so the single-GPU library does not
// depend on it.
#include "sharded.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>

#include "kernels.hpp"
#include "resident.hpp"
#include "train_schedule.hpp"

namespace cu2rec {

namespace {

// The RCCL entry points used.  TYPES and ENUM VALUES come from <rccl/rccl.h> at compile time (ncclUniqueId, ncclComm_t,
// ncclSum, ncclFloat32 / 64, ncclResult_t); the SYMBOLS are resolved at run time (dlopen: a copy PyTorch has already loaded
// is reused, otherwise /opt/rocm's), so the single-GPU library does not link RCCL.  The library found at run time must be
// of the header's major version (ncclGetVersion): the ABI of the calls below is what the header declares.
struct Rccl {
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;                  // optional
    decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;  // optional
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;        // optional (cu2rec_comm_info)
    decltype(&ncclCommUserRank) CommUserRank = nullptr;  // optional
    decltype(&ncclCommCuDevice) CommCuDevice = nullptr;  // optional
    int version = 0;
    bool ok = false;
    std::string why;
};

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        void *h = nullptr;
        for (const char *name : {"librccl.so.1", "librccl.so"}) {  // one already in the process (PyTorch's) first
            h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (h) break;
        }
        if (!h)
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (h) break;
            }
        if (!h) {
            r.why = std::string("cannot load librccl: ") + dlerror();
            return;
        }
        auto sym = [&](const char *name) { return dlsym(h, name); };
        r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(sym("ncclGetVersion"));
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(sym("ncclCommAbort"));
        r.CommGetAsyncError = reinterpret_cast<decltype(r.CommGetAsyncError)>(sym("ncclCommGetAsyncError"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
        r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(sym("ncclCommUserRank"));
        r.CommCuDevice = reinterpret_cast<decltype(r.CommCuDevice)>(sym("ncclCommCuDevice"));
        if (!(r.GetVersion && r.GetUniqueId && r.CommInitRank && r.AllReduce && r.CommDestroy)) {
            r.why = "librccl lacks ncclGetVersion / ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy";
            return;
        }
        if (r.GetVersion(&r.version) != ncclSuccess) {
            r.why = "ncclGetVersion failed";
            return;
        }
        // NCCL_VERSION_CODE = major * 10000 + minor * 100 + patch since 2.9 (rccl.h); the calls used here have been stable
        // across the 2.x line, another major version is not trusted
        const int major = r.version >= 10000 ? r.version / 10000 : r.version / 1000;
        if (major != NCCL_MAJOR) {
            r.why = "librccl reports version " + std::to_string(r.version) + ", the library was built against " +
                    std::to_string(NCCL_VERSION_CODE) + " (rccl/rccl.h): another major version";
            return;
        }
        r.ok = true;
    });
    if (!r.ok) fail(CU2REC_EUNSUPPORTED, "cu2rec_amd: RCCL is not available (" + r.why + ")");
    return r;
}

void rccl_check(ncclResult_t code, const char *what) {
    if (code == ncclSuccess) return;
    Rccl &r = rccl();
    fail(CU2REC_EHIP, std::string("RCCL error in ") + what + ": " + (r.GetErrorString ? r.GetErrorString(code) : "?"));
}

double comm_timeout_seconds() {  // how long a rank waits for a collective before it gives the communicator up
    static const double t = [] {
        const char *env = std::getenv("CU2REC_COMM_TIMEOUT_S");
        const double v = env ? std::atof(env) : 600.0;
        return v > 0 ? v : 600.0;
    }();
    return t;
}

}  // namespace

Comm::~Comm() {
    if (nccl && owns_nccl) {
        Rccl &r = rccl();
        // a communicator that has seen a failure is aborted, not destroyed: ncclCommDestroy waits for outstanding work
        if (broken && r.CommAbort) (void)r.CommAbort(static_cast<ncclComm_t>(nccl));
        else (void)r.CommDestroy(static_cast<ncclComm_t>(nccl));
    }
}

// Waits for everything queued on `stream` -- in particular a collective -- without hanging for ever on a peer that died: polls
// the stream and RCCL's asynchronous error state; on an error or after CU2REC_COMM_TIMEOUT_S (default 600 s) the communicator
// is aborted (ncclCommAbort: the local kernels are torn down) and the call fails, so that the process can exit non-zero
// instead of sitting in the collective.
void Comm::wait(hipStream_t stream) {
    if (!nccl) {
        CU2REC_HIP(hipStreamSynchronize(stream));
        return;
    }
    Rccl &r = rccl();
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    for (;;) {
        const hipError_t q = hipStreamQuery(stream);
        if (q == hipSuccess) return;
        if (q != hipErrorNotReady) CU2REC_HIP(q);
        std::string why;
        if (r.CommGetAsyncError) {
            ncclResult_t async = ncclSuccess;
            const ncclResult_t rc = r.CommGetAsyncError(static_cast<ncclComm_t>(nccl), &async);
            if (rc != ncclSuccess) why = std::string("ncclCommGetAsyncError: ") + (r.GetErrorString ? r.GetErrorString(rc) : "?");
            else if (async != ncclSuccess && async != ncclInProgress)
                why = std::string("asynchronous RCCL error: ") + (r.GetErrorString ? r.GetErrorString(async) : "?");
        }
        if (why.empty() && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > comm_timeout_seconds())
            why = "a collective did not complete within " + std::to_string(static_cast<int>(comm_timeout_seconds())) +
                  " s (CU2REC_COMM_TIMEOUT_S): did another rank die?";
        if (!why.empty()) {
            broken = true;
            if (r.CommAbort && owns_nccl) {
                (void)r.CommAbort(static_cast<ncclComm_t>(nccl));
                nccl = nullptr;
            }
            fail(CU2REC_EHIP, "cu2rec_comm (rank " + std::to_string(rank) + " of " + std::to_string(nranks) + "): " + why);
        }
        if (++spins < 2000) std::this_thread::yield();  // a collective of this path takes tens of microseconds
        else std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}

void Comm::allreduce(void *buf, size_t count, bool is_double, hipStream_t stream) const {
    if (count == 0 || (nranks == 1 && !nccl)) return;  // (a one-rank RCCL communicator exists for tests only)
    if (nccl) {
        rccl_check(rccl().AllReduce(buf, buf, count, is_double ? ncclFloat64 : ncclFloat32, ncclSum, static_cast<ncclComm_t>(nccl), stream),
                   "ncclAllReduce");
        return;
    }
    require(fn != nullptr, "cu2rec_comm: no communicator behind a multi-rank job");
    if (fn(ctx, buf, count, is_double ? 1 : 0, stream) != 0) fail(CU2REC_EHIP, "cu2rec_comm: the caller's all-reduce failed");
}

static_assert(sizeof(ncclUniqueId) == 128, "include/cu2rec_amd.h hands the unique id around as 128 bytes");

void comm_unique_id(void *id_out128) {
    ncclUniqueId id;
    rccl_check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
    std::memcpy(id_out128, id.internal, sizeof(id.internal));
}

Comm *comm_create_rccl(const void *id128, int rank, int nranks) {
    require(id128 && nranks >= 1 && rank >= 0 && rank < nranks, "cu2rec_comm_create: bad argument");
    require_device();
    Comm *c = new Comm;
    c->rank = rank;
    c->nranks = nranks;
    // CU2REC_RCCL_WORLD1=1: a real communicator even for one rank, so that a one-GPU box exercises ncclAllReduce
    const char *force = std::getenv("CU2REC_RCCL_WORLD1");
    if (nranks > 1 || (force && *force == '1')) {
        ncclUniqueId id;
        std::memcpy(id.internal, id128, sizeof(id.internal));
        try {
            ncclComm_t made = nullptr;
            rccl_check(rccl().CommInitRank(&made, nranks, id, rank), "ncclCommInitRank");
            c->nccl = made;
        } catch (...) {
            delete c;
            throw;
        }
        c->owns_nccl = true;
    }
    return c;
}

Comm *comm_adopt_rccl(void *nccl_comm, int rank, int nranks) {
    require(nranks >= 1 && rank >= 0 && rank < nranks && (nccl_comm || nranks == 1), "cu2rec_comm_from_nccl: bad argument");
    if (nranks > 1) (void)rccl();
    Comm *c = new Comm;
    c->rank = rank;
    c->nranks = nranks;
    c->nccl = nranks > 1 ? nccl_comm : nullptr;
    return c;
}

// What RCCL itself says about the attached communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice): the proof on a
// bench line that the collective spans N ranks, not what the caller asked for
void comm_info(const Comm &c, cu2rec_comm_info_t &out) {
    out = cu2rec_comm_info_t{c.rank, c.nranks, 0, -1, -1, 0, c.fn != nullptr ? 1 : 0};
    if (!c.nccl) return;
    Rccl &r = rccl();
    out.rccl_version = r.version;
    const ncclComm_t comm = static_cast<ncclComm_t>(c.nccl);
    if (r.CommCount) rccl_check(r.CommCount(comm, &out.rccl_nranks), "ncclCommCount");
    if (r.CommUserRank) rccl_check(r.CommUserRank(comm, &out.rccl_rank), "ncclCommUserRank");
    if (r.CommCuDevice) rccl_check(r.CommCuDevice(comm, &out.rccl_device), "ncclCommCuDevice");
}

Comm *comm_from_callback(cu2rec_allreduce_fn fn, void *ctx, int rank, int nranks) {
    require(nranks >= 1 && rank >= 0 && rank < nranks && (fn || nranks == 1), "cu2rec_comm_from_callback: bad argument");
    Comm *c = new Comm;
    c->rank = rank;
    c->nranks = nranks;
    c->fn = fn;
    c->ctx = ctx;
    return c;
}

// ----------------------------------------------------------------------------------------------- the driver's HIP backend
// (the driver itself -- cadence, exchange, merge weights, loss reduction, train() over all ranks -- is the template in
// shard_driver.hpp; tests/host_shard/ instantiates the same template over host memory)
void HipBackend::csr_structure(const DeviceCsr &c, std::vector<int> &indptr, std::vector<int> &indices) {
    indptr.resize(static_cast<size_t>(c.rows) + 1);
    indices.resize(static_cast<size_t>(std::max(c.nnz, 1)));
    CU2REC_HIP(hipMemcpy(indptr.data(), c.indptr.ptr, indptr.size() * sizeof(int), hipMemcpyDeviceToHost));
    if (c.nnz) CU2REC_HIP(hipMemcpy(indices.data(), c.indices.ptr, static_cast<size_t>(c.nnz) * sizeof(int), hipMemcpyDeviceToHost));
}

void HipBackend::wire_pack(const float *Q, const float *ib, const float *Q_base, const float *ib_base, const float *weight, int n_cols, int f,
                           int ldq, float *wire, hipStream_t s) {
    launch_items_wire_pack(Q, ib, Q_base, ib_base, weight, n_cols, f, ldq, wire, s);
    CU2REC_HIP(hipGetLastError());
}

void HipBackend::wire_apply(float *Q, float *ib, float *Q_base, float *ib_base, int n_cols, int f, int ldq, const float *wire, float scale,
                            hipStream_t s) {
    launch_items_wire_apply(Q, ib, Q_base, ib_base, n_cols, f, ldq, wire, scale, s);
    CU2REC_HIP(hipGetLastError());
}

void train_sharded(ShardJob &job, const DeviceCsr &test, cu2rec_config &cfg, int mode, bool verbose, float *losses,
                   cu2rec_train_stats *stats) {
    shard_train<HipBackend>(job, test, cfg, mode, verbose, losses, stats);
}

}  // namespace cu2rec

using namespace cu2rec;

extern "C" {

int cu2rec_comm_unique_id(void *id_out) {
    return guarded([&] {
        require(id_out, "cu2rec_comm_unique_id: null argument");
        comm_unique_id(id_out);
    });
}

int cu2rec_comm_create(const void *unique_id, int rank, int nranks, cu2rec_comm **out) {
    return guarded([&] {
        require(out, "out is null");
        *out = nullptr;
        *out = new cu2rec_comm{comm_create_rccl(unique_id, rank, nranks)};
    });
}

int cu2rec_comm_from_nccl(void *nccl_comm, int rank, int nranks, cu2rec_comm **out) {
    return guarded([&] {
        require(out, "out is null");
        *out = nullptr;
        *out = new cu2rec_comm{comm_adopt_rccl(nccl_comm, rank, nranks)};
    });
}

int cu2rec_comm_from_callback(cu2rec_allreduce_fn fn, void *ctx, int rank, int nranks, cu2rec_comm **out) {
    return guarded([&] {
        require(out, "out is null");
        *out = nullptr;
        *out = new cu2rec_comm{comm_from_callback(fn, ctx, rank, nranks)};
    });
}

int cu2rec_comm_info(const cu2rec_comm *c, cu2rec_comm_info_t *out) {
    return guarded([&] {
        require(c && c->impl && out, "cu2rec_comm_info: null argument");
        comm_info(*c->impl, *out);
    });
}

void cu2rec_comm_destroy(cu2rec_comm *c) {
    if (!c) return;
    delete c->impl;
    delete c;
}

int cu2rec_shard_job_create(cu2rec_comm *comm, cu2rec_model *model, const cu2rec_csr *train, int user_offset,
                            const cu2rec_shard_options *options, cu2rec_shard_job **out) {
    return guarded([&] {
        require(comm && comm->impl && model && train && out, "cu2rec_shard_job_create: null argument");
        *out = nullptr;
        cu2rec_shard_options opt{0, CU2REC_MERGE_ADAPTIVE};  // the one default everywhere (bin/mf, bench.py, the Python mirror)
        if (options) opt = *options;
        *out = new cu2rec_shard_job(*comm->impl, unwrap(model), unwrap(train), user_offset, opt);
    });
}

void cu2rec_shard_job_destroy(cu2rec_shard_job *job) { delete job; }

int cu2rec_shard_job_run(cu2rec_shard_job *job, const cu2rec_hyper *hyper, uint64_t seed, uint64_t iter0, int n_iters,
                         int mode, int update_items) {
    return guarded([&] {
        require(job && hyper, "cu2rec_shard_job_run: null argument");
        job->impl.run(*hyper, seed, iter0, n_iters, mode, update_items, nullptr);
    });
}

int cu2rec_shard_job_exchange(cu2rec_shard_job *job) {
    return guarded([&] {
        require(job, "cu2rec_shard_job_exchange: null argument");
        if (job->impl.since_sync > 0) job->impl.exchange(nullptr);
    });
}

int cu2rec_shard_job_loss(cu2rec_shard_job *job, const cu2rec_csr *ratings, double *sum_abs, double *sum_sq,
                          double *n_total, float *mae, float *rmse) {
    return guarded([&] {
        require(job && ratings, "cu2rec_shard_job_loss: null argument");
        job->impl.loss(unwrap(ratings), sum_abs, sum_sq, n_total, mae, rmse, nullptr);
    });
}

int cu2rec_shard_job_info(const cu2rec_shard_job *job, int *sync_every, int *exchanges, double *users_total,
                          double *nnz_total, size_t *wire_bytes) {
    return guarded([&] {
        require(job, "cu2rec_shard_job_info: null argument");
        if (sync_every) *sync_every = job->impl.sync_every;
        if (exchanges) *exchanges = job->impl.exchanges;
        if (users_total) *users_total = job->impl.users_total;
        if (nnz_total) *nnz_total = job->impl.nnz_total;
        if (wire_bytes) *wire_bytes = job->impl.wire_floats() * sizeof(float);
    });
}

int cu2rec_shard_job_exchange_stats(cu2rec_shard_job *job, int *timed, double *seconds, double *max_seconds) {
    return guarded([&] {
        require(job, "cu2rec_shard_job_exchange_stats: null argument");
        job->impl.exchange_stats(timed, seconds, max_seconds);
    });
}

int cu2rec_train_sharded(cu2rec_shard_job *job, const cu2rec_csr *test, cu2rec_config *cfg, int mode, int verbose,
                         float *losses, cu2rec_train_stats *stats) {
    return guarded([&] {
        require(job && test && cfg, "cu2rec_train_sharded: null argument");
        train_sharded(job->impl, unwrap(test), *cfg, mode, verbose != 0, losses, stats);
    });
}

}  // extern "C"
