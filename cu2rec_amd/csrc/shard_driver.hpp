// The user-sharded driver's logic -- what one rank of a sharded run does between the kernels: the exchange cadence, the wire
// exchange with its merge rules (mean / weighted / sum / adaptive: the per-item weights from every shard's expected update
// rates), the global loss reduction and train() over all ranks (training.h:12-15) -- as a template over a BACKEND that says
// where the arrays live and who runs the SGD and loss passes.
//   * the product: HipBackend in sharded.cpp (device memory, the HIP kernels, RCCL): cu2rec_shard_job_*, cu2rec_train_sharded,
//     bin/mf -g N.  There is no other backend in libcu2rec_amd.so: the compute path stays GPU-only.
//   * tests/host_shard/host_shard.cpp instantiates the SAME template with host arrays and the CPU oracle as the engine, so that
//     the world-2 gloo tests of tests/test_parallel_cpu.py run this code -- the product's exchange logic, not a restatement of
//     it -- on a box without a GPU.  Test infrastructure: never linked into the library.
// Backend B provides:
//   types     Model, Csr, Comm, Stream, Buffer<T> (allocate(n), ptr, upload(host, n), download(host, n)), Clock (train(); the
//             exchanges' timers: start(stream), stop(stream), done(), elapsed_ms())
//   shape     rows(m), cols(m), n_factors(m), ldq(m), Q(m), item_bias(m); csr_rows(c), csr_nnz(c), csr_max_item(c),
//             csr_users_with_ratings(c), csr_structure(c, indptr, indices)
//   memory    copy(dst, src, n) inside the backend's memory; to_backend(dst, src, bytes, stream), to_host(dst, src, bytes, stream)
//   compute   wire_pack(...), wire_apply(...), sgd(...), loss(...), require_ready()
//   comm      c.rank, c.nranks, c.collective() (a real collective is attached even at one rank), c.allreduce(ptr, count,
//             is_double, stream), c.wait(stream)
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <vector>

#include "common.hpp"
#include "train_schedule_core.hpp"

namespace cu2rec {

template <class B>
struct ShardDriver {
    using Model = typename B::Model;
    using Csr = typename B::Csr;
    using Comm = typename B::Comm;
    using Stream = typename B::Stream;
    template <class T>
    using Buffer = typename B::template Buffer<T>;

    Comm &comm;
    Model &model;
    const Csr &train;
    int user_offset;
    int sync_every;  // iterations between exchanges
    int merge;       // CU2REC_MERGE_*
    int since_sync = 0;
    int exchanges = 0;
    double users_total = 0, nnz_total = 0;
    Buffer<float> Q_base, ib_base, wire, weight;
    Buffer<double> sums;  // 3 doubles for the loss reduction
    // device time of the exchanges (cu2rec_shard_job_exchange_stats): a ring of event pairs, read when their slot comes round
    // again or when the statistics are asked for -- never a wait inside the data path
    static constexpr int kTimers = 32;
    std::vector<std::unique_ptr<typename B::Clock>> timers;
    std::vector<char> timer_pending;
    int exchanges_timed = 0;
    double exchange_seconds = 0, exchange_seconds_max = 0;

    // One rank's share of a sharded run: its users' CSR slice and model slice (P, user_bias for the local users; Q and
    // item_bias replicated), the snapshot the item deltas are taken against, the wire buffer.
    ShardDriver(Comm &comm_, Model &model_, const Csr &train_, int user_offset_, const cu2rec_shard_options &opt)
        : comm(comm_), model(model_), train(train_), user_offset(user_offset_), sync_every(opt.sync_every), merge(opt.merge) {
        const int cols = B::cols(model), f = B::n_factors(model), ldq = B::ldq(model);
        require(B::csr_rows(train) <= B::rows(model) && B::csr_max_item(train) < cols, "cu2rec_shard_job: ratings exceed the model's shape");
        require(merge == CU2REC_MERGE_MEAN || merge == CU2REC_MERGE_WEIGHTED || merge == CU2REC_MERGE_SUM || merge == CU2REC_MERGE_ADAPTIVE,
                "cu2rec_shard_job: unknown merge");
        require(user_offset >= 0, "cu2rec_shard_job: negative user offset");
        B::require_ready();
        const size_t nq = static_cast<size_t>(cols) * ldq;
        Q_base.allocate(std::max<size_t>(nq, 4));
        ib_base.allocate(std::max(cols, 1));
        wire.allocate(std::max<size_t>(static_cast<size_t>(cols) * (f + 1), 4));
        sums.allocate(4);
        B::copy(Q_base.ptr, B::Q(model), nq);
        B::copy(ib_base.ptr, B::item_bias(model), static_cast<size_t>(cols));
        // population totals (the epoch length) and, for the weighted merge, every item's expected updates per iteration
        double totals[2] = {static_cast<double>(B::csr_users_with_ratings(train)), static_cast<double>(B::csr_nnz(train))};
        sums.upload(totals, 2);
        comm.allreduce(sums.ptr, 2, true, Stream{});
        comm.wait(Stream{});
        sums.download(totals, 2);
        users_total = totals[0];
        nnz_total = totals[1];
        if (sync_every <= 0)  // "each epoch" (north star): one epoch = nnz / users iterations, SURVEY.md section 8e
            sync_every = std::max(1, static_cast<int>(std::lround(nnz_total / std::max(users_total, 1.0))));
        if ((merge == CU2REC_MERGE_WEIGHTED || merge == CU2REC_MERGE_ADAPTIVE) && comm.nranks > 1) {
            // w_k[y] = rate_k[y] / sum_j rate_j[y], rate = sum over the shard's raters of 1 / degree (host, double: the same
            // bits every run); items nobody rates anywhere keep weight 1 / N (their delta is zero anyway)
            std::vector<int> indptr, indices;
            B::csr_structure(train, indptr, indices);
            std::vector<double> rate(static_cast<size_t>(std::max(cols, 1)), 0.0);
            for (int u = 0; u < B::csr_rows(train); ++u) {
                const int low = indptr[u], high = indptr[u + 1];
                for (int k = low; k < high; ++k) rate[indices[k]] += 1.0 / (high - low);
            }
            Buffer<double> all;
            all.allocate(rate.size());
            all.upload(rate.data(), rate.size());
            comm.allreduce(all.ptr, rate.size(), true, Stream{});
            std::vector<double> total(rate.size());
            comm.wait(Stream{});
            all.download(total.data(), total.size());
            std::vector<float> w(rate.size());
            if (merge == CU2REC_MERGE_WEIGHTED) {
                for (size_t y = 0; y < rate.size(); ++y) w[y] = static_cast<float>(total[y] > 0 ? rate[y] / total[y] : 1.0 / comm.nranks);
            } else {
                // Adaptive: the deltas are SUMMED and scaled per item by alpha = phi(r_total) / sum_k phi(r_k), phi(r) = 1 - exp(-c r):
                // a shard's delta of an item row is, to first order, the progress phi of its own updates towards a common
                // target, and all shards' updates in sequence would make phi of the total -- alpha is 1 for an item that is
                // rarely updated (the sum is what the sequential run does) and 1 / N for an item every shard updates many times
                // per iteration (the mean).  c = 6 fitted on the ML-20M shape (tools/shard_study.py, profiles/r02_shard_study_*).
                // ... and scaled with the exchange period (round 6): what saturates an item row is its updates per PERIOD, so a run
                // that exchanges every sync_every iterations instead of once per epoch uses c = 6 sync_every / epoch (N = 8, converged,
                // equal LR histories: sync 2 -3.0e-3 instead of -1.1e-2, sync 1 -1.5e-3; profiles/r06_sharded_equal_schedule.txt).  At
                // the default period -- one epoch -- nothing changes.
                const double epoch = std::max(1.0, nnz_total / std::max(users_total, 1.0));
                double c = 6.0 * std::min(1.0, static_cast<double>(sync_every) / epoch);
                if (const char *env = std::getenv("CU2REC_MERGE_ADAPTIVE_C")) c = std::max(1e-3, std::atof(env));
                std::vector<double> phi(rate.size());
                for (size_t y = 0; y < rate.size(); ++y) phi[y] = -std::expm1(-c * rate[y]);
                all.upload(phi.data(), phi.size());
                comm.allreduce(all.ptr, phi.size(), true, Stream{});
                comm.wait(Stream{});
                all.download(phi.data(), phi.size());  // now the sum over the ranks
                for (size_t y = 0; y < rate.size(); ++y) w[y] = static_cast<float>(phi[y] > 0 ? -std::expm1(-c * total[y]) / phi[y] : 1.0);
            }
            weight.allocate(w.size());
            weight.upload(w.data(), w.size());
        }
    }

    float scale() const {
        return merge == CU2REC_MERGE_MEAN ? 1.f / static_cast<float>(comm.nranks) : 1.f;  // weighted / adaptive: in the weights
    }

    size_t wire_floats() const { return static_cast<size_t>(B::cols(model)) * (B::n_factors(model) + 1); }

    // wire = [w (Q - Q_base) | w (item_bias - ib_base)] without the rows' padding -> ONE sum-all-reduce -> Q = Q_base + scale wire,
    // which is also the new snapshot
    void exchange(Stream stream) {
        since_sync = 0;
        if (comm.nranks == 1 && !comm.collective()) return;
        if (timers.empty()) {
            timers.resize(kTimers);
            timer_pending.assign(kTimers, 0);
        }
        const int slot = exchanges % kTimers;
        harvest(slot);
        if (!timers[slot]) timers[slot] = std::make_unique<typename B::Clock>();
        const bool timed = !timer_pending[slot];  // (still in flight a whole ring later: that old sample is kept, this exchange goes untimed)
        if (timed) timers[slot]->start(stream);
        B::wire_pack(B::Q(model), B::item_bias(model), Q_base.ptr, ib_base.ptr, weight.ptr, B::cols(model), B::n_factors(model), B::ldq(model),
                     wire.ptr, stream);
        comm.allreduce(wire.ptr, wire_floats(), false, stream);
        B::wire_apply(B::Q(model), B::item_bias(model), Q_base.ptr, ib_base.ptr, B::cols(model), B::n_factors(model), B::ldq(model), wire.ptr,
                      scale(), stream);
        if (timed) {
            timers[slot]->stop(stream);
            timer_pending[slot] = 1;
        }
        ++exchanges;
    }

    void harvest(int slot) {
        if (!timer_pending[slot] || !timers[slot]->done()) return;
        const double s = 1e-3 * timers[slot]->elapsed_ms();
        exchange_seconds += s;
        exchange_seconds_max = std::max(exchange_seconds_max, s);
        ++exchanges_timed;
        timer_pending[slot] = 0;
    }

    void exchange_stats(int *timed, double *seconds, double *max_seconds) {
        for (size_t slot = 0; slot < timers.size(); ++slot) harvest(static_cast<int>(slot));
        if (timed) *timed = exchanges_timed;
        if (seconds) *seconds = exchange_seconds;
        if (max_seconds) *max_seconds = exchange_seconds_max;
    }

    // n_iters iterations on the local shard, an exchange every sync_every iterations (the cadence runs across calls)
    void run(const cu2rec_hyper &h, uint64_t seed, uint64_t iter0, int n_iters, int mode, int update_items, Stream stream) {
        require(n_iters >= 0, "cu2rec_shard_job_run: bad iteration count");
        require(mode != CU2REC_SGD_PINGPONG, "cu2rec_shard_job_run: the ping-pong mode swaps item buffers and is single-GPU only");
        int done = 0;
        while (done < n_iters) {
            const int n = std::min(n_iters - done, sync_every - since_sync);
            B::sgd(model, train, h, seed, iter0 + static_cast<uint64_t>(done), n, mode, update_items, stream, user_offset);
            done += n;
            since_sync += n;
            if (since_sync >= sync_every) {
                // the period ends whether or not anything moved: with frozen items nothing is exchanged, but the counter
                // starts over (or n would stay 0 for ever)
                if (update_items) exchange(stream);
                else since_sync = 0;
            }
        }
    }

    // global MAE / RMSE over all shards' slices of `ratings`
    void loss(const Csr &ratings, double *sum_abs, double *sum_sq, double *n_total, float *mae, float *rmse, Stream stream) {
        double host[3] = {0.0, 0.0, static_cast<double>(B::csr_nnz(ratings))};
        // the wire all-reduce of an exchange may still be queued on `stream`, and the loss pass ends in a plain stream
        // synchronisation: wait for it HERE, bounded (the product's Comm::wait polls ncclCommGetAsyncError and gives up after
        // CU2REC_COMM_TIMEOUT_S), or a dead peer would hold this rank inside that synchronisation for ever
        if (comm.nranks > 1) comm.wait(stream);
        B::loss(model, ratings, &host[0], &host[1], stream);
        if (comm.nranks > 1) {
            B::to_backend(sums.ptr, host, sizeof(host), stream);
            comm.allreduce(sums.ptr, 3, true, stream);
            B::to_host(host, sums.ptr, sizeof(host), stream);
            comm.wait(stream);  // (bounded: a dead peer ends the call, not the night)
        }
        if (sum_abs) *sum_abs = host[0];
        if (sum_sq) *sum_sq = host[1];
        if (n_total) *n_total = host[2];
        if (mae) *mae = static_cast<float>(host[0] / host[2]);              // loss.cu:189
        if (rmse) *rmse = static_cast<float>(std::sqrt(host[1] / host[2]));
    }
};

// train() (training.h:12-15) over all ranks: the same observable schedule as cu2rec_train -- loss on train and test at
// i == 0, every check_error and last, the TRAIN: / TEST: lines (rank 0), patience / learning-rate decay on the GLOBAL
// test RMSE (identical on every rank: the loss sums are all-reduced), cfg.learning_rate / cfg.cur_iterations updated.
template <class B>
void shard_train(ShardDriver<B> &job, const typename B::Csr &test, cu2rec_config &cfg, int mode, bool verbose, float *losses,
                 cu2rec_train_stats *stats) {
    require(cfg.n_factors == B::n_factors(job.model), "cu2rec_train_sharded: cfg.n_factors differs from the model's");
    require(B::csr_rows(test) <= B::rows(job.model) && B::csr_max_item(test) < B::cols(job.model),
            "cu2rec_train_sharded: test ratings name users/items the model does not have");
    typename B::Stream stream{};
    const uint64_t seed = static_cast<uint64_t>(static_cast<uint32_t>(cfg.seed));
    train_schedule_with<typename B::Clock>(
        cfg, verbose, job.comm.rank == 0, job.users_total, losses, stats, stream,
        [&](const cu2rec_hyper &h, uint64_t first, int n, typename B::Stream s) {
            job.run(h, seed, first, n, mode, cfg.is_train, s);
            // a loss check follows: every replica must hold the same item side (an exchange out of cadence, like the
            // Python driver's exchange(final=True))
            if (cfg.is_train && job.since_sync > 0) job.exchange(s);
        },
        [&](bool on_train, float *mae, float *rmse) {
            job.loss(on_train ? job.train : test, nullptr, nullptr, nullptr, mae, rmse, stream);
        },
        [] {});
}

}  // namespace cu2rec
