// The reference's training schedule (training.cu:101-177) as a template over "run these iterations", "evaluate the loss" and a
// CLOCK (how the SGD stretches are timed and how the run is drained at its end): cu2rec_train (one GPU), cu2rec_train_sharded (one
// process per GPU) and the host-memory test instantiation of the sharded driver (tests/host_shard/) are the same loop.  No HIP
// in this header: train_schedule.hpp adds the HIP clock.
#pragma once

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <limits>

#include "common.hpp"

namespace cu2rec {

inline bool is_check_iteration(int i, const cu2rec_config &cfg) {  // training.cu:118
    return (i + 1) % cfg.check_error == 0 || i == 0 || (i + 1) % cfg.total_iterations == 0;
}

// sgd(hyper, first_iteration, n, stream): queue n iterations; loss(train?, &mae, &rmse): evaluate on train / test (blocking);
// after_check(): what follows a loss check (CU2REC_SGD_PINGPONG's deferred swap).  `print`: this rank writes the lines.
// Clock: start(stream), stop(stream), elapsed_ms() (after a blocking loss call), drain() (training.cu:172).
template <class Clock, class Stream, class Sgd, class Loss, class After>
void train_schedule_with(cu2rec_config &cfg, bool verbose, bool print, double updates_per_iteration, float *losses,
                         cu2rec_train_stats *stats, Stream stream, Sgd sgd, Loss loss, After after_check) {
    require(cfg.total_iterations >= 0 && cfg.check_error > 0, "cu2rec_train: bad iteration counts");
    const int total = cfg.total_iterations;
    if (losses)
        for (int i = 0; i < total; ++i) losses[i] = std::numeric_limits<float>::quiet_NaN();
    Clock clock;
    cu2rec_train_stats st{};
    float train_mae = 0.f, train_rmse = 0.f, validation_mae, validation_rmse, last_validation_rmse;
    validation_rmse = validation_mae = std::numeric_limits<float>::max();  // training.cu:102
    int current_patience = static_cast<int>(cfg.patience);                 // training.cu:103
    const uint64_t iter_base = static_cast<uint64_t>(cfg.cur_iterations);

    const auto t0 = std::chrono::steady_clock::now();  // training.cu:106 (clock() there; wall clock here)
    int i = 0;
    while (i < total) {
        // queue every iteration up to and including the next loss check
        int seg_end = i;
        while (!is_check_iteration(seg_end, cfg)) ++seg_end;
        const int n = seg_end - i + 1;
        const cu2rec_hyper h{cfg.learning_rate, cfg.P_reg, cfg.Q_reg, cfg.user_bias_reg, cfg.item_bias_reg};
        clock.start(stream);
        sgd(h, iter_base + static_cast<uint64_t>(i), n, stream);
        clock.stop(stream);

        // training.cu:121-137: loss on train then test, printed in the reference's format
        loss(true, &train_mae, &train_rmse);
        last_validation_rmse = validation_rmse;  // training.cu:129
        loss(false, &validation_mae, &validation_rmse);
        st.seconds_sgd += 1e-3 * clock.elapsed_ms();
        st.n_checks += 1;
        if (verbose && print) {
            std::printf("TRAIN: Iteration %d GPU MAE: %f RMSE: %f\n", seg_end + 1, train_mae, train_rmse);
            std::printf("TEST: Iteration %d GPU MAE: %f RMSE: %f\n", seg_end + 1, validation_mae, validation_rmse);
        }
        // training.cu:146-155: patience is consumed when the test RMSE got worse, never restored on improvement
        if (last_validation_rmse < validation_rmse) current_patience--;
        if (current_patience <= 0) {
            current_patience = static_cast<int>(cfg.patience);
            cfg.learning_rate *= cfg.learning_rate_decay;
            if (verbose && print) std::printf("New Learning Rate: %f\n: ", cfg.learning_rate);
        }
        after_check();                                   // training.cu:164-165 (no-op outside CU2REC_SGD_PINGPONG)
        if (losses) losses[seg_end] = validation_rmse;  // training.cu:158
        cfg.cur_iterations += n;                         // training.cu:170
        i = seg_end + 1;
    }
    clock.drain();  // training.cu:172
    st.seconds_total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (verbose && print) std::printf("Time taken for %d of iterations is %lf\n", total, st.seconds_total);  // training.cu:177
    st.updates = updates_per_iteration * total;
    st.last_train_mae = train_mae;
    st.last_train_rmse = train_rmse;
    st.last_test_mae = validation_mae;
    st.last_test_rmse = validation_rmse;
    if (stats) *stats = st;
}

}  // namespace cu2rec
