// train(): the reference's training loop (training.cu:21-217) re-expressed for this library.
// Same observable schedule -- loss cadence, stdout lines, patience / learning-rate decay, what
// the clock covers -- with the SGD launches between two loss checks queued back to back on one
// stream and timed by HIP events.  No ping-pong buffers, no per-iteration memset, no RNG state:
// item rows are updated in place (mf_sequential.cu semantics, the parity target) -- except in
// CU2REC_SGD_PINGPONG, which keeps the reference GPU loop's two item buffers and its loss-before-swap order.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <limits>

#include "device.hpp"

namespace cu2rec {

namespace {

struct EventPair {
    hipEvent_t start = nullptr, stop = nullptr;
    EventPair() {
        CU2REC_HIP(hipEventCreate(&start));
        CU2REC_HIP(hipEventCreate(&stop));
    }
    ~EventPair() {
        if (start) (void)hipEventDestroy(start);
        if (stop) (void)hipEventDestroy(stop);
    }
};

bool is_check_iteration(int i, const cu2rec_config &cfg) {  // training.cu:118
    return (i + 1) % cfg.check_error == 0 || i == 0 || (i + 1) % cfg.total_iterations == 0;
}

}  // namespace

void train(const DeviceCsr &train_csr, const DeviceCsr &test_csr, cu2rec_config &cfg, DeviceModel &model, int mode,
           bool verbose, float *losses, cu2rec_train_stats *stats) {
    require(cfg.total_iterations >= 0 && cfg.check_error > 0, "cu2rec_train: bad iteration counts");
    require(cfg.n_factors == model.n_factors, "cu2rec_train: cfg.n_factors differs from the model's");
    require(train_csr.rows <= model.rows && train_csr.max_item < model.cols, "cu2rec_train: train ratings exceed the model");
    require(test_csr.rows <= model.rows && test_csr.max_item < model.cols,
            "cu2rec_train: test ratings name users/items the model does not have");
    const int total = cfg.total_iterations;
    if (losses)
        for (int i = 0; i < total; ++i) losses[i] = std::numeric_limits<float>::quiet_NaN();

    hipStream_t stream = nullptr;  // the reference runs everything on the default stream
    EventPair ev;
    cu2rec_train_stats st{};
    float train_mae = 0.f, train_rmse = 0.f, validation_mae, validation_rmse, last_validation_rmse;
    validation_rmse = validation_mae = std::numeric_limits<float>::max();  // training.cu:102
    int current_patience = static_cast<int>(cfg.patience);                 // training.cu:103
    const uint64_t iter_base = static_cast<uint64_t>(cfg.cur_iterations);
    const uint64_t seed = static_cast<uint64_t>(static_cast<uint32_t>(cfg.seed));

    const auto t0 = std::chrono::steady_clock::now();  // training.cu:106 (clock() there; wall clock here)
    int i = 0;
    while (i < total) {
        // queue every iteration up to and including the next loss check
        int seg_end = i;
        while (!is_check_iteration(seg_end, cfg)) ++seg_end;
        const int n = seg_end - i + 1;
        const cu2rec_hyper h{cfg.learning_rate, cfg.P_reg, cfg.Q_reg, cfg.user_bias_reg, cfg.item_bias_reg};
        CU2REC_HIP(hipEventRecord(ev.start, stream));
        // CU2REC_SGD_PINGPONG: the reference evaluates the loss BEFORE it swaps the item buffers (training.cu:121 vs
        // :164), i.e. on this iteration's P and the item side the iteration READ; so the last swap waits for the loss
        model.sgd(train_csr, h, seed, iter_base + static_cast<uint64_t>(i), n, mode, cfg.is_train, stream,
                  /*defer_last_swap=*/mode == CU2REC_SGD_PINGPONG);
        CU2REC_HIP(hipEventRecord(ev.stop, stream));

        // training.cu:121-137: loss on train then test, printed in the reference's format
        model.loss(train_csr, nullptr, nullptr, &train_mae, &train_rmse, stream);
        last_validation_rmse = validation_rmse;  // training.cu:129
        model.loss(test_csr, nullptr, nullptr, &validation_mae, &validation_rmse, stream);
        float ms = 0.f;
        CU2REC_HIP(hipEventElapsedTime(&ms, ev.start, ev.stop));
        st.seconds_sgd += 1e-3 * ms;
        st.n_checks += 1;
        if (verbose) {
            std::printf("TRAIN: Iteration %d GPU MAE: %f RMSE: %f\n", seg_end + 1, train_mae, train_rmse);
            std::printf("TEST: Iteration %d GPU MAE: %f RMSE: %f\n", seg_end + 1, validation_mae, validation_rmse);
        }
        // training.cu:146-155: patience is consumed when the test RMSE got worse, never restored on improvement
        if (last_validation_rmse < validation_rmse) current_patience--;
        if (current_patience <= 0) {
            current_patience = static_cast<int>(cfg.patience);
            cfg.learning_rate *= cfg.learning_rate_decay;
            if (verbose) std::printf("New Learning Rate: %f\n: ", cfg.learning_rate);
        }
        model.finish_swap();                             // training.cu:164-165 (no-op outside CU2REC_SGD_PINGPONG)
        if (losses) losses[seg_end] = validation_rmse;  // training.cu:158
        cfg.cur_iterations += n;                         // training.cu:170
        i = seg_end + 1;
    }
    CU2REC_HIP(hipDeviceSynchronize());  // training.cu:172
    st.seconds_total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (verbose) std::printf("Time taken for %d of iterations is %lf\n", total, st.seconds_total);  // training.cu:177
    st.updates = static_cast<double>(train_csr.users_with_ratings) * total;
    st.last_train_mae = train_mae;
    st.last_train_rmse = train_rmse;
    st.last_test_mae = validation_mae;
    st.last_test_rmse = validation_rmse;
    if (stats) *stats = st;
}

}  // namespace cu2rec

extern "C" int cu2rec_train(const cu2rec_csr *train, const cu2rec_csr *test, cu2rec_config *cfg, cu2rec_model *model,
                            int mode, int verbose, float *losses, cu2rec_train_stats *stats) {
    using namespace cu2rec;
    return guarded([&] {
        require(train && test && cfg && model, "cu2rec_train: null argument");
        cu2rec::train(unwrap(train), unwrap(test), *cfg, unwrap(model), mode, verbose != 0, losses, stats);
    });
}
