// train(): the reference's training loop (training.cu:21-217) re-expressed for this library.
// Same observable schedule -- loss cadence, stdout lines, patience / learning-rate decay, what
// the clock covers (train_schedule.hpp) -- with the SGD launches between two loss checks queued back to back on one
// stream and timed by HIP events.  No ping-pong buffers, no per-iteration memset, no RNG state:
// item rows are updated in place (mf_sequential.cu semantics, the parity target) -- except in
// CU2REC_SGD_PINGPONG, which keeps the reference GPU loop's two item buffers and its loss-before-swap order.
#include "train_schedule.hpp"

namespace cu2rec {

void train(const DeviceCsr &train_csr, const DeviceCsr &test_csr, cu2rec_config &cfg, DeviceModel &model, int mode,
           bool verbose, float *losses, cu2rec_train_stats *stats) {
    require(cfg.n_factors == model.n_factors, "cu2rec_train: cfg.n_factors differs from the model's");
    require(train_csr.rows <= model.rows && train_csr.max_item < model.cols, "cu2rec_train: train ratings exceed the model");
    require(test_csr.rows <= model.rows && test_csr.max_item < model.cols,
            "cu2rec_train: test ratings name users/items the model does not have");
    hipStream_t stream = nullptr;  // the reference runs everything on the default stream
    const uint64_t seed = static_cast<uint64_t>(static_cast<uint32_t>(cfg.seed));
    train_schedule(
        cfg, verbose, true, static_cast<double>(train_csr.users_with_ratings), losses, stats, stream,
        [&](const cu2rec_hyper &h, uint64_t first, int n, hipStream_t s) {
            // CU2REC_SGD_PINGPONG: the reference evaluates the loss BEFORE it swaps the item buffers (training.cu:121 vs
            // :164), i.e. on this iteration's P and the item side the iteration READ; so the last swap waits for the loss
            model.sgd(train_csr, h, seed, first, n, mode, cfg.is_train, s, /*defer_last_swap=*/mode == CU2REC_SGD_PINGPONG);
        },
        [&](bool on_train, float *mae, float *rmse) {
            model.loss(on_train ? train_csr : test_csr, nullptr, nullptr, mae, rmse, stream);
        },
        [&] { model.finish_swap(); });
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
