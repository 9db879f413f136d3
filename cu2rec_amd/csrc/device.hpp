// Owned device objects and the C++ form of the raw hot-path entry points.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include <memory>

#include "hip_check.hpp"
#include "ordered.hpp"

struct cu2rec_csr;
struct cu2rec_model;

namespace cu2rec {

// fills the launch argument block after validating shapes / alignment (shared by all SGD modes)
SgdArgs make_sgd_args(const int *indptr, const int *indices, const float *data, int n_rows, int n_cols, float *P,
                      int ldp, float *Q, int ldq, float *user_bias, float *item_bias, float global_bias, int n_factors,
                      const cu2rec_hyper &hyper, uint64_t seed, int update_items, int user_offset);

void sgd_update(const int *indptr, const int *indices, const float *data, int n_rows, int n_cols, float *P, int ldp,
                float *Q, int ldq, float *user_bias, float *item_bias, float global_bias, int n_factors,
                const cu2rec_hyper &hyper, uint64_t seed, uint64_t iter0, int n_iters, int mode, int update_items,
                int user_offset, hipStream_t stream, const void *sample_pairs = nullptr);

// CU2REC_SGD_PINGPONG on raw pointers (include/cu2rec_amd.h: cu2rec_sgd_update_pingpong); returns true if the item
// side's current values ended up in Q_target / item_bias_target (odd number of swaps)
bool sgd_update_pingpong(const int *indptr, const int *indices, const float *data, int n_rows, int n_cols, float *P,
                         int ldp, float *Q, float *Q_target, int ldq, float *user_bias, float *item_bias,
                         float *item_bias_target, unsigned long long *claim, float global_bias, int n_factors,
                         const cu2rec_hyper &hyper, uint64_t seed, uint64_t iter0, int n_iters, int update_items,
                         int user_offset, bool swap_last, hipStream_t stream);

// [nnz] x {item, rating} side by side for SgdArgs::pairs (8 bytes per rating, device memory)
void sample_pairs_build(const int *indices, const float *data, int nnz, void *pairs, hipStream_t stream);

void sgd_update_ordered(OrderedSchedule &schedule, const int *indptr, const int *indices, const float *data,
                        int n_rows, int n_cols, float *P, int ldp, float *Q, int ldq, float *user_bias,
                        float *item_bias, float global_bias, int n_factors, const cu2rec_hyper &hyper, uint64_t seed,
                        uint64_t iter0, int n_iters, int update_items, int user_offset, hipStream_t stream,
                        bool blocksolve = false);

void loss(const int *indptr, const int *indices, const float *data, int n_rows, int nnz, const float *P, int ldp,
          const float *Q, int ldq, const float *user_bias, const float *item_bias, float global_bias, int n_factors,
          float *errors_out, void *workspace, double *sum_abs, double *sum_sq, float *mae, float *rmse,
          hipStream_t stream);

void error_metrics(const float *errors, int n, void *workspace, float *mae, float *rmse, hipStream_t stream);

// Device CSR in the reference's layout (matrix.h:11-19): int32 indptr / indices, float32 data.
struct DeviceCsr {
    DeviceCsr(int rows, int cols, int nnz, const int *h_indptr, const int *h_indices, const float *h_data);
    DeviceBuffer<int> indptr, indices;
    DeviceBuffer<float> data;
    int rows, cols, nnz;
    int max_item;
    int users_with_ratings;
    mutable std::unique_ptr<OrderedSchedule> schedule;  // created on first CU2REC_SGD_ORDERED use
    mutable DeviceBuffer<uint2> pairs;                   // sample array, created on first Hogwild use (SgdArgs::pairs)
};

// P, Q, biases on the device with padded rows (see include/cu2rec_amd.h "Device data layout").
struct DeviceModel {
    DeviceModel(int rows, int cols, int n_factors, const float *P, const float *Q, const float *user_bias,
                const float *item_bias, float global_bias);
    void download(float *P, float *Q, float *user_bias, float *item_bias) const;
    // defer_last_swap (CU2REC_SGD_PINGPONG only): leave out the last iteration's swap; finish_swap() does it later
    // user_offset: global id of row 0 when the model holds one shard of a user-sharded set (enters the sampler only)
    void sgd(const DeviceCsr &train, const cu2rec_hyper &h, uint64_t seed, uint64_t iter0, int n_iters, int mode,
             int update_items, hipStream_t stream, bool defer_last_swap = false, int user_offset = 0);
    void finish_swap();
    void loss(const DeviceCsr &ratings, double *sum_abs, double *sum_sq, float *mae, float *rmse,
              hipStream_t stream) const;
    DeviceBuffer<float> P, Q, user_bias, item_bias;
    DeviceBuffer<float> Q_other, item_bias_other;  // CU2REC_SGD_PINGPONG: the second buffer pair, created on first use
    DeviceBuffer<unsigned long long> claim;
    bool other_in_sync = false;   // Q_other / item_bias_other were last written by the ping-pong mode itself
    bool swap_pending = false;
    mutable DeviceBuffer<double> workspace;
    int rows, cols, n_factors;
    int ld;   // row stride of P in floats: n_factors rounded up to 4 (16-byte aligned rows)
    int ldq;  // row stride of Q: rounded up to 32 floats, so that every item row is a whole number of 128-byte lines --
              // item rows are the ones several XCDs read AND write; rows that share a line cost extra coherence misses
              // and partial-line write-backs (resident launches: 18.0 -> 17.6 us per iteration at f = 100)
    float global_bias;
};

DeviceCsr &unwrap(cu2rec_csr *m);
const DeviceCsr &unwrap(const cu2rec_csr *m);
DeviceModel &unwrap(cu2rec_model *m);
const DeviceModel &unwrap(const cu2rec_model *m);

}  // namespace cu2rec
