// Launch interface between the host code and kernels.hip.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "common.hpp"

namespace cu2rec {

// Hyper-parameters travel by value in the kernel argument block (SGPRs on CDNA) instead of the
// reference's __constant__ symbols (config.h:9-18, config.cu:24-35).
struct SgdHyper {
    float lr, p_reg, q_reg, ub_reg, ib_reg;
};

struct SgdArgs {
    const int *indptr;    // [n_rows + 1]
    const int *indices;   // [nnz]
    const float *data;    // [nnz]
    float *P;             // [n_rows x ldp]
    float *Q;             // [n_cols x ldq]
    float *user_bias;     // [n_rows]
    float *item_bias;     // [n_cols]
    int n_rows;
    int ldp, ldq;         // row strides in floats, multiples of 4
    int nslots;           // float4 slots per row = ceil(n_factors / 4)
    float global_bias;
    SgdHyper h;
    uint64_t seed;
    uint64_t iter0;       // global iteration number of the first update
    int iters;            // updates per user in this launch
    int update_items;     // 0: Q and item_bias frozen (is_train == false)
    int user_offset;      // global id of row 0 (user-sharded runs); enters the sampler only
    // optional sample array: [nnz] x {int32 item, float32 rating}, i.e. indices[k] and data[k] side by side, so that a
    // draw is ONE 8-byte gather instead of two 4-byte gathers in different arrays (used by the resident launches,
    // where the draws are a third of the random accesses); nullptr: gather from indices / data
    const uint2 *pairs;
};

struct LossArgs {
    const int *indptr;
    const int *indices;
    const float *data;
    const float *P;
    const float *Q;
    const float *user_bias;
    const float *item_bias;
    int n_rows, nnz;
    int ldp, ldq, nslots;
    float global_bias;
    float *errors_out;    // [nnz] or nullptr
    double *partials;     // [2 * blocks]: per block {sum |e|, sum e^2}
};

// CU2REC_SGD_PINGPONG: the second item-side buffer pair and the first-writer claims (sgd.cu:22-75's Q_target,
// item_bias_target, item_is_updated)
struct PingPongArgs {
    float *Q_target;
    float *item_bias_target;
    unsigned long long *claim;  // [n_cols]: (~iteration << 32 | thread index) of the first claimant, by atomicMin
};

constexpr int kMaxPartialBlocks = 4096;

int slots_per_lane(int nslots);
void launch_sgd(const SgdArgs &args, int mode, hipStream_t stream);
// one iteration (args.iter0) of CU2REC_SGD_PINGPONG: claim kernel + update kernel; args.Q / item_bias are only read
void launch_sgd_pingpong(const SgdArgs &args, const PingPongArgs &pp, hipStream_t stream);
int loss_blocks(int nnz);
void launch_loss(const LossArgs &args, int blocks, hipStream_t stream);
// {sum |e|, sum e^2} of the `blocks` per-block partial sums, on the device, into partials[2 * kMaxPartialBlocks ..]: the
// workspace of a loss pass is 2 * kMaxPartialBlocks + 2 doubles
void launch_partials_reduce(double *partials, int blocks, hipStream_t stream);
int error_metrics_blocks(int n);
void launch_error_metrics(const float *errors, int n, double *partials, int blocks, hipStream_t stream);
void launch_sample_pairs_build(const int *indices, const float *data, size_t nnz, uint2 *pairs, hipStream_t stream);
void launch_items_delta_pack(const float *Q, const float *ib, const float *Q_base, const float *ib_base, int n_cols,
                             int ldq, float *buf, hipStream_t stream);
void launch_items_delta_apply_overlapped(float *Q, float *ib, float *Q_base, float *ib_base, const float *Q_snap,
                                         const float *ib_snap, int n_cols, int ldq, const float *buf, float scale,
                                         hipStream_t stream);
void launch_items_delta_pack_weighted(const float *Q, const float *ib, const float *Q_base, const float *ib_base,
                                      const float *weight, int n_cols, int ldq, float *buf, hipStream_t stream);
void launch_items_delta_apply(float *Q, float *ib, float *Q_base, float *ib_base, int n_cols, int ldq,
                              const float *buf, float scale, hipStream_t stream);
// wire format of the exchange: n_cols * f row deltas then n_cols bias deltas, unpadded; weight (n_cols) may be null
void launch_items_wire_pack(const float *Q, const float *ib, const float *Q_base, const float *ib_base, const float *weight,
                            int n_cols, int f, int ldq, float *wire, hipStream_t stream);
void launch_items_wire_apply(float *Q, float *ib, float *Q_base, float *ib_base, int n_cols, int f, int ldq, const float *wire,
                             float scale, hipStream_t stream);

}  // namespace cu2rec
