// Workspace + driver of the ordered (sequential-semantics, deterministic) SGD mode; see ordered.hip.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "hip_check.hpp"
#include "kernels.hpp"

namespace cu2rec {

struct OrderedSchedule {
    // d_indptr / d_indices: the device CSR the schedule will be used with (read once, on the host, to rank the
    // items by popularity and count the users that have ratings).
    OrderedSchedule(const int *d_indptr, const int *d_indices, int n_rows, int n_cols, int nnz);
    // Runs iterations [iter0, iter0 + n_iters) of `a` (pointers, hyper-parameters, seed, user_offset filled in).
    void run(SgdArgs a, uint64_t iter0, int n_iters, hipStream_t stream);

    int n_rows, n_cols, nnz;
    int n_active = 0;   // users with at least one rating = updates per iteration
    int item_bits = 0;  // key = iteration_in_batch << item_bits | popularity_rank(item)
    int max_batch = 1;  // iterations scheduled (sampled + sorted) per pass
    DeviceBuffer<int> item_rank, item_of_rank;
    DeviceBuffer<uint32_t> keys[2];
    DeviceBuffer<uint64_t> vals[2];
    DeviceBuffer<unsigned char> temp;
    size_t temp_bytes = 0;
};

}  // namespace cu2rec
