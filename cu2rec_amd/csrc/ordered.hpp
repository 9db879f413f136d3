// Workspace + driver of the ordered (sequential-semantics, deterministic) SGD mode; see ordered.hip.
#pragma once

#include <vector>

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "hip_check.hpp"
#include "blocksolve.hpp"
#include "kernels.hpp"

namespace cu2rec {

struct OrderedSchedule {
    // d_indptr / d_indices: the device CSR the schedule will be used with (read once, on the host, to rank the
    // items by popularity and count the users that have ratings).
    OrderedSchedule(const int *d_indptr, const int *d_indices, int n_rows, int n_cols, int nnz);
    // Runs iterations [iter0, iter0 + n_iters) of `a` (pointers, hyper-parameters, seed, user_offset filled in).
    // blocksolve: the chains of the hot items are solved block-wise (blocksolve.hip) instead of walked link by link:
    // the same sequential semantics up to float rounding, not bit for bit.
    void run(SgdArgs a, uint64_t iter0, int n_iters, hipStream_t stream, bool blocksolve = false);

    int n_rows, n_cols, nnz;
    int n_active = 0;   // users with at least one rating = updates per iteration
    int item_bits = 0;  // key = iteration_in_batch << item_bits | popularity_rank(item)
    int max_batch = 1;  // iterations scheduled (sampled + sorted) per pass
    DeviceBuffer<int> item_rank, item_of_rank;
    // Two schedule slots: the sample stream is value independent, so batch j + 1 is sampled and sorted on a stream of
    // its own while batch j's iterations run.  Per slot: the radix sort's two key / value buffers.
    DeviceBuffer<uint32_t> keys[2][2];
    DeviceBuffer<uint64_t> vals[2][2];
    hipStream_t sched = nullptr;
    hipEvent_t ev_ready[2] = {nullptr, nullptr}, ev_consumed[2] = {nullptr, nullptr};
    bool slot_used[2] = {false, false};
    // Schedule the next call's first batch ahead (it reads the CSR arrays after run() has returned): only where the
    // schedule lives and dies with the CSR it reads, i.e. the owned-object layer (cu2rec_csr), whose arrays are IMMUTABLE for the
    // object's lifetime -- windows are keyed on (seed, user offset, indptr address, mode), not on the arrays' contents; off for raw
    // pointers, whose windows never outlive the call.
    bool speculate = false;
    // A scheduled WINDOW of iterations per slot: [iter_begin, iter_begin + nb) of one (seed, user offset, CSR, mode).  A call runs the
    // iterations it asks for out of whichever window holds them, at whatever offset -- so calls shorter than a window (a bench's
    // 20 steps, a driver's periods) share one schedule of max_batch iterations instead of paying for a schedule each.
    struct Window {
        bool valid = false;
        uint64_t seed = 0, iter_begin = 0;
        int nb = 0, user_offset = 0;
        const int *indptr = nullptr;
        bool blocksolve = false;
        const uint32_t *sk = nullptr;
        const uint64_t *sv = nullptr;
    } win[2];
    DeviceBuffer<unsigned char> temp;
    DeviceBuffer<int> seg_offsets;  // [max_batch + 1] b * n_rows: the iterations of a batch as segments of the sort
    size_t temp_bytes = 0;
    // block-solve mode: items are ranked by their expected updates per iteration (sum over raters of 1 / degree);
    // ranks [0, n_hot_bs) -- at least blocksolve_min_rate() expected updates -- get the block-wise treatment
    int n_hot_bs = 0;
    int n_duo_bs = 0;        // ranks [n_hot_bs, n_duo_bs): the ordered mode's two-wave chains, beside the block solves
    int max_blocks = 0;      // blocks of 64 links (kBsLinks) per iteration, upper bound
    int la_ranks = 0;        // ranks [0, la_ranks): phase 2 in the look-ahead form (blocksolve.hip); their blocks lie below la_cap
    int la_cap = 0;
    DeviceBuffer<float> Nbuf;  // [la_cap][kBsCrossFloats] the cross blocks of the look-ahead chains
    int qstart_ld = 0;
    SgdHyper tables_for{};   // hyper-parameters the decay tables were computed from
    bool tables_valid = false;
    DeviceBuffer<int> chain_begin[2], walk_begin[2];
    // per schedule slot: [iteration of the batch][rank 0 .. n_range_ranks] first sorted position of the chains of the most
    // popular items (chain_ranges_kernel, behind the sort): the two-wave blocks read their chain's range instead of searching
    DeviceBuffer<int> chain_ranges[2];
    int n_range_ranks = 0;
    // rank_rate_prefix[r] = expected updates per iteration of the r most popular items together (host; [n_cols + 1]): what the walk
    // behind rank r has to cover is n_active minus that, give or take its square root (walk_bound)
    std::vector<double> rank_rate_prefix;
    int walk_bound(int n_hot) const;
    int blocks_bound() const;  // workgroups phase 3 of a block-solve iteration is launched with (BsIteration::launch_blocks)
    DeviceBuffer<BsChainDesc> bs_chains[2];
    DeviceBuffer<BsBlockDesc> bs_blocks[2];
    DeviceBuffer<float> tables, Mbuf, ebuf, qstart;
    // Block-solve mode: an iteration is four launches on two streams -- `stream` (the caller's): phases 1, 2, 3; `upd`: the other
    // items' chains (two-wave form + walk) beside phases 2 and 3, forked behind phase 1 and joined in front of the next phase 1
    // without events on the main stream (OrderedSchedule::run, blocksolve.hpp)
    DeviceBuffer<unsigned long long> solve_started, gram_done;
    unsigned long long gram_done_target = 0;  // phase-1 workgroups launched so far (what gram_done will reach)
    DeviceBuffer<unsigned long long> side_seq;  // [1] number of the last iteration whose side kernel is complete (bs_launch_signal)
    unsigned long long side_seq_host = 0;
    unsigned long long started_total = 0;  // phase-2 workgroups launched so far (what solve_started will reach)
    hipStream_t upd = nullptr;
    hipEvent_t ev_call = nullptr, ev_upd = nullptr, ev_gram = nullptr;
    // one stream at a time per schedule: a call on another stream than the last one first waits for that call's end
    hipEvent_t ev_last = nullptr;
    hipStream_t last_stream = nullptr;
    bool have_last = false;
    ~OrderedSchedule();
};

// minimum expected updates per iteration of an item for block-wise treatment (process-wide; schedules created later)
// rate > 0 sets it, rate < 0 returns to automatic (scaled with the set), 0 queries; returns the explicit value in force before
// the call or -1 for automatic
float blocksolve_min_rate(float rate);
float blocksolve_min_rate_base();   // the explicit value, or the default the automatic scaling starts from
bool blocksolve_min_rate_is_set();  // by the caller or the environment; otherwise a schedule scales the default with its set

// chains EXPECTED to be at least this many blocks of 64 links long run phase 2 in the look-ahead form (default 24: the top chains; 0: none;
// schedules created later); blocks < 0 only queries
int blocksolve_lookahead_blocks(int blocks);

}  // namespace cu2rec
