// Launch interface of the block-solve SGD mode (CU2REC_SGD_BLOCKSOLVE): sequential semantics
// (mf_sequential.cu:102-143) with the long item chains solved block-wise; see blocksolve.hip.
#pragma once

#include <string>

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "kernels.hpp"

namespace cu2rec {

constexpr int kBsLinks = 64;                        // updates ("links") of a chain per block
constexpr int kBsTableStride = 2 * kBsLinks + 1;    // tables: 1 - a^k, 1 - c^k, a^k, c^k for k = 0..2 kBsLinks (a block and its look-ahead)
constexpr int kBsTableFloats = 4 * kBsTableStride;
constexpr int kBsFactorFloats = 3 * 32 * 32;        // inverse factor of a block: tiles M11, M21, M22 (32 x 32 each, row major)
constexpr int kBsRecFloats = kBsFactorFloats + kBsLinks;  // a block's record from phase 1: the factor, then r - gb - ub per link
constexpr int kBsCrossFloats = kBsLinks * kBsLinks;  // look-ahead chains: the 64 x 64 block of lr L that couples a block (rows) to the one before it (columns)
constexpr int kBsMaxSlots = 63;                     // float4 slots per row the solver has LDS for (n_factors <= 252)

struct BsBlockDesc {  // up to 64 consecutive links of one hot chain
    int pos0;      // sorted position of the first link
    int n_valid;   // links in the block (the last block of a chain may be partial); 0 = unused table entry
    int chain;     // popularity rank of the item
    int m;         // block number inside the chain
};

struct BsChainDesc {  // the updates of one iteration that hit one hot item, users ascending
    int begin, len;  // sorted positions [begin, begin + len)
    int blk0;        // index of the chain's first block
    int item;
};

// One iteration's view of the sorted schedule (ordered.hip) and of the scratch buffers; all device pointers.
struct BsIteration {
    const uint32_t *keys;     // [n_active] sorted keys of this iteration
    const uint64_t *vals;     // [n_active] user << 32 | rating bits
    int n_active;
    int n_hot;                // popularity ranks [0, n_hot) are solved block-wise, the others walked link by link
    uint32_t item_mask;
    const BsChainDesc *chains;  // [n_hot]
    const BsBlockDesc *blocks;  // [max_blocks]
    const int *walk_begin;      // [1] sorted position where the walked chains start
    const int *item_of_rank;
    const float *tables;      // [kBsTableFloats]
    float log2a, log2c;       // log2(1 - lr * Q_reg), log2(1 - lr * item_bias_reg)
    float *Mbuf;              // [max_blocks][kBsRecFloats]  (I + lr L)^-1 of each block | r - gb - ub of its links
    float *ebuf;              // [max_blocks][kBsLinks]      error of each hot link
    float *qstart;            // [max_blocks][ldq]           item row at the start of each block
    int max_blocks;
    // workgroups phase 3 is LAUNCHED with (<= max_blocks; phase 1 keeps the full grid: blocksolve.hip): what an iteration is expected to have, mean + 8 sigma of the hot links
    // (OrderedSchedule::blocks_bound); the block table is dense and the workgroups stride through it, so a busier iteration is still
    // done completely
    int launch_blocks;
    // Look-ahead chains (blocksolve.hip, "look-ahead form"): popularity ranks [0, la_ranks) -- the long chains, whose length IS the
    // iteration's critical path.  Phase 1 also builds the block of lr L that couples each of their blocks to the one before it
    // (la_grid more workgroups in its launch), and phase 2 runs them with the item row OFF the dependent path.
    int la_ranks;
    int la_cap;               // ... whose blocks all lie below la_cap (Nbuf has room for that many; a chain beyond it runs in the plain form)
    int la_grid;              // workgroups of phase 1's launch that build the cross blocks
    float *Nbuf;              // [la_cap][kBsCrossFloats]
    unsigned *status;                 // [1] set by a wait that gave up (bounded spins); the host reports it
    // Fork and join of the side stream without events (ordered.hip): phase 1's workgroups count themselves through and phase 2's
    // count themselves in -- ONE wavefront queued in front of the side kernel (bs_launch_gate) ends when both counts have reached
    // what the host has launched so far (timing only, no data behind it); a signal kernel behind the side kernel stores the
    // iteration's number, and one extra workgroup of phase 3's launch waits for it (data: the next phase 1 reads rows the side kernel
    // writes).
    unsigned long long *solve_started;   // [1]  phase-2 workgroups started, over all iterations so far
    const unsigned long long *side_seq;  // (or null) the word the signal kernel sets ...
    unsigned long long side_target;      // ... and the number the extra workgroup of phase 3's launch waits for
    unsigned long long *gram_done;       // [32 x 16]  phase-1 workgroups through (32 shards, 128 bytes apart), over all iterations so far (or null)
    // development aid (cu2rec_debug_blocksolve_stamps): [0] = entries appended so far, then {kernel, id, start, end}
    // per wavefront in units of the 100 MHz s_memrealtime clock; nullptr in normal operation
    unsigned long long wait_ticks;    // bound of the device-side waits (bs_wait_ticks(): CU2REC_BS_WAIT_S, 2 s), 100 MHz ticks
    unsigned long long *stamps;
    int stamps_cap;
};

// process-wide stamp buffer for the launches that follow (nullptr = off)
void bs_set_stamps(unsigned long long *buf, int cap);
void bs_get_stamps(unsigned long long **buf, int *cap);
unsigned long long bs_wait_ticks();
bool bs_supported(int nslots);
bool bs_lookahead_supported(int nslots);  // rows of at most 31 float4 slots: the look-ahead form's rings fit the LDS
int bs_compute_units();       // of the current device
// bounded device-side waits report through one status word per device: its address, an asynchronous refresh of the host copy
// behind a call's launches, and the check (throws CU2REC_EHIP once if a wait gave up)
unsigned *bs_status_word();
void bs_report_status(hipStream_t stream);
void bs_check_fault();
// the fork / join topology of this device's iterations (blocksolve.hip, "the launch topology"): decided on first use -- environment,
// counter pass, or a two-stream handshake probe on the iterations' own streams -- and switched to events by a join that gave up
constexpr int kBsTopoEvents = 0, kBsTopoDevice = 2;
int bs_topology(hipStream_t stream, hipStream_t side);
int bs_topology_query(std::string *why);  // -1: not decided yet
void bs_launch_tables(const SgdHyper &h, float *tables, hipStream_t stream);
// one workgroup per iteration of the batch: chain and block descriptors
void bs_launch_plan(const uint32_t *keys, int n_active, int n_batch, int n_hot, int item_bits, int max_blocks,
                    const int *item_of_rank, int *chain_begin, BsChainDesc *chains, BsBlockDesc *blocks, int *walk_begin,
                    hipStream_t stream, size_t stride);
// the three phases of one iteration's hot chains: three launches queued one behind the other on one stream
void bs_launch_gram(const SgdArgs &a, const BsIteration &it, hipStream_t stream, hipEvent_t stop = nullptr);
void bs_launch_solve(const SgdArgs &a, const BsIteration &it, hipStream_t stream);
// one wavefront that ends once `count` has reached `target` (bounded): queued in front of a kernel that should not start before then
// one thread that stores `value` to `word` (write-through): queued behind a kernel, it announces that kernel's completion
void bs_launch_signal(unsigned long long *word, unsigned long long value, hipStream_t stream);
void bs_launch_gate(const unsigned long long *count, unsigned long long target, const unsigned long long *started,
                    unsigned long long started_target, hipStream_t stream);
void bs_launch_update(const SgdArgs &a, const BsIteration &it, hipStream_t stream);

}  // namespace cu2rec
