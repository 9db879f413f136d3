// Launch interface of the block-solve SGD mode (CU2REC_SGD_BLOCKSOLVE): sequential semantics
// (mf_sequential.cu:102-143) with the long item chains solved block-wise; see blocksolve.hip.
#pragma once

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
constexpr int kBsProgWords = 4;                     // 8-byte words of a chain's progress record (32 bytes)
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
    // Look-ahead chains (blocksolve.hip, "look-ahead form"): popularity ranks [0, la_ranks) -- the long chains, whose length IS the
    // iteration's critical path.  Phase 1 also builds the block of lr L that couples each of their blocks to the one before it
    // (la_grid more workgroups in its launch), and phase 2 runs them with the item row OFF the dependent path.
    int la_ranks;
    int la_cap;               // ... whose blocks all lie below la_cap (Nbuf has room for that many; a chain beyond it runs in the plain form)
    int la_grid;              // workgroups of phase 1's launch that build the cross blocks
    float *Nbuf;              // [la_cap][kBsCrossFloats]
    // Long chains (at least aff_min_blocks blocks; 0 = none; sequential topology only): every block's effect on the item state
    // (row, bias) as ONE affine map built in phase 1, so that phase 2 is one mat-vec per block with no meeting point
    // (blocksolve.hip, "affine form")
    int aff_min_blocks;
    int aff_head;             // blocks a long chain runs in the plain form while the maps of its other blocks are being built
    int aff_cap;              // blocks [0, aff_cap) may take the form: phase 2's launch has a workgroup for each of them
    unsigned *aff_flag;       // [max_blocks] epoch of the iteration whose T' / W of the block are in memory
    int aff_tails;            // chains [0, aff_tails) may take the form: the launch has a second workgroup for each of them
    unsigned long long *hstate;  // [n_hot][128] the item's row and bias behind the plain head for the chain's second workgroup: {epoch, value}
    float *Tbuf;              // [max_blocks][bs_affine_t_floats]  T' = H^T W, (4 nslots + 1) rows of bs_affine_ts floats (of 128)
    float *Wbuf;              // [max_blocks][64][bs_affine_ts]    W = M [G | r']
    float *bstart;            // [max_blocks]  item bias at the start of each block
    // Hand-over between the three phases, which run as three launches AT THE SAME TIME (ordered.hip): every word carries the
    // iteration's epoch, so nothing is ever reset.
    int concurrent;                   // 1: the phases run as launches side by side and wait for each other through the words
                                      // below; 0 (default): they are queued one behind the other and none of this is used
    unsigned epoch;                   // of this iteration (never 0)
    unsigned *gram_flag;              // [max_blocks]  == epoch: phase 1 has written block g's record (write-through)
    unsigned long long *chain_prog;   // [n_hot][kBsProgWords]  epoch << 32 | blocks of the chain whose errors / start rows have left phase 2 (a look-ahead
                                      //               chain: one word for each of the three wavefronts that store them)
    unsigned *status;                 // [1]           set by a wait that gave up (bounded spins); the host reports it
    // Phase 3's workgroups WAIT on the device, so phase 2's must hold their CUs before phase 3 is dispatched (a CU filled with
    // waiting phase-3 workgroups has no room for the chain they wait for).  Every phase-2 workgroup counts itself in when
    // it starts; the last workgroup of phase 1's grid leaves only when the count has reached this iteration's target --
    // and phase 3 is queued behind phase 1 on the same stream.
    // Pipelined topology (default on a device that runs the streams side by side, ordered.hip): phase 1, then phase 2 with phase 3 BESIDE it --
    // a launch of `pipe_grid` persistent workgroups on another stream that take the blocks in `order` and wait, block by block, for the
    // chains' progress words.  Phase 2's launch has one more workgroup that ends when phase 3's workgroups (pipe_done) and the side
    // stream's kernel (side_seq) are through: the next phase 1 follows it without an event.
    int pipe;
    const int *order;                    // [max_blocks] block indices, by block number inside the chain first
    unsigned long long *pipe_done;       // [1]  phase-3 workgroups through, over all iterations so far
    unsigned long long pipe_target;      //      ... including all of this iteration's
    unsigned long long *solve_started;   // [1]  phase-2 workgroups started, over all iterations so far
    const unsigned long long *side_seq;  // (or null) the word a signal kernel behind the side kernel sets to the iteration's number ...
    unsigned long long side_target;      // ... and the number the extra workgroup of the main stream's last launch waits for: the join without an event
    unsigned long long *gram_done;       // [32 x 16]  phase-1 workgroups through (32 shards, 128 bytes apart), over all iterations so far (or null): what the side
                                         //      stream's gate kernel waits for (bs_launch_gate) -- timing only, no data behind it
    unsigned long long started_target;   //      ... including all of this iteration's
    // development aid (cu2rec_debug_blocksolve_stamps): [0] = entries appended so far, then {kernel, id, start, end}
    // per wavefront in units of the 100 MHz s_memrealtime clock; nullptr in normal operation
    unsigned long long wait_ticks;    // bound of the device-side waits (bs_wait_ticks(): CU2REC_BS_WAIT_S, 2 s), 100 MHz ticks
    unsigned long long *stamps;
    int stamps_cap;
    int dbg;  // CU2REC_BS_DBG, timing experiments only (results undefined): 1 plain stores in phase 1, 2 phase 1 does not wait for
              // phase 2's start, 4 phase 2 does not wait for records, 8 phase 3 does not wait for progress; fault-path tests
              // (tests/test_gpu_blocksolve.py): 16 the side stream's signal is never sent, 32 the side stream's gate can never open
};

// process-wide stamp buffer for the launches that follow (nullptr = off)
void bs_set_stamps(unsigned long long *buf, int cap);
void bs_get_stamps(unsigned long long **buf, int *cap);
unsigned long long bs_wait_ticks();
bool bs_supported(int nslots);
bool bs_lookahead_supported(int nslots);  // rows of at most 31 float4 slots: the look-ahead form's rings fit the LDS
inline bool bs_pipe_supported(int nslots) { return nslots <= 32; }  // the pipelined topology: n_factors <= 128
int bs_compute_units();       // of the current device
int bs_solve_grid(int n_hot);  // workgroups of phase 2: one per chain, at most half the CUs of the current device
// bounded device-side waits report through one status word per device: its address, an asynchronous refresh of the host copy
// behind a call's launches, and the check (throws CU2REC_EHIP once if a wait gave up)
unsigned *bs_status_word();
void bs_report_status(hipStream_t stream);
void bs_check_fault();
// the affine form needs the state (row, bias) and the constant column inside 128 columns: n_factors <= 124
__host__ __device__ inline bool bs_affine_supported(int nslots) { return nslots <= 31; }
// floats per row of T': the columns 4 nslots + 2, in 16-byte pieces, an odd number of them (LDS bank spread)
__host__ __device__ inline int bs_affine_ts(int nslots) { return 4 * ((nslots + 1) | 1); }
// (room for 128 rows: phase 1 stores whole accumulator tiles, the rows past 4 nslots + 1 are never read)
__host__ __device__ inline size_t bs_affine_t_floats(int nslots) { return static_cast<size_t>(128) * bs_affine_ts(nslots); }
// floats per row of W: the columns 4 nslots + 2, in 16-byte pieces
__host__ __device__ inline int bs_affine_ws(int nslots) { return 4 * (nslots + 1); }
__host__ __device__ inline size_t bs_affine_w_floats(int nslots) { return static_cast<size_t>(kBsLinks) * bs_affine_ws(nslots); }
void bs_launch_tables(const SgdHyper &h, float *tables, hipStream_t stream);
// one workgroup per iteration of the batch: chain and block descriptors
void bs_launch_plan(const uint32_t *keys, int n_active, int n_batch, int n_hot, int item_bits, int max_blocks,
                    const int *item_of_rank, int *chain_begin, BsChainDesc *chains, BsBlockDesc *blocks, int *walk_begin,
                    hipStream_t stream, size_t stride, bool batch_keys, int *order);
// the three phases of one iteration's hot chains: three launches that may run at the same time on three streams (phase 2 waits
// for phase 1's records block by block, phase 3 for phase 2's progress chain by chain, through the words above)
void bs_launch_gram(const SgdArgs &a, const BsIteration &it, hipStream_t stream, hipEvent_t stop = nullptr);
void bs_launch_solve(const SgdArgs &a, const BsIteration &it, hipStream_t stream);
// one wavefront that ends once `count` has reached `target` (bounded): queued in front of a kernel that should not start before then
// one thread that stores `value` to `word` (write-through): queued behind a kernel, it announces that kernel's completion
void bs_launch_signal(unsigned long long *word, unsigned long long value, hipStream_t stream);
void bs_launch_gate(const unsigned long long *count, unsigned long long target, const unsigned long long *started,
                    unsigned long long started_target, hipStream_t stream);
void bs_launch_update(const SgdArgs &a, const BsIteration &it, hipStream_t stream);
// the pipelined topology's phase 3: `grid` persistent workgroups (the caller queues it behind a gate: phase 2's workgroups hold their CUs)
void bs_launch_update_pipe(const SgdArgs &a, const BsIteration &it, int grid, hipStream_t stream);



}  // namespace cu2rec
