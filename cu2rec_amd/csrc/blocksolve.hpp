// Launch interface of the block-solve SGD mode (CU2REC_SGD_BLOCKSOLVE): sequential semantics
// (mf_sequential.cu:102-143) with the long item chains solved block-wise; see blocksolve.hip.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "kernels.hpp"

namespace cu2rec {

constexpr int kBsLinks = 64;                        // updates ("links") of a chain per block
constexpr int kBsTableStride = kBsLinks + 1;        // tables: 1 - a^k, 1 - c^k, a^k, c^k for k = 0..kBsLinks
constexpr int kBsTableFloats = 4 * kBsTableStride;
constexpr int kBsFactorFloats = 3 * 32 * 32;        // inverse factor of a block: tiles M11, M21, M22 (32 x 32 each, row major)
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
    float *Mbuf;              // [max_blocks][kBsFactorFloats]  (I + lr L)^-1 of each block
    float *base;              // [n_active]  r - gb - ub per hot link
    float *ebuf;              // [n_active]  error of each hot link
    float *qstart;            // [max_blocks][ldq]  item row at the start of each block
    int max_blocks;
    // long chains (at least aff_min_blocks blocks; 0 = none): every block's effect on the item state (row, bias) as one
    // affine map built in phase 1, so that phase 2 is ONE mat-vec per block (blocksolve.hip, "affine form")
    int aff_min_blocks;
    float *Tbuf;              // [max_blocks][bs_affine_t_floats]  T' = H^T W, (4 nslots + 1) rows of bs_affine_ts floats
    float *Wbuf;              // [max_blocks][64][bs_affine_ts]    W = M [G | r']
    float *bstart;            // [max_blocks]  item bias at the start of each block
    // L2 warm-up beside the chains (bs_solve_kernel): pf_helpers workgroups for each of the pf_chains longest chains of
    // every XCD read the chain's user rows and factors ahead of its solver; 0 = none
    int pf_helpers, pf_chains;
    // development aid (cu2rec_debug_blocksolve_stamps): [0] = entries appended so far, then {kernel, id, start, end}
    // per wavefront in units of the 100 MHz s_memrealtime clock; nullptr in normal operation
    unsigned long long *stamps;
    int stamps_cap;
};

// process-wide stamp buffer for the launches that follow (nullptr = off)
void bs_set_stamps(unsigned long long *buf, int cap);
void bs_get_stamps(unsigned long long **buf, int *cap);
bool bs_supported(int nslots);
// the affine form needs the state (row, bias) and the constant column inside 128 columns: n_factors <= 124
__host__ __device__ inline bool bs_affine_supported(int nslots) { return nslots <= 31; }
// floats per row of T' and W: the columns 4 nslots + 2, in 16-byte pieces, an odd number of them (LDS bank spread)
__host__ __device__ inline int bs_affine_ts(int nslots) { return 4 * ((nslots + 1) | 1); }
__host__ __device__ inline size_t bs_affine_t_floats(int nslots) { return static_cast<size_t>(4 * nslots + 1) * bs_affine_ts(nslots); }
__host__ __device__ inline size_t bs_affine_w_floats(int nslots) { return static_cast<size_t>(kBsLinks) * bs_affine_ts(nslots); }
void bs_launch_tables(const SgdHyper &h, float *tables, hipStream_t stream);
// one workgroup per iteration of the batch: chain and block descriptors
void bs_launch_plan(const uint32_t *keys, int n_active, int n_batch, int n_hot, int item_bits, int max_blocks,
                    const int *item_of_rank, int *chain_begin, BsChainDesc *chains, BsBlockDesc *blocks, int *walk_begin,
                    hipStream_t stream);
// the three phases of one iteration's hot chains, in this order on one stream
void bs_launch_gram(const SgdArgs &a, const BsIteration &it, hipStream_t stream);
void bs_launch_solve(const SgdArgs &a, const BsIteration &it, hipStream_t stream);
void bs_launch_update(const SgdArgs &a, const BsIteration &it, hipStream_t stream);



}  // namespace cu2rec
