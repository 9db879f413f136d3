// Ordered SGD: the sequential semantics of mf_sequential.cu:102-143 at GPU speed, deterministic and
// bit-identical to the CPU oracle (TREE16 dot order), with no races.
//
// Why it is possible: the sampler is counter based, so WHICH rating user x trains on at iteration i is a
// pure function of (seed, x, i) -- known before any value is computed.  Inside one iteration every user
// appears exactly once, so two updates conflict only through a shared ITEM row, and the sequential order
// matters only among the users that sampled the same item.  Hence, per iteration:
//     for every item in parallel:  apply the updates of the users that sampled it, in ascending user order
// is exactly the sequential result.  An iteration becomes a set of independent "item chains".
//
// Pipeline for a batch of B iterations (B <= kMaxBatch):
//   1. schedule_keys_tile_kernel  for every (iteration b, user x): draw the rating, emit
//        key = popularity_rank(item)                          (users without ratings: one bit above all ranks)
//        val = user << 32 | rating bits
//      (64 users x 64 iterations per workgroup, a user's rating row read once for all its samples)
//   2. stable LSB radix sort of the (key, val) pairs, every iteration a segment of its own (hipCUB/rocPRIM
//      DeviceSegmentedRadixSort; stability keeps users ascending inside a chain).  Afterwards iteration b occupies
//      [b * n_rows, b * n_rows + n_active), chains are runs of equal keys, and because keys carry the item's popularity
//      RANK, the longest chains come first.  (The first form -- one thread per sample, keys b << item_bits | rank, ONE sort
//      of the whole batch, iteration b at [b * n_active, ...) -- is kept behind CU2REC_SCHED_KEYS_TILE=0 /
//      CU2REC_SCHED_SEGMENTED=0: what it cost the iterations running beside it is in DESIGN.md section 4.)
//   3. sgd_ordered_kernel, one launch per iteration (the kernel boundary carries the P-row dependency from
//      iteration b to b+1).  The item row and item bias stay in registers for a whole chain (read and written once
//      per chain, not once per update).  Two block roles:
//        hot blocks      the chains of the 256 most popular items.  Their length (thousands of dependent updates) IS
//                        the iteration's critical path, so for ld <= 128 a block splits each chain over wavefronts:
//                        a compute wave runs only the dependent arithmetic, two memory waves prefetch the users' rows,
//                        publish them through LDS and apply / store the user-side updates (run_hot_block_duo).
//        regular blocks  a 16-lane group owns kWindow sorted positions and runs every chain that STARTS there, the
//                        users' rows fetched four links ahead.
// The arithmetic of one update is the same device code as the Hogwild kernel (sgd_device.hpp).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <hipcub/hipcub.hpp>
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <map>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

#include "hip_check.hpp"
#include "kernels.hpp"
#include "ordered.hpp"
#include "sampler.hpp"
#include "sgd_device.hpp"
#include "grid_barrier.hpp"
#include "resident.hpp"

namespace cu2rec {

namespace {

using namespace dev;

// The segmented sort of a batch's schedule (one segment per iteration, run()) is rocPRIM's with a block of 1,024 threads instead of
// its 256: ONE workgroup sorts a segment of n_rows pairs, and with the default block the kernel sat on its 64 CUs for 6.6 ms of
// every 15 on the Netflix shape (480,189 pairs a segment) -- a quarter of the chip that the throughput-bound phases of that shape
// then lack (iterations beside it 306 instead of 238 us).  Four times the threads: a quarter of the time.
#ifndef CU2REC_SEG_BLOCK
#define CU2REC_SEG_BLOCK 1024
#endif
#ifndef CU2REC_SEG_ITEMS
#define CU2REC_SEG_ITEMS 8
#endif
using SegSortConfig = rocprim::segmented_radix_sort_config<8, rocprim::kernel_config<CU2REC_SEG_BLOCK, CU2REC_SEG_ITEMS>,
                                                           rocprim::WarpSortConfig<8, 4, 256, 64, 16, 8, 256>, true>;

constexpr int kHotChains = 256;  // most popular items: their chains run in the two-wave form (run_hot_block_duo)

__global__ __launch_bounds__(kBlock) void schedule_keys_kernel(const int *__restrict__ indptr,
                                                               const int *__restrict__ indices,
                                                               const float *__restrict__ data,
                                                               const int *__restrict__ item_rank, int n_rows,
                                                               int n_batch, int item_bits, uint32_t sentinel,
                                                               uint64_t seed, uint64_t iter0, int user_offset,
                                                               uint32_t *__restrict__ keys, uint64_t *__restrict__ vals, int batch_keys) {
    const size_t total = static_cast<size_t>(n_rows) * n_batch;
    for (size_t idx = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; idx < total;
         idx += static_cast<size_t>(gridDim.x) * kBlock) {
        const int b = static_cast<int>(idx / n_rows);
        const int x = static_cast<int>(idx - static_cast<size_t>(b) * n_rows);
        const int low = indptr[x], high = indptr[x + 1];
        uint32_t key = sentinel;
        uint64_t val = 0;
        if (low != high) {
            const int y_i = sampler_index(seed, static_cast<uint64_t>(user_offset + x), iter0 + b, low, high);
            const int y = indices[y_i];
            key = (batch_keys ? static_cast<uint32_t>(b) << item_bits : 0u) | static_cast<uint32_t>(item_rank[y]);
            val = (static_cast<uint64_t>(static_cast<uint32_t>(x)) << 32) | __float_as_uint(data[y_i]);
        }
        keys[idx] = key;
        vals[idx] = val;
    }
}

// first position in keys[0, n) whose key is >= target
__device__ __forceinline__ int lower_bound_key(const uint32_t *__restrict__ keys, int n, uint32_t target) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (keys[mid] < target) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// The same keys and values, 64 users x up to 64 iterations per workgroup, with the users' RATING ROWS read once instead of one
// random access per sample: schedule_keys_kernel pulls two cache lines (indices, data) out of memory for every (user, iteration)
// -- 2.0 GB per batch of 64 iterations on the ML-20M shape, a seventh of what the 64 iterations themselves move, and the
// iterations running beside it pay for that share of the memory system (tools/schedule_interference.py).  Here a wavefront
// takes 16 users one after the other: lane b draws iteration b's sample, the row comes in with coalesced loads (256 ratings
// per round: most users' whole row), the lanes pick their ratings out of the registers (ds_bpermute); rows of more than
// kRowRounds rounds fall back to two gathers per sample.  The tile leaves through LDS, so that every store is a run of 64
// consecutive users of one iteration.
constexpr int kRowRounds = 8;       // x 256 ratings: rows up to 2,048 ratings are read whole
constexpr int kTileStride = 65;     // LDS words per iteration of the tile (64 users + 1: lanes write columns)

__global__ __launch_bounds__(256) void schedule_keys_tile_kernel(const int *__restrict__ indptr, const int *__restrict__ indices,
                                                                 const float *__restrict__ data, const int *__restrict__ item_rank,
                                                                 int n_rows, int n_batch, int item_bits, uint32_t sentinel, uint64_t seed,
                                                                 uint64_t iter0, int user_offset, uint32_t *__restrict__ keys,
                                                                 uint64_t *__restrict__ vals, int batch_keys) {
    __shared__ uint32_t t_key[64 * kTileStride];
    __shared__ uint32_t t_rat[64 * kTileStride];  // the rating's bits; the user (the value's upper half) is known at the store
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int x0 = static_cast<int>(blockIdx.x) * 64;
    const int b = lane;  // this lane's iteration
    // the 16 users' row bounds: lane l < 17 holds indptr[x0 + 16 wave + l]
    const int xi = min(x0 + 16 * wave + min(lane, 16), n_rows);
    const int bound = indptr[xi];
    // four users at a time: their first 256 ratings (most rows whole) are requested together, the samples drawn while the loads
    // fly, and their four item ranks looked up together -- a wavefront walking its 16 users one load round trip after the other
    // kept the kernel (a chip-filling grid beside the iterations) on the CUs 1.6 times as long
    for (int u0 = 0; u0 < 16; u0 += 4) {
        int low[4], high[4], y_i[4], y[4], vi[4][4];
        uint32_t r[4], vr[4][4];
        bool live[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = x0 + 16 * wave + u0 + j;
            low[j] = __shfl(bound, u0 + j);
            high[j] = __shfl(bound, u0 + j + 1);
            live[j] = x < n_rows && low[j] != high[j];  // wavefront uniform
            const int last = max(high[j] - 1, low[j]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = live[j] ? min(low[j] + 64 * k + lane, last) : 0;
                vi[j][k] = indices[e];
                vr[j][k] = __float_as_uint(data[e]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = x0 + 16 * wave + u0 + j;
            y_i[j] = live[j] && b < n_batch ? sampler_index(seed, static_cast<uint64_t>(user_offset + x), iter0 + b, low[j], high[j]) : low[j];
            y[j] = 0;
            r[j] = 0;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!live[j]) continue;  // wavefront uniform
            if (high[j] - low[j] <= 256 * kRowRounds) {
                for (int base = low[j]; base < high[j]; base += 256) {
                    if (base > low[j]) {  // (rows above 256 ratings: the further rounds one after the other)
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int e = min(base + 64 * k + lane, high[j] - 1);
                            vi[j][k] = indices[e];
                            vr[j][k] = __float_as_uint(data[e]);
                        }
                    }
                    const int pos = y_i[j] - base, src = pos & 63, kk = pos >> 6;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int yi = __shfl(vi[j][k], src);
                        const uint32_t ri = __shfl(vr[j][k], src);
                        if (kk == k && pos >= 0) y[j] = yi, r[j] = ri;
                    }
                }
            } else {
                y[j] = indices[y_i[j]];
                r[j] = __float_as_uint(data[y_i[j]]);
            }
        }
        int rank[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) rank[j] = item_rank[y[j]];  // (item 0 for users without ratings: unused)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t key = live[j] ? (batch_keys ? static_cast<uint32_t>(b) << item_bits : 0u) | static_cast<uint32_t>(rank[j]) : sentinel;
            t_key[b * kTileStride + 16 * wave + u0 + j] = key;
            t_rat[b * kTileStride + 16 * wave + u0 + j] = live[j] ? r[j] : 0u;
        }
    }
    __syncthreads();
    const int x = x0 + lane;
    if (x < n_rows) {
        for (int bb = wave; bb < n_batch; bb += 4) {
            const uint32_t key = t_key[bb * kTileStride + lane];
            const size_t idx = static_cast<size_t>(bb) * n_rows + x;
            keys[idx] = key;
            vals[idx] = (key & sentinel) ? 0ull : (static_cast<uint64_t>(static_cast<uint32_t>(x)) << 32) | t_rat[bb * kTileStride + lane];
        }
    }
}

// Where the chains of the n_ranks most popular items begin in every iteration of a sorted batch: ranges[b][r] = first sorted
// position of iteration b whose popularity rank is >= r (r = 0 .. n_ranks).  Runs behind the sort on the schedule stream, so
// that a two-wave block finds its chains with two loads instead of two binary searches of ~17 dependent loads each at the
// head of every iteration (6-10 us of its ~50: the searches were the first thing the iteration's longest chains did).
__global__ __launch_bounds__(kBlock) void chain_ranges_kernel(const uint32_t *__restrict__ keys, int n_active, int n_ranks, int item_bits,
                                                              int *__restrict__ ranges, size_t stride, int batch_keys) {
    const int b = blockIdx.x;
    const uint32_t *kb = keys + static_cast<size_t>(b) * stride;
    const uint32_t key_base = batch_keys ? static_cast<uint32_t>(b) << item_bits : 0u;
    for (int r = threadIdx.x; r <= n_ranks; r += kBlock)
        ranges[static_cast<size_t>(b) * (n_ranks + 1) + r] = lower_bound_key(kb, n_active, key_base + static_cast<uint32_t>(r));
}

// ---- one update inside a chain: the user's row comes in, the item row / bias stay in registers -------------
template <int J>
__device__ __forceinline__ float chain_step(const SgdArgs &a, Row<J> &p, Row<J> &q, float ub, float &ib, float rating) {
    const float err = rating - predict<J>(p, q, ub, ib, a.global_bias);  // sgd.cu:45
    rank1_update<J>(p, q, err, a.h);                                      // mf_sequential.cu:129-137
    ib = ib + a.h.lr * (err - a.h.ib_reg * ib);                           // :141
    return ub + a.h.lr * (err - a.h.ub_reg * ub);                         // :140 (new user bias)
}

#ifndef CU2REC_ABLATE
#define CU2REC_ABLATE 0  // timing-only builds of the hot path: 8 = no user-side updates, 16 = no LDS publishes by the
#endif                   // compute wave, 32 = no row loads (results are wrong; never set in the shipped library)

constexpr int kWide = 32;  // lanes per chain in the wide layout (one float4 slot per lane, 65 <= ld <= 128)

typedef unsigned int uint2_t __attribute__((ext_vector_type(2)));

// Wide layout only: adds the two 16-lane rows of a half-wavefront, s_l + s_{l+16} of the canonical order.
__device__ __forceinline__ float cross_row_sum(float v) {
    // rows (r0, r1) -> both operands hold (r0 | r0) and (r1 | r1): the sum is r0 + r1 in every lane of both rows
    const uint2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}

// ---- hot chains: a compute wave and two memory waves per block --------------------------------------------------
// The chain is latency bound: per update a ~17-deep dependent chain (slot dot, cross-row add, butterfly, error,
// item-row update).  Everything that is NOT on that chain -- fetching the next users' rows, updating and storing
// the user's row, the user bias -- is moved to a second wavefront of the same workgroup, which talks to the first
// one through LDS, double buffered per tile of 8 updates, with ONE __syncthreads() per tile as the only
// synchronisation (no flags, no spinning: every wave of the block executes the same number of barriers).
//   wave 0 ("A"):      the block's two chains, one per half-wave: reads the tile's user rows / ratings / biases from
//                      LDS, computes the error, publishes the item row as it was BEFORE the update and the error,
//                      updates the item row and bias in registers.
//   waves 1, 2 ("B"):  the same two chains, each wave half of every tile: load tile k+2's rows from global memory
//                      (entries three tiles ahead), apply tile k-1's user-row and user-bias updates from what A
//                      published and store them, publish tile k+1.
// W = lanes per chain: 32 (one slot per lane, 65 <= ld <= 128) or 16 (ld <= 64); wave 0 computes 64 / W chains,
// waves 1 and 2 move their memory
// Shape of the two-wave form for a row of S float4 slots per lane:
//   W      lanes per chain: 32 with S == 1 (65 <= ld <= 128: one slot per lane, s_l + s_{l+16} by v_permlane16_swap),
//          16 otherwise (the 16-lane layout of the other kernels, S slots per lane)
//   tile   updates between two barriers; part = the share of each of the two memory waves
//   chains chains per hot block, limited by the 64 KB of static LDS a block may declare
template <int W, int S>
struct DuoShape {
    static_assert(W == kGroup || S == 1, "the 32-lane layout holds exactly one slot per lane");
    static constexpr int kTile = S == 1 ? 8 : 4;
    static constexpr int kPart = kTile / 2;
    static constexpr int kChains = W == kWide ? 2 : (S <= 3 ? 4 : (S <= 6 ? 2 : 1));
};

template <int S>
struct WideRow {  // a lane's share of a factor row: slots lane, lane + W, ...
    float4 v[S];
};

template <int W, int S>
struct DuoLds {
    static constexpr int T = DuoShape<W, S>::kTile;
    float4 p[2][T][S][W];     // user rows of the tile, as loaded
    float4 qold[2][T][S][W];  // item row before each update of the tile (written by A)
    float rating[2][T], ub[2][T], err[2][T];
    int user[2][T];
};

template <int W, int S>
struct DuoTileRegs {  // one memory wave's share of a tile
    WideRow<S> rows[DuoShape<W, S>::kPart];
    uint64_t val;  // lanes 0..part-1: entry (user << 32 | rating bits)
    float ub;      // lanes 0..part-1: user bias
};

template <int W, int S>
__device__ __forceinline__ WideRow<S> load_wide_row(const float *__restrict__ base, size_t row, int ld, int nslots, int lane) {
    const float4 *p = reinterpret_cast<const float4 *>(base + row * static_cast<size_t>(ld));
    WideRow<S> r;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int slot = lane + W * s;
        const float4 v = p[min(slot, nslots - 1)];  // branch free: out-of-row lanes re-read the last slot and drop it
        const bool ok = slot < nslots;
        r.v[s] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
    }
    return r;
}

template <int W, int S>
__device__ __forceinline__ void store_wide_row(float *__restrict__ base, size_t row, int ld, int nslots, int lane,
                                               const WideRow<S> &r) {
    float4 *p = reinterpret_cast<float4 *>(base + row * static_cast<size_t>(ld));
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int slot = lane + W * s;
        if (slot < nslots) p[slot] = r.v[s];
    }
}

// this wave's entries of the tile starting at tile_start, one per lane 0..part-1 (mirrored on the other lanes)
template <int W, int S>
__device__ __forceinline__ uint64_t duo_load_vals(const uint64_t *__restrict__ vals, int tile_start, int end, int part,
                                                  int lane) {
    constexpr int kPart = DuoShape<W, S>::kPart;
    return vals[min(tile_start + kPart * part + (lane & (kPart - 1)), end - 1)];  // past the chain: its last entry, never used
}

template <int W, int S>
__device__ __forceinline__ void duo_load_rows(DuoTileRegs<W, S> &r, const SgdArgs &a, uint64_t val, int lane) {
    r.val = val;
    const int my_user = static_cast<int>(val >> 32);
    r.ub = a.user_bias[my_user];
#pragma unroll
    for (int t = 0; t < DuoShape<W, S>::kPart; ++t) {
        const int x = __shfl(my_user, t, W);
        r.rows[t] = load_wide_row<W, S>(a.P, static_cast<size_t>(x), a.ldp, a.nslots, lane);
    }
}

template <int W, int S>
__device__ __forceinline__ void duo_publish(DuoLds<W, S> &l, int buf, const DuoTileRegs<W, S> &r, int part, int lane) {
    constexpr int kPart = DuoShape<W, S>::kPart;
#pragma unroll
    for (int t = 0; t < kPart; ++t)
#pragma unroll
        for (int s = 0; s < S; ++s) l.p[buf][kPart * part + t][s][lane] = r.rows[t].v[s];
    if (lane < kPart) {
        const int e = kPart * part + lane;
        l.rating[buf][e] = __uint_as_float(static_cast<uint32_t>(r.val));
        l.ub[buf][e] = r.ub;
        l.user[buf][e] = static_cast<int>(r.val >> 32);
    }
}

// B: the user side of this wave's share of a finished tile (mf_sequential.cu:133-135,140 with the item row as it
// was at that update)
template <int W, int S>
__device__ __forceinline__ void duo_update_users(DuoLds<W, S> &l, int buf, int n_valid, const SgdArgs &a, int part, int lane) {
    constexpr int kPart = DuoShape<W, S>::kPart;
#pragma unroll
    for (int t = 0; t < kPart; ++t) {
        const int e = kPart * part + t;
        if (e < n_valid) {
            const float err = l.err[buf][e];
            WideRow<S> pn;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const float4 po = l.p[buf][e][s][lane], qo = l.qold[buf][e][s][lane];
                pn.v[s].x = step(po.x, qo.x, err, a.h.lr, a.h.p_reg);
                pn.v[s].y = step(po.y, qo.y, err, a.h.lr, a.h.p_reg);
                pn.v[s].z = step(po.z, qo.z, err, a.h.lr, a.h.p_reg);
                pn.v[s].w = step(po.w, qo.w, err, a.h.lr, a.h.p_reg);
            }
            store_wide_row<W, S>(a.P, static_cast<size_t>(l.user[buf][e]), a.ldp, a.nslots, lane, pn);
        }
    }
    const int mine = kPart * part + lane;
    if (lane < kPart && mine < n_valid) {
        const float ub = l.ub[buf][mine], err = l.err[buf][mine];
        a.user_bias[l.user[buf][mine]] = ub + a.h.lr * (err - a.h.ub_reg * ub);
    }
}

// A: the dependent chain of one update (same operations, same order as predict<J> + rank1_update<J>'s item half); the
// user's row, rating and bias arrive in registers.
template <int W, int S>
__device__ __forceinline__ void duo_step_a(DuoLds<W, S> &l, int buf, int t, const WideRow<S> &po, float rating, float ub,
                                           const SgdArgs &a, WideRow<S> &q, float &ib, int lane) {
    float acc = slot_dot(q.v[0], po.v[0]);
#pragma unroll
    for (int s = 1; s < S; ++s) acc = acc + slot_dot(q.v[s], po.v[s]);
    const float dot = W == kWide ? row_sum16(cross_row_sum(acc)) : row_sum16(acc);  // canonical order either way
    const float err = rating - (((a.global_bias + ub) + ib) + dot);
#if !(CU2REC_ABLATE & 16)
#pragma unroll
    for (int s = 0; s < S; ++s) l.qold[buf][t][s][lane] = q.v[s];
    if (lane == 0) l.err[buf][t] = err;
#endif
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const float4 qo = q.v[s];
        q.v[s].x = step(qo.x, po.v[s].x, err, a.h.lr, a.h.q_reg);
        q.v[s].y = step(qo.y, po.v[s].y, err, a.h.lr, a.h.q_reg);
        q.v[s].z = step(qo.z, po.v[s].z, err, a.h.lr, a.h.q_reg);
        q.v[s].w = step(qo.w, po.v[s].w, err, a.h.lr, a.h.q_reg);
    }
    ib = ib + a.h.lr * (err - a.h.ib_reg * ib);
}

template <int W, int S>
__device__ __forceinline__ void duo_compute(DuoLds<W, S> &l, int buf, int n_valid, const SgdArgs &a, WideRow<S> &q,
                                            float &ib, int lane) {
    constexpr int T = DuoShape<W, S>::kTile;
    if (n_valid == T) {
        // full tile: fetch everything its updates need from LDS up front (one latency per tile, not per update);
        // the scheduling barrier keeps the compiler from sinking the reads back next to their uses
        WideRow<S> po[T];
        float rating[T], ub[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
#pragma unroll
            for (int s = 0; s < S; ++s) po[t].v[s] = l.p[buf][t][s][lane];
            rating[t] = l.rating[buf][t];
            ub[t] = l.ub[buf][t];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < T; ++t) duo_step_a<W, S>(l, buf, t, po[t], rating[t], ub[t], a, q, ib, lane);
    } else {
        for (int t = 0; t < n_valid; ++t) {
            WideRow<S> po;
#pragma unroll
            for (int s = 0; s < S; ++s) po.v[s] = l.p[buf][t][s][lane];
            duo_step_a<W, S>(l, buf, t, po, l.rating[buf][t], l.ub[buf][t], a, q, ib, lane);
        }
    }
}

// One hot block: up to 64 / W chains; wave 0 computes (W lanes per chain), waves 1 and 2 each move half of every
// tile's memory traffic, wave 3 only keeps the barriers company.
template <int W, int S>
__device__ __forceinline__ void run_hot_block_duo(const SgdArgs &a, const uint32_t *__restrict__ keys,
                                                  const uint64_t *__restrict__ vals, int n_active, uint32_t key_base,
                                                  int n_hot, const int *__restrict__ item_of_rank, int rank_lo,
                                                  const int *__restrict__ ranges) {
    constexpr int kChains = DuoShape<W, S>::kChains, kTile = DuoShape<W, S>::kTile;
    __shared__ DuoLds<W, S> lds[kChains];
    __shared__ int s_range[kChains][2];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & (W - 1);
    const bool role_a = wave == 0, role_b = wave == 1 || wave == 2;
    const int part = wave - 1;                       // memory waves: which half of the tile
    const int group = (threadIdx.x & 63) / W;        // lane group inside the wave
    const bool has_chain = group < kChains;          // blocks of fat rows use only some of the groups
    const int c = has_chain ? group : 0;
    const int r = rank_lo + blockIdx.x * kChains + c;  // ranks below rank_lo belong to somebody else (block-solve mode)
    if (role_a && has_chain) {
        int begin = 0, end = 0;
        if (r < n_hot) {
            if (ranges) {  // chain_ranges_kernel has looked them up behind the sort
                begin = ranges[r];
                end = ranges[r + 1];
            } else {
                begin = lower_bound_key(keys, n_active, key_base | static_cast<uint32_t>(r));
                end = lower_bound_key(keys, n_active, key_base | static_cast<uint32_t>(r + 1));
            }
        }
        if (lane == 0) {
            s_range[c][0] = begin;
            s_range[c][1] = end;
        }
    }
    __syncthreads();
    const int begin = s_range[c][0], end = s_range[c][1], len = end - begin;
    int longest = 0;
#pragma unroll
    for (int i = 0; i < kChains; ++i) longest = max(longest, s_range[i][1] - s_range[i][0]);
    const int n_tiles = (longest + kTile - 1) / kTile;  // block uniform: every wave runs the same barriers
    if (n_tiles == 0) return;
    const bool alive = len > 0 && has_chain && (role_a || role_b);
    DuoLds<W, S> &l = lds[c];
    const int y = alive ? item_of_rank[r] : 0;
    WideRow<S> q;
#pragma unroll
    for (int s = 0; s < S; ++s) q.v[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    float ib = 0.f;
    // A memory wave keeps two register tiles and alternates their roles each phase (no copies: a copy would wait for
    // the loads just issued): at the start of phase k, r0 / r1 (by parity of k) holds its share of tile k+1 -- loads
    // issued a whole phase ago -- and the other one receives tile k+2.
    DuoTileRegs<W, S> r0, r1;
    uint64_t v_next = 0;  // this wave's entries of tile k+2 at the start of phase k
    if (alive && role_b) {
        const uint64_t v0 = duo_load_vals<W, S>(vals, begin, end, part, lane);
        const uint64_t v1 = duo_load_vals<W, S>(vals, begin + kTile, end, part, lane);
        v_next = duo_load_vals<W, S>(vals, begin + 2 * kTile, end, part, lane);
        duo_load_rows<W, S>(r1, a, v0, lane);
        duo_load_rows<W, S>(r0, a, v1, lane);
        duo_publish<W, S>(l, 0, r1, part, lane);
    } else if (alive) {
        q = load_wide_row<W, S>(a.Q, static_cast<size_t>(y), a.ldq, a.nslots, lane);
        ib = a.item_bias[y];
    }
    __syncthreads();
    auto b_phase = [&](int k, DuoTileRegs<W, S> &ready, DuoTileRegs<W, S> &loading) {
        const uint64_t v_after = duo_load_vals<W, S>(vals, begin + (k + 3) * kTile, end, part, lane);
#if !(CU2REC_ABLATE & 32)
        duo_load_rows<W, S>(loading, a, v_next, lane);  // tile k + 2
#endif
#if !(CU2REC_ABLATE & 8)
        if (k >= 1) duo_update_users<W, S>(l, (k - 1) & 1, min(max(len - (k - 1) * kTile, 0), kTile), a, part, lane);
#endif
        duo_publish<W, S>(l, (k + 1) & 1, ready, part, lane);  // tile k + 1, after the reads of that buffer just above
        v_next = v_after;
    };
    auto a_phase = [&](int k) { duo_compute<W, S>(l, k & 1, min(max(len - k * kTile, 0), kTile), a, q, ib, lane); };
    for (int k = 0; k < n_tiles; k += 2) {
        if (alive) {
            if (role_b) b_phase(k, r0, r1); else a_phase(k);
        }
        __syncthreads();
        if (k + 1 < n_tiles) {  // block uniform
            if (alive) {
                if (role_b) b_phase(k + 1, r1, r0); else a_phase(k + 1);
            }
            __syncthreads();
        }
    }
    if (alive) {
        if (role_b) {
            duo_update_users<W, S>(l, (n_tiles - 1) & 1, min(max(len - (n_tiles - 1) * kTile, 0), kTile), a, part, lane);
        } else {
            store_wide_row<W, S>(a.Q, static_cast<size_t>(y), a.ldq, a.nslots, lane, q);
            if (lane == 0) a.item_bias[y] = ib;
        }
    }
}

// A 16-lane group owns ONE sorted position and runs the chain that starts there (rank >= n_hot; longer chains have the
// two-wave form).  A chain is a string of dependent updates on the item row, but nothing else about it depends on values:
// three dependent round trips bring everything in -- (1) the 16 keys around the position and the 16 schedule entries
// behind it, (2) the item's id and the first four users' rows, (3) the item row -- and the users' rows stay four links
// ahead from there (a walk that asks for a row only when it needs it pays the memory latency per link).  Loads past the
// end of the chain re-read its last row (a cache hit) instead of being predicated.
template <int J>
__device__ __forceinline__ void walk_group(const SgdArgs &a, const uint32_t *__restrict__ keys, const uint64_t *__restrict__ vals,
                                           int n_active, const int *__restrict__ item_of_rank, uint32_t item_mask, int n_hot,
                                           int start) {
#ifndef CU2REC_WALK_D
#define CU2REC_WALK_D 4  // user rows a walking group keeps in flight (tuning builds: tools/build_variant.sh)
#endif
    constexpr int D = CU2REC_WALK_D;
    const int lane = threadIdx.x & (kGroup - 1);
    const int gshift = threadIdx.x & 48;  // this group's lanes in the wavefront
    if (start >= n_active) return;
    const int kpos = start - 1 + lane;  // lane 0: the position in front, lane 1: this one, lanes 2..15: the next 14
    const uint32_t key_l = keys[min(max(kpos, 0), n_active - 1)];
    const uint64_t val0 = vals[min(start + lane, n_active - 1)];
    const uint32_t key = __shfl(key_l, gshift + 1), prev = __shfl(key_l, gshift);
    if (static_cast<int>(key & item_mask) < n_hot) return;  // a hot chain: the other role runs it
    if (start > 0 && prev == key) return;                    // the chain began at an earlier position
    const bool same0 = lane >= 2 && kpos < n_active && key_l == key;
    const uint32_t m0 = (static_cast<uint32_t>(__ballot(same0) >> gshift) & 0xffffu) >> 2;  // bit i: keys[start + 1 + i] == key
    int len = __ffs(static_cast<int>(~m0));  // 1 + leading run of equal keys: 1..15
    bool more = len == 15;
    auto link = [&](uint64_t mv, int i, int nw) -> uint64_t {
        const int src = gshift + min(i, nw - 1);
        const uint32_t lo = __shfl(static_cast<uint32_t>(mv), src), hi = __shfl(static_cast<uint32_t>(mv >> 32), src);
        return static_cast<uint64_t>(hi) << 32 | lo;
    };
    Row<J> cur[D], nxt[D];
    float cub[D], nub[D];
    auto prime = [&](uint64_t mv, int nw) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int x = static_cast<int>(link(mv, d, nw) >> 32);
            cur[d] = load_row<J>(a.P, static_cast<size_t>(x), a.ldp, a.nslots, lane);
            cub[d] = a.user_bias[x];
        }
    };
    prime(val0, len);
    const int y = item_of_rank[key & item_mask];
    while (more) {  // the chain's length: the group looks at 16 more keys at a time
        const int pos = start + len + lane;
        const bool same = pos < n_active && keys[pos] == key;
        const uint32_t m = static_cast<uint32_t>(__ballot(same) >> gshift) & 0xffffu;
        const int run = __ffs(static_cast<int>(~m)) - 1;  // 0..16
        len += run;
        more = run == kGroup;
    }
    Row<J> q = load_row<J>(a.Q, static_cast<size_t>(y), a.ldq, a.nslots, lane);
    float ib = a.item_bias[y];
    for (int wb = 0; wb < len; wb += kGroup) {
        const int nw = min(kGroup, len - wb);
        uint64_t myval = val0;
        if (wb > 0) {
            myval = vals[start + wb + min(lane, nw - 1)];
            prime(myval, nw);
        }
        for (int b0 = 0; b0 < nw; b0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int x = static_cast<int>(link(myval, b0 + D + d, nw) >> 32);
                nxt[d] = load_row<J>(a.P, static_cast<size_t>(x), a.ldp, a.nslots, lane);
                nub[d] = a.user_bias[x];
            }
#pragma unroll
            for (int d = 0; d < D; ++d) {
                if (b0 + d < nw) {
                    const uint64_t val = link(myval, b0 + d, nw);
                    const int x = static_cast<int>(val >> 32);
                    const float new_ub = chain_step<J>(a, cur[d], q, cub[d], ib, __uint_as_float(static_cast<uint32_t>(val)));
                    store_row<J>(a.P, static_cast<size_t>(x), a.ldp, a.nslots, lane, cur[d]);
                    if (lane == 0) a.user_bias[x] = new_ub;
                }
            }
#pragma unroll
            for (int d = 0; d < D; ++d) {
                cur[d] = nxt[d];
                cub[d] = nub[d];
            }
        }
    }
    store_row<J>(a.Q, static_cast<size_t>(y), a.ldq, a.nslots, lane, q);
    if (lane == 0) a.item_bias[y] = ib;
}

// One launch = one iteration.  Two roles, chosen per block:
//  * blocks [0, hot_blocks): the chains of popularity ranks r < n_hot, DuoShape::kChains per block, in the two-wave
//    form above.  These are the long chains (thousands of updates on the most rated items), i.e. the critical path.
//  * the other blocks: walk_group, one sorted position per 16-lane group.
template <int J>
__global__ __launch_bounds__(kBlock) void sgd_ordered_kernel(SgdArgs a, const uint32_t *__restrict__ keys,
                                                             const uint64_t *__restrict__ vals, int n_active,
                                                             const int *__restrict__ item_of_rank,
                                                             uint32_t item_mask, uint32_t key_base, int n_hot,
                                                             int hot_blocks, int rank_lo, const int *__restrict__ ranges) {
    if (static_cast<int>(blockIdx.x) < hot_blocks) {
        // two-wave form for every row width: 32 lanes x 1 slot when 65 <= ld <= 128, else 16 lanes x J slots
        run_hot_block_duo<(J == 2 ? kWide : kGroup), (J == 2 ? 1 : J)>(a, keys, vals, n_active, key_base, n_hot, item_of_rank, rank_lo,
                                                                         ranges);
        return;
    }
    walk_group<J>(a, keys, vals, n_active, item_of_rank, item_mask, n_hot,
                  static_cast<int>((blockIdx.x - hot_blocks) * kBlock + threadIdx.x) / kGroup);
}

// The walk alone (block-solve mode runs it beside the two-wave chains on a stream of its own: inside sgd_ordered_kernel
// every block reserves the two-wave role's 33 KB of LDS, which halves the walk's occupancy).
template <int J>
__global__ __launch_bounds__(kBlock) void sgd_walk_kernel(SgdArgs a, const uint32_t *__restrict__ keys,
                                                          const uint64_t *__restrict__ vals, int n_active,
                                                          const int *__restrict__ item_of_rank, uint32_t item_mask, int n_hot) {
    walk_group<J>(a, keys, vals, n_active, item_of_rank, item_mask, n_hot, static_cast<int>(blockIdx.x * kBlock + threadIdx.x) / kGroup);
}

// Small sets: ALL iterations of a schedule batch in ONE launch.  When a whole iteration's grid is co-resident (ML-1M shape: 378
// walk blocks + the two-wave blocks on 256 CUs) the kernel boundary between two iterations -- what carries the user rows from the
// workgroup that wrote them to the one that reads them next -- becomes the grid barrier of the resident Hogwild launches
// (grid_barrier.hpp: XCD-hierarchical, release = one L2 write-back per XCD, acquire = L1 invalidate, bounded spins) -- the form
// VERDICT r3 asked for against training.cu:107-115's cadence of one launch per iteration.  Same schedule, same roles, same
// arithmetic: bit-identical to the launch-per-iteration form.  Measured on the ML-1M shape (f = 50): 18.1 us per iteration
// against 15.6 with a launch per iteration -- the barrier costs more than the boundary -- so it is opt-in, not the default.
template <int J>
__global__ __launch_bounds__(kBlock) void sgd_ordered_persistent_kernel(SgdArgs a, const uint32_t *__restrict__ keys,
                                                                        const uint64_t *__restrict__ vals, size_t stride, int n_active,
                                                                        const int *__restrict__ item_of_rank, uint32_t item_mask,
                                                                        int n_hot, int hot_blocks, int rank_lo,
                                                                        const int *__restrict__ ranges, int ranges_stride, int n_iters,
                                                                        gridbar::Args ra) {
    __shared__ gridbar::BarrierShared s_barrier;
    if (threadIdx.x == 0) gridbar::barrier_census(ra, &s_barrier);
    __syncthreads();
    if (!s_barrier.ok) return;  // the grid is not co-resident: nothing is run, the host reports it
    for (int b = 0; b < n_iters; ++b) {
        const uint32_t *kb = keys + static_cast<size_t>(b) * stride;
        const uint64_t *vb = vals + static_cast<size_t>(b) * stride;
        const int *rb = ranges ? ranges + static_cast<size_t>(b) * ranges_stride : nullptr;
        if (static_cast<int>(blockIdx.x) < hot_blocks)
            run_hot_block_duo<(J == 2 ? kWide : kGroup), (J == 2 ? 1 : J)>(a, kb, vb, n_active, 0u, n_hot, item_of_rank, rank_lo, rb);
        else
            walk_group<J>(a, kb, vb, n_active, item_of_rank, item_mask, n_hot,
                          static_cast<int>((blockIdx.x - hot_blocks) * kBlock + threadIdx.x) / kGroup);
        if (b + 1 == n_iters) break;  // the launch's end is the last boundary
        gridbar::barrier_arrive(ra, static_cast<unsigned>(b + 1), &s_barrier);
        if (!gridbar::barrier_wait(ra, static_cast<unsigned>(b + 1), &s_barrier)) break;
    }
}

std::atomic<int> g_persistent_launches{0};

// true: the batch's iterations were queued as ONE persistent launch; false: not eligible / refused -- launch per iteration
template <int J>
bool launch_persistent(const SgdArgs &a, const uint32_t *keys, const uint64_t *vals, size_t stride, int n_active, const int *item_of_rank,
                       uint32_t item_mask, int n_hot, int rank_lo, const int *ranges, int ranges_stride, int n_iters, hipStream_t stream) {
    const int chains_per_block = DuoShape<(J == 2 ? kWide : kGroup), (J == 2 ? 1 : J)>::kChains;
    int hot_blocks = (std::max(n_hot - rank_lo, 0) + chains_per_block - 1) / chains_per_block;
    const int walk_blocks = (n_active + kGroupsPerBlock - 1) / kGroupsPerBlock;
    int blocks = hot_blocks + walk_blocks;
    static std::mutex mutex;
    static std::map<int, int> per_cu_by_device;  // co-resident workgroups per CU of this instantiation (occupancy query, once)
    int dev = 0;
    CU2REC_HIP(hipGetDevice(&dev));
    int per_cu = 0;
    {
        std::lock_guard<std::mutex> lock(mutex);
        auto it = per_cu_by_device.find(dev);
        if (it == per_cu_by_device.end()) {
            int q = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, sgd_ordered_persistent_kernel<J>, kBlock, 0) != hipSuccess) {
                (void)hipGetLastError();
                q = 0;
            }
            it = per_cu_by_device.emplace(dev, q).first;
        }
        per_cu = it->second;
    }
    if (per_cu < 1) return false;
    GridBarrierBlock bar = grid_barrier_begin(stream);
    // every block of the grid must hold a CU slot for the whole launch; leave one slot per CU to whatever else runs (the
    // schedule kernels of the next batch), or the grid could wait at its first barrier for workgroups that cannot start
    const long long capacity = static_cast<long long>(bar.cus) * std::max(per_cu - 1, 1);
    if (blocks > capacity || !bar.cooperative) {
        grid_barrier_end(stream);
        return false;
    }
    SgdArgs args = a;
    gridbar::Args ra{bar.words, bar.status, 0};
    uint32_t key_base = 0u;
    (void)key_base;
    void *params[] = {&args,   &keys,       &vals,   &stride,        &n_active, &item_of_rank, &item_mask, &n_hot,
                      &hot_blocks, &rank_lo, &ranges, &ranges_stride, &n_iters,  &ra};
    const hipError_t launched = hipLaunchCooperativeKernel(reinterpret_cast<const void *>(sgd_ordered_persistent_kernel<J>), dim3(blocks),
                                                           dim3(kBlock), params, 0, stream);
    grid_barrier_end(stream);
    if (launched == hipErrorCooperativeLaunchTooLarge || launched == hipErrorLaunchOutOfResources) {
        (void)hipGetLastError();  // a refusal, not an error: one launch per iteration instead
        std::lock_guard<std::mutex> lock(mutex);
        per_cu_by_device[dev] = 0;  // ... and do not ask again on this device
        return false;
    }
    CU2REC_HIP(launched);
    g_persistent_launches.fetch_add(1);
    return true;
}

bool launch_chains_persistent(const SgdArgs &a, const uint32_t *keys, const uint64_t *vals, size_t stride, int n_active,
                              const int *item_of_rank, uint32_t item_mask, int n_hot, int rank_lo, const int *ranges, int ranges_stride,
                              int n_iters, hipStream_t stream) {
    switch (slots_per_lane(a.nslots)) {
        case 1: return launch_persistent<1>(a, keys, vals, stride, n_active, item_of_rank, item_mask, n_hot, rank_lo, ranges, ranges_stride, n_iters, stream);
        case 2: return launch_persistent<2>(a, keys, vals, stride, n_active, item_of_rank, item_mask, n_hot, rank_lo, ranges, ranges_stride, n_iters, stream);
        case 3: return launch_persistent<3>(a, keys, vals, stride, n_active, item_of_rank, item_mask, n_hot, rank_lo, ranges, ranges_stride, n_iters, stream);
        case 4: return launch_persistent<4>(a, keys, vals, stride, n_active, item_of_rank, item_mask, n_hot, rank_lo, ranges, ranges_stride, n_iters, stream);
        default: return false;  // (wider rows: small sets of such rows keep the launch per iteration)
    }
}

// roles: kRoleDuo = the two-wave chains of ranks [rank_lo, n_hot), kRoleWalk = the walk of ranks >= n_hot; both in one launch
// (the ordered mode), or one launch each (block-solve mode, two streams)
constexpr int kRoleDuo = 1, kRoleWalk = 2;

template <int J>
void launch_chain(const SgdArgs &a, const uint32_t *keys, const uint64_t *vals, int n_active, const int *item_of_rank,
                  uint32_t item_mask, uint32_t key_base, int n_hot, hipStream_t stream, int rank_lo, int roles, const int *ranges,
                  hipEvent_t stop) {
    const int chains_per_block = DuoShape<(J == 2 ? kWide : kGroup), (J == 2 ? 1 : J)>::kChains;
    const int hot_blocks = (std::max(n_hot - rank_lo, 0) + chains_per_block - 1) / chains_per_block;
    const int walk_blocks = (n_active + kGroupsPerBlock - 1) / kGroupsPerBlock;
    if (roles == kRoleWalk) {
        hipLaunchKernelGGL(sgd_walk_kernel<J>, dim3(walk_blocks), dim3(kBlock), 0, stream, a, keys, vals, n_active, item_of_rank,
                           item_mask, n_hot);
        return;
    }
    const int blocks = hot_blocks + ((roles & kRoleWalk) ? walk_blocks : 0);
    if (blocks == 0) return;
    if (stop)  // the event rides on the kernel's completion signal (see bs_launch_gram)
        hipExtLaunchKernelGGL(sgd_ordered_kernel<J>, dim3(blocks), dim3(kBlock), 0, stream, nullptr, stop, 0, a, keys, vals, n_active,
                              item_of_rank, item_mask, key_base, n_hot, hot_blocks, rank_lo, ranges);
    else
        hipLaunchKernelGGL(sgd_ordered_kernel<J>, dim3(blocks), dim3(kBlock), 0, stream, a, keys, vals, n_active,
                           item_of_rank, item_mask, key_base, n_hot, hot_blocks, rank_lo, ranges);
}

void launch_chains(const SgdArgs &a, const uint32_t *kb, const uint64_t *vb, int n_active, const int *item_of_rank,
                   uint32_t item_mask, uint32_t key_base, int n_hot, hipStream_t stream, int rank_lo, int roles = kRoleDuo | kRoleWalk,
                   const int *ranges = nullptr, hipEvent_t stop = nullptr) {
    switch (slots_per_lane(a.nslots)) {
        case 1: launch_chain<1>(a, kb, vb, n_active, item_of_rank, item_mask, key_base, n_hot, stream, rank_lo, roles, ranges, stop); break;
        case 2: launch_chain<2>(a, kb, vb, n_active, item_of_rank, item_mask, key_base, n_hot, stream, rank_lo, roles, ranges, stop); break;
        case 3: launch_chain<3>(a, kb, vb, n_active, item_of_rank, item_mask, key_base, n_hot, stream, rank_lo, roles, ranges, stop); break;
        case 4: launch_chain<4>(a, kb, vb, n_active, item_of_rank, item_mask, key_base, n_hot, stream, rank_lo, roles, ranges, stop); break;
        case 5: launch_chain<5>(a, kb, vb, n_active, item_of_rank, item_mask, key_base, n_hot, stream, rank_lo, roles, ranges, stop); break;
        case 6: launch_chain<6>(a, kb, vb, n_active, item_of_rank, item_mask, key_base, n_hot, stream, rank_lo, roles, ranges, stop); break;
        case 7: launch_chain<7>(a, kb, vb, n_active, item_of_rank, item_mask, key_base, n_hot, stream, rank_lo, roles, ranges, stop); break;
        case 8: launch_chain<8>(a, kb, vb, n_active, item_of_rank, item_mask, key_base, n_hot, stream, rank_lo, roles, ranges, stop); break;
        default: fail(CU2REC_EUNSUPPORTED, "n_factors above 512 is not compiled in");
    }
}

int bits_for(int n) {  // bits needed to represent values in [0, n)
    int b = 1;
    while ((1ll << b) < n) ++b;
    return b;
}

}  // namespace

OrderedSchedule::OrderedSchedule(const int *d_indptr, const int *d_indices, int n_rows_, int n_cols_, int nnz_)
    : n_rows(n_rows_), n_cols(n_cols_), nnz(nnz_) {
    require(d_indptr && n_rows >= 0 && n_cols > 0 && nnz >= 0, "ordered schedule: bad argument");
    require_device();
    // one-off host pass: users with ratings, item popularity rank (descending rating count, ties by item id)
    std::vector<int> indptr(static_cast<size_t>(n_rows) + 1), indices(static_cast<size_t>(std::max(nnz, 1)));
    CU2REC_HIP(hipMemcpy(indptr.data(), d_indptr, indptr.size() * sizeof(int), hipMemcpyDeviceToHost));
    if (nnz) CU2REC_HIP(hipMemcpy(indices.data(), d_indices, static_cast<size_t>(nnz) * sizeof(int), hipMemcpyDeviceToHost));
    n_active = 0;
    for (int u = 0; u < n_rows; ++u) n_active += indptr[u + 1] > indptr[u];
    // popularity = expected updates per iteration: every user draws one of its ratings uniformly (sgd.cu:36-37), so an
    // item collects sum over its raters of 1 / degree; descending, ties by item id.  Ranks only order the chains
    // (longest first) and pick the hot ones; results do not depend on them.
    std::vector<double> rate(n_cols, 0.0);
    std::vector<int> order(n_cols), rank(n_cols);
    for (int u = 0; u < n_rows; ++u) {
        const int low = indptr[u], high = indptr[u + 1];
        require(low <= high && high <= nnz, "ordered schedule: bad indptr");
        const double w = high > low ? 1.0 / (high - low) : 0.0;
        for (int k = low; k < high; ++k) {
            require(indices[k] >= 0 && indices[k] < n_cols, "ordered schedule: item id out of range");
            rate[indices[k]] += w;
        }
    }
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int l, int r) { return rate[l] > rate[r]; });
    for (int r = 0; r < n_cols; ++r) rank[order[r]] = r;
    item_rank.allocate(n_cols);
    item_rank.upload(rank.data(), n_cols);
    item_of_rank.allocate(n_cols);
    item_of_rank.upload(order.data(), n_cols);
    item_bits = bits_for(n_cols);
    // batch size: keep key = b << item_bits | rank (+ the sentinel bit) inside 32 bits and the buffers modest
    int b_bits = std::min(6, 31 - item_bits - 1);
    require(b_bits >= 0, "ordered schedule: too many items for 32-bit keys");
    max_batch = 1 << b_bits;
    while (max_batch > 1 && static_cast<size_t>(max_batch) * n_rows > (size_t(1) << 27)) max_batch >>= 1;
    const size_t cap = static_cast<size_t>(max_batch) * std::max(n_rows, 1);
    for (int slot = 0; slot < 2; ++slot)
        for (int i = 0; i < 2; ++i) {
            keys[slot][i].allocate(cap);
            vals[slot][i].allocate(cap);
        }
    hipcub::DoubleBuffer<uint32_t> dk(keys[0][0].ptr, keys[0][1].ptr);
    hipcub::DoubleBuffer<uint64_t> dv(vals[0][0].ptr, vals[0][1].ptr);
    temp_bytes = 0;
    CU2REC_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, dk, dv, static_cast<int>(cap), 0, 32, nullptr));
    {
        // the segmented form (run()): every iteration of a batch is a segment of n_rows pairs
        std::vector<int> offs(static_cast<size_t>(max_batch) + 1);
        for (int b = 0; b <= max_batch; ++b) offs[b] = static_cast<int>(static_cast<size_t>(b) * n_rows);
        seg_offsets.allocate(offs.size());
        seg_offsets.upload(offs.data(), offs.size());
        size_t seg_bytes = 0;
        CU2REC_HIP(rocprim::segmented_radix_sort_pairs<SegSortConfig>(nullptr, seg_bytes, keys[0][0].ptr, keys[0][1].ptr, vals[0][0].ptr,
                                                                       vals[0][1].ptr, static_cast<unsigned>(cap), static_cast<unsigned>(max_batch),
                                                                       seg_offsets.ptr, seg_offsets.ptr + 1, 0u, 32u, nullptr));
        temp_bytes = std::max(temp_bytes, seg_bytes);
    }
    temp.allocate(temp_bytes + 16);
    // block-solve workspace
    // the threshold: what the caller set, else scaled with the set -- down (a long chain costs the same whatever the set,
    // the launch overhead it has to beat does not shrink with it; 1/1 .. 1/8 of the ML-20M shape,
    // profiles/r02_shard_size_probe_blocksolve.log) and up (Netflix shape, 480,189 users: 290 us per iteration at 800
    // against 353 at 240 -- the chains of several hundred hot items do not fit the CUs at once)
    float min_rate = blocksolve_min_rate_base();
    if (!blocksolve_min_rate_is_set()) {
        min_rate = std::max(30.f, min_rate * static_cast<float>(n_active) / 131072.f);
        // no chain long enough to pay for three launches and two events per iteration: the ordered walk alone is faster
        // (an eighth of the ML-20M shape, hottest item 247 updates per iteration: 40 against 53 us per iteration)
        if (n_cols > 0 && rate[order[0]] < 350.0) min_rate = std::numeric_limits<float>::infinity();
    }
    n_hot_bs = 0;
    while (n_hot_bs < n_cols && rate[order[n_hot_bs]] >= min_rate) ++n_hot_bs;
    n_hot_bs = std::min(n_hot_bs, (1 << item_bits) - 1);
    // beside the block solves: chains expected to be at least 12 links long run in the ordered mode's two-wave form
    n_duo_bs = n_hot_bs;
    while (n_duo_bs < n_cols && n_duo_bs < 4096 && rate[order[n_duo_bs]] >= 12.0) ++n_duo_bs;
    n_duo_bs = std::min(n_duo_bs, (1 << item_bits) - 1);
    max_blocks = n_active / kBsLinks + n_hot_bs + 1;
    // the look-ahead form of phase 2 (blocksolve.hip): the leading ranks whose chains are expected to be at least
    // blocksolve_lookahead_blocks() blocks long -- the chains whose length is the iteration's critical path.  Their blocks are
    // the first of every iteration's block table; la_cap bounds them with room to spare (a chain beyond it: the plain form).
    la_ranks = 0;
    la_cap = 0;
    {
        const int la_min = blocksolve_lookahead_blocks(-1);
        double blocks = 0.0;
        while (la_min > 0 && la_ranks < n_hot_bs && rate[order[la_ranks]] >= static_cast<double>(la_min) * kBsLinks) {
            blocks += std::ceil(rate[order[la_ranks]] / kBsLinks) + 1.0;
            ++la_ranks;
        }
        if (la_ranks > 0) la_cap = std::min(max_blocks, static_cast<int>(blocks * 1.25) + 4 * la_ranks + 8);
    }
    n_range_ranks = std::max(std::max(std::min(n_cols, kHotChains), n_duo_bs), n_hot_bs);
    n_range_ranks = std::min(n_range_ranks, (1 << item_bits) - 1);
    for (int slot = 0; slot < 2; ++slot) {
        chain_ranges[slot].allocate(static_cast<size_t>(max_batch) * (n_range_ranks + 1));
        chain_begin[slot].allocate(static_cast<size_t>(max_batch) * (n_hot_bs + 1));
        walk_begin[slot].allocate(max_batch);
        bs_chains[slot].allocate(static_cast<size_t>(max_batch) * std::max(n_hot_bs, 1));
        bs_blocks[slot].allocate(static_cast<size_t>(max_batch) * max_blocks);
        bs_order[slot].allocate(static_cast<size_t>(max_batch) * std::max(max_blocks, 1));
    }
    {
        // The next batch's schedule is built beside the iterations that consume this batch's, on a stream of the lowest priority
        // (measured: no different from the default priority -- what its kernels cost the iterations is DESIGN.md section 4)
        int lo = 0, hi = 0;
        CU2REC_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        CU2REC_HIP(hipStreamCreateWithPriority(&sched, hipStreamNonBlocking, lo));
    }
    for (int slot = 0; slot < 2; ++slot) {
        CU2REC_HIP(hipEventCreateWithFlags(&ev_ready[slot], hipEventDisableTiming));
        CU2REC_HIP(hipEventCreateWithFlags(&ev_consumed[slot], hipEventDisableTiming));
    }
    tables.allocate(kBsTableFloats);
    CU2REC_HIP(hipEventCreateWithFlags(&ev_last, hipEventDisableTiming));
    if (n_hot_bs > 0) {
        Mbuf.allocate(static_cast<size_t>(max_blocks) * kBsRecFloats);
        ebuf.allocate(static_cast<size_t>(max_blocks) * kBsLinks);
        gram_flag.allocate(max_blocks);
        gram_flag.zero();  // epochs start at 1
        chain_prog.allocate(static_cast<size_t>(n_hot_bs) * kBsProgWords);
        chain_prog.zero();
        pipe_done.allocate(16);
        pipe_done.zero();
        solve_started.allocate(16);
        solve_started.zero();
        gram_done.allocate(32 * 16);
        gram_done.zero();
        side_seq.allocate(16);
        side_seq.zero();
        if (la_cap > 0) Nbuf.allocate(static_cast<size_t>(la_cap) * kBsCrossFloats);
    }
}

OrderedSchedule::~OrderedSchedule() {
    // queued work may still read the buffers this object owns
    if (have_last) (void)hipEventSynchronize(ev_last);
    if (sched) (void)hipStreamSynchronize(sched);
    if (sched) (void)hipStreamDestroy(sched);
    for (int slot = 0; slot < 2; ++slot) {
        if (ev_ready[slot]) (void)hipEventDestroy(ev_ready[slot]);
        if (ev_consumed[slot]) (void)hipEventDestroy(ev_consumed[slot]);
    }
    if (solve) (void)hipStreamDestroy(solve);
    if (upd) (void)hipStreamDestroy(upd);
    for (hipEvent_t e : {ev_call, ev_upd, ev_solve, ev_gram, ev_last})
        if (e) (void)hipEventDestroy(e);
}

namespace {
std::atomic<int> g_bs_affine_blocks{-1};  // -1: not yet initialised from the environment
}

int ordered_persistent_launches() { return g_persistent_launches.load(); }

// How the block-solve mode forks and joins its side stream: 2 (default) gate kernel + device-side join, 1 gate kernel + event
// join, 0 events both ways.  CU2REC_BS_GATE in the environment decides; without it, a process running under a COUNTER pass of
// rocprofv3 (--pmc: the tool exports ROCPROF_COUNTER_COLLECTION) takes 0 -- a counter pass serialises kernels across streams, and
// the device-side join would wait for a signal kernel the profiler does not let run before the waiting kernel has ended.
int bs_gate_mode() {
    static const int mode = [] {
        if (const char *env = std::getenv("CU2REC_BS_GATE")) return std::atoi(env);
        if (const char *pmc = std::getenv("ROCPROF_COUNTER_COLLECTION"))
            if (*pmc && std::string(pmc) != "0" && std::string(pmc) != "False" && std::string(pmc) != "false") return 0;
        return 2;
    }();
    return mode;
}

int blocksolve_affine_blocks(int blocks) {
    int prev = g_bs_affine_blocks.load();
    if (prev < 0) {
        const char *env = std::getenv("CU2REC_BLOCKSOLVE_AFFINE");
        const int init = env ? std::max(0, std::atoi(env)) : 0;
        g_bs_affine_blocks.compare_exchange_strong(prev, init);
        prev = g_bs_affine_blocks.load();
    }
    if (blocks >= 0) g_bs_affine_blocks.store(blocks);
    return prev;
}

namespace {
std::atomic<int> g_bs_lookahead{-1};  // -1: not yet initialised from the environment
}

// Workgroups of the pipelined phase 3 (bs_update_pipe_kernel): persistent, so few enough to leave the side kernel its room, and enough
// to keep up with the chains (a block takes a workgroup a few microseconds; the chains together deliver a few dozen blocks per microsecond).
int blocksolve_pipe_grid(int set) {
    static int grid = [] {
        const char *e = std::getenv("CU2REC_BLOCKSOLVE_PIPE_GRID");
        return e && std::atoi(e) > 0 ? std::atoi(e) : 0;
    }();
    if (set > 0) grid = set;
    return grid > 0 ? grid : 2 * bs_compute_units();
}

int blocksolve_lookahead_blocks(int blocks) {
    int prev = g_bs_lookahead.load();
    if (prev < 0) {
        const char *env = std::getenv("CU2REC_BLOCKSOLVE_LOOKAHEAD");
        const int init = env ? std::max(0, std::atoi(env)) : 24;  // the top chains only: DESIGN.md section 4, "The look-ahead form"
        g_bs_lookahead.compare_exchange_strong(prev, init);
        prev = g_bs_lookahead.load();
    }
    if (blocks >= 0) g_bs_lookahead.store(blocks);
    return prev;
}

namespace {
std::atomic<int> g_bs_affine_head{-1};  // -1: not yet initialised from the environment
}

int blocksolve_affine_head(int blocks) {
    int prev = g_bs_affine_head.load();
    if (prev < 0) {
        const char *env = std::getenv("CU2REC_BLOCKSOLVE_AFFINE_HEAD");
        const int init = env ? std::max(1, std::atoi(env)) : 9;
        g_bs_affine_head.compare_exchange_strong(prev, init);
        prev = g_bs_affine_head.load();
    }
    if (blocks >= 1) g_bs_affine_head.store(blocks);
    return prev;
}

namespace {
// explicit threshold (> 0), or <= 0: automatic (240 per 131,072 rating users, scaled with the set); -2 = not yet initialised
// from the environment
std::atomic<float> g_bs_min_rate{-2.f};
constexpr float kBsDefaultRate = 240.f;

float bs_min_rate_state() {
    float cur = g_bs_min_rate.load();
    if (cur == -2.f) {  // first use: CU2REC_BLOCKSOLVE_RATE overrides the automatic threshold
        float init = -1.f;
        if (const char *env = std::getenv("CU2REC_BLOCKSOLVE_RATE")) init = std::max(0.01f, static_cast<float>(std::atof(env)));
        g_bs_min_rate.compare_exchange_strong(cur, init);
        cur = g_bs_min_rate.load();
    }
    return cur;
}
}  // namespace

bool blocksolve_min_rate_is_set() { return bs_min_rate_state() > 0.f; }

float blocksolve_min_rate_base() {
    const float cur = bs_min_rate_state();
    return cur > 0.f ? cur : kBsDefaultRate;
}

// rate > 0: explicit threshold; rate < 0: back to automatic; rate == 0: query.  Returns what was in force before the call:
// the explicit value, or -1 for automatic -- so that f(f(x)) restores either state.
float blocksolve_min_rate(float rate) {
    const float prev = bs_min_rate_state();
    if (rate > 0.f) g_bs_min_rate.store(rate);
    else if (rate < 0.f) g_bs_min_rate.store(-1.f);
    return prev > 0.f ? prev : -1.f;
}

void OrderedSchedule::run(SgdArgs a, uint64_t iter0, int n_iters, hipStream_t stream, bool blocksolve) {
    if (n_active == 0) return;
    if (blocksolve && n_hot_bs == 0) blocksolve = false;  // nothing to solve block-wise: the ordered walk, bit for bit
    // The workspace (key / value buffers, block records, events) exists once: a call on another stream than the last one
    // starts behind that call's end.
    if (have_last && last_stream != stream) CU2REC_HIP(hipStreamWaitEvent(stream, ev_last, 0));
    if (blocksolve) {
        if (!bs_supported(a.nslots)) fail(CU2REC_EUNSUPPORTED, "block-solve mode is compiled for n_factors <= 252");
        const bool same = tables_valid && tables_for.lr == a.h.lr && tables_for.q_reg == a.h.q_reg && tables_for.ib_reg == a.h.ib_reg;
        if (!same) {  // stream ordered: kernels already queued keep the old tables
            bs_launch_tables(a.h, tables.ptr, stream);
            tables_for = a.h;
            tables_valid = true;
        }
        if (qstart_ld != a.ldq) {
            CU2REC_HIP(hipStreamSynchronize(stream));
            qstart.allocate(static_cast<size_t>(max_blocks) * a.ldq);
            qstart_ld = a.ldq;
        }
        if (!solve) {
            // default priorities all round: a high-priority queue whose workgroups cannot all be placed (more hot chains than
            // CUs) held back the queues below it, including phase 1, which those workgroups wait for (a 300-chain test set
            // ran into the waits' timeout)
            CU2REC_HIP(hipStreamCreateWithFlags(&solve, hipStreamNonBlocking));
            CU2REC_HIP(hipStreamCreateWithFlags(&upd, hipStreamNonBlocking));
            for (hipEvent_t *e : {&ev_call, &ev_upd, &ev_solve, &ev_gram}) CU2REC_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
        }
    }
    const uint32_t item_mask = (1u << item_bits) - 1u;
    // ranks [0, n_hot) get a dedicated group each with deep prefetch; keep r + 1 representable in item_bits
    int hot_wanted = kHotChains;
    if (const char *env = std::getenv("CU2REC_ORDERED_HOT")) hot_wanted = std::max(0, std::atoi(env));  // tuning aid
    const int n_hot = std::min(std::min(n_cols, hot_wanted), static_cast<int>(item_mask));
    // Sample + sort (+ plan) of one batch into a slot, on the schedule stream: nothing in it depends on the model.

    // SEGMENTED (default): the keys carry the rank only, and every iteration of the batch is sorted as a segment of its own
    // (hipcub::DeviceSegmentedRadixSort: one workgroup per long segment; stable, so the order is the one the sort of the whole
    // batch by (iteration, rank) gives).  The whole-batch sort is three chip-filling passes of 130 us each per 64 iterations,
    // and the iterations that ran beside the schedule kernels took 175 instead of 92 us: a chain's workgroup needs most of a
    // CU and is not placed while a sort pass keeps refilling every CU (8 us per iteration on average,
    // tools/schedule_interference.py; stream priorities and more hardware queues change nothing; one slice of small sorts per
    // iteration is host bound).  CU2REC_SCHED_SEGMENTED=0: the whole-batch sort.
    static const bool segmented = !(std::getenv("CU2REC_SCHED_SEGMENTED") && std::atoi(std::getenv("CU2REC_SCHED_SEGMENTED")) == 0);
    static const bool tile_keys = !(std::getenv("CU2REC_SCHED_KEYS_TILE") && std::atoi(std::getenv("CU2REC_SCHED_KEYS_TILE")) == 0);  // max_batch <= 64
    const size_t seg = segmented ? static_cast<size_t>(n_rows) : static_cast<size_t>(n_active);  // an iteration's stride in the sorted arrays
    auto schedule = [&](int slot, uint64_t first_iter, int nb) {
        win[slot].valid = false;  // (until everything below has been queued: a launch that throws leaves no window behind)
        if (segmented) {
            const size_t n = static_cast<size_t>(nb) * n_rows;
            const int blocks = static_cast<int>(std::min<size_t>((n + kBlock - 1) / kBlock, 1 << 16));
            if (slot_used[slot]) CU2REC_HIP(hipStreamWaitEvent(sched, ev_consumed[slot], 0));  // its last batch has been run
            if (tile_keys)
                hipLaunchKernelGGL(schedule_keys_tile_kernel, dim3((n_rows + 63) / 64), dim3(256), 0, sched, a.indptr, a.indices, a.data,
                                   item_rank.ptr, n_rows, nb, item_bits, 1u << item_bits, a.seed, first_iter, a.user_offset, keys[slot][0].ptr,
                                   vals[slot][0].ptr, 0);
            else
                hipLaunchKernelGGL(schedule_keys_kernel, dim3(blocks), dim3(kBlock), 0, sched, a.indptr, a.indices, a.data, item_rank.ptr, n_rows, nb,
                                   item_bits, 1u << item_bits, a.seed, first_iter, a.user_offset, keys[slot][0].ptr, vals[slot][0].ptr, 0);
            CU2REC_HIP(hipGetLastError());
            size_t bytes = temp_bytes;
            CU2REC_HIP(rocprim::segmented_radix_sort_pairs<SegSortConfig>(temp.ptr, bytes, keys[slot][0].ptr, keys[slot][1].ptr, vals[slot][0].ptr,
                                                                           vals[slot][1].ptr, static_cast<unsigned>(n), static_cast<unsigned>(nb),
                                                                           seg_offsets.ptr, seg_offsets.ptr + 1, 0u,
                                                                           static_cast<unsigned>(item_bits + 1), sched));
            win[slot] = Window{true, a.seed, first_iter, nb, a.user_offset, a.indptr, blocksolve, keys[slot][1].ptr, vals[slot][1].ptr};
            hipLaunchKernelGGL(chain_ranges_kernel, dim3(nb), dim3(kBlock), 0, sched, win[slot].sk, n_active, n_range_ranks, item_bits,
                               chain_ranges[slot].ptr, seg, 0);
            CU2REC_HIP(hipGetLastError());
            if (blocksolve)
                bs_launch_plan(win[slot].sk, n_active, nb, n_hot_bs, item_bits, max_blocks, item_of_rank.ptr, chain_begin[slot].ptr,
                               bs_chains[slot].ptr, bs_blocks[slot].ptr, walk_begin[slot].ptr, sched, seg, false, bs_order[slot].ptr);
            CU2REC_HIP(hipEventRecord(ev_ready[slot], sched));
            slot_used[slot] = true;
            return;
        }
        const int b_bits = bits_for(nb);
        const uint32_t sentinel = 1u << (item_bits + b_bits);
        const size_t n = static_cast<size_t>(nb) * n_rows;
        const int blocks = static_cast<int>(std::min<size_t>((n + kBlock - 1) / kBlock, 1 << 16));
        if (slot_used[slot]) CU2REC_HIP(hipStreamWaitEvent(sched, ev_consumed[slot], 0));  // its last batch has been run
        if (tile_keys)
            hipLaunchKernelGGL(schedule_keys_tile_kernel, dim3((n_rows + 63) / 64), dim3(256), 0, sched, a.indptr, a.indices, a.data,
                               item_rank.ptr, n_rows, nb, item_bits, sentinel, a.seed, first_iter, a.user_offset, keys[slot][0].ptr,
                               vals[slot][0].ptr, 1);
        else
            hipLaunchKernelGGL(schedule_keys_kernel, dim3(blocks), dim3(kBlock), 0, sched, a.indptr, a.indices, a.data,
                               item_rank.ptr, n_rows, nb, item_bits, sentinel, a.seed, first_iter, a.user_offset,
                               keys[slot][0].ptr, vals[slot][0].ptr, 1);
        CU2REC_HIP(hipGetLastError());
        hipcub::DoubleBuffer<uint32_t> dk(keys[slot][0].ptr, keys[slot][1].ptr);
        hipcub::DoubleBuffer<uint64_t> dv(vals[slot][0].ptr, vals[slot][1].ptr);
        size_t bytes = temp_bytes;
        CU2REC_HIP(hipcub::DeviceRadixSort::SortPairs(temp.ptr, bytes, dk, dv, static_cast<int>(n), 0,
                                                      item_bits + b_bits + 1, sched));
        win[slot] = Window{true, a.seed, first_iter, nb, a.user_offset, a.indptr, blocksolve, dk.Current(), dv.Current()};
        hipLaunchKernelGGL(chain_ranges_kernel, dim3(nb), dim3(kBlock), 0, sched, win[slot].sk, n_active, n_range_ranks, item_bits,
                           chain_ranges[slot].ptr, seg, 1);
        CU2REC_HIP(hipGetLastError());
        if (blocksolve)
            bs_launch_plan(win[slot].sk, n_active, nb, n_hot_bs, item_bits, max_blocks, item_of_rank.ptr, chain_begin[slot].ptr,
                           bs_chains[slot].ptr, bs_blocks[slot].ptr, walk_begin[slot].ptr, sched, seg, true, bs_order[slot].ptr);
        CU2REC_HIP(hipEventRecord(ev_ready[slot], sched));
        slot_used[slot] = true;
    };
    // A call usually continues where the last one stopped (cu2rec_train's segments, a bench's steps, a sharded run's periods): its
    // iterations are then already scheduled -- in the window the earlier calls have been running out of, or in the one that was
    // scheduled ahead beside them.  Otherwise (first call, another seed / offset / mode, a jump in the iteration number) a window
    // for what this call needs is scheduled now, and the call's first iteration waits for it.
    auto holds = [&](const Window &w, uint64_t it) {
        return w.valid && w.seed == a.seed && w.user_offset == a.user_offset && w.indptr == a.indptr && w.blocksolve == blocksolve &&
               it >= w.iter_begin && it < w.iter_begin + static_cast<uint64_t>(w.nb);
    };
    if (blocksolve) {  // the other two streams of the mode start behind everything the caller has queued so far
        CU2REC_HIP(hipEventRecord(ev_call, stream));
        CU2REC_HIP(hipStreamWaitEvent(solve, ev_call, 0));
        CU2REC_HIP(hipStreamWaitEvent(upd, ev_call, 0));
    }
    bool upd_pending = false;  // phase 3 of an earlier iteration of THIS call has not been waited for by `stream` yet
    int done = 0;
    while (done < n_iters) {
        const uint64_t it_first = iter0 + static_cast<uint64_t>(done);
        int slot = holds(win[0], it_first) ? 0 : (holds(win[1], it_first) ? 1 : -1);
        if (slot < 0) {
            // the CSR / model pointers the schedule reads must be final: whatever the caller queued on `stream` so far
            CU2REC_HIP(hipEventRecord(ev_consumed[1], stream));
            CU2REC_HIP(hipStreamWaitEvent(sched, ev_consumed[1], 0));
            slot_used[1] = true;  // (that record also stands for "slot 1 free": nothing of it is queued behind this point)
            win[1].valid = false;
            schedule(0, it_first, std::min(max_batch, n_iters - done));
            slot = 0;
        }
        const int off = static_cast<int>(it_first - win[slot].iter_begin);  // the window's iterations [off, off + nb) are this batch
        const int nb = std::min(n_iters - done, win[slot].nb - off);
        // The window BEHIND this one is scheduled beside this batch's iterations unless it exists already: as far as this call goes
        // on, or -- for the calls to come (`speculate`) -- a whole max_batch.  It is queued (on its own stream) BEHIND this batch's
        // first iterations in host order: queueing it costs the host tens of microseconds (the segmented sort's launches), and at the
        // start of a call the device is idle -- with the schedule queued first, the call's first phase 1 started ~100 us after the
        // call did (kernel traces, round 4).
        const uint64_t next_begin = win[slot].iter_begin + static_cast<uint64_t>(win[slot].nb);
        const long long call_beyond = static_cast<long long>(iter0 + static_cast<uint64_t>(n_iters)) - static_cast<long long>(next_begin);
        const int next_nb = speculate ? max_batch : static_cast<int>(std::min<long long>(max_batch, call_beyond));
        bool next_scheduled = (holds(win[slot ^ 1], next_begin) && win[slot ^ 1].iter_begin == next_begin) || next_nb <= 0;
        auto schedule_next = [&] {
            if (next_scheduled) return;
            next_scheduled = true;
            schedule(slot ^ 1, next_begin, next_nb);
        };
        CU2REC_HIP(hipStreamWaitEvent(stream, ev_ready[slot], 0));
        const uint32_t *sk = win[slot].sk + static_cast<size_t>(off) * seg;   // (the batch's first iteration inside the window)
        const uint64_t *sv = win[slot].sv + static_cast<size_t>(off) * seg;
        a.iters = 1;
        if (blocksolve) {
            // One iteration = four launches: phase 1 (bs_gram_kernel: every block's inverse factor), phase 2 (bs_solve_kernel: one
            // workgroup per chain), phase 3 (bs_update_kernel: the user side), and the other items' chains (two-wave form / walk).
            // The kernels hand blocks over through device words tagged with the iteration's epoch (blocksolve.hpp), so they are
            // correct in EITHER launch topology:
            //  * sequential (default): `stream`: phase 1 -> phase 2 -> phase 3; second stream: the other chains, forked behind
            //    phase 1 and joined in front of the next phase 1.  Every producer has finished before its consumer starts: the
            //    device-side waits fall through.
            //  * concurrent (CU2REC_BS_CONCURRENT=1, opt-in): `stream`: phase 1 -> the other chains; `solve`: phase 2, which
            //    starts on a block as soon as phase 1 has announced it; second stream: phase 3 behind phase 1, its workgroups
            //    waiting chain by chain for phase 2's progress.  Measured at parity with the sequential topology (DESIGN.md
            //    section 4: the chain stays the critical path) and it NEEDS the three streams to be three hardware queues -- the
            //    HIP runtime multiplexes streams onto a small pool of queues, and two of these launches in one queue would wait
            //    for each other until the bounded waits give up -- so it is not the default.
            static const bool concurrent = std::getenv("CU2REC_BS_CONCURRENT") != nullptr && std::atoi(std::getenv("CU2REC_BS_CONCURRENT")) != 0;
            CU2REC_HIP(hipStreamWaitEvent(solve, ev_ready[slot], 0));  // (concurrent: phase 2; pipelined: phase 3)
            CU2REC_HIP(hipStreamWaitEvent(upd, ev_ready[slot], 0));
            bool pipe_used = false;
            const double la = std::log2(1.0 - static_cast<double>(a.h.lr) * static_cast<double>(a.h.q_reg));
            const double lc = std::log2(1.0 - static_cast<double>(a.h.lr) * static_cast<double>(a.h.ib_reg));
            unsigned *status = bs_status_word();
            // chains of at least this many blocks take the affine form (sequential topology, n_factors <= 124)
            // ... with their first aff_head blocks in the plain form (what the maps of the others take to build) in one workgroup and
            // the rest in another (at most aff_tails chains), and only as many blocks as the launch of phase 2 can hold a workgroup
            // for beside those: every workgroup of that launch must find a CU without another one of them leaving (the chains'
            // second workgroups wait for the builders and for the first ones)
            const int aff_head = blocksolve_affine_head(0);
            const int aff_want = blocksolve_affine_blocks(-1);
            const int aff_tails = std::min(n_hot_bs, 16);
            const int aff_cap = std::min(std::max(bs_compute_units() - n_hot_bs - aff_tails, 0), max_blocks);
            const int aff_min = (!concurrent && aff_want > 0 && bs_affine_supported(a.nslots) && aff_cap > aff_head + 1) ? std::max(aff_want, aff_head + 2) : 0;
            if (aff_min > 0 && aff_nslots != a.nslots) {
                CU2REC_HIP(hipStreamSynchronize(stream));
                Tbuf.allocate(static_cast<size_t>(max_blocks) * bs_affine_t_floats(a.nslots));
                Wbuf.allocate(static_cast<size_t>(max_blocks) * bs_affine_w_floats(a.nslots));
                bstart.allocate(max_blocks);
                hstate.allocate(static_cast<size_t>(std::max(n_hot_bs, 1)) * 128);
                hstate.zero();  // epochs start at 1
                aff_nslots = a.nslots;
            }
            for (int b = 0; b < nb; ++b) {
                if (b == std::min(2, nb - 1) && b > 0) schedule_next();  // (two iterations are queued: the device has work)
                a.iter0 = iter0 + done + b;
                if (++bs_epoch == 0) ++bs_epoch;
                BsIteration it{};
                it.keys = sk + static_cast<size_t>(b) * seg;
                it.vals = sv + static_cast<size_t>(b) * seg;
                it.n_active = n_active;
                it.n_hot = n_hot_bs;
                it.item_mask = item_mask;
                it.chains = bs_chains[slot].ptr + static_cast<size_t>(off + b) * std::max(n_hot_bs, 1);
                it.blocks = bs_blocks[slot].ptr + static_cast<size_t>(off + b) * max_blocks;
                it.order = bs_order[slot].ptr + static_cast<size_t>(off + b) * max_blocks;
                it.pipe = 0;
                it.pipe_done = pipe_done.ptr;
                it.pipe_target = 0;
                it.walk_begin = walk_begin[slot].ptr + off + b;
                it.item_of_rank = item_of_rank.ptr;
                it.tables = tables.ptr;
                it.log2a = static_cast<float>(la);
                it.log2c = static_cast<float>(lc);
                it.Mbuf = Mbuf.ptr;
                it.ebuf = ebuf.ptr;
                it.qstart = qstart.ptr;
                it.max_blocks = max_blocks;
                // (sequential topology, rows the rings have LDS for; the affine form and the look-ahead form exclude each other)
                const bool la_on = !concurrent && aff_min == 0 && la_ranks > 0 && bs_lookahead_supported(a.nslots);
                it.la_ranks = la_on ? la_ranks : 0;
                it.la_cap = la_cap;
                it.la_grid = la_on ? la_cap : 0;
                it.Nbuf = Nbuf.ptr;
                it.aff_min_blocks = aff_min;
                it.aff_head = aff_head;
                it.aff_cap = aff_cap;
                it.aff_flag = gram_flag.ptr;
                it.aff_tails = aff_tails;
                it.hstate = hstate.ptr;
                it.Tbuf = Tbuf.ptr;
                it.Wbuf = Wbuf.ptr;
                it.bstart = bstart.ptr;
                it.concurrent = concurrent ? 1 : 0;
                it.epoch = bs_epoch;
                it.gram_flag = gram_flag.ptr;
                it.chain_prog = chain_prog.ptr;
                it.status = status;
                it.solve_started = solve_started.ptr;
                it.gram_done = nullptr;
                it.side_seq = nullptr;
                it.side_target = 0;
                it.wait_ticks = bs_wait_ticks();
                bs_get_stamps(&it.stamps, &it.stamps_cap);
                static const int dbg = std::getenv("CU2REC_BS_DBG") ? std::atoi(std::getenv("CU2REC_BS_DBG")) : 0;  // timing experiments
                it.dbg = dbg;
                if (!concurrent) {
                    it.started_target = 0;  // phase 2 is queued behind phase 1: nothing to wait for
                    static const int gate_mode = bs_gate_mode();
                    const bool device_join = gate_mode >= 2 && n_hot_bs > 0 && max_blocks > 0 && std::getenv("CU2REC_BS_MARKERS") == nullptr;
                    if (upd_pending && !device_join) {  // join: the previous iteration's other chains may have written rows phase 1 reads
                        CU2REC_HIP(hipStreamWaitEvent(stream, ev_upd, 0));
                    }  // (device_join: the previous iteration's phase 3 has waited for them, see below)
                    // The other chains touch other items and other users than the hot ones: fork, run beside, join.  The fork
                    // sits behind phase 1: workgroups of the side kernel that already fill the CUs would keep phase 1's waiting
                    // (measured: 71 instead of 19 us).  Both events ride on their kernels' completion signals (CU2REC_BS_MARKERS=1:
                    // separate hipEventRecord markers as in round 2, for comparison).
                    static const bool markers = std::getenv("CU2REC_BS_MARKERS") != nullptr;
                    static const bool gate = bs_gate_mode() != 0;  // CU2REC_BS_GATE=0: the event
                    if (gate && !markers && n_hot_bs > 0 && max_blocks > 0) {  // (phase 1 is launched at all)
                        // The fork without an event: phase 1's workgroups count themselves through, and ONE wavefront queued in front
                        // of the side kernel (after phase 1 in host order: streams sharing a hardware queue serialise, they cannot
                        // wait for each other) ends when the whole grid is through.  Phase 1 is a plain launch then: phase 2 starts
                        // right behind it (an event riding on its completion signal costs 5 us before phase 2).
                        // (the host's counts move only behind a launch that succeeded: a launch that throws leaves gate and counters
                        // in step for the calls that follow)
                        it.gram_done = gram_done.ptr;
                        bs_launch_gram(a, it, stream);
                        gram_done_target += static_cast<unsigned long long>(max_blocks + it.la_grid);
                    } else {
                        bs_launch_gram(a, it, stream, markers ? nullptr : ev_gram);
                        if (markers) CU2REC_HIP(hipEventRecord(ev_gram, stream));
                        CU2REC_HIP(hipStreamWaitEvent(upd, ev_gram, 0));
                    }
                    // Pipelined (CU2REC_BS_PIPE=1, opt-in): phase 3 runs BESIDE phase 2 as a grid of persistent workgroups on the third
                    // stream, behind a gate of its own (phase 2's workgroups hold their CUs: a CU filled with waiting phase-3 workgroups
                    // would have no room for a chain), taking each block as soon as its chain has announced it; what is left of phase 3
                    // behind the longest chain is its last block, and the join -- with phase 3 AND the side kernel -- is one more
                    // workgroup in phase 2's launch.  Measured SLOWER than phase 3 behind phase 2 (round 4, ML-20M shape: 88.5 against
                    // 82.4 us per step): phase 3 moves 37 MB and the side kernel, which it then runs beside, is bound by the same memory
                    // system (57.6 instead of 44.9 us) -- DESIGN.md section 4.
                    static const bool pipe_ok = std::getenv("CU2REC_BS_PIPE") && std::atoi(std::getenv("CU2REC_BS_PIPE")) != 0;
                    const bool pipe = pipe_ok && device_join && it.gram_done && aff_min == 0 && bs_pipe_supported(a.nslots);
                    const int pipe_grid = std::min(max_blocks, blocksolve_pipe_grid(0));
                    // The join: a signal kernel behind the side kernel, and one more workgroup in the main stream's last launch of the
                    // iteration that waits for its word (the number is known before either is queued: the wait is on the device) -- the
                    // next phase 1 follows that launch without an event.
                    // (CU2REC_BS_DBG & 16, fault-path test: the signal is never sent -- the waiting workgroup gives up after
                    // 15 x the bound, the status word is set and the next entry point returns CU2REC_EHIP)
                    if (device_join) {
                        it.side_seq = side_seq.ptr;
                        it.side_target = side_seq_host + 1;
                    }
                    if (pipe) {
                        it.pipe = 1;
                        it.pipe_target = pipe_done_host + static_cast<unsigned long long>(pipe_grid);
                    }
                    bs_launch_solve(a, it, stream);
                    if (it.gram_done) started_total += static_cast<unsigned long long>(n_hot_bs + (pipe ? 1 : 0));  // (counted by its workgroups only then)
                    // (the gate behind phase 2 in host order: it also waits for phase 2's workgroups to hold their CUs)
                    // (CU2REC_BS_DBG & 32, fault-path test: a gate that can never be satisfied -- it gives up after the bound and the run
                    // goes on, results unchanged)
                    if (it.gram_done)
                        bs_launch_gate(gram_done.ptr, gram_done_target + ((dbg & 32) ? (1ull << 40) : 0ull), solve_started.ptr, started_total, upd);
                    // (the side kernel's completion event is what the main stream waits for at the END of the batch; with the device-side
                    // join only the batch's last side kernel carries it -- an event on a kernel's completion signal holds the next packet
                    // of its queue back, here the signal kernel: 5.4 us per iteration in the kernel traces of round 4)
                    const bool side_event = !markers && (!device_join || b == nb - 1);
                    launch_chains(a, it.keys, it.vals, n_active, item_of_rank.ptr, item_mask, segmented ? 0u : static_cast<uint32_t>(off + b) << item_bits,
                                  std::max(n_duo_bs, n_hot_bs), upd, n_hot_bs, kRoleDuo | kRoleWalk,
                                  chain_ranges[slot].ptr + static_cast<size_t>(off + b) * (n_range_ranks + 1), side_event ? ev_upd : nullptr);
                    if (markers) CU2REC_HIP(hipEventRecord(ev_upd, upd));
                    if (device_join) {
                        ++side_seq_host;
                        if (!(dbg & 16)) bs_launch_signal(side_seq.ptr, side_seq_host, upd);
                    }
                    if (pipe) {
                        bs_launch_gate(gram_done.ptr, gram_done_target, solve_started.ptr, started_total, solve);
                        bs_launch_update_pipe(a, it, pipe_grid, solve);
                        pipe_done_host += static_cast<unsigned long long>(pipe_grid);
                        pipe_used = true;
                    } else {
                        bs_launch_update(a, it, stream);
                    }
                    upd_pending = true;
                    continue;
                }
                // phase 2 first: its few workgroups take their CUs before anything else of this iteration asks for room
                started_total += static_cast<unsigned long long>(bs_solve_grid(n_hot_bs));
                it.started_target = started_total;
                bs_launch_solve(a, it, solve);
                // phase 1 reads user rows the previous iteration's phase 3 may have written
                if (upd_pending) {
                    CU2REC_HIP(hipStreamWaitEvent(stream, ev_upd, 0));
                }
                bs_launch_gram(a, it, stream);
                CU2REC_HIP(hipEventRecord(ev_gram, stream));
                // The other chains (ranks [n_hot_bs, n_duo_bs): the ordered mode's two-wave form; beyond: its walk): other items,
                // other users.  On the SAME stream, behind phase 1: beside it their thousands of workgroups take the CUs away from
                // phase 1's (measured: 71 instead of 19 us), and a cross-stream edge costs 13-14 us each way on this runtime
                // (phase 1 -> other chains -> next phase 1 on two streams: 106 us per iteration, 27 of them event latency).
                launch_chains(a, it.keys, it.vals, n_active, item_of_rank.ptr, item_mask, segmented ? 0u : static_cast<uint32_t>(off + b) << item_bits,
                              std::max(n_duo_bs, n_hot_bs), stream, n_hot_bs, kRoleDuo | kRoleWalk,
                              chain_ranges[slot].ptr + static_cast<size_t>(off + b) * (n_range_ranks + 1));
                // phase 3 on a stream of its own, behind phase 1 (hence behind the previous iteration's other chains, whose rows
                // it may rewrite, and behind the start of every phase-2 workgroup): its workgroups wait for the chains' progress
                CU2REC_HIP(hipStreamWaitEvent(upd, ev_gram, 0));
                bs_launch_update(a, it, upd);
                CU2REC_HIP(hipEventRecord(ev_upd, upd));
                upd_pending = true;
            }
            schedule_next();  // (a batch of one or two iterations)
            CU2REC_HIP(hipGetLastError());
            // the batch's slot is free once all three streams are through with it
            if (concurrent || pipe_used) {
                CU2REC_HIP(hipEventRecord(ev_solve, solve));
                CU2REC_HIP(hipStreamWaitEvent(stream, ev_solve, 0));
            }
            CU2REC_HIP(hipStreamWaitEvent(stream, ev_upd, 0));
            upd_pending = false;  // (waited for)
            CU2REC_HIP(hipEventRecord(ev_consumed[slot], stream));
            done += nb;
            continue;
        }
        // a small set: the batch's iterations in ONE persistent launch with a grid barrier where the kernel boundaries were
        // (sgd_ordered_persistent_kernel).  OPT-IN (CU2REC_ORDERED_PERSISTENT=1): measured SLOWER than the launch per iteration on
        // the ML-1M shape, 18.1 against 15.6 us per iteration (round 4) -- a grid barrier across eight XCDs (two atomic hops, an L2
        // write-back per XCD, an L1 invalidate per CU) costs more than the kernel boundary it replaces
        static const bool persistent_ok = std::getenv("CU2REC_ORDERED_PERSISTENT") && std::atoi(std::getenv("CU2REC_ORDERED_PERSISTENT")) != 0;
        const bool persistent = persistent_ok && segmented && nb >= 2 &&
                                launch_chains_persistent(a, sk, sv, seg, n_active, item_of_rank.ptr, item_mask, n_hot, 0,
                                                         n_hot <= n_range_ranks ? chain_ranges[slot].ptr + static_cast<size_t>(off) * (n_range_ranks + 1) : nullptr,
                                                         n_range_ranks + 1, nb, stream);
        for (int b = 0; b < nb && !persistent; ++b) {
            if (b == std::min(2, nb - 1) && b > 0) schedule_next();
            a.iter0 = iter0 + done + b;
            const uint32_t *kb = sk + static_cast<size_t>(b) * seg;
            const uint64_t *vb = sv + static_cast<size_t>(b) * seg;
            const uint32_t key_base = segmented ? 0u : static_cast<uint32_t>(off + b) << item_bits;
            launch_chains(a, kb, vb, n_active, item_of_rank.ptr, item_mask, key_base, n_hot, stream, 0, kRoleDuo | kRoleWalk,
                          n_hot <= n_range_ranks ? chain_ranges[slot].ptr + static_cast<size_t>(off + b) * (n_range_ranks + 1) : nullptr);
        }
        schedule_next();
        CU2REC_HIP(hipGetLastError());
        CU2REC_HIP(hipEventRecord(ev_consumed[slot], stream));
        done += nb;
    }
    // (raw pointers: the arrays behind them are the caller's and may change between calls -- no window outlives the call)
    if (!speculate) win[0].valid = win[1].valid = false;
    if (blocksolve) bs_report_status(stream);  // (every batch ended with `stream` behind the other two streams)
    CU2REC_HIP(hipEventRecord(ev_last, stream));
    last_stream = stream;
    have_last = true;
}

}  // namespace cu2rec
