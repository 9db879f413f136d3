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
//      RANK, the longest chains come first.
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

// first position in keys[0, n) whose key is >= target
__device__ __forceinline__ int lower_bound_key(const uint32_t *__restrict__ keys, int n, uint32_t target) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (keys[mid] < target) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Keys and values of a window, 64 users x up to 64 iterations per workgroup, with the users' RATING ROWS read once instead of one
// random access per sample (two cache lines -- indices, data -- for every (user, iteration): 2.0 GB per batch of 64 iterations on
// the ML-20M shape, which the iterations running beside it paid for: tools/schedule_interference.py).  Here a wavefront
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
                                                                 uint64_t *__restrict__ vals) {
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
            const uint32_t key = live[j] ? static_cast<uint32_t>(rank[j]) : sentinel;
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
__global__ __launch_bounds__(kBlock) void chain_ranges_kernel(const uint32_t *__restrict__ keys, int n_active, int n_ranks,
                                                              int *__restrict__ ranges, size_t stride) {
    const int b = blockIdx.x;
    const uint32_t *kb = keys + static_cast<size_t>(b) * stride;
    for (int r = threadIdx.x; r <= n_ranks; r += kBlock)
        ranges[static_cast<size_t>(b) * (n_ranks + 1) + r] = lower_bound_key(kb, n_active, static_cast<uint32_t>(r));
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

#ifndef CU2REC_WALK_NT
#define CU2REC_WALK_NT 0  // tuning builds (tools/build_variant.sh): 1 user rows of the chains' kernel STORED non-temporally, 2 loaded so, 3 both
#endif
// a user row is written once per iteration and not read again before the next one: the streaming form keeps it from displacing item
// rows in the XCD's L2 (measured on the Netflix shape in round 6: profiles/r06_netflix_walk_experiments.txt)
template <int W, int S>
__device__ __forceinline__ void store_user_wide_row(float *__restrict__ base, size_t row, int ld, int nslots, int lane, const WideRow<S> &r) {
#if CU2REC_WALK_NT & 1
    float4 *p = reinterpret_cast<float4 *>(base + row * static_cast<size_t>(ld));
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int slot = lane + W * s;
        if (slot < nslots) {
            const vec4f v = {r.v[s].x, r.v[s].y, r.v[s].z, r.v[s].w};
            __builtin_nontemporal_store(v, reinterpret_cast<vec4f *>(&p[slot]));
        }
    }
#else
    store_wide_row<W, S>(base, row, ld, nslots, lane, r);
#endif
}

template <int J>
__device__ __forceinline__ void store_user_row(float *__restrict__ base, size_t row, int ld, int nslots, int lane, const Row<J> &r) {
#if CU2REC_WALK_NT & 1
    store_row_stream<J>(base, row, ld, nslots, lane, r);
#else
    store_row<J>(base, row, ld, nslots, lane, r);
#endif
}

template <int J>
__device__ __forceinline__ Row<J> load_user_row(const float *__restrict__ base, size_t row, int ld, int nslots, int lane) {
#if CU2REC_WALK_NT & 2
    return load_row_stream<J>(base, row, ld, nslots, lane);
#else
    return load_row<J>(base, row, ld, nslots, lane);
#endif
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
            store_user_wide_row<W, S>(a.P, static_cast<size_t>(l.user[buf][e]), a.ldp, a.nslots, lane, pn);
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
                                                  const uint64_t *__restrict__ vals, int n_active,
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
                begin = lower_bound_key(keys, n_active, static_cast<uint32_t>(r));
                end = lower_bound_key(keys, n_active, static_cast<uint32_t>(r + 1));
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
            cur[d] = load_user_row<J>(a.P, static_cast<size_t>(x), a.ldp, a.nslots, lane);
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
                nxt[d] = load_user_row<J>(a.P, static_cast<size_t>(x), a.ldp, a.nslots, lane);
                nub[d] = a.user_bias[x];
            }
#pragma unroll
            for (int d = 0; d < D; ++d) {
                if (b0 + d < nw) {
                    const uint64_t val = link(myval, b0 + d, nw);
                    const int x = static_cast<int>(val >> 32);
                    const float new_ub = chain_step<J>(a, cur[d], q, cub[d], ib, __uint_as_float(static_cast<uint32_t>(val)));
                    store_user_row<J>(a.P, static_cast<size_t>(x), a.ldp, a.nslots, lane, cur[d]);
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
                                                             uint32_t item_mask, int n_hot,
                                                             int hot_blocks, int rank_lo, const int *__restrict__ ranges,
                                                             int walk_lo, int walk_blocks) {
    // above the schedule's kernels (priority 0), which run BESIDE the iterations: where a wavefront of theirs shares a SIMD with one
    // of an iteration's, the iteration's goes first (a kernel ends with its slowest workgroup: profiles/r05_handwritten_sort_experiment.txt)
    __builtin_amdgcn_s_setprio(1);
    if (static_cast<int>(blockIdx.x) < hot_blocks) {
        // two-wave form for every row width: 32 lanes x 1 slot when 65 <= ld <= 128, else 16 lanes x J slots
        run_hot_block_duo<(J == 2 ? kWide : kGroup), (J == 2 ? 1 : J)>(a, keys, vals, n_active, n_hot, item_of_rank, rank_lo, ranges);
        return;
    }
    // The walk: one 16-lane group per sorted position of [walk_lo, n_active) -- the tail of the sorted iteration, where the chains of
    // the unpopular items are (OrderedSchedule::walk_bound: the host knows how long that tail is, give or take a few hundred positions;
    // the positions in front of it belong to two-wave chains and used to cost a workgroup each just to find that out: 84 % of the walk's
    // workgroups on the Netflix shape).  One extra workgroup sweeps whatever walked chains an iteration has in FRONT of walk_lo.
    const int wb = static_cast<int>(blockIdx.x) - hot_blocks;
    if (wb < walk_blocks) {
        walk_group<J>(a, keys, vals, n_active, item_of_rank, item_mask, n_hot, walk_lo + (wb * kBlock + static_cast<int>(threadIdx.x)) / kGroup);
        return;
    }
    const int sweep_lo = ranges[n_hot];  // first sorted position whose rank is >= n_hot (chain_ranges_kernel)
    for (int p = sweep_lo + static_cast<int>(threadIdx.x) / kGroup; p < walk_lo; p += kGroupsPerBlock)
        walk_group<J>(a, keys, vals, n_active, item_of_rank, item_mask, n_hot, p);
}

// One iteration's chains of ranks >= rank_lo: the two-wave chains of ranks [rank_lo, n_hot) and the walk of the rest, one launch.
template <int J>
void launch_chain(const SgdArgs &a, const uint32_t *keys, const uint64_t *vals, int n_active, const int *item_of_rank,
                  uint32_t item_mask, int n_hot, hipStream_t stream, int rank_lo, const int *ranges, hipEvent_t stop, int walk_bound) {
    const int chains_per_block = DuoShape<(J == 2 ? kWide : kGroup), (J == 2 ? 1 : J)>::kChains;
    const int hot_blocks = (std::max(n_hot - rank_lo, 0) + chains_per_block - 1) / chains_per_block;
    // the walk covers the last walk_bound positions (whole workgroups); with the chains' ranges at hand one more workgroup sweeps what an
    // iteration may have in front of them (see the kernel); without them (ranges == nullptr) the walk covers every position
    int walk_lo = 0;
    if (ranges && walk_bound < n_active) walk_lo = ((n_active - walk_bound) / kGroupsPerBlock) * kGroupsPerBlock;
    const int walk_blocks = (n_active - walk_lo + kGroupsPerBlock - 1) / kGroupsPerBlock;
    const int blocks = hot_blocks + walk_blocks + (walk_lo > 0 ? 1 : 0);
    if (blocks == 0) return;
    if (stop)  // the event rides on the kernel's completion signal (see bs_launch_gram)
        hipExtLaunchKernelGGL(sgd_ordered_kernel<J>, dim3(blocks), dim3(kBlock), 0, stream, nullptr, stop, 0, a, keys, vals, n_active,
                              item_of_rank, item_mask, n_hot, hot_blocks, rank_lo, ranges, walk_lo, walk_blocks);
    else
        hipLaunchKernelGGL(sgd_ordered_kernel<J>, dim3(blocks), dim3(kBlock), 0, stream, a, keys, vals, n_active,
                           item_of_rank, item_mask, n_hot, hot_blocks, rank_lo, ranges, walk_lo, walk_blocks);
}

void launch_chains(const SgdArgs &a, const uint32_t *kb, const uint64_t *vb, int n_active, const int *item_of_rank,
                   uint32_t item_mask, int n_hot, hipStream_t stream, int rank_lo, const int *ranges, hipEvent_t stop, int walk_bound) {
    switch (slots_per_lane(a.nslots)) {
        case 1: launch_chain<1>(a, kb, vb, n_active, item_of_rank, item_mask, n_hot, stream, rank_lo, ranges, stop, walk_bound); break;
        case 2: launch_chain<2>(a, kb, vb, n_active, item_of_rank, item_mask, n_hot, stream, rank_lo, ranges, stop, walk_bound); break;
        case 3: launch_chain<3>(a, kb, vb, n_active, item_of_rank, item_mask, n_hot, stream, rank_lo, ranges, stop, walk_bound); break;
        case 4: launch_chain<4>(a, kb, vb, n_active, item_of_rank, item_mask, n_hot, stream, rank_lo, ranges, stop, walk_bound); break;
        case 5: launch_chain<5>(a, kb, vb, n_active, item_of_rank, item_mask, n_hot, stream, rank_lo, ranges, stop, walk_bound); break;
        case 6: launch_chain<6>(a, kb, vb, n_active, item_of_rank, item_mask, n_hot, stream, rank_lo, ranges, stop, walk_bound); break;
        case 7: launch_chain<7>(a, kb, vb, n_active, item_of_rank, item_mask, n_hot, stream, rank_lo, ranges, stop, walk_bound); break;
        case 8: launch_chain<8>(a, kb, vb, n_active, item_of_rank, item_mask, n_hot, stream, rank_lo, ranges, stop, walk_bound); break;
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
    rank_rate_prefix.assign(static_cast<size_t>(n_cols) + 1, 0.0);
    for (int r = 0; r < n_cols; ++r) rank_rate_prefix[r + 1] = rank_rate_prefix[r] + rate[order[r]];
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
    temp_bytes = 0;
    {
        // every iteration of a batch is a segment of n_rows pairs of the sort
        std::vector<int> offs(static_cast<size_t>(max_batch) + 1);
        for (int b = 0; b <= max_batch; ++b) offs[b] = static_cast<int>(static_cast<size_t>(b) * n_rows);
        seg_offsets.allocate(offs.size());
        seg_offsets.upload(offs.data(), offs.size());
        CU2REC_HIP(rocprim::segmented_radix_sort_pairs<SegSortConfig>(nullptr, temp_bytes, keys[0][0].ptr, keys[0][1].ptr, vals[0][0].ptr,
                                                                       vals[0][1].ptr, static_cast<unsigned>(cap), static_cast<unsigned>(max_batch),
                                                                       seg_offsets.ptr, seg_offsets.ptr + 1, 0u, 32u, nullptr));
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
        // a small set (one GPU's shard of a strong-scaling run): with the hottest item between 150 and 350 expected updates per
        // iteration a threshold of 60 still pays a little (an eighth of the ML-20M shape, hottest item 247: 32.5 us per iteration
        // against 35.3 walking everything; thresholds 40 / 60 / 90 / 130 / 180: 32.6 / 32.5 / 33.7 / 37.5 / 41.4, round 5); below
        // that no chain is long enough to pay for three launches per iteration and the mode IS the ordered walk (ML-1M shape)
        // Only for SMALL sets (at most a quarter of 131,072 users: where the rule was measured).  A large set with flat popularity --
        // many users, hottest item under 350 -- keeps the scaled threshold: there "every item above 60" would be hundreds of
        // 512-thread solve workgroups the side stream's gate has to see started (the regime that did not fit the CUs on the
        // Netflix shape), and nothing was measured there.
        if (n_cols > 0 && n_active <= 131072 / 4 && rate[order[0]] < 350.0)
            min_rate = rate[order[0]] >= 150.0 ? 60.f : std::numeric_limits<float>::infinity();
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
    if (upd) (void)hipStreamDestroy(upd);
    for (hipEvent_t e : {ev_call, ev_upd, ev_gram, ev_last})
        if (e) (void)hipEventDestroy(e);
}

namespace {
std::atomic<int> g_bs_lookahead{-1};  // -1: not yet initialised from the environment
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

// How many sorted positions the walk behind popularity rank n_hot has to cover in one iteration, as an upper bound that holds in all but
// astronomically rare iterations: a user lands behind rank n_hot with probability 1 - (its ratings of the n_hot most popular items) / (its
// ratings), independently; the count's mean is n_active - rank_rate_prefix[n_hot] and its variance at most that mean.  Mean + 8 sigma + 64.
// The launch covers exactly that many positions at the END of the sorted iteration (where the unpopular items' chains are); whatever an
// iteration has beyond it is swept by one extra workgroup (sgd_ordered_kernel), so the bound is about speed, never about results.
int OrderedSchedule::walk_bound(int n_hot) const {
#ifdef CU2REC_TEST_HOOKS  // (fault-path tests: a bound far too small, so that the sweeper does nearly all of the walk)
    if (const char *env = std::getenv("CU2REC_BS_DBG"))
        if (std::atoi(env) & 256) return 16;
#endif
    const int r = std::min(std::max(n_hot, 0), n_cols);
    const double mean = std::max(0.0, static_cast<double>(n_active) - rank_rate_prefix[static_cast<size_t>(r)]);
    const double bound = mean + 8.0 * std::sqrt(mean) + 64.0;
    return bound >= static_cast<double>(n_active) ? n_active : static_cast<int>(bound);
}

// Blocks of 64 links an iteration's hot chains are expected to fill, as a bound that holds in all but astronomically rare iterations
// (the hot links' count: mean rank_rate_prefix[n_hot_bs], variance at most the mean; one partial block per chain): what phase 3 is
// launched with.  Its workgroups stride through the dense block table, so an iteration beyond the bound is still done completely.
int OrderedSchedule::blocks_bound() const {
#ifdef CU2REC_TEST_HOOKS  // (fault-path tests: a launch far too small, so that every workgroup strides over many blocks)
    if (const char *env = std::getenv("CU2REC_BS_DBG"))
        if (std::atoi(env) & 512) return std::min(3, max_blocks);
#endif
    const double mean = rank_rate_prefix[static_cast<size_t>(std::min(std::max(n_hot_bs, 0), n_cols))];
    const double bound = (mean + 8.0 * std::sqrt(mean) + 64.0) / kBsLinks + n_hot_bs + 1;
    return std::max(1, bound >= static_cast<double>(max_blocks) ? max_blocks : static_cast<int>(bound));
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
        if (!upd) {
            // default priority: a high-priority queue whose workgroups cannot all be placed (more hot chains than CUs) held back
            // the queues below it (a 300-chain test set ran into the waits' timeout)
            CU2REC_HIP(hipStreamCreateWithFlags(&upd, hipStreamNonBlocking));
            for (hipEvent_t *e : {&ev_call, &ev_upd, &ev_gram}) CU2REC_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
        }
    }
    const uint32_t item_mask = (1u << item_bits) - 1u;
    // ranks [0, n_hot) get a dedicated group each with deep prefetch; keep r + 1 representable in item_bits
    const int n_hot = std::min(std::min(n_cols, kHotChains), static_cast<int>(item_mask));
    // positions the walk covers per iteration (its tail of the sorted order): behind the two-wave chains of either mode
    const int walk_bound_plain = walk_bound(n_hot), walk_bound_bs = walk_bound(std::max(n_duo_bs, n_hot_bs));
    const int launch_blocks = blocksolve ? blocks_bound() : 0;
    // Sample + sort (+ plan) of one window into a slot, on the schedule stream: nothing in it depends on the model.  The keys carry
    // the item's popularity rank only, and every iteration of the window is sorted as a segment of its own (stable, so users stay
    // ascending inside a chain); iteration b of the window occupies [b * n_rows, b * n_rows + n_active) of the sorted arrays.
    const size_t seg = static_cast<size_t>(n_rows);  // an iteration's stride in the sorted arrays
    auto schedule = [&](int slot, uint64_t first_iter, int nb) {
        win[slot].valid = false;
        // (the window is published only when everything below has been queued -- a launch that throws leaves none behind)
        const size_t n = static_cast<size_t>(nb) * n_rows;
        if (slot_used[slot]) CU2REC_HIP(hipStreamWaitEvent(sched, ev_consumed[slot], 0));  // its last window has been run
        hipLaunchKernelGGL(schedule_keys_tile_kernel, dim3((n_rows + 63) / 64), dim3(256), 0, sched, a.indptr, a.indices, a.data,
                           item_rank.ptr, n_rows, nb, item_bits, 1u << item_bits, a.seed, first_iter, a.user_offset, keys[slot][0].ptr,
                           vals[slot][0].ptr);
        CU2REC_HIP(hipGetLastError());
        size_t bytes = temp_bytes;
        CU2REC_HIP(rocprim::segmented_radix_sort_pairs<SegSortConfig>(temp.ptr, bytes, keys[slot][0].ptr, keys[slot][1].ptr, vals[slot][0].ptr,
                                                                       vals[slot][1].ptr, static_cast<unsigned>(n), static_cast<unsigned>(nb),
                                                                       seg_offsets.ptr, seg_offsets.ptr + 1, 0u,
                                                                       static_cast<unsigned>(item_bits + 1), sched));
        Window w{true, a.seed, first_iter, nb, a.user_offset, a.indptr, blocksolve, keys[slot][1].ptr, vals[slot][1].ptr};
        hipLaunchKernelGGL(chain_ranges_kernel, dim3(nb), dim3(kBlock), 0, sched, w.sk, n_active, n_range_ranks, chain_ranges[slot].ptr, seg);
        CU2REC_HIP(hipGetLastError());
        if (blocksolve)
            bs_launch_plan(w.sk, n_active, nb, n_hot_bs, item_bits, max_blocks, item_of_rank.ptr, chain_begin[slot].ptr,
                           bs_chains[slot].ptr, bs_blocks[slot].ptr, walk_begin[slot].ptr, sched, seg);
        CU2REC_HIP(hipEventRecord(ev_ready[slot], sched));
        slot_used[slot] = true;
        win[slot] = w;
    };
    // (raw pointers: the arrays behind them are the caller's and may change between calls -- no window outlives the call, whether
    // it returns or throws)
    struct DropWindows {
        OrderedSchedule *s;
        ~DropWindows() {
            if (!s->speculate) s->win[0].valid = s->win[1].valid = false;
        }
    } drop_windows{this};
    // A call usually continues where the last one stopped (cu2rec_train's segments, a bench's steps, a sharded run's periods): its
    // iterations are then already scheduled -- in the window the earlier calls have been running out of, or in the one that was
    // scheduled ahead beside them.  Otherwise (first call, another seed / offset / mode, a jump in the iteration number) a window
    // for what this call needs is scheduled now, and the call's first iteration waits for it.  A window is keyed on (seed, user
    // offset, indptr address, mode): with `speculate` (owned cu2rec_csr objects only) the CSR arrays are IMMUTABLE for the
    // object's lifetime, which is what lets a window outlive the call that scheduled it.
    auto holds = [&](const Window &w, uint64_t it) {
        return w.valid && w.seed == a.seed && w.user_offset == a.user_offset && w.indptr == a.indptr && w.blocksolve == blocksolve &&
               it >= w.iter_begin && it < w.iter_begin + static_cast<uint64_t>(w.nb);
    };
    if (blocksolve) {  // the side stream starts behind everything the caller has queued so far
        CU2REC_HIP(hipEventRecord(ev_call, stream));
        CU2REC_HIP(hipStreamWaitEvent(upd, ev_call, 0));
    }
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
        // first iterations in host order: queueing it costs the host tens of microseconds, and at the start of a call the device is
        // idle -- with the schedule queued first, the call's first phase 1 started ~100 us after the call did (kernel traces, round 4).
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
            // One iteration = four launches: `stream`: phase 1 (bs_gram_kernel: every block's inverse factor) -> phase 2
            // (bs_solve_kernel: one workgroup per chain) -> phase 3 (bs_update_kernel: the user side); side stream `upd`: the other
            // items' chains (two-wave form / walk), forked behind phase 1 -- beside it their thousands of workgroups take the CUs
            // from phase 1's (measured: 71 instead of 19 us) -- and joined in front of the next phase 1, which reads rows they write.
            //  * default (bs_topology() == kBsTopoDevice: blocksolve.hip, "the launch topology"), no event on the main stream: phase 1's workgroups count themselves through, phase 2's
            //    count themselves in, ONE wavefront queued in front of the side kernel (bs_gate_kernel) ends when both counts have
            //    reached what the host has launched so far; a signal kernel behind the side kernel stores the iteration's number and
            //    one extra workgroup of phase 3's launch waits for it.  Host order gram, solve, gate, side, signal, update: every
            //    launch behind what it waits for (streams sharing a hardware queue serialise, they cannot wait for each other).
            //  * events (kBsTopoEvents: a counter pass, streams that do not run side by side, after a join that gave up): fork on phase 1's completion signal, join by an event wait in
            //    front of the next phase 1 (5.4 + 2.5-3.3 us per iteration on the main stream: DESIGN.md section 4).
            const double la = std::log2(1.0 - static_cast<double>(a.h.lr) * static_cast<double>(a.h.q_reg));
            const double lc = std::log2(1.0 - static_cast<double>(a.h.lr) * static_cast<double>(a.h.ib_reg));
            unsigned *status = bs_status_word();
            const bool device_join = max_blocks > 0 && bs_topology(stream, upd) == kBsTopoDevice;
            bool upd_pending = false;  // (event join) the side kernel of an earlier iteration of this batch has not been waited for yet
#ifdef CU2REC_TEST_HOOKS  // fault-path tests only (build/test/libcu2rec_amd_hooks.so): 16 the signal is never sent, 32 the gate can never open
            static const int dbg = std::getenv("CU2REC_BS_DBG") ? std::atoi(std::getenv("CU2REC_BS_DBG")) : 0;
#else
            constexpr int dbg = 0;
#endif
            for (int b = 0; b < nb; ++b) {
                if (b == std::min(2, nb - 1) && b > 0) schedule_next();  // (two iterations are queued: the device has work)
                a.iter0 = iter0 + done + b;
                BsIteration it{};
                it.keys = sk + static_cast<size_t>(b) * seg;
                it.vals = sv + static_cast<size_t>(b) * seg;
                it.n_active = n_active;
                it.n_hot = n_hot_bs;
                it.item_mask = item_mask;
                it.chains = bs_chains[slot].ptr + static_cast<size_t>(off + b) * std::max(n_hot_bs, 1);
                it.blocks = bs_blocks[slot].ptr + static_cast<size_t>(off + b) * max_blocks;
                it.walk_begin = walk_begin[slot].ptr + off + b;
                it.item_of_rank = item_of_rank.ptr;
                it.tables = tables.ptr;
                it.log2a = static_cast<float>(la);
                it.log2c = static_cast<float>(lc);
                it.Mbuf = Mbuf.ptr;
                it.ebuf = ebuf.ptr;
                it.qstart = qstart.ptr;
                it.max_blocks = max_blocks;
                it.launch_blocks = launch_blocks;
                const bool la_on = la_ranks > 0 && bs_lookahead_supported(a.nslots);  // (rows the rings have LDS for)
                it.la_ranks = la_on ? la_ranks : 0;
                it.la_cap = la_cap;
                it.la_grid = la_on ? la_cap : 0;
                it.Nbuf = Nbuf.ptr;
                it.status = status;
                it.solve_started = solve_started.ptr;
                it.wait_ticks = bs_wait_ticks();
                bs_get_stamps(&it.stamps, &it.stamps_cap);
                const int *ranges = chain_ranges[slot].ptr + static_cast<size_t>(off + b) * (n_range_ranks + 1);
                if (device_join) {
                    // (the host's counts move only behind a launch that succeeded: a launch that throws leaves gate and counters in
                    // step for the calls that follow)
                    it.gram_done = gram_done.ptr;
                    it.side_seq = side_seq.ptr;
                    it.side_target = side_seq_host + 1;
                    bs_launch_gram(a, it, stream);
                    gram_done_target += static_cast<unsigned long long>(max_blocks + it.la_grid);
                    bs_launch_solve(a, it, stream);
                    started_total += static_cast<unsigned long long>(n_hot_bs);
                    bs_launch_gate(gram_done.ptr, gram_done_target + ((dbg & 32) ? (1ull << 40) : 0ull), solve_started.ptr, started_total, upd);
                    // (only the batch's last side kernel carries the event the main stream waits for at the END of the batch: an event
                    // on a kernel's completion signal holds the next packet of its queue back, here the signal kernel: 5.4 us)
                    launch_chains(a, it.keys, it.vals, n_active, item_of_rank.ptr, item_mask, std::max(n_duo_bs, n_hot_bs), upd, n_hot_bs, ranges,
                                  b == nb - 1 ? ev_upd : nullptr, walk_bound_bs);
                    ++side_seq_host;
                    if (!(dbg & 16)) bs_launch_signal(side_seq.ptr, side_seq_host, upd);
                    bs_launch_update(a, it, stream);
                } else {
                    // (test build, TIMING ONLY, results wrong: 64 the side kernel alone, 128 the three phases alone -- what each path costs
                    // on an otherwise idle chip, tools/path_isolation.sh)
                    if (upd_pending) CU2REC_HIP(hipStreamWaitEvent(stream, ev_upd, 0));  // join: rows the next phase 1 reads
                    if (!(dbg & 64)) {
                        bs_launch_gram(a, it, stream, ev_gram);
                        CU2REC_HIP(hipStreamWaitEvent(upd, ev_gram, 0));
                        bs_launch_solve(a, it, stream);
                    } else {
                        CU2REC_HIP(hipEventRecord(ev_gram, stream));
                        CU2REC_HIP(hipStreamWaitEvent(upd, ev_gram, 0));
                    }
                    if (!(dbg & 128))
                        launch_chains(a, it.keys, it.vals, n_active, item_of_rank.ptr, item_mask, std::max(n_duo_bs, n_hot_bs), upd, n_hot_bs, ranges, ev_upd,
                                      walk_bound_bs);
                    else
                        CU2REC_HIP(hipEventRecord(ev_upd, upd));
                    if (!(dbg & 64)) bs_launch_update(a, it, stream);
                    upd_pending = true;
                }
            }
            schedule_next();  // (a batch of one or two iterations)
            CU2REC_HIP(hipGetLastError());
            // the batch's slot is free once both streams are through with it
            CU2REC_HIP(hipStreamWaitEvent(stream, ev_upd, 0));
            CU2REC_HIP(hipEventRecord(ev_consumed[slot], stream));
            done += nb;
            continue;
        }
        for (int b = 0; b < nb; ++b) {
            if (b == std::min(2, nb - 1) && b > 0) schedule_next();
            a.iter0 = iter0 + done + b;
            launch_chains(a, sk + static_cast<size_t>(b) * seg, sv + static_cast<size_t>(b) * seg, n_active, item_of_rank.ptr, item_mask, n_hot, stream, 0,
                          n_hot <= n_range_ranks ? chain_ranges[slot].ptr + static_cast<size_t>(off + b) * (n_range_ranks + 1) : nullptr, nullptr,
                          walk_bound_plain);
        }
        schedule_next();
        CU2REC_HIP(hipGetLastError());
        CU2REC_HIP(hipEventRecord(ev_consumed[slot], stream));
        done += nb;
    }
    if (blocksolve) bs_report_status(stream);  // (every batch ended with `stream` behind the side stream)
    CU2REC_HIP(hipEventRecord(ev_last, stream));
    last_stream = stream;
    have_last = true;
}

}  // namespace cu2rec
