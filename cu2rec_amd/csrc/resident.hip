// Resident Hogwild SGD for gfx950: the reference's per-iteration launches (training.cu:107-113, sgd_update
// sgd.cu:22-75) folded into ONE persistent launch that keeps every user's factor row on the CU -- in the register
// file and, beyond that, in LDS -- for all iterations of the call.
//
// Why: in the streaming kernel (kernels.hip, sgd_hogwild_kernel) half of the 16 f + 32 algorithmic bytes per
// update are the user's own row going to HBM and back between two launches, although nobody else ever
// touches it.  An MI355X has 256 CUs x 512 KB of vector registers = 128 MB -- more than the whole P matrix
// of the ML-20M shape (55 MB).  So: one 512-thread workgroup per CU, each 16-lane group owns R users
// (R x groups >= users), loads their rows ONCE, runs all iterations of the launch on them and stores them once
// at the end.  Per iteration only the item side moves: sample -> item id / rating (one gather) -> item row read,
// rank-1 update, item row write.
//
// Semantics are those of one launch per iteration: between two iterations sits a grid-wide barrier with an
// agent-scope release (every item row written in iteration i has left the XCD's L2) and acquire (the CU's L1
// is invalidated), so iteration i + 1 reads what iteration i wrote, on every XCD -- exactly what the kernel
// boundary gives the streaming form.  Inside an iteration it is Hogwild as before (sgd.cu:18-21): all users in
// flight, two users that sampled the same item both read the row as it was, the later store wins.
//
// Layout inside a group (lanes 0..15 = one DPP row, as everywhere else):
//   p[r]          Row<J> of the group's r-th user in registers for the whole launch (r < R - RL; the last RL rows
//                 of a group live in LDS and pass through registers only during their step)
//   my_*[m]       per-user scalars (row start, row length, user bias, this iteration's sample) are
//                 LANE-distributed: lane l of set m holds them for user r = 16 m + l, so ONE Philox pass and
//                 ONE gather per set serve 16 users; a 16-wide shuffle hands them to the group when user r
//                 is processed.
// The update loop is straight-line code, software pipelined by hand: the item rows of D users are in flight while
// one is computed (why it has to be branch free: see sgd_resident_kernel).  The arithmetic of an update is the
// shared device code of sgd_device.hpp, bit for bit.
//
// The grid barrier follows MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup visibility" and
// its XCD-hierarchical form ("barrier-xcd"): every wave drains its stores, workgroup barrier, lane 0 arrives on its
// XCD's counter; the last arriver of an XCD does the ONE agent-scope release for that XCD's L2, reports to the top
// counter, waits for all XCDs and opens the XCD's generation word, which the other workgroups poll (relaxed sc1
// loads with s_sleep).  (A flat barrier with one release per workgroup cost 28 us per iteration here: 32 L2
// write-backs per XCD queue behind each other.)  The barrier is split in two halves; the next iteration's sample
// gathers and the acquire's L1 invalidate sit between them.  Every spin is bounded (wall clock): a grid that is
// not co-resident ends with the status word set instead of hanging, and the host reports it
// (resident_check_fault).  Residency comes from the grid size alone: at most one workgroup per CU.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "hip_check.hpp"
#include "kernels.hpp"
#include "resident.hpp"
#include "grid_barrier.hpp"
#include "sampler.hpp"
#include "sgd_device.hpp"

#ifndef CU2REC_RES_ABLATE
#define CU2REC_RES_ABLATE 0  // timing-only builds (tools/build_variant.sh), bits: 1 no grid barrier, 2 no updates, 4 no item-row
                             // stores, 64 / 128 barrier without its release / acquire fence; never set in the shipped library
#endif

#ifndef CU2REC_RES_D2
#define CU2REC_RES_D2 3  // item rows in flight per group at two float4 per lane (f = 68..128)
#endif

namespace cu2rec {

namespace {

using namespace dev;

constexpr int kResBlock = 512;                    // 8 wavefronts = 2 per SIMD: up to 256 VGPRs each
constexpr int kResGroups = kResBlock / kGroup;    // 32 user groups per workgroup

using namespace gridbar;  // the barrier: grid_barrier.hpp

struct ResidentArgs {
    unsigned *words;       // the barrier block above (device memory)
    unsigned *status;      // set to 1 by a workgroup that gave up waiting (the host reads and clears it)
    float *sink;           // kSinkFloats floats nobody reads: where the updates of users without ratings are written
    int streamed;          // partial residency: users per group BEHIND the resident ones whose rows stay in memory (0: none)
};
constexpr int kSinkBias = 512;                 // sink[0 .. 511]: an item row, sink[512]: an item bias
constexpr int kSinkFloats = kSinkBias + 16;

// A lane's float4 slots of a row, "wrapped": where lane + 16 j runs past the row (f = 100: the second slot of lanes
// 9..15) the lane holds a DUPLICATE of a real slot instead of padding.  Duplicates are loaded from, updated like
// and stored to the slot they copy -- same inputs, same operations, same bits, so the extra store is the same bytes
// to the same address in the same instruction -- and are masked out of the dot product, where they contribute the
// +0.0 the zero padding of the streaming layout contributes.  That keeps every load and store of the update loop
// unconditional: no exec-masked memory operation, hence no branch, hence exact s_waitcnt counts (see below).
template <int J>
struct Slots {
    int off[J];       // float4 index inside a row
    bool last_valid;  // does slot lane + 16 (J - 1) exist?  (all earlier ones do)
};

template <int J>
__device__ __forceinline__ Row<J> load_wrapped(const float *row, const Slots<J> &sl) {
    const float4 *p4 = reinterpret_cast<const float4 *>(row);
    Row<J> r;
#pragma unroll
    for (int j = 0; j < J; ++j) r.v[j] = p4[sl.off[j]];
    return r;
}

template <int J>
__device__ __forceinline__ void store_wrapped(float *row, const Slots<J> &sl, const Row<J> &r) {
    float4 *p4 = reinterpret_cast<float4 *>(row);
#pragma unroll
    for (int j = 0; j < J; ++j) p4[sl.off[j]] = r.v[j];
}

// predict<J> (sgd_device.hpp) on wrapped rows: same operations in the same order, the duplicate slot counts as +0.0
template <int J>
__device__ __forceinline__ float predict_wrapped(const Row<J> &p, const Row<J> &q, float ub, float ib, float gb,
                                                 bool last_valid) {
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        float sd = slot_dot(q.v[j], p.v[j]);
        if (j == J - 1) sd = last_valid ? sd : 0.f;
        acc = j == 0 ? sd : acc + sd;
    }
    const float dot = row_sum16(acc);
    return ((gb + ub) + ib) + dot;
}

// R (users per group) is a compile-time parameter and the whole update loop is straight-line code, because the
// compiler's s_waitcnt insertion counts outstanding memory operations conservatively across every branch: with a
// run-time `r < R`, an exec-masked load / store or an `if (update_items)` anywhere in the loop, each step ended up
// waiting for everything in flight (vmcnt(0)) -- no software pipelining at all (measured: D = 1..5 all took 19 us
// per iteration); early exits out of an unrolled chain made the register allocator spill hundreds of registers
// instead.  So: a few R per row width are compiled, the host rounds the count it needs up to the next one and
// shrinks the grid to match; frozen items (update_items == 0) take the streaming kernel, which needs no barrier.
//
// RL of a group's R rows live in the CU's LDS instead of registers (the last RL): 32 groups x RL rows x J x 256 B per
// workgroup, each lane reading and writing its own 16 bytes (conflict free).  Such a row is read at the start of its
// step and written back at the end; everything else is the same.  With the 160 KB of LDS a CU holds 9 more rows per
// group at f <= 128 (16 + 9 = 25: 204,000 users on 256 CUs).
//
// J: float4 slots per lane (row width), R: users per group, D: item rows in flight per group, RL: rows kept in LDS.
// PART (partial residency, round 4): a set that does not fit the chip.  Every group owns ra.streamed MORE users behind its R resident
// ones; their rows stay in memory and pass through the same pipeline -- row in with the item row, update, row out -- in chunks of 16
// users (one lane-distributed set of scalars per chunk, drawn and gathered while the chunk before is worked on; the pipeline is
// drained at a chunk's end, so that nothing is in flight across the loop's back edge: the compiler's s_waitcnt counts stay exact
// inside the straight-line chunk).  Netflix shape, 480,189 users at f = 128: 25 of a group's 59-60 rows resident.
template <int J, int R, int D, int RL, bool PART = false>
__global__ __launch_bounds__(kResBlock) void sgd_resident_kernel(SgdArgs a, ResidentArgs ra) {
    constexpr int RREG = R - RL;  // rows in registers
    static_assert(D >= 1 && D <= R && RL >= 0 && RREG >= 1, "shape");
    static_assert(sizeof(float4) * kResGroups * RL * J * kGroup + 64 <= 160 * 1024, "LDS rows exceed the CU's 160 KB");
    constexpr int M = (R + kGroup - 1) / kGroup;  // lane-distributed scalar sets
    __shared__ BarrierShared s_barrier;
    __shared__ float4 s_rows[RL > 0 ? kResGroups * RL * J * kGroup : 1];
    float4 *my_lds = s_rows + (threadIdx.x / kGroup) * (RL * J * kGroup) + (threadIdx.x & (kGroup - 1));
    auto lds_load = [&](int rl) {
        Row<J> row;
#pragma unroll
        for (int j = 0; j < J; ++j) row.v[j] = my_lds[(rl * J + j) * kGroup];
        return row;
    };
    auto lds_store = [&](int rl, const Row<J> &row) {
#pragma unroll
        for (int j = 0; j < J; ++j) my_lds[(rl * J + j) * kGroup] = row.v[j];
    };
    const int lane = threadIdx.x & (kGroup - 1);
    const int wg = blockIdx.x, n_wg = gridDim.x;
    bool alive = true;
    const int group = wg * kResGroups + (threadIdx.x / kGroup);
    const int n_groups = n_wg * kResGroups;

    Slots<J> sl;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int slot = lane + kGroup * j;
        // a duplicate copies a slot of the SAME 256-byte chunk of the row, so that instruction j stays on one region
        sl.off[j] = slot < a.nslots ? slot : kGroup * j + lane % (a.nslots - kGroup * j);
    }
    sl.last_valid = lane + kGroup * (J - 1) < a.nslots;

    // ---- prologue: the group's users move in -------------------------------------------------------
    int my_x[M], my_low[M], my_n[M];
    float my_ub[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const int r = m * kGroup + lane;
        const int x = group + r * n_groups;  // user r of the group; consecutive groups own consecutive users
        const bool valid = r < R && x < a.n_rows;
        my_x[m] = x;
        my_low[m] = valid ? a.indptr[x] : 0;
        my_n[m] = valid ? a.indptr[x + 1] - my_low[m] : 0;  // 0: no ratings, skipped (sgd.cu:34)
        my_ub[m] = valid ? a.user_bias[x] : 0.f;
    }
    Row<J> p[RREG];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int x = group + r * n_groups;
        Row<J> row;
        if (x < a.n_rows) {
            row = load_wrapped<J>(a.P + static_cast<size_t>(x) * a.ldp, sl);
        } else {
#pragma unroll
            for (int j = 0; j < J; ++j) row.v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (r < RREG) p[r] = row;
        else lds_store(r - RREG, row);
    }

    // sample of iteration `it` for the 16 users of every set: one draw + one gather per lane (sgd.cu:36-44)
    auto draw = [&](uint64_t it, int (&item)[M], float (&rating)[M]) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
            item[m] = 0;
            rating[m] = 0.f;
            if (my_n[m] > 0) {
                const int y_i = sampler_index(a.seed, static_cast<uint64_t>(a.user_offset + my_x[m]), it, my_low[m],
                                              my_low[m] + my_n[m]);
                if (a.pairs != nullptr) {  // one 8-byte gather (SgdArgs::pairs)
                    const uint2 ir = a.pairs[y_i];  // (a non-temporal hint here measured +1 us per iteration)
                    item[m] = static_cast<int>(ir.x);
                    rating[m] = __uint_as_float(ir.y);
                } else {
                    item[m] = a.indices[y_i];
                    rating[m] = a.data[y_i];
                }
            }
        }
    };

    int item[M];
    float rating[M];
    draw(a.iter0, item, rating);
    const bool synced = !(CU2REC_RES_ABLATE & 1) && a.iters > 1;
    if (synced) {
        if (threadIdx.x == 0) barrier_census(ra, &s_barrier);
        __syncthreads();
        alive = s_barrier.ok != 0;  // false: the grid is not co-resident; nothing is run, the host reports it
    }

    // ---- the iterations ----------------------------------------------------------------------------
    // One step = "compute the user whose item row was requested D steps ago, then request the next user's".  The
    // step bodies contain no branch: users without ratings and the unused slots of the last, partly filled row of
    // the user -> group assignment read item row 0 and write to a sink row.
    Row<J> q[D];
    float q_ib[D], q_rating[D];
    int q_item[D];
    bool q_active[D];

    const int64_t sink_row_off = reinterpret_cast<char *>(ra.sink) - reinterpret_cast<char *>(a.Q);
    const int64_t sink_bias_off = reinterpret_cast<char *>(ra.sink + kSinkBias) - reinterpret_cast<char *>(a.item_bias);

    // (item_l, rating_l, n_l: one lane-distributed set of 16 users; src: the user's lane in it)
    auto issue_from = [&](int item_l, float rating_l, int n_l, int src, int s) {
        q_item[s] = __shfl(item_l, src, kGroup);  // 0 for a user without ratings
        q_rating[s] = __shfl(rating_l, src, kGroup);
        q_active[s] = __shfl(n_l, src, kGroup) > 0;
        q[s] = load_wrapped<J>(a.Q + static_cast<size_t>(q_item[s]) * a.ldq, sl);
        q_ib[s] = a.item_bias[q_item[s]];
    };
    auto issue = [&](int r, int s) {  // r, s: constants after unrolling
        issue_from(item[r / kGroup], rating[r / kGroup], my_n[r / kGroup], r % kGroup, s);
    };
    auto consume_with = [&](Row<J> &pc, float &ub_l, int src, int s) {  // ub_l: the set's lane-distributed user biases
        const float ub = __shfl(ub_l, src, kGroup);
        const float ib = q_ib[s];
        const float err = q_rating[s] - predict_wrapped<J>(pc, q[s], ub, ib, a.global_bias, sl.last_valid);  // sgd.cu:45
        rank1_update<J>(pc, q[s], err, a.h);                                                                // sgd.cu:53-64
        // the new user row is not needed before the next iteration, and left alone the optimizer sinks its computation
        // to the end of the loop, keeping err * q (4 J registers) alive per user until then: pin it to this step
#pragma unroll
        for (int j = 0; j < J; ++j) asm volatile("" : "+v"(pc.v[j].x), "+v"(pc.v[j].y), "+v"(pc.v[j].z), "+v"(pc.v[j].w));
        if (!(CU2REC_RES_ABLATE & 4)) {                                                              // sgd.cu:61,70
            // the sink is addressed as a byte offset from Q / item_bias so that "real row or sink" is a select
            // between two integers (a select between two pointers comes back as a branch around the stores)
            const int64_t row_off = q_active[s] ? static_cast<int64_t>(q_item[s]) * a.ldq * 4 : sink_row_off;
            store_wrapped<J>(reinterpret_cast<float *>(reinterpret_cast<char *>(a.Q) + row_off), sl, q[s]);
            // all 16 lanes store the same value to the same address: one request, and no lane == 0 branch
            const int64_t bias_off = q_active[s] ? static_cast<int64_t>(q_item[s]) * 4 : sink_bias_off;
            *reinterpret_cast<float *>(reinterpret_cast<char *>(a.item_bias) + bias_off) =
                ib + a.h.lr * (err - a.h.ib_reg * ib);  // sgd.cu:71
        }
        const float ub_new = ub + a.h.lr * (err - a.h.ub_reg * ub);  // sgd.cu:67
        ub_l = (lane == src && q_active[s]) ? ub_new : ub_l;
    };
    auto consume = [&](Row<J> &pc, int c, int s) { consume_with(pc, my_ub[c / kGroup], c % kGroup, s); };
    [[maybe_unused]] const int64_t sink_p_off = reinterpret_cast<char *>(ra.sink) - reinterpret_cast<char *>(a.P);
    // a chunk of streamed users, lane-distributed: everything branch free (clamped addresses, selects), so that it can sit between
    // the steps of the chunk before without costing them their counted waits
    struct Chunk {
        int x, low, n, item;  // x: the user, or a valid row to read for a lane without one (n == 0: never written)
        float ub, rating;
    };
    auto chunk_bounds = [&](int ch, Chunk &c) {
        const int sidx = ch * kGroup + lane;
        const int xs = group + (R + sidx) * n_groups;
        const bool valid = sidx < ra.streamed && xs < a.n_rows;
        c.x = valid ? xs : 0;
        const int lo = a.indptr[c.x], hi = a.indptr[c.x + 1];
        const float ub = a.user_bias[c.x];
        c.low = lo;
        c.n = valid ? hi - lo : 0;
        c.ub = ub;
    };
    auto chunk_draw = [&](uint64_t it, Chunk &c) {
        const int y_i = sampler_index(a.seed, static_cast<uint64_t>(a.user_offset + c.x), it, c.low, c.low + max(c.n, 1));
        const int at = c.n > 0 ? y_i : 0;
        if (a.pairs != nullptr) {  // (uniform)
            const uint2 ir = a.pairs[at];
            c.item = c.n > 0 ? static_cast<int>(ir.x) : 0;
            c.rating = __uint_as_float(ir.y);
        } else {
            const int yi = a.indices[at];
            c.rating = a.data[at];
            c.item = c.n > 0 ? yi : 0;
        }
    };

    for (int k = 0; alive && k < a.iters; ++k) {
        // make the compiler wait for this iteration's samples HERE, before any item row is in flight (an operand use
        // is what its s_waitcnt insertion sees; later it could only wait for them together with the rows)
#pragma unroll
        for (int m = 0; m < M; ++m) asm volatile("" ::"v"(item[m]), "v"(rating[m]));

        if (!(CU2REC_RES_ABLATE & 2)) {
#pragma unroll
            for (int r = 0; r < R + D; ++r) {
                if (r >= D) {
                    const int c = r - D;
                    if (c < RREG) {
                        consume(p[c < RREG ? c : 0], c, c % D);
                    } else {  // an LDS row: in, one update, out
                        Row<J> row = lds_load(c - RREG);
                        consume(row, c, c % D);
                        lds_store(c - RREG, row);
                    }
                }
                if (r < R) issue(r, r % D);
                // keep a step's shuffles and address arithmetic inside the step: hoisted to the top of the
                // straight-line code they cost ~10 live registers per user, which the resident rows need
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (PART) {
                const uint64_t it_now = a.iter0 + static_cast<uint64_t>(k);
                const int n_chunks = (ra.streamed + kGroup - 1) / kGroup;
                Chunk cur, nxt;
                chunk_bounds(0, cur);
                chunk_draw(it_now, cur);
                for (int ch = 0; ch < n_chunks; ++ch) {
                    asm volatile("" ::"v"(cur.item), "v"(cur.rating), "v"(cur.ub), "v"(cur.n), "v"(cur.x));  // (the chunk's scalars are here)
                    Row<J> ps[D];
#pragma unroll
                    for (int u = 0; u < kGroup + D; ++u) {
                        if (u >= D) {
                            const int c = u - D;
                            Row<J> row = ps[c % D];
                            consume_with(row, cur.ub, c, c % D);
                            const int64_t p_off =
                                q_active[c % D] ? static_cast<int64_t>(__shfl(cur.x, c, kGroup)) * a.ldp * 4 : sink_p_off;
                            store_wrapped<J>(reinterpret_cast<float *>(reinterpret_cast<char *>(a.P) + p_off), sl, row);
                        }
                        if (u < kGroup) {
                            issue_from(cur.item, cur.rating, cur.n, u, u % D);
                            ps[u % D] = load_wrapped<J>(a.P + static_cast<size_t>(__shfl(cur.x, u, kGroup)) * a.ldp, sl);
                        }
                        // the next chunk's scalars and draw ride along (a chunk past the last one: clamped, unused)
                        if (u == 1) chunk_bounds(ch + 1, nxt);
                        if (u == kGroup / 2) chunk_draw(it_now, nxt);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // the users' biases go back (a lane without a user, or a user without ratings: its own old value, or nothing)
                    if (cur.n > 0) a.user_bias[cur.x] = cur.ub;
                    cur = nxt;
                }
            }
        }
        if (k + 1 == a.iters) break;
        // The sampler is value independent, so the next iteration's draws and CSR gathers need nothing from other
        // workgroups: they run between the two halves of the barrier, while the XCDs write back and arrive.
        const unsigned phase = static_cast<unsigned>(k) + 1;
        if (synced) barrier_arrive(ra, phase, &s_barrier);
        draw(a.iter0 + static_cast<uint64_t>(k) + 1, item, rating);
        if (synced && !barrier_wait(ra, phase, &s_barrier)) break;
    }

    // ---- epilogue: the users move out (a user without ratings was never changed and is not written) ----
    // The base pointer is made opaque so that the row addresses are computed again here: as common subexpressions of
    // the prologue's they would stay in registers (4 J per user) for the whole launch.
    float *P_out = a.P;
    asm volatile("" : "+s"(P_out));
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int x = group + r * n_groups;
        const bool has_ratings = __shfl(my_n[r / kGroup], r % kGroup, kGroup) > 0;
        const Row<J> row = r < RREG ? p[r < RREG ? r : 0] : lds_load(r - RREG);
        if (x < a.n_rows && has_ratings) store_wrapped<J>(P_out + static_cast<size_t>(x) * a.ldp, sl, row);
    }
#pragma unroll
    for (int m = 0; m < M; ++m)
        if (my_n[m] > 0) a.user_bias[my_x[m]] = my_ub[m];
}

// ---- host side --------------------------------------------------------------------------------------

// Compiled (J, R) variants.  J = float4 slots per lane (row width), R = users per group: what fits 256 VGPRs without
// spilling, plus (the _LDS variants) what the CU's 160 KB of LDS adds.  Capacity on 256 CUs: f <= 64: 385,000 users,
// f <= 128: 204,000, f <= 192: 114,000, f <= 256: 81,000.
struct Variant {
    int j, r, rl;  // rl of the r resident rows per group live in LDS
    bool part;     // partial residency: any number of further users per group, their rows streamed (ResidentArgs::streamed)
    const void *kernel;
};
#define CU2REC_RES_VARIANT(J, R, D) {J, R, 0, false, reinterpret_cast<const void *>(&sgd_resident_kernel<J, R, D, 0>)}
#define CU2REC_RES_VARIANT_LDS(J, R, D, RL) {J, R, RL, false, reinterpret_cast<const void *>(&sgd_resident_kernel<J, R, D, RL>)}
#define CU2REC_RES_VARIANT_PART(J, R, D, RL) {J, R, RL, true, reinterpret_cast<const void *>(&sgd_resident_kernel<J, R, D, RL, true>)}
const Variant kVariants[] = {
    CU2REC_RES_VARIANT(1, 4, 4),   CU2REC_RES_VARIANT(1, 8, 4),   CU2REC_RES_VARIANT(1, 12, 4),  CU2REC_RES_VARIANT(1, 16, 4),
    CU2REC_RES_VARIANT(1, 20, 4),  CU2REC_RES_VARIANT(1, 24, 4),  CU2REC_RES_VARIANT(1, 28, 4),  CU2REC_RES_VARIANT(1, 32, 4),
    CU2REC_RES_VARIANT_LDS(1, 36, 4, 8), CU2REC_RES_VARIANT_LDS(1, 40, 4, 12), CU2REC_RES_VARIANT_LDS(1, 44, 4, 16),
    CU2REC_RES_VARIANT_LDS(1, 47, 4, 19),
    CU2REC_RES_VARIANT(2, 4, CU2REC_RES_D2),  CU2REC_RES_VARIANT(2, 6, CU2REC_RES_D2),  CU2REC_RES_VARIANT(2, 8, CU2REC_RES_D2),
    CU2REC_RES_VARIANT(2, 10, CU2REC_RES_D2), CU2REC_RES_VARIANT(2, 12, CU2REC_RES_D2), CU2REC_RES_VARIANT(2, 14, CU2REC_RES_D2),
    CU2REC_RES_VARIANT(2, 16, CU2REC_RES_D2), CU2REC_RES_VARIANT(2, 17, CU2REC_RES_D2), CU2REC_RES_VARIANT(2, 18, CU2REC_RES_D2),
    CU2REC_RES_VARIANT_LDS(2, 20, CU2REC_RES_D2, 4), CU2REC_RES_VARIANT_LDS(2, 22, CU2REC_RES_D2, 6),
    CU2REC_RES_VARIANT_LDS(2, 24, CU2REC_RES_D2, 8), CU2REC_RES_VARIANT_LDS(2, 25, CU2REC_RES_D2, 9),
    CU2REC_RES_VARIANT(3, 4, 3),   CU2REC_RES_VARIANT(3, 6, 3),   CU2REC_RES_VARIANT(3, 8, 3),
    CU2REC_RES_VARIANT_LDS(3, 10, 3, 2), CU2REC_RES_VARIANT_LDS(3, 12, 3, 4), CU2REC_RES_VARIANT_LDS(3, 14, 3, 6),
    CU2REC_RES_VARIANT(4, 4, 2),   CU2REC_RES_VARIANT(4, 6, 2),
    CU2REC_RES_VARIANT_LDS(4, 8, 2, 2), CU2REC_RES_VARIANT_LDS(4, 10, 2, 4),
    // partial residency: one form per row width (a few register rows fewer than the largest resident form: the streamed rows'
    // pipeline and a chunk's scalars live in them)
    CU2REC_RES_VARIANT_PART(1, 39, 4, 19), CU2REC_RES_VARIANT_PART(2, 20, CU2REC_RES_D2, 9), CU2REC_RES_VARIANT_PART(3, 11, 3, 6),
    CU2REC_RES_VARIANT_PART(4, 8, 2, 4),
};
#undef CU2REC_RES_VARIANT
#undef CU2REC_RES_VARIANT_LDS
#undef CU2REC_RES_VARIANT_PART
constexpr int kNumVariants = static_cast<int>(sizeof(kVariants) / sizeof(kVariants[0]));

// smallest fully resident compiled variant with this J and at least `need` users per group; else (if `partial`) the partially
// resident form of this J, whose further users stream; -1 if none
int variant_for(int j, int need, bool partial = false) {
    int best = -1;
    for (int i = 0; i < kNumVariants; ++i)
        if (kVariants[i].j == j && !kVariants[i].part && kVariants[i].r >= need && (best < 0 || kVariants[i].r < kVariants[best].r)) best = i;
    if (best >= 0 || !partial) return best;
    for (int i = 0; i < kNumVariants; ++i)
        if (kVariants[i].j == j && kVariants[i].part) best = i;
    return best;
}

// Partial residency pays while a fair share of the rows is resident: at most kMaxStreamedPerResident streamed users per resident
// one (beyond that -- and with the policy off -- a set that does not fit streams, one launch per iteration).
constexpr int kMaxStreamedPerResident = 3;
bool resident_partial_allowed() { return true; }

struct DeviceState {
    bool ready = false;
    bool usable[kNumVariants] = {};   // the occupancy query admits one workgroup per CU
    int cus = 0;
    unsigned *words = nullptr;        // device: the barrier block (kBarrierWords), then the status word
    float *sink = nullptr;            // device: kSinkFloats floats
    unsigned *host_status = nullptr;  // pinned host copy of the status word, refreshed after every launch
    hipEvent_t done = nullptr;        // end of the latest resident launch
    hipStream_t last_stream = nullptr;
    bool have_last = false;
    bool cooperative = false;         // hipDeviceAttributeCooperativeLaunch
    int refused = 0;                  // launches the runtime refused (grid not co-resident): those calls streamed
};

std::mutex g_mutex;
std::vector<DeviceState> g_states;
std::atomic<int> g_policy{-1};

DeviceState &state_for_current_device() {
    int dev = 0;
    CU2REC_HIP(hipGetDevice(&dev));
    if (static_cast<int>(g_states.size()) <= dev) g_states.resize(dev + 1);
    DeviceState &s = g_states[dev];
    if (!s.ready) {
        hipDeviceProp_t prop;
        CU2REC_HIP(hipGetDeviceProperties(&prop, dev));
        s.cus = prop.multiProcessorCount;
        int coop = 0;
        if (hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev) != hipSuccess) (void)hipGetLastError();
        s.cooperative = coop != 0;
        CU2REC_HIP(hipMalloc(reinterpret_cast<void **>(&s.words), (kBarrierWords + kLine) * sizeof(unsigned)));
        CU2REC_HIP(hipMemset(s.words, 0, (kBarrierWords + kLine) * sizeof(unsigned)));
        CU2REC_HIP(hipMalloc(reinterpret_cast<void **>(&s.sink), kSinkFloats * sizeof(float)));
        CU2REC_HIP(hipMemset(s.sink, 0, kSinkFloats * sizeof(float)));
        CU2REC_HIP(hipHostMalloc(reinterpret_cast<void **>(&s.host_status), sizeof(unsigned), hipHostMallocDefault));
        *s.host_status = 0;
        CU2REC_HIP(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
        for (int i = 0; i < kNumVariants; ++i) {
            int per_cu = 0;
            const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kVariants[i].kernel, kResBlock, 0);
            if (e != hipSuccess) (void)hipGetLastError();
            s.usable[i] = e == hipSuccess && per_cu >= 1;
        }
        s.ready = true;
    }
    return s;
}

}  // namespace

int resident_policy(int set_to) {
    if (g_policy.load() < 0) {
        int from_env = kResidentAuto;
        if (const char *e = std::getenv("CU2REC_RESIDENT")) {
            const int v = std::atoi(e);
            if (v >= kResidentOff && v <= kResidentForce) from_env = v;
        }
        int expected = -1;
        g_policy.compare_exchange_strong(expected, from_env);
    }
    const int prev = g_policy.load();
    if (set_to >= kResidentOff && set_to <= kResidentForce) g_policy.store(set_to);
    return prev;
}

void resident_check_fault() {
    std::lock_guard<std::mutex> lock(g_mutex);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    if (dev < static_cast<int>(g_states.size()) && g_states[dev].ready &&
        *static_cast<volatile unsigned *>(g_states[dev].host_status) != 0) {
        // reported once: the words are cleared so that the process can go on (e.g. with cu2rec_hogwild_resident(0))
        DeviceState &s = g_states[dev];
        (void)hipDeviceSynchronize();
        (void)hipMemset(s.words + kBarrierWords, 0, sizeof(unsigned));
        *s.host_status = 0;
        fail(CU2REC_EHIP,
             "cu2rec_amd: a persistent SGD launch gave up at its grid barrier (the grid was not co-resident: is another "
             "process or stream using this GPU?); the model state is undefined.  cu2rec_hogwild_resident(0) / "
             "CU2REC_RESIDENT=0 selects the one-launch-per-iteration Hogwild kernel");
    }
}

// Pure arithmetic of the plan (no device needed): which compiled variant holds n_rows users of this width on n_cus
// CUs, and the grid that goes with it.  false: the rows do not fit registers + LDS.
namespace {

// variant index (or -1) and, through `need` / `blocks`, rows per group needed and the grid of the chosen variant
// ... `streamed`: users per group whose rows stream (0: fully resident)
int geometry(int n_rows, int nslots, int n_cus, int &need, int &blocks, int &streamed) {
    streamed = 0;
    if (n_rows < 1 || nslots < 1 || n_cus < 1) return -1;
    const long long max_groups = static_cast<long long>(n_cus) * kResGroups;
    need = static_cast<int>(std::min<long long>((n_rows + max_groups - 1) / max_groups, 1 << 20));
    const int v = variant_for(slots_per_lane(nslots), need, resident_partial_allowed());
    if (v < 0) return -1;  // the rows do not fit registers + LDS, and no partial form is compiled for this width
    if (kVariants[v].part) {  // the whole chip, every group its share: the first r users of a group resident, the others streamed
        streamed = std::max(need - kVariants[v].r, 0);
        if (streamed > kMaxStreamedPerResident * kVariants[v].r) return -1;  // too few rows would be resident: stream them all
        blocks = n_cus;
        return v;
    }
    const int r = kVariants[v].r;  // >= need: the grid shrinks instead of running empty steps
    const long long groups = (n_rows + static_cast<long long>(r) - 1) / r;
    blocks = static_cast<int>((groups + kResGroups - 1) / kResGroups);
    return v;
}

}  // namespace

bool resident_geometry(int n_rows, int n_factors, int n_cus, int *blocks, int *users_per_group, int *lds_rows) {
    int need = 0, b = 0, streamed = 0;
    const int v = n_factors >= 1 ? geometry(n_rows, (n_factors + 3) / 4, n_cus, need, b, streamed) : -1;
    if (v < 0) return false;
    if (blocks) *blocks = b;
    if (users_per_group) *users_per_group = kVariants[v].r + streamed;
    if (lds_rows) *lds_rows = kVariants[v].rl;
    return true;
}

int resident_streamed_rows(int n_rows, int n_factors, int n_cus) {
    int need = 0, b = 0, streamed = 0;
    const int v = n_factors >= 1 ? geometry(n_rows, (n_factors + 3) / 4, n_cus, need, b, streamed) : -1;
    return v < 0 ? -1 : streamed;
}

namespace {

// The launch geometry a call would get, or false if it would stream.  Caller holds g_mutex.
bool plan_locked(int n_rows, int nslots, int n_iters, int update_items, DeviceState *&state, int &variant, int &blocks,
                 int &users_per_group, int &streamed) {
    streamed = 0;
    const int policy = resident_policy(-1);
    if (policy == kResidentOff || n_iters < 1 || n_rows < 1 || !update_items) return false;
    DeviceState &s = state_for_current_device();
    state = &s;
    int need = 0;
    variant = geometry(n_rows, nslots, s.cus, need, blocks, streamed);
    if (variant < 0 || !s.usable[variant]) return false;  // the rows do not fit: stream them
    // Auto: the barrier costs a few microseconds where a kernel boundary costs one or two, so residency pays once
    // an iteration moves enough rows per group and the launch is long enough to amortise loading them.
    if (policy == kResidentAuto && (need < 4 || n_iters < 4)) return false;
    users_per_group = kVariants[variant].r + streamed;
#ifdef CU2REC_RES_TEST_OVERSUBSCRIBE  // fault-path check only (tools/build_variant.sh): a grid that cannot be co-resident
    blocks *= CU2REC_RES_TEST_OVERSUBSCRIBE;
#endif
    return true;
}

}  // namespace

int resident_refusals() {
    std::lock_guard<std::mutex> lock(g_mutex);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return dev < static_cast<int>(g_states.size()) ? g_states[dev].refused : 0;
}

bool resident_plan(int n_rows, int n_factors, int n_iters, int *blocks_out, int *users_per_group_out) {
    std::lock_guard<std::mutex> lock(g_mutex);
    DeviceState *s = nullptr;
    int variant = -1, blocks = 0, users_per_group = 0, streamed = 0;
    const bool yes = plan_locked(n_rows, (n_factors + 3) / 4, n_iters, 1, s, variant, blocks, users_per_group, streamed);
    if (blocks_out) *blocks_out = yes ? blocks : 0;
    if (users_per_group_out) *users_per_group_out = yes ? users_per_group : 0;
    return yes;
}

bool resident_launch(SgdArgs a, uint64_t iter0, int n_iters, hipStream_t stream) {
    std::lock_guard<std::mutex> lock(g_mutex);
    DeviceState *sp = nullptr;
    int variant = -1, blocks = 0, users_per_group = 0, streamed = 0;
    if (!plan_locked(a.n_rows, a.nslots, n_iters, a.update_items, sp, variant, blocks, users_per_group, streamed)) return false;
    DeviceState &s = *sp;

    if (s.have_last && s.last_stream != stream) CU2REC_HIP(hipStreamWaitEvent(stream, s.done, 0));  // never two at once
    a.iter0 = iter0;
    a.iters = n_iters;
    ResidentArgs ra{s.words, s.words + kBarrierWords, s.sink, streamed};
    void *args[] = {&a, &ra};
    CU2REC_HIP(hipMemsetAsync(s.words, 0, kBarrierWords * sizeof(unsigned), stream));  // counters start from zero
    // A cooperative launch: the runtime checks the grid against what can be co-resident and REFUSES a larger one up
    // front (hipErrorCooperativeLaunchTooLarge) instead of letting it wait at its first barrier for workgroups that
    // were never dispatched.  A refusal is not an error here: the call falls back to one launch per iteration, and the
    // variant is not tried again on this device.  (What a launch-time check cannot see -- another process taking CUs
    // while the grid runs -- is still caught by the barrier's timeout.)
    const hipError_t launched = s.cooperative
                                    ? hipLaunchCooperativeKernel(kVariants[variant].kernel, dim3(blocks), dim3(kResBlock), args, 0, stream)
                                    : hipLaunchKernel(kVariants[variant].kernel, dim3(blocks), dim3(kResBlock), args, 0, stream);
    if (launched == hipErrorCooperativeLaunchTooLarge || launched == hipErrorLaunchOutOfResources) {
        (void)hipGetLastError();  // a refusal: the grid cannot be co-resident on this device -- stream instead, and do not ask again
        s.usable[variant] = false;
        ++s.refused;
        return false;
    }
    CU2REC_HIP(launched);  // anything else (lost device, a sticky error of an earlier kernel, bad configuration) is an error
    CU2REC_HIP(hipMemcpyAsync(s.host_status, s.words + kBarrierWords, sizeof(unsigned), hipMemcpyDeviceToHost, stream));
    CU2REC_HIP(hipEventRecord(s.done, stream));
    s.last_stream = stream;
    s.have_last = true;
    return true;
}

}  // namespace cu2rec
