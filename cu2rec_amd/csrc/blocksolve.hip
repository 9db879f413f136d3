// Block-solve SGD (CU2REC_SGD_BLOCKSOLVE): the sequential semantics of mf_sequential.cu:102-143 at Hogwild-class
// speed.  Same schedule as the ordered mode (ordered.hip: counter-based sample stream -> per-iteration item chains,
// users ascending inside a chain), but a long chain is no longer walked one dependent update at a time.
//
// A chain is a sequence of affine rank-1 maps on the item's state (q, b).  For a block of B = 32 consecutive links with
// user rows p_k, a = 1 - lr*Q_reg, c = 1 - lr*item_bias_reg (tests/test_blocksolve_algebra.py pins these formulas):
//     (I + lr L) e = rhs,   L_kj  = c^(k-1-j) + a^(k-1-j) (p_k . p_j)  for j < k           unit lower triangular
//                           rhs_k = (r_k - gb - ub_k) - c^k b0 - a^k (p_k . q0)
//     q_end = a^B q0 + lr sum_j a^(B-1-j) e_j p_j,   b_end likewise
//     p_k'  = p_k + lr (e_k q^(k) - P_reg p_k),   q^(k) = a^k q0 + lr sum_{j<k} a^(k-1-j) e_j p_j
// Inside one iteration every user occurs once, so the user rows -- and with them L -- do not depend on any chain's
// progress.  Per iteration:
//   phase 1  bs_gram_kernel    every block of every hot chain in parallel: P_blk P_blk^T on the matrix cores
//                              (v_mfma_f32_32x32x2_f32: exact f32), scaled into lr*L
//   phase 2  bs_solve_kernel   one workgroup per hot chain walks its blocks: one mat-vec with the block's rows, a
//                              32-step scalar forward substitution, one transposed mat-vec -- the only sequential part;
//                              four more wavefronts of the workgroup stream the blocks' rows and L tiles into LDS, six
//                              blocks ahead (one CU's load path is what bounds a long chain)
//   phase 3  bs_update_kernel  every block in parallel: the item row as each link saw it = one 32x32 by 32xf product
//                              (matrix cores again), then the user rows and user biases
//   beside   bs_walk_kernel    the short chains, one update at a time as in the ordered mode (other items, other users:
//                              independent of the three phases, launched on a second stream)
// Results equal the sequential ones up to float rounding (not bit for bit: the sums are associated differently).
#include <hip/hip_runtime.h>

#include "blocksolve.hpp"
#include "hip_check.hpp"
#include "sgd_device.hpp"

namespace cu2rec {

namespace {

using namespace dev;

constexpr int kB = kBsLinks;
constexpr int kTabAdel = 0, kTabCdel = kBsTableStride, kTabApow = 2 * kBsTableStride, kTabCpow = 3 * kBsTableStride;
constexpr int kWalkWindow = 2;  // sorted positions per 16-lane group (short chains)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// v[l & 31] + v[(l & 31) + 32] in every lane
__device__ __forceinline__ float half_sum(float v) {
    const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}

__device__ __forceinline__ float lane_value(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

__device__ __forceinline__ int lane_value(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

// row of the 32x32 accumulator tile held in register `reg` of a lane of half h (column = lane & 31)
__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// development aid: one record per wavefront, written by its first lane when it ends.  Builds with -DCU2REC_BS_TRACE=1
// (tools/build_variant.sh) additionally drain the memory queue at four points of a kernel and record when (marks).
#ifndef CU2REC_BS_TRACE
#define CU2REC_BS_TRACE 0
#endif
constexpr int kStampWords = 8;
struct WaveStamp {
    unsigned long long t0;
#if CU2REC_BS_TRACE
    unsigned long long marks[4] = {0, 0, 0, 0};
#endif
    __device__ __forceinline__ explicit WaveStamp(const BsIteration &it) : t0(it.stamps ? wall_clock64() : 0) {}
    __device__ __forceinline__ void mark(const BsIteration &it, int i) {
#if CU2REC_BS_TRACE
        if (it.stamps) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            marks[i] = wall_clock64();
        }
#endif
    }
    __device__ __forceinline__ void done(const BsIteration &it, int kernel, int id) const {
        if (!it.stamps || (threadIdx.x & 63) != 0) return;
        const int seg = it.stamps_cap / 8;  // one segment of the buffer per kernel: no atomics, no contention
        if (id >= seg) return;
        unsigned long long *r = it.stamps + 1 + kStampWords * (static_cast<size_t>(kernel) * seg + id);
        r[0] = static_cast<unsigned long long>(kernel);
        r[1] = static_cast<unsigned long long>(id);
        r[2] = t0;
        r[3] = wall_clock64();
#if CU2REC_BS_TRACE
        for (int i = 0; i < 4; ++i) r[4 + i] = marks[i];
#endif
    }
};

// ---- decay tables: x * a^k is applied as x - adel[k] * x so that the rounding of a^k (one ulp of 1.0, the same sign
// every block) cannot bias a hot item's decay rate; the deltas are computed in double from the float hyper-parameters
__global__ void bs_tables_kernel(SgdHyper h, float *__restrict__ t) {
    const int k = threadIdx.x;
    if (k > kB) return;
    const double a = 1.0 - static_cast<double>(h.lr) * static_cast<double>(h.q_reg);
    const double c = 1.0 - static_cast<double>(h.lr) * static_cast<double>(h.ib_reg);
    double ak = 1.0, ck = 1.0;
    for (int i = 0; i < k; ++i) {
        ak *= a;
        ck *= c;
    }
    t[kTabAdel + k] = static_cast<float>(1.0 - ak);
    t[kTabCdel + k] = static_cast<float>(1.0 - ck);
    t[kTabApow + k] = static_cast<float>(ak);
    t[kTabCpow + k] = static_cast<float>(ck);
}

__device__ __forceinline__ int lower_bound_key(const uint32_t *__restrict__ keys, int n, uint32_t target) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (keys[mid] < target) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// ---- plan: one workgroup per iteration of the batch --------------------------------------------------------------
__global__ __launch_bounds__(256) void bs_plan_kernel(const uint32_t *__restrict__ keys, int n_active, int n_hot,
                                                      int item_bits, int max_blocks, const int *__restrict__ item_of_rank,
                                                      int *__restrict__ chain_begin, BsChainDesc *__restrict__ chains,
                                                      BsBlockDesc *__restrict__ blocks, int *__restrict__ walk_begin) {
    __shared__ int s_part[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const uint32_t *kb = keys + static_cast<size_t>(b) * n_active;
    const uint32_t key_base = static_cast<uint32_t>(b) << item_bits;
    int *cb = chain_begin + static_cast<size_t>(b) * (n_hot + 1);
    BsChainDesc *cd = chains + static_cast<size_t>(b) * max(n_hot, 1);
    BsBlockDesc *bd = blocks + static_cast<size_t>(b) * max_blocks;
    for (int r = tid; r <= n_hot; r += 256) cb[r] = lower_bound_key(kb, n_active, key_base + static_cast<uint32_t>(r));
    __syncthreads();
    if (tid == 0) walk_begin[b] = cb[n_hot];
    // blocks per chain -> exclusive prefix; thread t owns the chains [t * per, (t + 1) * per)
    const int per = (n_hot + 255) / 256;
    const int r0 = min(tid * per, n_hot), r1 = min(r0 + per, n_hot);
    int mine = 0;
    for (int r = r0; r < r1; ++r) mine += (cb[r + 1] - cb[r] + kB - 1) / kB;
    s_part[tid] = mine;
    __syncthreads();
    __shared__ int s_total;
    if (tid == 0) {
        int run = 0;
        for (int t = 0; t < 256; ++t) {
            const int v = s_part[t];
            s_part[t] = run;
            run += v;
        }
        s_total = run;
    }
    __syncthreads();
    int off = s_part[tid];
    for (int r = r0; r < r1; ++r) {
        const int begin = cb[r], len = cb[r + 1] - begin;
        cd[r] = BsChainDesc{begin, len, off, item_of_rank[r]};
        const int nb = (len + kB - 1) / kB;
        for (int m = 0; m < nb; ++m) bd[off + m] = BsBlockDesc{begin + kB * m, min(kB, len - kB * m), r, m};
        off += nb;
    }
    for (int g = s_total + tid; g < max_blocks; g += 256) bd[g] = BsBlockDesc{0, 0, 0, 0};
}

// ---- phase 1: lr * L of every block ---------------------------------------------------------------------------------
// One wavefront per block.  The 32 user rows are gathered with coalesced 128-byte pieces into the wavefront's own LDS
// tile (odd row stride: a column read is conflict free), then lane (k = l & 31, h = l >> 5) takes half h of row k:
// v_mfma_f32_32x32x2_f32 wants A[i = l & 31][kk = l >> 5] and B[kk = l >> 5][j = l & 31], so with B = A^T the SAME
// register is both operands and the contraction index pairs column c of half 0 with column c of half 1.
__global__ __launch_bounds__(256) void bs_gram_kernel(SgdArgs a, BsIteration it) {
    extern __shared__ float4 bs_smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + wave;
    if (g >= it.max_blocks) return;
    WaveStamp stamp(it);
    const BsBlockDesc bd = it.blocks[g];
    if (bd.n_valid == 0) return;
    stamp.mark(it, 0);
    const int nslots = a.nslots, RS = nslots | 1;
    float4 *tile = bs_smem + static_cast<size_t>(wave) * kB * RS;
    const int k = lane & 31, h = lane >> 5;
    const bool valid = k < bd.n_valid;
    const uint64_t val = valid ? it.vals[bd.pos0 + k] : 0;
    const int x = static_cast<int>(val >> 32);
    if (h == 0 && valid)  // what the link's error starts from: r - gb - ub (mf_sequential.cu:119-126 without b and p.q)
        it.base[bd.pos0 + k] = (__uint_as_float(static_cast<uint32_t>(val)) - a.global_bias) - a.user_bias[x];
    stamp.mark(it, 1);
    // gather: 8 lanes x 16 bytes per row piece, 8 rows per pass
    const int rsub = lane >> 3, cs = lane & 7;
    const int nch = (nslots + 7) >> 3;
    for (int c0 = 0; c0 < nch; c0 += 4) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = 8 * p + rsub;
            const int xr = __shfl(x, row);
            const bool rv = row < bd.n_valid;
            const float4 *src = reinterpret_cast<const float4 *>(a.P + static_cast<size_t>(xr) * a.ldp);
            // every load unconditional at a clamped address, zeroed afterwards: a predicated load makes the compiler
            // branch around it and wait for each one separately
            float4 v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = src[min(8 * (c0 + c) + cs, nslots - 1)];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int slot = 8 * (c0 + c) + cs;
                if (slot < nslots) tile[row * RS + slot] = rv ? v[c] : zero4();
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    stamp.mark(it, 2);
    const int S0 = (nslots + 1) >> 1;  // slots per half
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int c = 0; c < S0; c += 4) {
        float4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int slot = h * S0 + c + i;
            const float4 t4 = tile[k * RS + min(slot, nslots - 1)];
            v[i] = c + i < S0 && slot < nslots ? t4 : zero4();
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[i].x, v[i].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[i].y, v[i].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[i].z, v[i].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[i].w, v[i].w, acc, 0, 0, 0);
        }
    }
#if CU2REC_BS_TRACE
    asm volatile("" : "+v"(acc));  // the product chain ends before the mark
#endif
    stamp.mark(it, 3);
    // the tile is symmetric: read the lane as the link k and the register's row as j; Lbuf[j][k], k contiguous
    float *L = it.Lbuf + static_cast<size_t>(g) * (kB * kB);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int j = acc_row(reg, h);
        float v = 0.f;
        if (j < k && valid) {
            const float d = static_cast<float>(k - 1 - j);
            v = a.h.lr * (exp2f(d * it.log2c) + exp2f(d * it.log2a) * acc[reg]);
        }
        L[j * kB + k] = v;
    }
    stamp.done(it, 1, g);
}

// ---- phase 2: the chains ----------------------------------------------------------------------------------------------
// Workgroup = one hot chain, 8 wavefronts.
//   wavefronts 0-3 ("solver"): the f columns of the item row are split over them (CW = 4 * SW columns each): the
//     mat-vec with the block's rows needs one exchange of 32 partial dots per block (the only __syncthreads()), the
//     transposed mat-vec needs none.  Every solver wavefront runs the forward substitution itself (same inputs, same
//     bits) instead of waiting for one that does.  They issue no global loads inside the loop.
//   wavefronts 4-7 ("loader"): stream each block's 32 user rows, L tile and base errors global -> registers -> LDS ring
//     of three slots, kDepth blocks in flight in the register file: a chain of n links moves n * (4f + 132) bytes
//     through ONE CU's load path, and that, not the arithmetic, is what a long chain takes.
template <int SW>
struct SolveShape {
    static constexpr int kCols = 4 * SW;           // columns of the item row per solver wavefront
    static constexpr int kLoads = (SW + 1) / 2;    // float4 per loader thread and tile: slots tp, tp + 8, ...
};

constexpr int kRing = 3;   // LDS slots: the block being solved, the one before it (transposed mat-vec), the next one
constexpr int kDepth = 6;  // blocks a loader keeps in flight

template <int N>
struct Stage {  // one loader thread's share of a block
    float4 rows[N];
    float4 l4;
    float base;
    uint64_t next_val;  // the thread's schedule entry of the block this stage loads next (kDepth blocks on)
};

__host__ __device__ inline int solve_slot_f4(int nslots) { return kB * (nslots | 1) + kB * kB / 4 + kB / 4; }

__host__ __device__ inline size_t solve_lds_bytes(int nslots, int sw) {
    return (static_cast<size_t>(kRing) * solve_slot_f4(nslots) + 4 * sw + 2 * kB + kB + (kBsTableFloats + 3) / 4) * 16;
}

template <int SW>
__global__ __launch_bounds__(512) void bs_solve_kernel(SgdArgs a, BsIteration it) {
    extern __shared__ float4 bs_smem[];
    constexpr int CW = SolveShape<SW>::kCols, NL = SolveShape<SW>::kLoads;
    WaveStamp stamp(it);
    const BsChainDesc cd = it.chains[blockIdx.x];
    const int begin = cd.begin, len = cd.len;
    if (len <= 0) return;  // workgroup uniform
    const int nblk = (len + kB - 1) / kB;
    const int g0 = cd.blk0;
    const int nslots = a.nslots, RS = nslots | 1;  // odd row stride (in float4): conflict-free ds_read_b128 down a column
    const int S4 = solve_slot_f4(nslots);
    const bool long_chain = nblk > kDepth;
    const int n_intervals = long_chain ? (nblk + kDepth - 1) / kDepth * kDepth : kDepth;  // both roles: this many barriers
    float4 *smem = bs_smem;
    const int tid = threadIdx.x;

    if (tid >= 256) {
        // ------------------------------------------------------------------------------------------ loader
        const int lt = tid - 256, tr = lt >> 3, tp = lt & 7;
        auto load_val = [&](int t) -> uint64_t { return it.vals[begin + min(kB * t + tr, len - 1)]; };
        // every load is unconditional (clamped addresses): a branch or a predicated load inside the ring would make the
        // compiler's s_waitcnt pass fall back to vmcnt(0) and serialise the ring
        auto issue = [&](Stage<NL> &s, int t, uint64_t val) {
            s.next_val = load_val(t + kDepth);
            const float4 *row = reinterpret_cast<const float4 *>(a.P + static_cast<size_t>(static_cast<uint32_t>(val >> 32)) * a.ldp);
#pragma unroll
            for (int i = 0; i < NL; ++i) s.rows[i] = row[min(tp + 8 * i, nslots - 1)];
            s.l4 = reinterpret_cast<const float4 *>(it.Lbuf + static_cast<size_t>(g0 + min(t, nblk - 1)) * (kB * kB))[lt];
            s.base = it.base[begin + min(kB * t + (lt & 31), len - 1)];
        };
        auto commit = [&](const Stage<NL> &s, int t) {  // block t -> ring slot t % kRing; links beyond the chain: zero rows
            float4 *sl = smem + (t % kRing) * S4;
            const bool rv = kB * t + tr < len;
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int slot = tp + 8 * i;
                if (slot < nslots) sl[tr * RS + slot] = rv ? s.rows[i] : zero4();
            }
            sl[kB * RS + lt] = s.l4;
            if (lt < kB) reinterpret_cast<float *>(sl + kB * RS + kB * kB / 4)[lt] = kB * t + lt < len ? s.base : 0.f;
        };
        Stage<NL> st[kDepth];
        if (long_chain) {
            uint64_t v[kDepth];
#pragma unroll
            for (int t = 0; t < kDepth; ++t) v[t] = load_val(t);
#pragma unroll
            for (int t = 0; t < kDepth; ++t) issue(st[t], t, v[t]);
            commit(st[0], 0);
            issue(st[0], kDepth, st[0].next_val);
            __syncthreads();
            for (int m0 = 0; m0 < n_intervals; m0 += kDepth) {
#pragma unroll
                for (int u = 0; u < kDepth; ++u) {
                    const int m = m0 + u;
                    Stage<NL> &s = st[(u + 1) % kDepth];
                    if (m == 12) stamp.mark(it, 0);
                    commit(s, m + 1);
                    if (m == 12) stamp.mark(it, 1);
                    issue(s, m + 1 + kDepth, s.next_val);
                    __syncthreads();
                    if (m == 12) stamp.mark(it, 2);
                    if (m == 13) stamp.mark(it, 3);
                }
            }
        } else {  // everything the chain needs is requested at once
#pragma unroll
            for (int t = 0; t < kDepth; ++t)
                if (t < nblk) issue(st[t], t, load_val(t));
            commit(st[0], 0);
            __syncthreads();
#pragma unroll
            for (int u = 0; u < kDepth; ++u) {
                if (u + 1 < kDepth && u + 1 < nblk) commit(st[(u + 1) % kDepth], u + 1);
                __syncthreads();
            }
        }
        stamp.done(it, 3, static_cast<int>(blockIdx.x) * 8 + (tid >> 6));
        return;
    }

    // ---------------------------------------------------------------------------------------------- solver
    const int wave = tid >> 6, lane = tid & 63, k = lane & 31, h = lane >> 5;
    const int ncols = 4 * nslots;
    float4 *qrow = smem + kRing * S4;          // [4 * SW]
    float4 *dpart = qrow + 4 * SW;             // [2][kB]: partial dots of the four solver wavefronts
    float *wbuf = reinterpret_cast<float *>(dpart + 2 * kB);  // [4][kB]
    float *tab = wbuf + 4 * kB;                // [kBsTableFloats]
    float *qrow_f = reinterpret_cast<float *>(qrow);
    for (int i = tid; i < kBsTableFloats; i += 256) tab[i] = it.tables[i];
    const int y = cd.item;
    // the item row: one column per lane
    const int col = wave * CW + lane;
    const bool has_col = lane < CW && col < ncols;
    float qc = has_col ? a.Q[static_cast<size_t>(y) * a.ldq + col] : 0.f;
    if (lane < CW) qrow_f[col] = qc;
    float b = a.item_bias[y];
    const float lr = a.h.lr;
    __syncthreads();
    for (int m = 0; m < n_intervals; ++m) {
        const bool live = m < nblk;  // workgroup uniform
        const int n = min(kB, len - kB * m);
        const float4 *tile = smem + (m % kRing) * S4;
        const float *tile_f = reinterpret_cast<const float *>(tile);
        const float *Lt = reinterpret_cast<const float *>(tile + kB * RS);
        if (m == 12) stamp.mark(it, 0);
        if (live) {
            // (A) partial dots of the block's rows with this wavefront's columns of the item row
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < SW; ++i) {
                const int slot = wave * SW + i;
                if ((i & 1) == h && slot < nslots) {
                    const float4 p = tile[k * RS + slot], q = qrow[slot];
                    acc = __builtin_fmaf(p.x, q.x, acc);
                    acc = __builtin_fmaf(p.y, q.y, acc);
                    acc = __builtin_fmaf(p.z, q.z, acc);
                    acc = __builtin_fmaf(p.w, q.w, acc);
                }
            }
            acc = half_sum(acc);
            if (h == 0) reinterpret_cast<float *>(&dpart[(m & 1) * kB + k])[wave] = acc;
        }
        __syncthreads();
        if (m == 12) stamp.mark(it, 1);
        if (!live) continue;
        // (B) right-hand side and forward substitution: after step j lane j holds e_j, lanes k > j have it eliminated
        float Lr[kB];
#pragma unroll
        for (int j = 0; j < kB; ++j) Lr[j] = Lt[j * kB + k];
        const float base = Lt[kB * kB + k];
        const float4 dp = dpart[(m & 1) * kB + k];
        const float d = ((dp.x + dp.y) + dp.z) + dp.w;
        float rhs = k < n ? (base - (b - tab[kTabCdel + k] * b)) - (d - tab[kTabAdel + k] * d) : 0.f;
#pragma unroll
        for (int j = 0; j < kB; ++j) rhs = __builtin_fmaf(-Lr[j], lane_value(rhs, j), rhs);
        const float e = rhs;
        if (m == 12) stamp.mark(it, 2);
        if (wave == 0 && h == 0 && k < n) it.ebuf[begin + kB * m + k] = e;
        // (C) the state the block leaves behind; its start state goes to phase 3
        const int back = max(n - 1 - k, 0);
        const float wk = k < n ? lr * tab[kTabApow + back] * e : 0.f;
        const float bk = k < n ? lr * tab[kTabCpow + back] * e : 0.f;
        if (h == 0) wbuf[wave * kB + k] = wk;
        const float bs = row_sum16(bk);
        b = (b - tab[kTabCdel + n] * b) + (lane_value(bs, 0) + lane_value(bs, 16));
        if (has_col) it.qstart[static_cast<size_t>(g0 + m) * a.ldq + col] = qc;
        __builtin_amdgcn_wave_barrier();
        float upd = 0.f;
        if constexpr (CW <= 32) {  // lane (column l & 31, half): 16 links each
            const int cc = wave * CW + k;
            const bool ok = k < CW && cc < ncols;
            const float4 *wv = reinterpret_cast<const float4 *>(wbuf + wave * kB + 16 * h);
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
                const float4 w4 = wv[t4];
                const float *rowp = tile_f + static_cast<size_t>((16 * h + 4 * t4) * RS) * 4 + cc;
                upd = __builtin_fmaf(w4.x, ok ? rowp[0] : 0.f, upd);
                upd = __builtin_fmaf(w4.y, ok ? rowp[RS * 4] : 0.f, upd);
                upd = __builtin_fmaf(w4.z, ok ? rowp[RS * 8] : 0.f, upd);
                upd = __builtin_fmaf(w4.w, ok ? rowp[RS * 12] : 0.f, upd);
            }
            upd = half_sum(upd);
        } else {  // lane = column, all 32 links
            const float4 *wv = reinterpret_cast<const float4 *>(wbuf + wave * kB);
#pragma unroll
            for (int t4 = 0; t4 < 8; ++t4) {
                const float4 w4 = wv[t4];
                const float *rowp = tile_f + static_cast<size_t>((4 * t4) * RS) * 4 + col;
                upd = __builtin_fmaf(w4.x, has_col ? rowp[0] : 0.f, upd);
                upd = __builtin_fmaf(w4.y, has_col ? rowp[RS * 4] : 0.f, upd);
                upd = __builtin_fmaf(w4.z, has_col ? rowp[RS * 8] : 0.f, upd);
                upd = __builtin_fmaf(w4.w, has_col ? rowp[RS * 12] : 0.f, upd);
            }
        }
        if (has_col) qc = (qc - tab[kTabAdel + n] * qc) + upd;
        if (lane < CW) qrow_f[col] = qc;
        __builtin_amdgcn_wave_barrier();
        if (m == 12) stamp.mark(it, 3);
    }
    if (has_col) a.Q[static_cast<size_t>(y) * a.ldq + col] = qc;
    if (tid == 0) a.item_bias[y] = b;
    stamp.done(it, 2, static_cast<int>(blockIdx.x) * 8 + wave);
}

// row load without predicated loads: out-of-row slots re-read the last slot and are zeroed afterwards
template <int J>
__device__ __forceinline__ Row<J> load_row_all(const float *__restrict__ base, size_t row, int ld, int nslots, int lane) {
    const float4 *p = reinterpret_cast<const float4 *>(base + row * static_cast<size_t>(ld));
    Row<J> r;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int slot = lane + kGroup * j;
        const float4 v = p[min(slot, nslots - 1)];
        r.v[j] = slot < nslots ? v : zero4();
    }
    return r;
}

// ---- beside the phases: the other chains ------------------------------------------------------------------------------
// A 16-lane group walks every chain that starts in its window of sorted positions, one update at a time (the arithmetic
// of the ordered mode, sgd_device.hpp), the next link's row in flight while this one computes.
template <int J>
__global__ __launch_bounds__(256) void bs_walk_kernel(SgdArgs a, BsIteration it) {
    const int lane = threadIdx.x & (kGroup - 1);
    const int group = (blockIdx.x * 256 + threadIdx.x) / kGroup;
    const WaveStamp stamp(it);
    const int first = *it.walk_begin, n = it.n_active;
    const int w0 = first + group * kWalkWindow;
    for (int t = 0; t < kWalkWindow; ++t) {
        const int start = w0 + t;
        if (start >= n) break;
        const uint32_t key = it.keys[start];
        if (start > first && it.keys[start - 1] == key) continue;  // the chain began in an earlier window
        const int y = it.item_of_rank[key & it.item_mask];
        Row<J> q = load_row_all<J>(a.Q, static_cast<size_t>(y), a.ldq, a.nslots, lane);
        float ib = a.item_bias[y];
        int s = start;
        uint64_t val = it.vals[s];
        uint32_t k1 = s + 1 < n ? it.keys[s + 1] : ~key;
        uint64_t v1 = s + 1 < n ? it.vals[s + 1] : 0;
        int x = static_cast<int>(val >> 32);
        Row<J> p = load_row_all<J>(a.P, static_cast<size_t>(x), a.ldp, a.nslots, lane);
        float ub = a.user_bias[x];
        for (;;) {
            const bool more = k1 == key;
            int nx = 0;
            Row<J> np = p;
            float nub = 0.f;
            uint32_t k2 = ~key;
            uint64_t v2 = 0;
            if (more) {
                nx = static_cast<int>(v1 >> 32);
                np = load_row_all<J>(a.P, static_cast<size_t>(nx), a.ldp, a.nslots, lane);
                nub = a.user_bias[nx];
                if (s + 2 < n) {
                    k2 = it.keys[s + 2];
                    v2 = it.vals[s + 2];
                }
            }
            const float rating = __uint_as_float(static_cast<uint32_t>(val));
            const float err = rating - predict<J>(p, q, ub, ib, a.global_bias);  // sgd.cu:45
            rank1_update<J>(p, q, err, a.h);                                      // mf_sequential.cu:129-137
            ib = ib + a.h.lr * (err - a.h.ib_reg * ib);                           // :141
            store_row<J>(a.P, static_cast<size_t>(x), a.ldp, a.nslots, lane, p);
            if (lane == 0) a.user_bias[x] = ub + a.h.lr * (err - a.h.ub_reg * ub);  // :140
            if (!more) break;
            ++s;
            val = v1;
            x = nx;
            p = np;
            ub = nub;
            k1 = k2;
            v1 = v2;
        }
        store_row<J>(a.Q, static_cast<size_t>(y), a.ldq, a.nslots, lane, q);
        if (lane == 0) a.item_bias[y] = ib;
    }
    if (w0 < n) stamp.done(it, 5, (blockIdx.x * 256 + threadIdx.x) >> 6);
}

// ---- phase 3: the user side of every hot block ---------------------------------------------------------------------
// One wavefront per (block, 32 columns).  T[k][j] = lr a^(k-1-j) e_j (j < k) is the A operand, the block's user rows the
// B operand, the accumulator starts from a^k q0: the result is the item row as link k saw it.
__global__ __launch_bounds__(256) void bs_update_kernel(SgdArgs a, BsIteration it, int ntiles) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int unit = blockIdx.x * 4 + wave;
    const int g = unit / ntiles, ct = unit - g * ntiles;
    if (g >= it.max_blocks) return;
    const WaveStamp stamp(it);
    const BsBlockDesc bd = it.blocks[g];
    if (bd.n_valid == 0) return;
    const int c = lane & 31, h = lane >> 5;
    const bool valid = c < bd.n_valid;
    const uint64_t val = valid ? it.vals[bd.pos0 + c] : 0;
    const int x = valid ? static_cast<int>(val >> 32) : -1;  // lane l and l + 32: link l & 31
    const float e = valid ? it.ebuf[bd.pos0 + c] : 0.f;
    const float lr = a.h.lr;
    const int ncols = 4 * a.nslots;
    const int col = 32 * ct + c;
    const bool colok = col < ncols;
    float T[16], Bv[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int j = 16 * h + s;
        const float ej = h ? lane_value(e, 16 + s) : lane_value(e, s);
        const int xj = h ? lane_value(x, 16 + s) : lane_value(x, s);
        T[s] = j < c ? lr * exp2f(static_cast<float>(c - 1 - j) * it.log2a) * ej : 0.f;  // row k = c of T
        const float pv = a.P[static_cast<size_t>(max(xj, 0)) * a.ldp + min(col, ncols - 1)];  // unconditional, see bs_gram_kernel
        Bv[s] = xj >= 0 && colok ? pv : 0.f;
    }
    const float qs_all = it.qstart[static_cast<size_t>(g) * a.ldq + min(col, ncols - 1)];
    const float qs = colok ? qs_all : 0.f;
    f32x16 acc;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) acc[reg] = qs * exp2f(static_cast<float>(acc_row(reg, h)) * it.log2a);
    float pold[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int kk0 = acc_row(reg, 0);
        const int xr = h ? lane_value(x, kk0 + 4) : lane_value(x, kk0);
        pold[reg] = a.P[static_cast<size_t>(max(xr, 0)) * a.ldp + min(col, ncols - 1)];
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(T[s], Bv[s], acc, 0, 0, 0);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int kk0 = acc_row(reg, 0);
        const int xr = h ? lane_value(x, kk0 + 4) : lane_value(x, kk0);
        const float er = h ? lane_value(e, kk0 + 4) : lane_value(e, kk0);
        if (xr >= 0 && colok)  // mf_sequential.cu:133-134
            a.P[static_cast<size_t>(xr) * a.ldp + col] = pold[reg] + lr * (er * acc[reg] - a.h.p_reg * pold[reg]);
    }
    if (ct == 0 && h == 0 && valid) {
        const float ub = a.user_bias[x];
        a.user_bias[x] = ub + lr * (e - a.h.ub_reg * ub);  // mf_sequential.cu:140
    }
    stamp.done(it, 4, unit);
}

template <int SW>
void launch_solve(const SgdArgs &a, const BsIteration &it, hipStream_t stream) {
    const size_t lds = solve_lds_bytes(a.nslots, SW);
    static bool attr_set = false;  // one per instantiation
    if (!attr_set) {
        CU2REC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(bs_solve_kernel<SW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(bs_solve_kernel<SW>, dim3(it.n_hot), dim3(512), lds, stream, a, it);
}

template <int J>
void launch_walk(const SgdArgs &a, const BsIteration &it, hipStream_t stream) {
    const int groups = (it.n_active + kWalkWindow - 1) / kWalkWindow;  // upper bound: the walk starts at walk_begin >= 0
    hipLaunchKernelGGL(bs_walk_kernel<J>, dim3((groups + 15) / 16), dim3(256), 0, stream, a, it);
}

}  // namespace

namespace {
unsigned long long *g_stamps = nullptr;
int g_stamps_cap = 0;
}  // namespace

void bs_set_stamps(unsigned long long *buf, int cap) {
    g_stamps = buf;
    g_stamps_cap = buf ? cap : 0;
}

void bs_get_stamps(unsigned long long **buf, int *cap) {
    *buf = g_stamps;
    *cap = g_stamps_cap;
}

bool bs_supported(int nslots) { return nslots >= 1 && nslots <= kBsMaxSlots; }

void bs_launch_tables(const SgdHyper &h, float *tables, hipStream_t stream) {
    hipLaunchKernelGGL(bs_tables_kernel, dim3(1), dim3(64), 0, stream, h, tables);
}

void bs_launch_plan(const uint32_t *keys, int n_active, int n_batch, int n_hot, int item_bits, int max_blocks,
                    const int *item_of_rank, int *chain_begin, BsChainDesc *chains, BsBlockDesc *blocks, int *walk_begin,
                    hipStream_t stream) {
    hipLaunchKernelGGL(bs_plan_kernel, dim3(n_batch), dim3(256), 0, stream, keys, n_active, n_hot, item_bits, max_blocks,
                       item_of_rank, chain_begin, chains, blocks, walk_begin);
}

void bs_launch_hot(const SgdArgs &a, const BsIteration &it, hipStream_t stream) {
    if (it.n_hot <= 0 || it.max_blocks <= 0) return;
    const size_t gram_lds = static_cast<size_t>(4) * kB * (a.nslots | 1) * 16;
    static bool attr_set = false;
    if (!attr_set) {
        CU2REC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(bs_gram_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(bs_gram_kernel, dim3((it.max_blocks + 3) / 4), dim3(256), gram_lds, stream, a, it);
    const int sw = (a.nslots + 3) / 4;
    if (sw <= 1) launch_solve<1>(a, it, stream);
    else if (sw <= 2) launch_solve<2>(a, it, stream);
    else if (sw <= 4) launch_solve<4>(a, it, stream);
    else if (sw <= 7) launch_solve<7>(a, it, stream);
    else if (sw <= 8) launch_solve<8>(a, it, stream);
    else if (sw <= 12) launch_solve<12>(a, it, stream);
    else if (sw <= 16) launch_solve<16>(a, it, stream);
    else fail(CU2REC_EUNSUPPORTED, "block-solve mode is compiled for n_factors <= 256");
    const int ntiles = (4 * a.nslots + 31) / 32;
    const long units = static_cast<long>(it.max_blocks) * ntiles;
    hipLaunchKernelGGL(bs_update_kernel, dim3(static_cast<unsigned>((units + 3) / 4)), dim3(256), 0, stream, a, it, ntiles);
}

void bs_launch_walk(const SgdArgs &a, const BsIteration &it, hipStream_t stream) {
    switch (slots_per_lane(a.nslots)) {
        case 1: launch_walk<1>(a, it, stream); break;
        case 2: launch_walk<2>(a, it, stream); break;
        case 3: launch_walk<3>(a, it, stream); break;
        case 4: launch_walk<4>(a, it, stream); break;
        default: fail(CU2REC_EUNSUPPORTED, "block-solve mode is compiled for n_factors <= 256");
    }
}

}  // namespace cu2rec
