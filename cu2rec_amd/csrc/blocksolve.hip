// Block-solve SGD (CU2REC_SGD_BLOCKSOLVE): the sequential semantics of mf_sequential.cu:102-143 at Hogwild-class
// speed.  Same schedule as the ordered mode (ordered.hip: counter-based sample stream -> per-iteration item chains,
// users ascending inside a chain), but a long chain is no longer walked one dependent update at a time.
//
// A chain is a sequence of affine rank-1 maps on the item's state (q, b).  For a block of B = 64 consecutive links with
// user rows p_k, a = 1 - lr*Q_reg, c = 1 - lr*item_bias_reg (tests/test_blocksolve_algebra.py pins these formulas):
//     (I + lr L) e = rhs,   L_kj  = c^(k-1-j) + a^(k-1-j) (p_k . p_j)  for j < k           unit lower triangular
//                           rhs_k = (r_k - gb - ub_k) - c^k b0 - a^k (p_k . q0)
//     q_end = a^B q0 + lr sum_j a^(B-1-j) e_j p_j,   b_end likewise
//     p_k'  = p_k + lr (e_k q^(k) - P_reg p_k),   q^(k) = a^k q0 + lr sum_{j<k} a^(k-1-j) e_j p_j
// Inside one iteration every user occurs once, so the user rows -- and with them L -- do not depend on any chain's
// progress.  Per iteration:
//   phase 1  bs_gram_kernel    every block of every hot chain in parallel, one wavefront each: the Gram matrix of the
//                              block's 64 user rows on the matrix cores (v_mfma_f32_32x32x2_f32: exact f32), scaled into
//                              lr*L, and its inverse factor M = (I + lr L)^-1 (two 32x32 triangular inversions on the
//                              vector unit, the off-diagonal tile as two more matrix products) -- so that e = M rhs
//   phase 2  bs_solve_kernel   one workgroup per hot chain walks its blocks with ONE wavefront, lane = link: a mat-vec
//                              with the block's rows, a mat-vec with M, a transposed mat-vec -- the only sequential
//                              part, no cross-wavefront exchange in it; four more wavefronts stream the blocks' rows
//                              and factors global -> registers -> LDS, four blocks ahead (one CU's load path is what
//                              bounds a long chain)
//   phase 3  bs_update_kernel  every block in parallel: the item row as each link saw it = a 64x64 (lower triangular) by
//                              64xf product (matrix cores again), then the user rows and user biases
//   beside   the other chains  in the ordered mode's own kernel (ordered.hip: two-wave form for chains of a dozen links
//                              and more, windowed walk for the rest) on a second stream: other items, other users
// Results equal the sequential ones up to float rounding (not bit for bit: the sums are associated differently).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>

#include "blocksolve.hpp"
#include "hip_check.hpp"
#include "sgd_device.hpp"

namespace cu2rec {

namespace {

using namespace dev;

constexpr int kB = kBsLinks;
constexpr int kTabAdel = 0, kTabCdel = kBsTableStride, kTabApow = 2 * kBsTableStride, kTabCpow = 3 * kBsTableStride;
constexpr int kH = 32;   // tile edge: a block is two halves of 32 links
constexpr int kMS = 36;  // LDS row stride (floats) of a 32x32 tile: 16-byte aligned rows, odd in float4 units

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
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // LDS / scalar only: stores in flight are not part of a phase
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

// ---- phase 1: the inverse factor of every block --------------------------------------------------------------------
// One workgroup of four wavefronts per block.  The 64 user rows are gathered with coalesced 128-byte pieces into LDS
// (odd row stride: a column read is conflict free), 16 rows per wavefront, all of a lane's loads in flight together.
// v_mfma_f32_32x32x2_f32 wants A[i = l & 31][kk = l >> 5] and B[kk = l >> 5][j = l & 31]: lane (k = l & 31, h = l >> 5)
// feeds half h of the columns of row k, so for the two diagonal tiles (B = A^T) the SAME register is both operands.
// The three Gram tiles are three wavefronts' work (one matrix pipe each).  With N = lr L in tiles [[N00, 0], [N10, N11]]
// (strictly lower triangular diagonal tiles):
//     M = (I + N)^-1 = [[M11, 0], [M21, M22]],  M11 = (I + N00)^-1,  M22 = (I + N11)^-1,  M21 = -M22 N10 M11.
// The two triangular inversions run side by side in the two lane halves of wavefront 0 (lane = column, forward
// substitution with the N rows broadcast from LDS one row ahead of their use, four partial sums per row); the two
// products for M21 are matrix-core work again, the first one's accumulator tile being the second one's B operand as it
// stands.
constexpr int kGramTabFloats = 2 * kB;  // a^d | c^d, d in [0, 64)
__device__ __forceinline__ size_t gram_tile_float4s(int nslots) { return static_cast<size_t>(kB) * (nslots | 1); }

__global__ __launch_bounds__(256) void bs_gram_kernel(SgdArgs a, BsIteration it) {
    extern __shared__ float4 bs_smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = blockIdx.x;
    WaveStamp stamp(it);
    const BsBlockDesc bd = it.blocks[g];
    if (bd.n_valid == 0) return;
    const int nslots = a.nslots, RS = nslots | 1;
    float4 *tile = bs_smem;  // [kB][RS]
    float *Ns = reinterpret_cast<float *>(bs_smem + gram_tile_float4s(nslots));  // N00 | N11 | N10
    float *Ms = Ns + 3 * kH * kMS;                                                // M11 | M22
    float *tab = Ms + 2 * kH * kMS;
    const bool mine = lane < bd.n_valid;  // lane l: link l
    const uint64_t val = it.vals[bd.pos0 + min(lane, bd.n_valid - 1)];
    const int x = static_cast<int>(val >> 32);
    if (wave == 3) {
        if (mine)  // what the link's error starts from: r - gb - ub (mf_sequential.cu:119-126 without b and p.q)
            it.base[bd.pos0 + lane] = (__uint_as_float(static_cast<uint32_t>(val)) - a.global_bias) - a.user_bias[x];
        tab[lane] = it.tables[kTabApow + lane];
        tab[kB + lane] = it.tables[kTabCpow + lane];
    }
    stamp.mark(it, 0);
    // gather: 8 lanes x 16 bytes per row piece, 8 rows per pass, this wavefront's 16 rows.  Every load is unconditional at
    // a clamped address: a predicated load makes the compiler branch around it and wait for each one separately.
    const int rsub = lane >> 3, cs = lane & 7;
    const int nch = (nslots + 7) >> 3;
    for (int c0 = 0; c0 < nch; c0 += 4) {
        float4 v[2][4];
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const int xr = __shfl(x, 16 * wave + 8 * pp + rsub);
            const float4 *src = reinterpret_cast<const float4 *>(a.P + static_cast<size_t>(xr) * a.ldp);
#pragma unroll
            for (int c = 0; c < 4; ++c) v[pp][c] = src[min(8 * (c0 + c) + cs, nslots - 1)];
        }
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const int row = 16 * wave + 8 * pp + rsub;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int slot = 8 * (c0 + c) + cs;
                if (slot < nslots) tile[row * RS + slot] = row < bd.n_valid ? v[pp][c] : zero4();
            }
        }
    }
    __syncthreads();
    stamp.mark(it, 1);
    const int k = lane & 31, h = lane >> 5;
    if (wave < 3) {  // 0: G00 -> N00, 1: G11 -> N11, 2: G10 -> N10 (rows: links 32.., columns: links 0..31)
        const int S0 = (nslots + 1) >> 1;  // slots per lane half
        const float4 *ra = tile + (wave == 0 ? k : kH + k) * RS, *rb = tile + (wave == 1 ? kH + k : k) * RS;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        for (int c = 0; c < S0; c += 2) {
            float4 va[2], vb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int slot = h * S0 + c + i;
                const bool ok = c + i < S0 && slot < nslots;
                const float4 t0 = ra[min(slot, nslots - 1)], t1 = rb[min(slot, nslots - 1)];
                va[i] = ok ? t0 : zero4();
                vb[i] = ok ? t1 : zero4();
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float pa[4] = {va[i].x, va[i].y, va[i].z, va[i].w}, pb[4] = {vb[i].x, vb[i].y, vb[i].z, vb[i].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[e], pb[e], acc, 0, 0, 0);
            }
        }
        const float lr = a.h.lr;
        auto entry = [&](int d, float gram) { return lr * (tab[kB + d] + tab[d] * gram); };
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int j = acc_row(reg, h);
            if (wave < 2) {
                // a diagonal tile is symmetric: read the lane as the link k and the register's row as j
                const int d = max(k - 1 - j, 0);
                Ns[(wave * kH + k) * kMS + j] = j < k && wave * kH + k < bd.n_valid ? entry(d, acc[reg]) : 0.f;
            } else {
                // the off-diagonal tile: register row r = link 32 + r, lane column = link k
                Ns[(2 * kH + j) * kMS + k] = kH + j < bd.n_valid ? entry(kH + j - 1 - k, acc[reg]) : 0.f;
            }
        }
    }
    __syncthreads();
    stamp.mark(it, 2);
    if (wave != 0) {
        stamp.done(it, 1, 4 * g + wave);
        return;
    }
    // M11 (lanes 0-31) and M22 (lanes 32-63): lane = column, m[kk] = M[kk][column]
    float m[kH];
    {
        const float *Nh = Ns + h * kH * kMS;
        float4 cur[8], nxt[8];
        m[0] = k == 0 ? 1.f : 0.f;
        cur[0] = *reinterpret_cast<const float4 *>(Nh + kMS);
#pragma unroll
        for (int kk = 1; kk < kH; ++kk) {
            if (kk + 1 < kH) {
#pragma unroll
                for (int i4 = 0; i4 < (kk + 4) / 4; ++i4) nxt[i4] = *reinterpret_cast<const float4 *>(Nh + (kk + 1) * kMS + 4 * i4);
            }
            __builtin_amdgcn_sched_barrier(0);
            float s[4] = {kk == k ? 1.f : 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i4 = 0; i4 < (kk + 3) / 4; ++i4) {
                const float nv[4] = {cur[i4].x, cur[i4].y, cur[i4].z, cur[i4].w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * i4 + e < kk) s[e] = __builtin_fmaf(-nv[e], m[4 * i4 + e], s[e]);
            }
            m[kk] = (s[0] + s[1]) + (s[2] + s[3]);
            __builtin_amdgcn_sched_barrier(0);
            if (kk + 1 < kH) {
#pragma unroll
                for (int i4 = 0; i4 < (kk + 4) / 4; ++i4) cur[i4] = nxt[i4];
            }
        }
    }
    float *Mg = it.Mbuf + static_cast<size_t>(g) * kBsFactorFloats;
#pragma unroll
    for (int kk = 0; kk < kH; ++kk) {
        Mg[(h ? 2 : 0) * kH * kH + kk * kH + k] = m[kk];
        Ms[(h * kH + kk) * kMS + k] = m[kk];
    }
    __builtin_amdgcn_wave_barrier();
    // T = N10 M11, then M21 = -M22 T
    f32x16 tacc, macc;
#pragma unroll
    for (int i = 0; i < 16; ++i) tacc[i] = macc[i] = 0.f;
    {
        const float4 *arow = reinterpret_cast<const float4 *>(Ns + (2 * kH + k) * kMS + 16 * h);  // N10[k][16h + s]
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            const float4 a4 = arow[t4];
            const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                tacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], Ms[(16 * h + 4 * t4 + e) * kMS + k], tacc, 0, 0, 0);
        }
        // step s contracts over T's rows acc_row(s, 0) and acc_row(s, 1): exactly what lane half h holds in register s
        const float *mrow = Ms + (kH + k) * kMS + 4 * h;  // M22[k][8 t + 4 h + e] = M22[k][acc_row(4 t + e, h)]
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            const float4 a4 = *reinterpret_cast<const float4 *>(mrow + 8 * t4);
            const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) macc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], tacc[4 * t4 + e], macc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) Mg[kH * kH + acc_row(reg, h) * kH + k] = -macc[reg];
    stamp.mark(it, 3);
    stamp.done(it, 1, 4 * g);
}

// ---- phase 2: the chains ----------------------------------------------------------------------------------------------
// Workgroup = one hot chain: one solver wavefront (lane = link) and four loader wavefronts.
//   solver: per block  d = P_blk q (every lane its own row against the item row, broadcast from LDS),
//           rhs -> e = M rhs (every lane its row of M against rhs, broadcast from LDS), then the item row's columns
//           (lane = column) collect sum_k w_k P_blk[k][column].  No global loads in the loop, no exchange with other
//           wavefronts; one __syncthreads() per block hands over the next LDS slot.
//   loader: streams each block's 64 user rows, factor tiles and base errors global -> registers -> LDS ring of two slots,
//           kDepth blocks in flight in the register file: a chain of n links moves n * (4f + 196) bytes through ONE
//           CU's load path, and that is what a long chain takes.
constexpr int kRing = 2;   // LDS slots: the block being solved and the next one
constexpr int kDepth = 4;  // blocks a loader keeps in flight

template <int N>
struct Stage {  // one loader thread's share of a block
    float4 rows[N];
    float4 m4[4];
    float base;
    uint64_t next_val;  // the thread's schedule entry of the block this stage loads next (kDepth blocks on)
};

// LDS slot: [kB][RS] float4 user rows | three factor tiles, rows of kMS floats (column 32 of the first 64 rows: the base error)
__host__ __device__ inline int solve_slot_f4(int nslots) { return kB * (nslots | 1) + 3 * kH * kMS / 4; }

// after the ring: the item row (4 * sw + 8 float4, zero beyond the row), rhs, w, the decay tables
__host__ __device__ inline size_t solve_lds_bytes(int nslots, int sw) {
    return (static_cast<size_t>(kRing) * solve_slot_f4(nslots) + 4 * sw + 8 + kB / 4 + kB / 4 + (kBsTableFloats + 3) / 4) * 16;
}

// Four wavefronts = one per SIMD.
template <int SW>  // float4 slots per row, rounded up to a multiple of 4: 4 * SW >= nslots
__global__ __launch_bounds__(256) void bs_solve_kernel(SgdArgs a, BsIteration it) {
    extern __shared__ float4 bs_smem[];
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

    if (tid >= 64) {
        // ------------------------------------------------------------------------------------------ loader
        constexpr int NLT = (4 * SW + 2) / 3;  // float4 per loader thread and block: slots tp, tp + 3, ...
        const int lt = tid - 64, tr = lt / 3, tp = lt - 3 * tr;
        auto load_val = [&](int t) -> uint64_t { return it.vals[begin + min(kB * t + tr, len - 1)]; };
        // every load is unconditional (clamped addresses): a branch or a predicated load inside the ring would make the
        // compiler's s_waitcnt pass fall back to vmcnt(0) and serialise the ring
        auto issue = [&](Stage<NLT> &s, int t, uint64_t val) {
            s.next_val = load_val(t + kDepth);
            const float4 *row = reinterpret_cast<const float4 *>(a.P + static_cast<size_t>(static_cast<uint32_t>(val >> 32)) * a.ldp);
#pragma unroll
            for (int i = 0; i < NLT; ++i) s.rows[i] = row[min(tp + 3 * i, nslots - 1)];
            const float4 *mg = reinterpret_cast<const float4 *>(it.Mbuf + static_cast<size_t>(g0 + min(t, nblk - 1)) * kBsFactorFloats);
#pragma unroll
            for (int q = 0; q < 4; ++q) s.m4[q] = mg[q * 192 + lt];
            s.base = it.base[begin + min(kB * t + (lt & 63), len - 1)];
        };
        auto commit = [&](const Stage<NLT> &s, int t) {  // block t -> ring slot t % kRing; links beyond the chain: zero rows
            float4 *sl = smem + (t % kRing) * S4;
            const bool rv = kB * t + tr < len;
#pragma unroll
            for (int i = 0; i < NLT; ++i) {
                const int slot = tp + 3 * i;
                if (slot < nslots) sl[tr * RS + slot] = rv ? s.rows[i] : zero4();
            }
            float4 *mt = sl + kB * RS;  // three tiles of 32 rows x 8 float4, rows kMS floats apart
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int item = q * 192 + lt;  // float4 number inside the block's factor
                mt[(item >> 3) * (kMS / 4) + (item & 7)] = s.m4[q];
            }
            if (lt < kB) reinterpret_cast<float *>(mt)[lt * kMS + kH] = kB * t + lt < len ? s.base : 0.f;
        };
        Stage<NLT> st[kDepth];
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
                    Stage<NLT> &s = st[(u + 1) % kDepth];
                    if (m == 8) stamp.mark(it, 0);
                    commit(s, m + 1);
                    if (m == 8) stamp.mark(it, 1);
                    issue(s, m + 1 + kDepth, s.next_val);
                    __syncthreads();
                    if (m == 8) stamp.mark(it, 2);
                    if (m == 9) stamp.mark(it, 3);
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
        stamp.done(it, 3, static_cast<int>(blockIdx.x) * 4 + (tid >> 6));
        return;
    }

    // ---------------------------------------------------------------------------------------------- solver
    __builtin_amdgcn_s_setprio(3);  // the chain is the critical path of the iteration; whatever shares the CU is not
    const int k = tid;  // lane = link
    constexpr int NPS = (SW + 7) / 8;  // passes of 32 slots over the item row
    float4 *qrow = smem + kRing * S4;                              // [4 * SW + 8]: the item row, zero beyond it
    float *rbuf = reinterpret_cast<float *>(qrow + 4 * SW + 8);    // [kB]: rhs
    float *wbuf = rbuf + kB;                                       // [kB]: w
    float *tab = wbuf + kB;                                        // [kBsTableFloats]
    for (int i = k; i < kBsTableFloats; i += 64) tab[i] = it.tables[i];
    const int y = cd.item;
    const int sl = k & 31, kp = k >> 5;  // transposed mat-vec: lane (slot sl of the pass, link half kp)
    float4 q4[NPS];                      // the item row, slot 32 p + sl (both lane halves hold it)
    const float4 *qsrc = reinterpret_cast<const float4 *>(a.Q + static_cast<size_t>(y) * a.ldq);
#pragma unroll
    for (int p = 0; p < NPS; ++p) {
        const int slot = 32 * p + sl;
        const float4 v = qsrc[min(slot, nslots - 1)];
        q4[p] = slot < nslots ? v : zero4();
        if (kp == 0 && slot < 4 * SW + 8) qrow[slot] = q4[p];
    }
    float b = a.item_bias[y];
    const float lr = a.h.lr;
    const int kr = k & 31;
    const bool upper = k >= kH;
    __syncthreads();
    for (int m = 0; m < n_intervals; ++m) {
        if (m < nblk) {  // workgroup uniform
            const int n = min(kB, len - kB * m);
            const float4 *tile = smem + (m % kRing) * S4;
            const float *Mt = reinterpret_cast<const float *>(tile + kB * RS);
            // this lane's row of M: links 0-31 [M11 row | 0], links 32-63 [M21 row | M22 row]
            float4 ma[8], mb[8];
            {
                const float4 *ra = reinterpret_cast<const float4 *>(Mt + ((upper ? kH : 0) + kr) * kMS);
                const float4 *rb = reinterpret_cast<const float4 *>(Mt + (2 * kH + kr) * kMS);
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    ma[t] = ra[t];
                    const float4 v = rb[t];
                    mb[t] = upper ? v : zero4();
                }
            }
            const float base = Mt[k * kMS + kH];
            const int back = max(n - 1 - k, 0);
            const float adel_k = tab[kTabAdel + k], cdel_k = tab[kTabCdel + k];
            const float apr = tab[kTabApow + back], cpr = tab[kTabCpow + back];
            const float adel_n = tab[kTabAdel + n], cdel_n = tab[kTabCdel + n];
            if (m == 8) stamp.mark(it, 0);
            // (A) every link's row against the item row.  Chunks of eight slots, the next chunk's sixteen LDS reads issued
            // before this chunk's products (the scheduling barriers keep the compiler from sinking the reads next to
            // their uses: LDS latency, not issue, is what it would pay), four accumulators; slots beyond the row
            // multiply the item row's zero padding
            float d;
            {
                constexpr int NC = (4 * SW + 7) / 8;
                float4 pbuf[2][8], qbuf[2][8];
                const float4 *prow = tile + k * RS;
                float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    pbuf[0][i] = prow[min(i, nslots - 1)];
                    qbuf[0][i] = qrow[i];
                }
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    if (c + 1 < NC) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            pbuf[(c + 1) & 1][i] = prow[min(8 * (c + 1) + i, nslots - 1)];
                            qbuf[(c + 1) & 1][i] = qrow[8 * (c + 1) + i];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 8; i += 4) {
                        const float4 *pp = &pbuf[c & 1][i], *qq = &qbuf[c & 1][i];
                        d0 = __builtin_fmaf(pp[0].x, qq[0].x, d0);
                        d1 = __builtin_fmaf(pp[1].x, qq[1].x, d1);
                        d2 = __builtin_fmaf(pp[2].x, qq[2].x, d2);
                        d3 = __builtin_fmaf(pp[3].x, qq[3].x, d3);
                        d0 = __builtin_fmaf(pp[0].y, qq[0].y, d0);
                        d1 = __builtin_fmaf(pp[1].y, qq[1].y, d1);
                        d2 = __builtin_fmaf(pp[2].y, qq[2].y, d2);
                        d3 = __builtin_fmaf(pp[3].y, qq[3].y, d3);
                        d0 = __builtin_fmaf(pp[0].z, qq[0].z, d0);
                        d1 = __builtin_fmaf(pp[1].z, qq[1].z, d1);
                        d2 = __builtin_fmaf(pp[2].z, qq[2].z, d2);
                        d3 = __builtin_fmaf(pp[3].z, qq[3].z, d3);
                        d0 = __builtin_fmaf(pp[0].w, qq[0].w, d0);
                        d1 = __builtin_fmaf(pp[1].w, qq[1].w, d1);
                        d2 = __builtin_fmaf(pp[2].w, qq[2].w, d2);
                        d3 = __builtin_fmaf(pp[3].w, qq[3].w, d3);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                d = (d0 + d1) + (d2 + d3);
            }
            if (m == 8) stamp.mark(it, 1);
            // (B) e = M rhs
            const float rhs = k < n ? (base - (b - cdel_k * b)) - (d - adel_k * d) : 0.f;
            rbuf[k] = rhs;
            __builtin_amdgcn_wave_barrier();
            float e = 0.f;
            {
                const float4 *rv = reinterpret_cast<const float4 *>(rbuf);
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const float4 r0 = rv[t], r1 = rv[8 + t];
                    e = __builtin_fmaf(ma[t].x, r0.x, e);
                    e = __builtin_fmaf(ma[t].y, r0.y, e);
                    e = __builtin_fmaf(ma[t].z, r0.z, e);
                    e = __builtin_fmaf(ma[t].w, r0.w, e);
                    e = __builtin_fmaf(mb[t].x, r1.x, e);
                    e = __builtin_fmaf(mb[t].y, r1.y, e);
                    e = __builtin_fmaf(mb[t].z, r1.z, e);
                    e = __builtin_fmaf(mb[t].w, r1.w, e);
                }
            }
            if (m == 8) stamp.mark(it, 2);
            if (k < n) it.ebuf[begin + kB * m + k] = e;
            // (C) the state the block leaves behind; its start state goes to phase 3
            wbuf[k] = k < n ? lr * apr * e : 0.f;
            const float bs = row_sum16(k < n ? lr * cpr * e : 0.f);
            b = (b - cdel_n * b) + ((lane_value(bs, 0) + lane_value(bs, 16)) + (lane_value(bs, 32) + lane_value(bs, 48)));
            __builtin_amdgcn_wave_barrier();
            float4 *qdst = reinterpret_cast<float4 *>(it.qstart + static_cast<size_t>(g0 + m) * a.ldq);
#pragma unroll
            for (int p = 0; p < NPS; ++p) {  // lane (slot, half): sum over the half's 32 links of w_k P[k][slot]
                const int slot = 32 * p + sl;
                const bool ok = slot < nslots;
                if (ok && kp == 0) qdst[slot] = q4[p];
                const float4 *colp = tile + (kH * kp) * RS + min(slot, nslots - 1);
                const float4 *wv = reinterpret_cast<const float4 *>(wbuf + kH * kp);
                float4 tb[2][16], wb[2][4];  // all 32 rows of the half requested before the first product
#pragma unroll
                for (int gq = 0; gq < 2; ++gq) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) tb[gq][i] = colp[(16 * gq + i) * RS];
#pragma unroll
                    for (int t4 = 0; t4 < 4; ++t4) wb[gq][t4] = wv[4 * gq + t4];
                }
                __builtin_amdgcn_sched_barrier(0);
                float4 u = zero4();
#pragma unroll
                for (int gq = 0; gq < 2; ++gq) {
#pragma unroll
                    for (int t4 = 0; t4 < 4; ++t4) {
                        const float wl[4] = {wb[gq][t4].x, wb[gq][t4].y, wb[gq][t4].z, wb[gq][t4].w};
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float4 p4 = tb[gq][4 * t4 + i];
                            u.x = __builtin_fmaf(wl[i], p4.x, u.x);
                            u.y = __builtin_fmaf(wl[i], p4.y, u.y);
                            u.z = __builtin_fmaf(wl[i], p4.z, u.z);
                            u.w = __builtin_fmaf(wl[i], p4.w, u.w);
                        }
                    }
                }
                u = make_float4(half_sum(u.x), half_sum(u.y), half_sum(u.z), half_sum(u.w));
                const float4 qo = q4[p];
                q4[p] = ok ? make_float4((qo.x - adel_n * qo.x) + u.x, (qo.y - adel_n * qo.y) + u.y,
                                         (qo.z - adel_n * qo.z) + u.z, (qo.w - adel_n * qo.w) + u.w)
                           : zero4();
                if (kp == 0 && slot < 4 * SW + 8) qrow[slot] = q4[p];
            }
            if (m == 8) stamp.mark(it, 3);
        }
        __syncthreads();
    }
    if (kp == 0) {
        float4 *qdst = reinterpret_cast<float4 *>(a.Q + static_cast<size_t>(y) * a.ldq);
#pragma unroll
        for (int p = 0; p < NPS; ++p)
            if (32 * p + sl < nslots) qdst[32 * p + sl] = q4[p];
    }
    if (k == 0) a.item_bias[y] = b;
    stamp.done(it, 2, static_cast<int>(blockIdx.x) * 4);
}

// ---- phase 3: the user side of every hot block ---------------------------------------------------------------------
// One wavefront per (block, 32 columns).  T[k][j] = lr a^(k-1-j) e_j (j < k), 64 x 64 lower triangular in three 32 x 32
// tiles, is the A operand, the block's user rows the B operand, the accumulators start from a^k q0: the result is the
// item row as link k saw it.  The 64 x 32 piece of P comes in with eight 16-byte loads per lane into LDS, is both the B
// operand and the old value of the update, takes the new values and leaves with eight 16-byte stores per lane; errors,
// user ids and the powers of a are read from LDS as well (a lane-dependent readlane would turn into branches).
constexpr int kUpdStride = 36;                               // floats per tile row: 16-byte rows, 2-way conflicts at most
constexpr int kUpdWaveFloats = kB * kUpdStride + 2 * kB;     // tile | e | user ids
constexpr int kUpdPowPad = 32;                               // a^d, d in [-32, 64), zero below 0
__global__ __launch_bounds__(256) void bs_update_kernel(SgdArgs a, BsIteration it, int ntiles) {
    __shared__ __attribute__((aligned(16))) float upd_smem[4 * kUpdWaveFloats + kUpdPowPad + kB];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *pw = upd_smem + 4 * kUpdWaveFloats;
    if (threadIdx.x < kUpdPowPad + kB)
        pw[threadIdx.x] = threadIdx.x < kUpdPowPad ? 0.f : it.tables[kTabApow + threadIdx.x - kUpdPowPad];
    __syncthreads();
    const int unit = blockIdx.x * 4 + wave;
    const int g = unit / ntiles, ct = unit - g * ntiles;
    if (g >= it.max_blocks) return;
    WaveStamp stamp(it);
    const BsBlockDesc bd = it.blocks[g];
    if (bd.n_valid == 0) return;
    float *tile = upd_smem + wave * kUpdWaveFloats;
    float *se = tile + kB * kUpdStride;
    int *sx = reinterpret_cast<int *>(se + kB);
    const int c = lane & 31, h = lane >> 5;
    const bool mine = lane < bd.n_valid;  // lane l: link l
    // links past the end of a short block: the last link's row (finite, loaded, never stored) with e = 0
    const uint64_t val = it.vals[bd.pos0 + min(lane, bd.n_valid - 1)];
    const int x = static_cast<int>(val >> 32);
    const float e_all = it.ebuf[bd.pos0 + min(lane, bd.n_valid - 1)];
    const float e = mine ? e_all : 0.f;
    se[lane] = e;
    sx[lane] = x;
    const float lr = a.h.lr;
    const int ncols = 4 * a.nslots;
    const int col = 32 * ct + c, colc = min(col, ncols - 1);
    const float qs = it.qstart[static_cast<size_t>(g) * a.ldq + colc];
    __builtin_amdgcn_wave_barrier();
    const int rsub = lane >> 3, cs = lane & 7;
    const int slot = 8 * ct + cs, slotc = min(slot, a.nslots - 1);
    {
        float4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            v[i] = reinterpret_cast<const float4 *>(a.P + static_cast<size_t>(sx[8 * i + rsub]) * a.ldp)[slotc];
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<float4 *>(tile + (8 * i + rsub) * kUpdStride + 4 * cs) = v[i];
    }
    __builtin_amdgcn_wave_barrier();
    // operands of step s: the contraction index is 16 h + s inside a 32-link half
    float t00[16], t10[16], t11[16], b0[16], b1[16];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const float4 e0 = *reinterpret_cast<const float4 *>(se + 16 * h + 4 * s4);
        const float4 e1 = *reinterpret_cast<const float4 *>(se + kH + 16 * h + 4 * s4);
        const float e0v[4] = {e0.x, e0.y, e0.z, e0.w}, e1v[4] = {e1.x, e1.y, e1.z, e1.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int s = 4 * s4 + i, j = 16 * h + s;
            const float near = lr * pw[kUpdPowPad + c - 1 - j];      // a^(k-1-j) inside a diagonal tile, 0 for j >= k
            t00[s] = near * e0v[i];                                  // row k = c
            t11[s] = near * e1v[i];                                  // row 32 + c, column 32 + j
            t10[s] = lr * pw[kUpdPowPad + kH + c - 1 - j] * e0v[i];  // row 32 + c, column j
            b0[s] = tile[j * kUpdStride + c];
            b1[s] = tile[(kH + j) * kUpdStride + c];
        }
    }
    f32x16 acc0, acc1;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int kk = acc_row(reg, h);
        acc0[reg] = qs * pw[kUpdPowPad + kk];
        acc1[reg] = qs * pw[kUpdPowPad + kH + kk];
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(t00[s], b0[s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(t10[s], b0[s], acc1, 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(t11[s], b1[s], acc1, 0, 0, 0);
    __builtin_amdgcn_wave_barrier();  // every B operand has been read: the tile takes the new values
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float4 ea = *reinterpret_cast<const float4 *>(se + 8 * t + 4 * h);
        const float4 eb = *reinterpret_cast<const float4 *>(se + kH + 8 * t + 4 * h);
        const float eav[4] = {ea.x, ea.y, ea.z, ea.w}, ebv[4] = {eb.x, eb.y, eb.z, eb.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int reg = 4 * t + i, kk = acc_row(reg, h);
            float *p0 = tile + kk * kUpdStride + c, *p1 = tile + (kH + kk) * kUpdStride + c;
            const float o0 = *p0, o1 = *p1;
            *p0 = o0 + lr * (eav[i] * acc0[reg] - a.h.p_reg * o0);  // mf_sequential.cu:133-134
            *p1 = o1 + lr * (ebv[i] * acc1[reg] - a.h.p_reg * o1);
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (slot < a.nslots) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 8 * i + rsub;
            if (row < bd.n_valid)
                reinterpret_cast<float4 *>(a.P + static_cast<size_t>(sx[row]) * a.ldp)[slot] =
                    *reinterpret_cast<const float4 *>(tile + row * kUpdStride + 4 * cs);
        }
    }
    if (ct == 0 && mine) {
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
    hipLaunchKernelGGL(bs_solve_kernel<SW>, dim3(it.n_hot), dim3(256), lds, stream, a, it);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess)
        fail(CU2REC_EHIP, std::string("bs_solve_kernel launch failed: ") + hipGetErrorString(err) + " (SW " + std::to_string(SW) +
                              ", chains " + std::to_string(it.n_hot) + ", LDS " + std::to_string(lds) + " bytes)");
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
    hipLaunchKernelGGL(bs_tables_kernel, dim3(1), dim3(128), 0, stream, h, tables);
}

void bs_launch_plan(const uint32_t *keys, int n_active, int n_batch, int n_hot, int item_bits, int max_blocks,
                    const int *item_of_rank, int *chain_begin, BsChainDesc *chains, BsBlockDesc *blocks, int *walk_begin,
                    hipStream_t stream) {
    hipLaunchKernelGGL(bs_plan_kernel, dim3(n_batch), dim3(256), 0, stream, keys, n_active, n_hot, item_bits, max_blocks,
                       item_of_rank, chain_begin, chains, blocks, walk_begin);
}

void bs_launch_gram(const SgdArgs &a, const BsIteration &it, hipStream_t stream) {
    if (it.n_hot <= 0 || it.max_blocks <= 0) return;
    const size_t gram_lds = static_cast<size_t>(kB) * (a.nslots | 1) * 16 + (static_cast<size_t>(5) * kH * kMS + kGramTabFloats) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        CU2REC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(bs_gram_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(bs_gram_kernel, dim3(it.max_blocks), dim3(256), gram_lds, stream, a, it);
    CU2REC_HIP(hipGetLastError());
}

void bs_launch_solve(const SgdArgs &a, const BsIteration &it, hipStream_t stream) {
    if (it.n_hot <= 0 || it.max_blocks <= 0) return;
    const int sw = (a.nslots + 3) / 4;
    if (sw <= 1) launch_solve<1>(a, it, stream);
    else if (sw <= 2) launch_solve<2>(a, it, stream);
    else if (sw <= 4) launch_solve<4>(a, it, stream);
    else if (sw <= 7) launch_solve<7>(a, it, stream);
    else if (sw <= 8) launch_solve<8>(a, it, stream);
    else if (sw <= 12) launch_solve<12>(a, it, stream);
    else if (sw <= 16) launch_solve<16>(a, it, stream);
    else fail(CU2REC_EUNSUPPORTED, "block-solve mode is compiled for n_factors <= 256");
}

void bs_launch_update(const SgdArgs &a, const BsIteration &it, hipStream_t stream) {
    if (it.n_hot <= 0 || it.max_blocks <= 0) return;
    const int ntiles = (4 * a.nslots + 31) / 32;
    const long units = static_cast<long>(it.max_blocks) * ntiles;
    hipLaunchKernelGGL(bs_update_kernel, dim3(static_cast<unsigned>((units + 3) / 4)), dim3(256), 0, stream, a, it, ntiles);
}

}  // namespace cu2rec
