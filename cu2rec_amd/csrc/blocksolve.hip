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
//   phase 2  bs_solve_kernel   one workgroup per hot chain walks its blocks, lane = link, four solver wavefronts splitting
//                              each step (a mat-vec with the block's rows, a mat-vec with M, a transposed mat-vec) -- the only
//                              sequential part; four more wavefronts stream the blocks' rows and factors global -> registers
//                              -> LDS, four blocks ahead; the leading (longest) chains in the look-ahead form, whose dependent
//                              path per block is two 64 x 64 mat-vecs on one wavefront (chain_lookahead)
//   phase 3  bs_update_kernel  every block in parallel: one wavefront per 64 columns walks down the block's links with the
//                              reference's own recurrence, the item row as each link saw it in a register; user rows and biases
//   beside   the other chains  in the ordered mode's own kernel (ordered.hip: two-wave form for chains of a dozen links
//                              and more, windowed walk for the rest) on a second stream: other items, other users
// Results equal the sequential ones up to float rounding (not bit for bit: the sums are associated differently).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

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
// native vectors: a float4 (a struct) is copied with memcpy, and an array of them written that way stayed in scratch;
// pairs of them feed v_pk_fma_f32 (two FMAs an issue slot) straight from the registers a 16-byte LDS read filled
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 lds4(const float4 *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ f32x2 lo2(f32x4 v) { return __builtin_shufflevector(v, v, 0, 1); }
__device__ __forceinline__ f32x2 hi2(f32x4 v) { return __builtin_shufflevector(v, v, 2, 3); }
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 dup2(float v) { return f32x2{v, v}; }
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned ld_agent(const unsigned *p) {  // global_load sc1: never served from this CU's L1
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every device-side wait in here is bounded: BsIteration::wait_ticks (bs_wait_ticks(): CU2REC_BS_WAIT_S seconds, default 2) of the 100 MHz wall clock

// v[l] + v[l ^ 16] in every lane
__device__ __forceinline__ float row_pair_sum(float v) {
    const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}

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
#ifndef CU2REC_BS_ABLATE  // timing-only builds (tools/build_variant.sh): 1 no global stores in the solver's loop, 2 no (C) products,
                          // 4 the loaders do not write user rows into LDS, 8 no (A) products, 16 no (B) products
#define CU2REC_BS_ABLATE 0
#endif
constexpr int kStampWords = 8;
struct WaveStamp {
    unsigned long long t0;
#if CU2REC_BS_TRACE
    unsigned long long marks[4] = {0, 0, 0, 0};
    unsigned long long c0 = clock64();  // shader clock (s_memtime): with the 100 MHz wall clock, the frequency the wavefront ran at
#endif
    __device__ __forceinline__ explicit WaveStamp(const BsIteration &it) : t0(it.stamps ? wall_clock64() : 0) {}
    __device__ __forceinline__ void restart(const BsIteration &it) { t0 = it.stamps ? wall_clock64() : 0; }  // (a persistent workgroup's next unit)
    __device__ __forceinline__ void fine(const BsIteration &it, int i) {  // -DCU2REC_BS_TRACE=2: the inside of phases 1 and 3
#if CU2REC_BS_TRACE == 2
        if (it.stamps) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            marks[i] = wall_clock64();
        }
#endif
    }
    __device__ __forceinline__ void mark(const BsIteration &it, int i) {
#if CU2REC_BS_TRACE == 1
        if (it.stamps) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // LDS / scalar only: stores in flight are not part of a phase
            marks[i] = wall_clock64();
        }
#endif
    }
    __device__ __forceinline__ void span(int i, unsigned long long ticks) {  // trace builds: a duration, shown as a mark that far behind the start
#if CU2REC_BS_TRACE == 1
        marks[i] = t0 + ticks;
#endif
    }
    __device__ __forceinline__ void done(const BsIteration &it, int kernel, int id) const {
        if (!it.stamps || (threadIdx.x & 63) != 0) return;
        const int seg = it.stamps_cap / 8;  // one segment of the buffer per kernel: no atomics, no contention
        if (id >= seg) return;
        unsigned long long *r = it.stamps + 1 + kStampWords * (static_cast<size_t>(kernel) * seg + id);
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        r[0] = static_cast<unsigned long long>(kernel) | (static_cast<unsigned long long>(xcc & 15u) << 32);
        r[1] = static_cast<unsigned long long>(id);
        r[2] = t0;
        r[3] = wall_clock64();
#if CU2REC_BS_TRACE
        for (int i = 0; i < 4; ++i) r[4 + i] = marks[i];
        if (kernel == 3 || kernel == 4) r[7] = t0 + (clock64() - c0);  // loaders, update: shader cycles in the slot of the last mark
#endif
    }
};

// ---- decay tables: x * a^k is applied as x - adel[k] * x so that the rounding of a^k (one ulp of 1.0, the same sign
// every block) cannot bias a hot item's decay rate; the deltas are computed in double from the float hyper-parameters
__global__ void bs_tables_kernel(SgdHyper h, float *__restrict__ t) {
    const int k = threadIdx.x;
    if (k > 2 * kB) return;
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
                                                      BsBlockDesc *__restrict__ blocks, int *__restrict__ walk_begin, size_t stride) {
    __shared__ int s_part[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const uint32_t *kb = keys + static_cast<size_t>(b) * stride;  // iteration b's share of the sorted keys
    int *cb = chain_begin + static_cast<size_t>(b) * (n_hot + 1);
    BsChainDesc *cd = chains + static_cast<size_t>(b) * max(n_hot, 1);
    BsBlockDesc *bd = blocks + static_cast<size_t>(b) * max_blocks;
    for (int r = tid; r <= n_hot; r += 256) cb[r] = lower_bound_key(kb, n_active, static_cast<uint32_t>(r));
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

// ---- which chains take the look-ahead form (phase 2, chain_lookahead) -------------------------------------------------------
// The longest chains ARE the iteration's critical path.  For popularity ranks below la_ranks phase 1 also builds, for every block
// but the chain's first, the block of lr L that couples it to the block before it (gram_cross_block), and phase 2 then needs the
// item row for nothing on its dependent path.  The chain's blocks must lie below la_cap (the cross blocks' buffer).
__device__ __forceinline__ bool chain_is_lookahead(const BsIteration &it, int chain, const BsChainDesc &cd) {
    const int nblk = (cd.len + kB - 1) / kB;
    return chain < it.la_ranks && nblk >= 2 && cd.blk0 + nblk <= it.la_cap;
}

// ---- phase 1, look-ahead chains: the cross block N_i = lr L[block i][block i - 1] ---------------------------------------------------
//     N_i[k][j] = lr (c^(63 + k - j) + a^(63 + k - j) (p_(i,k) . p_(i-1,j)))        (tests/test_blocksolve_algebra.py, lookahead_chain)
// One workgroup of four wavefronts, one 32 x 32 tile each.  Both blocks' 128 rows are gathered into LDS the way phase 1 gathers its
// own (coalesced 128-byte pieces, every load unconditional at a clamped address) and the MFMA operands are read from there: with every
// lane reading its two rows straight from memory (the first form) a workgroup took 18 us -- 64 different rows per load instruction.
// Rows of links past the block's end are zero.  LDS: 2 x 64 rows (the launch asks for that much when there are such workgroups).
__device__ __forceinline__ void gram_cross_block(const SgdArgs &a, const BsIteration &it, int g, float4 *smem) {
    const BsBlockDesc bd = it.blocks[g];
    if (bd.n_valid == 0 || bd.m == 0 || !chain_is_lookahead(it, bd.chain, it.chains[bd.chain])) return;  // workgroup uniform
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, k = lane & 31, h = lane >> 5;
    const int rt = wave >> 1, ct = wave & 1;  // tile: rows 32 rt .. of this block, columns 32 ct .. of the previous one
    const int nslots = a.nslots, RS = nslots | 1, S0 = (nslots + 1) >> 1;
    float4 *tile_c = smem, *tile_p = smem + kB * RS;  // this block's rows | the previous block's
    {
        // lane l: link l of this block and of the one before it (all 64 of those exist)
        const int xc = static_cast<int>(it.vals[bd.pos0 + min(lane, bd.n_valid - 1)] >> 32);
        const int xp = static_cast<int>(it.vals[bd.pos0 - kB + lane] >> 32);
        const int rsub = lane >> 3, cs = lane & 7;
        const int nch = (nslots + 7) >> 3;
        for (int c0 = 0; c0 < nch; c0 += 2) {
            float4 vc[2][2], vp[2][2];
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const int src_lane = 16 * wave + 8 * pp + rsub;
                const float4 *sc = reinterpret_cast<const float4 *>(a.P + static_cast<size_t>(__shfl(xc, src_lane)) * a.ldp);
                const float4 *sp = reinterpret_cast<const float4 *>(a.P + static_cast<size_t>(__shfl(xp, src_lane)) * a.ldp);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    vc[pp][c] = sc[min(8 * (c0 + c) + cs, nslots - 1)];
                    vp[pp][c] = sp[min(8 * (c0 + c) + cs, nslots - 1)];
                }
            }
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const int row = 16 * wave + 8 * pp + rsub;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int slot = 8 * (c0 + c) + cs;
                    if (slot < nslots) {
                        tile_c[row * RS + slot] = row < bd.n_valid ? vc[pp][c] : zero4();
                        tile_p[row * RS + slot] = vp[pp][c];
                    }
                }
            }
        }
    }
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    {
        const float4 *ra = tile_c + (kH * rt + k) * RS, *rb = tile_p + (kH * ct + k) * RS;
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
    }
    const float lr = a.h.lr;
    float *dst = it.Nbuf + static_cast<size_t>(g) * kBsCrossFloats;
    const int j = kH * ct + k;  // this lane's column: link j of the previous block
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int i = kH * rt + acc_row(reg, h);  // link i of this block
        const float fd = static_cast<float>(kB - 1 + i - j);  // 0 .. 126
        const float v = lr * (__builtin_amdgcn_exp2f(fd * it.log2c) + __builtin_amdgcn_exp2f(fd * it.log2a) * acc[reg]);
        dst[i * kB + j] = i < bd.n_valid ? v : 0.f;
    }
    __syncthreads();  // (the next block of this workgroup's loop writes the tiles again)
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
__host__ __device__ inline size_t gram_tile_float4s(int nslots) { return static_cast<size_t>(kB) * (nslots | 1); }

// One block of phase 1 (the body of bs_gram_kernel): its rows -> LDS, the Gram tiles, (I + N)^-1, the record.  Workgroup uniform.
__device__ __forceinline__ void gram_block(const SgdArgs &a, const BsIteration &it, int g, float4 *bs_smem, WaveStamp &stamp) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const BsBlockDesc bd = it.blocks[g];
    if (bd.n_valid == 0) return;  // (the grid is an upper bound of the iteration's blocks)
    const int nslots = a.nslots, RS = nslots | 1;
    // The five 32 x 32 tiles TAKE THE PLACE of the user rows once the Gram products have read them (barrier below): 34.6 instead of
    // 57.6 KB of LDS at f = 128 -- three workgroups a CU instead of two on the Netflix shape, whose 7,500 blocks come in rounds
    float4 *tile = bs_smem;  // [kB][RS]
    float *Ms = reinterpret_cast<float *>(bs_smem);                               // M11 | M22
    float *Ns = Ms + 2 * kH * kMS;                                                // N00 (later M21) | N11 | N10
    float *tab = Ms + max(static_cast<int>(4 * gram_tile_float4s(nslots)), 5 * kH * kMS);
    float *basev = tab + kGramTabFloats;      // [kB] r - gb - ub, 0 past the end of a short block
    const bool mine = lane < bd.n_valid;  // lane l: link l
    const uint64_t val = it.vals[bd.pos0 + min(lane, bd.n_valid - 1)];
    const int x = static_cast<int>(val >> 32);
    if (wave == 3) {
        // what the link's error starts from: r - gb - ub (mf_sequential.cu:119-126 without b and p.q)
        const float r0 = (__uint_as_float(static_cast<uint32_t>(val)) - a.global_bias) - a.user_bias[x];
        basev[lane] = mine ? r0 : 0.f;
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
    stamp.fine(it, 0);  // rows gathered
    const int k = lane & 31, h = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    if (wave < 3) {  // 0: G00 -> N00, 1: G11 -> N11, 2: G10 -> N10 (rows: links 32.., columns: links 0..31)
        // D[i][j] = (A row i) . (B row j) lands in lane j = l & 31 at the rows i = acc_row(reg, h) of its registers.  The lane's own
        // link is the tile's ROW here, the registers' index its columns -- for the diagonal tiles by symmetry, for N10 because
        // A = links 0..31 and B = links 32..63: every lane then holds 4 x 4 CONTIGUOUS columns of its row and writes them with four
        // 16-byte LDS stores (round 4; it was sixteen scattered 4-byte stores behind two table reads each: 3.3 us of phase 1)
        const int S0 = (nslots + 1) >> 1;  // slots per lane half
        const float4 *ra = tile + (wave == 1 ? kH + k : k) * RS, *rb = tile + (wave == 0 ? k : kH + k) * RS;
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
    }
    stamp.fine(it, 1);  // Gram products issued (the accumulator is read right behind)
    __syncthreads();  // every wavefront is through with the rows: the tiles take their place
    if (wave < 3) {
        // N[K][J] = lr (c^d + a^d G[K][J]), d = K - 1 - J (links of the block), for J < K < n_valid; the powers by v_exp_f32 from
        // log2 a, log2 c (they multiply lr-sized terms only: an ulp of the power is 1e-9 of an entry)
        const float lr = a.h.lr;
        const int K = (wave == 0 ? 0 : kH) + k;            // this lane's link: the row
        const int J0 = (wave == 1 ? kH : 0) + 4 * h;       // first column of register 0
        const bool row_ok = K < bd.n_valid;
        float *dst = Ns + (wave * kH + k) * kMS + 4 * h;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            f32x4 out;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int d = K - 1 - (J0 + 8 * m + e);
                const float fd = static_cast<float>(d);
                const float v = lr * (__builtin_amdgcn_exp2f(fd * it.log2c) + __builtin_amdgcn_exp2f(fd * it.log2a) * acc[4 * m + e]);
                out[e] = d >= 0 && row_ok ? v : 0.f;
            }
            *reinterpret_cast<f32x4 *>(dst + 8 * m) = out;
        }
    }
    __syncthreads();
    stamp.mark(it, 2);
    stamp.fine(it, 2);  // lr L in LDS
    if (wave == 0) {
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
#pragma unroll
    for (int kk = 0; kk < kH; ++kk) Ms[(h * kH + kk) * kMS + k] = m[kk];
    __builtin_amdgcn_wave_barrier();
    stamp.fine(it, 3);  // the two triangular inverses
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
    for (int reg = 0; reg < 16; ++reg) Ns[acc_row(reg, h) * kMS + k] = -macc[reg];  // M21 where N00 was (its last reader is done)
    stamp.mark(it, 3);
    }  // wave 0
    __syncthreads();
    // The block's record leaves the workgroup: [M11 | M21 | M22 | base], 16 bytes per lane (phase 2 is queued behind this launch:
    // the kernel boundary publishes it).
    float *rec = it.Mbuf + static_cast<size_t>(g) * kBsRecFloats;
    {
        const int t = threadIdx.x, row = t >> 3, c4 = t & 7;
        const float *src[3] = {Ms, Ns, Ms + kH * kMS};
#pragma unroll
        for (int q = 0; q < 3; ++q) *reinterpret_cast<f32x4 *>(rec + q * kH * kH + 4 * t) = *reinterpret_cast<const f32x4 *>(src[q] + row * kMS + 4 * c4);
        if (t < kB / 4) *reinterpret_cast<f32x4 *>(rec + kBsFactorFloats + 4 * t) = *reinterpret_cast<const f32x4 *>(basev + 4 * t);
    }
    stamp.done(it, 1, 4 * g + wave);
}

__global__ __launch_bounds__(256, 4) void bs_gram_kernel(SgdArgs a, BsIteration it) {
    extern __shared__ float4 bs_smem[];
    const int wave = threadIdx.x >> 6;
    const int g = static_cast<int>(blockIdx.x) - it.la_grid;  // (the first la_grid workgroups of the launch: the cross blocks)
    __builtin_amdgcn_s_setprio(2);  // (above the walk beside us and the schedule's kernels: this launch is on the iteration's critical path)
    WaveStamp stamp(it);
    // (every workgroup counts itself through, used or not: the side stream's gate waits for the whole grid)
    struct Through {
        unsigned long long *count;
        __device__ ~Through() {
            // 32 shards, 128 bytes apart: two thousand atomics on ONE word took longer than phase 1 itself
            if (count && threadIdx.x == 0)
                __hip_atomic_fetch_add(count + 16 * (blockIdx.x & 31), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } through_count{it.gram_done};
    if (g < 0) {
        // the cross blocks of the look-ahead chains, whose blocks are the first of the table; dispatched first: they have nothing to
        // wait for and end long before the inversions do
        const BsChainDesc last = it.chains[it.la_ranks - 1];
        const int la_end = min(last.blk0 + (last.len + kB - 1) / kB, it.la_cap);
        for (int gb = g + it.la_grid; gb < la_end; gb += it.la_grid) gram_cross_block(a, it, gb, bs_smem);
        stamp.done(it, 7, 4 * (g + it.la_grid) + wave);
        return;
    }
    // (one workgroup per slot of the block table, max_blocks of them -- the worst case; most find their descriptor empty and leave.  Launching
    // only as many as an iteration is expected to fill needs a way to do the rest when an iteration has more: as a stride loop around
    // gram_block it cost this kernel 55 spilled registers, as an out-of-line function a scratch frame -- 92 instead of 79 us per
    // iteration on the ML-20M shape either way (round 6, profiles/r06_launch_bounds.txt).  Phase 3 and the walk, whose kernels have the
    // registers to spare, do stride: bs_update_kernel, sgd_ordered_kernel.)
    gram_block(a, it, g, bs_smem, stamp);
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
// (the plain form's copy of the decay tables in LDS: powers 0 .. 64 only -- at 63 slots a row its LDS has no room for more)
constexpr int kSolTabStride = kB + 1, kSolTabFloats = 4 * kSolTabStride;
constexpr int kSolTabAdel = 0, kSolTabCdel = kSolTabStride, kSolTabApow = 2 * kSolTabStride, kSolTabCpow = 3 * kSolTabStride;
constexpr int kRing = 2;   // LDS slots: the block being solved and the next one
constexpr int kDepth = 4;  // blocks a loader keeps in flight

template <int N>
struct Stage {  // one loader thread's share of a block
    f32x4 rows[N];
    f32x4 m4[3];
    float base;
    uint64_t next_val;  // the thread's schedule entry of the block this stage loads next (kDepth blocks on)
};

// LDS slot: [kB][RS] float4 user rows | three factor tiles, rows of kMS floats (column 32 of the first 64 rows: the base error)
__host__ __device__ inline int solve_slot_f4(int nslots) { return kB * (nslots | 1) + 3 * kH * kMS / 4; }

// after the ring: the item row (4 * sw + 8 float4, zero beyond the row), the partial dots and errors, four copies of rhs
// and w, the decay tables, a row of zeros
__host__ __device__ inline size_t solve_lds_bytes(int nslots, int sw) {
    return (static_cast<size_t>(kRing) * solve_slot_f4(nslots) + 4 * sw + 8 + 2 * kB + 8 * kB / 4 + (kSolTabFloats + 3) / 4 + 8 + 1) * 16;
}

// The four solver wavefronts meet twice inside a block; s_barrier would drag the loaders along (their work of an interval
// -- 25 KB of LDS writes, a few hundred cache-line requests -- would then sit on the chain's critical path), so they count
// arrivals in LDS instead.  LDS operations of a wavefront complete in order: the partial results written before the
// ds_add are visible to whoever sees the count.
__device__ __forceinline__ void solvers_meet(unsigned *meet, unsigned target, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_fetch_add(meet, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(meet, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Eight wavefronts: four solvers (one per SIMD) and four loaders.
//   loaders: stream the chain's blocks (64 user rows, the factor tiles, the base errors) into a ring of two LDS slots,
//            kDepth blocks in flight in their registers; every load unconditional at a clamped address.
//   solvers: all four hold the same 64 links (lane = link) and split every step of a block four ways:
//     (A) d_k = p_k . q      each wavefront a quarter of the row (SW slots), partial sums through LDS            -- meet
//     (B) e = M rhs          each wavefront 16 columns of M (rhs is cheap and computed four times), partials     -- meet
//     (C) q <- a^n q + sum_k w_k p_k   each wavefront its quarter of the row: lane = (slot, group of 8 links), the
//                            eight groups summed across lanes (DPP row rotate, permlane16/32 swaps)              -- barrier
//            (the barrier is the one per block all eight wavefronts share: it hands the ring slot back to the loaders)
//   One wavefront doing all of it issued ~500 instructions per block and took 1.35 us of the block's 1.8 us; the
//   loaders alone sustain a block per 0.42 us (tools/build_variant.sh ablations, profiles/README.md).
// SW: float4 slots per row quarter (4 * SW >= nslots).
template <int SW>
__device__ __forceinline__ void solve_chain(const SgdArgs &a, const BsIteration &it, float4 *bs_smem, WaveStamp &stamp, int chain,
                                            const BsChainDesc &cd) {
    const int begin = cd.begin, len = cd.len;
    if (len <= 0) return;  // workgroup uniform
    const int nblk = (len + kB - 1) / kB;
    const int g0 = cd.blk0;
    const int nslots = a.nslots, RS = nslots | 1;  // odd row stride (in float4): conflict-free ds_read_b128 down a column
    const int S4 = solve_slot_f4(nslots);
    const bool long_chain = nblk > kDepth;
    const int n_intervals = long_chain ? (nblk + kDepth - 1) / kDepth * kDepth : kDepth;  // both roles: 3 barriers each
    float4 *smem = bs_smem;
    const int tid = threadIdx.x;

    if (tid >= 256) {
        // ------------------------------------------------------------------------------------------ loader
        __builtin_amdgcn_s_setprio(2);
        constexpr int NLT = SW;  // float4 per loader thread and block: slots tp, tp + 4, ...
        const int lt = tid - 256, tr = lt >> 2, tp = lt & 3;
        auto load_val = [&](int t) -> uint64_t { return it.vals[begin + min(kB * t + tr, len - 1)]; };
        // every load is unconditional (clamped addresses): a branch or a predicated load inside the ring would make the
        // compiler's s_waitcnt pass fall back to vmcnt(0) and serialise the ring
        auto issue_rows = [&](Stage<NLT> &s, int t, uint64_t val) {  // the schedule and the user rows: nothing phase 1 writes
            s.next_val = load_val(t + kDepth);
            const f32x4 *row = reinterpret_cast<const f32x4 *>(a.P + static_cast<size_t>(static_cast<uint32_t>(val >> 32)) * a.ldp);
#pragma unroll
            for (int i = 0; i < NLT; ++i) s.rows[i] = (CU2REC_BS_ABLATE & 32) ? f32x4{0.01f, 0.01f, 0.01f, 0.01f} : row[min(tp + 4 * i, nslots - 1)];
        };
        auto issue_rec = [&](Stage<NLT> &s, int t) {  // phase 1's record of the block
            const float *rec = it.Mbuf + static_cast<size_t>(g0 + min(t, nblk - 1)) * kBsRecFloats;
            const f32x4 *mg = reinterpret_cast<const f32x4 *>(rec);
#pragma unroll
            for (int q = 0; q < 3; ++q) s.m4[q] = mg[q * 256 + lt];
            s.base = rec[kBsFactorFloats + (lt & 63)];
        };
        auto issue = [&](Stage<NLT> &s, int t, uint64_t val) {
            issue_rows(s, t, val);
            issue_rec(s, t);
        };
        auto commit = [&](const Stage<NLT> &s, int t) {  // block t -> ring slot t % kRing; links beyond the chain: zero rows
            f32x4 *sl = reinterpret_cast<f32x4 *>(smem + (t % kRing) * S4);
            const bool rv = kB * t + tr < len;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NLT; ++i) {
                const int slot = tp + 4 * i;
                if (!(CU2REC_BS_ABLATE & 4) && slot < nslots) sl[tr * RS + slot] = rv ? s.rows[i] : z;
            }
            f32x4 *mt = sl + kB * RS;  // three tiles of 32 rows x 8 float4, rows kMS floats apart
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int item = q * 256 + lt;  // float4 number inside the block's factor
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
                    commit(s, m + 1);
                    issue(s, m + 1 + kDepth, s.next_val);
                    __syncthreads();
                }
            }
        } else {  // everything the chain needs is requested at once
            uint64_t v[kDepth];
#pragma unroll
            for (int t = 0; t < kDepth; ++t) v[t] = load_val(min(t, nblk - 1));
#pragma unroll
            for (int t = 0; t < kDepth; ++t)
                if (t < nblk) issue(st[t], t, v[t]);
            commit(st[0], 0);
            __syncthreads();
#pragma unroll
            for (int u = 0; u < kDepth; ++u) {
                if (u + 1 < kDepth && u + 1 < nblk) commit(st[(u + 1) % kDepth], u + 1);
                __syncthreads();
            }
        }
        stamp.done(it, 3, chain * 4 + ((tid >> 6) & 3));
        return;
    }

    // ---------------------------------------------------------------------------------------------- solvers
    __builtin_amdgcn_s_setprio(3);  // the chain is the critical path of the iteration; whatever shares the CU is not
    const int w = tid >> 6, k = tid & 63;  // wavefront = quarter, lane = link
    constexpr int NP = (SW + 7) / 8;       // passes of 8 slots over this wavefront's quarter of the item row
    float4 *qrow = smem + kRing * S4;                              // [4 * SW + 8]: the item row, zero beyond it
    float4 *dpart = qrow + 4 * SW + 8;                             // [kB]: link k's four partial dot products
    float4 *epart = dpart + kB;                                    // [kB]: link k's four partial errors
    float *rbuf = reinterpret_cast<float *>(epart + kB) + w * kB;  // [4][kB]: rhs, one copy per wavefront
    float *wbuf = reinterpret_cast<float *>(epart + kB) + (4 + w) * kB;  // [4][kB]: w, one copy per wavefront
    float *tab = reinterpret_cast<float *>(epart + kB) + 8 * kB;   // [kSolTabFloats]
    float4 *zrow = reinterpret_cast<float4 *>(tab) + (kSolTabFloats + 3) / 4;  // [8] zeros
    unsigned *meet = reinterpret_cast<unsigned *>(zrow + 8);                    // [1] arrivals at the solvers' meeting points
    if (w == 0) {
        for (int i = k; i < kSolTabFloats; i += 64) tab[i] = it.tables[(i / kSolTabStride) * kBsTableStride + i % kSolTabStride];
        if (k < 8) zrow[k] = zero4();
        if (k == 0) *meet = 0u;
    }
    unsigned met = 0;
    const int y = cd.item;
    const int sl = k & 7, kg = k >> 3;  // (C): lane (slot sl of the pass, links 8 kg .. 8 kg + 7)
    f32x4 q4[NP];                       // this wavefront's quarter of the item row, slot w SW + 8 p + sl (every kg holds it)
    const f32x4 *qsrc = reinterpret_cast<const f32x4 *>(a.Q + static_cast<size_t>(y) * a.ldq);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int sq = 8 * p + sl, slot = w * SW + sq;
        const f32x4 v = qsrc[min(slot, nslots - 1)];
        q4[p] = sq < SW && slot < nslots ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        if (kg == 0 && sq < SW) *reinterpret_cast<f32x4 *>(qrow + slot) = q4[p];
    }
    float b = a.item_bias[y];
    const float lr = a.h.lr;
    const int kr = k & 31;
    const bool upper = k >= kH;
    stamp.mark(it, 0);
    __syncthreads();  // block 0 is in the ring
    stamp.mark(it, 1);
    for (int m = 0; m < n_intervals; ++m) {
        const bool live = m < nblk;  // workgroup uniform
        const int n = min(kB, len - kB * m);
        const float4 *tile = smem + (m % kRing) * S4;
        const float *Mt = reinterpret_cast<const float *>(tile + kB * RS);
        float base = 0.f, adel_k = 0.f, cdel_k = 0.f, apr = 0.f, cpr = 0.f, adel_n = 0.f, cdel_n = 0.f;
        f32x4 mq[4];
        if (live) {
            // this lane's 16 columns of its row of M: links 0-31 [M11 row | 0], links 32-63 [M21 row | M22 row]
            const float4 *mrow = w < 2 ? reinterpret_cast<const float4 *>(Mt + ((upper ? kH : 0) + kr) * kMS + 16 * w)
                                       : (upper ? reinterpret_cast<const float4 *>(Mt + (2 * kH + kr) * kMS + 16 * (w - 2)) : zrow);
#pragma unroll
            for (int t = 0; t < 4; ++t) mq[t] = lds4(mrow + t);
            base = Mt[k * kMS + kH];
            const int back = max(n - 1 - k, 0);
            adel_k = tab[kSolTabAdel + k], cdel_k = tab[kSolTabCdel + k];
            apr = tab[kSolTabApow + back], cpr = tab[kSolTabCpow + back];
            adel_n = tab[kSolTabAdel + n], cdel_n = tab[kSolTabCdel + n];
            // (A) this wavefront's quarter of every link's row against the item row; slots beyond the row multiply the
            // item row's zero padding
            const float4 *prow = tile + k * RS;
            f32x4 pv[SW], qv[SW];
#pragma unroll
            for (int i = 0; i < SW; ++i) {
                pv[i] = lds4(prow + min(w * SW + i, nslots - 1));
                qv[i] = lds4(qrow + w * SW + i);
            }
            f32x2 dacc[2] = {dup2(0.f), dup2(0.f)};
            if (!(CU2REC_BS_ABLATE & 8)) {
#pragma unroll
                for (int i = 0; i < SW; ++i) {
                    dacc[0] = fma2(lo2(pv[i]), lo2(qv[i]), dacc[0]);
                    dacc[1] = fma2(hi2(pv[i]), hi2(qv[i]), dacc[1]);
                }
            }
            const f32x2 ds = dacc[0] + dacc[1];
            reinterpret_cast<float *>(dpart)[4 * k + w] = ds.x + ds.y;
        }
        met += 4;
        solvers_meet(meet, met, k);
        if (live) {
            // (B) e = M rhs: 16 columns each
            const f32x4 d4 = lds4(dpart + k);
            const float d = (d4.x + d4.y) + (d4.z + d4.w);
            const float rhs = k < n ? (base - (b - cdel_k * b)) - (d - adel_k * d) : 0.f;
            rbuf[k] = rhs;
            __builtin_amdgcn_wave_barrier();
            const float4 *rv = reinterpret_cast<const float4 *>(rbuf) + 4 * w;
            f32x2 eacc[2] = {dup2(0.f), dup2(0.f)};
            if (!(CU2REC_BS_ABLATE & 16)) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const f32x4 r4 = lds4(rv + t);
                    eacc[0] = fma2(lo2(mq[t]), lo2(r4), eacc[0]);
                    eacc[1] = fma2(hi2(mq[t]), hi2(r4), eacc[1]);
                }
            }
            const f32x2 es = eacc[0] + eacc[1];
            reinterpret_cast<float *>(epart)[4 * k + w] = es.x + es.y;
        }
        met += 4;
        solvers_meet(meet, met, k);
        if (live) {
            const f32x4 e4 = lds4(epart + k);
            const float e = (e4.x + e4.y) + (e4.z + e4.w);
            if (!(CU2REC_BS_ABLATE & 1) && w == 0) {  // the block's errors, four to a lane, for phase 3
                const float e0 = k < n ? e : 0.f;
                const f32x4 e4v = {e0, dpp_move<0x55>(e0), dpp_move<0xAA>(e0), dpp_move<0xFF>(e0)};  // quad lanes 0..3
                if ((k & 3) == 0) {
                    float *dst = it.ebuf + static_cast<size_t>(g0 + m) * kB + k;
                    *reinterpret_cast<f32x4 *>(dst) = e4v;
                }
            }
            // (C) the state the block leaves behind; its start state goes to phase 3
            wbuf[k] = k < n ? lr * apr * e : 0.f;
            const float bs = row_sum16(k < n ? lr * cpr * e : 0.f);
            b = (b - cdel_n * b) + ((lane_value(bs, 0) + lane_value(bs, 16)) + (lane_value(bs, 32) + lane_value(bs, 48)));
            __builtin_amdgcn_wave_barrier();
            float *qdst = it.qstart + static_cast<size_t>(g0 + m) * a.ldq;
            const float4 *wv = reinterpret_cast<const float4 *>(wbuf) + 2 * kg;
#pragma unroll
            for (int p = 0; p < NP; ++p) {  // lane (slot, group): sum over the group's 8 links of w_k P[k][slot]
                const int sq = 8 * p + sl, slot = w * SW + sq;
                const bool ok = sq < SW && slot < nslots;
                if (!(CU2REC_BS_ABLATE & 1) && ok && kg == 0) *reinterpret_cast<f32x4 *>(qdst + 4 * slot) = q4[p];
                const float4 *colp = tile + (8 * kg) * RS + min(slot, nslots - 1);
                f32x4 tb[8], wb[2];
#pragma unroll
                for (int i = 0; i < 8; ++i) tb[i] = lds4(colp + i * RS);
                wb[0] = lds4(wv), wb[1] = lds4(wv + 1);
                f32x2 ulo = dup2(0.f), uhi = dup2(0.f);
                if (!(CU2REC_BS_ABLATE & 2)) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const f32x2 ww = dup2(wb[i >> 2][i & 3]);
                        ulo = fma2(ww, lo2(tb[i]), ulo);
                        uhi = fma2(ww, hi2(tb[i]), uhi);
                    }
                }
                float u[4] = {ulo.x, ulo.y, uhi.x, uhi.y};
#pragma unroll
                for (int c = 0; c < 4; ++c) {  // the eight groups: lanes l ^ 8, l ^ 16, l ^ 32
                    u[c] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(u[c]), 0x128, 0xf, 0xf, false));  // row_ror:8
                    u[c] = half_sum(row_pair_sum(u[c]));
                }
                const f32x4 qo = q4[p];
                q4[p] = ok ? f32x4{(qo.x - adel_n * qo.x) + u[0], (qo.y - adel_n * qo.y) + u[1], (qo.z - adel_n * qo.z) + u[2],
                                   (qo.w - adel_n * qo.w) + u[3]}
                           : f32x4{0.f, 0.f, 0.f, 0.f};
                if (kg == 0 && sq < SW) *reinterpret_cast<f32x4 *>(qrow + slot) = q4[p];
            }
        }
        __syncthreads();
        if (m == nblk / 2) stamp.mark(it, 2);
        if (m == nblk - 1) stamp.mark(it, 3);
    }
    {
        if (kg == 0) {
            f32x4 *qdst = reinterpret_cast<f32x4 *>(a.Q + static_cast<size_t>(y) * a.ldq);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int sq = 8 * p + sl, slot = w * SW + sq;
                if (sq < SW && slot < nslots) qdst[slot] = q4[p];
            }
        }
        if (tid == 0) a.item_bias[y] = b;
    }
    stamp.done(it, 2, chain * 4 + w);
}

// ---- phase 2, the longest chains: the look-ahead form ---------------------------------------------------------------------------
// The plain form above is a ring of three dependent mat-vecs per block, (A) d = P q -> (B) e = M rhs -> (C) q <- a^n q + P^T w, split
// four ways with two meeting points: 1.4-1.6 us per block, and the hottest chain (31 blocks on the ML-20M shape) is more than half
// of an iteration.  With the cross block N_i = lr L[block i][block i - 1] from phase 1 (gram_cross_block) the errors obey
//     e_i = M_i (pre_i - N_i e_(i-1)),     pre_i[k] = base_i[k] - c^(k+64) b_(i-1) - a^(k+64) (p_(i,k) . q_(i-1))
// (tests/test_blocksolve_algebra.py, lookahead_chain): the only DEPENDENT work per block is two 64 x 64 mat-vecs, done by ONE
// wavefront with no partner to meet; the item row follows one block behind -- q_(i+1) from e_i, then (p . q_(i+1)) for block
// i + 2 -- on other wavefronts, with two blocks' time to do it in.  Eight wavefronts, fixed roles, NO workgroup barrier after the
// prologue; every hand-over is a counter in LDS that only its producer writes (LDS operations of a wavefront complete in order:
// whoever sees the count sees the data written before it):
//   wavefront 0   chain   e_i: record i (M_i, N_i, base) LDS -> registers, t0 = N_i e_(i-1), waits for g_i, e_i = M_i (pre_i - t0),
//                         publishes e_i and w_i = lr a^(n-1-k) e_i[k]; carries the item bias
//   wavefronts 1-2 state  q_(i+1) = a^n q_i + sum_k w_i[k] p_(i,k): every other pass of 8 slots each, lane = (slot, group of 8 links);
//                         q_i goes to phase 3 (qstart)
//   wavefront 3   dots    g_i[k] = p_(i,k) . q_(max(i-1, 0)): lane = link
//   wavefronts 4-7 loaders  block i (i = w mod 4): its 64 user rows memory -> registers -> P ring (three slots), then its record
//                         memory -> registers -> record ring (two slots).  One block at a time per wavefront, waiting with its data
//                         in registers for the slot: no load is in flight across a wait, so the compiler's s_waitcnt
//                         bookkeeping has nothing to get wrong; four wavefronts cover the latency.
// NSM: float4 slots per row the instantiation has registers for (nslots <= NSM <= 31: the rings fit the 160 KB of LDS).
constexpr int kLaPRing = 3, kLaRRing = 2, kLaDepth = 2;
constexpr int kLaNS4 = 17;                                  // float4 per row of the cross block in LDS (odd: conflict-free row reads)
constexpr int kLaRec4 = 3 * kH * kMS / 4 + kB * kLaNS4;     // a record in LDS: three factor tiles (base errors in column 32), the cross block
constexpr int kLaQ4 = 32;                                   // float4 per item-row buffer
constexpr int kLaMaxSlots = 29;
__host__ __device__ inline bool la_supported(int nslots) { return nslots >= 1 && nslots <= kLaMaxSlots; }
__host__ __device__ inline size_t la_lds_bytes(int nslots) {
    return (static_cast<size_t>(kLaPRing) * kB * (nslots | 1) + kLaRRing * kLaRec4 + 2 * kLaQ4 + 7 * kB / 4 + (kBsTableFloats + 3) / 4 + 4 + 64) * 16;
}

enum { kSyC0 = 0, kSyC1 = 1, kSyE = 2, kSyG = 3, kSyLdP = 4, kSyLdR = 8, kSyR = 12, kSyAbort = 13 };  // (C0 | C1, LdP[4], LdR[4]: read together)

// waits until *word >= target; bounded: gives up after 2 s (or as soon as another wait of the workgroup has), sets the status word
__device__ __forceinline__ void la_wait(unsigned *sy, int word, unsigned target, const BsIteration &it, unsigned long long *waited = nullptr) {
#if CU2REC_BS_TRACE
    const unsigned long long tw = wall_clock64();
#endif
    unsigned polls = 0;
    unsigned long long t0 = 0;
    while (__hip_atomic_load(sy + word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
        __builtin_amdgcn_s_sleep(1);
        if ((++polls & 1023u) == 0) {
            if (t0 == 0) t0 = wall_clock64();
            if (__hip_atomic_load(sy + kSyAbort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u || wall_clock64() - t0 > it.wait_ticks) {
                __hip_atomic_store(sy + kSyAbort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if ((threadIdx.x & 63) == 0) __hip_atomic_store(it.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    asm volatile("" ::: "memory");  // nothing behind the wait is read in front of it
#if CU2REC_BS_TRACE
    if (waited) *waited += wall_clock64() - tw;
#endif
}
__device__ __forceinline__ void la_post(unsigned *sy, int word, unsigned value) {
    asm volatile("" ::: "memory");  // everything written so far is issued in front of the count
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(sy + word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ... until all four counters sy[word .. word + 3] (16-byte aligned) have reached target: ONE LDS read per poll
__device__ __forceinline__ void la_wait4(unsigned *sy, int word, unsigned target, const BsIteration &it, unsigned long long *waited = nullptr) {
#if CU2REC_BS_TRACE
    const unsigned long long tw = wall_clock64();
#endif
    unsigned polls = 0;
    unsigned long long t0 = 0;
    for (;;) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        // (an LDS pointer by type: through a generic volatile pointer this was a FLAT load that waited for every global store in flight)
        const u32x4 v = __builtin_nontemporal_load((const __attribute__((address_space(3))) u32x4 *)(sy + word));
        asm volatile("" ::: "memory");
        if (min(min(v.x, v.y), min(v.z, v.w)) >= target) break;
        __builtin_amdgcn_s_sleep(1);
        if ((++polls & 1023u) == 0) {
            if (t0 == 0) t0 = wall_clock64();
            if (__hip_atomic_load(sy + kSyAbort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u || wall_clock64() - t0 > it.wait_ticks) {
                __hip_atomic_store(sy + kSyAbort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if ((threadIdx.x & 63) == 0) __hip_atomic_store(it.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    asm volatile("" ::: "memory");
#if CU2REC_BS_TRACE
    if (waited) *waited += wall_clock64() - tw;
#endif
}
// ... until both sy[word] and sy[word + 1] (8-byte aligned) have
__device__ __forceinline__ void la_wait2(unsigned *sy, int word, unsigned target, const BsIteration &it, unsigned long long *waited = nullptr) {
#if CU2REC_BS_TRACE
    const unsigned long long tw = wall_clock64();
#endif
    unsigned polls = 0;
    unsigned long long t0 = 0;
    for (;;) {
        typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
        const u32x2v v = __builtin_nontemporal_load((const __attribute__((address_space(3))) u32x2v *)(sy + word));
        asm volatile("" ::: "memory");
        if (min(v.x, v.y) >= target) break;
        __builtin_amdgcn_s_sleep(1);
        if ((++polls & 1023u) == 0) {
            if (t0 == 0) t0 = wall_clock64();
            if (__hip_atomic_load(sy + kSyAbort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u || wall_clock64() - t0 > it.wait_ticks) {
                __hip_atomic_store(sy + kSyAbort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if ((threadIdx.x & 63) == 0) __hip_atomic_store(it.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    asm volatile("" ::: "memory");
#if CU2REC_BS_TRACE
    if (waited) *waited += wall_clock64() - tw;
#endif
}

template <int NSM>
__device__ __forceinline__ void chain_lookahead(const SgdArgs &a, const BsIteration &it, float4 *smem, WaveStamp &stamp, int chain,
                                                const BsChainDesc &cd) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nslots = a.nslots, RS = nslots | 1;
    const int begin = cd.begin, len = cd.len, nblk = (len + kB - 1) / kB, g0 = cd.blk0, y = cd.item;
    float4 *pring = smem;                                        // [kLaPRing][kB][RS]
    float4 *rring = pring + kLaPRing * kB * RS;                  // [kLaRRing][kLaRec4]
    float4 *qbuf = rring + kLaRRing * kLaRec4;                   // [2][kLaQ4] the item row in front of block i: buffer i & 1, zero beyond the row
    float *wbuf = reinterpret_cast<float *>(qbuf + 2 * kLaQ4);   // [2][kB] w_i
    float *ebl = wbuf + 2 * kB;                                  // [2][kB] e_i
    float *tbuf = ebl + 2 * kB;                                  // [kB]    the chain wavefront's own right-hand side
    float *gbuf = tbuf + kB;                                     // [2][kB] g_i
    float *tab = gbuf + 2 * kB;                                  // [kBsTableFloats]
    unsigned *sy = reinterpret_cast<unsigned *>(tab + ((kBsTableFloats + 3) & ~3));  // [16] the counters
    f32x4 *sink = reinterpret_cast<f32x4 *>(sy + 16);                               // [64] where the loaders' surplus lanes store
    const float lr = a.h.lr;

    if (wave >= 4) {
        // ---------------------------------------------------------------------------------------------------------- loaders
        // wavefront 4 + w: quarter w of every block's row image (float4 numbers 64 (w NQ + j) + lane, j < NQ) and its share of the
        // block's record (w 0, 1: the factor tiles, w 0 the base errors too; w 2, 3: the cross block), kLaDepth blocks in flight in
        // registers.  Every load and store is unconditional (clamped addresses, a sink in LDS for the surplus lanes): a branch
        // around a load makes the compiler wait for everything in flight.
        const int w = wave - 4;
        __builtin_amdgcn_s_setprio(2);
        unsigned long long wt[2] = {0, 0};
        constexpr int NQ = (NSM + 3) / 4;
        const int idx0 = w * kB * NQ + lane;
        const int row0 = idx0 / nslots, slot0 = idx0 - row0 * nslots;
        const int drow = kB / nslots, dslot = kB - drow * nslots;
        auto load_val = [&](int i) -> uint64_t { return it.vals[begin + min(kB * i + lane, len - 1)]; };
        struct LaStage {
            f32x4 v[NQ], r[8];
            uint64_t next_val;
        };
        auto issue = [&](LaStage &st, int i, uint64_t val) {  // block i (clamped: past the chain's end the last block again, never committed)
            const int ic = min(i, nblk - 1);
            st.next_val = load_val(min(i + kLaDepth, nblk - 1));
            const int x = static_cast<int>(val >> 32);
            int row = row0, slot = slot0;
#pragma unroll
            for (int j = 0; j < NQ; ++j) {  // past the block's image (float4 number >= 64 nslots): the last row again
                const int xr = __shfl(x, min(row, kB - 1));
                st.v[j] = reinterpret_cast<const f32x4 *>(a.P + static_cast<size_t>(xr) * a.ldp)[slot];
                slot += dslot, row += drow;
                if (slot >= nslots) slot -= nslots, ++row;
            }
            // the record: 768 float4 of factor tiles + 16 of base errors (Mbuf), 1,024 of the cross block (Nbuf; block 0 has none:
            // block 1's, never used)
            const f32x4 *src = w < 2 ? reinterpret_cast<const f32x4 *>(it.Mbuf + static_cast<size_t>(g0 + ic) * kBsRecFloats) + 384 * w
                                     : reinterpret_cast<const f32x4 *>(it.Nbuf + static_cast<size_t>(g0 + max(ic, 1)) * kBsCrossFloats) + 512 * (w - 2);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int at = w < 2 && j >= 6 ? 768 - 384 * w + (lane & 15) : lane + 64 * j;  // (w < 2: six tiles' pieces, then the base errors)
                st.r[j] = src[at];
            }
        };
        auto commit = [&](const LaStage &st, int i) {
            if (i >= kLaPRing) {  // the slot's last reader: the state wavefronts, block i - 3
                la_wait2(sy, kSyC0, static_cast<unsigned>(i - kLaPRing + 1), it, &wt[0]);
            }
            {
                f32x4 *tile = reinterpret_cast<f32x4 *>(pring + (i % kLaPRing) * kB * RS);
                int row = row0, slot = slot0;
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < NQ; ++j) {  // past the image: this lane's float4 of the sink
                    f32x4 *dst = row < kB ? tile + row * RS + slot : sink + lane;
                    *dst = kB * i + row < len ? st.v[j] : z;
                    slot += dslot, row += drow;
                    if (slot >= nslots) slot -= nslots, ++row;
                }
            }
            la_post(sy, kSyLdP + w, static_cast<unsigned>(i + 1));
            if (i >= kLaRRing) la_wait(sy, kSyR, static_cast<unsigned>(i - kLaRRing + 1), it, &wt[1]);  // the chain wavefront has taken record i - 2
            f32x4 *rr = reinterpret_cast<f32x4 *>(rring + (i % kLaRRing) * kLaRec4);
            if (w < 2) {
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const int idx = 384 * w + lane + 64 * j;  // tile idx >> 8, row (idx >> 3) & 31, float4 idx & 7
                    rr[((idx >> 8) * kH + ((idx >> 3) & 31)) * (kMS / 4) + (idx & 7)] = st.r[j];
                }
                if (w == 0 && lane < 16) {  // float4 number 768 + lane: the base errors of links 4 lane .. 4 lane + 3, column 32 of their rows
                    float *col = reinterpret_cast<float *>(rr) + (4 * lane) * kMS + kH;
#pragma unroll
                    for (int e = 0; e < 4; ++e) col[e * kMS] = st.r[6][e];
                }
            } else {
                f32x4 *nl = rr + 3 * kH * kMS / 4;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int idx = 512 * (w - 2) + lane + 64 * j;
                    nl[(idx >> 4) * kLaNS4 + (idx & 15)] = st.r[j];
                }
            }
            la_post(sy, kSyLdR + w, static_cast<unsigned>(i + 1));
        };
        static_assert(kLaDepth == 2, "two stages: four wavefronts x two blocks x 15 KB in flight");
        LaStage st[kLaDepth];
        {
            uint64_t v0[kLaDepth];
#pragma unroll
            for (int t = 0; t < kLaDepth; ++t) v0[t] = load_val(min(t, nblk - 1));
#pragma unroll
            for (int t = 0; t < kLaDepth; ++t) issue(st[t], t, v0[t]);
        }
        __syncthreads();  // the prologue's one barrier: counters and tables are set
        for (int m0 = 0; m0 < nblk; m0 += kLaDepth) {
#pragma unroll
            for (int u = 0; u < kLaDepth; ++u) {
                const int i = m0 + u;
                if (i < nblk) commit(st[u], i);  // wavefront uniform
                issue(st[u], i + kLaDepth, st[u].next_val);
            }
        }
        stamp.span(0, wt[0]);
        stamp.span(1, wt[1]);
        stamp.span(2, 1);
        stamp.span(3, 1);
        stamp.done(it, 3, chain * 4 + w);
        return;
    }

    if (wave == 3) {
        // ------------------------------------------------------------------------------------------------------------- dots
        __syncthreads();
        __builtin_amdgcn_s_setprio(2);
        unsigned long long wt[2] = {0, 0};
        for (int i = 0; i < nblk; ++i) {
            const int src = max(i - 1, 0);
            la_wait4(sy, kSyLdP, static_cast<unsigned>(i + 1), it, &wt[0]);
            if (src > 0) la_wait2(sy, kSyC0, static_cast<unsigned>(src), it, &wt[1]);
            const float4 *prow = pring + (i % kLaPRing) * kB * RS + lane * RS;
            const float4 *qv = qbuf + (src & 1) * kLaQ4;
            // two halves, every read of a half in flight before its first product: one LDS latency per half, not one per slot
            constexpr int NH2 = (NSM + 1) / 2;
            f32x2 acc[4] = {dup2(0.f), dup2(0.f), dup2(0.f), dup2(0.f)};
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                f32x4 pv[NH2], qq[NH2];
#pragma unroll
                for (int j = 0; j < NH2; ++j) {
                    pv[j] = lds4(prow + min(hh * NH2 + j, nslots - 1));
                    qq[j] = lds4(qv + min(hh * NH2 + j, kLaQ4 - 1));  // (the item row's buffer is zero beyond the row)
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < NH2; ++j) {
                    if (hh * NH2 + j < NSM) {
                        acc[2 * (j & 1)] = fma2(lo2(pv[j]), lo2(qq[j]), acc[2 * (j & 1)]);
                        acc[2 * (j & 1) + 1] = fma2(hi2(pv[j]), hi2(qq[j]), acc[2 * (j & 1) + 1]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            acc[0] += acc[2];
            acc[1] += acc[3];
            const f32x2 ds = acc[0] + acc[1];
            gbuf[(i & 1) * kB + lane] = ds.x + ds.y;
            la_post(sy, kSyG, static_cast<unsigned>(i + 1));
        }
        stamp.span(0, wt[0]);
        stamp.span(1, wt[1]);
        stamp.span(2, 1);
        stamp.span(3, 1);
        stamp.done(it, 2, chain * 4 + 3);
        return;
    }

    if (wave >= 1) {
        // ------------------------------------------------------------------------------------------------------------ state
        const int c = wave - 1;
        constexpr int NP = (NSM + 7) / 8, NPW = (NP + 1) / 2;  // passes of 8 slots; this wavefront: passes c, c + 2, ...
        const int sl = lane & 7, kg = lane >> 3;
        f32x4 q4[NPW];
        const f32x4 *qsrc = reinterpret_cast<const f32x4 *>(a.Q + static_cast<size_t>(y) * a.ldq);
#pragma unroll
        for (int pp = 0; pp < NPW; ++pp) {
            const int slot = 8 * (2 * pp + c) + sl;
            const f32x4 v = qsrc[min(slot, nslots - 1)];
            q4[pp] = slot < nslots ? v : f32x4{0.f, 0.f, 0.f, 0.f};
            if (kg < 2 && slot < kLaQ4) *reinterpret_cast<f32x4 *>(qbuf + kg * kLaQ4 + slot) = kg == 0 ? q4[pp] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
        __builtin_amdgcn_s_setprio(2);
        unsigned long long wt[1] = {0};
        for (int i = 0; i < nblk; ++i) {
            const int n = min(kB, len - kB * i);
            const float adel_n = tab[kTabAdel + n];
            la_wait(sy, kSyE, static_cast<unsigned>(i + 1), it, &wt[0]);
            const float4 *tile = pring + (i % kLaPRing) * kB * RS;
            const float4 *wv = reinterpret_cast<const float4 *>(wbuf + (i & 1) * kB) + 2 * kg;
            float *qdst = it.qstart + static_cast<size_t>(g0 + i) * a.ldq;
            const f32x4 wb0 = lds4(wv), wb1 = lds4(wv + 1);
            f32x4 tb[NPW][8];  // every read of the block in flight before the first product
#pragma unroll
            for (int pp = 0; pp < NPW; ++pp) {
                const float4 *colp = tile + (8 * kg) * RS + min(8 * (2 * pp + c) + sl, nslots - 1);
#pragma unroll
                for (int r = 0; r < 8; ++r) tb[pp][r] = lds4(colp + r * RS);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pp = 0; pp < NPW; ++pp) {
                const int slot = 8 * (2 * pp + c) + sl;
                const bool ok = slot < nslots;
                // the row in front of block i: phase 3's
                if (ok && kg == 0) *reinterpret_cast<f32x4 *>(qdst + 4 * slot) = q4[pp];
                f32x2 ulo = dup2(0.f), uhi = dup2(0.f);
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const f32x2 ww = dup2(r < 4 ? wb0[r & 3] : wb1[r & 3]);
                    ulo = fma2(ww, lo2(tb[pp][r]), ulo);
                    uhi = fma2(ww, hi2(tb[pp][r]), uhi);
                }
                float u[4] = {ulo.x, ulo.y, uhi.x, uhi.y};
#pragma unroll
                for (int e = 0; e < 4; ++e) {  // the eight groups: lanes l ^ 8, l ^ 16, l ^ 32
                    u[e] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(u[e]), 0x128, 0xf, 0xf, false));  // row_ror:8
                    u[e] = half_sum(row_pair_sum(u[e]));
                }
                const f32x4 qo = q4[pp];
                q4[pp] = ok ? f32x4{(qo.x - adel_n * qo.x) + u[0], (qo.y - adel_n * qo.y) + u[1], (qo.z - adel_n * qo.z) + u[2],
                                    (qo.w - adel_n * qo.w) + u[3]}
                            : f32x4{0.f, 0.f, 0.f, 0.f};
                if (ok && kg == 0) *reinterpret_cast<f32x4 *>(qbuf + ((i + 1) & 1) * kLaQ4 + slot) = q4[pp];
            }
            la_post(sy, kSyC0 + c, static_cast<unsigned>(i + 1));
        }
        if (kg == 0) {
            f32x4 *qd = reinterpret_cast<f32x4 *>(a.Q + static_cast<size_t>(y) * a.ldq);
#pragma unroll
            for (int pp = 0; pp < NPW; ++pp) {
                const int slot = 8 * (2 * pp + c) + sl;
                if (slot < nslots) qd[slot] = q4[pp];
            }
        }
        stamp.span(0, wt[0]);
        stamp.span(1, 1);
        stamp.span(2, 1);
        stamp.span(3, 1);
        stamp.done(it, 2, chain * 4 + wave);
        return;
    }

    // ------------------------------------------------------------------------------------------------------------------ chain
    const int k = lane;
    for (int i = k; i < kBsTableFloats; i += 64) tab[i] = it.tables[i];
    if (k < 16) sy[k] = 0u;
    float b_cur = a.item_bias[y], b_prev = b_cur;
    asm volatile("" ::"v"(b_cur));
    __syncthreads();
    __builtin_amdgcn_s_setprio(3);
    stamp.mark(it, 0);
    const int kr = k & 31;
    const bool upper = k >= kH;
    unsigned long long wt[2] = {0, 0};
    // record i in registers one block ahead: this lane's row of M (links 0-31: [M11 row | 0], links 32-63: [M21 row | M22 row]), its
    // row of the cross block, its base error -- read while the other wavefronts work on the item row, not on the dependent path
    f32x4 m4[16], n4[16];
    float base;
    auto take_record = [&](int i) {
        la_wait4(sy, kSyLdR, static_cast<unsigned>(i + 1), it, &wt[0]);
        const float4 *rr = rring + (i % kLaRRing) * kLaRec4;
        const float *Mt = reinterpret_cast<const float *>(rr);
        const float4 *mlo = reinterpret_cast<const float4 *>(Mt + ((upper ? kH : 0) + kr) * kMS);
        const float4 *mhi = reinterpret_cast<const float4 *>(Mt + (2 * kH + kr) * kMS);
        const float4 *nrow = rr + 3 * kH * kMS / 4 + k * kLaNS4;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            m4[j] = lds4(mlo + j);
            const f32x4 hi = lds4(mhi + j);
            m4[8 + j] = upper ? hi : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) n4[j] = lds4(nrow + j);
        base = Mt[k * kMS + kH];
        la_post(sy, kSyR, static_cast<unsigned>(i + 1));  // (the slot's reads are issued: LDS serves a wavefront in order)
    };
    take_record(0);
    for (int i = 0; i < nblk; ++i) {
        const int n = min(kB, len - kB * i);
        // t0 = N_i e_(i-1): this lane's row of the cross block against the errors of the block before (all 16 broadcast reads in
        // flight before the first product)
        f32x2 tacc[2] = {dup2(0.f), dup2(0.f)};
        if (i > 0) {
            const float4 *ev = reinterpret_cast<const float4 *>(ebl + ((i - 1) & 1) * kB);
            f32x4 e4[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) e4[j] = lds4(ev + j);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                tacc[0] = fma2(lo2(n4[j]), lo2(e4[j]), tacc[0]);
                tacc[1] = fma2(hi2(n4[j]), hi2(e4[j]), tacc[1]);
            }
        }
        const f32x2 ts = tacc[0] + tacc[1];
        const float t0 = ts.x + ts.y;
        // pre_i from the state in front of block max(i - 1, 0)
        const int kd = i > 0 ? k + kB : k;
        const float adel_k = tab[kTabAdel + kd], cdel_k = tab[kTabCdel + kd];
        const float bsrc = i > 0 ? b_prev : b_cur;
        la_wait(sy, kSyG, static_cast<unsigned>(i + 1), it, &wt[1]);
        const float gd = gbuf[(i & 1) * kB + k];
        const float pre = k < n ? (base - (bsrc - cdel_k * bsrc)) - (gd - adel_k * gd) : 0.f;
        tbuf[k] = pre - t0;
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        f32x2 eacc[2] = {dup2(0.f), dup2(0.f)};
        {
            const float4 *tv = reinterpret_cast<const float4 *>(tbuf);
            f32x4 t4[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) t4[j] = lds4(tv + j);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                eacc[0] = fma2(lo2(m4[j]), lo2(t4[j]), eacc[0]);
                eacc[1] = fma2(hi2(m4[j]), hi2(t4[j]), eacc[1]);
            }
        }
        const f32x2 es = eacc[0] + eacc[1];
        const float e = k < n ? es.x + es.y : 0.f;
        const int back = max(n - 1 - k, 0);
        ebl[(i & 1) * kB + k] = e;
        wbuf[(i & 1) * kB + k] = k < n ? lr * tab[kTabApow + back] * e : 0.f;
        la_post(sy, kSyE, static_cast<unsigned>(i + 1));
        it.ebuf[static_cast<size_t>(g0 + i) * kB + k] = e;
        if (i + 1 < nblk) take_record(i + 1);  // wavefront uniform
        // the item bias behind block i
        const float bs = row_sum16(k < n ? lr * tab[kTabCpow + back] * e : 0.f);
        const float cdel_n = tab[kTabCdel + n];
        const float b_next = (b_cur - cdel_n * b_cur) + ((lane_value(bs, 0) + lane_value(bs, 16)) + (lane_value(bs, 32) + lane_value(bs, 48)));
        b_prev = b_cur;
        b_cur = b_next;
    }
    stamp.span(1, wt[0]);
    stamp.span(2, wt[1]);
    stamp.mark(it, 3);
    if (k == 0) a.item_bias[y] = b_cur;
    stamp.done(it, 2, chain * 4);
}

// The join with the side stream: one thread of an extra workgroup in the LAST launch of the iteration on the main stream (phase 3).
// It ends when the signal kernel queued behind this iteration's side kernel has run -- i.e. that kernel is complete, its rows in
// memory -- so that the next iteration's phase 1, queued behind this launch, needs no event (a wait on one costs the stream 2.5 us
// per iteration).  (The side kernel's workgroups counting THEMSELVES through, each behind an agent-scope release of its rows, was
// tried in round 4: 9350 L2 write-backs per launch, the side kernel took 292 us instead of 46.)
// This wait guards DATA -- the next phase 1 reads rows the side kernel and phase 3 write -- so it gives up only when the GPU must be
// taken for wedged: 15 x the bound of the other waits, 30 s by default; then the status word is set and the call reports it.
__device__ __forceinline__ void await_iteration_end(const BsIteration &it) {
    const unsigned long long t0 = wall_clock64();
    unsigned polls = 0;
    while (it.side_seq && ld_agent(it.side_seq) < it.side_target) {
        __builtin_amdgcn_s_sleep(8);
        if ((++polls & 63u) == 0 && (ld_agent(it.status) != 0u || wall_clock64() - t0 > 15ull * it.wait_ticks)) {
            __hip_atomic_store(it.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
}

// One workgroup per hot chain: the look-ahead form for the leading chains (rows of at most kLaMaxSlots slots), the plain form for the rest.
template <int SW>
__global__ __launch_bounds__(512) void bs_solve_kernel(SgdArgs a, BsIteration it) {
    extern __shared__ float4 bs_smem[];
    WaveStamp stamp(it);
    if (it.gram_done && threadIdx.x == 0)  // (the side stream's gate lets the side kernel go once the chains hold their CUs)
        __hip_atomic_fetch_add(it.solve_started, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int chain = blockIdx.x;
    const BsChainDesc cd = it.chains[chain];
    if constexpr (SW <= 8) {
        if (chain_is_lookahead(it, chain, cd)) {  // workgroup uniform
            chain_lookahead<(4 * SW < kLaMaxSlots ? 4 * SW : kLaMaxSlots)>(a, it, bs_smem, stamp, chain, cd);
            return;
        }
    }
    solve_chain<SW>(a, it, bs_smem, stamp, chain, cd);
}

// ---- phase 3: the user side of every hot block ---------------------------------------------------------------------
// One wavefront per (block, 64 columns), lane = column.  The wavefront holds its 64 x 64 piece of the block's user rows in registers
// (one 256-byte row segment per load instruction, user id and error of link k broadcast from lane k) and walks down the links with
// the reference's own recurrence (mf_sequential.cu:133-134), the item row as link k saw it carried in one register:
//     p_k <- p_k + lr (e_k q - P_reg p_k),   q <- q + lr (e_k p_k - Q_reg q)        (p_k on the right: the old value)
// 64 dependent steps of a few VALU instructions (~0.4 us), no LDS, no staging.  (Rounds 2-3 built the rows "as link k saw them"
// as a triangular 64 x 64 product on the MFMA pipe: 48 fp32 MFMA instructions and ~200 LDS reads per 32 columns, 3 us of a wavefront's
// 9 -- and the kernel's duration IS one wavefront's latency, every block's workgroup being resident at once.)
constexpr int kUpdMaxWaves = (4 * kBsMaxSlots + 63) / 64;

__device__ __forceinline__ void update_block(const SgdArgs &a, const BsIteration &it, int g, int nwaves, WaveStamp &stamp) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const BsBlockDesc bd = it.blocks[g];
    if (bd.n_valid == 0) return;  // workgroup uniform
    const bool mine = lane < bd.n_valid;  // lane l: link l
    // links past the end of a short block: the last link's row (finite, loaded, never stored) with e = 0
    const uint64_t val = it.vals[bd.pos0 + min(lane, bd.n_valid - 1)];
    const int x = static_cast<int>(val >> 32);
    const float lr = a.h.lr;
    const int ncols = 4 * a.nslots;
    const int col = 64 * wave + lane, colc = min(col, ncols - 1);
    const float qs = it.qstart[static_cast<size_t>(g) * a.ldq + colc];
    const float e_all = it.ebuf[static_cast<size_t>(g) * kB + lane];
    const float e = mine ? e_all : 0.f;
    stamp.fine(it, 0);  // user ids
    float p[kB];
#pragma unroll
    for (int k = 0; k < kB; ++k) {
        const int xk = __builtin_amdgcn_readlane(x, k);
        p[k] = a.P[static_cast<size_t>(xk) * a.ldp + colc];
    }
    float q = qs;
    const float p_reg = a.h.p_reg, q_reg = a.h.q_reg;
    stamp.fine(it, 1);  // loads issued; errors, start row
#pragma unroll
    for (int k = 0; k < kB; ++k) {
        const float ek = lane_value(e, k);
        const float pk = p[k];
        p[k] = pk + lr * (ek * q - p_reg * pk);  // mf_sequential.cu:133-134
        q = q + lr * (ek * pk - q_reg * q);
    }
    // (the stores in a sequence of their own behind the walk: interleaved with it -- what the compiler makes of the two unrolled loops --
    // every step's wait for its row also waited for stores issued before it, 5 us per wavefront)
#pragma unroll
    for (int k = 0; k < kB; ++k) asm volatile("" : "+v"(p[k]));  // (the values exist HERE: no sinking of the walk into the stores)
    __builtin_amdgcn_sched_barrier(0);
    stamp.fine(it, 2);  // new rows
    if (col < ncols) {
#pragma unroll
        for (int k = 0; k < kB; ++k) {
            if (k < bd.n_valid) a.P[static_cast<size_t>(__builtin_amdgcn_readlane(x, k)) * a.ldp + col] = p[k];  // wavefront uniform
        }
    }
    if (wave == 0 && mine) {
        const float ub = a.user_bias[x];
        a.user_bias[x] = ub + lr * (e - a.h.ub_reg * ub);  // mf_sequential.cu:140
    }
    stamp.done(it, 4, g * nwaves + wave);
}

// One workgroup per block.
__global__ __launch_bounds__(64 * kUpdMaxWaves) void bs_update_kernel(SgdArgs a, BsIteration it, int nwaves) {
    const int g = blockIdx.x;
    WaveStamp stamp(it);
    __builtin_amdgcn_s_setprio(2);
    if (g == it.launch_blocks) {  // the extra workgroup of the launch (side_seq set)
        if (threadIdx.x == 0) await_iteration_end(it);
        return;
    }
    // (launch_blocks workgroups stride through the dense block table, like phase 1's; update_block keeps nothing in LDS)
    for (int blk = g; blk < it.max_blocks; blk += it.launch_blocks) {
        if (it.blocks[blk].n_valid == 0) break;  // workgroup uniform
        update_block(a, it, blk, nwaves, stamp);
    }
}

template <int SW>
void launch_solve(const SgdArgs &a, const BsIteration &it, hipStream_t stream) {
    size_t lds = solve_lds_bytes(a.nslots, SW);
    if (it.la_ranks > 0) lds = std::max(lds, la_lds_bytes(a.nslots));
    ensure_max_dynamic_lds(reinterpret_cast<const void *>(bs_solve_kernel<SW>));
    hipLaunchKernelGGL((bs_solve_kernel<SW>), dim3(it.n_hot), dim3(512), lds, stream, a, it);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess)
        fail(CU2REC_EHIP, std::string("bs_solve_kernel launch failed: ") + hipGetErrorString(err) + " (SW " + std::to_string(SW) +
                              ", chains " + std::to_string(it.n_hot) + ", LDS " + std::to_string(lds) + " bytes)");
}

}  // namespace

namespace {
unsigned long long *g_stamps = nullptr;
int g_stamps_cap = 0;

// One status word per device (set by a device-side wait that gave up) and its pinned host copy, refreshed after every
// block-solve call; like the resident launches' (resident.hip).
struct BsFault {
    unsigned *dev_word = nullptr;
    unsigned *host_word = nullptr;
};
std::mutex g_fault_mutex;
std::vector<BsFault> g_faults;

BsFault &fault_for_current_device() {  // caller holds g_fault_mutex
    int dev = 0;
    CU2REC_HIP(hipGetDevice(&dev));
    if (static_cast<int>(g_faults.size()) <= dev) g_faults.resize(dev + 1);
    BsFault &f = g_faults[dev];
    if (!f.dev_word) {
        CU2REC_HIP(hipMalloc(reinterpret_cast<void **>(&f.dev_word), 128));
        CU2REC_HIP(hipMemset(f.dev_word, 0, 128));
        CU2REC_HIP(hipHostMalloc(reinterpret_cast<void **>(&f.host_word), sizeof(unsigned), hipHostMallocDefault));
        *f.host_word = 0;
    }
    return f;
}
}  // namespace

// ---- the launch topology of an iteration's fork / join, per device: decided on first use, revisable ------------------------------
// kBsTopoDevice: gate kernel + device-side join (cross-stream spin waits: they need kernels of the two streams RUNNING SIDE BY SIDE);
// kBsTopoEvents: events both ways (works wherever HIP works; 5.4 + 2.5-3.3 us per iteration on the main stream).  What decides:
//   1. CU2REC_BS_GATE in the environment (0: events, anything else: device) -- a force, never revised;
//   2. a counter pass of rocprofv3 (--pmc exports ROCPROF_COUNTER_COLLECTION: kernels are serialised across streams): events;
//   3. otherwise a two-stream handshake PROBE on the very streams the iterations will use (bs_probe_streams below): a wait kernel queued
//      FIRST on one stream, satisfied by a signal kernel queued behind it on the other, bounded by 10 ms -- both directions.  Streams
//      that share a hardware queue (GPU_MAX_HW_QUEUES=1), a tool that serialises dispatches, a tenant holding the CUs: the waiter times
//      out, and the process uses events for good;
//   4. a join that gives up mid-run (bs_check_fault) switches the device to events for every later call, and says so in its error.
namespace {
struct BsTopology {
    int mode = -1;  // -1: not decided yet
    bool forced = false;
    std::string why = "not decided yet";
};
std::mutex g_topo_mutex;
std::vector<BsTopology> g_topo;  // per device

__global__ void bs_probe_wait_kernel(const unsigned *word, unsigned *result, unsigned long long ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    unsigned seen = 0;
    for (;;) {
        if (ld_agent(word) != 0u) {
            seen = 1;
            break;
        }
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > ticks) break;
    }
    __hip_atomic_store(result, seen ? 1u : 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void bs_probe_signal_kernel(unsigned *word) {
    if (threadIdx.x == 0) __hip_atomic_store(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Do kernels of `waiter` and `signaller` run side by side?  The waiter is queued FIRST: if the two streams are served one after the
// other it can only time out (10 ms), and the signal kernel runs behind it.  Every wave reaches its exit: the wait is bounded.
bool bs_probe_streams(hipStream_t waiter, hipStream_t signaller, unsigned *scratch) {
    CU2REC_HIP(hipMemsetAsync(scratch, 0, 64 * sizeof(unsigned), waiter));
    CU2REC_HIP(hipStreamSynchronize(waiter));
    CU2REC_HIP(hipStreamSynchronize(signaller));
    hipLaunchKernelGGL(bs_probe_wait_kernel, dim3(1), dim3(64), 0, waiter, scratch, scratch + 32, 1000000ull);  // 10 ms of the 100 MHz clock
    hipLaunchKernelGGL(bs_probe_signal_kernel, dim3(1), dim3(64), 0, signaller, scratch);
    CU2REC_HIP(hipGetLastError());
    CU2REC_HIP(hipStreamSynchronize(waiter));
    CU2REC_HIP(hipStreamSynchronize(signaller));
    unsigned result = 0;
    CU2REC_HIP(hipMemcpy(&result, scratch + 32, sizeof(unsigned), hipMemcpyDeviceToHost));
    return result == 1u;
}

BsTopology &topology_for_current_device() {  // caller holds g_topo_mutex
    int dev = 0;
    CU2REC_HIP(hipGetDevice(&dev));
    if (static_cast<int>(g_topo.size()) <= dev) g_topo.resize(dev + 1);
    return g_topo[dev];
}
}  // namespace

int bs_topology(hipStream_t stream, hipStream_t side) {
    std::lock_guard<std::mutex> lock(g_topo_mutex);
    BsTopology &t = topology_for_current_device();
    if (t.mode >= 0) return t.mode;
    if (const char *env = std::getenv("CU2REC_BS_GATE")) {
        t.mode = std::atoi(env) != 0 ? kBsTopoDevice : kBsTopoEvents;
        t.forced = true;
        t.why = std::string("CU2REC_BS_GATE=") + env;
        return t.mode;
    }
    if (const char *pmc = std::getenv("ROCPROF_COUNTER_COLLECTION"))
        if (*pmc && std::string(pmc) != "0" && std::string(pmc) != "False" && std::string(pmc) != "false") {
            t.mode = kBsTopoEvents;
            t.why = "a rocprofv3 counter pass (ROCPROF_COUNTER_COLLECTION) serialises kernels across streams";
            return t.mode;
        }
    unsigned *scratch = nullptr;
    CU2REC_HIP(hipMalloc(reinterpret_cast<void **>(&scratch), 64 * sizeof(unsigned)));
    bool forward = false, backward = false;
    try {
        forward = bs_probe_streams(stream, side, scratch);
        backward = forward && bs_probe_streams(side, stream, scratch);
    } catch (...) {
        (void)hipFree(scratch);
        throw;
    }
    (void)hipFree(scratch);
    if (forward && backward) {
        t.mode = kBsTopoDevice;
        t.why = "two-stream handshake probe passed in both directions";
    } else {
        t.mode = kBsTopoEvents;
        t.why = std::string("two-stream handshake probe: a wait kernel on the ") + (forward ? "side" : "main") +
                " stream was not reached by the other stream's signal kernel within 10 ms (the streams do not run side by side here)";
    }
    return t.mode;
}

int bs_topology_query(std::string *why) {
    std::lock_guard<std::mutex> lock(g_topo_mutex);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    if (dev >= static_cast<int>(g_topo.size())) return -1;
    if (why) *why = g_topo[dev].why;
    return g_topo[dev].mode;
}

void bs_set_stamps(unsigned long long *buf, int cap) {
    g_stamps = buf;
    g_stamps_cap = buf ? cap : 0;
}

void bs_get_stamps(unsigned long long **buf, int *cap) {
    *buf = g_stamps;
    *cap = g_stamps_cap;
}

unsigned *bs_status_word() {
    std::lock_guard<std::mutex> lock(g_fault_mutex);
    return fault_for_current_device().dev_word;
}

void bs_report_status(hipStream_t stream) {
    std::lock_guard<std::mutex> lock(g_fault_mutex);
    BsFault &f = fault_for_current_device();
    CU2REC_HIP(hipMemcpyAsync(f.host_word, f.dev_word, sizeof(unsigned), hipMemcpyDeviceToHost, stream));
}

void bs_check_fault() {
    std::lock_guard<std::mutex> lock(g_fault_mutex);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    if (dev >= static_cast<int>(g_faults.size()) || !g_faults[dev].host_word) return;
    BsFault &f = g_faults[dev];
    if (*static_cast<volatile unsigned *>(f.host_word) != 0) {
        (void)hipDeviceSynchronize();  // reported once: cleared so that the process can go on
        (void)hipMemset(f.dev_word, 0, sizeof(unsigned));
        *f.host_word = 0;
        // self-healing: whatever kept the launches from running side by side will do so again -- every later block-solve call of
        // this process on this device forks and joins with events (unless the environment forces the device form)
        bool switched = false;
        {
            std::lock_guard<std::mutex> topo_lock(g_topo_mutex);
            if (static_cast<int>(g_topo.size()) <= dev) g_topo.resize(dev + 1);
            if (!g_topo[dev].forced && g_topo[dev].mode != kBsTopoEvents) {
                g_topo[dev].mode = kBsTopoEvents;
                g_topo[dev].why = "a device-side join gave up mid-run: switched to the event fork / join";
                switched = true;
            }
        }
        fail(CU2REC_EHIP,
             std::string("cu2rec_amd: a block-solve iteration gave up waiting for one of its own launches (the three phases of an iteration "
                         "run side by side and hand blocks over through flags: is another process or stream holding this GPU's compute "
                         "units?); the model state is undefined.  ") +
                 (switched ? "From here on this process forks and joins the iteration's streams with events instead of device-side waits "
                             "(slower by a few microseconds per iteration, independent of concurrent execution): re-create the model and "
                             "run again.  "
                           : "(CU2REC_BS_GATE forces the device-side form: unset it to let the library fall back to events.)  ") +
                 "CU2REC_SGD_ORDERED is the same result without concurrent launches");
    }
}

unsigned long long bs_wait_ticks() {  // CU2REC_BS_WAIT_S (seconds, default 2): the bound of the device-side waits, in ticks of the 100 MHz clock
    static const unsigned long long ticks = [] {
        double sec = 2.0;
        if (const char *env = std::getenv("CU2REC_BS_WAIT_S")) sec = std::max(0.001, std::atof(env));
        return static_cast<unsigned long long>(sec * 1e8);
    }();
    return ticks;
}

bool bs_supported(int nslots) { return nslots >= 1 && nslots <= kBsMaxSlots; }
bool bs_lookahead_supported(int nslots) { return la_supported(nslots); }

int bs_compute_units() {
    static std::mutex mutex;
    static std::vector<int> cus;  // per device
    int dev = 0;
    CU2REC_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mutex);
    if (static_cast<int>(cus.size()) <= dev) cus.resize(dev + 1, 0);
    if (!cus[dev]) CU2REC_HIP(hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, dev));
    return cus[dev];
}

void bs_launch_tables(const SgdHyper &h, float *tables, hipStream_t stream) {
    hipLaunchKernelGGL(bs_tables_kernel, dim3(1), dim3(192), 0, stream, h, tables);
}

void bs_launch_plan(const uint32_t *keys, int n_active, int n_batch, int n_hot, int item_bits, int max_blocks,
                    const int *item_of_rank, int *chain_begin, BsChainDesc *chains, BsBlockDesc *blocks, int *walk_begin,
                    hipStream_t stream, size_t stride) {
    hipLaunchKernelGGL(bs_plan_kernel, dim3(n_batch), dim3(256), 0, stream, keys, n_active, n_hot, item_bits, max_blocks,
                       item_of_rank, chain_begin, chains, blocks, walk_begin, stride);
}

void bs_launch_gram(const SgdArgs &a, const BsIteration &it, hipStream_t stream, hipEvent_t stop) {
    if (it.n_hot <= 0 || it.max_blocks <= 0) return;
    size_t gram_lds = std::max(static_cast<size_t>(kB) * (a.nslots | 1) * 16, static_cast<size_t>(5) * kH * kMS * 4) +
                      static_cast<size_t>(kGramTabFloats + kB) * 4;
    if (it.la_ranks > 0) gram_lds = std::max(gram_lds, static_cast<size_t>(2) * kB * (a.nslots | 1) * 16);  // (the cross blocks' workgroups: two blocks' rows)
    ensure_max_dynamic_lds(reinterpret_cast<const void *>(bs_gram_kernel));
    // `stop`: an event completed by the kernel's own completion signal (hipExtLaunchKernelGGL) -- a hipEventRecord behind the
    // launch is a marker packet of its own and held the NEXT launch of the stream back by 6-7 us (kernel traces, round 3)
    const int grid = it.max_blocks + (it.la_ranks > 0 ? it.la_grid : 0);
    if (stop) hipExtLaunchKernelGGL(bs_gram_kernel, dim3(grid), dim3(256), static_cast<uint32_t>(gram_lds), stream, nullptr, stop, 0, a, it);
    else hipLaunchKernelGGL(bs_gram_kernel, dim3(grid), dim3(256), gram_lds, stream, a, it);
    CU2REC_HIP(hipGetLastError());
}

namespace {
// A gate that gives up just ENDS (the side kernel then starts a little early: a matter of timing, no data hangs on it): it must not
// set the status word -- that would declare the model state undefined for a delay that harms nothing.
__global__ void bs_gate_kernel(const unsigned long long *count, unsigned long long target, const unsigned long long *started,
                               unsigned long long started_target, unsigned long long wait_ticks) {
    // lanes 0-31: one shard of phase 1's count each; lane 32: phase 2's workgroups that hold their CU
    const int lane = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    unsigned polls = 0;
    for (;;) {
        unsigned long long v = lane < 32 ? ld_agent(count + 16 * lane) : 0ull;
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off);  // (lanes 32-63 sum zeros)
        const unsigned long long sum = __shfl(v, 0);
        const unsigned long long st = ld_agent(started);
        if (sum >= target && st >= started_target) break;
        __builtin_amdgcn_s_sleep(16);
        if ((++polls & 63u) == 0 && wall_clock64() - t0 > wait_ticks) break;
    }
}
}  // namespace

namespace {
__global__ void bs_signal_kernel(unsigned long long *word, unsigned long long value) {
    if (threadIdx.x == 0) __hip_atomic_store(word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
}  // namespace

void bs_launch_signal(unsigned long long *word, unsigned long long value, hipStream_t stream) {
    hipLaunchKernelGGL(bs_signal_kernel, dim3(1), dim3(64), 0, stream, word, value);
    CU2REC_HIP(hipGetLastError());
}

void bs_launch_gate(const unsigned long long *count, unsigned long long target, const unsigned long long *started,
                    unsigned long long started_target, hipStream_t stream) {
    hipLaunchKernelGGL(bs_gate_kernel, dim3(1), dim3(64), 0, stream, count, target, started, started_target, bs_wait_ticks());
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
    const int nwaves = (4 * a.nslots + 63) / 64;
    hipLaunchKernelGGL(bs_update_kernel, dim3(it.launch_blocks + (it.side_seq ? 1 : 0)), dim3(64 * nwaves), 0, stream, a, it, nwaves);
    CU2REC_HIP(hipGetLastError());
}

}  // namespace cu2rec
