// HIP kernels of the cu2rec hot path for gfx950 (MI355X / CDNA4).  Compiled with
// -ffp-contract=off: every fused multiply-add below is an explicit __builtin_fmaf, so the
// arithmetic is defined operation by operation and the CPU oracle can mirror it bit for bit.
//
// Geometry (all kernels): a 16-lane group -- one DPP row of a 64-wide wavefront -- owns one
// user (SGD) or one chunk of ratings (loss).  Lane l of the group owns the float4 "slots"
// l, l+16, l+32, ... of a factor row (row stride ld, a multiple of 4 floats, zero padded), so a
// row of f <= 64 floats is ONE coalesced 16 B-per-lane access and f = 100 is two.  A dot
// product is a per-lane fmaf chain over the lane's slots followed by a 4-step xor butterfly
// inside the row, done with DPP modifiers (quad_perm / row_half_mirror / row_mirror): no LDS,
// no ds_bpermute, and every lane of the group ends with the same bits.  Four users share a
// wavefront; 256-thread blocks hold 16 users.
//
// What each kernel replaces in the reference (matrix_factorization/):
//   sgd_hogwild_kernel / sgd_serial_kernel  sgd_update       sgd.cu:22-75   (+ initCurand :11-16: gone)
//   pingpong_claim_kernel + sgd_pingpong_kernel   sgd_update with its Q_target / item_is_updated semantics kept
//   loss_fused_kernel                       loss_kernel      loss.cu:19-35  + total_loss_kernel<B> :58-128 (x2)
//   error_metrics_kernel                    total_loss_kernel<B>  loss.cu:58-128 on an explicit array
//   items_delta_*                           (new: multi-GPU item-factor exchange)
#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "sampler.hpp"
#include "sgd_device.hpp"

#ifndef CU2REC_ABLATE_SAMPLER
#define CU2REC_ABLATE_SAMPLER 0  // timing-only builds (see sgd_one); never set in the shipped library
#endif
#ifndef CU2REC_STREAM_P
#define CU2REC_STREAM_P 0  // 1: non-temporal loads / stores for the user rows of the Hogwild kernel (A/B builds)
#endif

namespace cu2rec {

namespace {

using namespace dev;

// One SGD update of user x at iteration `it` (sgd.cu:27-73 / mf_sequential.cu:104-142), p and
// the user bias live in registers; the item row is read, updated and written back in place.
template <int J>
__device__ __forceinline__ void sgd_one(const SgdArgs &a, int x, int low, int high, uint64_t it, int lane,
                                        Row<J> &p, float &ub) {
#if CU2REC_ABLATE_SAMPLER == 1
    // timing only: no Philox, no CSR gathers -- a cheap in-register hash picks the item
    const int y = static_cast<int>((static_cast<uint32_t>(x) * 2654435761u + static_cast<uint32_t>(it) * 40503u) % 26744u);
    const float rating = 3.5f;
    (void)low; (void)high;
#elif CU2REC_ABLATE_SAMPLER == 2
    // timing only: Philox kept, CSR gathers dropped
    const int y_i = sampler_index(a.seed, static_cast<uint64_t>(a.user_offset + x), it, low, high);
    const int y = static_cast<int>(static_cast<uint32_t>(y_i) * 2654435761u % 26744u);
    const float rating = 3.5f;
#else
    const int y_i = sampler_index(a.seed, static_cast<uint64_t>(a.user_offset + x), it, low, high);  // sgd.cu:36-37
    const int y = a.indices[y_i];
    const float rating = a.data[y_i];
#endif
    Row<J> q = load_row<J>(a.Q, static_cast<size_t>(y), a.ldq, a.nslots, lane);
    const float ib = a.item_bias[y];
    const float err = rating - predict<J>(p, q, ub, ib, a.global_bias);  // sgd.cu:45
    rank1_update<J>(p, q, err, a.h);                                      // sgd.cu:53-64
    if (a.update_items) {                                                 // config::is_train, sgd.cu:61,70
        store_row<J>(a.Q, static_cast<size_t>(y), a.ldq, a.nslots, lane, q);
        if (lane == 0) a.item_bias[y] = ib + a.h.lr * (err - a.h.ib_reg * ib);  // sgd.cu:71
    }
    ub = ub + a.h.lr * (err - a.h.ub_reg * ub);  // sgd.cu:67
}

// ---- SGD, Hogwild: every user of the iteration in flight ------------------------------------
template <int J>
__global__ __launch_bounds__(kBlock) void sgd_hogwild_kernel(SgdArgs a) {
    const int lane = threadIdx.x & (kGroup - 1);
    const int group = (blockIdx.x * kBlock + threadIdx.x) / kGroup;
    const int n_groups = gridDim.x * kGroupsPerBlock;
    for (int x = group; x < a.n_rows; x += n_groups) {
        const int low = a.indptr[x], high = a.indptr[x + 1];
        if (low == high) continue;  // sgd.cu:34: users without ratings are skipped
#if CU2REC_STREAM_P
        Row<J> p = load_row_stream<J>(a.P, static_cast<size_t>(x), a.ldp, a.nslots, lane);
#else
        Row<J> p = load_row<J>(a.P, static_cast<size_t>(x), a.ldp, a.nslots, lane);
#endif
        float ub = a.user_bias[x];
        for (int k = 0; k < a.iters; ++k) sgd_one<J>(a, x, low, high, a.iter0 + static_cast<uint64_t>(k), lane, p, ub);
#if CU2REC_STREAM_P
        store_row_stream<J>(a.P, static_cast<size_t>(x), a.ldp, a.nslots, lane, p);
#else
        store_row<J>(a.P, static_cast<size_t>(x), a.ldp, a.nslots, lane, p);
#endif
        if (lane == 0) a.user_bias[x] = ub;
    }
}

// ---- SGD, serial order: one group walks the users ascending (mf_sequential.cu:102-143) -----
template <int J>
__global__ __launch_bounds__(64) void sgd_serial_kernel(SgdArgs a) {
    if (threadIdx.x >= kGroup || blockIdx.x != 0) return;
    const int lane = threadIdx.x;
    for (int k = 0; k < a.iters; ++k) {
        for (int x = 0; x < a.n_rows; ++x) {
            const int low = a.indptr[x], high = a.indptr[x + 1];
            if (low == high) continue;
            Row<J> p = load_row<J>(a.P, static_cast<size_t>(x), a.ldp, a.nslots, lane);
            float ub = a.user_bias[x];
            sgd_one<J>(a, x, low, high, a.iter0 + static_cast<uint64_t>(k), lane, p, ub);
            store_row<J>(a.P, static_cast<size_t>(x), a.ldp, a.nslots, lane, p);
            if (lane == 0) a.user_bias[x] = ub;
            __threadfence();  // the next user may read the item row this one just wrote
        }
    }
}

// ---- SGD, the reference GPU kernel's own semantics (CU2REC_SGD_PINGPONG): sgd.cu:22-75 -------------------------
// Who is the "early bird" of an item (sgd.cu:49-50)?  The reference races on a bool; here every user claims its item
// with a 64-bit atomicMin of (~iteration << 32 | thread index): newer iterations beat the leftovers of older ones (no
// per-iteration memset, training.cu:168), and inside an iteration the lowest thread index wins.  Thread gid handles
// user (gid + start_user) % rows with start_user = 250 * iteration (training.cu:97-98,115).
__device__ __forceinline__ unsigned long long pingpong_key(uint64_t it, int x, int n_rows) {
    const int start_user = static_cast<int>((250u * it) % static_cast<uint64_t>(n_rows));
    const int gid = x >= start_user ? x - start_user : x - start_user + n_rows;
    return (static_cast<unsigned long long>(~static_cast<uint32_t>(it)) << 32) | static_cast<unsigned>(gid);
}

__global__ __launch_bounds__(kBlock) void pingpong_claim_kernel(SgdArgs a, PingPongArgs pp) {
    for (int x = blockIdx.x * kBlock + threadIdx.x; x < a.n_rows; x += gridDim.x * kBlock) {
        const int low = a.indptr[x], high = a.indptr[x + 1];
        if (low == high) continue;
        const int y = a.indices[sampler_index(a.seed, static_cast<uint64_t>(a.user_offset + x), a.iter0, low, high)];
        atomicMin(&pp.claim[y], pingpong_key(a.iter0, x, a.n_rows));
    }
}

template <int J>
__global__ __launch_bounds__(kBlock) void sgd_pingpong_kernel(SgdArgs a, PingPongArgs pp) {
    const int lane = threadIdx.x & (kGroup - 1);
    const int group = (blockIdx.x * kBlock + threadIdx.x) / kGroup;
    const int n_groups = gridDim.x * kGroupsPerBlock;
    for (int x = group; x < a.n_rows; x += n_groups) {
        const int low = a.indptr[x], high = a.indptr[x + 1];
        if (low == high) continue;  // sgd.cu:34
        const int y_i = sampler_index(a.seed, static_cast<uint64_t>(a.user_offset + x), a.iter0, low, high);  // sgd.cu:36-37
        const int y = a.indices[y_i];
        Row<J> p = load_row<J>(a.P, static_cast<size_t>(x), a.ldp, a.nslots, lane);
        Row<J> q = load_row<J>(a.Q, static_cast<size_t>(y), a.ldq, a.nslots, lane);  // nobody writes a.Q in this launch
        const float ub = a.user_bias[x], ib = a.item_bias[y];
        const float err = a.data[y_i] - predict<J>(p, q, ub, ib, a.global_bias);  // sgd.cu:45
        const bool early_bird = pp.claim[y] == pingpong_key(a.iter0, x, a.n_rows);  // sgd.cu:49-50
        rank1_update<J>(p, q, err, a.h);                                            // sgd.cu:53-64
        store_row<J>(a.P, static_cast<size_t>(x), a.ldp, a.nslots, lane, p);
        if (lane == 0) a.user_bias[x] = ub + a.h.lr * (err - a.h.ub_reg * ub);  // sgd.cu:67
        if (a.update_items && early_bird) {                                     // sgd.cu:61,70
            store_row<J>(pp.Q_target, static_cast<size_t>(y), a.ldq, a.nslots, lane, q);
            if (lane == 0) pp.item_bias_target[y] = ib + a.h.lr * (err - a.h.ib_reg * ib);  // sgd.cu:71
        }
    }
}

// ---- fused loss: residual + sum|e| + sum e^2 in one pass ---------------------------------------
constexpr int kLossChunk = 64;  // ratings per group per step

// largest u in [0, n_rows) with indptr[u] <= k (k < nnz guarantees indptr[u+1] > k after skipping empties)
__device__ __forceinline__ int user_of_rating(const int *__restrict__ indptr, int n_rows, int k) {
    int lo = 0, hi = n_rows;  // invariant: indptr[lo] <= k < indptr[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (indptr[mid] <= k) lo = mid; else hi = mid;
    }
    return lo;
}

template <int J>
__global__ __launch_bounds__(kBlock) void loss_fused_kernel(LossArgs a) {
    __shared__ double s_abs[kGroupsPerBlock], s_sq[kGroupsPerBlock];
    const int lane = threadIdx.x & (kGroup - 1);
    const int group_in_block = threadIdx.x / kGroup;
    const int group = blockIdx.x * kGroupsPerBlock + group_in_block;
    const int n_groups = gridDim.x * kGroupsPerBlock;
    const int n_chunks = (a.nnz + kLossChunk - 1) / kLossChunk;
    double sum_abs = 0.0, sum_sq = 0.0;
    for (int chunk = group; chunk < n_chunks; chunk += n_groups) {
        const int k0 = chunk * kLossChunk;
        const int k1 = min(a.nnz, k0 + kLossChunk);
        int u = user_of_rating(a.indptr, a.n_rows, k0);
        int row_end = a.indptr[u + 1];
        Row<J> p = load_row<J>(a.P, static_cast<size_t>(u), a.ldp, a.nslots, lane);
        float ub = a.user_bias[u];
        for (int kb = k0; kb < k1; kb += kGroup) {
            // one coalesced 64 B read of 16 item ids and 16 ratings per group (loss.cu:29-31 walks them one by one)
            const int mine = kb + lane;
            const int my_item = mine < k1 ? a.indices[mine] : 0;
            const float my_rating = mine < k1 ? a.data[mine] : 0.f;
            const int n_here = min(kGroup, k1 - kb);
            for (int t = 0; t < n_here; ++t) {
                const int k = kb + t;
                while (k >= row_end) {  // next user (users without ratings are stepped over)
                    ++u;
                    row_end = a.indptr[u + 1];
                    if (k < row_end) {
                        p = load_row<J>(a.P, static_cast<size_t>(u), a.ldp, a.nslots, lane);
                        ub = a.user_bias[u];
                    }
                }
                const int item = __shfl(my_item, t, kGroup);
                const float rating = __shfl(my_rating, t, kGroup);
                const Row<J> q = load_row<J>(a.Q, static_cast<size_t>(item), a.ldq, a.nslots, lane);
                const float e = rating - predict<J>(p, q, ub, a.item_bias[item], a.global_bias);  // loss.cu:31
                if (a.errors_out != nullptr && lane == 0) a.errors_out[k] = e;
                sum_abs += static_cast<double>(fabsf(e));                    // loss.cu:70 (MAE pass)
                sum_sq += static_cast<double>(e) * static_cast<double>(e);   // loss.cu:70 (RMSE pass, pow(e,2) in double)
            }
        }
    }
    // every lane of a group carries the same sums; lane 0 publishes, thread 0 adds the 16 groups in order
    if (lane == 0) {
        s_abs[group_in_block] = sum_abs;
        s_sq[group_in_block] = sum_sq;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ta = 0.0, ts = 0.0;
        for (int g = 0; g < kGroupsPerBlock; ++g) {
            ta += s_abs[g];
            ts += s_sq[g];
        }
        a.partials[2 * blockIdx.x] = ta;
        a.partials[2 * blockIdx.x + 1] = ts;
    }
}

// ---- total loss on an explicit residual array (loss.cu:58-128, both error types at once) ----
__global__ __launch_bounds__(kBlock) void error_metrics_kernel(const float *__restrict__ errors, int n,
                                                               double *__restrict__ partials) {
    __shared__ double s_abs[kBlock], s_sq[kBlock];
    double sa = 0.0, ss = 0.0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const float e = errors[i];
        sa += static_cast<double>(fabsf(e));
        ss += static_cast<double>(e) * static_cast<double>(e);
    }
    s_abs[threadIdx.x] = sa;
    s_sq[threadIdx.x] = ss;
    __syncthreads();
    for (int stride = kBlock / 2; stride > 0; stride >>= 1) {  // every step guarded and barriered (cf. loss.cu:103-124)
        if (static_cast<int>(threadIdx.x) < stride) {
            s_abs[threadIdx.x] += s_abs[threadIdx.x + stride];
            s_sq[threadIdx.x] += s_sq[threadIdx.x + stride];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = s_abs[0];
        partials[2 * blockIdx.x + 1] = s_sq[0];
    }
}

// ---- sample array: indices[k] and data[k] side by side (SgdArgs::pairs) ----------------------------------------
__global__ __launch_bounds__(kBlock) void sample_pairs_build_kernel(const int *__restrict__ indices,
                                                                    const float *__restrict__ data, size_t nnz,
                                                                    uint2 *__restrict__ pairs) {
    for (size_t k = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; k < nnz;
         k += static_cast<size_t>(gridDim.x) * kBlock)
        pairs[k] = make_uint2(static_cast<unsigned>(indices[k]), __float_as_uint(data[k]));
}

// ---- multi-GPU item-factor exchange helpers -------------------------------------------------
__global__ __launch_bounds__(kBlock) void items_delta_pack_kernel(const float *__restrict__ Q,
                                                                  const float *__restrict__ ib,
                                                                  const float *__restrict__ Q_base,
                                                                  const float *__restrict__ ib_base, size_t nq,
                                                                  int n_cols, float *__restrict__ buf) {
    const size_t total = nq + static_cast<size_t>(n_cols);
    for (size_t i = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; i < total;
         i += static_cast<size_t>(gridDim.x) * kBlock)
        buf[i] = i < nq ? Q[i] - Q_base[i] : ib[i - nq] - ib_base[i - nq];
}

__global__ __launch_bounds__(kBlock) void items_delta_pack_weighted_kernel(const float *__restrict__ Q,
                                                                           const float *__restrict__ ib,
                                                                           const float *__restrict__ Q_base,
                                                                           const float *__restrict__ ib_base,
                                                                           const float *__restrict__ weight, size_t nq,
                                                                           int n_cols, int ldq, float *__restrict__ buf) {
    const size_t total = nq + static_cast<size_t>(n_cols);
    for (size_t i = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; i < total;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        if (i < nq) buf[i] = weight[i / ldq] * (Q[i] - Q_base[i]);
        else buf[i] = weight[i - nq] * (ib[i - nq] - ib_base[i - nq]);
    }
}

__global__ __launch_bounds__(kBlock) void items_delta_apply_kernel(float *__restrict__ Q, float *__restrict__ ib,
                                                                   float *__restrict__ Q_base,
                                                                   float *__restrict__ ib_base, size_t nq, int n_cols,
                                                                   const float *__restrict__ buf, float scale) {
    const size_t total = nq + static_cast<size_t>(n_cols);
    for (size_t i = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; i < total;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        if (i < nq) {
            const float v = Q_base[i] + scale * buf[i];
            Q[i] = v;
            Q_base[i] = v;
        } else {
            const float v = ib_base[i - nq] + scale * buf[i];
            ib[i - nq] = v;
            ib_base[i - nq] = v;
        }
    }
}

__global__ __launch_bounds__(kBlock) void items_delta_apply_overlapped_kernel(
    float *__restrict__ Q, float *__restrict__ ib, float *__restrict__ Q_base, float *__restrict__ ib_base,
    const float *__restrict__ Q_snap, const float *__restrict__ ib_snap, size_t nq, int n_cols,
    const float *__restrict__ buf, float scale) {
    const size_t total = nq + static_cast<size_t>(n_cols);
    for (size_t i = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; i < total;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        if (i < nq) {
            const float merged = Q_base[i] + scale * buf[i];
            Q[i] = merged + (Q[i] - Q_snap[i]);
            Q_base[i] = merged;
        } else {
            const size_t j = i - nq;
            const float merged = ib_base[j] + scale * buf[i];
            ib[j] = merged + (ib[j] - ib_snap[j]);
            ib_base[j] = merged;
        }
    }
}

// ---- the exchange in its wire format: n_cols * f item-row deltas, then n_cols bias deltas, no row padding --------------
// (the padded form above moves ldq floats per row: 13.8 MB instead of 10.8 MB at f = 100; what crosses xGMI is this one)
__global__ __launch_bounds__(kBlock) void items_wire_pack_kernel(const float *__restrict__ Q, const float *__restrict__ ib,
                                                                 const float *__restrict__ Q_base,
                                                                 const float *__restrict__ ib_base,
                                                                 const float *__restrict__ weight, int n_cols, int f, int ldq,
                                                                 float *__restrict__ wire) {
    const size_t nq = static_cast<size_t>(n_cols) * f, total = nq + static_cast<size_t>(n_cols);
    for (size_t i = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; i < total;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        if (i < nq) {
            const size_t y = i / f, c = i - y * f, at = y * ldq + c;
            const float d = Q[at] - Q_base[at];
            wire[i] = weight ? weight[y] * d : d;
        } else {
            const size_t y = i - nq;
            const float d = ib[y] - ib_base[y];
            wire[i] = weight ? weight[y] * d : d;
        }
    }
}

__global__ __launch_bounds__(kBlock) void items_wire_apply_kernel(float *__restrict__ Q, float *__restrict__ ib,
                                                                  float *__restrict__ Q_base, float *__restrict__ ib_base,
                                                                  int n_cols, int f, int ldq, const float *__restrict__ wire,
                                                                  float scale) {
    const size_t nq = static_cast<size_t>(n_cols) * f, total = nq + static_cast<size_t>(n_cols);
    for (size_t i = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; i < total;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        if (i < nq) {
            const size_t y = i / f, c = i - y * f, at = y * ldq + c;
            const float v = Q_base[at] + scale * wire[i];
            Q[at] = v;
            Q_base[at] = v;
        } else {
            const size_t y = i - nq;
            const float v = ib_base[y] + scale * wire[i];
            ib[y] = v;
            ib_base[y] = v;
        }
    }
}

inline int blocks_for(size_t work_items, size_t per_block, int cap) {
    size_t b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > static_cast<size_t>(cap)) b = static_cast<size_t>(cap);
    return static_cast<int>(b);
}

template <int J>
void launch_sgd_j(const SgdArgs &a, int mode, hipStream_t stream) {
    if (mode == CU2REC_SGD_SERIAL) {
        hipLaunchKernelGGL(sgd_serial_kernel<J>, dim3(1), dim3(64), 0, stream, a);
    } else {
        // one group per user; the dispatcher keeps the CUs fed with far more than 256 blocks
        const int blocks = blocks_for(static_cast<size_t>(a.n_rows), kGroupsPerBlock, 1 << 20);
        hipLaunchKernelGGL(sgd_hogwild_kernel<J>, dim3(blocks), dim3(kBlock), 0, stream, a);
    }
}

template <int J>
void launch_pingpong_j(const SgdArgs &a, const PingPongArgs &pp, hipStream_t stream) {
    const int blocks = blocks_for(static_cast<size_t>(a.n_rows), kGroupsPerBlock, 1 << 20);
    hipLaunchKernelGGL(sgd_pingpong_kernel<J>, dim3(blocks), dim3(kBlock), 0, stream, a, pp);
}

template <int J>
void launch_loss_j(const LossArgs &a, int blocks, hipStream_t stream) {
    hipLaunchKernelGGL(loss_fused_kernel<J>, dim3(blocks), dim3(kBlock), 0, stream, a);
}

}  // namespace

int slots_per_lane(int nslots) { return (nslots + kGroup - 1) / kGroup; }

void launch_sgd(const SgdArgs &a, int mode, hipStream_t stream) {
    switch (slots_per_lane(a.nslots)) {
        case 1: launch_sgd_j<1>(a, mode, stream); break;
        case 2: launch_sgd_j<2>(a, mode, stream); break;
        case 3: launch_sgd_j<3>(a, mode, stream); break;
        case 4: launch_sgd_j<4>(a, mode, stream); break;
        case 5: launch_sgd_j<5>(a, mode, stream); break;
        case 6: launch_sgd_j<6>(a, mode, stream); break;
        case 7: launch_sgd_j<7>(a, mode, stream); break;
        case 8: launch_sgd_j<8>(a, mode, stream); break;
        default: fail(CU2REC_EUNSUPPORTED, "n_factors above 512 is not compiled in");
    }
}

void launch_sgd_pingpong(const SgdArgs &a, const PingPongArgs &pp, hipStream_t stream) {
    hipLaunchKernelGGL(pingpong_claim_kernel, dim3(blocks_for(static_cast<size_t>(a.n_rows), kBlock, 4096)), dim3(kBlock), 0,
                       stream, a, pp);
    switch (slots_per_lane(a.nslots)) {
        case 1: launch_pingpong_j<1>(a, pp, stream); break;
        case 2: launch_pingpong_j<2>(a, pp, stream); break;
        case 3: launch_pingpong_j<3>(a, pp, stream); break;
        case 4: launch_pingpong_j<4>(a, pp, stream); break;
        case 5: launch_pingpong_j<5>(a, pp, stream); break;
        case 6: launch_pingpong_j<6>(a, pp, stream); break;
        case 7: launch_pingpong_j<7>(a, pp, stream); break;
        case 8: launch_pingpong_j<8>(a, pp, stream); break;
        default: fail(CU2REC_EUNSUPPORTED, "n_factors above 512 is not compiled in");
    }
}

int loss_blocks(int nnz) {
    const size_t chunks = (static_cast<size_t>(nnz) + kLossChunk - 1) / kLossChunk;
    return blocks_for(chunks, kGroupsPerBlock, kMaxPartialBlocks);
}

void launch_loss(const LossArgs &a, int blocks, hipStream_t stream) {
    switch (slots_per_lane(a.nslots)) {
        case 1: launch_loss_j<1>(a, blocks, stream); break;
        case 2: launch_loss_j<2>(a, blocks, stream); break;
        case 3: launch_loss_j<3>(a, blocks, stream); break;
        case 4: launch_loss_j<4>(a, blocks, stream); break;
        case 5: launch_loss_j<5>(a, blocks, stream); break;
        case 6: launch_loss_j<6>(a, blocks, stream); break;
        case 7: launch_loss_j<7>(a, blocks, stream); break;
        case 8: launch_loss_j<8>(a, blocks, stream); break;
        default: fail(CU2REC_EUNSUPPORTED, "n_factors above 512 is not compiled in");
    }
}

namespace {
// ---- the per-block partial sums of a loss pass -> two numbers, on the device (loss.cu:185-190 adds them on the host behind a
// blocking copy of all of them): one workgroup, thread t adds partials t, t + 256, ... in that order, then a guarded tree; fixed
// order, hence the same bits run after run.  out = partials + 2 * kMaxPartialBlocks: {sum |e|, sum e^2}.
__global__ __launch_bounds__(kBlock) void partials_reduce_kernel(const double *__restrict__ partials, int blocks, double *__restrict__ out) {
    __shared__ double s_abs[kBlock], s_sq[kBlock];
    double sa = 0.0, ss = 0.0;
    for (int b = threadIdx.x; b < blocks; b += kBlock) {
        sa += partials[2 * b];
        ss += partials[2 * b + 1];
    }
    s_abs[threadIdx.x] = sa;
    s_sq[threadIdx.x] = ss;
    __syncthreads();
    for (int stride = kBlock / 2; stride > 0; stride >>= 1) {
        if (static_cast<int>(threadIdx.x) < stride) {
            s_abs[threadIdx.x] += s_abs[threadIdx.x + stride];
            s_sq[threadIdx.x] += s_sq[threadIdx.x + stride];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = s_abs[0];
        out[1] = s_sq[0];
    }
}
}  // namespace

void launch_partials_reduce(double *partials, int blocks, hipStream_t stream) {
    hipLaunchKernelGGL(partials_reduce_kernel, dim3(1), dim3(kBlock), 0, stream, partials, blocks, partials + 2 * kMaxPartialBlocks);
}

int error_metrics_blocks(int n) { return blocks_for(static_cast<size_t>(n), kBlock * 4, kMaxPartialBlocks); }

void launch_error_metrics(const float *errors, int n, double *partials, int blocks, hipStream_t stream) {
    hipLaunchKernelGGL(error_metrics_kernel, dim3(blocks), dim3(kBlock), 0, stream, errors, n, partials);
}

void launch_sample_pairs_build(const int *indices, const float *data, size_t nnz, uint2 *pairs, hipStream_t stream) {
    const int blocks = blocks_for(nnz, kBlock * 8, 8192);
    hipLaunchKernelGGL(sample_pairs_build_kernel, dim3(blocks), dim3(kBlock), 0, stream, indices, data, nnz, pairs);
}

void launch_items_delta_pack(const float *Q, const float *ib, const float *Q_base, const float *ib_base, int n_cols,
                             int ldq, float *buf, hipStream_t stream) {
    const size_t nq = static_cast<size_t>(n_cols) * ldq;
    const int blocks = blocks_for(nq + n_cols, kBlock * 4, 4096);
    hipLaunchKernelGGL(items_delta_pack_kernel, dim3(blocks), dim3(kBlock), 0, stream, Q, ib, Q_base, ib_base, nq,
                       n_cols, buf);
}

void launch_items_delta_apply_overlapped(float *Q, float *ib, float *Q_base, float *ib_base, const float *Q_snap,
                                         const float *ib_snap, int n_cols, int ldq, const float *buf, float scale,
                                         hipStream_t stream) {
    const size_t nq = static_cast<size_t>(n_cols) * ldq;
    const int blocks = blocks_for(nq + n_cols, kBlock * 4, 4096);
    hipLaunchKernelGGL(items_delta_apply_overlapped_kernel, dim3(blocks), dim3(kBlock), 0, stream, Q, ib, Q_base, ib_base,
                       Q_snap, ib_snap, nq, n_cols, buf, scale);
}

void launch_items_delta_pack_weighted(const float *Q, const float *ib, const float *Q_base, const float *ib_base,
                                      const float *weight, int n_cols, int ldq, float *buf, hipStream_t stream) {
    const size_t nq = static_cast<size_t>(n_cols) * ldq;
    const int blocks = blocks_for(nq + n_cols, kBlock * 4, 4096);
    hipLaunchKernelGGL(items_delta_pack_weighted_kernel, dim3(blocks), dim3(kBlock), 0, stream, Q, ib, Q_base, ib_base,
                       weight, nq, n_cols, ldq, buf);
}

void launch_items_wire_pack(const float *Q, const float *ib, const float *Q_base, const float *ib_base, const float *weight,
                            int n_cols, int f, int ldq, float *wire, hipStream_t stream) {
    const int blocks = blocks_for(static_cast<size_t>(n_cols) * (f + 1), kBlock * 4, 4096);
    hipLaunchKernelGGL(items_wire_pack_kernel, dim3(blocks), dim3(kBlock), 0, stream, Q, ib, Q_base, ib_base, weight, n_cols, f,
                       ldq, wire);
}

void launch_items_wire_apply(float *Q, float *ib, float *Q_base, float *ib_base, int n_cols, int f, int ldq, const float *wire,
                             float scale, hipStream_t stream) {
    const int blocks = blocks_for(static_cast<size_t>(n_cols) * (f + 1), kBlock * 4, 4096);
    hipLaunchKernelGGL(items_wire_apply_kernel, dim3(blocks), dim3(kBlock), 0, stream, Q, ib, Q_base, ib_base, n_cols, f, ldq,
                       wire, scale);
}

void launch_items_delta_apply(float *Q, float *ib, float *Q_base, float *ib_base, int n_cols, int ldq,
                              const float *buf, float scale, hipStream_t stream) {
    const size_t nq = static_cast<size_t>(n_cols) * ldq;
    const int blocks = blocks_for(nq + n_cols, kBlock * 4, 4096);
    hipLaunchKernelGGL(items_delta_apply_kernel, dim3(blocks), dim3(kBlock), 0, stream, Q, ib, Q_base, ib_base, nq,
                       n_cols, buf, scale);
}

}  // namespace cu2rec
