// Scoring and ranking for bin/predict: predict_ratings (predict.cu:17-30) and
// get_recommendations (predict.cu:49-65) for MANY users at once, on the device.
//
//   scores[u][i] = ((gb + ub[u]) + ib[i]) + p_u . q_i        one dense product P_batch (B x f) * Q^T (f x I): the one
// place on this path where a contraction is wide on both sides, so it runs on the matrix cores
// (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, bit-for-bit an fmaf chain).  A workgroup owns 32 users x 128 items:
// the 32 user rows are staged once in LDS and shared by four wavefronts, each of which stages 32 item rows of its own
// (coalesced 128-byte pieces, odd row stride so that the operand reads -- one row per lane -- are conflict free) and
// accumulates one 32x32 tile.  Lane (r = l & 31, h = l >> 5) feeds half h of row r as A (users) and as B (items); the
// contraction index pairs column c of half 0 with column c of half 1, which only reorders the sum.
//
// Ranking = every item a user has NOT rated, best predicted rating first: rated items are masked with -inf, then one
// segmented radix sort (descending) over all users of a batch; the first k per user are the recommendations.
#include <hip/hip_runtime.h>

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

#include "device.hpp"
#include "hip_check.hpp"

namespace cu2rec {

namespace {

constexpr int kTile = 32;
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// 32 rows of `nslots` float4 from a row-major matrix into an LDS tile [32][RS], by one wavefront: 8 lanes x 16 bytes per
// row piece, 8 rows per pass; rows beyond n_rows are zero.  Loads are unconditional at clamped addresses.
__device__ __forceinline__ void stage_rows(float4 *tile, const float *__restrict__ base, int ld, int first_row, int n_rows,
                                           int nslots, int RS, int lane) {
    const int rsub = lane >> 3, cs = lane & 7;
    const int nch = (nslots + 7) >> 3;
    for (int c = 0; c < nch; ++c) {
        float4 v[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = min(first_row + 8 * p + rsub, n_rows - 1);
            v[p] = reinterpret_cast<const float4 *>(base + static_cast<size_t>(row) * ld)[min(8 * c + cs, nslots - 1)];
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int slot = 8 * c + cs;
            if (slot < nslots) tile[(8 * p + rsub) * RS + slot] = first_row + 8 * p + rsub < n_rows ? v[p] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

__global__ __launch_bounds__(256) void scores_kernel(const float *__restrict__ P, int ldp, const float *__restrict__ user_bias,
                                                     int n_users, const float *__restrict__ Q, int ldq,
                                                     const float *__restrict__ item_bias, int n_items, float global_bias,
                                                     int nslots, float *__restrict__ scores) {
    extern __shared__ float4 smem[];
    const int RS = nslots | 1;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int user0 = blockIdx.y * kTile, item0 = (blockIdx.x * 4 + wave) * kTile;
    float4 *utile = smem, *itile = smem + (1 + wave) * kTile * RS;
    // wave w stages rows 8 w .. 8 w + 7 of the user tile (all four together: 32 rows) and its own 32 item rows
    {
        const int rsub = lane >> 3, cs = lane & 7;
        for (int c = 0; c < (nslots + 7) >> 3; ++c) {
            const int row = min(user0 + 8 * wave + rsub, n_users - 1), slot = 8 * c + cs;
            const float4 v = reinterpret_cast<const float4 *>(P + static_cast<size_t>(row) * ldp)[min(slot, nslots - 1)];
            if (slot < nslots) utile[(8 * wave + rsub) * RS + slot] = user0 + 8 * wave + rsub < n_users ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (item0 < n_items) stage_rows(itile, Q, ldq, item0, n_items, nslots, RS, lane);
    __syncthreads();
    if (item0 >= n_items) return;
    const int r = lane & 31, h = lane >> 5;
    const int S0 = (nslots + 1) >> 1;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int c = 0; c < S0; c += 2) {
        float4 a4[2], b4[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int slot = h * S0 + c + i;
            const bool ok = c + i < S0 && slot < nslots;
            const float4 ua = utile[r * RS + min(slot, nslots - 1)], ib = itile[r * RS + min(slot, nslots - 1)];
            a4[i] = ok ? ua : make_float4(0.f, 0.f, 0.f, 0.f);
            b4[i] = ok ? ib : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].x, b4[i].x, acc, 0, 0, 0);  // rows: users, columns: items
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].y, b4[i].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].z, b4[i].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].w, b4[i].w, acc, 0, 0, 0);
        }
    }
    const int item = item0 + r;  // accumulator column = lane & 31
    if (item >= n_items) return;
    const float ibv = item_bias[item];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int u = user0 + acc_row(reg, h);
        if (u < n_users) scores[static_cast<size_t>(u) * n_items + item] = ((global_bias + user_bias[u]) + ibv) + acc[reg];  // predict.cu:24-27
    }
}

// get_recommendations only lists items the user has not rated (predict.cu:54-62): their scores drop out of the ranking
__global__ __launch_bounds__(256) void mask_rated_kernel(const int *__restrict__ indptr, const int *__restrict__ indices,
                                                         int user0, int n_users, int n_items, float *__restrict__ keys) {
    for (int u = blockIdx.x; u < n_users; u += gridDim.x) {
        const int low = indptr[user0 + u], high = indptr[user0 + u + 1];
        for (int k = low + threadIdx.x; k < high; k += blockDim.x)
            keys[static_cast<size_t>(u) * n_items + indices[k]] = -__builtin_inff();
    }
}

__global__ __launch_bounds__(256) void iota_items_kernel(int *__restrict__ items, size_t total, int n_items) {
    for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<size_t>(gridDim.x) * 256)
        items[i] = static_cast<int>(i % n_items);
}

__global__ __launch_bounds__(256) void take_top_kernel(const float *__restrict__ keys, const int *__restrict__ items, int n_users,
                                                       int n_items, int k, float *__restrict__ top_scores, int *__restrict__ top_items) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_users * k; i += gridDim.x * 256) {
        const int u = i / k, j = i - u * k;
        const bool have = j < n_items && keys[static_cast<size_t>(u) * n_items + j] != -__builtin_inff();
        top_scores[i] = have ? keys[static_cast<size_t>(u) * n_items + j] : __builtin_nanf("");
        top_items[i] = have ? items[static_cast<size_t>(u) * n_items + j] : -1;
    }
}

void launch_scores(const DeviceModel &m, int user0, int n_users, float *scores, hipStream_t stream) {
    const int nslots = (m.n_factors + 3) / 4;
    const size_t lds = static_cast<size_t>(5) * kTile * (nslots | 1) * 16;
    ensure_max_dynamic_lds(reinterpret_cast<const void *>(scores_kernel));
    const dim3 grid((m.cols + 4 * kTile - 1) / (4 * kTile), (n_users + kTile - 1) / kTile);
    hipLaunchKernelGGL(scores_kernel, grid, dim3(256), lds, stream, m.P.ptr + static_cast<size_t>(user0) * m.ld, m.ld,
                       m.user_bias.ptr + user0, n_users, m.Q.ptr, m.ldq, m.item_bias.ptr, m.cols, m.global_bias, nslots, scores);
    CU2REC_HIP(hipGetLastError());
}

}  // namespace

// predict_ratings (predict.cu:17-30) for every user of the model: scores [rows x cols], device memory
void model_scores(const DeviceModel &m, float *scores_device, hipStream_t stream) {
    require(scores_device != nullptr, "cu2rec_model_scores: null output");
    if (m.rows == 0 || m.cols == 0) return;
    require_device();
    launch_scores(m, 0, m.rows, scores_device, stream);
}

// get_recommendations (predict.cu:49-65) for every user: the k best items the user has not rated, best first
void model_recommend(const DeviceModel &m, const DeviceCsr *rated, int k, int *items_out, float *scores_out) {
    require(k >= 1 && items_out && scores_out, "cu2rec_model_recommend: bad argument");
    require(!rated || (rated->rows <= m.rows && rated->max_item < m.cols), "cu2rec_model_recommend: ratings exceed the model's shape");
    if (m.rows == 0) return;
    require_device();
    require(m.cols > 0, "cu2rec_model_recommend: no items");
    hipStream_t stream = nullptr;
    // users are ranked in batches: two key and two value arrays of batch x cols each
    const size_t per_user = static_cast<size_t>(m.cols);
    const int batch = static_cast<int>(std::max<size_t>(1, std::min<size_t>(m.rows, (size_t(1) << 28) / per_user)));
    const size_t cap = static_cast<size_t>(batch) * per_user;
    DeviceBuffer<float> keys_a(cap), keys_b(cap), top_s(static_cast<size_t>(batch) * k);
    DeviceBuffer<int> items_a(cap), items_b(cap), offsets(static_cast<size_t>(batch) + 1), top_i(static_cast<size_t>(batch) * k);
    std::vector<int> h_off(static_cast<size_t>(batch) + 1);
    for (int u = 0; u <= batch; ++u) h_off[u] = static_cast<int>(u * per_user);
    offsets.upload(h_off.data(), h_off.size());
    size_t temp_bytes = 0;
    hipcub::DoubleBuffer<float> dk(keys_a.ptr, keys_b.ptr);
    hipcub::DoubleBuffer<int> dv(items_a.ptr, items_b.ptr);
    CU2REC_HIP(hipcub::DeviceSegmentedRadixSort::SortPairsDescending(nullptr, temp_bytes, dk, dv, static_cast<int>(cap), batch,
                                                                      offsets.ptr, offsets.ptr + 1, 0, 32, stream));
    DeviceBuffer<unsigned char> temp(temp_bytes + 16);
    for (int user0 = 0; user0 < m.rows; user0 += batch) {
        const int n = std::min(batch, m.rows - user0);
        const size_t total = static_cast<size_t>(n) * per_user;
        launch_scores(m, user0, n, keys_a.ptr, stream);
        hipLaunchKernelGGL(iota_items_kernel, dim3(static_cast<unsigned>(std::min<size_t>((total + 255) / 256, 8192))), dim3(256), 0,
                           stream, items_a.ptr, total, m.cols);
        if (rated && rated->nnz > 0 && user0 < rated->rows)
            hipLaunchKernelGGL(mask_rated_kernel, dim3(std::min(std::min(n, rated->rows - user0), 4096)), dim3(256), 0, stream,
                               rated->indptr.ptr, rated->indices.ptr, user0, std::min(n, rated->rows - user0), m.cols, keys_a.ptr);
        CU2REC_HIP(hipGetLastError());
        hipcub::DoubleBuffer<float> k2(keys_a.ptr, keys_b.ptr);
        hipcub::DoubleBuffer<int> v2(items_a.ptr, items_b.ptr);
        size_t bytes = temp_bytes;
        CU2REC_HIP(hipcub::DeviceSegmentedRadixSort::SortPairsDescending(temp.ptr, bytes, k2, v2, static_cast<int>(total), n,
                                                                          offsets.ptr, offsets.ptr + 1, 0, 32, stream));
        hipLaunchKernelGGL(take_top_kernel, dim3(std::min((n * k + 255) / 256, 4096)), dim3(256), 0, stream, k2.Current(),
                           v2.Current(), n, m.cols, k, top_s.ptr, top_i.ptr);
        CU2REC_HIP(hipGetLastError());
        CU2REC_HIP(hipMemcpyAsync(scores_out + static_cast<size_t>(user0) * k, top_s.ptr, static_cast<size_t>(n) * k * sizeof(float),
                                  hipMemcpyDeviceToHost, stream));
        CU2REC_HIP(hipMemcpyAsync(items_out + static_cast<size_t>(user0) * k, top_i.ptr, static_cast<size_t>(n) * k * sizeof(int),
                                  hipMemcpyDeviceToHost, stream));
        CU2REC_HIP(hipStreamSynchronize(stream));
        if (k2.Current() != keys_a.ptr) {  // the next batch's scores go where the sort left its input
            keys_a.swap(keys_b);
            items_a.swap(items_b);
        }
    }
}

}  // namespace cu2rec

using namespace cu2rec;

extern "C" {

int cu2rec_model_scores(const cu2rec_model *m, float *scores_device, void *stream) {
    return guarded([&] {
        require(m, "model is null");
        model_scores(unwrap(m), scores_device, static_cast<hipStream_t>(stream));
    });
}

int cu2rec_model_scores_host(const cu2rec_model *m, float *scores_host) {
    return guarded([&] {
        require(m && scores_host, "cu2rec_model_scores_host: null argument");
        const DeviceModel &dm = unwrap(m);
        const size_t n = static_cast<size_t>(dm.rows) * dm.cols;
        if (n == 0) return;
        DeviceBuffer<float> d(n);
        model_scores(dm, d.ptr, nullptr);
        d.download(scores_host, n);
    });
}

int cu2rec_model_recommend(const cu2rec_model *m, const cu2rec_csr *rated, int k, int *items_out, float *scores_out) {
    return guarded([&] {
        require(m, "model is null");
        model_recommend(unwrap(m), rated ? &unwrap(rated) : nullptr, k, items_out, scores_out);
    });
}

}  // extern "C"
