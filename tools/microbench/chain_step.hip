// Timing-only microbenchmark of the two-wave chain's COMPUTE wavefront (tools/, never part of the library): how many nanoseconds per
// update does one wavefront need for (a) the plain dependent step of ordered.hip's duo_step_a and (b) a depth-8 look-ahead tile
// (eight dots with the row as it was at the tile's start, forward substitution through a pre-scaled Gram triangle read from LDS,
// progressive row update), with rows / Gram entries already in LDS?  One wavefront per workgroup, W = 32 lanes per chain (two chains
// per wavefront), tiles of 8 updates read round-robin from 4 LDS tile buffers.  Prints ns per update and a checksum.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I cu2rec_amd/csrc -I include tools/microbench/chain_step.hip -o build/chain_step
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "sgd_device.hpp"

using namespace cu2rec::dev;
typedef unsigned int uint2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float cross_row_sum(float v) {
    const uint2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float full_dot(const float4 &x, const float4 &y) { return row_sum16(cross_row_sum(slot_dot(x, y))); }

constexpr int T = 8, NB = 4;
struct Lds {
    float4 p[NB][T][32];
    float4 qold[NB][T][32];
    float g[NB][T][T];  // g[j][t], j < t, pre-scaled by A^(t-1-j)
    float base[NB][T], err[NB][T];
};

template <int MODE>
__global__ __launch_bounds__(64) void chain_kernel(const float4 *rows, int n_tiles, float lr, float reg, float *out) {
    __shared__ Lds lds[2];
    const int lane = threadIdx.x & 31, c = threadIdx.x >> 5;
    Lds &l = lds[c];
    for (int b = 0; b < NB; ++b)
        for (int t = 0; t < T; ++t) {
            l.p[b][t][lane] = rows[(b * T + t) * 32 + lane];
            if (lane < T) l.g[b][lane][t] = 0.001f * (lane + t + 1);
            if (lane == 0) l.base[b][t] = 0.01f * (t + 1);
        }
    __syncthreads();
    float4 q = rows[lane];
    float ib = 0.1f;
    const float A = 1.f - lr * reg, C = 1.f - lr * reg;
    for (int k = 0; k < n_tiles; ++k) {
        const int b = k & (NB - 1);
        float4 po[T];
        float base[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            po[t] = l.p[b][t][lane];
            base[t] = l.base[b][t];
        }
        if (MODE == 0) {  // the plain step (duo_step_a)
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const float dot = full_dot(q, po[t]);
                const float err = base[t] - (ib + dot);
                l.qold[b][t][lane] = q;
                if (lane == 0) l.err[b][t] = err;
                q.x = step(q.x, po[t].x, err, lr, reg);
                q.y = step(q.y, po[t].y, err, lr, reg);
                q.z = step(q.z, po[t].z, err, lr, reg);
                q.w = step(q.w, po[t].w, err, lr, reg);
                ib = ib + lr * (err - reg * ib);
            }
        } else {  // depth-8 look-ahead tile
            float g[T][T];
#pragma unroll
            for (int j = 0; j < T; ++j)
#pragma unroll
                for (int t = j + 1; t < T; ++t) g[j][t] = l.g[b][j][t];
            float acc[T];
            // eight independent dots with the row as it is now, written so that they interleave
            float part[T];
#pragma unroll
            for (int t = 0; t < T; ++t) part[t] = __builtin_fmaf(q.x, po[t].x, 0.f);
#pragma unroll
            for (int t = 0; t < T; ++t) part[t] = __builtin_fmaf(q.y, po[t].y, part[t]);
#pragma unroll
            for (int t = 0; t < T; ++t) part[t] = __builtin_fmaf(q.z, po[t].z, part[t]);
#pragma unroll
            for (int t = 0; t < T; ++t) part[t] = __builtin_fmaf(q.w, po[t].w, part[t]);
#pragma unroll
            for (int t = 0; t < T; ++t) part[t] = cross_row_sum(part[t]);
#pragma unroll
            for (int t = 0; t < T; ++t) part[t] = part[t] + dpp_move<0xB1>(part[t]);
#pragma unroll
            for (int t = 0; t < T; ++t) part[t] = part[t] + dpp_move<0x4E>(part[t]);
#pragma unroll
            for (int t = 0; t < T; ++t) part[t] = part[t] + dpp_move<0x141>(part[t]);
#pragma unroll
            for (int t = 0; t < T; ++t) part[t] = part[t] + dpp_move<0x140>(part[t]);
            float apow = 1.f;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                acc[t] = apow * part[t];
                apow *= A;
            }
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const float err = (base[t] - ib) - acc[t];
                l.qold[b][t][lane] = q;
                if (lane == 0) l.err[b][t] = err;
                const float w = lr * err;
#pragma unroll
                for (int u = t + 1; u < T; ++u) acc[u] = __builtin_fmaf(w, g[t][u], acc[u]);
                q.x = __builtin_fmaf(w, po[t].x, A * q.x);
                q.y = __builtin_fmaf(w, po[t].y, A * q.y);
                q.z = __builtin_fmaf(w, po[t].z, A * q.z);
                q.w = __builtin_fmaf(w, po[t].w, A * q.w);
                ib = __builtin_fmaf(C, ib, w);
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = q.x + q.y + q.z + q.w + ib;
}


// ---- MODE 2: the split form.  Wavefront 0 (A) never touches rows: per tile it reads the eight dots s_t = q0 . p_t that the row wavefront
// left in LDS, the pre-scaled Gram triangle and the bases, runs the forward substitution and leaves errors / weights in LDS.  Wavefront 3
// (H) owns the item row: behind the barrier it advances q through the tile's eight updates (publishing q as it was in front of each, for the
// user side), then computes the next tile's dots -- NDOTW of the workgroup's wavefronts share those dots (H keeps the first share).
template <int NDOTW>
__global__ __launch_bounds__(256) void split_kernel(const float4 *rows, int n_tiles, float lr, float reg, float *out) {
    __shared__ Lds lds[2];
    __shared__ float s_dot[2][2][T];     // [chain][parity][t]
    __shared__ float4 s_q0[2][32];        // the row at the tile's start, for the wavefronts that share the dots
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 31, c = (threadIdx.x & 63) >> 5;
    Lds &l = lds[c];
    if (wave == 0)
        for (int b = 0; b < NB; ++b)
            for (int t = 0; t < T; ++t) {
                l.p[b][t][lane] = rows[(b * T + t) * 32 + lane];
                if (lane < T) l.g[b][lane][t] = 0.001f * (lane + t + 1);
                if (lane == 0) l.base[b][t] = 0.01f * (t + 1);
            }
    float4 q = rows[lane];
    float ib = 0.1f;
    const float A = 1.f - lr * reg, C = 1.f - lr * reg;
    __syncthreads();
    // dots of tile 0
    auto dots = [&](int b, int par, const float4 &q0, int t_lo, int t_hi) {
        float part[T];
#pragma unroll
        for (int t = 0; t < T; ++t) part[t] = 0.f;
#pragma unroll
        for (int t = 0; t < T; ++t)
            if (t >= t_lo && t < t_hi) {
                const float4 po = l.p[b][t][lane];
                part[t] = slot_dot(q0, po);
            }
#pragma unroll
        for (int t = 0; t < T; ++t) if (t >= t_lo && t < t_hi) part[t] = cross_row_sum(part[t]);
#pragma unroll
        for (int t = 0; t < T; ++t) if (t >= t_lo && t < t_hi) part[t] = part[t] + dpp_move<0xB1>(part[t]);
#pragma unroll
        for (int t = 0; t < T; ++t) if (t >= t_lo && t < t_hi) part[t] = part[t] + dpp_move<0x4E>(part[t]);
#pragma unroll
        for (int t = 0; t < T; ++t) if (t >= t_lo && t < t_hi) part[t] = part[t] + dpp_move<0x141>(part[t]);
#pragma unroll
        for (int t = 0; t < T; ++t) if (t >= t_lo && t < t_hi) part[t] = part[t] + dpp_move<0x140>(part[t]);
#pragma unroll
        for (int t = 0; t < T; ++t) if (t >= t_lo && t < t_hi && lane == 0) s_dot[c][par][t] = part[t];
    };
    constexpr int kShare = (T + NDOTW - 1) / NDOTW;
    if (wave == 3) {
        s_q0[c][lane] = q;
    }
    __syncthreads();
    {
        const int w = wave == 3 ? 0 : wave;  // dot share index: H = 0, waves 1, 2 = 1, 2
        if ((wave == 3 || (wave >= 1 && wave < NDOTW)) ) dots(0, 0, s_q0[c][lane], w * kShare, min(T, (w + 1) * kShare));
    }
    __syncthreads();
    for (int k = 0; k < n_tiles; ++k) {
        const int b = k & (NB - 1), par = k & 1;
        if (wave == 0) {  // A: forward substitution on scalars
            float acc[T], base[T], g[T][T];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                acc[t] = s_dot[c][par][t];
                base[t] = l.base[b][t];
            }
#pragma unroll
            for (int j = 0; j < T; ++j)
#pragma unroll
                for (int t = j + 1; t < T; ++t) g[j][t] = l.g[b][j][t];
            float apow = 1.f;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                acc[t] = apow * acc[t];
                apow *= A;
            }
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const float err = (base[t] - ib) - acc[t];
                const float w = lr * err;
                if (lane == 0) l.err[b][t] = err;
#pragma unroll
                for (int u = t + 1; u < T; ++u) acc[u] = __builtin_fmaf(w, g[t][u], acc[u]);
                ib = __builtin_fmaf(C, ib, w);
            }
        }
        __syncthreads();
        if (wave == 3) {  // H: the row through the tile's updates, then it is the next tile's starting row
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const float w = lr * l.err[b][t];
                const float4 po = l.p[b][t][lane];
                l.qold[b][t][lane] = q;
                q.x = __builtin_fmaf(w, po.x, A * q.x);
                q.y = __builtin_fmaf(w, po.y, A * q.y);
                q.z = __builtin_fmaf(w, po.z, A * q.z);
                q.w = __builtin_fmaf(w, po.w, A * q.w);
            }
            if (NDOTW > 1) s_q0[c][lane] = q;
        }
        if (NDOTW > 1) __syncthreads();
        {
            const int nb = (k + 1) & (NB - 1), npar = (k + 1) & 1;
            const int w = wave == 3 ? 0 : wave;
            if (wave == 3 || (wave >= 1 && wave < NDOTW)) dots(nb, npar, NDOTW > 1 ? s_q0[c][lane] : q, w * kShare, min(T, (w + 1) * kShare));
        }
        __syncthreads();
    }
    if (wave == 3) out[blockIdx.x * 64 + (threadIdx.x & 63)] = q.x + q.y + q.z + q.w;
    if (wave == 0) out[4096 * 32 + blockIdx.x] = ib;
}

template <int NDOTW>
void run_split(const char *name, const float4 *d_rows, float *d_out, int blocks, int n_tiles) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(split_kernel<NDOTW>, dim3(blocks), dim3(256), 0, 0, d_rows, n_tiles, 0.01f, 0.02f, d_out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(split_kernel<NDOTW>, dim3(blocks), dim3(256), 0, 0, d_rows, n_tiles, 0.01f, 0.02f, d_out);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    float sum = 0;
    (void)hipMemcpy(&sum, d_out, sizeof(float), hipMemcpyDeviceToHost);
    printf("%-10s blocks %4d  %7.1f ns per update  (%.3f ms for %d updates, checksum %g)\n", name, blocks, 1e6 * ms / (n_tiles * T), ms, n_tiles * T, sum);
}

template <int MODE>
void run(const char *name, const float4 *d_rows, float *d_out, int blocks, int n_tiles) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(chain_kernel<MODE>, dim3(blocks), dim3(64), 0, 0, d_rows, n_tiles, 0.01f, 0.02f, d_out);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(chain_kernel<MODE>, dim3(blocks), dim3(64), 0, 0, d_rows, n_tiles, 0.01f, 0.02f, d_out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    float sum = 0;
    hipMemcpy(&sum, d_out, sizeof(float), hipMemcpyDeviceToHost);
    printf("%-10s blocks %4d  %7.1f ns per update  (%.3f ms for %d updates, checksum %g)\n", name, blocks, 1e6 * ms / (n_tiles * T), ms, n_tiles * T, sum);
}

int main() {
    std::vector<float4> rows(NB * T * 32);
    for (size_t i = 0; i < rows.size(); ++i) rows[i] = make_float4(0.01f * (i % 7), -0.02f * (i % 5), 0.015f * (i % 3), 0.01f);
    float4 *d_rows;
    float *d_out;
    hipMalloc(&d_rows, rows.size() * sizeof(float4));
    hipMalloc(&d_out, 4096 * 64 * sizeof(float));
    hipMemcpy(d_rows, rows.data(), rows.size() * sizeof(float4), hipMemcpyHostToDevice);
    for (int blocks : {1, 256, 1024}) {
        run<0>("plain", d_rows, d_out, blocks, 4096);
        run<1>("lookahead8", d_rows, d_out, blocks, 4096);
        run_split<1>("split/1", d_rows, d_out, blocks, 4096);
        run_split<3>("split/3", d_rows, d_out, blocks, 4096);
    }
    return 0;
}
