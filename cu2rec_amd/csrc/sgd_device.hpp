// Device-side building blocks shared by the SGD / loss kernels (kernels.hip) and the ordered
// (deterministic, sequential-semantics) SGD schedule (ordered.hip).  Everything here is
// __device__ __forceinline__; translation units that include it must be compiled with
// -ffp-contract=off so that the only fused operations are the explicit __builtin_fmaf below.
#pragma once

#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace cu2rec {
namespace dev {

constexpr int kGroup = 16;                    // lanes per user: one DPP row
constexpr int kBlock = 256;                   // threads per block: 4 wavefronts, 16 groups
constexpr int kGroupsPerBlock = kBlock / kGroup;

// ---- cross-lane sum inside a 16-lane row -------------------------------------------------
template <int kCtrl>
__device__ __forceinline__ float dpp_move(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), kCtrl, 0xF, 0xF, false));
}

// xor butterfly 1, 2, 4, 8.  After the quad steps every lane of a quad holds the quad's sum,
// so mirroring inside 8 and then 16 lanes fetches exactly what the xor-4 / xor-8 partner holds.
__device__ __forceinline__ float row_sum16(float v) {
    v = v + dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
    v = v + dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
    v = v + dpp_move<0x141>(v);  // row_half_mirror
    v = v + dpp_move<0x140>(v);  // row_mirror
    return v;
}

// ---- factor rows -------------------------------------------------------------------------
template <int J>
struct Row {
    float4 v[J];
};

template <int J>
__device__ __forceinline__ Row<J> load_row(const float *__restrict__ base, size_t row, int ld, int nslots, int lane) {
    const float4 *p = reinterpret_cast<const float4 *>(base + row * static_cast<size_t>(ld));
    Row<J> r;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int slot = lane + kGroup * j;
        r.v[j] = slot < nslots ? p[slot] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return r;
}

template <int J>
__device__ __forceinline__ void store_row(float *__restrict__ base, size_t row, int ld, int nslots, int lane,
                                          const Row<J> &r) {
    float4 *p = reinterpret_cast<float4 *>(base + row * static_cast<size_t>(ld));
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int slot = lane + kGroup * j;
        if (slot < nslots) p[slot] = r.v[j];
    }
}

// One float4 slot's share of the dot product: a 4-term fmaf chain starting from +0.
__device__ __forceinline__ float slot_dot(const float4 &q, const float4 &p) {
    float s = __builtin_fmaf(q.x, p.x, 0.f);
    s = __builtin_fmaf(q.y, p.y, s);
    s = __builtin_fmaf(q.z, p.z, s);
    s = __builtin_fmaf(q.w, p.w, s);
    return s;
}

typedef float vec4f __attribute__((ext_vector_type(4)));  // the builtin wants a native vector type

// Streaming variants for rows that are touched once per launch (the user rows of the Hogwild kernel): the
// non-temporal hint keeps them from displacing the item rows, which ARE re-read, in the XCD's L2.
template <int J>
__device__ __forceinline__ Row<J> load_row_stream(const float *__restrict__ base, size_t row, int ld, int nslots, int lane) {
    const float4 *p = reinterpret_cast<const float4 *>(base + row * static_cast<size_t>(ld));
    Row<J> r;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int slot = lane + kGroup * j;
        if (slot < nslots) {
            const vec4f v = __builtin_nontemporal_load(reinterpret_cast<const vec4f *>(&p[slot]));
            r.v[j] = make_float4(v.x, v.y, v.z, v.w);
        } else {
            r.v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    return r;
}

template <int J>
__device__ __forceinline__ void store_row_stream(float *__restrict__ base, size_t row, int ld, int nslots, int lane,
                                                 const Row<J> &r) {
    float4 *p = reinterpret_cast<float4 *>(base + row * static_cast<size_t>(ld));
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int slot = lane + kGroup * j;
        if (slot < nslots) {
            const vec4f v = {r.v[j].x, r.v[j].y, r.v[j].z, r.v[j].w};
            __builtin_nontemporal_store(v, reinterpret_cast<vec4f *>(&p[slot]));
        }
    }
}

// util.cu:199-204 get_prediction in the kernels' canonical order ("TREE16" in the oracle):
//   slot partial  s_k   = fmaf chain over the 4 floats of slot k
//   lane total    t_l   = ((s_l + s_{l+16}) + s_{l+32}) + ...        (l = 0..15)
//   dot                 = xor butterfly 1, 2, 4, 8 over t_0..t_15
//   prediction          = ((gb + ub) + ib) + dot
// Slot partials are independent, so a layout with one slot per lane over 32 lanes (ordered.hip, hot chains)
// produces the same bits: s_l + s_{l+16} is then one cross-row add.
template <int J>
__device__ __forceinline__ float predict(const Row<J> &p, const Row<J> &q, float ub, float ib, float gb) {
    float acc = slot_dot(q.v[0], p.v[0]);
#pragma unroll
    for (int j = 1; j < J; ++j) acc = acc + slot_dot(q.v[j], p.v[j]);
    const float dot = row_sum16(acc);
    return ((gb + ub) + ib) + dot;
}

// mf_sequential.cu:133-136 on one float: new = old + lr * (err * other_old - reg * old)
__device__ __forceinline__ float step(float old, float other_old, float err, float lr, float reg) {
    return old + lr * (err * other_old - reg * old);
}

template <int J>
__device__ __forceinline__ void rank1_update(Row<J> &p, Row<J> &q, float err, const SgdHyper &h) {
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const float4 po = p.v[j], qo = q.v[j];
        p.v[j].x = step(po.x, qo.x, err, h.lr, h.p_reg);
        p.v[j].y = step(po.y, qo.y, err, h.lr, h.p_reg);
        p.v[j].z = step(po.z, qo.z, err, h.lr, h.p_reg);
        p.v[j].w = step(po.w, qo.w, err, h.lr, h.p_reg);
        q.v[j].x = step(qo.x, po.x, err, h.lr, h.q_reg);
        q.v[j].y = step(qo.y, po.y, err, h.lr, h.q_reg);
        q.v[j].z = step(qo.z, po.z, err, h.lr, h.q_reg);
        q.v[j].w = step(qo.w, po.w, err, h.lr, h.q_reg);
    }
}

}  // namespace dev
}  // namespace cu2rec
