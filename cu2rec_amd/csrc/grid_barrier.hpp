// The grid-wide barrier of the persistent launches (resident.hip: Hogwild with the user rows in registers -- the one persistent
// launch left: the ordered walk's was removed in round 5): XCD-hierarchical, split in two halves, every spin bounded.
// MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility" ("barrier-xcd").  The functions take any
// argument block with .words (the barrier block below, zeroed before the launch) and .status (set by a wait that gave up).
#pragma once

#include <hip/hip_runtime.h>

#include "sgd_device.hpp"

#ifndef CU2REC_BARRIER_ABLATE
#define CU2REC_BARRIER_ABLATE 0  // timing-only builds: 64 / 128 barrier without its release / acquire fence
#endif

namespace cu2rec {
namespace gridbar {


constexpr unsigned long long kBarrierTimeoutTicks = 300000000ull;  // 3 s of the 100 MHz wall clock

// Barrier words, each on a 128-byte line of its own (indices into an array of unsigned, zeroed before every launch):
constexpr int kLine = 32;
constexpr int kMaxXcc = 16;
constexpr int kWTop = 0;                                  // XCD leaders that have released, summed over phases
constexpr int kWCensusTop = kLine;                        // workgroups that have reported their XCD
constexpr int kWCensus = 2 * kLine;                       // [x]: workgroups on XCD x
constexpr int kWArrive = kWCensus + kMaxXcc * kLine;      // [x]: arrivals on XCD x, summed over phases
constexpr int kWGen = kWArrive + kMaxXcc * kLine;         // [x]: last phase XCD x may leave
constexpr int kBarrierWords = kWGen + kMaxXcc * kLine;

struct BarrierShared {  // per workgroup, in LDS; written by thread 0 only
    int xcc, n_mine, n_xcds, leader, ok;
    int part, index;  // single-XCD launches: does this workgroup take part, and its number among those that do
};

__device__ __forceinline__ unsigned ld_relaxed(const unsigned *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_load sc1: never L1-served
}

// Spins until *p - target >= 0 (wrap-safe); false if the status word is set or 3 s have passed (then sets it).
__device__ __forceinline__ bool spin_until(const unsigned *p, unsigned target, unsigned *status) {
    const unsigned long long t0 = wall_clock64();
    unsigned polls = 0;
    while (static_cast<int>(ld_relaxed(p) - target) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if ((++polls & 63u) == 0 && (ld_relaxed(status) != 0 || wall_clock64() - t0 > kBarrierTimeoutTicks)) {
            __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
    return true;
}

// Once per launch, thread 0 of every workgroup: which XCD am I on, how many workgroups share it, how many XCDs
// are in use.  (Placement is the dispatcher's business; the barrier only needs the counts.)
template <class RA>
__device__ __forceinline__ void barrier_census(const RA &ra, BarrierShared *bs) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= kMaxXcc - 1;
    // The per-XCD count must be visible before the top count can reach gridDim.x: the second add is a RELEASE (it may
    // not overtake the first, which sits on another cache line), and a reader that has seen the full top count takes an
    // ACQUIRE before it reads the per-XCD counts.  With two relaxed adds a workgroup could see "everybody has arrived"
    // while the last per-XCD add was still in flight, undercount its XCD and open the barrier early.
    __hip_atomic_fetch_add(ra.words + kWCensus + xcc * kLine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(ra.words + kWCensusTop, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    bs->ok = spin_until(ra.words + kWCensusTop, gridDim.x, ra.status) ? 1 : 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    int n_xcds = 0;
    for (int x = 0; x < kMaxXcc; ++x) n_xcds += ld_relaxed(ra.words + kWCensus + x * kLine) != 0;
    bs->xcc = static_cast<int>(xcc);
    bs->n_mine = static_cast<int>(ld_relaxed(ra.words + kWCensus + xcc * kLine));
    bs->n_xcds = n_xcds;
}

// Grid barrier, split in two so that work which does not depend on other workgroups (the next iteration's sample
// gathers) runs while the chip drains.  XCD-hierarchical: the LAST workgroup to arrive on an XCD is its leader and
// does the one agent-scope release (L2 write-back) for that XCD -- every other workgroup of the XCD has drained its
// stores into that same L2 before it arrived -- then reports to the top counter, waits for all XCDs and opens its
// XCD's generation word; the others only poll that word.  Everybody ends with an agent-scope acquire (L1 invalidate).
template <class RA>
__device__ __forceinline__ void barrier_arrive(const RA &ra, unsigned phase, BarrierShared *bs) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's item-row stores have reached L2
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(ra.words + kWArrive + bs->xcc * kLine, 1u, __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT);
        const bool leader = old + 1 == phase * static_cast<unsigned>(bs->n_mine);
        if (leader) {
            if (!(CU2REC_BARRIER_ABLATE & 64))                        // 64: timing only, no L2 write-back
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // write back this XCD's dirty L2 lines
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (the compiler may drop its own wait here)
            __hip_atomic_fetch_add(ra.words + kWTop, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        bs->leader = leader ? 1 : 0;
        // The acquire half, early: invalidate this CU's L1 NOW, so that it overlaps the wait below.  Between here and the
        // end of barrier_wait no wave of this workgroup loads an item row or item bias (only the read-only sample arrays),
        // so nothing another CU rewrites can enter the L1 again before the barrier has passed, and the first item-row
        // loads after it miss the L1 and are served by the L2, which the XCD leaders' releases have made current.
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
}

// true: everybody arrived.  false: gave up (status word set); the caller leaves its iteration loop.
template <class RA>
__device__ __forceinline__ bool barrier_wait(const RA &ra, unsigned phase, BarrierShared *bs) {
    if (threadIdx.x == 0) {
        bool ok;
        unsigned *gen = ra.words + kWGen + bs->xcc * kLine;
        if (bs->leader) {
            ok = spin_until(ra.words + kWTop, phase * static_cast<unsigned>(bs->n_xcds), ra.status);
            __hip_atomic_store(gen, phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            ok = spin_until(gen, phase, ra.status);
        }
        if (!(CU2REC_BARRIER_ABLATE & 128))                         // 128: timing only, no L1 invalidate
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the invalidate issued in barrier_arrive has completed
        bs->ok = ok ? 1 : 0;
    }
    __syncthreads();
    return bs->ok != 0;
}


}  // namespace gridbar
}  // namespace cu2rec
