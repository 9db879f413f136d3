// Resident Hogwild SGD: one persistent launch runs many reference iterations with the users' rows held in
// the register file and a grid-wide barrier where the reference has its kernel boundary; see resident.hip.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "kernels.hpp"

namespace cu2rec {

enum ResidentPolicy { kResidentOff = 0, kResidentAuto = 1, kResidentForce = 2 };

// Process-wide policy: CU2REC_RESIDENT=0|1|2 in the environment at first use, or cu2rec_hogwild_resident().
int resident_policy(int set_to /* < 0: query only */);

// Runs iterations [iter0, iter0 + n_iters) of `a` (pointers, hyper-parameters, seed, user_offset filled in) in
// ONE launch if the policy allows it and every user row of the CSR fits the register file; returns false
// (nothing launched) otherwise and the caller falls back to one streaming launch per iteration.
bool resident_launch(SgdArgs a, uint64_t iter0, int n_iters, hipStream_t stream);

// Would a call with this shape be ONE resident launch on the current device under the current policy?  If so,
// *blocks = workgroups (one per CU) and *users_per_group = rows each 16-lane group keeps in registers.
bool resident_plan(int n_rows, int n_factors, int n_iters, int *blocks, int *users_per_group);

// The arithmetic behind resident_plan, device independent: geometry for n_rows users on n_cus CUs, or false if the rows
// do not fit registers + LDS (policy and call length are not considered).
bool resident_geometry(int n_rows, int n_factors, int n_cus, int *blocks, int *users_per_group, int *lds_rows);
// rows per group that stream in that geometry (partial residency): 0 = fully resident, -1 = no compiled form holds the set
int resident_streamed_rows(int n_rows, int n_factors, int n_cus);

// Launches the runtime refused on the current device because the grid could not be co-resident (those calls streamed).
int resident_refusals();

// Throws if an earlier resident launch on the current device gave up at a grid barrier (bounded spin).
void resident_check_fault();

}  // namespace cu2rec
