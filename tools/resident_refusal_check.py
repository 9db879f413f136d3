#!/usr/bin/env python3
"""What happens when a resident Hogwild launch cannot be co-resident.  Run with a library built to launch twice the
co-resident grid (tools/build_variant.sh oversub "-DCU2REC_RES_TEST_OVERSUBSCRIBE=2"):
  CU2REC_AMD_LIB=build/variants/oversub/libcu2rec_amd.so python tools/resident_refusal_check.py
Expected: the cooperative launch is REFUSED at launch time, the call runs one streaming launch per iteration instead,
finishes in milliseconds (not after the barrier's 3 s timeout), no error is raised and the model trains."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import cu2rec_amd as cu
    from cu2rec_amd import synth
    from cu2rec_amd._lib import check, lib
    # as many users as the ML-20M shape: the resident plan then uses (nearly) every CU, so twice that grid cannot fit
    tr, te = synth.make_ratings(138_493, 3000, 3_500_000, min_degree=5, seed=4)
    f, hyper = 100, (0.01, 0.02, 0.02, 0.02, 0.02)
    d_tr, d_te = cu.DeviceCSR(tr), cu.DeviceCSR(te)
    lib().cu2rec_hogwild_resident(2)
    import ctypes
    blocks = ctypes.c_int(0)
    planned = lib().cu2rec_hogwild_resident_plan(tr.rows, f, 200, ctypes.byref(blocks), None)
    model = cu.Model(tr.rows, tr.cols, f, tr.global_bias)
    before = model.loss(d_te)["rmse"]
    t0 = time.perf_counter()
    model.sgd(d_tr, hyper, 42, 0, 200, mode="hogwild")
    check(lib().cu2rec_check_faults())
    dt = time.perf_counter() - t0
    after = model.loss(d_te)["rmse"]
    refused = lib().cu2rec_hogwild_resident_refusals()
    print("planned resident: %d (%d workgroups), refused at launch: %d, 200 iterations in %.3f s, test rmse %.4f -> %.4f" %
          (planned, blocks.value, refused, dt, before, after))
    oversub = "oversub" in os.environ.get("CU2REC_AMD_LIB", "")
    assert planned == 1 and np.isfinite(after) and after < before
    assert (refused >= 1 and dt < 1.0) if oversub else refused == 0
    print("ok")


if __name__ == "__main__":
    main()
