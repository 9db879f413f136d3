#!/usr/bin/env python3
"""One call of the fused loss pass (cu2rec_loss through the object layer) on the ML-20M shape: host time of the call, device time
between two events around it, and the call right behind 50 SGD iterations."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, torch
import cu2rec_amd as cu, bench
tr, te = bench.load_dataset("ml-20m", 20240917, 0, lambda: None)
d = cu.DeviceCSR(tr); m = cu.Model(tr.rows, tr.cols, 100, tr.global_bias)
m.loss(d); torch.cuda.synchronize()
ts=[]; dev=[]
for _ in range(20):
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    t0=time.perf_counter(); e0.record(); r=m.loss(d); e1.record(); ts.append(time.perf_counter()-t0)
    torch.cuda.synchronize(); dev.append(e0.elapsed_time(e1))
print("loss call ms: host min %.3f median %.3f ; device (events) min %.3f median %.3f" % (1e3*min(ts), 1e3*np.median(ts), min(dev), np.median(dev)))
# keep the GPU busy right before (as inside a training loop): 50 hogwild iterations then loss
hyper=(0.01,0.02,0.02,0.02,0.02)
ts=[]
for i in range(10):
    m.sgd(d, hyper, 42, i*50, 50, mode="hogwild")
    t0=time.perf_counter(); r=m.loss(d); ts.append(time.perf_counter()-t0)
print("loss right behind 50 SGD iterations (includes draining them): min %.3f median %.3f" % (1e3*min(ts), 1e3*np.median(ts)))
