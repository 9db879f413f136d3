"""ctypes front end of the C++ multi-GPU driver (cu2rec_amd/csrc/sharded.cpp): one process per GPU, ratings sharded by
user, the item side reconciled by ONE RCCL all-reduce of the item deltas per period (include/cu2rec_amd.h,
"User-sharded training").  Everything that computes or communicates happens inside libcu2rec_amd.so; Python only
launches ranks, slices host arrays and hands the ncclUniqueId around.
"""
import ctypes as C
import os

import numpy as np

from . import api
from ._lib import CommInfo, Config, Hyper, TrainStats, check, lib

MERGES = {"mean": 0, "weighted": 1, "sum": 2, "adaptive": 3}


class ShardOptions(C.Structure):
    _fields_ = [("sync_every", C.c_int), ("merge", C.c_int)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)


class Comm:
    """cu2rec_comm: RCCL (rank 0's ncclUniqueId distributed by `share`), or a caller-supplied all-reduce (tests)."""

    def __init__(self, rank=0, nranks=1, share=None, allreduce=None):
        self.rank, self.nranks = int(rank), int(nranks)
        self._h = C.c_void_p()
        self._cb = None
        if allreduce is not None:
            self._cb = ALLREDUCE_FN(allreduce)
            check(lib().cu2rec_comm_from_callback(C.cast(self._cb, C.c_void_p), None, self.rank, self.nranks, C.byref(self._h)))
            return
        uid = (C.c_ubyte * 128)()
        if self.nranks == 1 and os.environ.get("CU2REC_RCCL_WORLD1") == "1":
            check(lib().cu2rec_comm_unique_id(uid))  # test aid: a real one-rank RCCL communicator
        if self.nranks > 1:
            if share is None:
                raise ValueError("a multi-rank RCCL communicator needs share(bytes_or_None) -> bytes to pass rank 0's id around")
            if self.rank == 0:
                check(lib().cu2rec_comm_unique_id(uid))
            got = share(bytes(uid) if self.rank == 0 else None)
            uid = (C.c_ubyte * 128).from_buffer_copy(got)
        check(lib().cu2rec_comm_create(uid, self.rank, self.nranks, C.byref(self._h)))

    def info(self):
        """cu2rec_comm_info: rank / nranks as created, and what RCCL reports (ncclCommCount, ncclCommUserRank, ncclCommCuDevice,
        ncclGetVersion) -- rccl_nranks 0 without an RCCL communicator (one rank, or the callback form)."""
        ci = CommInfo()
        check(lib().cu2rec_comm_info(self._h, C.byref(ci)))
        return {name: int(getattr(ci, name)) for name, _ in CommInfo._fields_}

    def close(self):
        if self._h:
            lib().cu2rec_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def share_through_torch(device=None):
    """share() for Comm on top of an initialised torch.distributed process group (broadcast from rank 0)."""
    import torch
    import torch.distributed as dist

    def share(uid):
        on_gpu = dist.get_backend() == "nccl"
        t = torch.zeros(128, dtype=torch.uint8, device=device if on_gpu else "cpu")
        if uid is not None:
            t.copy_(torch.frombuffer(bytearray(uid), dtype=torch.uint8))
        dist.broadcast(t, 0)
        return bytes(t.cpu().numpy().tobytes())
    return share


class ShardJob:
    """cu2rec_shard_job: one rank's share of a sharded run (model + train must outlive it)."""

    def __init__(self, comm, model, train, user_offset=0, sync_every=0, merge="adaptive"):
        self.comm, self.model, self.ratings = comm, model, train
        opt = ShardOptions(int(sync_every), MERGES[merge] if isinstance(merge, str) else int(merge))
        self._h = C.c_void_p()
        check(lib().cu2rec_shard_job_create(comm._h, model._h, train._h, int(user_offset), C.byref(opt), C.byref(self._h)))

    def run(self, hyper, seed, iter0, n_iters, mode=api.SGD_HOGWILD, update_items=True):
        h = hyper if isinstance(hyper, Hyper) else Hyper(*[float(x) for x in hyper])
        check(lib().cu2rec_shard_job_run(self._h, C.byref(h), int(seed), int(iter0), int(n_iters), api._mode(mode),
                                         1 if update_items else 0))

    def exchange(self):
        check(lib().cu2rec_shard_job_exchange(self._h))

    def loss(self, ratings):
        sa, ss, n, mae, rmse = C.c_double(), C.c_double(), C.c_double(), C.c_float(), C.c_float()
        check(lib().cu2rec_shard_job_loss(self._h, ratings._h, C.byref(sa), C.byref(ss), C.byref(n), C.byref(mae), C.byref(rmse)))
        return {"mae": mae.value, "rmse": rmse.value, "sum_abs": sa.value, "sum_sq": ss.value, "n": n.value}

    def info(self):
        se, ex, ut, nt, wb = C.c_int(), C.c_int(), C.c_double(), C.c_double(), C.c_size_t()
        check(lib().cu2rec_shard_job_info(self._h, C.byref(se), C.byref(ex), C.byref(ut), C.byref(nt), C.byref(wb)))
        return {"sync_every": se.value, "exchanges": ex.value, "users_total": ut.value, "nnz_total": nt.value,
                "wire_bytes": wb.value}

    def exchange_stats(self):
        """cu2rec_shard_job_exchange_stats: device time of the exchanges whose event pairs have completed (synchronise first)."""
        n, s, m = C.c_int(), C.c_double(), C.c_double()
        check(lib().cu2rec_shard_job_exchange_stats(self._h, C.byref(n), C.byref(s), C.byref(m)))
        return {"timed": n.value, "seconds": s.value, "max_seconds": m.value, "mean_seconds": (s.value / n.value) if n.value else None}

    def train(self, test, cfg, mode=api.SGD_HOGWILD, verbose=True):
        """cu2rec_train_sharded -> (losses, stats); cfg.learning_rate / cur_iterations updated like the reference's."""
        losses = np.empty(max(cfg.total_iterations, 1), np.float32)
        st = TrainStats()
        check(lib().cu2rec_train_sharded(self._h, test._h, C.byref(cfg), api._mode(mode), 1 if verbose else 0,
                                         losses.ctypes.data_as(C.c_void_p), C.byref(st)))
        return losses[:cfg.total_iterations], st

    def close(self):
        if self._h:
            lib().cu2rec_shard_job_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def plan_users(rows, nranks):
    return [int(v) for v in api.shard_plan(rows, nranks)]


def shard_of(train, test, rank, nranks):
    """This rank's contiguous user range of the full train / test sets -> (u0, u1, train slice, test slice)."""
    bounds = plan_users(train.rows, nranks)
    u0, u1 = bounds[rank], bounds[rank + 1]
    if test.rows < train.rows:  # a test file may name fewer users than train (mf.cu:50-51): pad with empty rows
        test = api.HostCSR(np.concatenate([test.indptr, np.full(train.rows - test.rows, test.nnz, np.int32)]),
                           test.indices, test.data, train.rows, train.cols, test.global_bias)
    return u0, u1, train.slice_users(u0, u1), test.slice_users(u0, u1)


def train_sharded(comm, train, test, cfg, mode=None, sync_every=0, merge="adaptive", verbose=True):
    """train() (training.h:12-15) over all ranks through the C++ driver.  train / test: the FULL HostCSR on every rank.
    mode None = api.default_mode(cfg.n_factors), as api.train, bin/mf and mf_mgpu (block-solve; `ordered` above 252 factors).
    Returns (P_local, Q, losses, user_bias_local, item_bias, (u0, u1), stats)."""
    u0, u1, tr, te = shard_of(train, test, comm.rank, comm.nranks)
    f = cfg.n_factors
    if mode is None:
        mode = api.default_mode(f)
    # every rank draws the reference's seed-42 initialisation and keeps its slice, so N ranks start exactly where
    # one rank would (training.cu:28,54,212-213)
    P0 = api.initialize_normal_array(train.rows * f, f).reshape(train.rows, f)[u0:u1]
    ub0 = api.initialize_normal_array(train.rows, f)[u0:u1]
    model = api.Model(u1 - u0, train.cols, f, train.global_bias, P=P0, user_bias=ub0)
    d_tr, d_te = api.DeviceCSR(tr), api.DeviceCSR(te)
    job = ShardJob(comm, model, d_tr, user_offset=u0, sync_every=sync_every, merge=merge)
    losses, stats = job.train(d_te, cfg, mode=mode, verbose=verbose)
    P, Q, ub, ib = model.download()
    job.close()
    return P, Q, losses, ub, ib, (u0, u1), stats
