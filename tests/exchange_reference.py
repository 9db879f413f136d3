"""TEST INFRASTRUCTURE: the reference implementation of the user-sharded exchange algebra, in Python over torch.distributed.

The PRODUCT's multi-GPU driver is C++ (cu2rec_amd/csrc/sharded.cpp: ncclAllReduce of the item deltas, cu2rec_train_sharded,
`bin/mf -g N`; Python launcher cu2rec_amd/sharded.py).  This module was round 1's driver; it is kept here, outside the package,
as the executable statement of the same algebra that (a) the world-2 gloo CPU tests run against the CPU oracle with a host
stand-in engine (tests/test_parallel_cpu.py) and (b) one GPU test uses to drive the product's pack / apply kernels through a
real world-1 RCCL all-reduce (tests/test_gpu_parity.py).  Nothing under cu2rec_amd/, bench.py or the CLIs imports it.

User-sharded training across the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI; "gloo" in the CPU tests).

The reference is single-GPU.  The path shards by user: P rows, user
biases, CSR rows and the sampler stream are keyed by user and touched only by that user's
update; Q and item_bias are shared.  Each rank owns a contiguous user range [u0, u1) -- its CSR
slice, its P / user_bias slice -- and a full replica of Q / item_bias.  Every `sync_every`
iterations the replicas are reconciled with ONE all-reduce of the fused buffer
    [Q - Q_base | item_bias - item_bias_base]          (n_items * (ld + 1) floats)
followed by  Q <- Q_base + scale * sum_of_deltas  (scale = 1/N "mean", or 1 "sum"), which also
becomes the next Q_base.  There is no other data-path collective: loss partial sums are three
doubles.  The exchange kernels are in libcu2rec_amd.so (cu2rec_items_delta_pack / _apply).

`engine` is duck-typed (sgd, snapshot_items, pack_item_delta, apply_item_delta, loss): the product
engine is cu2rec_amd.engine.Engine (HIP); the CPU gloo tests inject an oracle-backed stand-in.
"""
import torch
import torch.distributed as dist

from cu2rec_amd import api


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def plan_users(n_users, world_size):
    """Contiguous user ranges of (almost) equal size -- one SGD iteration is one update per user."""
    return [int(v) for v in api.shard_plan(n_users, world_size)]


class ShardedSGD:
    def __init__(self, engine, ratings, user_offset=0, sync_every=100, merge="mean", group=None, item_rates=None,
                 overlap=None):
        """merge: how the replicas' item deltas are combined at an exchange --
             "mean"      delta = (1/N) sum_k delta_k       right when every rank sees every item about equally often
             "sum"       delta = sum_k delta_k             right when every item is updated by (almost) one rank only
             "weighted"  delta[y] = sum_k w_k[y] delta_k[y], w_k[y] = rate_k[y] / sum_j rate_j[y] with rate_k[y] the
                         expected updates per iteration of rank k's users on item y (item_rates, a float64 array of
                         n_items from api.item_update_rates(shard)): a per-item average that leaves items touched by
                         one rank at full step and averages the shared ones."""
        if merge not in ("mean", "sum", "weighted"):
            raise ValueError("merge must be 'mean', 'sum' or 'weighted'")
        self.engine, self.ratings, self.user_offset = engine, ratings, int(user_offset)
        self.sync_every, self.merge, self.group = max(int(sync_every), 1), merge, group
        self.rank, self.world_size = world()
        self.since_sync = 0
        self.exchanges = 0
        self.item_weight = None
        # overlap=True: the all-reduce of period t runs while period t+1 trains; its result is folded in at the next
        # sync point (item deltas arrive one period late, local progress made meanwhile is kept).  Default: on for
        # RCCL with the streaming SGD kernel, where the collective runs on its own stream next to the SGD launches;
        # OFF when Hogwild calls may be resident launches (cu2rec_hogwild_resident != 0, the default): a persistent
        # grid that needs every CU and an RCCL kernel that needs its peers' kernels running must never wait for
        # each other's CUs, so there the exchange stays stream-ordered between two launches (it costs a fraction of
        # a millisecond per epoch; residency saves half of every iteration).  CU2REC_EXCHANGE_OVERLAP=0/1 overrides.
        import os
        if overlap is None:
            env = os.environ.get("CU2REC_EXCHANGE_OVERLAP")
            if env in ("0", "1"):
                overlap = env == "1"
            else:
                overlap = self.world_size > 1 and dist.get_backend(group) == "nccl"
                if overlap:
                    from cu2rec_amd._lib import lib
                    overlap = lib().cu2rec_hogwild_resident(-1) == 0
        self.overlap = bool(overlap) and self.world_size > 1 and hasattr(engine, "apply_item_delta_overlapped")
        self._pending = None
        if self.world_size > 1:
            engine.snapshot_items()
            if merge == "weighted":
                if item_rates is None:
                    raise ValueError("merge='weighted' needs item_rates")
                mine = torch.as_tensor(item_rates, dtype=torch.float64)
                total = mine.clone()
                dev = getattr(engine, "device", None)
                if dev is not None and dist.get_backend(group) == "nccl":
                    total = total.to(dev)
                dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
                total = total.cpu()
                w = torch.where(total > 0, mine / total.clamp_min(1e-300), torch.full_like(mine, 1.0 / self.world_size))
                self.item_weight = w.to(torch.float32)
                if dev is not None:
                    self.item_weight = self.item_weight.to(dev)

    def _scale(self):
        return 1.0 / self.world_size if self.merge == "mean" else 1.0  # weighted / sum: 1

    def finish_pending(self):
        """Fold in an all-reduce started by an earlier overlapped exchange (no-op if there is none)."""
        if self._pending is None:
            return
        work, host = self._pending
        work.wait()
        if host is not None:
            self.engine.exchange.copy_(host)
        self.engine.apply_item_delta_overlapped(self._scale())
        self._pending = None

    def exchange(self, final=False):
        """All-reduce the item-factor deltas and rebase every replica on the merged result.  With overlap the
        collective is only STARTED here (and the previous one folded in); final=True drains it, so that every
        replica holds the same item factors afterwards (used before a loss evaluation and at the end)."""
        self.since_sync = 0
        if self.world_size == 1:
            return
        if self.overlap:
            self.finish_pending()
            w = self.item_weight if self.merge == "weighted" else None
            buf = self.engine.pack_item_delta(w) if w is not None else self.engine.pack_item_delta()
            self.engine.snapshot_for_overlap()
            host = None
            if buf.is_cuda and dist.get_backend(self.group) != "nccl":
                host = buf.cpu()
                work = dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            else:
                work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._pending = (work, host)
            self.exchanges += 1
            if final:
                self.finish_pending()
            return
        buf = self.engine.pack_item_delta(self.item_weight) if self.merge == "weighted" else self.engine.pack_item_delta()
        if buf.is_cuda and dist.get_backend(self.group) != "nccl":
            # debugging aid (e.g. two ranks sharing one GPU under gloo): stage through the host
            host = buf.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            buf.copy_(host)
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
        self.engine.apply_item_delta(self._scale())
        self.since_sync = 0
        self.exchanges += 1

    def run(self, hyper, seed, iter0, n_iters, mode=api.SGD_HOGWILD, update_items=True):
        """n_iters iterations on the local shard, exchanging every sync_every iterations (the cadence
        runs across calls).  Same sampler stream as the unsharded run: draws are keyed by global user id."""
        done = 0
        while done < n_iters:
            n = min(n_iters - done, self.sync_every - self.since_sync)
            self.engine.sgd(self.ratings, hyper, seed, iter0 + done, n, mode, update_items, self.user_offset)
            done += n
            self.since_sync += n
            if self.since_sync >= self.sync_every:
                # the period ends whether or not there is anything to exchange: with frozen items (is_train == false)
                # no replica has moved, but the counter must still start over or n stays 0 forever
                if update_items:
                    self.exchange()
                else:
                    self.since_sync = 0
        return done

    def loss(self, ratings):
        """Global MAE / RMSE over all shards: all-reduce of {sum |e|, sum e^2, n}."""
        out = self.engine.loss(ratings)
        t = torch.tensor([out["sum_abs"], out["sum_sq"], float(ratings.nnz)], dtype=torch.float64)
        if self.world_size > 1:
            dev = getattr(self.engine, "device", None)
            if dev is not None and dist.get_backend(self.group) == "nccl":
                t = t.to(dev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            t = t.cpu()
        sa, ss, n = (float(v) for v in t)
        return {"mae": sa / n, "rmse": (ss / n) ** 0.5, "sum_abs": sa, "sum_sq": ss, "n": n}


def train_sharded(train, test, cfg, mode=api.SGD_HOGWILD, sync_every=0, merge="mean", verbose=True, device=None,
                  engine_factory=None):
    """train() (training.h:14-15) over all ranks of the default process group: the same observable schedule as
    cu2rec_train -- loss on train and test at i == 0, every check_error and last (training.cu:118), the TRAIN:/TEST:
    lines (rank 0), patience / learning-rate decay on the GLOBAL test RMSE (training.cu:146-155; identical on every
    rank because the loss sums are all-reduced), cfg.learning_rate / cfg.cur_iterations updated in place.

    train / test: the FULL HostCSR on every rank (each rank slices its own contiguous user range).
    Returns (P_local, Q, losses, user_bias_local, item_bias, (u0, u1)): the rank's user slice plus the merged item
    side.  sync_every == 0 means one epoch = nnz / users iterations."""
    import time

    import numpy as np

    rank, world_size = world()
    bounds = plan_users(train.rows, world_size)
    u0, u1 = bounds[rank], bounds[rank + 1]
    if test.rows < train.rows:  # a test file may name fewer users than train (mf.cu:50-51): pad with empty rows
        test = api.HostCSR(np.concatenate([test.indptr, np.full(train.rows - test.rows, test.nnz, np.int32)]),
                           test.indices, test.data, train.rows, train.cols, test.global_bias)
    tr, te = train.slice_users(u0, u1), test.slice_users(u0, u1)
    f = cfg.n_factors
    # every rank draws the reference's seed-42 initialisation and keeps its slice, so N ranks start exactly where
    # one rank would (training.cu:28,54,212-213)
    P0 = api.initialize_normal_array(train.rows * f, f).reshape(train.rows, f)[u0:u1]
    ub0 = api.initialize_normal_array(train.rows, f)[u0:u1]
    if engine_factory is None:
        from cu2rec_amd.engine import DeviceRatings, Engine
        eng = Engine(u1 - u0, train.cols, f, train.global_bias, P=P0, user_bias=ub0, device=device)
        d_tr, d_te = DeviceRatings(tr, eng.device), DeviceRatings(te, eng.device)
    else:
        eng, d_tr, d_te = engine_factory(u1 - u0, train.cols, f, train.global_bias, P0, ub0, tr, te)
    users_active = float(np.count_nonzero(np.diff(train.indptr)))
    every = sync_every or max(1, int(round(train.nnz / max(users_active, 1.0))))
    rates = api.item_update_rates(tr) if merge == "weighted" else None
    job = ShardedSGD(eng, d_tr, user_offset=u0, sync_every=every, merge=merge, item_rates=rates)

    total = cfg.total_iterations
    losses = np.full(max(total, 1), np.nan, np.float32)
    validation_rmse = float(np.finfo(np.float32).max)
    patience = int(cfg.patience)
    iter_base, seed = int(cfg.cur_iterations), int(cfg.seed) & 0xFFFFFFFF
    t_start = time.perf_counter()
    i = 0
    while i < total:
        seg_end = i
        while not ((seg_end + 1) % cfg.check_error == 0 or seg_end == 0 or (seg_end + 1) % total == 0):
            seg_end += 1
        n = seg_end - i + 1
        hyper = (cfg.learning_rate, cfg.P_reg, cfg.Q_reg, cfg.user_bias_reg, cfg.item_bias_reg)
        job.run(hyper, seed, iter_base + i, n, mode, bool(cfg.is_train))
        if world_size > 1 and cfg.is_train:
            job.exchange(final=True)  # the loss below is taken on reconciled item factors
        tr_loss = job.loss(d_tr)
        last = validation_rmse
        te_loss = job.loss(d_te)
        validation_rmse = float(np.float32(te_loss["rmse"]))
        if verbose and rank == 0:
            print("TRAIN: Iteration %d GPU MAE: %f RMSE: %f" % (seg_end + 1, tr_loss["mae"], tr_loss["rmse"]))
            print("TEST: Iteration %d GPU MAE: %f RMSE: %f" % (seg_end + 1, te_loss["mae"], te_loss["rmse"]), flush=True)
        if last < validation_rmse:
            patience -= 1
        if patience <= 0:
            patience = int(cfg.patience)
            cfg.learning_rate = float(np.float32(cfg.learning_rate) * np.float32(cfg.learning_rate_decay))
            if verbose and rank == 0:
                print("New Learning Rate: %f\n: " % cfg.learning_rate, end="")
        losses[seg_end] = validation_rmse
        cfg.cur_iterations += n
        i = seg_end + 1
    if verbose and rank == 0:
        print("Time taken for %d of iterations is %f" % (total, time.perf_counter() - t_start))
    P, Q, ub, ib = eng.download()
    return P, Q, losses[:total], ub, ib, (u0, u1)
