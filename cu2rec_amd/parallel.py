"""User-sharded training across the GPUs of one node: one process per GPU, torch.distributed
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

from . import api


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def plan_users(n_users, world_size):
    """Contiguous user ranges of (almost) equal size -- one SGD iteration is one update per user."""
    return [int(v) for v in api.shard_plan(n_users, world_size)]


class ShardedSGD:
    def __init__(self, engine, ratings, user_offset=0, sync_every=100, merge="mean", group=None):
        if merge not in ("mean", "sum"):
            raise ValueError("merge must be 'mean' or 'sum'")
        self.engine, self.ratings, self.user_offset = engine, ratings, int(user_offset)
        self.sync_every, self.merge, self.group = max(int(sync_every), 1), merge, group
        self.rank, self.world_size = world()
        self.since_sync = 0
        self.exchanges = 0
        if self.world_size > 1:
            engine.snapshot_items()

    def exchange(self):
        """All-reduce the item-factor deltas and rebase every replica on the merged result."""
        self.since_sync = 0
        if self.world_size == 1:
            return
        buf = self.engine.pack_item_delta()
        if buf.is_cuda and dist.get_backend(self.group) != "nccl":
            # debugging aid (e.g. two ranks sharing one GPU under gloo): stage through the host
            host = buf.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            buf.copy_(host)
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
        self.engine.apply_item_delta(1.0 / self.world_size if self.merge == "mean" else 1.0)
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
            if self.since_sync >= self.sync_every and update_items:
                self.exchange()
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
