"""Torch-tensor front end of the raw device-pointer C ABI.

PyTorch is plumbing here: it owns the device memory (so torch.distributed / RCCL can all-reduce
the item factors in place) and the stream.  Every computation is a call into libcu2rec_amd.so
on the tensors' data_ptr()s; nothing is computed with torch ops, and there is no CPU path --
constructing an Engine without a GPU raises.
"""
import numpy as np
import torch

from . import api
from ._lib import Hyper, lib


def _stream_ptr():
    return torch.cuda.current_stream().cuda_stream


class DeviceRatings:
    """CSR on the GPU in the reference layout (matrix.h:11-19) as three torch tensors."""

    def __init__(self, host, device):
        self.rows, self.cols, self.nnz = host.rows, host.cols, host.nnz
        if host.nnz and int(host.indices.max()) >= host.cols:
            raise ValueError("item id >= cols")
        self.users_with_ratings = int(np.count_nonzero(np.diff(host.indptr)))
        self.indptr = torch.from_numpy(host.indptr).to(device)
        self.indices = torch.from_numpy(host.indices if host.nnz else np.zeros(1, np.int32)).to(device)
        self.data = torch.from_numpy(host.data if host.nnz else np.zeros(1, np.float32)).to(device)
        self.global_bias = host.global_bias
        self._schedule = None
        self._pairs = None

    def sample_pairs(self):
        """The side-by-side {item, rating} sample array of this CSR (8 bytes per rating), built on first use."""
        if self._pairs is None and self.nnz:
            self._pairs = torch.empty(self.nnz, dtype=torch.int64, device=self.indices.device)
            api.sample_pairs_build(self.indices.data_ptr(), self.data.data_ptr(), self.nnz, self._pairs.data_ptr(),
                                   _stream_ptr())
        return self._pairs

    def schedule(self):
        """Ordered-mode workspace for this CSR, created on first use."""
        if self._schedule is None:
            self._schedule = api.Schedule(self.indptr.data_ptr(), self.indices.data_ptr(), self.rows, self.cols, self.nnz)
        return self._schedule


class Engine:
    """P, Q, user_bias, item_bias as padded torch tensors + the HIP hot path on them."""

    def __init__(self, rows, cols, n_factors, global_bias, P=None, Q=None, user_bias=None, item_bias=None,
                 device=None):
        if not torch.cuda.is_available() or api.device_count() < 1:
            raise RuntimeError("cu2rec_amd.Engine needs a GPU: the SGD / loss path has no CPU fallback")
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.rows, self.cols, self.f = int(rows), int(cols), int(n_factors)
        self.ld = (self.f + 3) & ~3     # user rows: 16-byte aligned, nothing more (they are streamed or resident)
        self.ldq = (self.f + 31) & ~31  # item rows: whole 128-byte lines, because several XCDs read AND write them and rows
        #                                 sharing a line cost coherence misses / partial write-backs
        self.global_bias = float(np.float32(global_bias))
        f = self.f

        def init(a, n, shape):  # training.cu:28,54,212-213: seed-42 normal init for whatever is not given
            if a is None:
                a = api.initialize_normal_array(n, f)
            return np.ascontiguousarray(a, np.float32).reshape(shape)

        self.P = self._padded(init(P, rows * f, (rows, f)), self.ld)
        self.Q = self._padded(init(Q, cols * f, (cols, f)), self.ldq)
        self.user_bias = torch.from_numpy(init(user_bias, rows, (rows,))).to(self.device)
        self.item_bias = torch.from_numpy(init(item_bias, cols, (cols,))).to(self.device)
        self.workspace = torch.empty(lib().cu2rec_loss_workspace_bytes() // 8, dtype=torch.float64, device=self.device)
        self.Q_base = None
        self.item_bias_base = None
        self.exchange = None

    def _padded(self, dense, ld):
        t = torch.zeros((max(dense.shape[0], 1), ld), dtype=torch.float32, device=self.device)
        if dense.shape[0]:
            t[:dense.shape[0], :self.f] = torch.from_numpy(dense).to(self.device)
        return t

    # ---- hot path -------------------------------------------------------------------------
    def sgd(self, ratings, hyper, seed, iter0, n_iters, mode=api.SGD_HOGWILD, update_items=True, user_offset=0):
        assert ratings.rows <= self.rows and ratings.cols <= self.cols
        if api._mode(mode) in (api.SGD_ORDERED, api.SGD_BLOCKSOLVE):
            fn = api.sgd_update_ordered if api._mode(mode) == api.SGD_ORDERED else api.sgd_update_blocksolve
            fn(ratings.schedule(), ratings.indptr.data_ptr(), ratings.indices.data_ptr(),
               ratings.data.data_ptr(), ratings.rows, self.cols, self.P.data_ptr(), self.ld,
               self.Q.data_ptr(), self.ldq, self.user_bias.data_ptr(), self.item_bias.data_ptr(),
               self.global_bias, self.f, hyper, seed, iter0, n_iters, update_items, user_offset, _stream_ptr())
            return
        pairs = None
        if (api._mode(mode) == api.SGD_HOGWILD and update_items and ratings.nnz and hasattr(ratings, "sample_pairs")
                and lib().cu2rec_hogwild_resident_plan(ratings.rows, self.f, n_iters, None, None) == 1):
            pairs = ratings.sample_pairs().data_ptr()  # resident launch: one 8-byte gather per draw
        api.sgd_update(ratings.indptr.data_ptr(), ratings.indices.data_ptr(), ratings.data.data_ptr(), ratings.rows,
                       self.cols, self.P.data_ptr(), self.ld, self.Q.data_ptr(), self.ldq, self.user_bias.data_ptr(),
                       self.item_bias.data_ptr(), self.global_bias, self.f, hyper, seed, iter0, n_iters, mode,
                       update_items, user_offset, _stream_ptr(), pairs)

    def loss(self, ratings, want_errors=False):
        assert ratings.rows <= self.rows and ratings.cols <= self.cols
        err = torch.empty(max(ratings.nnz, 1), dtype=torch.float32, device=self.device) if want_errors else None
        out = api.loss_raw(ratings.indptr.data_ptr(), ratings.indices.data_ptr(), ratings.data.data_ptr(),
                           ratings.rows, ratings.nnz, self.P.data_ptr(), self.ld, self.Q.data_ptr(), self.ldq,
                           self.user_bias.data_ptr(), self.item_bias.data_ptr(), self.global_bias, self.f,
                           self.workspace.data_ptr(), err.data_ptr() if want_errors else None, _stream_ptr())
        if want_errors:
            out["errors"] = err[:ratings.nnz].cpu().numpy()
        return out

    def error_metrics(self, errors):
        e = torch.as_tensor(np.ascontiguousarray(errors, np.float32)).to(self.device)
        return api.error_metrics_raw(e.data_ptr(), e.numel(), self.workspace.data_ptr(), _stream_ptr())

    def download(self):
        torch.cuda.synchronize(self.device)
        return (self.P[:self.rows, :self.f].cpu().numpy().copy(), self.Q[:self.cols, :self.f].cpu().numpy().copy(),
                self.user_bias[:self.rows].cpu().numpy().copy(), self.item_bias[:self.cols].cpu().numpy().copy())

    # ---- item-factor exchange (multi-GPU) ---------------------------------------------------
    def snapshot_items(self):
        self.Q_base = self.Q.clone()
        self.item_bias_base = self.item_bias.clone()
        self.exchange = torch.empty(self.Q.numel() + self.item_bias.numel(), dtype=torch.float32, device=self.device)

    def pack_item_delta(self, item_weight=None):
        """exchange <- [Q - Q_base | item_bias - item_bias_base] (device kernel), optionally times a per-item weight."""
        from ._lib import check
        if item_weight is not None:
            check(lib().cu2rec_items_delta_pack_weighted(self.Q.data_ptr(), self.item_bias.data_ptr(),
                                                         self.Q_base.data_ptr(), self.item_bias_base.data_ptr(),
                                                         item_weight.data_ptr(), self.item_bias.numel(), self.ldq,
                                                         self.exchange.data_ptr(), _stream_ptr()))
            return self.exchange
        check(lib().cu2rec_items_delta_pack(self.Q.data_ptr(), self.item_bias.data_ptr(), self.Q_base.data_ptr(),
                                            self.item_bias_base.data_ptr(), self.item_bias.numel(), self.ldq,
                                            self.exchange.data_ptr(), _stream_ptr()))
        return self.exchange

    def snapshot_for_overlap(self):
        """Remember Q / item_bias as they are now (called right after pack_item_delta, before training continues)."""
        if getattr(self, "Q_snap", None) is None:
            self.Q_snap, self.item_bias_snap = torch.empty_like(self.Q), torch.empty_like(self.item_bias)
        self.Q_snap.copy_(self.Q)
        self.item_bias_snap.copy_(self.item_bias)

    def apply_item_delta_overlapped(self, scale=1.0):
        """Q <- (Q_base + scale * exchange) + (Q - Q_snap); Q_base <- Q_base + scale * exchange (same for item_bias)."""
        from ._lib import check
        check(lib().cu2rec_items_delta_apply_overlapped(self.Q.data_ptr(), self.item_bias.data_ptr(), self.Q_base.data_ptr(),
                                                        self.item_bias_base.data_ptr(), self.Q_snap.data_ptr(),
                                                        self.item_bias_snap.data_ptr(), self.item_bias.numel(), self.ldq,
                                                        self.exchange.data_ptr(), float(scale), _stream_ptr()))

    def apply_item_delta(self, scale=1.0):
        """Q <- Q_base + scale * exchange_Q (same for item_bias); the result is the new snapshot."""
        from ._lib import check
        check(lib().cu2rec_items_delta_apply(self.Q.data_ptr(), self.item_bias.data_ptr(), self.Q_base.data_ptr(),
                                             self.item_bias_base.data_ptr(), self.item_bias.numel(), self.ldq,
                                             self.exchange.data_ptr(), float(scale), _stream_ptr()))


def hyper_of(cfg):
    return Hyper(cfg.learning_rate, cfg.P_reg, cfg.Q_reg, cfg.user_bias_reg, cfg.item_bias_reg)
