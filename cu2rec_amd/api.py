"""Python mirror of the reference's host interface for the hot path, on top of the C ABI.

Names follow the reference (matrix_factorization/): readCSV, createSparseMatrix,
initialize_normal_array, writeToFile, Config.read_config, train(), calculate_loss_gpu /
get_error_metrics_gpu, sgd_update -- same argument meaning and error behaviour, numpy arrays in
place of raw host pointers.  All GPU work happens inside libcu2rec_amd.so.
"""
import ctypes as C
import os

import numpy as np

from ._lib import Config, Cu2recError, Hyper, TrainStats, check, lib

SGD_HOGWILD, SGD_SERIAL, SGD_ORDERED, SGD_PINGPONG, SGD_BLOCKSOLVE = 0, 1, 2, 3, 4
MODES = {"hogwild": SGD_HOGWILD, "serial": SGD_SERIAL, "ordered": SGD_ORDERED, "pingpong": SGD_PINGPONG,
         "blocksolve": SGD_BLOCKSOLVE}


def _mode(mode):
    return MODES[mode] if isinstance(mode, str) else int(mode)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError("expected shape %s, got %s" % (shape, a.shape))
    return a


# ------------------------------------------------------------------------------------------ config

def default_config(**overrides):
    cfg = Config()
    check(lib().cu2rec_config_default(C.byref(cfg)))
    for k, v in overrides.items():
        if not hasattr(cfg, k):
            raise AttributeError("Config has no field " + k)
        setattr(cfg, k, v)
    return cfg


def read_config(path, cfg=None):
    """Config::read_config (config.cu:7-13)."""
    cfg = cfg or default_config()
    check(lib().cu2rec_config_read(os.fsencode(path), C.byref(cfg)))
    return cfg


def write_config(path, cfg):
    """Config::write_config (config.cu:15-22)."""
    check(lib().cu2rec_config_write(os.fsencode(path), C.byref(cfg)))


def print_config(cfg):
    check(lib().cu2rec_config_print(C.byref(cfg)))


# ------------------------------------------------------------------------------------------ host data

class Ratings:
    """std::vector<Rating> of readCSV: 0-based user / item ids and ratings, plus rows, cols, global_bias."""

    def __init__(self, user, item, rating, rows, cols, global_bias):
        self.user, self.item, self.rating = user, item, rating
        self.rows, self.cols, self.global_bias = rows, cols, global_bias

    def __len__(self):
        return len(self.user)


class HostCSR:
    """Host CSR in the reference layout (matrix.h:11-19)."""

    def __init__(self, indptr, indices, data, rows, cols, global_bias=0.0):
        self.indptr = np.ascontiguousarray(indptr, dtype=np.int32)
        self.indices = np.ascontiguousarray(indices, dtype=np.int32)
        self.data = np.ascontiguousarray(data, dtype=np.float32)
        self.rows, self.cols, self.nnz = int(rows), int(cols), int(self.indices.shape[0])
        self.global_bias = float(np.float32(global_bias))
        if self.indptr.shape[0] != self.rows + 1:
            raise ValueError("indptr must have rows + 1 entries")

    def slice_users(self, u0, u1):
        """Rows [u0, u1) as their own CSR (user-sharding; item ids unchanged)."""
        out = np.empty(u1 - u0 + 1, np.int32)
        off, nnz = C.c_int(), C.c_int()
        check(lib().cu2rec_csr_slice(_ptr(self.indptr), self.rows, u0, u1, _ptr(out), C.byref(off), C.byref(nnz)))
        return HostCSR(out, self.indices[off.value:off.value + nnz.value], self.data[off.value:off.value + nnz.value],
                       u1 - u0, self.cols, self.global_bias)


def readCSV(path):
    """readCSV (util.cu:17-45) -> Ratings.  Raises Cu2recError(CU2REC_EIO) if the file cannot be opened."""
    h = C.c_void_p()
    check(lib().cu2rec_ratings_read_csv(os.fsencode(path), C.byref(h)))
    try:
        n, rows, cols, gb = C.c_int(), C.c_int(), C.c_int(), C.c_float()
        check(lib().cu2rec_ratings_info(h, C.byref(n), C.byref(rows), C.byref(cols), C.byref(gb)))
        pu, pi, pr = C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.POINTER(C.c_float)()
        check(lib().cu2rec_ratings_view(h, C.byref(pu), C.byref(pi), C.byref(pr)))
        if n.value:
            user = np.ctypeslib.as_array(pu, (n.value,)).copy()
            item = np.ctypeslib.as_array(pi, (n.value,)).copy()
            rating = np.ctypeslib.as_array(pr, (n.value,)).copy()
        else:
            user, item, rating = np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32)
        return Ratings(user, item, rating, rows.value, cols.value, gb.value)
    finally:
        lib().cu2rec_ratings_free(h)


def createSparseMatrix(path_or_ratings, rows=None):
    """readCSV + createSparseMatrix's host half (util.cu:152-179) -> HostCSR."""
    if isinstance(path_or_ratings, Ratings):
        r = path_or_ratings
        # rebuild through the library from the COO arrays (same code path as a file)
        return _csr_from_coo(r.user, r.item, r.rating, rows if rows is not None else r.rows, r.cols, r.global_bias)
    h = C.c_void_p()
    check(lib().cu2rec_ratings_read_csv(os.fsencode(path_or_ratings), C.byref(h)))
    try:
        n, nrows, cols, gb = C.c_int(), C.c_int(), C.c_int(), C.c_float()
        check(lib().cu2rec_ratings_info(h, C.byref(n), C.byref(nrows), C.byref(cols), C.byref(gb)))
        R = nrows.value if rows is None else rows
        indptr = np.zeros(R + 1, np.int32)
        indices = np.zeros(n.value, np.int32)
        data = np.zeros(n.value, np.float32)
        check(lib().cu2rec_csr_build(h, R, _ptr(indptr), _ptr(indices), _ptr(data)))
        return HostCSR(indptr, indices, data, R, cols.value, gb.value)
    finally:
        lib().cu2rec_ratings_free(h)


def _csr_from_coo(user, item, rating, rows, cols, global_bias):
    user = np.asarray(user, np.int64)
    if len(user) and np.any(np.diff(user) < 0):
        raise Cu2recError(-1, "ratings must be sorted by userId")
    counts = np.bincount(user, minlength=rows) if len(user) else np.zeros(rows, np.int64)
    indptr = np.zeros(rows + 1, np.int32)
    np.cumsum(counts, out=indptr[1:])
    return HostCSR(indptr, np.asarray(item, np.int32), np.asarray(rating, np.float32), rows, cols, global_bias)


def initialize_normal_array(size, n_factors, mean=0.0, stddev=1.0, seed=42):
    """initialize_normal_array (util.cu:124-144): mt19937(seed), normal(mean, stddev / n_factors)."""
    out = np.empty(int(size), np.float32)
    check(lib().cu2rec_init_normal(_ptr(out), int(size), int(n_factors), float(mean), float(stddev), int(seed)))
    return out


def writeToFile(parent_dir, base_filename, component, data, rows, cols, factors):
    """writeToFile (util.cu:99-103) with extension csv."""
    data = _f32(data).reshape(rows, cols)
    check(lib().cu2rec_write_component(os.fsencode(parent_dir), os.fsencode(base_filename), os.fsencode(component),
                                       _ptr(data), rows, cols, factors))


def writeCSV(path, data):
    data = _f32(data)
    data = data.reshape(data.shape[0], -1)
    check(lib().cu2rec_write_csv(os.fsencode(path), _ptr(data), data.shape[0], data.shape[1]))


def read_array(path):
    """read_array (util.cu:52-81) -> 2-D float32 array."""
    p = C.POINTER(C.c_float)()
    rows, cols = C.c_int(), C.c_int()
    check(lib().cu2rec_read_array(os.fsencode(path), C.byref(p), C.byref(rows), C.byref(cols)))
    try:
        n = rows.value * cols.value
        return np.ctypeslib.as_array(p, (n,)).copy().reshape(rows.value, cols.value) if n else np.zeros((0, 0), np.float32)
    finally:
        lib().cu2rec_free(p)


def sampler_index(seed, user, iteration, low, high):
    return lib().cu2rec_sampler_index(seed, user, iteration, low, high)


def item_update_rates(csr):
    """Expected SGD updates per iteration on each item from the users of `csr` (sum over raters of 1 / degree)."""
    out = np.zeros(csr.cols, np.float64)
    check(lib().cu2rec_item_update_rates(_ptr(csr.indptr), _ptr(csr.indices), csr.rows, csr.cols, _ptr(out)))
    return out


def shard_plan(rows, nranks):
    out = np.zeros(nranks + 1, np.int32)
    check(lib().cu2rec_shard_plan(rows, nranks, _ptr(out)))
    return out


# ------------------------------------------------------------------------------------------ device objects

def device_count():
    return lib().cu2rec_device_count()


class DeviceCSR:
    """CudaCSRMatrix (matrix.h:11-19): device copy of a HostCSR."""

    def __init__(self, host):
        self.host = host
        self._h = C.c_void_p()
        check(lib().cu2rec_csr_create(host.rows, host.cols, host.nnz, _ptr(host.indptr), _ptr(host.indices),
                                      _ptr(host.data), C.byref(self._h)))
        self.rows, self.cols, self.nnz = host.rows, host.cols, host.nnz

    def device_ptrs(self):
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(lib().cu2rec_csr_device_ptrs(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def blocksolve_items(self):
        """cu2rec_csr_blocksolve_items: how many items the block-solve mode solves block-wise here (0: it is the ordered walk)."""
        n = lib().cu2rec_csr_blocksolve_items(self._h)
        if n < 0:
            check(-1)
        return n

    def close(self):
        if self._h:
            lib().cu2rec_csr_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Model:
    """Device-resident P, Q, user_bias, item_bias (+ global_bias).  None = the reference's
    seed-42 normal initialisation (training.cu:28,54,212-213)."""

    def __init__(self, rows, cols, n_factors, global_bias, P=None, Q=None, user_bias=None, item_bias=None):
        self.rows, self.cols, self.n_factors, self.global_bias = int(rows), int(cols), int(n_factors), float(global_bias)
        P, Q = _f32(P, (rows, n_factors)), _f32(Q, (cols, n_factors))
        user_bias, item_bias = _f32(user_bias, (rows,)), _f32(item_bias, (cols,))
        self._h = C.c_void_p()
        check(lib().cu2rec_model_create(rows, cols, n_factors, _ptr(P), _ptr(Q), _ptr(user_bias), _ptr(item_bias),
                                        float(global_bias), C.byref(self._h)))

    def sgd(self, train, hyper, seed, iter0, n_iters, mode=SGD_HOGWILD, update_items=True):
        """n_iters reference iterations of sgd_update (sgd.cu:22-75) starting at global iteration iter0."""
        h = hyper if isinstance(hyper, Hyper) else Hyper(*[float(x) for x in hyper])
        check(lib().cu2rec_model_sgd(self._h, train._h, C.byref(h), int(seed), int(iter0), int(n_iters), _mode(mode),
                                     1 if update_items else 0))

    def loss(self, ratings):
        """calculate_loss_gpu + get_error_metrics_gpu (loss.cu:40-49,195-200) -> dict(mae, rmse, sum_abs, sum_sq)."""
        sa, ss, mae, rmse = C.c_double(), C.c_double(), C.c_float(), C.c_float()
        check(lib().cu2rec_model_loss(self._h, ratings._h, C.byref(sa), C.byref(ss), C.byref(mae), C.byref(rmse)))
        return {"mae": mae.value, "rmse": rmse.value, "sum_abs": sa.value, "sum_sq": ss.value}

    def download(self):
        """-> P, Q, user_bias, item_bias as dense numpy arrays (training.cu:180-185)."""
        P = np.empty((self.rows, self.n_factors), np.float32)
        Q = np.empty((self.cols, self.n_factors), np.float32)
        ub, ib = np.empty(self.rows, np.float32), np.empty(self.cols, np.float32)
        check(lib().cu2rec_model_download(self._h, _ptr(P), _ptr(Q), _ptr(ub), _ptr(ib)))
        return P, Q, ub, ib

    def scores(self):
        """predict_ratings (predict.cu:17-30) for every user against every item -> (rows, cols) float32."""
        out = np.empty((self.rows, self.cols), np.float32)
        check(lib().cu2rec_model_scores_host(self._h, _ptr(out)))
        return out

    def recommend(self, rated, k):
        """get_recommendations (predict.cu:49-65) for every user: the k best unrated items -> (items, scores), each
        (rows, k); -1 / NaN where a user has fewer than k unrated items.  rated: DeviceCSR (row u = user u) or None."""
        items, scores = np.empty((self.rows, k), np.int32), np.empty((self.rows, k), np.float32)
        check(lib().cu2rec_model_recommend(self._h, rated._h if rated is not None else None, int(k), _ptr(items), _ptr(scores)))
        return items, scores

    def device_ptrs(self):
        a, b, c, d = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(lib().cu2rec_model_device_ptrs(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return a.value, b.value, c.value, d.value

    @property
    def ld(self):
        ld = C.c_int()
        check(lib().cu2rec_model_info(self._h, None, None, None, C.byref(ld), None))
        return ld.value

    def close(self):
        if self._h:
            lib().cu2rec_model_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def default_mode(n_factors):
    """The default SGD mode of train(), bin/mf and bench.py: block-solve (mf_sequential.cu's result within float rounding, the mode
    certified against the 1e-4 RMSE bar); above 252 factors, where it is not compiled, `ordered` (the same result bit for bit)."""
    return SGD_BLOCKSOLVE if n_factors <= 252 else SGD_ORDERED


def train(train_matrix, test_matrix, cfg, Q=None, item_bias=None, global_bias=None, mode=None, verbose=True,
          P=None, user_bias=None, return_stats=False):
    """train() (training.h:12-15).  mode None = default_mode(cfg.n_factors); "hogwild" opts into sgd.cu's racy semantics.  train_matrix / test_matrix: DeviceCSR (or HostCSR, uploaded here).
    Q / item_bias given = the 11-argument overload (caller-initialised item side, predict.cu:126);
    otherwise everything starts from the seed-42 normal init.  Returns (P, Q, losses, user_bias,
    item_bias) like the reference's out-pointers; cfg.learning_rate / cfg.cur_iterations are
    updated in place."""
    tr = train_matrix if isinstance(train_matrix, DeviceCSR) else DeviceCSR(train_matrix)
    te = test_matrix if isinstance(test_matrix, DeviceCSR) else DeviceCSR(test_matrix)
    gb = tr.host.global_bias if global_bias is None else global_bias
    model = Model(tr.rows, tr.cols, cfg.n_factors, gb, P=P, Q=Q, user_bias=user_bias, item_bias=item_bias)
    losses = np.empty(max(cfg.total_iterations, 1), np.float32)
    stats = TrainStats()
    mode = default_mode(cfg.n_factors) if mode is None else mode
    check(lib().cu2rec_train(tr._h, te._h, C.byref(cfg), model._h, _mode(mode), 1 if verbose else 0, _ptr(losses),
                             C.byref(stats)))
    Pn, Qn, ub, ib = model.download()
    model.close()
    out = (Pn, Qn, losses[:cfg.total_iterations], ub, ib)
    return out + (stats,) if return_stats else out


# ------------------------------------------------------------------------------------------ raw device pointers

def sgd_update(indptr, indices, data, n_rows, n_cols, P, ldp, Q, ldq, user_bias, item_bias, global_bias, n_factors,
               hyper, seed, iter0, n_iters, mode=SGD_HOGWILD, update_items=True, user_offset=0, stream=None,
               sample_pairs=None):
    """cu2rec_sgd_update(_ex) on raw device addresses (ints); sample_pairs: the optional side-by-side sample array."""
    h = hyper if isinstance(hyper, Hyper) else Hyper(*[float(x) for x in hyper])
    check(lib().cu2rec_sgd_update_ex(indptr, indices, data, n_rows, n_cols, P, ldp, Q, ldq, user_bias, item_bias,
                                     float(global_bias), n_factors, C.byref(h), int(seed), int(iter0), int(n_iters),
                                     _mode(mode), 1 if update_items else 0, int(user_offset), sample_pairs, stream))


def sample_pairs_build(indices, data, nnz, pairs, stream=None):
    """cu2rec_sample_pairs_build on raw device addresses: pairs[k] = {indices[k], data[k]}."""
    check(lib().cu2rec_sample_pairs_build(indices, data, int(nnz), pairs, stream))


class Schedule:
    """Workspace of the ordered mode for one device CSR (cu2rec_schedule)."""

    def __init__(self, indptr, indices, n_rows, n_cols, nnz):
        self._h = C.c_void_p()
        check(lib().cu2rec_schedule_create(indptr, indices, n_rows, n_cols, nnz, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().cu2rec_schedule_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def sgd_update_ordered(schedule, indptr, indices, data, n_rows, n_cols, P, ldp, Q, ldq, user_bias, item_bias,
                       global_bias, n_factors, hyper, seed, iter0, n_iters, update_items=True, user_offset=0,
                       stream=None):
    """cu2rec_sgd_update_ordered on raw device addresses."""
    h = hyper if isinstance(hyper, Hyper) else Hyper(*[float(x) for x in hyper])
    check(lib().cu2rec_sgd_update_ordered(schedule._h, indptr, indices, data, n_rows, n_cols, P, ldp, Q, ldq, user_bias,
                                          item_bias, float(global_bias), n_factors, C.byref(h), int(seed), int(iter0),
                                          int(n_iters), 1 if update_items else 0, int(user_offset), stream))


def sgd_update_blocksolve(schedule, indptr, indices, data, n_rows, n_cols, P, ldp, Q, ldq, user_bias, item_bias,
                          global_bias, n_factors, hyper, seed, iter0, n_iters, update_items=True, user_offset=0,
                          stream=None):
    """cu2rec_sgd_update_blocksolve on raw device addresses."""
    h = hyper if isinstance(hyper, Hyper) else Hyper(*[float(x) for x in hyper])
    check(lib().cu2rec_sgd_update_blocksolve(schedule._h, indptr, indices, data, n_rows, n_cols, P, ldp, Q, ldq,
                                             user_bias, item_bias, float(global_bias), n_factors, C.byref(h), int(seed),
                                             int(iter0), int(n_iters), 1 if update_items else 0, int(user_offset), stream))


def blocksolve_min_rate(rate=0.0):
    """cu2rec_blocksolve_min_rate: expected updates per iteration from which an item's chain is solved block-wise.
    rate > 0 sets it, rate < 0 returns to the automatic threshold, 0 queries; returns what was in force before (-1: automatic),
    so blocksolve_min_rate(previous) restores either state."""
    return lib().cu2rec_blocksolve_min_rate(float(rate))


def blocksolve_lookahead_blocks(blocks=-1):
    """cu2rec_blocksolve_lookahead_blocks: items expected to collect at least this many 64-update blocks per iteration run phase 2
    in the look-ahead form (default 24: the top chains; 0: off; read when a DeviceCSR's schedule is created)."""
    return lib().cu2rec_blocksolve_lookahead_blocks(int(blocks))


def blocksolve_topology():
    """cu2rec_blocksolve_topology -> ("device" | "events" | None before the first block-solve call, the reason)."""
    buf = C.create_string_buffer(512)
    mode = lib().cu2rec_blocksolve_topology(buf, len(buf))
    return {2: "device", 0: "events"}.get(mode), buf.value.decode(errors="replace")


def loss_raw(indptr, indices, data, n_rows, nnz, P, ldp, Q, ldq, user_bias, item_bias, global_bias, n_factors,
             workspace, errors_out=None, stream=None):
    sa, ss, mae, rmse = C.c_double(), C.c_double(), C.c_float(), C.c_float()
    check(lib().cu2rec_loss(indptr, indices, data, n_rows, nnz, P, ldp, Q, ldq, user_bias, item_bias,
                            float(global_bias), n_factors, errors_out, workspace, C.byref(sa), C.byref(ss),
                            C.byref(mae), C.byref(rmse), stream))
    return {"mae": mae.value, "rmse": rmse.value, "sum_abs": sa.value, "sum_sq": ss.value}


def error_metrics_raw(errors, n, workspace, stream=None):
    """get_error_metrics_gpu (loss.cu:195-200) on a device array of residuals -> (mae, rmse)."""
    mae, rmse = C.c_float(), C.c_float()
    check(lib().cu2rec_error_metrics(errors, n, workspace, C.byref(mae), C.byref(rmse), stream))
    return mae.value, rmse.value
