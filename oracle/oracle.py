"""ctypes binding of the CPU oracle (oracle/cu2rec_oracle.c).  TEST INFRASTRUCTURE ONLY.

Import this from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never from
the cu2rec_amd package.  The shared object is built by `make -C oracle` (also by
__graft_entry__.build()); on the GPU box the prebuilt oracle/_build/libcu2rec_oracle.so that
travelled with the snapshot is used, and rebuilt with gcc only if it is missing.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "libcu2rec_oracle.so")
REF_DIR = os.path.join(HERE, "_ref")

DOT_SEQ, DOT_TREE16 = 0, 1
ACC_F64, ACC_F32 = 0, 1
SCHED_SEQUENTIAL, SCHED_PATIENCE = 0, 1


class Config(C.Structure):
    _fields_ = [("cur_iterations", C.c_int), ("total_iterations", C.c_int), ("n_factors", C.c_int),
                ("learning_rate", C.c_float), ("seed", C.c_int), ("P_reg", C.c_float), ("Q_reg", C.c_float),
                ("user_bias_reg", C.c_float), ("item_bias_reg", C.c_float), ("is_train", C.c_int),
                ("n_threads", C.c_int), ("check_error", C.c_int), ("patience", C.c_float),
                ("learning_rate_decay", C.c_float)]


class Ratings(C.Structure):
    _fields_ = [("n", C.c_int), ("rows", C.c_int), ("cols", C.c_int), ("global_bias", C.c_float),
                ("user", C.POINTER(C.c_int)), ("item", C.POINTER(C.c_int)), ("rating", C.POINTER(C.c_float))]


class Hyper(C.Structure):
    _fields_ = [("learning_rate", C.c_float), ("P_reg", C.c_float), ("Q_reg", C.c_float),
                ("user_bias_reg", C.c_float), ("item_bias_reg", C.c_float)]


class LogEntry(C.Structure):
    _fields_ = [("iteration", C.c_int), ("train_mae", C.c_float), ("train_rmse", C.c_float),
                ("test_mae", C.c_float), ("test_rmse", C.c_float), ("lr", C.c_float)]


_lib = None


def build(force=False):
    src = os.path.join(HERE, "cu2rec_oracle.c")
    stale = (not os.path.exists(LIB_PATH)) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src)
    if force or stale:
        subprocess.run(["make", "-C", HERE, "-s"], check=True)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
        L.orc_read_csv.argtypes = [C.c_char_p, C.POINTER(Ratings)]
        L.orc_build_csr.argtypes = [C.POINTER(Ratings), C.c_int, ip, ip, fp]
        L.orc_normal_fill.argtypes = [fp, C.c_size_t, C.c_int, C.c_float, C.c_float, C.c_int]
        L.orc_philox4x32_10.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.orc_draw.restype = C.c_uint32
        L.orc_draw.argtypes = [C.c_uint64] * 3
        L.orc_uniform.restype = C.c_float
        L.orc_uniform.argtypes = [C.c_uint32]
        L.orc_sample.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int]
        L.orc_predict.restype = C.c_float
        L.orc_predict.argtypes = [C.c_int, fp, fp, C.c_float, C.c_float, C.c_float, C.c_int]
        L.orc_sgd_iterations.restype = None
        L.orc_sgd_iterations.argtypes = [ip, ip, fp, C.c_int, fp, fp, fp, fp, C.c_float, C.POINTER(Hyper),
                                         C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int]
        L.orc_loss.restype = None
        L.orc_loss.argtypes = [ip, ip, fp, C.c_int, C.c_int, fp, fp, fp, fp, C.c_float, C.c_int, C.c_int,
                               C.c_int, fp, C.POINTER(C.c_double), C.POINTER(C.c_double), fp, fp]
        L.orc_error_metrics.restype = None
        L.orc_error_metrics.argtypes = [fp, C.c_int, fp, fp]
        L.orc_train.argtypes = [ip, ip, fp, C.c_int, C.c_int, C.c_int, ip, ip, fp, C.c_int, C.c_int,
                                C.POINTER(Config), fp, fp, fp, fp, C.c_float, C.c_int, C.c_int, C.c_int,
                                C.POINTER(LogEntry), C.c_int]
        L.orc_write_csv.argtypes = [C.c_char_p, fp, C.c_int, C.c_int]
        L.orc_config_read.argtypes = [C.c_char_p, C.POINTER(Config)]
        L.orc_config_write.argtypes = [C.c_char_p, C.POINTER(Config)]
        L.orc_now_seconds.restype = C.c_double
        _lib = L
    return _lib


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def default_config(**kw):
    c = Config()
    lib().orc_config_default(C.byref(c))
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def read_config(path):
    c = default_config()
    rc = lib().orc_config_read(path.encode(), C.byref(c))
    if rc != 0:
        raise IOError("orc_config_read(%s) -> %d" % (path, rc))
    return c


class CSR:
    """Host CSR in the reference layout (matrix.h:11-19): int32 indptr/indices, f32 data."""

    def __init__(self, indptr, indices, data, rows, cols, global_bias=0.0):
        self.indptr = np.ascontiguousarray(indptr, dtype=np.int32)
        self.indices = np.ascontiguousarray(indices, dtype=np.int32)
        self.data = np.ascontiguousarray(data, dtype=np.float32)
        self.rows, self.cols, self.nnz = int(rows), int(cols), int(len(self.indices))
        self.global_bias = float(np.float32(global_bias))


def read_csv(path):
    """readCSV + createSparseMatrix (util.cu:17-45,152-179) -> CSR"""
    r = Ratings()
    if lib().orc_read_csv(path.encode(), C.byref(r)) != 0:
        raise IOError("cannot open " + path)
    try:
        indptr = np.zeros(r.rows + 1, np.int32)
        indices = np.zeros(r.n, np.int32)
        data = np.zeros(r.n, np.float32)
        rc = lib().orc_build_csr(C.byref(r), r.rows, _i(indptr), _i(indices), _f(data))
        if rc == -1:
            raise ValueError("ratings are not sorted by user: " + path)
        return CSR(indptr, indices, data, r.rows, r.cols, r.global_bias)
    finally:
        lib().orc_ratings_free(C.byref(r))


def normal_fill(size, n_factors, mean=0.0, stddev=1.0, seed=42):
    out = np.empty(size, np.float32)
    lib().orc_normal_fill(_f(out), size, n_factors, mean, stddev, seed)
    return out


def init_model(rows, cols, f):
    """mf_sequential.cu:91-94 / training.cu:28,54,212-213: all four arrays seeded with 42."""
    return (normal_fill(rows * f, f).reshape(rows, f), normal_fill(cols * f, f).reshape(cols, f),
            normal_fill(rows, f), normal_fill(cols, f))


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return list(o)


def sample(seed, user, iteration, low, high):
    return lib().orc_sample(seed, user, iteration, low, high)


def sgd_iterations(csr, P, Q, ub, ib, global_bias, hyper, seed, iter0, n_iters, dot_order=DOT_SEQ,
                   update_items=True):
    """In place on the given float32 arrays (mf_sequential.cu:102-143)."""
    f = P.shape[1]
    for a in (P, Q, ub, ib):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    h = Hyper(*[float(x) for x in hyper])
    lib().orc_sgd_iterations(_i(csr.indptr), _i(csr.indices), _f(csr.data), csr.rows, _f(P), _f(Q), _f(ub),
                             _f(ib), float(global_bias), C.byref(h), f, seed, iter0, n_iters, dot_order,
                             1 if update_items else 0)


def sgd_pingpong_iterations(csr, P, Q, Q_target, ub, ib, ib_target, global_bias, hyper, seed, iter0, n_iters,
                            dot_order=DOT_SEQ, update_items=True, swap_last=True):
    """The reference GPU kernel's own semantics (sgd.cu:22-75, training.cu:107-171), in place; see cu2rec_oracle.h."""
    L = lib()
    if L.orc_sgd_pingpong_iterations.argtypes is None:
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
        L.orc_sgd_pingpong_iterations.restype = None
        L.orc_sgd_pingpong_iterations.argtypes = [ip, ip, fp, C.c_int, C.c_int, fp, fp, fp, fp, fp, fp, C.c_float,
                                                  C.POINTER(Hyper), C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_int,
                                                  C.c_int, C.c_int]
    for a in (P, Q, Q_target, ub, ib, ib_target):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    h = Hyper(*[float(v) for v in hyper])
    L.orc_sgd_pingpong_iterations(_i(csr.indptr), _i(csr.indices), _f(csr.data), csr.rows, csr.cols, _f(P), _f(Q),
                                  _f(Q_target), _f(ub), _f(ib), _f(ib_target), float(global_bias), C.byref(h), P.shape[1],
                                  seed, iter0, n_iters, dot_order, 1 if update_items else 0, 1 if swap_last else 0)


def pingpong_swap(Q, Q_target, ib, ib_target):
    L = lib()
    fp = C.POINTER(C.c_float)
    L.orc_pingpong_swap.restype = None
    L.orc_pingpong_swap.argtypes = [fp, fp, fp, fp, C.c_int, C.c_int]
    L.orc_pingpong_swap(_f(Q), _f(Q_target), _f(ib), _f(ib_target), Q.shape[0], Q.shape[1])


def sgd_one(csr, x, P, Q, ub, ib, global_bias, hyper, seed, it, dot_order=DOT_SEQ, update_items=True):
    """One update of user x at iteration `it`, in place (the loop body, mf_sequential.cu:104-142)."""
    L = lib()
    if L.orc_sgd_one.argtypes is None:
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
        L.orc_sgd_one.restype = None
        L.orc_sgd_one.argtypes = [ip, ip, fp, C.c_int, fp, fp, fp, fp, C.c_float, C.POINTER(Hyper), C.c_int,
                                  C.c_uint64, C.c_uint64, C.c_int, C.c_int]
    h = Hyper(*[float(v) for v in hyper])
    L.orc_sgd_one(_i(csr.indptr), _i(csr.indices), _f(csr.data), int(x), _f(P), _f(Q), _f(ub), _f(ib),
                  float(global_bias), C.byref(h), P.shape[1], seed, it, dot_order, 1 if update_items else 0)


def sgd_iterations_parallel(csr, P, Q, ub, ib, global_bias, hyper, seed, iter0, n_iters, n_threads=0, dot_order=DOT_SEQ):
    """CPU-baseline only: Hogwild over users on n_threads host threads (0 = all).  Returns the thread count used."""
    L = lib()
    if L.orc_sgd_iterations_parallel.argtypes is None:
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
        L.orc_sgd_iterations_parallel.restype = C.c_int
        L.orc_sgd_iterations_parallel.argtypes = [ip, ip, fp, C.c_int, fp, fp, fp, fp, C.c_float, C.POINTER(Hyper),
                                                  C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int]
    h = Hyper(*[float(v) for v in hyper])
    return L.orc_sgd_iterations_parallel(_i(csr.indptr), _i(csr.indices), _f(csr.data), csr.rows, _f(P), _f(Q), _f(ub),
                                         _f(ib), float(global_bias), C.byref(h), P.shape[1], seed, iter0, n_iters,
                                         dot_order, int(n_threads))


def loss(csr, P, Q, ub, ib, global_bias, dot_order=DOT_SEQ, acc=ACC_F64, want_errors=False, rows=None):
    f = P.shape[1]
    rows = csr.rows if rows is None else rows
    err = np.empty(csr.nnz, np.float32) if want_errors else None
    sa, ss = C.c_double(), C.c_double()
    mae, rmse = C.c_float(), C.c_float()
    lib().orc_loss(_i(csr.indptr), _i(csr.indices), _f(csr.data), rows, csr.nnz, _f(P), _f(Q), _f(ub), _f(ib),
                   float(global_bias), f, dot_order, acc, _f(err) if want_errors else None, C.byref(sa),
                   C.byref(ss), C.byref(mae), C.byref(rmse))
    out = {"mae": mae.value, "rmse": rmse.value, "sum_abs": sa.value, "sum_sq": ss.value}
    if want_errors:
        out["errors"] = err
    return out


def error_metrics(errors):
    errors = np.ascontiguousarray(errors, np.float32)
    mae, rmse = C.c_float(), C.c_float()
    lib().orc_error_metrics(_f(errors), len(errors), C.byref(mae), C.byref(rmse))
    return mae.value, rmse.value


def train(train_csr, test_csr, cfg, P, Q, ub, ib, global_bias, dot_order=DOT_SEQ, acc=ACC_F64,
          schedule=SCHED_PATIENCE):
    cap = cfg.total_iterations // max(cfg.check_error, 1) + 4
    log = (LogEntry * cap)()
    n = lib().orc_train(_i(train_csr.indptr), _i(train_csr.indices), _f(train_csr.data), train_csr.rows,
                        train_csr.cols, train_csr.nnz, _i(test_csr.indptr), _i(test_csr.indices),
                        _f(test_csr.data), test_csr.rows, test_csr.nnz, C.byref(cfg), _f(P), _f(Q), _f(ub), _f(ib),
                        float(global_bias), dot_order, acc, schedule, log, cap)
    return [dict(iteration=e.iteration, train_mae=e.train_mae, train_rmse=e.train_rmse, test_mae=e.test_mae,
                 test_rmse=e.test_rmse, lr=e.lr) for e in log[:n]]


def write_csv(path, arr):
    arr = np.ascontiguousarray(arr, np.float32)
    a2 = arr.reshape(arr.shape[0], -1)
    return lib().orc_write_csv(path.encode(), _f(a2), a2.shape[0], a2.shape[1])


def ref_binary(name="mf_cpu"):
    """Path of the compiled reference CPU twin (oracle/build_ref.py), or None."""
    p = os.path.join(REF_DIR, name)
    return p if os.path.exists(p) else None
