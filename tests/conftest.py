import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# the reference's bundled MovieLens latest-small file; exists only in the build container
ML_SMALL = "/root/reference/ratings_mapped.csv"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_path(name):
    return ML_SMALL if name == "ML_SMALL" else os.path.join(GOLDEN, name)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# ---- the CPU oracle's long runs of the full-shape block-solve tests, started when the session's tests are known and computed on
# spare host cores WHILE the GPU tests in front of them run (the oracle is test infrastructure; ctypes calls release the GIL).  Each is
# 1e8-scale sequential updates = 45-75 s of one core; the tests that need them join here.
_BACKGROUND = {}
_FULL_SHAPE_ORACLE_RUNS = {
    # test name fragment -> (key, workload, factors, iterations, dot orders)
    "test_blocksolve_full_shape_ml20m_1000_iterations_within_1e4_of_oracle": ("ml-20m", 100, 1000, ("TREE16", "SEQ")),
    "test_blocksolve_full_shape_netflix_f128_against_the_cpu_oracle": ("netflix", 128, 72, ("TREE16",)),
}


# ... and the CONVERGED run of configs[2] in the reference's own arithmetic (tools/oracle_converged.py: orc_train, sequential dot,
# float / double loss accumulators; 6.9e8 sequential updates = 3-5 minutes of one core each): test name fragment -> runs
_CONVERGED_ORACLE_RUNS = {
    "test_converged_ml20m_blocksolve_against_the_cpu_oracle_in_the_references_own_arithmetic":
        [("ml-20m", 100, 5000, "SEQ", "F32"), ("ml-20m", 100, 5000, "SEQ", "F64")],
}

_DATASET_LOCK = __import__("threading").Lock()


def _oracle_job(workload, f, iters, order_name):
    import bench
    from oracle import oracle as orc
    with _DATASET_LOCK:  # (one generation of a missing set, not one per job)
        tr, _ = bench.load_dataset(workload, 20240917, 0, lambda: None)
    state = orc.init_model(tr.rows, tr.cols, f)
    orc.sgd_iterations(orc.CSR(tr.indptr, tr.indices, tr.data, tr.rows, tr.cols, tr.global_bias), *state, tr.global_bias,
                       (0.01, 0.02, 0.02, 0.02, 0.02), 42, 0, iters, dot_order=getattr(orc, "DOT_" + order_name))
    return state


def _converged_job(workload, f, iters, dot_name, acc_name):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench
    import oracle_converged
    with _DATASET_LOCK:
        data = bench.load_dataset(workload, 20240917, 0, lambda: None)
    return oracle_converged.run(workload, f, iters, dot_name, acc_name, data=data)


def pytest_collection_modifyitems(config, items):
    # the tests that join the background oracle runs go LAST (the longest runs last of all): the runs then have the whole session
    # to finish in
    late = [it for it in items if any(name in it.nodeid for name in _FULL_SHAPE_ORACLE_RUNS)]
    last = [it for it in items if any(name in it.nodeid for name in _CONVERGED_ORACLE_RUNS)]
    if late or last:
        items[:] = [it for it in items if it not in late and it not in last] + late + last


def pytest_collection_finish(session):
    from concurrent.futures import ThreadPoolExecutor
    wanted = [(name, spec) for name, spec in _FULL_SHAPE_ORACLE_RUNS.items() if any(name in item.nodeid for item in session.items)]
    converged = [run for name, runs in _CONVERGED_ORACLE_RUNS.items() if any(name in item.nodeid for item in session.items) for run in runs]
    if not (wanted or converged) or session.config.option.collectonly:
        return
    pool = ThreadPoolExecutor(max_workers=5)
    for run in converged:  # (the longest first)
        _BACKGROUND[run] = pool.submit(_converged_job, *run)
    for name, (workload, f, iters, orders) in wanted:
        for order_name in orders:
            _BACKGROUND[(workload, f, iters, order_name)] = pool.submit(_oracle_job, workload, f, iters, order_name)


def _join(fut, what, limit_s=1500):
    """Wait for a background oracle run, leaving a line in gpurun_out/pytest_oracle_wait.log every 30 s (a GPU box's watchdog takes
    seven silent minutes for a hang; the wait is legitimate host work and bounded)."""
    import time
    from concurrent.futures import TimeoutError as FutureTimeout
    t0 = time.time()
    while True:
        try:
            return fut.result(timeout=30)
        except FutureTimeout:
            waited = time.time() - t0
            try:
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                with open(os.path.join(ROOT, "gpurun_out", "pytest_oracle_wait.log"), "a") as fh:
                    fh.write("%s waiting %.0f s for the CPU oracle's %s\n" % (time.strftime("%H:%M:%S"), waited, what))
            except OSError:
                pass
            if waited > limit_s:
                raise RuntimeError("the CPU oracle's %s did not finish within %d s" % (what, limit_s))


def oracle_state(workload, f, iters, order_name):
    """(P, Q, user_bias, item_bias) after `iters` sequential iterations of the oracle from the seed-42 initialisation: the background
    run started at collection time, or computed here if the test was selected some other way."""
    fut = _BACKGROUND.get((workload, f, iters, order_name))
    return _join(fut, "%s f=%d %d iterations %s" % (workload, f, iters, order_name)) if fut is not None else _oracle_job(workload, f, iters, order_name)


def oracle_converged_run(workload, f, iters, dot_name, acc_name):
    """(record, (P, Q, user_bias, item_bias)) of tools/oracle_converged.py's run: the oracle's train() under the reference's schedule."""
    fut = _BACKGROUND.get((workload, f, iters, dot_name, acc_name))
    if fut is None:
        from concurrent.futures import ThreadPoolExecutor
        fut = ThreadPoolExecutor(max_workers=1).submit(_converged_job, workload, f, iters, dot_name, acc_name)
    return _join(fut, "converged run %s f=%d %d iterations %s %s" % (workload, f, iters, dot_name, acc_name))
