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


def pytest_collection_modifyitems(config, items):
    # the tests that join the background oracle runs go LAST: the runs then have the whole session to finish in
    late = [it for it in items if any(name in it.nodeid for name in _FULL_SHAPE_ORACLE_RUNS)]
    if late:
        items[:] = [it for it in items if it not in late] + late


def pytest_collection_finish(session):
    from concurrent.futures import ThreadPoolExecutor
    wanted = [(name, spec) for name, spec in _FULL_SHAPE_ORACLE_RUNS.items() if any(name in item.nodeid for item in session.items)]
    if not wanted or session.config.option.collectonly:
        return
    pool = ThreadPoolExecutor(max_workers=3)
    for name, (workload, f, iters, orders) in wanted:
        for order_name in orders:
            _BACKGROUND[(workload, f, iters, order_name)] = pool.submit(_oracle_job, workload, f, iters, order_name)


def oracle_state(workload, f, iters, order_name):
    """(P, Q, user_bias, item_bias) after `iters` sequential iterations of the oracle from the seed-42 initialisation: the background
    run started at collection time, or computed here if the test was selected some other way."""
    fut = _BACKGROUND.get((workload, f, iters, order_name))
    return fut.result() if fut is not None else _oracle_job(workload, f, iters, order_name)
