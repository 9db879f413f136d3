"""the data-preparation equivalents produce files the trainer's reader accepts."""
import numpy as np

import cu2rec_amd as cu
from cu2rec_amd import prep


def test_map_split_config_roundtrip(tmp_path):
    raw = tmp_path / "raw.csv"
    rng = np.random.RandomState(0)
    rows = [(int(u), int(i), float(rng.choice([1, 2.5, 3, 4.5, 5])), 9999)
            for u in rng.permutation([7, 7, 7, 1000, 1000, 42, 42, 42, 42, 5, 5]) for i in rng.choice([10, 99, 3, 512, 77], 2, replace=False)]
    raw.write_text("userId,movieId,rating,timestamp\n" + "\n".join("%d,%d,%s,%d" % r for r in rows) + "\n")
    mapped, n_users, n_items, n = prep.map_ids(str(raw))
    assert (n_users, n) == (4, len(rows)) and n_items <= 5
    m = cu.createSparseMatrix(mapped)  # the product's reader: sorted by user, ids 1..N
    assert m.rows == n_users and m.cols == n_items and m.nnz == n
    first_user = rows[0][0]
    assert all(int(np.diff(m.indptr)[0]) == sum(1 for r in rows if r[0] == first_user) for _ in [0])  # first-seen user is id 1
    train, test = prep.split(mapped, test_fraction=0.3, seed=42)
    tr, te = cu.createSparseMatrix(train), cu.createSparseMatrix(test)
    assert tr.nnz + te.nnz == n and tr.rows == n_users and te.rows <= n_users and np.all(np.diff(tr.indptr) >= 1)
    cfg_path = prep.write_config(str(tmp_path / "c.cfg"), total_iterations=100, n_factors=10)
    cfg = cu.read_config(cfg_path)
    assert (cfg.total_iterations, cfg.n_factors, cfg.seed) == (100, 10, 42) and abs(cfg.learning_rate - 0.01) < 1e-9
    assert open(cfg_path).read() == "0 100 10 0.010000 42 0.020000 0.020000 0.020000 0.020000\n"  # create_config.py:13-16
