"""Scoring and ranking on the device (-m gpu): cu2rec_model_scores / cu2rec_model_recommend and bin/predict -u against
the oracle's prediction (predict.cu:17-30 restated in oracle/cu2rec_oracle.c: orc_predict) and a host sort
(predict.cu:49-65).  The dense product associates the sum over the factors differently from the reference's sequential
loop: scores agree within 1e-5 (values are O(1..5)), rankings agree wherever neighbouring scores are further apart."""
import os
import subprocess

import numpy as np
import pytest

import cu2rec_amd as cu
from cu2rec_amd import synth
from conftest import ROOT
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _oracle_scores(P, Q, ub, ib, gb):
    f = P.shape[1]
    fp = orc.C.POINTER(orc.C.c_float)
    out = np.empty((P.shape[0], Q.shape[0]), np.float32)
    for u in range(P.shape[0]):
        for i in range(Q.shape[0]):
            out[u, i] = orc.lib().orc_predict(f, P[u].ctypes.data_as(fp), Q[i].ctypes.data_as(fp), float(ub[u]), float(ib[i]),
                                              gb, orc.DOT_SEQ)
    return out


@pytest.mark.parametrize("users,items,f", [(1, 5, 4), (33, 100, 10), (70, 300, 100), (40, 257, 128), (17, 64, 200)])
def test_scores_match_oracle_prediction(users, items, f):
    rng = np.random.RandomState(users + f)
    P, Q = (rng.randn(users, f) * 0.3).astype(np.float32), (rng.randn(items, f) * 0.3).astype(np.float32)
    ub, ib = (rng.randn(users) * 0.3).astype(np.float32), (rng.randn(items) * 0.3).astype(np.float32)
    model = cu.Model(users, items, f, 3.5, P, Q, ub, ib)
    got = model.scores()
    want = _oracle_scores(P, Q, ub, ib, 3.5)
    assert float(np.abs(got - want).max()) <= 1e-5


def test_recommend_is_the_sorted_unrated_items():
    tr, _ = synth.make_ratings(90, 400, 6000, min_degree=3, seed=12)
    f = 50
    rng = np.random.RandomState(5)
    P, Q = (rng.randn(tr.rows, f) * 0.3).astype(np.float32), (rng.randn(tr.cols, f) * 0.3).astype(np.float32)
    ub, ib = (rng.randn(tr.rows) * 0.3).astype(np.float32), (rng.randn(tr.cols) * 0.3).astype(np.float32)
    model = cu.Model(tr.rows, tr.cols, f, tr.global_bias, P, Q, ub, ib)
    scores = model.scores()
    for k in (1, 10, tr.cols):
        items, top = model.recommend(cu.DeviceCSR(tr), k)
        for u in range(tr.rows):
            rated = set(tr.indices[tr.indptr[u]:tr.indptr[u + 1]].tolist())
            cand = [(scores[u, i], i) for i in range(tr.cols) if i not in rated]
            cand.sort(key=lambda t: -t[0])
            n = min(k, len(cand))
            np.testing.assert_array_equal(top[u, :n], np.array([c[0] for c in cand[:n]], np.float32))  # the device's own scores, sorted
            for j in range(n):
                assert items[u, j] not in rated and scores[u, items[u, j]] == top[u, j]
            assert (items[u, n:] == -1).all() and np.isnan(top[u, n:]).all()
    items, top = model.recommend(None, 3)  # nothing excluded
    np.testing.assert_array_equal(top, -np.sort(-scores, axis=1)[:, :3])


def test_bin_predict_many_users_matches_oracle(tmp_path):
    """bin/predict -u on a file of three users (not sorted by user id): every distinct userId is a new user = one row of
    a fresh model (row u draws the sample stream of user u), all fitted in one batch with the item side frozen -- exactly
    the oracle's frozen-item SGD on the same rows -- then scored and ranked on the device: the printed recommendations
    are the oracle's predictions of the unrated items, best first."""
    mf, predict = os.path.join(ROOT, "bin", "mf"), os.path.join(ROOT, "bin", "predict")
    tr, te = synth.make_ratings(200, 40, 3000, min_degree=3, seed=2)
    synth.write_csv(str(tmp_path / "train.csv"), tr)
    synth.write_csv(str(tmp_path / "test.csv"), te)
    (tmp_path / "train.cfg").write_text("0 60 8 0.02 42 0.02 0.02 0.02 0.02\n")
    subprocess.run([mf, "-c", str(tmp_path / "train.cfg"), "-m", "ordered", str(tmp_path / "train.csv"), str(tmp_path / "test.csv")],
                   stdout=subprocess.PIPE, check=True)
    (tmp_path / "predict.cfg").write_text("0 150 8 0.05 42 0.02 0.02 0.02 0.02\n")
    model_args = ["-c", str(tmp_path / "predict.cfg"), "-i", str(tmp_path / "train_f8_item_bias.csv"), "-g",
                  str(tmp_path / "train_f8_global_bias.csv"), "-q", str(tmp_path / "train_f8_q.csv")]
    rng = np.random.RandomState(3)
    users = {7: rng.choice(40, 6, replace=False), 3: rng.choice(40, 4, replace=False), 12: rng.choice(40, 9, replace=False)}
    ratings = {uid: [(int(i), rng.randint(1, 11) / 2.0) for i in its] for uid, its in users.items()}
    lines = ["userId,itemId,rating"]
    for uid in (7, 3, 12):  # deliberately not sorted by user id
        lines += ["%d,%d,%.1f" % (uid, i + 1, r) for i, r in ratings[uid]]
    (tmp_path / "many.csv").write_text("\n".join(lines) + "\n")
    out = subprocess.run([predict] + model_args + ["-u", "-k", "5", str(tmp_path / "many.csv")], stdout=subprocess.PIPE, text=True,
                         check=True).stdout
    blocks = out.split("User: ")[1:]
    order = [3, 7, 12]
    assert [int(b.split("\n")[0]) for b in blocks] == order  # ascending user id
    # oracle: rows 0..2 = users 3, 7, 12, every row from the one-user seed-42 draw, 150 frozen-item iterations
    Q = cu.read_array(str(tmp_path / "train_f8_q.csv"))
    ib = cu.read_array(str(tmp_path / "train_f8_item_bias.csv")).ravel()
    gb = float(cu.read_array(str(tmp_path / "train_f8_global_bias.csv")).ravel()[0])
    indptr = np.cumsum([0] + [len(ratings[u]) for u in order]).astype(np.int32)
    indices = np.array([i for u in order for i, _ in ratings[u]], np.int32)
    data = np.array([r for u in order for _, r in ratings[u]], np.float32)
    P = np.tile(orc.normal_fill(8, 8), (3, 1))
    ub = np.tile(orc.normal_fill(1, 8), 3)
    Qc, ibc = Q.copy(), ib.copy()
    orc.sgd_iterations(orc.CSR(indptr, indices, data, 3, Q.shape[0]), P, Qc, ub, ibc, gb, (0.05, 0.02, 0.02, 0.02, 0.02), 42, 0, 150,
                       dot_order=orc.DOT_TREE16, update_items=False)
    want = _oracle_scores(P, Q, ub, ib, gb)
    for row, b in enumerate(blocks):
        rated = {i for i, _ in ratings[order[row]]}
        cand = sorted([(want[row, i], i) for i in range(Q.shape[0]) if i not in rated], key=lambda t: -t[0])[:5]
        got = [l.split("\t") for l in b.split("\n") if l.startswith("Rank:")]
        assert len(got) == 5
        for j, (fields, (score, item)) in enumerate(zip(got, cand)):
            assert fields[0] == "Rank: %d" % (j + 1) and abs(float(fields[2].split()[-1]) - score) <= 1e-5
            if j + 1 < len(cand) and cand[j][0] - cand[j + 1][0] > 1e-5 and (j == 0 or cand[j - 1][0] - cand[j][0] > 1e-5):
                assert int(fields[1].split()[1]) == item
