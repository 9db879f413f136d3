"""Data preparation equivalents of the reference's preprocessing scripts, so raw
`userId,itemId,rating[,...]` files can be taken to the trainer's input format without the reference tree.
The outputs equal the reference scripts' byte for byte on the same input (tests/test_prep.py against files produced by the
reference's own scripts, oracle/gen_prep_golden.py): ids 1..N in first-seen order, rows stably sorted by user, one header line,
ratings printed as Python prints a float ("4.0", "2.5"), the split = `random.seed(seed)`, one shuffle of all rows, the first
int(n * (1 - test_ratio)) of them for training, both halves stably sorted by user; 9-field config without a newline.

  python -m cu2rec_amd.prep map    ratings.csv            -> ratings_mapped.csv   (preprocessing/map_items.py:21-89)
  python -m cu2rec_amd.prep split  ratings_mapped.csv     -> *_train.csv, *_test.csv (split_to_test_train.py:39-78)
  python -m cu2rec_amd.prep config out.cfg [--iters ...]  -> 9-field config       (create_config.py:10-32)
  python -m cu2rec_amd.prep sort   ratings.csv            -> ratings_sorted.csv   (sort_ratings.py:11-42: by (userId, itemId))
  python -m cu2rec_amd.prep to-npy p.csv [q.csv ...]      -> p.npy ...            (convert_to_np.py:6-23: float64 .npy)
  python -m cu2rec_amd.prep map-netflix train.txt test.txt -> ratings_mapped_train.csv, ratings_mapped_test.csv next to train.txt
                                                                                  (map_netflix.py:9-28)
"""
import argparse
import os
import random
import sys

import numpy as np


def _read_triples(path):
    """userId,itemId,rating[,anything] with one header line -> (user, item, rating) as read (any id values)."""
    users, items, ratings = [], [], []
    with open(path) as fh:
        fh.readline()
        for line in fh:
            parts = line.strip().split(",")
            if len(parts) < 3 or not parts[0]:
                continue
            users.append(int(parts[0]))
            items.append(int(parts[1]))
            ratings.append(float(parts[2]))
    return np.asarray(users, np.int64), np.asarray(items, np.int64), np.asarray(ratings, np.float64)


def _first_seen_ids(values):
    """Dense 1-based ids in order of first appearance (map_items.py:40-54)."""
    uniq, first = np.unique(values, return_index=True)
    order = np.argsort(first, kind="stable")
    new_id = np.empty(len(uniq), np.int64)
    new_id[order] = np.arange(1, len(uniq) + 1)
    return new_id[np.searchsorted(uniq, values)]


def _write(path, user, item, rating):
    """map_items.py:80-89: header, then `user,item,rating` with the rating as str(float) ("4.0", "2.5")."""
    with open(path, "w", newline="") as fh:
        fh.write("userId,itemId,rating\n")
        fh.write("".join("%d,%d,%s\n" % (u, i, repr(float(r))) for u, i, r in zip(user.tolist(), item.tolist(), rating.tolist())))


def map_ids(src, dst=None):
    """Remap user and item ids to 1..N (first-seen order) and sort rows by user, stably (map_items.py:21-89)."""
    user, item, rating = _read_triples(src)
    user, item = _first_seen_ids(user), _first_seen_ids(item)
    order = np.argsort(user, kind="stable")
    dst = dst or os.path.splitext(src)[0] + "_mapped.csv"
    _write(dst, user[order], item[order], rating[order])
    return dst, int(user.max(initial=0)), int(item.max(initial=0)), len(user)


def sort_ratings(src, dst=None):
    """sort_ratings.py:11-42: rows ordered by (userId, itemId), ids left as they are, stable; -> <name>_sorted<ext>."""
    user, item, rating = _read_triples(src)
    order = np.lexsort((item, user))  # (stable: rows equal in both keys keep their file order, like sorted())
    base, ext = os.path.splitext(src)
    dst = dst or base + "_sorted" + ext
    _write(dst, user[order], item[order], rating[order])
    return dst


def to_npy(src, dst=None):
    """convert_to_np.py:6-14: a CSV of floats (the trainer's p / q / bias files) as a float64 .npy next to it."""
    dst = dst or os.path.splitext(src)[0] + ".npy"
    np.save(dst, np.genfromtxt(src, delimiter=","))
    return dst


def _read_netflix(path):
    """The raw Netflix split files (map_netflix.py:9-13): no header, `user item  rating` separated by single spaces -- two in front
    of the rating, so the rating is the FOURTH field of a line split at every space."""
    users, items, ratings = [], [], []
    with open(path) as fh:
        for line in fh:
            parts = line.rstrip("\r\n").split(" ")
            if len(parts) < 4 or not parts[0]:
                continue
            users.append(int(parts[0]))
            items.append(int(parts[1]))
            ratings.append(float(parts[3]))
    return np.asarray(users, np.int64), np.asarray(items, np.int64), np.asarray(ratings, np.float64)


def map_netflix(train_src, test_src, train_dst=None, test_dst=None):
    """map_netflix.py:15-28: ONE id mapping for both files, built from the training file in first-seen order; test rows whose user or
    item the training file never names are dropped (add_missing=False, map_items.py:39-52); both outputs stably sorted by user."""
    tu, ti, tr = _read_netflix(train_src)
    uniq_u, first_u = np.unique(tu, return_index=True)
    uniq_i, first_i = np.unique(ti, return_index=True)
    id_u = np.empty(len(uniq_u), np.int64)
    id_u[np.argsort(first_u, kind="stable")] = np.arange(1, len(uniq_u) + 1)
    id_i = np.empty(len(uniq_i), np.int64)
    id_i[np.argsort(first_i, kind="stable")] = np.arange(1, len(uniq_i) + 1)
    su, si, sr = _read_netflix(test_src)
    pu, pi = np.searchsorted(uniq_u, su), np.searchsorted(uniq_i, si)
    known_u = (pu < len(uniq_u)) & (uniq_u[np.minimum(pu, max(len(uniq_u) - 1, 0))] == su) if len(uniq_u) else np.zeros(len(su), bool)
    known_i = (pi < len(uniq_i)) & (uniq_i[np.minimum(pi, max(len(uniq_i) - 1, 0))] == si) if len(uniq_i) else np.zeros(len(si), bool)
    # (the reference looks the user up first and skips the row at once: a row with an unknown user AND an unknown item counts as a user)
    missing_users, missing_items = int((~known_u).sum()), int((known_u & ~known_i).sum())
    if missing_users:
        print("Skipped %d rows because of missing users" % missing_users)
    if missing_items:
        print("Skipped %d rows because of missing items" % missing_items)
    keep = known_u & known_i
    base = os.path.dirname(os.path.abspath(train_src))
    train_dst = train_dst or os.path.join(base, "ratings_mapped_train.csv")
    test_dst = test_dst or os.path.join(base, "ratings_mapped_test.csv")
    mu, mi = id_u[np.searchsorted(uniq_u, tu)], id_i[np.searchsorted(uniq_i, ti)]
    order = np.argsort(mu, kind="stable")
    _write(train_dst, mu[order], mi[order], tr[order])
    mu, mi, mr = id_u[pu[keep]], id_i[pi[keep]], sr[keep]
    order = np.argsort(mu, kind="stable")
    _write(test_dst, mu[order], mi[order], mr[order])
    return train_dst, test_dst


def split(src, test_fraction=0.2, seed=42, train_dst=None, test_dst=None, keep_users=False):
    """split_to_test_train.py:39-49,69-78 (split_true): random.seed(seed), ONE shuffle of all rows, the first
    int(n * (1 - test_fraction)) rows train, the rest test, both stably sorted by user -- the same files as the reference's
    script.  keep_users=True (not the reference's behaviour) moves one rating of every user that ended up without a training
    rating back from test to train, so that the test file never names a user the model has no ratings for."""
    user, item, rating = _read_triples(src)
    n = len(user)
    order = list(range(n))
    random.Random(seed).shuffle(order)  # the algorithm behind the reference's module-level random.seed / random.shuffle
    order = np.asarray(order, np.int64)
    n_train = int(n * (1 - test_fraction))
    train_idx, test_idx = order[:n_train], order[n_train:]
    if keep_users and n:
        missing = np.setdiff1d(user[test_idx], user[train_idx])
        if len(missing):
            pos = np.asarray([np.flatnonzero(user[test_idx] == u)[0] for u in missing], np.int64)
            train_idx = np.concatenate([train_idx, test_idx[pos]])
            test_idx = np.delete(test_idx, pos)
    base = os.path.splitext(src)[0]
    train_dst, test_dst = train_dst or base + "_train.csv", test_dst or base + "_test.csv"
    for dst, idx in ((train_dst, train_idx), (test_dst, test_idx)):
        idx = idx[np.argsort(user[idx], kind="stable")]
        _write(dst, user[idx], item[idx], rating[idx])
    return train_dst, test_dst


def write_config(path, total_iterations=5000, n_factors=50, learning_rate=0.01, seed=42, P_reg=0.02, Q_reg=0.02,
                 user_bias_reg=0.02, item_bias_reg=0.02):
    """create_config.py:10-19: `0 %d %d %f %d %f %f %f %f`, no newline."""
    with open(path, "w") as fh:
        fh.write("0 %d %d %f %d %f %f %f %f" % (total_iterations, n_factors, learning_rate, seed, P_reg, Q_reg,
                                               user_bias_reg, item_bias_reg))
    return path


def main(argv=None):
    ap = argparse.ArgumentParser(prog="cu2rec_amd.prep")
    sub = ap.add_subparsers(dest="cmd", required=True)
    m = sub.add_parser("map")
    m.add_argument("src")
    m.add_argument("dst", nargs="?")
    s = sub.add_parser("split")
    s.add_argument("src")
    s.add_argument("--test-fraction", type=float, default=0.2)
    s.add_argument("--seed", type=int, default=42)
    s.add_argument("--keep-users", action="store_true", help="every user keeps at least one training rating (not the reference's behaviour)")
    so = sub.add_parser("sort")
    so.add_argument("src")
    so.add_argument("dst", nargs="?")
    tn = sub.add_parser("to-npy")
    tn.add_argument("files", nargs="+")
    nf = sub.add_parser("map-netflix")
    nf.add_argument("train")
    nf.add_argument("test")
    c = sub.add_parser("config")
    c.add_argument("dst")
    c.add_argument("--iters", type=int, default=5000)
    c.add_argument("--factors", type=int, default=50)
    c.add_argument("--lr", type=float, default=0.01)
    c.add_argument("--seed", type=int, default=42)
    c.add_argument("--reg", type=float, default=0.02)
    args = ap.parse_args(argv)
    if args.cmd == "map":
        print("%s: %d users, %d items, %d ratings" % map_ids(args.src, args.dst))
    elif args.cmd == "sort":
        print(sort_ratings(args.src, args.dst))
    elif args.cmd == "to-npy":
        print(" ".join(to_npy(f) for f in args.files))
    elif args.cmd == "map-netflix":
        print("%s %s" % map_netflix(args.train, args.test))
    elif args.cmd == "split":
        print("%s %s" % split(args.src, args.test_fraction, args.seed, keep_users=args.keep_users))
    else:
        print(write_config(args.dst, args.iters, args.factors, args.lr, args.seed, args.reg, args.reg, args.reg, args.reg))
    return 0


if __name__ == "__main__":
    sys.exit(main())
