"""Data preparation equivalents of the reference's preprocessing scripts, so raw
`userId,itemId,rating[,...]` files can be taken to the trainer's input format without the reference tree.
Only the FILE FORMATS matter (ids 1..N in first-seen order, rows sorted by user, one header line,
9-field config); bit parity with Python's `random` shuffle in the reference's splitter is not a goal.

  python -m cu2rec_amd.prep map    ratings.csv            -> ratings_mapped.csv   (preprocessing/map_items.py:21-89)
  python -m cu2rec_amd.prep split  ratings_mapped.csv     -> *_train.csv, *_test.csv (split_to_test_train.py:39-78)
  python -m cu2rec_amd.prep config out.cfg [--iters ...]  -> 9-field config       (create_config.py:10-32)
"""
import argparse
import os
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
    return np.asarray(users, np.int64), np.asarray(items, np.int64), np.asarray(ratings, np.float32)


def _first_seen_ids(values):
    """Dense 1-based ids in order of first appearance (map_items.py:40-54)."""
    uniq, first = np.unique(values, return_index=True)
    order = np.argsort(first, kind="stable")
    new_id = np.empty(len(uniq), np.int64)
    new_id[order] = np.arange(1, len(uniq) + 1)
    return new_id[np.searchsorted(uniq, values)]


def _write(path, user, item, rating):
    with open(path, "w") as fh:
        fh.write("userId,itemId,rating\n")
        np.savetxt(fh, np.column_stack([user, item, rating]), fmt=["%d", "%d", "%g"], delimiter=",")


def map_ids(src, dst=None):
    """Remap user and item ids to 1..N (first-seen order) and sort rows by user, stably (map_items.py:21-89)."""
    user, item, rating = _read_triples(src)
    user, item = _first_seen_ids(user), _first_seen_ids(item)
    order = np.argsort(user, kind="stable")
    dst = dst or os.path.splitext(src)[0] + "_mapped.csv"
    _write(dst, user[order], item[order], rating[order])
    return dst, int(user.max(initial=0)), int(item.max(initial=0)), len(user)


def split(src, test_fraction=0.2, seed=42, train_dst=None, test_dst=None):
    """Random per-rating split, both halves sorted by user (split_to_test_train.py:39-49,76-78).  Every user keeps
    at least one training rating, so the test file never names a user the model does not have."""
    user, item, rating = _read_triples(src)
    rng = np.random.RandomState(seed)
    is_test = rng.rand(len(user)) < test_fraction
    first = np.zeros(len(user), bool)
    first[np.unique(user, return_index=True)[1]] = True
    is_test &= ~first
    base = os.path.splitext(src)[0]
    train_dst, test_dst = train_dst or base + "_train.csv", test_dst or base + "_test.csv"
    for dst, mask in ((train_dst, ~is_test), (test_dst, is_test)):
        order = np.argsort(user[mask], kind="stable")
        _write(dst, user[mask][order], item[mask][order], rating[mask][order])
    return train_dst, test_dst


def write_config(path, total_iterations=5000, n_factors=50, learning_rate=0.01, seed=42, P_reg=0.02, Q_reg=0.02,
                 user_bias_reg=0.02, item_bias_reg=0.02):
    """create_config.py:10-19: `0 %d %d %f %d %f %f %f %f`."""
    with open(path, "w") as fh:
        fh.write("0 %d %d %f %d %f %f %f %f\n" % (total_iterations, n_factors, learning_rate, seed, P_reg, Q_reg,
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
    elif args.cmd == "split":
        print("%s %s" % split(args.src, args.test_fraction, args.seed))
    else:
        print(write_config(args.dst, args.iters, args.factors, args.lr, args.seed, args.reg, args.reg, args.reg, args.reg))
    return 0


if __name__ == "__main__":
    sys.exit(main())
