#!/usr/bin/env python3
"""Generate tests/golden/prep_* from the REFERENCE's own preprocessing scripts.

TEST INFRASTRUCTURE.  Runs only in the build container (needs /root/reference/preprocessing).  Nothing here is imported at
test time; tests/test_prep.py reads the files this script writes.  The reference's scripts are executed as they are, as
child processes, on a small raw file this script makes up; only their OUTPUTS are kept:

  prep_raw.csv            the made-up input: `userId,movieId,rating,timestamp`, sparse unordered ids, 60 ratings
  prep_ref_mapped.csv     preprocessing/map_items.py prep_raw.csv           (ids 1..N in first-seen order, rows sorted by user)
  prep_ref_train.csv      preprocessing/split_to_test_train.py prep_ref_mapped.csv 0.2   (random.seed(42) shuffle, first 80 %,
  prep_ref_test.csv         both halves stably sorted by user)
  prep_ref_config.cfg     preprocessing/create_config.py -n 500 -f 50 -l 0.005 -s 7 -p 0.03 -q 0.04 -u 0.05 -i 0.06
  prep_ref_sorted.csv     preprocessing/sort_ratings.py prep_raw.csv           (rows sorted by (userId, itemId), ids as they are)
  prep_matrix.csv         a made-up 5 x 4 float matrix in the trainer's `%f` output format (util.cu:86-97), and
  prep_ref_matrix.npy       preprocessing/convert_to_np.py prep_matrix.csv     (np.genfromtxt -> np.save: float64)
  prep_netflix_train.txt  made-up raw Netflix split files (`user item  rating`, no header; the test file names two users and an item
  prep_netflix_test.txt     the training file does not)
  prep_ref_netflix_train.csv, prep_ref_netflix_test.csv   preprocessing/map_netflix.py on those (run where its hard-wired relative
                          paths point: ../data/datasets/netflix/)
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")
REF = "/root/reference/preprocessing"


def make_raw(path):
    rng = np.random.RandomState(20241004)
    users = [907, 12, 4400, 31, 5, 77777, 640]           # sparse ids, first seen in this order (not ascending)
    items = [2571, 1, 318, 99999, 50, 260, 4993, 7, 1210, 33, 858]
    rows, seen = [], set()
    while len(rows) < 60:
        u, i = users[rng.randint(len(users))], items[rng.randint(len(items))]
        if (u, i) in seen:
            continue
        seen.add((u, i))
        rows.append((u, i, float(rng.choice([0.5, 1.0, 1.5, 2.0, 2.5, 3.0, 3.5, 4.0, 4.5, 5.0])), 1260759144 + len(rows)))
    with open(path, "w") as fh:
        fh.write("userId,movieId,rating,timestamp\n")
        for r in rows:
            fh.write("%d,%d,%s,%d\n" % r)


def make_netflix(train_path, test_path):
    rng = np.random.RandomState(20241005)
    users = [1488844, 822109, 885013, 30878, 823519, 893988, 124105, 1248029]
    items = [1, 17770, 8, 30, 4500, 571, 175]
    seen, rows = set(), []
    while len(rows) < 48:
        u, i = users[rng.randint(len(users))], items[rng.randint(len(items))]
        if (u, i) not in seen:
            seen.add((u, i))
            rows.append((u, i, int(rng.randint(1, 6))))
    with open(train_path, "w") as fh:
        for u, i, r in rows:
            fh.write("%d %d  %d\n" % (u, i, r))
    test = [(rows[k][0], rows[(k * 7 + 3) % len(rows)][1], int(rng.randint(1, 6))) for k in range(0, 24, 2)]
    test += [(2000000, 1, 4), (30878, 9999, 2), (2000001, 9999, 5), (822109, 8, 3)]  # unknown user, unknown item, both, known
    with open(test_path, "w") as fh:
        for u, i, r in test:
            fh.write("%d %d  %d\n" % (u, i, r))


def main():
    if not os.path.isdir(REF):
        sys.exit("needs the reference tree at /root/reference (build container only)")
    os.makedirs(GOLD, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        raw = os.path.join(tmp, "prep_raw.csv")
        make_raw(raw)
        run = lambda *a: subprocess.run([sys.executable] + list(a), check=True, cwd=tmp)
        run(os.path.join(REF, "map_items.py"), raw)                                        # -> prep_raw_mapped.csv
        mapped = os.path.join(tmp, "prep_raw_mapped.csv")
        run(os.path.join(REF, "split_to_test_train.py"), mapped, "0.2")                    # -> *_train.csv, *_test.csv
        run(os.path.join(REF, "create_config.py"), os.path.join(tmp, "c.cfg"), "-n", "500", "-f", "50", "-l", "0.005", "-s", "7",
            "-p", "0.03", "-q", "0.04", "-u", "0.05", "-i", "0.06")
        run(os.path.join(REF, "sort_ratings.py"), raw)                                     # -> prep_raw_sorted.csv
        rng = np.random.RandomState(20241006)
        with open(os.path.join(tmp, "prep_matrix.csv"), "w") as fh:
            for row in rng.normal(0, 0.3, (5, 4)):
                fh.write(",".join("%f" % v for v in row) + "\n")
        run(os.path.join(REF, "convert_to_np.py"), os.path.join(tmp, "prep_matrix.csv"))   # -> prep_matrix.npy
        for src, dst in (("prep_raw_sorted.csv", "prep_ref_sorted.csv"), ("prep_matrix.csv", "prep_matrix.csv"), ("prep_matrix.npy", "prep_ref_matrix.npy")):
            shutil.copyfile(os.path.join(tmp, src), os.path.join(GOLD, dst))
            print("wrote tests/golden/" + dst)
        for src, dst in (("prep_raw.csv", "prep_raw.csv"), ("prep_raw_mapped.csv", "prep_ref_mapped.csv"),
                         ("prep_raw_mapped_train.csv", "prep_ref_train.csv"), ("prep_raw_mapped_test.csv", "prep_ref_test.csv"),
                         ("c.cfg", "prep_ref_config.cfg")):
            shutil.copyfile(os.path.join(tmp, src), os.path.join(GOLD, dst))
            print("wrote tests/golden/" + dst)
        # map_netflix.py reads and writes ../data/datasets/netflix/ relative to the directory it is run in
        nf = os.path.join(tmp, "data", "datasets", "netflix")
        os.makedirs(nf)
        os.makedirs(os.path.join(tmp, "pp"))
        make_netflix(os.path.join(nf, "netflix_train.txt"), os.path.join(nf, "netflix_test.txt"))
        subprocess.run([sys.executable, os.path.join(REF, "map_netflix.py")], check=True, cwd=os.path.join(tmp, "pp"))
        for src, dst in (("netflix_train.txt", "prep_netflix_train.txt"), ("netflix_test.txt", "prep_netflix_test.txt"),
                         ("ratings_mapped_train.csv", "prep_ref_netflix_train.csv"), ("ratings_mapped_test.csv", "prep_ref_netflix_test.csv")):
            shutil.copyfile(os.path.join(nf, src), os.path.join(GOLD, dst))
            print("wrote tests/golden/" + dst)


if __name__ == "__main__":
    main()
