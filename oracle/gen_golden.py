#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/ from the REFERENCE itself.

TEST INFRASTRUCTURE.  Runs only in the build container (needs oracle/_ref/* built by
oracle/build_ref.py from /root/reference).  Nothing here is imported at test time; the tests
read the JSON / CSV files this script writes.

What is produced
  toy_*.csv                 the toy rating files the reference's own tests use
                            (data/test/*.csv; values restated from tests/test_util.cu:98-189
                            and the other fixtures) -- inputs, i.e. data.
  ref_lr0_known_answers.json  stdout TRAIN/TEST lines of the UNMODIFIED reference CPU twin
                            (oracle/_ref/mf_cpu) with learning_rate=0: a pure function of
                            reader + CSR + seed-42 init + loss (mf_sequential.cu:146-201).
  ref_sgd_golden.json       P/Q/bias dumps + stdout lines of oracle/_ref/mf_cpu_philox: the
                            reference CPU twin whose random_device draw (mf_sequential.cu:109-112)
                            is replaced by orc_sample(); everything else -- reader, CSR, init,
                            the update arithmetic, the float-accumulated loss -- is the
                            reference's own compiled code.
"""
import base64
import hashlib
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
REF_ML_SMALL = "/root/reference/ratings_mapped.csv"

# (userId, itemId, rating) -- data/test/test_ratings.csv as pinned by tests/test_util.cu:98-142
TOY = {
    "toy_ratings.csv": [(1, 1, 1), (1, 2, 1), (1, 3, 1), (1, 5, 5), (2, 1, 3), (2, 2, 3), (2, 3, 3), (3, 1, 4),
                        (3, 2, 4), (3, 3, 4), (4, 1, 5), (4, 2, 5), (4, 3, 5), (5, 2, 2), (5, 4, 4), (5, 5, 4),
                        (6, 4, 5), (6, 5, 5)],
    # tests/test_util.cu:146-189: user 2 absent
    "toy_missing_user.csv": [(1, 1, 1), (1, 2, 1), (1, 3, 1), (1, 5, 5), (3, 1, 4), (3, 2, 4), (3, 3, 4), (4, 1, 5),
                             (4, 2, 5), (4, 3, 5), (5, 2, 2), (5, 4, 4), (5, 5, 4), (6, 4, 5), (6, 5, 5)],
    "toy_ratings2.csv": [(1, 2, 1), (1, 3, 1), (1, 5, 5), (2, 1, 3), (2, 3, 3), (3, 1, 4), (3, 2, 4), (4, 1, 5),
                         (4, 2, 5), (4, 3, 5), (4, 4, 1), (5, 2, 2), (5, 4, 4), (5, 5, 4), (6, 4, 5)],
    "toy_ratings3.csv": [(1, 2, 1), (1, 3, 1), (1, 5, 5), (2, 1, 3), (2, 3, 3), (3, 1, 4), (3, 2, 4), (4, 1, 5),
                         (4, 2, 5), (4, 3, 5), (4, 4, 1), (5, 2, 2), (5, 4, 4), (5, 5, 4), (6, 4, 5), (7, 1, 5),
                         (7, 2, 5), (7, 3, 5), (7, 4, 1), (7, 5, 1), (8, 1, 1), (8, 2, 1), (8, 3, 1), (8, 4, 5),
                         (8, 5, 5)],
}


def write_toys():
    for name, rows in TOY.items():
        with open(os.path.join(GOLD, name), "w") as fh:
            fh.write("userId,itemId,rating\n")
            # no trailing newline, like the reference's fixtures
            fh.write("\n".join("%d,%d,%.1f" % r for r in rows))
    # data/test/test_user_ratings.csv: spaces after the commas
    with open(os.path.join(GOLD, "toy_user_spaces.csv"), "w") as fh:
        fh.write("userId,itemId,rating\n1, 1, 1.0\n1, 2, 1.0\n1, 4, 5.0")
    # data/test/test_Q.csv (tests/test_util.cu:36-46)
    with open(os.path.join(GOLD, "toy_Q.csv"), "w") as fh:
        fh.write("0, 1.0, 2.0, 3.0, 4.0\n5.0, 6.0, 7.0, 8.0, 9.0\n")


LINE = re.compile(r"^(TRAIN|TEST): Iteration (\d+) MAE: (\S+) RMSE: (\S+)")


def run_ref(binary, cfg_fields, train, test, dump=None):
    with tempfile.TemporaryDirectory() as td:
        cfg = os.path.join(td, "c.cfg")
        with open(cfg, "w") as fh:
            fh.write(" ".join(str(x) for x in cfg_fields) + "\n")
        env = dict(os.environ)
        if dump:
            env["ORC_DUMP"] = dump
        out = subprocess.run([binary, "-c", cfg, train, test], check=True, stdout=subprocess.PIPE, env=env,
                             text=True).stdout
    lines = []
    for l in out.split("\n"):
        m = LINE.match(l)
        if m:
            lines.append({"split": m.group(1), "iteration": int(m.group(2)), "mae": m.group(3), "rmse": m.group(4)})
    return lines


def read_dump(path):
    raw = open(path, "rb").read()
    rows, cols, f = struct.unpack("iii", raw[:12])
    a = np.frombuffer(raw[12:], dtype=np.float32)
    P = a[:rows * f]
    Q = a[rows * f:rows * f + cols * f]
    ub = a[rows * f + cols * f:rows * f + cols * f + rows]
    ib = a[rows * f + cols * f + rows:]
    assert len(ib) == cols
    return rows, cols, f, P, Q, ub, ib


def pack(arr, full):
    b = np.ascontiguousarray(arr, np.float32).tobytes()
    d = {"sha256": hashlib.sha256(b).hexdigest(), "n": int(arr.size),
         "head": [float(np.float32(x)) for x in arr[:8]]}
    if full:
        d["f32_le_b64"] = base64.b64encode(b).decode()
    return d


def main():
    os.makedirs(GOLD, exist_ok=True)
    write_toys()
    mf_cpu = os.path.join(HERE, "_ref", "mf_cpu")
    mf_phx = os.path.join(HERE, "_ref", "mf_cpu_philox")
    if not (os.path.exists(mf_cpu) and os.path.exists(mf_phx)):
        sys.exit("run oracle/build_ref.py first")
    toy = os.path.join(GOLD, "toy_ratings.csv")
    toy2 = os.path.join(GOLD, "toy_ratings2.csv")
    toy3 = os.path.join(GOLD, "toy_ratings3.csv")
    have_ml = os.path.exists(REF_ML_SMALL)

    # ---- lr = 0 known answers from the unmodified reference
    lr0 = []
    for f in (2, 10, 50, 100):
        fields = [0, 1, f, 0.0, 42, 0.02, 0.02, 0.02, 0.02]
        lr0.append({"train": "toy_ratings.csv", "test": "toy_ratings2.csv", "cfg": fields,
                    "lines": run_ref(mf_cpu, fields, toy, toy2)})
        if have_ml:
            lr0.append({"train": "ML_SMALL", "test": "ML_SMALL", "cfg": fields,
                        "lines": run_ref(mf_cpu, fields, REF_ML_SMALL, REF_ML_SMALL)})
    with open(os.path.join(GOLD, "ref_lr0_known_answers.json"), "w") as fh:
        json.dump({"generator": "oracle/gen_golden.py", "binary": "oracle/_ref/mf_cpu (unmodified reference)",
                   "cases": lr0}, fh, indent=1)

    # ---- SGD goldens from the philox-sampler build of the reference
    cases = []
    grid = [
        # cur, total, f, lr, seed, regs...
        ("toy_ratings.csv", "toy_ratings2.csv", [0, 1, 1, 0.07, 1, 0.1, 0.1, 0.1, 0.1]),      # tests/test_sgd.cu hyper-params
        ("toy_ratings.csv", "toy_ratings.csv", [0, 10, 2, 0.001, 42, 0.1, 0.1, 0.1, 0.1]),    # tests/test_training.cu
        ("toy_ratings.csv", "toy_ratings2.csv", [0, 10, 2, 0.1, 42, 0.2, 0.1, 0.1, 0.1]),     # data/test/train.cfg
        ("toy_ratings.csv", "toy_ratings2.csv", [0, 100, 10, 0.01, 42, 0.02, 0.02, 0.02, 0.02]),
        ("toy_ratings.csv", "toy_ratings2.csv", [0, 1000, 10, 0.01, 7, 0.02, 0.02, 0.02, 0.02]),
        ("toy_ratings.csv", "toy_ratings2.csv", [0, 50, 50, 0.01, 42, 0.02, 0.02, 0.02, 0.02]),
        ("toy_ratings.csv", "toy_ratings2.csv", [0, 50, 100, 0.01, 42, 0.02, 0.02, 0.02, 0.02]),
        ("toy_ratings.csv", "toy_ratings2.csv", [0, 20, 128, 0.01, 42, 0.02, 0.02, 0.02, 0.02]),
        ("toy_ratings.csv", "toy_ratings2.csv", [0, 20, 300, 0.01, 42, 0.02, 0.02, 0.02, 0.02]),
        ("toy_ratings.csv", "toy_ratings2.csv", [5, 20, 10, 0.01, 42, 0.02, 0.02, 0.02, 0.02]),  # resumed stream
        ("toy_missing_user.csv", "toy_ratings2.csv", [0, 30, 10, 0.01, 42, 0.02, 0.02, 0.02, 0.02]),
        ("toy_ratings3.csv", "toy_ratings2.csv", [0, 30, 10, 0.05, 42, 0.02, 0.02, 0.02, 0.02]),
    ]
    if have_ml:
        grid += [
            ("ML_SMALL", "ML_SMALL", [0, 1, 10, 0.01, 42, 0.02, 0.02, 0.02, 0.02]),
            ("ML_SMALL", "ML_SMALL", [0, 20, 10, 0.01, 42, 0.02, 0.02, 0.02, 0.02]),
            ("ML_SMALL", "ML_SMALL", [0, 5, 50, 0.01, 42, 0.02, 0.02, 0.02, 0.02]),
            ("ML_SMALL", "ML_SMALL", [0, 3, 100, 0.01, 42, 0.02, 0.02, 0.02, 0.02]),
        ]
    with tempfile.TemporaryDirectory() as td:
        for train, test, fields in grid:
            tr = REF_ML_SMALL if train == "ML_SMALL" else os.path.join(GOLD, train)
            te = REF_ML_SMALL if test == "ML_SMALL" else os.path.join(GOLD, test)
            dump = os.path.join(td, "dump.bin")
            lines = run_ref(mf_phx, fields, tr, te, dump)
            rows, cols, f, P, Q, ub, ib = read_dump(dump)
            full = train != "ML_SMALL"
            cases.append({"train": train, "test": test, "cfg": fields, "rows": rows, "cols": cols, "f": f,
                          "lines": lines, "P": pack(P, full), "Q": pack(Q, full), "user_bias": pack(ub, full),
                          "item_bias": pack(ib, full)})
            print("golden:", train, fields, lines[-1] if lines else None)
    with open(os.path.join(GOLD, "ref_sgd_golden.json"), "w") as fh:
        json.dump({"generator": "oracle/gen_golden.py",
                   "binary": "oracle/_ref/mf_cpu_philox (reference CPU twin, sampler lines 109-112 -> orc_sample)",
                   "ML_SMALL": "the reference's bundled ratings_mapped.csv (610 x 9724, 100836 ratings); not shipped",
                   "cases": cases}, fh, indent=1)

    # ---- Tier 3: the UNMODIFIED reference binary's own run-to-run band.  Its sampler draws from a fresh
    # std::random_device per update (mf_sequential.cu:109-112: no seed reaches it; inclusive range), so identical runs differ;
    # the oracle's trajectory (counter-based sampler, half-open range) must lie inside that band, statistically
    # (tests/test_oracle_golden.py::test_oracle_trajectory_lies_in_the_unmodified_reference_band).
    if have_ml:
        fields = [0, 500, 10, 0.01, 42, 0.02, 0.02, 0.02, 0.02]
        runs = [run_ref(mf_cpu, fields, REF_ML_SMALL, REF_ML_SMALL) for _ in range(TIER3_RUNS)]
        band = {}
        for it in (1, 500):
            vals = {"rmse": [], "mae": []}
            for lines in runs:
                for l in lines:
                    if l["split"] == "TRAIN" and l["iteration"] == it:
                        vals["rmse"].append(float(l["rmse"]))
                        vals["mae"].append(float(l["mae"]))
            assert len(vals["rmse"]) == TIER3_RUNS, (it, vals)
            band[str(it)] = {k: {"min": min(v), "max": max(v), "runs": v} for k, v in vals.items()}
            print("tier 3 band, iteration", it, band[str(it)]["rmse"]["min"], band[str(it)]["rmse"]["max"])
        with open(os.path.join(GOLD, "ref_tier3_band.json"), "w") as fh:
            json.dump({"generator": "oracle/gen_golden.py", "binary": "oracle/_ref/mf_cpu (unmodified reference)",
                       "train": "ML_SMALL", "test": "ML_SMALL", "cfg": fields, "runs": TIER3_RUNS, "band": band}, fh, indent=1)


TIER3_RUNS = 8


if __name__ == "__main__":
    main()
