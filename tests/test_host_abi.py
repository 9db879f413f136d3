"""CPU-only checks of the product's host side through the C ABI (no GPU compute calls):
the library loads, exports every symbol include/cu2rec_amd.h declares, and its reader / CSR
builder / init / sampler / writer agree with the oracle and the reference's known answers."""
import os
import re

import numpy as np
import pytest

import cu2rec_amd as cu
from cu2rec_amd import _lib
from conftest import GOLDEN, ROOT
from oracle import oracle as orc


def test_header_symbols_are_exported():
    header = open(os.path.join(ROOT, "include", "cu2rec_amd.h")).read()
    declared = set(re.findall(r"\b(cu2rec_[a-z0-9_]+)\s*\(", header))
    declared -= {"cu2rec_status"}
    assert len(declared) >= 35
    L = cu.lib()
    for name in sorted(declared):
        assert hasattr(L, name), "missing symbol " + name
        assert name in _lib.SIGNATURES, "no ctypes signature for " + name
    assert L.cu2rec_version() == 100


def test_compute_fails_loudly_without_gpu():
    if cu.device_count() > 0:
        pytest.skip("a GPU is present")
    m = cu.createSparseMatrix(os.path.join(GOLDEN, "toy_ratings.csv"))
    with pytest.raises(cu.Cu2recError) as e:
        cu.DeviceCSR(m)
    assert e.value.status == -4  # CU2REC_ENODEVICE: no CPU fallback
    with pytest.raises(cu.Cu2recError):
        cu.Model(m.rows, m.cols, 4, m.global_bias)


def test_csr_goldens():
    m = cu.createSparseMatrix(os.path.join(GOLDEN, "toy_ratings.csv"))  # tests/test_util.cu:20-34,98-142
    assert (m.rows, m.cols, m.nnz) == (6, 5, 18) and abs(m.global_bias - 3.5556) < 1e-3
    assert m.indptr.tolist() == [0, 4, 7, 10, 13, 16, 18]
    assert m.indices.tolist() == [0, 1, 2, 4, 0, 1, 2, 0, 1, 2, 0, 1, 2, 1, 3, 4, 3, 4]
    assert m.data.tolist() == [1, 1, 1, 5, 3, 3, 3, 4, 4, 4, 5, 5, 5, 2, 4, 4, 5, 5]
    m = cu.createSparseMatrix(os.path.join(GOLDEN, "toy_missing_user.csv"))  # tests/test_util.cu:146-189
    assert m.indptr.tolist() == [0, 4, 4, 7, 10, 13, 15]
    m = cu.createSparseMatrix(os.path.join(GOLDEN, "toy_user_spaces.csv"))
    assert m.indices.tolist() == [0, 1, 3] and m.data.tolist() == [1, 1, 5]


def test_read_array_golden():
    a = cu.read_array(os.path.join(GOLDEN, "toy_Q.csv"))  # tests/test_util.cu:36-46
    assert a.shape == (2, 5) and a.ravel().tolist() == list(range(10))


def test_reader_matches_oracle_on_random_files(tmp_path):
    rng = np.random.RandomState(3)
    for trial in range(5):
        n_users, n_items = int(rng.randint(1, 60)), int(rng.randint(1, 40))
        rows = []
        for u in range(1, n_users + 1):
            if rng.rand() < 0.15:
                continue
            for i in rng.permutation(n_items)[: rng.randint(1, n_items + 1)] + 1:
                rows.append((u, int(i), float(rng.choice([0.5, 1, 1.5, 2, 2.5, 3, 3.5, 4, 4.5, 5]))))
        p = tmp_path / ("r%d.csv" % trial)
        sep = ", " if trial % 2 else ","
        p.write_text("userId,itemId,rating\n" + "\n".join("%d%s%d%s%s" % (u, sep, i, sep, repr(r)) for u, i, r in rows)
                     + ("\n" if trial % 3 == 0 else ""))
        if not rows:
            continue
        a, b = cu.createSparseMatrix(str(p)), orc.read_csv(str(p))
        assert (a.rows, a.cols, a.nnz) == (b.rows, b.cols, b.nnz)
        assert np.float32(a.global_bias) == np.float32(b.global_bias)
        np.testing.assert_array_equal(a.indptr, b.indptr)
        np.testing.assert_array_equal(a.indices, b.indices)
        np.testing.assert_array_equal(a.data, b.data)


def test_unsorted_and_missing_files_are_errors(tmp_path):
    p = tmp_path / "bad.csv"
    p.write_text("userId,itemId,rating\n2,1,3.0\n1,1,4.0\n")
    with pytest.raises(cu.Cu2recError):
        cu.createSparseMatrix(str(p))
    with pytest.raises(cu.Cu2recError) as e:
        cu.readCSV(str(tmp_path / "nope.csv"))
    assert e.value.status == -2


def test_init_matches_reference_known_answers():
    np.testing.assert_array_equal(cu.initialize_normal_array(6, 10),
                                  np.array([0.122192137, -0.051696416, 0.086963594, 0.0721332654, 0.158855632,
                                            0.161821708], np.float32))
    for f in (1, 2, 50, 100, 300):
        np.testing.assert_array_equal(cu.initialize_normal_array(1000, f), orc.normal_fill(1000, f))
    np.testing.assert_array_equal(cu.initialize_normal_array(64, 7, 0.5, 2.0, 9), orc.normal_fill(64, 7, 0.5, 2.0, 9))


def test_sampler_matches_oracle():
    rng = np.random.RandomState(5)
    L = cu.lib()
    for _ in range(5000):
        seed, user, it = int(rng.randint(0, 2**31)), int(rng.randint(0, 2**31)), int(rng.randint(0, 2**31))
        low, n = int(rng.randint(0, 10**6)), int(rng.randint(1, 20000))
        assert L.cu2rec_sampler_draw(seed, user, it) == orc.lib().orc_draw(seed, user, it)
        assert L.cu2rec_sampler_index(seed, user, it, low, low + n) == orc.sample(seed, user, it, low, low + n)


def test_config_roundtrip_and_defaults(tmp_path):
    c = cu.default_config()
    assert (c.total_iterations, c.n_factors, c.seed, c.n_threads, c.check_error) == (5000, 50, 42, 32, 500)
    assert abs(c.learning_rate - 0.01) < 1e-9 and abs(c.P_reg - 0.02) < 1e-9 and c.patience == 2.0
    p = tmp_path / "t.cfg"
    p.write_text("0 100 10 0.0001 42 0.2 0.1 0.1 0.1")  # data/test/test_config.cfg
    c = cu.read_config(str(p))
    assert c.total_iterations == 100 and abs(c.P_reg - 0.2) < 1e-7  # tests/test_config.cu:20-24
    q = tmp_path / "u.cfg"
    cu.write_config(str(q), c)
    d, o = cu.read_config(str(q)), orc.read_config(str(q))
    for name, _ in cu.Config._fields_[:9]:
        assert getattr(c, name) == getattr(d, name) == getattr(o, name)
    with pytest.raises(cu.Cu2recError):
        cu.read_config(str(tmp_path / "missing.cfg"))


def test_writer_matches_oracle(tmp_path):
    a = (np.random.RandomState(0).randn(7, 5) * 3).astype(np.float32)
    cu.writeToFile(str(tmp_path), "base", "p", a, 7, 5, 5)
    orc.write_csv(str(tmp_path / "o.csv"), a)
    assert (tmp_path / "base_f5_p.csv").read_text() == (tmp_path / "o.csv").read_text()  # util.cu:86-103
    assert (tmp_path / "base_f5_p.csv").read_text().split("\n")[0].count(",") == 4


def test_shard_plan_and_slice():
    plan = cu.shard_plan(10, 3)
    assert plan.tolist() == [0, 3, 6, 10]
    m = cu.createSparseMatrix(os.path.join(GOLDEN, "toy_ratings.csv"))
    s = m.slice_users(2, 5)
    assert s.indptr.tolist() == [0, 3, 6, 9] and s.nnz == 9 and s.indices.tolist() == m.indices[7:16].tolist()
    whole = [m.slice_users(a, b) for a, b in zip(cu.shard_plan(m.rows, 4)[:-1], cu.shard_plan(m.rows, 4)[1:])]
    assert sum(w.nnz for w in whole) == m.nnz


def test_threaded_reader_binary_cache_and_fallback(tmp_path):
    """files above 1 MB are parsed on several threads (same result as the sequential grammar and as
    the oracle's fscanf restatement); a record that spans lines forces the sequential path; the binary cache round-trips."""
    import ctypes as C
    rng = np.random.RandomState(4)
    n_users = 3000
    rows = []
    for u in range(1, n_users + 1):
        for i in np.sort(rng.choice(5000, rng.randint(20, 60), replace=False)) + 1:
            rows.append("%d,%d,%s" % (u, i, rng.choice(["0.5", "3", "4.5", "5.0", "2.5"])))
    p = tmp_path / "big.csv"
    p.write_text("userId,itemId,rating\n" + "\n".join(rows))  # no trailing newline
    assert p.stat().st_size > (1 << 20)
    want = orc.read_csv(str(p))
    for threads in ("1", "3", "8"):
        os.environ["CU2REC_READER_THREADS"] = threads
        got = cu.createSparseMatrix(str(p))
        np.testing.assert_array_equal(got.indptr, want.indptr)
        np.testing.assert_array_equal(got.indices, want.indices)
        np.testing.assert_array_equal(got.data, want.data)
        assert np.float32(got.global_bias) == np.float32(want.global_bias)
    # a record broken over two lines in the middle: only the sequential grammar accepts it
    broken = rows[: len(rows) // 2] + ["%s,\n%s" % tuple(rows[len(rows) // 2].split(",", 1))] + rows[len(rows) // 2 + 1:]
    q = tmp_path / "broken.csv"
    q.write_text("userId,itemId,rating\n" + "\n".join(broken) + "\n")
    os.environ["CU2REC_READER_THREADS"] = "4"
    got = cu.createSparseMatrix(str(q))
    np.testing.assert_array_equal(got.indices, want.indices)
    del os.environ["CU2REC_READER_THREADS"]
    # binary cache
    L = cu.lib()
    h, h2 = C.c_void_p(), C.c_void_p()
    assert L.cu2rec_ratings_read_csv(str(p).encode(), C.byref(h)) == 0
    assert L.cu2rec_ratings_save_binary(h, str(tmp_path / "big.bin").encode()) == 0
    assert L.cu2rec_ratings_load_binary(str(tmp_path / "big.bin").encode(), C.byref(h2)) == 0
    n1, n2, gb1, gb2 = C.c_int(), C.c_int(), C.c_float(), C.c_float()
    L.cu2rec_ratings_info(h, C.byref(n1), None, None, C.byref(gb1))
    L.cu2rec_ratings_info(h2, C.byref(n2), None, None, C.byref(gb2))
    assert n1.value == n2.value == len(rows) and gb1.value == gb2.value
    indptr, indices, data = np.zeros(n_users + 1, np.int32), np.zeros(len(rows), np.int32), np.zeros(len(rows), np.float32)
    assert L.cu2rec_csr_build(h2, n_users, indptr.ctypes.data_as(C.c_void_p), indices.ctypes.data_as(C.c_void_p),
                              data.ctypes.data_as(C.c_void_p)) == 0
    np.testing.assert_array_equal(indices, want.indices)
    L.cu2rec_ratings_free(h)
    L.cu2rec_ratings_free(h2)
    assert L.cu2rec_ratings_load_binary(str(p).encode(), C.byref(h2)) == -2  # a text file is not a cache
    # a cache whose ids do not fit its own header (corrupt or edited) is refused, not handed to the CSR build
    raw = bytearray((tmp_path / "big.bin").read_bytes())
    n_rec = len(rows)
    header = len(raw) - 12 * n_rec
    for column, bad in ((0, n_users + 5), (0, -3), (1, 1 << 30)):
        mod = bytearray(raw)
        at = header + column * 4 * n_rec + 4 * (n_rec // 2)
        mod[at:at + 4] = int(bad).to_bytes(4, "little", signed=True)
        (tmp_path / "bad.bin").write_bytes(bytes(mod))
        assert L.cu2rec_ratings_load_binary(str(tmp_path / "bad.bin").encode(), C.byref(h2)) == -2


def test_resident_geometry_arithmetic():
    """Planning of the resident Hogwild launches, device independent (cu2rec_amd/csrc/resident.hip): which shapes fit
    the registers + LDS of a 256-CU MI355X and the grid they get."""
    import ctypes as C
    L = cu.lib()

    def geometry(rows, f, cus=256):
        b, r, l = C.c_int(-1), C.c_int(-1), C.c_int(-1)
        ok = L.cu2rec_hogwild_resident_geometry(rows, f, cus, C.byref(b), C.byref(r), C.byref(l))
        return ok, b.value, r.value, l.value

    assert geometry(138493, 100) == (1, 255, 17, 0)     # ML-20M shape: 17 rows per group, all in registers, 255 workgroups
    assert geometry(147000, 128) == (1, 256, 18, 0)     # the most the registers hold at two float4 per lane
    assert geometry(162541, 100) == (1, 254, 20, 4)     # ML-25M shape: 4 of 20 rows in LDS
    assert geometry(204800, 128) == (1, 256, 25, 9)     # registers + 147 KB of LDS
    streamed = L.cu2rec_hogwild_resident_streamed_rows
    assert streamed(204800, 128, 256) == 0 and streamed(138493, 100, 256) == 0
    # one more does not fit: partial residency (round 4) -- the whole chip, 20 resident rows per group (11 in registers, 9 in LDS)
    # and 6 streamed ones
    assert geometry(204801, 128) == (1, 256, 26, 9) and streamed(204801, 128, 256) == 6
    assert geometry(480189, 128) == (1, 256, 59, 9) and streamed(480189, 128, 256) == 39  # Netflix shape, one GPU: 20 of 59 resident
    assert geometry(60024, 128) == (1, 235, 8, 0)       # ... and one of its eight shards
    assert geometry(6040, 50) == (1, 48, 4, 0)          # small sets round up to 4 rows per group on a smaller grid
    assert geometry(385024, 64)[0] == 1 and streamed(385024, 64, 256) == 0 and streamed(385025, 64, 256) == 9
    assert geometry(39 * 4 * 8192, 64)[0] == 1 and geometry(39 * 4 * 8192 + 1, 64)[0] == 0  # at most three streamed users per resident one
    assert streamed(39 * 4 * 8192 + 1, 64, 256) == -1
    assert geometry(20000, 300)[0] == 0                 # five float4 per lane: no resident variant compiled
    assert geometry(1000, 100, cus=0)[0] == 0
    for rows, f in ((138493, 100), (162541, 100), (60024, 128), (6040, 50), (60000, 200), (77777, 256)):
        ok, b, r, l = geometry(rows, f)
        assert ok and b <= 256 and b * 32 * r >= rows > (b - 1) * 32 * r and 0 <= l < r


def test_integration_shim_compiles(tmp_path):
    """The reference-side train() shim shown in INTEGRATION.md (Option B) compiles against include/cu2rec_amd.h."""
    import re
    import subprocess
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```cpp\n(.*?)```", text, re.S).group(1)
    stub = ("namespace config { struct Config { int cur_iterations, total_iterations, n_factors, seed, check_error; "
            "float learning_rate, P_reg, Q_reg, user_bias_reg, item_bias_reg, patience, learning_rate_decay; bool is_train; }; }\n"
            "#include <cstddef>\n")
    src = tmp_path / "training_amd.cpp"
    src.write_text(stub + code)
    res = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert res.returncode == 0, res.stdout


def test_bin_mf_multi_gpu_rank_failure_ends_every_rank(tmp_path):
    """bin/mf -g N (one process per GPU): a rank that dies must not leave the others waiting in a collective for ever.  Rank 1 is
    made to exit with code 7 before it touches anything, rank 0 (the parent) and rank 2 to sit still like ranks inside an
    all-reduce whose peer is gone: the program ends at once, non-zero, and no rank is left behind.  (Fault injection hooks in
    csrc/mf_main.cpp, compiled into the test binary build/test/mf_hooks only -- bin/mf itself ignores the variables; no GPU
    involved.)"""
    import subprocess
    import time
    exe = os.path.join(ROOT, "build", "test", "mf_hooks")
    if not os.path.exists(exe):
        pytest.skip("build/test/mf_hooks not built (make -C cu2rec_amd/csrc test-hooks)")
    env = dict(os.environ, CU2REC_TEST_RANK_EXIT="1:7", CU2REC_TEST_RANK_HANG="0")
    t0 = time.time()
    res = subprocess.run([exe, "-g", "3", "train.csv", "test.csv"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         timeout=60)
    assert time.time() - t0 < 20, "the surviving ranks were not stopped"
    assert res.returncode == 3 and "rank 1 ended abnormally (exit code 7)" in res.stderr, (res.returncode, res.stderr)
    # every child is gone (they die with the parent even if it had been killed itself)
    for _ in range(40):
        out = subprocess.run(["pgrep", "-x", "mf_hooks"], stdout=subprocess.PIPE, text=True).stdout.split()
        if not out:
            break
        time.sleep(0.05)
    assert out == [], out
