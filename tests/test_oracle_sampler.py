"""Sampler stream pins: Random123 known answers and rocRAND's own engine (host-callable header)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import oracle as orc

# Random123 kat_vectors, philox4x32 10 rounds
KAT = [
    ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
     [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
]


@pytest.mark.parametrize("ctr,key,want", KAT)
def test_philox_known_answers(ctr, key, want):
    assert orc.philox(ctr, key) == want


def test_sample_range_and_formula():
    L = orc.lib()
    rng = np.random.RandomState(1)
    for _ in range(2000):
        seed, user, it = int(rng.randint(0, 2**31)), int(rng.randint(0, 10**6)), int(rng.randint(0, 10**5))
        low = int(rng.randint(0, 1000))
        n = int(rng.randint(1, 10000))
        y = L.orc_sample(seed, user, it, low, low + n)
        assert low <= y < low + n
        x = L.orc_draw(seed, user, it)
        u = np.float32(2.0 ** -32) + np.float32(x) * np.float32(2.0 ** -32)
        assert 0.0 < u <= 1.0
        assert y == int(np.ceil(np.float32(u * np.float32(n)))) - 1 + low  # sgd.cu:36-37
    # extremes of the draw
    assert orc.lib().orc_uniform(0) > 0.0 and orc.lib().orc_uniform(0xFFFFFFFF) == 1.0


ROCRAND_SRC = r"""
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_kernel.h>
int main(int argc, char** argv) {
    // argv: seed user iteration (triples)
    for (int i = 1; i + 2 < argc; i += 3) {
        unsigned long long seed = strtoull(argv[i], 0, 10), user = strtoull(argv[i+1], 0, 10), it = strtoull(argv[i+2], 0, 10);
        rocrand_state_philox4x32_10 st;
        rocrand_init(seed, user, 4ULL * it, &st);
        rocrand_state_philox4x32_10 st2 = st;
        unsigned int x = rocrand(&st);
        float u = rocrand_uniform(&st2);
        printf("%u %.9g\n", x, u);
    }
    return 0;
}
"""


def test_stream_equals_rocrand(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc) or not os.path.exists("/opt/rocm/include/rocrand/rocrand_kernel.h"):
        pytest.skip("hipcc / rocRAND headers not available")
    src = tmp_path / "rr.cpp"
    src.write_text(ROCRAND_SRC)
    exe = tmp_path / "rr"
    subprocess.run([hipcc, "-O1", "--offload-arch=gfx950", "-w", "-o", str(exe), str(src)], check=True)
    rng = np.random.RandomState(7)
    triples = [(42, 0, 0), (42, 1, 0), (42, 0, 1), (1, 5, 123456), (2**40 + 17, 2**33 + 5, 2**34 + 1)]
    triples += [(int(rng.randint(0, 2**31)), int(rng.randint(0, 2**31)), int(rng.randint(0, 2**31))) for _ in range(50)]
    args = [str(v) for t in triples for v in t]
    out = subprocess.run([str(exe)] + args, check=True, stdout=subprocess.PIPE, text=True).stdout.split("\n")
    for (seed, user, it), line in zip(triples, out):
        x, u = line.split()
        assert int(x) == orc.lib().orc_draw(seed, user, it)
        assert np.float32(float(u)) == np.float32(orc.lib().orc_uniform(int(x)))
