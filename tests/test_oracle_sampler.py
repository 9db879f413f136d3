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


ROCRAND_DEVICE_SRC = r"""
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_kernel.h>
__global__ void draw(const unsigned long long* in, unsigned int* out_x, float* out_u, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    rocrand_state_philox4x32_10 st;
    rocrand_init(in[3 * i], in[3 * i + 1], 4ULL * in[3 * i + 2], &st);   // device API, on the GPU
    rocrand_state_philox4x32_10 st2 = st;
    out_x[i] = rocrand(&st);
    out_u[i] = rocrand_uniform(&st2);
}
int main(int argc, char** argv) {
    int n = (argc - 1) / 3;
    std::vector<unsigned long long> h(3 * n);
    for (int i = 0; i < 3 * n; ++i) h[i] = strtoull(argv[1 + i], 0, 10);
    unsigned long long* d_in; unsigned int* d_x; float* d_u;
    if (hipMalloc(&d_in, h.size() * 8) != hipSuccess) return 2;
    hipMalloc(&d_x, n * 4); hipMalloc(&d_u, n * 4);
    hipMemcpy(d_in, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    draw<<<(n + 63) / 64, 64>>>(d_in, d_x, d_u, n);
    std::vector<unsigned int> x(n); std::vector<float> u(n);
    if (hipMemcpy(x.data(), d_x, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return 3;
    hipMemcpy(u.data(), d_u, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("%u %.9g\n", x[i], u[i]);
    return 0;
}
"""


@pytest.mark.gpu
def test_stream_equals_rocrand_device_api(tmp_path):
    """The sampler the kernels compute in registers is rocRAND's Philox4x32-10 device stream (north star: cuRAND ->
    rocRAND): a kernel that calls rocrand_init / rocrand / rocrand_uniform on the GPU draws the same words."""
    import cu2rec_amd as cu
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = tmp_path / "rrd.hip"
    src.write_text(ROCRAND_DEVICE_SRC)
    exe = tmp_path / "rrd"
    subprocess.run([hipcc, "-O2", "--offload-arch=gfx950", "-w", "-o", str(exe), str(src)], check=True)
    rng = np.random.RandomState(11)
    triples = [(42, 0, 0), (42, 138492, 9999), (7, 1, 2**33)]
    triples += [(int(rng.randint(0, 2**31)), int(rng.randint(0, 2**31)), int(rng.randint(0, 2**31))) for _ in range(200)]
    out = subprocess.run([str(exe)] + [str(v) for t in triples for v in t], check=True, stdout=subprocess.PIPE,
                         text=True).stdout.split("\n")
    for (seed, user, it), line in zip(triples, out):
        x, u = line.split()
        assert int(x) == cu.lib().cu2rec_sampler_draw(seed, user, it) == orc.lib().orc_draw(seed, user, it)
        assert np.float32(float(u)) == np.float32(orc.lib().orc_uniform(int(x)))


def test_sampler_is_uniform_over_a_users_ratings():
    """sgd.cu:36-37 draws each of a user's n ratings with probability 1/n: chi-square over 200,000 draws of
    (user, iteration) streams for a few n, and no visible correlation between neighbouring users / iterations."""
    import cu2rec_amd as cu
    L = cu.lib()
    for n, user0 in ((7, 0), (20, 1000), (144, 138000)):
        counts = np.zeros(n)
        draws = np.array([L.cu2rec_sampler_index(42, user0 + (k % 500), k // 500, 10, 10 + n) - 10 for k in range(200_000)])
        counts = np.bincount(draws, minlength=n).astype(float)
        expected = len(draws) / n
        chi2 = float(((counts - expected) ** 2 / expected).sum())
        assert chi2 < n + 6 * np.sqrt(2 * n), (n, chi2)  # mean n-1, sd sqrt(2(n-1))
        assert abs(np.corrcoef(draws[:-1], draws[1:])[0, 1]) < 0.01
