#!/usr/bin/env python3
"""Build the reference's own CPU twin (mf_sequential.cu) from the sources where they lie.

TEST INFRASTRUCTURE (oracle/).  Runs only where /root/reference exists (the build container);
the GPU box uses the prebuilt binaries that travel under oracle/_ref/ (git-ignored).

Outputs (only these, only under oracle/_ref/):
  mf_cpu         unmodified reference CPU twin: the file list of the reference's own unity
                 build (matrix_factorization/makefile:8), hipify-perl (CUDA runtime names ->
                 HIP runtime names; the program never calls the runtime on this path), hipcc
                 host compile.  `-include random` because nvcc pulled <random> in transitively;
                 -I<rocm>/include/hipblas so the hipified (unused) cuBLAS include resolves.
  mf_cpu_philox  the same translation unit with ONE behavioural change, used to generate
                 bit-exact golden vectors: the four lines mf_sequential.cu:109-112 that draw
                 the rating index from a fresh std::random_device (not reproducible; no seed
                 reaches it) are replaced by a call to the oracle's counter-based sampler
                 orc_sample(seed, user, iteration, low, high), and the final P/Q/bias arrays
                 are dumped to the file named by $ORC_DUMP before they are freed.

No reference source is written into the repo: the concatenated / hipified / patched
translation units live in a temporary directory that is removed afterwards.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("CU2REC_REFERENCE", "/root/reference")
SRC = os.path.join(REF, "matrix_factorization")
OUT = os.path.join(HERE, "_ref")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
UNITY = ["config.h", "matrix.h", "util.h", "config.cu", "matrix.cu", "util.cu", "mf_sequential.cu"]
LOCAL_INCLUDE = re.compile(r'^\s*#include "[a-zA-Z./]*"\s*$')


def unity_source():
    """makefile:8 -- drop local #include "x.h" lines, concatenate in order."""
    lines = []
    for name in UNITY:
        with open(os.path.join(SRC, name)) as fh:
            for line in fh.read().split("\n"):
                if not LOCAL_INCLUDE.match(line):
                    lines.append(line)
    return lines


def patch_sampler(lines):
    """Replace the random_device draw (4 lines) and add the dump hook."""
    start = next(i for i, l in enumerate(lines) if "std::random_device" in l)
    end = next(i for i, l in enumerate(lines) if "distr(eng)" in l)
    assert end == start + 3, "reference sampler block is not the expected 4 lines"
    assert "mt19937" in lines[start + 1] and "uniform_int_distribution" in lines[start + 2]
    lines[start:end + 1] = [
        "                int y_i = orc_sample((unsigned long long)(unsigned)cfg->seed, (unsigned long long)x,",
        "                                     (unsigned long long)cfg->cur_iterations + (unsigned long long)i, low, high);",
    ]
    free_at = next(i for i, l in enumerate(lines) if l.strip() == "delete cfg;")
    dump = [
        '    if(const char *orc_dump_path = getenv("ORC_DUMP")) {',
        '        FILE *orc_fp = fopen(orc_dump_path, "wb");',
        "        int orc_hdr[3] = {rows, cols, cfg->n_factors};",
        "        fwrite(orc_hdr, sizeof(int), 3, orc_fp);",
        "        fwrite(P, sizeof(float), (size_t)rows * cfg->n_factors, orc_fp);",
        "        fwrite(Q, sizeof(float), (size_t)cols * cfg->n_factors, orc_fp);",
        "        fwrite(user_bias, sizeof(float), rows, orc_fp);",
        "        fwrite(item_bias, sizeof(float), cols, orc_fp);",
        "        fclose(orc_fp);",
        "    }",
    ]
    lines[free_at:free_at] = dump
    proto = ['extern "C" int orc_sample(unsigned long long seed, unsigned long long user, '
             'unsigned long long iteration, int low, int high);']
    return proto + lines


def compile_tu(tmp, tag, lines, extra):
    cu = os.path.join(tmp, tag + "_all.cu")
    with open(cu, "w") as fh:
        fh.write("\n".join(lines))
    hip = os.path.join(tmp, tag + ".hip.cpp")
    with open(hip, "w") as fh:
        subprocess.run([os.path.join(ROCM, "bin", "hipify-perl"), cu], check=True, stdout=fh,
                       stderr=subprocess.DEVNULL)
    exe = os.path.join(OUT, tag)
    obj = os.path.join(tmp, tag + ".o")
    hipcc = os.path.join(ROCM, "bin", "hipcc")
    run([hipcc, "-std=c++14", "-O2", "-include", "random", "-I" + os.path.join(ROCM, "include", "hipblas"),
         "--offload-arch=gfx950", "-w", "-c", "-o", obj, hip])
    run([hipcc, "-o", exe, obj] + extra)
    return exe


def run(cmd):
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout[-4000:])
        raise SystemExit("build_ref: command failed: " + " ".join(cmd[:3]) + " ...")


def main():
    if not os.path.isdir(SRC):
        print("build_ref: %s not present; keeping prebuilt oracle/_ref (if any)" % SRC)
        return 0
    os.makedirs(OUT, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="cu2rec_ref_")
    try:
        compile_tu(tmp, "mf_cpu", unity_source(), [])
        obj = os.path.join(tmp, "orc.o")
        run(["gcc", "-std=c99", "-O2", "-ffp-contract=off", "-c", "-o", obj,
             os.path.join(HERE, "cu2rec_oracle.c")])
        compile_tu(tmp, "mf_cpu_philox", patch_sampler(unity_source()), [obj, "-lm"])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print("build_ref: wrote", sorted(os.listdir(OUT)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
