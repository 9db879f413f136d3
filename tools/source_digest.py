#!/usr/bin/env python3
"""sha256 over the kernel sources of the library (cu2rec_amd/csrc/*.hip, *.hpp, *.cpp + the C header), in name order: what a committed
rocprofv3 summary must have been taken from for bench.py to put it beside a live timing.
usage: tools/source_digest.py [--write profiles/rNN_profile_meta.json file ...]   (records the digest and the profile files it covers)"""
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_digest():
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "cu2rec_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "cu2rec_amd", "csrc", "*.hpp")) +
                   glob.glob(os.path.join(ROOT, "cu2rec_amd", "csrc", "*.cpp")) + [os.path.join(ROOT, "include", "cu2rec_amd.h")])
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    d = kernel_source_digest()
    if len(sys.argv) > 2 and sys.argv[1] == "--write":
        with open(sys.argv[2], "w") as fh:
            json.dump({"source_digest": d, "covers": [os.path.basename(p) for p in sys.argv[3:]],
                       "what": "sha256[:16] of cu2rec_amd/csrc/*.{hip,hpp,cpp} + include/cu2rec_amd.h at the time these rocprofv3 summaries were taken"}, fh, indent=1)
    print(d)
