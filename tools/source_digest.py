#!/usr/bin/env python3
"""sha256 over the kernel sources of the library (cu2rec_amd/csrc/*.hip and the local headers they include, transitively), in name order: what a committed
rocprofv3 summary must have been taken from for bench.py to put it beside a live timing.
usage: tools/source_digest.py [--write profiles/rNN_profile_meta.json file ...]   (records the digest and the profile files it covers)"""
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_files():
    """The .hip translation units of the library and every local header they include, transitively: the code the GPU runs and the
    launch logic beside it.  Host-only sources (CSV reader, sharded driver's cadence, CLIs) are not part of a kernel profile's identity."""
    import re
    csrc = os.path.join(ROOT, "cu2rec_amd", "csrc")
    todo, seen = sorted(glob.glob(os.path.join(csrc, "*.hip"))), []
    while todo:
        f = todo.pop(0)
        if f in seen or not os.path.exists(f):
            continue
        seen.append(f)
        with open(f) as fh:
            for inc in re.findall(r'^\s*#include\s+"([^"]+)"', fh.read(), flags=re.M):
                for base in (csrc, os.path.join(ROOT, "include")):
                    cand = os.path.normpath(os.path.join(base, inc))
                    if os.path.exists(cand) and cand not in seen:
                        todo.append(cand)
    return sorted(seen)


def kernel_source_digest():
    h = hashlib.sha256()
    for f in kernel_source_files():
        h.update(os.path.relpath(f, ROOT).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    d = kernel_source_digest()
    if len(sys.argv) > 2 and sys.argv[1] == "--write":
        with open(sys.argv[2], "w") as fh:
            json.dump({"source_digest": d, "covers": [os.path.basename(p) for p in sys.argv[3:]],
                       "files": [os.path.relpath(f, ROOT) for f in kernel_source_files()],
                       "what": "sha256[:16] of the library's .hip translation units and every local header they include, at the time these rocprofv3 summaries were taken"}, fh, indent=1)
    print(d)
