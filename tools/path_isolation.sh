#!/bin/bash
# What each path of a block-solve iteration costs on an otherwise idle chip (TIMING ONLY: the test build of the library skips launches,
# the results are wrong): the whole iteration, the side kernel alone (CU2REC_BS_DBG=64), the three phases alone (128); event topology.
#   gpurun -- bash tools/path_isolation.sh NAME [bench.py arguments]   -> gpurun_out/iso/NAME_{all,side,main}_kernel_stats.csv
set -uo pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/iso
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export CU2REC_BS_GATE=0 CU2REC_AMD_LIB=$R/build/test/libcu2rec_amd_hooks.so
name=$1; shift
for case in all:0 side:64 main:128; do
  tag=${case%%:*}; export CU2REC_BS_DBG=${case##*:}
  rm -rf $O/${name}_$tag
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${name}_$tag -- python3 $R/bench.py "$@" --no-side-modes --no-cpu-baseline > $O/${name}_$tag.log 2>&1 || echo "$tag failed"
  f=$(find $O/${name}_$tag -name '*kernel_stats.csv' | head -1)
  cp "$f" $O/${name}_${tag}_kernel_stats.csv
  echo "== $tag"; python3 - "$O/${name}_${tag}_kernel_stats.csv" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:7]:
    print("%-46s calls %6s avg %9.1f us" % (r["Name"].replace("void ", "").replace("cu2rec::(anonymous namespace)::", "")[:46], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  grep -o '"ms_per_step": [0-9.]*' $O/${name}_$tag.log | head -1
  find $O/${name}_$tag -name "*kernel_trace.csv" -delete
done
