#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/w1; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/w1 -- python3 $R/bench.py "$@" --no-side-modes --no-cpu-baseline > /dev/null 2>&1
python3 $R/tools/kernel_window.py $(find /tmp/w1 -name '*kernel_trace.csv' | head -1) --after-gram 300 --us ${WIN_US:-170}
