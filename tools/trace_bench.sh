#!/bin/bash
# Kernel trace of bench.py on the GPU box + the analyses built on it (iteration timeline, schedule interference, kernel stats):
#   gpurun -- bash tools/trace_bench.sh NAME [bench.py arguments]   ->  gpurun_out/trace_NAME/{timeline.txt,interference.txt,kernel_stats.csv,bench.json}
set -uo pipefail
R=$GRAFT_REPO_ROOT
name=$1; shift
O=$R/gpurun_out/trace_$name
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -- python3 $R/bench.py "$@" --no-side-modes --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { echo "trace failed"; tail -5 $O/bench.err; exit 1; }
trace=$(find $O/raw -name '*kernel_trace.csv' | head -1)
stats=$(find $O/raw -name '*kernel_stats.csv' | head -1)
cp $stats $O/kernel_stats.csv
python3 $R/tools/iteration_timeline.py $trace ${CAP:-100} > $O/timeline.txt 2>&1
python3 $R/tools/schedule_interference.py $trace --skip ${SKIP:-50} > $O/interference.txt 2>&1
python3 $R/tools/kernel_overlap.py $trace ${OVERLAP:-sched_sort} > $O/overlap.txt 2>&1
rm -rf $O/raw
cat $O/timeline.txt $O/interference.txt $O/overlap.txt
