#!/bin/bash
# rocprofv3 --pmc passes of bench.py on the GPU box (FETCH_SIZE, WRITE_SIZE, TCC_HIT_sum + TCC_MISS_sum: one pass each) + tools/pmc_summary.py:
#   gpurun -- bash tools/pmc_passes.sh NAME [bench.py arguments]      ->  gpurun_out/pmc/pmc_NAME.json
#   PMC_PROGRAM=tools/loss_pmc_probe.py gpurun -- bash tools/pmc_passes.sh NAME [its arguments]   (another program of the repo, run as it is)
# A counter pass SERIALISES kernels across streams, so the default topology's device-side join (phase 3's waiting workgroup, the side
# stream's gate) would wait for kernels the profiler has not let run yet: CU2REC_BS_GATE=0 selects the event fork / join (same
# kernels, same bytes) for these passes.
set -uo pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export CU2REC_BS_GATE=0  # (the library also takes this by itself when it sees ROCPROF_COUNTER_COLLECTION, which rocprofv3 --pmc exports)
name=$1; shift
for ctr in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $ctr | tr ' ' '_')
  rm -rf $O/pmc_${name}_$tag
  if [ -n "${PMC_PROGRAM:-}" ]; then prog=("$R/$PMC_PROGRAM" "$@"); else prog=("$R/bench.py" "$@" --no-side-modes --no-cpu-baseline); fi
  timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_${name}_$tag -- python3 "${prog[@]}" > $O/pmc_${name}_$tag.log 2>&1 || echo "pmc $name $ctr failed"
  echo "$name $ctr done: $(grep -c . $(find $O/pmc_${name}_$tag -name '*counter_collection.csv' | head -1)) rows"
done
python3 $R/tools/pmc_summary.py $O/pmc_$name.json $O/pmc_${name}_FETCH_SIZE $O/pmc_${name}_WRITE_SIZE $O/pmc_${name}_TCC_HIT_sum_TCC_MISS_sum > $O/pmc_$name.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
tail -3 $O/pmc_${name}_FETCH_SIZE.log | cut -c1-200
