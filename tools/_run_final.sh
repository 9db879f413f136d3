export TMPDIR=/tmp
mkdir -p gpurun_out/final2
R=$PWD
timeout -k 10 500 python bench.py > gpurun_out/final2/bench_ml20m.json 2> gpurun_out/final2/bench_ml20m.err || { tail -20 gpurun_out/final2/bench_ml20m.err; exit 1; }
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/final2/bench_ml20m_steps20.json 2> gpurun_out/final2/bench_ml20m_steps20.err
timeout -k 10 500 python bench.py --workload netflix --factors 128 --steps 500 --warmup 128 --no-cpu-baseline > gpurun_out/final2/bench_netflix.json 2> gpurun_out/final2/bench_netflix.err
timeout -k 10 300 python bench.py --workload ml-1m --factors 50 --steps 10000 --warmup 500 --no-cpu-baseline > gpurun_out/final2/bench_ml1m.json 2> gpurun_out/final2/bench_ml1m.err
for f in ml20m ml20m_steps20 netflix ml1m; do python -c "
import json
d=json.loads(open('gpurun_out/final2/bench_$f.json').read().strip().splitlines()[-1])
print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('rmse_gap_vs_sequential',{}).get('gap'), {k:(d[k]['value'], d[k]['ms_per_step']) for k in ('ordered_mode','hogwild_resident_mode','hogwild_streaming_mode') if k in d})"; done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final2/stats -o s -- python3 bench.py --steps 1000 --warmup 200 --no-cpu-baseline --no-side-modes > gpurun_out/final2/bench_prof.json 2> gpurun_out/final2/bench_prof.err
f=$(find $R/gpurun_out/final2/stats -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" gpurun_out/final2/kernel_stats.csv; fi
f=$(find $R/gpurun_out/final2/stats -name "*kernel_trace.csv" | head -1)
if [ -n "$f" ]; then python tools/schedule_interference.py "$f" > gpurun_out/final2/interference.txt; python tools/kernel_timeline.py "$f" --skip 100 --count 400 > gpurun_out/final2/timeline.txt; cat gpurun_out/final2/interference.txt gpurun_out/final2/timeline.txt; fi
rm -rf $R/gpurun_out/final2/stats
exit 0
