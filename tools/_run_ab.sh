mkdir -p gpurun_out/ab2
for rep in 1 2; do
for v in off any anyaff; do
  unset CU2REC_BLOCKSOLVE_AFFINE CU2REC_BLOCKSOLVE_AFFINE_HEAD CU2REC_BS_ANYORDER
  if [ $v = any ]; then export CU2REC_BS_ANYORDER=1; fi
  if [ $v = anyaff ]; then export CU2REC_BS_ANYORDER=1 CU2REC_BLOCKSOLVE_AFFINE=24 CU2REC_BLOCKSOLVE_AFFINE_HEAD=8; fi
  timeout -k 10 200 python bench.py --steps 1000 --warmup 200 --no-cpu-baseline --no-side-modes > gpurun_out/ab2/bench_${v}_$rep.json 2> gpurun_out/ab2/bench_${v}_$rep.err && python -c "
import json,sys
d=json.loads(open('gpurun_out/ab2/bench_${v}_$rep.json').read().strip().splitlines()[-1])
print('$v',$rep,d['value'],d['ms_per_step'],d['rmse_gap_vs_sequential']['gap'])"
done
done
exit 0
