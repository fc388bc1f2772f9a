mkdir -p gpurun_out/occ; : > gpurun_out/occ/pro.log
for cfg in "pattern500k_n120 16" "pattern500k_n120 128" "cifar10_n150 128"; do set -- $cfg
for np in 0 1; do
  EGT_BWD_TL=16 EGT_NO_BWD_PROLOGUE=$np EGT_BENCH_B=$2 timeout 300 python bench.py --workload $1 --no-cpu-baseline --no-graph-leg --steps 20 --warmup 5 2>>gpurun_out/occ/err | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); k = d['roofline']['kernels']
        print('$1 B=$2 no_pro=$np', round(d['value']), 'graphs/s', round(d['ms_per_step'], 3), 'ms |', ' '.join(f'{n}={v[\"avg_us\"]:.1f}x{v[\"launches\"]}' for n, v in k.items()))
" >> gpurun_out/occ/pro.log
done; done
cat gpurun_out/occ/pro.log
