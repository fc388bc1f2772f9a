# batch sweep of a De = 8 workload: per-launch kernel times against B (tail / quantisation of the workgroup rounds)
mkdir -p gpurun_out/occ; : > gpurun_out/occ/bsweep.log
for B in ${BS:-64 96 102 104 112 128 154 160 205}; do
  EGT_BENCH_B=$B timeout 300 python bench.py --workload ${WL:-cifar10_n150} --no-cpu-baseline --no-graph-leg --steps 20 --warmup 5 2>>gpurun_out/occ/err | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); k = d['roofline']['kernels']
        print('B=$B', round(d['value']), 'graphs/s', round(d['ms_per_step'], 3), 'ms |', ' '.join(f'{n}={v[\"avg_us\"]:.1f}' for n, v in k.items() if n in ('k_block_bwd','k_block_fwd')), '| us/graph bwd', round(k['k_block_bwd']['avg_us']/$B, 3), 'fwd', round(k['k_block_fwd']['avg_us']/$B, 3))
" >> gpurun_out/occ/bsweep.log
done
cat gpurun_out/occ/bsweep.log
