mkdir -p gpurun_out/occ; : > gpurun_out/occ/tlsweep.log
for cfg in "cifar10_n150 128" "pattern500k_n120 128" "pattern500k_n120 64" "pattern500k_n120 32" "pattern500k_n120 16" "cifar10_n150 64"; do set -- $cfg
for tl in ${TLS:-16 15 14 13 12 11 10 9 8 6}; do
  EGT_BWD_TL=$tl EGT_BENCH_B=$2 timeout 300 python bench.py --workload $1 --no-cpu-baseline --no-graph-leg --steps 12 --warmup 3 2>>gpurun_out/occ/err | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); k = d['roofline']['kernels']
        print('$1 B=$2 TL=$tl', round(d['value']), 'graphs/s', round(d['ms_per_step'], 3), 'ms |', ' '.join(f'{n}={v[\"avg_us\"]:.1f}' for n, v in k.items() if n in ('k_block_bwd','k_block_fwd','k_sum_segments','k_node_bwd')))
" >> gpurun_out/occ/tlsweep.log
done; done
cat gpurun_out/occ/tlsweep.log
