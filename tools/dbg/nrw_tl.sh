mkdir -p gpurun_out/nrwm gpurun_out/occ
[ -n "$SKIP_TESTS" ] || { timeout 1800 python -m pytest tests/test_narrow_gpu.py tests/test_block_gpu.py -m gpu -x -q > gpurun_out/nrwm/pytest3.log 2>&1; tail -3 gpurun_out/nrwm/pytest3.log; }
: > gpurun_out/occ/tl.log
for B in 16 32 64; do for tl in 16 8 4 0; do
  EGT_BWD_TL=$tl EGT_BENCH_B=$B timeout 300 python bench.py --workload pattern500k_n120 --no-cpu-baseline --no-graph-leg --steps 20 --warmup 5 2>>gpurun_out/occ/err | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); k = d['roofline']['kernels']
        print('B=$B TL=$tl', round(d['value']), 'graphs/s', round(d['ms_per_step'], 3), 'ms |', ' '.join(f'{n}={v[\"avg_us\"]:.1f}' for n, v in k.items()))
" >> gpurun_out/occ/tl.log
done; done
cat gpurun_out/occ/tl.log
