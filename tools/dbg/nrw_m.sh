mkdir -p gpurun_out/nrwm; : > gpurun_out/nrwm/log
timeout 900 python -m pytest tests/test_narrow_gpu.py -m gpu -x -q > gpurun_out/nrwm/pytest.log 2>&1; tail -5 gpurun_out/nrwm/pytest.log
for w in ${WLS:-cifar10_n150 cifar10_n150_fp32 pattern500k_n120_b128}; do
for quad in 1 0; do
  EGT_NARROW_BWD=1 EGT_NRW_BWD_QUAD=$quad timeout 300 python bench.py --workload $w --no-cpu-baseline --no-graph-leg --steps 30 --warmup 5 2>>gpurun_out/nrwm/err | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); k = d['roofline']['kernels']
        print('$w quad=$quad', round(d['value']), 'graphs/s', round(d['ms_per_step'], 3), 'ms |', ' '.join(f'{n}={v[\"avg_us\"]:.1f}' for n, v in k.items() if n in ('k_block_bwd','k_block_fwd')))
" >> gpurun_out/nrwm/log
done; done
cat gpurun_out/nrwm/log
