mkdir -p gpurun_out/nrwm; : > gpurun_out/nrwm/flog
timeout 900 python -m pytest tests/test_narrow_gpu.py -m gpu -x -q > gpurun_out/nrwm/pytest_f.log 2>&1; tail -3 gpurun_out/nrwm/pytest_f.log
for w in cifar10_n150 pattern500k_n120_b128 pattern500k_n120; do
for v in quad base fo4; do
  lib=egt_amd/lib/libegt_amd.so; qd=0; [ $v = quad ] && qd=1; [ $v = fo4 ] && lib=egt_amd/lib/var/libegt_fo4.so
  EGT_AMD_LIB=$lib EGT_NRW_FWD_QUAD=$qd timeout 300 python bench.py --workload $w --no-cpu-baseline --no-graph-leg --steps 30 --warmup 5 2>>gpurun_out/nrwm/err | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); k = d['roofline']['kernels']
        print('$w $v', round(d['value']), 'graphs/s', round(d['ms_per_step'], 3), 'ms |', ' '.join(f'{n}={v[\"avg_us\"]:.1f}' for n, v in k.items() if n in ('k_block_bwd','k_block_fwd')))
" >> gpurun_out/nrwm/flog
done; done
cat gpurun_out/nrwm/flog
