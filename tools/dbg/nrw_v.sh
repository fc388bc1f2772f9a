# usage: nrw_v.sh <lib variants...>: bench cifar10_n150 (bf16) with each variant library
mkdir -p gpurun_out/nrwm; : > gpurun_out/nrwm/vlog
for v in "$@"; do
for w in ${WLS:-cifar10_n150}; do
  lib=egt_amd/lib/var/libegt_$v.so; [ "$v" = base ] && lib=egt_amd/lib/libegt_amd.so
  EGT_AMD_LIB=$lib EGT_NARROW_BWD=1 timeout 300 python bench.py --workload $w --no-cpu-baseline --no-graph-leg --steps 30 --warmup 5 2>>gpurun_out/nrwm/err | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); k = d['roofline']['kernels']
        print('$w $v', round(d['value']), 'graphs/s', round(d['ms_per_step'], 3), 'ms |', ' '.join(f'{n}={v[\"avg_us\"]:.1f}' for n, v in k.items() if n in ('k_block_bwd','k_block_fwd')))
" >> gpurun_out/nrwm/vlog
done; done
cat gpurun_out/nrwm/vlog
