mkdir -p gpurun_out/occ; : > gpurun_out/occ/log
for pad in ${PADS:-0 45000 75000}; do
  EGT_NRW_LDS_PAD=$pad timeout 300 python bench.py --workload cifar10_n150 --no-cpu-baseline --steps 30 --warmup 5 2>>gpurun_out/occ/err | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); k = d['roofline']['kernels']
        print('pad=$pad', round(d['value']), 'graphs/s', round(d['ms_per_step'], 3), 'ms |', ' '.join(f'{n}={v[\"avg_us\"]:.1f}x{v[\"launches\"]}' for n, v in k.items()))
" >> gpurun_out/occ/log
done
cat gpurun_out/occ/log
