# bench lines of the De = 8 workloads (eager + hipGraph replay figure)
mkdir -p gpurun_out/nrwm; : > gpurun_out/nrwm/lines.log
for w in ${WLS:-cifar10_n150 cifar10_n150_fp32 pattern500k_n120_b128 pattern500k_n120}; do
  python bench.py --workload $w --no-cpu-baseline --steps 30 --warmup 5 2>>gpurun_out/nrwm/err | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); k = d['roofline']['kernels']
        print('$w', round(d['value']), 'graphs/s', round(d['ms_per_step'], 3), 'ms | graph', d.get('hipgraph_replay'), '|', ' '.join(f'{n}={v[\"avg_us\"]:.1f}' for n, v in k.items()))
" >> gpurun_out/nrwm/lines.log
done
cat gpurun_out/nrwm/lines.log
