timeout 300 python -m pytest tests/test_graph_gpu.py -x -q 2>&1 | grep -v "amdgpu.ids" | cut -c1-300 | tail -12
for args in "--scope model --workload pattern500k_n120" "--scope model" "--scope model --workload cifar10_n150_fp32" "--scope model --workload pattern500k_n120_b128"; do
 for ov in "" "--overlap-ffn"; do for g in off on; do
  timeout 300 python bench.py $args $ov --graph $g --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$args', '| $ov', '| graph $g', round(d['value'],1), round(d['ms_per_step'],3))
"
 done; done
done
