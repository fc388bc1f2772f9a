#!/bin/bash
# round 5: the reworked bench line (events inside the graph, median, CPU legs) + the SURVEY 8(d) workload variants
out=gpurun_out/r05_bench; mkdir -p $out
timeout 1200 python -m pytest tests/test_bench_gpu.py tests/test_graph_gpu.py -x -q -m gpu > $out/pytest_bench.log 2>&1; echo "pytest bench rc=$?"; tail -4 $out/pytest_bench.log
timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "default rc=$?"
timeout 300 python bench.py --graph on --no-cpu-baseline --no-graph-leg > $out/bench_graph_on.json 2> $out/bench_graph_on.err
timeout 300 python bench.py --graph off --no-cpu-baseline --no-graph-leg > $out/bench_graph_off.json 2> $out/bench_graph_off.err
timeout 300 python bench.py --steps 20 --warmup 10 --no-cpu-baseline > $out/bench_driver_style.json 2> $out/bench_driver_style.err
for wl in zinc500k_n64_full pattern500k_bmax pattern500k_bmax_b128 pattern500k_n188 pattern500k_n188_b128 pattern500k_n120 pattern500k_n120_b128; do
  timeout 300 python bench.py --workload $wl --no-cpu-baseline > $out/bench_$wl.json 2> $out/bench_$wl.err; echo "$wl rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_bench/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f,'ERR',e); continue
    r=d.get('roofline') or {}
    print(f.split('bench_')[-1][:-5], round(d['value']), 'g/s', round(d['ms_per_step'],3),'ms median',d.get('median_ms_per_step'), 'mode',(d['config'].get('step_mode') or {}).get('chosen'),
          'dom',r.get('kernel'),round(r.get('avg_launch_us') or 0,1),'us n=',r.get('launches'),'in_region',r.get('timed_in_region'),'frac',round(r.get('frac') or 0,3),
          'sum_kernels',round(r.get('kernels_sum_ms_per_step') or 0,3), 'N',d['config'].get('N'), 'cpu',(d.get('cpu_baseline') or {}).get('sample','')[-160:])
PY
