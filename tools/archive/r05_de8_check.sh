#!/bin/bash
# round 5: De = 8 backward without spills + k_sum_segments on 16-byte loads: parity, then the configs that run them
out=gpurun_out/r05_de8; mkdir -p $out
timeout 1500 python -m pytest tests/test_narrow_gpu.py tests/test_block_gpu.py tests/test_fullsize_gpu.py tests/test_model.py -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest.log
timeout 600 python tools/sweep_de8.py 8 > $out/sweep_de8.log 2>&1; echo "sweep rc=$?"; tail -3 $out/sweep_de8.log
for wl in cifar10_n150 cifar10_n150_fp32 pattern500k_n120_b128 pattern500k_n120 zinc500k_n64 zinc100k_n37; do
  timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-graph-leg > $out/bench_$wl.json 2> $out/bench_$wl.err; echo "$wl rc=$?"
done
timeout 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "default rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_de8/bench_*.json')):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,'ERR',e); continue
    r=d.get('roofline') or {}; k=r.get('kernels') or {}
    print(f.split('bench_')[-1][:-5], round(d['value']), 'g/s', round(d['ms_per_step'],3),'ms med',round(d.get('median_ms_per_step') or 0,3), 'mode',(d['config'].get('step_mode') or {}).get('chosen'),
          'dom',round(r.get('avg_launch_us') or 0,1),'us', 'sum',round(r.get('kernels_sum_ms_per_step') or 0,3),'net',r.get('kernels_sum_net_ms_per_step'),'ov',r.get('event_pair_overhead_us'),
          {n:round(x['avg_us'],1) for n,x in list(k.items())[:5]}, (d.get('cpu_baseline') or {}).get('sample','')[-200:])
PY
