#!/bin/bash
# A/B of --graph on / off on one box (hipGraph replay of forward + backward vs eager launches).
# usage (GPU box): bash tools/graph_ab.sh > gpurun_out/graph_ab.jsonl
for args in "" "--workload pattern500k_n120" "--workload pattern500k_n120_b128" "--workload cifar10_n150" \
            "--scope model" "--scope model --workload pattern500k_n120" "--scope model --workload pattern500k_n120_b128" \
            "--scope model --workload cifar10_n150_fp32"; do
  for g in off on; do
    timeout 300 python bench.py $args --graph $g --steps 30 --warmup 5 --no-cpu-baseline 2> gpurun_out/graph_ab.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print(json.dumps({'args': '$args', 'graph': '$g', 'graphs_per_s': round(d['value'], 1), 'ms_per_step': round(d['ms_per_step'], 4),
                          'hipgraph': d['config'].get('hipgraph')}))
" || echo "{\"args\": \"$args\", \"graph\": \"$g\", \"error\": \"$(tail -1 gpurun_out/graph_ab.err | tr '"' "'")\"}"
  done
done
