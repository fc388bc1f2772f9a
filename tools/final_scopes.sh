mkdir -p gpurun_out/r03_scopes; : > gpurun_out/r03_scopes/scopes.jsonl
for args in "--scope model --workload cifar10_n150_fp32" "--scope model --workload cifar10_n150" "--scope model --workload pattern500k_n120_b128" "--scope model" "--scope model --ffn-matmul bf16x3" "--workload cifar10_n150" "--workload cifar10_n150_fp32" "--workload pattern500k_n120_b128" "--workload pattern500k_n120" ""; do
  timeout 300 python bench.py $args --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print(json.dumps(dict(args='$args', graphs_per_s=round(d['value'],1), ms_per_step=round(d['ms_per_step'],4), kernels={k:round(v['avg_us'],1) for k,v in list((r.get('kernels') or {}).items())[:6]})))" >> gpurun_out/r03_scopes/scopes.jsonl
done
cat gpurun_out/r03_scopes/scopes.jsonl
