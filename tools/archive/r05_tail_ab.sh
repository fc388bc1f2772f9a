#!/bin/bash
# round 5: step tail -- k_sum_segments (16 groups x 8 loads per round, one wait per round), k_edge_param_grads (inputs staged in one
# round), k_edge_prep folded into k_node_pre -- against the build before (variant "pretail"), same box
out=gpurun_out/r05_tail; mkdir -p $out
timeout 1200 python -m pytest tests/test_block_gpu.py tests/test_fullsize_gpu.py tests/test_narrow_gpu.py tests/test_graph_gpu.py tests/test_capi_graph_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $out/pytest.txt
for i in 1 2; do
for v in default pretail; do
  if [ "$v" = default ]; then unset EGT_AMD_LIB; else export EGT_AMD_LIB=$PWD/egt_amd/lib/var/libegt_$v.so; fi
  python bench.py --no-cpu-baseline --no-graph-leg --graph off --steps 50 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); k=d['roofline']['kernels']
        print('$v', round(d['value']), 'graphs/s', round(d['ms_per_step'],4), 'ms median', round(d['median_ms_per_step'],4), {n:round(x['avg_us'],1) for n,x in k.items() if not n.startswith('k_block')})
" | tee -a $out/ab.txt
  python bench.py --no-cpu-baseline --no-graph-leg --graph on --steps 50 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); k=d['roofline']['kernels']
        print('$v graph', round(d['value']), 'graphs/s', round(d['ms_per_step'],4), 'ms median', round(d['median_ms_per_step'],4), {n:round(x['avg_us'],1) for n,x in k.items() if not n.startswith('k_block')})
" | tee -a $out/ab.txt
done; done
unset EGT_AMD_LIB
for wl in pattern500k_n120 zinc100k_n37; do
for v in default pretail; do
  if [ "$v" = default ]; then unset EGT_AMD_LIB; else export EGT_AMD_LIB=$PWD/egt_amd/lib/var/libegt_$v.so; fi
  python bench.py --workload $wl --no-cpu-baseline --no-graph-leg --steps 30 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); k=d['roofline']['kernels']
        print('$wl $v', round(d['value']), 'graphs/s', round(d['ms_per_step'],4), 'ms', {n:round(x['avg_us'],1) for n,x in k.items() if not n.startswith('k_block')})
" | tee -a $out/ab.txt
done; done
