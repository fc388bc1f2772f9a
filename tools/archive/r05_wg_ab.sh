#!/bin/bash
# round 5: k_node_wgrads with the LayerNorm on the prefetched registers (one barrier less per 32-row step) against the build before ("prewg")
out=gpurun_out/r05_wg; mkdir -p $out
timeout 1200 python -m pytest tests/test_block_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $out/pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee -a $out/pytest.txt
for i in 1 2; do
for v in default prewg; do
  if [ "$v" = default ]; then unset EGT_AMD_LIB; else export EGT_AMD_LIB=$PWD/egt_amd/lib/var/libegt_$v.so; fi
  python bench.py --no-cpu-baseline --no-graph-leg --graph off --steps 50 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); k=d['roofline']['kernels']
        print('$v', round(d['value']), 'graphs/s', round(d['ms_per_step'],4), 'ms median', round(d['median_ms_per_step'],4), {n:round(x['avg_us'],1) for n,x in k.items() if not n.startswith('k_block')})
" | tee -a $out/ab.txt
done; done
