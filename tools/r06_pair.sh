#!/bin/bash
# round 6: the fused pair operator (config 5): parity on the box, the block-scope line with its kernel table, phase stamps
OUT=gpurun_out/r06_pair; mkdir -p $OUT
timeout 900 python -m pytest tests/test_pair_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
timeout 600 python -m pytest tests/test_fullsize_gpu.py -m gpu -x -q -k "n512_block" >> $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
grep -E "passed|failed|rc=|Error|error" $OUT/pytest.log | tail -8
timeout 300 python bench.py --workload synthetic_n512_block --no-cpu-baseline > $OUT/bench_block.json 2> $OUT/bench_block_err.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_pair/bench_block.json').read().strip().splitlines()[-1])
print(round(d['value']), d['ms_per_step'])
r=d['roofline']
for k,v in r['kernels'].items(): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items()})
PY
if [ -n "$1" ]; then
EGT_ATTN_FLAGS=-DEGT_ATTN_STAMPS python -c "from egt_amd import build as B; B.build()" > $OUT/build.log 2>&1
EGT_ATTN_FLAGS=-DEGT_ATTN_STAMPS python tools/pair_stamps.py > $OUT/stamps.txt 2>&1
tail -50 $OUT/stamps.txt
fi
# the node-side GEMM library: hipBLASLt preferred instead of the default
EGT_BENCH_BLAS=default timeout 300 python bench.py --workload synthetic_n512_block --no-cpu-baseline --no-prof > $OUT/bench_block_lt.json 2>> $OUT/bench_block_err.log
python -c "
import json
d=json.loads(open('gpurun_out/r06_pair/bench_block_lt.json').read().strip().splitlines()[-1]); print('default BLAS:', round(d['value']), d['ms_per_step'])"
