#!/bin/bash
# round 6: (1) parity of the pair kernels with the drain-free in-wave LDS hand-offs + four-piece DMA blocks, A/B against the drains;
#          (2) config 4 as specified (B = 16): rows per backward workgroup / forward waves (zero-code switches)
OUT=gpurun_out/r06_ab2; mkdir -p $OUT
python -c "from egt_amd import build as B; B.build()" > $OUT/build.log 2>&1
timeout 900 python -m pytest tests/test_pair_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q -k "pair or n512_block" > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
line() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=(d.get('roofline') or {}).get('kernels') or {}
print(sys.argv[2], round(d['value']), round(d['ms_per_step'],4), {n:round(v['avg_us'],1) for n,v in list(k.items())[:3]})
PY
}
for rep in 1 2; do
  timeout 300 python bench.py --workload synthetic_n512_block --no-cpu-baseline > $OUT/b_nowait_$rep.json 2>> $OUT/err.log; line $OUT/b_nowait_$rep.json nowait+dma4
done
EGT_ATTN_FLAGS="-DPAIR_LDS_NOWAIT=0" python -c "from egt_amd import build as B; B.build()" >> $OUT/build.log 2>&1
for rep in 1 2; do
  EGT_ATTN_FLAGS="-DPAIR_LDS_NOWAIT=0" timeout 300 python bench.py --workload synthetic_n512_block --no-cpu-baseline > $OUT/b_drain_$rep.json 2>> $OUT/err.log; line $OUT/b_drain_$rep.json drains
done
python -c "from egt_amd import build as B; B.build()" >> $OUT/build.log 2>&1
for v in default 4 6; do
  if [ $v = default ]; then unset EGT_BWD_TL; else export EGT_BWD_TL=$v; fi
  timeout 300 python bench.py --workload pattern500k_n120 --no-cpu-baseline > $OUT/p_tl_$v.json 2>> $OUT/err.log; line $OUT/p_tl_$v.json "pattern500k_n120 EGT_BWD_TL=$v"
done
unset EGT_BWD_TL
EGT_NRW_FWD_WAVES=8 timeout 300 python bench.py --workload pattern500k_n120 --no-cpu-baseline > $OUT/p_fw8.json 2>> $OUT/err.log; line $OUT/p_fw8.json "pattern500k_n120 EGT_NRW_FWD_WAVES=8"
EGT_NRW_FWD_WAVES=4 timeout 300 python bench.py --workload pattern500k_n120 --no-cpu-baseline > $OUT/p_fw4.json 2>> $OUT/err.log; line $OUT/p_fw4.json "pattern500k_n120 EGT_NRW_FWD_WAVES=4"
