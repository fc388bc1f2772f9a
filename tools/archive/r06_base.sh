#!/bin/bash
# round 6 baseline on today's box: the new full-size oracle tests + the deep-stack case, then the config-5 lines before any kernel change
OUT=gpurun_out/r06_base; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_fullsize_gpu.py tests/test_block_gpu.py -m gpu -x -q -k "fullsize or zinc500k or deep_stack or shared_workspace" > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
timeout 600 python -m pytest tests/test_attn_gpu.py -m gpu -x -q -k "mfma" >> $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
for WL in synthetic_n512 synthetic_n512_block; do
  timeout 300 python bench.py --workload $WL --no-cpu-baseline > $OUT/bench_$WL.json 2> $OUT/bench_${WL}_err.log
done
timeout 300 python bench.py --steps 20 --warmup 10 --no-cpu-baseline > $OUT/bench_headline.json 2> $OUT/bench_headline_err.log
tail -5 $OUT/pytest.log
