#!/bin/bash
# round 6: the fused pair operator (config 5): parity on the box, then the block-scope line
OUT=gpurun_out/r06_pair; mkdir -p $OUT
timeout 900 python -m pytest tests/test_pair_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -30 $OUT/pytest.log
timeout 600 python -m pytest tests/test_fullsize_gpu.py -m gpu -x -q -k "n512_block" >> $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
timeout 300 python bench.py --workload synthetic_n512_block --no-cpu-baseline > $OUT/bench_block.json 2> $OUT/bench_block_err.log
tail -12 $OUT/pytest.log; tail -c 1500 $OUT/bench_block.json
