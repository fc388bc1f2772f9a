#!/bin/bash
OUT=gpurun_out/r06_fulltest; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
