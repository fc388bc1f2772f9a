#!/bin/bash
OUT=gpurun_out/r06_stamps; mkdir -p $OUT
EGT_ATTN_FLAGS=-DEGT_ATTN_STAMPS python -c "from egt_amd import build as B; B.build()" > $OUT/build.log 2>&1
EGT_ATTN_FLAGS=-DEGT_ATTN_STAMPS python tools/pair_stamps.py > $OUT/stamps.txt 2>&1
cat $OUT/stamps.txt | tail -60
