#!/bin/bash
# GPU box: the two BASELINE config files that need positional encodings, run unchanged (keys as shipped by the reference) on
# synthetic graphs: 2 epochs of training each, then the evaluation reports.  (tests/configs/ holds the key/value DATA of
# configs/main/zinc/100k/egt_epe.json and configs/main/cifar10/100k/egt_spe.json with three keys changed for the box: num_epochs 2,
# distributed false (one process), save_path under gpurun_out/; tests/test_training.py loads the shipped keys verbatim.)
set -e
for cfg in tests/configs/zinc_100k_egt_epe.json tests/configs/cifar10_100k_egt_spe.json; do
  echo "== $cfg"
  python -m egt_amd.training $cfg --synthetic 256 2>&1 | grep -E "CHECKPOINT|DONE|Error|error" | tail -4
  python -m egt_amd.training $cfg --synthetic 256 --evaluate 2>&1 | grep -E "MAE|accuracy|crossentropy|LOADED" | tail -8
done
