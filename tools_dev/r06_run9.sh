#!/bin/bash
cd "$(dirname "$0")/.."
for v in "new A=1" "old GGA_SP_OFFSET_SUMS=0"; do
  set -- $v; tag=$1; shift
  env "$@" python -m pytest tests/test_model_gpu.py -q -s -m gpu -k "second_train_step and (3-2 or 4-2 or 5-2)" > gpurun_out/r06_gradratio_$tag.log 2>&1
  echo "$tag rc $?"; grep -h "GRAD_RATIOS\|LOSSES" gpurun_out/r06_gradratio_$tag.log | cut -c1-400
done
