#!/bin/bash
cd "$(dirname "$0")/.."
for v in "new A=1" "old GGA_SP_OFFSET_SUMS=0"; do
  set -- $v; tag=$1; shift
  env "$@" python -m pytest tests/test_sparse_gpu.py -q -s -m gpu -k "full_grid_vs_pair_list" > gpurun_out/r06_fullgrid_$tag.log 2>&1
  echo "$tag rc $?"; grep -h "FULL_GRID_GRADS\|FULL_GRID_ROWS" gpurun_out/r06_fullgrid_$tag.log | cut -c1-6000
done
