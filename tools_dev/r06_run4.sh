#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for v in "new A=1" "old GGA_SP_OFFSET_SUMS=0 GGA_SP_HALO=1" "new_halo GGA_SP_HALO=1" "old_nohalo GGA_SP_OFFSET_SUMS=0"; do
  set -- $v; tag=$1; shift
  env "$@" python -m pytest tests/test_sparse_gpu.py -q -s -m gpu -k "full_grid_vs_pair_list" > gpurun_out/r06_fullgrid_$tag.log 2>&1
  echo "$tag rc $?"; grep -h "FULL_GRID_OFFENDERS" gpurun_out/r06_fullgrid_$tag.log | cut -c1-900
done
python -m pytest tests/test_sparse_gpu.py tests/test_trained_regime_gpu.py tests/test_label_gen.py tests/test_pgd_gpu.py tests/test_pipelines.py tests/test_postproc_gpu.py tests/test_precision_gpu.py tests/test_pseudo_labels.py -q -m gpu --durations=12 > gpurun_out/r06_suite_rest.log 2>&1; echo "rest rc $?"; tail -25 gpurun_out/r06_suite_rest.log | cut -c1-200
