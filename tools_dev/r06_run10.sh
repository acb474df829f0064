#!/bin/bash
cd "$(dirname "$0")/.."
python tools_dev/dbg_fullgrid_fwd.py 2 2>&1 | tail -21 | cut -c1-300
python -m pytest tests/test_sparse_gpu.py -q -s -m gpu > gpurun_out/r06_sparse_tests.log 2>&1; echo "sparse rc $?"; grep -h "FULL_GRID_INSITU\|FULL_GRID_GRADS\|passed\|failed" gpurun_out/r06_sparse_tests.log | cut -c1-300
python -m pytest tests/test_model_gpu.py tests/test_trained_regime_gpu.py -q -s -m gpu --durations=8 -k "second_train_step or sweep or trained_regime" > gpurun_out/r06_sweep2.log 2>&1; echo "sweep rc $?"; grep -h "SWEEP second,\|passed\|failed\|^[0-9.]*s call" gpurun_out/r06_sweep2.log | cut -c1-400
