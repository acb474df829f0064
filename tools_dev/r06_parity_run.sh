#!/bin/bash
# round 6: precision table, 8-seed sweep of the shipped config, trained-regime parity - logs into gpurun_out/
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_precision_gpu.py -x -q -s -m gpu > gpurun_out/r06_precision.log 2>&1; echo "precision rc $?"
python -m pytest tests/test_model_gpu.py -x -q -s -m gpu -k "second_train_step or seed_sweep or pp_train_step" > gpurun_out/r06_seed_sweep.log 2>&1; echo "sweep rc $?"
python -m pytest tests/test_trained_regime_gpu.py -x -q -s -m gpu > gpurun_out/r06_trained_regime.log 2>&1; echo "trained rc $?"
grep -h "PRECISION {" gpurun_out/r06_precision.log | cut -c1-600
grep -h "SWEEP\|passed\|failed" gpurun_out/r06_seed_sweep.log | tail -14
grep -h "RESYNC\|TRAINED\|GUARD\|passed\|failed\|Error" gpurun_out/r06_trained_regime.log | cut -c1-400 | tail -40
