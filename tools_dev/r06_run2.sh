#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_trained_regime_gpu.py -q -s -m gpu > gpurun_out/r06_trained_regime.log 2>&1; echo "trained rc $?"
python -m pytest tests/test_model_gpu.py -q -s -m gpu -k "bench_size or 9-2 or 10-2" > gpurun_out/r06_bench_size.log 2>&1; echo "bench-size rc $?"
python bench.py --no-pgd --no-fcaf3d --no-loader-fed --no-planes3 --steps 10 --warmup 5 > gpurun_out/bench_r06_a.json 2> gpurun_out/bench_r06_a.err; echo "bench rc $?"
grep -h "RESYNC\|TRAINED\|passed\|failed\|^E  " gpurun_out/r06_trained_regime.log | cut -c1-330 | tail -40
grep -h "BENCH_SIZE\|LOSSES\|passed\|failed\|^E  " gpurun_out/r06_bench_size.log | cut -c1-330 | tail
tail -c 1500 gpurun_out/bench_r06_a.err
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/bench_r06_a.json').read().strip().splitlines()[-1])
    print('pp', d['ms_per_step'], 'second', d['second_trunk']['ms_per_step'])
    print(json.dumps(d.get('parity_at_bench_size'))[:1500])
    print(json.dumps(d.get('inference'))[:3000])
    print(json.dumps({k:v for k,v in d['cpu_baseline'].items() if k!='pieces_ms_per_16_frames'})[:1200])
except Exception as e: print('bench parse failed', e)
PY
