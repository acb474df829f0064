#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_postproc_gpu.py tests/test_loader.py -q -x -m gpu > gpurun_out/r06_postproc.log 2>&1; echo "postproc rc $?"; tail -3 gpurun_out/r06_postproc.log
python -m pytest tests/test_model_gpu.py -q -s -m gpu -k "10-2" > gpurun_out/r06_seed10.log 2>&1; echo "seed10 rc $?"
python bench.py --no-pgd --no-fcaf3d --no-loader-fed --no-planes3 --no-cpu-baseline --steps 10 --warmup 5 > gpurun_out/bench_r06_b.json 2> gpurun_out/bench_r06_b.err; echo "bench rc $?"
GGA_DETECT_BATCHED=0 python bench.py --no-pgd --no-fcaf3d --no-loader-fed --no-planes3 --no-cpu-baseline --no-second-trunk --steps 10 --warmup 5 > gpurun_out/bench_r06_b_loop.json 2> gpurun_out/bench_r06_b_loop.err; echo "bench rc $?"
grep -h "passed\|failed\|^E  " gpurun_out/r06_postproc.log gpurun_out/r06_seed10.log | cut -c1-300 | tail
tail -c 800 gpurun_out/bench_r06_b.err
python - <<'PY'
import json
for f in ('gpurun_out/bench_r06_b.json','gpurun_out/bench_r06_b_loop.json'):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, 'pp', d['ms_per_step'], 'second', d.get('second_trunk',{}).get('ms_per_step'))
        inf=d['inference']
        for k in ('samples_per_gpu_1','samples_per_gpu_16'):
            print('  pp', k, json.dumps(inf[k]))
        print('  match', inf['match_ms_per_frame'], inf.get('pseudo_labels'), inf.get('vs_train_step_frames_per_s'))
        if 'second_trunk' in inf:
            for k in ('samples_per_gpu_1','samples_per_gpu_16'):
                print('  second', k, json.dumps(inf['second_trunk'][k]))
    except Exception as e: print('bench parse failed', f, e)
PY
