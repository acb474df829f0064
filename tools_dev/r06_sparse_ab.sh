#!/bin/bash
# precision + time of the sparse products, and the sparse config's step, for the accumulator-chain variants
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python tools_dev/sparse_precision.py new > gpurun_out/sp_prec_new.log 2>&1
GGA_SP_HALO=0 python tools_dev/sparse_precision.py new_nohalo > gpurun_out/sp_prec_new_nohalo.log 2>&1
GGA_SP_OFFSET_SUMS=0 python tools_dev/sparse_precision.py old > gpurun_out/sp_prec_old.log 2>&1
B="python bench.py --no-pgd --no-fcaf3d --no-loader-fed --no-planes3 --no-cpu-baseline --steps 20 --warmup 8"
$B > gpurun_out/bench_sp_new.json 2> gpurun_out/bench_sp_new.err
GGA_SP_HALO=0 $B > gpurun_out/bench_sp_new_nohalo.json 2> gpurun_out/bench_sp_new_nohalo.err
GGA_SP_OFFSET_SUMS=0 $B > gpurun_out/bench_sp_old.json 2> gpurun_out/bench_sp_old.err
grep -h ratio gpurun_out/sp_prec_*.log | grep -v weight_grad
python - <<'PY'
import json
for t in ('new','new_nohalo','old'):
    try:
        d=json.loads(open(f'gpurun_out/bench_sp_{t}.json').read().strip().splitlines()[-1])
        print(t,'pp',d['ms_per_step'],'second',d['second_trunk']['ms_per_step'], d['second_trunk']['dominant_kernels_ms_per_step'])
    except Exception as e: print(t,'failed',e)
PY
