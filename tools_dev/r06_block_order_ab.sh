#!/bin/bash
# sparse config: mask order within spatial blocks (GGA_SP_BLOCK_ROWS) x XCD-major tile walk (GGA_SP_TILE_ORDER=1)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
B="python bench.py --no-pgd --no-fcaf3d --no-loader-fed --no-planes3 --no-cpu-baseline --no-inference --steps 10 --warmup 8"
run() { tag=$1; shift; env "$@" $B > gpurun_out/bench_blk_$tag.json 2> gpurun_out/bench_blk_$tag.err; python - "$tag" <<'PY'
import json, sys
t=sys.argv[1]
try:
    d=json.loads(open(f'gpurun_out/bench_blk_{t}.json').read().strip().splitlines()[-1])
    print(t, 'second', d['second_trunk']['ms_per_step'], d['second_trunk']['dominant_kernels_ms_per_step'])
except Exception as e: print(t, 'failed', e)
PY
}
run base A=1
run tile1 GGA_SP_TILE_ORDER=1
run b4096 GGA_SP_BLOCK_ROWS=4096
run b4096_tile1 GGA_SP_BLOCK_ROWS=4096 GGA_SP_TILE_ORDER=1
run b16384_tile1 GGA_SP_BLOCK_ROWS=16384 GGA_SP_TILE_ORDER=1
run b1024_tile1 GGA_SP_BLOCK_ROWS=1024 GGA_SP_TILE_ORDER=1
run spatial GGA_SP_SPATIAL_ROWS=1
run spatial_tile1 GGA_SP_SPATIAL_ROWS=1 GGA_SP_TILE_ORDER=1
run spatial_b4096_tile1 GGA_SP_SPATIAL_ROWS=1 GGA_SP_BLOCK_ROWS=4096 GGA_SP_TILE_ORDER=1
