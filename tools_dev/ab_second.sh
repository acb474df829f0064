run() { python bench.py --config configs/gga/gga_kitti_config.py --batch 8 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
(cd _r02; run r02)
run HEAD_prefetch_first
GGA_PREFETCH_FIRST=0 run HEAD_prefetch_after
(cd _r02; run r02)
run HEAD_prefetch_first
GGA_PREFETCH_FIRST=0 run HEAD_prefetch_after
GGA_SP_HALO=1 run HEAD_halo
