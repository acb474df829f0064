run() { python bench.py --config configs/gga/gga_kitti_config.py --batch 8 --steps $2 --warmup 0 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 steps $2', d['config']['final_loss'])"; }
for i in 1 2 3 4 5 6 7 8; do run default 2; done
for i in 1 2 3 4 5 6; do run default 4; done
