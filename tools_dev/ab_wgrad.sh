for g in 0 1 0 1; do echo plain_grid=$g; GGA_SP_WGRAD_PLAIN_GRID=$g python tools_dev/exp_sparse_locality.py 2>/dev/null | grep -E "base .*weight gradient"; done
python -m pytest tests/test_sparse_gpu.py -q -m gpu -k "single_conv" 2>&1 | tail -2
for g in 1 0 1 0; do GGA_SP_WGRAD_PLAIN_GRID=$g python bench.py --config configs/gga/gga_kitti_config.py --batch 8 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain_grid=$g', d['ms_per_step'])"; done
