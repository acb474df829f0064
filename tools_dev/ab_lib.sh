# A/B of an experiment library against the product library on ONE box: the PointPillars and the sparse-config step, alternating,
# 30 steps after 10 each. Usage (GPU box, repo root): bash tools_dev/ab_lib.sh tools_dev/exp_libs/libgga_<name>.so [reps]
R=$GRAFT_REPO_ROOT
LIB=$1; REPS=${2:-2}
for rep in $(seq $REPS); do
for lib in $R/gga_amd/libgga_hip.so $R/$LIB; do
  pp=$(python3 $R/tools_dev/run_with_lib.py $lib $R/bench.py --steps 30 --warmup 10 --no-second-trunk --no-pgd --no-planes3 --no-fcaf3d --no-cpu-baseline --no-roofline --no-loader-fed 2>/dev/null | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  se=$(python3 $R/tools_dev/run_with_lib.py $lib $R/bench.py --config $R/configs/gga/gga_kitti_config.py --batch 8 --steps 30 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$(basename $lib) rep $rep: pointpillars $pp ms, sparse config $se ms"
done
done
