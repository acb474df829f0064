# A/B of the round-5 "fewer vector instructions beside the matrix instructions" changes (dense + sparse weight gradient,
# default sparse forward kernel) on ONE box: per-step kernel tables of both LiDAR legs with tools_dev/exp_libs/libgga_prediet.so
# (dense_conv.hip / sparse_conv.hip of commit 2e1be0f) and with the product library, alternating.
# GPU box, from the repo root: bash tools_dev/ab_valu_diet.sh > gpurun_out/r05_valu_diet_ab.txt
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for rep in 1; do
for lib in $R/tools_dev/exp_libs/libgga_prediet.so $R/gga_amd/libgga_hip.so; do
  tag=$(basename $lib .so)
  rm -rf /tmp/ab_s /tmp/ab_p
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ab_s -- python3 $R/tools_dev/run_with_lib.py $lib $R/bench.py --config $R/configs/gga/gga_kitti_config.py --batch 8 --steps 8 --warmup 4 --no-cpu-baseline --no-roofline > /tmp/ab_s.log 2>&1
  python3 $R/tools_dev/trace_summary.py /tmp/ab_s --steps 3 --top 90 --out /tmp/ab_s.csv > /dev/null
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ab_p -- python3 $R/tools_dev/run_with_lib.py $lib $R/bench.py --no-second-trunk --no-pgd --no-planes3 --no-fcaf3d --no-cpu-baseline --no-roofline --no-loader-fed --steps 8 --warmup 4 > /tmp/ab_p.log 2>&1
  python3 $R/tools_dev/trace_summary.py /tmp/ab_p --steps 3 --top 90 --out /tmp/ab_p.csv > /dev/null
  for leg in s p; do
    echo "== $tag rep $rep leg $leg: $(head -1 /tmp/ab_$leg.csv)"
    grep -E "dense_wgrad3x3_x9_kernel|sp_conv_wgrad_x9_kernel|sp_conv_x9_kernel|dense_conv3x3_ws_kernel|sp_conv_halo" /tmp/ab_$leg.csv | cut -c1-75
  done
done
done
