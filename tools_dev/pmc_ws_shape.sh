# One PMC pass (clock + matrix-pipe busy) over three PointPillars steps for each consumer MFMA shape of dense_conv_ws.hip.
# Usage (on the GPU box): bash tools_dev/pmc_ws_shape.sh -> gpurun_out/r05_ws_shape_pmc_{32,16}.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in 32 16; do
  rm -rf /tmp/pmcws$m
  export GGA_DC_WS_MFMA=$m
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmcws$m -- python3 $R/bench.py --steps 3 --warmup 2 --no-second-trunk --no-pgd --no-fcaf3d --no-planes3 --no-cpu-baseline --no-roofline --no-loader-fed > /tmp/pmcws$m.log 2>&1 || echo "pass failed"
  python3 $R/tools_dev/mfma_busy.py /tmp/pmcws$m $R/gpurun_out/r05_ws_shape_pmc_$m.json dense_conv3x3_ws dense_wgrad3x3 | head -6
done
