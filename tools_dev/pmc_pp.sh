# PMC passes over a few PointPillars steps (bench.py, main leg only), summarised per kernel into gpurun_out/${TAG}_pp_pmc.json:
# duration, clock, matrix-pipe busy, HBM bytes per launch (FETCH_SIZE doubled as on gfx950), wave-cycle shares.
# Separate --pmc passes with --kernel-trace only, as the MI355X guide prescribes. Usage: bash tools_dev/pmc_pp.sh r04
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r04}
rm -rf /tmp/pmcpp; mkdir -p /tmp/pmcpp
i=0; dirs=""
for c in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcpp/s$i -- python3 $R/bench.py --steps 3 --warmup 2 --no-second-trunk --no-pgd --no-fcaf3d --no-planes3 --no-cpu-baseline --no-roofline --no-loader-fed --no-inference > /tmp/pmcpp/log$i.txt 2>&1 || echo "pass $i failed: $c"
  dirs="$dirs /tmp/pmcpp/s$i"
done
python3 $R/tools_dev/pmc_kernels_summary.py $R/gpurun_out/${TAG}_pp_pmc.json $dirs | head -14
