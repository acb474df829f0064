# PMC passes over tools_dev/exp_headconv.py (head output conv forward / weight gradient as the step calls them), one counter
# group per pass (known-good groups of pmc_second.sh; every pass under its own timeout), summarised per kernel into
# gpurun_out/r03_headconv_pmc.json (copy to profiles/).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p /tmp/pmc
i=0; dirs=""
for c in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 100 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc/h$i -- python3 $R/tools_dev/exp_headconv.py > /tmp/pmc/hlog$i.txt 2>&1 || echo "pass $i failed: $c"
  dirs="$dirs /tmp/pmc/h$i"
done
python3 $R/tools_dev/pmc_kernels_summary.py $R/gpurun_out/r03_headconv_pmc.json $dirs -- headconv_fwd16 headconv_wgrad16 | head -8
