# PMC passes over a few steps of the shipped config (tools_dev/pmc_target_second.py), one counter group per pass, summarised
# per kernel into gpurun_out/r04_second_pmc.json (copy to profiles/). Raw outputs stay on the box (/tmp/pmc).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p /tmp/pmc
i=0; dirs=""
for c in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc/s$i -- python3 $R/tools_dev/pmc_target_second.py 3 > /tmp/pmc/slog$i.txt 2>&1 || echo "pass $i failed: $c"
  dirs="$dirs /tmp/pmc/s$i"
done
python3 $R/tools_dev/pmc_kernels_summary.py $R/gpurun_out/r04_second_pmc.json $dirs | head -12
