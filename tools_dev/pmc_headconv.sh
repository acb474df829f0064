cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p /tmp/pmc
i=0
for c in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc/p$i -- python3 $R/tools_dev/exp_headconv.py > /tmp/pmc/log$i.txt 2>&1 || echo "pass $i failed: $c"
done
python3 $R/tools_dev/pmc_kernels_summary.py $R/gpurun_out/headconv_pmc.json /tmp/pmc/p1 /tmp/pmc/p2 /tmp/pmc/p3 /tmp/pmc/p4 -- headconv_fwd16 headconv_wgrad_kernel | head -8
