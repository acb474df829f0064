# Round-5 end-of-round artefacts on the final tree (GPU box, from the repo root: bash tools_dev/final_profiles_r05.sh):
#   r05_bench_default_output.json / _traced_output.json / _kernel_stats.csv   the driver's command, untraced and under rocprofv3 --kernel-trace --stats
#   r05_pp_bs16_channels_last_steady_state.csv, r05_second_bs8_steady_state.csv  per-step kernel tables (last 3 of 8 steps)
#   r05_pp_pmc.json, r05_second_pmc.json, r05_scatter_pmc.json                 PMC passes (separate --pmc runs, kernel-trace only)
#   (r05_sp_halo2.txt was taken before GGA_SP_HALO=2 left the tree)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 python3 $R/bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $R/gpurun_out/r05_bench_default_output.json
python3 -c "
import json; d=json.load(open('$R/gpurun_out/r05_bench_default_output.json'))
print('untraced', d['ms_per_step'], d['value'], {k: d[k].get('ms_per_step') for k in ('second_trunk','pgd_trunk','fcaf3d_trunk','planes3')}, d['roofline']['frac'], {k: v['value'] for k, v in d['loader_fed'].items() if k.startswith('workers')})"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-loader-fed > $R/gpurun_out/r05_bench_default_traced_output.json 2> /tmp/tr_bench.err
f=$(ls /tmp/tr_bench/*/*_kernel_stats.csv | head -1); cp "$f" $R/gpurun_out/r05_bench_default_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_second -- python3 $R/bench.py --config $R/configs/gga/gga_kitti_config.py --batch 8 --steps 8 --warmup 4 --no-cpu-baseline --no-roofline > /tmp/tr_second.log 2>&1
python3 $R/tools_dev/trace_summary.py /tmp/tr_second --steps 3 --top 90 --out $R/gpurun_out/r05_second_bs8_steady_state.csv | head -2
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_pp -- python3 $R/bench.py --no-second-trunk --no-pgd --no-planes3 --no-fcaf3d --no-cpu-baseline --no-roofline --no-loader-fed --steps 8 --warmup 4 > /tmp/tr_pp.log 2>&1
python3 $R/tools_dev/trace_summary.py /tmp/tr_pp --steps 3 --top 90 --out $R/gpurun_out/r05_pp_bs16_channels_last_steady_state.csv | head -2
bash $R/tools_dev/pmc_scatter.sh r05 | tail -4
bash $R/tools_dev/pmc_pp.sh r05 | tail -8
sed -e 's/r04_second_pmc.json/r05_second_pmc.json/' $R/tools_dev/pmc_second.sh > /tmp/pmc_second_r05.sh; bash /tmp/pmc_second_r05.sh | tail -10
