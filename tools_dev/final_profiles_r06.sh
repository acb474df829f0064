# Round-6 end-of-round artefacts on the final tree (GPU box, from the repo root: bash tools_dev/final_profiles_r06.sh):
#   r06_bench_default_output.json / _traced_output.json / _kernel_stats.csv   the driver's command, untraced and under rocprofv3 --kernel-trace --stats
#   r06_pp_bs16_channels_last_steady_state.csv, r06_second_bs8_steady_state.csv  per-step kernel tables (last 3 of 8 steps)
#   r06_pp_pmc.json, r06_second_pmc.json, r06_scatter_pmc.json                 PMC passes (separate --pmc runs, kernel-trace only)
#   (r06_sp_halo2.txt was taken before GGA_SP_HALO=2 left the tree)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 900 python3 $R/bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $R/gpurun_out/r06_bench_default_output.json
python3 -c "
import json; d=json.load(open('$R/gpurun_out/r06_bench_default_output.json'))
print('untraced', d['ms_per_step'], d['value'], {k: d[k].get('ms_per_step') for k in ('second_trunk','pgd_trunk','fcaf3d_trunk','planes3')}, d['roofline']['frac'], {k: v['value'] for k, v in d['loader_fed'].items() if k.startswith('workers')})"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-loader-fed --no-inference > $R/gpurun_out/r06_bench_default_traced_output.json 2> /tmp/tr_bench.err
f=$(ls /tmp/tr_bench/*/*_kernel_stats.csv | head -1); cp "$f" $R/gpurun_out/r06_bench_default_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_second -- python3 $R/bench.py --config $R/configs/gga/gga_kitti_config.py --batch 8 --steps 8 --warmup 4 --no-cpu-baseline --no-roofline --no-inference > /tmp/tr_second.log 2>&1
python3 $R/tools_dev/trace_summary.py /tmp/tr_second --steps 3 --top 90 --out $R/gpurun_out/r06_second_bs8_steady_state.csv | head -2
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_pp -- python3 $R/bench.py --no-second-trunk --no-pgd --no-planes3 --no-fcaf3d --no-cpu-baseline --no-roofline --no-loader-fed --no-inference --steps 8 --warmup 4 > /tmp/tr_pp.log 2>&1
python3 $R/tools_dev/trace_summary.py /tmp/tr_pp --steps 3 --top 90 --out $R/gpurun_out/r06_pp_bs16_channels_last_steady_state.csv | head -2
bash $R/tools_dev/pmc_scatter.sh r06 | tail -4
bash $R/tools_dev/pmc_pp.sh r06 | tail -8
sed -e 's/r04_second_pmc.json/r06_second_pmc.json/' $R/tools_dev/pmc_second.sh > /tmp/pmc_second_r06.sh; bash /tmp/pmc_second_r06.sh | tail -10
# the pseudo-label run under the kernel tracer (PointPillars, both samples_per_gpu values; the train legs cut to 3 steps)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_inf -- python3 $R/bench.py --no-second-trunk --no-pgd --no-planes3 --no-fcaf3d --no-cpu-baseline --no-roofline --no-loader-fed --steps 3 --warmup 2 > $R/gpurun_out/r06_inference_traced_output.json 2> /tmp/tr_inf.err
f=$(ls /tmp/tr_inf/*/*_kernel_stats.csv | head -1); cp "$f" $R/gpurun_out/r06_inference_kernel_stats.csv
python3 -c "
import json; d=json.load(open('$R/gpurun_out/r06_bench_default_output.json'))
json.dump(d.get('inference'), open('$R/gpurun_out/r06_inference.json', 'w'), indent=1)
json.dump(d.get('parity_at_bench_size'), open('$R/gpurun_out/r06_parity_at_bench_size.json', 'w'), indent=1)
print('inference', {k: v['value'] for k, v in d['inference'].items() if k.startswith('samples')}, 'second', {k: v['value'] for k, v in d['inference']['second_trunk'].items() if k.startswith('samples')})
print('parity_at_bench_size', d['parity_at_bench_size']['planes2'])"
# the shipped config at the reference's own samples_per_gpu = 32 (configs/gga/gga_kitti_config.py:185)
timeout 600 python3 $R/bench.py --no-pgd --no-fcaf3d --no-loader-fed --no-planes3 --no-cpu-baseline --no-inference --second-batch 32 --steps 10 --warmup 8 2>/dev/null | tail -1 > /tmp/bs32.json
python3 -c "
import json; d=json.load(open('/tmp/bs32.json')); s=d['second_trunk']; s.pop('roofline', None)
json.dump(s, open('$R/gpurun_out/r06_second_bs32.json','w'), indent=1); print('bs32', s['ms_per_step'], s['value'])"
