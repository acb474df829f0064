#!/bin/bash
# the driver's command on the final tree + the inference leg under the kernel tracer (the PMC / steady-state tables of final_profiles_r06.sh stay valid: the train kernels did not change)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 1200 python3 $R/bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $R/gpurun_out/r06_bench_default_output.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_inf -- python3 $R/bench.py --no-second-trunk --no-pgd --no-planes3 --no-fcaf3d --no-cpu-baseline --no-roofline --no-loader-fed --steps 3 --warmup 2 > $R/gpurun_out/r06_inference_traced_output.json 2> /tmp/tr_inf.err
f=$(ls /tmp/tr_inf/*/*_kernel_stats.csv | head -1); cp "$f" $R/gpurun_out/r06_inference_kernel_stats.csv
python3 -c "
import json; d=json.load(open('$R/gpurun_out/r06_bench_default_output.json'))
json.dump(d.get('inference'), open('$R/gpurun_out/r06_inference.json', 'w'), indent=1)
json.dump(d.get('parity_at_bench_size'), open('$R/gpurun_out/r06_parity_at_bench_size.json', 'w'), indent=1)
print('pp', d['ms_per_step'], d['value'], 'scatter', d['roofline']['frac'], 'second', d['second_trunk']['ms_per_step'], d['second_trunk']['roofline'].get('traffic'), d['second_trunk']['roofline'].get('hbm_roofline', {}).get('traffic'))
print('inference', {k: v['value'] for k, v in d['inference'].items() if k.startswith('samples')}, 'second', {k: v['value'] for k, v in d['inference']['second_trunk'].items() if k.startswith('samples')}, d['inference']['vs_train_step_frames_per_s'])
print('parity_at_bench_size', d['parity_at_bench_size']['planes2'], d['parity_at_bench_size']['fp32_floor'])"
head -12 $R/gpurun_out/r06_inference_kernel_stats.csv | cut -c1-160
