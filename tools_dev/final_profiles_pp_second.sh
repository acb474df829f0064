# Round-end artefacts after a change that only touches the KITTI legs: the default bench under the tracer (kernel stats of the
# SAME command the driver runs) and the steady-state per-step kernel tables of the PointPillars and SECOND legs.
# Outputs: gpurun_out/r04_* (copy to profiles/). The PGD / FCAF3D tables come from tools_dev/final_profiles.sh.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_bench -- python3 $R/bench.py --no-loader-fed --no-inference > $R/gpurun_out/r04_bench_default_traced_output.json 2> /tmp/tr_bench.err
f=$(ls /tmp/tr_bench/*/*_kernel_stats.csv | head -1); cp "$f" $R/gpurun_out/r04_bench_default_kernel_stats.csv
tail -c 300 $R/gpurun_out/r04_bench_default_traced_output.json; echo
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_second -- python3 $R/bench.py --no-loader-fed --no-inference --config $R/configs/gga/gga_kitti_config.py --batch 8 --steps 8 --warmup 4 --no-cpu-baseline --no-roofline > /tmp/tr_second.log 2>&1
python3 $R/tools_dev/trace_summary.py /tmp/tr_second --steps 3 --top 90 --out $R/gpurun_out/r04_second_bs8_steady_state.csv | head -2
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_pp -- python3 $R/bench.py --no-loader-fed --no-inference --no-second-trunk --no-pgd --no-planes3 --no-fcaf3d --no-cpu-baseline --no-roofline --steps 8 --warmup 4 > /tmp/tr_pp.log 2>&1
python3 $R/tools_dev/trace_summary.py /tmp/tr_pp --steps 3 --top 90 --out $R/gpurun_out/r04_pp_bs16_channels_last_steady_state.csv | head -2
timeout 400 python3 $R/bench.py 2>/dev/null | tail -1 > $R/gpurun_out/r04_bench_default_output.json
python3 -c "
import json; d=json.load(open('$R/gpurun_out/r04_bench_default_output.json'))
print('untraced', d['ms_per_step'], d['value'], {k: d[k].get('ms_per_step') for k in ('second_trunk','pgd_trunk','fcaf3d_trunk','planes3')}, d['roofline'])"
