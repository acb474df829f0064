# Round-end artefacts (run on the GPU box from the repo root): the default bench under the tracer (kernel stats of the SAME
# command the driver runs) and steady-state per-step kernel tables of the legs. Outputs: gpurun_out/r04_* (copy to profiles/).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_bench -- python3 $R/bench.py --no-loader-fed --no-inference > $R/gpurun_out/r04_bench_default_traced_output.json 2> /tmp/tr_bench.err
f=$(ls /tmp/tr_bench/*/*_kernel_stats.csv | head -1); cp "$f" $R/gpurun_out/r04_bench_default_kernel_stats.csv
tail -c 400 $R/gpurun_out/r04_bench_default_traced_output.json; echo
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_second -- python3 $R/bench.py --no-loader-fed --no-inference --config $R/configs/gga/gga_kitti_config.py --batch 8 --steps 8 --warmup 4 --no-cpu-baseline --no-roofline > /tmp/tr_second.log 2>&1
python3 $R/tools_dev/trace_summary.py /tmp/tr_second --steps 3 --top 60 --out $R/gpurun_out/r04_second_bs8_steady_state.csv | head -4
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_pp -- python3 $R/bench.py --no-loader-fed --no-inference --no-second-trunk --no-pgd --no-planes3 --no-fcaf3d --no-cpu-baseline --no-roofline --steps 8 --warmup 4 > /tmp/tr_pp.log 2>&1
python3 $R/tools_dev/trace_summary.py /tmp/tr_pp --steps 3 --top 60 --out $R/gpurun_out/r04_pp_bs16_channels_last_steady_state.csv | head -4
