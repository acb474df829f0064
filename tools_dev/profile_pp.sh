# PP leg only: three untraced timings, then the steady-state per-kernel table (gpurun_out/pp_steady_state.csv)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for i in 1 2 3; do timeout 200 python3 $R/bench.py --no-second-trunk --no-pgd --no-planes3 --no-fcaf3d --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; print('pp ms/step', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; done
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_pp -- python3 $R/bench.py --no-loader-fed --no-inference --no-second-trunk --no-pgd --no-planes3 --no-fcaf3d --no-cpu-baseline --no-roofline --steps 8 --warmup 4 > /tmp/tr_pp.log 2>&1
python3 $R/tools_dev/trace_summary.py /tmp/tr_pp --steps 3 --top 60 --out $R/gpurun_out/pp_steady_state.csv | head -4
grep -E "headconv|headtail" $R/gpurun_out/pp_steady_state.csv | cut -c1-70
