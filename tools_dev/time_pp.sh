# PP leg only, three untraced timings
R=$GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 200 python3 $R/bench.py --no-second-trunk --no-pgd --no-planes3 --no-fcaf3d --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; print('pp ms/step', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; done
