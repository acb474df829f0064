#!/bin/bash
# Start-up cost of MIOpen's first-call solver search for the bench step under different MIOpen settings.
cd $GRAFT_REPO_ROOT
run() { local t0=$(date +%s%N); "$@" 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; local t1=$(date +%s%N); echo "  wall $(( (t1 - t0) / 1000000 )) ms"; }
echo "== default (fresh user db)"; run timeout 600 python3 bench.py --no-cpu-baseline
echo "== default again (user db warm)"; run timeout 600 python3 bench.py --no-cpu-baseline
rm -rf ~/.config/miopen ~/.cache/miopen
echo "== naive off (db cleared)"; MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW=0 run timeout 600 python3 bench.py --no-cpu-baseline
rm -rf ~/.config/miopen ~/.cache/miopen
echo "== find mode 2 (db cleared)"; MIOPEN_FIND_MODE=2 run timeout 600 python3 bench.py --no-cpu-baseline
ls -la ~/.config/miopen ~/.cache/miopen 2>&1 | head
