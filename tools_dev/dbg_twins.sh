t() { echo "== $1"; env $1 GGA_REPLAY_TWINS=28 python tools_dev/dbg_replay2.py 2>&1 | grep -E "twins differ|differs from" | tail -4; }
t GGA_X=0
t GGA_SP_HALO=0
t GGA_DBG_HALO_SYNC=1
t GGA_DBG_HALO_SYNC=2
t GGA_INPUTS_READY=0
