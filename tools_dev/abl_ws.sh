# Ablation builds of the producer / consumer dense 3x3 kernel (tools_dev/exp_libs/libgga_ws_*.so, dense_conv_ws.hip with -DWS_ABL_*=1)
# next to the shipped library and the lock-step kernel (GGA_DC_WS=0): tools_dev/ws_time.py shapes, us (error column is only
# meaningful for the shipped builds).
for rep in 1 2; do
printf "%-10s" ws; python tools_dev/ws_time.py 2>&1 | tail -1
printf "%-10s" lockstep; GGA_DC_WS=0 python tools_dev/ws_time.py 2>&1 | tail -1
for v in ${VARIANTS:-noprod nostore nohalo now both}; do
printf "%-10s" $v; python tools_dev/run_with_lib.py tools_dev/exp_libs/libgga_ws_$v.so tools_dev/ws_time.py 2>&1 | tail -1
done
done
