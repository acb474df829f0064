for i in 1 2; do
echo depth2; GGA_EXP_FEW=1 python tools_dev/exp_sparse_locality.py 2>/dev/null | grep -E "base .*weight gradient"
echo nosplit; GGA_EXP_FEW=1 python tools_dev/run_with_lib.py tools_dev/exp_libs/libgga_wgrad_nosplit.so tools_dev/exp_sparse_locality.py 2>/dev/null | grep -E "base .*weight gradient"
done
