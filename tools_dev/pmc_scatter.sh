# HBM traffic of the pillar-scatter kernels from the PMC counters (separate --pmc passes, as the MI355X guide prescribes),
# on the GPU box from the repo root: bash tools_dev/pmc_scatter.sh r04  ->  gpurun_out/r04_scatter_pmc.json (+ the two csv)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r04}
cd /tmp
rm -rf /tmp/pmc_f /tmp/pmc_w
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -- python3 $R/tools_dev/pmc_target_scatter.py > /tmp/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -- python3 $R/tools_dev/pmc_target_scatter.py > /tmp/pmc_w.log 2>&1
python3 $R/tools_dev/pmc_summary.py /tmp/pmc_f /tmp/pmc_w $R/gpurun_out/${TAG}_scatter_pmc.json scatter | tail -30
cp $(ls /tmp/pmc_f/*/*counter_collection.csv | head -1) $R/gpurun_out/${TAG}_scatter_pmc_FETCH_SIZE.csv
cp $(ls /tmp/pmc_w/*/*counter_collection.csv | head -1) $R/gpurun_out/${TAG}_scatter_pmc_WRITE_SIZE.csv
