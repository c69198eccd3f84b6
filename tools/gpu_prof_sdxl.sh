#!/bin/bash
# rocprofv3 kernel trace of the SDXL bench line (steady window), tuned choices taken from the committed cache of the same round.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; rm -rf gpurun_out/prof_sdxl
export TMPDIR=/tmp
cp profiles/r01_tune_cache_run44.json /tmp/tune_cache.json
export SFAST_TUNE_CACHE=/tmp/tune_cache.json
R=$PWD
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_sdxl -o bench -- python $R/bench.py --config sdxl --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/rocprof_sdxl.log 2>&1 )
echo "rocprof exit=$? $(grep '^{' gpurun_out/rocprof_sdxl.log | cut -c1-160)"
for db in $(find gpurun_out/prof_sdxl -name "*.db"); do python tools/rocpd_summary.py $db --csv gpurun_out/kernel_stats_sdxl.csv --top 40 --step-marker cfg_ddim --steps 4 > gpurun_out/kernel_stats_sdxl.txt; done
rm -rf gpurun_out/prof_sdxl
head -30 gpurun_out/kernel_stats_sdxl.txt | cut -c1-170
