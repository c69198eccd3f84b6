#!/bin/bash
# Round 6, last session: full GPU suite, smoke, the default bench line, the standalone SDXL / 8-image / SVD-XT lines and the SVD-XT counter
# passes (on a tune cache populated by the bench run right before them: without it the passes time SVD's own autotuning under counters).
export SFAST_COMMIT=50ebe39
O=gpurun_out/r06g; mkdir -p $O
timeout -k 10 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu_full.log 2>&1; tail -2 $O/pytest_gpu_full.log
cp gpurun_out/parity.jsonl $O/parity.jsonl 2>/dev/null
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout -k 10 600 python bench.py > $O/bench_sd15_default.json.log 2> $O/bench_sd15_default.err
timeout -k 10 400 python bench.py --config sdxl > $O/bench_sdxl.json.log 2>/dev/null
timeout -k 10 400 python bench.py --config sd15 --images 8 > $O/bench_bs8.json.log 2>/dev/null
export SFAST_TUNE_CACHE=/tmp/tune_svd.json
timeout -k 10 900 python bench.py --config svd > $O/bench_svd.json.log 2>/dev/null
cut -c1-200 $O/bench_svd.json.log
timeout -k 10 2400 bash tools/gpu_pmc_bench.sh svd 2 > $O/pmc_svd.log 2>&1
cp gpurun_out/pmcb/traffic_by_symbol_svd.json $O/ 2>/dev/null
cp gpurun_out/pmcb/trace.txt $O/kernel_stats_svd.txt 2>/dev/null
cp gpurun_out/pmcb/trace.csv $O/kernel_stats_svd.csv 2>/dev/null
tail -4 $O/pmc_svd.log | cut -c1-160
