#!/bin/bash
# Round 6, last session: full GPU suite, smoke, the default bench line, the standalone SDXL / 8-image / SVD-XT lines, then the counter
# passes of all four configurations (tools/gpu_r6_pmc.sh).
export SFAST_COMMIT=a10eeba
O=gpurun_out/r06g; mkdir -p $O
timeout -k 10 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu_full.log 2>&1; tail -2 $O/pytest_gpu_full.log
cp gpurun_out/parity.jsonl $O/parity.jsonl 2>/dev/null
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout -k 10 600 python bench.py > $O/bench_sd15_default.json.log 2> $O/bench_sd15_default.err
timeout -k 10 400 python bench.py --config sdxl > $O/bench_sdxl.json.log 2>/dev/null
timeout -k 10 400 python bench.py --config sd15 --images 8 > $O/bench_bs8.json.log 2>/dev/null
timeout -k 10 900 python bench.py --config svd > $O/bench_svd.json.log 2>/dev/null
cut -c1-200 $O/bench_svd.json.log
# counter + trace-only passes of all four configurations (gpurun_out/r06f)
bash tools/gpu_r6_pmc.sh
