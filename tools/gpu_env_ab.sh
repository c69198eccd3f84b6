#!/bin/bash
# A/B of HIP runtime knobs on the graph-replay step time (same box, tuned choices from the committed cache).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
cp profiles/r01_tune_cache_run44.json /tmp/tune_cache.json
export SFAST_TUNE_CACHE=/tmp/tune_cache.json
B="python bench.py --steps 60 --warmup 10 --no-roofline --no-cpu-baseline"
run() { echo "== $1"; shift; env "$@" $B 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2),'it/s', round(d['ms_per_step'],3),'ms')"; }
{
run baseline A=1
run HIP_FORCE_DEV_KERNARG=1 HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0 HIP_FORCE_DEV_KERNARG=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run baseline2 A=1
} > gpurun_out/env_ab.log 2>&1
cat gpurun_out/env_ab.log
