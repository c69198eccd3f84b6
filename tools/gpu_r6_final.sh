#!/bin/bash
# Round 6, evidence session on the final code and kernel choices: full GPU suite, smoke, the counter passes of all four configurations
# (tools/gpu_r6_pmc.sh -> gpurun_out/r06f; pre-heat off), then trace-only passes of the same commands BEHIND bench.py's 3 s pre-heat (steady
# clocks) whose per-symbol durations tools/merge_trace_avg.py puts into the traffic files. The bench lines are taken in the next session,
# once the traffic files of these kernel choices are in profiles/.
export SFAST_COMMIT=db167da
O=gpurun_out/r06g; mkdir -p $O
timeout -k 10 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu_full.log 2>&1; tail -2 $O/pytest_gpu_full.log
cp gpurun_out/parity.jsonl $O/parity.jsonl 2>/dev/null
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
bash tools/gpu_r6_pmc.sh
export TMPDIR=/tmp; R=$PWD
for cfg in sd15 bs8 sdxl svd; do
  case $cfg in sd15) A="--config sd15"; N=20;; bs8) A="--config sd15 --images 8"; N=10;; sdxl) A="--config sdxl"; N=10;; svd) A="--config svd"; N=3;; esac
  [ $cfg = svd ] && export SFAST_GRAPH_CALIBRATE=0
  rm -rf $O/tr; mkdir -p $O/tr
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace -d $R/$O/tr -o trace -- python $R/bench.py $A --steps $N --warmup 2 --no-cpu-baseline --no-roofline --no-end-to-end --no-variants > $R/$O/trace_$cfg.log 2>&1 )
  for db in $(find $O/tr -name "*.db"); do python tools/rocpd_summary.py $db --csv $O/steady_$cfg.csv --top 200 --step-marker cfg_ddim --steps $(( N - 1 )) > $O/steady_$cfg.txt; rm -f $db; done
  head -1 $O/steady_$cfg.txt
done
