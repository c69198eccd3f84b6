export SFAST_COMMIT=db167da
mkdir -p gpurun_out/r06f
for cfg in sd15:6 bs8:4 sdxl:4 svd:2; do
  c=${cfg%%:*}; n=${cfg##*:}
  timeout -k 10 1500 bash tools/gpu_pmc_bench.sh $c $n > gpurun_out/r06f/pmc_$c.log 2>&1
  suf=""; [ "$c" != "sd15" ] && suf="_$c"
  cp gpurun_out/pmcb/traffic_by_symbol$suf.json gpurun_out/r06f/ 2>/dev/null
  cp gpurun_out/pmcb/trace.txt gpurun_out/r06f/kernel_stats_$c.txt 2>/dev/null
  cp gpurun_out/pmcb/trace.csv gpurun_out/r06f/kernel_stats_$c.csv 2>/dev/null
  tail -3 gpurun_out/r06f/pmc_$c.log
done
