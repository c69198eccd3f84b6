cd $GRAFT_REPO_ROOT
python bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/bench_long.log 2>&1 &
BP=$!
sleep 45
for i in 1 2 3 4 5 6 7 8; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | head -6; echo ---; sleep 0.4; done > gpurun_out/clocks.log 2>&1
wait $BP
tail -1 gpurun_out/bench_long.log | cut -c1-200
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | head -3
