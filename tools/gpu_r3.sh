#!/bin/bash
# Round-3 GPU sessions (one gpurun call each): tools/gpu_r3.sh <stage>
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
STAGE=${1:-s1}
export TMPDIR=/tmp
PYT="python -m pytest -q --no-header --tb=short -p no:cacheprovider --timeout=900 --maxfail=30 -m gpu"
run() { # name, timeout, command...
  local name=$1 to=$2; shift 2
  echo "=== $name ===" | tee -a gpurun_out/session.log
  local t0=$(date +%s)
  timeout $to "$@" > gpurun_out/$name.log 2>&1
  echo "exit=$? $(( $(date +%s) - t0 ))s $(tail -n 1 gpurun_out/$name.log | cut -c1-300)" | tee -a gpurun_out/session.log
}
rm -f gpurun_out/parity.jsonl gpurun_out/session.log
rocminfo 2>/dev/null | grep -E "gfx|Compute Unit" | head -2 >> gpurun_out/session.log
case $STAGE in
s1)  # review items: new parity tests, int8 binding, eager cliffs, smoke, default bench (+ end-to-end leg)
  run t_parity_r3 1500 $PYT tests/test_parity_r3_gpu.py
  run t_qlinear 600 $PYT tests/test_reference_api_gpu.py -k "linear_dynamic or qlinear or auto_graph"
  run t_unet_new 900 $PYT tests/test_unet_gpu.py -k "lcm or compile_drop_in or sd15_unet_parity or scheduler or live_weight"
  run smoke 600 python __graft_entry__.py smoke
  run bench 900 python bench.py --steps 50 --warmup 10 --dump-kernels gpurun_out/kernels.json
  ;;
s2)  # attention_q64: parity + same-process A/B against the 32-row kernel; fixes of session 1
  run t_attn 900 $PYT tests/test_ops_gpu.py -k "attention" --durations=5
  run t_qlinear 600 $PYT tests/test_reference_api_gpu.py -k "linear_dynamic or qlinear"
  run t_fix 600 $PYT tests/test_unet_gpu.py tests/test_parity_r3_gpu.py -k "compile_drop_in or peaked or tiny_trained" --durations=5
  run attn_ab 600 python tools/attn_ab.py --variants 32,62,64 --more-shapes
  ;;
s3)  # where the tile loop of attention_q64 spends its time
  run attn_ablate 600 python tools/attn_ablate.py
  run attn_ab 600 python tools/attn_ab.py --variants 32,64
  run attn_fixed 600 python tools/attn_ab.py --variants 32,64 --fixed-cost
  run t_attn 900 $PYT tests/test_ops_gpu.py tests/test_parity_r3_gpu.py -k "attention or peaked"
  ;;
s4)  # conv_patch.hip: parity + A/B against the implicit-im2col pipes
  run t_patch 900 $PYT tests/test_ops_gpu.py -k "conv_patch" --durations=3
  run conv_ab 900 python tools/conv_ab.py
  ;;
s5)  # conv_patch with separate weight / patch producer waves and the deeper ring; attention PMC pass (32-row kernel vs attention_q64)
  run t_patch 900 $PYT tests/test_ops_gpu.py -k "conv_patch" --durations=3
  run conv_ab 900 python tools/conv_ab.py
  run pmc_attn 700 bash tools/gpu_pmc_attn.sh
  ;;
s6)  # fragment reads pipelined across the tile barrier (igemm_glds_ws + conv_patch): parity, conv A/B, bench with the packaged choices and with a fresh tune
  run t_gemm 900 $PYT tests/test_ops_gpu.py -k "gemm or conv or linear or geglu"
  run conv_ab 900 python tools/conv_ab.py
  run bench_pk 600 python bench.py --steps 50 --warmup 10 --no-end-to-end --no-cpu-baseline
  SFAST_TUNE_PACKAGED=0 SFAST_TUNE_CACHE=$PWD/gpurun_out/tune_cache.json run bench_tuned 900 python bench.py --steps 50 --warmup 10 --no-end-to-end --no-cpu-baseline --dump-kernels gpurun_out/kernels.json
  ;;
s7)  # K-loop experiments of the wave-specialised conv kernel; addmm epilogue parameters
  run ws_probe 600 python tools/ws_loop_probe.py
  run t_addmm 600 $PYT tests/test_reference_api_gpu.py -k "addmm or lowp"
  ;;
s8)  # split-K finished inside the GEMM kernel: parity of every GEMM / conv test, then the bench with the join on and off (own tune caches)
  run t_gemm 1200 $PYT tests/test_ops_gpu.py -k "gemm or conv or linear or geglu or split"
  SFAST_TUNE_PACKAGED=0 SFAST_TUNE_CACHE=$PWD/gpurun_out/tune_cache_join.json run bench_join 900 python bench.py --steps 50 --warmup 10 --no-end-to-end --no-cpu-baseline --dump-kernels gpurun_out/kernels.json
  SFAST_SPLITK_JOIN=0 SFAST_TUNE_PACKAGED=0 SFAST_TUNE_CACHE=$PWD/gpurun_out/tune_cache_nojoin.json run bench_nojoin 900 python bench.py --steps 50 --warmup 10 --no-end-to-end --no-cpu-baseline
  run t_unet 900 $PYT tests/test_unet_gpu.py -k "sd15_unet_parity or tiny" 
  ;;
s9)  # K-loop experiments of the LDS-patch conv kernel
  run patch_probe 600 python tools/ws_loop_probe.py --patch
  ;;
full)  # everything but the full-size SVD-XT parity case (run on its own: `svdxt`), slowest tests listed
  run t_all 1700 $PYT tests --durations=15
  run smoke 600 python __graft_entry__.py smoke
  ;;
final)  # the round's SD1.5 evidence: PMC traffic by symbol, the default bench line (roofline + cpu_baseline + end-to-end), rocprofv3 kernel stats
  run pmc_traffic 900 bash tools/gpu_pmc_bench.sh sd15 6
  cp gpurun_out/pmcb/traffic_by_symbol.json profiles/r03_pmc_traffic_by_symbol.json
  run bench_default 900 python bench.py --dump-kernels gpurun_out/kernels.json
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof -o bench -- python $OLDPWD/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-end-to-end > $OLDPWD/gpurun_out/rocprof.log 2>&1 )
  echo "rocprof exit=$?" >> gpurun_out/session.log
  for db in $(find gpurun_out/prof -name "*.db"); do python tools/rocpd_summary.py $db --csv gpurun_out/kernel_stats.csv --top 60 --step-marker cfg_ddim --steps 12 > gpurun_out/kernel_stats.txt; done
  rm -rf gpurun_out/prof
  run bench_torchrun 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-end-to-end
  ;;
final2)  # SDXL and SVD-XT lines with their own PMC traffic files
  run pmc_sdxl 900 bash tools/gpu_pmc_bench.sh sdxl 4
  cp gpurun_out/pmcb/traffic_by_symbol_sdxl.json profiles/r03_pmc_traffic_by_symbol_sdxl.json; cp gpurun_out/pmcb/traffic_by_symbol_sdxl.json gpurun_out/
  run bench_sdxl 900 python bench.py --config sdxl --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end
  run pmc_svd 1200 bash tools/gpu_pmc_bench.sh svd 3
  cp gpurun_out/pmcb/traffic_by_symbol_svd.json profiles/r03_pmc_traffic_by_symbol_svd.json; cp gpurun_out/pmcb/traffic_by_symbol_svd.json gpurun_out/
  run bench_svd 900 python bench.py --config svd --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end
  ;;
svdxt)
  run t_svdxt 1500 $PYT tests/test_parity_r3_gpu.py -k "svd_xt_full_size" --durations=5
  ;;
esac
cut -c1-400 gpurun_out/session.log
