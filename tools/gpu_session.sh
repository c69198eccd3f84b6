#!/bin/bash
# One GPU session: per-family parity tests (separate processes so a fault in one family cannot hide the
# others), whole-UNet tests, smoke, bench, rocprofv3 kernel trace. Everything lands in gpurun_out/.
# usage: tools/gpu_session.sh [quick|full|final|...]  (final = all tests, smoke, the bench lines and the steady-state rocprof summary)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
MODE=${1:-full}
export TMPDIR=/tmp
PYT="python -m pytest -q --no-header --tb=short -p no:cacheprovider --timeout=300 --maxfail=40 -m gpu"
run() { # name, timeout, command...
  local name=$1 to=$2; shift 2
  echo "=== $name ===" | tee -a gpurun_out/session.log
  timeout $to "$@" > gpurun_out/$name.log 2>&1
  echo "exit=$? $(tail -n 1 gpurun_out/$name.log | cut -c1-200)" | tee -a gpurun_out/session.log
}
rm -f gpurun_out/parity.jsonl gpurun_out/session.log
rocminfo 2>/dev/null | grep -E "gfx|Compute Unit" | head -4 >> gpurun_out/session.log
nproc >> gpurun_out/session.log
if [ "$MODE" = "vae" ]; then
  run t_vae    900 $PYT tests/test_vae_gpu.py
  run bench_vae 900 python bench.py --config vae --steps 30 --warmup 5
  cut -c1-600 gpurun_out/session.log
  exit 0
fi
if [ "$MODE" = "micro" ]; then
  run t_gemm   900 $PYT tests/test_ops_gpu.py -k "geglu or linear or gemv or dma_pipe"
  run t_conv   900 $PYT tests/test_ops_gpu.py -k "conv"
  run t_refapi 900 $PYT tests/test_reference_api_gpu.py
  run trace 600 python tools/trace_igemm.py
  if [ "$MODE" = "final" ]; then
  export SFAST_TUNE_CACHE=$PWD/gpurun_out/tune_cache.json
  run bench_sdxl 900 python bench.py --config sdxl --steps 10 --warmup 2 --no-cpu-baseline
  run bench_default 900 python bench.py --no-cpu-baseline --no-roofline
  run bench_vae 900 python bench.py --config vae --steps 30 --warmup 5
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof -o bench -- python $OLDPWD/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline > $OLDPWD/gpurun_out/rocprof.log 2>&1 )
  echo "rocprof exit=$?" >> gpurun_out/session.log
  for db in $(find gpurun_out/prof -name "*.db"); do python tools/rocpd_summary.py $db --csv gpurun_out/kernel_stats.csv --top 70 --step-marker cfg_ddim --steps 8 > gpurun_out/kernel_stats.txt; done
  rm -rf gpurun_out/prof
  cut -c1-300 gpurun_out/session.log
  exit 0
fi
run bench2 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-roofline
  cut -c1-300 gpurun_out/session.log
  exit 0
fi
if [ "$MODE" = "testprof" ]; then
  run t_norm   600 $PYT tests/test_ops_gpu.py -k "group_norm or layer_norm"
  run t_gemm   900 $PYT tests/test_ops_gpu.py -k "geglu or linear or gemv or dma_pipe"
  run t_conv   900 $PYT tests/test_ops_gpu.py -k "conv"
  run t_misc   600 $PYT tests/test_ops_gpu.py -k "copy or timestep or cfg or golden"
  run t_refapi 900 $PYT tests/test_reference_api_gpu.py
  run t_unet  1200 $PYT tests/test_unet_gpu.py
  MODE=prof
fi
if [ "$MODE" = "prof" ]; then
  export SFAST_TUNE_CACHE=$PWD/gpurun_out/tune_cache.json
  run bench    900 python bench.py --steps 30 --warmup 5 --dump-kernels gpurun_out/kernels.json
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof -o bench -- python $OLDPWD/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline > $OLDPWD/gpurun_out/rocprof.log 2>&1 )
  echo "rocprof exit=$?" >> gpurun_out/session.log
  for db in $(find gpurun_out/prof -name "*.db"); do python tools/rocpd_summary.py $db --csv gpurun_out/kernel_stats.csv --top 70 --step-marker cfg_ddim --steps 8 > gpurun_out/kernel_stats.txt; done
  rm -rf gpurun_out/prof
  cut -c1-300 gpurun_out/session.log
  exit 0
fi
run t_norm   600 $PYT tests/test_ops_gpu.py -k "group_norm or layer_norm"
run t_gemm   900 $PYT tests/test_ops_gpu.py -k "geglu or linear or gemv or dma_pipe"
run t_conv   900 $PYT tests/test_ops_gpu.py -k "conv"
run t_attn   900 $PYT tests/test_ops_gpu.py -k "attention"
run t_misc   600 $PYT tests/test_ops_gpu.py -k "copy or timestep or cfg or golden"
run t_refapi 900 $PYT tests/test_reference_api_gpu.py
run t_unet  1200 $PYT tests/test_unet_gpu.py
run t_vae    900 $PYT tests/test_vae_gpu.py
run smoke    600 python __graft_entry__.py smoke
run bench    900 python bench.py --steps 30 --warmup 5 --dump-kernels gpurun_out/kernels.json
if [ "$MODE" = "final" ]; then
  export SFAST_TUNE_CACHE=$PWD/gpurun_out/tune_cache.json
  run bench_sdxl 900 python bench.py --config sdxl --steps 10 --warmup 2 --no-cpu-baseline
  run bench_default 900 python bench.py --no-cpu-baseline --no-roofline
  run bench_vae 900 python bench.py --config vae --steps 30 --warmup 5
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof -o bench -- python $OLDPWD/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline > $OLDPWD/gpurun_out/rocprof.log 2>&1 )
  echo "rocprof exit=$?" >> gpurun_out/session.log
  for db in $(find gpurun_out/prof -name "*.db"); do python tools/rocpd_summary.py $db --csv gpurun_out/kernel_stats.csv --top 70 --step-marker cfg_ddim --steps 8 > gpurun_out/kernel_stats.txt; done
  rm -rf gpurun_out/prof
  cut -c1-300 gpurun_out/session.log
  exit 0
fi
run bench2 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-roofline
run bench_sdxl 900 python bench.py --config sdxl --steps 10 --warmup 2 --no-cpu-baseline
run bench_default 900 python bench.py --no-cpu-baseline --no-roofline
run bench_vae 900 python bench.py --config vae --steps 30 --warmup 5
run bench_torchrun 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-roofline
if [ "$MODE" = "full" ] || [ "$MODE" = "tune" ]; then
  run tune 900 bash -c "python tools/tune_igemm.py > gpurun_out/tune.json 2> gpurun_out/tune.txt"
fi
if [ "$MODE" = "full" ]; then
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof -o bench -- python $OLDPWD/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline > $OLDPWD/gpurun_out/rocprof.log 2>&1 )
  echo "rocprof exit=$?" >> gpurun_out/session.log
  for db in $(find gpurun_out/prof -name "*.db"); do python tools/rocpd_summary.py $db --csv gpurun_out/kernel_stats.csv --top 60 > gpurun_out/kernel_stats.txt; done
  rm -rf gpurun_out/prof
  find gpurun_out/prof -name "*stats*" | head >> gpurun_out/session.log
fi
cut -c1-300 gpurun_out/session.log
