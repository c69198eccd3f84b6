#!/bin/bash
# Round-2 GPU sessions. usage: tools/gpu_r2.sh <mode>   (everything lands in gpurun_out/)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
MODE=${1:-a}
export TMPDIR=/tmp
export SFAST_TUNE_CACHE=$PWD/gpurun_out/tune_cache_r2.json
PYT="python -m pytest -q --no-header --tb=short -p no:cacheprovider --timeout=900 --maxfail=30 -m gpu"
run() { # name, timeout, command...
  local name=$1 to=$2; shift 2
  echo "=== $name ===" | tee -a gpurun_out/session.log
  local t0=$(date +%s)
  timeout $to "$@" > gpurun_out/$name.log 2>&1
  echo "exit=$? $(( $(date +%s) - t0 ))s $(tail -n 1 gpurun_out/$name.log | cut -c1-300)" | tee -a gpurun_out/session.log
}
prof() { # name, bench args...
  local name=$1; shift
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof_$name -o bench -- python $OLDPWD/bench.py "$@" --no-cpu-baseline --no-roofline > $OLDPWD/gpurun_out/rocprof_$name.log 2>&1 )
  echo "rocprof $name exit=$?" >> gpurun_out/session.log
  for db in $(find gpurun_out/prof_$name -name "*.db"); do python tools/rocpd_summary.py $db --csv gpurun_out/kernel_stats_$name.csv --top 70 --step-marker cfg_ddim --steps 8 > gpurun_out/kernel_stats_$name.txt; done
  rm -rf gpurun_out/prof_$name
}
rm -f gpurun_out/parity.jsonl gpurun_out/session.log gpurun_out/error_budget.jsonl
rocminfo 2>/dev/null | grep -E "gfx|Compute Unit" | head -2 >> gpurun_out/session.log
case $MODE in
a)  # first session of the round: whole suite, floor probe, bench through both paths, error budget, dispatch profile
  run t_new    900 $PYT tests/test_ops_gpu.py -k "grouped" tests/test_reference_api_gpu.py -k "grouped or convolution_op or bmm"
  run floor    300 python tools/launch_floor.py
  run bench    900 python bench.py --steps 30 --warmup 5 --through-compile --dump-kernels gpurun_out/kernels.json
  run t_all   1800 $PYT tests --deselect tests/test_sdxl_gpu.py
  run budget   600 python tools/error_budget.py --config sd15 --batch 2
  run t_sdxl  1500 $PYT tests/test_sdxl_gpu.py
  prof sd15 --steps 10 --warmup 2
  ;;
b)  # statistics hand-over: new op tests, whole-UNet tests, A/B of the fusion and of staged stores, dispatch profile
  run t_stats  900 $PYT tests/test_ops_gpu.py -k "statistics or out_scale or gemv_grouped or layer_norm or group_norm"
  run t_unet  1500 $PYT tests/test_unet_gpu.py tests/test_vae_gpu.py
  run bench_fused   600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --dump-kernels gpurun_out/kernels.json
  SFAST_GN_FUSE=0 run bench_nofuse 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline
  SFAST_STAGE_OUT=1 run bench_staged 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline
  run bench_fused2  600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline
  prof sd15 --steps 10 --warmup 2
  ;;
c)  # after the merge-prologue / slot-combine fixes: statistics tests, attention bias tests, A/B of the fusion modes
  run t_stats  900 $PYT tests/test_ops_gpu.py -k "statistics or out_scale or attention"
  run t_unet   900 $PYT tests/test_unet_gpu.py -k "sd15 or tiny or compile"
  SFAST_GN_FUSE=0 run bench_nofuse 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline
  run bench_fused   600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --dump-kernels gpurun_out/kernels.json
  SFAST_GN_FUSE=nosplit run bench_nosplit 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline
  SFAST_GN_FUSE=0 run bench_nofuse2 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline
  run bench_fused2  600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline
  prof sd15 --steps 10 --warmup 2
  ;;
d)  # attention bias / masks, SVD, int8 linear, then the whole suite (without the 4-minute SDXL test) and the SVD bench line
  run t_new   1200 $PYT tests/test_ops_gpu.py -k "attention" tests/test_svd_gpu.py tests/test_reference_api_gpu.py -k "attention or svd or mix_rows or qlinear or compile_unet"
  run t_mask   600 $PYT tests/test_unet_gpu.py -k "encoder_attention_mask"
  run t_all   1800 $PYT tests --deselect tests/test_sdxl_gpu.py
  run bench_svd 1200 python bench.py --config svd --steps 3 --warmup 1 --no-cpu-baseline
  run bench    600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --dump-kernels gpurun_out/kernels.json
  ;;
e)  # re-run of what session d failed on + SVD bench + SD1.5 line
  run t_new   1200 $PYT tests/test_ops_gpu.py -k "attention or igemm or variant" tests/test_svd_gpu.py tests/test_reference_api_gpu.py -k "attention or svd or qlinear or compile_unet or auto_graph"
  run t_mask   600 $PYT tests/test_unet_gpu.py -k "encoder_attention_mask"
  run bench_svd 1200 python bench.py --config svd --steps 3 --warmup 1 --no-cpu-baseline
  run bench    600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --dump-kernels gpurun_out/kernels.json
  ;;
f)  # XCD box map (choose_xcd_map): whole suite, then A/B against the legacy order on SD1.5 / SDXL (own tune cache per mode), SVD line
  run t_all   1800 $PYT tests --deselect tests/test_sdxl_gpu.py
  export SFAST_TUNE_CACHE=$PWD/gpurun_out/tune_cache_xmap1.json
  run bench_map    600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --dump-kernels gpurun_out/kernels.json
  SFAST_XCD_MAP=0 SFAST_TUNE_CACHE=$PWD/gpurun_out/tune_cache_xmap0.json run bench_legacy 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline
  run bench_map2   600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline
  SFAST_XCD_MAP=0 SFAST_TUNE_CACHE=$PWD/gpurun_out/tune_cache_xmap0.json run bench_legacy2 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline
  run sdxl_map     900 python bench.py --config sdxl --steps 10 --warmup 2 --no-cpu-baseline --no-roofline
  SFAST_XCD_MAP=0 SFAST_TUNE_CACHE=$PWD/gpurun_out/tune_cache_xmap0.json run sdxl_legacy 900 python bench.py --config sdxl --steps 10 --warmup 2 --no-cpu-baseline --no-roofline
  run bench_svd   1200 python bench.py --config svd --steps 3 --warmup 1 --no-cpu-baseline
  ;;
g)  # GroupNorm apply: workgroup-count sweep (prologue repeats vs streaming width), same tune cache for all
  export SFAST_TUNE_CACHE=$PWD/gpurun_out/tune_cache_xmap1.json
  run bench_w256   600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline
  SFAST_GN_APPLY_WGS=64  run bench_w64  600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline
  SFAST_GN_APPLY_WGS=128 run bench_w128 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline
  SFAST_GN_APPLY_WGS=512 run bench_w512 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline
  SFAST_GN_FUSE=0 run bench_nofuse 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline
  run bench_w256b  600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline
  ;;
final)  # the round's evidence: default bench line (packaged tune cache), through-compile gap, rocprof steady window, PMC traffic,
        # SDXL / VAE / SVD lines, the whole GPU suite as the driver runs it, smoke
  unset SFAST_TUNE_CACHE
  run smoke      600 python __graft_entry__.py smoke
  run bench_default 1200 python bench.py --dump-kernels gpurun_out/kernels.json
  run bench_compile  900 python bench.py --steps 30 --warmup 5 --through-compile --no-cpu-baseline --no-roofline
  run bench_torchrun 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline
  prof sd15 --steps 10 --warmup 2
  export SFAST_TUNE_CACHE=$PWD/gpurun_out/tune_cache.json
  run pmc        1500 bash tools/gpu_pmc_bench.sh
  unset SFAST_TUNE_CACHE
  run bench_sdxl 1200 python bench.py --config sdxl --steps 20 --warmup 3 --no-cpu-baseline
  run bench_vae   900 python bench.py --config vae --steps 20 --warmup 3 --no-cpu-baseline
  run bench_svd  1200 python bench.py --config svd --steps 3 --warmup 1 --no-cpu-baseline
  run t_all      2400 $PYT tests
  ;;
h)  # XCD-aware block order of the flash kernel: tests, A/B (SFAST_XCD_MAP=0 = round-1 order for GEMMs and attention), traffic
  run t_attn  1200 $PYT tests/test_ops_gpu.py -k "attention" tests/test_unet_gpu.py tests/test_svd_gpu.py tests/test_reference_api_gpu.py -k "attention or sd15 or tiny or svd or compile"
  run bench_map    600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --dump-kernels gpurun_out/kernels.json
  SFAST_XCD_MAP=0 run bench_legacy 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline
  run bench_map2   600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline
  SFAST_XCD_MAP=0 run bench_legacy2 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline
  run sdxl_map     900 python bench.py --config sdxl --steps 10 --warmup 2 --no-cpu-baseline --no-roofline
  SFAST_XCD_MAP=0 run sdxl_legacy 900 python bench.py --config sdxl --steps 10 --warmup 2 --no-cpu-baseline --no-roofline
  export SFAST_TUNE_CACHE=$PWD/gpurun_out/tune_cache.json
  run pmc        1500 bash tools/gpu_pmc_bench.sh
  ;;
i)  # attention: consecutive heads per XCD -- tests, bench, traffic
  run t_attn   900 $PYT tests/test_ops_gpu.py -k "attention" tests/test_unet_gpu.py -k "sd15 or tiny"
  run bench_map    600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --dump-kernels gpurun_out/kernels.json
  SFAST_XCD_MAP=0 run bench_legacy 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline
  run bench_map2   600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline
  export SFAST_TUNE_CACHE=$PWD/gpurun_out/tune_cache.json
  run pmc        1500 bash tools/gpu_pmc_bench.sh
  ;;
j)  # new edge-case / grouped-conv tests, and the B = 16 (8 images per GPU, BASELINE configs[3] per-GPU shape) autotune pass
  run t_new   900 $PYT tests/test_ops_gpu.py -k "empty or xcd" tests/test_reference_api_gpu.py -k "grouped or conv_bias"
  SFAST_TUNE_CACHE=$PWD/gpurun_out/tune_b16.json run bench_b16 1200 python bench.py --images 8 --steps 10 --warmup 2 --no-cpu-baseline
  ;;
lite)  # bench line + steady-window profile + new backward test (a second evidence set when the first landed on a slow box)
  unset SFAST_TUNE_CACHE
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | head -6 >> gpurun_out/session.log
  run t_bwd     600 $PYT tests/test_reference_api_gpu.py -k "backward or grouped"
  run bench_default 1200 python bench.py --dump-kernels gpurun_out/kernels.json
  run bench_compile  900 python bench.py --steps 30 --warmup 5 --through-compile --no-cpu-baseline --no-roofline
  prof sd15 --steps 10 --warmup 2
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | head -6 >> gpurun_out/session.log
  ;;
nograph)  # are kernels slower inside the replayed graph than in an eager replay? rocprofv3 trace of both
  unset SFAST_TUNE_CACHE
  prof sd15 --steps 10 --warmup 2
  mv gpurun_out/kernel_stats_sd15.csv gpurun_out/kernel_stats_graph.csv; mv gpurun_out/kernel_stats_sd15.txt gpurun_out/kernel_stats_graph.txt
  prof sd15 --steps 10 --warmup 2 --no-graph
  mv gpurun_out/kernel_stats_sd15.csv gpurun_out/kernel_stats_eager.csv; mv gpurun_out/kernel_stats_sd15.txt gpurun_out/kernel_stats_eager.txt
  ;;
k)  # ws 64x64 with a 3-stage ring (three workgroups per CU): op tests, then SD1.5 / SDXL with a fresh tuning pass against the packaged choices
  run t_v26   900 $PYT tests/test_ops_gpu.py -k "variants or split_k or statistics"
  run bench_pkg    600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline
  SFAST_TUNE_PACKAGED=0 SFAST_TUNE_CACHE=$PWD/gpurun_out/tune_v26.json run bench_tuned 900 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --dump-kernels gpurun_out/kernels.json
  run bench_pkg2   600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline
  SFAST_TUNE_PACKAGED=0 SFAST_TUNE_CACHE=$PWD/gpurun_out/tune_v26.json run bench_tuned2 900 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline
  SFAST_TUNE_PACKAGED=0 SFAST_TUNE_CACHE=$PWD/gpurun_out/tune_v26.json run sdxl_tuned 1200 python bench.py --config sdxl --steps 10 --warmup 2 --no-cpu-baseline --no-roofline
  run sdxl_pkg     900 python bench.py --config sdxl --steps 10 --warmup 2 --no-cpu-baseline --no-roofline
  ;;
l)  # conv_in padded to 8 channels (MFMA) vs the small-channel kernel
  run t_unet  1200 $PYT tests/test_unet_gpu.py tests/test_vae_gpu.py -k "not sdxl_full"
  run bench_pad    600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --dump-kernels gpurun_out/kernels.json
  SFAST_CONV_IN_PAD=0 run bench_nopad 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline
  run bench_pad2   600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline
  SFAST_CONV_IN_PAD=0 run bench_nopad2 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline
  run smoke      600 python __graft_entry__.py smoke
  ;;
final2)  # last evidence pass of the round: smoke, default bench line + through-compile, steady-window profile, whole GPU suite
  unset SFAST_TUNE_CACHE
  run smoke      600 python __graft_entry__.py smoke
  run bench_default 1200 python bench.py --dump-kernels gpurun_out/kernels.json
  run bench_compile  900 python bench.py --steps 30 --warmup 5 --through-compile --no-cpu-baseline --no-roofline
  prof sd15 --steps 10 --warmup 2
  run t_all      2400 $PYT tests
  ;;
quick)
  run t_quick  900 $PYT tests/test_ops_gpu.py tests/test_unet_gpu.py -k "${2:-not zzz}"
  run bench    900 python bench.py --steps 30 --warmup 5 --dump-kernels gpurun_out/kernels.json --no-cpu-baseline
  ;;
bench)
  run bench    900 python bench.py --steps 30 --warmup 5 --through-compile --dump-kernels gpurun_out/kernels.json
  prof sd15 --steps 10 --warmup 2
  ;;
esac
cut -c1-400 gpurun_out/session.log
